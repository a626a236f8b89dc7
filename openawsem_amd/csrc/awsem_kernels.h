// awsem_kernels.h -- CDNA4 (gfx950) kernels of the replica-resident AWSEM path.
//
// Execution model: ONE WORKGROUP OWNS ONE REPLICA.  The replica's coordinates, velocities and fixed-point force
// accumulators live in LDS for the whole launch (a 1r69 replica is ~36 KB, so 4 replicas are resident per CU); index
// arrays and parameter tables are shared by every replica and stay L2-resident.  A launch evaluates energies/forces
// once (awsem_eval_kernel) or advances `nsteps` Langevin steps without touching HBM in between (awsem_step_kernel).
// Nothing here is a dense contraction, so no MFMA: the work is distance screens, transcendental switching functions
// and LDS atomics.
//
// Every term body is written against a *view* `V` of one replica (positions in, forces out):
//   Lds<float>   production: float4 positions (one ds_read_b128 per gathered atom) and per-atom 16-byte force
//                records {xy: two Q19.12 components packed in one 64-bit word, z: Q19.12}, so a force lands with one
//                ds_add_u64 + one ds_add_u32 per atom (half the LDS bytes of three 64-bit accumulators)
//   Lds<double>  check mode: SoA doubles and three Q31.32 accumulators per atom
//   CellView     awsem_cell.h: the same API over global memory for systems that do not fit LDS
//
// Determinism: every force, energy and density contribution is converted to fixed point *before* it is added and
// accumulated with integer atomics (integer adds commute), so results are bit-identical from run to run regardless
// of wave order; a pair term adds +F to one atom and -F to the other bit for bit (Newton's third law is exact).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "awsem_device.h"

typedef unsigned long long u64;

// Phase timing of the step kernel (experiment builds only: -DAWSEM_PROF): thread 0 of every workgroup adds the
// cycles between consecutive stamps to DevSys::prof[phase]; dumped by the host when AWSEM_PROF_DUMP is set.
#ifdef AWSEM_PROF
// AWSEM_ARRIVE(k), placed in front of the barrier that ends phase k, books the time up to it as the wave's own work; the stamp behind the
// barrier then books the rest as waiting.  Every wave keeps its own sums (the table behind the per-replica stamps, [wave][work, wait][10]).
struct AwsemProf { long long last, acc[10], wait[10]; };
#define AWSEM_ARRIVE(k) do { long long _t = __builtin_readcyclecounter(); awsem_prof.acc[k] += _t - awsem_prof.last; awsem_prof.last = _t; } while (0)
#define AWSEM_STAMP(S, k) do { long long _t = __builtin_readcyclecounter(); awsem_prof.wait[k] += _t - awsem_prof.last; awsem_prof.last = _t; } while (0)
#define AWSEM_STAMP_INIT AwsemProf awsem_prof; awsem_prof.last = __builtin_readcyclecounter(); for (int _k = 0; _k < 10; _k++) awsem_prof.acc[_k] = awsem_prof.wait[_k] = 0
#define AWSEM_STAMP_FLUSH(S) do { if (threadIdx.x == 0) for (int _k = 0; _k < 10; _k++) atomicAdd((unsigned long long*)&(S).prof[_k], (unsigned long long)(awsem_prof.acc[_k] + awsem_prof.wait[_k])); \
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 1024) for (int _k = 0; _k < 10; _k++) { \
        unsigned long long* _w = (unsigned long long*)(S).prof + 16 + 3 * (size_t)gridDim.x + (threadIdx.x >> 6) * 20; \
        atomicAdd(_w + _k, (unsigned long long)awsem_prof.acc[_k]); atomicAdd(_w + 10 + _k, (unsigned long long)awsem_prof.wait[_k]); } } while (0)
#define AWSEM_STAMP_ARG , AwsemProf& awsem_prof
#define AWSEM_STAMP_PASS , awsem_prof
#else
#define AWSEM_ARRIVE(k) do { } while (0)
#define AWSEM_STAMP(S, k) do { } while (0)
#define AWSEM_STAMP_INIT do { } while (0)
#define AWSEM_STAMP_FLUSH(S) do { } while (0)
#define AWSEM_STAMP_ARG
#define AWSEM_STAMP_PASS
#endif

// -DAWSEM_WALL (implied by -DAWSEM_PROF): wall-clock start and end of every workgroup of the step kernel, and where it ran
#if defined(AWSEM_PROF) || defined(AWSEM_WALL)
#define AWSEM_STAMP_WALL(S, k) do { if (threadIdx.x == 0) { (S).prof[16 + 3 * blockIdx.x + (k)] = (long long)wall_clock64(); \
    if ((k) == 0) (S).prof[16 + 3 * blockIdx.x + 2] = ((long long)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4); } } while (0)
#else
#define AWSEM_STAMP_WALL(S, k) do { } while (0)
#endif

// ------------------------------------------------------------------ scalar helpers
template <class real> struct M;
template <> struct M<float> {
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
    // raw v_rsq_f32 / v_sqrt_f32 (1 ulp): the arguments are squared distances of distinct atoms, never denormal
    static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float floor(float x) { return floorf(x); }
    static __device__ __forceinline__ float log(float x) { return __logf(x); }
    // v_sin_f32 / v_cos_f32 take their argument in revolutions: sin(2 pi u) for u in [0,1) in one instruction each
    static __device__ __forceinline__ void sincos2pi(float u, float* s, float* c) { *s = __builtin_amdgcn_sinf(u); *c = __builtin_amdgcn_cosf(u); }
};
template <> struct M<double> {
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / ::sqrt(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double floor(double x) { return ::floor(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ void sincos2pi(double u, double* s, double* c) { *s = sinpi(2.0 * u); *c = cospi(2.0 * u); }
};

// value * 2^32 as a two's complement 64-bit integer (energies, densities, fp64 forces).  fp32: adding the magic
// constant 1.5 * 2^52 in f64 leaves the rounded (to nearest even) integer in the low mantissa bits; the constant's low
// word is zero, so only the high word needs correcting (cvt, add, sub after the scaling multiply).  |v| < 2^19.
__device__ __forceinline__ u64 to_fixed_scaled(float v32) {
    const u64 b = (u64)__double_as_longlong((double)v32 + 6755399441055744.0);
    return ((u64)((unsigned)(b >> 32) - 0x43380000u) << 32) | (u64)(unsigned)b;
}
__device__ __forceinline__ u64 to_fixed_scaled(double v32) { return (u64)(long long)v32; }
__device__ __forceinline__ u64 to_fixed(float v) { return to_fixed_scaled(v * 4294967296.0f); }
__device__ __forceinline__ u64 to_fixed(double v) { return (u64)(long long)(v * AWSEM_FIXED_SCALE); }
__device__ __forceinline__ double fixed_to_double(u64 u) { return (double)(long long)u * (1.0 / AWSEM_FIXED_SCALE); }
// 2^-32 fixed point -> float without the f64 detour: the integer part and the fraction convert separately
__device__ __forceinline__ float fixed_to_real(u64 u, float) { return (float)(int)(u >> 32) + (float)(unsigned)u * 2.3283064365386963e-10f; }
__device__ __forceinline__ double fixed_to_real(u64 u, double) { return fixed_to_double(u); }

// ds_add / ds_sub (LDS views) or global_atomic_add / _sub (cell view), result unused
__device__ __forceinline__ void fixed_add(u64* p, u64 v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fixed_sub(u64* p, u64 v) { (void)__hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fixed_add(int* p, int v) { (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fixed_sub(int* p, int v) { (void)__hip_atomic_fetch_sub(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <class real> struct R3 { real x, y, z; };
template <class real> __device__ __forceinline__ R3<real> operator-(const R3<real>& a, const R3<real>& b) { return R3<real>{a.x - b.x, a.y - b.y, a.z - b.z}; }
// fixed evaluation order, fma in the handle's own precision (v_fma_f32 / v_fma_f64): the pair set is bit-exact
__device__ __forceinline__ float dist2(float dx, float dy, float dz) { return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)); }
__device__ __forceinline__ double dist2(double dx, double dy, double dz) { return __builtin_fma(dz, dz, __builtin_fma(dy, dy, dx * dx)); }
template <class real> __device__ __forceinline__ real dist2(const R3<real>& d) { return dist2(d.x, d.y, d.z); }

// ------------------------------------------------------------------ views of one replica
// Members every view carries besides coordinates and forces (the contact passes' scratch, energies, pair lists).
template <class real> struct ViewCommon {
    static constexpr int cflag = 0;       // (lean views: the contact term's flags, known at compile time)
    static constexpr bool lean = false;   // LdsLean: the view of the step kernel's lean instantiation (contact variants fold away: cflags)
    int nthr;                // threads of the workgroup.  blockDim.x is a 16-bit field of the implicit kernel arguments: every place the compiler
                             // rematerialises it costs a VECTOR load and a wait on it (there is no scalar 16-bit load); the LDS views read it once and
                             // pin it in a scalar register
    u64 *rho, *drho;         // contact density and dE/drho accumulators (2^-32 fixed point)
    real* rhov;              // per-site water switch s_i = sig(-2 eta_s (rho_i - rho0))
    real *alpha, *dalpha;    // z-dependent membrane switch
    real* gamma;             // gamma tables of the current contact term (env-major)
    real* ramap;             // LDS views: parameter sets of the merged Ramachandran list (RamaMergedD), [set][well][8]
    u64* eg;                 // per-group energies (fixed point)
    u64* cv;                 // sums of the collective-variable terms (fixed point), DevSys::n_cv of them
    u64* cm;                 // [2][3] total momentum of the massive particles (fixed point, CmFix<real>::scale), double buffered by step parity;
                             // behind them [2][4] reals: the mean velocity each sum stands for, converted once per step
    int* satom;              // contact site -> atom, [contact term][max_contact_sites]
    unsigned* lcount;        // pair-list entry counts, one per list
    unsigned* flag;          // [2]: largest squared displacement since the last list build (float bits), double buffered
    unsigned* lists;         // GLOBAL: this replica's pair-list entries
};

template <class real> struct Lds;

// fp32 force accumulator: Q19.12 fixed point, x and y packed in one 64-bit word (x * 2^32 + sign-extended y; the
// packed sums stay exact because integer addition commutes: y = low word, x = (sum - y) >> 32).  The xy words and the z
// words live in two planes: an atomic of 64 lanes on consecutive atoms then touches consecutive banks (an array of
// 16-byte {xy, z, pad} records put lanes l and l+8 of the 64-bit atomic, and l, l+8, l+16, l+24 of the 32-bit one, on
// the same banks).
#define AWSEM_F32_FORCE_SCALE 4096.0f
__device__ __forceinline__ long long pack_xy(int X, int Y) { return ((long long)X << 32) + (long long)Y; }

template <> struct Lds<float> : ViewCommon<float> {
    typedef float real_t;
    float4* p;       // positions (w unused)
    long long* fxy;  // force accumulators: packed x, y  [natoms]
    int* fz;         //                     z           [natoms] (directly behind fxy; 16 bytes per atom are reserved in all)
    float4* v;       // velocities (step kernel only)
    float4* ref;     // positions at the last pair-list build (step kernel only)
    __device__ __forceinline__ R3<float> pos(int a) const { const float4 q = p[a]; return R3<float>{q.x, q.y, q.z}; }
    __device__ __forceinline__ void set_pos(int a, const R3<float>& r) const { p[a] = make_float4(r.x, r.y, r.z, 0.0f); }
    __device__ __forceinline__ R3<float> vel(int a) const { const float4 q = v[a]; return R3<float>{q.x, q.y, q.z}; }
    __device__ __forceinline__ void set_vel(int a, const R3<float>& r) const { v[a] = make_float4(r.x, r.y, r.z, 0.0f); }
    __device__ __forceinline__ R3<float> refpos(int a) const { const float4 q = ref[a]; return R3<float>{q.x, q.y, q.z}; }
    __device__ __forceinline__ void set_ref(int a, const R3<float>& r) const { ref[a] = make_float4(r.x, r.y, r.z, 0.0f); }
    __device__ __forceinline__ void zero_force(int a) const { fxy[a] = 0; fz[a] = 0; }
    // the accumulators are all zero while the pair lists are (re)built: 16 bytes per atom of scratch for the builders
    __device__ __forceinline__ unsigned* scratch() const { return (unsigned*)fxy; }
    __device__ __forceinline__ R3<float> force(int a) const {
        const int2 q = *(const int2*)&fxy[a];   // {xy.lo, xy.hi}
        const int y = q.x, x = q.y + (int)((unsigned)y >> 31);   // undo the borrow of a negative y
        const float s = 1.0f / AWSEM_F32_FORCE_SCALE;
        return R3<float>{(float)x * s, (float)y * s, (float)fz[a] * s};
    }
    __device__ __forceinline__ void add_force(int a, float x, float y, float z) const {
        const float s = AWSEM_F32_FORCE_SCALE;
        fixed_add((u64*)&fxy[a], (u64)pack_xy(__float2int_rn(x * s), __float2int_rn(y * s)));
        fixed_add(&fz[a], __float2int_rn(z * s));
    }
    __device__ __forceinline__ void add_fz(int a, float z) const { fixed_add(&fz[a], __float2int_rn(z * AWSEM_F32_FORCE_SCALE)); }
    // force of a pair potential: d = x_i - x_j, s = -dE/dr / r; +F on i, the bit-identical -F on j
    __device__ __forceinline__ void pair_force(int i, int j, float dx, float dy, float dz, float s) const {
        const float t = s * AWSEM_F32_FORCE_SCALE;
        const int Z = __float2int_rn(t * dz);
        const u64 XY = (u64)pack_xy(__float2int_rn(t * dx), __float2int_rn(t * dy));
        fixed_add((u64*)&fxy[i], XY); fixed_add(&fz[i], Z);
        fixed_sub((u64*)&fxy[j], XY); fixed_sub(&fz[j], Z);
    }
    // the same with the +F of atom i collected in integer registers (exactly what the atomics would have summed) and
    // added once by add_acc: for a lane that walks several bonds of one atom
    struct acc_t { u64 xy; int z; };
    static __device__ __forceinline__ acc_t acc_zero() { return acc_t{0, 0}; }
    __device__ __forceinline__ void pair_force_owner(acc_t& A, int j, float dx, float dy, float dz, float s) const {
        const float t = s * AWSEM_F32_FORCE_SCALE;
        const int Z = __float2int_rn(t * dz);
        const u64 XY = (u64)pack_xy(__float2int_rn(t * dx), __float2int_rn(t * dy));
        A.xy += XY; A.z += Z;
        fixed_sub((u64*)&fxy[j], XY); fixed_sub(&fz[j], Z);
    }
    __device__ __forceinline__ void add_acc(int i, const acc_t& A) const { fixed_add((u64*)&fxy[i], A.xy); fixed_add(&fz[i], A.z); }
};

// SoA coordinates and three Q31.32 accumulators per atom: the fp64 check mode (in LDS) and the cell-list path (in HBM)
template <class real> struct SoaOps {
    real *px, *py, *pz;
    u64 *fx, *fy, *fz;
    __device__ __forceinline__ R3<real> pos(int a) const { return R3<real>{px[a], py[a], pz[a]}; }
    __device__ __forceinline__ void set_pos(int a, const R3<real>& r) const { px[a] = r.x; py[a] = r.y; pz[a] = r.z; }
    __device__ __forceinline__ void zero_force(int a) const { fx[a] = 0; fy[a] = 0; fz[a] = 0; }
    __device__ __forceinline__ R3<real> force(int a) const { return R3<real>{fixed_to_real(fx[a], (real)0), fixed_to_real(fy[a], (real)0), fixed_to_real(fz[a], (real)0)}; }
    __device__ __forceinline__ void add_force(int a, real x, real y, real z) const {
        fixed_add(&fx[a], to_fixed(x)); fixed_add(&fy[a], to_fixed(y)); fixed_add(&fz[a], to_fixed(z));
    }
    __device__ __forceinline__ void add_fz(int a, real z) const { fixed_add(&fz[a], to_fixed(z)); }
    __device__ __forceinline__ void pair_force(int i, int j, real dx, real dy, real dz, real s) const {
        const real t = s * (real)AWSEM_FIXED_SCALE;
        const u64 X = to_fixed_scaled(t * dx), Y = to_fixed_scaled(t * dy), Z = to_fixed_scaled(t * dz);
        fixed_add(&fx[i], X); fixed_add(&fy[i], Y); fixed_add(&fz[i], Z);
        fixed_sub(&fx[j], X); fixed_sub(&fy[j], Y); fixed_sub(&fz[j], Z);
    }
    struct acc_t { u64 x, y, z; };
    static __device__ __forceinline__ acc_t acc_zero() { return acc_t{0, 0, 0}; }
    __device__ __forceinline__ void pair_force_owner(acc_t& A, int j, real dx, real dy, real dz, real s) const {
        const real t = s * (real)AWSEM_FIXED_SCALE;
        const u64 X = to_fixed_scaled(t * dx), Y = to_fixed_scaled(t * dy), Z = to_fixed_scaled(t * dz);
        A.x += X; A.y += Y; A.z += Z;
        fixed_sub(&fx[j], X); fixed_sub(&fy[j], Y); fixed_sub(&fz[j], Z);
    }
    __device__ __forceinline__ void add_acc(int i, const acc_t& A) const { fixed_add(&fx[i], A.x); fixed_add(&fy[i], A.y); fixed_add(&fz[i], A.z); }
};
template <> struct Lds<double> : ViewCommon<double>, SoaOps<double> {
    typedef double real_t;
    __device__ __forceinline__ unsigned* scratch() const { return (unsigned*)fx; }   // fx, fy, fz are contiguous: 24 bytes per atom
    double *vx, *vy, *vz, *rx, *ry, *rz;
    __device__ __forceinline__ R3<double> vel(int a) const { return R3<double>{vx[a], vy[a], vz[a]}; }
    __device__ __forceinline__ void set_vel(int a, const R3<double>& r) const { vx[a] = r.x; vy[a] = r.y; vz[a] = r.z; }
    __device__ __forceinline__ R3<double> refpos(int a) const { return R3<double>{rx[a], ry[a], rz[a]}; }
    __device__ __forceinline__ void set_ref(int a, const R3<double>& r) const { rx[a] = r.x; ry[a] = r.y; rz[a] = r.z; }
};

// The view of the step kernel's lean instantiation: the same LDS image; `lean` tells the term bodies that the system holds only the plain
// contact_term (flags 0), so the variants' branches are not compiled in (cflags).
template <class real> struct LdsLean : Lds<real> {
    static constexpr bool lean = true;
    __device__ __forceinline__ explicit LdsLean(const Lds<real>& b) : Lds<real>(b) {}
};
// ... and of the membrane flavour: the one contact term is the z-dependent one (flags 1)
template <class real> struct LdsLeanZ : Lds<real> {
    static constexpr bool lean = true;
    static constexpr int cflag = 1;
    __device__ __forceinline__ explicit LdsLeanZ(const Lds<real>& b) : Lds<real>(b) {}
};
template <class V> __device__ __forceinline__ int cflags(const ContactD<typename V::real_t>& d, const V&) { return V::lean ? V::cflag : d.flags; }

// Energies are summed as 2^-32 fixed point from the first addition on (integer adds commute, so a sum does not
// depend on the order in which a pair list happens to hold its entries): `Energy` is a per-lane accumulator.
struct Energy {
    u64 v = 0;
    __device__ __forceinline__ void operator+=(double x) { v += (u64)__double2ll_rn(x * AWSEM_FIXED_SCALE); }   // round to nearest: no bias over 10^4 terms
};
template <bool ENERGY, class V> __device__ __forceinline__ void flush_energy(const V& L, int group, Energy& e) {
    if (ENERGY) {
        u64 s = e.v;
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if ((threadIdx.x & 63) == 0 && s != 0) fixed_add(&L.eg[group], s);
        e.v = 0;
    }
}

// sigmoid(x) = 1/(1+exp(-x)) = (1+tanh(x/2))/2.  All tanh switches of the reference are evaluated in this form:
// it keeps full relative accuracy in the tails where 1+tanh cancels, and costs one exp + one rcp.
template <class real> __device__ __forceinline__ real sigmoid(real x) { return M<real>::rcp((real)1 + M<real>::exp(-x)); }

// Item loops are block-strided; `rot` shifts which wave takes the first 64 items, so that the short terms of one
// phase (a few dozen dihedrals, the per-site contact pass ...) land on different waves instead of all on wave 0.
__device__ __forceinline__ int rot_tid_n(int rot, int nthr) {
    int t = (int)threadIdx.x + 64 * rot;
    return t >= nthr ? t - nthr : t;
}
#define rot_tid(rot) rot_tid_n((rot), L.nthr)   // every caller has its view `L` in scope
struct Rotor {   // hands out starting waves: each job continues where the previous one stopped
    int nthr, acc = 0;
    __device__ __forceinline__ explicit Rotor(int nthreads) : nthr(nthreads) {}
    __device__ __forceinline__ int next(int nitems) {
        const int nw = nthr >> 6, r = acc & (nw - 1);   // workgroups hold a power-of-two number of waves (Engine::finalize): no scalar division
        acc += (nitems + 63) >> 6;
        return r;
    }
};

// ------------------------------------------------------------------ bonded terms
template <bool ENERGY, class V> __device__ __forceinline__ void term_harmonic(const HarmD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const HarmItem<real> it = d.items[b];
        const R3<real> dd = L.pos(it.i) - L.pos(it.j);
        const real r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z, ir = M<real>::rsqrt(r2), r = r2 * ir, dr = r - it.r0;
        if (ENERGY) e += (double)((real)0.5 * it.k * dr * dr);
        L.pair_force(it.i, it.j, dd.x, dd.y, dd.z, -it.k * dr * ir);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class V> __device__ __forceinline__ void term_chi(const ChiD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const Int4 a = d.atoms[b];   // CA, C, N, CB
        const R3<real> pca = L.pos(a.x), u = pca - L.pos(a.y), v = L.pos(a.z) - pca, w = pca - L.pos(a.w);
        const real ux = u.x, uy = u.y, uz = u.z, vx = v.x, vy = v.y, vz = v.z, wx = w.x, wy = w.y, wz = w.z;
        real uvx = uy * vz - uz * vy, uvy = uz * vx - ux * vz, uvz = ux * vy - uy * vx;
        real vwx = vy * wz - vz * wy, vwy = vz * wx - vx * wz, vwz = vx * wy - vy * wx;
        real wux = wy * uz - wz * uy, wuy = wz * ux - wx * uz, wuz = wx * uy - wy * ux;
        real u2 = ux * ux + uy * uy + uz * uz, v2 = vx * vx + vy * vy + vz * vz, w2 = wx * wx + wy * wy + wz * wz;
        real norm = M<real>::rsqrt(u2 * v2 * w2);
        real chi = (uvx * wx + uvy * wy + uvz * wz) * norm, dchi = chi - d.chi0;
        if (ENERGY) e += (double)(d.k * dchi * dchi);
        real c = (real)-2 * d.k * dchi, cu = chi * M<real>::rcp(u2), cv = chi * M<real>::rcp(v2), cw = chi * M<real>::rcp(w2);
        real gux = vwx * norm - cu * ux, guy = vwy * norm - cu * uy, guz = vwz * norm - cu * uz;
        real gvx = wux * norm - cv * vx, gvy = wuy * norm - cv * vy, gvz = wuz * norm - cv * vz;
        real gwx = uvx * norm - cw * wx, gwy = uvy * norm - cw * wy, gwz = uvz * norm - cw * wz;
        L.add_force(a.x, c * (gux - gvx + gwx), c * (guy - gvy + gwy), c * (guz - gvz + gwz));
        L.add_force(a.y, -c * gux, -c * guy, -c * guz);
        L.add_force(a.z, c * gvx, c * gvy, c * gvz);
        L.add_force(a.w, -c * gwx, -c * gwy, -c * gwz);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// cos/sin of the IUPAC dihedral p0-p1-p2-p3 and the gradient of the angle (Blondel & Karplus form).
template <class real> struct Dihedral { real c, s; real g[4][3]; };
template <class real> __device__ __forceinline__ void dihedral(const R3<real>& q0, const R3<real>& q1, const R3<real>& q2, const R3<real>& q3, Dihedral<real>& D) {
    const R3<real> F = q0 - q1, G = q1 - q2, H = q3 - q2;
    real Ax = F.y * G.z - F.z * G.y, Ay = F.z * G.x - F.x * G.z, Az = F.x * G.y - F.y * G.x;
    real Bx = H.y * G.z - H.z * G.y, By = H.z * G.x - H.x * G.z, Bz = H.x * G.y - H.y * G.x;
    real A2 = Ax * Ax + Ay * Ay + Az * Az, B2 = Bx * Bx + By * By + Bz * Bz, G2 = G.x * G.x + G.y * G.y + G.z * G.z;
    real iAB = M<real>::rsqrt(A2 * B2), iG = M<real>::rsqrt(G2), Gn = G2 * iG;
    D.c = (Ax * Bx + Ay * By + Az * Bz) * iAB;
    real Cx = By * Az - Bz * Ay, Cy = Bz * Ax - Bx * Az, Cz = Bx * Ay - By * Ax;   // B x A
    D.s = (Cx * G.x + Cy * G.y + Cz * G.z) * iAB * iG;
    real FG = F.x * G.x + F.y * G.y + F.z * G.z, HG = H.x * G.x + H.y * G.y + H.z * G.z;
    const real iA2 = M<real>::rcp(A2), iB2 = M<real>::rcp(B2);
    real a = Gn * iA2, b = Gn * iB2, ta = FG * iA2 * iG, tb = HG * iB2 * iG;
    real A[3] = {Ax, Ay, Az}, B[3] = {Bx, By, Bz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        real dA = -a * A[k], dB = b * B[k], t = ta * A[k] - tb * B[k];
        D.g[0][k] = dA; D.g[3][k] = dB; D.g[1][k] = -dA + t; D.g[2][k] = -dB - t;
    }
}

template <bool ENERGY, class V> __device__ __forceinline__ void term_rama(const RamaD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const Int5 a = d.atoms[b];
        real W[AWSEM_MAX_WELLS];
#pragma unroll
        for (int w = 0; w < AWSEM_MAX_WELLS; w++) W[w] = w < d.nwell ? d.bw[b * d.nwell + w] : (real)0;   // -k * w * weight, folded
        const R3<real> q0 = L.pos(a.a[0]), q1 = L.pos(a.a[1]), q2 = L.pos(a.a[2]), q3 = L.pos(a.a[3]), q4 = L.pos(a.a[4]);
        Dihedral<real> P, S;
        dihedral(q0, q1, q2, q3, P);
        dihedral(q1, q2, q3, q4, S);
        real dEdphi = 0, dEdpsi = 0;
#pragma unroll
        for (int w = 0; w < AWSEM_MAX_WELLS; w++) {
            if (w >= d.nwell) break;
            real cp = P.c * d.cphi0[w] + P.s * d.sphi0[w] - (real)1, sp = P.s * d.cphi0[w] - P.c * d.sphi0[w];
            real cs = S.c * d.cpsi0[w] + S.s * d.spsi0[w] - (real)1, ss = S.s * d.cpsi0[w] - S.c * d.spsi0[w];
            real ex = W[w] * M<real>::exp(-d.sigma[w] * (d.omega_phi[w] * cp * cp + d.omega_psi[w] * cs * cs));
            if (ENERGY) e += (double)ex;
            dEdphi += ex * (real)2 * d.sigma[w] * d.omega_phi[w] * cp * sp;
            dEdpsi += ex * (real)2 * d.sigma[w] * d.omega_psi[w] * cs * ss;
        }
        // atoms 1..3 receive both contributions
        L.add_force(a.a[0], -dEdphi * P.g[0][0], -dEdphi * P.g[0][1], -dEdphi * P.g[0][2]);
#pragma unroll
        for (int m = 1; m < 4; m++)
            L.add_force(a.a[m], -dEdphi * P.g[m][0] - dEdpsi * S.g[m - 1][0], -dEdphi * P.g[m][1] - dEdpsi * S.g[m - 1][1],
                        -dEdphi * P.g[m][2] - dEdpsi * S.g[m - 1][2]);
        L.add_force(a.a[4], -dEdpsi * S.g[3][0], -dEdpsi * S.g[3][1], -dEdpsi * S.g[3][2]);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// All Ramachandran terms of a group as one item list: the same arithmetic as term_rama, item by item, with the term's well
// parameters read per lane from the LDS copy of the parameter sets.
template <bool ENERGY, class V> __device__ __forceinline__ void term_rama_merged(const RamaMergedD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const RamaItem a = d.items[b];
        real W[AWSEM_MAX_WELLS];
#pragma unroll
        for (int w = 0; w < AWSEM_MAX_WELLS; w++) W[w] = w < d.nwell ? d.bw[b * d.nwell + w] : (real)0;
        const R3<real> q0 = L.pos(a.a[0]), q1 = L.pos(a.a[1]), q2 = L.pos(a.a[2]), q3 = L.pos(a.a[3]), q4 = L.pos(a.a[4]);
        Dihedral<real> P, S;
        dihedral(q0, q1, q2, q3, P);
        dihedral(q1, q2, q3, q4, S);
        const real* par = L.ramap + a.set * d.nwell * 8;
        real dEdphi = 0, dEdpsi = 0;
#pragma unroll
        for (int w = 0; w < AWSEM_MAX_WELLS; w++) {
            if (w >= d.nwell) break;
            const real sigma = par[8 * w], omega_phi = par[8 * w + 1], omega_psi = par[8 * w + 2];
            const real cphi0 = par[8 * w + 3], sphi0 = par[8 * w + 4], cpsi0 = par[8 * w + 5], spsi0 = par[8 * w + 6];
            real cp = P.c * cphi0 + P.s * sphi0 - (real)1, sp = P.s * cphi0 - P.c * sphi0;
            real cs = S.c * cpsi0 + S.s * spsi0 - (real)1, ss = S.s * cpsi0 - S.c * spsi0;
            real ex = W[w] * M<real>::exp(-sigma * (omega_phi * cp * cp + omega_psi * cs * cs));
            if (ENERGY) e += (double)ex;
            dEdphi += ex * (real)2 * sigma * omega_phi * cp * sp;
            dEdpsi += ex * (real)2 * sigma * omega_psi * cs * ss;
        }
        L.add_force(a.a[0], -dEdphi * P.g[0][0], -dEdphi * P.g[0][1], -dEdphi * P.g[0][2]);
#pragma unroll
        for (int m = 1; m < 4; m++)
            L.add_force(a.a[m], -dEdphi * P.g[m][0] - dEdpsi * S.g[m - 1][0], -dEdphi * P.g[m][1] - dEdpsi * S.g[m - 1][1],
                        -dEdphi * P.g[m][2] - dEdpsi * S.g[m - 1][2]);
        L.add_force(a.a[4], -dEdpsi * S.g[3][0], -dEdpsi * S.g[3][1], -dEdpsi * S.g[3][2]);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// theta(x;y,z) = exp(-(|x-y|-r_ON)^2/(2 s_NO^2) - (|x-z|-r_OH)^2/(2 s_HO^2)); g1,g2 = d theta / d r1, d r2
template <class real> __device__ __forceinline__ real hb_theta(real r1, real r2, real r_ON, real r_OH, real inv_sNO2, real inv_sHO2, real& g1, real& g2) {
    real a = r1 - r_ON, b = r2 - r_OH;
    real th = M<real>::exp((real)-0.5 * (a * a * inv_sNO2 + b * b * inv_sHO2));
    g1 = -th * a * inv_sNO2; g2 = -th * b * inv_sHO2;
    return th;
}

template <bool ENERGY, class V> __device__ __forceinline__ void term_helical(const HelicalD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const HelicalItem<real> it = d.items[b];
        const R3<real> po = L.pos(it.o), A = po - L.pos(it.n), B = po - L.pos(it.h);
        real r1s = A.x * A.x + A.y * A.y + A.z * A.z, r2s = B.x * B.x + B.y * B.y + B.z * B.z;
        real i1 = M<real>::rsqrt(r1s), i2 = M<real>::rsqrt(r2s), g1, g2;
        real th = hb_theta(r1s * i1, r2s * i2, d.r_ON, d.r_OH, d.inv_sNO2, d.inv_sHO2, g1, g2);
        real coef = it.coef;
        if (d.zitems) {   // uniform: z_dependent_helical_term, coef = (1 - alpha) coef_water + alpha coef_membrane, alpha(z of CA_i)
            const HelicalZItem<real> zi = d.zitems[b];
            const real z = L.pos(zi.ca).z - d.center;
            // 1/2 tanh(u) + 1/2 tanh(v) = sig(2u) + sig(2v) - 1
            const real su = sigmoid(d.k_m2 * (z + d.z_m)), sv = sigmoid(d.k_m2 * (d.z_m - z));
            coef += zi.dcoef * (su + sv - (real)1);
            L.add_fz(zi.ca, th * zi.dcoef * d.k_m2 * (su * ((real)1 - su) - sv * ((real)1 - sv)));   // -dE/dz, E = -coef theta
        }
        if (ENERGY) e += (double)(-coef * th);
        L.pair_force(it.o, it.n, A.x, A.y, A.z, coef * g1 * i1);
        L.pair_force(it.o, it.h, B.x, B.y, B.z, coef * g2 * i2);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// Fragment memory: per bond two neighbouring floats of an L2-resident table row.  U bonds are kept in flight per
// thread so that the dependent chain (pair word -> LDS coordinates -> table gather) overlaps across bonds; all index
// arithmetic is unsigned 32-bit so that the loads use the scalar-base + 32-bit-offset addressing form.
template <bool ENERGY, int U, bool PACKED, class V> __device__ __forceinline__ void fm_body(const FmD<typename V::real_t>& d, const V& L, unsigned b0, unsigned stride, Energy& e) {
    typedef typename V::real_t real;
    unsigned ai[U], aj[U], i1[U], bb[U];
    real r[U], ir[U];
    real2_t<real> v[U];
    const unsigned nrow = (unsigned)d.ntab - 1u, n = (unsigned)d.n;
#pragma unroll
    for (int u = 0; u < U; u++) {
        bb[u] = b0 + u * stride;
        const unsigned b = bb[u] < n ? bb[u] : b0;   // lanes past the end re-read their first bond and skip the bookkeeping
        if (PACKED) {
            const unsigned w = d.pairs[b];           // atom_i | atom_j << 16
            ai[u] = w & 0xffffu; aj[u] = w >> 16;
        } else {
            ai[u] = (unsigned)d.pairs2[2 * b]; aj[u] = (unsigned)d.pairs2[2 * b + 1];
        }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        const unsigned b = bb[u] < n ? bb[u] : b0;
        const R3<real> dd = L.pos(ai[u]) - L.pos(aj[u]);
        const real r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;
        ir[u] = M<real>::rsqrt(r2); r[u] = r2 * ir[u];
        const unsigned bin = (unsigned)(int)(r[u] * d.inv_dr);   // r >= 0: truncation is floor
        i1[u] = bin < nrow - 1u ? bin : nrow - 1u;
#ifdef AWSEM_EXP_FM_NOGATHER
        v[u].x = r[u]; v[u].y = r[u] + (real)1;
#else
        v[u] = d.table2[b * nrow + i1[u]];                       // (t[i], t[i+1]) in one load
#endif
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
        if (bb[u] >= n) continue;
        const real slope = (v[u].y - v[u].x) * d.inv_dr;
        if (ENERGY) e += (double)(-d.k * (v[u].x + slope * (r[u] - d.dr * (real)i1[u])));
#ifdef AWSEM_EXP_FM_NOATOMIC
        if (slope * ir[u] == (real)123.456) L.zero_force(0);
#else
        const R3<real> dd = L.pos(ai[u]) - L.pos(aj[u]);   // read again rather than kept in registers across the gather
        L.pair_force(ai[u], aj[u], dd.x, dd.y, dd.z, d.k * slope * ir[u]);
#endif
    }
}
template <bool ENERGY, bool PACKED = true, class V> __device__ __forceinline__ void term_fm(const FmD<typename V::real_t>& d, const V& L, int rot = 0) {
    Energy e;
    constexpr int U = 6;
    const unsigned stride = L.nthr, n = (unsigned)d.n;
    for (unsigned b = (unsigned)rot_tid(rot); b < n; b += U * stride) fm_body<ENERGY, U, PACKED>(d, L, b, stride, e);
    flush_energy<ENERGY>(L, d.group, e);
}

// Bundled walk (replica-resident path): one lane per bundle of <= AWSEM_FM_BUNDLE bonds around one owner atom.  The owner's
// position is read once, its force is summed in integer registers (bit-identical to the atomics it replaces) and added once:
// a third fewer LDS reads and atomics than bond-by-bond, all table gathers of the bundle in flight together.
// two neighbouring floats in one 8-byte load at 4-byte alignment (global_load_dwordx2 needs dword alignment only)
struct __attribute__((packed, aligned(4))) Float2U { float x, y; };
template <bool ENERGY, class V> __device__ __forceinline__ void term_fm_bundled(const FmD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    constexpr int K = AWSEM_FM_BUNDLE;
    // fp32 force walks read the cropped rows (FmD::crop): a bin outside a bond's window has k |slope| < 2^-15 and adds exact zeros to the
    // Q19.12 accumulators, so it is skipped -- bit-identical forces from a table a quarter of the size (2xov: 27.6 MB -> 3.8 MB, inside L2).
    // Energy evaluations and the fp64 mode read the full rows.
    constexpr bool CROP = !ENERGY && sizeof(real) == 4;
    Energy e;
    const unsigned nrow = (unsigned)d.ntab - 1u;
    const unsigned clen = (unsigned)d.crop_len;
    for (int k = rot_tid(rot); k < d.nbundles; k += L.nthr) {
        const FmBundle h = d.bundles[k];
        const unsigned o = h.head & 0xffffu, cnt = h.head >> 16;
        unsigned pj[K], i1[K];
        real r[K], ir[K];
        real2_t<real> v[K];
#pragma unroll
        for (int u = 0; u < K; u++) pj[u] = d.partners[(unsigned)u * (unsigned)d.nb_pad + (unsigned)k];   // atom | lo << 16; slots past the count repeat the first partner
        const R3<real> po = L.pos(o);
#pragma unroll
        for (int u = 0; u < K; u++) {
            const R3<real> dd = po - L.pos(pj[u] & 0xffffu);
            const real r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;
            ir[u] = M<real>::rsqrt(r2); r[u] = r2 * ir[u];
            const unsigned bin = (unsigned)(int)(r[u] * d.inv_dr);   // r >= 0: truncation is floor
            i1[u] = bin < nrow - 1u ? bin : nrow - 1u;
            const unsigned bond = h.first + ((unsigned)u < cnt ? (unsigned)u : 0u);
            if (CROP) {
                const unsigned c = i1[u] - (pj[u] >> 16);            // wraps below the window
                i1[u] = c;                                           // kept as the in-window test
                const Float2U w = *(const Float2U*)((const float*)d.crop + bond * clen + (c < clen - 1u ? c : 0u));
                v[u].x = (real)w.x; v[u].y = (real)w.y;
            } else
                v[u] = d.table2[bond * nrow + i1[u]];                // (t[i], t[i+1]) in one load
        }
        typename V::acc_t A = V::acc_zero();
#pragma unroll
        for (int u = 0; u < K; u++) {
            if ((unsigned)u >= cnt) continue;
            if (CROP && i1[u] >= clen - 1u) continue;
            const real slope = (v[u].y - v[u].x) * d.inv_dr;
            if (ENERGY) e += (double)(-d.k * (v[u].x + slope * (r[u] - d.dr * (real)i1[u])));
            const R3<real> dd = po - L.pos(pj[u] & 0xffffu);   // read again rather than kept in registers across the gather
            L.pair_force_owner(A, pj[u] & 0xffffu, dd.x, dd.y, dd.z, d.k * slope * ir[u]);
        }
        L.add_acc(o, A);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class V> __device__ __forceinline__ void term_memb(const MembD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const MembItem<real> it = d.items[b];
        if (d.kind != 0) {   // uniform: the other external-field terms (awsem_add_external)
            const R3<real> x = L.pos(it.atom);
            if (d.kind == 4) {   // k |z - center - z_i|
                const real u = x.z - d.center - it.scale;
                if (ENERGY) e += (double)(d.k * (u < 0 ? -u : u));
                L.add_fz(it.atom, u > 0 ? -d.k : (u < 0 ? d.k : (real)0));
            } else {             // -k s_i x (or y, z): a constant force k s_i along the axis
                const real c = d.kind == 1 ? x.x : (d.kind == 2 ? x.y : x.z), fc = d.k * it.scale;
                if (ENERGY) e += (double)(-fc * c);
                L.add_force(it.atom, d.kind == 1 ? fc : (real)0, d.kind == 2 ? fc : (real)0, d.kind == 3 ? fc : (real)0);
            }
            continue;
        }
        real z = L.pos(it.atom).z - d.center;
        // 1/2 tanh(u) + 1/2 tanh(v) = sig(2u) + sig(2v) - 1
        real su = sigmoid(d.k_m2 * (z + d.z_m)), sv = sigmoid(d.k_m2 * (d.z_m - z));
        if (ENERGY) e += (double)(d.k * it.scale * (su + sv - (real)1));
        real dEdz = d.k * it.scale * d.k_m2 * (su * ((real)1 - su) - sv * ((real)1 - sv));
        L.add_fz(it.atom, -dEdz);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class V> __device__ __forceinline__ void term_dh(const DhD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const DhItem<real> it = d.items[b];
        const R3<real> dd = L.pos(it.i) - L.pos(it.j);
        real r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z, ir = M<real>::rsqrt(r2), r = r2 * ir;
        real E = d.k * it.qq * ir * M<real>::exp(-r * d.inv_lam);
        if (ENERGY) e += (double)E;
        L.pair_force(it.i, it.j, dd.x, dd.y, dd.z, E * (ir + d.inv_lam) * ir);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// Bond sums: er_term templateTerms.py:643-693, additive_amhgo_term :600-640, q_value biasTerms.py:146-188 (CustomBondForce)
// and the CustomCVForce wrappers over one: qbias_term biasTerms.py:270-302, rg_term :461-486, rg_bias_term :489-528.
// bond_value: g and s = -(dg/dr)/r of one bond.
template <class real> __device__ __forceinline__ void bond_value(const GaussD<real>& d, const GaussItem<real>& it, const R3<real>& dd, real& g, real& s) {
    const real r2 = dd.x * dd.x + dd.y * dd.y + dd.z * dd.z;
    if (d.kind != 0) { g = it.c * r2; s = (real)-2 * it.c; return; }   // r^2 kinds (kind 2: the callers have projected dd on the xy plane)
    const real ir = M<real>::rsqrt(r2), dr = r2 * ir - it.r0;
    g = -it.c * M<real>::exp(-it.a * dr * dr);
    s = (real)2 * it.a * dr * g * ir;   // dg/dr = -2 a dr g
}
template <bool ENERGY, class V> __device__ __forceinline__ void term_gauss(const GaussD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    Energy e;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const GaussItem<real> it = d.items[b];
        R3<real> dd = L.pos(it.i) - L.pos(it.j);
        if (d.kind == 2) dd.z = 0;
        real g, s;
        bond_value(d, it, dd, g, s);
        if (ENERGY) e += (double)g;
        L.pair_force(it.i, it.j, dd.x, dd.y, dd.z, s);
    }
    flush_energy<ENERGY>(L, d.group, e);
}
// first pass of a collective-variable term: S into L.cv[d.slot] (zero on entry; fixed point, so order-independent).
// 2^-48 resolution (|S| < 32768): the 2^-32 of the energy accumulators would round the ~10^4 nearly equal bonds of a
// native structure coherently and shift Q by 2e-7, which a stiff umbrella turns into 1e-5 kJ/mol.
#define AWSEM_CV_SCALE 281474976710656.0
template <class V> __device__ __forceinline__ void cv_sum(const GaussD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    u64 t = 0;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const GaussItem<real> it = d.items[b];
        real g, s;
        R3<real> dd = L.pos(it.i) - L.pos(it.j);
        if (d.kind == 2) dd.z = 0;
        bond_value(d, it, dd, g, s);
        t += (u64)__double2ll_rn((double)g * AWSEM_CV_SCALE);
    }
    for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off);
    if ((threadIdx.x & 63) == 0 && t != 0) fixed_add(&L.cv[d.slot], t);
}
// second pass (after a barrier): E = f(S) once, forces f'(S) * (-dg/dr) along every bond
template <bool ENERGY, class V> __device__ __forceinline__ void cv_apply(const GaussD<typename V::real_t>& d, const V& L, int rot = 0) {
    typedef typename V::real_t real;
    const double S = (double)(long long)L.cv[d.slot] * (1.0 / AWSEM_CV_SCALE), k = (double)d.cv_k, x0 = (double)d.cv_x0;
    double E, fp;
    if (d.cv == 1) { E = 0.5 * k * (S - x0) * (S - x0); fp = k * (S - x0); }
    else {
        const double q = S > 0 ? ::sqrt(S) : 0.0, iq = q > 0 ? 1.0 / q : 0.0;
        if (d.cv == 2) { E = k * q; fp = 0.5 * k * iq; }
        else { E = k * (q - x0) * (q - x0); fp = k * (q - x0) * iq; }
    }
    if (ENERGY && threadIdx.x == 0) fixed_add(&L.eg[d.group], (u64)__double2ll_rn(E * AWSEM_FIXED_SCALE));
    const real fpr = (real)fp;
    for (int b = rot_tid(rot); b < d.n; b += L.nthr) {
        const GaussItem<real> it = d.items[b];
        R3<real> dd = L.pos(it.i) - L.pos(it.j);
        if (d.kind == 2) dd.z = 0;
        real g, s;
        bond_value(d, it, dd, g, s);
        L.pair_force(it.i, it.j, dd.x, dd.y, dd.z, fpr * s);
    }
}

// ------------------------------------------------------------------ Verlet pair lists
// Every pair term (excl, contact, beta, pap) is evaluated from a per-replica pair list instead of an all-pairs
// scan.  BUILD (amortised: only when some atom has moved more than skin/2 since the last build) enumerates the
// candidate pairs along *diagonals* -- in one wave iteration lane l tests pair (i0 + l, i0 + l + d) -- so that the
// entries one iteration appends have pairwise distinct first and distinct second members; the wave compacts the
// passing lanes (ballot + mbcnt) behind one LDS counter bump and stores them to the replica's global list with one
// coalesced store.  EVALUATE reads the list 64 entries per wave iteration with every lane busy, and books both
// sides of the pair with fixed-point LDS atomics (conflict-light thanks to the diagonal order).  The evaluators
// re-apply the exact cutoff of the reference, so the result does not depend on when a list was built.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
#define AWSEM_BUILD_UNROLL 1   // measured: 4 chunks in flight do not help (the builders are LDS-throughput bound) and cost small systems 10 %

__device__ __forceinline__ void list_append(unsigned* __restrict__ e, unsigned* cnt, int cap, int* status, bool pass, unsigned entry) {
    const u64 m = __ballot(pass);
    if (m == 0) return;   // wave-uniform
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(cnt, (unsigned)__popcll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    const unsigned off = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (pass) {
        if (base + off < (unsigned)cap) e[base + off] = entry;
        else *status = 1;   // cannot happen while cap = number of candidate pairs; kept as a guard
    }
}

// the same for the hydrogen-bond lists' records
template <class real> __device__ __forceinline__ void list_append_entry(HbEntry<real>* __restrict__ e, unsigned* cnt, int cap, int* status, bool pass, const HbEntry<real>& ent) {
    const u64 m = __ballot(pass);
    if (m == 0) return;   // wave-uniform
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(cnt, (unsigned)__popcll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    const unsigned off = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (pass) {
        if (base + off < (unsigned)cap) e[base + off] = ent;
        else *status = 1;
    }
}

#define AWSEM_WAVE_VARS const int nw = L.nthr >> 6, wave = uniform(threadIdx.x >> 6), lane = threadIdx.x & 63
// list chunks of 64 entries are dealt to the waves round-robin, starting at a different wave for every list so
// that short lists do not all land on wave 0
#define AWSEM_LIST_CHUNKS(c0, count, rot) for (int c0 = ((wave - (rot)) & (nw - 1)) * 64; c0 < (count); c0 += nw * 64)   // nw is a power of two

struct ListView { const unsigned* e; int count; };
template <class V> __device__ __forceinline__ ListView list_view(const V& L, const ListSlot& ls) {
    const int n = (int)L.lcount[ls.list_id];
    return ListView{L.lists + ls.list_off, n < ls.list_cap ? n : ls.list_cap};
}

// ---- excluded volume: same-class site pairs that are not bonded partners
// The builders' candidate loops read their per-site tables (site -> atom, site info, acceptor / donor words) from LDS:
// a table is staged into the force accumulators' memory, which is all zero and unused while lists are built
// (build_lists clears it again afterwards).  A global load per candidate would put ~1 us of latency into every
// iteration of these loops.
// site records of the lean builders: position and a word (atom, or site info) side by side, so that a candidate costs ONE conflict-free
// 16-byte LDS read (the site -> atom table and a float4 at a stride of six atoms were two dependent reads, the second on shared banks)
template <class real> struct __attribute__((aligned(16))) SiteRec { real x, y, z; unsigned w; };
template <bool LEAN = false, class V> __device__ __forceinline__ void build_excl(const ExclD<typename V::real_t>& d, const V& L, int* status, int natoms) {
    typedef typename V::real_t real;
    AWSEM_WAVE_VARS;
    unsigned* e = L.lists + d.ls.list_off;
    unsigned* cnt = &L.lcount[d.ls.list_id];
    // LEAN (host: the records fit the staging area): a lane keeps site i of the class in registers and meets j = i + dd (mod n), dd = 1 ..
    // n/2 dealt to the waves: every unordered pair once (for even n the last dd would come twice: its second half is skipped), the
    // entries of one append still pairwise distinct in both members, and one LDS read per candidate instead of four.
    if (LEAN || (sizeof(real) == 4 && d.nsites <= natoms)) {   // (the full fp32 kernel: when the 16-byte records fit the accumulators' 16 bytes per atom, uniform)
        SiteRec<real>* rec = (SiteRec<real>*)L.scratch();
        for (int i = threadIdx.x; i < d.nsites; i += L.nthr) { const int a = d.atom[i]; const R3<real> p = L.pos(a); rec[i] = SiteRec<real>{p.x, p.y, p.z, (unsigned)a}; }
        __syncthreads();
        for (int c = 0; c < d.nclass; c++) {
            const int base = d.class_start[c], n = d.class_start[c + 1] - base, half = n >> 1;
            for (int i0 = 0; i0 < n; i0 += 64) {
                const int i = i0 + lane;
                const bool vi = i < n;
                const SiteRec<real> Ri = rec[base + (vi ? i : 0)];
                const bool low = 2 * i < n;   // (i < n / 2 for even n)
                const Int4 p = d.partners[base + (vi ? i : 0)];   // the lane's bonded partners: once per site, not per survivor
                for (int dd = 1 + wave; dd <= half; dd += nw) {
                    int j = i + dd;
                    j = j >= n ? j - n : j;
                    const SiteRec<real> Rj = rec[base + (vi ? j : 0)];
                    const int b = (int)Rj.w;
                    const bool pass = vi && dist2(Ri.x - Rj.x, Ri.y - Rj.y, Ri.z - Rj.z) < d.list_rc2 && (2 * dd != n || low) &&
                                      p.x != b && p.y != b && p.z != b && p.w != b;
                    list_append(e, cnt, d.ls.list_cap, status, pass, (Ri.w << 16) | Rj.w);
                }
            }
        }
        __syncthreads();
        return;
    }
    unsigned* tab = L.scratch();
    for (int i = threadIdx.x; i < d.nsites; i += L.nthr) tab[i] = (unsigned)d.atom[i];
    __syncthreads();
    constexpr int U = AWSEM_BUILD_UNROLL;   // candidate chunks in flight: the LDS round trips of one chunk hide behind the others
    for (int c = 0; c < d.nclass; c++) {
        const int base = d.class_start[c], n = d.class_start[c + 1] - base;
        // diagonal dd (n - dd pairs) and diagonal n - dd (dd pairs) share an iteration: n candidates, every lane busy when n is near 64
        for (int dd = 1 + wave; dd <= (n >> 1); dd += nw) {
            const int n1 = n - dd, nboth = n1 + (n1 != dd ? dd : 0);
            for (int i0 = 0; i0 < nboth; i0 += 64 * U) {
                int a[U], b[U], si[U];
                bool pass[U];
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int idx = i0 + 64 * u + lane;
                    const bool v = idx < nboth;
                    const int i = idx < n1 ? idx : idx - n1, j = idx < n1 ? idx + dd : idx;   // second part: (idx - n1, idx - n1 + n - dd)
                    si[u] = base + (v ? i : 0);
                    a[u] = (int)tab[si[u]]; b[u] = (int)tab[base + (v ? j : 0)];
                    pass[u] = v;
                }
#pragma unroll
                for (int u = 0; u < U; u++) pass[u] = pass[u] && dist2(L.pos(a[u]) - L.pos(b[u])) < d.list_rc2;
#pragma unroll
                for (int u = 0; u < U; u++) {
                    if (__ballot(pass[u]) == 0) continue;
                    if (pass[u]) {
                        const Int4 p = d.partners[si[u]];
                        pass[u] = p.x != b[u] && p.y != b[u] && p.z != b[u] && p.w != b[u];
                    }
                    list_append(e, cnt, d.ls.list_cap, status, pass[u], ((unsigned)a[u] << 16) | (unsigned)b[u]);
                }
            }
        }
    }
    __syncthreads();
}
// one candidate pair per lane (v = lane holds a pair); shared by the replica-resident and the cell-list path
template <bool ENERGY, class V> __device__ __forceinline__ void excl_body(const ExclD<typename V::real_t>& d, const V& L, int a, int b, bool v, Energy& e) {
    typedef typename V::real_t real;
    const R3<real> dd = L.pos(a) - L.pos(b);
    const real r2 = dist2(dd);
    if (v && r2 < d.r0 * d.r0) {   // clashes are rare: divergent body
        const real ir = M<real>::rsqrt(r2), dr = r2 * ir - d.r0;
        if (ENERGY) e += (double)(d.k * dr * dr);
        L.pair_force(a, b, dd.x, dd.y, dd.z, (real)-2 * d.k * dr * ir);
    }
}
template <bool ENERGY, class V> __device__ __forceinline__ void eval_excl(const ExclD<typename V::real_t>& d, const V& L) {
    AWSEM_WAVE_VARS;
    Energy e;
    const ListView lv = list_view(L, d.ls);
    AWSEM_LIST_CHUNKS(c0, lv.count, d.ls.list_id) {
        const int idx = c0 + lane;
        const bool v = idx < lv.count;
        const unsigned ent = v ? lv.e[idx] : 0u;
        excl_body<ENERGY>(d, L, ent >> 16, ent & 0xffff, v, e);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// ---- contact: site pairs with |resid_i - resid_j| >= 2 (the density's own exclusion; the pair energy needs more)
template <class V> __device__ __forceinline__ void build_contact(const ContactD<typename V::real_t>& d, const V& L, const int* satom, int* status) {
    typedef typename V::real_t real;
    AWSEM_WAVE_VARS;
    unsigned* e = L.lists + d.ls.list_off;
    unsigned* cnt = &L.lcount[d.ls.list_id];
    const int n = d.nsites;
    unsigned* tab = L.scratch();
    for (int i = threadIdx.x; i < n; i += L.nthr) tab[i] = (unsigned)d.info[i];
    __syncthreads();
    constexpr int U = AWSEM_BUILD_UNROLL;
    for (int dd = 1 + wave; dd <= (n >> 1); dd += nw) {   // diagonals dd and n - dd together, as in build_excl
        const int n1 = n - dd, nboth = n1 + (n1 != dd ? dd : 0);
        for (int i0 = 0; i0 < nboth; i0 += 64 * U) {
            int si[U], sj[U], info_i[U], info_j[U], a[U], b[U];
            bool v[U];
            real r2[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int idx = i0 + 64 * u + lane;
                const int i = idx < n1 ? idx : idx - n1, j = idx < n1 ? idx + dd : idx;
                v[u] = idx < nboth;
                si[u] = v[u] ? i : 0; sj[u] = v[u] ? j : 0;
                info_i[u] = (int)tab[si[u]]; info_j[u] = (int)tab[sj[u]]; a[u] = satom[si[u]]; b[u] = satom[sj[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++) r2[u] = dist2(L.pos(a[u]) - L.pos(b[u]));
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int dres = (info_j[u] & 0xffff) - (info_i[u] & 0xffff);
                // the density counts |resid_i - resid_j| >= 2 whatever the chains (contactTerms.py:255); the pair energy
                // counts different chains or |resid_i - resid_j| >= min_seq_sep (res_table, contactTerms.py:200-210)
                const bool in_rho = dres >= 2 || dres <= -2;
                const bool pair_e = !(cflags(d, L) & 2) && ((((info_i[u] ^ info_j[u]) >> 16) & 0x7ff) != 0 || dres >= d.min_seq_sep || -dres >= d.min_seq_sep);
                const bool pass = v[u] && (in_rho || pair_e) && r2[u] < d.list_rc2;
                const unsigned ty = (unsigned)(((info_i[u] >> 27) & 31) * 20 + ((info_j[u] >> 27) & 31));
                list_append(e, cnt, d.ls.list_cap, status, pass,
                            (unsigned)si[u] | ((unsigned)sj[u] << 10) | (ty << 20) | (in_rho ? 0x40000000u : 0u) | (pair_e ? 0x80000000u : 0u));
            }
        }
    }
    __syncthreads();
}
// pass 1: density rho_i = sum_j theta_I(r_ij) inside the cutoff
template <class V> __device__ __forceinline__ void density_body(const ContactD<typename V::real_t>& d, const V& L, int i, int j, int a, int b, bool v) {
    typedef typename V::real_t real;
    const real r2 = dist2(L.pos(a) - L.pos(b));
    const bool pass = v && r2 < d.cutoff2;
    if (__ballot(pass) == 0) return;
    const real r = M<real>::sqrt(r2);
    const real th = sigmoid(d.eta2 * (r - d.rmin)) * sigmoid(d.eta2 * (d.rmax - r));
    if (pass) {
        const u64 t = to_fixed(th);
        fixed_add(&L.rho[i], t);
        fixed_add(&L.rho[j], t);
    }
}
template <class V> __device__ __forceinline__ void contact_density(const ContactD<typename V::real_t>& d, const V& L, const int* satom) {
    AWSEM_WAVE_VARS;
    const ListView lv = list_view(L, d.ls);
    AWSEM_LIST_CHUNKS(c0, lv.count, d.ls.list_id) {
        const int idx = c0 + lane;
        const bool v = idx < lv.count;
        const unsigned ent = v ? lv.e[idx] : 0u;
        const int i = ent & 0x3ff, j = (ent >> 10) & 0x3ff;
        density_body(d, L, i, j, satom[i], satom[j], v && (ent & 0x40000000u));
    }
}
// Between pass 1 and 2: evaluate the burial wells (single-particle energy), seed dE/drho, and precompute the
// per-site water switch s_i = sig(-2 eta_s (rho_i - rho0)) (sigma_water_ij = s_i s_j) into L.rhov.
template <bool ENERGY, class V> __device__ __forceinline__ void single_body(const ContactD<typename V::real_t>& d, const V& L, int i, int atom, Energy& e) {
    typedef typename V::real_t real;
    real rho = fixed_to_real(L.rho[i], (real)0);
    L.rhov[i] = sigmoid(-d.eta_s2 * (rho - d.rho0));
    real dE = 0, al = 0, dal = 0;
    if (cflags(d, L) & 1) {
        real z = L.pos(atom).z - d.zc;
        real su = sigmoid(d.eta_sw2 * (z + d.z_m)), sv = sigmoid(d.eta_sw2 * (d.z_m - z));
        al = su + sv - (real)1;
        dal = d.eta_sw2 * (su * ((real)1 - su) - sv * ((real)1 - sv));
        L.alpha[i] = al;
        L.dalpha[i] = dal;
    }
    if (d.has_burial) {
        int t = (cflags(d, L) & 16) ? i : (d.info[i] >> 27) & 31;   // index based: wells per site
        real dz = 0;
#pragma unroll
        for (int m = 0; m < 3; m++) {
            real gb = d.bgamma[t * 3 + m];
            // tanh(a) + tanh(b) = 2 (sig(2a) + sig(2b) - 1)
            real sa = sigmoid(d.kappa2 * (rho - d.rho_lo[m])), sb = sigmoid(d.kappa2 * (d.rho_hi[m] - rho));
            if (cflags(d, L) & 8) {   // hybrid contact: the wells blend water and membrane depths with the site's own alpha(z)
                const real gm = d.bgamma[60 + t * 3 + m];
                dz += -d.kb * (gm - gb) * (sa + sb - (real)1);
                gb = ((real)1 - al) * gb + al * gm;
            }
            if (ENERGY) e += (double)(-d.kb * gb * (sa + sb - (real)1));
            dE += -d.kb * gb * d.kappa2 * (sa * ((real)1 - sa) - sb * ((real)1 - sb));
        }
        if (cflags(d, L) & 8) L.add_fz(atom, -dz * dal);
    }
    L.drho[i] = to_fixed(dE);
}
template <bool ENERGY, class V> __device__ __forceinline__ void contact_single(const ContactD<typename V::real_t>& d, const V& L, const int* satom, int rot = 0) {
    Energy e;
    for (int i = rot_tid(rot); i < d.nsites; i += L.nthr) single_body<ENERGY>(d, L, i, satom[i], e);
    flush_energy<ENERGY>(L, d.group, e);
}
// pass 2: pair energy (direct + mediated), its explicit radial force, and dE/drho through sigma_water.
// contact_pair_math: the arithmetic for one pair at distance r (shared by the list-driven passes of both paths):
// E, dE/dr, dE/d(sigma_water) and, for the z-dependent variant, M - W (membrane minus water energy).
template <class real> struct PairTerms { real E, dEdr, dEdsw, MW; };
// INDEXED (index_based_contact_term): g0 is the global [3][nsites][nsites] table and ty = site_i * nsites + site_j.
template <bool INDEXED = false, int PLAIN = 0, class real> __device__ __forceinline__ PairTerms<real> contact_pair_math(const ContactD<real>& d, const real* g0, int ty, real r, real sw, real aa) {
    // one environment is three planes (direct, protein, water) of 400 type pairs, or of nsites^2 site pairs (INDEXED); the membrane
    // environment of a z-dependent term lies behind the water one
    const int plane = INDEXED ? d.nsites * d.nsites : 400;
    const real* g1 = g0 + 3 * plane;
    const real one = (real)1;
    const real s1 = sigmoid(d.eta2 * (r - d.rmin)), s2 = sigmoid(d.eta2 * (d.rmax - r));
    const real s3 = sigmoid(d.eta2 * (r - d.rminII)), s4 = sigmoid(d.eta2 * (d.rmaxII - r));
    const real th1 = s1 * s2, dth1 = d.eta2 * th1 * (s2 - s1);
    const real th2 = s3 * s4, dth2 = d.eta2 * th2 * (s4 - s3);
    PairTerms<real> o;
    if (PLAIN == 1 || (PLAIN == 0 && !(d.flags & 1))) {   // PLAIN 1 / 2: the caller knows the term is not / is z-dependent (lean step kernels)
        const real gd = g0[ty], gp = g0[plane + ty], gw = g0[2 * plane + ty];
        const real med = sw * gw + (one - sw) * gp;
        o.E = -d.k * (gd * th1 + th2 * med);
        o.dEdr = -d.k * (gd * dth1 + dth2 * med);
        o.dEdsw = -d.k * th2 * (gw - gp);
        o.MW = 0;
    } else {
        const real med0 = sw * g0[2 * plane + ty] + (one - sw) * g0[plane + ty];
        const real med1 = sw * g1[2 * plane + ty] + (one - sw) * g1[plane + ty];
        const real W = -d.k * (g0[ty] * th1 + th2 * med0), Mm = -d.k * (g1[ty] * th1 + th2 * med1);
        const real dW = -d.k * (g0[ty] * dth1 + dth2 * med0), dM = -d.k * (g1[ty] * dth1 + dth2 * med1);
        o.E = (one - aa) * W + aa * Mm;
        o.dEdr = (one - aa) * dW + aa * dM;
        o.dEdsw = -d.k * th2 * ((one - aa) * (g0[2 * plane + ty] - g0[plane + ty]) + aa * (g1[2 * plane + ty] - g1[plane + ty]));
        o.MW = Mm - W;
    }
    return o;
}
template <bool ENERGY, class V> __device__ __forceinline__ void pairs_body(const ContactD<typename V::real_t>& d, const V& L, int i, int j, int ty, int a, int b, bool v, Energy& e) {
    typedef typename V::real_t real;
    const bool zdep = cflags(d, L) & 1;
    const real one = (real)1;
    const R3<real> dd = L.pos(a) - L.pos(b);
    const real r2 = dist2(dd);
    const bool pass = v && r2 < d.cutoff2;
    if (__ballot(pass) == 0) return;
    const real ir = M<real>::rsqrt(pass ? r2 : one), r = r2 * ir;
    const real si = L.rhov[i], sj = L.rhov[j];
    real ai = 0, aj = 0;
    if (zdep) { ai = L.alpha[i]; aj = L.alpha[j]; }
    const PairTerms<real> t = (cflags(d, L) & 16) ? contact_pair_math<true>(d, d.gamma, pass ? i * d.nsites + j : 0, r, si * sj, ai * aj)
                                             : contact_pair_math<false, V::lean ? 1 + (V::cflag & 1) : 0>(d, L.gamma, ty, r, si * sj, ai * aj);
    if (pass) {
        if (ENERGY) e += (double)t.E;
        L.pair_force(a, b, dd.x, dd.y, dd.z, -t.dEdr * ir);
        fixed_add(&L.drho[i], to_fixed(t.dEdsw * sj * (-d.eta_s2 * si * (one - si))));
        fixed_add(&L.drho[j], to_fixed(t.dEdsw * si * (-d.eta_s2 * sj * (one - sj))));
        if (zdep) {
            L.add_fz(a, -t.MW * aj * L.dalpha[i]);
            L.add_fz(b, -t.MW * ai * L.dalpha[j]);
        }
    }
}
template <bool ENERGY, class V> __device__ __forceinline__ void contact_pairs(const ContactD<typename V::real_t>& d, const V& L, const int* satom) {
    AWSEM_WAVE_VARS;
    Energy e;
    const ListView lv = list_view(L, d.ls);
    AWSEM_LIST_CHUNKS(c0, lv.count, d.ls.list_id + 1) {
        const int idx = c0 + lane;
        const bool v = idx < lv.count;
        const unsigned ent = v ? lv.e[idx] : 0u;
        const int i = ent & 0x3ff, j = (ent >> 10) & 0x3ff, ty = (ent >> 20) & 0x1ff;
        pairs_body<ENERGY>(d, L, i, j, ty, satom[i], satom[j], v && (ent >> 31), e);
    }
    flush_energy<ENERGY>(L, d.group, e);
}
// pass 3: chain rule through the densities: F_j += (dE/drho_i + dE/drho_j) theta_I'(r) (x_i - x_j)/r
template <class V> __device__ __forceinline__ void chain_body(const ContactD<typename V::real_t>& d, const V& L, int i, int j, int a, int b, bool v) {
    typedef typename V::real_t real;
    const R3<real> dd = L.pos(a) - L.pos(b);
    const real r2 = dist2(dd);
    const bool pass = v && r2 < d.cutoff2;
    if (__ballot(pass) == 0) return;
    const real ir = M<real>::rsqrt(pass ? r2 : (real)1), r = r2 * ir;
    const real s1 = sigmoid(d.eta2 * (r - d.rmin)), s2 = sigmoid(d.eta2 * (d.rmax - r));
    const real g = fixed_to_real(L.drho[i] + L.drho[j], (real)0);
    if (pass) L.pair_force(a, b, dd.x, dd.y, dd.z, -g * d.eta2 * s1 * s2 * (s2 - s1) * ir);
}
template <class V> __device__ __forceinline__ void contact_chain_rule(const ContactD<typename V::real_t>& d, const V& L, const int* satom) {
    AWSEM_WAVE_VARS;
    const ListView lv = list_view(L, d.ls);
    AWSEM_LIST_CHUNKS(c0, lv.count, d.ls.list_id + 2) {
        const int idx = c0 + lane;
        const bool v = idx < lv.count;
        const unsigned ent = v ? lv.e[idx] : 0u;
        const int i = ent & 0x3ff, j = (ent >> 10) & 0x3ff;
        chain_body(d, L, i, j, satom[i], satom[j], v && (ent & 0x40000000u));
    }
}

// ---- acceptor x donor terms (CustomHbondForce shape)
// In fp32 the lists use a conservative pre-screen (Gaussian factors below e^-30 of their maximum are skipped, far
// under fp32 resolution); the fp64 check mode applies only the reference's cutoff.
// Entry (donor residue i, acceptor residue j) of a lambda / gamma table in its functional form, WITHOUT the -k factor; the
// formulas and their order of operations are those of the reference's table builders (hydrogenBondTerms.py:160-359), so the
// fp64 instantiation reproduces the dense tables to the last bits.
template <class real> __host__ __device__ __forceinline__ real hb_fn_value(const HbFn<real>& f, int i, int j) {
    const int ri = f.rinfo[i], rj = f.rinfo[j];
    if (f.kind <= 3) {
        const int d = i > j ? i - j : j - i;
        if (d < 4) return (real)0;
        const int c = d >= 45 ? 2 : (d >= 18 ? 1 : 0);
        if (f.kind == 1) return f.cc[0][c];
        if ((ri | rj) & (1 << 15)) return (real)0;   // a chain edge: the table builders leave the row / column zero
        const int ai = ri & 31, aip = (ri >> 5) & 31, ain = (ri >> 10) & 31, aj = rj & 31, ajp = (rj >> 5) & 31, ajn = (rj >> 10) & 31;
        real Lm = f.cc[0][c];
        if (f.kind == 2) {
            Lm = Lm + f.cc[1][c] * f.t1[ai * 20 + aj];
            Lm = Lm + f.cc[2][c] * (f.t2[ain * 20 + ajp] + f.t2[aip * 20 + ajn]);
            Lm = Lm + f.cc[3][c] * (f.ob[ai] + f.ob[aj]);
        } else {
            Lm = Lm + f.cc[1][c] * f.t1[ain * 20 + aj];
            Lm = Lm + f.cc[2][c] * f.ob[ain];
            Lm = Lm + f.cc[1][c] * f.ob[aj];
        }
        return Lm;
    }
    const int d = i - j;   // donor - acceptor
    const bool other = (ri >> 17) != (rj >> 17);
    const real s = ((ri & rj) >> 16) & 1 ? (real)1.5 : (real)1;
    if (f.kind == 4) return ((d >= 13 && d <= 16 && !other) ? f.g[0] : (real)0) + ((d >= 17 || other) ? s * f.g[1] : (real)0);
    return (d >= 9 || other) ? s * f.g[2] : (real)0;
}

template <bool LEAN = false, class V, class Desc> __device__ __forceinline__ void build_hbond(const Desc& d, const V& L, int* status) {
    typedef typename V::real_t real;
    AWSEM_WAVE_VARS;
    HbEntry<real>* e = (HbEntry<real>*)(L.lists + d.ls.list_off);
    unsigned* cnt = &L.lcount[d.ls.list_id];
    const int resw = Desc::res_slot;   // which Int4 component carries the residue index
    // staged words: screening atom (a1 / d1) | residue << 16
    unsigned* sa = L.scratch();
    unsigned* sd = sa + d.nacc;
    for (int i = threadIdx.x; i < d.nacc; i += L.nthr) { const Int4 A = d.acc[i]; sa[i] = (unsigned)A.x | ((unsigned)(resw == 3 ? A.w : A.z) << 16); }
    for (int i = threadIdx.x; i < d.ndon; i += L.nthr) { const Int4 D = d.don[i]; sd[i] = (unsigned)D.x | ((unsigned)(resw == 3 ? D.w : D.z) << 16); }
    // which (donor residue, acceptor residue) pairs have a non-zero table entry: one bit each, staged as well when it fits
    const unsigned* nz = d.nzmask;
    if (d.nz_lds && (LEAN || !d.fn)) {
        unsigned* sm = sd + d.ndon;
        const int words = (d.nres * d.nres + 31) >> 5;
        for (int i = threadIdx.x; i < words; i += L.nthr) sm[i] = d.nzmask[i];
        nz = sm;
    }
    __syncthreads();
    // LEAN (host: every term's ndon <= 64 waves' worth, dense tables): two passes per chunk of acceptors.  The first tests distance, zero
    // band and the table's non-zero bit from LDS alone and keeps the survivors as bits of a 64-bit mask in a register; the second walks
    // the set bits -- a handful per lane -- and only there touches global memory (coefficient, atom records, the list).  The one-pass
    // loop below pays that round trip in nearly every iteration, because some lane of the 64 nearly always has a pair in range: 2xov's
    // five scans of 276 x 276 candidates were 23 % of its run time that way.
    if (LEAN || (!d.fn && d.ndon <= 64 * nw)) {   // (the full kernel: when the term qualifies, uniform)
        for (int a0 = 0; a0 < d.nacc; a0 += 64) {
            const int ia = a0 + lane;
            const bool va = ia < d.nacc;
            const unsigned A = sa[va ? ia : 0];
            const R3<real> pa = L.pos((int)(A & 0xffffu));
            const int ares = (int)(A >> 16);
            const int q0 = (va ? ia : 0) % d.ndon;
            u64 mm = 0;
            for (int dd = wave, k = 0; dd < d.ndon; dd += nw, k++) {
                const int t = q0 + dd, q = t >= d.ndon ? t - d.ndon : t;
                const unsigned D = sd[q];
                const int dres = (int)(D >> 16), dq = dres - ares;
                if (va && dist2(pa - L.pos((int)(D & 0xffffu))) < d.list_rc2 && (dq <= d.band_lo || dq >= d.band_hi)) {
                    const unsigned bit = (unsigned)(dres * d.nres + ares);
                    if ((nz[bit >> 5] >> (bit & 31)) & 1u) mm |= 1ull << k;
                }
            }
            while (__ballot(mm != 0) != 0) {
                const bool pass = mm != 0;
                const int k = pass ? __ffsll((long long)mm) - 1 : 0;
                mm &= mm - 1;
                const int t = q0 + wave + k * nw, q = t >= d.ndon ? t - d.ndon : t;
                HbEntry<real> ent;
                ent.w0 = ent.w1 = ent.w2 = 0u; ent.c = (real)0;
                if (pass) {
                    const int dres = (int)(sd[q] >> 16);
                    ent.c = d.table()[(size_t)dres * d.nres + ares];
                    const Int4 Aq = d.acc[ia], Dn = d.don[q];
                    ent.w0 = (unsigned)Aq.x | (d.entry_flag() << 15) | ((unsigned)Dn.x << 16);
                    if (resw == 3) { ent.w1 = ((unsigned)Aq.y & 0xffffu) | ((unsigned)Aq.z << 16); ent.w2 = ((unsigned)Dn.y & 0xffffu) | ((unsigned)Dn.z << 16); }
                    else { ent.w1 = (unsigned)Aq.y & 0xffffu; ent.w2 = (unsigned)Dn.y & 0xffffu; }
                }
                list_append_entry(e, cnt, d.ls.list_cap, status, pass, ent);
            }
        }
        __syncthreads();
        return;
    }
    for (int a0 = 0; a0 < d.nacc; a0 += 64) {
        const int ia = a0 + lane;
        const bool va = ia < d.nacc;
        const unsigned A = sa[va ? ia : 0];
        const R3<real> pa = L.pos((int)(A & 0xffffu));
        const int ares = (int)(A >> 16);
        const int q0 = (va ? ia : 0) % d.ndon;
        constexpr int U = AWSEM_BUILD_UNROLL;
        for (int dd0 = wave; dd0 < d.ndon; dd0 += nw * U) {
            int q[U];
            unsigned D[U];
            bool pass[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int dd = dd0 + u * nw;
                pass[u] = va && dd < d.ndon;
                int t = q0 + (dd < d.ndon ? dd : 0);
                q[u] = t >= d.ndon ? t - d.ndon : t;
                D[u] = sd[q[u]];
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int dres = (int)(D[u] >> 16), dq = dres - ares;
                const real r2 = dist2(pa - L.pos((int)(D[u] & 0xffffu)));
                pass[u] = pass[u] && r2 < d.list_rc2 && (dq <= d.band_lo || dq >= d.band_hi);
                HbEntry<real> ent;
                ent.w0 = ent.w1 = ent.w2 = 0u; ent.c = (real)0;
                if (pass[u]) {   // a surviving pair's record: the atoms of both sides and the coefficient, exactly as the pair bodies used to look them up
                    if (!LEAN && d.fn) { const real v = hb_fn_value(*d.fn, dres, ares); pass[u] = v != (real)0; ent.c = d.fn->scale * v; }   // uniform branch
                    else {
                        const unsigned bit = (unsigned)(dres * d.nres + ares);
                        pass[u] = (nz[bit >> 5] >> (bit & 31)) & 1u;
                        if (pass[u]) ent.c = d.table()[(size_t)dres * d.nres + ares];
                    }
                    if (pass[u]) {
                        const Int4 A = d.acc[ia], Dn = d.don[q[u]];
                        ent.w0 = (unsigned)A.x | (d.entry_flag() << 15) | ((unsigned)Dn.x << 16);
                        if (resw == 3) { ent.w1 = ((unsigned)A.y & 0xffffu) | ((unsigned)A.z << 16); ent.w2 = ((unsigned)Dn.y & 0xffffu) | ((unsigned)Dn.z << 16); }
                        else { ent.w1 = (unsigned)A.y & 0xffffu; ent.w2 = (unsigned)Dn.y & 0xffffu; }
                    }
                }
                list_append_entry(e, cnt, d.ls.list_cap, status, pass[u], ent);
            }
        }
    }
    __syncthreads();
}

// The terms that share a list (beta_term_1/2/3; pap_term_1/2) screen the same atom pair (a1, d1) against the same radius: ONE distance
// scan serves them all.  A lane keeps its acceptor; wave w tests the donors w, w + nw, ... and records the survivors as bits of a 64-bit
// mask in a register (so at most 64 nw donors); every member then walks the set bits only -- a handful per lane instead of every donor
// -- for its zero band, its table and its records.  (Three earlier forms of a shared scan, which staged the survivors in LDS or in the
// list, lost to the per-term scans: DESIGN.md, rejected (xvi).  This one adds nothing to the chain of a surviving pair.)
template <class V, class Desc> __device__ __forceinline__ void build_hbond_group(const Desc& d0, const Desc* __restrict__ terms, int lead, int nterms, const V& L, int* status) {
    typedef typename V::real_t real;
    AWSEM_WAVE_VARS;
    const int resw = Desc::res_slot, list_id = d0.ls.list_id;
    HbEntry<real>* e = (HbEntry<real>*)(L.lists + d0.ls.list_off);
    unsigned* cnt = &L.lcount[list_id];
    unsigned* sa = L.scratch();
    unsigned* sd = sa + d0.nacc;
    unsigned* sm = sd + d0.ndon;
    const int words = (d0.nres * d0.nres + 31) >> 5;
    for (int i = threadIdx.x; i < d0.nacc; i += L.nthr) { const Int4 A = d0.acc[i]; sa[i] = (unsigned)A.x | ((unsigned)(resw == 3 ? A.w : A.z) << 16); }
    for (int i = threadIdx.x; i < d0.ndon; i += L.nthr) { const Int4 D = d0.don[i]; sd[i] = (unsigned)D.x | ((unsigned)(resw == 3 ? D.w : D.z) << 16); }
    for (int u = lead, k = 0; u < nterms; u++) {
        if (terms[u].ls.list_id != list_id || !terms[u].scan_mode) continue;
        const unsigned* src = terms[u].nzmask;
        if (terms[u].nz_lds && !terms[u].fn) for (int i = threadIdx.x; i < words; i += L.nthr) sm[k * words + i] = src[i];
        k++;
    }
    __syncthreads();
    for (int a0 = 0; a0 < d0.nacc; a0 += 64) {
        const int ia = a0 + lane;
        const bool va = ia < d0.nacc;
        const unsigned A = sa[va ? ia : 0];
        const R3<real> pa = L.pos((int)(A & 0xffffu));
        const int ares = (int)(A >> 16);
        const int q0 = (va ? ia : 0) % d0.ndon;
        u64 mask = 0;
        for (int dd = wave, k = 0; dd < d0.ndon; dd += nw, k++) {
            const int t = q0 + dd, q = t >= d0.ndon ? t - d0.ndon : t;
            const unsigned D = sd[q];
            if (va && dist2(pa - L.pos((int)(D & 0xffffu))) < d0.list_rc2) mask |= 1ull << k;
        }
        for (int u = lead, member = 0; u < nterms; u++) {
            if (terms[u].ls.list_id != list_id || !terms[u].scan_mode) continue;
            const Desc d = terms[u];   // uniform: scalar loads from device memory
            const unsigned* nz = d.nz_lds && !d.fn ? sm + member * words : d.nzmask;
            member++;
            u64 mm = mask;
            while (__ballot(mm != 0) != 0) {
                bool pass = mm != 0;
                const int k = pass ? __ffsll((long long)mm) - 1 : 0;
                mm &= mm - 1;
                const int t = q0 + wave + k * nw, q = t >= d0.ndon ? t - d0.ndon : t;
                const unsigned D = sd[q];
                const int dres = (int)(D >> 16), dq = dres - ares;
                pass = pass && (dq <= d.band_lo || dq >= d.band_hi);
                HbEntry<real> ent;
                ent.w0 = ent.w1 = ent.w2 = 0u; ent.c = (real)0;
                if (pass) {
                    if (d.fn) { const real v = hb_fn_value(*d.fn, dres, ares); pass = v != (real)0; ent.c = d.fn->scale * v; }   // uniform branch
                    else {
                        const unsigned bit = (unsigned)(dres * d.nres + ares);
                        pass = (nz[bit >> 5] >> (bit & 31)) & 1u;
                        if (pass) ent.c = d.table()[(size_t)dres * d.nres + ares];
                    }
                    if (pass) {
                        const Int4 Aq = d.acc[ia], Dn = d.don[q];
                        ent.w0 = (unsigned)Aq.x | (d.entry_flag() << 15) | ((unsigned)Dn.x << 16);
                        if (resw == 3) { ent.w1 = ((unsigned)Aq.y & 0xffffu) | ((unsigned)Aq.z << 16); ent.w2 = ((unsigned)Dn.y & 0xffffu) | ((unsigned)Dn.z << 16); }
                        else { ent.w1 = (unsigned)Aq.y & 0xffffu; ent.w2 = (unsigned)Dn.y & 0xffffu; }
                    }
                }
                list_append_entry(e, cnt, d.ls.list_cap, status, pass, ent);
            }
        }
    }
    __syncthreads();
}

// one (acceptor, donor) pair of a beta term; shared by the pair-list walk below and the cell-list rows (awsem_cell.h)
// LISTED: the pair comes from an HbEntry record, which carries its coefficient and the two_theta flag of its term
template <bool ENERGY, bool LISTED = false, class V> __device__ __forceinline__ void beta_pair(const BetaD<typename V::real_t>& d, const V& L, const Int4& A, const Int4& D, Energy& e,
                                                                                           typename V::real_t c_listed = 0, bool two_theta_listed = false) {
    typedef typename V::real_t real;
    const R3<real> po = L.pos(A.x), a = po - L.pos(D.x);
    const real r1s = dist2(a);
    if (!(r1s < d.screen2)) return;
    const R3<real> b = po - L.pos(D.y);
    const real r2s = dist2(b);
    const real i1 = M<real>::rsqrt(r1s), i2 = M<real>::rsqrt(r2s);
    const real da = r1s * i1 - d.r_ON, db = r2s * i2 - d.r_OH;
    const real ex1 = (real)0.5 * (da * da * d.inv_sNO2 + db * db * d.inv_sHO2);
    if (ex1 > d.max_expo) return;
    const real c = LISTED ? c_listed : (d.fn ? d.fn->scale * hb_fn_value(*d.fn, D.w, A.w) : d.lambda[(size_t)D.w * d.nres + A.w]);   // -k * lambda[don_res][acc_res]
    const real th1 = M<real>::exp(-ex1), g1 = -th1 * da * d.inv_sNO2, g2 = -th1 * db * d.inv_sHO2;
    if (!(LISTED ? two_theta_listed : d.two_theta != 0)) {
        if (ENERGY) e += (double)(c * th1);
        L.pair_force(A.x, D.x, a.x, a.y, a.z, -c * g1 * i1);
        L.pair_force(A.x, D.y, b.x, b.y, b.z, -c * g2 * i2);
    } else {
        const R3<real> pp = L.pos(D.z), cc = pp - L.pos(A.y), ee = pp - L.pos(A.z);
        const real r3s = dist2(cc), r4s = dist2(ee);
        const real i3 = M<real>::rsqrt(r3s), i4 = M<real>::rsqrt(r4s);
        real h1, h2;
        const real th2 = hb_theta(r3s * i3, r4s * i4, d.r_ON, d.r_OH, d.inv_sNO2, d.inv_sHO2, h1, h2);
        if (ENERGY) e += (double)(c * th1 * th2);
        L.pair_force(A.x, D.x, a.x, a.y, a.z, -c * g1 * th2 * i1);
        L.pair_force(A.x, D.y, b.x, b.y, b.z, -c * g2 * th2 * i2);
        L.pair_force(D.z, A.y, cc.x, cc.y, cc.z, -c * th1 * h1 * i3);
        L.pair_force(D.z, A.z, ee.x, ee.y, ee.z, -c * th1 * h2 * i4);
    }
}
template <bool ENERGY, class V> __device__ __forceinline__ void eval_beta(const BetaD<typename V::real_t>& d, const V& L) {
    AWSEM_WAVE_VARS;
    Energy e;
    const ListView lv = list_view(L, d.ls);
    const HbEntry<typename V::real_t>* ent = (const HbEntry<typename V::real_t>*)lv.e;
    AWSEM_LIST_CHUNKS(c0, lv.count, d.ls.list_id) {
        const int idx = c0 + lane;
        if (idx >= lv.count) continue;
        const HbEntry<typename V::real_t> E = ent[idx];
        const Int4 A{(int)(E.w0 & 0x7fffu), (int)(E.w1 & 0xffffu), (int)(E.w1 >> 16), 0}, D{(int)((E.w0 >> 16) & 0x7fffu), (int)(E.w2 & 0xffffu), (int)(E.w2 >> 16), 0};
        beta_pair<ENERGY, true>(d, L, A, D, e, E.c, ((E.w0 >> 15) & 1u) != 0);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, bool LISTED = false, class V> __device__ __forceinline__ void pap_pair(const PapD<typename V::real_t>& d, const V& L, const Int4& A, const Int4& D, Energy& e,
                                                                                          typename V::real_t c_listed = 0) {
    typedef typename V::real_t real;
    const real one = (real)1;
    const R3<real> a1 = L.pos(A.x), a = a1 - L.pos(D.x);
    const real r1s = dist2(a);
    if (!(r1s < d.cutoff2)) return;
    const R3<real> a2 = L.pos(A.y), ee = a1 - a2;
    const real r3s = dist2(ee), i3 = M<real>::rsqrt(r3s);
    const real arg3 = d.eta2 * (r3s * i3 - d.d_i4);
    if (arg3 < d.min_arg) return;
    const R3<real> b = a2 - L.pos(D.y);
    const real r2s = dist2(b);
    const real i1 = M<real>::rsqrt(r1s), i2 = M<real>::rsqrt(r2s);
    const real arg2 = d.eta2 * (d.r0 - r2s * i2);
    if (arg2 < d.min_arg) return;
    const real c = LISTED ? c_listed : (d.fn ? d.fn->scale * hb_fn_value(*d.fn, D.z, A.z) : d.gamma[(size_t)D.z * d.nres + A.z]);   // -k * gamma[don_res][acc_res]
    const real s1 = sigmoid(d.eta2 * (d.r0 - r1s * i1)), s2 = sigmoid(arg2), s3 = sigmoid(arg3);
    if (ENERGY) e += (double)(c * s1 * s2 * s3);
    L.pair_force(A.x, D.x, a.x, a.y, a.z, c * d.eta2 * s1 * (one - s1) * s2 * s3 * i1);
    L.pair_force(A.y, D.y, b.x, b.y, b.z, c * s1 * d.eta2 * s2 * (one - s2) * s3 * i2);
    L.pair_force(A.x, A.y, ee.x, ee.y, ee.z, -c * s1 * s2 * d.eta2 * s3 * (one - s3) * i3);
}
template <bool ENERGY, class V> __device__ __forceinline__ void eval_pap(const PapD<typename V::real_t>& d, const V& L) {
    AWSEM_WAVE_VARS;
    Energy e;
    const ListView lv = list_view(L, d.ls);
    const HbEntry<typename V::real_t>* ent = (const HbEntry<typename V::real_t>*)lv.e;
    AWSEM_LIST_CHUNKS(c0, lv.count, d.ls.list_id) {
        const int idx = c0 + lane;
        if (idx >= lv.count) continue;
        const HbEntry<typename V::real_t> E = ent[idx];
        const Int4 A{(int)(E.w0 & 0x7fffu), (int)(E.w1 & 0xffffu), 0, 0}, D{(int)((E.w0 >> 16) & 0x7fffu), (int)(E.w2 & 0xffffu), 0, 0};
        pap_pair<ENERGY, true>(d, L, A, D, e, E.c);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// ------------------------------------------------------------------ virtual sites, whole-system evaluation
template <class V> __device__ __forceinline__ void compute_virtual_sites(const DevSys<typename V::real_t>& S, const V& L) {
    typedef typename V::real_t real;
    for (int i = threadIdx.x; i < S.nvs; i += L.nthr) {
        const Int4 v = S.vs[i];
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        const R3<real> a = L.pos(v.y), b = L.pos(v.z), c = L.pos(v.w);
        L.set_pos(v.x, R3<real>{w0 * a.x + w1 * b.x + w2 * c.x, w0 * a.y + w1 * b.y + w2 * c.y, w0 * a.z + w1 * b.z + w2 * c.z});
    }
}
// fold the force on each virtual site onto its three parents (ThreeParticleAverageSite chain rule); evaluation API
// only -- the step kernel gathers instead (every massive atom collects from the sites it parents, no atomics)
// step kernel: virtual sites from their (new) parents, their force accumulators -- already gathered by the integrator --
// cleared, as are those of massless particles that are no virtual sites
template <class V> __device__ __forceinline__ void update_virtual_sites(const DevSys<typename V::real_t>& S, const V& L) {
    typedef typename V::real_t real;
    for (int i = threadIdx.x; i < S.nvs; i += L.nthr) {
        const Int4 v = S.vs[i];
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        const R3<real> pa = L.pos(v.y), pb = L.pos(v.z), pc = L.pos(v.w);
        L.set_pos(v.x, R3<real>{w0 * pa.x + w1 * pb.x + w2 * pc.x, w0 * pa.y + w1 * pb.y + w2 * pc.y, w0 * pa.z + w1 * pb.z + w2 * pc.z});
        L.zero_force(v.x);
    }
    for (int i = threadIdx.x; i < S.nfixed; i += L.nthr) L.zero_force(S.fixed[i]);
}
template <class V> __device__ __forceinline__ void redistribute_forces(const DevSys<typename V::real_t>& S, const V& L) {
    typedef typename V::real_t real;
    for (int i = threadIdx.x; i < S.nvs; i += L.nthr) {
        const Int4 v = S.vs[i];
        const R3<real> f = L.force(v.x);
        L.zero_force(v.x);
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        L.add_force(v.y, w0 * f.x, w0 * f.y, w0 * f.z);
        L.add_force(v.z, w1 * f.x, w1 * f.y, w1 * f.z);
        L.add_force(v.w, w2 * f.x, w2 * f.y, w2 * f.z);
    }
}

// Once per launch: contact site -> atom tables and (when the system has a single contact term) its gamma tables.
template <class V> __device__ __forceinline__ void stage_contact_gamma(const ContactD<typename V::real_t>& d, const V& L) {
    typedef typename V::real_t real;
    if (cflags(d, L) & (2 | 16)) return;   // burial only: no pair tables; index based: read from global memory
    const int ng = (cflags(d, L) & 1) ? 2400 : 1200;
    const real* src = d.gamma + ((cflags(d, L) & 1) ? 0 : d.env * 1200);
    for (int i = threadIdx.x; i < ng; i += L.nthr) L.gamma[i] = src[i];
}
template <class V> __device__ __forceinline__ void stage_tables(const DevSys<typename V::real_t>& S, const V& L) {
    for (int t = 0; t < S.n_contact; t++)
        for (int i = threadIdx.x; i < S.contact[t].nsites; i += L.nthr) L.satom[t * S.max_contact_sites + i] = S.contact[t].atom[i];
    if (S.n_contact == 1) stage_contact_gamma(S.contact[0], L);
    if (S.ramam.n) for (int i = threadIdx.x; i < S.ramam.nset * S.ramam.nwell * 8; i += L.nthr) L.ramap[i] = S.ramam.params[i];
}

#define AWSEM_ON(d) ((mask >> (d).group) & 1u)
// (Re)build the pair list of every pair term in `mask` from the current positions.  The caller zeroes L.lcount
// before (and synchronises), and synchronises after.
template <bool LEAN = false, class V> __device__ __forceinline__ void build_lists(const DevSys<typename V::real_t>& S, const V& L, unsigned mask) {
    for (int t = 0; t < (LEAN ? (S.n_contact < 1 ? S.n_contact : 1) : S.n_contact); t++) if (AWSEM_ON(S.contact[t])) build_contact(S.contact[t], L, L.satom + t * S.max_contact_sites, S.status);
    for (int t = 0; t < (LEAN ? (S.n_excl < 1 ? S.n_excl : 1) : S.n_excl); t++) if (AWSEM_ON(S.excl[t])) build_excl<LEAN>(S.excl[t], L, S.status, S.natoms);
    for (int t = 0; t < (LEAN ? (S.n_beta < 3 ? S.n_beta : 3) : S.n_beta); t++) if (AWSEM_ON(S.beta[t])) {
        if (LEAN || !S.beta[t].scan_mode) build_hbond<LEAN>(S.beta[t], L, S.status);   // (LEAN: chosen only for systems whose terms scan for themselves, dense tables)
        else if (S.beta[t].list_owner) build_hbond_group(S.beta[t], S.beta_dev, t, S.n_beta, L, S.status);
    }
    for (int t = 0; t < (LEAN ? (S.n_pap < 2 ? S.n_pap : 2) : S.n_pap); t++) if (AWSEM_ON(S.pap[t])) {
        if (LEAN || !S.pap[t].scan_mode) build_hbond<LEAN>(S.pap[t], L, S.status);
        else if (S.pap[t].list_owner) build_hbond_group(S.pap[t], S.pap_dev, t, S.n_pap, L, S.status);
    }
    for (int i = threadIdx.x; i < S.natoms; i += L.nthr) L.zero_force(i);   // the builders staged their tables in the accumulators
}

#ifdef AWSEM_PROF
// -DAWSEM_PROF experiment builds: the rebuild split by builder -- contact into slot 1, excluded volume into slot 3, beta into slot 8
// (three slots no phase of the step kernel uses), P-AP and the clearing of the staging area stay in slot 2 with the caller's stamp
template <bool LEAN = false, class V> __device__ __forceinline__ void build_lists_prof(const DevSys<typename V::real_t>& S, const V& L, unsigned mask, AwsemProf& awsem_prof) {
    for (int t = 0; t < S.n_contact; t++) if (AWSEM_ON(S.contact[t])) build_contact(S.contact[t], L, L.satom + t * S.max_contact_sites, S.status);
    AWSEM_ARRIVE(1);
    for (int t = 0; t < S.n_excl; t++) if (AWSEM_ON(S.excl[t])) build_excl<LEAN>(S.excl[t], L, S.status, S.natoms);
    AWSEM_ARRIVE(3);
    for (int t = 0; t < S.n_beta; t++) if (AWSEM_ON(S.beta[t])) {
        if (LEAN || !S.beta[t].scan_mode) build_hbond<LEAN>(S.beta[t], L, S.status);
        else if (S.beta[t].list_owner) build_hbond_group(S.beta[t], S.beta_dev, t, S.n_beta, L, S.status);
    }
    AWSEM_ARRIVE(8);
    for (int t = 0; t < S.n_pap; t++) if (AWSEM_ON(S.pap[t])) {
        if (LEAN || !S.pap[t].scan_mode) build_hbond<LEAN>(S.pap[t], L, S.status);
        else if (S.pap[t].list_owner) build_hbond_group(S.pap[t], S.pap_dev, t, S.n_pap, L, S.status);
    }
    for (int i = threadIdx.x; i < S.natoms; i += L.nthr) L.zero_force(i);
}
#endif
// Evaluate every term whose group is in `mask` from the current pair lists.  Forces accumulate into the view and the
// density of the first contact term into L.rho: both zeroed by the caller and visible (call after a __syncthreads());
// ends with all contributions visible to the workgroup.
// Everything that takes the descriptor block by reference is force-inlined: an out-of-line call would need its
// address, and the compiler then copies the whole by-value kernarg block to scratch (2.7 KB per lane, half the speed).
// The contact passes need workgroup barriers between them, which cuts the evaluation into four phases; every other
// term is issued in one of the phases so that each phase keeps all waves busy.
// STEP (the Langevin kernel): the Ramachandran and chirality terms move from the first phase to the second, and with
// `fused_vs` the first phase also places the virtual sites -- its other work (density, fragment memory) reads massive
// particles only (DevSys::fuse_vs), so the sites need no phase and no barrier of their own.
// LEAN: the step kernel's second instantiation for systems that hold none of the rarer term shapes -- Gaussian bonds and collective
// variables, helical, membrane / external fields, a second contact term, unbundled fragment memory, unmerged Ramachandran lists (the
// default forces_setup.py list qualifies; Engine::finalize checks).  The code for them is not compiled in, which the register
// allocator of the one big step kernel feels.  LEAN == 2: the same plus one helical term (forces_setup.py with the helical term
// switched on, as the many-memory configurations run it).  LEAN == 3: lean plus one membrane term, the contact term being the
// z-dependent one (the membrane-protein set-up, configs[3]).
template <bool ENERGY, bool STEP = false, int LEAN = 0, class V> __device__ __forceinline__ void evaluate_terms(const DevSys<typename V::real_t>& S, const V& L, unsigned mask, bool fused_vs AWSEM_STAMP_ARG) {
    typedef typename V::real_t real;
    int first = -1;   // the contact term whose passes define the phases
    for (int t = S.n_contact - 1; t >= 0; t--) if (AWSEM_ON(S.contact[t])) first = t;
    const ContactD<real>* c0 = first >= 0 ? &S.contact[first] : nullptr;
    const int* satom0 = L.satom + (first >= 0 ? first : 0) * S.max_contact_sites;
    const bool multi = !LEAN && S.n_contact > 1;
    const bool bar = c0 || (!LEAN && S.n_cv > 0);   // the phases need barriers: contact passes, or a collective variable's sum before its forces
    if (c0 && multi) { stage_contact_gamma(*c0, L); __syncthreads(); }
    Rotor rotor(L.nthr);
    // phase A
    if (STEP && fused_vs) update_virtual_sites(S, L);
    if (c0) contact_density(*c0, L, satom0);
    for (int u = 0; u < (LEAN ? (S.n_fm < 1 ? S.n_fm : 1) : S.n_fm); u++) if (AWSEM_ON(S.fm[u])) {
        if (LEAN || S.fm[u].nbundles) term_fm_bundled<ENERGY>(S.fm[u], L, rotor.next(S.fm[u].nbundles));
        else term_fm<ENERGY>(S.fm[u], L, rotor.next(S.fm[u].n));
    }
    if (!STEP) {
        if (LEAN || S.ramam.n) { if (AWSEM_ON(S.ramam)) term_rama_merged<ENERGY>(S.ramam, L, rotor.next(S.ramam.n)); }
        else for (int u = 0; u < S.n_rama; u++) if (AWSEM_ON(S.rama[u])) term_rama<ENERGY>(S.rama[u], L, rotor.next(S.rama[u].n));
        for (int u = 0; u < (LEAN ? (S.n_chi < 1 ? S.n_chi : 1) : S.n_chi); u++) if (AWSEM_ON(S.chi[u])) term_chi<ENERGY>(S.chi[u], L, rotor.next(S.chi[u].n));
    }
    if (!LEAN) for (int u = 0; u < S.n_gauss; u++) { const GaussD<real> d = S.gauss[u]; if (d.cv && AWSEM_ON(d)) cv_sum(d, L, rotor.next(d.n)); }
    AWSEM_ARRIVE(4);
    if (bar || STEP) __syncthreads();
    AWSEM_STAMP(S, 4);
    // phase B
    rotor.acc = 0;
    if (c0) contact_single<ENERGY>(*c0, L, satom0, rotor.next(c0->nsites));
    for (int u = 0; u < (LEAN ? (S.n_harm < 1 ? S.n_harm : 1) : S.n_harm); u++) if (AWSEM_ON(S.harm[u])) term_harmonic<ENERGY>(S.harm[u], L, rotor.next(S.harm[u].n));
    for (int u = 0; u < (LEAN ? (S.n_dh < 1 ? S.n_dh : 1) : S.n_dh); u++) if (AWSEM_ON(S.dh[u])) term_dh<ENERGY>(S.dh[u], L, rotor.next(S.dh[u].n));
    if (!LEAN) for (int u = 0; u < S.n_gauss; u++) {
        const GaussD<real> d = S.gauss[u];   // uniform: scalar loads
        if (!AWSEM_ON(d)) continue;
        if (d.cv) cv_apply<ENERGY>(d, L, rotor.next(d.n));
        else term_gauss<ENERGY>(d, L, rotor.next(d.n));
    }
    if (LEAN != 1) for (int u = 0; u < (LEAN ? (S.n_helical < 1 ? S.n_helical : 1) : S.n_helical); u++) if (AWSEM_ON(S.helical[u])) term_helical<ENERGY>(S.helical[u], L, rotor.next(S.helical[u].n));
    if (!LEAN || LEAN == 3) for (int u = 0; u < (LEAN ? (S.n_memb < 1 ? S.n_memb : 1) : S.n_memb); u++) if (AWSEM_ON(S.memb[u])) term_memb<ENERGY>(S.memb[u], L, rotor.next(S.memb[u].n));
    if (STEP) {
        if (LEAN || S.ramam.n) { if (AWSEM_ON(S.ramam)) term_rama_merged<ENERGY>(S.ramam, L, rotor.next(S.ramam.n)); }
        else for (int u = 0; u < S.n_rama; u++) if (AWSEM_ON(S.rama[u])) term_rama<ENERGY>(S.rama[u], L, rotor.next(S.rama[u].n));
        for (int u = 0; u < (LEAN ? (S.n_chi < 1 ? S.n_chi : 1) : S.n_chi); u++) if (AWSEM_ON(S.chi[u])) term_chi<ENERGY>(S.chi[u], L, rotor.next(S.chi[u].n));
    }
    AWSEM_ARRIVE(5);
    if (bar) __syncthreads();
    AWSEM_STAMP(S, 5);
    // phase C
    if (!LEAN && threadIdx.x < S.n_cv) L.cv[threadIdx.x] = 0;   // read for the last time in phase B; the next evaluation starts from zero
    if (c0 && !(cflags(*c0, L) & 2)) contact_pairs<ENERGY>(*c0, L, satom0);
    for (int u = 0; u < (LEAN ? (S.n_beta < 1 ? S.n_beta : 1) : S.n_beta); u++) if (AWSEM_ON(S.beta[u]) && S.beta[u].list_owner) eval_beta<ENERGY>(S.beta[u], L);   // one walk per shared list
    AWSEM_ARRIVE(6);
    if (bar) __syncthreads();
    AWSEM_STAMP(S, 6);
    // phase D
    if (c0) contact_chain_rule(*c0, L, satom0);
    for (int t = 0; t < (LEAN ? (S.n_excl < 1 ? S.n_excl : 1) : S.n_excl); t++) if (AWSEM_ON(S.excl[t])) eval_excl<ENERGY>(S.excl[t], L);
    for (int t = 0; t < (LEAN ? (S.n_pap < 1 ? S.n_pap : 1) : S.n_pap); t++) if (AWSEM_ON(S.pap[t]) && S.pap[t].list_owner) eval_pap<ENERGY>(S.pap[t], L);
    // further contact terms (e.g. a standalone burial term next to the contact term) reuse the density scratch
    if (!LEAN) for (int t = first + 1; first >= 0 && t < S.n_contact; t++) {
        const ContactD<real>& d = S.contact[t];
        if (!AWSEM_ON(d)) continue;   // uniform: S and mask are
        const int* satom = L.satom + t * S.max_contact_sites;
        __syncthreads();
        stage_contact_gamma(d, L);
        for (int i = threadIdx.x; i < d.nsites; i += L.nthr) L.rho[i] = 0;
        __syncthreads();
        contact_density(d, L, satom);
        __syncthreads();
        contact_single<ENERGY>(d, L, satom);
        __syncthreads();
        if (!(d.flags & 2)) contact_pairs<ENERGY>(d, L, satom);
        __syncthreads();
        contact_chain_rule(d, L, satom);
    }
    AWSEM_ARRIVE(7);
    __syncthreads();
    AWSEM_STAMP(S, 7);
}

// ------------------------------------------------------------------ LDS carve-up
template <class real> struct LdsLayout;
template <> struct LdsLayout<float> { static constexpr int pos = 16, force = 16, vel = 16, ref = 16; };      // float4, {xy plane, z plane, pad}, float4, float4
template <> struct LdsLayout<double> { static constexpr int pos = 24, force = 24, vel = 24, ref = 24; };     // three 8-byte planes each
// what sizes the LDS image of a replica: velocities exist for the massive particles only (indexed by their position in
// DevSys::massive), the gamma staging holds one environment (1200) or both (2400, z-dependent contact)
struct LdsShape { int natoms, nsites, ncontact, nmassive, ngamma, ncv, nramap; };
template <class real> __host__ __device__ inline LdsShape lds_shape(const DevSys<real>& S) {
    return LdsShape{S.natoms, S.max_contact_sites, S.n_contact, S.nmassive, S.any_zdep ? 2400 : 1200, S.n_cv, S.ramam.n ? S.ramam.nset * S.ramam.nwell * 8 : 0};
}
template <class real> __host__ __device__ inline size_t lds_bytes(const LdsShape& sh, bool with_vel) {
    const int natoms = sh.natoms, nsites = sh.nsites, ncontact = sh.ncontact;
    typedef LdsLayout<real> Y;
    size_t b = 0;
    b += (size_t)natoms * (Y::pos + Y::force + (with_vel ? Y::ref : 0)) + (with_vel ? (size_t)sh.nmassive * Y::vel : 0);
    b += (size_t)2 * nsites * sizeof(u64);            // rho, drho
    b += (AWSEM_NGROUPS + (size_t)sh.ncv + 6 + 8) * sizeof(u64);   // group energies, collective-variable sums, momentum sums and the mean velocities they stand for
    b += (size_t)3 * nsites * sizeof(real);           // rhov, alpha, dalpha
    b += (size_t)(sh.ngamma + sh.nramap) * sizeof(real);   // gamma staging, Ramachandran parameter sets
    b += (size_t)(ncontact > 0 ? ncontact : 1) * nsites * sizeof(int);   // contact site -> atom
    b += (AWSEM_MAX_LISTS + 4) * sizeof(unsigned);    // list counters, flags
    return (b + 15) & ~(size_t)15;
}
__device__ __forceinline__ int pinned_block_size() {
    int n = __builtin_amdgcn_readfirstlane((int)blockDim.x);
    asm volatile("" : "+s"(n));   // opaque from here on: kept in (or spilled from) a scalar register, never reloaded
    return n;
}
template <class real> __device__ inline void carve_common(ViewCommon<real>& L, unsigned char* q, const LdsShape& sh) {
    L.nthr = pinned_block_size();
    const int nsites = sh.nsites, ncontact = sh.ncontact;
    u64* w = (u64*)q;
    L.rho = w; w += nsites; L.drho = w; w += nsites;
    L.eg = w; w += AWSEM_NGROUPS;
    L.cv = w; w += sh.ncv;
    if ((int)threadIdx.x < sh.ncv) L.cv[threadIdx.x] = 0;   // every kernel has a barrier between the carve-up and its first evaluation
    L.cm = w; w += 6 + 8;
    if (threadIdx.x < 6) L.cm[threadIdx.x] = 0;
    real* r = (real*)w;
    L.rhov = r; r += nsites; L.alpha = r; r += nsites; L.dalpha = r; r += nsites;
    L.gamma = r; r += sh.ngamma;
    L.ramap = r; r += sh.nramap;
    L.satom = (int*)r;
    L.lcount = (unsigned*)(L.satom + (ncontact > 0 ? ncontact : 1) * nsites);
    L.flag = L.lcount + AWSEM_MAX_LISTS;
    L.lists = nullptr;
}
__device__ inline Lds<float> carve_lds(float, unsigned char* base, const LdsShape& sh, bool with_vel) {
    Lds<float> L;
    const int natoms = sh.natoms;
    unsigned char* q = base;
    L.p = (float4*)q; q += (size_t)16 * natoms;
    L.fxy = (long long*)q; L.fz = (int*)(q + (size_t)8 * natoms); q += (size_t)16 * natoms;
    L.v = nullptr; L.ref = nullptr;
    if (with_vel) { L.v = (float4*)q; q += (size_t)16 * sh.nmassive; L.ref = (float4*)q; q += (size_t)16 * natoms; }
    carve_common<float>(L, q, sh);
    return L;
}
__device__ inline Lds<double> carve_lds(double, unsigned char* base, const LdsShape& sh, bool with_vel) {
    Lds<double> L;
    const int natoms = sh.natoms, nm = sh.nmassive;
    u64* w = (u64*)base;
    L.fx = w; w += natoms; L.fy = w; w += natoms; L.fz = w; w += natoms;
    double* r = (double*)w;
    L.px = r; r += natoms; L.py = r; r += natoms; L.pz = r; r += natoms;
    L.vx = L.vy = L.vz = L.rx = L.ry = L.rz = nullptr;
    if (with_vel) {
        L.vx = r; r += nm; L.vy = r; r += nm; L.vz = r; r += nm;
        L.rx = r; r += natoms; L.ry = r; r += natoms; L.rz = r; r += natoms;
    }
    carve_common<double>(L, (unsigned char*)r, sh);
    return L;
}

// replica state in global memory: SoA per replica, [replica][xyz][natoms]
template <class V> __device__ inline void load_positions(const typename V::real_t* g, const V& L, int natoms) {
    typedef typename V::real_t real;
    for (int i = threadIdx.x; i < natoms; i += L.nthr) { L.set_pos(i, R3<real>{g[i], g[natoms + i], g[2 * natoms + i]}); L.zero_force(i); }
}

// largest squared displacement of the workgroup's particles since the last list build, into an LDS word
__device__ __forceinline__ void book_displacement(unsigned* flag, float m) {
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off));
    if ((threadIdx.x & 63) == 0) atomicMax(flag, __float_as_uint(m));   // non-negative floats order like their bits
}

// ------------------------------------------------------------------ evaluation kernel (energies and/or forces)
// The fp64 check mode compiles with a register budget of its own: at most 512 threads per workgroup (two waves per SIMD -> 256 VGPRs); under
// the fp32 kernels' 1024-thread cap (128 VGPRs) its step kernel spilled 176 VGPRs.  Engine::finalize keeps fp64 workgroups within the bound.
#ifndef AWSEM_FP64_THREADS
#define AWSEM_FP64_THREADS 512
#endif
#define AWSEM_BOUNDS(real) __launch_bounds__(sizeof(real) == 8 ? AWSEM_FP64_THREADS : 1024)
// LEAN: as for the step kernel (evaluate_terms): the instantiation without the code of the term shapes the system does not hold.
template <bool ENERGY, class real, int LEAN = 0>
__global__ AWSEM_BOUNDS(real) void awsem_eval_kernel(const DevSys<real> S, const real* __restrict__ pos, unsigned mask, int redistribute,
                                  double* __restrict__ e_out /* [R][32] or null */, double* __restrict__ f_out /* [R][natoms][3] or null */,
                                  const double* __restrict__ xyz_aos /* null, or [R][natoms][3] in the caller's atom order instead of `pos` */,
                                  int reuse /* S.lists are the lists the last step launch left for these very positions: walk them instead of building */,
                                  int rep0 /* first replica of the launch (a grid of one workgroup evaluates one replica of the ensemble) */,
                                  double* __restrict__ e_sum /* null, or [R]: the sum over the masked groups (what awsem_group_sum_kernel makes of e_out) */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int natoms = S.natoms, rep = (int)blockIdx.x + rep0;
    typedef typename std::conditional<LEAN == 3, LdsLeanZ<real>, typename std::conditional<LEAN != 0, LdsLean<real>, Lds<real>>::type>::type View;
    View L(carve_lds((real)0, smem, lds_shape(S), false));
    L.lists = S.lists + (size_t)rep * S.list_stride;
    if (xyz_aos) {   // uniform; awsem_energy_by_group_noalloc: the caller's device buffer, converted like the host does ((real) of the double)
        const double* x = xyz_aos + (size_t)rep * 3 * natoms;
        for (int i = threadIdx.x; i < natoms; i += L.nthr) {
            const int o = S.orig ? S.orig[i] : i;
            L.set_pos(i, R3<real>{(real)x[3 * o], (real)x[3 * o + 1], (real)x[3 * o + 2]});
            L.zero_force(i);
        }
    } else load_positions(pos + (size_t)rep * 3 * natoms, L, natoms);
    for (int i = threadIdx.x; i < S.max_contact_sites; i += L.nthr) L.rho[i] = 0;
    if (threadIdx.x < AWSEM_NGROUPS) L.eg[threadIdx.x] = 0;
    if (threadIdx.x < AWSEM_MAX_LISTS) L.lcount[threadIdx.x] = reuse ? S.list_counts[(size_t)rep * AWSEM_MAX_LISTS + threadIdx.x] : 0u;
    if (threadIdx.x == 0) L.flag[0] = 0;
    stage_tables(S, L);
    __syncthreads();
    // A query right after a step launch (a reporter's energies, the energies of a replica-exchange sweep) finds that launch's lists in
    // device memory, built for positions at most skin/2 away unless the launch's last step went further: the evaluators apply the exact
    // cutoffs themselves, so the lists serve as they are.  A replica that HAS gone further rebuilds them here, in place, and leaves
    // them (lengths, reference positions) as the next step launch expects them -- the rebuild that launch would have started with.
    bool build = true;
    if (reuse) {   // uniform
        real* gr = S.list_ref + (size_t)rep * 3 * S.nmassive;
        float m = 0;
        for (int k = threadIdx.x; k < S.nmassive; k += L.nthr)
            m = fmaxf(m, (float)dist2(L.pos(S.massive[k].atom) - R3<real>{gr[3 * k], gr[3 * k + 1], gr[3 * k + 2]}));
        book_displacement(&L.flag[0], m);
        __syncthreads();
        build = __uint_as_float(L.flag[0]) > (float)S.skin_half2;
        if (build) {
            __syncthreads();   // everyone has read the flag and the old reference positions
            if (threadIdx.x < AWSEM_MAX_LISTS) L.lcount[threadIdx.x] = 0;
            for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {
                const R3<real> x = L.pos(S.massive[k].atom);
                gr[3 * k] = x.x; gr[3 * k + 1] = x.y; gr[3 * k + 2] = x.z;
            }
            __syncthreads();
        }
    }
    if (build) {
        build_lists<LEAN != 0>(S, L, reuse ? 0xffffffffu : mask);
        __syncthreads();
        if (reuse && threadIdx.x < AWSEM_MAX_LISTS) S.list_counts[(size_t)rep * AWSEM_MAX_LISTS + threadIdx.x] = L.lcount[threadIdx.x];
    }
    AWSEM_STAMP_INIT;
    evaluate_terms<ENERGY, false, LEAN>(S, L, mask, false AWSEM_STAMP_PASS);
    if (f_out) {
        if (redistribute) { redistribute_forces(S, L); __syncthreads(); }
        double* f = f_out + (size_t)rep * 3 * natoms;
        for (int i = threadIdx.x; i < natoms; i += L.nthr) {
            const R3<real> F = L.force(i);
            const int o = S.orig ? S.orig[i] : i;
            f[3 * o] = (double)F.x; f[3 * o + 1] = (double)F.y; f[3 * o + 2] = (double)F.z;
        }
    }
    if (ENERGY && e_out && threadIdx.x < AWSEM_NGROUPS) e_out[(size_t)rep * AWSEM_NGROUPS + threadIdx.x] = fixed_to_double(L.eg[threadIdx.x]);
    if (ENERGY && e_sum && threadIdx.x == 0) {   // groups in ascending order, as awsem_group_sum_kernel and the host sum them: the same double
        double s = 0;
        for (int g = 0; g < AWSEM_NGROUPS; g++) if ((mask >> g) & 1u) s += fixed_to_double(L.eg[g]);
        e_sum[rep] = s;
    }
}

// Pair-list dump for awsem_neighbors: builds the lists of the terms in `mask` and writes the per-list counts.  LEAN: with the builders
// of the lean step / evaluation kernels, for the handles that run those (the list-parity tests then pin the builders that are used).
template <class real, bool LEAN = false>
__global__ AWSEM_BOUNDS(real) void awsem_list_kernel(const DevSys<real> S, const real* __restrict__ pos, unsigned mask, unsigned* __restrict__ counts /* [R][AWSEM_MAX_LISTS] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int natoms = S.natoms, rep = blockIdx.x;
    Lds<real> L = carve_lds((real)0, smem, lds_shape(S), false);
    L.lists = S.lists + (size_t)rep * S.list_stride;
    load_positions(pos + (size_t)rep * 3 * natoms, L, natoms);
    if (threadIdx.x < AWSEM_MAX_LISTS) L.lcount[threadIdx.x] = 0;
    stage_tables(S, L);
    __syncthreads();
    build_lists<LEAN>(S, L, mask);
    __syncthreads();
    if (threadIdx.x < AWSEM_MAX_LISTS) counts[(size_t)rep * AWSEM_MAX_LISTS + threadIdx.x] = L.lcount[threadIdx.x];
}

// ------------------------------------------------------------------ Philox4x32-10 Gaussian stream (shared with the oracle)
__device__ __forceinline__ void philox4x32_10(unsigned c[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        unsigned h0 = __umulhi(0xD2511F53u, c[0]), l0 = 0xD2511F53u * c[0];
        unsigned h1 = __umulhi(0xCD9E8D57u, c[2]), l1 = 0xCD9E8D57u * c[2];
        unsigned n0 = h1 ^ c[1] ^ k0, n2 = h0 ^ c[3] ^ k1;
        c[0] = n0; c[1] = l1; c[2] = n2; c[3] = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
template <class real> __device__ __forceinline__ void gauss3(unsigned long long seed, unsigned stream, int rep, long long step, int atom, real g[3]) {
    unsigned c[4] = {(unsigned)atom, (unsigned)(unsigned long long)step, (unsigned)((unsigned long long)step >> 32), (unsigned)rep};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32) ^ stream);
    const real inv = (real)(1.0 / 16777216.0);
    real u0 = ((real)(c[0] >> 8) + (real)0.5) * inv, u1 = ((real)(c[1] >> 8) + (real)0.5) * inv;
    real u2 = ((real)(c[2] >> 8) + (real)0.5) * inv, u3 = ((real)(c[3] >> 8) + (real)0.5) * inv;
    real r0 = M<real>::sqrt((real)-2 * M<real>::log(u0)), r1 = M<real>::sqrt((real)-2 * M<real>::log(u2));
    real s, co;
    M<real>::sincos2pi(u1, &s, &co);
    g[0] = r0 * co; g[1] = r0 * s;
    M<real>::sincos2pi(u3, &s, &co);
    g[2] = r1 * co;
}

#define AWSEM_STREAM_VEL 0x56454c30u
#define AWSEM_STREAM_LANGEVIN 0x4c414e47u
#define AWSEM_KB 0.00831446261815324

template <class real>
__global__ void awsem_maxwell_kernel(const DevSys<real> S, real* __restrict__ vel, const double* __restrict__ T,
                                     unsigned long long seed) {
    const int natoms = S.natoms, rep = blockIdx.x;
    real* v = vel + (size_t)rep * 3 * natoms;
    for (int i = threadIdx.x; i < natoms; i += blockDim.x) {
        real g[3] = {0, 0, 0}, s = 0;
        if (S.inv_mass[i] != (real)0) {
            gauss3<real>(seed, AWSEM_STREAM_VEL, rep, 0, S.orig ? S.orig[i] : i, g);
            s = (real)::sqrt(AWSEM_KB * T[rep]) * S.sqrt_inv_mass[i];
        }
        v[i] = s * g[0]; v[natoms + i] = s * g[1]; v[2 * natoms + i] = s * g[2];
    }
}

// the pending factors of a replica-exchange sweep applied on their own (a consumer other than the step kernel asks for the velocities)
template <class real>
__global__ void awsem_take_scale_kernel(real* __restrict__ vel, double* __restrict__ scale, int n3) {
    const real s = (real)scale[blockIdx.x];
    real* v = vel + (size_t)blockIdx.x * n3;
    for (int i = threadIdx.x; i < n3; i += blockDim.x) v[i] *= s;
    __syncthreads();
    if (threadIdx.x == 0) scale[blockIdx.x] = 1.0;
}
template <class real>
__global__ void awsem_scale_velocities_kernel(real* __restrict__ vel, const double* __restrict__ scale, int n3) {
    real* v = vel + (size_t)blockIdx.x * n3;
    const real s = (real)scale[blockIdx.x];
    for (int i = threadIdx.x; i < n3; i += blockDim.x) v[i] *= s;
}

// ------------------------------------------------------------------ temperature replica exchange on the device
// potential energy of every local replica for the masked groups, written where the caller wants it (awsem_remd_energies)
__global__ void awsem_group_sum_kernel(const double* __restrict__ e_by_group /* [R][32] */, unsigned mask, double* __restrict__ out, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    double s = 0;
    for (int g = 0; g < AWSEM_NGROUPS; g++) if ((mask >> g) & 1u) s += e_by_group[(size_t)r * AWSEM_NGROUPS + g];
    out[r] = s;
}
// U(0,1) with 53 random bits from the counter (pair, iteration) and the key `seed`: openawsem_amd/ensemble.py: counter_uniform
// computes the same number on the host, so every rank -- and the host-side sweep the CPU tests run -- takes the same decisions
#define AWSEM_STREAM_REMD 0x52454d44u
__device__ __forceinline__ double remd_uniform(unsigned long long seed, long long iteration, int pair) {
    unsigned c[4] = {(unsigned)pair, (unsigned)(unsigned long long)iteration, (unsigned)((unsigned long long)iteration >> 32), AWSEM_STREAM_REMD};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    return (double)((((unsigned long long)c[0] << 32) | c[1]) >> 11) * (1.0 / 9007199254740992.0);
}
// One sweep of neighbour swaps (scripts/mm_remd.py:133-149 hands this to openmmtools' ReplicaExchangeSampler): pairs (i, i+1) of
// the ladder with i of the iteration's parity; Metropolis on (beta_i - beta_{i+1})(U_a - U_b); accepted pairs swap their
// LADDER INDEX (slot[replica]), coordinates never move.  One workgroup; every rank runs it on the same gathered energies and
// arrives at the same slots.  Then, for the local replicas: new temperature and velocity factor sqrt(T_new / T_old).
__global__ void awsem_remd_sweep_kernel(const double* __restrict__ U /* gathered energies */, const int* __restrict__ where /* [N] replica -> index in U */,
                                        const double* __restrict__ ladder /* [N] K */, int* __restrict__ slot /* [N] replica -> ladder index */,
                                        unsigned long long* __restrict__ attempted, unsigned long long* __restrict__ accepted /* [N-1] */, int N,
                                        long long iteration, unsigned long long seed, int local_start, int nlocal,
                                        double* __restrict__ T_local /* [nlocal] */, double* __restrict__ scale_local /* [nlocal] */) {
    extern __shared__ int sh[];   // holder[N] (ladder index -> replica), old_slot[N]
    int* holder = sh;
    int* old = sh + N;
    for (int r = threadIdx.x; r < N; r += blockDim.x) { const int s = slot[r]; old[r] = s; holder[s] = r; }
    __syncthreads();
    const int parity = (int)(iteration & 1);
    for (int i = parity + 2 * (int)threadIdx.x; i < N - 1; i += 2 * blockDim.x) {   // disjoint pairs: one thread each
        const int a = holder[i], b = holder[i + 1];   // a runs at T_i, b at T_{i+1}
        const double delta = (1.0 / (AWSEM_KB * ladder[i]) - 1.0 / (AWSEM_KB * ladder[i + 1])) * (U[where[a]] - U[where[b]]);
        attempted[i] += 1;
        if (delta >= 0.0 || remd_uniform(seed, iteration, i) < ::exp(delta)) { slot[a] = i + 1; slot[b] = i; accepted[i] += 1; }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nlocal; k += blockDim.x) {
        const int r = local_start + k;
        const double Tn = ladder[slot[r]], To = ladder[old[r]];
        T_local[k] = Tn;
        scale_local[k] *= ::sqrt(Tn / To);   // (1 unless an earlier sweep's factor has not been taken in yet: Engine::vel)
    }
}

// openmmtools' DEFAULT mixing scheme, 'swap-all' (ReplicaExchangeSampler(replica_mixing_scheme='swap-all'), which scripts/mm_remd.py:133-140
// inherits by passing none): N^3 attempts per iteration, each on a uniformly drawn pair of REPLICAS (i, j) -- not of neighbouring
// rungs --, accepted when log_p = (beta_si - beta_sj)(U_i - U_j) >= 0 or u < exp(log_p); an accepted pair swaps its ladder indices.
// The attempts form a chain (each sees the swaps before it), so they are resolved by ONE thread; the other 255 draw the chain's
// random numbers a batch ahead: attempt a of this iteration uses Philox4x32-10(counter (a, iteration, 'RMSA'), key seed): word 0 ->
// i = (w0 * N) >> 32, word 1 -> j likewise, words 2,3 -> u with 53 bits (ensemble.py:_swap_all_draws is the same stream on the
// host).  The test is ln u < log_p (no exp in the chain).  Counters: attempts / acceptances between NEIGHBOURING rungs only (the
// pair (s, s+1) in either order), which keeps awsem_remd_state's meaning; i == j is drawn and counted as nothing, as in openmmtools.
#define AWSEM_STREAM_REMD_ALL 0x524d5341u
#define AWSEM_MIX_BATCH 256
__global__ void awsem_remd_mix_all_kernel(const double* __restrict__ U, const int* __restrict__ where, const double* __restrict__ ladder, int* __restrict__ slot,
                                          unsigned long long* __restrict__ attempted, unsigned long long* __restrict__ accepted, int N,
                                          long long iteration, unsigned long long seed, int local_start, int nlocal,
                                          double* __restrict__ T_local, double* __restrict__ scale_local) {
    extern __shared__ double shd[];   // beta[N], Us[N], lnu[BATCH]; then ints: s[N], old[N], pi[BATCH], pj[BATCH], att[N], acc[N]
    double* beta = shd;
    double* Us = shd + N;
    double* lnu = shd + 2 * N;
    int* s = (int*)(lnu + AWSEM_MIX_BATCH);
    int* old = s + N;
    int* pi = old + N;
    int* pj = pi + AWSEM_MIX_BATCH;
    int* att = pj + AWSEM_MIX_BATCH;
    int* acc = att + N;
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
        const int v = slot[r];
        s[r] = v; old[r] = v; att[r] = 0; acc[r] = 0;
        beta[r] = 1.0 / (AWSEM_KB * ladder[r]);      // by ladder index
        Us[r] = U[where[r]];                         // by replica
    }
    const long long total = (long long)N * N * N;
    for (long long a0 = 0; a0 < total; a0 += AWSEM_MIX_BATCH) {
        __syncthreads();
        const long long a = a0 + threadIdx.x;
        if (threadIdx.x < AWSEM_MIX_BATCH && a < total) {
            unsigned c[4] = {(unsigned)a, (unsigned)(unsigned long long)iteration, (unsigned)((unsigned long long)iteration >> 32) ^ (unsigned)((unsigned long long)a >> 32), AWSEM_STREAM_REMD_ALL};
            philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
            pi[threadIdx.x] = (int)(((unsigned long long)c[0] * (unsigned)N) >> 32);
            pj[threadIdx.x] = (int)(((unsigned long long)c[1] * (unsigned)N) >> 32);
            const double u = (double)((((unsigned long long)c[2] << 32) | c[3]) >> 11) * (1.0 / 9007199254740992.0);
            lnu[threadIdx.x] = u > 0.0 ? ::log(u) : -1e300;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const int nb = (int)((total - a0) < AWSEM_MIX_BATCH ? (total - a0) : AWSEM_MIX_BATCH);
            for (int k = 0; k < nb; k++) {
                const int i = pi[k], j = pj[k];
                if (i == j) continue;
                const int si = s[i], sj = s[j];
                const double log_p = (beta[si] - beta[sj]) * (Us[i] - Us[j]);
                const int lo = si < sj ? si : sj, near = (si - sj == 1) | (sj - si == 1);
                if (near) att[lo]++;
                if (log_p >= 0.0 || lnu[k] < log_p) { s[i] = sj; s[j] = si; if (near) acc[lo]++; }
            }
        }
    }
    __syncthreads();
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
        slot[r] = s[r];
        if (r < N - 1) { attempted[r] += (unsigned long long)att[r]; accepted[r] += (unsigned long long)acc[r]; }
    }
    for (int k = threadIdx.x; k < nlocal; k += blockDim.x) {
        const int r = local_start + k;
        const double Tn = ladder[s[r]], To = ladder[old[r]];
        T_local[k] = Tn;
        scale_local[k] *= ::sqrt(Tn / To);   // (1 unless an earlier sweep's factor has not been taken in yet: Engine::vel)
    }
}

// kinetic energy of every replica: sum over the massive particles of 1/2 m v^2 (StateDataReporter's temperature column)
template <class real>
__global__ void awsem_kinetic_kernel(const DevSys<real> S, const real* __restrict__ vel, double* __restrict__ ke) {
    const int natoms = S.natoms, rep = blockIdx.x;
    const real* v = vel + (size_t)rep * 3 * natoms;
    __shared__ double part[16];
    double s = 0;
    for (int k = threadIdx.x; k < S.nmassive; k += blockDim.x) {
        const MassiveItem<real> mi = S.massive[k];
        const double vx = v[mi.atom], vy = v[natoms + mi.atom], vz = v[2 * natoms + mi.atom];
        s += 0.5 * (vx * vx + vy * vy + vz * vz) / (double)mi.inv_mass;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += part[w];   // fixed order: reproducible
        ke[rep] = t;
    }
}

template <class real>
__global__ void awsem_vsite_kernel(const DevSys<real> S, real* __restrict__ pos) {
    const int natoms = S.natoms;
    real* p = pos + (size_t)blockIdx.x * 3 * natoms;
    for (int i = threadIdx.x; i < S.nvs; i += blockDim.x) {
        Int4 v = S.vs[i];
        real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        for (int k = 0; k < 3; k++) p[k * natoms + v.x] = w0 * p[k * natoms + v.y] + w1 * p[k * natoms + v.z] + w2 * p[k * natoms + v.w];
    }
}

// ------------------------------------------------------------------ fused multi-step Langevin kernel
// OpenMM LangevinIntegrator update (see oracle/awsem_oracle.c: awsem_oracle_step) with the state LDS-resident.
// Per step:  [pair lists, if an atom moved more than skin/2]  ->  forces (all groups, four barrier-separated phases)
//   ->  INTEGRATE: every massive particle gathers the forces of the virtual sites it parents (owner computes: no
//       atomics), clears its accumulator, takes the Langevin step and books its displacement since the last list build
//   ->  VIRTUAL SITES: positions from the new parents, accumulators cleared, displacement booked.
// The displacement maximum is double buffered (L.flag[s & 1] is read while L.flag[~s & 1] is being written).
// ---- centre-of-mass motion removal (OpenMM CMMotionRemover, frequency 1)
// The total momentum sum_i m_i v_i of a replica is accumulated in fixed point (integer adds commute: the sum does not depend on
// which lane owns which particle, so a trajectory stays bit-identical under any workgroup size or atom order).  fp32: Q15.16 in
// 32-bit lanes (one lane-step resolves 1.5e-5 amu nm/ps, i.e. < 1e-7 nm/ps in the mean velocity, below an fp32 ulp of a
// thermal velocity), reduced over the wave with six DPP adds per component; fp64: Q31.32, 64-bit shuffles.
__device__ __forceinline__ int wave_sum_lane63(int v) {   // inclusive scan over the wave: lane 63 ends with the total
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);    // row_shr:1  (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);    // row_shr:8  -> lane 15 of every row holds the row's sum
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}
template <class real> struct CmAcc;
template <> struct CmAcc<float> {
    static constexpr float scale = 65536.0f;
    int x = 0, y = 0, z = 0;
    __device__ __forceinline__ void add(float m, const R3<float>& v) {
        const float s = m * scale;
        x += __float2int_rn(s * v.x); y += __float2int_rn(s * v.y); z += __float2int_rn(s * v.z);
    }
    __device__ __forceinline__ void flush(u64* cm) const {   // whole wave, converged
        const int X = wave_sum_lane63(x), Y = wave_sum_lane63(y), Z = wave_sum_lane63(z);
        if ((threadIdx.x & 63) == 63) { fixed_add(&cm[0], (u64)(long long)X); fixed_add(&cm[1], (u64)(long long)Y); fixed_add(&cm[2], (u64)(long long)Z); }
    }
};
template <> struct CmAcc<double> {
    static constexpr double scale = AWSEM_FIXED_SCALE;
    u64 x = 0, y = 0, z = 0;
    __device__ __forceinline__ void add(double m, const R3<double>& v) {
        x += (u64)__double2ll_rn(m * v.x * scale); y += (u64)__double2ll_rn(m * v.y * scale); z += (u64)__double2ll_rn(m * v.z * scale);
    }
    __device__ __forceinline__ void flush(u64* cm) const {
        u64 s[3] = {x, y, z};
        for (int q = 0; q < 3; q++) {
            for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off);
            if ((threadIdx.x & 63) == 0) fixed_add(&cm[q], s[q]);
        }
    }
};
// Pacing (DevSys::prio mode 3): the replicas that share a CU should END a launch together -- a replica that ends early leaves its
// issue slots to fewer and fewer waves, and the launch waits for the slowest replica on the chip.  How long a replica takes is
// mostly its own structure (compact chains: residency of consecutive launches correlates 0.94), so every replica keeps a priority
// bias that the NEXT launch corrects by how far behind (ahead of) its CU's mean end it came in.  No communication inside a launch:
// thread 0 reads the previous launch's records of the workgroups it shared the CU with (b, b + #CUs, ... for a grid of one
// wave; the recorded CU id vouches for it) and writes its own at the end.  Records: [launch parity][replica] {end time, bias |
// CU id << 32 | duration << 44} behind the profiling area of DevSys::prof, wall-clock ticks of 10 ns.
__device__ __forceinline__ unsigned pace_cu_key() {
    return ((unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) << 8) | (((unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4) >> 8) & 0xffu);   // XCC_ID, HW_ID's CU/SH/SE bits
}
template <class real> __device__ __forceinline__ long long* pace_area(const DevSys<real>& S, int nrep, int parity) {
    return S.prof + 16 + 3 * (size_t)nrep + 16 * 20 + (size_t)parity * 2 * nrep;
}
// Placement: which replica a workgroup takes (perm) and what a replica cost last time (the lengths of its pair lists at the end of
// the launch: they explain 84 % of the variance of a compact replica's residency) -- awsem_place_kernel deals the replicas to the
// CUs so that every CU gets the same total, every few launches.
__host__ __device__ inline size_t pace_perm_offset(int nrep) { return 16 + (7 + (size_t)AWSEM_MAX_LISTS) * nrep + 16 * 20; }   // in long longs from DevSys::prof
template <class real> __device__ __forceinline__ int* pace_perm(const DevSys<real>& S, int nrep) { return (int*)(S.prof + pace_perm_offset(nrep)); }
template <class real> __device__ __forceinline__ int* pace_cost(const DevSys<real>& S, int nrep) { return pace_perm(S, nrep) + nrep; }
template <class real> __device__ __forceinline__ float pace_bias(const DevSys<real>& S, int rep, int nrep, int launch, unsigned cu) {
    const long long* prev = pace_area(S, nrep, (launch & 1) ^ 1);
    const int* perm = pace_perm(S, nrep);
    const long long own_end = prev[2 * rep], own_w = prev[2 * rep + 1];
    if (own_end == 0) return 0.f;                                      // first launch
    float bias = __uint_as_float((unsigned)own_w);
    const unsigned own_cu = (unsigned)(own_w >> 32) & 0xfffu;
    const float dur = (float)(int)((unsigned long long)own_w >> 44);
    if (own_cu != cu || dur <= 0.f) return bias;                       // dispatched elsewhere this time: keep, do not learn
    const int group = S.pace_group;
    long long sum = 0; int n = 0;
    for (int b = (int)blockIdx.x % group; b < nrep; b += group) {       // a replica that was elsewhere last time fails the CU test
        const int q = perm[b];
        const long long w = prev[2 * q + 1];
        if (((unsigned)(w >> 32) & 0xfffu) == own_cu && prev[2 * q] != 0) { sum += prev[2 * q] - own_end; n++; }
    }
    if (n < 2) return bias;
    const float lag = -(float)sum / ((float)n * dur);                  // own end behind the CU's mean end, as a fraction of the launch
    bias += (float)((S.prio >> 8) & 0xffff) * 0.0625f * lag;
    return bias < -3.f ? -3.f : bias > 3.f ? 3.f : bias;
}
template <class real> __device__ __forceinline__ void pace_record(const DevSys<real>& S, int rep, int nrep, int launch, unsigned cu, float bias, long long t0) {
    long long* cur = pace_area(S, nrep, launch & 1);
    const long long t1 = (long long)wall_clock64();
    long long dur = t1 - t0;
    dur = dur < 1 ? 1 : dur > 0xfffff ? 0xfffff : dur;
    cur[2 * rep] = t1;
    cur[2 * rep + 1] = (long long)(((unsigned long long)dur << 44) | ((unsigned long long)(cu & 0xfffu) << 32) | (unsigned long long)__float_as_uint(bias));
}
// Issue priority of a wave (s_setprio takes an immediate).  The SIMD arbiter serves the highest priority first and, among equals,
// the OLDEST wave: left alone, the replica dispatched first on a CU runs at nearly its solo speed, the three behind it get what
// is left, and the CU ends the launch with one, two, three of its four replicas already gone (measured with -DAWSEM_WALL: the
// replicas of a CU ended 157 us apart in a 873 us launch, finishing in dispatch order on 231 of 256 CUs).  Rotating the priority
// step by step makes them advance together (40 us apart, +6 % throughput).  A feedback form -- the replica furthest behind goes
// first, step counts shared through one L2 atomic per wave and step -- ends them 28 us apart but costs more than it gains.
__device__ __forceinline__ void set_wave_priority(int p) {
    switch (p & 3) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}
#ifndef AWSEM_STEP_ATTR
#define AWSEM_STEP_ATTR
#endif
template <class real, int LEAN = 0>
__global__ AWSEM_STEP_ATTR AWSEM_BOUNDS(real) void awsem_step_kernel(const DevSys<real> S, real* __restrict__ pos, real* __restrict__ vel,
                                  const double* __restrict__ T, int nsteps, double dt, double friction, unsigned long long seed,
                                  long long step0, int launch, int reuse, int stage_len) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    AWSEM_STAMP_WALL(S, 0);
    const long long pace_t0 = (S.prio & 0xff) == 3 ? (long long)wall_clock64() : 0;
    const int natoms = S.natoms, rep = (S.prio & 0xff) == 3 ? uniform(pace_perm(S, (int)gridDim.x)[blockIdx.x]) : (int)blockIdx.x;
    typedef typename std::conditional<LEAN == 3, LdsLeanZ<real>, typename std::conditional<LEAN != 0, LdsLean<real>, Lds<real>>::type>::type View;
    View L(carve_lds((real)0, smem, lds_shape(S), true));
    L.lists = S.lists + (size_t)rep * S.list_stride;
    real* gp = pos + (size_t)rep * 3 * natoms;
    real* gv = vel + (size_t)rep * 3 * natoms;
    load_positions(gp, L, natoms);
    // a replica-exchange sweep leaves the factor its new temperature asks for here instead of spending a launch on it: the product is
    // what awsem_take_scale_kernel would have stored, and the word goes back to 1 behind the next barrier
    const real vsc = S.vel_scale ? (real)S.vel_scale[rep] : (real)1;
    for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {   // velocities live in LDS for the massive particles only, by massive index
        const int i = S.massive[k].atom;
        L.set_vel(k, R3<real>{gv[i] * vsc, gv[natoms + i] * vsc, gv[2 * natoms + i] * vsc});
    }
    for (int i = threadIdx.x; i < S.max_contact_sites; i += L.nthr) L.rho[i] = 0;
    const double a_d = ::exp(-friction * dt);
    const real a = (real)a_d, fs = (real)(friction == 0.0 ? dt : (1.0 - a_d) / friction);
    // The launch runs nsteps / stage_len temperature stages back to back (awsem_step_stages: T is [stage][replica]); the noise
    // amplitude is set at the first step of each, from the same expression a launch of that stage alone evaluates.
    const real dtr = (real)dt;
    real ns = 0;
    int stage_next = 0;
    stage_tables(S, L);
    if (threadIdx.x < 2) L.flag[threadIdx.x] = 0;
    if (threadIdx.x == 0) L.flag[2] = __builtin_amdgcn_s_getreg((3 << 11) | 4);   // HW_ID[3:0]: wave 0's slot on its SIMD -- distinct among the replicas that share the CU (measured: 0..3 in dispatch order)
    const bool paced = (S.prio & 0xff) == 3;
    const unsigned pace_cu = paced ? pace_cu_key() : 0;
    if (paced && threadIdx.x == 0) L.flag[3] = __float_as_uint(pace_bias(S, rep, (int)gridDim.x, launch, pace_cu));
    __syncthreads();
    if (S.vel_scale && threadIdx.x == 0) S.vel_scale[rep] = 1.0;
    const int prio_slot = uniform((int)L.flag[2]);
    const float prio_bias = paced ? __uint_as_float(uniform((int)L.flag[3])) : 0.f;
    compute_virtual_sites(S, L);
    // `reuse`: the previous launch left this replica's lists, their lengths and the positions they were built at (host: step_impl keeps
    // track of what invalidates them) -- go on with them; the displacement since then decides about the first rebuild like any other.
    if (reuse) {
        if (threadIdx.x < AWSEM_MAX_LISTS) L.lcount[threadIdx.x] = S.list_counts[(size_t)rep * AWSEM_MAX_LISTS + threadIdx.x];
        const real* gr = S.list_ref + (size_t)rep * 3 * S.nmassive;
        float m = 0;
        for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {
            const int i = S.massive[k].atom;
            const R3<real> r{gr[3 * k], gr[3 * k + 1], gr[3 * k + 2]};
            L.set_ref(i, r);
            m = fmaxf(m, (float)dist2(L.pos(i) - r));
        }
        book_displacement(&L.flag[0], m);
    }
    const bool cmr = LEAN || S.remove_cm != 0;   // (lean: chosen only with the centre-of-mass motion remover on, forces_setup.py's default)
    if (cmr) {   // momentum of the incoming velocities: what the first step removes (L.cm[0..2]; zeroed at the carve-up, a barrier ago)
        CmAcc<real> acc;
        for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) acc.add(M<real>::rcp(S.massive[k].inv_mass), L.vel(k));
        acc.flush(L.cm);
    }
    __syncthreads();
    AWSEM_STAMP_INIT;
    // The lists are rebuilt when a MASSIVE particle has moved further than skin/2 since the last build; lists screened on a
    // virtual site carry a correspondingly larger margin (host: list_margin), so the sites need no displacement test.
    const bool fuse = LEAN || S.fuse_vs != 0;   // (the lean instantiation is only chosen for systems that fuse, and whose random numbers are drawn ahead: `pre` below)
    int boost_left = 0;
    for (int s = 0; s < nsteps; s++) {
        const int cur = s & 1;
        const bool rebuild = (s == 0 && !reuse) || __uint_as_float(L.flag[cur]) > (float)S.skin_half2;   // uniform across the workgroup
        if (s == stage_next) {
            ns = (real)::sqrt(AWSEM_KB * T[rep] * (1.0 - a_d * a_d));
            T += gridDim.x; stage_next += stage_len;
        }
        if (S.prio == 1) set_wave_priority(prio_slot + s);   // this replica's turn at each of the four levels comes round every four steps
        else if (paced) {   // the same rotation, shifted by the replica's bias (dithered step by step, so a fractional bias acts as such)
            const int shift = (int)floorf(prio_bias + (float)(((unsigned)s * 2654435769u) >> 8) * (1.f / 16777216.f));
            const int p = ((prio_slot + s) & 3) + shift;
            // a rebuild sets the replica back by about one step against its companions, which no earlier launch could foresee: it goes
            // first for the next few steps
            if (rebuild && s > 0) boost_left = (S.prio >> 24) & 15;
            if (boost_left > 0) { boost_left--; set_wave_priority(3); }
            else set_wave_priority(p < 0 ? 0 : p > 3 ? 3 : p);
        }
        else if (!LEAN && S.prio == 2) set_wave_priority(prio_slot + (int)(wall_clock64() >> 11));   // by the clock instead (100 MHz: turns of 20 us); measured equal
        if (cmr && threadIdx.x < 3) {
            L.cm[(cur ^ 1) * 3 + threadIdx.x] = 0;   // read for the last time in the previous step's integrator; this step's adds to it
            // the mean velocity this step removes, as reals, for every lane to pick up in the integrator (several barriers from here)
            ((real*)(L.cm + 6))[cur * 4 + threadIdx.x] = (real)(long long)L.cm[cur * 3 + threadIdx.x] * (S.inv_total_mass * (real)(1.0 / (double)CmAcc<real>::scale));
        }
        if (rebuild || !fuse) {   // the virtual sites in a phase of their own: the hydrogen-bond list builders read them
            update_virtual_sites(S, L);
            __syncthreads();
            AWSEM_STAMP(S, 0);
        }
        if (rebuild) {
            if (threadIdx.x < AWSEM_MAX_LISTS) L.lcount[threadIdx.x] = 0;
            for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) { const int i = S.massive[k].atom; L.set_ref(i, L.pos(i)); }
            __syncthreads();
#ifdef AWSEM_PROF
            AWSEM_ARRIVE(2);
            build_lists_prof<LEAN != 0>(S, L, 0xffffffffu, awsem_prof);
#else
            build_lists<LEAN != 0>(S, L, 0xffffffffu);
#endif
            AWSEM_ARRIVE(2);
            __syncthreads();
            AWSEM_STAMP(S, 2);
#ifdef AWSEM_PROF
            if (threadIdx.x == 0) atomicAdd((unsigned long long*)&S.prof[10], 1ull);   // number of list rebuilds
#endif
        }
        // the step's random numbers depend on nothing but the counter: generated here, ahead of the force phases
        // (for up to two particles per lane; larger systems draw them in the integrator as before)
        real g_pre[2][3] = {{0, 0, 0}, {0, 0, 0}}, im_pre[2] = {0, 0}, ms_pre[2] = {0, 0};
        int at_pre[2] = {0, 0};
        const bool pre = LEAN || S.nmassive <= 2 * L.nthr;
        if (pre) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int k = (int)threadIdx.x + q * L.nthr;
                if (k < S.nmassive) {   // the particle's static record is consumed here too: noise amplitude, force factor, atom
                    const MassiveItem<real> mi = S.massive[k];
                    gauss3<real>(seed, AWSEM_STREAM_LANGEVIN, rep, step0 + s, mi.id, g_pre[q]);
                    const real sq = ns * mi.sqrt_inv_mass;
                    g_pre[q][0] *= sq; g_pre[q][1] *= sq; g_pre[q][2] *= sq;
                    im_pre[q] = fs * mi.inv_mass; at_pre[q] = mi.atom;
                    if (cmr) ms_pre[q] = M<real>::rcp(mi.inv_mass);
                }
            }
        }
        evaluate_terms<false, true, LEAN>(S, L, 0xffffffffu, fuse && !rebuild AWSEM_STAMP_PASS);
        if (threadIdx.x == 0) L.flag[cur] = 0;   // read by everyone before the barriers of the evaluation; written again next step
        {
            float m = 0;
            R3<real> vcm{0, 0, 0};
            if (cmr) { const real* q = (const real*)(L.cm + 6) + cur * 4; vcm = R3<real>{q[0], q[1], q[2]}; }   // the momentum the previous step (or the launch start) left
            CmAcc<real> acc;
            for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {   // compact list: no lane idles on a virtual site
                const GatherItem<real> gi = S.gather[k];
                const bool second = k >= L.nthr;
                real g[3], im, ms = 0;
                int i;
                if (pre) {
                    i = second ? at_pre[1] : at_pre[0]; im = second ? im_pre[1] : im_pre[0]; ms = second ? ms_pre[1] : ms_pre[0];
                    g[0] = second ? g_pre[1][0] : g_pre[0][0]; g[1] = second ? g_pre[1][1] : g_pre[0][1]; g[2] = second ? g_pre[1][2] : g_pre[0][2];
                } else {
                    const MassiveItem<real> mi = S.massive[k];
                    i = mi.atom; im = fs * mi.inv_mass;
                    if (cmr) ms = M<real>::rcp(mi.inv_mass);
                    gauss3<real>(seed, AWSEM_STREAM_LANGEVIN, rep, step0 + s, mi.id, g);
                    const real sq = ns * mi.sqrt_inv_mass;
                    g[0] *= sq; g[1] *= sq; g[2] *= sq;
                }
                R3<real> F = L.force(i);
                L.zero_force(i);
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (gi.vs[q] < 0) continue;
                    const R3<real> fv = L.force(gi.vs[q]);
                    F.x += gi.w[q] * fv.x; F.y += gi.w[q] * fv.y; F.z += gi.w[q] * fv.z;
                }
                // fp32 accumulators hold +-5.2e5 kJ/mol/nm per component (Q19.12): a force beyond half of that means overlapping
                // atoms -- flag it (bit 2 of the status word) instead of letting the fixed-point sums wrap silently
                if (sizeof(real) == 4 && !(fabsf((float)F.x) < 2.6e5f && fabsf((float)F.y) < 2.6e5f && fabsf((float)F.z) < 2.6e5f)) atomicOr(S.status, 4);
                R3<real> v = L.vel(k), x = L.pos(i);
                v.x = a * (v.x - vcm.x) + im * F.x + g[0];
                v.y = a * (v.y - vcm.y) + im * F.y + g[1];
                v.z = a * (v.z - vcm.z) + im * F.z + g[2];
                x.x += dtr * v.x; x.y += dtr * v.y; x.z += dtr * v.z;
                L.set_vel(k, v); L.set_pos(i, x);
                if (cmr) acc.add(ms, v);
                m = fmaxf(m, (float)dist2(x - L.refpos(i)));
            }
            for (int i = threadIdx.x; i < S.max_contact_sites; i += L.nthr) L.rho[i] = 0;   // last read in the per-site pass
            if (cmr) acc.flush(&L.cm[(cur ^ 1) * 3]);
            book_displacement(&L.flag[cur ^ 1], m);
        }
        AWSEM_ARRIVE(9);
        __syncthreads();
        AWSEM_STAMP(S, 9);
    }
    compute_virtual_sites(S, L);   // the sites of the final coordinates (a third inlined copy of update_virtual_sites here tips the
    __syncthreads();               // compiler into spilling the kernarg block: __graft_entry__.build() checks for that)
    AWSEM_STAMP_FLUSH(S);
    for (int i = threadIdx.x; i < natoms; i += L.nthr) {
        const R3<real> x = L.pos(i);
        gp[i] = x.x; gp[natoms + i] = x.y; gp[2 * natoms + i] = x.z;
    }
    for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {
        const int i = S.massive[k].atom;
        const R3<real> v = L.vel(k);
        gv[i] = v.x; gv[natoms + i] = v.y; gv[2 * natoms + i] = v.z;
    }
    {   // for the next launch (see `reuse` above)
        real* gr = S.list_ref + (size_t)rep * 3 * S.nmassive;
        for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {
            const R3<real> r = L.refpos(S.massive[k].atom);
            gr[3 * k] = r.x; gr[3 * k + 1] = r.y; gr[3 * k + 2] = r.z;
        }
        if (threadIdx.x < AWSEM_MAX_LISTS) S.list_counts[(size_t)rep * AWSEM_MAX_LISTS + threadIdx.x] = L.lcount[threadIdx.x];
    }
    AWSEM_STAMP_WALL(S, 1);
#if defined(AWSEM_PROF) || defined(AWSEM_WALL)
    if ((int)threadIdx.x < S.n_lists) S.prof[16 + 7 * (size_t)gridDim.x + 16 * 20 + (size_t)rep * AWSEM_MAX_LISTS + threadIdx.x] = L.lcount[threadIdx.x];
#endif
    if (threadIdx.x == 0) {
        if (paced) pace_record(S, rep, (int)gridDim.x, launch, pace_cu, prio_bias, pace_t0);
        int c = 0;
        for (int l = 0; l < S.n_lists; l++) c += (int)L.lcount[l];
        pace_cost(S, (int)gridDim.x)[rep] = c;   // read by the placement and by the skin rule (awsem_list_mean_kernel)
    }
}
// Mean number of listed pairs per replica after the last launch, for the host's skin rule: written to pinned host memory as
// {mean, tag} -- the tag last, behind a system-scope fence, so that a host that sees the tag of its request sees that request's mean.
__global__ void awsem_list_mean_kernel(const long long* prof, int nrep, double tag, volatile double* out) {
    __shared__ double part[256];
    const int* cost = (const int*)(prof + pace_perm_offset(nrep)) + nrep;
    double s = 0;
    for (int r = threadIdx.x; r < nrep; r += 256) s += (double)cost[r];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) { if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off]; __syncthreads(); }
    if (threadIdx.x == 0) { out[0] = part[0] / (double)nrep; __threadfence_system(); out[1] = tag; }
}
// One workgroup: rank the replicas by cost (heaviest first) and deal them to the CUs boustrophedon -- row 0 left to right, row 1 right
// to left, ... -- so that the workgroups b, b + group, b + 2 group, ... of a CU hold one replica of every weight class.
__global__ void awsem_place_kernel(long long* prof, int nrep, int group) {
    extern __shared__ int place_cost[];
    int* perm = (int*)(prof + pace_perm_offset(nrep));
    const int* cost = perm + nrep;
    for (int r = threadIdx.x; r < nrep; r += blockDim.x) place_cost[r] = cost[r];
    __syncthreads();
    for (int r = threadIdx.x; r < nrep; r += blockDim.x) {
        const int c = place_cost[r];
        int rank = 0;
        for (int q = 0; q < nrep; q++) { const int cq = place_cost[q]; rank += (cq > c || (cq == c && q < r)) ? 1 : 0; }
        const int row = rank / group, col = rank - row * group;
        const int width = nrep - row * group < group ? nrep - row * group : group;
        perm[row * group + ((row & 1) ? width - 1 - col : col)] = r;
    }
}

// ------------------------------------------------------------------ energy minimiser (simulation.minimizeEnergy())
// FIRE (Bitzek et al. 2006) on the massive particles with unit masses, every replica in its own workgroup with the
// state LDS-resident and the same force phases as the step kernel; restated in fp64 in oracle/awsem_oracle.c
// (awsem_oracle_minimize).  OpenMM's LocalEnergyMinimizer is L-BFGS; the reference pins nothing about the minimiser
// beyond its contract (a local minimum with RMS force <= tolerance), which is what the tests check.
// The three sums an iteration needs (F.v, F.F, v.v) are fixed-point LDS accumulators (L.eg[0..2]), so a minimisation is
// bit-reproducible; F.v and F.F are scaled by 2^-10 / 2^-20 to stay inside Q31.32 for the clashing start structures.
template <class real>
__global__ AWSEM_BOUNDS(real) void awsem_minimize_kernel(const DevSys<real> S, real* __restrict__ pos, double tolerance, int max_iterations,
                                      int* __restrict__ iterations /* [R] */, double* __restrict__ rms_out /* [R] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int natoms = S.natoms, rep = blockIdx.x;
    Lds<real> L = carve_lds((real)0, smem, lds_shape(S), true);
    L.lists = S.lists + (size_t)rep * S.list_stride;
    real* gp = pos + (size_t)rep * 3 * natoms;
    load_positions(gp, L, natoms);
    for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) L.set_vel(k, R3<real>{0, 0, 0});
    for (int i = threadIdx.x; i < S.max_contact_sites; i += L.nthr) L.rho[i] = 0;
    stage_tables(S, L);
    if (threadIdx.x < 2) L.flag[threadIdx.x] = 0;
    __syncthreads();
    compute_virtual_sites(S, L);
    __syncthreads();
    real dt = (real)0.001, alpha = (real)0.1;
    const real dt_max = (real)0.01, dmax2 = (real)(0.02 * 0.02);
    const double sP = 1.0 / 1024.0, sF = 1.0 / (1024.0 * 1024.0);
    int npos = 0, it = 0;
    double rms = 0;
    AWSEM_STAMP_INIT;
    for (; it < max_iterations; it++) {
        const int cur = it & 1;
        if (it == 0 || __uint_as_float(L.flag[cur]) > (float)S.skin_half2) {   // uniform across the workgroup
            if (threadIdx.x < AWSEM_MAX_LISTS) L.lcount[threadIdx.x] = 0;
            for (int i = threadIdx.x; i < natoms; i += L.nthr) L.set_ref(i, L.pos(i));
            __syncthreads();
            build_lists(S, L, 0xffffffffu);
            __syncthreads();
        }
        if (threadIdx.x < 3) L.eg[threadIdx.x] = 0;   // visible before anyone adds: the evaluation has barriers
        evaluate_terms<false>(S, L, 0xffffffffu, false AWSEM_STAMP_PASS);
        if (threadIdx.x == 0) L.flag[cur] = 0;
        {   // pass 1: the sums
            Energy eP, eF, eV;
            for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {
                const GatherItem<real> gi = S.gather[k];
                const int i = S.massive[k].atom;
                R3<real> F = L.force(i);
#pragma unroll
                for (int q = 0; q < AWSEM_MAX_GATHER; q++) {
                    if (gi.vs[q] < 0) continue;
                    const R3<real> fv = L.force(gi.vs[q]);
                    F.x += gi.w[q] * fv.x; F.y += gi.w[q] * fv.y; F.z += gi.w[q] * fv.z;
                }
                const R3<real> v = L.vel(k);
                eP += (double)(F.x * v.x + F.y * v.y + F.z * v.z) * sP;
                eF += (double)(F.x * F.x + F.y * F.y + F.z * F.z) * sF;
                eV += (double)(v.x * v.x + v.y * v.y + v.z * v.z);
            }
            flush_energy<true>(L, 0, eP); flush_energy<true>(L, 1, eF); flush_energy<true>(L, 2, eV);
        }
        __syncthreads();
        const double P = fixed_to_double(L.eg[0]) / sP, FF = fixed_to_double(L.eg[1]) / sF, VV = fixed_to_double(L.eg[2]);
        rms = ::sqrt(FF / (3.0 * (S.nmassive > 0 ? S.nmassive : 1)));
        if (rms <= tolerance) break;   // uniform
        real keep = 0, mix = 0;
        if (P >= 0) {   // not uphill (the first iteration has v = 0)
            keep = (real)1 - alpha; mix = alpha * (real)::sqrt(VV / FF);
            if (++npos > 5) { dt = dt * (real)1.1 < dt_max ? dt * (real)1.1 : dt_max; alpha *= (real)0.99; }
        } else { npos = 0; dt *= (real)0.5; alpha = (real)0.1; }
        {   // pass 2: velocity mixing, Euler step with the displacement clamp
            float m = 0;
            for (int k = threadIdx.x; k < S.nmassive; k += L.nthr) {
                const GatherItem<real> gi = S.gather[k];
                const int i = S.massive[k].atom;
                R3<real> F = L.force(i);
                L.zero_force(i);
#pragma unroll
                for (int q = 0; q < AWSEM_MAX_GATHER; q++) {
                    if (gi.vs[q] < 0) continue;
                    const R3<real> fv = L.force(gi.vs[q]);
                    F.x += gi.w[q] * fv.x; F.y += gi.w[q] * fv.y; F.z += gi.w[q] * fv.z;
                }
                R3<real> v = L.vel(k), x = L.pos(i);
                v.x = keep * v.x + mix * F.x + dt * F.x;
                v.y = keep * v.y + mix * F.y + dt * F.y;
                v.z = keep * v.z + mix * F.z + dt * F.z;
                real dx = dt * v.x, dy = dt * v.y, dz = dt * v.z;
                const real d2 = dx * dx + dy * dy + dz * dz;
                if (d2 > dmax2) { const real sc = M<real>::sqrt(dmax2 / d2); dx *= sc; dy *= sc; dz *= sc; }
                x.x += dx; x.y += dy; x.z += dz;
                L.set_vel(k, v); L.set_pos(i, x);
                m = fmaxf(m, (float)dist2(x - L.refpos(i)));
            }
            book_displacement(&L.flag[cur ^ 1], m);
        }
        __syncthreads();
        {
            float m = 0;
            for (int i = threadIdx.x; i < S.nvs; i += L.nthr) {
                const Int4 v = S.vs[i];
                const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
                const R3<real> pa = L.pos(v.y), pb = L.pos(v.z), pc = L.pos(v.w);
                const R3<real> x = R3<real>{w0 * pa.x + w1 * pb.x + w2 * pc.x, w0 * pa.y + w1 * pb.y + w2 * pc.y, w0 * pa.z + w1 * pb.z + w2 * pc.z};
                L.set_pos(v.x, x);
                L.zero_force(v.x);
                m = fmaxf(m, (float)dist2(x - L.refpos(v.x)));
            }
            for (int i = threadIdx.x; i < S.nfixed; i += L.nthr) L.zero_force(S.fixed[i]);
            for (int i = threadIdx.x; i < S.max_contact_sites; i += L.nthr) L.rho[i] = 0;
            book_displacement(&L.flag[cur ^ 1], m);
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < natoms; i += L.nthr) {
        const R3<real> x = L.pos(i);
        gp[i] = x.x; gp[natoms + i] = x.y; gp[2 * natoms + i] = x.z;
    }
    if (threadIdx.x == 0) { iterations[rep] = it; rms_out[rep] = rms; }
}
