// awsem_kernels.h -- CDNA4 (gfx950) kernels of the replica-resident AWSEM path.
//
// Execution model: ONE WORKGROUP OWNS ONE REPLICA.  The replica's coordinates, velocities and int64 fixed-point
// force accumulators live in LDS for the whole launch (a 1r69 replica is ~20 KB, so 4-8 replicas are resident per
// CU); index arrays and parameter tables are shared by every replica and stay L2-resident.  A launch evaluates
// energies/forces once (awsem_eval_kernel) or advances `nsteps` Langevin steps without touching HBM in between
// (awsem_step_kernel).  Nothing here is a dense contraction, so no MFMA: the work is distance screens,
// transcendental switching functions and LDS atomics; wave-uniform "row" operands are LDS broadcasts.
//
// Determinism: every force, energy and density contribution is converted to 2^-32 fixed point and accumulated
// with integer LDS atomics (ds_add_u64), so results are bit-identical from run to run regardless of wave order.
#pragma once
#include <hip/hip_runtime.h>
#include "awsem_device.h"

typedef unsigned long long u64;

// ------------------------------------------------------------------ scalar helpers
template <class real> struct M;
template <> struct M<float> {
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
    static __device__ __forceinline__ float rsqrt(float x) { return rsqrtf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float floor(float x) { return floorf(x); }
    static __device__ __forceinline__ float log(float x) { return __logf(x); }
    static __device__ __forceinline__ void sincos2pi(float u, float* s, float* c) { *s = sinpif(2.0f * u); *c = cospif(2.0f * u); }
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

// value * 2^32 as a two's complement 64-bit integer.  fp32: floor + fraction split (5 VALU ops, exact for |v| < 2^31).
__device__ __forceinline__ u64 to_fixed(float v) {
    float fl = floorf(v);
    int hi = (int)fl;
    unsigned lo = (unsigned)((v - fl) * 4294967296.0f);
    return ((u64)(unsigned)hi << 32) | (u64)lo;
}
__device__ __forceinline__ u64 to_fixed(double v) { return (u64)(long long)(v * AWSEM_FIXED_SCALE); }
__device__ __forceinline__ double fixed_to_double(u64 u) { return (double)(long long)u * (1.0 / AWSEM_FIXED_SCALE); }

template <class real> struct Lds {
    real *px, *py, *pz;      // positions (SoA)
    u64 *fx, *fy, *fz;       // fixed-point force accumulators
    real *vx, *vy, *vz;      // velocities (step kernel only)
    u64 *rho, *drho;         // contact density and dE/drho accumulators (fixed point)
    real *rhov, *drhov;      // converted density / dE/drho
    real *alpha, *dalpha;    // z-dependent membrane switch
    real *gamma;             // staged gamma tables of the current contact term (env-major)
    u64* eg;                 // per-group energies (fixed point)
};

template <class real> __device__ __forceinline__ void add_force(const Lds<real>& L, int a, real x, real y, real z) {
    atomicAdd(&L.fx[a], to_fixed(x));
    atomicAdd(&L.fy[a], to_fixed(y));
    atomicAdd(&L.fz[a], to_fixed(z));
}
// force of a pair potential: d = x_i - x_j, s = -dE/dr / r
template <class real> __device__ __forceinline__ void pair_force(const Lds<real>& L, int i, int j, real dx, real dy, real dz, real s) {
    real X = s * dx, Y = s * dy, Z = s * dz;
    add_force(L, i, X, Y, Z);
    add_force(L, j, -X, -Y, -Z);
}

template <bool ENERGY, class real> __device__ __forceinline__ void flush_energy(const Lds<real>& L, int group, double& e) {
    if (ENERGY) {
        for (int off = 32; off > 0; off >>= 1) e += __shfl_down(e, off);
        if ((threadIdx.x & 63) == 0 && e != 0.0) atomicAdd(&L.eg[group], to_fixed(e));
        e = 0.0;
    }
}

// sigmoid(x) = 1/(1+exp(-x)) = (1+tanh(x/2))/2.  All tanh switches of the reference are evaluated in this form:
// it keeps full relative accuracy in the tails where 1+tanh cancels, and costs one exp + one rcp.
template <class real> __device__ __forceinline__ real sigmoid(real x) { return M<real>::rcp((real)1 + M<real>::exp(-x)); }

// ------------------------------------------------------------------ bonded terms
template <bool ENERGY, class real> __device__ void term_harmonic(const HarmD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        HarmItem<real> it = d.items[b];
        real dx = L.px[it.i] - L.px[it.j], dy = L.py[it.i] - L.py[it.j], dz = L.pz[it.i] - L.pz[it.j];
        real r2 = dx * dx + dy * dy + dz * dz, ir = M<real>::rsqrt(r2), r = r2 * ir, dr = r - it.r0;
        if (ENERGY) e += (double)((real)0.5 * it.k * dr * dr);
        pair_force(L, it.i, it.j, dx, dy, dz, -it.k * dr * ir);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class real> __device__ void term_chi(const ChiD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        Int4 a = d.atoms[b];   // CA, C, N, CB
        real ux = L.px[a.x] - L.px[a.y], uy = L.py[a.x] - L.py[a.y], uz = L.pz[a.x] - L.pz[a.y];
        real vx = L.px[a.z] - L.px[a.x], vy = L.py[a.z] - L.py[a.x], vz = L.pz[a.z] - L.pz[a.x];
        real wx = L.px[a.x] - L.px[a.w], wy = L.py[a.x] - L.py[a.w], wz = L.pz[a.x] - L.pz[a.w];
        real uvx = uy * vz - uz * vy, uvy = uz * vx - ux * vz, uvz = ux * vy - uy * vx;
        real vwx = vy * wz - vz * wy, vwy = vz * wx - vx * wz, vwz = vx * wy - vy * wx;
        real wux = wy * uz - wz * uy, wuy = wz * ux - wx * uz, wuz = wx * uy - wy * ux;
        real u2 = ux * ux + uy * uy + uz * uz, v2 = vx * vx + vy * vy + vz * vz, w2 = wx * wx + wy * wy + wz * wz;
        real norm = M<real>::rsqrt(u2 * v2 * w2);
        real chi = (uvx * wx + uvy * wy + uvz * wz) * norm, dchi = chi - d.chi0;
        if (ENERGY) e += (double)(d.k * dchi * dchi);
        real c = (real)-2 * d.k * dchi, cu = chi / u2, cv = chi / v2, cw = chi / w2;
        real gux = vwx * norm - cu * ux, guy = vwy * norm - cu * uy, guz = vwz * norm - cu * uz;
        real gvx = wux * norm - cv * vx, gvy = wuy * norm - cv * vy, gvz = wuz * norm - cv * vz;
        real gwx = uvx * norm - cw * wx, gwy = uvy * norm - cw * wy, gwz = uvz * norm - cw * wz;
        add_force(L, a.x, c * (gux - gvx + gwx), c * (guy - gvy + gwy), c * (guz - gvz + gwz));
        add_force(L, a.y, -c * gux, -c * guy, -c * guz);
        add_force(L, a.z, c * gvx, c * gvy, c * gvz);
        add_force(L, a.w, -c * gwx, -c * gwy, -c * gwz);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// cos/sin of the IUPAC dihedral p0-p1-p2-p3 and the gradient of the angle (Blondel & Karplus form).
template <class real> struct Dihedral { real c, s; real g[4][3]; };
template <class real> __device__ __forceinline__ void dihedral(const Lds<real>& L, int p0, int p1, int p2, int p3, Dihedral<real>& D) {
    real Fx = L.px[p0] - L.px[p1], Fy = L.py[p0] - L.py[p1], Fz = L.pz[p0] - L.pz[p1];
    real Gx = L.px[p1] - L.px[p2], Gy = L.py[p1] - L.py[p2], Gz = L.pz[p1] - L.pz[p2];
    real Hx = L.px[p3] - L.px[p2], Hy = L.py[p3] - L.py[p2], Hz = L.pz[p3] - L.pz[p2];
    real Ax = Fy * Gz - Fz * Gy, Ay = Fz * Gx - Fx * Gz, Az = Fx * Gy - Fy * Gx;
    real Bx = Hy * Gz - Hz * Gy, By = Hz * Gx - Hx * Gz, Bz = Hx * Gy - Hy * Gx;
    real A2 = Ax * Ax + Ay * Ay + Az * Az, B2 = Bx * Bx + By * By + Bz * Bz, G2 = Gx * Gx + Gy * Gy + Gz * Gz;
    real iAB = M<real>::rsqrt(A2 * B2), iG = M<real>::rsqrt(G2), Gn = G2 * iG;
    D.c = (Ax * Bx + Ay * By + Az * Bz) * iAB;
    real Cx = By * Az - Bz * Ay, Cy = Bz * Ax - Bx * Az, Cz = Bx * Ay - By * Ax;   // B x A
    D.s = (Cx * Gx + Cy * Gy + Cz * Gz) * iAB * iG;
    real FG = Fx * Gx + Fy * Gy + Fz * Gz, HG = Hx * Gx + Hy * Gy + Hz * Gz;
    real a = Gn / A2, b = Gn / B2, ta = FG / (A2 * Gn), tb = HG / (B2 * Gn);
    real A[3] = {Ax, Ay, Az}, B[3] = {Bx, By, Bz};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        real dA = -a * A[k], dB = b * B[k], t = ta * A[k] - tb * B[k];
        D.g[0][k] = dA; D.g[3][k] = dB; D.g[1][k] = -dA + t; D.g[2][k] = -dB - t;
    }
}

template <bool ENERGY, class real> __device__ void term_rama(const RamaD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        Int5 a = d.atoms[b];
        Dihedral<real> P, S;
        dihedral(L, a.a[0], a.a[1], a.a[2], a.a[3], P);
        dihedral(L, a.a[1], a.a[2], a.a[3], a.a[4], S);
        real dEdphi = 0, dEdpsi = 0;
        for (int w = 0; w < d.nwell; w++) {
            real W = d.bw[b * d.nwell + w];   // -k * w * weight, folded
            real cp = P.c * d.cphi0[w] + P.s * d.sphi0[w] - (real)1, sp = P.s * d.cphi0[w] - P.c * d.sphi0[w];
            real cs = S.c * d.cpsi0[w] + S.s * d.spsi0[w] - (real)1, ss = S.s * d.cpsi0[w] - S.c * d.spsi0[w];
            real ex = W * M<real>::exp(-d.sigma[w] * (d.omega_phi[w] * cp * cp + d.omega_psi[w] * cs * cs));
            if (ENERGY) e += (double)ex;
            dEdphi += ex * (real)2 * d.sigma[w] * d.omega_phi[w] * cp * sp;
            dEdpsi += ex * (real)2 * d.sigma[w] * d.omega_psi[w] * cs * ss;
        }
        // atoms 1..3 receive both contributions
        add_force(L, a.a[0], -dEdphi * P.g[0][0], -dEdphi * P.g[0][1], -dEdphi * P.g[0][2]);
#pragma unroll
        for (int m = 1; m < 4; m++)
            add_force(L, a.a[m], -dEdphi * P.g[m][0] - dEdpsi * S.g[m - 1][0], -dEdphi * P.g[m][1] - dEdpsi * S.g[m - 1][1],
                      -dEdphi * P.g[m][2] - dEdpsi * S.g[m - 1][2]);
        add_force(L, a.a[4], -dEdpsi * S.g[3][0], -dEdpsi * S.g[3][1], -dEdpsi * S.g[3][2]);
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

template <bool ENERGY, class real> __device__ void term_helical(const HelicalD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        HelicalItem<real> it = d.items[b];
        real ax = L.px[it.o] - L.px[it.n], ay = L.py[it.o] - L.py[it.n], az = L.pz[it.o] - L.pz[it.n];
        real bx = L.px[it.o] - L.px[it.h], by = L.py[it.o] - L.py[it.h], bz = L.pz[it.o] - L.pz[it.h];
        real r1s = ax * ax + ay * ay + az * az, r2s = bx * bx + by * by + bz * bz;
        real i1 = M<real>::rsqrt(r1s), i2 = M<real>::rsqrt(r2s), g1, g2;
        real th = hb_theta(r1s * i1, r2s * i2, d.r_ON, d.r_OH, d.inv_sNO2, d.inv_sHO2, g1, g2);
        if (ENERGY) e += (double)(-it.coef * th);
        pair_force(L, it.o, it.n, ax, ay, az, it.coef * g1 * i1);
        pair_force(L, it.o, it.h, bx, by, bz, it.coef * g2 * i2);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class real> __device__ void term_fm(const FmD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        int i = d.pairs[2 * b], j = d.pairs[2 * b + 1];
        real dx = L.px[i] - L.px[j], dy = L.py[i] - L.py[j], dz = L.pz[i] - L.pz[j];
        real r2 = dx * dx + dy * dy + dz * dz, ir = M<real>::rsqrt(r2), r = r2 * ir;
        real fi = M<real>::floor(r * d.inv_dr);
        fi -= (r - fi * d.dr < (real)0) ? (real)1 : (real)0;   // r*inv_dr may round up across a bin edge: keep r >= fi*dr
        int i1 = fi > (real)(d.ntab - 2) ? d.ntab - 2 : (int)fi;
        const real* row = d.table + (size_t)b * d.ntab + i1;
        real v1 = row[0], v2 = row[1];
        real slope = (v2 - v1) * d.inv_dr;
        if (ENERGY) e += (double)(-d.k * (v1 + slope * (r - d.dr * (real)i1)));
        pair_force(L, i, j, dx, dy, dz, d.k * slope * ir);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class real> __device__ void term_memb(const MembD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        MembItem<real> it = d.items[b];
        real z = L.pz[it.atom] - d.center;
        // 1/2 tanh(u) + 1/2 tanh(v) = sig(2u) + sig(2v) - 1
        real su = sigmoid(d.k_m2 * (z + d.z_m)), sv = sigmoid(d.k_m2 * (d.z_m - z));
        if (ENERGY) e += (double)(d.k * it.scale * (su + sv - (real)1));
        real dEdz = d.k * it.scale * d.k_m2 * (su * ((real)1 - su) - sv * ((real)1 - sv));
        atomicAdd(&L.fz[it.atom], to_fixed(-dEdz));
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class real> __device__ void term_dh(const DhD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int b = threadIdx.x; b < d.n; b += blockDim.x) {
        DhItem<real> it = d.items[b];
        real dx = L.px[it.i] - L.px[it.j], dy = L.py[it.i] - L.py[it.j], dz = L.pz[it.i] - L.pz[it.j];
        real r2 = dx * dx + dy * dy + dz * dz, ir = M<real>::rsqrt(r2), r = r2 * ir;
        real E = d.k * it.qq * ir * M<real>::exp(-r * d.inv_lam);
        if (ENERGY) e += (double)E;
        pair_force(L, it.i, it.j, dx, dy, dz, E * (ir + d.inv_lam) * ir);
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// ------------------------------------------------------------------ all-pairs (triangular) scans
// Rows of a strictly-upper-triangular n x n pair matrix are folded two by two (row s with row n-1-s) into
// "super-rows" of exactly n-1 pairs, so that a 64-lane wave walking a super-row stays full and the row index i is
// uniform over long lane runs (its coordinates come from LDS as a broadcast).  Waves take super-rows round-robin.
#define TRI_FOREACH(n, I, J, ...)                                                    \
    {                                                                                \
        const int _n = (n), _nw = blockDim.x >> 6, _lane = threadIdx.x & 63;         \
        const int _half = (_n + 1) >> 1;                                             \
        for (int _s = threadIdx.x >> 6; _s < _half; _s += _nw) {                     \
            const int _len0 = _n - 1 - _s;                                           \
            const int _i1 = _n - 1 - _s;                                             \
            const int _tot = (_i1 == _s) ? _len0 : _n - 1;                           \
            for (int _l = _lane; _l < _tot; _l += 64) {                              \
                int I, J;                                                            \
                if (_l < _len0) { I = _s; J = _s + 1 + _l; }                         \
                else { I = _i1; J = _i1 + 1 + (_l - _len0); }                        \
                __VA_ARGS__                                                          \
            }                                                                        \
        }                                                                            \
    }

template <bool ENERGY, class real> __device__ void term_excl(const ExclD<real>& d, const Lds<real>& L) {
    double e = 0;
    const real r02 = d.r0 * d.r0;
    for (int c = 0; c < d.nclass; c++) {
        const int base = d.class_start[c], n = d.class_start[c + 1] - base;
        TRI_FOREACH(n, i, j, {
            int a = d.atom[base + i], b = d.atom[base + j];
            real dx = L.px[a] - L.px[b], dy = L.py[a] - L.py[b], dz = L.pz[a] - L.pz[b];
            real r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < r02) {
                Int4 p = d.partners[base + i];
                if (p.x != b && p.y != b && p.z != b && p.w != b) {
                    real ir = M<real>::rsqrt(r2), dr = r2 * ir - d.r0;
                    if (ENERGY) e += (double)(d.k * dr * dr);
                    pair_force(L, a, b, dx, dy, dz, (real)-2 * d.k * dr * ir);
                }
            }
        })
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// Contact pass 1: density rho_i = sum_j theta_I(r_ij) over |resid_i - resid_j| >= 2 inside the cutoff.
template <class real> __device__ void contact_density(const ContactD<real>& d, const Lds<real>& L) {
    TRI_FOREACH(d.nsites, i, j, {
        int ri = d.resid[i] - d.resid[j];
        if (ri >= 2 || ri <= -2) {
            int a = d.atom[i], b = d.atom[j];
            real dx = L.px[a] - L.px[b], dy = L.py[a] - L.py[b], dz = L.pz[a] - L.pz[b];
            real r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < d.cutoff2) {
                real r = M<real>::sqrt(r2);
                real th = sigmoid(d.eta2 * (r - d.rmin)) * sigmoid(d.eta2 * (d.rmax - r));
                u64 f = to_fixed(th);
                if (f) { atomicAdd(&L.rho[i], f); atomicAdd(&L.rho[j], f); }
            }
        }
    })
}
// Between pass 1 and 2: convert densities, evaluate the burial wells (single-particle energy) and seed dE/drho.
template <bool ENERGY, class real> __device__ void contact_single(const ContactD<real>& d, const Lds<real>& L) {
    double e = 0;
    for (int i = threadIdx.x; i < d.nsites; i += blockDim.x) {
        real rho = (real)fixed_to_double(L.rho[i]);
        L.rhov[i] = rho;
        real dE = 0;
        if (d.has_burial) {
            int t = d.restype[i];
#pragma unroll
            for (int m = 0; m < 3; m++) {
                real gb = d.bgamma[t * 3 + m];
                // tanh(a) + tanh(b) = 2 (sig(2a) + sig(2b) - 1)
                real sa = sigmoid(d.kappa2 * (rho - d.rho_lo[m])), sb = sigmoid(d.kappa2 * (d.rho_hi[m] - rho));
                if (ENERGY) e += (double)(-d.kb * gb * (sa + sb - (real)1));
                dE += -d.kb * gb * d.kappa2 * (sa * ((real)1 - sa) - sb * ((real)1 - sb));
            }
        }
        L.drho[i] = to_fixed(dE);
        if (d.flags & 1) {
            real z = L.pz[d.atom[i]] - d.zc;
            real su = sigmoid(d.eta_sw2 * (z + d.z_m)), sv = sigmoid(d.eta_sw2 * (d.z_m - z));
            L.alpha[i] = su + sv - (real)1;
            L.dalpha[i] = d.eta_sw2 * (su * ((real)1 - su) - sv * ((real)1 - sv));
        }
    }
    flush_energy<ENERGY>(L, d.group, e);
}
// Contact pass 2: pair energy (direct + mediated), its explicit radial force, and dE/drho through sigma_water.
template <bool ENERGY, class real> __device__ void contact_pairs(const ContactD<real>& d, const Lds<real>& L) {
    double e = 0;
    const bool zdep = d.flags & 1;
    const real* g0 = L.gamma + (zdep ? 0 : 0);
    TRI_FOREACH(d.nsites, i, j, {
        int ri = d.resid[i] - d.resid[j];
        bool ok = d.chain[i] != d.chain[j] || ri >= d.min_seq_sep || -ri >= d.min_seq_sep;
        if (ok) {
            int a = d.atom[i], b = d.atom[j];
            real dx = L.px[a] - L.px[b], dy = L.py[a] - L.py[b], dz = L.pz[a] - L.pz[b];
            real r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < d.cutoff2) {
                real ir = M<real>::rsqrt(r2), r = r2 * ir;
                real s1 = sigmoid(d.eta2 * (r - d.rmin)), s2 = sigmoid(d.eta2 * (d.rmax - r));
                real s3 = sigmoid(d.eta2 * (r - d.rminII)), s4 = sigmoid(d.eta2 * (d.rmaxII - r));
                real th1 = s1 * s2, dth1 = d.eta2 * th1 * (s2 - s1);
                real th2 = s3 * s4, dth2 = d.eta2 * th2 * (s4 - s3);
                // sigma_water = 1/4 (1-tanh(eta_s(rho_i-rho0)))(1-tanh(...)) = sig(-2 eta_s (rho_i - rho0)) sig(...)
                real si = sigmoid(-d.eta_s2 * (L.rhov[i] - d.rho0)), sj = sigmoid(-d.eta_s2 * (L.rhov[j] - d.rho0));
                real sw = si * sj, dsi = -d.eta_s2 * si * ((real)1 - si), dsj = -d.eta_s2 * sj * ((real)1 - sj);
                int ty = d.restype[i] * 20 + d.restype[j];
                real E, dEdr, dEdsw;
                if (!zdep) {
                    real gd = g0[ty], gp = g0[400 + ty], gw = g0[800 + ty];
                    real med = sw * gw + ((real)1 - sw) * gp;
                    E = -d.k * (gd * th1 + th2 * med);
                    dEdr = -d.k * (gd * dth1 + dth2 * med);
                    dEdsw = -d.k * th2 * (gw - gp);
                } else {
                    const real* g1 = L.gamma + 1200;
                    real med0 = sw * g0[800 + ty] + ((real)1 - sw) * g0[400 + ty];
                    real med1 = sw * g1[800 + ty] + ((real)1 - sw) * g1[400 + ty];
                    real W = -d.k * (g0[ty] * th1 + th2 * med0), Mm = -d.k * (g1[ty] * th1 + th2 * med1);
                    real dW = -d.k * (g0[ty] * dth1 + dth2 * med0), dM = -d.k * (g1[ty] * dth1 + dth2 * med1);
                    real ai = L.alpha[i], aj = L.alpha[j], aa = ai * aj;
                    E = ((real)1 - aa) * W + aa * Mm;
                    dEdr = ((real)1 - aa) * dW + aa * dM;
                    dEdsw = -d.k * th2 * (((real)1 - aa) * (g0[800 + ty] - g0[400 + ty]) + aa * (g1[800 + ty] - g1[400 + ty]));
                    atomicAdd(&L.fz[a], to_fixed(-(Mm - W) * L.dalpha[i] * aj));
                    atomicAdd(&L.fz[b], to_fixed(-(Mm - W) * ai * L.dalpha[j]));
                }
                if (ENERGY) e += (double)E;
                atomicAdd(&L.drho[i], to_fixed(dEdsw * dsi * sj));
                atomicAdd(&L.drho[j], to_fixed(dEdsw * si * dsj));
                pair_force(L, a, b, dx, dy, dz, -dEdr * ir);
            }
        }
    })
    flush_energy<ENERGY>(L, d.group, e);
}
// Contact pass 3: chain rule through the densities: F += -(dE/drho_i + dE/drho_j) theta_I'(r) r_hat
template <class real> __device__ void contact_chain_rule(const ContactD<real>& d, const Lds<real>& L) {
    TRI_FOREACH(d.nsites, i, j, {
        int ri = d.resid[i] - d.resid[j];
        if (ri >= 2 || ri <= -2) {
            int a = d.atom[i], b = d.atom[j];
            real dx = L.px[a] - L.px[b], dy = L.py[a] - L.py[b], dz = L.pz[a] - L.pz[b];
            real r2 = dx * dx + dy * dy + dz * dz;
            if (r2 < d.cutoff2) {
                real ir = M<real>::rsqrt(r2), r = r2 * ir;
                real s1 = sigmoid(d.eta2 * (r - d.rmin)), s2 = sigmoid(d.eta2 * (d.rmax - r));
                real dth = d.eta2 * s1 * s2 * (s2 - s1);
                real c = (L.drhov[i] + L.drhov[j]) * dth;
                if (c != (real)0) pair_force(L, a, b, dx, dy, dz, -c * ir);
            }
        }
    })
}

// ------------------------------------------------------------------ acceptor x donor scans (CustomHbondForce shape)
// A wave takes acceptors round-robin (acceptor operands are wave-uniform LDS broadcasts), its lanes walk the donors.
template <bool ENERGY, class real> __device__ void term_beta(const BetaD<real>& d, const Lds<real>& L) {
    double e = 0;
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    for (int ia = threadIdx.x >> 6; ia < d.nacc; ia += nw) {
        Int4 A = d.acc[ia];
        real ox = L.px[A.x], oy = L.py[A.x], oz = L.pz[A.x];
        const real* lam_col = d.lambda + A.w;
        for (int id = lane; id < d.ndon; id += 64) {
            Int4 D = d.don[id];
            real ax = ox - L.px[D.x], ay = oy - L.py[D.x], az = oz - L.pz[D.x];
            real r1s = ax * ax + ay * ay + az * az;
            if (r1s >= d.cutoff2) continue;
            real c = lam_col[(size_t)D.w * d.nres];   // -k * lambda[don_res][acc_res]
            if (c == (real)0) continue;
            real bx = ox - L.px[D.y], by = oy - L.py[D.y], bz = oz - L.pz[D.y];
            real r2s = bx * bx + by * by + bz * bz;
            real i1 = M<real>::rsqrt(r1s), i2 = M<real>::rsqrt(r2s), g1, g2;
            real th1 = hb_theta(r1s * i1, r2s * i2, d.r_ON, d.r_OH, d.inv_sNO2, d.inv_sHO2, g1, g2);
            if (!d.two_theta) {
                if (ENERGY) e += (double)(c * th1);
                pair_force(L, A.x, D.x, ax, ay, az, -c * g1 * i1);
                pair_force(L, A.x, D.y, bx, by, bz, -c * g2 * i2);
            } else {
                real px = L.px[D.z], py = L.py[D.z], pz = L.pz[D.z];
                real cx = px - L.px[A.y], cy = py - L.py[A.y], cz = pz - L.pz[A.y];
                real ex = px - L.px[A.z], ey = py - L.py[A.z], ez = pz - L.pz[A.z];
                real r3s = cx * cx + cy * cy + cz * cz, r4s = ex * ex + ey * ey + ez * ez;
                real i3 = M<real>::rsqrt(r3s), i4 = M<real>::rsqrt(r4s), h1, h2;
                real th2 = hb_theta(r3s * i3, r4s * i4, d.r_ON, d.r_OH, d.inv_sNO2, d.inv_sHO2, h1, h2);
                if (ENERGY) e += (double)(c * th1 * th2);
                pair_force(L, A.x, D.x, ax, ay, az, -c * g1 * th2 * i1);
                pair_force(L, A.x, D.y, bx, by, bz, -c * g2 * th2 * i2);
                pair_force(L, D.z, A.y, cx, cy, cz, -c * th1 * h1 * i3);
                pair_force(L, D.z, A.z, ex, ey, ez, -c * th1 * h2 * i4);
            }
        }
    }
    flush_energy<ENERGY>(L, d.group, e);
}

template <bool ENERGY, class real> __device__ void term_pap(const PapD<real>& d, const Lds<real>& L) {
    double e = 0;
    const int nw = blockDim.x >> 6, lane = threadIdx.x & 63;
    for (int ia = threadIdx.x >> 6; ia < d.nacc; ia += nw) {
        Int4 A = d.acc[ia];
        real a1x = L.px[A.x], a1y = L.py[A.x], a1z = L.pz[A.x];
        real a2x = L.px[A.y], a2y = L.py[A.y], a2z = L.pz[A.y];
        real ex = a1x - a2x, ey = a1y - a2y, ez = a1z - a2z;
        real r3s = ex * ex + ey * ey + ez * ez, i3 = M<real>::rsqrt(r3s);
        real s3 = sigmoid(d.eta2 * (r3s * i3 - d.d_i4));   // 1/2 (1 + tanh(eta (|a1-a2| - d_i4)))
        const real* gam_col = d.gamma + A.z;
        for (int id = lane; id < d.ndon; id += 64) {
            Int4 D = d.don[id];
            real ax = a1x - L.px[D.x], ay = a1y - L.py[D.x], az = a1z - L.pz[D.x];
            real r1s = ax * ax + ay * ay + az * az;
            if (r1s >= d.cutoff2) continue;
            real c = gam_col[(size_t)D.z * d.nres];   // -k * gamma[don_res][acc_res]
            if (c == (real)0) continue;
            real bx = a2x - L.px[D.y], by = a2y - L.py[D.y], bz = a2z - L.pz[D.y];
            real r2s = bx * bx + by * by + bz * bz;
            real i1 = M<real>::rsqrt(r1s), i2 = M<real>::rsqrt(r2s);
            real s1 = sigmoid(d.eta2 * (d.r0 - r1s * i1)), s2 = sigmoid(d.eta2 * (d.r0 - r2s * i2));
            if (ENERGY) e += (double)(c * s1 * s2 * s3);
            // d s1/dr1 = -eta2 s1 (1-s1) etc.
            pair_force(L, A.x, D.x, ax, ay, az, c * d.eta2 * s1 * ((real)1 - s1) * s2 * s3 * i1);
            pair_force(L, A.y, D.y, bx, by, bz, c * s1 * d.eta2 * s2 * ((real)1 - s2) * s3 * i2);
            pair_force(L, A.x, A.y, ex, ey, ez, -c * s1 * s2 * d.eta2 * s3 * ((real)1 - s3) * i3);
        }
    }
    flush_energy<ENERGY>(L, d.group, e);
}

// ------------------------------------------------------------------ virtual sites, whole-system evaluation
template <class real> __device__ void compute_virtual_sites(const DevSys<real>& S, const Lds<real>& L) {
    for (int i = threadIdx.x; i < S.nvs; i += blockDim.x) {
        Int4 v = S.vs[i];
        real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        L.px[v.x] = w0 * L.px[v.y] + w1 * L.px[v.z] + w2 * L.px[v.w];
        L.py[v.x] = w0 * L.py[v.y] + w1 * L.py[v.z] + w2 * L.py[v.w];
        L.pz[v.x] = w0 * L.pz[v.y] + w1 * L.pz[v.z] + w2 * L.pz[v.w];
    }
}
// fold the force on each virtual site onto its three parents (ThreeParticleAverageSite chain rule)
template <class real> __device__ void redistribute_forces(const DevSys<real>& S, const Lds<real>& L) {
    for (int i = threadIdx.x; i < S.nvs; i += blockDim.x) {
        Int4 v = S.vs[i];
        real fx = (real)fixed_to_double(L.fx[v.x]), fy = (real)fixed_to_double(L.fy[v.x]), fz = (real)fixed_to_double(L.fz[v.x]);
        L.fx[v.x] = 0; L.fy[v.x] = 0; L.fz[v.x] = 0;
        real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        add_force(L, v.y, w0 * fx, w0 * fy, w0 * fz);
        add_force(L, v.z, w1 * fx, w1 * fy, w1 * fz);
        add_force(L, v.w, w2 * fx, w2 * fy, w2 * fz);
    }
}

// Evaluate every term whose group is in `mask`.  Forces accumulate into L.f* (must be zeroed by the caller and
// be visible: call after a __syncthreads()); ends with all contributions visible to the whole workgroup.
template <bool ENERGY, class real> __device__ void evaluate_terms(const DevSys<real>& S, const Lds<real>& L, unsigned mask) {
#define ON(d) ((mask >> (d).group) & 1u)
    for (int t = 0; t < S.n_harm; t++) if (ON(S.harm[t])) term_harmonic<ENERGY>(S.harm[t], L);
    for (int t = 0; t < S.n_fm; t++) if (ON(S.fm[t])) term_fm<ENERGY>(S.fm[t], L);
    for (int t = 0; t < S.n_chi; t++) if (ON(S.chi[t])) term_chi<ENERGY>(S.chi[t], L);
    for (int t = 0; t < S.n_rama; t++) if (ON(S.rama[t])) term_rama<ENERGY>(S.rama[t], L);
    for (int t = 0; t < S.n_helical; t++) if (ON(S.helical[t])) term_helical<ENERGY>(S.helical[t], L);
    for (int t = 0; t < S.n_memb; t++) if (ON(S.memb[t])) term_memb<ENERGY>(S.memb[t], L);
    for (int t = 0; t < S.n_dh; t++) if (ON(S.dh[t])) term_dh<ENERGY>(S.dh[t], L);
    for (int t = 0; t < S.n_excl; t++) if (ON(S.excl[t])) term_excl<ENERGY>(S.excl[t], L);
    for (int t = 0; t < S.n_beta; t++) if (ON(S.beta[t])) term_beta<ENERGY>(S.beta[t], L);
    for (int t = 0; t < S.n_pap; t++) if (ON(S.pap[t])) term_pap<ENERGY>(S.pap[t], L);
    // contact terms: three passes separated by workgroup barriers (uniform control flow: S and mask are uniform)
    for (int t = 0; t < S.n_contact; t++) {
        const ContactD<real>& d = S.contact[t];
        if (!ON(d)) continue;
        __syncthreads();   // previous user of the contact scratch (or previous contact term) is done
        for (int i = threadIdx.x; i < d.nsites; i += blockDim.x) L.rho[i] = 0;
        if (!(d.flags & 2)) {
            const int ng = (d.flags & 1) ? 2400 : 1200;
            const real* src = d.gamma + ((d.flags & 1) ? 0 : d.env * 1200);
            for (int i = threadIdx.x; i < ng; i += blockDim.x) L.gamma[i] = src[i];
        }
        __syncthreads();
        contact_density(d, L);
        __syncthreads();
        contact_single<ENERGY>(d, L);
        __syncthreads();
        if (!(d.flags & 2)) {
            contact_pairs<ENERGY>(d, L);
            __syncthreads();
        }
        for (int i = threadIdx.x; i < d.nsites; i += blockDim.x) L.drhov[i] = (real)fixed_to_double(L.drho[i]);
        __syncthreads();
        contact_chain_rule(d, L);
    }
#undef ON
    __syncthreads();
}

// ------------------------------------------------------------------ LDS carve-up
template <class real> __host__ __device__ inline size_t lds_bytes(int natoms, int nsites, bool with_vel) {
    size_t b = 0;
    b += (size_t)3 * natoms * sizeof(u64);            // forces
    b += (size_t)2 * nsites * sizeof(u64);            // rho, drho
    b += AWSEM_NGROUPS * sizeof(u64);                 // group energies
    b += (size_t)3 * natoms * sizeof(real);           // positions
    if (with_vel) b += (size_t)3 * natoms * sizeof(real);
    b += (size_t)4 * nsites * sizeof(real);           // rhov, drhov, alpha, dalpha
    b += (size_t)2400 * sizeof(real);                 // gamma staging
    return (b + 15) & ~(size_t)15;
}
template <class real> __device__ inline Lds<real> carve_lds(unsigned char* base, int natoms, int nsites, bool with_vel) {
    Lds<real> L;
    u64* q = (u64*)base;
    L.fx = q; q += natoms; L.fy = q; q += natoms; L.fz = q; q += natoms;
    L.rho = q; q += nsites; L.drho = q; q += nsites;
    L.eg = q; q += AWSEM_NGROUPS;
    real* r = (real*)q;
    L.px = r; r += natoms; L.py = r; r += natoms; L.pz = r; r += natoms;
    if (with_vel) { L.vx = r; r += natoms; L.vy = r; r += natoms; L.vz = r; r += natoms; }
    else { L.vx = L.vy = L.vz = nullptr; }
    L.rhov = r; r += nsites; L.drhov = r; r += nsites; L.alpha = r; r += nsites; L.dalpha = r; r += nsites;
    L.gamma = r;
    return L;
}

// replica state in global memory: SoA per replica, [replica][xyz][natoms]
template <class real> __device__ inline void load_positions(const real* g, const Lds<real>& L, int natoms) {
    for (int i = threadIdx.x; i < natoms; i += blockDim.x) { L.px[i] = g[i]; L.py[i] = g[natoms + i]; L.pz[i] = g[2 * natoms + i]; }
}

// ------------------------------------------------------------------ evaluation kernel (energies and/or forces)
template <bool ENERGY, class real>
__global__ void awsem_eval_kernel(const DevSys<real>* __restrict__ Sp, const real* __restrict__ pos, unsigned mask, int redistribute,
                                  double* __restrict__ e_out /* [R][32] or null */, double* __restrict__ f_out /* [R][natoms][3] or null */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevSys<real>& S = *Sp;
    const int natoms = S.natoms, rep = blockIdx.x;
    Lds<real> L = carve_lds<real>(smem, natoms, S.max_contact_sites, false);
    load_positions(pos + (size_t)rep * 3 * natoms, L, natoms);
    for (int i = threadIdx.x; i < natoms; i += blockDim.x) { L.fx[i] = 0; L.fy[i] = 0; L.fz[i] = 0; }
    if (threadIdx.x < AWSEM_NGROUPS) L.eg[threadIdx.x] = 0;
    __syncthreads();
    evaluate_terms<ENERGY>(S, L, mask);
    if (f_out) {
        if (redistribute) { redistribute_forces(S, L); __syncthreads(); }
        double* f = f_out + (size_t)rep * 3 * natoms;
        for (int i = threadIdx.x; i < natoms; i += blockDim.x) {
            f[3 * i] = fixed_to_double(L.fx[i]); f[3 * i + 1] = fixed_to_double(L.fy[i]); f[3 * i + 2] = fixed_to_double(L.fz[i]);
        }
    }
    if (ENERGY && e_out && threadIdx.x < AWSEM_NGROUPS) e_out[(size_t)rep * AWSEM_NGROUPS + threadIdx.x] = fixed_to_double(L.eg[threadIdx.x]);
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
__global__ void awsem_maxwell_kernel(const DevSys<real>* __restrict__ Sp, real* __restrict__ vel, const double* __restrict__ T,
                                     unsigned long long seed) {
    const DevSys<real>& S = *Sp;
    const int natoms = S.natoms, rep = blockIdx.x;
    real* v = vel + (size_t)rep * 3 * natoms;
    for (int i = threadIdx.x; i < natoms; i += blockDim.x) {
        real g[3] = {0, 0, 0}, s = 0;
        if (S.inv_mass[i] != (real)0) {
            gauss3<real>(seed, AWSEM_STREAM_VEL, rep, 0, i, g);
            s = (real)::sqrt(AWSEM_KB * T[rep]) * S.sqrt_inv_mass[i];
        }
        v[i] = s * g[0]; v[natoms + i] = s * g[1]; v[2 * natoms + i] = s * g[2];
    }
}

template <class real>
__global__ void awsem_vsite_kernel(const DevSys<real>* __restrict__ Sp, real* __restrict__ pos) {
    const DevSys<real>& S = *Sp;
    const int natoms = S.natoms;
    real* p = pos + (size_t)blockIdx.x * 3 * natoms;
    for (int i = threadIdx.x; i < S.nvs; i += blockDim.x) {
        Int4 v = S.vs[i];
        real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        for (int k = 0; k < 3; k++) p[k * natoms + v.x] = w0 * p[k * natoms + v.y] + w1 * p[k * natoms + v.z] + w2 * p[k * natoms + v.w];
    }
}

// ------------------------------------------------------------------ fused multi-step Langevin kernel
// OpenMM LangevinIntegrator update (see oracle/awsem_oracle.c: awsem_oracle_step) with the state LDS-resident:
// per step: virtual sites -> forces (all groups) -> virtual-site force redistribution -> v, x update.
template <class real>
__global__ void awsem_step_kernel(const DevSys<real>* __restrict__ Sp, real* __restrict__ pos, real* __restrict__ vel,
                                  const double* __restrict__ T, int nsteps, double dt, double friction, unsigned long long seed,
                                  long long step0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const DevSys<real>& S = *Sp;
    const int natoms = S.natoms, rep = blockIdx.x;
    Lds<real> L = carve_lds<real>(smem, natoms, S.max_contact_sites, true);
    real* gp = pos + (size_t)rep * 3 * natoms;
    real* gv = vel + (size_t)rep * 3 * natoms;
    load_positions(gp, L, natoms);
    for (int i = threadIdx.x; i < natoms; i += blockDim.x) { L.vx[i] = gv[i]; L.vy[i] = gv[natoms + i]; L.vz[i] = gv[2 * natoms + i]; }
    const double a_d = ::exp(-friction * dt);
    const real a = (real)a_d, fs = (real)(friction == 0.0 ? dt : (1.0 - a_d) / friction);
    const real ns = (real)::sqrt(AWSEM_KB * T[rep] * (1.0 - a_d * a_d)), dtr = (real)dt;
    __syncthreads();
    for (int s = 0; s < nsteps; s++) {
        compute_virtual_sites(S, L);
        for (int i = threadIdx.x; i < natoms; i += blockDim.x) { L.fx[i] = 0; L.fy[i] = 0; L.fz[i] = 0; }
        __syncthreads();
        evaluate_terms<false>(S, L, 0xffffffffu);
        redistribute_forces(S, L);
        __syncthreads();
        for (int i = threadIdx.x; i < natoms; i += blockDim.x) {
            real im = S.inv_mass[i];
            if (im == (real)0) continue;
            real g[3];
            gauss3<real>(seed, AWSEM_STREAM_LANGEVIN, rep, step0 + s, i, g);
            real sq = S.sqrt_inv_mass[i];
            real fx = (real)fixed_to_double(L.fx[i]), fy = (real)fixed_to_double(L.fy[i]), fz = (real)fixed_to_double(L.fz[i]);
            real vx = a * L.vx[i] + fs * im * fx + ns * sq * g[0];
            real vy = a * L.vy[i] + fs * im * fy + ns * sq * g[1];
            real vz = a * L.vz[i] + fs * im * fz + ns * sq * g[2];
            L.vx[i] = vx; L.vy[i] = vy; L.vz[i] = vz;
            L.px[i] += dtr * vx; L.py[i] += dtr * vy; L.pz[i] += dtr * vz;
        }
        __syncthreads();
    }
    compute_virtual_sites(S, L);
    __syncthreads();
    for (int i = threadIdx.x; i < natoms; i += blockDim.x) {
        gp[i] = L.px[i]; gp[natoms + i] = L.py[i]; gp[2 * natoms + i] = L.pz[i];
        gv[i] = L.vx[i]; gv[natoms + i] = L.vy[i]; gv[2 * natoms + i] = L.vz[i];
    }
}
