// awsem_cell.h -- cell-list path: systems whose replica does not fit the LDS of one CU (tens of thousands of residues).
//
// State lives in HBM (SoA positions / velocities, int64 fixed-point force accumulators); one evaluation is a short
// sequence of grid-wide kernels:
//   [when an atom has moved more than skin/2 since the last build]
//     bounding box -> cell grid (edge >= cutoff + skin) -> counting sort of the atoms into cells ->
//     neighbour lists of the contact and the excluded-volume term (27-cell scan per atom, wave-compacted appends)
//   zero accumulators -> bonded terms (one lane per item) -> contact density -> per-site pass -> pair pass ->
//   chain-rule pass -> excluded volume -> [virtual-site force redistribution -> Langevin update]
// The per-pair arithmetic is the same code as on the replica-resident path (the *_body functions of awsem_kernels.h
// work on an `Lds` view, here filled with global pointers), so both paths compute the same contributions;
// all sums are fixed point, so the result does not depend on the (non-deterministic) order of the list entries.
#pragma once
#include "awsem_kernels.h"

#define AWSEM_CELL_MAXDIM 64
#define AWSEM_CELL_MAXCELLS (AWSEM_CELL_MAXDIM * AWSEM_CELL_MAXDIM * AWSEM_CELL_MAXDIM)

template <class real> struct CellSys {
    int natoms, nrep, nsites;   // nsites: sites of the contact term (0 if none)
    // static roles of the atoms
    const int* c_site;          // [natoms] site index in contact term 0, or -1
    const int* x_slot;          // [natoms] slot in the class-sorted table of excl term 0 (ExclD::atom / partners), or -1
    const int* x_cls;           // [natoms] excluded-volume class, or -1
    // cell grid, per replica
    int* bbox;                  // [R][8]  order-preserving int images of min xyz (0..2) and max xyz (4..6)
    real* grid;                 // [R][4]  origin xyz, 1/edge
    int* dims;                  // [R][4]  nx, ny, nz, ncells
    int* cell_count;            // [R][MAXCELLS]
    int* cell_start;            // [R][MAXCELLS + 1]
    int* cell_of;               // [R][natoms]
    int* cell_atoms;            // [R][natoms] atom ids sorted by cell
    // neighbour lists, per replica
    u64* clist;                 // contact: site_i | site_j << 24 | type pair << 48 | in_density << 57 | has_pair_energy << 58
    u64* xlist;                 // excl: atom_a | atom_b << 32
    unsigned* counts;           // [R][2] entries in clist, xlist
    long long ccap, xcap;
    // accumulators, per replica
    u64* f;                     // [R][3][natoms]
    u64* eg;                    // [R][32]
    u64 *rho, *drho;            // [R][nsites]
    real *rhov, *alpha, *dalpha;
    // list validity
    real* ref;                  // [R][3][natoms] positions at the last build
    unsigned* flag;             // [R][2]: [0] != 0 -> an atom moved further than skin/2, [1] != 0 -> rebuild in progress
    real rc_cell;               // minimum cell edge = largest (cutoff + skin)
    real skin_half2;
    int* status;                // bit 1: neighbour list overflow
};

__device__ __forceinline__ int ordered_int(float f) { int o = __float_as_int(f); return o >= 0 ? o : o ^ 0x7fffffff; }
__device__ __forceinline__ float ordered_float(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

// View of replica `rep` in global memory (same API as the LDS views of awsem_kernels.h)
template <class real> struct CellView : ViewCommon<real>, SoaOps<real> { typedef real real_t; };
template <class real> __device__ inline CellView<real> cell_view(const DevSys<real>& S, const CellSys<real>& C, const real* pos, int rep) {
    CellView<real> L;
    const size_t n = C.natoms;
    real* p = const_cast<real*>(pos) + (size_t)rep * 3 * n;
    L.px = p; L.py = p + n; L.pz = p + 2 * n;
    u64* f = C.f + (size_t)rep * 3 * n;
    L.fx = f; L.fy = f + n; L.fz = f + 2 * n;
    const size_t so = (size_t)rep * C.nsites;
    L.rho = C.rho + so; L.drho = C.drho + so; L.rhov = C.rhov + so; L.alpha = C.alpha + so; L.dalpha = C.dalpha + so;
    L.gamma = nullptr;
    if (S.n_contact > 0) {
        const ContactD<real>& d = S.contact[0];
        L.gamma = const_cast<real*>(d.gamma) + ((d.flags & 1) ? 0 : d.env * 1200);
        L.satom = const_cast<int*>(d.atom);
    } else L.satom = nullptr;
    L.eg = C.eg + (size_t)rep * AWSEM_NGROUPS;
    L.lcount = nullptr; L.flag = nullptr; L.lists = nullptr;
    return L;
}

// ------------------------------------------------------------------ list validity and cell grid
// grid (blocks, R).  Largest squared displacement since the last build; `force` marks the lists stale regardless.
template <class real> __global__ void cell_displacement_kernel(CellSys<real> C, const real* __restrict__ pos, int force) {
    const int rep = blockIdx.y, n = C.natoms;
    const real* p = pos + (size_t)rep * 3 * n;
    const real* q = C.ref + (size_t)rep * 3 * n;
    bool moved = force != 0;
    if (!force)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
            moved |= (float)dist2(p[i] - q[i], p[n + i] - q[n + i], p[2 * n + i] - q[2 * n + i]) > (float)C.skin_half2;
    if (__ballot(moved) && (threadIdx.x & 63) == 0) C.flag[2 * rep] = 1u;   // same value from every writer: no atomic needed
}
// one thread per replica: decide, and reset what a rebuild accumulates into
template <class real> __global__ void cell_decide_kernel(CellSys<real> C) {
    const int rep = blockIdx.x * blockDim.x + threadIdx.x;
    if (rep >= C.nrep) return;
    const bool rebuild = C.flag[2 * rep] != 0;
    C.flag[2 * rep + 1] = rebuild ? 1u : 0u;
    C.flag[2 * rep] = 0;
    if (rebuild) {
        for (int k = 0; k < 3; k++) { C.bbox[8 * rep + k] = 0x7fffffff; C.bbox[8 * rep + 4 + k] = (int)0x80000000; }
        C.counts[2 * rep] = 0; C.counts[2 * rep + 1] = 0;
    }
}
template <class real> __global__ void cell_bbox_kernel(CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y, n = C.natoms;
    if (!C.flag[2 * rep + 1]) return;
    const real* p = pos + (size_t)rep * 3 * n;
    real* q = C.ref + (size_t)rep * 3 * n;
    __shared__ int slo[3], shi[3];
    if (threadIdx.x < 3) { slo[threadIdx.x] = 0x7fffffff; shi[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        for (int k = 0; k < 3; k++) {
            const real x = p[k * n + i];
            q[k * n + i] = x;   // new reference positions
            lo[k] = fminf(lo[k], (float)x); hi[k] = fmaxf(hi[k], (float)x);
        }
    for (int k = 0; k < 3; k++) {
        for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_down(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], ordered_int(lo[k])); atomicMax(&shi[k], ordered_int(hi[k])); }
    }
    __syncthreads();
    if (threadIdx.x < 3) { atomicMin(&C.bbox[8 * rep + threadIdx.x], slo[threadIdx.x]); atomicMax(&C.bbox[8 * rep + 4 + threadIdx.x], shi[threadIdx.x]); }
}
// grid (ceil(MAXCELLS / block), R): grid geometry (every thread derives the same numbers) and zeroed cell counts
template <class real> __global__ void cell_grid_kernel(CellSys<real> C) {
    const int rep = blockIdx.y;
    if (!C.flag[2 * rep + 1]) return;
    float lo[3], ext = 0;
    for (int k = 0; k < 3; k++) {
        lo[k] = ordered_float(C.bbox[8 * rep + k]) - 1e-3f;
        ext = fmaxf(ext, ordered_float(C.bbox[8 * rep + 4 + k]) + 1e-3f - lo[k]);
    }
    const float edge = fmaxf((float)C.rc_cell, ext / (float)AWSEM_CELL_MAXDIM * 1.0001f);
    int nd[3];
    for (int k = 0; k < 3; k++) {
        nd[k] = (int)floorf((ordered_float(C.bbox[8 * rep + 4 + k]) + 1e-3f - lo[k]) / edge) + 1;
        nd[k] = nd[k] > AWSEM_CELL_MAXDIM ? AWSEM_CELL_MAXDIM : nd[k];
    }
    const int ncell = nd[0] * nd[1] * nd[2], c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c == 0) {
        for (int k = 0; k < 3; k++) { C.grid[4 * rep + k] = (real)lo[k]; C.dims[4 * rep + k] = nd[k]; }
        C.grid[4 * rep + 3] = (real)(1.0f / edge);
        C.dims[4 * rep + 3] = ncell;
    }
    if (c < ncell) C.cell_count[(size_t)rep * AWSEM_CELL_MAXCELLS + c] = 0;
}
template <class real> __device__ __forceinline__ void cell_coords(const CellSys<real>& C, int rep, real x, real y, real z, int c[3]) {
    const real inv = C.grid[4 * rep + 3];
    const real v[3] = {x, y, z};
    for (int k = 0; k < 3; k++) {
        int t = (int)M<real>::floor((v[k] - C.grid[4 * rep + k]) * inv);
        const int nk = C.dims[4 * rep + k];
        c[k] = t < 0 ? 0 : (t >= nk ? nk - 1 : t);
    }
}
template <class real> __global__ void cell_count_kernel(CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y, n = C.natoms;
    if (!C.flag[2 * rep + 1]) return;
    const real* p = pos + (size_t)rep * 3 * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (C.c_site[i] < 0 && C.x_cls[i] < 0) { C.cell_of[(size_t)rep * n + i] = -1; continue; }   // only atoms that are a site of a listed term
        int c[3];
        cell_coords(C, rep, p[i], p[n + i], p[2 * n + i], c);
        const int cell = (c[2] * C.dims[4 * rep + 1] + c[1]) * C.dims[4 * rep] + c[0];
        C.cell_of[(size_t)rep * n + i] = cell;
        atomicAdd(&C.cell_count[(size_t)rep * AWSEM_CELL_MAXCELLS + cell], 1);
    }
}
// one 1024-thread workgroup per replica: exclusive scan of the cell counts; the counts are zeroed for the fill pass
template <class real> __global__ void cell_scan_kernel(CellSys<real> C) {
    const int rep = blockIdx.x;
    if (!C.flag[2 * rep + 1]) return;
    __shared__ int part[1024];
    const int ncell = C.dims[4 * rep + 3], per = (ncell + 1023) / 1024;
    int* cnt = C.cell_count + (size_t)rep * AWSEM_CELL_MAXCELLS;
    int* start = C.cell_start + (size_t)rep * (AWSEM_CELL_MAXCELLS + 1);
    const int b = threadIdx.x * per, e = min(b + per, ncell);
    int s = 0;
    for (int c = b; c < e; c++) s += cnt[c];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partial sums
        int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = threadIdx.x ? part[threadIdx.x - 1] : 0;
    for (int c = b; c < e; c++) { start[c] = run; run += cnt[c]; cnt[c] = 0; }
    if (threadIdx.x == 1023) start[ncell] = part[1023];
}
template <class real> __global__ void cell_fill_kernel(CellSys<real> C) {
    const int rep = blockIdx.y, n = C.natoms;
    if (!C.flag[2 * rep + 1]) return;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int cell = C.cell_of[(size_t)rep * n + i];
        if (cell < 0) continue;
        const int slot = C.cell_start[(size_t)rep * (AWSEM_CELL_MAXCELLS + 1) + cell] + atomicAdd(&C.cell_count[(size_t)rep * AWSEM_CELL_MAXCELLS + cell], 1);
        C.cell_atoms[(size_t)rep * n + slot] = i;
    }
}

// wave-compacted append to a global list
__device__ __forceinline__ void cell_append(u64* __restrict__ e, unsigned* cnt, long long cap, int* status, bool pass, u64 entry) {
    const u64 m = __ballot(pass);
    if (m == 0) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
    unsigned base = 0;
    if (lane == leader) base = atomicAdd(cnt, (unsigned)__popcll(m));
    base = (unsigned)__builtin_amdgcn_readlane((int)base, leader);
    const unsigned off = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
    if (pass) {
        if ((long long)base + off < cap) e[base + off] = entry;
        else atomicOr(status, 2);
    }
}

// One lane per binned atom, in cell order: scan the HALF shell of cells around it (the rest of its own cell and of
// the x-row behind it, then the 4 rows of 3 cells "above"), so that every pair is met exactly once.  Both lists in
// one sweep.  rc2c / rc2x: (cutoff + skin)^2 of the contact / excluded-volume term (<= 0 disables that list).
template <class real> __global__ void cell_list_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos, real rc2c, real rc2x) {
    const int rep = blockIdx.y, n = C.natoms;
    if (!C.flag[2 * rep + 1]) return;
    const real* p = pos + (size_t)rep * 3 * n;
    const int* catoms = C.cell_atoms + (size_t)rep * n;
    const int* cstart = C.cell_start + (size_t)rep * (AWSEM_CELL_MAXCELLS + 1);
    u64* cl = C.clist + (size_t)rep * C.ccap;
    u64* xl = C.xlist + (size_t)rep * C.xcap;
    const int nx = C.dims[4 * rep], ny = C.dims[4 * rep + 1], nz = C.dims[4 * rep + 2];
    const int nb = cstart[C.dims[4 * rep + 3]];   // binned atoms
    const int ntot = (nb + 63) & ~63;             // whole waves: the appends are wave collectives
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ntot; k += gridDim.x * blockDim.x) {
        const bool vi = k < nb;
        const int i = vi ? catoms[k] : 0;
        const real xi = p[i], yi = p[n + i], zi = p[2 * n + i];
        const int si = vi && rc2c > (real)0 ? C.c_site[i] : -1;
        const int xi_cls = vi && rc2x > (real)0 ? C.x_cls[i] : -1;
        int c[3] = {0, 0, 0};
        if (vi) cell_coords(C, rep, xi, yi, zi, c);
        int info_i = 0;
        Int4 part = {-1, -1, -1, -1};
        if (si >= 0) info_i = S.contact[0].info[si];
        if (xi_cls >= 0) part = S.excl[0].partners[C.x_slot[i]];
        for (int row = 0; row < 5; row++) {
            // row 0: own row, from the entry after this atom to the end of cell x+1; rows 1..4: (dy,dz) = (1,0), (-1,1), (0,1), (1,1)
            const int dy = row == 0 ? 0 : (row == 1 ? 1 : row - 3), dz = row < 2 ? 0 : 1;
            const int cz = c[2] + dz, cy = c[1] + dy;
            const bool vrow = vi && (si >= 0 || xi_cls >= 0) && cz < nz && cy >= 0 && cy < ny;
            const int x0 = max(c[0] - 1, 0), x1 = min(c[0] + 1, nx - 1);
            const int base = (cz * ny + cy) * nx;
            int q = vrow ? (row == 0 ? k + 1 : cstart[base + x0]) : 0;
            const int qe = vrow ? cstart[base + x1 + 1] : 0;
            while (__ballot(q < qe)) {
                const bool vq = q < qe;
                const int j = vq ? catoms[q] : i;
                q++;
                const real r2 = dist2(xi - p[j], yi - p[n + j], zi - p[2 * n + j]);
                bool pc = false;
                u64 ec = 0;
                if (vq && si >= 0 && r2 < rc2c) {
                    const int sj = C.c_site[j];
                    if (sj >= 0) {
                        const ContactD<real>& d = S.contact[0];
                        const int info_j = d.info[sj], dres = (info_j & 0xffff) - (info_i & 0xffff);
                        const bool in_rho = dres >= 2 || dres <= -2;
                        const bool pair_e = !(d.flags & 2) && ((((info_i ^ info_j) >> 16) & 0x7ff) != 0 || dres >= d.min_seq_sep || -dres >= d.min_seq_sep);
                        pc = in_rho || pair_e;
                        const u64 ty = (u64)(((info_i >> 27) & 31) * 20 + ((info_j >> 27) & 31));
                        ec = (u64)si | ((u64)sj << 24) | (ty << 48) | ((u64)in_rho << 57) | ((u64)pair_e << 58);
                    }
                }
                cell_append(cl, &C.counts[2 * rep], C.ccap, C.status, pc, ec);
                const bool px = vq && xi_cls >= 0 && r2 < rc2x && C.x_cls[j] == xi_cls &&
                                part.x != j && part.y != j && part.z != j && part.w != j;
                cell_append(xl, &C.counts[2 * rep + 1], C.xcap, C.status, px, (u64)(unsigned)i | ((u64)(unsigned)j << 32));
            }
        }
    }
}

// ------------------------------------------------------------------ evaluation from the lists (grid (blocks, R))
template <class real> __global__ void cell_density_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ContactD<real>& d = S.contact[0];
    const u64* e = C.clist + (size_t)rep * C.ccap;
    const int count = (int)min((long long)C.counts[2 * rep], C.ccap), ntot = (count + 63) & ~63;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ntot; k += gridDim.x * blockDim.x) {
        const bool v = k < count;
        const u64 ent = v ? e[k] : 0;
        const int i = (int)(ent & 0xffffff), j = (int)((ent >> 24) & 0xffffff);
        density_body(d, L, i, j, L.satom[i], L.satom[j], v && ((ent >> 57) & 1));
    }
}
template <bool ENERGY, class real> __global__ void cell_single_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ContactD<real>& d = S.contact[0];
    Energy e;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.nsites; i += gridDim.x * blockDim.x) single_body<ENERGY>(d, L, i, L.satom[i], e);
    flush_energy<ENERGY>(L, d.group, e);
}
template <bool ENERGY, class real> __global__ void cell_pairs_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ContactD<real>& d = S.contact[0];
    const u64* e = C.clist + (size_t)rep * C.ccap;
    const int count = (int)min((long long)C.counts[2 * rep], C.ccap), ntot = (count + 63) & ~63;
    Energy en;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ntot; k += gridDim.x * blockDim.x) {
        const bool v = k < count;
        const u64 ent = v ? e[k] : 0;
        const int i = (int)(ent & 0xffffff), j = (int)((ent >> 24) & 0xffffff), ty = (int)((ent >> 48) & 0x1ff);
        pairs_body<ENERGY>(d, L, i, j, ty, L.satom[i], L.satom[j], v && ((ent >> 58) & 1), en);
    }
    flush_energy<ENERGY>(L, d.group, en);
}
template <class real> __global__ void cell_chain_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ContactD<real>& d = S.contact[0];
    const u64* e = C.clist + (size_t)rep * C.ccap;
    const int count = (int)min((long long)C.counts[2 * rep], C.ccap), ntot = (count + 63) & ~63;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ntot; k += gridDim.x * blockDim.x) {
        const bool v = k < count;
        const u64 ent = v ? e[k] : 0;
        const int i = (int)(ent & 0xffffff), j = (int)((ent >> 24) & 0xffffff);
        chain_body(d, L, i, j, L.satom[i], L.satom[j], v && ((ent >> 57) & 1));
    }
}
template <bool ENERGY, class real> __global__ void cell_excl_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ExclD<real>& d = S.excl[0];
    const u64* e = C.xlist + (size_t)rep * C.xcap;
    const int count = (int)min((long long)C.counts[2 * rep + 1], C.xcap), ntot = (count + 63) & ~63;
    Energy en;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ntot; k += gridDim.x * blockDim.x) {
        const bool v = k < count;
        const u64 ent = v ? e[k] : 0;
        excl_body<ENERGY>(d, L, (int)(ent & 0xffffffffu), (int)(ent >> 32), v, en);
    }
    flush_energy<ENERGY>(L, d.group, en);
}

// Bonded terms: one workgroup per 256 items of one term (job table built at finalize).
struct CellJob { int kind, term, first; };
enum { CJ_HARM, CJ_CHI, CJ_RAMA, CJ_HELICAL, CJ_FM, CJ_MEMB, CJ_DH };
template <bool ENERGY, class real> __global__ void cell_bonded_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos, const CellJob* __restrict__ jobs, unsigned mask) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const CellJob job = jobs[blockIdx.x];
    const int f = job.first;
    switch (job.kind) {
    case CJ_HARM: { HarmD<real> d = S.harm[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_harmonic<ENERGY>(d, L); } break;
    case CJ_CHI: { ChiD<real> d = S.chi[job.term]; if (!AWSEM_ON(d)) return; d.atoms += f; d.n = min(256, d.n - f); term_chi<ENERGY>(d, L); } break;
    case CJ_RAMA: { RamaD<real> d = S.rama[job.term]; if (!AWSEM_ON(d)) return; d.atoms += f; d.bw += (size_t)f * d.nwell; d.n = min(256, d.n - f); term_rama<ENERGY>(d, L); } break;
    case CJ_HELICAL: { HelicalD<real> d = S.helical[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_helical<ENERGY>(d, L); } break;
    case CJ_FM: { FmD<real> d = S.fm[job.term]; if (!AWSEM_ON(d)) return; d.pairs2 += 2 * (size_t)f; d.table2 += (size_t)f * (d.ntab - 1); d.n = min(256, d.n - f); term_fm<ENERGY, false>(d, L); } break;
    case CJ_MEMB: { MembD<real> d = S.memb[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_memb<ENERGY>(d, L); } break;
    case CJ_DH: { DhD<real> d = S.dh[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_dh<ENERGY>(d, L); } break;
    }
}

// ------------------------------------------------------------------ outputs, virtual sites, integrator
template <class real> __global__ void cell_vsite_kernel(DevSys<real> S, real* __restrict__ pos) {
    const int n = S.natoms;
    real* p = pos + (size_t)blockIdx.y * 3 * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S.nvs; i += gridDim.x * blockDim.x) {
        const Int4 v = S.vs[i];
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        for (int k = 0; k < 3; k++) p[k * n + v.x] = w0 * p[k * n + v.y] + w1 * p[k * n + v.z] + w2 * p[k * n + v.w];
    }
}
template <class real> __global__ void cell_redistribute_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S.nvs; i += gridDim.x * blockDim.x) {
        const Int4 v = S.vs[i];
        const R3<real> F = L.force(v.x);
        const real fx = F.x, fy = F.y, fz = F.z;
        L.zero_force(v.x);   // only this lane touches the accumulators of its virtual site here
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        L.add_force(v.y, w0 * fx, w0 * fy, w0 * fz);
        L.add_force(v.z, w1 * fx, w1 * fy, w1 * fz);
        L.add_force(v.w, w2 * fx, w2 * fy, w2 * fz);
    }
}
template <class real> __global__ void cell_output_kernel(CellSys<real> C, double* __restrict__ f_out, double* __restrict__ e_out) {
    const int rep = blockIdx.y, n = C.natoms;
    const u64* f = C.f + (size_t)rep * 3 * n;
    if (f_out)
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
            for (int k = 0; k < 3; k++) f_out[((size_t)rep * n + i) * 3 + k] = fixed_to_double(f[(size_t)k * n + i]);
    if (e_out && blockIdx.x == 0 && threadIdx.x < AWSEM_NGROUPS) e_out[(size_t)rep * AWSEM_NGROUPS + threadIdx.x] = fixed_to_double(C.eg[(size_t)rep * AWSEM_NGROUPS + threadIdx.x]);
}
template <class real>
__global__ void cell_integrate_kernel(DevSys<real> S, CellSys<real> C, real* __restrict__ pos, real* __restrict__ vel, const double* __restrict__ T,
                                      double dt, double friction, unsigned long long seed, long long step) {
    const int rep = blockIdx.y, n = C.natoms;
    real* p = pos + (size_t)rep * 3 * n;
    real* v = vel + (size_t)rep * 3 * n;
    const u64* f = C.f + (size_t)rep * 3 * n;
    const double a_d = ::exp(-friction * dt);
    const real a = (real)a_d, fs = (real)(friction == 0.0 ? dt : (1.0 - a_d) / friction);
    const real ns = (real)::sqrt(AWSEM_KB * T[rep] * (1.0 - a_d * a_d)), dtr = (real)dt;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S.nmassive; k += gridDim.x * blockDim.x) {
        const MassiveItem<real> mi = S.massive[k];
        const int i = mi.atom;
        real g[3];
        gauss3<real>(seed, AWSEM_STREAM_LANGEVIN, rep, step, i, g);
        for (int c = 0; c < 3; c++) {
            const real vn = a * v[c * n + i] + fs * mi.inv_mass * (real)fixed_to_double(f[(size_t)c * n + i]) + ns * mi.sqrt_inv_mass * g[c];
            v[c * n + i] = vn;
            p[c * n + i] += dtr * vn;
        }
    }
}
