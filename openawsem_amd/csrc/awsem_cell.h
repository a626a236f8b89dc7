// awsem_cell.h -- cell-list path: systems whose replica does not fit the LDS of one CU (tens of thousands of residues).
//
// State lives in HBM (SoA positions / velocities, int64 fixed-point force accumulators).  The pair terms are
// evaluated OWNER-COMPUTES from FULL neighbour lists: every contact site (excluded-volume atom) owns one row of a
// fixed-stride list holding all its partners, a group of 16 lanes (one DPP row of the wavefront) walks that row,
// reduces its partial sums across the 16 lanes and the row leader adds the result to the owner's accumulator with a
// plain read-modify-write -- no atomics in any pair pass, at the price of meeting every pair from both sides.
// One MD step is a short fixed sequence of grid-wide kernels on the handle's stream:
//   pre (virtual sites, displacement test, bounding box, packed site records)
//   [grid -> count -> scan -> fill -> neighbour rows -> hydrogen-bond rows: all exit at once unless an atom moved more than skin/2]
//   bonded terms -> density + per-site pass + excluded volume -> pair pass -> chain-rule pass ->
//   hydrogen-bond terms (per-acceptor donor rows, one launch for all beta / P-AP terms; global fixed-point atomics) ->
//   integrate (gathers the virtual-site forces of each massive particle, Langevin update)
// The per-pair arithmetic is the same code as on the replica-resident path (awsem_kernels.h); all sums are fixed
// point, so the result does not depend on the (non-deterministic) order of the list entries.
#pragma once
#include "awsem_kernels.h"

#define AWSEM_CELL_MAXDIM 64
#define AWSEM_CELL_MAXCELLS (AWSEM_CELL_MAXDIM * AWSEM_CELL_MAXDIM * AWSEM_CELL_MAXDIM)
#define AWSEM_ROW 16   // lanes that share one list row
#define AWSEM_ROW_X 4   // ... of the excluded-volume bin: an atom has about ten partners inside 0.35 nm + skin, sixteen lanes idled on them (round 3)
#define AWSEM_WIN 256        // atoms per window of the bonded-term kernel
#define AWSEM_WIN_SPAN 384   // atoms a window's LDS holds: its own and the next half window (items reach forward along the chain)

// One binning = the members of one listed term (bin 0: contact sites, bin 1: excluded-volume atoms) sorted into a
// uniform grid whose cell edge is at least that term's cutoff + skin.
template <class real> struct CellBin {
    int n;                      // members
    const int* atom;            // [n] member -> atom
    real edge_min;              // cutoff + skin
    real rc2;                   // (cutoff + skin)^2 of the list
    int stride;                 // list row length (multiple of AWSEM_ROW)
    // per replica
    real* grid;                 // [R][4]  origin xyz, 1/edge
    int* dims;                  // [R][4]  nx, ny, nz, ncells
    int* cell_count;            // [R][MAXCELLS]
    int* cell_start;            // [R][MAXCELLS + 1]
    int* tile_sum;              // [R][MAXCELLS / 1024] members per tile of 1024 cells
    int* cell_of;               // [R][n]
    int* sorted;                // [R][n] member ids in cell order
    real* rec;                  // [R][n][4] packed member records: x, y, z, tag (bits)
    unsigned* rows;             // [R][n][stride] neighbour rows (member ids, flags in the top bits); row k belongs to member sorted[k] (round 3: the
                                // walks read rows and counts by their own loop index, one dependent load fewer than through the member id)
    int* row_count;             // [R][n], in the same order
};

template <class real> struct CellSys {
    int natoms, nrep, nsites;   // nsites: sites of the contact term (0 if none)
    CellBin<real> bin[2];
    const Int4* x_partners;     // [bin[1].n] excluded partner atoms of each excluded-volume member
    const int* x_class;         // [bin[1].n] excluded-volume class of each member
    int* bbox;                  // [R][nblk_pre][8] per-workgroup bounding boxes of the pre kernel: order-preserving int images of min xyz (0..2), max xyz (4..6)
    int nblk_pre;
    unsigned* flag;             // [R][2]   != 0 -> a member moved further than skin/2 since the last build, double buffered
    // accumulators, per replica
    u64* f;                     // [R][3][natoms]
    u64* eg;                    // [R][32]
    u64* cv;                    // [R][n_cv] sums of the collective-variable terms (2^-48 fixed point); cleared by the pre kernel
    int n_cv;
    u64 *rho, *drho;            // [R][nsites]
    real *rhov, *alpha, *dalpha;
    real* ref;                  // [R][3][natoms] positions at the last build
    real skin_half2;
    int* status;                // bit 1: neighbour row overflow
    u64* cm;                    // [2][R][nblk_int][3] per-workgroup momentum sums of the integrator (Q31.32), by step parity (CMMotionRemover)
    int nblk_int;               // workgroups per replica of the integrator's grid
    // round 3: the step as five launches.  The integrator also does the next step's "pre" work (member records,
    // displacement flag, bounding box), the window kernel places the virtual sites it stages, and ONE launch rebuilds the lists.
    const Int4* pre_items;      // [nmassive] per massive particle: member of bin 0 (-1: none), member of bin 1, the bin-0 record's tag, -
    const int* vs_of;           // [natoms] atom -> index into DevSys::vs, -1 if not a virtual site
    int nblk_box;               // bounding boxes kept per replica: max(nblk_pre, nblk_int); unused ones stay neutral
    unsigned* bar;              // [R][2][4] arrival counters of the rebuild kernel's grid barriers, per (replica, bin)
};

__device__ __forceinline__ int ordered_int(float f) { int o = __float_as_int(f); return o >= 0 ? o : o ^ 0x7fffffff; }
__device__ __forceinline__ float ordered_float(int o) { return __int_as_float(o >= 0 ? o : o ^ 0x7fffffff); }

// View of replica `rep` in global memory (same API as the LDS views of awsem_kernels.h)
template <class real> struct CellView : ViewCommon<real>, SoaOps<real> { typedef real real_t; };
template <class real> __device__ inline CellView<real> cell_view(const DevSys<real>& S, const CellSys<real>& C, const real* pos, int rep) {
    CellView<real> L;
    L.nthr = blockDim.x; L.ramap = nullptr;
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
    L.cv = C.cv ? C.cv + (size_t)rep * C.n_cv : nullptr;
    L.lcount = nullptr; L.flag = nullptr; L.lists = nullptr;
    return L;
}

template <class real> struct alignas(4 * sizeof(real)) Rec { real x, y, z, tag; };
__device__ __forceinline__ int tag_bits(float t) { return __float_as_int(t); }
__device__ __forceinline__ int tag_bits(double t) { return (int)__double_as_longlong(t); }
__device__ __forceinline__ float make_tag(int b, float) { return __int_as_float(b); }
__device__ __forceinline__ double make_tag(int b, double) { return __longlong_as_double((long long)(unsigned)b); }

// sum over the 16 lanes of a row; every lane of the row receives the total
template <int ROW = AWSEM_ROW> __device__ __forceinline__ u64 row_sum(u64 v) {
#pragma unroll
    for (int off = 1; off < ROW; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

// ------------------------------------------------------------------ pre kernel (every step / evaluation)
// grid (blocks, R), one pass.  (1) STEP only: virtual sites placed, their accumulators cleared for the coming
// evaluation, as are those of massless particles that are no sites; (2) for the list members: has one moved further than skin/2
// since the last build, and the bounding box; (3) packed member records.  `cur` selects the flag this step
// writes and the rebuild kernels read; the other one is reset for the next step.
// List members are massive particles (checked at finalize), so (1) and (2) touch different atoms.
template <bool STEP, class real> __global__ void cell_pre_kernel(DevSys<real> S, CellSys<real> C, real* __restrict__ pos, int cur) {
    const int rep = blockIdx.y, n = C.natoms;
    real* p = pos + (size_t)rep * 3 * n;
    const real* q = C.ref + (size_t)rep * 3 * n;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (STEP) {   // the evaluation API uses the stored site coordinates as given (as OpenMM does after setPositions)
        u64* f = C.f + (size_t)rep * 3 * n;
        if (t < S.nvs) {
            const Int4 v = S.vs[t];
            const real w0 = S.vs_w[3 * t], w1 = S.vs_w[3 * t + 1], w2 = S.vs_w[3 * t + 2];
            for (int k = 0; k < 3; k++) p[(size_t)k * n + v.x] = w0 * p[(size_t)k * n + v.y] + w1 * p[(size_t)k * n + v.z] + w2 * p[(size_t)k * n + v.w];
            f[v.x] = 0; f[n + v.x] = 0; f[2 * (size_t)n + v.x] = 0;
        }
        if (t < S.nfixed) { const int a = S.fixed[t]; f[a] = 0; f[n + a] = 0; f[2 * (size_t)n + a] = 0; }
    }
    __shared__ int slo[3], shi[3];
    if (threadIdx.x < 3) { slo[threadIdx.x] = 0x7fffffff; shi[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    const int n0 = C.bin[0].n, n1 = C.bin[1].n;
    const bool member = t < n0 + n1;
    bool moved = false;
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f};
    if (member) {
        const int b = t < n0 ? 0 : 1, m = t - (b ? n0 : 0);
        const int a = C.bin[b].atom[m];
        const real x = p[a], y = p[n + a], z = p[2 * (size_t)n + a];
        moved = (float)dist2(x - q[a], y - q[n + a], z - q[2 * (size_t)n + a]) > (float)C.skin_half2;
        lo[0] = hi[0] = (float)x; lo[1] = hi[1] = (float)y; lo[2] = hi[2] = (float)z;
        const int tag = b == 0 ? S.contact[0].info[m] : a;   // excluded volume: the atom index (partners are listed by atom)
        Rec<real>* rec = (Rec<real>*)C.bin[b].rec + (size_t)rep * C.bin[b].n;
        rec[m] = Rec<real>{x, y, z, make_tag(tag, (real)0)};
    }
    if (__ballot(moved) && (threadIdx.x & 63) == 0) C.flag[2 * rep + cur] = 1u;   // same value from every writer: no atomic needed
    if (__ballot(member)) {
        for (int k = 0; k < 3; k++) {
            for (int off = 32; off > 0; off >>= 1) { lo[k] = fminf(lo[k], __shfl_down(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off)); }
            if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], ordered_int(lo[k])); atomicMax(&shi[k], ordered_int(hi[k])); }
        }
    }
    __syncthreads();
    int* bb = C.bbox + ((size_t)rep * C.nblk_box + blockIdx.x) * 8;   // read by the geometry pass of this step only
    if (threadIdx.x < 3) { bb[threadIdx.x] = slo[threadIdx.x]; bb[4 + threadIdx.x] = shi[threadIdx.x]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) C.flag[2 * rep + (cur ^ 1)] = 0;   // nobody reads or writes the other flag during this step
    if (blockIdx.x == 0 && (int)threadIdx.x < C.n_cv) C.cv[(size_t)rep * C.n_cv + threadIdx.x] = 0;
}

// ------------------------------------------------------------------ rebuild (all kernels exit at once unless flagged)
// blockIdx.y = 2 * rep + bin
template <class real> __device__ __forceinline__ bool cell_rebuild(const CellSys<real>& C, int rep, int cur, int force) { return force || C.flag[2 * rep + cur] != 0; }

// one 1024-thread workgroup per (replica, bin): bounding box from the pre kernel's per-workgroup boxes, grid geometry,
// zeroed cell and tile counts
template <class real> __global__ void cell_geom_kernel(CellSys<real> C, int cur, int force) {
    const int rep = blockIdx.x >> 1, b = blockIdx.x & 1;
    const CellBin<real>& B = C.bin[b];
    if (B.n == 0 || !cell_rebuild(C, rep, cur, force)) return;
    __shared__ int slo[3], shi[3];
    if (threadIdx.x < 3) { slo[threadIdx.x] = 0x7fffffff; shi[threadIdx.x] = (int)0x80000000; }
    __syncthreads();
    for (int k = 0; k < 3; k++) {
        int lo = 0x7fffffff, hi = (int)0x80000000;
        for (int w = threadIdx.x; w < C.nblk_box; w += blockDim.x) {
            const int* bb = C.bbox + ((size_t)rep * C.nblk_box + w) * 8;
            lo = min(lo, bb[k]); hi = max(hi, bb[4 + k]);
        }
        for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off)); hi = max(hi, __shfl_down(hi, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], lo); atomicMax(&shi[k], hi); }
    }
    __syncthreads();
    float lo[3], ext = 0;
    for (int k = 0; k < 3; k++) {
        lo[k] = ordered_float(slo[k]) - 1e-3f;
        ext = fmaxf(ext, ordered_float(shi[k]) + 1e-3f - lo[k]);
    }
    const float edge = fmaxf((float)B.edge_min, ext / (float)AWSEM_CELL_MAXDIM * 1.0001f);
    int nd[3];
    for (int k = 0; k < 3; k++) {
        nd[k] = (int)floorf((ordered_float(shi[k]) + 1e-3f - lo[k]) / edge) + 1;
        nd[k] = nd[k] > AWSEM_CELL_MAXDIM ? AWSEM_CELL_MAXDIM : nd[k];
    }
    const int ncell = nd[0] * nd[1] * nd[2];
    if (threadIdx.x == 0) {
        for (int k = 0; k < 3; k++) { B.grid[4 * rep + k] = (real)lo[k]; B.dims[4 * rep + k] = nd[k]; }
        B.grid[4 * rep + 3] = (real)(1.0f / edge);
        B.dims[4 * rep + 3] = ncell;
    }
    for (int c = threadIdx.x; c < ncell; c += blockDim.x) B.cell_count[(size_t)rep * AWSEM_CELL_MAXCELLS + c] = 0;
    for (int c = threadIdx.x; c < AWSEM_CELL_MAXCELLS / 1024; c += blockDim.x) B.tile_sum[(size_t)rep * (AWSEM_CELL_MAXCELLS / 1024) + c] = 0;
}
// cell coordinates from the grid geometry stored by cell_geom_kernel
template <class real> __device__ __forceinline__ void cell_coords(const CellBin<real>& B, int rep, real x, real y, real z, int c[3]) {
    const real inv = B.grid[4 * rep + 3];
    const real v[3] = {x, y, z};
    for (int k = 0; k < 3; k++) {
        int t = (int)M<real>::floor((v[k] - B.grid[4 * rep + k]) * inv);
        const int nk = B.dims[4 * rep + k];
        c[k] = t < 0 ? 0 : (t >= nk ? nk - 1 : t);
    }
}
// grid (blocks, 2R): cell of every member, cell populations, new reference positions
template <class real> __global__ void cell_count_kernel(CellSys<real> C, const real* __restrict__ pos, int cur, int force) {
    const int rep = blockIdx.y >> 1, b = blockIdx.y & 1, n = C.natoms;
    const CellBin<real>& B = C.bin[b];
    if (!cell_rebuild(C, rep, cur, force)) return;
    const real* p = pos + (size_t)rep * 3 * n;
    real* q = C.ref + (size_t)rep * 3 * n;
    // tile populations: same-address global atomics from many workgroups serialise (~0.1 us each), so each workgroup
    // first histograms its members' tiles in LDS and then books every tile it met once
    __shared__ int hist[AWSEM_CELL_MAXCELLS / 1024];
    for (int t = threadIdx.x; t < AWSEM_CELL_MAXCELLS / 1024; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < B.n; m += gridDim.x * blockDim.x) {
        const int a = B.atom[m];
        const real x = p[a], y = p[n + a], z = p[2 * (size_t)n + a];
        q[a] = x; q[n + a] = y; q[2 * (size_t)n + a] = z;
        int c[3];
        cell_coords(B, rep, x, y, z, c);
        const int cell = (c[2] * B.dims[4 * rep + 1] + c[1]) * B.dims[4 * rep] + c[0];
        B.cell_of[(size_t)rep * B.n + m] = cell;
        atomicAdd(&B.cell_count[(size_t)rep * AWSEM_CELL_MAXCELLS + cell], 1);
        atomicAdd(&hist[cell >> 10], 1);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < AWSEM_CELL_MAXCELLS / 1024; t += blockDim.x)
        if (hist[t]) atomicAdd(&B.tile_sum[(size_t)rep * (AWSEM_CELL_MAXCELLS / 1024) + t], hist[t]);
}
// grid (MAXCELLS / 1024, 2R), 1024 threads: exclusive scan of the cell counts, one tile of 1024 cells per workgroup (its
// offset is the sum of the tile populations before it, booked by cell_count_kernel); the counts are zeroed for the fill pass
template <class real> __global__ void cell_scan_kernel(CellSys<real> C, int cur, int force) {
    const int rep = blockIdx.y >> 1, b = blockIdx.y & 1;
    const CellBin<real>& B = C.bin[b];
    if (B.n == 0 || !cell_rebuild(C, rep, cur, force)) return;
    const int ncell = B.dims[4 * rep + 3], tile = blockIdx.x, c = tile * 1024 + threadIdx.x;
    if (tile * 1024 >= ncell) return;
    __shared__ int wsum[16], base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int* ts = B.tile_sum + (size_t)rep * (AWSEM_CELL_MAXCELLS / 1024);
    int before = threadIdx.x < tile ? ts[threadIdx.x] : 0;   // at most 256 tiles
    if (threadIdx.x < 256) {
        for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
        if ((threadIdx.x & 63) == 0 && before) atomicAdd(&base, before);
    }
    int* cnt = B.cell_count + (size_t)rep * AWSEM_CELL_MAXCELLS;
    int* start = B.cell_start + (size_t)rep * (AWSEM_CELL_MAXCELLS + 1);
    const int mine = c < ncell ? cnt[c] : 0;
    int incl = mine;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off); if (lane >= off) incl += v; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; w++) woff += wsum[w];
    const int excl = base + woff + incl - mine;
    if (c < ncell) { start[c] = excl; cnt[c] = 0; }
    if (c == ncell - 1) start[ncell] = excl + mine;
}
template <class real> __global__ void cell_fill_kernel(CellSys<real> C, int cur, int force) {
    const int rep = blockIdx.y >> 1, b = blockIdx.y & 1;
    const CellBin<real>& B = C.bin[b];
    if (!cell_rebuild(C, rep, cur, force)) return;
    for (int m = blockIdx.x * blockDim.x + threadIdx.x; m < B.n; m += gridDim.x * blockDim.x) {
        const int cell = B.cell_of[(size_t)rep * B.n + m];
        const int slot = B.cell_start[(size_t)rep * (AWSEM_CELL_MAXCELLS + 1) + cell] + atomicAdd(&B.cell_count[(size_t)rep * AWSEM_CELL_MAXCELLS + cell], 1);
        B.sorted[(size_t)rep * B.n + slot] = m;
    }
}

// restores the invariant (zero counts) for the next sort; called by whoever builds the rows of (rep, bin) -- nobody reads the counts then
template <class real> __device__ __forceinline__ void cell_zero_counts(const CellSys<real>& C, int rep, int b, int bx, int nbx) {
    const CellBin<real>& B = C.bin[b];
    const int ncell = B.dims[4 * rep + 3];
    int* cnt = B.cell_count + (size_t)rep * AWSEM_CELL_MAXCELLS;
    for (int c = bx * (int)blockDim.x + (int)threadIdx.x; c < ncell; c += nbx * (int)blockDim.x) cnt[c] = 0;
    if (bx == 0) for (int t = threadIdx.x; t < AWSEM_CELL_MAXCELLS / 1024; t += blockDim.x) B.tile_sum[(size_t)rep * (AWSEM_CELL_MAXCELLS / 1024) + t] = 0;
}

// Neighbour rows.  16 lanes per member (members taken in cell order, so a wave works on four neighbours in space):
// the 27 cells around the member are 9 runs of 3 consecutive cells in the sorted member array; the 16 lanes test 16
// candidates at a time and the passing ones are compacted into the member's row with a ballot -- the row count is a
// register of the row, no atomics.  blockIdx.y = 2 * rep + bin.
//   bin 0 (contact): entry = site_j | in_density << 30 | has_pair_energy << 31
//   bin 1 (excl):    entry = member_j   (same class, not an excluded partner)
template <int ROW, class real> __device__ __forceinline__ void cell_rows_body_n(const DevSys<real>& S, const CellSys<real>& C, int rep, int b, int bx, int nbx) {
    const CellBin<real>& B = C.bin[b];
    const Rec<real>* rec = (const Rec<real>*)B.rec + (size_t)rep * B.n;
    const int* sorted = B.sorted + (size_t)rep * B.n;
    const int* cstart = B.cell_start + (size_t)rep * (AWSEM_CELL_MAXCELLS + 1);
    unsigned* rows = B.rows + (size_t)rep * B.n * B.stride;
    const int nx = B.dims[4 * rep], ny = B.dims[4 * rep + 1], nz = B.dims[4 * rep + 2];
    const int sub = threadIdx.x & (ROW - 1), row_in_wave = (threadIdx.x & 63) / ROW;
    const int nrows_total = (B.n + 64 / ROW - 1) & ~(64 / ROW - 1);   // whole waves: the ballots are wave collectives
    for (int k = (bx * (int)blockDim.x + (int)threadIdx.x) / ROW; k < nrows_total; k += nbx * (int)blockDim.x / ROW) {
        const bool vi = k < B.n;
        const int i = vi ? sorted[k] : 0;
        const Rec<real> ri = rec[i];
        const int tag_i = tag_bits(ri.tag);
        int c[3] = {0, 0, 0};
        if (vi) cell_coords(B, rep, ri.x, ri.y, ri.z, c);
        Int4 part = {-1, -1, -1, -1};
        int cls_i = 0;
        if (b == 1 && vi) { part = C.x_partners[i]; cls_i = C.x_class[i]; }
        int cnt = 0;
        for (int cell_row = 0; cell_row < 9; cell_row++) {
            const int cy = c[1] + cell_row % 3 - 1, cz = c[2] + cell_row / 3 - 1;
            const bool vrow = vi && cy >= 0 && cy < ny && cz >= 0 && cz < nz;
            const int x0 = max(c[0] - 1, 0), x1 = min(c[0] + 1, nx - 1);
            const int base = (cz * ny + cy) * nx;
            int q = vrow ? cstart[base + x0] + sub : 0;
            const int qe = vrow ? cstart[base + x1 + 1] : 0;
            while (__ballot(q < qe)) {
                const bool vq = q < qe;
                const int j = vq ? sorted[q] : i;
                q += ROW;
                const Rec<real> rj = rec[j];
                const real r2 = dist2(ri.x - rj.x, ri.y - rj.y, ri.z - rj.z);
                bool pass = vq && j != i && r2 < B.rc2;
                unsigned ent = (unsigned)j;
                if (b == 0) {
                    const ContactD<real>& d = S.contact[0];
                    const int tag_j = tag_bits(rj.tag), dres = (tag_j & 0xffff) - (tag_i & 0xffff);
                    const bool in_rho = dres >= 2 || dres <= -2;
                    const bool pair_e = !(d.flags & 2) && ((((tag_i ^ tag_j) >> 16) & 0x7ff) != 0 || dres >= d.min_seq_sep || -dres >= d.min_seq_sep);
                    pass = pass && (in_rho || pair_e);
                    ent |= (in_rho ? 0x40000000u : 0u) | (pair_e ? 0x80000000u : 0u);
                } else if (pass) {
                    const int aj = tag_bits(rj.tag);
                    pass = C.x_class[j] == cls_i && part.x != aj && part.y != aj && part.z != aj && part.w != aj;
                }
                const unsigned rowmask = (unsigned)(__ballot(pass) >> (row_in_wave * ROW)) & ((1u << ROW) - 1u);
                if (pass) {
                    const int slot = cnt + __popc(rowmask & ((1u << sub) - 1u));
                    if (slot < B.stride) rows[(size_t)k * B.stride + slot] = ent;
                    else atomicOr(C.status, 2);
                }
                cnt += __popc(rowmask);
            }
        }
        if (vi && sub == 0) B.row_count[(size_t)rep * B.n + k] = min(cnt, B.stride);
    }
}
// bin 0 (contact sites) walks with 16 lanes per row, bin 1 (excluded-volume atoms) with 4
template <class real> __device__ __forceinline__ void cell_rows_body(const DevSys<real>& S, const CellSys<real>& C, int rep, int b, int bx, int nbx) {
    if (b == 0) cell_rows_body_n<AWSEM_ROW>(S, C, rep, b, bx, nbx);
    else cell_rows_body_n<AWSEM_ROW_X>(S, C, rep, b, bx, nbx);
}
template <class real> __global__ void cell_rows_kernel(DevSys<real> S, CellSys<real> C, int cur, int force) {
    const int rep = blockIdx.y >> 1, b = blockIdx.y & 1;
    if (!cell_rebuild(C, rep, cur, force)) return;
    cell_zero_counts(C, rep, b, (int)blockIdx.x, (int)gridDim.x);
    cell_rows_body(S, C, rep, b, (int)blockIdx.x, (int)gridDim.x);
}

// ------------------------------------------------------------------ evaluation from the rows (grid (blocks, R))
// The leader of a row adds the row's force to the owner's accumulator: a plain read-modify-write, because within one
// of these kernels nobody else touches that atom (kernels on the stream run one after the other).
template <class real> __device__ __forceinline__ void owner_add_force(const CellView<real>& L, int atom, u64 X, u64 Y, u64 Z) {
    L.fx[atom] += X; L.fy[atom] += Y; L.fz[atom] += Z;
}
// Workgroups go to the eight XCDs round robin (workgroup b to XCD b mod 8), and every XCD has an L2 of its own: with rows dealt to
// workgroups in order, each XCD walked every eighth block of rows -- the whole system -- and pulled every member record, density and
// switch value of the system into its L2, eight copies per pass.  xcd_block turns the index round so that XCD x gets a CONTIGUOUS
// eighth of the blocks: rows are in cell order, so that is a slab of space, and an XCD's L2 holds the slab and its rim only.
#ifndef AWSEM_XCDS
#define AWSEM_XCDS 8
#endif
__device__ __forceinline__ int xcd_block(int b, int nb) {
    const int q = nb / AWSEM_XCDS, r = nb - q * AWSEM_XCDS, x = b % AWSEM_XCDS, y = b / AWSEM_XCDS;
    return x * q + (x < r ? x : r) + y;
}
#define AWSEM_ROW_LOOP(B, k)                                                                                              \
    const int sub = threadIdx.x & (AWSEM_ROW - 1);                                                                        \
    const int nrows_ = ((B).n + 3) & ~3;                                                                                  \
    for (int k = (xcd_block((int)blockIdx.x, (int)gridDim.x) * blockDim.x + threadIdx.x) / AWSEM_ROW; k < nrows_; k += gridDim.x * blockDim.x / AWSEM_ROW)

// Bonded terms, window by window: a workgroup stages the positions of AWSEM_WIN_SPAN consecutive atoms in LDS, runs
// every item whose lowest atom lies in its window (the item lists are sorted by window at finalize) with LDS
// accumulators, and flushes the touched atoms with one global atomic per component -- a fifth of the global atomics of
// the item-by-item kernel above, which keeps the items that do not fit a window.
// wstart[slot][w]: first item of window w in the slot's term; slots run over harm, chi, rama, helical, fm, memb.
template <class real> struct WinView : ViewCommon<real>, SoaOps<real> { typedef real real_t; };
// VS (dynamics, round 3): the window PLACES the virtual sites among the atoms it stages, from the parents' coordinates in global memory
// (massive particles, final since the integrator), and stores those of its own 512 atoms back for the kernels that follow -- the
// separate virtual-site pass of the pre kernel is gone.  Sites in the forward half window are recomputed by the next window as well:
// the same three products in the same order, the same bits.
// Dynamics (round 3) runs the windows as extra workgroups of cell_density_excl_kernel: the bonded terms need nothing from the contact
// passes and the contact passes nothing from them, so the two share a launch.
template <bool ENERGY, bool VS, class real> __device__ void cell_window_body(const DevSys<real>& S, const CellSys<real>& C, real* __restrict__ pos, const int* __restrict__ wstart, int nwin, unsigned mask, int rep, int w) {
    __shared__ real sp[3][AWSEM_WIN_SPAN];
    __shared__ u64 sf[3][AWSEM_WIN_SPAN];
    const int n = C.natoms;
    const int a0 = w * AWSEM_WIN, na = min(AWSEM_WIN_SPAN, n - a0);
    real* p = pos + (size_t)rep * 3 * n;
    for (int i = threadIdx.x; i < na; i += blockDim.x) {
        const int t = VS ? C.vs_of[a0 + i] : -1;
        if (t >= 0) {
            const Int4 v = S.vs[t];
            const real w0 = S.vs_w[3 * t], w1 = S.vs_w[3 * t + 1], w2 = S.vs_w[3 * t + 2];
            for (int k = 0; k < 3; k++) {
                const real x = w0 * p[(size_t)k * n + v.y] + w1 * p[(size_t)k * n + v.z] + w2 * p[(size_t)k * n + v.w];
                sp[k][i] = x; sf[k][i] = 0;
                if (i < AWSEM_WIN) p[(size_t)k * n + a0 + i] = x;
            }
        } else
            for (int k = 0; k < 3; k++) { sp[k][i] = p[(size_t)k * n + a0 + i]; sf[k][i] = 0; }
    }
    WinView<real> L;
    L.nthr = blockDim.x; L.ramap = nullptr;
    L.px = &sp[0][0] - a0; L.py = &sp[1][0] - a0; L.pz = &sp[2][0] - a0;
    L.fx = &sf[0][0] - a0; L.fy = &sf[1][0] - a0; L.fz = &sf[2][0] - a0;
    L.rho = L.drho = nullptr; L.rhov = L.alpha = L.dalpha = L.gamma = nullptr; L.satom = nullptr; L.cv = nullptr;
    L.lcount = nullptr; L.flag = nullptr; L.lists = nullptr;
    L.eg = C.eg + (size_t)rep * AWSEM_NGROUPS;
    __syncthreads();
    const int* ws = wstart + w;
    const int stride = nwin + 1;
    for (int t = 0; t < S.n_harm; t++, ws += stride) { HarmD<real> d = S.harm[t]; if (!AWSEM_ON(d) || ws[1] == ws[0]) continue; d.items += ws[0]; d.n = ws[1] - ws[0]; term_harmonic<ENERGY>(d, L); }
    for (int t = 0; t < S.n_chi; t++, ws += stride) { ChiD<real> d = S.chi[t]; if (!AWSEM_ON(d) || ws[1] == ws[0]) continue; d.atoms += ws[0]; d.n = ws[1] - ws[0]; term_chi<ENERGY>(d, L); }
    for (int t = 0; t < S.n_rama; t++, ws += stride) { RamaD<real> d = S.rama[t]; if (!AWSEM_ON(d) || ws[1] == ws[0]) continue; d.atoms += ws[0]; d.bw += (size_t)ws[0] * d.nwell; d.n = ws[1] - ws[0]; term_rama<ENERGY>(d, L); }
    for (int t = 0; t < S.n_helical; t++, ws += stride) { HelicalD<real> d = S.helical[t]; if (!AWSEM_ON(d) || ws[1] == ws[0]) continue; d.items += ws[0]; if (d.zitems) d.zitems += ws[0]; d.n = ws[1] - ws[0]; term_helical<ENERGY>(d, L); }
    for (int t = 0; t < S.n_fm; t++, ws += stride) { FmD<real> d = S.fm[t]; if (!AWSEM_ON(d) || ws[1] == ws[0]) continue; d.pairs2 += 2 * (size_t)ws[0]; d.table2 += (size_t)ws[0] * (d.ntab - 1); d.n = ws[1] - ws[0]; term_fm<ENERGY, false>(d, L); }
    for (int t = 0; t < S.n_memb; t++, ws += stride) { MembD<real> d = S.memb[t]; if (!AWSEM_ON(d) || ws[1] == ws[0]) continue; d.items += ws[0]; d.n = ws[1] - ws[0]; term_memb<ENERGY>(d, L); }
    __syncthreads();
    u64* f = C.f + (size_t)rep * 3 * n;
    for (int i = threadIdx.x; i < na; i += blockDim.x)
        for (int k = 0; k < 3; k++) { const u64 v = sf[k][i]; if (v) fixed_add(&f[(size_t)k * n + a0 + i], v); }
}

template <bool ENERGY, bool VS, class real> __global__ __launch_bounds__(256) void cell_window_kernel(DevSys<real> S, CellSys<real> C, real* __restrict__ pos, const int* __restrict__ wstart, int nwin, unsigned mask) {
    cell_window_body<ENERGY, VS>(S, C, pos, wstart, nwin, mask, (int)blockIdx.y, (int)blockIdx.x);
}

// pass 1: rho_i, then the per-site pass (burial wells, water switch, dE/drho seed) by the row leader;
// blocks from `nb_contact` on evaluate the excluded volume (its rows live in bin 1).
// rows_cur >= 0 (dynamics, round 3): when the replica's flag is set, every 16-lane group first BUILDS the rows it is about to walk (the
// sort ran in the launch before); the same mapping of groups to rows serves both loops.
// WIN (dynamics): blocks from nb_contact + nb_excl on are the windows of the bonded terms (cell_window_body: places the virtual sites); the
// excluded-volume leaders then add with atomics, because a window may be adding to the same atom.
template <bool ENERGY, bool WIN, class real> __global__ __launch_bounds__(256) void cell_density_excl_kernel(DevSys<real> S, CellSys<real> C, real* __restrict__ pos, int nb_contact, int nb_excl, unsigned mask,
                                                                            int rows_cur, int rows_force, const int* __restrict__ wstart, int nwin) {
    const int rep = blockIdx.y;
    // the windows take the FIRST block indices: theirs is the longest chain of dependent loads in this launch, so they start at once and
    // the row walks fill in around them (dispatched last they ran after everybody else: measured, 28.6 us against 17.3 + 11)
    if (WIN && (int)blockIdx.x < nwin) {
        cell_window_body<ENERGY, true>(S, C, pos, wstart, nwin, mask, rep, (int)blockIdx.x);
        return;
    }
    // (the index within the contact / excluded-volume range turned round for the XCDs: xcd_block.  The window blocks in front shift
    // which XCD a block lands on by nwin mod 8 -- a rotation of the slabs, harmless.)
    const int bx0 = (int)blockIdx.x - (WIN ? nwin : 0);
    const int bx = bx0 < nb_contact ? xcd_block(bx0, nb_contact) : nb_contact + xcd_block(bx0 - nb_contact, nb_excl);
    const CellView<real> L = cell_view(S, C, pos, rep);
    Energy en;
    const bool build = rows_cur >= 0 && cell_rebuild(C, rep, rows_cur, rows_force);   // uniform over the replica
    if (bx < nb_contact) {
        const ContactD<real>& d = S.contact[0];
        const CellBin<real>& B = C.bin[0];
        if (build) {
            cell_zero_counts(C, rep, 0, bx, nb_contact);
            cell_rows_body(S, C, rep, 0, bx, nb_contact);
            __syncthreads();   // the rows and their counts are read back by other lanes of the group
        }
        const Rec<real>* rec = (const Rec<real>*)B.rec + (size_t)rep * B.n;
        const int* sorted = B.sorted + (size_t)rep * B.n;
        const unsigned* rows = B.rows + (size_t)rep * B.n * B.stride;
        const int sub = threadIdx.x & (AWSEM_ROW - 1), nrows = (B.n + 3) & ~3;
        for (int k = (bx * (int)blockDim.x + (int)threadIdx.x) / AWSEM_ROW; k < nrows; k += nb_contact * (int)blockDim.x / AWSEM_ROW) {
            const bool vi = k < B.n;
            const int i = vi ? sorted[k] : 0;
            const Rec<real> ri = rec[i];
            const int cnt = vi ? B.row_count[(size_t)rep * B.n + k] : 0;
            u64 acc = 0;
            for (int s = sub; s < cnt; s += AWSEM_ROW) {
                const unsigned ent = rows[(size_t)k * B.stride + s];
                if (!(ent & 0x40000000u)) continue;
                const Rec<real> rj = rec[ent & 0xffffffu];
                const real r2 = dist2(ri.x - rj.x, ri.y - rj.y, ri.z - rj.z);
                if (!(r2 < d.cutoff2)) continue;
                const real r = M<real>::sqrt(r2);
                acc += to_fixed(sigmoid(d.eta2 * (r - d.rmin)) * sigmoid(d.eta2 * (d.rmax - r)));
            }
            acc = row_sum(acc);
            if (vi && sub == 0) {
                L.rho[i] = acc;
                single_body<ENERGY>(d, L, i, L.satom[i], en);
            }
        }
        flush_energy<ENERGY>(L, d.group, en);
    } else {
        const ExclD<real>& d = S.excl[0];
        if (!AWSEM_ON(d)) return;
        const CellBin<real>& B = C.bin[1];
        if (build) {
            cell_zero_counts(C, rep, 1, bx - nb_contact, nb_excl);
            cell_rows_body(S, C, rep, 1, bx - nb_contact, nb_excl);
            __syncthreads();
        }
        const Rec<real>* rec = (const Rec<real>*)B.rec + (size_t)rep * B.n;
        const int* sorted = B.sorted + (size_t)rep * B.n;
        const unsigned* rows = B.rows + (size_t)rep * B.n * B.stride;
        constexpr int RX = AWSEM_ROW_X;
        const int sub = threadIdx.x & (RX - 1), nrows = (B.n + 64 / RX - 1) & ~(64 / RX - 1), nb = nb_excl;
        const real r02 = d.r0 * d.r0;
        for (int k = ((bx - nb_contact) * blockDim.x + threadIdx.x) / RX; k < nrows; k += nb * blockDim.x / RX) {
            const bool vi = k < B.n;
            const int i = vi ? sorted[k] : 0;
            const Rec<real> ri = rec[i];
            const int cnt = vi ? B.row_count[(size_t)rep * B.n + k] : 0;
            u64 X = 0, Y = 0, Z = 0;
            for (int s = sub; s < cnt; s += RX) {
                const Rec<real> rj = rec[rows[(size_t)k * B.stride + s]];
                const real dx = ri.x - rj.x, dy = ri.y - rj.y, dz = ri.z - rj.z;
                const real r2 = dist2(dx, dy, dz);
                if (!(r2 < r02)) continue;
                const real ir = M<real>::rsqrt(r2), dr = r2 * ir - d.r0;
                if (ENERGY) en += (double)((real)0.5 * d.k * dr * dr);   // the pair is met from both sides
                const real t = (real)-2 * d.k * dr * ir * (real)AWSEM_FIXED_SCALE;
                X += to_fixed_scaled(t * dx); Y += to_fixed_scaled(t * dy); Z += to_fixed_scaled(t * dz);
            }
            if (__ballot((X | Y | Z) != 0) == 0) continue;   // clashes are rare
            X = row_sum<RX>(X); Y = row_sum<RX>(Y); Z = row_sum<RX>(Z);
            if (vi && sub == 0 && (X | Y | Z) != 0) {
                const int a = tag_bits(ri.tag);
                if (WIN) { fixed_add(&L.fx[a], X); fixed_add(&L.fy[a], Y); fixed_add(&L.fz[a], Z); }
                else owner_add_force(L, a, X, Y, Z);
            }
        }
        flush_energy<ENERGY>(L, d.group, en);
    }
}
// pass 2: pair energy, its radial force on the owner, the owner's dE/drho through sigma_water
template <bool ENERGY, class real> __global__ void cell_pairs_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ContactD<real>& d = S.contact[0];
    const CellBin<real>& B = C.bin[0];
    const Rec<real>* rec = (const Rec<real>*)B.rec + (size_t)rep * B.n;
    const int* sorted = B.sorted + (size_t)rep * B.n;
    const unsigned* rows = B.rows + (size_t)rep * B.n * B.stride;
    const bool zdep = d.flags & 1;
    const real one = (real)1;
    Energy en;
    AWSEM_ROW_LOOP(B, k) {
        const bool vi = k < B.n;
        const int i = vi ? sorted[k] : 0;
        const Rec<real> ri = rec[i];
        const int cnt = vi ? B.row_count[(size_t)rep * B.n + k] : 0;
        const int ti = (tag_bits(ri.tag) >> 27) & 31;
        const real si = L.rhov[i], ai = zdep ? L.alpha[i] : (real)0;
        u64 X = 0, Y = 0, Z = 0, D = 0, FZ = 0;
        for (int s = sub; s < cnt; s += AWSEM_ROW) {
            const unsigned ent = rows[(size_t)k * B.stride + s];
            if (!(ent >> 31)) continue;
            const int j = ent & 0xffffffu;
            const Rec<real> rj = rec[j];
            const real dx = ri.x - rj.x, dy = ri.y - rj.y, dz = ri.z - rj.z;
            const real r2 = dist2(dx, dy, dz);
            if (!(r2 < d.cutoff2)) continue;
            const real ir = M<real>::rsqrt(r2), r = r2 * ir;
            const int tj = (tag_bits(rj.tag) >> 27) & 31;
            const real sj = L.rhov[j], aj = zdep ? L.alpha[j] : (real)0;
            // the type pair as the lower site sees it (what the half lists of the resident path hold)
            const int ty = i < j ? ti * 20 + tj : tj * 20 + ti;
            const PairTerms<real> t = (d.flags & 16) ? contact_pair_math<true>(d, d.gamma, (i < j ? i : j) * d.nsites + (i < j ? j : i), r, si * sj, ai * aj)
                                                     : contact_pair_math(d, L.gamma, ty, r, si * sj, ai * aj);
            if (ENERGY) en += (double)((real)0.5 * t.E);
            const real c = -t.dEdr * ir * (real)AWSEM_FIXED_SCALE;
            X += to_fixed_scaled(c * dx); Y += to_fixed_scaled(c * dy); Z += to_fixed_scaled(c * dz);
            D += to_fixed(t.dEdsw * sj * (-d.eta_s2 * si * (one - si)));
            if (zdep) FZ += to_fixed(-t.MW * aj * L.dalpha[i]);
        }
        X = row_sum(X); Y = row_sum(Y); Z = row_sum(Z); D = row_sum(D);
        if (zdep) FZ = row_sum(FZ);
        if (vi && sub == 0) {
            owner_add_force(L, L.satom[i], X, Y, Z + FZ);
            L.drho[i] += D;
        }
    }
    flush_energy<ENERGY>(L, d.group, en);
}
// pass 3: chain rule through the densities: F_i -= (dE/drho_i + dE/drho_j) theta_I'(r) (x_i - x_j)/r
template <class real> __global__ void cell_chain_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const ContactD<real>& d = S.contact[0];
    const CellBin<real>& B = C.bin[0];
    const Rec<real>* rec = (const Rec<real>*)B.rec + (size_t)rep * B.n;
    const int* sorted = B.sorted + (size_t)rep * B.n;
    const unsigned* rows = B.rows + (size_t)rep * B.n * B.stride;
    AWSEM_ROW_LOOP(B, k) {
        const bool vi = k < B.n;
        const int i = vi ? sorted[k] : 0;
        const Rec<real> ri = rec[i];
        const int cnt = vi ? B.row_count[(size_t)rep * B.n + k] : 0;
        const u64 di = L.drho[i];
        u64 X = 0, Y = 0, Z = 0;
        for (int s = sub; s < cnt; s += AWSEM_ROW) {
            const unsigned ent = rows[(size_t)k * B.stride + s];
            if (!(ent & 0x40000000u)) continue;
            const int j = ent & 0xffffffu;
            const Rec<real> rj = rec[j];
            const real dx = ri.x - rj.x, dy = ri.y - rj.y, dz = ri.z - rj.z;
            const real r2 = dist2(dx, dy, dz);
            if (!(r2 < d.cutoff2)) continue;
            const real ir = M<real>::rsqrt(r2), r = r2 * ir;
            const real s1 = sigmoid(d.eta2 * (r - d.rmin)), s2 = sigmoid(d.eta2 * (d.rmax - r));
            const real g = fixed_to_real(di + L.drho[j], (real)0);
            const real c = -g * d.eta2 * s1 * s2 * (s2 - s1) * ir * (real)AWSEM_FIXED_SCALE;
            X += to_fixed_scaled(c * dx); Y += to_fixed_scaled(c * dy); Z += to_fixed_scaled(c * dz);
        }
        X = row_sum(X); Y = row_sum(Y); Z = row_sum(Z);
        if (vi && sub == 0) owner_add_force(L, L.satom[i], X, Y, Z);
    }
}

// ------------------------------------------------------------------ hydrogen-bond terms (beta 1-3, P-AP 1-2)
// Every acceptor owns a fixed-stride row of the donors whose screening atom (N) lies within the term's screening
// distance of the acceptor's (O resp. CA) and whose table entry is non-zero.  The rows are rebuilt with the other
// lists by a brute-force scan -- 16 lanes per acceptor test 16 donors at a time and compact the passing ones with a
// ballot, the row count is a register: O(acceptors x donors) per REBUILD, which is what OpenMM's CustomHbondForce
// does every step.  The rebuild flag watches the list members of the contact and excluded-volume terms (CA, CB, O);
// a donor's N is a virtual site of CA_{i-1}, CA_i, O_{i-1} with |weights| summing to 1.373, so while no member has
// moved further than skin/2 an O-N or CA-N distance has changed by less than 1.19 skin: the rows are cut at the
// screening distance + 1.2 skin.  Evaluation: 16 lanes walk a row, same per-pair code as the resident path
// (beta_pair / pap_pair), forces through global fixed-point atomics (a pair touches up to five atoms).
#define AWSEM_HB_BLOCK 16   // donors per bounding sphere: one iteration of the row builder's fine test (= AWSEM_ROW)
// [R][nacc][stride] donor indices, [R][nacc]; [R][nblocks][4] bounding spheres (centre, radius; fp32) of the donors' screening atoms in
// blocks of AWSEM_HB_BLOCK consecutive donors, renewed at every rebuild (cell_hb_spheres_kernel)
struct HbRows { unsigned* rows; int* row_count; int stride; float* spheres; int nblocks; };

template <class real> __device__ inline CellView<real> cell_view_plain(const CellSys<real>& C, const real* pos, int rep) {
    CellView<real> L;
    L.nthr = blockDim.x; L.ramap = nullptr;
    const size_t n = C.natoms;
    real* p = const_cast<real*>(pos) + (size_t)rep * 3 * n;
    L.px = p; L.py = p + n; L.pz = p + 2 * n;
    u64* f = C.f + (size_t)rep * 3 * n;
    L.fx = f; L.fy = f + n; L.fz = f + 2 * n;
    L.rho = L.drho = nullptr; L.rhov = L.alpha = L.dalpha = L.gamma = nullptr; L.satom = nullptr; L.cv = nullptr;
    L.eg = C.eg + (size_t)rep * AWSEM_NGROUPS;
    L.lcount = nullptr; L.flag = nullptr; L.lists = nullptr;
    return L;
}
// The cell-list view with the coordinates read from packed {x, y, z, -} records (HbTerms::pos4)
template <class real> struct CellViewP : CellView<real> {
    const Rec<real>* p4;
    __device__ __forceinline__ CellViewP(const CellView<real>& b, const Rec<real>* q) : CellView<real>(b), p4(q) {}
    __device__ __forceinline__ R3<real> pos(int a) const { const Rec<real> q = p4[a]; return R3<real>{q.x, q.y, q.z}; }
};
// ... and with the forces collected per workgroup before they reach the force array.  The sixteen lanes of a row walk the partners of
// ONE acceptor, and the sixteen acceptors of a workgroup are neighbours along a chain that see the same donors: most atomics of this
// kernel hit an address that other lanes of the same instruction, or other rows of the same workgroup, hit at the same moment, and
// same-address atomics are served one after the other (measured with the atomics compiled out: 62 of the kernel's 90 us at 20 008
// residues were theirs, 39 of them the acceptor atoms').  The contributions go to a workgroup-local table in LDS instead -- open
// addressing on the atom index, AWSEM_HB_SLOTS slots, a neighbourhood of AWSEM_HB_PROBES probes, beyond that the global atomic --
// and every occupied slot is added to the force array once, when the workgroup's rows are done: the same fixed-point sums bit for
// bit.  (Summing the acceptor atoms' share in registers and across the row with shuffles, the first form, was slower than the table.)
#define AWSEM_HB_SLOTS 512
#define AWSEM_HB_PROBES 8
template <class real> struct CellViewHb : CellViewP<real> {
    int* tkey;    // LDS [AWSEM_HB_SLOTS] atom index, -1 = free
    u64* tacc;    // LDS [AWSEM_HB_SLOTS][3]
    __device__ __forceinline__ CellViewHb(const CellView<real>& b, const Rec<real>* q, int* key, u64* acc) : CellViewP<real>(b, q), tkey(key), tacc(acc) {}
    // every thread of the workgroup, before the first and after the last row
    __device__ __forceinline__ void table_clear() const {
        for (int sl = threadIdx.x; sl < AWSEM_HB_SLOTS; sl += blockDim.x) { tkey[sl] = -1; tacc[3 * sl] = 0; tacc[3 * sl + 1] = 0; tacc[3 * sl + 2] = 0; }
        __syncthreads();
    }
    __device__ __forceinline__ void table_flush() const {
        __syncthreads();
        for (int sl = threadIdx.x; sl < AWSEM_HB_SLOTS; sl += blockDim.x) {
            const int atom = tkey[sl];
            if (atom < 0) continue;
            const u64 X = tacc[3 * sl], Y = tacc[3 * sl + 1], Z = tacc[3 * sl + 2];
            if (X) fixed_add(&this->fx[atom], X);
            if (Y) fixed_add(&this->fy[atom], Y);
            if (Z) fixed_add(&this->fz[atom], Z);
        }
    }
    __device__ __forceinline__ void book(int i, u64 X, u64 Y, u64 Z) const {   // += on atom i
        unsigned h = ((unsigned)i * 2654435761u) >> 23;   // 9 bits: AWSEM_HB_SLOTS
#pragma unroll 1
        for (int probe = 0; probe < AWSEM_HB_PROBES; probe++) {
            const int old = atomicCAS(&tkey[h], -1, i);
            if (old == -1 || old == i) { fixed_add(&tacc[3 * h], X); fixed_add(&tacc[3 * h + 1], Y); fixed_add(&tacc[3 * h + 2], Z); return; }
            h = (h + 1) & (AWSEM_HB_SLOTS - 1);
        }
        fixed_add(&this->fx[i], X); fixed_add(&this->fy[i], Y); fixed_add(&this->fz[i], Z);
    }
    __device__ __forceinline__ void pair_force(int i, int j, real dx, real dy, real dz, real s) const {
        const real t = s * (real)AWSEM_FIXED_SCALE;
        const u64 X = to_fixed_scaled(t * dx), Y = to_fixed_scaled(t * dy), Z = to_fixed_scaled(t * dz);
        book(i, X, Y, Z);
        book(j, (u64)0 - X, (u64)0 - Y, (u64)0 - Z);
    }
};
static_assert((AWSEM_HB_SLOTS & (AWSEM_HB_SLOTS - 1)) == 0 && AWSEM_HB_SLOTS == 512, "CellViewHb::book hashes to 9 bits");
template <bool ENERGY, class V> __device__ __forceinline__ void hb_pair(const BetaD<typename V::real_t>& d, const V& L, const Int4& A, const Int4& D, Energy& e) { beta_pair<ENERGY>(d, L, A, D, e); }
template <bool ENERGY, class V> __device__ __forceinline__ void hb_pair(const PapD<typename V::real_t>& d, const V& L, const Int4& A, const Int4& D, Energy& e) { pap_pair<ENERGY>(d, L, A, D, e); }

// All hydrogen-bond terms of a system go through ONE launch of each kernel (grid (blocks, R, terms)): their descriptors,
// rows and list radii sit in device memory and are fetched with scalar loads (blockIdx.z is uniform).
// pos4: [R][natoms] the coordinates once more as packed {x, y, z, -} records, written by cell_hb_prepare_kernel before every walk.  The
// hydrogen-bond kernels gather up to six atoms per pair from all over the system; from the SoA arrays that is three cache lines per atom
// and lane, from the records one (a third of the kernel's vector-memory instructions; what it waited for, though, were its atomics).
template <class real> struct HbTerms { const BetaD<real>* beta; const PapD<real>* pap; const HbRows* rows; const real* rc2; int n_beta, n_pap; Rec<real>* pos4; };

// Donor rows of one hydrogen-bond term: for every acceptor the donors whose screening atom lies inside the list radius (and whose table
// entry is not zero), in ascending donor order.  Rounds 1 and 2 tested every acceptor against every donor -- 4 10^8 distance tests per
// term for the 20 008-residue system, 4.3 ms per rebuild.  Now the donors come in blocks of 16 consecutive ones (neighbours along a
// chain, so a block is a blob of ~1 nm) with a bounding sphere each (hb_spheres_body, renewed at every rebuild): a wave tests 64 spheres
// per iteration against its four acceptors and runs the fine test only on the blocks that can hold a partner.  The rows are the same
// rows (same donors, same order): |a - d| >= |a - c| - r_block, so a block further than list radius + r_block holds nothing.
template <class real, class Desc> __device__ __forceinline__ void hb_spheres_body(const Desc& d, const CellSys<real>& C, const HbRows& H, const real* __restrict__ pos, int rep) {
    const size_t n = C.natoms;
    const real* p = pos + (size_t)rep * 3 * n;
    float* sp = H.spheres + (size_t)rep * H.nblocks * 4;
    const int sub = threadIdx.x & (AWSEM_HB_BLOCK - 1);
    const int nb_pad = (H.nblocks + 3) & ~3;   // whole waves: the shuffles are wave collectives
    for (int b = (blockIdx.x * blockDim.x + threadIdx.x) / AWSEM_HB_BLOCK; b < nb_pad; b += gridDim.x * blockDim.x / AWSEM_HB_BLOCK) {
        const bool vb = b < H.nblocks;
        const int q = b * AWSEM_HB_BLOCK + sub;
        const Int4 D = d.don[vb ? (q < d.ndon ? q : b * AWSEM_HB_BLOCK) : 0];   // a short last block repeats its first donor
        const float x = (float)p[D.x], y = (float)p[n + D.x], z = (float)p[2 * n + D.x];
        float cx = x, cy = y, cz = z;
        for (int m = AWSEM_HB_BLOCK / 2; m > 0; m >>= 1) { cx += __shfl_xor(cx, m, AWSEM_HB_BLOCK); cy += __shfl_xor(cy, m, AWSEM_HB_BLOCK); cz += __shfl_xor(cz, m, AWSEM_HB_BLOCK); }
        cx *= 1.0f / AWSEM_HB_BLOCK; cy *= 1.0f / AWSEM_HB_BLOCK; cz *= 1.0f / AWSEM_HB_BLOCK;
        float r2 = (x - cx) * (x - cx) + (y - cy) * (y - cy) + (z - cz) * (z - cz);
        for (int m = AWSEM_HB_BLOCK / 2; m > 0; m >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, m, AWSEM_HB_BLOCK));
        if (vb && sub == 0) { sp[4 * b] = cx; sp[4 * b + 1] = cy; sp[4 * b + 2] = cz; sp[4 * b + 3] = sqrtf(r2); }
    }
}
template <class real, class Desc> __device__ __forceinline__ void hb_rows_body(const Desc& d, const CellSys<real>& C, const HbRows& H, const Rec<real>* __restrict__ pos4, real rc2, int rep, int bx) {
    const Rec<real>* p4 = pos4 + (size_t)rep * C.natoms;
    const float* sp = H.spheres + (size_t)rep * H.nblocks * 4;
    const int lane = threadIdx.x & 63, sub = threadIdx.x & (AWSEM_ROW - 1), row_in_wave = lane / AWSEM_ROW;
    const int nrows_total = (d.nacc + 3) & ~3;   // whole waves: the ballots are wave collectives
    const float reach = sqrtf((float)rc2) + 1e-3f;   // list radius plus a margin for the spheres' fp32 arithmetic
    unsigned* rows = H.rows + (size_t)rep * d.nacc * H.stride;
    for (int k = (bx * blockDim.x + threadIdx.x) / AWSEM_ROW; k < nrows_total; k += gridDim.x * blockDim.x / AWSEM_ROW) {
        const bool vi = k < d.nacc;
        const Int4 A = d.acc[vi ? k : 0];
        const int ares = Desc::res_slot == 3 ? A.w : A.z;
        const Rec<real> pa = p4[A.x];
        const real ax = pa.x, ay = pa.y, az = pa.z;
        float gx[4], gy[4], gz[4];   // the wave's four acceptors, for the sphere test (an invalid row repeats acceptor 0: harmless)
        for (int g = 0; g < 4; g++) { gx[g] = __shfl((float)ax, g * AWSEM_ROW); gy[g] = __shfl((float)ay, g * AWSEM_ROW); gz[g] = __shfl((float)az, g * AWSEM_ROW); }
        int cnt = 0;
        for (int b0 = 0; b0 < H.nblocks; b0 += 64) {   // the same trip count for every lane
            const int b = b0 + lane;
            bool near = false;
            if (b < H.nblocks) {
                const float cx = sp[4 * b], cy = sp[4 * b + 1], cz = sp[4 * b + 2], lim = reach + sp[4 * b + 3];
                for (int g = 0; g < 4; g++) near = near || (gx[g] - cx) * (gx[g] - cx) + (gy[g] - cy) * (gy[g] - cy) + (gz[g] - cz) * (gz[g] - cz) < lim * lim;
            }
            unsigned long long blocks = __ballot(near);
            while (blocks) {   // uniform: every lane walks the same blocks, in ascending order
                const int q0 = (b0 + __builtin_ctzll(blocks)) * AWSEM_HB_BLOCK;
                blocks &= blocks - 1;
                const int q = q0 + sub;
                const bool vq = vi && q < d.ndon;
                const Int4 D = d.don[vq ? q : 0];
                const int dres = Desc::res_slot == 3 ? D.w : D.z, dq = dres - ares;
                const Rec<real> pd = p4[D.x];
                const real r2 = dist2(ax - pd.x, ay - pd.y, az - pd.z);
                bool pass = vq && r2 < rc2 && (dq <= d.band_lo || dq >= d.band_hi);
                if (pass) {
                    if (d.fn) pass = hb_fn_value(*d.fn, dres, ares) != (real)0;
                    else { const size_t bit = (size_t)dres * d.nres + ares; pass = (d.nzmask[bit >> 5] >> (bit & 31)) & 1u; }
                }
                const unsigned rowmask = (unsigned)(__ballot(pass) >> (row_in_wave * AWSEM_ROW)) & 0xffffu;
                if (pass) {
                    const int slot = cnt + __popc(rowmask & ((1u << sub) - 1u));
                    if (slot < H.stride) rows[(size_t)k * H.stride + slot] = (unsigned)q;
                    else atomicOr(C.status, 2);
                }
                cnt += __popc(rowmask);
            }
        }
        if (vi && sub == 0) H.row_count[(size_t)rep * d.nacc + k] = min(cnt, H.stride);
    }
}
// grid (blocks, R, terms + 1): slice `terms` packs the coordinates (every step), the others renew the spheres of their term's donors on
// a rebuild step -- from the SoA coordinates, which are what this launch can rely on.  Launched before the rows are built / walked.
template <class real> __global__ void cell_hb_prepare_kernel(HbTerms<real> T, CellSys<real> C, const real* __restrict__ pos, int cur, int force) {
    const int rep = blockIdx.y, t = blockIdx.z;
    if (t == T.n_beta + T.n_pap) {
        const size_t n = C.natoms;
        const real* p = pos + (size_t)rep * 3 * n;
        Rec<real>* p4 = T.pos4 + (size_t)rep * n;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C.natoms; i += gridDim.x * blockDim.x) p4[i] = Rec<real>{p[i], p[n + i], p[2 * n + i], (real)0};
        return;
    }
    if (!cell_rebuild(C, rep, cur, force)) return;
    const HbRows H = T.rows[t];
    if (t < T.n_beta) { const BetaD<real> d = T.beta[t]; hb_spheres_body(d, C, H, pos, rep); }
    else { const PapD<real> d = T.pap[t - T.n_beta]; hb_spheres_body(d, C, H, pos, rep); }
}
template <class real> __global__ void cell_hb_rows_kernel(HbTerms<real> T, CellSys<real> C, const real* __restrict__ pos, int cur, int force) {
    const int rep = blockIdx.y, t = blockIdx.z;
    if (!cell_rebuild(C, rep, cur, force)) return;
    const HbRows H = T.rows[t];
    const real rc2 = T.rc2[t];
    if (t < T.n_beta) { const BetaD<real> d = T.beta[t]; hb_rows_body(d, C, H, T.pos4, rc2, rep, (int)blockIdx.x); }
    else { const PapD<real> d = T.pap[t - T.n_beta]; hb_rows_body(d, C, H, T.pos4, rc2, rep, (int)blockIdx.x); }
}

// ------------------------------------------------------------------ the rebuild without launches of its own (round 3)
// The five kernels above cost a launch each on every step although they exit at once on all but one step in twenty or more.  Now:
//   * SORT (geometry -> count -> scan -> fill) is the work of G = 32 small workgroups per (replica, bin) that synchronise among
//     themselves with three grid barriers of their own (cell_grid_barrier: arrival counters in device memory, agent-scope fences;
//     a wait is BOUNDED -- after ~1 s bit 3 of the status word is set, every later phase is skipped and the next synchronisation
//     raises: a hang is impossible).  They ride along as extra workgroups of the window kernel (cell_window_kernel; a launch of their own,
//     cell_sort_kernel, where there is no window kernel) and leave at once unless the replica's flag is set.  (All workgroups of one
//     launch doing all four phases was measured first: with 768 workgroups per group a barrier costs ~280 us -- same-address arrivals
//     serialise at ~0.1 us each and the pollers crowd them out; with 32 it is a few us.)
//     (Measured and rejected: the sort riding in the PREVIOUS step's integrator launch, its workgroups waiting for an arrival counter --
//     235 same-address arrivals that all come at once plus the pollers made that launch 53 us instead of 14.)
//   * ROWS are per owner and need nothing but the sorted arrays: the 16 lanes that are about to walk row i in the density pass (the
//     excluded-volume pass, the hydrogen-bond pass) build it first when the flag is set -- full occupancy, no launch.
// Invariant between rebuilds: cell_count[0 .. ncell) and tile_sum are zero (set at allocation, restored by whoever builds the rows).
#define AWSEM_BAR_SPINS (1u << 20)
#define AWSEM_SORT_G 32
__device__ __forceinline__ bool cell_grid_barrier(unsigned* counter, unsigned nwg, unsigned* reset, int* status, int* sh_ok) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // every wave: its stores are visible device-wide before the arrival
    __syncthreads();
    if (threadIdx.x == 0) {
        int ok = 1;
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < nwg) {
            __builtin_amdgcn_s_sleep(16);
            if (++spins > AWSEM_BAR_SPINS) { ok = 0; atomicOr(status, 8); break; }
        }
        *sh_ok = ok;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // every wave: no stale lines of what the others wrote
    // everyone has passed the PREVIOUS barrier (they all arrived at this one): its counter goes back to zero
    if (*sh_ok && reset && threadIdx.x == 0) __hip_atomic_store(reset, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *sh_ok != 0;
}
template <class real> struct CellGeo { real lo[3], inv; int nd[3], ncell; };
template <class real> __device__ __forceinline__ void cell_coords_geo(const CellGeo<real>& g, real x, real y, real z, int c[3]) {
    const real v[3] = {x, y, z};
    for (int k = 0; k < 3; k++) {
        int t = (int)M<real>::floor((v[k] - g.lo[k]) * g.inv);
        c[k] = t < 0 ? 0 : (t >= g.nd[k] ? g.nd[k] - 1 : t);
    }
}
// workgroup g of G (256 threads each) of the group (rep, bin b); every workgroup of the group must call it (grid barriers)
template <class real> __device__ void cell_sort_body(const CellSys<real>& C, const real* __restrict__ pos, int rep, int b, int g, int G, int cur, int force) {
    const int n = C.natoms;
    const CellBin<real>& B = C.bin[b];
    if (B.n == 0 || !cell_rebuild(C, rep, cur, force)) return;
    unsigned* bar = C.bar + ((size_t)rep * 2 + b) * 4;
    __shared__ int slo[3], shi[3], sh_ok, sh_base;
    __shared__ int hist[AWSEM_CELL_MAXCELLS / 1024];
    __shared__ int wsum[4];
    __shared__ CellGeo<real> geo;
    // ---- geometry: every workgroup works it out for itself from the per-workgroup boxes (a few KB out of L2)
    if (threadIdx.x < 3) { slo[threadIdx.x] = 0x7fffffff; shi[threadIdx.x] = (int)0x80000000; }
    for (int t = threadIdx.x; t < AWSEM_CELL_MAXCELLS / 1024; t += blockDim.x) hist[t] = 0;
    __syncthreads();
    for (int k = 0; k < 3; k++) {
        int lo = 0x7fffffff, hi = (int)0x80000000;
        for (int w = threadIdx.x; w < C.nblk_box; w += blockDim.x) {
            const int* bb = C.bbox + ((size_t)rep * C.nblk_box + w) * 8;
            lo = min(lo, bb[k]); hi = max(hi, bb[4 + k]);
        }
        for (int off = 32; off > 0; off >>= 1) { lo = min(lo, __shfl_down(lo, off)); hi = max(hi, __shfl_down(hi, off)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&slo[k], lo); atomicMax(&shi[k], hi); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {   // the arithmetic of cell_geom_kernel
        float lo[3], ext = 0;
        for (int k = 0; k < 3; k++) { lo[k] = ordered_float(slo[k]) - 1e-3f; ext = fmaxf(ext, ordered_float(shi[k]) + 1e-3f - lo[k]); }
        const float edge = fmaxf((float)B.edge_min, ext / (float)AWSEM_CELL_MAXDIM * 1.0001f);
        for (int k = 0; k < 3; k++) {
            int nd = (int)floorf((ordered_float(shi[k]) + 1e-3f - lo[k]) / edge) + 1;
            geo.nd[k] = nd > AWSEM_CELL_MAXDIM ? AWSEM_CELL_MAXDIM : nd;
            geo.lo[k] = (real)lo[k];
        }
        geo.inv = (real)(1.0f / edge);
        geo.ncell = geo.nd[0] * geo.nd[1] * geo.nd[2];
        if (g == 0) {
            for (int k = 0; k < 3; k++) { B.grid[4 * rep + k] = geo.lo[k]; B.dims[4 * rep + k] = geo.nd[k]; }
            B.grid[4 * rep + 3] = geo.inv; B.dims[4 * rep + 3] = geo.ncell;
        }
    }
    __syncthreads();
    const int ncell = geo.ncell, nx = geo.nd[0], ny = geo.nd[1];
    int* cnt = B.cell_count + (size_t)rep * AWSEM_CELL_MAXCELLS;
    int* start = B.cell_start + (size_t)rep * (AWSEM_CELL_MAXCELLS + 1);
    int* ts = B.tile_sum + (size_t)rep * (AWSEM_CELL_MAXCELLS / 1024);
    // ---- count: cell of every member, cell and tile populations, new reference positions
    {
        const real* p = pos + (size_t)rep * 3 * n;
        real* q = C.ref + (size_t)rep * 3 * n;
        for (int m = g * blockDim.x + threadIdx.x; m < B.n; m += G * blockDim.x) {
            const int a = B.atom[m];
            const real x = p[a], y = p[n + a], z = p[2 * (size_t)n + a];
            q[a] = x; q[n + a] = y; q[2 * (size_t)n + a] = z;
            int c[3];
            cell_coords_geo(geo, x, y, z, c);
            const int cell = (c[2] * ny + c[1]) * nx + c[0];
            B.cell_of[(size_t)rep * B.n + m] = cell;
            atomicAdd(&cnt[cell], 1);
            atomicAdd(&hist[cell >> 10], 1);
        }
        __syncthreads();
        for (int t = threadIdx.x; t < AWSEM_CELL_MAXCELLS / 1024; t += blockDim.x) if (hist[t]) atomicAdd(&ts[t], hist[t]);
    }
    if (!cell_grid_barrier(&bar[0], (unsigned)G, &bar[2], C.status, &sh_ok)) return;
    // ---- scan: exclusive prefix of the cell populations, a tile of 1024 cells at a time (four cells per thread); the counts go
    // back to zero for the fill pass
    for (int tile = g; tile * 1024 < ncell; tile += G) {
        if (threadIdx.x == 0) sh_base = 0;
        __syncthreads();
        int before = (int)threadIdx.x < tile ? ts[threadIdx.x] : 0;   // at most 256 tiles
        for (int off = 32; off > 0; off >>= 1) before += __shfl_down(before, off);
        if ((threadIdx.x & 63) == 0 && before) atomicAdd(&sh_base, before);
        const int c0 = tile * 1024 + 4 * threadIdx.x;
        int v[4], mine = 0;
        for (int u = 0; u < 4; u++) { v[u] = c0 + u < ncell ? cnt[c0 + u] : 0; mine += v[u]; }
        int incl = mine;
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int run = sh_base + incl - mine;
        for (int w = 0; w < wave; w++) run += wsum[w];
        for (int u = 0; u < 4; u++) {
            if (c0 + u < ncell) { start[c0 + u] = run; cnt[c0 + u] = 0; }
            run += v[u];
            if (c0 + u == ncell - 1) start[ncell] = run;
        }
        __syncthreads();
    }
    if (!cell_grid_barrier(&bar[1], (unsigned)G, &bar[0], C.status, &sh_ok)) return;
    // ---- fill: members into cell order
    for (int m = g * blockDim.x + threadIdx.x; m < B.n; m += G * blockDim.x) {
        const int cell = B.cell_of[(size_t)rep * B.n + m];
        B.sorted[(size_t)rep * B.n + start[cell] + atomicAdd(&cnt[cell], 1)] = m;
    }
    // Nobody waits for the fill (the kernel boundary orders the rows after it); the third counter only closes the cycle of counter
    // resets: whoever arrives LAST knows that everyone has passed barrier 1 and puts its counter back to zero.  (Counter 2 itself
    // goes back to zero behind barrier 0 of the next rebuild.)
    __syncthreads();
    if (threadIdx.x == 0 && __hip_atomic_fetch_add(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)G - 1u)
        __hip_atomic_store(&bar[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// grid (AWSEM_SORT_G, 2 R): the sort as a launch of its own (systems without window-local bonded terms; the evaluation API)
template <class real> __global__ __launch_bounds__(256) void cell_sort_kernel(CellSys<real> C, const real* __restrict__ pos, int cur, int force) {
    cell_sort_body(C, pos, (int)blockIdx.y >> 1, (int)blockIdx.y & 1, (int)blockIdx.x, (int)gridDim.x, cur, force);
}
template <bool ENERGY, class real, class Desc> __device__ __forceinline__ void hb_eval_body(const Desc& d, const CellSys<real>& C, const HbRows& H, const real* __restrict__ pos, const Rec<real>* __restrict__ pos4, int rep, int bx, int* hb_key, u64* hb_acc) {
    CellViewHb<real> L(cell_view_plain(C, pos, rep), pos4 + (size_t)rep * C.natoms, hb_key, hb_acc);
    L.table_clear();
    const int sub = threadIdx.x & (AWSEM_ROW - 1);
    const unsigned* rows = H.rows + (size_t)rep * d.nacc * H.stride;
    const int* rc = H.row_count + (size_t)rep * d.nacc;
    Energy e;
    for (int k = (bx * blockDim.x + threadIdx.x) / AWSEM_ROW; k < d.nacc; k += gridDim.x * blockDim.x / AWSEM_ROW) {
        const int cnt = rc[k];
        if (cnt == 0) continue;
        const Int4 A = d.acc[k];
        for (int s = sub; s < cnt; s += AWSEM_ROW) hb_pair<ENERGY>(d, L, A, d.don[rows[(size_t)k * H.stride + s]], e);
    }
    L.table_flush();
    flush_energy<ENERGY>(L, d.group, e);
}
template <bool ENERGY, class real> __global__ void cell_hb_eval_kernel(HbTerms<real> T, CellSys<real> C, const real* __restrict__ pos, unsigned mask,
                                                                       int rows_cur, int rows_force) {
    const int rep = blockIdx.y, t = blockIdx.z;
    const HbRows H = T.rows[t];
    const bool build = rows_cur >= 0 && cell_rebuild(C, rep, rows_cur, rows_force);   // as in cell_density_excl_kernel
    const int bx = xcd_block((int)blockIdx.x, (int)gridDim.x);   // acceptors are in chain order and a chain is a blob in space: a contiguous range per XCD
    __shared__ int hb_key[AWSEM_HB_SLOTS];       // the workgroup's table of forces (CellViewHb)
    __shared__ u64 hb_acc[3 * AWSEM_HB_SLOTS];
    if (t < T.n_beta) {
        const BetaD<real> d = T.beta[t];
        if (build) { hb_rows_body(d, C, H, T.pos4, T.rc2[t], rep, bx); __syncthreads(); }
        if (AWSEM_ON(d)) hb_eval_body<ENERGY>(d, C, H, pos, T.pos4, rep, bx, hb_key, hb_acc);
    } else {
        const PapD<real> d = T.pap[t - T.n_beta];
        if (build) { hb_rows_body(d, C, H, T.pos4, T.rc2[t], rep, bx); __syncthreads(); }
        if (AWSEM_ON(d)) hb_eval_body<ENERGY>(d, C, H, pos, T.pos4, rep, bx, hb_key, hb_acc);
    }
}

// Bonded terms: one workgroup per 256 items of one term (job table built at finalize).
struct CellJob { int kind, term, first; };
enum { CJ_HARM, CJ_CHI, CJ_RAMA, CJ_HELICAL, CJ_FM, CJ_MEMB, CJ_DH, CJ_GAUSS, CJ_CV_SUM, CJ_CV_APPLY };
template <bool ENERGY, class real> __global__ void cell_bonded_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos, const CellJob* __restrict__ jobs, unsigned mask) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const CellJob job = jobs[blockIdx.x];
    const int f = job.first;
    switch (job.kind) {
    case CJ_HARM: { HarmD<real> d = S.harm[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_harmonic<ENERGY>(d, L); } break;
    case CJ_CHI: { ChiD<real> d = S.chi[job.term]; if (!AWSEM_ON(d)) return; d.atoms += f; d.n = min(256, d.n - f); term_chi<ENERGY>(d, L); } break;
    case CJ_RAMA: { RamaD<real> d = S.rama[job.term]; if (!AWSEM_ON(d)) return; d.atoms += f; d.bw += (size_t)f * d.nwell; d.n = min(256, d.n - f); term_rama<ENERGY>(d, L); } break;
    case CJ_HELICAL: { HelicalD<real> d = S.helical[job.term]; if (!AWSEM_ON(d)) return; d.items += f; if (d.zitems) d.zitems += f; d.n = min(256, d.n - f); term_helical<ENERGY>(d, L); } break;
    case CJ_FM: { FmD<real> d = S.fm[job.term]; if (!AWSEM_ON(d)) return; d.pairs2 += 2 * (size_t)f; d.table2 += (size_t)f * (d.ntab - 1); d.n = min(256, d.n - f); term_fm<ENERGY, false>(d, L); } break;
    case CJ_MEMB: { MembD<real> d = S.memb[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_memb<ENERGY>(d, L); } break;
    case CJ_DH: { DhD<real> d = S.dh[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_dh<ENERGY>(d, L); } break;
    case CJ_GAUSS: { GaussD<real> d = S.gauss[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); term_gauss<ENERGY>(d, L); } break;
    // collective-variable terms: the sum in a launch of its own (cell_cv_kernel), then energy (first job only) and forces
    case CJ_CV_SUM: { GaussD<real> d = S.gauss[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f); cv_sum(d, L); } break;
    case CJ_CV_APPLY: { GaussD<real> d = S.gauss[job.term]; if (!AWSEM_ON(d)) return; d.items += f; d.n = min(256, d.n - f);
                        if (f == 0) cv_apply<ENERGY>(d, L); else cv_apply<false>(d, L); } break;
    }
}

template <class T> __global__ void cell_set_word_kernel(T* p, T v) { *p = v; }

// ------------------------------------------------------------------ outputs, virtual sites, integrator
template <class real> __global__ void cell_vsite_kernel(DevSys<real> S, real* __restrict__ pos) {
    const int n = S.natoms;
    real* p = pos + (size_t)blockIdx.y * 3 * n;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S.nvs; i += gridDim.x * blockDim.x) {
        const Int4 v = S.vs[i];
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        for (int k = 0; k < 3; k++) p[(size_t)k * n + v.x] = w0 * p[(size_t)k * n + v.y] + w1 * p[(size_t)k * n + v.z] + w2 * p[(size_t)k * n + v.w];
    }
}
// evaluation API only (getState(getForces=True) reports the forces folded onto the parents)
template <class real> __global__ void cell_redistribute_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos) {
    const int rep = blockIdx.y;
    const CellView<real> L = cell_view(S, C, pos, rep);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < S.nvs; i += gridDim.x * blockDim.x) {
        const Int4 v = S.vs[i];
        const R3<real> F = L.force(v.x);
        L.zero_force(v.x);   // only this lane touches the accumulators of its virtual site here
        const real w0 = S.vs_w[3 * i], w1 = S.vs_w[3 * i + 1], w2 = S.vs_w[3 * i + 2];
        L.add_force(v.y, w0 * F.x, w0 * F.y, w0 * F.z);
        L.add_force(v.z, w1 * F.x, w1 * F.y, w1 * F.z);
        L.add_force(v.w, w2 * F.x, w2 * F.y, w2 * F.z);
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
// Centre-of-mass motion removal on the cell-list path.  A same-address global atomic from every workgroup costs ~0.1 us each
// (DESIGN.md section 8), so the integrator's workgroups leave their momentum sums side by side -- C.cm[parity][rep][workgroup][3],
// plain stores -- and the next step's integrator adds them up again in every workgroup (a few KB out of L2).  Q31.32 integers:
// the total is the same whatever the grid.  cell_block_sum: sum of three per-thread values over the workgroup, to every thread.
__device__ __forceinline__ void cell_block_sum(u64 s[3], u64* sh /* [3][16] shared */) {
    for (int q = 0; q < 3; q++) {
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off);
        if ((threadIdx.x & 63) == 0) sh[q * 16 + (threadIdx.x >> 6)] = s[q];
    }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int q = 0; q < 3; q++) { u64 t = 0; for (int w = 0; w < nw; w++) t += sh[q * 16 + w]; s[q] = t; }
    __syncthreads();
}
// momentum of the velocities as they stand (first step of an awsem_step call): partial sums into the parity the first step reads
template <class real>
__global__ void cell_momentum_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ vel, const long long* __restrict__ step_base) {
    __shared__ u64 sh[48];
    const int rep = blockIdx.y, n = C.natoms, par = (int)(*step_base & 1);
    const real* v = vel + (size_t)rep * 3 * n;
    u64 s[3] = {0, 0, 0};
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S.nmassive; k += gridDim.x * blockDim.x) {
        const MassiveItem<real> mi = S.massive[k];
        const double m = 1.0 / (double)mi.inv_mass;
        s[0] += (u64)__double2ll_rn(m * (double)v[mi.atom] * AWSEM_FIXED_SCALE);
        s[1] += (u64)__double2ll_rn(m * (double)v[n + mi.atom] * AWSEM_FIXED_SCALE);
        s[2] += (u64)__double2ll_rn(m * (double)v[2 * (size_t)n + mi.atom] * AWSEM_FIXED_SCALE);
    }
    cell_block_sum(s, sh);
    if (threadIdx.x < 3) C.cm[(((size_t)par * C.nrep + rep) * C.nblk_int + blockIdx.x) * 3 + threadIdx.x] = s[threadIdx.x];
}
// Langevin update; every massive particle gathers the forces of the virtual sites it parents (no atomics) and clears
// its own accumulator (those of the sites are cleared by the next step's pre kernel, once every parent has read them).
// next_cur >= 0 (round 3): the integrator also does what the NEXT step's pre kernel would do -- member records, displacement flag
// (written to flag[next_cur]; flag[next_cur ^ 1], which this step's rebuild has read, is reset), per-workgroup bounding boxes -- and
// clears the accumulators of the virtual sites and the massless particles in the OTHER parity's force array `f_other`, the one the next
// step sums into (this step's are still being gathered by the sites' parents in other workgroups).
template <class real>
__global__ void cell_integrate_kernel(DevSys<real> S, CellSys<real> C, real* __restrict__ pos, real* __restrict__ vel, const double* __restrict__ T,
                                      double dt, double friction, unsigned long long seed, const long long* __restrict__ step_base, int step_offset,
                                      int next_cur, u64* __restrict__ f_other) {
    const long long step = *step_base + step_offset;   // the base lives in device memory so that a captured graph of several steps can be replayed
    const int rep = blockIdx.y, n = C.natoms;
    const CellView<real> L = cell_view(S, C, pos, rep);
    real* v = vel + (size_t)rep * 3 * n;
    const double a_d = ::exp(-friction * dt);
    const real a = (real)a_d, fs = (real)(friction == 0.0 ? dt : (1.0 - a_d) / friction);
    const real ns = (real)::sqrt(AWSEM_KB * T[rep] * (1.0 - a_d * a_d)), dtr = (real)dt;
    __shared__ u64 sh[48];
    const bool cmr = S.remove_cm != 0;   // uniform
    const int par = (int)(step & 1);
    real vcx = 0, vcy = 0, vcz = 0;
    u64 mom[3] = {0, 0, 0};
    if (cmr) {   // total momentum = sum of the partial sums every workgroup of the previous step (or cell_momentum_kernel) left
        const u64* part = C.cm + ((size_t)par * C.nrep + rep) * C.nblk_int * 3;
        u64 s[3] = {0, 0, 0};
        for (int b = threadIdx.x; b < C.nblk_int; b += blockDim.x) { s[0] += part[3 * b]; s[1] += part[3 * b + 1]; s[2] += part[3 * b + 2]; }
        cell_block_sum(s, sh);
        const double f = (double)S.inv_total_mass / AWSEM_FIXED_SCALE;
        vcx = (real)((double)(long long)s[0] * f); vcy = (real)((double)(long long)s[1] * f); vcz = (real)((double)(long long)s[2] * f);
    }
    const bool pre = next_cur >= 0;   // uniform
    if (pre) {   // first, so that these stores and the loads behind them are under way while the particles are integrated
        u64* fo = f_other + (size_t)rep * 3 * n;
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < S.nvs; t += gridDim.x * blockDim.x) { const int a = S.vs[t].x; fo[a] = 0; fo[n + a] = 0; fo[2 * (size_t)n + a] = 0; }
        for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < S.nfixed; t += gridDim.x * blockDim.x) { const int a = S.fixed[t]; fo[a] = 0; fo[n + a] = 0; fo[2 * (size_t)n + a] = 0; }
    }
    bool moved = false;
    float blo[3] = {3e38f, 3e38f, 3e38f}, bhi[3] = {-3e38f, -3e38f, -3e38f};
    const real* qref = C.ref + (size_t)rep * 3 * n;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S.nmassive; k += gridDim.x * blockDim.x) {
        const MassiveItem<real> mi = S.massive[k];
        const GatherItem<real> gi = S.gather[k];
        const int i = mi.atom;
        R3<real> F = L.force(i);
        L.zero_force(i);
#pragma unroll
        for (int q = 0; q < AWSEM_MAX_GATHER; q++) {
            if (gi.vs[q] < 0) continue;
            const R3<real> fv = L.force(gi.vs[q]);
            F.x += gi.w[q] * fv.x; F.y += gi.w[q] * fv.y; F.z += gi.w[q] * fv.z;
        }
        real g[3];
        gauss3<real>(seed, AWSEM_STREAM_LANGEVIN, rep, step, mi.id, g);
        const real im = fs * mi.inv_mass, sq = ns * mi.sqrt_inv_mass;
        const real vx = a * (v[i] - vcx) + im * F.x + sq * g[0];
        const real vy = a * (v[n + i] - vcy) + im * F.y + sq * g[1];
        const real vz = a * (v[2 * (size_t)n + i] - vcz) + im * F.z + sq * g[2];
        v[i] = vx; v[n + i] = vy; v[2 * (size_t)n + i] = vz;
        const real x = L.px[i] + dtr * vx, y = L.py[i] + dtr * vy, z = L.pz[i] + dtr * vz;
        L.px[i] = x; L.py[i] = y; L.pz[i] = z;
        if (cmr) {
            const double m = 1.0 / (double)mi.inv_mass;
            mom[0] += (u64)__double2ll_rn(m * (double)vx * AWSEM_FIXED_SCALE);
            mom[1] += (u64)__double2ll_rn(m * (double)vy * AWSEM_FIXED_SCALE);
            mom[2] += (u64)__double2ll_rn(m * (double)vz * AWSEM_FIXED_SCALE);
        }
        if (pre) {   // list members are massive particles (checked at finalize): every member passes through here
            const Int4 pi = C.pre_items[k];   // one coalesced record instead of three dependent gathers
            const int m0 = pi.x, m1 = pi.y;
            if (m0 >= 0 || m1 >= 0) {
                moved = moved || (float)dist2(x - qref[i], y - qref[n + i], z - qref[2 * (size_t)n + i]) > (float)C.skin_half2;
                blo[0] = fminf(blo[0], (float)x); bhi[0] = fmaxf(bhi[0], (float)x);
                blo[1] = fminf(blo[1], (float)y); bhi[1] = fmaxf(bhi[1], (float)y);
                blo[2] = fminf(blo[2], (float)z); bhi[2] = fmaxf(bhi[2], (float)z);
                if (m0 >= 0) ((Rec<real>*)C.bin[0].rec + (size_t)rep * C.bin[0].n)[m0] = Rec<real>{x, y, z, make_tag(pi.z, (real)0)};
                if (m1 >= 0) ((Rec<real>*)C.bin[1].rec + (size_t)rep * C.bin[1].n)[m1] = Rec<real>{x, y, z, make_tag(i, (real)0)};
            }
        }
    }
    if (pre) {
        __shared__ int slo[3], shi[3];
        if (threadIdx.x < 3) { slo[threadIdx.x] = 0x7fffffff; shi[threadIdx.x] = (int)0x80000000; }
        __syncthreads();
        if (__ballot(moved) && (threadIdx.x & 63) == 0) C.flag[2 * rep + next_cur] = 1u;   // same value from every writer: no atomic needed
        for (int k = 0; k < 3; k++) {
            for (int off = 32; off > 0; off >>= 1) { blo[k] = fminf(blo[k], __shfl_down(blo[k], off)); bhi[k] = fmaxf(bhi[k], __shfl_down(bhi[k], off)); }
            if ((threadIdx.x & 63) == 0 && blo[k] <= bhi[k]) { atomicMin(&slo[k], ordered_int(blo[k])); atomicMax(&shi[k], ordered_int(bhi[k])); }
        }
        __syncthreads();
        int* bb = C.bbox + ((size_t)rep * C.nblk_box + blockIdx.x) * 8;
        if (threadIdx.x < 3) { bb[threadIdx.x] = slo[threadIdx.x]; bb[4 + threadIdx.x] = shi[threadIdx.x]; }
        if (blockIdx.x == 0 && threadIdx.x == 0) C.flag[2 * rep + (next_cur ^ 1)] = 0;
        if (blockIdx.x == 0 && (int)threadIdx.x < C.n_cv) C.cv[(size_t)rep * C.n_cv + threadIdx.x] = 0;
    }
    if (cmr) {   // this workgroup's share of the new momentum, for the next step (the other parity: nobody reads it during this kernel)
        cell_block_sum(mom, sh);
        if (threadIdx.x < 3) C.cm[(((size_t)(par ^ 1) * C.nrep + rep) * C.nblk_int + blockIdx.x) * 3 + threadIdx.x] = mom[threadIdx.x];
    }
}

// ------------------------------------------------------------------ energy minimiser on the cell-list path
// The FIRE iteration of awsem_minimize_kernel (awsem_kernels.h) spread over grid-wide kernels: after a force pass,
// cell_fire_sums_kernel reduces F.v, F.F, v.v over the massive particles of every replica (fixed point: order-independent),
// cell_fire_update_kernel takes the per-replica decision -- every thread derives the same one from the sums and the
// previous state, so no broadcast is needed -- and moves the particles.  State and sums are double buffered by iteration
// parity, so nobody reads what another workgroup is writing.
template <class real> struct FireState { real dt, alpha; int npos, done, iters, pad; double rms; };

template <class real> __device__ __forceinline__ R3<real> fire_force(const DevSys<real>& S, const CellView<real>& L, int k, int i) {
    const GatherItem<real> gi = S.gather[k];
    R3<real> F = L.force(i);
#pragma unroll
    for (int q = 0; q < AWSEM_MAX_GATHER; q++) {
        if (gi.vs[q] < 0) continue;
        const R3<real> fv = L.force(gi.vs[q]);
        F.x += gi.w[q] * fv.x; F.y += gi.w[q] * fv.y; F.z += gi.w[q] * fv.z;
    }
    return F;
}
#define AWSEM_FIRE_SP (1.0 / 1024.0)
#define AWSEM_FIRE_SF (1.0 / (1024.0 * 1024.0))
template <class real> __global__ void cell_fire_sums_kernel(DevSys<real> S, CellSys<real> C, const real* __restrict__ pos, const real* __restrict__ vel,
                                                            u64* __restrict__ sums /* [R][3] of this parity */, const FireState<real>* __restrict__ state) {
    const int rep = blockIdx.y, n = C.natoms;
    if (state[rep].done) return;
    const CellView<real> L = cell_view(S, C, pos, rep);
    const real* v = vel + (size_t)rep * 3 * n;
    Energy eP, eF, eV;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S.nmassive; k += gridDim.x * blockDim.x) {
        const int i = S.massive[k].atom;
        const R3<real> F = fire_force(S, L, k, i);
        const real vx = v[i], vy = v[n + i], vz = v[2 * (size_t)n + i];
        eP += (double)(F.x * vx + F.y * vy + F.z * vz) * AWSEM_FIRE_SP;
        eF += (double)(F.x * F.x + F.y * F.y + F.z * F.z) * AWSEM_FIRE_SF;
        eV += (double)(vx * vx + vy * vy + vz * vz);
    }
    __shared__ u64 part[3];
    if (threadIdx.x < 3) part[threadIdx.x] = 0;
    __syncthreads();
    u64 s[3] = {eP.v, eF.v, eV.v};
    for (int q = 0; q < 3; q++) {
        for (int off = 32; off > 0; off >>= 1) s[q] += __shfl_down(s[q], off);
        if ((threadIdx.x & 63) == 0 && s[q] != 0) fixed_add(&part[q], s[q]);
    }
    __syncthreads();
    if (threadIdx.x < 3 && part[threadIdx.x] != 0) fixed_add(&sums[3 * rep + threadIdx.x], part[threadIdx.x]);   // one atomic per workgroup and sum
}
template <class real> __global__ void cell_fire_update_kernel(DevSys<real> S, CellSys<real> C, real* __restrict__ pos, real* __restrict__ vel,
                                                              const u64* __restrict__ sums, u64* __restrict__ sums_next,
                                                              const FireState<real>* __restrict__ state, FireState<real>* __restrict__ state_next, double tolerance) {
    const int rep = blockIdx.y, n = C.natoms;
    const bool writer = blockIdx.x == 0 && threadIdx.x == 0;
    if (blockIdx.x == 0 && threadIdx.x < 3) sums_next[3 * rep + threadIdx.x] = 0;   // the other parity: idle during this kernel
    FireState<real> st = state[rep];
    if (st.done) { if (writer) state_next[rep] = st; return; }
    const double P = fixed_to_double(sums[3 * rep]) / AWSEM_FIRE_SP, FF = fixed_to_double(sums[3 * rep + 1]) / AWSEM_FIRE_SF, VV = fixed_to_double(sums[3 * rep + 2]);
    st.rms = ::sqrt(FF / (3.0 * (S.nmassive > 0 ? S.nmassive : 1)));
    if (st.rms <= tolerance) { st.done = 1; if (writer) state_next[rep] = st; return; }   // the coordinates stay as they are
    const real dt_max = (real)0.01, dmax2 = (real)(0.02 * 0.02);
    real keep = 0, mix = 0;
    if (P >= 0) {
        keep = (real)1 - st.alpha; mix = st.alpha * (real)::sqrt(VV / FF);
        if (++st.npos > 5) { st.dt = st.dt * (real)1.1 < dt_max ? st.dt * (real)1.1 : dt_max; st.alpha *= (real)0.99; }
    } else { st.npos = 0; st.dt *= (real)0.5; st.alpha = (real)0.1; }
    st.iters++;
    if (writer) state_next[rep] = st;
    const real dt = st.dt;
    const CellView<real> L = cell_view(S, C, pos, rep);
    real* v = vel + (size_t)rep * 3 * n;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < S.nmassive; k += gridDim.x * blockDim.x) {
        const int i = S.massive[k].atom;
        const R3<real> F = fire_force(S, L, k, i);
        L.zero_force(i);   // the sites' accumulators are cleared by the next pre kernel
        const real vx = keep * v[i] + mix * F.x + dt * F.x;
        const real vy = keep * v[n + i] + mix * F.y + dt * F.y;
        const real vz = keep * v[2 * (size_t)n + i] + mix * F.z + dt * F.z;
        v[i] = vx; v[n + i] = vy; v[2 * (size_t)n + i] = vz;
        real dx = dt * vx, dy = dt * vy, dz = dt * vz;
        const real d2 = dx * dx + dy * dy + dz * dz;
        if (d2 > dmax2) { const real sc = M<real>::sqrt(dmax2 / d2); dx *= sc; dy *= sc; dz *= sc; }
        L.px[i] += dx; L.py[i] += dy; L.pz[i] += dz;
    }
}
