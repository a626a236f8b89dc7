// Microbenchmark for DESIGN.md section 3 ("small ensembles"): what would it cost to let K workgroups on K CUs of ONE XCD share a replica?
// Every step of such a split needs the partial sums of the K workgroups combined three times (contact densities, dE/drho, forces).
// An exchange here = every workgroup adds `nwords` 64-bit fixed-point words into the cluster's accumulator with L2 atomics, the K
// workgroups meet at a counter (bounded spin), everyone reads the sums back (L2 atomic reads or agent-scope loads) and clears its share.
// Launched cooperatively (256 workgroups of 1024 threads must be co-resident); every spin is bounded, so the grid always drains.
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/ubench/cluster_sync scratch/ubench/cluster_sync.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void exchange_kernel(u64* acc, unsigned* counters, int* failed, int K, int nwords, int iters, int read_mode, long long* clocks) {
    const int b = blockIdx.x, x = b & 7, rest = b >> 3, j = rest / K, k = rest % K;   // workgroups b, b + 8, ... share XCD x
    const int cluster = x + 8 * j;
    u64* A = acc + (size_t)cluster * 2 * nwords;            // double buffered by iteration parity
    unsigned* cnt = counters + cluster * 64;
    u64 sink = 0;
    __shared__ int stop;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; it++) {
        u64* a = A + (size_t)(it & 1) * nwords;
        for (int i = threadIdx.x; i < nwords; i += blockDim.x)
            (void)__hip_atomic_fetch_add(&a[i], (u64)(i + k + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();                                      // s_waitcnt vmcnt(0): this workgroup's adds have been performed at L2
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = (unsigned)K * (unsigned)(it + 1);
            int polls = 0;
            while (__hip_atomic_fetch_add(cnt, 0u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (++polls > 20000) { *failed = 1; stop = 1; break; }   // never reached in a co-resident launch; keeps the grid finite
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (stop) break;   // uniform: the partners are gone too, each after its own bounded spin
        for (int i = threadIdx.x; i < nwords; i += blockDim.x) {
            if (read_mode == 0) sink += __hip_atomic_fetch_add(&a[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else sink += __hip_atomic_load(&a[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // clear the OTHER buffer's share for the iteration after next (everyone is past reading it: it was read before this barrier)
        u64* o = A + (size_t)((it + 1) & 1) * nwords;
        for (int i = threadIdx.x + k * blockDim.x; i < nwords; i += K * blockDim.x) __hip_atomic_store(&o[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) clocks[b] = t1 - t0;
    if (sink == 0x123456789ull) acc[0] = sink;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    int ncu = 0;
    CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs %d\n", ncu);
    if (ncu < 256) { printf("needs 256 CUs\n"); return 0; }
    const int grid = 256, block = 1024;
    for (int K : {2, 4}) for (int nwords : {276, 2430}) for (int read_mode : {0, 1}) {
        const int nclusters = grid / K;
        u64* acc; unsigned* counters; int* failed; long long* clocks;
        CHECK(hipMalloc(&acc, (size_t)nclusters * 2 * nwords * sizeof(u64)));
        CHECK(hipMalloc(&counters, (size_t)nclusters * 64 * sizeof(unsigned)));
        CHECK(hipMalloc(&failed, sizeof(int)));
        CHECK(hipMalloc(&clocks, grid * sizeof(long long)));
        CHECK(hipMemset(acc, 0, (size_t)nclusters * 2 * nwords * sizeof(u64)));
        CHECK(hipMemset(counters, 0, (size_t)nclusters * 64 * sizeof(unsigned)));
        CHECK(hipMemset(failed, 0, sizeof(int)));
        int Kc = K, nw = nwords, itc = iters, rm = read_mode;
        void* args[] = {&acc, &counters, &failed, &Kc, &nw, &itc, &rm, &clocks};
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, 0));
        CHECK(hipLaunchCooperativeKernel((const void*)exchange_kernel, dim3(grid), dim3(block), args, 0, 0));
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        int f = 0; CHECK(hipMemcpy(&f, failed, sizeof f, hipMemcpyDeviceToHost));
        std::vector<long long> c(grid); CHECK(hipMemcpy(c.data(), clocks, grid * sizeof(long long), hipMemcpyDeviceToHost));
        long long mx = 0; for (long long v : c) mx = v > mx ? v : mx;
        printf("K %d  words %5d  read %s : %.2f us per exchange (events), %.2f us (slowest workgroup, wall clock)%s\n", K, nwords,
               read_mode == 0 ? "L2 atomic   " : "agent load  ", ms * 1e3 / iters, mx * 0.01 / iters, f ? "  SPIN LIMIT HIT" : "");
        CHECK(hipFree(acc)); CHECK(hipFree(counters)); CHECK(hipFree(failed)); CHECK(hipFree(clocks));
    }
    return 0;
}
