// Microbenchmark: throughput of LDS 64-bit / 32-bit atomics and plain LDS ops under different address patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
#define N 1104
template <int MODE, class T>
__global__ void k(const int* __restrict__ idx, long long* out, int iters) {
    __shared__ T acc[N * 3];
    for (int i = threadIdx.x; i < N * 3; i += blockDim.x) acc[i] = 0;
    __syncthreads();
    int a[8];
    for (int u = 0; u < 8; u++) a[u] = idx[(MODE * 8 + u) * 256 + threadIdx.x];
    long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) atomicAdd(&acc[a[u]], (T)(it + u));
    }
    __syncthreads();
    long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc[threadIdx.x] == (T)12345) out[0] = 0;
}
int main() {
    // modes: 0 consecutive, 1 random in 1104, 2 stride 6, 3 all same, 4 random in 3*1104
    std::vector<int> h(5 * 8 * 256);
    srand(1);
    for (int m = 0; m < 5; m++) for (int u = 0; u < 8; u++) for (int t = 0; t < 256; t++) {
        int v;
        if (m == 0) v = (t + 64 * u) % N;
        else if (m == 1) v = rand() % N;
        else if (m == 2) v = (6 * t + u) % N;
        else if (m == 3) v = u;
        else v = rand() % (3 * N);
        h[(m * 8 + u) * 256 + t] = v;
    }
    int* d; long long* o;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o, 8 * 1024);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int iters = 2000;
    const char* names[5] = {"consecutive", "random", "stride6", "same8", "random3N"};
    for (int nblk : {1, 4}) {   // 1 block (4 waves) or 4 blocks per CU sharing the LDS pipe: launch 256*nblk blocks
        for (int m = 0; m < 5; m++) {
            long long r64 = 0, r32 = 0;
            for (int rep = 0; rep < 2; rep++) {
                switch (m) {
#define RUN(M) case M: hipLaunchKernelGGL((k<M, u64>), dim3(256 * nblk), dim3(256), 0, 0, d, o, iters); hipMemcpy(&r64, o, 8, hipMemcpyDeviceToHost); \
                       hipLaunchKernelGGL((k<M, unsigned>), dim3(256 * nblk), dim3(256), 0, 0, d, o, iters); hipMemcpy(&r32, o, 8, hipMemcpyDeviceToHost); break;
                    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
                }
            }
            // per block: 4 waves x iters x 8 atomics; with nblk blocks per CU the CU executes nblk x that in the same time
            double per64 = (double)r64 / (iters * 8.0 * 4 * nblk), per32 = (double)r32 / (iters * 8.0 * 4 * nblk);
            printf("blocks/CU %d  %-12s  u64: %.1f clk per wave-instr (CU-level)   u32: %.1f\n", nblk, names[m], per64, per32);
        }
    }
    return 0;
}
