// micro-benchmark: latency of a DEPENDENT LDS read chain (pointer chasing, random addresses per lane) with 1..16
// waves per CU, 150 KB of dynamic LDS (one block per CU), i.e. the kd-tree descent pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 2048
__global__ void chase(unsigned* out, unsigned long long* cyc, int words, int same) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = (unsigned)((i * 2654435761u + 12345u) % (unsigned)words);
    __syncthreads();
    unsigned p = same ? 7u : (threadIdx.x * 977u) % (unsigned)words;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < N; ++i) p = lds[p];
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = p;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    unsigned* out; unsigned long long* cyc; hipMalloc(&out, 4 * 1024 * 256); hipMalloc(&cyc, 8);
    const int words = 150 * 1024 / 4;
    hipFuncSetAttribute((const void*)chase, hipFuncAttributeMaxDynamicSharedMemorySize, words * 4);
    for (int same : {0, 1})
        for (int th : {64, 256, 512, 1024}) {
            for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(chase, dim3(256), dim3(th), words * 4, 0, out, cyc, words, same);
            unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s addresses, %4d threads/CU: %7.1f cycles per dependent ds_read_b32\n", same ? "uniform" : "random ", th, (double)h / N);
        }
    return 0;
}
