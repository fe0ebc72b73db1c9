// micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 on one SIMD (cycles per MFMA, one and four accumulators), with and
// without a second wave on the same SIMD that runs dependent v_fma_f64 (do the matrix and the vector pipe overlap for f64?).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64 mfma_f64.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 2048
// MODE 0: every wave runs MFMAs (NACC accumulators). MODE 1: waves 0..3 run MFMAs, waves 4..7 (second wave of each SIMD) run v_fma_f64.
// MODE 2: every wave runs v_fma_f64 only.
template <int NACC, int MODE>
__global__ void k(double* out, double din, unsigned long long* cyc) {
    const int wave = threadIdx.x >> 6;
    d4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    double a = din + threadIdx.x * 1e-3, b = din * 1.0001, c = 0.5;
    unsigned long long t0 = __builtin_readcyclecounter();
    const bool do_mfma = MODE == 0 || (MODE == 1 && wave < 4);
    if (do_mfma) {
#pragma unroll 8
        for (int i = 0; i < N; ++i) acc[i % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i % NACC], 0, 0, 0);
    } else {
#pragma unroll 16
        for (int i = 0; i < N * 8; ++i) c = fma(c, b, a);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    double s = c;
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
}
template <int NACC, int MODE> void run(const char* name, int threads) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 8 * 16); hipMemset(cyc, 0, 128);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, out, 1.25, cyc);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(threads), 0, 0, out, 1.25, cyc);
    unsigned long long h[16]; hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
    printf("%-52s %4d thr: wave0 %8.2f cyc per MFMA (or per 8 FMA)", name, threads, (double)h[0] / N);
    if (threads > 256) printf("   wave4 %8.2f cyc per 8 v_fma_f64", (double)h[4] / N);
    printf("\n");
    hipFree(out); hipFree(cyc);
}
int main() {
    run<1, 0>("mfma_f64_16x16x4, 1 accumulator, 1 wave/SIMD", 256);
    run<4, 0>("mfma_f64_16x16x4, 4 accumulators, 1 wave/SIMD", 256);
    run<4, 0>("mfma_f64_16x16x4, 4 accumulators, 2 waves/SIMD", 512);
    run<4, 2>("v_fma_f64 dependent only, 1 wave/SIMD", 256);
    run<4, 2>("v_fma_f64 dependent only, 2 waves/SIMD", 512);
    run<4, 1>("mfma (waves 0-3) beside v_fma_f64 (waves 4-7)", 512);
    return 0;
}
