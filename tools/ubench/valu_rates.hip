// micro-benchmark: issue cost (cycles per wave-instruction) of the VALU ops the kd search is made of, on one
// wave per SIMD and on 4 waves per SIMD. Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 4096
template <int OP>
__global__ void k(double* out, float fin, double din, unsigned long long* cyc) {
    double a = din + threadIdx.x, b = din * 1.0001, c = 0;
    float f = fin + threadIdx.x;
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        if (OP == 0) { a = a + b; }                              // v_add_f64 (dependent)
        if (OP == 1) { a = a * b; }                              // v_mul_f64
        if (OP == 2) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(c) : "v"(f)); f += 1.0f; a += c; }   // cvt + add f32 + add f64
        if (OP == 3) { f = f + fin; }                            // v_add_f32
        if (OP == 4) { a = (a > b) ? a - b : a + b; }            // cmp_f64 + 2 adds + cndmask
        if (OP == 5) { a = fma(a, b, c); }                       // v_fma_f64
        if (OP == 6) { a = a / b; }                              // f64 division sequence
        if (OP == 7) { a = sqrt(a) + b; }                        // f64 sqrt sequence
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + f + c;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int OP> void run(const char* name, int threads) {
    double* out; unsigned long long* cyc; hipMalloc(&out, 8 * 1024 * 256); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, 1.5f, 1.25, cyc);
    hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 0, 0, out, 1.5f, 1.25, cyc);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-34s %4d thr/CU: %7.2f cycles per loop trip (wave 0)\n", name, threads, (double)h / N);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int th : {256, 1024}) {
        run<0>("v_add_f64 dependent", th); run<1>("v_mul_f64 dependent", th); run<5>("v_fma_f64 dependent", th);
        run<2>("cvt_f64_f32 + add_f32 + add_f64", th); run<3>("v_add_f32 dependent", th); run<4>("cmp_f64 + 2 add_f64 + cndmask", th);
        run<6>("f64 division", th); run<7>("f64 sqrt + add", th);
    }
    return 0;
}
