// micro-benchmark: cost of a wave64 gather through the vector L1 (TA/TCP) by address pattern, data resident in L2
// (8 arrays of 10 k float4 = 160 KB, one per XCD), 16 waves per CU like the frame kernel (2 blocks x 512 threads).
// Reported: CU clock cycles per wave-level load instruction (all 16 waves of the CU issue concurrently).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather gather.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define NP 10000
#define ITERS 2048
enum { DW_RANDOM, DW_LANESEQ, X2_RANDOM, X4_RANDOM, X4_PAIRS, X4_QUADS, X4_OCTS, X4_COALESCED, X4_LANESEQ, X4_LANESEQ_PAIRS, DW_COALESCED, NPAT };
template <int PAT>
__global__ __launch_bounds__(512) void k(const float4* __restrict__ base, float* out, unsigned long long* cyc) {
    const float4* a4 = base + (size_t)(blockIdx.x % 8) * NP;
    const float* a1 = (const float*)a4;
    const float2* a2 = (const float2*)a4;
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    const unsigned lane = threadIdx.x & 63u;
    float acc = 0.f;
    unsigned seq = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITERS; i += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = s * 1664525u + 1013904223u;
            unsigned r = (s >> 8);
            unsigned ws = __builtin_amdgcn_readfirstlane(s) >> 8;   // wave-uniform random
            v[u] = float4{0, 0, 0, 0};
            if (PAT == DW_RANDOM) v[u].x = a1[r % (4 * NP)];
            if (PAT == DW_COALESCED) v[u].x = a1[(ws % (4 * NP - 64)) + lane];
            if (PAT == DW_LANESEQ) { if (((i + u) & 31) == 0) seq = r % (4 * NP - 32); v[u].x = a1[seq + ((i + u) & 31)]; }
            if (PAT == X2_RANDOM) { float2 t = a2[r % (2 * NP)]; v[u].x = t.x; v[u].y = t.y; }
            if (PAT == X4_RANDOM) v[u] = a4[r % NP];
            if (PAT == X4_PAIRS) { unsigned rr = __shfl(r, lane & ~1u); v[u] = a4[(rr % (NP - 2)) + (lane & 1u)]; }
            if (PAT == X4_QUADS) { unsigned rr = __shfl(r, lane & ~3u); v[u] = a4[(rr % (NP - 4)) + (lane & 3u)]; }
            if (PAT == X4_OCTS) { unsigned rr = __shfl(r, lane & ~7u); v[u] = a4[(rr % (NP - 8)) + (lane & 7u)]; }
            if (PAT == X4_COALESCED) v[u] = a4[(ws % (NP - 64)) + lane];
            if (PAT == X4_LANESEQ) { if (((i + u) & 31) == 0) seq = r % (NP - 32); v[u] = a4[seq + ((i + u) & 31)]; }
            if (PAT == X4_LANESEQ_PAIRS) { if (((i + u) & 15) == 0) { unsigned rr = __shfl(r, lane & ~1u); seq = rr % (NP - 32); } v[u] = a4[seq + 2 * ((i + u) & 15) + (lane & 1u)]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int PAT> void run(const char* name, const float4* base, float* out, unsigned long long* cyc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(512), dim3(512), 0, 0, base, out, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(512), dim3(512), 0, 0, base, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // 512 blocks of 8 waves over 256 CUs = 16 waves per CU, each issuing ITERS loads
    printf("%-44s wave0: %8.1f clk per own load  => %6.2f CU-clk per wave-load;  kernel %.3f ms => %.2f ns per wave-load per CU\n", name,
           (double)h / ITERS, (double)h / ITERS / 16.0, ms, ms * 1e6 / (ITERS * 16.0));
}
int main() {
    float4* base; float* out; unsigned long long* cyc;
    hipMalloc(&base, sizeof(float4) * NP * 8); hipMemset(base, 0, sizeof(float4) * NP * 8);
    hipMalloc(&out, 4 * 512 * 512); hipMalloc(&cyc, 8);
    run<DW_RANDOM>("dword, every lane a random address", base, out, cyc);
    run<DW_LANESEQ>("dword, lane walks 32 consecutive dwords", base, out, cyc);
    run<DW_COALESCED>("dword, coalesced (64 consecutive)", base, out, cyc);
    run<X2_RANDOM>("dwordx2, every lane random", base, out, cyc);
    run<X4_RANDOM>("dwordx4, every lane random", base, out, cyc);
    run<X4_PAIRS>("dwordx4, lane pairs adjacent", base, out, cyc);
    run<X4_QUADS>("dwordx4, lane quads adjacent (64 B)", base, out, cyc);
    run<X4_OCTS>("dwordx4, 8 lanes adjacent (128 B)", base, out, cyc);
    run<X4_COALESCED>("dwordx4, coalesced (1 KB)", base, out, cyc);
    run<X4_LANESEQ>("dwordx4, lane walks 32 consecutive float4", base, out, cyc);
    run<X4_LANESEQ_PAIRS>("dwordx4, lane pair walks 32 consecutive", base, out, cyc);
    return 0;
}
