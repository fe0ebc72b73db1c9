// iba_nn_list_kernel<WHICH, STEPS> (round 6; OPT-IN, IBA_NN_LIST=1 with IBA_DEBUG_ENV=1: measured slower than iba_nn_kernel — see the end of this note):
// the search kernel's pass over the anchored neighbour lists as a PERSISTENT grid.
//
// What iba_nn_kernel<WHICH, 0, 1> computes — ComputeAlignmentDist's 1-NN (iba_global.cpp:111-156) and the association path's neighbour
// (iba_local.cpp:238-289) of every list entry, picked from the keypoint's anchored list when the certificate holds, searched in the tree
// otherwise; one record of kNNPartial doubles per (candidate, keyframe, slice), summed in the same fixed order (same bits: tests/test_gpu_nn_list.py) —
// with the launch reshaped:
//
//   * iba_nn_kernel runs one block per (keyframe, group of candidates, slice of 128 list positions): 11 200 blocks at 200 keyframes x 64
//     candidates. Cut short (tools/nn_exp.sh) it shows block start-up 28 us, loads 28 us, picks 48 us, sums 6 us of its 110.
//   * Here a block belongs to one (XCD, group of candidates) for its whole life and walks the (keyframe, slice) items of that pair,
//     every R-th one. A thread owns ONE candidate of the group (cc = tid % CG) and STEPS entries per item. Its loads run ahead of its
//     arithmetic: the entry words two items ahead, the list row + MapPoint of every entry a whole item ahead (one register set per step,
//     re-fetched right behind its pick — or all four behind the item's picks, IBA_NN_LIST_BULK). The candidate constants are staged once per block.
//   * Per item: one LDS-only barrier, then the fixed-order sums of rounds 2-5 over the item's result slots (two slot buffers alternate) and the
//     record. An item whose lists leave entries over ends the inner loop; its tree searches (iba_nn_kernel's code) run between two runs of it.
//
// What it took to make the loads actually run ahead (each of these was an s_waitcnt vmcnt(0) or a spill in the loop; all are in the ISA now as meant):
// buffer-load intrinsics (the compiler regroups plain loads of a row's pieces and parks the moves behind them), out-of-range offsets instead of
// branches for lanes without an entry, the keyframe header through the constant address space and pinned in scalar registers, an LDS-only barrier
// (__syncthreads is a workgroup fence first: vmcnt(0)), the rare other pick finishing inside its own branch, the tree search outside the loop.
//
// MEASURED (same box, 200 keyframes x 64 candidates, tools/nn_list_ab.sh): 0.195 ms against iba_nn_kernel's 0.110 ms, within 0.01 ms of that whatever
// the depth of the prefetch (one pick ahead at 3 waves / SIMD: 0.174; a whole item ahead at 2: 0.195-0.215), with or without any of the fixes above.
// The counters say why (SQ_INSTS_*: 51 k instructions per SIMD, the same work as iba_nn_kernel's 45 k): a wave of this code issues one instruction per
// ~18 cycles however its loads are arranged — the pick is one dependent chain of f64 operations (8 cycles each) through ~100 exec-mask branches —, and
// 232 registers leave two waves per SIMD: one instruction per 9 cycles per SIMD. iba_nn_kernel's waves are no faster (one per 24 cycles) but there are
// four of them: one per 6 cycles. The search is bound by instruction issue over dependent chains, not by the latencies the persistent form hides; what
// would pay is the same walk at >= 4 waves / SIMD (rows through LDS instead of registers) or two entries' chains interleaved in one wave.
//
// grid: hand-eye blocks (as iba_nn_kernel) + 8 XCDs x NG groups x R workers. Launched only when the anchored lists are there, the planes are
// memoised, a group holds at least 4 candidates (STEPS = kSliceW * CG / 256 = 2 or 4) and every buffer stays below 4 GB (32-bit offsets).
#pragma once
#include "iba_split_kernels.hpp"

namespace iba {

// the persistent kernel keeps the block shape it was built and measured with (four waves, up to eight candidates), whatever iba_nn_kernel's is
constexpr int kNLThreads = 256, kNLMaxGroup = 8, kNLCoopLimit = kNLThreads / 2;
#ifndef IBA_NN_LIST_SEARCH_ATTR
#define IBA_NN_LIST_SEARCH_ATTR __forceinline__
#endif
// The tree searches of the entries an item's lists left over (iba_nn_kernel's code: a group of lanes per entry when they are few, persistent lanes when they are
// many), as a function of its own: inlined into the walk's loop it cost that loop 140 spilled registers (its 100 live registers meet the walk's two
// register sets in one allocation); as a call the walk saves what it holds only when a search actually happens.
template <int WHICH>
__device__ IBA_NN_LIST_SEARCH_ATTR void nn_list_search(const int f, const int sl, const int g, const int CG, const int cg_shift, const uint32_t c_end, const int tid_in,
                                                         uint4* __restrict__ flist, const int flist_stride, double* s_res, const bool do_stage, const int dbg) {
    extern __shared__ __align__(16) unsigned char smem[];
    typedef __attribute__((address_space(4))) const NNArgs NNArgsC;
    NNArgsC* ka = (NNArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
#define dp (ka->dp)
#define prm (ka->prm)
#define lay (ka->lay)
    constexpr int T = kNLThreads;
    const int tid = tid_in, lane = tid & 63;
    const int nf = dp.n_frames;
    TreeNode* s_nodes = (TreeNode*)(smem + lay.off_nodes);
    uint32_t* s_ctr = (uint32_t*)(smem + lay.off_misc) + kNLMaxGroup;
    double* s_cd = (double*)(smem + lay.off_cd);
    uint32_t* s_ovf = (uint32_t*)(smem + lay.off_ovf);
    typedef __attribute__((address_space(4))) const FrameHdr FrameHdrC;
    FrameHdrC& h = *(FrameHdrC*)(unsigned long long)(dp.frames + f);
    const uint32_t i_lo = (uint32_t)sl * kSliceW;
    {
    const uint32_t P = h.P, D = h.depth;
    const float4* p4 = dp.pts4 + h.pt_base;
    const float4* kmp = dp.kp_mp + h.kp_base;
    const uint32_t* perm_g = dp.perm + h.pt_base;
    const PlaneRec* planes_cost = dp.plane_cost + h.pt_base;
    const PlaneRec* planes_local = dp.plane_local + h.pt_base;
    if (do_stage) {
        const uint32_t nnodes = (1u << D) - 1u;
        for (uint32_t i = tid; i < nnodes; i += T) s_nodes[i] = dp.nodes[h.node_base + i];
    }
    __syncthreads();   // the tree, and s_ctr[0]
    IBA_LANE_NN_DECL;
    auto entry_at = [&](uint32_t wn) -> size_t { return ((size_t)((uint32_t)(g * CG) + (wn & ((1u << cg_shift) - 1u))) * nf + f) * (size_t)flist_stride + (i_lo + (wn >> cg_shift)); };
    auto make_queries = [&](uint32_t c2, const uint4& e, const float4& mp) {
        actC = (WHICH & 2) && (e.w & kFlagC);
        actA = (WHICH & 1) && (e.w & kFlagA);
        const double* cdl = s_cd + c2 * kCdDoubles;
        const double cs = cdl[0];
        double Ri[9], ti[3];
#pragma unroll
        for (int q = 0; q < 9; ++q) Ri[q] = cdl[1 + q];
#pragma unroll
        for (int q = 0; q < 3; ++q) ti[q] = cdl[10 + q];
        const float s32 = *(const float*)(cdl + 13);
        ax = NAN; ay = NAN; az = NAN; qx = NAN; qy = NAN; qz = NAN;
        if (actA) {
            const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
            const double mx = ((h.Tcw[0] * w0 + h.Tcw[1] * w1) + h.Tcw[2] * w2) + h.Tcw[3];
            const double my = ((h.Tcw[4] * w0 + h.Tcw[5] * w1) + h.Tcw[6] * w2) + h.Tcw[7];
            const double mz = ((h.Tcw[8] * w0 + h.Tcw[9] * w1) + h.Tcw[10] * w2) + h.Tcw[11];
            const double sx = mx * cs, sy = my * cs, sz = mz * cs;
            ax = ((Ri[0] * sx + Ri[1] * sy) + Ri[2] * sz) + ti[0];
            ay = ((Ri[3] * sx + Ri[4] * sy) + Ri[5] * sz) + ti[1];
            az = ((Ri[6] * sx + Ri[7] * sy) + Ri[8] * sz) + ti[2];
        }
        if (actC) {
            const double ts0 = h.Tcw[3] * cs, ts1 = h.Tcw[7] * cs, ts2 = h.Tcw[11] * cs;
            const float m0 = mp.x * s32, m1 = mp.y * s32, m2 = mp.z * s32;
            const double a0 = (double)m0, a1 = (double)m1, a2 = (double)m2;
            const double cx_ = ((h.Tcw[0] * a0 + h.Tcw[1] * a1) + h.Tcw[2] * a2) + ts0;
            const double cy_ = ((h.Tcw[4] * a0 + h.Tcw[5] * a1) + h.Tcw[6] * a2) + ts1;
            const double cz_ = ((h.Tcw[8] * a0 + h.Tcw[9] * a1) + h.Tcw[10] * a2) + ts2;
            qx = ((Ri[0] * cx_ + Ri[1] * cy_) + Ri[2] * cz_) + ti[0];
            qy = ((Ri[3] * cx_ + Ri[4] * cy_) + Ri[5] * cz_) + ti[1];
            qz = ((Ri[6] * cx_ + Ri[7] * cy_) + Ri[8] * cz_) + ti[2];
        }
    };
    auto finish = [&](uint32_t wn) {   // a finished tree search: the plane records are fetched here
        const size_t at = entry_at(wn);
        if ((WHICH & 1) && actA && !(bestA > prm.max_3d_dist2)) {
            const PlaneRec r2 = planes_local[bposA];
            const bool state = local_neigh_ok(prm, r2) && local_plane_ok(prm, r2);   // pointcloud.h:699-717
            flist[at].z = bposA | (state ? 0x80000000u : 0u);
        }
        double res = NAN;
        if ((WHICH & 2) && actC) {
            const float4 pv = p4[bposC];
            const double ex = (double)pv.x - qx, ey = (double)pv.y - qy, ez = (double)pv.z - qz;
            PlaneRec rec; rec.k = 0;
            if (prm.use_plane) rec = planes_cost[bposC];
            res = cost_res(prm, prm.use_plane != 0, rec, ex, ey, ez);
        }
        if (WHICH & 2) s_res[wn] = res;
    };
    auto seed = [&](uint32_t wn) {   // the first bound of a left-over entry's search: its nearest listed points (their positions wait in the entry's result slot)
        const uint2 sd = ((const uint2*)s_res)[wn];
        if ((WHICH & 1) && actA && sd.x != kNone) { const float4 pv = p4[sd.x]; const double dx = ax - (double)pv.x, dy = ay - (double)pv.y, dz = az - (double)pv.z; bestA = (dx * dx + dy * dy) + dz * dz; bposA = sd.x; }
        if ((WHICH & 2) && actC && sd.y != kNone) { const float4 pv = p4[sd.y]; const double dx = qx - (double)pv.x, dy = qy - (double)pv.y, dz = qz - (double)pv.z; bestC = (dx * dx + dy * dy) + dz * dz; bposC = sd.y; }
    };
    const bool coop = c_end <= (uint32_t)kNLCoopLimit;
    if (coop && dbg != 4) {   // a few entries: a group of lanes each
        int G = 64; while ((uint32_t)(T / G) < c_end && G > 2) G >>= 1;
        for (uint32_t q = (uint32_t)tid / (uint32_t)G; q < c_end; q += (uint32_t)(T / G)) {
            const uint32_t wn = s_ovf[q];
            const uint4 e = flist[entry_at(wn)]; const float4 mp = kmp[e.x];
            make_queries(wn & ((1u << cg_shift) - 1u), e, mp);
            lane_nn_begin(IBA_LANE_NN_PASS);
            seed(wn);
            const int round_cap = min(kRoundLeaves, (int)((kSliceW * (uint32_t)kNLMaxGroup - (uint32_t)kNLCoopLimit) / (uint32_t)(T / G)));
            uint32_t* s_leaf = s_ovf + kNLCoopLimit + ((uint32_t)tid / (uint32_t)G) * (uint32_t)round_cap;
            if (dbg == 6) {}
            else if (ka->nn_rounds) do wave_nn_round<WHICH>(IBA_LANE_NN_PASS, tid & (G - 1), G, s_nodes, p4, perm_g, P, D, s_leaf, round_cap); while (go >= 0 && dbg != 7);
            else do wave_nn_visit<WHICH>(IBA_LANE_NN_PASS, tid & (G - 1), G, s_nodes, p4, perm_g, P, D); while (go >= 0 && dbg != 7);
            if (dbg != 5) finish(wn);   // every lane of the group writes the same values
        }
    } else if (!coop) {   // many: persistent lanes, refilled from the queue
        bool have = false, exhausted = false;
        uint32_t w = 0u;
        for (;;) {
            const unsigned long long idle = __ballot(!have);
            if (idle != 0ull && !exhausted) {
                const uint32_t nidle = (uint32_t)__popcll(idle);
                uint32_t base = 0u;
                if (lane == 0) base = atomicAdd(s_ctr, nidle);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                exhausted = base + nidle >= c_end;
                if (!have) {
                    const uint32_t wq = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                    if (wq < c_end) {
                        const uint32_t wn = s_ovf[wq];
                        const uint4 e = flist[entry_at(wn)]; const float4 mp = kmp[e.x];
                        w = wn;
                        make_queries(wn & ((1u << cg_shift) - 1u), e, mp);
                        lane_nn_begin(IBA_LANE_NN_PASS);
                        seed(wn);
                        have = true;
                    }
                }
            }
            if (__ballot(have) == 0ull) { if (exhausted) break; continue; }
            if (have && dbg == 4) { have = false; continue; }
            if (have) lane_nn_visit<WHICH>(IBA_LANE_NN_PASS, s_nodes, p4, perm_g, P, D);
            if (have && go < 0) {
                if (dbg != 5) finish(w);
                have = false;
            }
        }
    }
        __syncthreads();
    }
#undef dp
#undef prm
#undef lay
}

#ifndef IBA_NN_LIST_BULK
#define IBA_NN_LIST_BULK 1
#endif
#ifndef IBA_NN_LIST_WAVES
#define IBA_NN_LIST_WAVES 2   /* waves per SIMD iba_nn_list_kernel is compiled for (256 VGPRs: the rows of a whole item in flight beside a pick; its own loads run ahead of its arithmetic) */
#endif
template <int WHICH, int STEPS>
__global__ __launch_bounds__(kNLThreads) __attribute__((amdgpu_waves_per_eu(IBA_NN_LIST_WAVES, IBA_NN_LIST_WAVES))) void iba_nn_list_kernel(
    NNArgs ka_by_value, const Cand* __restrict__ cands, int B, int CG, int NS, int R, double* __restrict__ nn_partials, int nn_nrec, uint4* __restrict__ flist,
    const uint32_t* __restrict__ lcount, int flist_stride, int dbg, const SetPt* __restrict__ anchor, double* __restrict__ he_out, int he_blocks) {
    extern __shared__ __align__(16) unsigned char smem[];
    if ((int)blockIdx.x < he_blocks) {   // K7 rides in front, as in iba_nn_kernel
        const int nfh = ka_by_value.dp.n_frames;
        const int i = (int)blockIdx.x * (kNLThreads / 2) + (int)(threadIdx.x >> 1);
        const bool live = i < B * nfh;
        const int ii = live ? i : 0;
        const double v = he_term(ka_by_value.dp.frames[ii % nfh], cands[ii / nfh], threadIdx.x & 1, live);
        if (live && !(threadIdx.x & 1)) he_out[i] = v;
        return;
    }
    if (blockIdx.x < (uint32_t)((he_blocks + 7) & ~7)) return;   // (padding: worker i still runs on XCD i % 8)
    const uint32_t bid = blockIdx.x - (uint32_t)((he_blocks + 7) & ~7);
    typedef __attribute__((address_space(4))) const NNArgs NNArgsC;
    NNArgsC* ka = (NNArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)ka_by_value;
#define dp (ka->dp)
#define prm (ka->prm)
#define lay (ka->lay)
    constexpr int T = kNLThreads;
    constexpr uint32_t kWantMask = ((WHICH & 2) ? kFlagC : 0u) | ((WHICH & 1) ? kFlagA : 0u);
    const int tid = threadIdx.x;
#ifdef IBA_DIAG_COUNTERS
    unsigned long long nl_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, nl_last = __builtin_readcyclecounter();   // cycles of thread 0 per phase -> dp.diag[8 + 2 i] (iba_debug_phase_cycles; tools/nn_phase_probe.py)
#define NL_TICK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); nl_t[i] += _t - nl_last; nl_last = _t; } while (0)
#else
#define NL_TICK(i) do { } while (0)
#endif
    // Block-uniform values the walk reads in every pick are pinned in scalar registers (`keep`): left to itself the compiler re-loads them from the kernel arguments
    // and the keyframe header wherever it runs short of scalar registers — 55 scalar loads per item, each followed by its own wait, in a kernel whose two waves
    // per SIMD cannot hide them.
    auto keep = [](int v) -> int { asm volatile("" : "+s"(v)); return v; };
    auto keep_u = [](uint32_t v) -> uint32_t { asm volatile("" : "+s"(v)); return v; };
    auto keep_d = [](double v) -> double { unsigned long long u = (unsigned long long)__double_as_longlong(v); asm volatile("" : "+s"(u)); return __longlong_as_double((long long)u); };
    const int nf = keep(dp.n_frames);
    flist_stride = keep(flist_stride); NS = keep(NS); R = keep(R);
    const uint32_t max_k = keep_u(dp.max_k);
    const double max_3d_dist2 = keep_d(prm.max_3d_dist2), corr_threshold = keep_d(prm.corr_3d_3d_threshold);
    const int per_xcd = (nf + 7) / 8;
    const int NG = (B + CG - 1) / CG;
    const int xcd = bid & 7, g = (int)(bid >> 3) % NG, r = (int)(bid >> 3) / NG;
    const int M = per_xcd * NS;                      // items of this (XCD, group): (keyframe xcd + 8 (m / NS), slice m % NS)
    const int K = r < M ? (M - r + R - 1) / R : 0;   // ... of which this block takes r, r + R, ...
    if (K == 0) return;
    int cg_shift = 0; while ((1 << cg_shift) < CG) ++cg_shift;   // CG is a power of two (host); STEPS * T == kSliceW << cg_shift
    const uint32_t W = kSliceW << cg_shift;          // result slots of an item
    const int cands_here = min(CG, B - g * CG);
    const uint32_t cc = (uint32_t)tid & ((1u << cg_shift) - 1u);   // this thread's candidate of the group
    const bool cand_ok = (int)cc < cands_here;
    const size_t b_safe = (size_t)(g * CG + (cand_ok ? (int)cc : 0));

    double* s_res0 = (double*)(smem + lay.off_res);
    double* s_res1 = (double*)(smem + lay.off_res2);
    uint32_t* s_ctr = (uint32_t*)(smem + lay.off_misc) + kNLMaxGroup;   // [0]: next unclaimed left-over entry; [1..3]: left-over counts of the items k % 3
    double* s_cd = (double*)(smem + lay.off_cd);
    uint32_t* s_ovf = (uint32_t*)(smem + lay.off_ovf);

    if (tid < cands_here * kCdDoubles) s_cd[tid] = ((const double*)&cands[g * CG + tid / kCdDoubles])[12 + tid % kCdDoubles];
    if (tid < 4) s_ctr[tid] = 0u;
    const uint32_t sel = cand_ok ? (uint32_t)ka->anchor_sel[g * CG + (int)cc] : 255u;   // the lists this thread's candidate reads (255: none is near — its entries go to the tree)
    const unsigned char* set_base = (const unsigned char*)anchor + (size_t)(sel != 255u ? sel : 0u) * ka->anchor_set_bytes;
    __syncthreads();
    if (dbg == 1) return;

    // (block-uniform values, but the division runs on the vector ALU: without the readfirstlane the keyframe's header — Tcw: 24 registers — is fetched per lane)
    // the keyframe headers through the constant address space: nothing writes them while a kernel runs, but behind the stores of the walk's loop the compiler
    // would fetch a header per LANE (Tcw: 24 vector registers and 6 vector loads per pick) instead of once per wave into scalar registers
    typedef __attribute__((address_space(4))) const FrameHdr FrameHdrC;
    auto frame_hdr = [&](int f) -> FrameHdrC& { return *(FrameHdrC*)(unsigned long long)(dp.frames + f); };
    struct Item { int f, sl; uint32_t kp_base; bool live; };   // an item of the walk: keyframe, slice, where the keyframe's MapPoints start; live: it exists (block-uniform, all of it)
    auto make_item = [&](int k) -> Item {
        Item it;
        const int m = r + k * R, fr = __builtin_amdgcn_readfirstlane(xcd + 8 * (m / NS));   // (the division runs on the vector ALU)
        it.live = k < K && fr < nf;
        it.f = keep(it.live ? fr : 0);
        it.sl = keep(__builtin_amdgcn_readfirstlane(m % NS));
        it.kp_base = keep_u(frame_hdr(it.f).kp_base);
        return it;
    };
    // the entry of step s of item k: list position, address of its words
    auto list_pos = [&](const Item& it, int s) -> uint32_t { return (uint32_t)it.sl * kSliceW + (((uint32_t)s * (uint32_t)T + (uint32_t)tid) >> cg_shift); };

    // The loads that run ahead are BUFFER loads (raw_buffer_load intrinsics): the compiler neither splits nor regroups them (plain loads of a row's six 16-byte
    // pieces came back as seven overlapping loads plus register moves placed right behind them — a wait for the load where the next pick should have covered it),
    // and a lane without an entry asks for an offset beyond the buffer: zeros, no memory request, no branch around the load (a load under a condition is waited for
    // where its value merges). Offsets are 32 bits: the host launches this kernel only while the lists, the MapPoints and the anchored sets stay below 4 GB each.
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    constexpr uint32_t kNoOffset = 0xFFFFFF00u;   // (+ the 96 bytes of a row's pieces: no wrap; every buffer below is shorter)
    const __amdgpu_buffer_rsrc_t rs_list = __builtin_amdgcn_make_buffer_rsrc((void*)flist, 0, (int)min((size_t)B * nf * (size_t)flist_stride * 16u, (size_t)0xFFFFFF00u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cnt = __builtin_amdgcn_make_buffer_rsrc((void*)lcount, 0, (int)((size_t)B * nf * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_mp = __builtin_amdgcn_make_buffer_rsrc((void*)dp.kp_mp, 0, (int)0x80000000u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_row = __builtin_amdgcn_make_buffer_rsrc((void*)anchor, 0, (int)min((size_t)kAnchorSets * (size_t)ka->anchor_set_bytes, (size_t)0xFFFFFF00u), 0x00020000);
    const uint32_t set_off = sel != 255u ? sel * (uint32_t)ka->anchor_set_bytes : kNoOffset;   // this thread's set within the anchored sets (none: every row reads as empty — count 0 — and its entries go to the tree)

    // ---- stage E: the entry words (keypoint, flags) of the STEPS entries of an item and the list length of this thread's candidate ----
    struct EWords { uint32_t x[STEPS], w[STEPS]; uint32_t n; };
    auto fetch_entries = [&](const Item& it, EWords& ew) {
        const bool live = it.live && cand_ok;
        const uint32_t row0 = (uint32_t)b_safe * (uint32_t)nf + (uint32_t)it.f;
        ew.n = __builtin_amdgcn_raw_buffer_load_b32(rs_cnt, live ? row0 * 4u : kNoOffset, 0, 0);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const uint32_t off = live ? (row0 * (uint32_t)flist_stride + list_pos(it, s)) * 16u : kNoOffset;   // (a position beyond the list's stride lands in the next list: never used — entry_valid — and inside the buffer or zero)
            ew.x[s] = __builtin_amdgcn_raw_buffer_load_b32(rs_list, off, 0, 0);
            ew.w[s] = __builtin_amdgcn_raw_buffer_load_b32(rs_list, off + 12u, 0, 0);
        }
    };
    auto entry_valid = [&](const Item& it, int s, const EWords& ew) -> bool { return it.live && cand_ok && list_pos(it, s) < ew.n && (ew.w[s] & kWantMask) != 0u; };

    // ---- stage S: the MapPoint of the entry's keypoint and the first line of its list row (header + nearest listed point) ----
    // (kept as the 16-byte pieces they are loaded as: fields picked out of a piece behind the load are register moves the compiler places right there — a wait for
    //  the load where the next pick should have covered it)
    struct Row { v4u q[6]; v4u m; };   // q[0..2]: AnchorHdr, q[3..5]: SetPt of the nearest listed point; m: the MapPoint (float bits)
    static_assert(sizeof(AnchorHdr) == 48 && sizeof(SetPt) == 48 && offsetof(AnchorHdr, dM) == 32 && offsetof(AnchorHdr, count) == 40 && offsetof(AnchorHdr, da1_lo) == 44 &&
                  offsetof(SetPt, pos) == 12 && offsetof(SetPt, nx) == 16 && offsetof(SetPt, nz) == 32 && offsetof(SetPt, flags) == 40 && offsetof(SetPt, da_lo) == 44, "Row's pieces");
    auto row_of = [&](int f, uint32_t kp) -> const unsigned char* { return anchor_row(set_base, (size_t)f * max_k + kp); };
    auto fetch_row = [&](const Item& it, uint32_t kp, bool valid, Row& rw) {
        rw.m = __builtin_amdgcn_raw_buffer_load_b128(rs_mp, valid ? (it.kp_base + kp) * 16u : kNoOffset, 0, 0);
        const uint32_t ro = (valid && sel != 255u) ? set_off + ((uint32_t)it.f * max_k + kp) * (uint32_t)kAnchorRowBytes : kNoOffset;   // header + nearest neighbour: the first 96 bytes of the row's first 128-byte line
#pragma unroll
        for (int i = 0; i < 6; ++i) rw.q[i] = __builtin_amdgcn_raw_buffer_load_b128(rs_row, ro + 16u * (uint32_t)i, 0, 0);
    };
    auto dbl = [](uint32_t lo, uint32_t hi) -> double { return __hiloint2double((int)hi, (int)lo); };

    // ---- stage P: the pick (see iba_nn_kernel) ----
    auto pick = [&](const Item& it, const double (&Tcw)[12], int s, uint32_t kp, uint32_t ew_w, bool valid, const Row& rw, double* s_res, uint32_t* ovf_ctr) {
        const uint32_t wn = (uint32_t)s * (uint32_t)T + (uint32_t)tid;
        if (!valid) { if ((WHICH & 2) && wn < W) s_res[wn] = NAN; return; }
        const int f = it.f;
        const bool actC = (WHICH & 2) && (ew_w & kFlagC), actA = (WHICH & 1) && (ew_w & kFlagA);
        double ax = NAN, ay = NAN, az = NAN, qx = NAN, qy = NAN, qz = NAN;
        {   // the two MapPoint -> LiDAR-frame queries of the entry under this thread's candidate (iba_local.cpp:238-239,282 and iba_global.cpp:231-234):
            // the keyframe's pose first, then the candidate's rotation ROW BY ROW for both queries (4 of its 13 constants live at a time, not 26 registers of them)
            const double* cdl = s_cd + cc * kCdDoubles;
            const double cs = cdl[0];
            const float s32 = *(const float*)(cdl + 13);
            double sx = 0, sy = 0, sz = 0, cx_ = 0, cy_ = 0, cz_ = 0;
            if (actA) {
                const double w0 = (double)__uint_as_float(rw.m.x), w1 = (double)__uint_as_float(rw.m.y), w2 = (double)__uint_as_float(rw.m.z);
                const double mx = ((Tcw[0] * w0 + Tcw[1] * w1) + Tcw[2] * w2) + Tcw[3];
                const double my = ((Tcw[4] * w0 + Tcw[5] * w1) + Tcw[6] * w2) + Tcw[7];
                const double mz = ((Tcw[8] * w0 + Tcw[9] * w1) + Tcw[10] * w2) + Tcw[11];
                sx = mx * cs; sy = my * cs; sz = mz * cs;
            }
            if (actC) {
                const double ts0 = Tcw[3] * cs, ts1 = Tcw[7] * cs, ts2 = Tcw[11] * cs;   // TcwRS translation *= scale (:208)
                const float m0 = __uint_as_float(rw.m.x) * s32, m1 = __uint_as_float(rw.m.y) * s32, m2 = __uint_as_float(rw.m.z) * s32;             // CV_32F product (:232)
                const double a0 = (double)m0, a1 = (double)m1, a2 = (double)m2;
                cx_ = ((Tcw[0] * a0 + Tcw[1] * a1) + Tcw[2] * a2) + ts0;
                cy_ = ((Tcw[4] * a0 + Tcw[5] * a1) + Tcw[6] * a2) + ts1;
                cz_ = ((Tcw[8] * a0 + Tcw[9] * a1) + Tcw[10] * a2) + ts2;
            }
            double oa[3], oc[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double r0 = cdl[1 + 3 * i], r1 = cdl[2 + 3 * i], r2 = cdl[3 + 3 * i], t0 = cdl[10 + i];
                oa[i] = ((r0 * sx + r1 * sy) + r2 * sz) + t0;
                oc[i] = ((r0 * cx_ + r1 * cy_) + r2 * cz_) + t0;
            }
            if (actA) { ax = oa[0]; ay = oa[1]; az = oa[2]; }
            if (actC) { qx = oc[0]; qy = oc[1]; qz = oc[2]; }
        }
        NL_TICK(8);   // (inside a pick) the queries: MapPoint, candidate constants, keyframe pose
        auto leave = [&](uint32_t pa, uint32_t pc) { ((uint2*)s_res)[wn] = make_uint2(pa, pc); s_ovf[atomicAdd(ovf_ctr, 1u)] = wn; };
        if (sel == 255u || rw.q[2].z == 0u) { leave(kNone, kNone); return; }
        struct { double qa[3], d1, dM; uint32_t count; float da1_lo; } hd;
        hd.qa[0] = dbl(rw.q[0].x, rw.q[0].y); hd.qa[1] = dbl(rw.q[0].z, rw.q[0].w); hd.qa[2] = dbl(rw.q[1].x, rw.q[1].y); hd.d1 = dbl(rw.q[1].z, rw.q[1].w);
        hd.dM = dbl(rw.q[2].x, rw.q[2].y); hd.count = rw.q[2].z; hd.da1_lo = __uint_as_float(rw.q[2].w);
        SetPt p0;   // the nearest listed point (overwritten by the rare other pick of the cost path)
        p0.x = __uint_as_float(rw.q[3].x); p0.y = __uint_as_float(rw.q[3].y); p0.z = __uint_as_float(rw.q[3].z); p0.pos = rw.q[3].w;
        p0.nx = dbl(rw.q[4].x, rw.q[4].y); p0.ny = dbl(rw.q[4].z, rw.q[4].w); p0.nz = dbl(rw.q[5].x, rw.q[5].y); p0.flags = rw.q[5].z; p0.da_lo = __uint_as_float(rw.q[5].w);
        double S2 = 0.0;
        if ((WHICH & 1) && actA) { const double dx = ax - hd.qa[0], dy = ay - hd.qa[1], dz = az - hd.qa[2]; S2 = (dx * dx + dy * dy) + dz * dz; }
        if ((WHICH & 2) && actC) { const double dx = qx - hd.qa[0], dy = qy - hd.qa[1], dz = qz - hd.qa[2]; S2 = fmax(S2, (dx * dx + dy * dy) + dz * dz); }
        auto sqrt_up = [](double v2) -> double { return (double)(__builtin_sqrtf((float)v2) * 1.000001f + 1e-18f); };   // >= sqrt(v2); NaN stays NaN
        const double S = sqrt_up(S2);
        const double radius = (hd.d1 + 2.0 * S) * (1.0 + 1e-12) + 1e-12;
        const bool quick = radius < hd.dM;   // (NaN queries fail both tests)
        NL_TICK(9);   // the certificate (the row's header)
        double bestA = INFINITY, bestC = INFINITY; uint32_t bposA = kNone, bposC = kNone, ia = 0u, ic = 0u;
        // nn_merge (least d^2, ties to the lowest original index) with the index table fetched only on a tie: two dependent scalar loads per pick otherwise
        auto merge = [&](double& best, uint32_t& bpos, double od, uint32_t op) {
            if (od < best) { best = od; bpos = op; }
            else if (od == best && op != kNone && op != bpos) { const uint32_t* perm_g = dp.perm + frame_hdr(f).pt_base; if (bpos == kNone || perm_g[op] < perm_g[bpos]) bpos = op; }
        };
        const float rf = quick ? (float)radius * 1.000001f + 1e-30f : INFINITY;   // >= radius
        if (p0.da_lo <= rf) {   // the nearest listed point (its own lower bound is <= d_1 <= radius)
            const double x = (double)p0.x, y = (double)p0.y, z = (double)p0.z;
            if ((WHICH & 1) && actA) { const double dx = ax - x, dy = ay - y, dz = az - z; merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, p0.pos); }
            if ((WHICH & 2) && actC) { const double dx = qx - x, dy = qy - y, dz = qz - z; merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, p0.pos); }
            if (hd.count > 1u && hd.da1_lo <= rf) {   // (the usual case: the second neighbour cannot qualify and its line is not even fetched)
                for (uint32_t si = 1u; si < hd.count; ++si) {
                    const SetPt pv = *anchor_pt(row_of(f, kp), si);
                    if (!(pv.da_lo <= rf)) break;            // the list is sorted by distance to the anchor query: nothing further qualifies
                    const double x2 = (double)pv.x, y2 = (double)pv.y, z2 = (double)pv.z;
                    if ((WHICH & 1) && actA) { const double dx = ax - x2, dy = ay - y2, dz = az - z2; const uint32_t was = bposA; merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, pv.pos); if (bposA != was) ia = si; }
                    if ((WHICH & 2) && actC) { const double dx = qx - x2, dy = qy - y2, dz = qz - z2; const uint32_t was = bposC; merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, pv.pos); if (bposC != was) ic = si; }
                }
            }
        }
        NL_TICK(10);   // the nearest listed point(s)
        if (!quick) {   // second chance, from the whole row (see iba_nn_kernel)
            double r2 = 0.0;
            if ((WHICH & 1) && actA) r2 = bestA;
            if ((WHICH & 2) && actC) r2 = fmax(r2, bestC);
            if (!((sqrt_up(r2) + S) * (1.0 + 1e-12) + 1e-12 < hd.dM)) { leave(bposA, bposC); return; }   // (NaN queries end here, unseeded)
        }
        if (dbg == 5) return;
        // the results. The nearest listed point is the answer of nearly every lane and is in registers; the rare other pick fetches its entry INSIDE its own branch and
        // finishes there: a value loaded under a condition and merged behind it is waited for behind it, by every lane — s_waitcnt vmcnt(0), with it every row in flight.
        uint4* at = flist + (((size_t)(g * CG + (int)cc) * nf + f) * (size_t)flist_stride + list_pos(it, s));
        auto finish = [&](const uint32_t fa, const SetPt& pc) {
            if ((WHICH & 1) && actA && !(bestA > max_3d_dist2)) at->z = bposA | ((fa & 1u) ? 0x80000000u : 0u);   // the association keeps its neighbour only within max_3d_dist (iba_local.cpp:289)
            double res = NAN;
            if ((WHICH & 2) && actC) {
                const double ex = (double)pc.x - qx, ey = (double)pc.y - qy, ez = (double)pc.z - qz;
                if (pc.flags & 2u) res = fabs(ex * pc.nx + ey * pc.ny + ez * pc.nz);   // cost_res with the plane's verdict already taken (iba_anchor_kernel): the same expressions
                else res = -sqrt((ex * ex + ey * ey) + ez * ez);
            }
            if (WHICH & 2) s_res[wn] = res;
        };
        if ((ia | ic) == 0u) finish(p0.flags, p0);
        else {
            const uint32_t fa = ia != 0u ? anchor_pt(row_of(f, kp), ia)->flags : p0.flags;
            const SetPt pc = ic != 0u ? *anchor_pt(row_of(f, kp), ic) : p0;
            finish(fa, pc);
        }
    };

    // ---- the end of an item: left-over searches (iba_nn_kernel's), fixed-order sums, record ----
    int staged_f = -1;   // the keyframe whose tree is in LDS
    auto item_sync = [&](int k, uint32_t* ovf_ctr) -> uint32_t {   // the item's barrier; returns the number of entries its lists left over
        // (an LDS-only barrier: __syncthreads() is a workgroup-scope fence first — s_waitcnt vmcnt(0) —, which drained every row in flight for the next item
        //  at every item. Nothing this kernel stores to global memory is read by another wave of the block.)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const uint32_t c_end = *ovf_ctr;
        if (tid == 0) { s_ctr[0] = 0u; s_ctr[1 + (k + 2) % 3] = 0u; }   // (the count of item k - 1: read by every wave before this item's barrier, used again behind the next one's)
        return c_end;
    };
    auto item_search = [&](const Item& it, double* s_res, const uint32_t c_end) {
        const int f = it.f;
        nn_list_search<WHICH>(f, it.sl, g, CG, cg_shift, c_end, tid, flist, flist_stride, s_res, staged_f != f, dbg);
        staged_f = f;
    };
    auto item_sums = [&](const Item& it, const double* s_res, const uint32_t c_end) {
        const int f = it.f, sl = it.sl;
        // fixed-order sums: 32 threads per candidate, each over a strided subset of its entries, then a butterfly
        double fin_sum = 0.0; uint32_t fin_c = 0, fin_v = 0, fin_pl = 0, fin_pt = 0;
        const uint32_t c2 = (uint32_t)tid >> 5, q = (uint32_t)tid & 31u;
        if ((WHICH & 2) && c2 < (uint32_t)cands_here) {
            for (uint32_t wi = c2 + (q << cg_shift); wi < W; wi += 32u << cg_shift) {
                const double rr = s_res[wi];
                if (rr == rr) {
                    const double dist = fabs(rr);
                    const bool is_pt = __double_as_longlong(rr) < 0;
                    ++fin_c;
                    if (dist < corr_threshold) { fin_sum += dist; ++fin_v; fin_pl += is_pt ? 0u : 1u; fin_pt += is_pt ? 1u : 0u; }
                }
            }
        }
        unsigned long long cnt = (unsigned long long)fin_c | ((unsigned long long)fin_v << 32);
        unsigned long long cnt2 = (unsigned long long)fin_pl | ((unsigned long long)fin_pt << 32);
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { fin_sum += __shfl_xor(fin_sum, off); cnt += __shfl_xor(cnt, off); cnt2 += __shfl_xor(cnt2, off); }
        if (c2 < (uint32_t)cands_here && q < (uint32_t)kNNPartial) {
            double out = 0.0;
            if (q == 0) out = fin_sum;
            else if (q == 1) out = (double)(cnt & 0xffffffffull);
            else if (q == 2) out = (double)(cnt >> 32);
            else if (q == 3) out = (double)(cnt2 & 0xffffffffull);
            else if (q == 4) out = (double)(cnt2 >> 32);
            else if (q == 5 && c2 == 0) out = (double)c_end;
            nn_partials[((size_t)(g * CG + (int)c2) * nn_nrec + (size_t)f * NS + sl) * kNNPartial + q] = out;
        }
    };

    // ---- the walk ----
    // An item whose lists leave entries over ends the inner loop: its tree searches run between two runs of the loop, where nothing of the walk is in
    // flight (inside the loop their ~100 live registers met the walk's two register sets in one allocation: 140 spilled registers), and the walk
    // starts again at the next item (its first fetches exposed once more: the price of a search, not of an item).
    int k = 0;
    for (;;) {
        Item it0 = make_item(k), it1 = make_item(k + 1), it2;
        if (!it0.live) break;
        EWords e0, e1, e2;   // the entries of the current item, of the next one (its rows are being fetched) and of the one after (in flight)
        Row rs[STEPS];       // the row of each step's entry: picked for item k, then re-fetched at once for item k + 1 — a row has a whole item's picks to arrive
        fetch_entries(it0, e0);
        fetch_entries(it1, e1);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) fetch_row(it0, e0.x[s], entry_valid(it0, s, e0), rs[s]);
        uint32_t c_end = 0u;
        double* s_res = s_res0;
        for (; it0.live; ++k) {   // (the keyframes of this XCD beyond the last end the walk: items are in keyframe order)
            s_res = (k & 1) ? s_res1 : s_res0;
            uint32_t* ovf_ctr = s_ctr + 1 + k % 3;
            NL_TICK(0);   // (start-up, and whatever lies between two items)
            it2 = make_item(k + 2);
            fetch_entries(it2, e2);
            double Tcw[12];
            {
                FrameHdrC& h = frame_hdr(it0.f);
#pragma unroll
                for (int q = 0; q < 12; ++q) Tcw[q] = keep_d(h.Tcw[q]);
            }
            NL_TICK(1);   // the item's scalars, the entries two items ahead
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                pick(it0, Tcw, s, e0.x[s], e0.w[s], entry_valid(it0, s, e0), rs[s], s_res, ovf_ctr);
                NL_TICK(11);   // the results of a pick (2: what is left of a pick outside 8-11: nothing)
#if IBA_NN_LIST_BULK == 0
                fetch_row(it1, e1.x[s], entry_valid(it1, s, e1), rs[s]);
                NL_TICK(3);   // issuing the next item's row
#endif
            }
#if IBA_NN_LIST_BULK
            // the next item's rows, all at once BEHIND the picks: a pick that looks at further listed points waits for its own loads with s_waitcnt vmcnt(0) — loads
            // return in order — and would wait for every row issued since; behind the picks the rows have the barrier, the sums and the next item's scalars to arrive
#pragma unroll
            for (int s = 0; s < STEPS; ++s) fetch_row(it1, e1.x[s], entry_valid(it1, s, e1), rs[s]);
            NL_TICK(3);
#endif
            c_end = item_sync(k, ovf_ctr);
            NL_TICK(4);   // the item's barrier
            if (c_end != 0u) break;
            item_sums(it0, s_res, 0u);
            NL_TICK(5);   // sums + record
            e0 = e1; e1 = e2; it0 = it1; it1 = it2;
        }
        if (c_end == 0u) break;
#ifndef NL_NO_SEARCH
        item_search(it0, s_res, c_end);
#endif
        NL_TICK(6);   // tree searches
        item_sums(it0, s_res, c_end);
        NL_TICK(5);
        ++k;
    }
#ifdef IBA_DIAG_COUNTERS
    if (tid == 0 && dp.diag) {
        unsigned long long* d64 = (unsigned long long*)(dp.diag + 8);
        for (int i = 0; i < 7; ++i) atomicAdd(d64 + i, nl_t[i]);
        atomicAdd(d64 + 7, 1ull);   // blocks that got here
        for (int i = 8; i < 12; ++i) atomicAdd(d64 + i, nl_t[i]);   // inside the picks: queries, certificate, nearest listed point, results
    }
#endif
#undef NL_TICK
#undef dp
#undef prm
#undef lay
}

}  // namespace iba
