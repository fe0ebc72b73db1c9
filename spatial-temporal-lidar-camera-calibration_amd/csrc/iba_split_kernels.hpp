// CDNA4 (gfx950) kernels of the IBA cross-modality evaluation path, part 2: the evaluation chain of one batch of candidates.
//
//   iba_assoc_kernel   one workgroup per (keyframe, candidate): frustum cull of 64-point chunks, float32 pre-cull against the
//                      keypoint bitmap, float32 walk of the keypoint grid -> (point, keypoint) pairs, exact f64 projection and
//                      d^2 per pair, ties, corrset size, ONE work list per (candidate, keyframe) in keypoint order, local
//                      plane at the matched point, 3d-2d covisible residuals -> partial record + list handed to
//                      iba_nn_kernel (and, through it, to iba_factor_kernel).
//                      = TransformPointCloud + FindProjectCorrespondences (pointcloud.h:82-86, iba_global.cpp:55-96 =
//                      iba_local.cpp:17-58), the 3d-2d loop (iba_global.cpp:291-328), ComputeLocalNeighbor validity
//                      (iba_local.cpp:207-231).
//   iba_nn_kernel      one workgroup per (keyframe, group of up to 8 candidates, slice of 128 list positions): the MapPoint ->
//                      scan 1-NN of iba_global.cpp:231-234,116-122 and iba_local.cpp:282-290, ONE LANE per (MapPoint, candidate)
//                      in a persistent loop without barriers (LaneNN / lane_nn_visit), then per entry the point-to-plane /
//                      point-to-point cost distance (ComputeAlignmentDist, iba_global.cpp:111-156, 241-249), the kind of the
//                      3d-3d residual block (pointcloud.h:699-717), and fixed-order sums.
//   iba_fit_kernel     plane_cache = 0 only: the planes one evaluation needs, fitted for that evaluation between the kernels
//                      above (stage 1: at the matched points; stage 2: at the searched neighbours).
//   iba_reduce2_kernel fixed-order sums of the records of both kernels (and of iba_factor_kernel) per candidate.
//
// Why two kernels: in round 1's one-kernel form 43 % of the time went into the 1-NN phase, each (keyframe, candidate) block
// descending the tree for ~264 MapPoints in barrier-separated rounds with half of its lanes idle, and 13 % into a finalize pass
// that recomputed the queries. Splitting also frees the association kernel from the search's registers and LDS.
#pragma once
#include "iba_kernels.hpp"
#include "iba_pair_plan.hpp"

namespace iba {

#ifndef IBA_PAIRS_CUT
#define IBA_PAIRS_CUT 0   /* timing experiment (tools/pairs_cuts.sh): iba_pairs_kernel ends behind its k-th phase; results invalid */
#endif
#ifndef IBA_NN_EXP
#define IBA_NN_EXP 0
#endif
#ifndef IBA_NN_THREADS
#define IBA_NN_THREADS 256
#endif
// (round 6) iba_nn_kernel is also built with ONE wave per block and two candidates per block (kNNThreadsSmall), which the host picks while a keyframe's kd tree is small (its
// LDS copy is reserved per block: 4 KB at 10 k points per scan, 64 KB at 120 k): 0.1058 -> 0.0909 ms at the bench shape (128 threads / 4 candidates: 0.0955; 256 / 8, rounds
// 2-5: 0.1058; 512 / 16: 0.128 — tools/experiments/README.md). The kernel is bound by instruction issue over dependent chains with the blocks of a CU in step; single-wave
// blocks come and go one by one, their barriers cost nothing, and 44 800 of them fill the tail of the launch evenly. At 120 k points per scan the same shape is 18 % slower
// (1.52 -> 1.80 ms per call: two blocks of one wave per CU beside their trees), hence the choice per handle. (The sums use 32 threads per candidate: a wave holds two.)
constexpr int kNNThreadsSmall = 64;
constexpr int kNNThreads = IBA_NN_THREADS;
constexpr int kNNWaves = kNNThreads / 64;
#ifndef IBA_NN_SLICE
#define IBA_NN_SLICE 128
#endif
constexpr uint32_t kSliceW = IBA_NN_SLICE;   // list positions per search block: a constant, so that the order of every sum is the same whatever the batch
constexpr int kNNPartial = 8;   // doubles per (candidate, nn record): sum3d, cnt, valid, valid_pl, valid_pt, 3 spare
constexpr int kMaxGroup = kNNThreads / 32 < 16 ? kNNThreads / 32 : 16;   // candidates per block (32 threads each in the final sums)
static_assert(4 * (2 * kMaxGroup + 4) <= 128, "iba_nn_kernel's misc slab (layout_nn: 128 bytes) holds s_n[kMaxGroup], s_ctr[4] and s_sel[kMaxGroup]");

struct NNLayout {   // byte offsets into the dynamic LDS of iba_nn_kernel
    uint32_t off_nodes, off_res, off_misc, off_cd, off_ovf, total;
    uint32_t off_res2;   // the second result-slot buffer of iba_nn_list_kernel (items alternate)
};
constexpr int kCdDoubles = 14;   // per candidate in LDS: s, Ri[9], ti[3], (s32, pad) = doubles 12..25 of Cand
#ifndef IBA_NN_WAVES
#define IBA_NN_WAVES 5   /* waves per SIMD the search kernel is compiled for (<= 96 VGPRs) */
#endif
#ifndef IBA_NN_ROUND_LEAVES
#define IBA_NN_ROUND_LEAVES 8   /* leaves a group of lanes walks to before it scans them together (wave_nn_round) */
#endif
#ifndef IBA_NN_LEAF_BATCH
#define IBA_NN_LEAF_BATCH 4   /* points of a leaf scan whose loads are in flight together */
#endif
#ifndef IBA_NN_SETS_WAVES
#define IBA_NN_SETS_WAVES 4   /* waves per SIMD the search kernel with the anchored lists is compiled for (4: 128 VGPRs, what its LDS allows) */
#endif
constexpr int kAnchorSets = 2;      // anchored neighbour lists are kept around up to this many anchor extrinsics (an optimiser polls around two incumbents)
struct NNArgs {
    DevProblem dp; DevParams prm; NNLayout lay;
    // which set of anchored neighbour lists each candidate of the batch reads (255: none — its lanes search the tree), and the
    // distance in bytes between two sets
    unsigned long long anchor_set_bytes;
    uint32_t nn_rounds;   // the entries the lists leave over are searched in rounds of leaves (wave_nn_round; 0: leaf by leaf, wave_nn_visit)
    uint8_t anchor_sel[kMaxChain];
};

// list entry flags (uint4.w) written by iba_assoc_kernel
constexpr uint32_t kFlagC = 1u;   // cost-path 1-NN wanted (keypoint owns a MapPoint, frame counts for BAError, 3d-3d enabled)
constexpr int kRefitSearch = 1, kRefitSums = 2;
// Diagnostic counters (iba_debug_counters; tools/rescan_probe.py) are compiled in by `make diag` only (-DIBA_DIAG_COUNTERS ->
// libiba_diag.so): three cold atomics in iba_assoc2_kernel moved its code generation enough to cost 8 us of 171 (profiles r04e vs r04f).
#ifdef IBA_DIAG_COUNTERS
#define IBA_DIAG_COUNT(cond, slot) do { if (cond) atomicAdd(dp.diag + (slot), 1u); } while (0)
#else
#define IBA_DIAG_COUNT(cond, slot) do { } while (0)
#endif
constexpr int kHardLds = 256;   // iba_assoc_kernel: scan points within 0.1 m of the camera plane that a block notes for its tie pass (more: the full rescan)
constexpr uint32_t kFlagA = 2u;   // association-path 1-NN wanted (ComputeLocalNeighbor at the matched point is valid)

// ------------------------------------------------------------------------------------------------------------------
// the gates on a local-plane record, shared by the memoised and the refit paths
template <class PRM> __device__ __forceinline__ bool local_neigh_ok(const PRM& prm, const PlaneRec& rec) { return !(rec.k < prm.neigh_min_pts || rec.far_d2 < prm.local_min_diff_dist2); }
template <class PRM> __device__ __forceinline__ bool local_plane_ok(const PRM& prm, const PlaneRec& rec) { return rec.reg_sum / (double)(rec.k - 1) < prm.local_norm_reg_threshold; }
// point-to-plane / point-to-point distance of the 3d-3d cost term (iba_global.cpp:111-156, 241-249) for e = neighbour - query;
// rec: the plane at the neighbour (has_rec false when use_plane is off). dist >= 0: the sign bit carries the kind (set: point-to-point)
template <class PRM> __device__ __forceinline__ double cost_res(const PRM& prm, bool has_rec, const PlaneRec& rec, double ex, double ey, double ez) {
    double dist = sqrt((ex * ex + ey * ey) + ez * ez);
    bool is_plane = false;
    if (has_rec) {
        if (!(rec.far_d2 < prm.min_diff_dist2) && !(rec.k < prm.norm_min_pts) && !(rec.reg_sum / (double)(rec.k - 1) > prm.norm_reg_threshold)) {
            dist = fabs(ex * rec.nx + ey * rec.ny + ez * rec.nz);
            is_plane = true;
        }
    }
    return is_plane ? dist : -dist;
}

// ------------------------------------------------------------------------------------------------------------------
// The second half of an association block, shared by iba_assoc_kernel and iba_assoc2_kernel: from the winners of the 2d-3d
// association (s_best_idx[k] = original index of keypoint k's scan point, or kNone) to corrset size, the work list in keypoint
// order, the local plane at the matched point, the 3d-2d covisible residuals (iba_global.cpp:291-328) and the partial record.
// One function, one order of every sum: which kernel ran the first half cannot be told from the results.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(4))) const KArgs KArgsC;
// The tail walks the frame's FLAGGED keypoints — the ones that own a MapPoint or have a covisible match (DevProblem::fkp: ~40 % of the
// keypoints of a frame, ascending ids) — not every keypoint (round 5): a keypoint without either can own no term, and corrset.size(), the
// one thing that needed all of them, is counted where the winners are made (`first_hits`: the number of keypoints this thread was the first
// to reach; see grid_match). Q > 0 (at most Q x 512 flagged keypoints per frame): the thread's Q list entries (keypoint, flag word) arrive in
// registers (rfa, rfb: loaded at the start of the kernel), their winners are gathered once and stay in registers for the list pass, and the
// covisible-slot mask of a list entry is parked next to it (s_msk, over the winners' array). Q = 0: any number, read from the list where needed.
template <int Q, bool MANY = true, int T = IBA_THREADS>   // MANY: a frame may have more covisible keyframes than the flag word has match bits (the second word, kp_fl2); T: threads of the block
__device__ __forceinline__ void assoc_tail(KArgsC* ka, const FrameHdr& h, const Cand& cd, const FrameCtx& c, uint32_t* s_best_idx, const uint4 rfa, const uint4 rfb, uint32_t* s_list,
                                           double* s_red, const double* s_rel, const uint32_t K, const int want, const int dbg, const bool refit, const int b, const int f, const int nf,
                                           double* __restrict__ part, uint4* __restrict__ flist,
                                           uint32_t* __restrict__ fcount, uint32_t* __restrict__ lcount, const int flist_stride, const uint32_t first_hits) {
#define dp (ka->dp)
#define prm (ka->prm)
    constexpr int kThreads = T, kWaves = T / 64;   // (of THIS block: iba_assoc2_kernel runs blocks of 256 or of 512 threads)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t* inv_perm = dp.inv_perm + h.pt_base;
    const uint2* fk = dp.fkp + h.fk_base;
    const uint32_t Kw = h.n_fk;

    const double s = cd.s;
    uint32_t n3 = 0;
    // ---- phase 3: the work list in keypoint order, in one pass over the flagged keypoints: every thread owns a run of q consecutive list
    //      entries, counts the ones that go to the cost list / the association list, an inclusive DPP scan of the packed counts gives its
    //      place in the wave, the waves' totals go through LDS (the one barrier), and the entries are written. Work list = keypoints with a
    //      correspondence that can own a term: a MapPoint and/or a covisible match for the cost (iba_global.cpp:225, 295-300), both for a
    //      residual block (iba_local.cpp:213, 259-260); entry = k | w << 16 (bits 16,17: MapPoint / covisible-match flags). ----
    // which keypoints the association list holds: a MapPoint AND a covisible match for BuildProblem's blocks (iba_local.cpp:213, 259-260); a
    // covisible match alone when the 3d-2d residual is IBATestEdge (factor_3d2d_kind = 1: the edge set of BAError's 3d-2d loop, iba_global.cpp:295-300)
    const uint32_t amask = prm.p2pix ? 2u : 3u;
    constexpr int QR = Q > 0 ? Q : 1;
    const uint32_t q = Q > 0 ? (uint32_t)Q : ((Kw + (uint32_t)kThreads * 2u - 1u) / ((uint32_t)kThreads * 2u)) * 2u;   // list entries per thread (even: read two at a time)
    const uint32_t e0 = (uint32_t)tid * q;
    uint32_t n_corr = 0u;
    // counts of this thread's run: cost-list | association-list << 16 entries (K < 65 535: each fits 16 bits, so do the block totals)
    uint32_t mine = 0u;
    uint32_t kk[QR], ff[QR], bw[QR];      // Q > 0: keypoint, flag word and winner of the thread's entries
    uint32_t mC_keep = 0u, mA_keep = 0u;  // ... and which of them go to the cost / the association list
    if (Q > 0) {
        const uint32_t rk4[4] = {rfa.x, rfa.z, rfb.x, rfb.z}, rf4[4] = {rfa.y, rfa.w, rfb.y, rfb.w};
#pragma unroll
        for (int j = 0; j < QR; ++j) {
            kk[j] = rk4[j]; ff[j] = rf4[j];
            bw[j] = e0 + (uint32_t)j < Kw ? s_best_idx[kk[j]] : kNone;
            const uint32_t w = bw[j] != kNone ? ff[j] : 0u;
            mC_keep |= (w != 0u ? 1u : 0u) << j; mA_keep |= ((w & amask) == amask ? 1u : 0u) << j;
        }
        mine = (uint32_t)__popc(mC_keep) | ((uint32_t)__popc(mA_keep) << 16);
    } else {
        for (uint32_t g = 0; g < q; g += 2u) {   // (block-uniform trip count)
            uint4 two = make_uint4(kNone, 0u, kNone, 0u);
            if (e0 + g < Kw) two = *(const uint4*)(fk + e0 + g);
            const uint32_t k2[2] = {two.x, two.z}, f2[2] = {two.y, two.w};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const uint32_t bwv = e0 + g + (uint32_t)j < Kw ? s_best_idx[k2[j]] : kNone;
                const uint32_t w = bwv != kNone ? f2[j] : 0u;
                mine += (w != 0u ? 1u : 0u) | (((w & amask) == amask ? 1u : 0u) << 16);
            }
        }
    }
    const uint32_t incl = wave_sum_u32(mine);         // inclusive prefix over the lanes of the wave (the total in lane 63)
    const uint32_t fhw = wave_sum_u32(first_hits);    // keypoints this wave was the first to reach
    uint2* s_cnt = (uint2*)s_red;   // (list counts, first hits) of every wave (the reduction slab is not in use yet)
    if (lane == 63) s_cnt[wave] = make_uint2(incl, fhw);
    __syncthreads();
    uint32_t before = 0u, total = 0u;
    for (int w = 0; w < kWaves; ++w) { const uint2 t = s_cnt[w]; total += t.x; n_corr += t.y; if (w < wave) before += t.x; }
    const bool usedA = (want & 1) && !((int)n_corr < prm.num_min_corr);        // iba_local.cpp:192
    const bool usedC = (want & 2) && !((int)n_corr < prm.num_min_corr_cost);   // iba_global.cpp:203
    uint4* fl = flist + ((size_t)b * nf + f) * (size_t)flist_stride;
    uint32_t* s_pos = s_list + K;        // per list item: matched scan point (tree position); aliases the 2nd half of best_d2
    uint32_t* s_msk = s_best_idx;        // Q > 0: per list item its covisible-slot mask (every winner was read before the barrier above)
    {
        const int sh = usedC ? 0 : 16;   // which of the two lists this evaluation builds
        n3 = (usedC || usedA) ? ((total >> sh) & 0xffffu) : 0u;
        uint32_t at = ((before + (incl - mine)) >> sh) & 0xffffu;   // entries of the threads before this one
        if ((usedC || usedA) && Q > 0) {
            const uint32_t keep = usedC ? mC_keep : mA_keep;
            uint32_t ip[QR];
#pragma unroll
            for (int j = 0; j < QR; ++j) ip[j] = ((keep >> j) & 1u) ? inv_perm[bw[j]] : 0u;   // the gathers are in flight together
#pragma unroll
            for (int j = 0; j < QR; ++j)
                if ((keep >> j) & 1u) { s_list[at] = kk[j] | ((ff[j] & 3u) << 16); s_pos[at] = ip[j]; s_msk[at] = ff[j] >> 2; ++at; }
        } else if (usedC || usedA) {
            for (uint32_t g = 0; g < q && e0 + g < Kw; g += 2u) {
                const uint4 two = *(const uint4*)(fk + e0 + g);
                const uint32_t k2[2] = {two.x, two.z}, f2[2] = {two.y, two.w};
                uint32_t ip[2], bv2[2]; bool wk[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    bv2[j] = e0 + g + (uint32_t)j < Kw ? s_best_idx[k2[j]] : kNone;
                    const uint32_t w = bv2[j] != kNone ? f2[j] : 0u;
                    wk[j] = usedC ? w != 0u : (w & amask) == amask;
                    ip[j] = wk[j] ? inv_perm[bv2[j]] : 0u;
                }
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (wk[j]) { s_list[at] = k2[j] | ((f2[j] & 3u) << 16); s_pos[at] = ip[j]; ++at; }   // (the slot mask is read from kp_fl where it is needed)
            }
        }
    }
    __syncthreads();
    if (dbg == 6) return;
    // The 3d-2d sum keeps the order of a block of IBA_THREADS threads whatever T is (a candidate's bits may not depend on the block size its
    // launch was given): a thread of a smaller block stands for V = IBA_THREADS / T threads of the big one — entry i belongs to thread
    // i % IBA_THREADS there — with an accumulator for each, and their wave sums go to the slots those threads' waves would have filled.
    constexpr int V = IBA_THREADS / T;
    static_assert(V == 1 || V == 2, "assoc_tail: blocks of IBA_THREADS or IBA_THREADS / 2 threads");
    double sum2d = 0.0, sum2d_hi = 0.0;
    uint32_t c2 = 0, v2 = 0;
    // association: local plane at the matched point (iba_local.cpp:207-231); list entry for the search / factor kernels
    for (uint32_t i = tid; i < n3; i += kThreads) {
        const uint32_t e = s_list[i], k = e & 0xffffu;
        uint32_t ax_ = kNone, flags = 0u;
        if (usedA && ((e >> 16) & amask) == amask) {
            const uint32_t pos = s_pos[i];
            // IBATestEdge mode: the 3d-2d edges need no plane (.y = the matched point, always); the 3d-3d block keeps BuildProblem's
            // conditions (MapPoint, valid neighbourhood) and exists only when the 3d-3d term is on (err_weight[1], iba_global.cpp:214-220)
            const bool want3 = !prm.p2pix || (prm.use_3d3d && ((e >> 16) & 1u));
            if (refit) { ax_ = pos; if (want3) flags |= kFlagA; }
            else if (prm.p2pix) { ax_ = pos; if (want3 && (dp.plane_ok[h.pt_base + pos] & 1u)) flags |= kFlagA; }
            else {   // the two verdicts of the memoised local plane at the matched point: one byte instead of the 48-byte record
                const uint32_t v = dp.plane_ok[h.pt_base + pos];
                if ((v & 3u) == 3u) ax_ = pos;      // ComputeLocalNeighbor valid (pointcloud.h:752) and bvalid_plane (iba_local.cpp:231)
                if (v & 1u) flags |= kFlagA;        // no 3d-3d block either otherwise (the `continue` at :209-211)
            }
        }
        if (usedC && prm.use_3d3d && ((e >> 16) & 1u)) flags |= kFlagC;
        // (.z, the 3d-3d block, is filled in by iba_nn_kernel. Round 4 measured the 16-byte entry against 8 bytes — the store skipped
        //  altogether: 2 us of 186; 8 bytes + a 4-byte array for the search kernel's pick: search + 7 us, factors + 8 us (two loads per
        //  entry instead of one); one packed 8-byte word: search + 10 us (spills at its 128-VGPR cap), factors + 3 us, association
        //  unchanged. These kernels do not wait for HBM: tools/experiments/r04_list_entry_8_bytes.patch.)
        fl[i] = make_uint4(k, ax_, kNone, flags);
    }
    if (dbg == 7) return;
    // K6: 3d-2d covisible reprojection residuals (iba_global.cpp:291-328): only the slots whose match bit is set
    if (usedC) {
        for (uint32_t i = tid; i < n3; i += kThreads) {
            const uint32_t k = s_list[i] & 0xffffu, pos = s_pos[i];
            const uint32_t mask_lo = Q > 0 ? s_msk[i] : dp.kp_fl[h.kp_base + k] >> 2;
            const uint32_t mask_hi = (MANY && h.n_slots > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + k] : 0u;   // (block-uniform: a frame with more than 30 covisible keyframes)
            if (!(mask_lo | mask_hi)) continue;
            float xf_, yf_, zf_; load_pt<true>(c, pos, xf_, yf_, zf_);
            const double x = (double)xf_, y = (double)yf_, z = (double)zf_;
            const double p0x = ((c.R[0] * x + c.R[1] * y) + c.R[2] * z) + c.t[0];
            const double p0y = ((c.R[3] * x + c.R[4] * y) + c.R[5] * z) + c.t[1];
            const double p0z = ((c.R[6] * x + c.R[7] * y) + c.R[8] * z) + c.t[2];
            const float2* mrow = dp.match_uv + h.match_base + k;
#pragma unroll 1
            for (int wi = 0; wi < ((MANY && mask_hi) ? 2 : 1); ++wi) {   // the slots of one flag word after the other, in slot order (one copy of the loop body)
                uint32_t mask = (!MANY || wi == 0) ? mask_lo : mask_hi;
                const uint32_t base = (!MANY || wi == 0) ? 0u : (uint32_t)kCovisWord;
                if (!mask) continue;
                float2 mm = mrow[(size_t)(base + (uint32_t)__ffs((int)mask) - 1u) * K];
                while (mask) {
                    const uint32_t sl = base + (uint32_t)__ffs((int)mask) - 1u;
                    mask &= mask - 1u;
                    const float2 cur = mm;
                    if (mask) mm = mrow[(size_t)(base + (uint32_t)__ffs((int)mask) - 1u) * K];   // next match is in flight during the arithmetic
                    const double* rel = s_rel + sl * 12;
                    const double p1x = ((rel[0] * p0x + rel[1] * p0y) + rel[2] * p0z) + rel[3] * s;
                    const double p1y = ((rel[4] * p0x + rel[5] * p0y) + rel[6] * p0z) + rel[7] * s;
                    const double p1z = ((rel[8] * p0x + rel[9] * p0y) + rel[10] * p0z) + rel[11] * s;
                    double qu, qv;
                    div2(h.fx * p1x, h.fy * p1y, p1z, qu, qv);
                    const double ou = qu + h.cx, ov = qv + h.cy;
                    if (!(ou >= 0 && ou < h.W && ov >= 0 && ov < h.H)) continue;
                    const double eu = ou - (double)cur.x, ev = ov - (double)cur.y;
                    const double dist = sqrt(eu * eu + ev * ev);
                    if (dist < prm.corr_3d_2d_threshold) { if (V == 2 && ((i / (uint32_t)kThreads) & 1u)) sum2d_hi += dist; else sum2d += dist; ++v2; }
                    ++c2;
                }
            }
        }
    }
    if (tid == 0) { fcount[(size_t)b * nf + f] = usedA ? n3 : 0u; lcount[(size_t)b * nf + f] = n3; }
    // K8: reduction -> record. The 3d-3d sums of this (candidate, frame) come from iba_nn_kernel's own records.
    {
        const double w2d = wave_sum_f64(sum2d);
        const double w2d_hi = V == 2 ? wave_sum_f64(sum2d_hi) : 0.0;
        const unsigned long long wa = wave_sum_u64((unsigned long long)c2 | ((unsigned long long)v2 << 32));
        unsigned long long* s_redu = (unsigned long long*)s_red;
        __syncthreads();
        if (lane == 63) { s_red[wave * 4 + 0] = w2d; s_redu[wave * 4 + 2] = wa; if (V == 2) { s_red[(wave + kWaves) * 4 + 0] = w2d_hi; s_redu[(wave + kWaves) * 4 + 2] = 0ull; } }
        __syncthreads();
        if (tid < kPartialStride) {
            double out = 0.0;
            if (usedC) {
                if (tid == P_SUM_3D2D) { for (int w = 0; w < kWaves * V; ++w) out += s_red[w * 4 + 0]; }
                else if (tid == P_CNT_3D2D || tid == P_VALID_3D2D) {
                    unsigned long long a = 0;
                    for (int w = 0; w < kWaves; ++w) a += s_redu[w * 4 + 2];
                    out = (double)(tid == P_CNT_3D2D ? (a & 0xffffffffull) : (a >> 32));
                }
                else if ((tid == P_CNT_3D3D || tid == P_VALID_3D3D) && !prm.use_3d3d) out = 1.0;   // iba_global.cpp:214-220
                else if (tid == P_FRAMES) out = 1.0;
                else if (tid == P_NCORR) out = (double)n_corr;
                // (the hand-eye VALUE of a counted (candidate, frame) is evaluated by the summing kernel, iba_reduce2_kernel: K7 left the head of the chain in round 5)
                else if (tid == P_HE_CNT) out = h.he_valid ? 1.0 : 0.0;
            }
            if (tid == P_FRAMES_N) out = usedA ? 1.0 : 0.0;
            else if (tid == P_NCORR_N) out = usedA ? (double)n_corr : 0.0;
            part[tid] = out;
        }
    }
#undef dp
#undef prm
}

// THE HEAD OF A CHAIN (round 5). Every kernel of a chain but the first reads the candidates from device memory; rounds 1-4 copied them
// there with a staging launch of their own (12 us of latency at the head of every evaluation, or a second stream and two event hops to
// hide it). Now the FIRST kernel of the chain carries the copy in a few spare workgroups: `head_n16` 16-byte words from the pinned ring
// (`head_src`, as the device sees it) to `head_dst`. When that kernel is an association kernel its own blocks read their candidate (13
// doubles) from the pinned ring directly; when the pair search runs first it carries the copy and the association reads the device copy.
__device__ __forceinline__ void chain_head_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16, uint32_t block, uint32_t n_blocks, uint32_t threads) {
    for (uint32_t i = block * threads + threadIdx.x; i < n16; i += n_blocks * threads) dst[i] = src[i];
}
constexpr uint32_t kHeadBlocks = 16;   // spare workgroups appended to an association kernel's grid for the copy (64 candidates = 4096 words: half a pass)

// iba_assoc_kernel. want: bit 0 = BuildProblem association wanted, bit 1 = BAError cost wanted, bit 2 = planes are refitted (plane_cache = 0). corr_out != nullptr: dump
// the correspondences and return (iba_get_correspondences).
// grid: 8 * ceil(n_frames/8) * B blocks of kThreads; block i runs on XCD i%8, so all candidates of a frame share an L2.
// ------------------------------------------------------------------------------------------------------------------
#ifdef IBA_ASSOC_WAVES
#define IBA_ASSOC_ATTR __attribute__((amdgpu_waves_per_eu(IBA_ASSOC_WAVES, IBA_ASSOC_WAVES)))
#else
#define IBA_ASSOC_ATTR
#endif
template <int Q>   // flagged keypoints per thread that the tail keeps in registers (2, 4; 0: any number): see assoc_tail
__global__ __launch_bounds__(kThreads) IBA_ASSOC_ATTR void iba_assoc_kernel(KArgs ka_by_value, const Cand* __restrict__ cands, int B, int want,
                                                             double* __restrict__ frame_partials, int nrec, uint32_t* __restrict__ corr_out,
                                                             uint4* __restrict__ flist, uint32_t* __restrict__ fcount,
                                                             uint32_t* __restrict__ lcount, int flist_stride, const uint4* __restrict__ head_src, uint4* __restrict__ head_dst, uint32_t head_n16) {
    extern __shared__ __align__(16) unsigned char smem[];
    KArgsC* ka = (KArgsC*)__builtin_amdgcn_kernarg_segment_ptr();   // see iba_frame_kernel: parameter blocks are read where they are used
    (void)ka_by_value;
    {   // the chain's head rides in the blocks behind the (frame, candidate) grid
        const uint32_t main_blocks = 8u * (uint32_t)((ka->dp.n_frames + 7) / 8) * (uint32_t)B;
        if (blockIdx.x >= main_blocks) { chain_head_copy(head_src, head_dst, head_n16, blockIdx.x - main_blocks, gridDim.x - main_blocks, (uint32_t)kThreads); return; }
    }
#define dp (ka->dp)
#define prm (ka->prm)
#define lay (ka->lay)
#define IBA_RELOAD() asm volatile("" : "+s"(ka))
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nf = dp.n_frames;
    const int per_xcd = (nf + 7) / 8;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int f = xcd + 8 * (jj / B), b = jj % B;
    if (f >= nf || jj / B >= per_xcd) return;
    const FrameHdr& h = dp.frames[f];
    const Cand& cd = cands[b];
    double* part = frame_partials + ((size_t)b * nrec + f) * kPartialStride;

    unsigned long long* s_best_d2 = (unsigned long long*)(smem + lay.off_best_d2);
    uint32_t* s_best_idx = (uint32_t*)(smem + lay.off_best_idx);
    uint32_t* s_bitmap = (uint32_t*)(smem + lay.off_bitmap);
    uint16_t* s_cstart = (uint16_t*)(smem + lay.off_cstart);
    double* s_red = (double*)(smem + lay.off_red);
    double* s_rel = s_red + kWaves * 4;
    uint32_t* s_wcnt = (uint32_t*)(s_rel + lay.rel_slots * 12u);
    uint32_t* s_misc = s_wcnt + kWaves;
    uint32_t* s_hard = s_misc + 4;                            // [kHardLds] tree positions of the points the float pass could not decide (s_misc[3] of them)
    uint32_t* s_cand = (uint32_t*)(smem + lay.off_cand);
    uint32_t* s_list = (uint32_t*)(smem + lay.off_best_d2);   // aliases best_d2 after phase 2
    float2* s_kuv = (float2*)(smem + lay.off_kuv);            // (u, v) of every keypoint; keypoint ids are in grid-record order

    const uint32_t P = h.P, Ppad = h.Ppad, K = h.K;
    const float* gxs = dp.xs + h.pt_base; const float* gys = dp.ys + h.pt_base; const float* gzs = dp.zs + h.pt_base;

    // ---- phase 0: LDS init. The static tables come from global memory (L2) in batches: all loads of a batch are issued before
    //      the first store waits for one, so the phase costs about two memory round trips instead of one per table row ----
    const uint32_t nbw = (h.gw * h.gh + 31u) >> 5;
    const uint32_t ncs = h.gwc * h.ghc + 1u;
    const uint32_t ut = (uint32_t)tid;
    double rv = 0.0;
    if (ut < h.n_slots * 12u) rv = dp.slots[h.slot_base + ut / 12].rel[ut % 12];
    const float4* boxes = (const float4*)(dp.chunk_box + 8 * h.box_base);   // two 16-byte loads per chunk; the first pass of the frustum test is fetched here
    const uint32_t nchunks = (P + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;
    float4 blo_n = make_float4(0.f, 0.f, 0.f, 0.f), bhi_n = blo_n;
    if (ut < nchunks) { blo_n = boxes[2 * (size_t)ut]; bhi_n = boxes[2 * (size_t)ut + 1]; }
    {
        const uint32_t* gbm = dp.bitmap + h.bitmap_base; const uint32_t* gcs = dp.coarse_start + h.coarse_base;
        const float2* guv = dp.kp_uv + h.kp_base;
        uint32_t bw[8], cw[4]; float2 uv[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t i = ut + (uint32_t)j * kThreads; bw[j] = i < nbw ? gbm[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = ut + (uint32_t)j * kThreads; cw[j] = i < ncs ? gcs[i] : 0u; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = ut + (uint32_t)j * kThreads; uv[j] = make_float2(0.f, 0.f); if (i < K) uv[j] = guv[i]; }
#pragma unroll
        for (int j = 0; j < 8; ++j) { const uint32_t i = ut + (uint32_t)j * kThreads; if (i < nbw) s_bitmap[i] = bw[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = ut + (uint32_t)j * kThreads; if (i < ncs) s_cstart[i] = (uint16_t)cw[j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint32_t i = ut + (uint32_t)j * kThreads; if (i < K) { s_best_d2[i] = ~0ull; s_best_idx[i] = kNone; s_kuv[i] = uv[j]; } }
        // what the first batch did not cover (more than 2048 keypoints, a larger image)
        for (uint32_t i = ut + 8u * kThreads; i < nbw; i += kThreads) s_bitmap[i] = gbm[i];
        for (uint32_t i = ut + 4u * kThreads; i < ncs; i += kThreads) s_cstart[i] = (uint16_t)gcs[i];
        for (uint32_t i = ut + 4u * kThreads; i < K; i += kThreads) { s_best_d2[i] = ~0ull; s_best_idx[i] = kNone; s_kuv[i] = guv[i]; }
    }
    if (ut < h.n_slots * 12u) s_rel[ut] = rv;
    for (uint32_t i = ut + kThreads; i < h.n_slots * 12u; i += kThreads) s_rel[i] = dp.slots[h.slot_base + i / 12u].rel[i % 12u];   // more than 42 covisible keyframes: beyond one store per thread
    if (tid < 4) s_misc[tid] = 0u;
    __syncthreads();

    const int dbg = want >> 8;   // diagnostic: cut the kernel short after a phase (timing attribution; results are garbage)
    const bool refit = (want & 4) != 0;   // plane_cache = 0: the local planes are fitted after this kernel (iba_fit_kernel<.., 1>), which then settles .y and kFlagA
    if (dbg == 1) return;
    uint32_t first = 0u;   // keypoints this thread was the first to reach with a point inside max_pixel_dist: their block sum is corrset.size() (see grid_match)
    uint4 rfa = make_uint4(kNone, 0u, kNone, 0u), rfb = rfa;   // Q > 0: this thread's Q entries of the frame's flagged-keypoint list, in flight until the tail
    if (Q > 0) {
        const uint2* fk = dp.fkp + h.fk_base;
        const uint32_t e0 = ut * (uint32_t)Q;
        if (e0 < h.n_fk) rfa = *(const uint4*)(fk + e0);
        if (Q > 2 && e0 + 2u < h.n_fk) rfb = *(const uint4*)(fk + e0 + 2u);
    }
    FrameCtx c;
    c.xs = gxs; c.ys = gys; c.zs = gzs;
    c.nodes = nullptr; c.bitmap = s_bitmap; c.best_d2 = s_best_d2; c.best_idx = s_best_idx;
    c.cstart = s_cstart; c.gwc = (int)h.gwc; c.crec = dp.crec + h.kp_base;
    c.perm = dp.perm + h.pt_base;
    c.p4 = dp.pts4 + h.pt_base;
    c.gw = (int)h.gw; c.gh = (int)h.gh; c.margin = (float)prm.grid_margin; c.gate2 = prm.gate2;
    c.fx = h.fx; c.cx = h.cx; c.cy = h.cy; c.W = h.W; c.H = h.H;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = cd.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = cd.t[i];

    IBA_RELOAD();
    const uint32_t cand_cap = lay.cand_cap;
    // ---- phase 0.5: conservative frustum test of the static per-chunk boxes (see iba_frame_kernel) ----
    uint32_t* s_vis = (uint32_t*)(smem + lay.off_vis);
    {
        const float m = 8.0f;
        const float A[5][3] = {{(float)c.fx, 0.f, (float)c.cx + m}, {-(float)c.fx, 0.f, (float)c.W + m - (float)c.cx},
                               {0.f, (float)c.fx, (float)c.cy + m}, {0.f, -(float)c.fx, (float)c.H + m - (float)c.cy}, {0.f, 0.f, 1.f}};
        float N[5][3], Dd[5], N1[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int j = 0; j < 3; ++j) N[q][j] = (float)c.R[0 * 3 + j] * A[q][0] + (float)c.R[1 * 3 + j] * A[q][1] + (float)c.R[2 * 3 + j] * A[q][2];
            Dd[q] = (float)c.t[0] * A[q][0] + (float)c.t[1] * A[q][1] + (float)c.t[2] * A[q][2] + (q == 4 ? 0.2f : 0.f);
            N1[q] = fabsf(N[q][0]) + fabsf(N[q][1]) + fabsf(N[q][2]);
        }
        uint32_t vis_cnt = 0u;
        for (uint32_t ch0 = 0; ch0 < nchunks; ch0 += kThreads) {
            const uint32_t ch = ch0 + (uint32_t)tid;
            bool vis = false;
            if (ch < nchunks) {
                const float4 blo = ch0 == 0u ? blo_n : boxes[2 * (size_t)ch], bhi = ch0 == 0u ? bhi_n : boxes[2 * (size_t)ch + 1];
                const float lo3[3] = {blo.x, blo.y, blo.z}, hi3[3] = {bhi.x, bhi.y, bhi.z};
                vis = true;
                const float rmax = bhi.w;   // largest |coordinate| of the box: |n . x| <= |n|_1 rmax bounds the slack term
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    float smax = Dd[q];
#pragma unroll
                    for (int j = 0; j < 3; ++j) smax += fmaxf(N[q][j] * hi3[j], N[q][j] * lo3[j]);   // the farthest corner along n
                    const float mag = fmaf(N1[q], rmax, fabsf(Dd[q]));
                    vis = vis && !(smax < -1e-3f * mag - 1e-6f);   // NaN boxes (empty chunk) compare false -> kept, harmless
                }
            }
            const unsigned long long bal = __ballot(vis);
            vis_cnt += (uint32_t)__popcll(bal);
            if (lane == 0) { s_vis[(ch0 >> 5) + 2u * (uint32_t)wave] = (uint32_t)bal; s_vis[(ch0 >> 5) + 2u * (uint32_t)wave + 1u] = (uint32_t)(bal >> 32); }
        }
        if (lane == 0) s_wcnt[wave] = vis_cnt;
    }
    __syncthreads();
    // visible chunks, compacted into a u16 list right behind the ballot words: every wave appends the chunks it tested
    const uint32_t nchunks_all = (P + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;
    uint16_t* s_vlist = (uint16_t*)(s_vis + lay.vis_words);
    uint32_t n_vis = 0u;
    {
        uint32_t at = 0u;
        for (int w = 0; w < kWaves; ++w) { const uint32_t cw = s_wcnt[w]; n_vis += cw; if (w < wave) at += cw; }
        for (uint32_t ch0 = 0; ch0 < nchunks_all; ch0 += kThreads) {
            const unsigned long long bal = (unsigned long long)s_vis[(ch0 >> 5) + 2u * (uint32_t)wave] | ((unsigned long long)s_vis[(ch0 >> 5) + 2u * (uint32_t)wave + 1u] << 32);
            if ((bal >> lane) & 1ull) s_vlist[at + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)(ch0 + (uint32_t)tid);
            at += (uint32_t)__popcll(bal);
        }
    }
    __syncthreads();
    if (dbg == 2) return;
    // ---- phase 1a: stream the visible chunks (16 B/lane), float32 projection, one bit of the dilated reject bitmap ----
    {
        const float r0 = (float)c.R[0], r1 = (float)c.R[1], r2 = (float)c.R[2], r3 = (float)c.R[3], r4 = (float)c.R[4], r5 = (float)c.R[5],
                    r6 = (float)c.R[6], r7 = (float)c.R[7], r8 = (float)c.R[8], t0 = (float)c.t[0], t1 = (float)c.t[1], t2 = (float)c.t[2];
        const float fxf = (float)c.fx, cxf = (float)c.cx, cyf = (float)c.cy, Wf = (float)c.W + 1.0f, Hf = (float)c.H + 1.0f;
        constexpr float kInvCell = 1.0f / (float)kGridCell;
        const uint32_t n_groups = n_vis * (uint32_t)(kChunk / 4);
        const uint32_t n_iter = (n_groups + kThreads - 1u) / kThreads;
        const float qn = __builtin_nanf("");
        const float4 nan4 = make_float4(qn, qn, qn, qn);
        auto group_base = [&](uint32_t g) -> uint32_t {
            if (g >= n_groups) return kNone;
            const uint32_t b4 = (uint32_t)s_vlist[g / (uint32_t)(kChunk / 4)] * (uint32_t)kChunk + (g % (uint32_t)(kChunk / 4)) * 4u;
            return b4 < Ppad ? b4 : kNone;
        };
        // two passes ahead: a frame streams ~2 passes per lane, so both are in flight before the first is looked at
        float4 X = nan4, Y = nan4, Z = nan4, X1 = nan4, Y1 = nan4, Z1 = nan4;
        uint32_t base = group_base((uint32_t)tid), base1 = group_base((uint32_t)tid + kThreads);
        if (base != kNone) { X = *(const float4*)(gxs + base); Y = *(const float4*)(gys + base); Z = *(const float4*)(gzs + base); }
        if (base1 != kNone) { X1 = *(const float4*)(gxs + base1); Y1 = *(const float4*)(gys + base1); Z1 = *(const float4*)(gzs + base1); }
        for (uint32_t it = 0; it < n_iter; ++it) {
            const uint32_t nbase = group_base((uint32_t)tid + (it + 2u) * kThreads);
            float4 Xn = nan4, Yn = nan4, Zn = nan4;
            if (nbase != kNone) { Xn = *(const float4*)(gxs + nbase); Yn = *(const float4*)(gys + nbase); Zn = *(const float4*)(gzs + nbase); }
            const float px[4] = {X.x, X.y, X.z, X.w}, py[4] = {Y.x, Y.y, Y.z, Y.w}, pz[4] = {Z.x, Z.y, Z.z, Z.w};
            bool pass[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float zc = fmaf(r6, px[j], fmaf(r7, py[j], fmaf(r8, pz[j], t2)));
                pass[j] = false;
                if (zc > 0.1f) {
                    const float xc = fmaf(r0, px[j], fmaf(r1, py[j], fmaf(r2, pz[j], t0)));
                    const float yc = fmaf(r3, px[j], fmaf(r4, py[j], fmaf(r5, pz[j], t1)));
                    const float rz = __builtin_amdgcn_rcpf(zc);
                    const float uf = fmaf(fxf * xc, rz, cxf), vf = fmaf(fxf * yc, rz, cyf);
                    if (uf > -1.0f && uf < Wf && vf > -1.0f && vf < Hf) {   // inside the padded image: the cell needs no clamping
                        const uint32_t cell = (uint32_t)((int)floorf(vf * kInvCell) + 1) * (uint32_t)c.gw + (uint32_t)((int)floorf(uf * kInvCell) + 1);
                        pass[j] = (s_bitmap[cell >> 5] >> (cell & 31)) & 1u;
                    }
                } else if (zc > -0.1f) pass[j] = true;   // undecidable in f32 (NaN padding fails both tests): exact path decides
            }
            const unsigned long long b0 = __ballot(pass[0]), b1 = __ballot(pass[1]), b2 = __ballot(pass[2]), b3 = __ballot(pass[3]);
            const uint32_t n0 = (uint32_t)__popcll(b0), n1 = (uint32_t)__popcll(b1), n2 = (uint32_t)__popcll(b2), n3q = (uint32_t)__popcll(b3);
            if (n0 + n1 + n2 + n3q) {
                uint32_t wb = 0;
                if (lane == 0) wb = atomicAdd(&s_misc[0], n0 + n1 + n2 + n3q);
                wb = __shfl(wb, 0);
                const unsigned long long lt = (1ull << lane) - 1ull;
                const uint32_t off[4] = {wb + (uint32_t)__popcll(b0 & lt), wb + n0 + (uint32_t)__popcll(b1 & lt), wb + n0 + n1 + (uint32_t)__popcll(b2 & lt),
                                         wb + n0 + n1 + n2 + (uint32_t)__popcll(b3 & lt)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (!pass[j]) continue;
                    if (off[j] < cand_cap) s_cand[off[j]] = base + j;
                    else {   // queue full: exact path inline, full rescan in phase 2 (speed only)
                        double u, v;
                        if (project_uv(c, px[j], py[j], pz[j], u, v)) first += grid_match<1>(c, u, v, base + j);
                        s_misc[1] = 1u;
                    }
                }
            }
            X = X1; Y = Y1; Z = Z1; base = base1; X1 = Xn; Y1 = Yn; Z1 = Zn; base1 = nbase;
        }
    }
    __syncthreads();
    if (dbg == 3) return;
    IBA_RELOAD();
    // ---- phase 1b: float32 walk of the keypoint grid for the queued points -> compact list of (point, keypoint) PAIRS that
    //      may be within max_pixel_dist. The f32 projection errs by < 0.3 px per axis for depth > 0.1 m (see phase 1a), so a
    //      pair whose f32 distance exceeds max_pixel_dist + 0.45 px cannot pass the exact test. About one queued point in
    //      four yields a pair: the f64 projection and the exact distance below run on those only, one lane per pair.
    const uint32_t ncand = min(s_misc[0], cand_cap);
    uint4* s_pair = (uint4*)(smem + lay.off_pair);   // {tree position, keypoint record -> keypoint id, d^2 bits}
    const uint32_t pair_cap = lay.pair_cap;
    {
        const float r0 = (float)c.R[0], r1 = (float)c.R[1], r2 = (float)c.R[2], r3 = (float)c.R[3], r4 = (float)c.R[4], r5 = (float)c.R[5],
                    r6 = (float)c.R[6], r7 = (float)c.R[7], r8 = (float)c.R[8], t0 = (float)c.t[0], t1 = (float)c.t[1], t2 = (float)c.t[2];
        const float fxf = (float)c.fx, cxf = (float)c.cx, cyf = (float)c.cy;
        const float rB = (float)prm.bitmap_margin, rB2 = rB * rB * 1.00001f, wm = rB + 0.01f;
        // the exact test of a pair when the pair list is full (speed only: ties are settled by the full rescan)
        auto test_now = [&](const float4& pv, uint32_t e) {
            const float2 rec = s_kuv[e];
            double u, v;
            if (project_uv(c, pv.x, pv.y, pv.z, u, v)) {
                const double du_ = (double)rec.x - u, dv_ = (double)rec.y - v;
                const double d2 = du_ * du_ + dv_ * dv_;
                if (d2 <= c.gate2) first += atomicMin(&s_best_d2[e], d2bits(d2)) == ~0ull ? 1u : 0u;
            }
            s_misc[1] = 1u;
        };
        // the gather of a pass is issued one pass ahead: every exposed global-memory round trip costs this kernel about 4 % of its time
        uint32_t pos_n = 0u; float4 pv_n = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((uint32_t)tid < ncand) { pos_n = s_cand[tid]; pv_n = c.p4[pos_n]; }
        for (uint32_t i0 = 0; i0 < ncand; i0 += kThreads) {   // wave-uniform trip count: the pairs of a wave are appended together
            const uint32_t i = i0 + (uint32_t)tid;
            uint32_t h0 = 0u, h1 = 0u, h2 = 0u, h3 = 0u, nh = 0u;   // up to four hits of this lane's point wait in registers
            const uint32_t pos = pos_n; const float4 pv = pv_n;
            if (i + kThreads < ncand) { pos_n = s_cand[i + kThreads]; pv_n = c.p4[pos_n]; }
            if (i < ncand) {
                const float zc = fmaf(r6, pv.x, fmaf(r7, pv.y, fmaf(r8, pv.z, t2)));
                if (!(zc > 0.1f)) {   // undecidable in f32 (queued by phase 1a for that reason): exact path inline; the point is noted for the tie pass
                    double u, v;
                    if (project_uv(c, pv.x, pv.y, pv.z, u, v)) first += grid_match<1>(c, u, v, pos);
                    // (Rounds 2-3 raised the overflow flag here, and with it a rescan of EVERY scan point in f64 for the ties: a 360-degree
                    //  scan always has a few dozen points within 0.1 m of the camera plane, so every block rescanned — 255 of the kernel's
                    //  716 us on a box-wide batch, tools/wide_cuts.sh. Now only these points are walked again.)
                    const uint32_t hs = atomicAdd(&s_misc[3], 1u);
                    if (hs < (uint32_t)kHardLds) s_hard[hs] = pos; else s_misc[1] = 1u;
                } else {
                    const float xc = fmaf(r0, pv.x, fmaf(r1, pv.y, fmaf(r2, pv.z, t0)));
                    const float yc = fmaf(r3, pv.x, fmaf(r4, pv.y, fmaf(r5, pv.z, t1)));
                    const float rz = __builtin_amdgcn_rcpf(zc);
                    const float uf = fmaf(fxf * xc, rz, cxf), vf = fmaf(fxf * yc, rz, cyf);
                    const int x0 = grid_cell(uf - wm, c.gw) >> kCoarseShift, x1 = grid_cell(uf + wm, c.gw) >> kCoarseShift;
                    const int y0 = grid_cell(vf - wm, c.gh) >> kCoarseShift, y1 = grid_cell(vf + wm, c.gh) >> kCoarseShift;
                    for (int yy = y0; yy <= y1; ++yy) {
                        const uint32_t e0 = c.cstart[yy * c.gwc + x0], e1 = c.cstart[yy * c.gwc + x1 + 1];
                        for (uint32_t e = e0; e < e1; ++e) {
                            const float2 rec = s_kuv[e];
                            const float du = rec.x - uf, dv = rec.y - vf;
                            if (fmaf(dv, dv, du * du) <= rB2) {
                                if (nh >= 4u) test_now(pv, h3);   // a fifth hit: the oldest one is settled right here
                                h3 = h2; h2 = h1; h1 = h0; h0 = e; ++nh;
                            }
                        }
                    }
                }
            }
            // one LDS atomic reserves the slots of all the hits of the wave
            const uint32_t nk = min(nh, 4u);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const unsigned long long b0 = __ballot(nk & 1u), b1 = __ballot(nk & 2u), b2 = __ballot(nk & 4u);
            const uint32_t tot = (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
            if (tot) {
                uint32_t sb = 0u;
                if (lane == 0) sb = atomicAdd(&s_misc[2], tot);
                sb = (uint32_t)__shfl((int)sb, 0);
                uint32_t slot = sb + (uint32_t)__popcll(b0 & lt) + 2u * (uint32_t)__popcll(b1 & lt) + 4u * (uint32_t)__popcll(b2 & lt);
                const uint32_t hh[4] = {h0, h1, h2, h3};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((uint32_t)q < nk) {
                        if (slot < pair_cap) s_pair[slot] = make_uint4(pos, hh[q], 0u, 0u);
                        else test_now(pv, hh[q]);
                        ++slot;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (dbg == 4 || dbg >= 29) return;
    // ---- phase 1c: exact f64 projection + FOV test (K1 + K2) and exact d^2 of every pair, ds_min_u64 on the keypoint's best ----
    const uint32_t npair = min(s_misc[2], pair_cap);
    const bool overflow = s_misc[1] != 0u;
    uint4 pr_n = make_uint4(0u, 0u, 0u, 0u); float4 pq_n = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((uint32_t)tid < npair) { pr_n = s_pair[tid]; pq_n = c.p4[pr_n.x]; }
    for (uint32_t i = tid; i < npair; i += kThreads) {
        const uint4 pr = pr_n;
        const float4 pv = pq_n;   // .w: original index of the point
        if (i + kThreads < npair) { pr_n = s_pair[i + kThreads]; pq_n = c.p4[pr_n.x]; }
        const float2 rec = s_kuv[pr.y];
        double u, v;
        uint32_t k = kNone; unsigned long long bits = 0ull;
        if (project_uv(c, pv.x, pv.y, pv.z, u, v)) {
            const double du = (double)rec.x - u, dv = (double)rec.y - v;
            const double d2 = du * du + dv * dv;
            if (d2 <= c.gate2) { k = pr.y; bits = d2bits(d2); first += atomicMin(&s_best_d2[k], bits) == ~0ull ? 1u : 0u; }
        }
        s_pair[i] = make_uint4(__float_as_uint(pv.w), k, (uint32_t)bits, (uint32_t)(bits >> 32));
    }
    __syncthreads();
    // ---- phase 2: the winner of each keypoint records its original index; exact ties -> lowest index ----
    for (uint32_t i = tid; i < npair; i += kThreads) {
        const uint4 pr = s_pair[i];
        if (pr.y != kNone && s_best_d2[pr.y] == ((unsigned long long)pr.z | ((unsigned long long)pr.w << 32))) atomicMin(&s_best_idx[pr.y], pr.x);
    }
    {   // the points whose depth the float pass could not decide: their exact walk again, for the ties
        const uint32_t nhard = min(s_misc[3], (uint32_t)kHardLds);
        for (uint32_t i = tid; i < nhard; i += kThreads) {
            const uint32_t pos = s_hard[i];
            double u, v;
            if (project_pos<true>(c, pos, u, v)) grid_match<2>(c, u, v, pos);
        }
    }
    IBA_DIAG_COUNT(overflow && tid == 0, 0);   // (iba_debug_counters: how often the fallback below runs)
    if (overflow) {   // some other exact tests ran inline (a full queue or pair list, a fifth hit of one point, more undecidable points than the list holds): every point again, for the ties
        for (uint32_t pos = tid; pos < P; pos += kThreads) {
            double u, v;
            if (project_pos<true>(c, pos, u, v)) grid_match<2>(c, u, v, pos);
        }
    }
    __syncthreads();

    if (corr_out) {   // dense dump: corr_out[kp_base + k] = original point index or kNone
        for (uint32_t k = tid; k < K; k += kThreads) corr_out[h.kp_base + k] = s_best_idx[k];
        return;
    }

    if (dbg == 5) return;
    IBA_RELOAD();
    assoc_tail<Q>(ka, h, cd, c, s_best_idx, rfa, rfb, s_list, s_red, s_rel, K, want, dbg, refit, b, f, nf, part, flist, fcount, lcount, flist_stride, first);
#undef dp
#undef prm
#undef lay
#undef IBA_RELOAD
}

// ------------------------------------------------------------------------------------------------------------------
// The association of a BATCH of nearby candidates: common pairs.
//
// iba_assoc_kernel spends 56 % of its vector instructions (r03 counters, tools/pmc_cuts.sh) on finding, for ONE candidate, the
// (scan point, keypoint) pairs that can be within max_pixel_dist of each other: the float stream of the visible chunks and the
// walk of the keypoint grid. The candidates of one call are usually close to each other (a poll of the mesh search late in its
// run, the perturbations of a line search, bench.py's batch): their projections of a point differ by a few pixels. So the
// pairs are found ONCE per keyframe for the whole batch, around a reference candidate, with a search radius per point that
// rigorously covers every candidate of the batch; each candidate then runs only the exact f64 test on that list.
//
// The bound. Candidate b maps a point p to q_b = R_b p + t_b = A_b q_0 + a_b with the reference's q_0 = R_0 p + t_0,
// A_b = R_b R_0^T, a_b = t_b - A_b t_0. With rho_ij = max_b |A_b - I|_ij and tau_i = max_b |a_b|_i (host, once per call),
// |q_b - q_0|_i <= delta_i = sum_j rho_ij |q_0|_j + tau_i for every b. For z_0 > 2 delta_z the pinhole projection
// (iba_global.cpp:72-73; v uses fx as the reference does) moves by
//     |u_b - u_0| <= Du = fx (delta_x z_0 + |x_0| delta_z) / (z_0 (z_0 - delta_z))            (and Dv alike),
// so a keypoint within max_pixel_dist of candidate b's projection is within r = max_pixel_dist + sqrt(Du^2 + Dv^2) (+ slack
// for rounding) of the reference's, and a point whose reference projection is more than (Du, Dv) outside the image is inside
// it for no candidate. Points with z_0 < -delta_z are behind every candidate's camera. What is left — depth not bounded away
// from zero and not provably outside the image — goes to a short "hard" list that every candidate projects exactly.
// Every decision of a candidate (visibility, d^2 <= gate^2, nearest point, ties) is still made by its own exact f64
// arithmetic in iba_assoc2_kernel, in the reference's expression order: the pair list only has to be a superset.
// ------------------------------------------------------------------------------------------------------------------
// rel[b] = (M_b row-major, a_b) as floats: candidate b relative to the reference, q_b = q_0 + M_b q_0 + a_b with M_b = R_b R_0^T - I,
// a_b = t_b - R_b R_0^T t_0 (the pair search bounds a block's motion over the batch by the candidates' own motions). The whole
// struct travels in the kernel arguments: the pair search depends on nothing the staging launch produces and runs beside it.
//
// GROUPS (round 4). A batch need not be tight as a whole: an optimiser's batch is typically two polls — around its feasible and
// its infeasible incumbent (iba_mads.hpp) — each tight once the mesh has shrunk, far from each other (tools/mads_trace_stats.py:
// 376 of 1276 batches of a calibration, half of its time). The host clusters the candidates into at most kMaxPairGroups groups
// (plan_pairs), every group has its own reference + bound (GroupRef) and its own pair lists in one of kMaxPairGroups list SLOTS
// of the handle; ONE launch of the pair search builds the lists of every group that needs new ones (blockIdx.z), and a
// candidate's association block reads the slot of its group (Assoc2Map). A slot whose lists were built with an inflated bound
// stays valid for later calls (the reuse of round 3, now per slot: both poll centres keep theirs).
// (kMaxPairGroups, GroupRef: iba_pair_plan.hpp, shared with the host-only planner)
// what the pair search reads of the problem (the whole DevProblem would not fit the kernel argument segment beside four groups)
struct PairsProblem { const FrameHdr* frames; const float4* pts4; const float* chunk_box; const float2* kp_uv; const uint32_t* coarse_start; };
struct PairsPlan {
    GroupRef g[kMaxPairGroups];                  // the groups whose lists this launch builds (blockIdx.z)
    float rel[kOwnBoundMax][12];                 // the candidates' own motions relative to their group's reference, the rows of a group consecutive (groups of at most kOwnBoundMax candidates: a larger group is bounded entrywise only)
    uint32_t cnt_off[kMaxPairGroups];            // per built group: offset (u32) of the counter set its lists use ...
    uint32_t next_off[kMaxPairGroups];           // ... and of the set cleared for the slot's next build
    uint8_t first[kMaxPairGroups], count[kMaxPairGroups];   // rows of rel[] of the group; count 0: entrywise bound only (a lone candidate, a reusable list)
    uint8_t slot[kMaxPairGroups];                // the list slot the group's lists go to
    uint8_t pad[4];
};
struct PairsArgs { PairsProblem dp; PairsPlan pl; };
static_assert(sizeof(PairsArgs) + 96 <= 4096, "the pair search's arguments must fit the 4 KB kernel argument segment");
// which list slot a candidate's association block reads, and where the slots' current counters are
struct Assoc2Map { uint32_t cnt_off[kMaxPairGroups]; uint8_t slot[kMaxChain]; };
struct K2Args { KArgs k; Assoc2Map m; };   // first argument of iba_assoc2_kernel: read in place through the kernarg segment pointer
struct PairRec { float x, y, z; uint32_t idx; float u, v; uint32_t k, pad; };   // scan point (+ original index), keypoint (+ id): 32 B, streamed
constexpr int kPairsThreads = 512;    // measured at the bench shape: 1024 threads 50 us, 512 threads 39 us, 256 threads 58 us per batch
constexpr int kCountStride = 32;   // u32 per frame (one 128-byte line: the frames' counters do not share a line): pairs, hard points, overflow flag
constexpr int kPairStage = 2048;   // (point, keypoint) hits a block parks in LDS before ONE global reservation writes them out

// grid: (ceil(max P / kPairsThreads), frames); one scan point per thread. The keypoint grid of the frame (coarse CSR + the
// keypoints' (u, v)) sits in LDS, so a thread's walk costs LDS round trips, not L2 ones; the hits of a block are parked in LDS
// and written out behind one atomic reservation per block.
__global__ __launch_bounds__(kPairsThreads) void iba_pairs_kernel(PairsArgs pa_by_value, double max_pixel_dist, uint32_t lds_kuv_off, int n_frames,
                                                                  PairRec* __restrict__ pairs_all, uint32_t* __restrict__ hard_all,
                                                                  uint32_t* __restrict__ counts_all, int pair_cap, int hard_cap,
                                                                  const uint4* __restrict__ head_src, uint4* __restrict__ head_dst, uint32_t head_n16, uint32_t dense_min_pts) {
    extern __shared__ __align__(16) unsigned char smem[];
    if (head_n16 != 0u && blockIdx.z + 1u == gridDim.z) {   // the chain's head: one more z-plane of the grid carries the candidates to the device (see chain_head_copy)
        chain_head_copy(head_src, head_dst, head_n16, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, (uint32_t)kPairsThreads);
        return;
    }
    typedef __attribute__((address_space(4))) const PairsArgs PairsArgsC;
    PairsArgsC* pa = (PairsArgsC*)__builtin_amdgcn_kernarg_segment_ptr();   // (the argument block is read in place: the group is indexed per block, rel[] per lane)
    (void)pa_by_value;
    const PairsProblem dp{pa->dp.frames, pa->dp.pts4, pa->dp.chunk_box, pa->dp.kp_uv, pa->dp.coarse_start};
    const int grp = blockIdx.z;                    // which of the plan's groups this block searches for
    GroupRef cr;
#pragma unroll
    for (int i = 0; i < 9; ++i) { cr.R[i] = pa->pl.g[grp].R[i]; cr.rho[i] = pa->pl.g[grp].rho[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { cr.t[i] = pa->pl.g[grp].t[i]; cr.tau[i] = pa->pl.g[grp].tau[i]; }
    const int B = (int)pa->pl.count[grp], rel0 = (int)pa->pl.first[grp];
    const size_t slot = (size_t)pa->pl.slot[grp];
    PairRec* pairs = pairs_all + slot * (size_t)n_frames * (size_t)pair_cap;
    uint32_t* hard = hard_all + slot * (size_t)n_frames * (size_t)hard_cap;
    uint32_t* counts = counts_all + pa->pl.cnt_off[grp]; uint32_t* counts_next = counts_all + pa->pl.next_off[grp];
    const int f = blockIdx.y;
    const FrameHdr& h = dp.frames[f];
    const uint32_t P = h.P, K = h.K;
    const uint32_t begin = blockIdx.x * (uint32_t)kPairsThreads;
    // the counters of the NEXT call's lists are cleared here (two sets, used in turn: no memset between the kernels of a call)
    if (blockIdx.x == 0 && threadIdx.x < 4) counts_next[(size_t)f * kCountStride + threadIdx.x] = 0u;
    if (begin >= P) return;
    uint2* s_hit = (uint2*)smem;                                   // [kPairStage] (tree position, keypoint)
    uint32_t* s_n = (uint32_t*)(smem + 8u * kPairStage);            // [0] hits parked, [1] base of the block's global reservation, [4..] one flag per wave
    uint16_t* s_cstart = (uint16_t*)(smem + 8u * kPairStage + 128u);
    float2* s_kuv = (float2*)(smem + lds_kuv_off);
    const float4* p4 = dp.pts4 + h.pt_base;
    const float2* guv = dp.kp_uv + h.kp_base;
    uint32_t* cnt = counts + (size_t)f * kCountStride;
    PairRec* out = pairs + (size_t)f * (size_t)pair_cap;
    uint32_t* hout = hard + (size_t)f * (size_t)hard_cap;
    const int gw = (int)h.gw, gh = (int)h.gh, gwc = (int)h.gwc;
    const double fx = h.fx, cx = h.cx, cy = h.cy, W = h.W, H = h.H;
    const uint32_t pos = begin + threadIdx.x;
    // ---- the block's culling chunks (static AABBs of kChunk consecutive tree positions) against every candidate's frustum: a
    //      block of kPairsThreads consecutive tree positions is a compact piece of the scene, and most pieces are seen by no candidate.
    //      For a point p of a box with centre c and half extent e: q_b(p)_i lies within m_i = (|R_0| e)_i + delta_i of q_0(c)_i,
    //      delta from the batch bound at |q_0(c)| + |R_0| e. The piece is invisible when it is behind the camera (z + m_z <= 0) or
    //      wholly beyond one image border: u >= W <=> fx x + (cx - W) z >= 0 (z > 0), u < 0 <=> fx x + cx z < 0, v alike. ----
    constexpr uint32_t kBlkChunks = (uint32_t)kPairsThreads / (uint32_t)kChunk;
    bool chunk_vis = false;
    float4 blo = make_float4(INFINITY, INFINITY, INFINITY, 0.f), bhi = make_float4(-INFINITY, -INFINITY, -INFINITY, 0.f);   // this lane's chunk box (lanes < kBlkChunks)
    if (threadIdx.x < kBlkChunks) {
        const uint32_t ch = begin / (uint32_t)kChunk + threadIdx.x;
        if (ch * (uint32_t)kChunk < P) {
            const float4* bx = (const float4*)(dp.chunk_box + 8 * (h.box_base + ch));
            const float4 lo = bx[0], hi = bx[1];
            blo = lo; bhi = hi;
            const double c3[3] = {0.5 * ((double)lo.x + (double)hi.x), 0.5 * ((double)lo.y + (double)hi.y), 0.5 * ((double)lo.z + (double)hi.z)};
            const double e3[3] = {0.5 * ((double)hi.x - (double)lo.x), 0.5 * ((double)hi.y - (double)lo.y), 0.5 * ((double)hi.z - (double)lo.z)};
            double qc[3], ex[3], m[3];
            for (int i = 0; i < 3; ++i) {
                qc[i] = ((cr.R[i * 3] * c3[0] + cr.R[i * 3 + 1] * c3[1]) + cr.R[i * 3 + 2] * c3[2]) + cr.t[i];
                ex[i] = (fabs(cr.R[i * 3]) * e3[0] + fabs(cr.R[i * 3 + 1]) * e3[1]) + fabs(cr.R[i * 3 + 2]) * e3[2];
            }
            const double a3[3] = {fabs(qc[0]) + ex[0], fabs(qc[1]) + ex[1], fabs(qc[2]) + ex[2]};
            for (int i = 0; i < 3; ++i)
                m[i] = (ex[i] + ((cr.rho[i * 3] * a3[0] + cr.rho[i * 3 + 1] * a3[1]) + cr.rho[i * 3 + 2] * a3[2]) + cr.tau[i]) * (1.0 + 1e-9) + 1e-9 * ((a3[0] + a3[1]) + a3[2]) + 1e-9;
            const double zhi = qc[2] + m[2];
            const bool behind = zhi <= 0.0;
            const bool right = fx * (qc[0] - m[0]) + (cx - W) * zhi >= 1e-6 * (fx * a3[0] + W * a3[2]);
            const bool left = fx * (qc[0] + m[0]) + cx * zhi < -1e-6 * (fx * a3[0] + W * a3[2]);
            const bool below = fx * (qc[1] - m[1]) + (cy - H) * zhi >= 1e-6 * (fx * a3[1] + H * a3[2]);
            const bool above = fx * (qc[1] + m[1]) + cy * zhi < -1e-6 * (fx * a3[1] + H * a3[2]);
            chunk_vis = !(behind || right || left || below || above);   // a NaN box (empty chunk) compares false everywhere: kept, harmless
        }
    }
    // A DENSE scan (the reference's own: 120 k points, 235 blocks per keyframe, three in four of them behind or beside every candidate's
    // camera) decides first and loads afterwards: the point and the keypoint grid of a block that returns below were 20 KB of traffic per
    // block — 0.9 GB per pair search at 200 keyframes x 120 k points (r05: 356 -> 325 us per search there). A sparse scan keeps round 3's order: everything in
    // flight before the first barrier (one exposed round trip less for the blocks that stay, and few blocks to spare).
    const bool dense = P >= dense_min_pts;   // (block-uniform: a property of the keyframe; 32768 points by default)
    float4 pv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!dense && pos < P) pv = p4[pos];
    // the frame's keypoint grid is fetched into registers now and parked in LDS only if some point of the block needs it: a block
    // of consecutive tree positions is a compact piece of the scene, and most pieces lie outside every candidate's image
    const uint32_t* gcs = dp.coarse_start + h.coarse_base;
    const uint32_t ncs = h.gwc * h.ghc + 1u;
    uint32_t cs_r[2] = {0u, 0u}; float2 uv_r[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    if (!dense) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t i = threadIdx.x + (uint32_t)j * kPairsThreads;
            cs_r[j] = i < ncs ? gcs[i] : 0u;
            uv_r[j] = i < K ? guv[i] : make_float2(0.f, 0.f);
        }
    }
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0u;
    if (threadIdx.x < 64) {   // wave 0 holds the chunk tests
        const unsigned long long anyc = __ballot(chunk_vis);
        if (threadIdx.x == 0) s_n[3] = anyc != 0ull ? 1u : 0u;
    }
    __syncthreads();
    if (!s_n[3]) return;   // no candidate sees any point of this block
#if IBA_PAIRS_CUT == 1
    return;
#endif
    if (dense && pos < P) pv = p4[pos];
    // ---- how far the candidates move the points of THIS block (512 consecutive tree positions: a box with centre c and half extent
    //      e in the LiDAR frame, qc = R_0 c + t_0, ex = |R_0| e under the reference): candidate b moves q_0 by M_b q_0 + a_b
    //      (Cand::rel), so |q_b - q_0|_i <= |(M_b qc + a_b)_i| + sum_j |M_b|_ij ex_j for every point of the box. Wave 0: lane b
    //      evaluates candidate b, the maximum per axis goes to LDS (as floats rounded up). The entrywise bound below (rho, tau: every
    //      term maximised over the batch on its own) is about twice as wide for a batch of random perturbations; the smaller of
    //      the two is used. ----
    float* s_delta = (float*)(s_n + 16);
    double wdx = INFINITY, wdy = INFINITY, wdz = INFINITY;
    if (B > 0) {   // (B = 0: a lone candidate — the entrywise bound is zero but for rounding — or IBA_PAIR_BOUND=0)
        if (threadIdx.x < 64) {
            float l[3] = {blo.x, blo.y, blo.z}, u[3] = {bhi.x, bhi.y, bhi.z};   // union of the block's chunk boxes (a NaN box of an empty chunk drops out of fminf / fmaxf)
#pragma unroll
            for (int o = 1; o < (int)kBlkChunks; o <<= 1)
#pragma unroll
                for (int i = 0; i < 3; ++i) { l[i] = fminf(l[i], __shfl_xor(l[i], o)); u[i] = fmaxf(u[i], __shfl_xor(u[i], o)); }
#pragma unroll
            for (int i = 0; i < 3; ++i) { l[i] = __shfl(l[i], 0); u[i] = __shfl(u[i], 0); }
            const double c3[3] = {0.5 * ((double)l[0] + (double)u[0]), 0.5 * ((double)l[1] + (double)u[1]), 0.5 * ((double)l[2] + (double)u[2])};
            const double e3[3] = {0.5 * ((double)u[0] - (double)l[0]), 0.5 * ((double)u[1] - (double)l[1]), 0.5 * ((double)u[2] - (double)l[2])};
            double qc[3], ex[3];
            for (int i = 0; i < 3; ++i) {
                qc[i] = ((cr.R[i * 3] * c3[0] + cr.R[i * 3 + 1] * c3[1]) + cr.R[i * 3 + 2] * c3[2]) + cr.t[i];
                ex[i] = ((fabs(cr.R[i * 3]) * e3[0] + fabs(cr.R[i * 3 + 1]) * e3[1]) + fabs(cr.R[i * 3 + 2]) * e3[2]) * (1.0 + 1e-9) + 1e-9 * ((fabs(c3[0]) + fabs(c3[1])) + fabs(c3[2])) + 1e-12;   // (+ the rounding of qc itself)
            }
            const double scale = (fabs(qc[0]) + fabs(qc[1])) + fabs(qc[2]) + (ex[0] + ex[1]) + ex[2];
            float m[3] = {0.f, 0.f, 0.f};
            const int b = (int)threadIdx.x;
            if (b < B) {
                double rl[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) rl[i] = (double)pa->pl.rel[rel0 + b][i];
                for (int i = 0; i < 3; ++i) {
                    // (the entries are floats of the host's doubles: each is off by at most 2^-24 of itself, which the last term covers)
                    const double lin = (fabs(rl[i * 3]) * (fabs(qc[0]) + ex[0]) + fabs(rl[i * 3 + 1]) * (fabs(qc[1]) + ex[1])) + fabs(rl[i * 3 + 2]) * (fabs(qc[2]) + ex[2]);
                    const double mi = (fabs(((rl[i * 3] * qc[0] + rl[i * 3 + 1] * qc[1]) + rl[i * 3 + 2] * qc[2]) + rl[9 + i]) + ((fabs(rl[i * 3]) * ex[0] + fabs(rl[i * 3 + 1]) * ex[1]) + fabs(rl[i * 3 + 2]) * ex[2])) * (1.0 + 1e-9) + 1e-12 * scale + 1e-13
                                      + 6.1e-8 * (lin + fabs(rl[9 + i]));
                    m[i] = (float)mi * 1.0000002f + 1e-30f;   // >= mi (NaN stays NaN)
                }
            }
#pragma unroll
            for (int o = 1; o < 64; o <<= 1)
#pragma unroll
                for (int i = 0; i < 3; ++i) { const float other = __shfl_xor(m[i], o); m[i] = (m[i] != m[i] || other != other) ? __builtin_nanf("") : fmaxf(m[i], other); }
            if (threadIdx.x == 0) {   // NaN (an empty block's box, a NaN candidate): the entrywise bound alone
                s_delta[0] = m[0] == m[0] ? m[0] : INFINITY; s_delta[1] = m[1] == m[1] ? m[1] : INFINITY; s_delta[2] = m[2] == m[2] ? m[2] : INFINITY;
            }
        }
        __syncthreads();
        wdx = (double)s_delta[0]; wdy = (double)s_delta[1]; wdz = (double)s_delta[2];
    }
#if IBA_PAIRS_CUT == 2
    return;
#endif
    // ---- the point under the reference candidate, the batch's bound on its motion, its search window ----
    int kind = 0;   // 0: nothing to do, 1: walk the grid, 2: hard point
    double u0 = 0, v0 = 0, r = 0;
    if (pos < P) {
        const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
        const double x0 = ((cr.R[0] * x + cr.R[1] * y) + cr.R[2] * z) + cr.t[0];
        const double y0 = ((cr.R[3] * x + cr.R[4] * y) + cr.R[5] * z) + cr.t[1];
        const double z0 = ((cr.R[6] * x + cr.R[7] * y) + cr.R[8] * z) + cr.t[2];
        const double ax = fabs(x0), ay = fabs(y0), az = fabs(z0);
        const double round_off = 1e-13 * ((ax + ay) + az) + 1e-13;   // of q_0 itself and of the candidates' own q_b (a few ulps of |q|)
        const double dx = fmin(((cr.rho[0] * ax + cr.rho[1] * ay) + cr.rho[2] * az) + cr.tau[0], wdx) + round_off;
        const double dy = fmin(((cr.rho[3] * ax + cr.rho[4] * ay) + cr.rho[5] * az) + cr.tau[1], wdy) + round_off;
        const double dz = fmin(((cr.rho[6] * ax + cr.rho[7] * ay) + cr.rho[8] * az) + cr.tau[2], wdz) + round_off;
        if (!(z0 == z0) || z0 < -dz) kind = 0;   // NaN: fails every candidate's own depth test; z0 < -dz: behind every candidate's camera
        else if (!(z0 > 2.0 * dz)) {
            // depth not bounded away from zero: z_b in (0, z0 + dz]. Far enough to the side, every candidate still sees it outside
            const double zmax = z0 + dz;
            const bool out_u = fx * (ax - dx) > zmax * (fmax(cx, W - cx) + 1.0), out_v = fx * (ay - dy) > zmax * (fmax(cy, H - cy) + 1.0);
            kind = (out_u || out_v) ? 0 : 2;
        } else {
            const double den = z0 * (z0 - dz);
            const double Du = fx * (dx * z0 + ax * dz) / den, Dv = fx * (dy * z0 + ay * dz) / den;
            u0 = fx * x0 / z0 + cx; v0 = fx * y0 / z0 + cy;
            if (u0 + Du < -1.0 || u0 - Du >= W + 1.0 || v0 + Dv < -1.0 || v0 - Dv >= H + 1.0) kind = 0;   // inside the image for no candidate
            else {
                r = max_pixel_dist + sqrt(Du * Du + Dv * Dv) + 0.02;
                kind = (r <= 64.0) ? 1 : 2;   // a window wider than that is cheaper as one exact projection per candidate
            }
        }
    }
    if (kind == 2) {
        const uint32_t slot = atomicAdd(&cnt[1], 1u);
        if (slot < (uint32_t)hard_cap) hout[slot] = pos; else cnt[2] = 1u;
    }
    {   // does any point of the block walk the grid? (one flag per wave; hipcc's __syncthreads_or would add static LDS of its own)
        const unsigned long long any = __ballot(kind == 1);
        if ((threadIdx.x & 63) == 0) s_n[4 + (threadIdx.x >> 6)] = any != 0ull ? 1u : 0u;
        __syncthreads();
        uint32_t walk = 0u;
#pragma unroll
        for (int w = 0; w < kPairsThreads / 64; ++w) walk |= s_n[4 + w];
        if (!walk) return;   // no point of this block can meet a keypoint under any candidate
    }
#if IBA_PAIRS_CUT == 3
    return;
#endif
    if (dense) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {   // (the loads of the two passes in flight together)
            const uint32_t i = threadIdx.x + (uint32_t)j * kPairsThreads;
            cs_r[j] = i < ncs ? gcs[i] : 0u;
            uv_r[j] = i < K ? guv[i] : make_float2(0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t i = threadIdx.x + (uint32_t)j * kPairsThreads;
        if (i < ncs) s_cstart[i] = (uint16_t)cs_r[j];
        if (i < K) s_kuv[i] = uv_r[j];
    }
    for (uint32_t i = threadIdx.x + 2u * kPairsThreads; i < ncs; i += kPairsThreads) s_cstart[i] = (uint16_t)gcs[i];
    for (uint32_t i = threadIdx.x + 2u * kPairsThreads; i < K; i += kPairsThreads) s_kuv[i] = guv[i];
    __syncthreads();
#if IBA_PAIRS_CUT == 4
    return;
#endif
    if (kind == 1) {
        // keypoints within r of (u0, v0): the coarse CSR of the keypoint grid (record e of the grid IS keypoint e)
        const int x0c = grid_cell((float)(u0 - r) - 0.01f, gw) >> kCoarseShift, x1c = grid_cell((float)(u0 + r) + 0.01f, gw) >> kCoarseShift;
        const int y0c = grid_cell((float)(v0 - r) - 0.01f, gh) >> kCoarseShift, y1c = grid_cell((float)(v0 + r) + 0.01f, gh) >> kCoarseShift;
        const double r2 = r * r;
        for (int yy = y0c; yy <= y1c; ++yy) {
            const uint32_t e0 = s_cstart[yy * gwc + x0c], e1 = s_cstart[yy * gwc + x1c + 1];
            for (uint32_t e = e0; e < e1; ++e) {
                const float2 kv = s_kuv[e];
                const double du = (double)kv.x - u0, dv = (double)kv.y - v0;
                if (du * du + dv * dv <= r2) {
                    const uint32_t slot = atomicAdd(&s_n[0], 1u);
                    if (slot < (uint32_t)kPairStage) s_hit[slot] = make_uint2(pos, e);
                    else {   // the stage is full (a dense frame): straight to the list
                        const uint32_t g = atomicAdd(&cnt[0], 1u);
                        if (g < (uint32_t)pair_cap) { PairRec pr; pr.x = pv.x; pr.y = pv.y; pr.z = pv.z; pr.idx = __float_as_uint(pv.w); pr.u = kv.x; pr.v = kv.y; pr.k = e; pr.pad = 0u; out[g] = pr; }
                        else cnt[2] = 1u;
                    }
                }
            }
        }
    }
    __syncthreads();
#if IBA_PAIRS_CUT == 5
    return;
#endif
    const uint32_t n = min(s_n[0], (uint32_t)kPairStage);
    if (n == 0u) return;
    if (threadIdx.x == 0) s_n[1] = atomicAdd(&cnt[0], n);
    __syncthreads();
    const uint32_t base = s_n[1];
    for (uint32_t i = threadIdx.x; i < n; i += kPairsThreads) {
        const uint2 hit = s_hit[i];
        const uint32_t g = base + i;
        if (g < (uint32_t)pair_cap) {
            const float4 q = p4[hit.x]; const float2 kv = s_kuv[hit.y];
            PairRec pr; pr.x = q.x; pr.y = q.y; pr.z = q.z; pr.idx = __float_as_uint(q.w); pr.u = kv.x; pr.v = kv.y; pr.k = hit.y; pr.pad = 0u;
            out[g] = pr;
        } else cnt[2] = 1u;
    }
}

// exact association of one scan point against the keypoint grid in GLOBAL memory (hard points, overflow rescans): grid_match
// with the coarse CSR read through L2. PASS 1: ds_min_u64 on the keypoint's best d^2; PASS 2: ties -> lowest original index.
template <int PASS>
__device__ __forceinline__ uint32_t grid_match_g(const FrameCtx& c, const uint32_t* __restrict__ gcs, double u, double v, uint32_t pos) {
    uint32_t hit = 0u;   // (PASS 1) keypoints this point was the first to reach: see grid_match
    const float uf = (float)u, vf = (float)v;
    const int x0 = grid_cell(uf - c.margin, c.gw) >> kCoarseShift, x1 = grid_cell(uf + c.margin, c.gw) >> kCoarseShift;
    const int y0 = grid_cell(vf - c.margin, c.gh) >> kCoarseShift, y1 = grid_cell(vf + c.margin, c.gh) >> kCoarseShift;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t e0 = gcs[yy * c.gwc + x0], e1 = gcs[yy * c.gwc + x1 + 1];
        for (uint32_t e = e0; e < e1; ++e) {
            const float4 rec = c.crec[e];
            if (fabsf(rec.x - uf) > c.margin || fabsf(rec.y - vf) > c.margin) continue;
            const double du = (double)rec.x - u, dv = (double)rec.y - v;
            const double d2 = du * du + dv * dv;
            if (d2 <= c.gate2) {
                const uint32_t k = __float_as_uint(rec.z);
                if (PASS == 1) hit += atomicMin(&c.best_d2[k], d2bits(d2)) == ~0ull ? 1u : 0u;
                else if (c.best_d2[k] == d2bits(d2)) atomicMin(&c.best_idx[k], c.perm[pos]);
            }
        }
    }
    return hit;
}

// iba_assoc2_kernel: one workgroup per (keyframe, candidate), as iba_assoc_kernel, with the first half replaced by the exact
// f64 test of the batch's common pairs (K1 + K2 + K3 of the reference: TransformPointCloud pointcloud.h:82-86, the projection
// and FOV test iba_global.cpp:68-81, the 1-NN within max_pixel_dist :86-95) streamed from the list iba_pairs_kernel left.
// LDS: 16 B per keypoint (best d^2, best index, flags) + the relative poses: ~33 KB at 2000 keypoints.
#ifndef IBA_PAIR_REGS
#define IBA_PAIR_REGS 4
#endif
#ifndef IBA_PAIR_REGS_256
#define IBA_PAIR_REGS_256 6   /* ... of a block of 256 threads (6 x 256 = 1536 pairs) */
#endif
constexpr int kPairRegs = IBA_PAIR_REGS;   // pairs per thread whose d^2 waits in registers for the tie pass (4 x 512 = 2048 pairs; of the others, the possible winners are re-evaluated)
constexpr int kPairNote = 2048;  // possible winners beyond the register window a block can note (u16 pair numbers, 4 KB of LDS) when the scans are dense; 512 otherwise (layout_assoc2)
// BLOCK SIZE (round 5). T = 512 threads was round 1's choice for the per-candidate kernel (two blocks of 80 KB LDS per CU) and this kernel
// inherited it. With a keyframe's ~1.2 k common pairs and ~800 flagged keypoints a block of 256 threads (four waves behind every barrier
// instead of eight, six pairs per thread in registers) takes the association of 64 candidates x 200 keyframes from 0.219 to 0.183 ms; a
// block of 128 falls off the register path of the tail (0.274 ms), 384 is no power of two (0.234 ms). Few blocks (one candidate: 200) are
// faster with 512 threads each, and dense scans (9.8 k pairs per keyframe) do not care: the host picks per launch (assoc2_threads).
template <int Q, bool MANY, int T>   // Q: flagged keypoints per thread that the tail keeps in registers (2: at most 2 T per frame, 4: at most 4 T; 0: any number, read where needed; see assoc_tail); MANY: more than 30 covisible keyframes possible; T: threads per block
__global__ __launch_bounds__(T) void iba_assoc2_kernel(K2Args ka_by_value, const Cand* __restrict__ cands, int B, int want, double* __restrict__ frame_partials, int nrec,
                                                              uint4* __restrict__ flist, uint32_t* __restrict__ fcount,
                                                              uint32_t* __restrict__ lcount, int flist_stride, const PairRec* __restrict__ pairs_all, const uint32_t* __restrict__ hard_all,
                                                              const uint32_t* __restrict__ counts_all, int pair_cap, int hard_cap,
                                                              const uint4* __restrict__ head_src, uint4* __restrict__ head_dst, uint32_t head_n16) {
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int kThreads = T, kPairRegs = T <= 256 ? IBA_PAIR_REGS_256 : IBA_PAIR_REGS;   // (of THIS block)
    typedef __attribute__((address_space(4))) const K2Args K2ArgsC;
    K2ArgsC* ka2 = (K2ArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
    KArgsC* ka = &ka2->k;
    (void)ka_by_value;
    {   // the chain's head rides in the blocks behind the (frame, candidate) grid (a call that reuses earlier pair lists has no pair search to carry it)
        const uint32_t main_blocks = 8u * (uint32_t)((ka->dp.n_frames + 7) / 8) * (uint32_t)B;
        if (blockIdx.x >= main_blocks) {
            // (round 6) head_n16 with bit 31 set: the spare blocks carry the DERIVATIVE half of the candidates instead — words [w0, w1) of every Cand, B | w0 << 16 | w1 << 24 —
            // for the factor kernel at the end of this chain (the pair search in front of this kernel has carried the values; rounds 3-5: iba_fetch_jets_kernel, a launch of
            // its own in stream order, 5 us)
            if (head_n16 >> 31) {
                const uint32_t nb = head_n16 & 0xFFFFu, w0 = (head_n16 >> 16) & 0x7Fu, w1 = (head_n16 >> 24) & 0x7Fu, per = w1 - w0, n = nb * per;
                for (uint32_t i = (blockIdx.x - main_blocks) * (uint32_t)kThreads + threadIdx.x; i < n; i += (gridDim.x - main_blocks) * (uint32_t)kThreads) {
                    const uint32_t w = (i / per) * w1 + w0 + i % per;
                    head_dst[w] = head_src[w];
                }
            } else chain_head_copy(head_src, head_dst, head_n16, blockIdx.x - main_blocks, gridDim.x - main_blocks, (uint32_t)kThreads);
            return;
        }
    }
#define dp (ka->dp)
#define prm (ka->prm)
#define lay (ka->lay)
    const int tid = threadIdx.x;
    const int nf = dp.n_frames;
    const int per_xcd = (nf + 7) / 8;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int f = xcd + 8 * (jj / B), b = jj % B;
    if (f >= nf || jj / B >= per_xcd) return;
    const FrameHdr& h = dp.frames[f];
    const Cand& cd = cands[b];
    double* part = frame_partials + ((size_t)b * nrec + f) * kPartialStride;
    // the lists of this candidate's group: one of the handle's list slots
    const uint32_t lslot = (uint32_t)ka2->m.slot[b];
    const PairRec* pairs = pairs_all + (size_t)lslot * (size_t)nf * (size_t)pair_cap;
    const uint32_t* hard = hard_all + (size_t)lslot * (size_t)nf * (size_t)hard_cap;
    const uint32_t* counts = counts_all + ka2->m.cnt_off[lslot];

    unsigned long long* s_best_d2 = (unsigned long long*)(smem + lay.off_best_d2);
    uint32_t* s_best_idx = (uint32_t*)(smem + lay.off_best_idx);
    double* s_red = (double*)(smem + lay.off_red);
    double* s_rel = s_red + (IBA_THREADS / 64) * 4;   // (the reduction slab is laid out for the largest block: layout_assoc2)
    uint32_t* s_list = (uint32_t*)(smem + lay.off_best_d2);   // aliases best_d2 once the winners are known
    uint32_t* s_qn = (uint32_t*)(s_rel + lay.rel_slots * 12u) + IBA_THREADS / 64;   // [0]: pairs noted for the tie pass
    uint16_t* s_q = (uint16_t*)(smem + lay.off_pair);                // their numbers (pair lists hold at most 65 536 records)
    const uint32_t K = h.K, P = h.P;
    const uint32_t ut = (uint32_t)tid;

    // ---- the counts of this frame's common lists, the first pairs of this thread, the tables ----
    const uint32_t* cnt = counts + (size_t)f * kCountStride;
    uint4 rfa = make_uint4(kNone, 0u, kNone, 0u), rfb = rfa;   // Q > 0: this thread's Q entries of the frame's flagged-keypoint list (keypoint, flag word)
    uint32_t first = 0u;   // keypoints this thread is the first to reach (their block sum is corrset.size(): see grid_match)
    const bool overflow = cnt[2] != 0u;   // a list did not hold everything: every point again, exactly (speed only)
    const uint32_t npair = overflow ? 0u : min(cnt[0], (uint32_t)pair_cap), nhard = overflow ? 0u : min(cnt[1], (uint32_t)hard_cap);
    const float4* prq = (const float4*)(pairs + (size_t)f * (size_t)pair_cap);   // two 16-byte halves per record
    {
        double rv = 0.0;
        if (ut < h.n_slots * 12u) rv = dp.slots[h.slot_base + ut / 12].rel[ut % 12];
        if (Q > 0) {   // this thread's list entries: in flight until the tail
            const uint2* fk = dp.fkp + h.fk_base;
            const uint32_t e0 = ut * (uint32_t)Q;
            if (e0 < h.n_fk) rfa = *(const uint4*)(fk + e0);                    // (the lists are padded to four entries)
            if (Q > 2 && e0 + 2u < h.n_fk) rfb = *(const uint4*)(fk + e0 + 2u);
        }
        for (uint32_t i = ut; i < K; i += kThreads) { s_best_d2[i] = ~0ull; s_best_idx[i] = kNone; }
        if (ut < h.n_slots * 12u) s_rel[ut] = rv;
        for (uint32_t i = ut + kThreads; i < h.n_slots * 12u; i += kThreads) s_rel[i] = dp.slots[h.slot_base + i / 12u].rel[i % 12u];   // more than 42 covisible keyframes: beyond one store per thread
        if (tid == 0) s_qn[0] = 0u;
    }
    const int dbg = want >> 8;
    const bool refit = (want & 4) != 0;
    FrameCtx c;
    c.xs = dp.xs + h.pt_base; c.ys = dp.ys + h.pt_base; c.zs = dp.zs + h.pt_base;
    c.nodes = nullptr; c.bitmap = nullptr; c.best_d2 = s_best_d2; c.best_idx = s_best_idx;
    c.cstart = nullptr; c.gwc = (int)h.gwc; c.crec = dp.crec + h.kp_base;
    c.perm = dp.perm + h.pt_base;
    c.p4 = dp.pts4 + h.pt_base;
    c.gw = (int)h.gw; c.gh = (int)h.gh; c.margin = (float)prm.grid_margin; c.gate2 = prm.gate2;
    c.fx = h.fx; c.cx = h.cx; c.cy = h.cy; c.W = h.W; c.H = h.H;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = cd.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = cd.t[i];
    const uint32_t* gcs = dp.coarse_start + h.coarse_base;
    __syncthreads();
    if (dbg == 1) return;

    // ---- pass 1: exact f64 projection + FOV test and exact d^2 of every common pair, ds_min_u64 on the keypoint's best ----
    auto eval_loaded = [&](const float4& a, const float4& q, uint32_t& k, uint32_t& idx) -> unsigned long long {
        double u, v;
        k = kNone; idx = __float_as_uint(a.w);
        if (project_uv(c, a.x, a.y, a.z, u, v)) {
            const double du = (double)q.x - u, dv = (double)q.y - v;
            const double d2 = du * du + dv * dv;
            if (d2 <= c.gate2) { k = __float_as_uint(q.z); return d2bits(d2); }
        }
        return 0ull;
    };
    auto eval_pair = [&](uint32_t i, uint32_t& k, uint32_t& idx) -> unsigned long long {
        const float4 a = prq[2 * (size_t)i], q = prq[2 * (size_t)i + 1];
        return eval_loaded(a, q, k, idx);
    };
    // The pairs are streamed ONE AHEAD (r04): the record of the thread's next pair is in flight while this one is evaluated. (As written
    // in round 3 the compiler issued load, wait, evaluate, load, wait, ... per pair — and, beyond the register window, the record's
    // second half only after the projection had passed: three exposed round trips per block at the bench shape, two per pair and
    // nineteen pairs per thread at 120 k points.)
    float4 na = make_float4(0.f, 0.f, 0.f, 0.f), nq = na;
    if (ut < npair) { na = prq[2 * (size_t)ut]; nq = prq[2 * (size_t)ut + 1]; }
    uint32_t rk[kPairRegs], ri[kPairRegs]; unsigned long long rb[kPairRegs];
#pragma unroll
    for (int j = 0; j < kPairRegs; ++j) {
        rk[j] = kNone; ri[j] = 0u; rb[j] = 0ull;
        const uint32_t i = ut + (uint32_t)j * kThreads;
        const float4 a = na, q = nq;
        if (i + (uint32_t)kThreads < npair) { na = prq[2 * (size_t)(i + kThreads)]; nq = prq[2 * (size_t)(i + kThreads) + 1]; }
        if (i < npair) { rb[j] = eval_loaded(a, q, rk[j], ri[j]); if (rk[j] != kNone) first += atomicMin(&s_best_d2[rk[j]], rb[j]) == ~0ull ? 1u : 0u; }
    }
    // pairs beyond the register window (a dense scan: 9 k pairs per keyframe at 120 k points): one that is at most the keypoint's best
    // so far MAY be the winner and is noted for the tie pass (a keypoint sees ~1.3 such pairs); the others cannot win any more
    {
        const unsigned long long lt = (1ull << (tid & 63)) - 1ull;
        for (uint32_t i0 = (uint32_t)kPairRegs * kThreads; i0 < npair; i0 += kThreads) {   // block-uniform trip count
            const uint32_t i = i0 + ut;
            const float4 a = na, q = nq;
            if (i + (uint32_t)kThreads < npair) { na = prq[2 * (size_t)(i + kThreads)]; nq = prq[2 * (size_t)(i + kThreads) + 1]; }
            bool note = false;
            if (i < npair) {
                uint32_t k, idx;
                const unsigned long long bits = eval_loaded(a, q, k, idx);
                if (k != kNone) { const unsigned long long was = atomicMin(&s_best_d2[k], bits); note = bits <= was; first += was == ~0ull ? 1u : 0u; }
            }
            const unsigned long long bal = __ballot(note);
            if (bal != 0ull) {
                uint32_t base = 0u;
                if ((tid & 63) == 0) base = atomicAdd(&s_qn[0], (uint32_t)__popcll(bal));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                const uint32_t slot = base + (uint32_t)__popcll(bal & lt);
                if (note && slot < lay.pair_cap) s_q[slot] = (uint16_t)i;
            }
        }
    }
    for (uint32_t i = ut; i < nhard; i += kThreads) {
        const uint32_t pos = hard[(size_t)f * (size_t)hard_cap + i];
        double u, v;
        if (project_pos<true>(c, pos, u, v)) first += grid_match_g<1>(c, gcs, u, v, pos);
    }
    IBA_DIAG_COUNT(overflow && tid == 0, 0);
    if (overflow)
        for (uint32_t pos = ut; pos < P; pos += kThreads) { double u, v; if (project_pos<true>(c, pos, u, v)) first += grid_match_g<1>(c, gcs, u, v, pos); }
    __syncthreads();
    if (dbg == 4) return;
    // ---- pass 2: the winner of each keypoint records its original index; exact ties -> lowest index ----
#pragma unroll
    for (int j = 0; j < kPairRegs; ++j)
        if (rk[j] != kNone && s_best_d2[rk[j]] == rb[j]) atomicMin(&s_best_idx[rk[j]], ri[j]);
    {
        const uint32_t nq = s_qn[0];
        if (nq <= lay.pair_cap) {   // the noted pairs alone (lay.pair_cap: the note list's capacity, layout_assoc2)
            for (uint32_t t = ut; t < nq; t += kThreads) {
                uint32_t k, idx;
                const unsigned long long bits = eval_pair((uint32_t)s_q[t], k, idx);
                if (k != kNone && s_best_d2[k] == bits) atomicMin(&s_best_idx[k], idx);
            }
        } else {   // more than the note list holds: every pair beyond the window again
            IBA_DIAG_COUNT(tid == 0, 1);
            for (uint32_t i = ut + (uint32_t)kPairRegs * kThreads; i < npair; i += kThreads) {
                uint32_t k, idx;
                const unsigned long long bits = eval_pair(i, k, idx);
                if (k != kNone && s_best_d2[k] == bits) atomicMin(&s_best_idx[k], idx);
            }
        }
    }
    for (uint32_t i = ut; i < nhard; i += kThreads) {
        const uint32_t pos = hard[(size_t)f * (size_t)hard_cap + i];
        double u, v;
        if (project_pos<true>(c, pos, u, v)) grid_match_g<2>(c, gcs, u, v, pos);
    }
    if (overflow)
        for (uint32_t pos = ut; pos < P; pos += kThreads) { double u, v; if (project_pos<true>(c, pos, u, v)) grid_match_g<2>(c, gcs, u, v, pos); }
    __syncthreads();
    if (dbg == 5) return;
    assoc_tail<Q, MANY, T>(ka, h, cd, c, s_best_idx, rfa, rfb, s_list, s_red, s_rel, K, want, dbg, refit, b, f, nf, part, flist, fcount, lcount, flist_stride, first);
#undef dp
#undef prm
#undef lay
}

// ---- one lane's exact 1-NN search of a query pair in the implicit balanced kd-tree: a = the association path's MapPoint query,
//      c = the cost path's (the same MapPoint through different float/double islands of the reference, ~1e-7 apart). The walk is
//      float32 and conservative around the steering point o (|q - o| <= del per axis for both queries: for a node with split s and
//      d = fl(o - s), max(|d| - del', 0)^2 (1 - 2^-19) is a lower bound of the squared plane distance of either query); a leaf is
//      scanned in float and only the arg-min is confirmed in double when the runner-up is separated by more than the error bound
//      E(u), otherwise every point within the bound. The visited path stays in registers from one leaf to the next. ----
// The state of a lane's search is a set of plain variables (a struct would be kept in scratch: the compiler turns the selects
// on neighbouring members into indexed loads): declare them with IBA_LANE_NN_DECL, hand them over with IBA_LANE_NN_PASS.
//   ax..qz: the queries (set by the caller together with actA / actC); go < 0 after a visit: the search has ended
#define IBA_LANE_NN_DECL \
    double ax = NAN, ay = NAN, az = NAN, qx = NAN, qy = NAN, qz = NAN, bestA = INFINITY, bestC = INFINITY; uint32_t bposA = kNone, bposC = kNone; \
    float o0 = 0.f, o1 = 0.f, o2 = 0.f, delc = 0.f, e_lin = 0.f, e_const = 0.f; float pd2[kPathMax]; \
    uint32_t side = 0u, done = 0u, node = 0u; int go = -1; bool actA = false, actC = false; \
    _Pragma("unroll") for (int L_ = 0; L_ < kPathMax; ++L_) pd2[L_] = INFINITY
#define IBA_LANE_NN_PARAMS \
    double& ax, double& ay, double& az, double& qx, double& qy, double& qz, double& bestA, double& bestC, uint32_t& bposA, uint32_t& bposC, \
    float& o0, float& o1, float& o2, float& delc, float& e_lin, float& e_const, float (&pd2)[kPathMax], uint32_t& side, uint32_t& done, uint32_t& node, int& go, \
    bool& actA, bool& actC
#define IBA_LANE_NN_PASS ax, ay, az, qx, qy, qz, bestA, bestC, bposA, bposC, o0, o1, o2, delc, e_lin, e_const, pd2, side, done, node, go, actA, actC
// The next level whose far child a search enters (-1: the search has ended). The cell of the far child at level L lies beyond the
// splitting plane of L AND beyond the planes of every level above it at which the current path has already turned to the far side:
// per coordinate the largest of those plane distances, summed over the coordinates, is a lower bound of the squared distance from the
// query to the cell (nanoflann's `dists` vector, nanoflann.hpp searchLevel). Rounds 2-3 tested the plane of L alone — correct, but a
// query far from the scan (a MapPoint 2 m in front of a wall, a candidate far from the truth) then opens every cell within reach of
// ONE plane at a time: 24 leaves and more. pd2[] are float lower bounds of the plane distances; their sum keeps a margin of 1e-6.
__device__ __forceinline__ int nn_next_level(const float (&pd2)[kPathMax], const uint32_t side, uint32_t& done, const uint32_t node, const uint32_t D, const float bestf, const TreeNode* s_nodes) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;   // per coordinate: the largest plane distance^2 among the far turns of the path so far
    uint32_t cnd = 0u;
#pragma unroll
    for (int L = 0; L < kPathMax; ++L) {
        if (L >= (int)D) break;
        const uint32_t dm = s_nodes[((node + 1u) >> (D - (uint32_t)L)) - 1u].dim;
        const float p = pd2[L];
        const float w0 = dm == 0u ? fmaxf(a0, p) : a0, w1 = dm == 1u ? fmaxf(a1, p) : a1, w2 = dm == 2u ? fmaxf(a2, p) : a2;
        cnd |= (((w0 + w1) + w2) * 0.999999f <= bestf ? 1u : 0u) << L;
        if ((side >> (16 + L)) & 1u) { a0 = w0; a1 = w1; a2 = w2; }
    }
    cnd &= ~done & ((1u << D) - 1u);
    done |= ~cnd;
    return cnd ? 31 - __clz((int)cnd) : -1;
}
__device__ __forceinline__ void lane_nn_begin(IBA_LANE_NN_PARAMS) {
    // float steering point o and the radius del >= |q - o| per axis of both queries
    o0 = (float)(actC ? qx : ax); o1 = (float)(actC ? qy : ay); o2 = (float)(actC ? qz : az);
    double m = 0.0;
    if (actA) m = fmax(fmax(fabs(ax - (double)o0), fabs(ay - (double)o1)), fabs(az - (double)o2));
    if (actC) m = fmax(m, fmax(fmax(fabs(qx - (double)o0), fabs(qy - (double)o1)), fabs(qz - (double)o2)));
    const float del = (float)m * 1.00001f + 1e-30f;
    delc = del * 0.999999f;
    e_lin = 2.01f * del; e_const = 3.01f * del * del + 2.1e-19f * e_lin + 1e-37f;
    bestA = INFINITY; bestC = INFINITY; bposA = kNone; bposC = kNone;
    side = 0u; done = 0u; node = 0u; go = -1;
}
// one leaf visit
template <int WHICH>
__device__ __forceinline__ void lane_nn_visit(IBA_LANE_NN_PARAMS, const TreeNode* s_nodes, const float4* __restrict__ p4, const uint32_t* __restrict__ perm_g, uint32_t P, uint32_t D) {
    const uint32_t first_leaf = (1u << D) - 1u;
    auto lower_bound = [delc](float d) { const float a = fmaxf(fmaf(fabsf(d), 0.999999f, -delc), 0.f); return a * a; };
    int start = 0;
    if (go >= 0) {   // enter the far child at level go
        const uint32_t anc = ((node + 1u) >> (D - (uint32_t)go)) - 1u;
        done |= 1u << go; side ^= 1u << go; side |= 0x10000u << go;   // (bits 16..: the levels at which the current path has turned to the FAR side)
        node = 2u * anc + 1u + ((side >> go) & 1u);
        start = go + 1;
    }
    {
        const uint32_t keep = (1u << start) - 1u;
        side &= keep | (keep << 16); done &= keep;
        uint32_t n1 = node + 1u;
#pragma unroll
        for (int L = 0; L < kPathMax; ++L) {
            if (L >= (int)D) break;
            if (L >= start) {
                const TreeNode n = s_nodes[n1 - 1u];
                const float d = (n.dim == 0 ? o0 : (n.dim == 1 ? o1 : o2)) - n.split;
                const uint32_t r = (~__float_as_uint(d)) >> 31;
                pd2[L] = lower_bound(d);
                side |= r << L;
                n1 = (n1 << 1) | r;
            }
        }
        node = n1 - 1u;
    }
    {
        const uint32_t j = node - first_leaf;
        const uint32_t lo = (uint32_t)(((uint64_t)j * P) >> D), hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
        float m1 = INFINITY, m2 = INFINITY; uint32_t mi = kNone;
        auto err_of = [e_lin, e_const](float u) { return fmaf(1.001f * e_lin, __builtin_amdgcn_sqrtf(3.f * u), fmaf(1.5e-6f, u, e_const)); };
        constexpr int kLeafBatch = IBA_NN_LEAF_BATCH;
        for (uint32_t i0 = lo; i0 < hi; i0 += (uint32_t)kLeafBatch) {
            float X[kLeafBatch], Y[kLeafBatch], Z[kLeafBatch];
#pragma unroll
            for (int u = 0; u < kLeafBatch; ++u) {
                const uint32_t iu = i0 + (uint32_t)u, ic = iu < hi ? iu : hi - 1u;
                const float4 v = p4[ic]; X[u] = v.x; Y[u] = v.y; Z[u] = v.z;
            }
#pragma unroll
            for (int u = 0; u < kLeafBatch; ++u) {
                const uint32_t i = i0 + (uint32_t)u;
                const float dx = o0 - X[u], dy = o1 - Y[u], dz = o2 - Z[u];
                float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                uu = i < hi ? uu : INFINITY;
                mi = uu < m1 ? i : mi;
                m2 = __builtin_amdgcn_fmed3f(m1, m2, uu);
                m1 = vmin(m1, uu);
            }
        }
        const float mono = 4.f * e_lin * e_lin;
        const float thi = m1 + err_of(m1);
        bool single = m2 >= mono && m2 - err_of(m2) > thi;
        const float bnear = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
        const float bmax = fmaf(fabsf(bnear), 1.2e-7f, bnear);
        const bool skip = m1 >= mono && m1 - err_of(m1) > bmax;
#ifdef IBA_LEAF_FORCE_SINGLE
        single = true;
#endif
        if (mi != kNone && !skip) {
            uint32_t i = single ? mi : lo;
            while (single || i < hi) {
                const float4 pv = p4[i];
                bool take = single;
                if (!single) {
                    const float dx = o0 - pv.x, dy = o1 - pv.y, dz = o2 - pv.z;
                    const float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    take = uu - err_of(uu) <= thi;
                }
                if (take) {
                    const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
                    if (WHICH & 1) {
                        const double dx = ax - x, dy = ay - y, dz = az - z;
                        nn_merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, i, perm_g);
                    }
                    if (WHICH & 2) {
                        const double dx = qx - x, dy = qy - y, dz = qz - z;
                        nn_merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, i, perm_g);
                    }
                }
                if (single) break;
                ++i;
            }
        }
    }
    {   // deepest level whose far side may still be within reach of either query
        const float bestf = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
        go = nn_next_level(pd2, side, done, node, D, bestf, s_nodes);
    }
}

// ---- the same search run by a GROUP of G lanes (8 .. 64, a whole wave) for one query pair: every lane carries the same state and walks the same path, the points
//      of a leaf are spread over the lanes (one coalesced load instead of a lane's 59 dependent gathers at 120 k points per scan) and
//      the leaf's minimum, runner-up and arg-min are combined across the wave exactly as the sequential scan defines them (arg-min =
//      lowest position among the minima, runner-up = second smallest of the multiset). The exact confirmations merge through
//      nn_merge, whose outcome (least d^2, ties to the lowest original index) does not depend on the order. For the handful of
//      entries per block that the anchored lists leave over: their searches, not their number, set the block's run time. ----
// (minimum over the G consecutive lanes of a group, G a power of two: a lane's partners are in its own group, which branches as one)
__device__ __forceinline__ float group_min_f32(float v, const int G) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) if (o < G) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v, const int G) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) if (o < G) v = min(v, (uint32_t)__shfl_xor((int)v, o));
    return v;
}
__device__ __forceinline__ double group_min_f64(double v, const int G) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) if (o < G) v = fmin(v, __shfl_xor(v, o));
    return v;
}
template <int WHICH>
__device__ __forceinline__ void wave_nn_visit(IBA_LANE_NN_PARAMS, const int lane, const int G, const TreeNode* s_nodes, const float4* __restrict__ p4, const uint32_t* __restrict__ perm_g, uint32_t P, uint32_t D) {
    const uint32_t first_leaf = (1u << D) - 1u;
    auto lower_bound = [delc](float d) { const float a = fmaxf(fmaf(fabsf(d), 0.999999f, -delc), 0.f); return a * a; };
    int start = 0;
    if (go >= 0) {   // enter the far child at level go
        const uint32_t anc = ((node + 1u) >> (D - (uint32_t)go)) - 1u;
        done |= 1u << go; side ^= 1u << go; side |= 0x10000u << go;   // (bits 16..: the levels at which the current path has turned to the FAR side)
        node = 2u * anc + 1u + ((side >> go) & 1u);
        start = go + 1;
    }
    {
        const uint32_t keep = (1u << start) - 1u;
        side &= keep | (keep << 16); done &= keep;
        uint32_t n1 = node + 1u;
#pragma unroll
        for (int L = 0; L < kPathMax; ++L) {
            if (L >= (int)D) break;
            if (L >= start) {
                const TreeNode n = s_nodes[n1 - 1u];
                const float d = (n.dim == 0 ? o0 : (n.dim == 1 ? o1 : o2)) - n.split;
                const uint32_t r = (~__float_as_uint(d)) >> 31;
                pd2[L] = lower_bound(d);
                side |= r << L;
                n1 = (n1 << 1) | r;
            }
        }
        node = n1 - 1u;
    }
    {
        const uint32_t j = node - first_leaf;
        const uint32_t lo = (uint32_t)(((uint64_t)j * P) >> D), hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
        auto err_of = [e_lin, e_const](float u) { return fmaf(1.001f * e_lin, __builtin_amdgcn_sqrtf(3.f * u), fmaf(1.5e-6f, u, e_const)); };
        // this lane's share of the leaf, then the wave's (m1, m2, mi) of the whole leaf
        float l1 = INFINITY, l2 = INFINITY; uint32_t li = kNone;
        for (uint32_t i0 = lo + (uint32_t)lane; i0 < hi; i0 += 4u * (uint32_t)G) {   // four loads in flight per lane (a small group scans 15 points per lane)
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const uint32_t iu = i0 + (uint32_t)u * (uint32_t)G; v[u] = p4[iu < hi ? iu : hi - 1u]; }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + (uint32_t)u * (uint32_t)G;
                const float dx = o0 - v[u].x, dy = o1 - v[u].y, dz = o2 - v[u].z;
                float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                uu = i < hi ? uu : INFINITY;
                li = uu < l1 ? i : li;
                l2 = __builtin_amdgcn_fmed3f(l1, l2, uu);
                l1 = vmin(l1, uu);
            }
        }
        const float m1 = group_min_f32(l1, G);
        const uint32_t mi = group_min_u32(l1 == m1 ? li : kNone, G);
        const float m2 = group_min_f32((li == mi && mi != kNone) ? l2 : l1, G);
        const float mono = 4.f * e_lin * e_lin;
        const float thi = m1 + err_of(m1);
        bool single = m2 >= mono && m2 - err_of(m2) > thi;
        const float bnear = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
        const float bmax = fmaf(fabsf(bnear), 1.2e-7f, bnear);
        const bool skip = m1 >= mono && m1 - err_of(m1) > bmax;
#ifdef IBA_LEAF_FORCE_SINGLE
        single = true;
#endif
        if (mi != kNone && !skip) {
            if (single) {
                const float4 pv = p4[mi];
                const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
                if (WHICH & 1) { const double dx = ax - x, dy = ay - y, dz = az - z; nn_merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, mi, perm_g); }
                if (WHICH & 2) { const double dx = qx - x, dy = qy - y, dz = qz - z; nn_merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, mi, perm_g); }
            } else {
                double lbA = INFINITY, lbC = INFINITY; uint32_t lpA = kNone, lpC = kNone;
                for (uint32_t i0 = lo + (uint32_t)lane; i0 < hi; i0 += 4u * (uint32_t)G) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const uint32_t iu = i0 + (uint32_t)u * (uint32_t)G; v[u] = p4[iu < hi ? iu : hi - 1u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = i0 + (uint32_t)u * (uint32_t)G;
                        const float4 pv = v[u];
                        const float dx = o0 - pv.x, dy = o1 - pv.y, dz = o2 - pv.z;
                        const float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (i < hi && uu - err_of(uu) <= thi) {
                            const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
                            if (WHICH & 1) { const double ex = ax - x, ey = ay - y, ez = az - z; nn_merge(lbA, lpA, (ex * ex + ey * ey) + ez * ez, i, perm_g); }
                            if (WHICH & 2) { const double ex = qx - x, ey = qy - y, ez = qz - z; nn_merge(lbC, lpC, (ex * ex + ey * ey) + ez * ez, i, perm_g); }
                        }
                    }
                }
                if (WHICH & 1) {
                    const double g = group_min_f64(lbA, G);
                    const uint32_t key = group_min_u32((lbA == g && lpA != kNone) ? perm_g[lpA] : kNone, G);
                    const uint32_t pw = group_min_u32((lbA == g && lpA != kNone && perm_g[lpA] == key) ? lpA : kNone, G);
                    if (pw != kNone) nn_merge(bestA, bposA, g, pw, perm_g);
                }
                if (WHICH & 2) {
                    const double g = group_min_f64(lbC, G);
                    const uint32_t key = group_min_u32((lbC == g && lpC != kNone) ? perm_g[lpC] : kNone, G);
                    const uint32_t pw = group_min_u32((lbC == g && lpC != kNone && perm_g[lpC] == key) ? lpC : kNone, G);
                    if (pw != kNone) nn_merge(bestC, bposC, g, pw, perm_g);
                }
            }
        }
    }
    {   // deepest level whose far side may still be within reach of either query
        const float bestf = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
        go = nn_next_level(pd2, side, done, node, D, bestf, s_nodes);
    }
}

// ---- ROUNDS of leaves (round 5). A left-over entry starts from a good bound (the nearest LISTED point: seed), so its search seldom
//      improves the bound — it has to LOOK into every leaf the ball of that radius reaches (24 and more for a MapPoint 2 m in front
//      of a wall at 120 k points per scan), and wave_nn_visit pays three dependent round trips per leaf (the leaf's points, the exact
//      confirmation, the walk to the next leaf). Here the group first walks on, WITHOUT looking, to the next leaves the current bound
//      reaches (LDS only: the nodes), then scans them together — the loads of two leaves in flight at a time — and confirms once. A
//      stale bound can only add leaves; the result (least d^2, ties to the lowest original index) does not depend on the leaves'
//      order or on looking into one too many. s_leaf: round_cap <= kRoundLeaves words of LDS of this group. ----
constexpr int kRoundLeaves = IBA_NN_ROUND_LEAVES;
template <int WHICH>
__device__ __forceinline__ void wave_nn_round(IBA_LANE_NN_PARAMS, const int lane, const int G, const TreeNode* s_nodes, const float4* __restrict__ p4, const uint32_t* __restrict__ perm_g, uint32_t P, uint32_t D,
                                              uint32_t* s_leaf, const int round_cap) {
    const uint32_t first_leaf = (1u << D) - 1u;
    auto lower_bound = [delc](float d) { const float a = fmaxf(fmaf(fabsf(d), 0.999999f, -delc), 0.f); return a * a; };
    int nl = 0;
    {
        const float bestf = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
        do {
            int start = 0;
            if (go >= 0) {   // enter the far child at level go
                const uint32_t anc = ((node + 1u) >> (D - (uint32_t)go)) - 1u;
                done |= 1u << go; side ^= 1u << go; side |= 0x10000u << go;
                node = 2u * anc + 1u + ((side >> go) & 1u);
                start = go + 1;
            }
            const uint32_t keep = (1u << start) - 1u;
            side &= keep | (keep << 16); done &= keep;
            uint32_t n1 = node + 1u;
#pragma unroll
            for (int L = 0; L < kPathMax; ++L) {
                if (L >= (int)D) break;
                if (L >= start) {
                    const TreeNode n = s_nodes[n1 - 1u];
                    const float d = (n.dim == 0 ? o0 : (n.dim == 1 ? o1 : o2)) - n.split;
                    const uint32_t r = (~__float_as_uint(d)) >> 31;
                    pd2[L] = lower_bound(d);
                    side |= r << L;
                    n1 = (n1 << 1) | r;
                }
            }
            node = n1 - 1u;
            s_leaf[nl++] = node - first_leaf;   // (every lane of the group writes the same word)
            go = nn_next_level(pd2, side, done, node, D, bestf, s_nodes);
        } while (go >= 0 && nl < round_cap);
    }
    auto err_of = [e_lin, e_const](float u) { return fmaf(1.001f * e_lin, __builtin_amdgcn_sqrtf(3.f * u), fmaf(1.5e-6f, u, e_const)); };
    auto range_of = [P, D](uint32_t j, uint32_t& lo, uint32_t& hi) { lo = (uint32_t)(((uint64_t)j * P) >> D); hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D); };
    // this lane's share of the round's leaves, two leaves at a time, then the group's (m1, m2, mi) of all of them
    float l1 = INFINITY, l2 = INFINITY; uint32_t li = kNone;
    for (int m = 0; m < nl; m += 2) {
        uint32_t loa, hia, lob, hib;
        range_of(s_leaf[m], loa, hia);
        range_of(s_leaf[m + 1 < nl ? m + 1 : m], lob, hib);
        if (m + 1 >= nl) hib = lob;   // an odd last leaf: no partner
        const uint32_t len = max(hia - loa, hib - lob);
        for (uint32_t k0 = (uint32_t)lane; k0 < len; k0 += 4u * (uint32_t)G) {
            float4 va[4], vb[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + (uint32_t)u * (uint32_t)G;
                va[u] = p4[loa + k < hia ? loa + k : hia - 1u];
                vb[u] = p4[lob + k < hib ? lob + k : lob];   // (an absent partner reads the leaf's own first point and is masked)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t k = k0 + (uint32_t)u * (uint32_t)G;
                {
                    const float dx = o0 - va[u].x, dy = o1 - va[u].y, dz = o2 - va[u].z;
                    float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    uu = loa + k < hia ? uu : INFINITY;
                    li = uu < l1 ? loa + k : li;
                    l2 = __builtin_amdgcn_fmed3f(l1, l2, uu);
                    l1 = vmin(l1, uu);
                }
                {
                    const float dx = o0 - vb[u].x, dy = o1 - vb[u].y, dz = o2 - vb[u].z;
                    float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    uu = lob + k < hib ? uu : INFINITY;
                    li = uu < l1 ? lob + k : li;
                    l2 = __builtin_amdgcn_fmed3f(l1, l2, uu);
                    l1 = vmin(l1, uu);
                }
            }
        }
    }
    const float m1 = group_min_f32(l1, G);
    const uint32_t mi = group_min_u32(l1 == m1 ? li : kNone, G);
    const float m2 = group_min_f32((li == mi && mi != kNone) ? l2 : l1, G);
    const float mono = 4.f * e_lin * e_lin;
    const float thi = m1 + err_of(m1);
    const bool single = m2 >= mono && m2 - err_of(m2) > thi;
    const float bnear = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
    const float bmax = fmaf(fabsf(bnear), 1.2e-7f, bnear);
    const bool skip = m1 >= mono && m1 - err_of(m1) > bmax;
    if (mi != kNone && !skip) {
        if (single) {
            const float4 pv = p4[mi];
            const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
            if (WHICH & 1) { const double dx = ax - x, dy = ay - y, dz = az - z; nn_merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, mi, perm_g); }
            if (WHICH & 2) { const double dx = qx - x, dy = qy - y, dz = qz - z; nn_merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, mi, perm_g); }
        } else {   // several points within the float scan's error of the least: each of them exactly (rare: one leaf at a time)
            double lbA = INFINITY, lbC = INFINITY; uint32_t lpA = kNone, lpC = kNone;
            for (int m = 0; m < nl; ++m) {
                uint32_t lo, hi;
                range_of(s_leaf[m], lo, hi);
                for (uint32_t i0 = lo + (uint32_t)lane; i0 < hi; i0 += 4u * (uint32_t)G) {
                    float4 v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const uint32_t iu = i0 + (uint32_t)u * (uint32_t)G; v[u] = p4[iu < hi ? iu : hi - 1u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const uint32_t i = i0 + (uint32_t)u * (uint32_t)G;
                        const float4 pv = v[u];
                        const float dx = o0 - pv.x, dy = o1 - pv.y, dz = o2 - pv.z;
                        const float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        if (i < hi && uu - err_of(uu) <= thi) {
                            const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
                            if (WHICH & 1) { const double ex = ax - x, ey = ay - y, ez = az - z; nn_merge(lbA, lpA, (ex * ex + ey * ey) + ez * ez, i, perm_g); }
                            if (WHICH & 2) { const double ex = qx - x, ey = qy - y, ez = qz - z; nn_merge(lbC, lpC, (ex * ex + ey * ey) + ez * ez, i, perm_g); }
                        }
                    }
                }
            }
            if (WHICH & 1) {
                const double g = group_min_f64(lbA, G);
                const uint32_t key = group_min_u32((lbA == g && lpA != kNone) ? perm_g[lpA] : kNone, G);
                const uint32_t pw = group_min_u32((lbA == g && lpA != kNone && perm_g[lpA] == key) ? lpA : kNone, G);
                if (pw != kNone) nn_merge(bestA, bposA, g, pw, perm_g);
            }
            if (WHICH & 2) {
                const double g = group_min_f64(lbC, G);
                const uint32_t key = group_min_u32((lbC == g && lpC != kNone) ? perm_g[lpC] : kNone, G);
                const uint32_t pw = group_min_u32((lbC == g && lpC != kNone && perm_g[lpC] == key) ? lpC : kNone, G);
                if (pw != kNone) nn_merge(bestC, bposC, g, pw, perm_g);
            }
        }
    }
    if (go >= 0) {   // the walk stopped at a full round: which far side comes next, under the bound as it is NOW (the level picked above has not been entered)
        const float bestf = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
        go = nn_next_level(pd2, side, done, node, D, bestf, s_nodes);
    }
}

// diagnostic (iba_debug_nn): the search of iba_nn_kernel on caller-supplied LiDAR-frame queries, one lane per query, run to its
// end. mode 1: the query is the association path's (a) alone; 2: the cost path's (c) alone; 3 / 4: both paths are searched
// together, the query as a (3) or as c (4), its partner 1e-7 beside it as the reference's two float/double islands are.
// out_idx = ORIGINAL point index, out_d2 = exact squared distance.
__global__ __launch_bounds__(256) void iba_nn_probe_kernel(DevProblem dp, int frame, const double* __restrict__ q, int n, int mode,
                                                           uint32_t* __restrict__ out_idx, double* __restrict__ out_d2) {
    extern __shared__ __align__(16) unsigned char smem[];
    TreeNode* s_nodes = (TreeNode*)smem;
    const FrameHdr& h = dp.frames[frame];
    const uint32_t P = h.P, D = h.depth;
    for (uint32_t i = threadIdx.x; i < (1u << D) - 1u; i += blockDim.x) s_nodes[i] = dp.nodes[h.node_base + i];
    __syncthreads();
    const int e = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (e >= n || P == 0) return;
    const float4* p4 = dp.pts4 + h.pt_base;
    const uint32_t* perm = dp.perm + h.pt_base;
    const double x = q[3 * e], y = q[3 * e + 1], z = q[3 * e + 2];
    const double ox = x + 1e-7, oy = y - 1e-7, oz = z + 0.5e-7;
    IBA_LANE_NN_DECL;
    actA = mode != 2; actC = mode != 1;
    if (actA) { const bool own = mode != 4; ax = own ? x : ox; ay = own ? y : oy; az = own ? z : oz; }
    if (actC) { const bool own = mode != 3; qx = own ? x : ox; qy = own ? y : oy; qz = own ? z : oz; }
    lane_nn_begin(IBA_LANE_NN_PASS);
    do {
        if (mode == 1) lane_nn_visit<1>(IBA_LANE_NN_PASS, s_nodes, p4, perm, P, D);
        else if (mode == 2) lane_nn_visit<2>(IBA_LANE_NN_PASS, s_nodes, p4, perm, P, D);
        else lane_nn_visit<3>(IBA_LANE_NN_PASS, s_nodes, p4, perm, P, D);
    } while (go >= 0);
    const bool a = mode == 1 || mode == 3;
    const uint32_t bp = a ? bposA : bposC;
    out_idx[e] = bp != kNone ? perm[bp] : kNone; out_d2[e] = a ? bestA : bestC;
}

// ------------------------------------------------------------------------------------------------------------------
// Anchored neighbour lists: the 1-NN search memoised around an anchor extrinsic.
//
// The 1-NN search of iba_nn_kernel answers, per (MapPoint, candidate): which scan point is nearest to the MapPoint moved into the
// LiDAR frame by this candidate (iba_global.cpp:231-234, 116-122; iba_local.cpp:238-239, 282-290). The MapPoint is the keypoint's,
// not the candidate's, and an optimiser's candidates stay near each other: under candidate b the query q_b = s_b Ri_b m + ti_b
// (m = the MapPoint in the camera frame) lies S = |q_b - q_a| from the query q_a of an ANCHOR extrinsic a. Let p_1 .. p_M be the
// M scan points nearest to q_a (distances d_1 <= .. <= d_M, exact). The nearest point p of q_b has |q_b - p| <= |q_b - p_1| <=
// d_1 + S, hence |q_a - p| <= d_1 + 2 S: if d_1 + 2 S < d_M (or the scan has fewer than M points), every nearest point of q_b —
// and every point tied with it — is among the listed points within d_1 + 2 S of q_a. A lane of iba_nn_kernel evaluates its own
// S from its own exact queries, checks that certificate, and picks its nearest point among those few entries with the search's own
// f64 expressions and tie rule: the same bits as the tree search, which remains the path of every lane whose certificate fails.
// The lists are built once per anchor (iba_anchor_kernel: exact kNN(M) per MapPoint keypoint) and rebuilt when the candidates
// have drifted away from it (host, run_split); each entry carries what the memoised planes say at its point, so a lane that
// picks it needs no further gather. At the bench shape a candidate 0.5 mrad / 5 mm / 0.1 % from the anchor has S ~ 4 cm against
// d_M ~ 30 cm: one or two entries qualify, and no lane searches the tree.
// ------------------------------------------------------------------------------------------------------------------
constexpr int kCoopMax = kNNThreads / 2; // entries the lists left over that a block searches with a group of 2 .. 64 lanes each (more: one lane each)
constexpr int kCoopLimit = kCoopMax;     // (r05: up to four passes of two-lane groups instead of one lane each beyond kCoopMax, i.e. 4 * kCoopMax here — measured 0.147 -> 0.152 ms at 40 KF x 120 k points: 52 of 2240 blocks take that path and their time is the same either way)
static_assert((int)(kSliceW * (uint32_t)kMaxGroup) - kCoopLimit >= kNNThreads / 2 * 4, "the rounds of leaves of a cooperative block (wave_nn_round: at least four per two-lane group) live behind its queue");
constexpr int kSetM = 8;              // neighbours a list holds
// a listed neighbour: the scan point, a float lower bound of its distance to the anchor query, and (plane_cache = 1) what the
// memoised planes at it say: flags bit 0 = the local plane is valid (pointcloud.h:699-717), bit 1 = the cost term is
// point-to-plane (iba_global.cpp:136-148) with normal n
struct SetPt { float x, y, z; uint32_t pos; double nx, ny, nz; uint32_t flags; float da_lo; };   // 48 B
struct AnchorHdr { double qa[3]; double d1, dM; uint32_t count; float da1_lo; };                // 48 B: anchor query, nearest / farthest listed distance, count, lower bound of the second neighbour's distance
static_assert(sizeof(SetPt) == 48 && sizeof(AnchorHdr) == 48, "anchor rows are read as 16-byte pieces");
// A (frame, keypoint) row is 512 bytes, 128-byte aligned: header and nearest neighbour share the FIRST 128-byte line (96 bytes:
// what nearly every lane needs, and all a lane fetches up front); the neighbours 1..7 follow from byte 128. (As nine
// consecutive 48-byte pieces a lane's header + two neighbours straddled 2.6 lines on average, and the lines, not the bytes, are
// what the L2 moves.)
constexpr size_t kAnchorRowBytes = 512;
__host__ __device__ inline const unsigned char* anchor_row(const void* base, size_t row) { return (const unsigned char*)base + row * kAnchorRowBytes; }
__device__ __forceinline__ const AnchorHdr* anchor_hdr(const unsigned char* r) { return (const AnchorHdr*)r; }
__device__ __forceinline__ const SetPt* anchor_pt(const unsigned char* r, uint32_t i) { return (const SetPt*)(r + (i == 0u ? 48u : 128u + (i - 1u) * 48u)); }
struct AnchorRef { double M[9], t[3]; };   // s_a Ri_a, ti_a of the anchor extrinsic: q_a = M m + t
struct AnchorArgs { DevProblem dp; DevParams prm; AnchorRef ar; };
// grid: (ceil(max MapPoint keypoints of a frame / kAnchorThreads), frames); one lane per MapPoint keypoint; dynamic LDS: the tree nodes
constexpr int kAnchorThreads = 256;
__global__ __launch_bounds__(kAnchorThreads) void iba_anchor_kernel(AnchorArgs a, SetPt* __restrict__ rows) {
    extern __shared__ __align__(16) unsigned char smem[];
    const DevProblem& dp = a.dp;
    const DevParams& prm = a.prm;
    const int f = blockIdx.y;
    const FrameHdr& h = dp.frames[f];
    if (blockIdx.x * (uint32_t)kAnchorThreads >= h.n_mpk) return;
    TreeNode* s_nodes = (TreeNode*)smem;
    const uint32_t P = h.P, D = h.depth;
    for (uint32_t i = threadIdx.x; i < (1u << D) - 1u; i += kAnchorThreads) s_nodes[i] = dp.nodes[h.node_base + i];
    __syncthreads();
    const uint32_t j = blockIdx.x * (uint32_t)kAnchorThreads + threadIdx.x;
    if (j >= h.n_mpk) return;
    const uint32_t k = dp.mpk[h.mpk_base + j];
    unsigned char* row = (unsigned char*)anchor_row(rows, (size_t)f * dp.max_k + k);
    AnchorHdr hd;
    const float4* p4 = dp.pts4 + h.pt_base;
    const float4 mp = dp.kp_mp[h.kp_base + k];
    const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
    const double m0 = ((h.Tcw[0] * w0 + h.Tcw[1] * w1) + h.Tcw[2] * w2) + h.Tcw[3];
    const double m1 = ((h.Tcw[4] * w0 + h.Tcw[5] * w1) + h.Tcw[6] * w2) + h.Tcw[7];
    const double m2 = ((h.Tcw[8] * w0 + h.Tcw[9] * w1) + h.Tcw[10] * w2) + h.Tcw[11];
    const double qx = ((a.ar.M[0] * m0 + a.ar.M[1] * m1) + a.ar.M[2] * m2) + a.ar.t[0];
    const double qy = ((a.ar.M[3] * m0 + a.ar.M[4] * m1) + a.ar.M[5] * m2) + a.ar.t[1];
    const double qz = ((a.ar.M[6] * m0 + a.ar.M[7] * m1) + a.ar.M[8] * m2) + a.ar.t[2];
    hd.qa[0] = qx; hd.qa[1] = qy; hd.qa[2] = qz; hd.d1 = 0; hd.dM = 0; hd.count = 0u; hd.da1_lo = INFINITY;
    const double aq = (fabs(qx) + fabs(qy)) + fabs(qz);
    if (P == 0 || !(aq <= 1e30)) { hd.count = 0u; hd.dM = -1.0; *(AnchorHdr*)row = hd; return; }   // dM < 0: no certificate, the lanes search the tree
    // ---- exact kNN(M): the M smallest (d^2, tree position) in registers, kept sorted by an unrolled insertion; register-path
    //      traversal with float lower bounds of the plane distances against the current M-th distance ----
    double bd[kSetM]; uint32_t bp[kSetM];
#pragma unroll
    for (int i = 0; i < kSetM; ++i) { bd[i] = INFINITY; bp[i] = kNone; }
    const float o0 = (float)qx, o1 = (float)qy, o2 = (float)qz;
    const float slop = 1e-4f + 2e-6f * (float)aq;   // float(o) and the float plane differences: far below a millimetre
    const uint32_t first_leaf = (1u << D) - 1u;
    float pd2[kPathMax];
#pragma unroll
    for (int L = 0; L < kPathMax; ++L) pd2[L] = INFINITY;
    uint32_t side = 0u, done = 0u, node = 0u; int go = -1;
    do {
        int start = 0;
        if (go >= 0) {
            const uint32_t anc = ((node + 1u) >> (D - (uint32_t)go)) - 1u;
            done |= 1u << go; side ^= 1u << go; side |= 0x10000u << go;   // (bits 16..: far turns of the path, see nn_next_level)
            node = 2u * anc + 1u + ((side >> go) & 1u);
            start = go + 1;
        }
        {
            const uint32_t keep = (1u << start) - 1u;
            side &= keep | (keep << 16); done &= keep;
            uint32_t n1 = node + 1u;
#pragma unroll
            for (int L = 0; L < kPathMax; ++L) {
                if (L >= (int)D) break;
                if (L >= start) {
                    const TreeNode n = s_nodes[n1 - 1u];
                    const float dd = (n.dim == 0 ? o0 : (n.dim == 1 ? o1 : o2)) - n.split;
                    const uint32_t r = (~__float_as_uint(dd)) >> 31;
                    const float ad = fmaxf(fabsf(dd) - slop, 0.f);
                    pd2[L] = ad * ad * 0.999999f;   // <= the squared distance from the query to the splitting plane
                    side |= r << L;
                    n1 = (n1 << 1) | r;
                }
            }
            node = n1 - 1u;
        }
        const uint32_t lj = node - first_leaf;
        const uint32_t lo = (uint32_t)(((uint64_t)lj * P) >> D), hi = (uint32_t)(((uint64_t)(lj + 1) * P) >> D);
        for (uint32_t i0 = lo; i0 < hi; i0 += 8u) {   // eight loads in flight per step
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const uint32_t iu = i0 + (uint32_t)u; v[u] = p4[iu < hi ? iu : hi - 1u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t i = i0 + (uint32_t)u;
                const double dx = qx - (double)v[u].x, dy = qy - (double)v[u].y, dz = qz - (double)v[u].z;
                double d2 = (dx * dx + dy * dy) + dz * dz;
                uint32_t pi = i;
                if (i < hi && d2 < bd[kSetM - 1]) {   // insert: a strictly closer point only (ties at the M-th distance stay out; the certificate is strict)
#pragma unroll
                    for (int t = 0; t < kSetM; ++t) {
                        const bool lt = d2 < bd[t] || (d2 == bd[t] && pi < bp[t]);
                        const double td = lt ? bd[t] : d2; const uint32_t tp = lt ? bp[t] : pi;
                        bd[t] = lt ? d2 : bd[t]; bp[t] = lt ? pi : bp[t];
                        d2 = td; pi = tp;
                    }
                }
            }
        }
        const float bound = bd[kSetM - 1] < INFINITY ? (float)bd[kSetM - 1] * 1.000001f + 1e-30f : INFINITY;
        go = nn_next_level(pd2, side, done, node, D, bound, s_nodes);
    } while (go >= 0);
    const PlaneRec* planes_cost = dp.plane_cost + h.pt_base;
    const PlaneRec* planes_local = dp.plane_local + h.pt_base;
    uint32_t cnt = 0u;
#pragma unroll
    for (int i = 0; i < kSetM; ++i) {
        if (bp[i] == kNone) continue;
        const float4 v = p4[bp[i]];
        SetPt sp; sp.x = v.x; sp.y = v.y; sp.z = v.z; sp.pos = bp[i]; sp.nx = 0; sp.ny = 0; sp.nz = 0; sp.flags = 0u;
        const float dl = (float)sqrt(bd[i]);
        sp.da_lo = fmaxf(dl * 0.999999f - 1e-30f, 0.f);   // <= the exact distance
        if (prm.plane_cache) {   // with refitted planes the fit kernels decide
            const PlaneRec rl = planes_local[bp[i]];
            sp.flags = (local_neigh_ok(prm, rl) && local_plane_ok(prm, rl)) ? 1u : 0u;
            if (prm.use_plane) {
                const PlaneRec rc = planes_cost[bp[i]];
                if (!(rc.far_d2 < prm.min_diff_dist2) && !(rc.k < prm.norm_min_pts) && !(rc.reg_sum / (double)(rc.k - 1) > prm.norm_reg_threshold)) { sp.flags |= 2u; sp.nx = rc.nx; sp.ny = rc.ny; sp.nz = rc.nz; }
            }
        }
        *(SetPt*)anchor_pt(row, (uint32_t)i) = sp;
        if (i == 0) hd.d1 = sqrt(bd[0]);
        if (i == 1) hd.da1_lo = sp.da_lo;
        hd.dM = sqrt(bd[i]);
        ++cnt;
    }
    hd.count = cnt;
    if (cnt < (uint32_t)kSetM) hd.dM = INFINITY;   // the whole scan is listed: every radius is certified
    *(AnchorHdr*)row = hd;
}

// ------------------------------------------------------------------------------------------------------------------
// iba_nn_kernel<WHICH>. WHICH bit 0: association-path queries present, bit 1: cost-path queries.
// grid: 8 * ceil(n_frames/8) * NG * NS blocks (NG = ceil(B / CG) candidate groups, NS keypoint slices) of kNNThreads.
//
// One LANE per (MapPoint, candidate): the exact 1-NN searches of one list entry (association-path query a, cost-path query c:
// the same MapPoint through different float/double islands of the reference, 1e-7 apart) — the float32-conservative tree walk
// and float-filtered, f64-confirmed leaf scans of nn_dual_step with one lane per query pair. The waves of a block run
// INDEPENDENTLY of each other, without barriers: a lane that has finished its entry takes the next one from the block's
// work list (one LDS atomic per wave and iteration), so every lane of a wave is searching in every iteration however uneven
// the searches are (72 % end in their first leaf, a few need 5..24), and the visited path stays in registers from one leaf
// to the next. The work list of a block = the list entries, of up to CG candidates, that want a search and whose keypoint
// falls into the block's slice, compacted in (candidate, list) order.
// Results: flist[b][f][i].z (kind | tree position of the 3d-3d block's scan point) directly; the cost distance of an entry
// goes to its slot of an LDS array (sign bit = point-to-point), which is summed per candidate in a FIXED order afterwards —
// which lane ran which search does not matter, so the sums are bitwise reproducible. One record of kNNPartial doubles per
// (candidate, frame, slice).
// ------------------------------------------------------------------------------------------------------------------
// SETS 1: the anchored neighbour lists are in use (anchor != nullptr); compiled for 4 waves per SIMD — what its LDS allows anyway — so that the
// direct pass keeps two entries' loads in registers
template <int WHICH, int REFIT, int SETS, int T = kNNThreads>   // REFIT: 0 = planes memoised; plane_cache = 0 runs the kernel twice around iba_fit_kernel<.., 2>: kRefitSearch, then kRefitSums; T: threads per block (kNNThreads, or kNNThreadsSmall: see there)
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(SETS ? IBA_NN_SETS_WAVES : IBA_NN_WAVES, SETS ? IBA_NN_SETS_WAVES : IBA_NN_WAVES))) void iba_nn_kernel(NNArgs ka_by_value, const Cand* __restrict__ cands, int B, int CG, int NS,
                                                                                                  double* __restrict__ nn_partials, int nn_nrec, uint4* __restrict__ flist,
                                                                                                  const uint32_t* __restrict__ lcount, int flist_stride, int dbg, double4* __restrict__ frefit,
                                                                                                  const SetPt* __restrict__ anchor, double* __restrict__ he_out, int he_blocks) {
    extern __shared__ __align__(16) unsigned char smem[];
    // K7 rides in front (round 5): the first he_blocks workgroups evaluate the hand-eye term of every (candidate, frame) — two lanes per
    // term, iba_global.cpp:264-276 — into he_out[b][f] for the summing kernel, beside the searches instead of in a launch of their own at
    // the head of the chain (rounds 3-4) or inside the summing kernel's single block per candidate (13 us of dependent f64 there).
    if ((int)blockIdx.x < he_blocks) {
        const int nfh = ka_by_value.dp.n_frames;
        const int i = (int)blockIdx.x * (T / 2) + (int)(threadIdx.x >> 1);
        const bool live = i < B * nfh;
        const int ii = live ? i : 0;
        const double v = he_term(ka_by_value.dp.frames[ii % nfh], cands[ii / nfh], threadIdx.x & 1, live);
        if (live && !(threadIdx.x & 1)) he_out[i] = v;
        return;
    }
    const uint32_t bid = blockIdx.x - (uint32_t)((he_blocks + 7) & ~7);   // (the hand-eye blocks are padded to a multiple of 8: block i of the search still runs on XCD i % 8)
    // SETS: the anchored neighbour lists (iba_anchor_kernel): a lane whose certificate holds picks its nearest points from its
    // keypoint's list instead of searching the tree.
    // kRefitSearch = the searches only (neighbour and query offset of every entry -> flist.z / frefit), kRefitSums = the fixed-order
    // sums over the distances the fit kernel left in frefit
    constexpr int refit = REFIT;
    typedef __attribute__((address_space(4))) const NNArgs NNArgsC;
    NNArgsC* ka = (NNArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)ka_by_value;
#define dp (ka->dp)
#define prm (ka->prm)
#define lay (ka->lay)
    constexpr int kMaxGroup = T / 32 < 16 ? T / 32 : 16, kCoopLimit = T / 2;   // (of THIS block shape: they shadow the namespace's, which are kNNThreads')
    static_assert((int)(kSliceW * (uint32_t)kMaxGroup) - kCoopLimit >= T / 2 * 4 && 4 * (2 * kMaxGroup + 4) <= 128, "LDS plan of the block shape");
    const int tid = threadIdx.x, lane = tid & 63;
#ifdef IBA_DIAG_COUNTERS
    unsigned long long nn_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nn_last = __builtin_readcyclecounter();   // cycles of thread 0 per phase -> dp.diag[8 + 2 i] (64-bit sums)
#define NN_TICK(i) do { const unsigned long long _t = __builtin_readcyclecounter(); nn_t[i] += _t - nn_last; nn_last = _t; } while (0)
#else
#define NN_TICK(i) do { } while (0)
#endif
    const int nf = dp.n_frames;
    const int per_xcd = (nf + 7) / 8;
    const int NG = (B + CG - 1) / CG;
    const int per_frame = NG * NS;
    if (blockIdx.x < (uint32_t)((he_blocks + 7) & ~7)) return;   // (padding)
    const int xcd = bid & 7, jj = bid >> 3;
    const int f = xcd + 8 * (jj / per_frame);
    if (f >= nf || jj / per_frame >= per_xcd) return;
    const int g = (jj % per_frame) / NS, sl = (jj % per_frame) % NS;
    const FrameHdr& h = dp.frames[f];
    const uint32_t P = h.P, D = h.depth;
    const int cands_here = min(CG, B - g * CG);
    constexpr uint32_t kWantMask = ((WHICH & 2) ? kFlagC : 0u) | ((WHICH & 1) ? kFlagA : 0u);

    TreeNode* s_nodes = (TreeNode*)(smem + lay.off_nodes);
    double* s_res = (double*)(smem + lay.off_res);           // cost distance of a work entry (NaN: none; sign bit: point-to-point)
    uint32_t* s_n = (uint32_t*)(smem + lay.off_misc);        // [kMaxGroup] list length per candidate of the group
    uint32_t* s_ctr = s_n + kMaxGroup;                       // [1] next unclaimed work entry
    double* s_cd = (double*)(smem + lay.off_cd);             // [kMaxGroup][kCdDoubles] candidate constants of the query transform
    int cg_shift = 0; while ((1 << cg_shift) < CG) ++cg_shift;   // CG is a power of two (host)

    // (round 6) what the block stages — its candidates' list lengths, transform constants and anchor sets — is requested BEFORE the list lengths of the
    // empty-slice test are waited for: the block's start-up was two dependent round trips (the scalar loads of the test, then these), 28 us of the kernel
    // when it is cut short behind its first barrier (tools/nn_exp.sh)
    uint32_t my_n = 0u;
    if (tid < cands_here) my_n = lcount[(size_t)(g * CG + tid) * nf + f];
    double my_cd = 0.0;
    if (tid < cands_here * kCdDoubles) my_cd = ((const double*)&cands[g * CG + tid / kCdDoubles])[12 + tid % kCdDoubles];
    uint32_t my_sel = 255u;
    if (SETS && tid < cands_here) my_sel = (uint32_t)ka->anchor_sel[g * CG + tid];
    // ---- a slice beyond every list of the group has nothing to search or sum: zero record, done (the lists hold ~575 of
    //      up to 7 x 128 positions at the bench shape: two blocks in seven) ----
    uint32_t nmax_u = 0u;
    for (int cc = 0; cc < cands_here; ++cc) nmax_u = max(nmax_u, lcount[(size_t)(g * CG + cc) * nf + f]);   // uniform addresses: scalar loads
    if ((uint32_t)sl * kSliceW >= nmax_u) {
        const int cc = tid >> 5, q = tid & 31;
        if (cc < cands_here && q < kNNPartial) nn_partials[((size_t)(g * CG + cc) * nn_nrec + (size_t)f * NS + sl) * kNNPartial + q] = 0.0;
        return;
    }
    // ---- kd nodes -> LDS; list lengths of the group's candidates ----
    const uint32_t nnodes = (1u << D) - 1u;
    if (!SETS) for (uint32_t i = tid; i < nnodes; i += T) s_nodes[i] = dp.nodes[h.node_base + i];   // with the batch's neighbour sets the tree is staged only if a lane needs it
    if (tid < kMaxGroup) s_n[tid] = my_n;
    uint32_t* s_sel = s_ctr + 4;                             // [kMaxGroup] anchor set of each candidate of the group (255: none)
    if (SETS && tid < kMaxGroup) s_sel[tid] = my_sel;
    if (tid < cands_here * kCdDoubles) s_cd[tid] = my_cd;
    // (r04: the counters and the result slots are cleared HERE, before the block's first barrier — the slice's extent follows from the
    //  list lengths every thread has just read — instead of behind two more barriers of their own)
    if (tid == 0) { s_ctr[0] = 0u; s_ctr[1] = 0u; }
    if ((WHICH & 2) && refit != kRefitSums) {
        const uint32_t lo_u = min(nmax_u, (uint32_t)sl * kSliceW), hi_u = min(nmax_u, lo_u + kSliceW);
        for (uint32_t i = tid; i < ((hi_u - lo_u) << cg_shift); i += T) s_res[i] = NAN;   // entries that turn out to want no cost search
    }
    __syncthreads();
    NN_TICK(0);   // block start-up: list lengths, candidate constants -> LDS, first barrier
    if (dbg == 1) return;
    // Work entry w of the block = (candidate w % CG, list position i_lo + w / CG): the candidates' lists interleaved, so that the
    // lanes of a wave search for (nearly) the same MapPoints under different candidates and walk the same leaves. The slices
    // of a (frame, group) split the list positions.
    uint32_t nmax = 0u;
#pragma unroll
    for (int cc = 0; cc < kMaxGroup; ++cc) nmax = max(nmax, s_n[cc]);
    const uint32_t i_lo = min(nmax, (uint32_t)sl * kSliceW), i_hi = min(nmax, i_lo + kSliceW);
    const uint32_t W = (i_hi - i_lo) << cg_shift;   // <= kSliceW * kMaxGroup result slots

    const float4* p4 = dp.pts4 + h.pt_base;
    const float4* kmp = dp.kp_mp + h.kp_base;   // the MapPoint of a list entry's keypoint (shared by the candidates: L2)
    const uint32_t* perm_g = dp.perm + h.pt_base;
    const PlaneRec* planes_cost = dp.plane_cost + h.pt_base;
    const PlaneRec* planes_local = dp.plane_local + h.pt_base;

    // per-thread partial sums of the final (fixed-order) pass: thread t sums entries of candidate t / 32
    double fin_sum = 0.0; uint32_t fin_c = 0, fin_v = 0, fin_pl = 0, fin_pt = 0;
    uint32_t left_to_tree = 0u;   // diagnostic (record slot 5): entries of this block the anchored lists could not settle

    {
        const uint32_t c0 = 0u, c1 = W;
        if (refit == kRefitSums) {
            if (WHICH & 2) for (uint32_t i = tid; i < c1 - c0; i += T) {
                double r = NAN;
                const uint32_t cc = i & ((1u << cg_shift) - 1u), il = i_lo + (i >> cg_shift);
                if (il < s_n[cc]) r = frefit[((size_t)(g * CG + (int)cc) * nf + f) * (size_t)flist_stride + il].x;
                s_res[i] = r;
            }
            __syncthreads();
        }
        if (dbg == 2) return;

        // ---- the searches ----
        if (refit != kRefitSums) {
            bool have = false;
            uint32_t w = 0u;   // the lane's work entry: candidate w % CG of the group, list position i_lo + w / CG
            IBA_LANE_NN_DECL;
            auto entry_at = [&](uint32_t wn) -> size_t { return ((size_t)((uint32_t)(g * CG) + (wn & ((1u << cg_shift) - 1u))) * nf + f) * (size_t)flist_stride + (i_lo + (wn >> cg_shift)); };
            // the two MapPoint -> LiDAR-frame queries of an entry under candidate cc (iba_local.cpp:238-239,282 and iba_global.cpp:231-234)
            auto make_queries_to = [&](uint32_t cc, const uint4& e, const float4& mp, bool& actA, bool& actC, double& ax, double& ay, double& az, double& qx, double& qy, double& qz) {
                actC = (WHICH & 2) && (e.w & kFlagC);
                actA = (WHICH & 1) && (e.w & kFlagA);
                // (round 6) both queries are computed by every lane and an unwanted one is set to NaN behind it: nearly every entry wants both, and a branch
                // around each was two exec-mask round trips per pick in a kernel bound by instruction issue; the candidate's rotation row by row (4 of its
                // 13 constants live at a time)
                const double* cdl = s_cd + cc * kCdDoubles;
                const double s = cdl[0];
                const float s32 = *(const float*)(cdl + 13);
                double sx = 0, sy = 0, sz = 0, cx_ = 0, cy_ = 0, cz_ = 0;
                if (WHICH & 1) {
                    const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
                    const double mx = ((h.Tcw[0] * w0 + h.Tcw[1] * w1) + h.Tcw[2] * w2) + h.Tcw[3];
                    const double my = ((h.Tcw[4] * w0 + h.Tcw[5] * w1) + h.Tcw[6] * w2) + h.Tcw[7];
                    const double mz = ((h.Tcw[8] * w0 + h.Tcw[9] * w1) + h.Tcw[10] * w2) + h.Tcw[11];
                    sx = mx * s; sy = my * s; sz = mz * s;
                }
                if (WHICH & 2) {
                    const double ts0 = h.Tcw[3] * s, ts1 = h.Tcw[7] * s, ts2 = h.Tcw[11] * s;   // TcwRS translation *= scale (:208)
                    const float m0 = mp.x * s32, m1 = mp.y * s32, m2 = mp.z * s32;       // CV_32F product (:232)
                    const double a0 = (double)m0, a1 = (double)m1, a2 = (double)m2;
                    cx_ = ((h.Tcw[0] * a0 + h.Tcw[1] * a1) + h.Tcw[2] * a2) + ts0;
                    cy_ = ((h.Tcw[4] * a0 + h.Tcw[5] * a1) + h.Tcw[6] * a2) + ts1;
                    cz_ = ((h.Tcw[8] * a0 + h.Tcw[9] * a1) + h.Tcw[10] * a2) + ts2;
                }
                double oa[3], oc[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double r0 = cdl[1 + 3 * i], r1 = cdl[2 + 3 * i], r2 = cdl[3 + 3 * i], t0 = cdl[10 + i];
                    oa[i] = ((r0 * sx + r1 * sy) + r2 * sz) + t0;
                    oc[i] = ((r0 * cx_ + r1 * cy_) + r2 * cz_) + t0;
                }
                ax = actA ? oa[0] : NAN; ay = actA ? oa[1] : NAN; az = actA ? oa[2] : NAN;
                qx = actC ? oc[0] : NAN; qy = actC ? oc[1] : NAN; qz = actC ? oc[2] : NAN;
            };
            auto make_queries = [&](uint32_t cc, const uint4& e, const float4& mp) { make_queries_to(cc, e, mp, actA, actC, ax, ay, az, qx, qy, qz); };   // into the lane's search state
            // the finished searches of entry wn: the association's neighbour (+ kind), the cost distance. From the tree search
            // the plane records are fetched here; a set lane brings them along (sa / sc: the set entries of its two neighbours)
            auto finish = [&](uint32_t wn, const SetPt* sa, const SetPt* sc) {
                const size_t at = entry_at(wn);
                if ((WHICH & 1) && actA && !(bestA > prm.max_3d_dist2)) {   // the association keeps its neighbour only within max_3d_dist (iba_local.cpp:289)
                    bool state = false;   // refit: settled by the fit kernel
                    if (!refit) {
                        if (sa) state = (sa->flags & 1u) != 0u;
                        else { const PlaneRec r2 = planes_local[bposA]; state = local_neigh_ok(prm, r2) && local_plane_ok(prm, r2); }   // pointcloud.h:699-717
                    }
                    flist[at].z = bposA | (state ? 0x80000000u : 0u);
                }
                double res = NAN;
                if ((WHICH & 2) && actC) {
                    float px, py, pz;
                    if (sc) { px = sc->x; py = sc->y; pz = sc->z; } else { const float4 pv = p4[bposC]; px = pv.x; py = pv.y; pz = pv.z; }
                    const double ex = (double)px - qx, ey = (double)py - qy, ez = (double)pz - qz;
                    if (refit) frefit[at] = make_double4(ex, ey, ez, __longlong_as_double((long long)bposC));
                    else if (sc) {   // cost_res with the plane's verdict already taken (iba_anchor_kernel): the same expressions
                        const double dist = (sc->flags & 2u) ? fabs(ex * sc->nx + ey * sc->ny + ez * sc->nz) : sqrt((ex * ex + ey * ey) + ez * ez);
                        res = (sc->flags & 2u) ? dist : -dist;
                    } else {
                        PlaneRec rec; rec.k = 0;
                        if (prm.use_plane) rec = planes_cost[bposC];   // the whole record in one round trip
                        res = cost_res(prm, prm.use_plane != 0, rec, ex, ey, ez);
                    }
                }
                if (WHICH & 2) s_res[wn - c0] = res;   // (NaN when the planes are refitted: the sums come from the fit kernel's distances; a left-over entry's seed is cleared either way)
            };
            uint32_t* s_ovf = (uint32_t*)(smem + lay.off_ovf);   // work entries whose keypoint has no usable set: searched in the tree below
            // the first bound of a left-over entry's search: its nearest listed points (their positions wait in the entry's result slot)
            auto seed = [&](uint32_t wn) {
                const uint2 sd = ((const uint2*)s_res)[wn - c0];
                if ((WHICH & 1) && actA && sd.x != kNone) { const float4 pv = p4[sd.x]; const double dx = ax - (double)pv.x, dy = ay - (double)pv.y, dz = az - (double)pv.z; bestA = (dx * dx + dy * dy) + dz * dz; bposA = sd.x; }
                if ((WHICH & 2) && actC && sd.y != kNone) { const float4 pv = p4[sd.y]; const double dx = qx - (double)pv.x, dy = qy - (double)pv.y, dz = qz - (double)pv.z; bestC = (dx * dx + dy * dy) + dz * dz; bposC = sd.y; }
            };
            uint32_t c_end = c1;                                 // entries the persistent loop hands out
            if (SETS) {
                // ---- the anchored neighbour lists of the MapPoint keypoints are there (iba_anchor_kernel).
                //      One pass, no claiming: thread t takes the entries t, t + T, ...; per entry one dependent chain of three loads
                //      (entry -> set -> plane records), the next entry's first load in flight meanwhile ----
                // two entries per step: their loads are issued stage by stage (entries, then list rows), so a thread's four entries
                // cost four dependent round trips instead of eight
                // (round 6, measured and reverted: the ISA waits behind each conditional load of this step — a load merged with a default behind a branch is
                //  waited for at the merge —, six dependent round trips where two were meant. With every load unconditional on a safe address, pinned by
                //  compiler barriers, one entry per step and the next entry's fetch in flight (two round trips per entry, one exposed): 38 % more vector-memory
                //  instructions — lanes without an entry load too — and the kernel 0.121 -> 0.125 ms on the same box, twice. The pass is not bound by its
                //  round trips: 2.7e10 L2 requests/s against the ~1e11 the association and factor kernels reach, VALU-active 0.36.)
                auto fetch = [&](uint32_t wn, uint4& e, float4& mp) {
                    e = make_uint4(0u, 0u, 0u, 0u); mp = make_float4(0.f, 0.f, 0.f, 0.f);
                    const uint32_t cc = wn & ((1u << cg_shift) - 1u), il = i_lo + (wn >> cg_shift);
                    if (wn < c1 && il < s_n[cc]) { e = flist[entry_at(wn)]; if (e.w & kWantMask) mp = kmp[e.x]; }
                };
                // (round 6: the pick keeps the INDEX of its nearest listed points, not copies of their 48-byte entries — the nearest listed point to the anchor,
                //  p0, is the answer of nearly every lane and is in registers already; two copies of it per entry were 50 register moves and 24 live registers.
                //  The certificate's S comes from a float square root rounded UP: a larger S can only send an entry to the second chance or the tree.)
                auto pick = [&](uint32_t wn, const uint4& e, const float4& mp, const AnchorHdr& hd, SetPt p0, const unsigned char* row) {
                    // (its own queries and bests, not the lane's search state: that lives across the loop for the left-over searches, and every pick merging into it was a register move per word)
                    bool actA, actC; double ax, ay, az, qx, qy, qz, bestA, bestC; uint32_t bposA, bposC;
                    make_queries_to(wn & ((1u << cg_shift) - 1u), e, mp, actA, actC, ax, ay, az, qx, qy, qz);
                    // the certificate: this candidate's queries are S from the anchor's; its nearest points lie within d_1 + 2 S of the
                    // anchor query, and the list is complete out to d_M (exclusive)
                    double S2 = 0.0;   // (an unwanted query is NaN: fmax drops it, the select keeps the 0)
                    if (WHICH & 1) { const double dx = ax - hd.qa[0], dy = ay - hd.qa[1], dz = az - hd.qa[2]; const double dA = (dx * dx + dy * dy) + dz * dz; S2 = actA ? dA : 0.0; }
                    if (WHICH & 2) { const double dx = qx - hd.qa[0], dy = qy - hd.qa[1], dz = qz - hd.qa[2]; const double dC = (dx * dx + dy * dy) + dz * dz; S2 = actC ? fmax(S2, dC) : S2; }
                    auto sqrt_up = [](double v2) -> double { return (double)(__builtin_sqrtf((float)v2) * 1.000001f + 1e-18f); };   // >= sqrt(v2) (float rounding 6e-8, flushed denormals < 1.1e-19); NaN stays NaN
                    const double S = sqrt_up(S2);
                    const double radius = (hd.d1 + 2.0 * S) * (1.0 + 1e-12) + 1e-12;
                    const bool quick = radius < hd.dM;   // (NaN queries fail both tests)
                    // Second chance, from the whole row: with r = the distance to the nearest LISTED point, every point within r of this
                    // query is within r + S of the anchor's; if that is inside d_M they are all listed and the nearest listed point is the
                    // nearest point. (Dense scans: d_M shrinks with the point spacing, S does not.) An entry that fails this too goes to
                    // the tree search WITH its nearest listed points as the first bound: a query far from the scan (a MapPoint 2 m in front
                    // of a wall) otherwise opens every leaf within its first, poor, bound.
                    auto leave = [&](uint32_t pa, uint32_t pc) { ((uint2*)s_res)[wn - c0] = make_uint2(pa, pc); s_ovf[atomicAdd(&s_ctr[1], 1u)] = wn; };
                    if (hd.count == 0u) { leave(kNone, kNone); return; }
                    bestA = INFINITY; bestC = INFINITY; bposA = kNone; bposC = kNone;
                    uint32_t ia = 0u, ic = 0u;   // which listed point each search ended on
                    const float rf = quick ? (float)radius * 1.000001f + 1e-30f : INFINITY;   // >= radius
                    {   // the nearest listed point (its own lower bound is <= d_1 <= radius: it always qualifies)
                        const double x = (double)p0.x, y = (double)p0.y, z = (double)p0.z;
                        // (an unwanted query is NaN: its distance compares false with everything and nn_merge leaves best / bpos alone)
                        if (WHICH & 1) { const double dx = ax - x, dy = ay - y, dz = az - z; nn_merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, p0.pos, perm_g); }
                        if (WHICH & 2) { const double dx = qx - x, dy = qy - y, dz = qz - z; nn_merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, p0.pos, perm_g); }
                    }
                    if (hd.count > 1u && hd.da1_lo <= rf) {   // (the usual case: the second neighbour cannot qualify and its line is not even fetched)
                        for (uint32_t si = 1u; si < hd.count; ++si) {
                            const SetPt pv = *anchor_pt(row, si);
                            if (!(pv.da_lo <= rf)) break;            // the list is sorted by distance to the anchor query: nothing further qualifies
                            const double x = (double)pv.x, y = (double)pv.y, z = (double)pv.z;
                            if ((WHICH & 1) && actA) { const double dx = ax - x, dy = ay - y, dz = az - z; const uint32_t was = bposA; nn_merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, pv.pos, perm_g); if (bposA != was) ia = si; }
                            if ((WHICH & 2) && actC) { const double dx = qx - x, dy = qy - y, dz = qz - z; const uint32_t was = bposC; nn_merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, pv.pos, perm_g); if (bposC != was) ic = si; }
                        }
                    }
                    if (!quick) {
                        double r2 = 0.0;
                        if ((WHICH & 1) && actA) r2 = bestA;
                        if ((WHICH & 2) && actC) r2 = fmax(r2, bestC);
                        if (!((sqrt_up(r2) + S) * (1.0 + 1e-12) + 1e-12 < hd.dM)) { leave(bposA, bposC); return; }   // (NaN queries end here, unseeded)
                    }
                    if (dbg == 5) return;
                    // ---- the results (finish() with the listed points' own verdicts and normals). The rare other pick fetches its entry and finishes INSIDE its own
                    //      branch: a value loaded under a condition and merged behind it is waited for behind it, by every lane (s_waitcnt vmcnt(0)) ----
                    const size_t at = entry_at(wn);
                    auto results = [&](const uint32_t fa, const SetPt& pc) {
                        if ((WHICH & 1) && actA && !(bestA > prm.max_3d_dist2))   // the association keeps its neighbour only within max_3d_dist (iba_local.cpp:289)
                            flist[at].z = bposA | ((!refit && (fa & 1u)) ? 0x80000000u : 0u);   // (refit: the kind bit is settled by the fit kernel)
                        double res = NAN;
                        if ((WHICH & 2) && actC) {
                            const double ex = (double)pc.x - qx, ey = (double)pc.y - qy, ez = (double)pc.z - qz;
                            if (refit) frefit[at] = make_double4(ex, ey, ez, __longlong_as_double((long long)bposC));
                            else if (pc.flags & 2u) res = fabs(ex * pc.nx + ey * pc.ny + ez * pc.nz);   // cost_res with the plane's verdict already taken (iba_anchor_kernel): the same expressions
                            else res = -sqrt((ex * ex + ey * ey) + ez * ez);
                        }
                        if (WHICH & 2) s_res[wn - c0] = res;   // (NaN when the planes are refitted: the sums come from the fit kernel's distances)
                    };
                    if ((ia | ic) == 0u) results(p0.flags, p0);
                    else {
                        const uint32_t fa = ia != 0u ? anchor_pt(row, ia)->flags : p0.flags;
                        const SetPt pc = ic != 0u ? *anchor_pt(row, ic) : p0;
                        results(fa, pc);
                    }
                };
                for (uint32_t wn = (uint32_t)tid; wn < c1; wn += 2u * (uint32_t)T) {
                    uint4 e0, e1; float4 mq0, mq1;
                    fetch(wn, e0, mq0); fetch(wn + (uint32_t)T, e1, mq1);
#ifdef IBA_DIAG_COUNTERS
                    asm volatile("" :: "v"(e0.x), "v"(e1.x), "v"(mq0.x), "v"(mq1.x));   // (diag build only: pinning the values here costs the regular build 33 spilled registers)
#endif
                    NN_TICK(1);   // entries + MapPoints of the step
                    const bool w0 = (e0.w & kWantMask) != 0u, w1 = (e1.w & kWantMask) != 0u;
                    // the lists this lane's candidate reads: the set built around the anchor nearest to it (255: none is near — straight to the tree search)
                    const uint32_t sel0 = s_sel[wn & ((1u << cg_shift) - 1u)], sel1 = s_sel[(wn + (uint32_t)T) & ((1u << cg_shift) - 1u)];
                    const unsigned char* r0 = anchor_row((const unsigned char*)anchor + (size_t)(sel0 != 255u ? sel0 : 0u) * ka->anchor_set_bytes, (size_t)f * dp.max_k + (w0 ? e0.x : 0u));
                    const unsigned char* r1 = anchor_row((const unsigned char*)anchor + (size_t)(sel1 != 255u ? sel1 : 0u) * ka->anchor_set_bytes, (size_t)f * dp.max_k + (w1 ? e1.x : 0u));
                    AnchorHdr h0, h1; SetPt a0, b0;
                    h0.count = h1.count = 0u; h0.dM = h1.dM = -1.0; h0.d1 = h1.d1 = 0.0; h0.da1_lo = h1.da1_lo = INFINITY;
                    a0.flags = b0.flags = 0u;
#if IBA_NN_EXP == 1   /* timing experiment: the nearest neighbour's 48 bytes are not fetched (results invalid) */
                    if (w0 && sel0 != 255u) { h0 = *anchor_hdr(r0); }
                    if (w1 && sel1 != 255u) { h1 = *anchor_hdr(r1); }
#else
                    if (w0 && sel0 != 255u) { h0 = *anchor_hdr(r0); a0 = *anchor_pt(r0, 0u); }   // header + nearest neighbour: ONE 128-byte line, all that most lanes need
                    if (w1 && sel1 != 255u) { h1 = *anchor_hdr(r1); b0 = *anchor_pt(r1, 0u); }
#endif
#if IBA_NN_EXP == 2   /* timing experiment: the loads of the list pass alone, no picks (results invalid) */
                    asm volatile("" :: "v"(h0.count), "v"(h1.count), "v"(a0.pos), "v"(b0.pos), "v"(h0.dM), "v"(h1.dM), "v"(a0.nz), "v"(b0.nz), "v"(mq0.x), "v"(mq1.x));
                    continue;
#endif
#ifdef IBA_DIAG_COUNTERS
                    asm volatile("" :: "v"(h0.count), "v"(h1.count), "v"(a0.pos), "v"(b0.pos));
#endif
                    NN_TICK(2);   // list rows of the step
                    if (w0) pick(wn, e0, mq0, h0, a0, r0);
                    if (w1) pick(wn + (uint32_t)T, e1, mq1, h1, b0, r1);
                    NN_TICK(3);   // the picks (certificate, nearest listed points, results)
                }
                __syncthreads();
                NN_TICK(4);   // waiting for the block's other waves at the end of the list pass
                c_end = s_ctr[1];
                left_to_tree = c_end;
                if (tid == 0) *s_ctr = 0u;
                if (c_end != 0u) for (uint32_t i = tid; i < nnodes; i += T) s_nodes[i] = dp.nodes[h.node_base + i];   // some lanes search the tree after all
                __syncthreads();
            }
            // ---- a few entries left over by the lists: a group of lanes each — a whole wave when there are at most 8, 8 lanes when
            //      there are 64, 2 when there are 256 (their searches would otherwise run one lane each while 500 lanes of the block wait for the slowest) ----
            const bool coop = SETS && c_end != 0u && c_end <= (uint32_t)kCoopLimit;
            if (coop && dbg != 4) {
                int G = 64; while ((uint32_t)(T / G) < c_end && G > 2) G >>= 1;   // the largest group that takes all of them in one pass
                for (uint32_t q = (uint32_t)tid / (uint32_t)G; q < c_end; q += (uint32_t)(T / G)) {
                    const uint32_t wn = s_ovf[q];
                    const uint32_t cc = wn & ((1u << cg_shift) - 1u);
                    const size_t at = entry_at(wn);
                    const uint4 e = flist[at]; const float4 mp = kmp[e.x];   // (an entry is queued only if it exists and wants a search)
                    make_queries(cc, e, mp);
                    lane_nn_begin(IBA_LANE_NN_PASS);
                    seed(wn);
                    // this group's leaves of a round: in the queue's own array, behind the at most kCoopLimit entries of a cooperative block
                    const int round_cap = min(kRoundLeaves, (int)((kSliceW * (uint32_t)kMaxGroup - (uint32_t)kCoopLimit) / (uint32_t)(T / G)));
                    uint32_t* s_leaf = s_ovf + kCoopLimit + ((uint32_t)tid / (uint32_t)G) * (uint32_t)round_cap;
                    // (IBA_NN_DBG 6: no search at all, 7: one round / leaf — timing cuts, the results are garbage)
                    if (dbg == 6) {}
                    else if (ka->nn_rounds) do wave_nn_round<WHICH>(IBA_LANE_NN_PASS, tid & (G - 1), G, s_nodes, p4, perm_g, P, D, s_leaf, round_cap); while (go >= 0 && dbg != 7);
                    else do wave_nn_visit<WHICH>(IBA_LANE_NN_PASS, tid & (G - 1), G, s_nodes, p4, perm_g, P, D); while (go >= 0 && dbg != 7);
                    if (dbg != 5) finish(wn, nullptr, nullptr);   // every lane of the group writes the same values
                }
            }
            // ---- persistent lanes, refilled from the work list (all of it, or what the sets left over) ----
            bool exhausted = false;   // wave-uniform: the work list has been handed out
            if (c_end != 0u && !coop) for (;;) {
                // ---- refill: idle lanes claim the next entries (one LDS atomic per wave) ----
                const unsigned long long idle = __ballot(!have);
                if (idle != 0ull && !exhausted) {
                    const uint32_t nidle = (uint32_t)__popcll(idle);
                    uint32_t base = 0u;
                    if (lane == 0) base = atomicAdd(s_ctr, nidle);
                    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                    exhausted = base + nidle >= c_end;
                    if (!have) {
                        const uint32_t wq = base + (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
                        const uint32_t wn = SETS ? (wq < c_end ? s_ovf[wq] : c1) : wq;
                        const uint32_t cc = wn & ((1u << cg_shift) - 1u), il = i_lo + (wn >> cg_shift);
                        uint4 e = make_uint4(0u, 0u, 0u, 0u);
                        float4 mp = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (wn < c1 && il < s_n[cc]) { e = flist[entry_at(wn)]; if (e.w & kWantMask) mp = kmp[e.x]; }
                        if (e.w & kWantMask) {
                            w = wn;
                            make_queries(cc, e, mp);
                            lane_nn_begin(IBA_LANE_NN_PASS);
                            if (SETS) seed(wn);
                            have = true;
                        }
                    }
                }
                if (__ballot(have) == 0ull) { if (exhausted) break; continue; }   // a claim of 64 entries none of which wants a search: claim again
                if (have && dbg == 4) { have = false; continue; }
                if (have) lane_nn_visit<WHICH>(IBA_LANE_NN_PASS, s_nodes, p4, perm_g, P, D);
                if (have && go < 0) {
                    if (dbg != 5) finish(w, nullptr, nullptr);
                    have = false;
                }
            }
        }
        __syncthreads();
        NN_TICK(5);   // left-over tree searches (and the barriers around them)
        if (dbg == 3) return;
        // ---- fixed-order sums of the chunk: 32 threads per candidate, each over a strided subset of its entries ----
        if (WHICH & 2) {
            const uint32_t cc = (uint32_t)tid >> 5, q = (uint32_t)tid & 31u;
            if (cc < (uint32_t)cands_here) {
                for (uint32_t wi = c0 + cc + (q << cg_shift); wi < c1; wi += 32u << cg_shift) {   // c0 is a multiple of CG
                    const double r = s_res[wi - c0];
                    if (r == r) {
                        const double dist = fabs(r);
                        const bool is_pt = __double_as_longlong(r) < 0;
                        ++fin_c;
                        if (dist < prm.corr_3d_3d_threshold) { fin_sum += dist; ++fin_v; fin_pl += is_pt ? 0u : 1u; fin_pt += is_pt ? 1u : 0u; }
                    }
                }
            }
        }
        __syncthreads();   // s_res is rewritten by the next chunk
    }

    // ---- per-candidate totals: butterfly over the 32 threads of a candidate (fixed order) ----
    {
        unsigned long long cnt = (unsigned long long)fin_c | ((unsigned long long)fin_v << 32);
        unsigned long long cnt2 = (unsigned long long)fin_pl | ((unsigned long long)fin_pt << 32);
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) { fin_sum += __shfl_xor(fin_sum, off); cnt += __shfl_xor(cnt, off); cnt2 += __shfl_xor(cnt2, off); }
        const int cc = tid >> 5, q = tid & 31;
        if (cc < cands_here && q < kNNPartial) {
            double out = 0.0;
            if (q == 0) out = fin_sum;
            else if (q == 1) out = (double)(cnt & 0xffffffffull);
            else if (q == 2) out = (double)(cnt >> 32);
            else if (q == 3) out = (double)(cnt2 & 0xffffffffull);
            else if (q == 4) out = (double)(cnt2 >> 32);
            else if (q == 5 && cc == 0) out = (double)left_to_tree;
            nn_partials[((size_t)(g * CG + cc) * nn_nrec + (size_t)f * NS + sl) * kNNPartial + q] = out;
        }
    }
#ifdef IBA_DIAG_COUNTERS
    NN_TICK(6);   // sums + records
    if (tid == 0 && dp.diag) {
        unsigned long long* d64 = (unsigned long long*)(dp.diag + 8);
        for (int i = 0; i < 7; ++i) atomicAdd(d64 + i, nn_t[i]);
        atomicAdd(d64 + 7, 1ull);   // blocks that got here
    }
#endif
#undef dp
#undef prm
#undef lay
}


// ------------------------------------------------------------------------------------------------------------------
// iba_fit_kernel<SLOTS, STAGE> (plane_cache = 0): the planes one evaluation needs, fitted for exactly that evaluation as the
// reference does (ComputeLocalNeighbor / ComputeLocalNormalSingleThre, pointcloud.h:699-760; ComputeAlignmentDist,
// iba_global.cpp:125-153), and the decisions that hang on them.
//   STAGE 1 (after iba_assoc_kernel): local plane at the matched scan point of every association entry -> .y (the plane
//            factor's point, or none) and kFlagA (no neighbourhood: no residual block at all) of the entry.
//   STAGE 2 (after the searches): local plane at the association path's neighbour -> kind bit of .z; cost plane at the cost
//            path's neighbour -> the entry's cost distance into frefit[..].x (NaN: no cost term), which the search kernel's
//            second launch sums. When both planes use the same radius and size and the neighbour is the same point, it is
//            one fit.
// Plane records a residual block will read (iba_factor_kernel, iba_residual_kernel) go to the candidate's private scratch
// slot. grid: (ceil(list stride / 64), frames, B) workgroups of one wave: 64 consecutive list entries; fits are packed four
// to a round (fit_list_rows), then finished one per lane (fit_finish_lane).
// ------------------------------------------------------------------------------------------------------------------
template <int SLOTS, int STAGE>
__global__ __launch_bounds__(64) void iba_fit_kernel(DevProblem dp, DevParams prm, uint4* __restrict__ flist, double4* __restrict__ frefit, const uint32_t* __restrict__ lcount,
                                                     int flist_stride, int slot_base, int want) {
    __shared__ FitLds<SLOTS> s_fit;
    __shared__ uint32_t s_job[64];
    const int nf = dp.n_frames, f = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const uint32_t n = lcount[(size_t)b * nf + f];
    if (blockIdx.x * 64u >= n) return;
    const FrameHdr& h = dp.frames[f];
    const uint32_t i = blockIdx.x * 64u + (uint32_t)lane;
    const size_t at = ((size_t)b * nf + f) * (size_t)flist_stride + i;
    const float4* p4 = dp.pts4 + h.pt_base;
    const TreeNode* nodes = dp.nodes + h.node_base;
    PlaneRec* scratch = dp.scratch_local + (size_t)(slot_base + b) * (size_t)dp.n_pt_total + h.pt_base;
    const bool live = i < n;
    uint4 e = make_uint4(0u, kNone, kNone, 0u);
    if (live) e = flist[at];
    const double r2l = prm.neigh_radius2, r2c = prm.norm_radius2;
    const bool same = (r2l == r2c && prm.neigh_max_pts == prm.norm_max_pts);
    bool needA, needC = false; uint32_t posA, posC = kNone;
    double ex = 0, ey = 0, ez = 0;
    if (STAGE == 1) { needA = live && (e.w & kFlagA) && e.y != kNone; posA = e.y; }
    else {
        needA = live && (want & 1) && (e.w & kFlagA) && e.z != kNone; posA = e.z;
        if (live && (want & 2) && (e.w & kFlagC)) {
            const double4 q = frefit[at];
            ex = q.x; ey = q.y; ez = q.z; posC = (uint32_t)__double_as_longlong(q.w);
            needC = prm.use_plane != 0;
        }
    }
    // pass 0: local-plane parameters; pass 1: cost-plane parameters, for the cost planes pass 0 did not produce
    PlaneRec rec0, rec1;
    rec0.k = 0; rec1.k = 0;
    for (int pass = 0; pass < (STAGE == 1 ? 1 : 2); ++pass) {
        const bool need = pass == 0 ? (needA || (same && needC)) : (needC && !(same && (!needA || posA == posC)));
        const uint32_t pos = pass == 0 ? (needA ? posA : posC) : posC;
        const unsigned long long nm = __ballot(need);
        if (nm == 0ull) continue;
        const int c = __popcll(nm & ((1ull << lane) - 1ull)), njobs = __popcll(nm);
        if (need) s_job[c] = pos;
        __syncthreads();
        const double r2 = pass == 0 ? r2l : r2c; const int max_pts = pass == 0 ? prm.neigh_max_pts : prm.norm_max_pts;
        for (int r = 0; 4 * r < njobs; ++r) {
            const int j = 4 * r + (lane >> 4);
            fit_list_rows<SLOTS>(p4, nodes, h.P, h.depth, j < njobs ? s_job[j] : kNone, r2, max_pts, s_fit, 4 * r);
        }
        __syncthreads();
        if (need) {
            const PlaneRec rec = fit_finish_lane(p4, pos, s_fit.list[c], s_fit.count[c], s_fit.far_d2[c]);
            if (pass == 0) rec0 = rec; else rec1 = rec;
        }
        __syncthreads();   // the lists are rewritten by the next pass
    }
    if (!live) return;
    if (STAGE == 1) {
        if (needA) {
            const bool neigh_ok = local_neigh_ok(prm, rec0);                           // pointcloud.h:752
            const bool plane_ok = neigh_ok && local_plane_ok(prm, rec0);              // bvalid_plane (iba_local.cpp:231)
            e.y = (plane_ok || prm.p2pix) ? posA : kNone;   // (IBATestEdge mode: the edge needs no plane)
            if (!neigh_ok) e.w &= ~kFlagA;   // no 3d-3d block either (the `continue` at iba_local.cpp:209-211)
            flist[at] = e;
            if (plane_ok) scratch[posA] = rec0;
        }
    } else {
        if (needA) {
            const bool state = local_neigh_ok(prm, rec0) && local_plane_ok(prm, rec0);   // pointcloud.h:699-717
            flist[at].z = posA | (state ? 0x80000000u : 0u);
            if (state) scratch[posA] = rec0;
        }
        if (want & 2) {
            double res = NAN;
            if (e.w & kFlagC) {
                const bool from0 = same && (!needA || posA == posC);
                PlaneRec rc;
                rc.nx = from0 ? rec0.nx : rec1.nx; rc.ny = from0 ? rec0.ny : rec1.ny; rc.nz = from0 ? rec0.nz : rec1.nz;
                rc.reg_sum = from0 ? rec0.reg_sum : rec1.reg_sum; rc.far_d2 = from0 ? rec0.far_d2 : rec1.far_d2; rc.k = from0 ? rec0.k : rec1.k; rc.pad = 0;
                res = cost_res(prm, needC, rc, ex, ey, ez);
            }
            frefit[at].x = res;
        }
    }
}

// sums the per-frame records of each candidate in a fixed order, plus the search kernel's own (narrow) records.
// grid: B blocks of kReduceThreads threads: 16 record groups x 64 slots for the wide records; 128 record lanes x 8 slots
// for the narrow ones.
// the association's two questions to the memoised local plane of every scan point (local_neigh_ok / local_plane_ok), answered once per
// parameter set: the tail of the association kernels gathers one byte per matched point
__global__ __launch_bounds__(256) void iba_verdict_kernel(const PlaneRec* __restrict__ planes_local, DevParams prm, uint8_t* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const PlaneRec rec = planes_local[i];
    const bool neigh_ok = local_neigh_ok(prm, rec);
    out[i] = (uint8_t)((neigh_ok ? 1u : 0u) | (neigh_ok && local_plane_ok(prm, rec) ? 2u : 0u));
}

// K7 rides here (round 5): the hand-eye term of every counted (candidate, frame) — iba_global.cpp:264-276, two lanes per term as in
// rounds 3-4's staging launch — is evaluated by this block right before it is summed, from the device copy of the candidates. The
// sum over the frames keeps its place and its order in the record sums (slot P_HE_SUM of the association's records, which now hold
// the COUNT alone): the same numbers added in the same order as when the association kernel wrote them into its records.
constexpr int kHeLds = 2048;   // HE_MODE 1: frames whose terms wait in LDS (more: through `he_scratch`, read back past the L1)
// HE_MODE 0: an evaluation without the cost tuple; 2: the search kernel's spare blocks have left the terms in he_scratch[b][f] (the usual
// chain); 1: no search kernel ran in this chain — the terms are evaluated here, two lanes each
template <int HE_MODE>
__global__ __launch_bounds__(kReduceThreads) void iba_reduce2_kernel(const double* __restrict__ frame_partials, int nrec, int nfr, int n_fact, const double* __restrict__ nn_partials, int nn_nrec,
                                                                     double* __restrict__ out, const FrameHdr* __restrict__ frames, const Cand* __restrict__ cands, double* __restrict__ he_scratch,
                                                                     unsigned long long* __restrict__ done_flag, uint32_t* __restrict__ done_ctr, unsigned long long done_seq) {
    constexpr int NG = kReduceThreads / kPartialStride;
    constexpr int NL = kReduceThreads / kNNPartial;
    __shared__ double s[NG][kPartialStride];
    __shared__ double s2[NL][kNNPartial];
    __shared__ double s_he[HE_MODE == 1 ? kHeLds : 1];
    const int b = blockIdx.x, i = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // (a wave is one record group: its ranges and base addresses are scalars)
    const double* src = frame_partials + (size_t)b * nrec * kPartialStride;
    const bool he_lds = HE_MODE == 1 && nfr <= kHeLds;
    double* he_g = he_scratch + (size_t)b * (size_t)nfr;
    if (HE_MODE == 1) {
        const Cand& cd = cands[b];
        for (int t0 = 0; t0 < 2 * nfr; t0 += kReduceThreads) {   // (block-uniform trip count: he_term pairs lanes)
            const int t = t0 + (int)threadIdx.x, f = t >> 1;
            const bool live = f < nfr && src[(size_t)(f < nfr ? f : 0) * kPartialStride + P_HE_CNT] != 0.0;   // the frame counts for this candidate (corrset test passed) and has a next keyframe
            const double v = he_term(frames[f < nfr ? f : 0], cd, t & 1, live);
            if (f < nfr && !(t & 1)) { if (he_lds) s_he[f] = v; else __hip_atomic_store(he_g + f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        __syncthreads();
    }
    // the first nfr records are the association's (one per frame), the others — when the evaluation has them — the factor kernel's:
    // a group sums the same frame range of either half, so that a slot only one half fills (every cost slot, every H / b slot) gets
    // the same bits whether or not the other half exists (iba_eval_cost = the cost tuple of iba_eval_full, bit for bit)
    const int per = (nfr + NG - 1) / NG, f0 = g * per, f1 = min(nfr, f0 + per);
    // slot P_HE_SUM of an association record is empty; its lane sums the hand-eye terms of the counted frames instead (the count is the
    // neighbouring lane's own value: one shuffle; the term one more load of that lane alone — straight-line code, no divergent loop)
    const bool he_lane = HE_MODE != 0 && i == P_HE_SUM;
    // (round 6) every load of a thread — its share of the search kernel's records, of the association's and of the factor kernel's — is issued before
    // the first addition: the kernel was a chain of nine dependent round trips to records the previous kernels had just written (13 + 13 keyframes in
    // steps of four, eleven search records one by one), 15.7 us at 64 x 200 keyframes. Loads past the end of a range re-read its last record
    // (an unconditional load: one under a condition is waited for where its value merges with the default) and are left out of the sum; the
    // additions keep the order of rounds 2-5.
    // A range's last step reads the CH records that END at the range's end (one base address, immediate offsets) and leaves out the ones below
    // its start; the additions keep the order of rounds 2-5.
    constexpr int CH = HE_MODE == 1 ? 8 : 16, CN = 12;
    double vn[CN];
    const int qn = threadIdx.x & (kNNPartial - 1), rl = threadIdx.x / kNNPartial;
    const double* ns = nn_partials ? nn_partials + (size_t)b * nn_nrec * kNNPartial : nullptr;
    const int n_mine = ns && rl < nn_nrec ? (nn_nrec - rl + NL - 1) / NL : 0;   // records rl, rl + NL, ... of slot qn
    if (n_mine > 0) {
#pragma unroll
        for (int k = 0; k < CN; ++k) vn[k] = ns[(size_t)(rl + (k < n_mine ? k : 0) * NL) * kNNPartial + qn];   // (past the lane's last record: its first one again, left out below)
    }
    const double* srcB = src + (size_t)nfr * kPartialStride;
    const int nB = n_fact > 0 ? min(nfr, n_fact) : 0;   // (n_fact <= nfr records of the factor kernel: one per keyframe, or one per range of iba_factor2_kernel)
    const int hiB = n_fact > 0 ? min(f1, n_fact) : f0;
    double x = 0, xb = 0;
    auto he_fix = [&](double v, int f) -> double {   // slot P_HE_SUM of an association record is empty; its lane sums the hand-eye term of every counted frame instead (the count is the neighbouring lane's own value)
        if (HE_MODE == 0) return v;
        double hv = 0.0;
        if (HE_MODE == 2) hv = he_g[f];
        else if (he_lane) hv = he_lds ? s_he[f] : __hip_atomic_load(he_g + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double c = __shfl(v, P_HE_CNT);
        return he_lane ? (c != 0.0 ? hv : 0.0) : v;
    };
    if (nfr >= CH) {
        for (int f = f0; f < f1; f += CH) {
            double va[CH], vb[CH], hv[CH];
            const int wa = min(f, nfr - CH);
            const double* pa = src + (size_t)wa * kPartialStride + i;
#pragma unroll
            for (int q = 0; q < CH; ++q) va[q] = pa[(size_t)q * kPartialStride];
            if (HE_MODE == 2) {
#pragma unroll
                for (int q = 0; q < CH; ++q) hv[q] = he_g[wa + q];   // (left in he_scratch by the search kernel's spare blocks; one address per wave: scalar loads)
            }
            const bool any_b = f < hiB && nB >= CH;
            const int wb = min(f, nB - CH);
            if (any_b) {
                const double* pb = srcB + (size_t)wb * kPartialStride + i;
#pragma unroll
                for (int q = 0; q < CH; ++q) vb[q] = pb[(size_t)q * kPartialStride];
            }
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                double v = va[q];
                if (HE_MODE == 2) { const double c = __shfl(v, P_HE_CNT); if (he_lane) v = c != 0.0 ? hv[q] : 0.0; }
                else if (HE_MODE == 1) v = he_fix(v, wa + q);
                if (wa + q >= f && wa + q < f1) x += v;
            }
            if (any_b) {
#pragma unroll
                for (int q = 0; q < CH; ++q) if (wb + q >= f && wb + q < hiB) xb += vb[q];
            } else {
                for (int fb = f; fb < min(f + CH, hiB); ++fb) xb += srcB[(size_t)fb * kPartialStride + i];
            }
        }
    } else {
        for (int f = f0; f < f1; ++f) x += he_fix(src[(size_t)f * kPartialStride + i], f);
        for (int f = f0; f < hiB; ++f) xb += srcB[(size_t)f * kPartialStride + i];
    }
    if (n_fact > 0) x += xb;
    s[g][i] = x;
    if (nn_partials) {
        double y = 0;
        asm volatile("" : "+v"(y));   // (the first addition, 0 + vn[0], was scheduled right behind its load — a wait before the other loads were issued)
        if (n_mine > 0) {
#pragma unroll
            for (int k = 0; k < CN; ++k) if (k < n_mine) y += vn[k];
            for (int k = CN; k < n_mine; ++k) y += ns[(size_t)(rl + k * NL) * kNNPartial + qn];
        }
        s2[rl][qn] = y;
    }
    __syncthreads();
    if (g == 0) {
        double t = s[0][i];
#pragma unroll
        for (int qq = 1; qq < NG; ++qq) t += s[qq][i];
        // the search kernel's slot q belongs to partial slot nn_slot[q]
        int q = -1;
        if (i == P_SUM_3D3D) q = 0; else if (i == P_CNT_3D3D) q = 1; else if (i == P_VALID_3D3D) q = 2; else if (i == P_VALID_PL) q = 3; else if (i == P_VALID_PT) q = 4;
        if (q >= 0 && nn_partials) {
            double y = 0;
            for (int rl = 0; rl < NL; ++rl) y += s2[rl][q];
            t += y;
        }
        out[(size_t)b * kPartialStride + i] = t;
        if (done_flag) __threadfence_system();   // (see below)
    }
    // END OF A BLOCKING CALL (round 5): `out` is pinned host memory and the caller polls `done_flag` (pinned, too) instead of the stream — the
    // stream's own completion signal reaches the host 3-4 us after the last wave has retired, a tenth of a one-candidate call. Every writer
    // fences its sums to system scope, the block's thread 0 counts the block in, and the last block of the grid publishes the call's sequence number.
    if (done_flag) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t prev = __hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (prev + 1u == gridDim.x) {
                __hip_atomic_store(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the next chain's sums run behind this kernel in stream order)
                __threadfence_system();
                __hip_atomic_store(done_flag, done_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

}  // namespace iba
