// C-ABI of the MI355X IBA evaluation path (include/iba_mi355x.h). Host side: flat-problem packing,
// static index builds, kernel launches, host finalisation. There is NO CPU compute fallback: every
// entry point that evaluates anything needs a HIP device and fails with IBA_ERR_NO_DEVICE otherwise.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/iba_mi355x.h"
#include "../../include/iba_mi355x_debug.h"
#include "iba_build.hpp"
#include "iba_host_math.hpp"
#include "iba_internal.hpp"
namespace iba {
const char* debug_env(const char* name) {   // (read at every handle creation: a test may switch it on after the library was loaded)
    const char* e = std::getenv("IBA_DEBUG_ENV");
    return (e && std::atoi(e) != 0) ? std::getenv(name) : nullptr;
}
}  // namespace iba
#include "iba_kernels.hpp"
#include "iba_lm.hpp"
#include "iba_mads.hpp"
#include "iba_mads_glue.hpp"
#include "iba_split_kernels.hpp"
#include "iba_factor2_kernel.hpp"
#include "iba_nn_list_kernel.hpp"
#include "iba_types.hpp"

using namespace iba;

namespace {
// Environment overrides are DEBUG aids (A/B runs of an unmodified caller, timing cuts, fault injection: include/iba_mi355x_debug.h): they are read only
// when IBA_DEBUG_ENV=1 is set as well, so that a stray variable in an integrator's environment cannot change launch shapes or kernels (round 6)
const char* dbg_env(const char* name) { return iba::debug_env(name); }

thread_local std::string g_create_error = "";

constexpr uint32_t kLdsBytes = 160u * 1024u;
constexpr int kRing = 4;

template <class T>
struct DevBuf {
    T* p = nullptr; size_t n = 0;
    hipError_t alloc(size_t count) { n = count; return hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)); }
    hipError_t upload(const std::vector<T>& v) {
        hipError_t e = alloc(v.size()); if (e != hipSuccess) return e;
        if (!v.empty()) e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct iba_handle {
    int device = 0;
    int n_frames = 0;          // owned frames
    int global_frames = 0;
    int frame_begin = 0;
    iba_params params{};
    DevParams dprm{};
    std::string err;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
    bool timing = false, timing_recorded = false, timing_split = false;
    float last_frame_ms = 0.f, last_total_ms = 0.f;
    int64_t n_points = 0, n_keypoints = 0;
    uint32_t maxP = 0, maxPpad = 0, maxK = 0, maxNodes = 0, maxBitmapWords = 0, maxCoarse = 0;
    // the evaluation chain: association kernel + search kernel (+ fit kernels with plane_cache = 0) (iba_split_kernels.hpp)
    LdsLayout alay{};                     // LDS plan of iba_assoc_kernel
    uint32_t maxKw = 0;                   // max over frames of the keypoints that can own a term (MapPoint or covisible match)
    int assoc_dbg = 0;
    bool factor_valu = true;              // IBA_FACTOR_MFMA=1 selects the matrix-core variant of the factor kernel (slower on gfx950: see iba_kernels.hpp)
    // iba_factor2_kernel (r06): one wave per equal share of a candidate's whole work list (iba_factor2_kernel.hpp)
    bool factor_v2 = false;               // IBA_FACTOR_V2=1 (debug): iba_factor2_kernel instead of the one-wave-per-(keyframe, candidate) kernel (measured: no faster, see the header)
    int factor_slots = 2048;              // wave slots of the device at two waves per SIMD: CUs x 8
    int factor_dbg = 0;                   // IBA_FACTOR_DBG (debug): timing cuts of iba_factor2_kernel (results invalid)
    int factor_waves_forced = 0;          // IBA_FACTOR_WAVES_PER_CAND (debug): ranges per candidate instead of the rule in factor_waves()
    F2Layout f2lay{};
    DevBuf<KpRec> d_kp_rec;               // the factor kernel's keypoint records (r06: one cache line per keypoint)
    DevBuf<ScanRec> d_scan_rec;           // ... and scan-point records (point + memoised local-plane normal), rebuilt by compute_plane_cache; empty when the allocation failed
    bool scan_rec_valid = false;          // d_scan_rec holds the normals of the current plane memo (plane_cache = 1)
    bool jets_fold = true;                // IBA_JETS_FOLD=0 (debug): the derivative half of a batch always by a copy launch of its own (rounds 3-5)
    bool factor_rec = true;               // IBA_FACTOR_REC=0 (debug): gather from the separate arrays as rounds 1-5 did
    DevBuf<double> d_ffr;                 // per keyframe: camera, pose, table offsets, relative poses of its covisible slots — one contiguous record (kFfrHead + 12 max_slots doubles)
    DevBuf<double2> d_kp_c;               // ((u - cx) / fx, (v - cy) / fy) of every keypoint: IBA_PlaneFactor's ray (IBACalib2.hpp:165), divided once here instead of per residual block
    int nn_dbg = 0;                       // IBA_NN_DBG: cut the search kernel short (timing attribution; results are garbage)
    bool nn_list = false;                 // IBA_NN_LIST=1 (opt-in, measured slower: see iba_nn_list_kernel.hpp): the anchored lists are walked by iba_nn_list_kernel's persistent grid instead of iba_nn_kernel's one block per slice
    int nn_list_workers = 0;              // IBA_NN_LIST_WORKERS: blocks of iba_nn_list_kernel per CU (0: what the LDS and 4 waves per SIMD allow)
    int n_cus = 256;                      // compute units of the device (MI355X: 256)
    int last_nn_threads = 0;              // diagnostic: threads per block of the last iba_nn_kernel launch
    int last_nn_list = 0;                 // diagnostic: workers per (XCD, group) of the last iba_nn_list_kernel launch (0: iba_nn_kernel ran)
    int nn_cg_max = kMaxGroup < 8 ? kMaxGroup : 8;   // candidates per search block (power of two <= kMaxGroup)
    int nn_small_min_b = 12;              // ... from this many candidates per launch (IBA_NN_SMALL_MIN_B)
    bool nn_small = false;                // the search kernel runs one-wave blocks (kNNThreadsSmall): decided at create from the size of the keyframes' kd trees; IBA_NN_SMALL=0 / 1 forces it
    bool nn_cg_fixed = false;             // IBA_NN_CG given: no adaptation to the batch size
    bool nn_rounds = true;                // IBA_NN_ROUNDS=0: the entries the anchored lists leave over are searched leaf by leaf (rounds 3-4) instead of in rounds of leaves
    DevBuf<uint32_t> d_lcount, d_lcount_frozen;   // work-list length per (candidate, frame)
    uint32_t max_slots = 0;               // covisible keyframes of the busiest frame
    uint32_t lstride = 1;                 // entries per (candidate, frame) row of the lists: no list is longer than maxKw
    int nn_ns = 1;                        // search blocks per (frame, candidate group): ceil(maxKw / kSliceW)
    DevBuf<double> d_nn_partials;         // part_cap * n_frames * nn_ns * kNNPartial (ensure_lists)
    hipEvent_t ev_mid = nullptr;
    float last_assoc_ms = 0.f, last_nn_ms = 0.f;
    // common pairs of a batch of nearby candidates (iba_pairs_kernel + iba_assoc2_kernel)
    LdsLayout alay2{};                    // LDS plan of iba_assoc2_kernel
    int assoc2_flreg_on = 1;              // IBA_ASSOC2_FLREG
    DevBuf<PairRec> d_pairs;              // n_frames x pair_cap
    DevBuf<uint32_t> d_hard, d_pcounts;   // n_frames x hard_cap; n_frames x kCountStride
    DevBuf<uint32_t> mpk;                 // per frame: keypoints that own a MapPoint
    DevBuf<uint2> fkp;                    // per frame: (keypoint, flag word) of the keypoints that can own a term (FrameHdr::fk_base, n_fk)
    uint32_t max_mpk = 0;
    DevBuf<SetPt> d_anchor;               // anchored neighbour lists: n_frames x maxK rows of kAnchorRowBytes (512) bytes
    // up to kAnchorSets anchors (an optimiser polls around two incumbents, the feasible and the infeasible one): each with its own set of lists
    bool anchor_valid[kAnchorSets] = {false, false}; AnchorRef anchor_ref[kAnchorSets]{}; int calls_since_anchor[kAnchorSets] = {0, 0}; unsigned long long anchor_used[kAnchorSets] = {0, 0};
    unsigned long long anchor_clock = 0; size_t anchor_set_elems = 0;
    uint8_t anchor_sel[kMaxChain] = {0};       // this call: the set each candidate reads (255: none)
    double anchor_reach = 0.06;           // IBA_ANCHOR_REACH (m): a batch whose reference candidate moves a nominal MapPoint further than this from the anchor's query gets a new anchor
    int anchor_builds = 0;
    // list slots of the common pairs: each holds the pair lists of ONE group of candidates (reference + bound) and may outlive the call
    struct PairSlot { GroupRef ref{}; bool valid = false; unsigned epoch = 0; unsigned long long last_use = 0; };
    PairSlot pslot[kMaxPairGroups];
    unsigned long long pair_clock = 0;
    PairsPlan pplan{}; int n_build = 0;   // this call's pair search: the groups that need new lists
    Assoc2Map amap{};                     // this call's candidate -> slot map and the slots' current counter sets
    int n_groups = 0;                     // groups of this call's batch (1: the whole batch shares one search)
    int max_groups = kMaxPairGroups;      // IBA_PAIR_GROUPS: 1 = no clustering of wide batches (round 3's behaviour)
    int last_mean_pairs_slot = -1;
    int pair_cap = 0, hard_cap = 0;
    int last_assoc2_threads = 0;          // block size of the last iba_assoc2_kernel launch (iba_debug_last_assoc2_threads)
    int assoc2_threads_forced = 0;        // IBA_ASSOC2_THREADS: 256 / 512 (0: chosen per launch, assoc2_threads)
    int assoc2_small_min_blocks = 1024;   // launches of at least this many (candidate, keyframe) blocks run iba_assoc2_kernel with 256 threads per block (IBA_ASSOC2_SMALL_MIN)
    uint32_t pairs_dense_min = 32768u;    // scans of at least this many points: the pair search tests a block's boxes before it loads the block's points and the keypoint grid (IBA_PAIRS_DENSE_MIN)
    int common_mode = 1;                  // IBA_COMMON_PAIRS: 0 = never, 1 = when the batch is tight (default), 2 = whenever the bound allows
    bool spin_wait = true;                // IBA_SPIN_WAIT=0: blocking waits only
    bool nn_sets = true;                  // IBA_NN_SETS=0: no anchored neighbour lists, every lane searches the tree (diagnostic)
    int pair_bound = 1;                   // IBA_PAIR_BOUND: 0 = the pair search bounds the batch's motion entrywise only (diagnostic)
    int common_min_batch = 1;             // IBA_COMMON_MIN_BATCH
    double common_max_px = 20.0;          // IBA_COMMON_MAX_PX: nominal spread of the batch's projections beyond which every candidate searches for itself
    double max_fx = 0.0;
    const Cand* last_hc = nullptr;        // host copy of the candidate block staged last (pinned ring)
    bool cref_ok = false;                 // this call's batch shares pair searches (plan_pairs at staging time)
    bool plan_wide = false;               // ... or was found wide everywhere (an exploratory poll: every candidate for itself)
    int pair_memo = 1; double pair_infl = 1.25, pair_rho_floor = 1e-4, pair_tau_floor = 1e-3;   // IBA_PAIR_MEMO, IBA_PAIR_INFL
    int pairs_builds = 0; int pair_memo_max_b = 40; double pair_memo_max_px = 8.0;   // IBA_PAIR_MEMO_MAX_B
    const double* jets_x = nullptr; int jets_B = 0, jets_slot = 0;   // candidates whose derivative half is still to be computed (finish_jets)
    int last_nn_nrec = 0, last_nn_B = 0;   // shape of the search kernel's records of the last evaluation (iba_debug_nn_left_to_tree)
    const Cand* jets_src = nullptr; const std::atomic<int>* jets_flag = nullptr;   // ... or is being computed by the group's calling thread: the block to copy once *jets_flag is set
    int last_path = 0;                    // 1: the last evaluation chain used the common pairs

    DevBuf<FrameHdr> frames; DevBuf<SlotHdr> slots;
    DevBuf<float> xs, ys, zs, chunk_box; DevBuf<uint32_t> perm, inv_perm; DevBuf<TreeNode> nodes;
    DevBuf<float4> pts4;   // the same scan points as (x, y, z, original index bits): one 16 B gather per point where lanes diverge
    DevBuf<uint32_t> d_diag;              // diagnostic counters (DevProblem::diag)
    DevBuf<float2> kp_uv; DevBuf<float4> kp_mp; DevBuf<uint32_t> kp_fl, kp_fl2;   // kp_fl2: match bits of the covisible slots 30..61 (allocated only when a frame has that many)
    DevBuf<uint32_t> coarse_start, bitmap; DevBuf<float4> crec;
    DevBuf<float2> match_uv;
    DevBuf<PlaneRec> plane_cost, plane_local;
    DevBuf<uint8_t> plane_ok;             // the association's verdicts on plane_local per scan point (iba_verdict_kernel), rebuilt with every parameter set
    DevBuf<double4> d_frefit;   // plane_cache = 0: per list entry, query offset + neighbour of the cost path between the search and the fit kernel
    DevBuf<PlaneRec> scratch_cost, scratch_local;   // plane_cache = 0: (scratch_cap + 1) x n_pt_total records
    int scratch_cap = -1; bool scratch_local_aliases = false; int64_t n_pt_total = 0;
    bool plane_local_aliases_cost = false;
    double plane_cost_r2 = -1, plane_local_r2 = -1; int plane_cost_max = -1, plane_local_max = -1;
    DevBuf<Cand> d_cands;                 // kRing * chain_cap
    DevBuf<double> d_frame_partials;      // part_cap * nrec * kPartialStride (ensure_lists)
    DevBuf<double> d_partials;            // chain_cap * kPartialStride
    DevBuf<uint32_t> d_corr;              // n_keypoints
    DevBuf<double> d_he;                  // part_cap * n_frames: hand-eye terms of handles with more than kHeLds frames (iba_reduce2_kernel)
    int assoc_cap = 0;
    DevBuf<uint2> d_assoc_frozen;         // n_keypoints (iba_build_problem)
    DevBuf<uint4> d_flist, d_flist_frozen;       // dense residual-block lists: [cand][frame][maxK] / [frame][maxK]
    DevBuf<uint32_t> d_fcount, d_fcount_frozen;  // entries per (cand, frame)
    bool frozen_valid = false; int32_t frozen_frames = 0, frozen_ncorr = 0;
    int nfb = 0;                          // factor-kernel records per candidate (= n_frames)
    int nrec = 0;                         // partial records per candidate = n_frames + nfb
    Cand* h_cands = nullptr;              // pinned, kRing * chain_cap
    double* h_partials = nullptr;         // pinned
    double* h_partials_dev = nullptr;     // the same buffer as the kernels see it: the last kernel of a chain writes the sums there (no D2H copy)
    unsigned long long* h_done = nullptr; unsigned long long* h_done_dev = nullptr;   // pinned: the sequence number of the last blocking call whose sums have landed (iba_reduce2_kernel)
    DevBuf<uint32_t> d_done_ctr;          // blocks of the summing kernel that have finished (reset by the last one)
    unsigned long long done_seq = 0; bool done_armed = false;   // the call in flight publishes done_seq; done_flag (IBA_DONE_FLAG=0: poll the stream as rounds 3-4 did)
    bool done_flag_on = true;
    Cand* h_cands_dev = nullptr;          // the pinned candidate ring as the fetch kernel sees it
    hipEvent_t ring_ev[kRing] = {nullptr, nullptr, nullptr, nullptr};
    // The head of a chain (round 5): the candidate block reaches the device through spare blocks of the chain's FIRST kernel (the pair
    // search when one runs, else the association kernel, whose own blocks then read their candidate from the pinned ring), the hand-eye
    // terms are evaluated by the summing kernel: no staging launch, no second stream, no event between kernels. chain_fold = 0: a
    // staging launch (iba_fetch_kernel) at the head of every chain.
    int chain_fold = 1;                   // iba_create_options.chain_fold / IBA_CHAIN_FOLD
    int chain_cap = kMaxChain;            // candidates one launch chain takes (iba_create_options.max_chain_batch / IBA_MAX_CHAIN)
    bool head_pending = false; int head_slot = -1, head_B = 0;   // this call's candidate block still waits in the pinned ring for the first kernel of run_split
    int part_cap = 0;                     // candidates the per-candidate record buffers hold (ensure_lists)
    bool ring_used[kRing] = {false, false, false, false};
    int ring_next = 0;
    std::vector<FrameHdr> h_frames;
    std::vector<uint64_t> h_kp_off;       // local frame -> kp offset (K+1)
    std::vector<uint32_t> h_kp_ext;       // internal (Morton) keypoint id -> reference keypoint id
    std::vector<float> h_kp_uv;           // (u, v) of every keypoint in internal order: the reject bitmap is rebuilt from it when max_pixel_dist changes

    DevProblem dev_problem() const {
        DevProblem dp{};
        dp.frames = frames.p; dp.slots = slots.p; dp.xs = xs.p; dp.ys = ys.p; dp.zs = zs.p; dp.perm = perm.p; dp.inv_perm = inv_perm.p; dp.chunk_box = chunk_box.p; dp.pts4 = pts4.p;
        dp.nodes = nodes.p; dp.kp_uv = kp_uv.p; dp.kp_mp = kp_mp.p; dp.kp_fl = kp_fl.p; dp.coarse_start = coarse_start.p; dp.crec = crec.p;
        dp.bitmap = bitmap.p; dp.match_uv = match_uv.p; dp.plane_cost = plane_cost.p;
        dp.kp_rec = d_kp_rec.p; dp.scan_rec = (scan_rec_valid && factor_rec) ? d_scan_rec.p : nullptr;
        dp.plane_local = plane_local_aliases_cost ? plane_cost.p : plane_local.p; dp.plane_ok = plane_ok.p; dp.n_frames = n_frames; dp.n_kp_total = n_keypoints;
        dp.scratch_cost = scratch_cost.p; dp.scratch_local = scratch_local_aliases ? scratch_cost.p : scratch_local.p; dp.n_pt_total = n_pt_total; dp.scratch_slot_base = 1;
        dp.mpk = mpk.p; dp.max_k = std::max(maxK, 1u); dp.kp_fl2 = kp_fl2.p; dp.diag = d_diag.p; dp.fkp = fkp.p;
        return dp;
    }
};

namespace {

#define HIP_TRY(h, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) {                                                                            \
            (h)->err = std::string(#expr) + ": " + hipGetErrorString(_e);                                  \
            return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? IBA_ERR_NO_DEVICE : IBA_ERR_HIP; \
        }                                                                                                  \
    } while (0)

iba_status fail(iba_handle* h, iba_status s, const std::string& msg) { if (h) h->err = msg; else g_create_error = msg; return s; }

void to_dev_params(const iba_params& p, DevParams& d) {
    d.gate2 = p.max_pixel_dist * p.max_pixel_dist;
    d.grid_margin = p.max_pixel_dist + 0.01;
    d.bitmap_margin = p.max_pixel_dist + 0.45;
    d.num_min_corr_cost = p.num_min_corr_cost;
    d.corr_3d_2d_threshold = p.corr_3d_2d_threshold; d.corr_3d_3d_threshold = p.corr_3d_3d_threshold;
    d.norm_max_pts = p.norm_max_pts; d.norm_min_pts = p.norm_min_pts;
    d.norm_radius2 = p.norm_radius * p.norm_radius; d.norm_reg_threshold = p.norm_reg_threshold; d.min_diff_dist2 = p.min_diff_dist * p.min_diff_dist;
    d.use_plane = p.use_plane; d.use_3d3d = p.err_weight[1] > 1e-10 ? 1 : 0;
    d.num_min_corr = p.num_min_corr; d.max_3d_dist2 = p.max_3d_dist * p.max_3d_dist; d.neigh_radius2 = p.neigh_radius * p.neigh_radius;
    d.neigh_max_pts = p.neigh_max_pts; d.neigh_min_pts = p.neigh_min_pts;
    d.local_min_diff_dist2 = p.local_min_diff_dist * p.local_min_diff_dist; d.local_norm_reg_threshold = p.local_norm_reg_threshold;
    d.robust_kernel_delta = p.robust_kernel_delta; d.robust_kernel_3ddelta = p.robust_kernel_3ddelta; d.plane_cache = p.plane_cache;
    d.p2pix = p.factor_3d2d_kind == 1 ? 1 : 0;
}

iba_status check_params(iba_handle* h, const iba_params& p) {
    if (!(p.max_pixel_dist > 0) || p.max_pixel_dist > 64.0)
        return fail(h, IBA_ERR_UNSUPPORTED, "max_pixel_dist must be in (0, 64] px");
    if (p.norm_max_pts < 1 || p.norm_max_pts > 64 || p.neigh_max_pts < 1 || p.neigh_max_pts > 64)
        return fail(h, IBA_ERR_UNSUPPORTED, "norm_max_pts / neigh_max_pts must be in [1, 64] (the neighbour list lives one entry per lane of a wave)");
    if (p.factor_3d2d_kind != 0 && p.factor_3d2d_kind != 1) return fail(h, IBA_ERR_INVALID_ARG, "factor_3d2d_kind must be 0 (IBA_PlaneFactor) or 1 (IBATestEdge)");
    return IBA_OK;
}

template <class F>
void parallel_for(int n, F fn) {
    const int nt = std::max(1, std::min<int>((int)std::thread::hardware_concurrency(), n));
    if (nt <= 1) { for (int i = 0; i < n; ++i) fn(i); return; }
    std::atomic<int> next(0);
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t) th.emplace_back([&]() { for (int i; (i = next.fetch_add(1)) < n;) fn(i); });
    for (auto& t : th) t.join();
}

uint32_t align_up(uint32_t v, uint32_t a) { return (v + a - 1) / a * a; }

// LDS plan of iba_assoc_kernel: per-keypoint tables, the candidate queue (4 B per queued point), and the reject bitmap, whose
// storage — dead after the streaming pass — becomes the head of the pair list (16 B per (point, keypoint) pair). What is left
// of half of the LDS (two blocks per CU) is split between queue and pair list; a full queue or list only costs speed (inline
// exact tests + a rescan), never correctness.
bool layout_assoc(iba_handle* h, LdsLayout& L) {
    const uint32_t rel_slots = std::max<uint32_t>(h->max_slots, 1u);   // (62 slots' worth for every handle took 3 KB from the queue and the pair list of handles with 3)
    const uint32_t red_bytes = 8u * kWaves * 4u + 8u * rel_slots * 12u + 4u * kWaves + 16u + 4u * (uint32_t)kHardLds;   // (+ the undecidable-depth points of a block)
    L = LdsLayout{};
    L.rel_slots = rel_slots;
    uint32_t off = 0;
    const uint32_t Kp = (std::max(h->maxK, 1u) + 3u) & ~3u;   // per-keypoint tables hold a multiple of 4 entries: the tail reads them 16 bytes at a time
    L.off_best_d2 = off; off += 8u * Kp;
    L.off_best_idx = off; off += 4u * Kp;
    off = align_up(off, 8); L.off_nodes = off;
    off = align_up(off, 16); L.off_red = off; off += red_bytes;
    L.off_vis = off; L.vis_words = 2u * kWaves * ((h->maxPpad / (uint32_t)kChunk + kThreads) / kThreads) + 2u;
    off += 4u * L.vis_words + 2u * (h->maxPpad / (uint32_t)kChunk + 2u);
    off = align_up(off, 4); L.off_cstart = off; off += 2u * std::max(h->maxCoarse, 1u);
    off = align_up(off, 16); L.off_kuv = off; off += 8u * Kp;
    L.off_kfl = off;   // (no LDS copy of the flags since round 5: the tail reads the flagged-keypoint list)
    off = align_up(off, 16);
    const uint32_t bm_bytes = align_up(4u * std::max(h->maxBitmapWords, 1u), 16u);
    if (off + bm_bytes + 4u * 512u > kLdsBytes) return false;
    const uint32_t want_q = std::max<uint32_t>(h->maxPpad / 2u, 512u) * 4u;      // nothing queues more than half of a scan
    const uint32_t want_p = std::max<uint32_t>(2u * h->maxK, 256u) * 16u;         // ~1 pair per matched keypoint, x 1.7 for the f32 margin
    int nb_max = 2;
    if (const char* e = dbg_env("IBA_ASSOC_BLOCKS")) nb_max = std::max(1, std::atoi(e));   // diagnostic
    uint32_t room = (kLdsBytes - off - bm_bytes) & ~15u;   // beyond the tables and the bitmap
    for (int nb = nb_max; nb >= 2; --nb) {
        const uint32_t share = (kLdsBytes / (uint32_t)nb) & ~255u;
        if (off + bm_bytes < share && share - off - bm_bytes >= std::max<uint32_t>(h->maxPpad / 16u, 512u) * 4u) { room = (share - off - bm_bytes) & ~15u; break; }
    }
    // the queue first (a scan queues ~8 % of its points at 2000 keypoints), the pair list gets the bitmap and the rest
    uint32_t bytes_q = std::min(want_q, std::max<uint32_t>(std::min<uint32_t>(room, std::max<uint32_t>(h->maxPpad / 8u, 1024u) * 4u), room / 2u) & ~15u);
    uint32_t bytes_p = std::min(want_p, bm_bytes + ((room - bytes_q) & ~15u));
    if (const char* e = dbg_env("IBA_CAND_BYTES")) bytes_q = std::min<uint32_t>((uint32_t)std::atoi(e) & ~15u, bytes_q);   // diagnostic
    if (const char* e = dbg_env("IBA_PAIR_BYTES")) bytes_p = std::min<uint32_t>((uint32_t)std::atoi(e) & ~15u, bytes_p);   // diagnostic
    L.off_cand = off; L.cand_cap = bytes_q / 4u; off += bytes_q;
    L.off_bitmap = off; L.off_pair = off; L.pair_cap = bytes_p / 16u; off += std::max(bm_bytes, bytes_p);
    L.total = off;
    return L.total <= kLdsBytes;
}

// LDS plan of iba_nn_kernel: kd nodes, counters, the candidates' transform constants and one result slot (8 B) per work entry
bool layout_nn(const iba_handle* h, NNLayout& L, const int max_group = kMaxGroup) {   // max_group: candidates per block of the block shape (kMaxGroup, or 2 for one-wave blocks)
    L = NNLayout{};
    uint32_t off = 0;
    L.off_nodes = off; off += 8u * std::max(h->maxNodes, 1u);
    off = align_up(off, 16); L.off_misc = off; off += 128u;
    off = align_up(off, 16); L.off_cd = off; off += 8u * (uint32_t)kCdDoubles * (uint32_t)max_group;
    off = align_up(off, 16); L.off_res = off; off += 8u * kSliceW * (uint32_t)max_group;
    L.off_ovf = off; off += 4u * kSliceW * (uint32_t)max_group;   // work entries left to the tree search when the batch's neighbour sets are in use
    L.total = off;
    off = align_up(off, 16); L.off_res2 = off;   // (iba_nn_list_kernel launches with total + one more result buffer)
    return L.total <= kLdsBytes;
}

// iba_assoc2_kernel<Q>: flagged keypoints per thread that the tail keeps in registers — 2 (at most 1024 flagged keypoints per frame), 4 (at most
// 2048), 0 = any number, read from the list where needed (IBA_ASSOC2_FLREG=0: always 0)
int assoc2_q(const iba_handle* h, int threads = kThreads) { return !h->assoc2_flreg_on ? 0 : (h->maxKw <= 2u * (uint32_t)threads ? 2 : (h->maxKw <= 4u * (uint32_t)threads ? 4 : 0)); }
// threads per block of iba_assoc2_kernel for a launch of `blocks` (candidate, keyframe) blocks: 256 once the launch fills the machine more than
// twice over (see the kernel), 512 below that (a one-candidate call's 200 blocks are latency, not throughput) and whenever 256 threads would
// push the tail off its register path. IBA_ASSOC2_THREADS=256 / 512 forces one.
int assoc2_threads(const iba_handle* h, int blocks) {
    if (h->assoc2_threads_forced) return h->assoc2_threads_forced;
    if (blocks < h->assoc2_small_min_blocks) return kThreads;
    return assoc2_q(h, 256) != 0 || assoc2_q(h, kThreads) == 0 ? 256 : kThreads;
}
// LDS plan of iba_assoc2_kernel: best d^2, best index and flag word per keypoint, the reduction slab
bool layout_assoc2(const iba_handle* h, LdsLayout& L) {
    const uint32_t rel_slots = std::max<uint32_t>(h->max_slots, 1u);
    const uint32_t red_bytes = 8u * kWaves * 4u + 8u * rel_slots * 12u + 4u * kWaves + 16u;
    L = LdsLayout{};
    L.rel_slots = rel_slots;
    uint32_t off = 0;
    const uint32_t Kp = (std::max(h->maxK, 1u) + 3u) & ~3u;
    L.off_best_d2 = off; off += 8u * Kp;
    L.off_best_idx = off; off += 4u * Kp;
    L.off_kfl = off;   // (no LDS copy of the flags since round 5: the tail reads the flagged-keypoint list)
    off = align_up(off, 16); L.off_red = off; off += red_bytes;
    // possible winners beyond the register window (1536 / 2048 pairs): a sparse scan's ~1.2 k pairs per keyframe never get there, and at 2000
    // keypoints the 3 KB decide whether a CU holds five or six blocks of 256 threads
    L.pair_cap = h->maxP >= 32768u ? (uint32_t)kPairNote : 512u;
    off = align_up(off, 16); L.off_pair = off; off += 2u * L.pair_cap;
    L.total = align_up(off, 16);
    return L.total <= kLdsBytes;
}

// The pair lists of a batch (see iba_pairs_kernel), their REUSE and the GROUPS of a wide batch. A list built around a reference
// candidate with the entrywise bound (rho, tau) holds every (scan point, keypoint) pair that any candidate within that bound of the
// reference can match — not only the candidates it was built for. The handle has kMaxPairGroups list slots. Planning a call:
//   1. the whole batch against the valid slots: inside the bound of one -> no pair search at all (an optimiser's late polls and
//      line searches hover around one point);
//   2. the whole batch as one group when its nominal projection spread (at a point 12 m out, 10 m deep) is at most common_max_px;
//   3. otherwise greedy clustering (farthest-point seeds, nearest seed wins) into up to max_groups groups, accepted when EVERY
//      group is that tight: a MADS batch is two polls, around the feasible and the infeasible incumbent, each tight, far apart;
//      each group is matched against the valid slots like a batch of its own, or gets new lists in the least recently used slot;
//   4. otherwise (a batch wide everywhere) every candidate searches for itself (iba_assoc_kernel): returns false.
// With IBA_PAIR_MEMO (default) the bound of a small (<= pair_memo_max_b), tight (<= pair_memo_max_px) group's new lists is its own
// times pair_infl plus a floor and the candidates' own per-block bound is left out (it is specific to the batch): such a slot
// stays valid for later calls. Sets h->pplan / h->n_build (the pair search of this call), h->amap, h->n_groups.
// is every member of the group inside the bound of a valid slot? (-1: no)
static int covering_slot(const iba_handle* h, const Cand* hc, const GroupPick& g) {
    for (int sl = 0; sl < kMaxPairGroups; ++sl) {
        const iba_handle::PairSlot& ps = h->pslot[sl];
        if (!ps.valid) continue;
        double rho[9], tau[3];
        if (!batch_spread(hc, g.n, ps.ref.R, ps.ref.t, rho, tau, nullptr, g.idx)) continue;
        bool fits = true;
        for (int i = 0; i < 9; ++i) fits = fits && rho[i] <= ps.ref.rho[i];
        for (int i = 0; i < 3; ++i) fits = fits && tau[i] <= ps.ref.tau[i];
        if (fits) return sl;
    }
    return -1;
}
inline bool plan_pairs(iba_handle* h, const Cand* hc, int B) {
    h->n_build = 0; h->n_groups = 0;
    const size_t set_words = (size_t)std::max(h->n_frames, 1) * kCountStride;
    auto cnt_off = [&](int sl, unsigned e) { return (uint32_t)(((size_t)sl * 2 + (e & 1u)) * set_words); };
    static thread_local GroupPick gp[kMaxPairGroups];
    int ng = 1;
    gp[0].n = B; for (int b = 0; b < B; ++b) gp[0].idx[b] = b;
    const bool memo_whole = h->pair_memo && B <= h->pair_memo_max_b;
    int whole_slot = -1;
    if (memo_whole && (whole_slot = covering_slot(h, hc, gp[0])) >= 0) {
        // (1) the lists of an earlier call cover this batch
        h->pslot[whole_slot].last_use = ++h->pair_clock;
        for (int b = 0; b < B; ++b) h->amap.slot[b] = (uint8_t)whole_slot;
        for (int sl = 0; sl < kMaxPairGroups; ++sl) h->amap.cnt_off[sl] = cnt_off(sl, h->pslot[sl].epoch - 1u);
        h->n_groups = 1; h->last_mean_pairs_slot = whole_slot;
        return true;
    }
    if (!pick_group(h->max_fx, hc, gp[0])) return false;
    const double max_px = h->common_mode >= 2 ? INFINITY : h->common_max_px;
    if (gp[0].px > max_px) {
        // (3) cluster (iba_pair_plan.hpp)
        ng = cluster_batch(h->max_fx, hc, B, max_px, h->max_groups, gp);
        if (ng == 0) return false;   // (4) wide everywhere
    }
    // (2) / (3): every group reuses a covering slot or gets new lists in the least recently used slot this call does not use
    bool taken[kMaxPairGroups] = {false, false, false, false};
    int slot_of[kMaxPairGroups];
    for (int g = 0; g < ng; ++g) {
        slot_of[g] = -1;
        if (ng > 1 && h->pair_memo && gp[g].n <= h->pair_memo_max_b) { const int sl = covering_slot(h, hc, gp[g]); if (sl >= 0 && !taken[sl]) slot_of[g] = sl; }   // (a single group was tried above)
        if (slot_of[g] >= 0) taken[slot_of[g]] = true;
    }
    int rel_at = 0;
    for (int g = 0; g < ng; ++g) {
        int sl = slot_of[g];
        if (sl < 0) {
            unsigned long long oldest = ~0ull;
            for (int c = 0; c < kMaxPairGroups; ++c) {
                if (taken[c]) continue;
                const unsigned long long age = h->pslot[c].valid ? h->pslot[c].last_use : 0ull;
                if (age < oldest) { oldest = age; sl = c; }
            }
            taken[sl] = true; slot_of[g] = sl;
            iba_handle::PairSlot& ps = h->pslot[sl];
            // a reusable (inflated, entrywise-bounded) list only for a small tight group: a list grows with the square of its window,
            // and at 16 px of nominal spread the inflated one overflows its capacity where the group's own fits (tools/wide_probe.py)
            const bool memo = h->pair_memo && gp[g].n <= h->pair_memo_max_b && gp[g].px <= h->pair_memo_max_px;
            if (memo) {
                for (int i = 0; i < 9; ++i) gp[g].gr.rho[i] = gp[g].gr.rho[i] * h->pair_infl + h->pair_rho_floor;
                for (int i = 0; i < 3; ++i) gp[g].gr.tau[i] = gp[g].gr.tau[i] * h->pair_infl + h->pair_tau_floor;
            }
            const int k = h->n_build++;
            PairsPlan& pl = h->pplan;
            pl.g[k] = gp[g].gr; pl.slot[k] = (uint8_t)sl;
            pl.cnt_off[k] = cnt_off(sl, ps.epoch); pl.next_off[k] = cnt_off(sl, ps.epoch + 1u);
            ++ps.epoch;
            // (lists that later calls may reuse are bounded entrywise only: the candidates' own per-block bound is this batch's)
            const bool own = h->pair_bound && gp[g].n > 1 && gp[g].n <= kOwnBoundMax && !memo;   // (a larger group: entrywise only — the candidates' own motions take a lane each in the pair search)
            pl.first[k] = (uint8_t)rel_at; pl.count[k] = (uint8_t)(own ? gp[g].n : 0);
            if (own && rel_at + gp[g].n > kOwnBoundMax) { pl.count[k] = 0; }   // (the rows of this launch's groups share PairsPlan::rel)
            else if (own) { std::memcpy(pl.rel[rel_at], gp[g].rel, sizeof(float) * 12 * (size_t)gp[g].n); rel_at += gp[g].n; }
            ps.ref = gp[g].gr; ps.valid = memo;
        }
        h->pslot[sl].last_use = ++h->pair_clock;
        for (int j = 0; j < gp[g].n; ++j) h->amap.slot[gp[g].idx[j]] = (uint8_t)sl;
    }
    for (int sl = 0; sl < kMaxPairGroups; ++sl) h->amap.cnt_off[sl] = cnt_off(sl, h->pslot[sl].epoch - 1u);
    h->n_groups = ng; h->last_mean_pairs_slot = slot_of[0];
    return true;
}

iba_status ensure_scratch(iba_handle* h);
iba_status compute_plane_cache(iba_handle* h) {
    const iba_params& p = h->params;
    h->scan_rec_valid = false;
    if (!p.plane_cache) return ensure_scratch(h);   // planes are refitted inside every evaluation, into per-candidate scratch
    const DevProblem dp = h->dev_problem();
    auto run = [&](double r2, int max_pts, PlaneRec* out) -> hipError_t {
        dim3 grid((h->maxP + 63) / 64, h->n_frames);
        if (h->maxP == 0 || h->n_frames == 0) return hipSuccess;
        if (max_pts <= 32) hipLaunchKernelGGL(iba_plane_kernel<2>, grid, dim3(64), 0, h->stream, dp, r2, max_pts, out);
        else hipLaunchKernelGGL(iba_plane_kernel<4>, grid, dim3(64), 0, h->stream, dp, r2, max_pts, out);
        hipError_t e = hipGetLastError(); if (e != hipSuccess) return e;
        return hipStreamSynchronize(h->stream);
    };
    const double r2c = p.norm_radius * p.norm_radius, r2l = p.neigh_radius * p.neigh_radius;
    if (h->plane_cost_r2 != r2c || h->plane_cost_max != p.norm_max_pts) {
        HIP_TRY(h, run(r2c, p.norm_max_pts, h->plane_cost.p));
        h->plane_cost_r2 = r2c; h->plane_cost_max = p.norm_max_pts;
    }
    h->plane_local_aliases_cost = (r2l == r2c && p.neigh_max_pts == p.norm_max_pts);
    if (!h->plane_local_aliases_cost && (h->plane_local_r2 != r2l || h->plane_local_max != p.neigh_max_pts)) {
        if (!h->plane_local.p) HIP_TRY(h, h->plane_local.alloc(h->plane_cost.n));
        HIP_TRY(h, run(r2l, p.neigh_max_pts, h->plane_local.p));
        h->plane_local_r2 = r2l; h->plane_local_max = p.neigh_max_pts;
    }
    if (h->plane_cost.n) {   // the thresholds of the two verdicts are parameters too: every call of this function rebuilds them
        if (!h->plane_ok.p) HIP_TRY(h, h->plane_ok.alloc(h->plane_cost.n));
        hipLaunchKernelGGL(iba_verdict_kernel, dim3((unsigned)((h->plane_cost.n + 255) / 256)), dim3(256), 0, h->stream, h->plane_local_aliases_cost ? h->plane_cost.p : h->plane_local.p, h->dprm, h->plane_ok.p, h->plane_cost.n);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        // the factor kernel's scan-point records: the point and the normal of its memoised LOCAL plane side by side (one cache line). An optional memo:
        // 64 B per scan point; when the allocation fails the kernel gathers from the separate arrays
        if (!h->d_scan_rec.p && h->d_scan_rec.alloc(h->plane_cost.n) != hipSuccess) { (void)hipGetLastError(); h->d_scan_rec.p = nullptr; h->d_scan_rec.n = 0; }
        if (h->d_scan_rec.p) {
            hipLaunchKernelGGL(iba_scanrec_kernel, dim3((unsigned)((h->plane_cost.n + 255) / 256)), dim3(256), 0, h->stream, h->pts4.p, h->plane_local_aliases_cost ? h->plane_cost.p : h->plane_local.p, h->d_scan_rec.p, h->plane_cost.n);
            HIP_TRY(h, hipGetLastError());
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            h->scan_rec_valid = true;
        }
    }
    return IBA_OK;
}

// plane_cache = 0: private plane records for IBA_MAX_BATCH candidates (+ slot 0 for the frozen problem of iba_build_problem),
// allocated when the mode is entered (iba_create / iba_set_params) and never during an evaluation: the *_partial entry points
// do not synchronise, and the frozen problem's planes (slot 0) survive evaluations of any batch size.
iba_status ensure_scratch(iba_handle* h) {
    if (h->params.plane_cache) return IBA_OK;
    const bool alias = (h->params.norm_radius == h->params.neigh_radius && h->params.norm_max_pts == h->params.neigh_max_pts);
    if (h->scratch_cap >= IBA_MAX_BATCH && h->scratch_local_aliases == alias) return IBA_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->scratch_cost.release(); h->scratch_local.release();
    const size_t n = (size_t)(IBA_MAX_BATCH + 1) * (size_t)std::max<int64_t>(h->n_pt_total, 1);
    HIP_TRY(h, h->scratch_cost.alloc(n));
    if (!alias) HIP_TRY(h, h->scratch_local.alloc(n));
    h->scratch_cap = IBA_MAX_BATCH; h->scratch_local_aliases = alias;
    h->frozen_valid = false;   // slot 0 (the frozen problem's planes) went with the old buffer
    return IBA_OK;
}

// Stages B candidates into a slot of the pinned ring. The block is computed from x here, or copied from `pre` when the caller
// (iba_group) has already computed it for all its devices. jets = 0: the values only (cost evaluations never read the derivatives);
// jets = 2: the values now, the derivatives later (finish_jets: the host differentiates the exponentials while the GPU runs the
// association and search kernels on the values) — a small batch computes them at once (0.6 us each: less than a second copy).
// plan: the chain goes through run_split, which carries the block to the device in spare blocks of its first kernel (chain_fold);
// otherwise — and for the callers outside run_split — a staging launch copies it here. *d_out = where the device copy will be.
iba_status stage_cands(iba_handle* h, const double* x, int B, hipStream_t st, Cand** d_out, const Cand* pre = nullptr, int jets = 1, const std::atomic<int>* pre_flag = nullptr, bool plan = false) {
    const int slot = h->ring_next; h->ring_next = (h->ring_next + 1) % kRing;
    if (h->ring_used[slot]) HIP_TRY(h, hipEventSynchronize(h->ring_ev[slot]));
    Cand* hc = h->h_cands + (size_t)slot * h->chain_cap;
    Cand* dc = h->d_cands.p + (size_t)slot * h->chain_cap;
    h->last_hc = hc;
    h->jets_x = nullptr; h->jets_src = nullptr; h->jets_flag = nullptr;
    if (!pre && jets == 2 && B <= 16) jets = 1;   // a small batch: its derivatives cost the host less (0.6 us each) than the second copy of the block
    if (pre) {   // the group's block: complete, or (pre_flag) with the derivative half still being computed by the calling thread
        if (pre_flag && jets == 2) {
            // the calling thread is writing the derivative half of `pre` right now: only the value half is read here (the other is
            // copied by finish_jets once *pre_flag is set). The stale derivative words of the ring slot that the head's copy carries
            // along are overwritten on the device by finish_jets' own copy, stream-ordered before the factor kernel.
            for (int b = 0; b < B; ++b) std::memcpy((void*)&hc[b], (const void*)&pre[b], offsetof(Cand, dR));
            h->jets_src = pre; h->jets_flag = pre_flag; h->jets_B = B; h->jets_slot = slot;
        } else std::memcpy(hc, pre, sizeof(Cand) * (size_t)B);
        jets = 1;
    }
    else if (jets == 1) for (int b = 0; b < B; ++b) make_cand(x + 7 * b, hc[b]);
    else for (int b = 0; b < B; ++b) make_cand_values(x + 7 * b, hc[b]);
    const bool planned = plan && h->common_mode > 0 && B >= h->common_min_batch && h->d_pairs.p;
    h->cref_ok = planned && plan_pairs(h, hc, B);
    h->plan_wide = planned && !h->cref_ok;   // the planner looked at this batch and found it wide everywhere
    h->head_pending = false; h->head_slot = slot; h->head_B = B;
    if (plan && h->chain_fold) h->head_pending = true;   // run_split's first kernel carries the block (and records the slot's event at the end of the chain)
    else {
        // the block crosses PCIe by a kernel that reads the pinned ring (a copy-engine transfer of these 70 KB costs ~15 us of latency)
        const uint32_t n16 = (uint32_t)(sizeof(Cand) * (size_t)B / 16);
        hipLaunchKernelGGL(iba_fetch_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, (const uint4*)(h->h_cands_dev + (size_t)slot * h->chain_cap), (uint4*)dc, n16);
        HIP_TRY(h, hipGetLastError());
        HIP_TRY(h, hipEventRecord(h->ring_ev[slot], st));
    }
    if (jets == 2) { h->jets_x = x; h->jets_B = B; h->jets_slot = slot; }
    h->ring_used[slot] = true;
    *d_out = dc;
    return IBA_OK;
}
// the derivative half of the staged candidates: computed (or taken over from the group's calling thread) and copied now, in stream
// order, before the factor kernel is enqueued
// what is left to copy of a staged batch's derivative half (pinned ring -> device), in 16-byte words [w0, w1) of every Cand
struct JetsCopy { const uint4* src = nullptr; uint4* dst = nullptr; uint32_t B = 0, w0 = 0, w1 = 0; };
// the host side: the derivatives are computed (or taken over from the group's calling thread) into the pinned slot. false: nothing is pending
bool prepare_jets(iba_handle* h, JetsCopy& jc) {
    if (!h->jets_x && !h->jets_src) return false;
    Cand* hc = h->h_cands + (size_t)h->jets_slot * h->chain_cap;
    Cand* dc = h->d_cands.p + (size_t)h->jets_slot * h->chain_cap;
    if (h->jets_src) {   // the group's calling thread has been differentiating while this device's kernels ran on the values
        while (h->jets_flag->load(std::memory_order_acquire) == 0) { /* microseconds */ }
        for (int b = 0; b < h->jets_B; ++b)   // the derivative half alone: the values stay as staged
            std::memcpy((char*)&hc[b] + offsetof(Cand, dR), (const char*)&h->jets_src[b] + offsetof(Cand, dR), sizeof(Cand) - offsetof(Cand, dR));
        h->jets_src = nullptr; h->jets_flag = nullptr;
    } else
    for (int b = 0; b < h->jets_B; ++b) make_cand_jets(h->jets_x + 7 * b, hc[b]);
    static_assert(offsetof(Cand, dR) % 16 == 0 && sizeof(Cand) % 16 == 0, "the halves of a Cand are copied as 16-byte words");
    jc.src = (const uint4*)(h->h_cands_dev + (size_t)h->jets_slot * h->chain_cap); jc.dst = (uint4*)dc;
    jc.B = (uint32_t)h->jets_B; jc.w0 = (uint32_t)(offsetof(Cand, dR) / 16); jc.w1 = (uint32_t)(sizeof(Cand) / 16);
    h->jets_x = nullptr;
    return true;
}
// the derivative half of the staged candidates: computed and copied now, in stream order, before the factor kernel is enqueued (a chain with a search
// kernel lets that kernel's spare blocks do the copy instead: run_split)
iba_status finish_jets(iba_handle* h, hipStream_t st) {
    JetsCopy jc;
    if (!prepare_jets(h, jc)) return IBA_OK;
    // the derivative half alone (the kernels in flight read the value half), on the chain's own stream
    const uint32_t n = jc.B * (jc.w1 - jc.w0);
    hipLaunchKernelGGL(iba_fetch_jets_kernel, dim3((n + 255) / 256), dim3(256), 0, st, jc.src, jc.dst, jc.B, jc.w0, jc.w1, jc.w1);
    HIP_TRY(h, hipGetLastError());
    return IBA_OK;
}

// End of an evaluation: the host polls the stream (hipStreamQuery) for up to 2 ms before it falls back to the blocking
// wait — an evaluation takes 0.1 .. 0.6 ms and the blocking wait's wake-up costs 10-20 us of it (IBA_SPIN_WAIT=0: always block).
hipError_t wait_stream(iba_handle* h, hipStream_t st) {
    if (h->spin_wait) {
        const auto t0 = std::chrono::steady_clock::now();
        int polls = 0;
        for (;;) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
            if ((++polls & 15) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    return hipStreamSynchronize(st);
}

// End of a BLOCKING evaluation whose sums land in the handle's pinned block: the host polls the sequence number the summing kernel's last
// block publishes there (iba_reduce2_kernel), not the stream — whose completion reaches the host 3-4 us later. Falls back to the stream
// after 2 ms, and whenever the chain did not arm the flag (event timing on, an empty handle, IBA_DONE_FLAG=0).
hipError_t wait_done(iba_handle* h, hipStream_t st) {
    const bool armed = h->done_armed;
    h->done_armed = false;
    if (armed && h->spin_wait) {   // (a caller that asked for blocking waits only — spin_wait = 0 — never polls: not the stream, not this word)
        const volatile unsigned long long* f = h->h_done;
        const auto t0 = std::chrono::steady_clock::now();
        for (int polls = 1;; ++polls) {
            if (*f == h->done_seq) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
            if ((polls & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    return wait_stream(h, st);
}

// ranges (= waves = records) per candidate of iba_factor2_kernel for a batch of B: the device's wave slots shared out, at most one per keyframe
// (a range of less than a keyframe's list is mostly start-up), at least one. 0: the batch runs on the kernels that write one record per keyframe.
int factor_waves(const iba_handle* h, int B) {
    if (!h->factor_v2 || !h->factor_valu || h->n_frames == 0 || h->n_frames > kF2MaxFrames || !h->d_ffr.p || h->f2lay.total == 0) return 0;
    // a lane counts the blocks it evaluates in 16 bits (NAccP): a range may hold at most 64 x 60000 list entries
    const int w_min = (int)std::min<uint64_t>((uint64_t)h->n_frames, ((uint64_t)h->n_frames * h->lstride + 64ull * 60000ull - 1ull) / (64ull * 60000ull));
    if (h->factor_waves_forced > 0) return std::max(w_min, std::min(h->factor_waves_forced, h->n_frames));
    return std::max(std::max(1, w_min), std::min(h->n_frames, h->factor_slots / std::max(B, 1)));
}
// records per candidate the factor kernel of this launch writes
int factor_records(const iba_handle* h, int B) { const int w = factor_waves(h, B); return w > 0 ? w : h->n_frames; }

iba_status launch_factors(iba_handle* h, const Cand* dc, int B, int per_cand, double* partials, int nrec, int rec_base, hipStream_t st) {
    if (h->n_frames == 0) return IBA_OK;
    if (const int W = factor_waves(h, B)) {
        const uint4* fl2 = per_cand ? h->d_flist.p : h->d_flist_frozen.p; const uint32_t* fc2 = per_cand ? h->d_fcount.p : h->d_fcount_frozen.p;
        const dim3 grid2(8u * (uint32_t)((W + 7) / 8) * (uint32_t)B);
        const bool many = h->max_slots > (uint32_t)kCovisWord;
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid2, dim3(64), h->f2lay.total, st, h->dev_problem(), h->dprm, dc, fl2, fc2, (int)h->lstride, per_cand, partials, nrec, rec_base, B, W,
                                                      (const double*)h->d_ffr.p, (const double2*)h->d_kp_c.p, h->f2lay, h->factor_dbg); };
        if (h->dprm.p2pix) { if (many) go(iba_factor2_kernel<true, true>); else go(iba_factor2_kernel<false, true>); }
        else { if (many) go(iba_factor2_kernel<true, false>); else go(iba_factor2_kernel<false, false>); }
        HIP_TRY(h, hipGetLastError());
        return IBA_OK;
    }
    // (a handle whose frames hold no keypoint at all still launches: every list is empty and the kernel writes zero records,
    // which is what the sums over the records expect)
    const dim3 grid(h->n_frames, B);
    // iba_factor_kernel with a keyframe count that is no multiple of 8: the XCD-aware mapping of the association kernels (see the kernel)
    const bool xmap = (h->n_frames % 8) != 0;
    const dim3 grid1 = xmap ? dim3(8 * ((h->n_frames + 7) / 8) * B) : grid;
    const int Bx = xmap ? B : 0;
    const uint4* fl = per_cand ? h->d_flist.p : h->d_flist_frozen.p; const uint32_t* fc = per_cand ? h->d_fcount.p : h->d_fcount_frozen.p;
    if (h->factor_valu && h->scan_rec_valid && h->factor_rec && h->d_kp_rec.p) {   // the gathers from the one-line records (planes memoised)
        const uint32_t lds = 96u * std::max<uint32_t>(h->max_slots, 1u);
        auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid1, dim3(kFactorThreads), lds, st, h->dev_problem(), h->dprm, dc, fl, fc, (int)h->lstride, per_cand, partials, nrec, rec_base, Bx); };
        const bool many = h->max_slots > (uint32_t)kCovisWord;
        if (h->dprm.p2pix) { if (many) go(iba_factor_kernel<true, true, true>); else go(iba_factor_kernel<false, true, true>); }
        else { if (many) go(iba_factor_kernel<true, false, true>); else go(iba_factor_kernel<false, false, true>); }
        HIP_TRY(h, hipGetLastError());
        return IBA_OK;
    }
    if (h->dprm.p2pix) {   // IBATestEdge edges (factor_3d2d_kind = 1): an instantiation of its own (the matrix-core variant has none)
        if (h->max_slots > (uint32_t)kCovisWord) hipLaunchKernelGGL((iba_factor_kernel<true, true>), grid1, dim3(kFactorThreads), 96u * h->max_slots, st, h->dev_problem(), h->dprm, dc, fl, fc, (int)h->lstride, per_cand, partials, nrec, rec_base, Bx);
        else hipLaunchKernelGGL((iba_factor_kernel<false, true>), grid1, dim3(kFactorThreads), 96u * std::max<uint32_t>(h->max_slots, 1u), st, h->dev_problem(), h->dprm, dc, fl, fc, (int)h->lstride, per_cand, partials, nrec, rec_base, Bx);
    }
    else if (h->factor_valu && h->max_slots > (uint32_t)kCovisWord) hipLaunchKernelGGL(iba_factor_kernel<true>, grid1, dim3(kFactorThreads), 96u * h->max_slots, st, h->dev_problem(), h->dprm, dc, fl, fc, (int)h->lstride, per_cand, partials, nrec, rec_base, Bx);
    else if (h->factor_valu) hipLaunchKernelGGL(iba_factor_kernel<false>, grid1, dim3(kFactorThreads), 96u * std::max<uint32_t>(h->max_slots, 1u), st, h->dev_problem(), h->dprm, dc, fl, fc, (int)h->lstride, per_cand, partials, nrec, rec_base, Bx);
    else hipLaunchKernelGGL(iba_factor_mfma_kernel, grid, dim3(64), 0, st, h->dev_problem(), h->dprm, dc, fl, fc, (int)h->lstride, per_cand, partials, nrec, rec_base);
    HIP_TRY(h, hipGetLastError());
    return IBA_OK;
}


// per-candidate buffers of a chain — work lists (they also carry the residual blocks to the factor kernel), list lengths, partial
// records, hand-eye scratch — sized for the largest batch a call has passed so far (a chain takes up to chain_cap candidates: 16 B x
// keypoints x keyframes each for the lists alone, which a handle that is only ever asked for 64 never pays)
iba_status ensure_lists(iba_handle* h, int B, hipStream_t st) {
    if (h->assoc_cap >= B && (h->params.plane_cache || h->d_frefit.p)) return IBA_OK;
    B = std::max(B, h->assoc_cap);
    HIP_TRY(h, hipStreamSynchronize(st)); HIP_TRY(h, hipStreamSynchronize(h->stream));
    const size_t nf1 = (size_t)std::max(h->n_frames, 1);
    // a growth that fails half-way must not leave the old capacities standing over released buffers (the next call of at most the old size
    // would launch on null pointers): the handle forgets its capacities first and learns the new ones only when every buffer exists
    const int old_part_cap = h->part_cap;
    h->assoc_cap = 0;
    if (old_part_cap < B) h->part_cap = 0;
    h->d_flist.release(); h->d_fcount.release(); h->d_lcount.release();
    HIP_TRY(h, h->d_flist.alloc((size_t)B * nf1 * h->lstride));
    HIP_TRY(h, h->d_fcount.alloc((size_t)B * nf1));
    HIP_TRY(h, h->d_lcount.alloc((size_t)B * nf1));
    h->d_frefit.release();
    if (!h->params.plane_cache) HIP_TRY(h, h->d_frefit.alloc((size_t)std::min(B, (int)IBA_MAX_BATCH) * nf1 * h->lstride));   // (a chain of refitted planes takes at most IBA_MAX_BATCH candidates)
    if (old_part_cap < B) {
        h->d_frame_partials.release(); h->d_nn_partials.release(); h->d_he.release();
        HIP_TRY(h, h->d_frame_partials.alloc((size_t)B * std::max(h->nrec, 1) * kPartialStride));
        HIP_TRY(h, h->d_nn_partials.alloc((size_t)B * nf1 * h->nn_ns * kNNPartial));
        HIP_TRY(h, h->d_he.alloc((size_t)B * nf1));
        h->part_cap = B;
    }
    h->assoc_cap = B;
    return IBA_OK;
}
// candidates one chain takes on this handle right now: with the planes refitted per evaluation every candidate of a chain owns a
// private set of plane records (48 B x scan points), allocated for IBA_MAX_BATCH of them
inline int chain_limit(const iba_handle* h) { return h->params.plane_cache ? h->chain_cap : std::min(h->chain_cap, (int)IBA_MAX_BATCH); }


// Two-kernel evaluation chain (plane_cache = 1) on stream st: [hand-eye] -> association -> grouped 1-NN -> [factors] -> sums.
// want: bit 0 = BuildProblem association (+ normal equations when `factors`), bit 1 = BAError cost.
// frozen: B = 1, the lists go to the frozen problem's buffers (iba_build_problem).
iba_status run_split(iba_handle* h, const Cand* dc, int B, int want, bool frozen, bool factors, double* d_partials, hipStream_t st) {
    const int nf = h->n_frames;
    if (!frozen) { iba_status es = ensure_lists(h, B, st); if (es != IBA_OK) return es; }
    uint4* fl = frozen ? h->d_flist_frozen.p : h->d_flist.p;
    uint32_t* fc = frozen ? h->d_fcount_frozen.p : h->d_fcount.p;
    uint32_t* lc = frozen ? h->d_lcount_frozen.p : h->d_lcount.p;
    const DevProblem dp = h->dev_problem();
    const int nrec = factors ? h->nrec : nf;
    // candidates per search block: a power of two; list positions are cut into slices of a fixed width
    // candidates per search block: 8 fill a wave with neighbours that walk the same leaves, but a small batch then leaves the GPU
    // short of blocks and every block waits for its slowest search: fewer per block below 24 candidates (measured at 200 keyframes:
    // 8 candidates 0.122 -> 0.082 ms, 14 candidates 0.112 -> 0.101 ms; the sums do not depend on the grouping)
    // (r04, with the anchored lists' direct pass: 14 candidates 81.4 -> 84.5 k evaluations/s with 8 per block instead of 4, 8 candidates best with 4)
    // (round 6) the block shape of the search kernel: one wave and two candidates per block while the keyframes' trees are small (nn_small: see kNNThreadsSmall), else four waves
    // and up to eight; the refit chain (plane_cache = 0) keeps the four-wave shape
    const bool nn_small = h->nn_small && h->params.plane_cache && B >= h->nn_small_min_b;   // (a small batch keeps the four-wave blocks: 1 candidate 0.035 vs 0.038 ms, 8: 0.082 vs 0.088, 14: 0.081 vs 0.078, 64: 0.265 vs 0.252 — tools/latency_probe.py, cost tuple)
    const int nn_threads = nn_small ? kNNThreadsSmall : kNNThreads, nn_mg = nn_threads / 32 < 16 ? nn_threads / 32 : 16;
    const int cg_cap = std::min(nn_mg, h->nn_cg_fixed ? h->nn_cg_max : std::min(h->nn_cg_max, B >= 12 ? 8 : (B >= 6 ? 4 : 2)));
    int CG = 1; while (CG < std::min(B, cg_cap)) CG <<= 1;
    const int ngroups = (B + CG - 1) / CG;
    const int NS = h->nn_ns;
    NNLayout nl;
    if (!layout_nn(h, nl, nn_mg)) return fail(h, IBA_ERR_UNSUPPORTED, "kd tree exceeds the LDS plan of the search kernel");
    if (h->part_cap < B) { iba_status es = ensure_lists(h, B, st); if (es != IBA_OK) return es; }   // (the frozen problem's chain: one candidate)
    // the candidate block of this chain: still in the pinned ring (the first kernel below carries it to the device: chain_fold) or copied already
    const bool head = h->head_pending; h->head_pending = false;
    const uint4* head_src = head ? (const uint4*)(h->h_cands_dev + (size_t)h->head_slot * h->chain_cap) : nullptr;
    const uint32_t head_n16 = head ? (uint32_t)(sizeof(Cand) * (size_t)B / 16) : 0u;
    auto chain_done = [&]() -> iba_status {   // the ring slot is free again once everything enqueued so far has run: the head's copy AND finish_jets' copy,
        // which reads the slot long after a staging launch (chain_fold = 0) has recorded its own event
        if (!frozen && h->head_slot >= 0) HIP_TRY(h, hipEventRecord(h->ring_ev[h->head_slot], st));
        return IBA_OK;
    };
    if (nf == 0) {
        HIP_TRY(h, hipMemsetAsync(d_partials, 0, sizeof(double) * (size_t)B * kPartialStride, st));
        if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev0, st)); HIP_TRY(h, hipEventRecord(h->ev_mid, st)); HIP_TRY(h, hipEventRecord(h->ev1, st)); HIP_TRY(h, hipEventRecord(h->ev2, st)); h->timing_recorded = true; h->timing_split = true; }
        return chain_done();
    }
    const int per_xcd = (nf + 7) / 8;
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev0, st));
    const bool search = (want & 1) || h->dprm.use_3d3d;
    const int nn_nrec = nf * NS;
    // (Splitting the batch into chunks so that the association kernel of chunk i + 1 runs beside the search kernel of chunk i
    // on a second stream was measured: 0.93 ms -> 0.99 / 1.07 / 1.22 ms for 2 / 4 / 8 chunks. The kernels do not overlap enough
    // to pay for the extra launches and event hops; one launch each it is.)
    // plane_cache = 0: the planes are fitted per evaluation, between the kernels that produce the points and those that read
    // the planes: association -> fit<1> (local planes at the matched points) -> searches -> fit<2> (planes at the neighbours,
    // cost distances) -> the search kernel again for its fixed-order sums. Slot 0 of the plane scratch belongs to the frozen problem.
    const bool refit = !h->params.plane_cache;
    const int slot_base = frozen ? 0 : 1;
    if (refit && h->scratch_cap < B) return fail(h, IBA_ERR_STATE, "plane scratch not allocated");
    if (refit && frozen && !h->d_frefit.p) { iba_status es = ensure_lists(h, 1, st); if (es != IBA_OK) return es; }
    const bool wide_fit = std::max(h->params.norm_max_pts, h->params.neigh_max_pts) > 32;
    const dim3 fit_grid((h->lstride + 63) / 64, nf, B);
    // 2d-3d association: a batch of nearby candidates shares ONE search for the (scan point, keypoint) pairs per keyframe
    // (iba_pairs_kernel) and every candidate runs the exact test on that list (iba_assoc2_kernel); a lone candidate, a small or a
    // wide batch searches per candidate (iba_assoc_kernel). Same results either way, bit for bit.
    const bool common = !frozen && h->last_hc && h->cref_ok;
    h->last_path = common ? (h->n_groups > 1 ? 2 : 1) : 0;
    bool head_open = head;   // the first kernel launched below takes the head along
    if (common && h->n_build > 0) {   // (n_build == 0: the lists of earlier calls cover every group of this batch)
        // LDS of the pairs kernel: hit stage, coarse CSR (u16), the keypoints' (u, v)
        const uint32_t kuv_off = align_up(8u * (uint32_t)kPairStage + 128u + 2u * std::max(h->maxCoarse, 1u), 16u);
        const uint32_t lds = kuv_off + 8u * std::max(h->maxK, 1u);
        const PairsProblem pp{dp.frames, dp.pts4, dp.chunk_box, dp.kp_uv, dp.coarse_start};
        // the pair search needs nothing but its arguments: it goes first and, in one more z-plane of its grid, carries the candidates to the device
        hipLaunchKernelGGL(iba_pairs_kernel, dim3(std::max(1u, (h->maxP + (uint32_t)kPairsThreads - 1u) / (uint32_t)kPairsThreads), nf, h->n_build + (head_open ? 1 : 0)), dim3(kPairsThreads), lds, st, PairsArgs{pp, h->pplan}, h->params.max_pixel_dist, kuv_off,
                           nf, h->d_pairs.p, h->d_hard.p, h->d_pcounts.p, h->pair_cap, h->hard_cap, head_open ? head_src : nullptr, (uint4*)dc, head_open ? head_n16 : 0u, h->pairs_dense_min);
        HIP_TRY(h, hipGetLastError());
        h->pairs_builds += h->n_build;
        head_open = false;
    }
    // Anchored neighbour lists (iba_anchor_kernel): the scan points nearest to every MapPoint's query under an ANCHOR extrinsic,
    // built once and reused by every evaluation whose candidates stay near the anchor (each lane certifies its own pick, or
    // searches the tree). The anchor follows the optimiser: when the candidate of this call that is nearest the batch mean has
    // moved a nominal MapPoint (30 m out) further than anchor_reach from the anchor's query, the lists are rebuilt around it —
    // at most every fourth call, so that a wide exploratory phase does not rebuild for nothing.
    bool sets = false;
    const bool search_wanted = (want & 1) || h->dprm.use_3d3d;
    if (h->nn_sets && h->d_anchor.p && h->max_mpk > 0 && search_wanted && h->last_hc) {
        const Cand* hc = h->last_hc;
        auto drift = [&](const Cand& c, const AnchorRef& ar) {   // how far the query of a MapPoint 30 m out moves between an anchor and c (row-sum bound; |m| = 30 m / scale)
            const double nominal = 30.0 / std::max(std::fabs(c.s), 1e-12);
            double worst = 0;
            for (int r = 0; r < 3; ++r) {
                double v = std::fabs(c.ti[r] - ar.t[r]);
                for (int q = 0; q < 3; ++q) v += nominal * std::fabs(c.s * c.Ri[r * 3 + q] - ar.M[r * 3 + q]) / std::sqrt(3.0);
                worst = std::max(worst, v);
            }
            return worst;
        };
        // the poll centres of this batch: the groups of the pair-search plan (one per incumbent) — or the whole batch; of each, the member
        // nearest the group's mean query transform
        static thread_local int grp_of[kMaxChain]; int n_grp = 1;
        for (int b = 0; b < B; ++b) grp_of[b] = 0;
        if (common && h->n_groups > 1) {
            int slot_grp[kMaxPairGroups] = {-1, -1, -1, -1}; n_grp = 0;
            for (int b = 0; b < B; ++b) { int& g = slot_grp[h->amap.slot[b]]; if (g < 0) g = n_grp++; grp_of[b] = g; }
        }
        int ref_of[kMaxPairGroups], size_of[kMaxPairGroups];
        for (int g = 0; g < n_grp; ++g) {
            double mean[12] = {0}; int n = 0;
            for (int b = 0; b < B; ++b) if (grp_of[b] == g) { for (int i = 0; i < 9; ++i) mean[i] += hc[b].s * hc[b].Ri[i]; for (int i = 0; i < 3; ++i) mean[9 + i] += hc[b].ti[i]; ++n; }
            double best = INFINITY; int ref = -1;
            for (int b = 0; b < B; ++b) if (grp_of[b] == g) {
                double dd = 0;
                for (int i = 0; i < 9; ++i) dd = std::max(dd, 3.0 * std::fabs(hc[b].s * hc[b].Ri[i] - mean[i] / n));
                for (int i = 0; i < 3; ++i) dd = std::max(dd, std::fabs(hc[b].ti[i] - mean[9 + i] / n));
                if (ref < 0 || dd < best) { best = dd; ref = b; }
            }
            ref_of[g] = ref; size_of[g] = n;
        }
        for (int a = 0; a < kAnchorSets; ++a) ++h->calls_since_anchor[a];
        // The anchors follow the optimiser: a centre none of whose anchors is within anchor_reach gets new lists — in a free set at once,
        // else in the least recently used set no centre of THIS batch sits on, and then at most every fourth call of that set (a wide
        // exploratory phase does not rebuild for nothing). At most one build per call: the biggest centre first.
        bool in_use[kAnchorSets] = {false, false};
        int want_build = -1;
        for (int g = 0; g < n_grp; ++g) {
            double near = INFINITY; int na = -1;
            for (int a = 0; a < kAnchorSets; ++a) if (h->anchor_valid[a]) { const double d = drift(hc[ref_of[g]], h->anchor_ref[a]); if (d < near) { near = d; na = a; } }
            if (na >= 0 && near <= h->anchor_reach) in_use[na] = true;
            else if (std::isfinite(hc[ref_of[g]].s) && (want_build < 0 || size_of[g] > size_of[want_build])) want_build = g;
        }
        // (a batch that is wide everywhere — tens of pixels between any two candidates — is an exploratory poll: lists around its
        //  centre would serve one or two of its candidates; 216 -> fewer builds of 0.16 ms in a recorded calibration)
        if (want_build >= 0 && !h->plan_wide) {
            int slot = -1;
            for (int a = 0; a < kAnchorSets && slot < 0; ++a) if (!h->anchor_valid[a]) slot = a;
            if (slot < 0) {
                unsigned long long oldest = ~0ull;
                for (int a = 0; a < kAnchorSets; ++a) if (!in_use[a] && h->calls_since_anchor[a] >= 4 && h->anchor_used[a] < oldest) { oldest = h->anchor_used[a]; slot = a; }
            }
            if (slot >= 0) {
                const Cand& c = hc[ref_of[want_build]];
                AnchorRef& ar = h->anchor_ref[slot];
                for (int i = 0; i < 9; ++i) ar.M[i] = c.s * c.Ri[i];
                for (int i = 0; i < 3; ++i) ar.t[i] = c.ti[i];
                hipLaunchKernelGGL(iba_anchor_kernel, dim3((h->max_mpk + kAnchorThreads - 1) / kAnchorThreads, nf), dim3(kAnchorThreads), 8u * std::max(h->maxNodes, 1u), st, AnchorArgs{dp, h->dprm, ar}, h->d_anchor.p + (size_t)slot * h->anchor_set_elems);
                HIP_TRY(h, hipGetLastError());
                h->anchor_valid[slot] = true; h->calls_since_anchor[slot] = 0; ++h->anchor_builds;
            }
        }
        // every candidate reads the set of the anchor nearest to it — further than 4 x anchor_reach the certificates fail anyway: none
        for (int b = 0; b < B; ++b) {
            double near = INFINITY; int na = 255;
            for (int a = 0; a < kAnchorSets; ++a) if (h->anchor_valid[a]) { const double d = drift(hc[b], h->anchor_ref[a]); if (d < near) { near = d; na = a; } }
            h->anchor_sel[b] = (uint8_t)((na != 255 && near <= 4.0 * h->anchor_reach) ? na : 255);
            if (h->anchor_sel[b] != 255) { sets = true; h->anchor_used[na] = ++h->anchor_clock; }
        }
    }
    // an association kernel that is the first of its chain carries the head in kHeadBlocks spare blocks, and its own blocks read their
    // candidate (R, t, s: 13 doubles) where it lies in the pinned ring
    uint32_t head_blocks = head_open ? std::min<uint32_t>(kHeadBlocks, (head_n16 + (uint32_t)kThreads - 1u) / (uint32_t)kThreads) : 0u;
    const Cand* assoc_cands = head_open ? (const Cand*)head_src : dc;
    const uint4* a_src = head_open ? head_src : nullptr; uint32_t a_n16 = head_open ? head_n16 : 0u;
    // (round 6) the pair search has carried the values: the spare blocks of the shared-pair association are free for the DERIVATIVE half, which the host finishes now — the
    // GPU is busy with the pair search for 45 us — instead of a copy launch of its own in front of the factor kernel (5 us in stream order). A chain whose association kernel
    // carries the head itself, or the per-candidate association, keeps the copy launch (finish_jets below).
    if (common && !head_open && factors && h->chain_fold && h->jets_fold) {
        JetsCopy jc;
        if (prepare_jets(h, jc) && jc.B <= 0xFFFFu && jc.w1 < 128u) {
            a_src = jc.src; a_n16 = 0x80000000u | jc.B | (jc.w0 << 16) | (jc.w1 << 24);
            head_blocks = std::min<uint32_t>(kHeadBlocks, (jc.B * (jc.w1 - jc.w0) + 255u) / 256u);
        }
    }
    if (common) {
        // (twelve instantiations: two / four / any number of flagged keypoints per thread x at most 30 covisible keyframes or more x 256 / 512 threads)
        const int at = assoc2_threads(h, 8 * per_xcd * B);
        h->last_assoc2_threads = at;
        auto launch_assoc2 = [&](auto qtag, auto ttag) {
            constexpr int QQ = decltype(qtag)::value, TT = decltype(ttag)::value;
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(8 * per_xcd * B + head_blocks), dim3(TT), h->alay2.total, st, K2Args{KArgs{dp, h->dprm, h->alay2}, h->amap}, assoc_cands, B, want | (refit ? 4 : 0) | (h->assoc_dbg << 8),
                                                          h->d_frame_partials.p, nrec, fl, fc, lc, (int)h->lstride, h->d_pairs.p, h->d_hard.p, h->d_pcounts.p, h->pair_cap, h->hard_cap, a_src, (uint4*)dc, a_n16); };
            if (h->max_slots > (uint32_t)kCovisWord) go(iba_assoc2_kernel<QQ, true, TT>); else go(iba_assoc2_kernel<QQ, false, TT>);
        };
        auto launch_t = [&](auto ttag) {
            const int aq = assoc2_q(h, decltype(ttag)::value);
            if (aq == 2) launch_assoc2(std::integral_constant<int, 2>{}, ttag);
            else if (aq == 4) launch_assoc2(std::integral_constant<int, 4>{}, ttag);
            else launch_assoc2(std::integral_constant<int, 0>{}, ttag);
        };
        if (at == 256) launch_t(std::integral_constant<int, 256>{}); else launch_t(std::integral_constant<int, kThreads>{});
    } else {
    auto go1 = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(8 * per_xcd * B + head_blocks), dim3(kThreads), h->alay.total, st, KArgs{dp, h->dprm, h->alay}, assoc_cands, B, want | (refit ? 4 : 0) | (h->assoc_dbg << 8),
                       h->d_frame_partials.p, nrec, (uint32_t*)nullptr, fl, fc, lc, (int)h->lstride, a_src, (uint4*)dc, a_n16); };
    const int aq1 = assoc2_q(h);
    if (aq1 == 2) go1(iba_assoc_kernel<2>); else if (aq1 == 4) go1(iba_assoc_kernel<4>); else go1(iba_assoc_kernel<0>);
    }
    head_open = false;
    HIP_TRY(h, hipGetLastError());
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev_mid, st));
    if (refit && (want & 1)) {
        if (wide_fit) hipLaunchKernelGGL((iba_fit_kernel<4, 1>), fit_grid, dim3(64), 0, st, dp, h->dprm, fl, h->d_frefit.p, lc, (int)h->lstride, slot_base, want);
        else hipLaunchKernelGGL((iba_fit_kernel<2, 1>), fit_grid, dim3(64), 0, st, dp, h->dprm, fl, h->d_frefit.p, lc, (int)h->lstride, slot_base, want);
        HIP_TRY(h, hipGetLastError());
    }
    bool he_in_search = false;
    if (search) {
        // the hand-eye terms of a cost evaluation ride in front of the search's grid (two lanes per term), padded to a multiple of 8 blocks
        const int he_blocks = (want & 2) ? (B * nf + nn_threads / 2 - 1) / (nn_threads / 2) : 0;
        he_in_search = he_blocks > 0;
        const dim3 grid(8 * per_xcd * ngroups * NS + ((he_blocks + 7) & ~7)), block(nn_threads);
        NNArgs na{dp, h->dprm, nl, (unsigned long long)(h->anchor_set_elems * sizeof(SetPt)), h->nn_rounds ? 1u : 0u, {0}};
        std::memcpy(na.anchor_sel, h->anchor_sel, sizeof(na.anchor_sel));
        const bool wA = (want & 1) != 0, wC = (want & 2) && h->dprm.use_3d3d;
        const SetPt* anchor = sets ? h->d_anchor.p : nullptr;
        auto launch_nn = [&](auto mode_tag) {
            constexpr int MODE = decltype(mode_tag)::value;
            h->last_nn_nrec = nn_nrec; h->last_nn_B = B;
            const int heb = MODE == kRefitSums ? 0 : he_blocks;   // (the second launch of a refit chain only sums)
            const dim3 grid_m(MODE == kRefitSums ? 8 * per_xcd * ngroups * NS : grid.x);
            auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid_m, block, nl.total, st, na, dc, B, CG, NS, h->d_nn_partials.p, nn_nrec, fl, lc, (int)h->lstride, h->nn_dbg, h->d_frefit.p, anchor, h->d_he.p, heb); };
            h->last_nn_list = 0; h->last_nn_threads = nn_threads;
            const bool small_bufs = (size_t)B * nf * h->lstride * 16u < 0xFFFFFF00ull && (size_t)kAnchorSets * h->anchor_set_elems * sizeof(SetPt) < 0xFFFFFF00ull && (size_t)dp.n_kp_total * 16u < 0x80000000ull;   // (32-bit buffer offsets)
            // (the opt-in persistent form keeps its own block shape: 256 threads, up to 8 candidates per block; its LDS plan and hand-eye blocks follow that shape)
            int CGl = 1; { const int capl = B >= 12 ? 8 : (B >= 6 ? 4 : 2); while (CGl < std::min(B, capl)) CGl <<= 1; }
            if (sets && MODE == 0 && h->nn_list && small_bufs && CGl >= 4 && (kSliceW * (uint32_t)CGl) % (2u * (uint32_t)kNLThreads) == 0u && (kSliceW * (uint32_t)CGl) / (uint32_t)kNLThreads <= 4u) {
                NNLayout ll{};
                {
                    uint32_t off = 0;
                    ll.off_nodes = off; off += 8u * std::max(h->maxNodes, 1u);
                    off = align_up(off, 16); ll.off_misc = off; off += 128u;
                    off = align_up(off, 16); ll.off_cd = off; off += 8u * (uint32_t)kCdDoubles * (uint32_t)kNLMaxGroup;
                    off = align_up(off, 16); ll.off_res = off; off += 8u * kSliceW * (uint32_t)kNLMaxGroup;
                    ll.off_ovf = off; off += 4u * kSliceW * (uint32_t)kNLMaxGroup;
                    ll.total = off; off = align_up(off, 16); ll.off_res2 = off;
                }
                const uint32_t lds = align_up(ll.off_res2 + 8u * kSliceW * (uint32_t)kNLMaxGroup, 16);
                const int ngl = (B + CGl - 1) / CGl;
                const int hebl = (want & 2) ? (B * nf + kNLThreads / 2 - 1) / (kNLThreads / 2) : 0;
                const int per_cu = h->nn_list_workers > 0 ? h->nn_list_workers : std::max(1, std::min(4 * IBA_NN_LIST_WAVES * 64 / kNLThreads, (int)(kLdsBytes / lds)));
                const int R = std::max(1, std::min(per_xcd * NS, (h->n_cus * per_cu) / (8 * ngl)));
                h->last_nn_list = R;
                NNArgs nal = na; nal.lay = ll;
                const dim3 grid_l(8 * ngl * R + ((hebl + 7) & ~7));
                auto gol = [&](auto kern) { hipLaunchKernelGGL(kern, grid_l, dim3(kNLThreads), lds, st, nal, dc, B, CGl, NS, R, h->d_nn_partials.p, nn_nrec, fl, lc, (int)h->lstride, h->nn_dbg, anchor, h->d_he.p, hebl); };
                const bool four = (kSliceW * (uint32_t)CGl) / (uint32_t)kNLThreads == 4u;
                if (wA && wC) { if (four) gol(iba_nn_list_kernel<3, 4>); else gol(iba_nn_list_kernel<3, 2>); }
                else if (wA) { if (four) gol(iba_nn_list_kernel<1, 4>); else gol(iba_nn_list_kernel<1, 2>); }
                else { if (four) gol(iba_nn_list_kernel<2, 4>); else gol(iba_nn_list_kernel<2, 2>); }
            }
            else if (MODE == 0 && nn_small) {   // (one-wave blocks: instantiated for the memoised-plane chain only)
                if (sets) { if (wA && wC) go(iba_nn_kernel<3, 0, 1, kNNThreadsSmall>); else if (wA) go(iba_nn_kernel<1, 0, 1, kNNThreadsSmall>); else go(iba_nn_kernel<2, 0, 1, kNNThreadsSmall>); }
                else { if (wA && wC) go(iba_nn_kernel<3, 0, 0, kNNThreadsSmall>); else if (wA) go(iba_nn_kernel<1, 0, 0, kNNThreadsSmall>); else go(iba_nn_kernel<2, 0, 0, kNNThreadsSmall>); }
            }
            else if (sets && MODE != kRefitSums) { if (wA && wC) go(iba_nn_kernel<3, MODE, 1>); else if (wA) go(iba_nn_kernel<1, MODE, 1>); else go(iba_nn_kernel<2, MODE, 1>); }
            else { if (wA && wC) go(iba_nn_kernel<3, MODE, 0>); else if (wA) go(iba_nn_kernel<1, MODE, 0>); else go(iba_nn_kernel<2, MODE, 0>); }
        };
        if (refit) launch_nn(std::integral_constant<int, kRefitSearch>{}); else launch_nn(std::integral_constant<int, 0>{});
        HIP_TRY(h, hipGetLastError());
        if (refit) {
            const int w2 = (wA ? 1 : 0) | (wC ? 2 : 0);
            if (wide_fit) hipLaunchKernelGGL((iba_fit_kernel<4, 2>), fit_grid, dim3(64), 0, st, dp, h->dprm, fl, h->d_frefit.p, lc, (int)h->lstride, slot_base, w2);
            else hipLaunchKernelGGL((iba_fit_kernel<2, 2>), fit_grid, dim3(64), 0, st, dp, h->dprm, fl, h->d_frefit.p, lc, (int)h->lstride, slot_base, w2);
            HIP_TRY(h, hipGetLastError());
            if (wC) { launch_nn(std::integral_constant<int, kRefitSums>{}); HIP_TRY(h, hipGetLastError()); }
        }
    }
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev1, st));
    if (factors) {
        iba_status s = finish_jets(h, st); if (s != IBA_OK) return s;   // the GPU has been busy with the values since stage_cands
        s = launch_factors(h, dc, B, 1, h->d_frame_partials.p, h->nrec, nf, st); if (s != IBA_OK) return s;
    }
    const int n_fact = factors ? factor_records(h, B) : 0;   // records the factor kernel wrote behind the association's nf (iba_factor2_kernel: one per range)
    // the sums — and, for a cost evaluation, K7: the hand-eye term of every counted (candidate, frame), evaluated where it is summed
    {
        const double* nnp = search ? h->d_nn_partials.p : (const double*)nullptr;
        const int he_mode = (want & 2) ? (he_in_search ? 2 : 1) : 0;
        // a blocking entry point's sums go to pinned memory and its caller polls the flag the last block publishes (wait_done)
        const bool flag = h->done_flag_on && d_partials == h->h_partials_dev && !h->timing;
        unsigned long long* df = flag ? h->h_done_dev : nullptr;
        if (flag) ++h->done_seq;
        h->done_armed = flag;
        if (he_mode == 2) hipLaunchKernelGGL(iba_reduce2_kernel<2>, dim3(B), dim3(kReduceThreads), 0, st, h->d_frame_partials.p, nrec, nf, n_fact, nnp, nn_nrec, d_partials, dp.frames, dc, h->d_he.p, df, h->d_done_ctr.p, h->done_seq);
        else if (he_mode == 1) hipLaunchKernelGGL(iba_reduce2_kernel<1>, dim3(B), dim3(kReduceThreads), 0, st, h->d_frame_partials.p, nrec, nf, n_fact, nnp, nn_nrec, d_partials, dp.frames, dc, h->d_he.p, df, h->d_done_ctr.p, h->done_seq);
        else hipLaunchKernelGGL(iba_reduce2_kernel<0>, dim3(B), dim3(kReduceThreads), 0, st, h->d_frame_partials.p, nrec, nf, n_fact, nnp, nn_nrec, d_partials, dp.frames, dc, h->d_he.p, df, h->d_done_ctr.p, h->done_seq);
    }
    HIP_TRY(h, hipGetLastError());
    if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev2, st)); h->timing_recorded = true; h->timing_split = true; }
    return chain_done();
}

// plan_pairs commits a list slot (reference, bound, valid, epoch) when the candidates are staged — before iba_pairs_kernel has been
// enqueued. A chain that fails between the two (staging launch, event, pair search) would leave slots valid for lists that were never
// built, and a later call inside their bound would associate on stale or empty lists: a failed chain invalidates every slot.
iba_status chain_status(iba_handle* h, iba_status s) {
    if (s != IBA_OK && h) {
        for (auto& ps : h->pslot) ps.valid = false;
        // a chain that failed after its first launch never recorded its ring slot's event (chain_done runs at the successful end), but a kernel
        // that reads the pinned candidates may be in flight: drain the device before the slot can be staged again (error path only)
        (void)hipDeviceSynchronize();
    }
    return s;
}

// batches larger than one launch chain takes (chain_limit) run as consecutive chains
template <class F>
iba_status chunked(const iba_handle* h, int B, F f) {
    const int cap = chain_limit(h);
    for (int b0 = 0; b0 < B; b0 += cap) { const iba_status s = f(b0, std::min(cap, B - b0)); if (s != IBA_OK) return s; }
    return IBA_OK;
}

iba_status eval_cost_partial_impl(iba_handle* h, const double* x, int B, double* d_partials, hipStream_t st, const Cand* pre = nullptr) {
    if (!h || (!x && !pre) || B < 1 || B > chain_limit(h)) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments (a chain takes 1 .. max_chain_batch candidates)");
    HIP_TRY(h, hipSetDevice(h->device));
    Cand* dc = nullptr;
    iba_status s = stage_cands(h, x, B, st, &dc, pre, 0, nullptr, true); if (s != IBA_OK) return chain_status(h, s);   // the cost tuple never reads the derivatives
    return chain_status(h, run_split(h, dc, B, 2, false, false, d_partials, st));
}

}  // namespace

extern "C" {

int32_t iba_abi_version(void) { return IBA_ABI_VERSION; }
const char* iba_last_error(const iba_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }
int64_t iba_num_points(const iba_handle* h) { return h ? h->n_points : 0; }
int64_t iba_num_keypoints(const iba_handle* h) { return h ? h->n_keypoints : 0; }

void iba_destroy(iba_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->frames.release(); h->slots.release(); h->xs.release(); h->ys.release(); h->zs.release(); h->perm.release(); h->inv_perm.release(); h->chunk_box.release(); h->pts4.release();
    h->d_ffr.release(); h->d_kp_c.release(); h->d_kp_rec.release(); h->d_scan_rec.release();
    h->nodes.release(); h->kp_uv.release(); h->kp_mp.release(); h->kp_fl.release(); h->coarse_start.release(); h->bitmap.release(); h->crec.release();
    h->match_uv.release(); h->plane_cost.release(); h->plane_local.release(); h->plane_ok.release(); h->scratch_cost.release(); h->scratch_local.release(); h->d_assoc_frozen.release(); h->d_flist.release(); h->d_flist_frozen.release(); h->d_fcount.release(); h->d_fcount_frozen.release(); h->d_cands.release(); h->d_frame_partials.release(); h->d_partials.release(); h->d_corr.release(); h->d_he.release(); h->d_lcount.release(); h->d_lcount_frozen.release(); h->d_nn_partials.release(); h->d_frefit.release(); h->d_pairs.release(); h->d_hard.release(); h->d_pcounts.release(); h->mpk.release(); h->fkp.release(); h->kp_fl2.release(); h->d_diag.release(); h->d_anchor.release();
    if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
    if (h->h_cands) (void)hipHostFree(h->h_cands);
    if (h->h_partials) (void)hipHostFree(h->h_partials);
    if (h->h_done) (void)hipHostFree(h->h_done);
    for (int i = 0; i < kRing; ++i) if (h->ring_ev[i]) (void)hipEventDestroy(h->ring_ev[i]);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->ev2) (void)hipEventDestroy(h->ev2);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

iba_status iba_create(const iba_problem_desc* d, const iba_params* params, int device, int32_t frame_begin, int32_t frame_end, iba_handle** out) {
    return iba_create_ex(d, params, device, frame_begin, frame_end, nullptr, out);
}

iba_status iba_create_ex(const iba_problem_desc* d, const iba_params* params, int device, int32_t frame_begin, int32_t frame_end, const iba_create_options* user_opt, iba_handle** out) {
    g_create_error.clear();
    iba_create_options opt;
    iba_default_create_options(&opt);
    if (user_opt) {   // (a caller compiled against an older, shorter struct: the fields it does not know keep their defaults)
        if (user_opt->struct_size < (int32_t)(2 * sizeof(int32_t)) || user_opt->struct_size > (int32_t)sizeof(opt)) return fail(nullptr, IBA_ERR_INVALID_ARG, "iba_create_options.struct_size is not set (use iba_default_create_options)");
        std::memcpy(&opt, user_opt, (size_t)user_opt->struct_size);
        opt.struct_size = (int32_t)sizeof(opt);
        if (opt.common_pairs < 0 || opt.common_pairs > 2 || opt.max_pair_groups < 1 || opt.max_pair_groups > kMaxPairGroups || !(opt.common_max_px >= 0) || !(opt.pair_inflation >= 1.0) ||
            !(opt.anchor_reach >= 0) || opt.pair_list_capacity < 0 || opt.pair_memo_max_batch < 0 || opt.max_chain_batch < 1 || opt.max_chain_batch > IBA_MAX_CHAIN)
            return fail(nullptr, IBA_ERR_INVALID_ARG, "iba_create_options: a field is out of range");
    }
    if (!d || !params || !out) return fail(nullptr, IBA_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    const int F = d->n_frames;
    if (F < 0 || frame_begin < 0 || frame_end < frame_begin || frame_end > F) return fail(nullptr, IBA_ERR_INVALID_ARG, "bad frame range");
    {
        iba_status ps = check_params(nullptr, *params);
        if (ps != IBA_OK) return ps;
    }
    // ---- validation: every CSR array of the descriptor (covisible keyframes outside the owned slice are dereferenced too) ----
    {
        auto monotonic = [](const uint64_t* o, int64_t n) { if (!o) return false; for (int64_t i = 0; i < n; ++i) if (o[i + 1] < o[i]) return false; return true; };
        if (F > 0 && (!d->pt_offset || !d->kp_offset || !d->covis_offset || !d->intrinsics || !d->Tcw || !d->Tc_next || !d->Tl_next)) { return fail(nullptr, IBA_ERR_INVALID_ARG, "null array in the problem descriptor"); }
        if (F > 0 && (!monotonic(d->pt_offset, F) || !monotonic(d->kp_offset, F) || !monotonic(d->covis_offset, F))) { return fail(nullptr, IBA_ERR_INVALID_ARG, "pt_offset / kp_offset / covis_offset must be non-decreasing"); }
        const uint64_t S = F > 0 ? d->covis_offset[F] : 0, Ntot = F > 0 ? d->pt_offset[F] : 0, Ktot = F > 0 ? d->kp_offset[F] : 0;
        if ((Ntot > 0 && !d->pts_xyz) || (Ktot > 0 && (!d->kp_uv || !d->kp_has_mappoint || !d->kp_mappoint_w))) { return fail(nullptr, IBA_ERR_INVALID_ARG, "null point / keypoint array in the problem descriptor"); }
        if (S > 0 && (!d->covis_frame || !d->covis_relpose || !d->match_offset || !monotonic(d->match_offset, (int64_t)S))) { return fail(nullptr, IBA_ERR_INVALID_ARG, "covisibility arrays missing or match_offset not non-decreasing"); }
        if (S > 0 && d->match_offset[S] > 0 && (!d->match_kp_ref || !d->match_kp_covis)) { return fail(nullptr, IBA_ERR_INVALID_ARG, "null match arrays in the problem descriptor"); }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, IBA_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(nullptr, IBA_ERR_NO_DEVICE, "device index out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return fail(nullptr, IBA_ERR_NO_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));

    iba_handle* h = new iba_handle;
    h->device = device; h->params = *params; to_dev_params(*params, h->dprm);
    h->global_frames = F; h->frame_begin = frame_begin; h->n_frames = frame_end - frame_begin;
    const int nf = h->n_frames;
    bool many_slots = false;   // some frame has more covisible keyframes than the flag word has match bits

    for (int f = frame_begin; f < frame_end; ++f) {
        const uint64_t P = d->pt_offset[f + 1] - d->pt_offset[f], K = d->kp_offset[f + 1] - d->kp_offset[f];
        if (P >= (1ull << 22) || K >= 65535ull) { delete h; return fail(nullptr, IBA_ERR_UNSUPPORTED, "scan (>= 2^22 points) or keypoint count (>= 65535) too large"); }
        const uint64_t ns = d->covis_offset[f + 1] - d->covis_offset[f];
        if (ns > (uint64_t)kMaxCovis) { delete h; return fail(nullptr, IBA_ERR_UNSUPPORTED, "more than 62 covisible keyframes per frame"); }
        if (ns > (uint64_t)kCovisWord) many_slots = true;
        h->max_slots = std::max<uint32_t>(h->max_slots, (uint32_t)ns);
    }


    // ---- per-frame host build (parallel over frames; reference: omp parallel for at iba_global.cpp:363) ----
    // keypoints are stored in Morton order of their pixel (internal id j -> reference id kp_order[j]): queries that
    // share a wave then walk neighbouring kd-tree leaves (coherent LDS reads, less lane divergence)
    struct FrameBuild { std::vector<uint32_t> idx; std::vector<TreeNode> nodes; KpGrid grid; uint32_t D = 0; std::vector<uint32_t> kp_order, kp_inv; std::vector<float> uv; };
    std::vector<FrameBuild> fb(nf);
    const double margin = h->dprm.bitmap_margin;
    parallel_for(nf, [&](int lf) {
        const int f = frame_begin + lf;
        const uint32_t P = (uint32_t)(d->pt_offset[f + 1] - d->pt_offset[f]);
        const uint32_t K = (uint32_t)(d->kp_offset[f + 1] - d->kp_offset[f]);
        FrameBuild& b = fb[lf];
        b.D = tree_depth_for(P);
        build_tree(d->pts_xyz + 3 * d->pt_offset[f], P, b.D, b.idx, b.nodes);
        const double* in = d->intrinsics + 6 * f;
        auto part1by1 = [](uint32_t v) { v &= 0xFFFFu; v = (v | (v << 8)) & 0x00FF00FFu; v = (v | (v << 4)) & 0x0F0F0F0Fu; v = (v | (v << 2)) & 0x33333333u; v = (v | (v << 1)) & 0x55555555u; return v; };
        const float* uv0 = d->kp_uv + 2 * d->kp_offset[f];
        // internal keypoint order = coarse cell of the keypoint grid (row-major), Morton order of the pixel inside a cell: the
        // grid's record e IS keypoint e, so the kernels keep (u, v) of all keypoints in LDS and need no id lookup; neighbours
        // in the list are neighbours in the image, and their MapPoint queries walk neighbouring kd leaves
        const int gw_ = (int)std::ceil(in[4] / kGridCell) + 3, gh_ = (int)std::ceil(in[5] / kGridCell) + 3;
        const uint32_t gwc_ = ((uint32_t)gw_ + (1u << kCoarseShift) - 1u) >> kCoarseShift;
        std::vector<std::pair<uint64_t, uint32_t>> key(K);
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t qx = (uint32_t)std::min(65535.f, std::max(0.f, uv0[2 * k] * 8.f)), qy = (uint32_t)std::min(65535.f, std::max(0.f, uv0[2 * k + 1] * 8.f));
            const uint32_t cell = (uint32_t)(grid_cell(uv0[2 * k + 1], gh_) >> kCoarseShift) * gwc_ + (uint32_t)(grid_cell(uv0[2 * k], gw_) >> kCoarseShift);
            key[k] = {((uint64_t)cell << 32) | (part1by1(qx) | (part1by1(qy) << 1)), k};
        }
        std::sort(key.begin(), key.end());
        b.kp_order.resize(K); b.kp_inv.resize(K); b.uv.resize(2 * (size_t)K);
        for (uint32_t j = 0; j < K; ++j) { const uint32_t k = key[j].second; b.kp_order[j] = k; b.kp_inv[k] = j; b.uv[2 * j] = uv0[2 * k]; b.uv[2 * j + 1] = uv0[2 * k + 1]; }
        build_kp_grid(b.uv.data(), K, in[4], in[5], margin, b.grid);
    });

    // ---- flatten ----
    std::vector<FrameHdr>& hdr = h->h_frames; hdr.resize(nf);
    uint64_t pt_base = 0, kp_base = 0, coarse_base = 0, bm_base = 0, match_base = 0, box_base = 0; uint32_t node_base = 0, slot_base = 0;
    for (int lf = 0; lf < nf; ++lf) {
        const int f = frame_begin + lf;
        FrameHdr& x = hdr[lf]; std::memset(&x, 0, sizeof(x));
        x.P = (uint32_t)(d->pt_offset[f + 1] - d->pt_offset[f]); x.Ppad = (x.P + 3u) & ~3u; x.pt_base = pt_base; pt_base += x.Ppad;
        x.box_base = box_base; box_base += (x.P + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;
        x.depth = fb[lf].D; x.node_base = node_base; node_base += (uint32_t)fb[lf].nodes.size();
        x.K = (uint32_t)(d->kp_offset[f + 1] - d->kp_offset[f]); x.kp_base = kp_base; kp_base += x.K;
        x.gw = fb[lf].grid.gw; x.gh = fb[lf].grid.gh; x.gwc = fb[lf].grid.gwc; x.ghc = fb[lf].grid.ghc;
        x.coarse_base = coarse_base; coarse_base += (uint64_t)x.gwc * x.ghc + 1;
        h->maxCoarse = std::max<uint32_t>(h->maxCoarse, x.gwc * x.ghc + 1);
        x.bitmap_base = bm_base; bm_base += fb[lf].grid.bitmap.size();
        x.n_slots = (uint32_t)(d->covis_offset[f + 1] - d->covis_offset[f]); x.slot_base = slot_base; slot_base += x.n_slots;
        x.match_base = match_base; match_base += (uint64_t)x.n_slots * x.K;
        const double* in = d->intrinsics + 6 * f;
        x.fx = in[0]; x.fy = in[1]; x.cx = in[2]; x.cy = in[3]; x.W = in[4]; x.H = in[5];
        for (int i = 0; i < 12; ++i) { x.Tcw[i] = (double)d->Tcw[12 * f + i]; x.Tc_next[i] = (double)d->Tc_next[12 * f + i]; x.Tl_next[i] = d->Tl_next[12 * f + i]; }
        x.he_valid = f < F - 1 ? 1 : 0; x.global_frame = f;
        h->maxP = std::max(h->maxP, x.P); h->maxPpad = std::max(h->maxPpad, x.Ppad); h->maxK = std::max(h->maxK, x.K);
        h->max_fx = std::max(h->max_fx, std::max(std::fabs(x.fx), std::fabs(x.fy)));
        h->maxNodes = std::max<uint32_t>(h->maxNodes, (uint32_t)fb[lf].nodes.size());
        h->maxBitmapWords = std::max<uint32_t>(h->maxBitmapWords, (uint32_t)fb[lf].grid.bitmap.size());
    }
    h->n_points = 0; for (auto& x : hdr) h->n_points += x.P;
    h->n_pt_total = (int64_t)pt_base;
    h->n_keypoints = (int64_t)kp_base;
    h->h_kp_off.resize(nf + 1); for (int lf = 0; lf < nf; ++lf) h->h_kp_off[lf] = hdr[lf].kp_base; h->h_kp_off[nf] = kp_base;

    const float qnan = std::numeric_limits<float>::quiet_NaN();
    std::vector<float> xs(pt_base, qnan), ys(pt_base, qnan), zs(pt_base, qnan);
    std::vector<uint32_t> perm(pt_base, 0u), inv_perm(pt_base, 0u);
    std::vector<float4> pts4(pt_base, float4{qnan, qnan, qnan, 0.f});
    std::vector<float> chunk_box(8 * (size_t)box_base, qnan);   // [chunk][8]: min xyz, -, max xyz, - (two 16-byte loads)
    std::vector<TreeNode> nodes(node_base);
    std::vector<float2> kp_uv(kp_base); std::vector<float4> kp_mp(kp_base), crec(kp_base);
    std::vector<uint32_t>& kp_ext = h->h_kp_ext; kp_ext.resize(kp_base);
    std::vector<uint32_t> coarse_start(coarse_base), bitmap(bm_base);
    std::vector<float2> match_uv(match_base, float2{qnan, qnan});
    std::vector<SlotHdr> slots(slot_base);
    std::vector<uint32_t> kp_fl(kp_base, 0u);   // flag word of every keypoint (see the match loop)
    std::vector<uint32_t> kp_fl2(many_slots ? (size_t)kp_base : 0, 0u);
    std::atomic<bool> bad_match(false);
    parallel_for(nf, [&](int lf) {
        const int f = frame_begin + lf; const FrameHdr& x = hdr[lf]; const FrameBuild& b = fb[lf];
        const float* src = d->pts_xyz + 3 * d->pt_offset[f];
        for (uint32_t i = 0; i < x.P; ++i) {
            const uint32_t o = b.idx[i];
            xs[x.pt_base + i] = src[3 * (size_t)o]; ys[x.pt_base + i] = src[3 * (size_t)o + 1]; zs[x.pt_base + i] = src[3 * (size_t)o + 2];
            perm[x.pt_base + i] = o; inv_perm[x.pt_base + o] = i;
            float ob; std::memcpy(&ob, &o, 4);
            pts4[x.pt_base + i] = float4{src[3 * (size_t)o], src[3 * (size_t)o + 1], src[3 * (size_t)o + 2], ob};
        }
        for (uint32_t c0 = 0; c0 < x.P; c0 += (uint32_t)kChunk) {   // static AABB of every kChunk consecutive tree positions
            float* bx = &chunk_box[8 * (size_t)(x.box_base + c0 / (uint32_t)kChunk)];
            float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = c0; i < std::min<uint32_t>(c0 + (uint32_t)kChunk, x.P); ++i) {
                const float v[3] = {xs[x.pt_base + i], ys[x.pt_base + i], zs[x.pt_base + i]};
                for (int a = 0; a < 3; ++a) { if (v[a] < mn[a]) mn[a] = v[a]; if (v[a] > mx[a]) mx[a] = v[a]; }   // NaN coordinates never narrow a box
            }
            for (int a = 0; a < 3; ++a) { bx[a] = mn[a]; bx[4 + a] = mx[a]; }
            float rmax = 0.f;   // largest |coordinate| of the box (slack term of the frustum test); NaN for an empty chunk
            for (int a = 0; a < 3; ++a) rmax = std::max(rmax, std::max(std::fabs(mn[a]), std::fabs(mx[a])));
            bx[7] = (mn[0] <= mx[0]) ? rmax : qnan;
        }
        std::copy(b.nodes.begin(), b.nodes.end(), nodes.begin() + x.node_base);
        const uint64_t k0 = d->kp_offset[f];
        for (uint32_t k = 0; k < x.K; ++k) {   // k = internal (Morton) id, e = reference id
            const uint64_t e = k0 + b.kp_order[k];
            kp_ext[x.kp_base + k] = b.kp_order[k];
            kp_uv[x.kp_base + k] = float2{d->kp_uv[2 * e], d->kp_uv[2 * e + 1]};
            const bool has = d->kp_has_mappoint[e] != 0;
            kp_mp[x.kp_base + k] = has ? float4{d->kp_mappoint_w[3 * e], d->kp_mappoint_w[3 * e + 1], d->kp_mappoint_w[3 * e + 2], 1.0f} : float4{0.f, 0.f, 0.f, 0.f};
            crec[x.kp_base + k] = float4{b.grid.crec[4 * (size_t)k], b.grid.crec[4 * (size_t)k + 1], b.grid.crec[4 * (size_t)k + 2], 0.f};
        }
        std::copy(b.grid.coarse_start.begin(), b.grid.coarse_start.end(), coarse_start.begin() + x.coarse_base);
        std::copy(b.grid.bitmap.begin(), b.grid.bitmap.end(), bitmap.begin() + x.bitmap_base);
        for (uint32_t sl = 0; sl < x.n_slots; ++sl) {
            const uint64_t gs = d->covis_offset[f] + sl;
            for (int i = 0; i < 12; ++i) slots[x.slot_base + sl].rel[i] = (double)d->covis_relpose[12 * gs + i];
            const int cf = d->covis_frame[gs];
            if (cf < 0 || cf >= F) { bad_match = true; continue; }
            const uint64_t ck0 = d->kp_offset[cf], cK = d->kp_offset[cf + 1] - ck0;
            for (uint64_t m = d->match_offset[gs]; m < d->match_offset[gs + 1]; ++m) {
                const int kr = d->match_kp_ref[m], kc = d->match_kp_covis[m];
                if (kr < 0 || (uint32_t)kr >= x.K || kc < 0 || (uint64_t)kc >= cK) { bad_match = true; continue; }
                match_uv[x.match_base + (uint64_t)sl * x.K + b.kp_inv[kr]] = float2{d->kp_uv[2 * (ck0 + kc)], d->kp_uv[2 * (ck0 + kc) + 1]};
                // flag word = 1 (owns a MapPoint) | 2 (matched in >= 1 covisible KF) | per-slot match bits << 2 (up to 30 slots)
                if (sl < (uint32_t)kCovisWord) kp_fl[x.kp_base + b.kp_inv[kr]] |= 2u | (4u << sl);
                else { kp_fl[x.kp_base + b.kp_inv[kr]] |= 2u; kp_fl2[x.kp_base + b.kp_inv[kr]] |= 1u << (sl - (uint32_t)kCovisWord); }
            }
        }
    });
    bool crec_ok = true;
    for (size_t k = 0; k < (size_t)kp_base; ++k) { if (kp_mp[k].w != 0.f) kp_fl[k] |= 1u; kp_mp[k].w = (float)(kp_fl[k] & 3u); }
    std::vector<uint32_t> mpk;   // per frame: the keypoints that own a MapPoint (the only ones a 1-NN search is ever run for)
    for (int lf = 0; lf < nf; ++lf) {
        hdr[lf].mpk_base = mpk.size();
        for (uint32_t k = 0; k < hdr[lf].K; ++k) if (kp_fl[hdr[lf].kp_base + k] & 1u) mpk.push_back(k);
        hdr[lf].n_mpk = (uint32_t)(mpk.size() - hdr[lf].mpk_base);
        h->max_mpk = std::max(h->max_mpk, hdr[lf].n_mpk);
    }
    std::vector<uint2> fkp;   // per frame: the keypoints whose flag word is not zero, ascending (the association tail's walk)
    for (int lf = 0; lf < nf; ++lf) {
        uint32_t cnt = 0;
        for (uint32_t k = 0; k < hdr[lf].K; ++k) { uint32_t idb; std::memcpy(&idb, &crec[hdr[lf].kp_base + k].z, 4); crec_ok = crec_ok && idb == k; }
        hdr[lf].fk_base = fkp.size();
        for (uint32_t k = 0; k < hdr[lf].K; ++k) if (kp_fl[hdr[lf].kp_base + k] != 0u) { fkp.push_back(make_uint2(k, kp_fl[hdr[lf].kp_base + k])); ++cnt; }
        hdr[lf].n_fk = cnt;
        while (fkp.size() & 3u) fkp.push_back(make_uint2(kNone, 0u));   // (every frame's list starts 32-byte aligned and is read two or four entries at a time)
        h->maxKw = std::max(h->maxKw, cnt);
    }
    fkp.resize(fkp.size() + 4, make_uint2(kNone, 0u));
    fb.clear(); fb.shrink_to_fit();
    if (bad_match) { delete h; return fail(nullptr, IBA_ERR_INVALID_ARG, "covisibility / match index out of range"); }
    if (!crec_ok) { delete h; return fail(nullptr, IBA_ERR_STATE, "internal: keypoint grid records are not in keypoint order"); }

    // ---- LDS plans ----
    // the options struct first, then the environment as a debug override (process-global: for A/B runs of an unmodified caller)
    h->common_mode = opt.common_pairs; h->common_max_px = opt.common_max_px; h->max_groups = opt.max_pair_groups; h->pair_memo = opt.pair_memo;
    h->pair_memo_max_b = opt.pair_memo_max_batch; h->pair_infl = opt.pair_inflation; h->nn_sets = opt.anchored_lists != 0; h->anchor_reach = opt.anchor_reach;
    h->spin_wait = opt.spin_wait != 0; h->factor_valu = opt.factor_mfma == 0;
    h->chain_fold = opt.chain_fold != 0; h->chain_cap = opt.max_chain_batch;
    if (const char* e = dbg_env("IBA_NN_ROUNDS")) h->nn_rounds = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_DONE_FLAG")) h->done_flag_on = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_NN_CG")) { h->nn_cg_max = std::max(1, std::min(kMaxGroup, std::atoi(e))); h->nn_cg_fixed = true; }
    if (const char* e = dbg_env("IBA_NN_DBG")) h->nn_dbg = std::atoi(e);
    if (const char* e = dbg_env("IBA_NN_LIST")) h->nn_list = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_NN_LIST_WORKERS")) h->nn_list_workers = std::max(0, std::atoi(e));
    if (const char* e = dbg_env("IBA_ASSOC_DBG")) h->assoc_dbg = std::atoi(e);
    if (const char* e = dbg_env("IBA_FACTOR_MFMA")) h->factor_valu = std::atoi(e) == 0;
    if (const char* e = dbg_env("IBA_JETS_FOLD")) h->jets_fold = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_FACTOR_REC")) h->factor_rec = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_FACTOR_V2")) h->factor_v2 = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_FACTOR_DBG")) h->factor_dbg = std::atoi(e);
    if (const char* e = dbg_env("IBA_FACTOR_WAVES_PER_CAND")) h->factor_waves_forced = std::max(0, std::atoi(e));
    if (const char* e = dbg_env("IBA_COMMON_PAIRS")) h->common_mode = std::atoi(e);
    if (const char* e = dbg_env("IBA_NN_SETS")) h->nn_sets = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_SPIN_WAIT")) h->spin_wait = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_ANCHOR_REACH")) h->anchor_reach = std::atof(e);
    if (const char* e = dbg_env("IBA_PAIR_BOUND")) h->pair_bound = std::atoi(e);
    if (const char* e = dbg_env("IBA_ASSOC2_FLREG")) h->assoc2_flreg_on = std::atoi(e);
    if (const char* e = dbg_env("IBA_PAIR_MEMO")) h->pair_memo = std::atoi(e);
    if (const char* e = dbg_env("IBA_PAIR_INFL")) h->pair_infl = std::max(1.0, std::atof(e));
    if (const char* e = dbg_env("IBA_PAIR_MEMO_MAX_B")) h->pair_memo_max_b = std::atoi(e);
    if (const char* e = dbg_env("IBA_COMMON_MIN_BATCH")) h->common_min_batch = std::max(1, std::atoi(e));
    if (const char* e = dbg_env("IBA_COMMON_MAX_PX")) h->common_max_px = std::atof(e);
    if (const char* e = dbg_env("IBA_PAIRS_DENSE_MIN")) h->pairs_dense_min = (uint32_t)std::max(0, std::atoi(e));
    if (const char* e = dbg_env("IBA_ASSOC2_THREADS")) { const int v = std::atoi(e); h->assoc2_threads_forced = (v == 256 || v == kThreads) ? v : 0; }
    if (const char* e = dbg_env("IBA_ASSOC2_SMALL_MIN")) h->assoc2_small_min_blocks = std::max(0, std::atoi(e));
    if (const char* e = dbg_env("IBA_PAIR_GROUPS")) h->max_groups = std::max(1, std::min(kMaxPairGroups, std::atoi(e)));
    if (const char* e = dbg_env("IBA_CHAIN_FOLD")) h->chain_fold = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_MAX_CHAIN")) h->chain_cap = std::atoi(e);
    h->chain_cap = std::max(1, std::min(h->chain_cap, (int)kMaxChain));
    if (!layout_assoc(h, h->alay)) { delete h; return fail(nullptr, IBA_ERR_UNSUPPORTED, "keypoints per frame exceed the LDS plan"); }
    if (!layout_assoc2(h, h->alay2) || 8u * (uint32_t)kPairStage + 160u + 2u * std::max(h->maxCoarse, 1u) + 8u * std::max(h->maxK, 1u) > kLdsBytes) h->common_mode = 0;
    if (dbg_env("IBA_LAYOUT_DEBUG")) std::fprintf(stderr, "[iba] assoc LDS: total %u B, queue %u entries, pairs %u, bitmap@%u; maxK %u maxKw %u\n", h->alay.total, h->alay.cand_cap, h->alay.pair_cap, h->alay.off_bitmap, h->maxK, h->maxKw);
    { NNLayout probe; if (!layout_nn(h, probe)) { delete h; return fail(nullptr, IBA_ERR_UNSUPPORTED, "kd tree exceeds the LDS plan of the search kernel"); } }
    h->lstride = std::max(1u, std::min(h->maxK, h->maxKw));
    // ... rounded up to an ODD count (r05): with 8 candidates per block a keyframe has 8 x NS search blocks, and a power of two of them per keyframe
    // is a pathology of the block -> XCD / memory-channel pattern — measured at 200 x 10 k points with the count forced: 4 slices 0.177 ms, 5: 0.118,
    // 6: 0.129, 7 (what this shape needs): 0.120, 8: 0.184, 9: 0.126; at 40 x 120 k points (needs 8) 0.169 -> 0.161 ms with 9. The extra slice's
    // blocks return at once.
    h->nn_ns = (int)((h->lstride + kSliceW - 1u) / kSliceW) | 1;
    // one-wave search blocks while a block's LDS plan (the tree's nodes + ~3 KB) lets a CU hold 16 of them (see kNNThreadsSmall)
    h->nn_small = 8u * std::max(h->maxNodes, 1u) <= 6144u;
    if (const char* e = dbg_env("IBA_NN_SMALL")) h->nn_small = std::atoi(e) != 0;
    if (const char* e = dbg_env("IBA_NN_SMALL_MIN_B")) h->nn_small_min_b = std::max(1, std::atoi(e));

    auto bail = [&](const char* what, hipError_t er) { std::string m = std::string(what) + ": " + hipGetErrorString(er); iba_destroy(h); return fail(nullptr, IBA_ERR_HIP, m); };
#define UP(buf, vec) do { hipError_t _e = h->buf.upload(vec); if (_e != hipSuccess) return bail("upload " #buf, _e); } while (0)
    UP(frames, hdr); UP(slots, slots); UP(xs, xs); UP(ys, ys); UP(zs, zs); UP(perm, perm); UP(inv_perm, inv_perm); UP(nodes, nodes); UP(chunk_box, chunk_box); UP(pts4, pts4);
    h->h_kp_uv.resize(2 * (size_t)kp_base);
    for (size_t k = 0; k < (size_t)kp_base; ++k) { h->h_kp_uv[2 * k] = kp_uv[k].x; h->h_kp_uv[2 * k + 1] = kp_uv[k].y; }
    UP(kp_uv, kp_uv); UP(kp_mp, kp_mp); UP(kp_fl, kp_fl); UP(coarse_start, coarse_start); UP(crec, crec); UP(bitmap, bitmap); UP(match_uv, match_uv); UP(mpk, mpk); UP(fkp, fkp);
    if (many_slots) UP(kp_fl2, kp_fl2);
    {   // iba_factor2_kernel's per-keyframe records and per-keypoint rays (iba_factor2_kernel.hpp)
        const uint32_t gstride = (uint32_t)kFfrHead + 12u * h->max_slots;
        std::vector<double> ffr((size_t)std::max(nf, 1) * gstride, 0.0);
        std::vector<double2> kp_c(kp_base);
        parallel_for(nf, [&](int lf) {
            const FrameHdr& x = hdr[lf];
            double* r = ffr.data() + (size_t)lf * gstride;
            r[0] = x.fx; r[1] = x.fy; r[2] = x.cx; r[3] = x.cy;
            for (int i = 0; i < 12; ++i) r[4 + i] = x.Tcw[i];
            const uint64_t u64s[3] = {x.kp_base, x.pt_base, x.match_base};
            std::memcpy(&r[16], u64s, 24);
            const uint32_t u32s[2] = {x.K, x.n_slots};
            std::memcpy(&r[19], u32s, 8);
            for (uint32_t sl = 0; sl < x.n_slots; ++sl) for (int i = 0; i < 12; ++i) r[kFfrHead + 12 * sl + i] = slots[x.slot_base + sl].rel[i];
            for (uint32_t k = 0; k < x.K; ++k) {   // (u0 - cx) / fx with u0 the float pixel widened: the quotient IBA_PlaneFactor forms (IBACalib2.hpp:165-166)
                const float2 uv = kp_uv[x.kp_base + k];
                kp_c[x.kp_base + k] = double2{((double)uv.x - x.cx) / x.fx, ((double)uv.y - x.cy) / x.fy};
            }
        });
        UP(d_ffr, ffr); UP(d_kp_c, kp_c);
        {   // the factor kernel's keypoint records: ray, first three matches (as load_match_pre reads them), flag word, MapPoint — one cache line
            std::vector<KpRec> kr(kp_base);
            parallel_for(nf, [&](int lf) {
                const FrameHdr& x = hdr[lf];
                for (uint32_t k = 0; k < x.K; ++k) {
                    KpRec& r = kr[x.kp_base + k];
                    r.cxz = kp_c[x.kp_base + k].x; r.cyz = kp_c[x.kp_base + k].y;
                    float2 m[3];
                    for (uint32_t i = 0; i < 3u; ++i) m[i] = x.n_slots ? match_uv[x.match_base + (uint64_t)std::min(i, x.n_slots - 1u) * x.K + k] : float2{qnan, qnan};
                    r.m0u = m[0].x; r.m0v = m[0].y; r.m1u = m[1].x; r.m1v = m[1].y; r.m2u = m[2].x; r.m2v = m[2].y;
                    r.fl = kp_fl[x.kp_base + k]; r.pad0 = 0u;
                    r.mpx = kp_mp[x.kp_base + k].x; r.mpy = kp_mp[x.kp_base + k].y; r.mpz = kp_mp[x.kp_base + k].z; r.pad1 = 0.f;
                }
            });
            UP(d_kp_rec, kr);
        }
        // LDS plan of one wave: prefix sums | derivative halves of the candidate | three block queues | keyframe ring ; the final reduction (21 x 65 doubles) over queues + ring
        F2Layout& L = h->f2lay;
        L.ffr_stride = gstride; L.ring_stride = (uint32_t)kFfrHead + (uint32_t)kFfrSlotRing * h->max_slots;
        L.ring_slots = 8u; while (L.ring_slots > 2u && L.ring_slots * L.ring_stride * 8u > 8192u) L.ring_slots >>= 1;
        L.off_pre = 0u; L.off_cand = align_up(4u * ((uint32_t)nf + 1u), 16u); L.off_q = L.off_cand + 90u * 8u;
        L.off_ring = L.off_q + 3u * kF2Queue * 8u; L.off_tr = L.off_q;
        L.total = std::max(L.off_ring + L.ring_slots * L.ring_stride * 8u, L.off_tr + 21u * 65u * 8u);
        if (L.total > 64u * 1024u) L.total = 0u;   // (a handle with that many keyframes or covisible slots keeps the one-record-per-keyframe kernel)
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && cus > 0) { h->factor_slots = cus * 8; h->n_cus = cus; }
    }
#undef UP
    hipError_t er;
    if ((er = h->plane_cost.alloc(pt_base)) != hipSuccess) return bail("alloc plane_cost", er);
    if ((er = h->d_cands.alloc((size_t)kRing * h->chain_cap)) != hipSuccess) return bail("alloc cands", er);
    h->nfb = nf; h->nrec = nf + h->nfb;   // one factor-kernel record per frame
    if ((er = h->d_assoc_frozen.alloc((size_t)std::max<int64_t>(h->n_keypoints, 1))) != hipSuccess) return bail("alloc assoc", er);
    if ((er = h->d_flist_frozen.alloc((size_t)std::max(nf, 1) * h->lstride)) != hipSuccess) return bail("alloc flist", er);
    if ((er = h->d_diag.alloc(32)) != hipSuccess || (er = hipMemset(h->d_diag.p, 0, 128)) != hipSuccess) return bail("alloc diag", er);   // [0..3] counters, [8..23] eight 64-bit cycle sums (diag build)
    if ((er = h->d_fcount_frozen.alloc((size_t)std::max(nf, 1))) != hipSuccess) return bail("alloc fcount", er);
    if ((er = h->d_lcount_frozen.alloc((size_t)std::max(nf, 1))) != hipSuccess) return bail("alloc lcount", er);
    if ((er = hipEventCreate(&h->ev_mid)) != hipSuccess) return bail("hipEventCreate", er);
    if (h->nn_sets && h->max_mpk > 0) {
        // 512 B per (frame, keypoint): 205 MB at 200 x 2000 keypoints, linear in the keyframes. The lists are an
        // optional memo: when the allocation fails the handle runs without them (every lane searches the tree), it does not fail.
        h->anchor_set_elems = (((size_t)std::max(nf, 1) * std::max(h->maxK, 1u) * kAnchorRowBytes + sizeof(SetPt) - 1) / sizeof(SetPt) + 1 + 7) / 8 * 8;   // (a multiple of 384 B: every set starts 128-byte aligned)
        if (h->d_anchor.alloc((size_t)kAnchorSets * h->anchor_set_elems) != hipSuccess) { (void)hipGetLastError(); h->d_anchor.p = nullptr; h->d_anchor.n = 0; h->nn_sets = false; }
    }
    if (h->common_mode > 0) {   // common lists of one batch: (scan point, keypoint) pairs and hard points per frame
        // the pairs of a batch grow with the scan density (points per pixel) and with the batch's spread: 4 per keypoint serve 10 k-point
        // scans, a full KITTI scan (120 k points) needs ~8 (r03: 14 k pairs per keyframe at the bench spread). A full list only costs speed.
        h->pair_cap = (int)std::min<uint32_t>(65536u, std::max<uint32_t>(std::max<uint32_t>(2048u, 4u * h->maxK), h->maxP / 4u));
        h->hard_cap = 1024;
        if (opt.pair_list_capacity > 0) { h->pair_cap = opt.pair_list_capacity; h->hard_cap = std::max(1, opt.pair_list_capacity / 8); }
        if (const char* e = dbg_env("IBA_DEBUG_PAIR_CAP")) { h->pair_cap = std::max(1, std::atoi(e)); h->hard_cap = std::max(1, std::atoi(e) / 8); }   // tests: force the overflow path
        h->pair_cap = std::min(h->pair_cap, 65536);   // (iba_assoc2_kernel notes pair numbers as u16)
        // kMaxPairGroups list slots (one group of candidates each); two counter sets per slot, used in turn (the pair search clears the set of the slot's NEXT build)
        if ((er = h->d_pairs.alloc((size_t)kMaxPairGroups * std::max(nf, 1) * h->pair_cap)) != hipSuccess) return bail("alloc pairs", er);
        if ((er = h->d_hard.alloc((size_t)kMaxPairGroups * std::max(nf, 1) * h->hard_cap)) != hipSuccess) return bail("alloc hard list", er);
        if ((er = h->d_pcounts.alloc(2 * (size_t)kMaxPairGroups * std::max(nf, 1) * kCountStride)) != hipSuccess) return bail("alloc pair counts", er);
        if ((er = hipMemset(h->d_pcounts.p, 0, sizeof(uint32_t) * 2 * (size_t)kMaxPairGroups * std::max(nf, 1) * kCountStride)) != hipSuccess) return bail("clear pair counts", er);
    }
    if ((er = h->d_partials.alloc((size_t)h->chain_cap * kPartialStride)) != hipSuccess) return bail("alloc partials", er);
    if ((er = h->d_corr.alloc((size_t)std::max<int64_t>(h->n_keypoints, 1))) != hipSuccess) return bail("alloc corr", er);
    if ((er = hipHostMalloc((void**)&h->h_cands, sizeof(Cand) * kRing * h->chain_cap)) != hipSuccess) return bail("hipHostMalloc", er);
    if ((er = hipHostMalloc((void**)&h->h_partials, sizeof(double) * h->chain_cap * kPartialStride)) != hipSuccess) return bail("hipHostMalloc", er);
    if ((er = hipHostMalloc((void**)&h->h_done, 64)) != hipSuccess) return bail("hipHostMalloc", er);
    *h->h_done = 0ull;
    if ((er = hipHostGetDevicePointer((void**)&h->h_done_dev, h->h_done, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", er);
    if ((er = h->d_done_ctr.alloc(1)) != hipSuccess || (er = hipMemset(h->d_done_ctr.p, 0, sizeof(uint32_t))) != hipSuccess) return bail("done counter", er);
    if ((er = hipHostGetDevicePointer((void**)&h->h_partials_dev, h->h_partials, 0)) != hipSuccess || (er = hipHostGetDevicePointer((void**)&h->h_cands_dev, h->h_cands, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", er);
    if ((er = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", er);
    if ((er = hipEventCreate(&h->ev0)) != hipSuccess || (er = hipEventCreate(&h->ev1)) != hipSuccess || (er = hipEventCreate(&h->ev2)) != hipSuccess) return bail("hipEventCreate", er);
    for (int i = 0; i < kRing; ++i) if ((er = hipEventCreateWithFlags(&h->ring_ev[i], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", er);
    // > 64 KB of dynamic LDS must be opted into per kernel
    for (const void* fn : {(const void*)iba_assoc_kernel<0>, (const void*)iba_assoc_kernel<2>, (const void*)iba_assoc_kernel<4>})
        if ((er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->alay.total)) != hipSuccess) return bail("hipFuncSetAttribute", er);
    if (h->common_mode > 0 && (er = hipFuncSetAttribute((const void*)iba_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes)) != hipSuccess) return bail("hipFuncSetAttribute", er);
    {
        const void* a2[12] = {(const void*)iba_assoc2_kernel<2, false, kThreads>, (const void*)iba_assoc2_kernel<2, true, kThreads>, (const void*)iba_assoc2_kernel<4, false, kThreads>, (const void*)iba_assoc2_kernel<4, true, kThreads>,
                              (const void*)iba_assoc2_kernel<0, false, kThreads>, (const void*)iba_assoc2_kernel<0, true, kThreads>,
                              (const void*)iba_assoc2_kernel<2, false, 256>, (const void*)iba_assoc2_kernel<2, true, 256>, (const void*)iba_assoc2_kernel<4, false, 256>, (const void*)iba_assoc2_kernel<4, true, 256>,
                              (const void*)iba_assoc2_kernel<0, false, 256>, (const void*)iba_assoc2_kernel<0, true, 256>};
        for (const void* fn : a2) if (h->common_mode > 0 && (er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->alay2.total)) != hipSuccess) return bail("hipFuncSetAttribute", er);
    }
    if ((er = hipFuncSetAttribute((const void*)iba_anchor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes)) != hipSuccess) return bail("hipFuncSetAttribute", er);
    const void* nfns[15] = {(const void*)iba_nn_kernel<1, 0, 0>, (const void*)iba_nn_kernel<2, 0, 0>, (const void*)iba_nn_kernel<3, 0, 0>, (const void*)iba_nn_kernel<1, 1, 0>, (const void*)iba_nn_kernel<2, 1, 0>,
                            (const void*)iba_nn_kernel<3, 1, 0>, (const void*)iba_nn_kernel<1, 2, 0>, (const void*)iba_nn_kernel<2, 2, 0>, (const void*)iba_nn_kernel<3, 2, 0>,
                            (const void*)iba_nn_kernel<1, 0, 1>, (const void*)iba_nn_kernel<2, 0, 1>, (const void*)iba_nn_kernel<3, 0, 1>, (const void*)iba_nn_kernel<1, 1, 1>, (const void*)iba_nn_kernel<2, 1, 1>,
                            (const void*)iba_nn_kernel<3, 1, 1>};
    for (const void* fn : nfns)
        if ((er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes)) != hipSuccess) return bail("hipFuncSetAttribute", er);
    const void* sfns[6] = {(const void*)iba_nn_kernel<1, 0, 0, kNNThreadsSmall>, (const void*)iba_nn_kernel<2, 0, 0, kNNThreadsSmall>, (const void*)iba_nn_kernel<3, 0, 0, kNNThreadsSmall>,
                           (const void*)iba_nn_kernel<1, 0, 1, kNNThreadsSmall>, (const void*)iba_nn_kernel<2, 0, 1, kNNThreadsSmall>, (const void*)iba_nn_kernel<3, 0, 1, kNNThreadsSmall>};
    for (const void* fn : sfns)
        if ((er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes)) != hipSuccess) return bail("hipFuncSetAttribute", er);
    const void* lfns[6] = {(const void*)iba_nn_list_kernel<1, 2>, (const void*)iba_nn_list_kernel<2, 2>, (const void*)iba_nn_list_kernel<3, 2>, (const void*)iba_nn_list_kernel<1, 4>, (const void*)iba_nn_list_kernel<2, 4>, (const void*)iba_nn_list_kernel<3, 4>};
    for (const void* fn : lfns)
        if ((er = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes)) != hipSuccess) return bail("hipFuncSetAttribute", er);

    {   // the association kernel's view of its kernarg segment (see iba_kernarg_probe_kernel)
        DevBuf<int32_t> okb; std::vector<int32_t> z(1, 0);
        er = okb.upload(z);
        if (er != hipSuccess) return bail("kernarg probe", er);
        hipLaunchKernelGGL(iba_kernarg_probe_kernel, dim3(1), dim3(1), 0, h->stream, KArgs{h->dev_problem(), h->dprm, h->alay}, okb.p);
        int32_t okv = 0;
        er = hipStreamSynchronize(h->stream);
        if (er == hipSuccess) er = hipMemcpy(&okv, okb.p, sizeof(okv), hipMemcpyDeviceToHost);
        okb.release();
        if (er != hipSuccess) return bail("kernarg probe", er);
        if (!okv) { iba_destroy(h); return fail(nullptr, IBA_ERR_UNSUPPORTED, "kernel arguments do not start at offset 0 of the kernarg segment"); }
    }
    iba_status ps = compute_plane_cache(h);
    if (ps != IBA_OK) { g_create_error = h->err; iba_destroy(h); return ps; }
    *out = h;
    return IBA_OK;
}

iba_status iba_set_params(iba_handle* h, const iba_params* p) {
    if (!h || !p) return IBA_ERR_INVALID_ARG;
    iba_status s = check_params(h, *p); if (s != IBA_OK) return s;
    HIP_TRY(h, hipSetDevice(h->device));
    if (p->max_pixel_dist != h->params.max_pixel_dist && h->bitmap.n) {
        // the 1-bit reject bitmap of the per-candidate association is the keypoints dilated by max_pixel_dist (+ the float slack):
        // rebuilt here from the host copy of the keypoints. The coarse CSR and the grid dimensions do not depend on it.
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::vector<uint32_t> bm(h->bitmap.n, 0u);
        std::atomic<bool> bad(false);
        const double margin = p->max_pixel_dist + 0.45;
        parallel_for(h->n_frames, [&](int lf) {
            const FrameHdr& x = h->h_frames[lf];
            KpGrid g;
            build_kp_grid(h->h_kp_uv.data() + 2 * x.kp_base, x.K, x.W, x.H, margin, g);
            if (g.gw != x.gw || g.gh != x.gh || x.bitmap_base + g.bitmap.size() > bm.size()) { bad = true; return; }
            std::copy(g.bitmap.begin(), g.bitmap.end(), bm.begin() + x.bitmap_base);
        });
        if (bad) return fail(h, IBA_ERR_STATE, "internal: keypoint grid changed shape");
        HIP_TRY(h, hipMemcpy(h->bitmap.p, bm.data(), sizeof(uint32_t) * bm.size(), hipMemcpyHostToDevice));
    }
    h->params = *p; to_dev_params(*p, h->dprm); h->frozen_valid = false;
    for (bool& v : h->anchor_valid) v = false;   // the lists carry the planes' verdicts under the old parameters
    for (auto& ps : h->pslot) ps.valid = false;    // the pair lists were cut for the old max_pixel_dist
    return compute_plane_cache(h);
}

iba_status iba_set_timing(iba_handle* h, int32_t enable) { if (!h) return IBA_ERR_INVALID_ARG; h->timing = enable != 0; return IBA_OK; }
iba_status iba_last_kernel_ms(iba_handle* h, float* frame_kernel_ms, float* total_ms) {
    if (!h) return IBA_ERR_INVALID_ARG;
    if (h->timing && h->timing_recorded) {   // events of the last launch (any stream): wait for them, then read
        HIP_TRY(h, hipEventSynchronize(h->ev2));
        HIP_TRY(h, hipEventElapsedTime(&h->last_frame_ms, h->ev0, h->ev1));
        HIP_TRY(h, hipEventElapsedTime(&h->last_total_ms, h->ev0, h->ev2));
        if (h->timing_split) { HIP_TRY(h, hipEventElapsedTime(&h->last_assoc_ms, h->ev0, h->ev_mid)); HIP_TRY(h, hipEventElapsedTime(&h->last_nn_ms, h->ev_mid, h->ev1)); }
        else { h->last_assoc_ms = h->last_frame_ms; h->last_nn_ms = 0.f; }
    }
    if (frame_kernel_ms) *frame_kernel_ms = h->last_frame_ms;
    if (total_ms) *total_ms = h->last_total_ms;
    return IBA_OK;
}

iba_status iba_last_phase_ms(iba_handle* h, float* assoc_kernel_ms, float* nn_kernel_ms, float* rest_ms) {
    if (!h) return IBA_ERR_INVALID_ARG;
    float fk = 0.f, tot = 0.f;
    iba_status s = iba_last_kernel_ms(h, &fk, &tot); if (s != IBA_OK) return s;
    if (assoc_kernel_ms) *assoc_kernel_ms = h->last_assoc_ms;
    if (nn_kernel_ms) *nn_kernel_ms = h->last_nn_ms;
    if (rest_ms) *rest_ms = tot - fk;
    return IBA_OK;
}

iba_status iba_eval_cost_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream) {
    if (!h || !d_partials || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) { return eval_cost_partial_impl(h, x + 7 * b0, Bc, (double*)d_partials + (size_t)b0 * kPartialStride, stream ? (hipStream_t)stream : h->stream); });
}

iba_status iba_eval_cost(iba_handle* h, const double* x, int32_t B, iba_cost_out* out) {
    if (!h || !out || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) {
        iba_status s = eval_cost_partial_impl(h, x + 7 * b0, Bc, h->h_partials_dev, h->stream); if (s != IBA_OK) return s;   // the sums land in pinned host memory: no copy behind the last kernel
        HIP_TRY(h, wait_done(h, h->stream));
        if (h->timing) { HIP_TRY(h, hipEventElapsedTime(&h->last_frame_ms, h->ev0, h->ev1)); HIP_TRY(h, hipEventElapsedTime(&h->last_total_ms, h->ev0, h->ev2)); }
        return iba_finalize_cost(&h->params, h->h_partials, Bc, out + b0);
    });
}

iba_status iba_eval_bbo(iba_handle* h, const double* x, int32_t B, double he_threshold, double valid_rate, iba_bbo* out) {
    if (!h || !out || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments");
    return chunked(h, B, [&](int b0, int Bc) {
        std::vector<iba_cost_out> cv((size_t)Bc); iba_cost_out* c = cv.data();
        iba_status s = iba_eval_cost(h, x + 7 * b0, Bc, c); if (s != IBA_OK) return s;
        for (int b = 0; b < Bc; ++b) {   // iba_global.cpp:386-392
            out[b0 + b].f = c[b].f1 * h->params.err_weight[0] + c[b].f2 * h->params.err_weight[1];
            out[b0 + b].c1 = c[b].C - he_threshold; out[b0 + b].c2 = -c[b].C - he_threshold;
            out[b0 + b].c3 = valid_rate - static_cast<double>(c[b].valid_cnt_3d_2d) / (c[b].cnt_3d_2d + 1);
        }
        return IBA_OK;
    });
}

iba_status iba_get_correspondences(iba_handle* h, const double* x, int32_t frame, uint32_t* kp_idx, uint32_t* pt_idx, int32_t cap, int32_t* n_out) {
    if (!h || !x || !n_out) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    const int lf = frame - h->frame_begin;
    if (lf < 0 || lf >= h->n_frames) return fail(h, IBA_ERR_INVALID_ARG, "frame not owned by this handle");
    HIP_TRY(h, hipSetDevice(h->device));
    Cand* dc = nullptr;
    { iba_status es = ensure_lists(h, 1, h->stream); if (es != IBA_OK) return es; }
    iba_status s = stage_cands(h, x, 1, h->stream, &dc); if (s != IBA_OK) return s;
    {
        hipLaunchKernelGGL(iba_assoc_kernel<0>, dim3(8 * ((h->n_frames + 7) / 8)), dim3(kThreads), h->alay.total, h->stream, KArgs{h->dev_problem(), h->dprm, h->alay}, dc, 1, 0,
                           h->d_frame_partials.p, h->n_frames, h->d_corr.p, h->d_flist_frozen.p, h->d_fcount_frozen.p, h->d_lcount_frozen.p, (int)h->lstride, (const uint4*)nullptr, (uint4*)nullptr, 0u);
        HIP_TRY(h, hipGetLastError());
    }
    const uint64_t k0 = h->h_kp_off[lf], K = h->h_kp_off[lf + 1] - k0;
    std::vector<uint32_t> tmp(K);
    HIP_TRY(h, hipMemcpyAsync(tmp.data(), h->d_corr.p + k0, sizeof(uint32_t) * K, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    for (uint64_t k = 0; k < K; ++k)
        if (tmp[k] != kNone) pairs.emplace_back(h->h_kp_ext[k0 + k], tmp[k]);
    std::sort(pairs.begin(), pairs.end());   // corrset is ordered by keypoint id (iba_global.cpp:85-95)
    int n = 0;
    for (auto const& pr : pairs) { if (n < cap && kp_idx && pt_idx) { kp_idx[n] = pr.first; pt_idx[n] = pr.second; } ++n; }
    *n_out = n;
    return IBA_OK;
}

iba_status iba_debug_plane(iba_handle* h, int32_t frame, uint32_t point, int32_t which, double out5[5], int32_t* k) {
    if (!h || !out5 || !k) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    const int lf = frame - h->frame_begin;
    if (lf < 0 || lf >= h->n_frames || point >= h->h_frames[lf].P) return fail(h, IBA_ERR_INVALID_ARG, "frame / point out of range");
    // plane_cache = 0: the planes are fitted per evaluation; slot 0 of the scratch records holds those of the FROZEN problem
    // (iba_build_problem), the ones iba_eval_residuals / iba_eval_factors read — only the local planes (which = 1) exist there
    if (!h->params.plane_cache && (which != 1 || !h->frozen_valid || !h->scratch_cost.p)) return fail(h, IBA_ERR_STATE, "plane_cache = 0: only the local planes of a frozen problem (which = 1 after iba_build_problem)");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    const FrameHdr& fh = h->h_frames[lf];
    uint32_t pos = 0;
    HIP_TRY(h, hipMemcpy(&pos, h->inv_perm.p + fh.pt_base + point, sizeof(pos), hipMemcpyDeviceToHost));
    const PlaneRec* src = !h->params.plane_cache ? (h->scratch_local_aliases ? h->scratch_cost.p : h->scratch_local.p)
                        : (which == 0 || h->plane_local_aliases_cost) ? h->plane_cost.p : h->plane_local.p;
    PlaneRec rec;
    HIP_TRY(h, hipMemcpy(&rec, src + fh.pt_base + pos, sizeof(rec), hipMemcpyDeviceToHost));
    out5[0] = rec.nx; out5[1] = rec.ny; out5[2] = rec.nz; out5[3] = rec.reg_sum; out5[4] = rec.far_d2; *k = rec.k;
    return IBA_OK;
}

// ---- (R, t, s) <-> x (iba_global.cpp:511-515: x[0:6] = g2o::SE3Quat(R, t).log(), x[6] = scale; Sim3Exp g2o_tools.h:105-140) ----
iba_status iba_sim3_to_x(const double rigid12[12], double scale, double x[7]) {
    if (!rigid12 || !x) return IBA_ERR_INVALID_ARG;
    const double R[9] = {rigid12[0], rigid12[1], rigid12[2], rigid12[4], rigid12[5], rigid12[6], rigid12[8], rigid12[9], rigid12[10]};
    const double t[3] = {rigid12[3], rigid12[7], rigid12[11]};
    dev_se3log(R, t, x);
    x[6] = scale;
    return IBA_OK;
}
iba_status iba_x_to_sim3(const double x[7], double rigid12[12], double* scale) {
    if (!rigid12 || !x || !scale) return IBA_ERR_INVALID_ARG;
    double R[9], t[3];
    se3_exp<double>(x, R, t);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) rigid12[r * 4 + c] = R[r * 3 + c]; rigid12[r * 4 + 3] = t[r]; }
    *scale = x[6];
    return IBA_OK;
}

// debug: host copy of the last summed partial blocks (B x iba_partial_stride() doubles)
iba_status iba_debug_last_partials(iba_handle* h, double* out, int32_t B) {
    if (!h || !out || B < 1 || B > h->chain_cap) return IBA_ERR_INVALID_ARG;   // the last launch chain's block
    std::memcpy(out, h->h_partials, sizeof(double) * B * kPartialStride);
    return IBA_OK;
}

// debug: 1 when the last evaluation chain shared the 2d-3d pair search over the batch (iba_pairs_kernel + iba_assoc2_kernel)
int32_t iba_debug_last_path(const iba_handle* h) { return h ? h->last_path : -1; }

// debug: how many times the anchored neighbour lists have been (re)built on this handle
// entries of the last evaluation (all candidates) that the anchored neighbour lists left to the tree search (-1: no search ran)
double iba_debug_nn_left_to_tree(iba_handle* h) {
    if (!h || h->last_nn_nrec <= 0) return -1.0;
    std::vector<double> v((size_t)h->last_nn_B * h->last_nn_nrec * kNNPartial);
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1.0;
    if (hipMemcpy(v.data(), h->d_nn_partials.p, v.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    double t = 0; for (size_t i = 5; i < v.size(); i += kNNPartial) t += v[i];
    if (dbg_env("IBA_DEBUG_LEFT_HIST")) {   // blocks by the number of entries they searched in the tree
        const int edges[] = {0, 1, 9, 17, 33, 65, 129, 257, 1 << 30};
        int hist[8] = {0};
        for (size_t i = 5; i < v.size(); i += kNNPartial) for (int k = 0; k < 8; ++k) if (v[i] >= edges[k] && v[i] < edges[k + 1]) ++hist[k];
        std::fprintf(stderr, "left-over entries per block: 0:%d 1-8:%d 9-16:%d 17-32:%d 33-64:%d 65-128:%d 129-256:%d more:%d\n", hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7]);
    }
    return t;
}
// mean length of the keyframes' common pair lists of the last evaluation that shared the pair search (-1: none has)
double iba_debug_mean_pairs(iba_handle* h) {
    if (!h || !h->d_pcounts.p || h->last_mean_pairs_slot < 0 || h->n_frames == 0) return -1.0;
    std::vector<uint32_t> v((size_t)h->n_frames * kCountStride);
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1.0;
    if (hipMemcpy(v.data(), h->d_pcounts.p + h->amap.cnt_off[h->last_mean_pairs_slot], v.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    double t = 0; for (int f = 0; f < h->n_frames; ++f) t += v[(size_t)f * kCountStride];
    return t / h->n_frames;
}
// debug: of the (list slot, frame) pair lists the last call's candidates read, how many had overflowed (a list or a hard list that did not
// hold everything: its blocks rescan every scan point exactly — speed only) and the largest list; out3 = {overflowed lists, lists, longest list}
iba_status iba_debug_pair_lists(iba_handle* h, int32_t out3[3]) {
    if (!h || !out3) return IBA_ERR_INVALID_ARG;
    out3[0] = out3[1] = out3[2] = 0;
    if (!h->d_pcounts.p || h->n_frames == 0 || h->last_path == 0) return IBA_OK;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, IBA_ERR_HIP, "sync");
    bool used[kMaxPairGroups] = {false, false, false, false};
    for (int b = 0; b < kMaxChain; ++b) if (h->amap.slot[b] < kMaxPairGroups) used[h->amap.slot[b]] = true;   // (stale entries beyond the last batch only add slots)
    std::vector<uint32_t> v((size_t)h->n_frames * kCountStride);
    for (int sl = 0; sl < kMaxPairGroups; ++sl) {
        if (!used[sl]) continue;
        if (hipMemcpy(v.data(), h->d_pcounts.p + h->amap.cnt_off[sl], v.size() * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) return fail(h, IBA_ERR_HIP, "copy");
        for (int f = 0; f < h->n_frames; ++f) { out3[0] += v[(size_t)f * kCountStride + 2] != 0u; ++out3[1]; out3[2] = std::max<int32_t>(out3[2], (int32_t)v[(size_t)f * kCountStride]); }
    }
    return IBA_OK;
}
// debug: (candidate, keyframe) association blocks, since the last reset, that took a speed-only fallback — a full candidate queue or pair
// list, a fifth hit of one point, an overflowed common list: every scan point again, exactly (results unaffected). reset != 0 clears it.
int64_t iba_debug_rescans(iba_handle* h, int32_t reset) {
    uint32_t v[4] = {0, 0, 0, 0};
    if (iba_debug_counters(h, v, reset) != IBA_OK) return -1;
    return (int64_t)v[0];
}
iba_status iba_debug_counters(iba_handle* h, uint32_t out4[4], int32_t reset) {
    if (!h || !h->d_diag.p || !out4) return IBA_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, IBA_ERR_HIP, "sync");
    if (hipMemcpy(out4, h->d_diag.p, 16, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, IBA_ERR_HIP, "copy");
    if (reset && hipMemset(h->d_diag.p, 0, 16) != hipSuccess) return fail(h, IBA_ERR_HIP, "memset");
    return IBA_OK;
}
int32_t iba_debug_last_assoc2_threads(const iba_handle* h) { return h ? h->last_assoc2_threads : -1; }
int32_t iba_debug_factor_ranges(const iba_handle* h, int32_t B) { return h ? factor_waves(h, B) : -1; }
// debug (diag build only: make -C csrc diag): eight 64-bit sums the search kernel's thread 0 of every block adds up — cycles per phase, blocks
iba_status iba_debug_phase_cycles(iba_handle* h, uint64_t out8[8], int32_t reset) {
    if (!h || !h->d_diag.p || !out8) return IBA_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, IBA_ERR_HIP, "sync");
    if (hipMemcpy(out8, h->d_diag.p + 8, 64, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, IBA_ERR_HIP, "copy");
    if (reset && hipMemset(h->d_diag.p + 8, 0, 64) != hipSuccess) return fail(h, IBA_ERR_HIP, "memset");
    return IBA_OK;
}
iba_status iba_debug_phase_cycles12(iba_handle* h, uint64_t out12[12], int32_t reset) {   // (iba_nn_list_kernel: slots 8-11 split its picks)
    if (!h || !h->d_diag.p || !out12) return IBA_ERR_INVALID_ARG;
    if (hipSetDevice(h->device) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return fail(h, IBA_ERR_HIP, "sync");
    if (hipMemcpy(out12, h->d_diag.p + 8, 96, hipMemcpyDeviceToHost) != hipSuccess) return fail(h, IBA_ERR_HIP, "copy");
    if (reset && hipMemset(h->d_diag.p + 8, 0, 96) != hipSuccess) return fail(h, IBA_ERR_HIP, "memset");
    return IBA_OK;
}
int32_t iba_debug_last_nn_threads(const iba_handle* h) { return h ? h->last_nn_threads : -1; }   // threads per block of the last search launch (64: one-wave blocks, 256)
int32_t iba_debug_last_nn_list(const iba_handle* h) { return h ? h->last_nn_list : -1; }   // workers per (XCD, group) of the last search launch if it was iba_nn_list_kernel's (0: iba_nn_kernel)
int32_t iba_debug_pairs_builds(const iba_handle* h) { return h ? h->pairs_builds : -1; }
int32_t iba_debug_anchor_builds(const iba_handle* h) { return h ? h->anchor_builds : -1; }

// debug: exact 1-NN of n LiDAR-frame queries in the scan of a local frame, through the frame kernels' own search
iba_status iba_debug_nn(iba_handle* h, int32_t frame, const double* q, int32_t n, int32_t mode, uint32_t* out_idx, double* out_d2) {
    if (!h || !q || !out_idx || !out_d2 || n < 1 || frame < 0 || frame >= h->n_frames) return IBA_ERR_INVALID_ARG;
    if (mode < 1 || mode > 4) return fail(h, IBA_ERR_INVALID_ARG, "mode must be 1 (association query), 2 (cost query), 3 or 4 (both, the query as the first / the second)");
    (void)hipSetDevice(h->device);
    DevBuf<double> dq, dd; DevBuf<uint32_t> di;
    std::vector<double> hq(q, q + 3 * (size_t)n);
    hipError_t er = dq.upload(hq);
    if (er == hipSuccess) er = dd.alloc((size_t)n);
    if (er == hipSuccess) er = di.alloc((size_t)n);
    if (er != hipSuccess) return fail(h, IBA_ERR_HIP, hipGetErrorString(er));
    const size_t lds = 8u * (size_t)std::max(h->maxNodes, 1u);
    const int blocks = (int)(((size_t)n + 255) / 256);
    if (er == hipSuccess) er = hipMemsetAsync(di.p, 0xFF, sizeof(uint32_t) * (size_t)n, h->stream);   // (an empty scan: the kernel writes nothing, every query answers "none")
    hipLaunchKernelGGL(iba_nn_probe_kernel, dim3(blocks), dim3(256), lds, h->stream, h->dev_problem(), frame, dq.p, n, mode, di.p, dd.p);
    er = hipStreamSynchronize(h->stream);
    if (er == hipSuccess) er = hipMemcpy(out_idx, di.p, sizeof(uint32_t) * n, hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(out_d2, dd.p, sizeof(double) * n, hipMemcpyDeviceToHost);
    dq.release(); dd.release(); di.release();
    if (er != hipSuccess) return fail(h, IBA_ERR_HIP, hipGetErrorString(er));
    return IBA_OK;
}

// GeoCalib.h:18-33 on the evaluation path's own kd search (see the header)
iba_status iba_geo_correspondences(iba_handle* h, int32_t frame, const double* src_xyz, int32_t n_src, double max_distance, uint32_t* out_src, uint32_t* out_tgt, int32_t* n_out) {
    if (!h || !n_out || n_src < 0 || frame < 0 || frame >= h->n_frames || (n_src > 0 && (!src_xyz || !out_src || !out_tgt))) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments");
    *n_out = 0;
    if (n_src == 0) return IBA_OK;
    std::vector<uint32_t> idx((size_t)n_src); std::vector<double> d2((size_t)n_src);
    const iba_status s = iba_debug_nn(h, frame, src_xyz, n_src, 1, idx.data(), d2.data());
    if (s != IBA_OK) return s;
    int32_t n = 0;
    for (int32_t i = 0; i < n_src; ++i)   // num_res > 0 && sq_dist[0] <= maxDistance (GeoCalib.h:29; an empty target cloud answers "none")
        if (idx[(size_t)i] != 0xFFFFFFFFu && d2[(size_t)i] <= max_distance) { out_src[n] = (uint32_t)i; out_tgt[n] = idx[(size_t)i]; ++n; }
    *n_out = n;
    return IBA_OK;
}

// debug: the wall time of a blocking entry point as a C caller sees it — `iters` back-to-back calls of iba_eval_cost (kind 0), iba_eval_full (1) or
// iba_eval_factors (2) on the same candidates, timed around each call with the steady clock here, no language binding in the clock
iba_status iba_debug_call_latency(iba_handle* h, const double* x, int32_t B, int32_t kind, int32_t iters, double* median_ms, double* min_ms) {
    if (!h || !x || !median_ms || B < 1 || iters < 1 || iters > 100000 || kind < 0 || kind > 2) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments");
    std::vector<iba_cost_out> c((size_t)B); std::vector<iba_normal_out> n((size_t)B); std::vector<double> t((size_t)iters);
    for (int i = -3; i < iters; ++i) {   // three calls that do not count
        const auto t0 = std::chrono::steady_clock::now();
        const iba_status s = kind == 0 ? iba_eval_cost(h, x, B, c.data()) : (kind == 1 ? iba_eval_full(h, x, B, c.data(), n.data()) : iba_eval_factors(h, x, B, n.data()));
        if (s != IBA_OK) return s;
        if (i >= 0) t[(size_t)i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(t.begin(), t.end());
    *median_ms = t[t.size() / 2];
    if (min_ms) *min_ms = t[0];
    return IBA_OK;
}

// debug: div2 (the projections' two quotients by one depth, iba_kernels.hpp) beside the compiler's f64 division, on the caller's operands
iba_status iba_debug_div2_selftest(int32_t device, const double* num0, const double* num1, const double* den, int64_t n, double* q0, double* q1, double* ref0, double* ref1, int64_t* n_fast) {
    if (!num0 || !num1 || !den || !q0 || !q1 || !ref0 || !ref1 || !n_fast || n < 1 || n > (1ll << 28)) return fail(nullptr, IBA_ERR_INVALID_ARG, "bad arguments");
    hipError_t er = hipSetDevice(device);
    if (er != hipSuccess) return fail(nullptr, IBA_ERR_HIP, hipGetErrorString(er));
    DevBuf<double> d[7]; DevBuf<unsigned long long> dn;
    for (auto& b : d) if (er == hipSuccess) er = b.alloc((size_t)n);
    if (er == hipSuccess) er = dn.alloc(1);
    const double* src[3] = {num0, num1, den};
    for (int i = 0; i < 3 && er == hipSuccess; ++i) er = hipMemcpy(d[i].p, src[i], sizeof(double) * (size_t)n, hipMemcpyHostToDevice);
    if (er == hipSuccess) er = hipMemset(dn.p, 0, sizeof(unsigned long long));
    if (er == hipSuccess) {
        hipLaunchKernelGGL(iba_div2_selftest_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d[0].p, d[1].p, d[2].p, (long long)n, d[3].p, d[4].p, d[5].p, d[6].p, dn.p);
        er = hipGetLastError();
    }
    if (er == hipSuccess) er = hipDeviceSynchronize();
    double* dst[4] = {q0, q1, ref0, ref1};
    for (int i = 0; i < 4 && er == hipSuccess; ++i) er = hipMemcpy(dst[i], d[3 + i].p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost);
    unsigned long long nf = 0;
    if (er == hipSuccess) er = hipMemcpy(&nf, dn.p, sizeof(nf), hipMemcpyDeviceToHost);
    for (auto& b : d) b.release();
    dn.release();
    if (er != hipSuccess) return fail(nullptr, IBA_ERR_HIP, hipGetErrorString(er));
    *n_fast = (int64_t)nf;
    return IBA_OK;
}

// debug: the kNN lists of the plane fits (fit_list_rows, the list builder of iba_plane_kernel / iba_fit_kernel) around n scan points
// (ORIGINAL indices) of a local frame: up to k neighbours with d^2 < r2 each, nearest first (the point itself first, at 0)
iba_status iba_debug_knn(iba_handle* h, int32_t frame, const uint32_t* points, int32_t n, int32_t k, double r2, uint32_t* out_idx, double* out_d2, int32_t* out_cnt) {
    if (!h || !points || !out_idx || !out_d2 || !out_cnt || n < 1 || frame < 0 || frame >= h->n_frames) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments");
    if (k < 1 || k > 64 || !(r2 > 0)) return fail(h, IBA_ERR_INVALID_ARG, "k must be in [1, 64] and r2 > 0 (INFINITY: no radius clip)");
    HIP_TRY(h, hipSetDevice(h->device));
    const FrameHdr& fh = h->h_frames[frame];
    const uint32_t P = fh.P;
    for (int i = 0; i < n; ++i) if (points[i] >= P) return fail(h, IBA_ERR_INVALID_ARG, "point index out of range");
    DevBuf<uint32_t> di; DevBuf<double> dd; DevBuf<int32_t> dc;
    hipError_t er = di.alloc((size_t)P * k);
    if (er == hipSuccess) er = dd.alloc((size_t)P * k);
    if (er == hipSuccess) er = dc.alloc((size_t)P);
    std::vector<uint32_t> inv(P), hi((size_t)P * k); std::vector<double> hd((size_t)P * k); std::vector<int32_t> hc(P);
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    if (er == hipSuccess) {
        const dim3 grid((P + 63) / 64);
        if (k <= 32) hipLaunchKernelGGL(iba_knn_dump_kernel<2>, grid, dim3(64), 0, h->stream, h->dev_problem(), frame, r2, k, di.p, dd.p, dc.p);
        else hipLaunchKernelGGL(iba_knn_dump_kernel<4>, grid, dim3(64), 0, h->stream, h->dev_problem(), frame, r2, k, di.p, dd.p, dc.p);
        er = hipGetLastError();
    }
    if (er == hipSuccess) er = hipStreamSynchronize(h->stream);
    if (er == hipSuccess) er = hipMemcpy(inv.data(), h->inv_perm.p + fh.pt_base, sizeof(uint32_t) * P, hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(hi.data(), di.p, sizeof(uint32_t) * hi.size(), hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(hd.data(), dd.p, sizeof(double) * hd.size(), hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(hc.data(), dc.p, sizeof(int32_t) * P, hipMemcpyDeviceToHost);
    di.release(); dd.release(); dc.release();
    if (er != hipSuccess) return fail(h, IBA_ERR_HIP, hipGetErrorString(er));
    for (int i = 0; i < n; ++i) {
        const uint32_t pos = inv[points[i]];
        out_cnt[i] = hc[pos];
        for (int j = 0; j < k; ++j) {
            const bool have = j < hc[pos];
            out_idx[(size_t)i * k + j] = have ? hi[(size_t)pos * k + j] : kNone;
            out_d2[(size_t)i * k + j] = have ? hd[(size_t)pos * k + j] : -1.0;
        }
    }
    return IBA_OK;
}

static iba_status eval_normal_partial_impl(iba_handle* h, const double* x, int B, double* d_partials, hipStream_t st, const Cand* pre = nullptr, const std::atomic<int>* pre_flag = nullptr) {
    if (!h || (!x && !pre) || B < 1 || B > chain_limit(h)) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments (a chain takes 1 .. max_chain_batch candidates)");
    HIP_TRY(h, hipSetDevice(h->device));
    { iba_status es = ensure_lists(h, B, st); if (es != IBA_OK) return es; }
    Cand* dc = nullptr;
    iba_status s = stage_cands(h, x, B, st, &dc, pre, 2, pre_flag, true); if (s != IBA_OK) return chain_status(h, s);
    return chain_status(h, run_split(h, dc, B, 1, false, true, d_partials, st));
}

iba_status iba_eval_normal_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream) {
    if (!h || !d_partials || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) { return eval_normal_partial_impl(h, x + 7 * b0, Bc, (double*)d_partials + (size_t)b0 * kPartialStride, stream ? (hipStream_t)stream : h->stream); });
}

iba_status iba_eval_normal(iba_handle* h, const double* x, int32_t B, iba_normal_out* out) {
    if (!h || !out || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) {
        iba_status s = eval_normal_partial_impl(h, x + 7 * b0, Bc, h->h_partials_dev, h->stream); if (s != IBA_OK) return s;
        HIP_TRY(h, wait_done(h, h->stream));
        if (h->timing) { HIP_TRY(h, hipEventElapsedTime(&h->last_frame_ms, h->ev0, h->ev1)); HIP_TRY(h, hipEventElapsedTime(&h->last_total_ms, h->ev0, h->ev2)); }
        return iba_finalize_normal(&h->params, h->h_partials, Bc, out + b0);
    });
}

// BAError tuple AND re-associated normal equations of the same candidates from ONE pass over the scans:
// the two paths share projection + 2d-3d association (iba_global.cpp:55-96 = iba_local.cpp:17-58).
static iba_status eval_full_partial_impl(iba_handle* h, const double* x, int B, double* d_partials, hipStream_t st, const Cand* pre = nullptr, const std::atomic<int>* pre_flag = nullptr) {
    if (!h || (!x && !pre) || B < 1 || B > chain_limit(h)) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments (a chain takes 1 .. max_chain_batch candidates)");
    HIP_TRY(h, hipSetDevice(h->device));
    { iba_status es = ensure_lists(h, B, st); if (es != IBA_OK) return es; }
    Cand* dc = nullptr;
    iba_status s = stage_cands(h, x, B, st, &dc, pre, 2, pre_flag, true); if (s != IBA_OK) return chain_status(h, s);
    return chain_status(h, run_split(h, dc, B, 3, false, true, d_partials, st));
}

iba_status iba_eval_full_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream) {
    if (!h || !d_partials || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) { return eval_full_partial_impl(h, x + 7 * b0, Bc, (double*)d_partials + (size_t)b0 * kPartialStride, stream ? (hipStream_t)stream : h->stream); });
}

iba_status iba_eval_full(iba_handle* h, const double* x, int32_t B, iba_cost_out* cost, iba_normal_out* normal) {
    if (!h || !cost || !normal || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) {
        iba_status s = eval_full_partial_impl(h, x + 7 * b0, Bc, h->h_partials_dev, h->stream); if (s != IBA_OK) return s;
        HIP_TRY(h, wait_done(h, h->stream));
        s = iba_finalize_cost(&h->params, h->h_partials, Bc, cost + b0); if (s != IBA_OK) return s;
        return iba_finalize_normal(&h->params, h->h_partials, Bc, normal + b0);
    });
}

static iba_status build_problem_impl(iba_handle* h, const double* x, const Cand* pre) {
    if (!h || (!x && !pre)) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    HIP_TRY(h, hipSetDevice(h->device));
    Cand* dc = nullptr;
    iba_status s = stage_cands(h, x, 1, h->stream, &dc, pre); if (s != IBA_OK) return s;
    s = run_split(h, dc, 1, 1, true, false, h->h_partials_dev, h->stream); if (s != IBA_OK) return s;
    h->done_armed = false;   // (waited for on the stream: the flag this chain publishes is nobody's to poll)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->frozen_frames = (int32_t)h->h_partials[P_FRAMES_N]; h->frozen_ncorr = (int32_t)h->h_partials[P_NCORR_N]; h->frozen_valid = true;
    return IBA_OK;
}
iba_status iba_build_problem(iba_handle* h, const double* x) { return build_problem_impl(h, x, nullptr); }

static iba_status eval_factors_partial_impl(iba_handle* h, const double* x, int B, double* d_partials, hipStream_t st, const Cand* pre = nullptr) {
    if (!h || (!x && !pre) || B < 1 || B > chain_limit(h)) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments");
    if (!h->frozen_valid) return fail(h, IBA_ERR_STATE, "iba_eval_factors called before iba_build_problem");
    HIP_TRY(h, hipSetDevice(h->device));
    { iba_status es = ensure_lists(h, B, st); if (es != IBA_OK) return es; }
    Cand* dc = nullptr;
    iba_status s = stage_cands(h, x, B, st, &dc, pre); if (s != IBA_OK) return s;
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev0, st));
    const int nfr = factor_records(h, B);   // (records per candidate: the stride of this launch's records and the count of its sums)
    s = launch_factors(h, dc, B, 0, h->d_frame_partials.p, nfr, 0, st); if (s != IBA_OK) return s;
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev1, st));
    hipLaunchKernelGGL(iba_reduce_kernel, dim3(B), dim3(kReduceThreads), 0, st, h->d_frame_partials.p, nfr, d_partials);
    HIP_TRY(h, hipGetLastError());
    // the frozen problem's frame / correspondence counts (iba_build_problem) ride in their slots of the block
    hipLaunchKernelGGL(iba_set_slots_kernel, dim3((B + 63) / 64), dim3(64), 0, st, d_partials, B, (int)P_FRAMES_N, (double)h->frozen_frames, (int)P_NCORR_N, (double)h->frozen_ncorr);
    HIP_TRY(h, hipGetLastError());
    if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev2, st)); h->timing_recorded = true; h->timing_split = false; }
    return IBA_OK;
}

iba_status iba_eval_factors_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream) {
    if (!h || !d_partials || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) { return eval_factors_partial_impl(h, x + 7 * b0, Bc, (double*)d_partials + (size_t)b0 * kPartialStride, stream ? (hipStream_t)stream : h->stream); });
}

iba_status iba_eval_factors(iba_handle* h, const double* x, int32_t B, iba_normal_out* out) {
    if (!h || !out || !x || B < 1) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    return chunked(h, B, [&](int b0, int Bc) {
        iba_status s = eval_factors_partial_impl(h, x + 7 * b0, Bc, h->h_partials_dev, h->stream); if (s != IBA_OK) return s;
        HIP_TRY(h, wait_stream(h, h->stream));   // (this chain ends in iba_set_slots_kernel, not in the summing kernel that publishes the flag of wait_done)
        if (h->timing) { HIP_TRY(h, hipEventElapsedTime(&h->last_frame_ms, h->ev0, h->ev1)); HIP_TRY(h, hipEventElapsedTime(&h->last_total_ms, h->ev0, h->ev2)); }
        return iba_finalize_normal(&h->params, h->h_partials, Bc, out + b0);
    });
}

iba_status iba_eval_whitened(iba_handle* h, const double* x, double r[8], double J[56]) {
    if (!h || !x || !r || !J) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    iba_normal_out n;
    iba_status s = iba_eval_factors(h, x, 1, &n); if (s != IBA_OK) return s;
    return iba_whiten_normal(&n, r, J);
}

iba_status iba_calibrate_lm(iba_handle* h, const double* x0, const iba_lm_options* opt, iba_lm_result* res) {
    if (!h || !x0 || !res) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    LmOptions o;
    if (opt) {
        o.max_outer_iterations = opt->max_outer_iterations; o.max_inner_iterations = opt->max_inner_iterations; o.min_diff = opt->min_diff;
        o.function_tolerance = opt->function_tolerance; o.gradient_tolerance = opt->gradient_tolerance; o.parameter_tolerance = opt->parameter_tolerance;
        o.initial_trust_region_radius = opt->initial_trust_region_radius;
    }
    iba_status st = IBA_OK;
    LmResult r;
    const bool ok = calibrate_lm(x0, o,
        [&](const double* x) { st = iba_build_problem(h, x); return st == IBA_OK; },
        [&](const double* x, double* H, double* g, double& cost) {
            iba_normal_out n; st = iba_eval_factors(h, x, 1, &n);
            if (st != IBA_OK) return false;
            std::memcpy(H, n.H, sizeof(n.H)); std::memcpy(g, n.b, sizeof(n.b)); cost = n.cost;
            return true;
        }, r);
    if (!ok) return st == IBA_OK ? IBA_ERR_STATE : st;
    std::memcpy(res->x, r.x, sizeof(r.x));
    res->outer_iterations = r.outer_iterations; res->inner_iterations = r.inner_iterations; res->evaluations = r.evaluations; res->converged = r.converged;
    res->initial_cost = r.initial_cost; res->final_cost = r.final_cost;
    return IBA_OK;
}

iba_status iba_calibrate_mads_record(iba_handle* h, const double* x0, const iba_mads_options* opt, iba_mads_result* res, double* trace, int32_t cap, int32_t* n_trace,
                                     int32_t* batch_sizes, int32_t cap_batches, int32_t* n_batches) {
    if (!h || !x0 || !res) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    iba_mads_options dflt;
    if (!opt) { iba_default_mads_options(x0, &dflt); opt = &dflt; }
    if (!mads_options_ok(opt)) return fail(h, IBA_ERR_INVALID_ARG, "bad MADS options (bounds, frame sizes, budget)");
    MadsOptions o; to_mads(opt, o);
    std::vector<double> tr;
    if (trace || n_trace) o.trace = &tr;
    std::vector<int> bs;
    if (batch_sizes || n_batches) o.batch_sizes = &bs;
    iba_status st = IBA_OK;
    MadsResult r;
    const bool ok = mads_minimize(x0, o, [&](const double* X, int B, MadsPoint* out) {
        iba_bbo bbo[IBA_MAX_BATCH];
        st = iba_eval_bbo(h, X, B, opt->he_threshold, opt->valid_rate, bbo);
        if (st != IBA_OK) return false;
        for (int b = 0; b < B; ++b) { out[b].f = bbo[b].f; out[b].c[0] = bbo[b].c1; out[b].c[1] = bbo[b].c2; out[b].c[2] = bbo[b].c3; }
        return true;
    }, r);
    if (!ok) return st == IBA_OK ? IBA_ERR_STATE : st;
    from_mads(r, res);
    hand_over_trace(tr, trace, cap, n_trace);
    if (n_batches) *n_batches = (int32_t)bs.size();
    if (batch_sizes && cap_batches > 0) std::memcpy(batch_sizes, bs.data(), sizeof(int32_t) * (size_t)std::min<int32_t>((int32_t)bs.size(), cap_batches));
    return IBA_OK;
}
iba_status iba_calibrate_mads_trace(iba_handle* h, const double* x0, const iba_mads_options* opt, iba_mads_result* res, double* trace, int32_t cap, int32_t* n_trace) {
    return iba_calibrate_mads_record(h, x0, opt, res, trace, cap, n_trace, nullptr, 0, nullptr);
}
iba_status iba_calibrate_mads(iba_handle* h, const double* x0, const iba_mads_options* opt, iba_mads_result* res) {
    return iba_calibrate_mads_trace(h, x0, opt, res, nullptr, 0, nullptr);
}
iba_status iba_eval_residuals(iba_handle* h, const double* x, double* r, double* J, int32_t* block_id, int32_t* block_kind, int64_t* n_rows) {
    if (!h || !x || !n_rows) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    if (!h->frozen_valid) return fail(h, IBA_ERR_STATE, "iba_eval_residuals called before iba_build_problem");
    HIP_TRY(h, hipSetDevice(h->device));
    const size_t KT = (size_t)h->n_keypoints;
    std::vector<uint2> a(KT);
    {   // the association is kept as per-frame block lists: expand them to one row per keypoint
        std::fill(a.begin(), a.end(), make_uint2(kNone, kNone));
        std::vector<uint32_t> fcn((size_t)std::max(h->n_frames, 1));
        HIP_TRY(h, hipMemcpy(fcn.data(), h->d_fcount_frozen.p, sizeof(uint32_t) * (size_t)h->n_frames, hipMemcpyDeviceToHost));
        std::vector<uint4> row(h->lstride);
        for (int lf = 0; lf < h->n_frames; ++lf) {
            const uint32_t n = fcn[lf];
            if (!n) continue;
            HIP_TRY(h, hipMemcpy(row.data(), h->d_flist_frozen.p + (size_t)lf * h->lstride, sizeof(uint4) * n, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n; ++i) a[h->h_frames[lf].kp_base + row[i].x] = make_uint2(row[i].y, row[i].z);
        }
        if (KT) HIP_TRY(h, hipMemcpy(h->d_assoc_frozen.p, a.data(), KT * sizeof(uint2), hipMemcpyHostToDevice));
    }
    std::vector<float2> muv; std::vector<float2> dummy;
    // covisible-match counts per keypoint decide the number of plane-factor rows (2 per matched covisible KF)
    std::vector<int> nconv(KT, 0);
    {
        std::vector<float2> m(h->match_uv.n);
        if (h->match_uv.n) HIP_TRY(h, hipMemcpy(m.data(), h->match_uv.p, h->match_uv.n * sizeof(float2), hipMemcpyDeviceToHost));
        for (int lf = 0; lf < h->n_frames; ++lf) {
            const FrameHdr& fh = h->h_frames[lf];
            for (uint32_t sl = 0; sl < fh.n_slots; ++sl)
                for (uint32_t k = 0; k < fh.K; ++k) if (m[fh.match_base + (size_t)sl * fh.K + k].x == m[fh.match_base + (size_t)sl * fh.K + k].x) nconv[fh.kp_base + k]++;
        }
    }
    // rows ordered like the oracle / BuildProblem: frame, then reference keypoint id, plane factor before the 3d-3d factor
    std::vector<long long> row_off(KT, -1);
    std::vector<int32_t> bid, bkind;
    long long rows = 0; int32_t blk = 0;
    for (int lf = 0; lf < h->n_frames; ++lf) {
        const FrameHdr& fh = h->h_frames[lf];
        std::vector<uint32_t> inv(fh.K);
        for (uint32_t j = 0; j < fh.K; ++j) inv[h->h_kp_ext[fh.kp_base + j]] = j;
        for (uint32_t e = 0; e < fh.K; ++e) {
            const size_t g = fh.kp_base + inv[e];
            if (a[g].x == kNone && a[g].y == kNone) continue;
            row_off[g] = rows;
            if (a[g].x != kNone && h->dprm.p2pix) { for (int i = 0; i < nconv[g]; ++i) { bid.push_back(blk); bid.push_back(blk); bkind.push_back(3); bkind.push_back(3); rows += 2; ++blk; } }   // one IBATestEdge per matched covisible keyframe
            else if (a[g].x != kNone) { const int nr = 2 * nconv[g]; for (int i = 0; i < nr; ++i) { bid.push_back(blk); bkind.push_back(0); } rows += nr; ++blk; }
            if (a[g].y != kNone) { const bool pl = (a[g].y >> 31) != 0; const int nr = pl ? 1 : 3; for (int i = 0; i < nr; ++i) { bid.push_back(blk); bkind.push_back(pl ? 1 : 2); } rows += nr; ++blk; }
        }
    }
    *n_rows = rows;
    if (!r || !J) return IBA_OK;
    if (rows == 0) return IBA_OK;
    DevBuf<long long> d_off; DevBuf<double> d_r, d_J;
    std::vector<long long> offv(row_off);
    HIP_TRY(h, d_off.upload(offv)); HIP_TRY(h, d_r.alloc((size_t)rows)); HIP_TRY(h, d_J.alloc((size_t)rows * 7));
    Cand* dc = nullptr;
    iba_status s = stage_cands(h, x, 1, h->stream, &dc); if (s != IBA_OK) return s;
    hipLaunchKernelGGL(iba_residual_kernel, dim3((h->maxK + 63) / 64, h->n_frames), dim3(64), 0, h->stream, h->dev_problem(), h->dprm, dc, h->d_assoc_frozen.p, d_off.p, d_r.p, d_J.p);
    HIP_TRY(h, hipGetLastError());
    HIP_TRY(h, hipMemcpyAsync(r, d_r.p, sizeof(double) * rows, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipMemcpyAsync(J, d_J.p, sizeof(double) * rows * 7, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    d_off.release(); d_r.release(); d_J.release();
    if (block_id) std::memcpy(block_id, bid.data(), sizeof(int32_t) * rows);
    if (block_kind) std::memcpy(block_kind, bkind.data(), sizeof(int32_t) * rows);
    return IBA_OK;
}

}  // extern "C"

// ---- library-internal interface of iba_group (iba_internal.hpp) ----
namespace iba {
void make_cands_jets_host(const double* x, int B, Cand* out) { for (int b = 0; b < B; ++b) make_cand_jets(x + 7 * b, out[b]); }
void make_cands_host(const double* x, int B, Cand* out, bool jets) { for (int b = 0; b < B; ++b) { make_cand_values(x + 7 * b, out[b]); if (jets) make_cand_jets(x + 7 * b, out[b]); } }
iba_status eval_partial_cands(iba_handle* h, const Cand* hc, int B, EvalKind kind, double* d_partials, hipStream_t st, const std::atomic<int>* jets_ready) {
    if (!h || !hc || !d_partials) return fail(h, IBA_ERR_INVALID_ARG, "null argument");
    if (!st) st = h->stream;
    switch (kind) {
        case kEvalCost: return eval_cost_partial_impl(h, nullptr, B, d_partials, st, hc);
        case kEvalNormal: return eval_normal_partial_impl(h, nullptr, B, d_partials, st, hc, jets_ready);
        case kEvalFull: return eval_full_partial_impl(h, nullptr, B, d_partials, st, hc, jets_ready);
        case kEvalFactors: return eval_factors_partial_impl(h, nullptr, B, d_partials, st, hc);
    }
    return fail(h, IBA_ERR_INVALID_ARG, "bad evaluation kind");
}
iba_status build_problem_cands(iba_handle* h, const Cand* hc) { return build_problem_impl(h, nullptr, hc); }
int chain_capacity(const iba_handle* h) { return h ? chain_limit(h) : 0; }
iba_status reserve_batch(iba_handle* h, int B) {
    if (!h || B < 1 || B > chain_limit(h)) return fail(h, IBA_ERR_INVALID_ARG, "bad arguments");   // (the evaluators' own limit: chain_limit)
    HIP_TRY(h, hipSetDevice(h->device));
    return ensure_lists(h, B, h->stream);
}
}  // namespace iba
