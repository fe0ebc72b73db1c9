/*
 * iba_mi355x_debug.h — diagnostics of the MI355X IBA evaluation path: timing probes, counters, the kernels' own searches and
 * plane fits exposed for parity tests, and the host-only self-tests of the optimiser logic. Test and benchmark tooling; none of
 * it is part of the drop-in surface (iba_mi355x.h), none of it changes a result, and an integrator never needs it.
 *
 * DEBUG-ONLY ENVIRONMENT VARIABLES (read at iba_create / iba_group_create, AND ONLY WHEN IBA_DEBUG_ENV=1 IS SET AS WELL — round 6: a stray
 * variable in an integrator's environment changes nothing; an integrator sets iba_create_options instead — where a variable has a field
 * there, the FIELD is the interface and the variable only overrides it for A/B runs of an unmodified caller):
 *
 *   variable                field of iba_create_options   meaning
 *   IBA_COMMON_PAIRS        common_pairs                  0 never share the 2d-3d pair search, 1 when tight, 2 always
 *   IBA_COMMON_MAX_PX       common_max_px                 nominal spread (px) up to which a batch shares one pair search
 *   IBA_PAIR_GROUPS         max_pair_groups               clustering of a wide batch into 1..4 tight groups
 *   IBA_PAIR_MEMO           pair_memo                     reuse of pair lists across calls
 *   IBA_PAIR_MEMO_MAX_B     pair_memo_max_batch           largest batch whose lists are built reusable
 *   IBA_PAIR_INFL           pair_inflation                inflation of a reusable list's bound
 *   IBA_NN_SETS             anchored_lists                anchored neighbour lists on / off
 *   IBA_ANCHOR_REACH        anchor_reach                  drift (m) that moves an anchor
 *   IBA_SPIN_WAIT           spin_wait                     polling wait at the end of a call
 *   IBA_FACTOR_MFMA         factor_mfma                   normal-equation sums on the matrix cores
 *   IBA_DEBUG_PAIR_CAP      pair_list_capacity            entries per pair list (tests force the overflow path)
 *   IBA_CHAIN_FOLD          chain_fold                    staging / reduction launches folded into their neighbours
 *   IBA_MAX_CHAIN           max_chain_batch               candidates one launch chain takes
 *   -- no field: pure diagnostics, results unaffected unless stated --
 *   IBA_NN_CG, IBA_PAIRS_DENSE_MIN, IBA_COMMON_MIN_BATCH, IBA_PAIR_BOUND, IBA_ASSOC2_FLREG, IBA_ASSOC2_THREADS (256 / 512 threads per
 *   block of the shared-pair association, else chosen per launch), IBA_ASSOC2_SMALL_MIN, IBA_ASSOC_BLOCKS, IBA_CAND_BYTES,
 *   IBA_PAIR_BYTES                                        launch-shape / LDS-plan knobs of single kernels (A/B timing)
 *   IBA_NN_ROUNDS                                         0: the entries the anchored lists leave over are searched leaf by leaf (rounds 3-4) instead of
 *                                                         in rounds of leaves (same results; A/B timing)
 *   IBA_DONE_FLAG                                         0: a blocking call polls its stream (rounds 3-4) instead of the sequence number the summing
 *                                                         kernel's last block publishes in pinned memory (same results; A/B timing)
 *   IBA_NN_SMALL, IBA_NN_SMALL_MIN_B                      0 / 1: the search kernel in blocks of four waves / of one wave whatever the size of the kd trees; from how many candidates per launch
 *   IBA_NN_LIST                                           1: the anchored lists are walked by the persistent list kernel, measured slower: csrc/iba_nn_list_kernel.hpp;
 *                                                         IBA_NN_LIST_WORKERS=n: its blocks per CU
 *   IBA_FACTOR_V2                                         1: the normal equations by iba_factor2_kernel — one wave per equal share of a candidate's whole work
 *                                                         list (csrc/iba_factor2_kernel.hpp) — instead of one wave per (keyframe, candidate); same sums to
 *                                                         summation order, measured no faster; IBA_FACTOR_WAVES_PER_CAND forces its ranges per candidate
 *   IBA_NN_DBG, IBA_ASSOC_DBG, IBA_FACTOR_DBG             cut a kernel short after a phase (timing attribution; RESULTS ARE GARBAGE)
 *   IBA_LAYOUT_DEBUG, IBA_DEBUG_LEFT_HIST                 print the LDS plan / a histogram of left-over searches to stderr
 *   IBA_GROUP_TIMEOUT_MS                                  bound (ms, default 20 000; x4 for a group's first call) of a device thread's
 *                                                         wait for its stream before the group is declared broken
 *   IBA_GROUP_REDUCE_HOST (flag of iba_group_create_ex, not a variable), IBA_DEBUG_FAIL_RANK / IBA_DEBUG_FAIL_PHASE: inject one
 *                                                         failure into a group's evaluation (tests of the fail-not-hang path)
 */
#ifndef IBA_MI355X_DEBUG_H
#define IBA_MI355X_DEBUG_H

#include "iba_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Introspection for benchmarks: device-side duration of the last evaluation's dominant kernel
 * measured with HIP events on the launch stream (ms), and the frame-kernel launch shape. */
iba_status iba_last_kernel_ms(iba_handle* h, float* frame_kernel_ms, float* total_ms);
iba_status iba_set_timing(iba_handle* h, int32_t enable);
/* the same split by kernel: association kernel (projection, 2d-3d association, 3d-2d residuals), grouped 1-NN search kernel
 * (3d-3d terms), and everything after them (factor kernel, sums) */
iba_status iba_last_phase_ms(iba_handle* h, float* assoc_kernel_ms, float* nn_kernel_ms, float* rest_ms);
/* debug (library built with -DIBA_DIAG_COUNTERS only, `make -C csrc diag`; zeros otherwise): cycle sums per phase of the search kernel (thread 0 of every block:
 * start-up, entries + MapPoints, list rows, picks, wait at the end of the list pass, left-over searches, sums) and the number of blocks */
iba_status iba_debug_phase_cycles(iba_handle* h, uint64_t out8[8], int32_t reset);
iba_status iba_debug_phase_cycles12(iba_handle* h, uint64_t out12[12], int32_t reset);   /* the same with iba_nn_list_kernel's four slots inside its picks */
/* debug: threads per block of the last iba_assoc2_kernel launch on this handle (256 or 512: chosen per launch from the number of (candidate,
 * keyframe) blocks; 0 = no shared-pair association has run) */
int32_t iba_debug_last_assoc2_threads(const iba_handle* h);
/* debug: ranges per candidate iba_factor2_kernel would cut a batch of B into on this handle; 0 = the batch runs on the one-wave-per-(keyframe,
 * candidate) factor kernel (the default) */
int32_t iba_debug_factor_ranges(const iba_handle* h, int32_t B);
/* debug: host copy of the summed partial blocks of the last iba_eval_* call */
iba_status iba_debug_last_partials(iba_handle* h, double* out, int32_t B);
/* debug: which 2d-3d association ran in the last evaluation chain: 1 = the batch shared one pair search per keyframe
 * (any batch, a single candidate included, whose nominal projection spread stays under IBA_COMMON_MAX_PX = 20 px), 0 = every
 * candidate searched for itself. The results are the same bits either way.
 * Environment (read at iba_create): IBA_COMMON_PAIRS=0 never share, 2 share whenever the bound allows; IBA_COMMON_MAX_PX;
 * IBA_PAIR_MEMO=0 no reuse of pair lists across calls, IBA_PAIR_MEMO_MAX_B (40) the largest batch that reuses, IBA_PAIR_INFL (1.25)
 * the inflation of a reusable list's bound; IBA_SIDE_STREAM=0 one stream only; IBA_SPIN_WAIT=0 blocking waits. */
int32_t iba_debug_last_path(const iba_handle* h);   /* 2: the batch was clustered into several tight groups with one pair search each */
/* The planner behind that decision on B <= IBA_MAX_BATCH candidates, host only (no GPU): a batch whose nominal projection spread
 * (at a point 12 m out, 10 m deep, focal length max_fx) exceeds max_px is clustered into at most max_groups (<= 4) groups, accepted
 * when every group is within max_px. group_of[B] (may be NULL) receives each candidate's group, group_px[4] (may be NULL) the
 * groups' nominal spreads, *n_groups 1 (one shared search), 2..4 (clustered) or 0 (wide everywhere: every candidate for itself). */
iba_status iba_debug_plan_groups(const double* x, int32_t B, double max_fx, double max_px, int32_t max_groups, int32_t* group_of, double* group_px, int32_t* n_groups);
/* debug: how often the anchored neighbour lists (the 1-NN search memoised around an anchor extrinsic that follows the
 * optimiser's candidates; IBA_NN_SETS=0 disables them, IBA_ANCHOR_REACH sets the drift in metres that moves the anchor) have
 * been built on this handle. Results do not depend on the anchor: every lane certifies its pick or searches the tree. */
int32_t iba_debug_anchor_builds(const iba_handle* h);
/* diagnostic: how many times the shared pair search has run on this handle (an evaluation whose batch stays inside the bound of
 * the lists an earlier call built reuses them: IBA_PAIR_MEMO, default on) */
int32_t iba_debug_pairs_builds(const iba_handle* h);
/* diagnostic: list entries of the last evaluation (all candidates) that the anchored neighbour lists could not settle and the
 * tree search took over; -1 when no search ran */
double iba_debug_nn_left_to_tree(iba_handle* h);
int32_t iba_debug_last_nn_threads(const iba_handle* h);   /* threads per block of the last search launch: 64 = one-wave blocks (small kd trees, 12 candidates or more), 256 otherwise */
int32_t iba_debug_last_nn_list(const iba_handle* h);   /* > 0: the last search launch was the opt-in persistent list kernel, IBA_NN_LIST=1, with that many workers per XCD and group */
/* diagnostic: mean number of (scan point, keypoint) pairs per keyframe that the last shared pair search listed; -1: none ran */
double iba_debug_mean_pairs(iba_handle* h);
/* debug (host only): R[9], t[3], dR/d omega_k [3][9], dt/dx_k [6][3], s of a candidate as the factor kernel reads them (58 doubles) */
iba_status iba_debug_cand(const double x[7], double out58[58]);
/* debug: association blocks since the last reset that rescanned every scan point (a speed-only fallback: full queue / pair list) */
int64_t iba_debug_rescans(iba_handle* h, int32_t reset);
/* debug: out4 = {those rescans, iba_assoc2_kernel blocks whose note list of possible winners overflowed (speed only), 0, 0} since the last reset */
iba_status iba_debug_counters(iba_handle* h, uint32_t out4[4], int32_t reset);
/* debug: {pair lists of the last call that had overflowed (their blocks rescan every point: speed only), lists read, longest list} */
iba_status iba_debug_pair_lists(iba_handle* h, int32_t out3[3]);
/* debug: exact 1-NN (nanoflann semantics with the lowest-index tie rule, iba_global.cpp:116-122) of n LiDAR-frame query
 * points in the scan of local frame `frame`, run through the search kernel's own kd search, one lane per query: original point
 * index and exact squared distance. mode 1: as the association path's query alone; 2: as the cost path's alone; 3 / 4: both paths
 * searched together as the kernel does, the query as the first (3) or the second (4) with its partner 1e-7 beside it.
 * For parity tests of the search itself. */
iba_status iba_debug_nn(iba_handle* h, int32_t frame, const double* q_xyz, int32_t n, int32_t mode, uint32_t* out_idx, double* out_d2);
/* debug: the memoised local plane (plane_cache = 1) at scan point `point` (ORIGINAL index) of owned frame `frame`: which = 0 the cost
 * path's planes (norm_radius / norm_max_pts), 1 the association / Jacobian path's (neigh_radius / neigh_max_pts). With plane_cache = 0
 * only which = 1 after iba_build_problem: the plane the frozen problem's residual blocks read (fitted where a block needed one; other
 * points hold stale records). out5 = unit normal
 * (3), sum of |(p_i - c) . n|, squared distance of the farthest kept neighbour; *k = kept neighbours. The parity tests substitute this
 * normal into the oracle's residual block: the device fits planes with its own libm, and an ill-conditioned block amplifies the
 * last-bit difference of the two normals (tests/parity_explain.py). */
iba_status iba_debug_plane(iba_handle* h, int32_t frame, uint32_t point, int32_t which, double out5[5], int32_t* k);

/* the same driver on built-in analytic black boxes (host only; for tests of the search logic without a GPU):
 * 0 smooth bowl, 1 bowl with an active constraint and an infeasible start, 2 nonsmooth with two constraints,
 * 3 shallow bowl covered with narrow local basins (for the variable-neighbourhood restarts) */
iba_status iba_mads_selftest(int32_t problem, const double* x0, const iba_mads_options* opt, iba_mads_result* res);
/* the same two calls with the sequence of black-box evaluations recorded: `trace` receives up to `cap` rows of 8 doubles
 * (x[7], f) in evaluation order, *n_trace the number of evaluations (tests diff the sequence against oracle/mads.py) */
iba_status iba_calibrate_mads_trace(iba_handle* h, const double* x0, const iba_mads_options* opt, iba_mads_result* res, double* trace, int32_t cap, int32_t* n_trace);
/* ... and with the black-box CALLS recorded as well: batch_sizes receives up to cap_batches call sizes in order (they split the
 * rows of `trace` into the batches iba_eval_bbo was given), *n_batches their number. bench.py replays such a record through
 * iba_eval_bbo (extras.mads_trace_replay). */
iba_status iba_calibrate_mads_record(iba_handle* h, const double* x0, const iba_mads_options* opt, iba_mads_result* res, double* trace, int32_t cap, int32_t* n_trace,
                                     int32_t* batch_sizes, int32_t cap_batches, int32_t* n_batches);
iba_status iba_mads_selftest_trace(int32_t problem, const double* x0, const iba_mads_options* opt, iba_mads_result* res, double* trace, int32_t cap, int32_t* n_trace);


/* debug: the sorted neighbour lists of the plane fits — the list builder of the plane kernels, i.e. ComputeAlignmentDist's
 * kNN(norm_max_pts) around nn_pt (iba_global.cpp:125-133; ComputeLocalNeighbor, pointcloud.h:733-760) — around n scan points (ORIGINAL
 * indices) of owned frame `frame`: up to k <= 64 neighbours with d^2 < r2 each (INFINITY: no clip), nearest first, the point itself
 * first at distance 0. out_idx / out_d2 are n x k (0xFFFFFFFF / -1 beyond out_cnt[i]). Equal distances keep the visiting order of THIS
 * library's tree (nanoflann keeps its own tree's): only exact ties can differ from the reference's lists. */
iba_status iba_debug_knn(iba_handle* h, int32_t frame, const uint32_t* points, int32_t n, int32_t k, double r2, uint32_t* out_idx, double* out_d2, int32_t* out_cnt);

/* debug: what a blocking entry point costs a C caller (no language binding in the clock): `iters` back-to-back calls — after three that do not
 * count — of iba_eval_cost (kind 0), iba_eval_full (kind 1) or iba_eval_factors (kind 2; needs iba_build_problem) on the same B candidates, each
 * timed here with the steady clock. *median_ms, *min_ms (may be NULL) per call. The unmodified caller of the reference is such a loop
 * (BALoss::eval_x, iba_global.cpp:385: one candidate per call). */
iba_status iba_debug_call_latency(iba_handle* h, const double* x, int32_t B, int32_t kind, int32_t iters, double* median_ms, double* min_ms);

/* debug: the kernels' shared-reciprocal division (two quotients by one depth: csrc/iba_kernels.hpp, div2) beside the compiler's IEEE f64
 * division, on n operand triples from the caller, on `device`. q0/q1 = num0/den, num1/den as the projections compute them, ref0/ref1 = as
 * the plain division does; *n_fast = triples that took the shared-reciprocal path (the others fall back to the plain division inside). The
 * two must agree bit for bit on every operand (tests/test_gpu_division.py; iba_global.cpp:70-75, :308-313 are the divisions they stand for). */
iba_status iba_debug_div2_selftest(int32_t device, const double* num0, const double* num1, const double* den, int64_t n, double* q0, double* q1, double* ref0, double* ref1, int64_t* n_fast);

#ifdef __cplusplus
}
#endif
#endif /* IBA_MI355X_DEBUG_H */
