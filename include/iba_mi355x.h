/*
 * iba_mi355x.h — C ABI of the MI355X-native IBA cross-modality evaluation path.
 *
 * This is the drop-in boundary for ONE hot path of
 * gitouni/Spatial-Temporal-LiDAR-camera-Calibration: the per-candidate-extrinsic
 * evaluation that the reference runs on the CPU in
 *   BAError()            src/examples/iba_global.cpp:169-344  (= iba_func.cpp:179-354)
 *   BALoss::eval_x()     src/examples/iba_global.cpp:377-396
 *   BuildProblem()       src/examples/iba_local.cpp:145-323   (association for the Jacobian path)
 *   IBA_PlaneFactor / Point2Point_Factor / Point2Plane_Factor
 *                        include/IBACalib2.hpp:152-184, 570-584, 611-625 (g2o twin: IBACalib.hpp:103-140)
 *
 * Only PODs cross the boundary: no Eigen / OpenCV / ORB_SLAM2 / torch types.
 * All pointers in iba_problem_desc are HOST pointers borrowed for the duration
 * of iba_create() only. Errors are return codes (the reference throws or
 * returns DBL_MAX sentinels; the sentinels are kept, see iba_cost_out).
 * A handle is thread-compatible: one evaluation at a time per handle
 * (BALoss::eval_x is called from NOMAD's single worker thread, iba_global.cpp:385).
 */
#ifndef IBA_MI355X_H
#define IBA_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IBA_ABI_VERSION 2 /* 2: iba_params.factor_3d2d_kind */
#define IBA_MAX_BATCH 64 /* the batch unit of the callers in this library (one MADS poll block, the planner's diagnostics); NOT a limit of the evaluators */
#define IBA_MAX_CHAIN 512 /* most candidates ONE launch chain takes (iba_create_options.max_chain_batch <= this); a call with more runs as consecutive chains */

typedef enum iba_status {
    IBA_OK = 0,
    IBA_ERR_INVALID_ARG = 1,
    IBA_ERR_NO_DEVICE = 2,   /* no gfx950 device / HIP runtime unusable: there is NO CPU fallback */
    IBA_ERR_HIP = 3,
    IBA_ERR_UNSUPPORTED = 4, /* problem shape outside what the kernels handle */
    IBA_ERR_STATE = 5,       /* e.g. iba_eval_factors before iba_build_problem */
    IBA_ERR_IO = 6           /* a dataset file is missing, truncated or not in the reference's format */
} iba_status;

typedef struct iba_handle iba_handle;

/*
 * Flat description of what the reference holds in
 *   std::vector<VecVector3d> PointClouds, std::vector<Eigen::Isometry3d> vTwl,
 *   std::vector<ORB_SLAM2::KeyFrame*> KeyFrames          (iba_global.cpp:169-173)
 * Frames are the keyframes sorted by mnId (iba_global.cpp:507). CSR offsets are
 * element counts (not bytes).
 */
typedef struct iba_problem_desc {
    int32_t n_frames; /* F */

    /* PointClouds[Fi] — raw scan in the LiDAR frame, float32 as in KITTI .bin (io_tools.h:170-187). */
    const uint64_t* pt_offset; /* [F+1] */
    const float* pts_xyz;      /* [N*3] AoS x,y,z */

    /* pKF->fx, fy, cx, cy, mnMaxX (W), mnMaxY (H)  (iba_global.cpp:64-65) */
    const double* intrinsics; /* [F*6] */

    /* pKF->mvKeysUn[i].pt (cv::Point2f) */
    const uint64_t* kp_offset; /* [F+1] */
    const float* kp_uv;        /* [K*2] */

    /* mapKpt2Mpt (inverse of pKF->mmapMpt2Kpt, iba_global.cpp:210-213):
     * MapPoint::GetWorldPos() (CV_32F 3x1) of the MapPoint owned by keypoint i. */
    const uint8_t* kp_has_mappoint; /* [K] 0/1 */
    const float* kp_mappoint_w;     /* [K*3] world position, ignored where kp_has_mappoint==0 */

    /* pKF->GetPoseSafe() (CV_32F 4x4, top 3 rows, row-major) */
    const float* Tcw; /* [F*12] */

    /* pKF->GetBestCovisibilityKeyFramesSafe(num_best_covis) (or ByWeight), iba_global.cpp:254-258.
     * One "slot" per (reference KF, covisible KF) pair; at most 62 per reference KF (30 match bits in the keypoint's flag word, a
     * second word for the slots beyond them; more: IBA_ERR_UNSUPPORTED). */
    const uint64_t* covis_offset; /* [F+1] slots of frame f are covis_offset[f]..covis_offset[f+1] */
    const int32_t* covis_frame;   /* [S] frame index of the covisible KF */
    /* pKFConv->GetPose() * InvRefCVPose evaluated in CV_32F (iba_global.cpp:280), top 3 rows
     * row-major, translation NOT multiplied by scale. */
    const float* covis_relpose; /* [S*12] */
    /* pKF->GetUordMatchedKptIds(pKFConv): keypoint id in the reference KF -> keypoint id in
     * the covisible KF (KeyFrame.cc:528-538 semantics). */
    const uint64_t* match_offset;  /* [S+1] */
    const int32_t* match_kp_ref;   /* [M] */
    const int32_t* match_kp_covis; /* [M] */

    /* Hand-eye term inputs (iba_global.cpp:264-276), one per frame, ignored for the last frame:
     *   Tc_next[f] = KeyFrames[f+1]->GetPose() * InvRefCVPose   in CV_32F, unscaled (:267)
     *   Tl_next[f] = vTwl[f+1].inverse() * vTwl[f]              double            (:269) */
    const float* Tc_next;  /* [F*12] */
    const double* Tl_next; /* [F*12] */
} iba_problem_desc;

/*
 * Thresholds. Field names follow IBAGlobalParams (iba_global.cpp:26-52) for the cost path
 * and IBALocalParams (IBACalib2.hpp:108-137) for the Jacobian path ("local_" prefix where the
 * two structs share a name but are configured independently).
 */
typedef struct iba_params {
    /* shared association (FindProjectCorrespondences) */
    double max_pixel_dist; /* 1.5 */

    /* cost path: BAError */
    int32_t num_min_corr_cost;   /* 30, hard-coded at iba_global.cpp:203 */
    double corr_3d_2d_threshold; /* 40 */
    double corr_3d_3d_threshold; /* 5 (yml: 10) */
    int32_t norm_max_pts;        /* 30 (<= 64 supported: the neighbour list lives one entry per lane of a wave) */
    int32_t norm_min_pts;        /* 5 */
    double norm_radius;          /* 0.6 */
    double norm_reg_threshold;   /* 0.04 (yml: 0.02) */
    double min_diff_dist;        /* 0.01 (yml: 0.2) */
    double err_weight[2];        /* {1,1} */
    int32_t use_plane;           /* 1 */

    /* Jacobian path: BuildProblem + Ceres losses */
    int32_t num_min_corr;             /* 30 */
    double max_3d_dist;               /* 1.0 */
    double neigh_radius;              /* 0.6 */
    int32_t neigh_max_pts;            /* 30 (<= 64 supported) */
    int32_t neigh_min_pts;            /* 5 */
    double local_min_diff_dist;       /* 0.2 */
    double local_norm_reg_threshold;  /* 0.001 */
    double robust_kernel_delta;       /* 2.98 */
    double robust_kernel_3ddelta;     /* 1.0 */

    /* engine knob (no reference counterpart): 1 = memoise the x-independent local-plane fit per
     * scan point (bit-identical results); 0 = refit inside every evaluation as the reference does. */
    int32_t plane_cache;

    /* Which 3d-2d residual the Jacobian path builds (ABI version 2):
     *   0  IBA_PlaneFactor (IBACalib2.hpp:152-184; g2o twin IBAPlaneEdge, IBACalib.hpp:103-140): the keypoint's ray intersected with the
     *      local plane at its scan point, reprojected into every covisible keyframe — one block of 2 NConv rows per keypoint [default];
     *   1  IBATestEdge (IBACalib.hpp:14-71, functor :40-58): the DIRECT point-to-pixel term — the matched scan point itself,
     *      p1 = R_i (R_cl p0 + t_cl) + s t_i, one 2-row block per (correspondence, matched covisible keyframe), Huber(robust_kernel_delta)
     *      each. The reference declares the edge and instantiates it nowhere; the edge set here is that of BAError's 3d-2d loop
     *      (iba_global.cpp:291-328: every correspondence of a used frame x every covisible keyframe that matches its keypoint; no plane, no
     *      MapPoint, no neighbourhood test), whose cost these are the normal equations of. The 3d-3d blocks are built as in mode 0 when
     *      err_weight[1] > 1e-10 and not at all otherwise (BAError's switch, :214-220): err_weight = {1, 0} is BASELINE's
     *      "point-to-pixel only" configuration on the Jacobian path. Mode 1 is an EXTENSION without a reference-run counterpart: the
     *      reference never constructs the edge, so there is no reference output to validate against (it is checked against the oracle's
     *      restatement of the functor and an independent autograd evaluation only), and its cost follows this library's Ceres convention
     *      (0.5 rho) although the edge's reference twin is a g2o edge. */
    int32_t factor_3d2d_kind;
} iba_params;

/* GeoCalib.h:18-33 computeCorrespondence, the cloud-to-cloud 1-NN correspondence north_star's "GeoCalib call surface" names: for every
 * source point the nearest target point (nanoflann KDTreeSingleIndexAdaptor, L2_Simple, max_leaf 15: the leaf size does not change a result),
 * kept when  sq_dist <= max_distance  — the SQUARED distance against the un-squared parameter, exactly as the reference writes it (:29);
 * pairs (source index, target index) in ascending source order (:30).
 * The target cloud is the scan of local keyframe `frame` of the handle (the float32 points as loaded, io_tools.h:170-187; indices are the scan's
 * ORIGINAL order), searched by the kd search of the evaluation path itself (exact f64 distances; a tie between two target points goes to the
 * lower index, nanoflann keeps the first visited: only duplicate points can tie). src_xyz: n_src points (x, y, z doubles) in that scan's frame —
 * the caller applies its transform, as computeInitialGeoError does (GeoCalib.h:76-105). out_src / out_tgt: room for n_src entries each.
 * The header is an orphan of the reference (nothing includes it, and std::vector<Eigen::Vector3d> is no nanoflann dataset: it does not compile
 * there); this entry point exists so that the one name north_star lists is not missing, and is pinned against the reference's own nanoflann. */
iba_status iba_geo_correspondences(iba_handle* h, int32_t frame, const double* src_xyz, int32_t n_src, double max_distance,
                                   uint32_t* out_src, uint32_t* out_tgt, int32_t* n_out);

/* The ABI version the LIBRARY was built with (IBA_ABI_VERSION of its header). iba_params carries no struct_size: a caller compiled against
 * an older header would pass a shorter struct. Callers compare iba_abi_version() with their own IBA_ABI_VERSION before iba_create(). */
int32_t iba_abi_version(void);

/* Output of one BAError() call. The first five fields are the reference's returned tuple
 * (iba_global.cpp:343); the rest are the counters it prints with verborse (:341-342). */
typedef struct iba_cost_out {
    double f1;               /* mean 3d-2d distance over valid edges, DBL_MAX sentinel (:330-333) */
    double f2;               /* mean 3d-3d distance over valid edges, DBL_MAX sentinel (:334-337) */
    double C;                /* mean hand-eye constraint value, NaN if no frame was processed (:338) */
    int32_t valid_cnt_3d_2d;
    int32_t cnt_3d_2d;
    int32_t cnt_3d_3d;
    int32_t valid_cnt_3d_3d;
    int32_t valid_pl_3d_3d;
    int32_t valid_pt_3d_3d;
    int32_t frames_used;     /* frames that passed the >= num_min_corr_cost test */
    int32_t n_corr;          /* sum of corrset.size() over used frames */
} iba_cost_out;

/* Gauss-Newton normal equations of the iba_local problem at x:
 *   H = sum_blocks w J^T J,  b = sum_blocks w J^T r,  cost = 1/2 sum_blocks rho(|r|^2)
 * with Huber IRLS weights per residual block as Ceres applies them (corrector with rho''<=0). */
typedef struct iba_normal_out {
    double H[49]; /* row-major symmetric 7x7 */
    double b[7];
    double cost;          /* Ceres convention: 1/2 sum rho(s) */
    double chi2;          /* sum |r|^2 (un-robustified) */
    int32_t n_factor_3d2d; /* IBA_PlaneFactor blocks (factor_3d2d_kind = 1: IBATestEdge blocks) */
    int32_t n_factor_p2pl; /* Point2Plane_Factor blocks */
    int32_t n_factor_p2pt; /* Point2Point_Factor blocks */
    int32_t n_residuals;   /* total scalar residuals */
    int32_t frames_used;
    int32_t n_corr;
} iba_normal_out;

/* NOMAD black-box outputs of BALoss::eval_x (iba_global.cpp:386-392). */
typedef struct iba_bbo {
    double f, c1, c2, c3;
} iba_bbo;

iba_status iba_default_params(iba_params* p); /* IBAGlobalParams / IBALocalParams defaults */

/* Uploads the frames [frame_begin, frame_end) of the problem to HIP device `device`, builds the
 * static per-scan 3-D indices (reference: KDTree3D per scan, iba_global.cpp:361-367) and the
 * per-frame keypoint grids. The full descriptor must be given on every rank (covisible keypoints
 * are resolved at creation time); only owned frames are uploaded and evaluated. */
iba_status iba_create(const iba_problem_desc* desc, const iba_params* params, int device,
                      int32_t frame_begin, int32_t frame_end, iba_handle** out);
/*
 * Engine options of a handle (no reference counterpart: none of them changes a result bit, they steer how the work is shared
 * between candidates and calls). Fill with iba_default_create_options, change fields, pass to iba_create_ex; iba_create uses the
 * defaults. The IBA_* environment variables of earlier rounds are still read at creation as DEBUG overrides of these fields
 * (process-global, for A/B runs without recompiling a caller); an integrator sets the struct.
 */
typedef struct iba_create_options {
    int32_t struct_size;          /* sizeof(iba_create_options) of the caller: the struct may grow at its end */
    int32_t common_pairs;         /* 2d-3d pair search shared by a batch: 0 never, 1 when the batch (or each of its groups) is tight [default], 2 always */
    double common_max_px;         /* nominal projection spread (px) up to which candidates share one pair search [20] */
    int32_t max_pair_groups;      /* a wider batch is clustered into up to this many tight groups, 1..4 [4]; 1 = no clustering */
    int32_t pair_memo;            /* pair lists built for an inflated bound and reused by later calls that stay inside it [1] */
    int32_t pair_memo_max_batch;  /* largest batch (group) whose lists are built reusable [40] */
    double pair_inflation;        /* inflation of a reusable list's bound [1.25] */
    int32_t anchored_lists;       /* 3-D 1-NN memoised around an anchor extrinsic that follows the candidates [1] */
    double anchor_reach;          /* drift (m) of a MapPoint query 30 m out that moves the anchor [0.06] */
    int32_t side_stream;          /* (round 3-4: staging launch on a second stream of the handle) no effect since round 5: the chain has no staging launch [1] */
    int32_t spin_wait;            /* the host polls the stream at the end of a call instead of blocking [1] */
    int32_t factor_mfma;          /* normal-equation sums on the matrix cores (v_mfma_f64_16x16x4; measured slower) [0] */
    int32_t pair_list_capacity;   /* entries per keyframe of a pair list; 0 = automatic. A full list only costs speed [0] */
    int32_t max_chain_batch;      /* candidates one launch chain takes, 1..IBA_MAX_CHAIN [512]: a shard of few keyframes (one rank of an 8-GPU job) fills the
                                     device only with many candidates per chain. Work lists are allocated for the largest batch a call has passed so far
                                     (16 B x keypoints x keyframes per candidate); with plane_cache = 0 a chain takes at most IBA_MAX_BATCH */
    int32_t chain_fold;           /* 1: the candidate block reaches the device through spare blocks of the first kernel of the chain and the hand-eye terms
                                     are evaluated inside the summing kernel (no staging launch, no second stream, no event between kernels) [1];
                                     0: a staging launch of its own at the head of every chain */
} iba_create_options;
iba_status iba_default_create_options(iba_create_options* o);
iba_status iba_create_ex(const iba_problem_desc* desc, const iba_params* params, int device,
                         int32_t frame_begin, int32_t frame_end, const iba_create_options* options, iba_handle** out);
void iba_destroy(iba_handle* h);
iba_status iba_set_params(iba_handle* h, const iba_params* params);
const char* iba_last_error(const iba_handle* h); /* never NULL; h may be NULL for creation errors */

/* BAError() for B candidate x = [omega(3), upsilon(3), s] (row-major B x 7). */
iba_status iba_eval_cost(iba_handle* h, const double* x, int32_t B, iba_cost_out* out);
/* BALoss::eval_x packing on top of iba_eval_cost. */
iba_status iba_eval_bbo(iba_handle* h, const double* x, int32_t B, double he_threshold, double valid_rate,
                        iba_bbo* out);

/* BuildProblem() at x_assoc followed by one evaluation of every residual block at the same x:
 * association + residuals + Jacobians + normal equations, B candidates per call. */
iba_status iba_eval_normal(iba_handle* h, const double* x, int32_t B, iba_normal_out* out);

/* BAError tuple AND the re-associated normal equations of the same B candidates from one pass over the scans
 * (the two paths share projection + 2d-3d association). Counters are identical to those of iba_eval_cost and
 * iba_eval_normal called separately, sums agree to summation order (1e-15). A candidate's results do not depend on what
 * else is in the batch. */
iba_status iba_eval_full(iba_handle* h, const double* x, int32_t B, iba_cost_out* cost, iba_normal_out* normal);

/* The two halves separately, as Ceres uses them (iba_local.cpp:443-445): freeze the association
 * at x_assoc, then evaluate the frozen residual blocks at B other x. */
iba_status iba_build_problem(iba_handle* h, const double* x_assoc);
iba_status iba_eval_factors(iba_handle* h, const double* x, int32_t B, iba_normal_out* out);
/* The frozen problem as ONE residual block for a solver that wants residuals and Jacobians (Ceres: the blocks
 * BuildProblem() adds, iba_local.cpp:263-308; g2o: a unary edge on VertexSim3, IBACalib.hpp:74-155): 8 rows with
 *   J^T J = H,  J^T r = b,  |r|^2 = 2 cost      ([J | r] = upper Cholesky factor of [[H, b], [b^T, 2 cost]])
 * so that the solver's Gauss-Newton model AND its step-acceptance cost are those of the whole problem, robust weights
 * included. r has 8 entries, J is 8 x 7 row-major. iba_whiten_normal is the host-only half (any iba_normal_out). */
iba_status iba_eval_whitened(iba_handle* h, const double* x, double r[8], double J[56]);
iba_status iba_whiten_normal(const iba_normal_out* normal, double r[8], double J[56]);
/* Caller of the Jacobian path: the outer re-association loop of iba_local
 * (iba_local.cpp:434-460) around a Ceres-style LM on the device-reduced 7x7 normal equations. */
typedef struct iba_lm_options {
    int32_t max_outer_iterations; /* max_iba_iter */
    int32_t max_inner_iterations; /* 30, iba_local.cpp:437 */
    double min_diff;              /* iba_min_diff for allClose (iba_local.cpp:454) */
    double function_tolerance, gradient_tolerance, parameter_tolerance; /* Ceres defaults 1e-6, 1e-10, 1e-8 */
    double initial_trust_region_radius;                                 /* 1e4 */
} iba_lm_options;
typedef struct iba_lm_result {
    double x[7];
    int32_t outer_iterations, inner_iterations, evaluations, converged;
    double initial_cost, final_cost;
} iba_lm_result;
iba_status iba_default_lm_options(iba_lm_options* o);
iba_status iba_calibrate_lm(iba_handle* h, const double* x0, const iba_lm_options* opt, iba_lm_result* res);

/* Per-residual values and Jacobians of the frozen problem (for Ceres / g2o adaptors and tests).
 * Call with r == NULL to query *n_rows. J is n_rows x 7 row-major, block_id[n_rows] identifies the
 * residual block, block_kind: 0 = IBA_PlaneFactor, 1 = Point2Plane, 2 = Point2Point, 3 = IBATestEdge (factor_3d2d_kind = 1). */
iba_status iba_eval_residuals(iba_handle* h, const double* x, double* r, double* J, int32_t* block_id,
                              int32_t* block_kind, int64_t* n_rows);

/* 2d-3d correspondences (corrset of FindProjectCorrespondences, iba_global.cpp:55-96) of one owned
 * frame at x: pairs (keypoint id, scan point id) sorted by keypoint id. cap = capacity in pairs. */
iba_status iba_get_correspondences(iba_handle* h, const double* x, int32_t frame, uint32_t* kp_idx,
                                   uint32_t* pt_idx, int32_t cap, int32_t* n_out);

/*
 * Multi-GPU building blocks (frames shard across ranks; one sum all-reduce per evaluation).
 * iba_eval_*_partial writes this rank's partial sums for B candidates into DEVICE memory
 * `d_partials` (B * iba_partial_stride() doubles, counters carried as doubles) on HIP stream
 * `stream` (a hipStream_t passed as void*; NULL = the handle's own NON-BLOCKING stream, NOT the legacy default stream: a caller whose
 * other work — the all-reduce, copies — sits on the default stream must pass a stream of its own) without synchronising.
 * After the caller has summed the partial blocks over ranks (ncclAllReduce, sum, f64),
 * iba_finalize_* turns HOST copies of the summed blocks into the outputs above.
 */
int32_t iba_partial_stride(void);
iba_status iba_eval_cost_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream);
iba_status iba_eval_normal_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream);
/* cost and normal sums share one block (disjoint slots): finalize the summed block with BOTH iba_finalize_* */
iba_status iba_eval_full_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream);
iba_status iba_finalize_cost(const iba_params* params, const double* partials, int32_t B, iba_cost_out* out);
iba_status iba_finalize_normal(const iba_params* params, const double* partials, int32_t B, iba_normal_out* out);

/* The frozen problem's residual blocks, as a partial block (the Jacobian-path half of the LM caller on several GPUs). */
iba_status iba_eval_factors_partial(iba_handle* h, const double* x, int32_t B, void* d_partials, void* stream);
/* Calls on one handle must be issued in order, on one stream at a time: the handle's work buffers (candidate ring, lists,
 * records) are reused from call to call and are ordered by that stream only.
 * A cost evaluation also uses a second stream that belongs to the handle: its staging launch (candidates -> device, hand-eye terms)
 * and the later copy of the candidates' derivatives run there, beside the pair search / the search kernel on `stream`. That stream
 * is ordered behind everything `stream` held when the call was made and `stream` waits for it before the first kernel that reads
 * its results, so a caller sees one stream's ordering (IBA_SIDE_STREAM=0 in the environment: everything on `stream`). */

/* Diagnostics, timing probes and self-tests (iba_debug_*, iba_last_*_ms, iba_set_timing, iba_*_selftest*) are declared in
 * iba_mi355x_debug.h: test and benchmark tooling, not part of the drop-in surface. */
int64_t iba_num_points(const iba_handle* h);
int64_t iba_num_keypoints(const iba_handle* h);

/*
 * Batch-aware mesh adaptive direct search for the global stage [SURVEY.md 8(f) row 2]: the caller the reference gets
 * from NOMAD 4 (iba_global.cpp:551-602: 7 variables, bounds x0 + lb / x0 + ub, OBJ + 3 progressive-barrier
 * constraints from BALoss::eval_x, OrthoMADS 2N, INITIAL_POLL_SIZE, MIN_MESH_SIZE, MAX_BB_EVAL). One iteration =
 * one iba_eval_bbo batch (full polls around the feasible and the infeasible incumbent). csrc/iba_mads.hpp.
 */
typedef struct iba_mads_options {
    int32_t max_bb_eval;     /* max_bbeval, 5000 */
    double lb[7], ub[7];     /* ABSOLUTE bounds (the reference adds its yml lb/ub to x0, iba_global.cpp:530-533) */
    double init_frame[7];    /* init_frame, 0.5 each */
    double min_mesh;         /* min_mesh, 1e-6 */
    double he_threshold;     /* constraint |C| <= he_threshold (iba_global.cpp:387) */
    double valid_rate;       /* constraint valid/(cnt+1) >= valid_rate (:388) */
    int32_t seed;
    int32_t bases_per_poll;  /* orthogonal 2n-direction sets per poll centre and iteration (1 = OrthoMADS 2N) */
    int32_t speculative;     /* 1: one extra point along the last successful direction */
    int32_t vns_max_idle;    /* variable-neighbourhood restarts (use_vns): stop after this many in a row without gain; 0 = none */
} iba_mads_options;
typedef struct iba_mads_result {
    double x[7];
    double f, c1, c2, c3;
    int32_t feasible;        /* 1: x satisfies the three constraints (findBestFeas, iba_global.cpp:593-599) */
    int32_t evaluations, iterations, batches, cache_hits, restarts;
    int32_t stop_reason;     /* 1 converged (min mesh, restarts exhausted), 2 evaluation budget */
} iba_mads_result;
/* defaults of config/calib/00/iba_calib_global.yml:21-47 around x0 (lb/ub = x0 -/+ (0.1,0.1,0.1,0.3,0.3,0.3,1.0)) */
iba_status iba_default_mads_options(const double* x0, iba_mads_options* o);
iba_status iba_calibrate_mads(iba_handle* h, const double* x0, const iba_mads_options* opt, iba_mads_result* res);
/*
 * Multi-GPU inside one process: the keyframes sharded over n devices of a node (contiguous ranges balanced by points), one
 * handle, one issuing thread and one RCCL communicator per device (ncclCommInitAll). The reference's one parallel strategy is the
 * frame loop with critical-section sums (iba_global.cpp:193, 239, 318; iba_func.cpp:203; iba_local.cpp:162); here every device
 * evaluates its frames, ONE ncclAllReduce(sum, f64) of the B x iba_partial_stride() block over xGMI adds them on the devices,
 * and device 0's copy is finalised on the host. Same outputs and callers as the single-device entry points. The candidate
 * block (Sim3Exp and its derivatives) is computed once per call, the devices are issued concurrently, and the caller's current
 * HIP device is left alone. librccl is loaded (dlopen) when the first communicator is needed: single-device and host-only
 * users of this library do not need it installed.
 */
typedef struct iba_group iba_group;
iba_status iba_group_create(const iba_problem_desc* desc, const iba_params* params, const int32_t* devices, int32_t n_devices, iba_group** out);
/* flags: IBA_GROUP_REDUCE_HOST = the partial blocks are copied to the host and summed there in rank order (bitwise
 * reproducible, no RCCL needed, and the same device may appear more than once in `devices`) instead of one ncclAllReduce */
#define IBA_GROUP_REDUCE_HOST 1
iba_status iba_group_create_ex(const iba_problem_desc* desc, const iba_params* params, const int32_t* devices, int32_t n_devices, int32_t flags, iba_group** out);
void iba_group_destroy(iba_group* g);
const char* iba_group_last_error(const iba_group* g); /* g may be NULL for creation errors */
int32_t iba_group_size(const iba_group* g);
int32_t iba_group_comm_ranks(const iba_group* g);     /* ncclCommCount of the group's communicator; 0 with IBA_GROUP_REDUCE_HOST */
double iba_group_last_issue_us(const iba_group* g);   /* host wall time of the last chunk: candidate block, hand-over to the device threads, wait */
double iba_group_last_enqueue_us(const iba_group* g); /* of which: until the last device's launch chain + collective were enqueued (host issue time) */
iba_status iba_group_frame_range(const iba_group* g, int32_t rank, int32_t* frame_begin, int32_t* frame_end);
iba_status iba_group_set_params(iba_group* g, const iba_params* params);
iba_status iba_group_eval_cost(iba_group* g, const double* x, int32_t B, iba_cost_out* out);
iba_status iba_group_eval_bbo(iba_group* g, const double* x, int32_t B, double he_threshold, double valid_rate, iba_bbo* out);
iba_status iba_group_eval_normal(iba_group* g, const double* x, int32_t B, iba_normal_out* out);
iba_status iba_group_eval_full(iba_group* g, const double* x, int32_t B, iba_cost_out* cost, iba_normal_out* normal);
iba_status iba_group_build_problem(iba_group* g, const double* x_assoc);
iba_status iba_group_eval_factors(iba_group* g, const double* x, int32_t B, iba_normal_out* out);
iba_status iba_group_calibrate_lm(iba_group* g, const double* x0, const iba_lm_options* opt, iba_lm_result* res);
iba_status iba_group_calibrate_mads(iba_group* g, const double* x0, const iba_mads_options* opt, iba_mads_result* res);
/* One process per GPU with a communicator of the caller's (MPI / torchrun style): the one collective of the path on the
 * caller's ncclComm_t (passed as void*), in place on the device block written by iba_eval_*_partial. */
iba_status iba_comm_allreduce(void* nccl_comm, void* d_partials, int32_t B, void* stream);
/* For callers without RCCL headers of their own: communicators over devices of this process (comms[i] belongs to
 * devices[i]; ncclCommInitAll), their rank count, their release. */
iba_status iba_comm_init_all(void** comms, const int32_t* devices, int32_t n);
int32_t iba_comm_count(void* nccl_comm);
iba_status iba_comm_destroy(void* nccl_comm);
/* Which librccl this process runs — "path=<file> runtime=<code> header=<code> match=<0|1>" — and the two version codes
 * (ncclGetVersion of the loaded library, NCCL_VERSION_CODE of the headers this library was compiled against). Inside a
 * torch process the already-mapped torch/lib/librccl.so is the one that is used. */
iba_status iba_rccl_info(char* buf, int32_t cap, int32_t* runtime_version, int32_t* header_version);

/*
 * ---- On-disk formats of the reference pipeline -> problem descriptor [SURVEY.md 8(f) row 1] ----
 * Host-only (no GPU needed). Replaces, for the IBA path, what the reference does with OpenCV/ORB-SLAM2 objects in
 * main(): iba_global.cpp:398-505, iba_local.cpp:325-406, System::RestoreSystemFromFile (System.cc:612-694),
 * KeyFrameConstInfo (KeyFrame.cc:31-80), Map::RestoreMap (Map.cc:162-170), MapPoint(FileNode) (MapPoint.cc:435-451).
 */
typedef struct iba_dataset iba_dataset;
typedef struct iba_dataset_paths {
    const char* frame_id_file;    /* FrameId.yml: "mnId", "mnFrameId" (System.cc:597-609) */
    const char* lidar_pose_file;  /* LOFile: 12 numbers per pose, row-major 3x4 (kitti_tools.h:66-87) */
    const char* pointcloud_dir;   /* KITTI velodyne .bin files; file k (sorted by name, kitti_tools.h:48-62) = frame k */
    const char* keyframe_dir;     /* KeyFrames/NNNNNN.yml written by KeyFrame::saveData (KeyFrame.cc:209-252); the .bin
                                     twins hold only BoW vectors (KeyFrame.h:97-101) and are not read */
    const char* map_file;         /* Map.yml (Map.cc:213-231, MapPoint.cc:454-476) */
    int32_t pointcloud_skip;      /* readPointCloud `skip` (io_tools.h:142-196); iba_global passes 1 (iba_global.cpp:494) */
    int32_t only_positive_x;      /* readPointCloud `only_positive_x`; iba_local passes its config value (iba_local.cpp:394) */
    int32_t num_best_covis;       /* > 0: first N ordered covisible KFs (KeyFrame.cc:417-424); else by weight */
    int32_t min_covis_weight;     /* GetCovisiblesByWeightSafe (KeyFrame.cc:426-439) */
} iba_dataset_paths;

/* Loads and packs a dataset; the descriptor (and everything it points to) lives until iba_dataset_free. */
iba_status iba_dataset_load(const iba_dataset_paths* paths, iba_dataset** out);
const iba_problem_desc* iba_dataset_desc(const iba_dataset* d);
/* mnId / mnFrameId of keyframe f (FrameId.yml order = KeyFrame::lId order) */
iba_status iba_dataset_frame_ids(const iba_dataset* d, int32_t frame, int32_t* mn_id, int32_t* mn_frame_id);
void iba_dataset_free(iba_dataset* d);
/* message of the last failing iba_dataset_* / iba_read_* / iba_write_* call on this thread */
const char* iba_io_last_error(void);

/*
 * The reference's RUN CONFIGURATION (config/calib/NN/iba_calib_global.yml and its iba_func / iba_local siblings): what main()
 * reads with yaml-cpp — the maps io / orb / runtime (iba_global.cpp:412-471, iba_func.cpp:356-406, iba_local.cpp:325-378) — turned
 * into this header's structs, so that a run on the reference pipeline's artefacts takes the reference's own config file.
 * csrc/iba_config.cpp; host only. A missing key is an error (IBA_ERR_IO), as yaml-cpp's .as<T>() throws.
 */
typedef struct iba_run_config iba_run_config;
iba_status iba_run_config_load(const char* yaml_file, iba_run_config** out);
void iba_run_config_free(iba_run_config* c);
const char* iba_run_config_last_error(void);   /* of the last failing iba_run_config_* call on this thread */
/* iba_default_params + the file's runtime keys. local_stage = 0: IBAGlobalParams as iba_global / iba_func fill it
 * (iba_global.cpp:436-459); 1: IBALocalParams as iba_local fills it (iba_local.cpp:358-377). */
iba_status iba_run_config_params(const iba_run_config* c, int32_t local_stage, iba_params* out);
/* The dataset files as main() derives them: BaseDir (+ '/') + VOIdFile / LOFile, PointCloudDir, orb.KeyFrameDir, orb.MapFile,
 * num_best_covis, min_covis_weight. local_stage = 0 ignores PointCloudskip / PointCloudOnlyPositiveX exactly as iba_global does
 * (it reads them and then calls readPointCloud without them, iba_global.cpp:450-451 vs :494); 1 passes them (iba_local.cpp:394).
 * The strings belong to `c` and live until iba_run_config_free. */
iba_status iba_run_config_paths(iba_run_config* c, int32_t local_stage, iba_dataset_paths* out);
/* The NOMAD set-up of the global stage (iba_global.cpp: lb / ub added to x0 :530-533, init_frame, min_mesh, max_bbeval, he_threshold, valid_rate, seed,
 * use_vns) on top of iba_default_mads_options(x0). */
iba_status iba_run_config_mads(const iba_run_config* c, const double* x0, iba_mads_options* out);
/* any entry as text, "section.key" ("io.init_sim3", "runtime.direction_type"; sequences as "[a, b]"); NULL when absent */
const char* iba_run_config_get(const iba_run_config* c, const char* dotted_key);
/* BaseDir (+ '/') + io.<io_key> ("init_sim3", "gt_sim3", "ResFile", "VOFile"); NULL when absent; valid until the next call with the same key */
const char* iba_run_config_path(const iba_run_config* c, const char* io_key);

/* The numbers of one TOP-LEVEL entry of a cv::FileStorage YAML file ("%YAML:1.0": KeyFrames/NNNNNN.yml, Map.yml, ORB-SLAM2 settings
 * such as config/orb_ori/KITTI00-02.yaml, whose Camera.fx.. become KeyFrame::fx..): a scalar, a flow sequence or the data of an
 * !!opencv-matrix node, through the reader iba_dataset_load uses. *n_out = how many there are; at most cap are written. */
iba_status iba_read_cv_yaml_numbers(const char* file, const char* key, double* out, int32_t cap, int32_t* n_out);
/* readPointCloud for .bin (io_tools.h:142-196): XYZI float32 records; with skip > 1 the reference advances its counter
 * by `skip` but reads CONSECUTIVE records, i.e. it keeps the first floor((n - skip) / skip) + 1 points — reproduced.
 * *xyz is malloc'ed (release with iba_io_free). */
iba_status iba_read_kitti_bin(const char* file, int32_t skip, int32_t only_positive_x, float** xyz, int64_t* n_points);
/* ReadPoseList (kitti_tools.h:66-87): complete 12-number records only (the reference additionally appends one junk
 * pose when the file ends with a newline; nothing on the IBA path indexes it). *poses12 is malloc'ed. */
iba_status iba_read_pose_list(const char* file, double** poses12, int64_t* n_poses);
/* readSim3 / writeSim3 (kitti_tools.h:96-158): 12 row-major 3x4 numbers + scale, max_digits10 precision */
iba_status iba_read_sim3(const char* file, double rigid12[12], double* scale);
iba_status iba_write_sim3(const char* file, const double rigid12[12], double scale);
void iba_io_free(void* p);
/* (R, t, s) <-> the 7-vector the evaluators take: x[0:6] = g2o::SE3Quat(R, t).log() (rotation first), x[6] = s raw
 * (iba_global.cpp:511-515, iba_local.cpp:414); inverse = Sim3Exp (g2o_tools.h:105-140). */
iba_status iba_sim3_to_x(const double rigid12[12], double scale, double x[7]);
iba_status iba_x_to_sim3(const double x[7], double rigid12[12], double* scale);

/*
 * ---- Hand-eye initialiser [SURVEY.md 8(f) row 3]: the init_sim3 of the IBA stages without g2o (host only) ----
 * Ta = camera motions (scale-free), Tb = LiDAR motions, each n x 12 (row-major 3x4); result: T_AB (B -> A) and the
 * monocular scale. csrc/iba_handeye.cpp.
 */
iba_status iba_pose_to_motion(const double* poses12, int64_t n, double* motions12 /* (n-1) x 12 */); /* kitti_tools.h:160-165 */
iba_status iba_handeye(const double* Ta12, const double* Tb12, int64_t n, double rigid12[12], double* scale); /* HECalib.h:12-57 */
/* DGHECalib (HECalib.h:66-120), the reference's initialiser for degenerate motion: the rotation as iba_handeye's, translation zero (:109), scale =
 * sum |ta| |tb| / sum |ta|^2 over the pairs whose camera rotation angle is below dg_threshold (reference default 0.01 rad, :66, :82, :112-119;
 * NaN when there is none, as in the reference); *n_degenerate (may be NULL): how many pairs that were. iba_handeye returns IBA_ERR_UNSUPPORTED
 * exactly when its 4 x 4 system is singular — the case this one is for. */
iba_status iba_handeye_degenerate(const double* Ta12, const double* Tb12, int64_t n, double dg_threshold, double rigid12[12], double* scale, int64_t* n_degenerate);
/* The cost HECalibRobustKernelg2o minimises (NLHECalib.hpp:121-163): EdgeHE residual (:27-48), Huber(delta) per pair,
 * optional regulariser on upsilon with information n * ratio; Levenberg-Marquardt with a numerical Jacobian instead of
 * g2o's Dogleg on the reference's hand-written one (see csrc/iba_handeye.cpp). he_calib.cpp: 10 iterations. */
iba_status iba_handeye_robust(const double* Ta12, const double* Tb12, int64_t n, const double rigid12_init[12], double scale_init,
                              double robust_kernel_size, int32_t regulation, double regulation_ratio, int32_t iterations,
                              double rigid12[12], double* scale);
/* HECalibLineProcessg2o (NLHECalib.hpp:189-277): the same residual without a robust kernel but with a per-pair scalar
 * information w^2, w = mu / (mu + chi2), re-estimated between LM solves while mu anneals from mu0 by divid_factor until
 * below min_mu or ex_max_iter outer rounds; the regulariser follows the sum of the weights. The reference ignores its
 * in_max_iter argument and runs 10 inner iterations per solve: pass inner_iterations = 10 for its behaviour. */
iba_status iba_handeye_lineprocess(const double* Ta12, const double* Tb12, int64_t n, const double rigid12_init[12], double scale_init,
                                   int32_t inner_iterations, double mu0, double divid_factor, double min_mu, int32_t ex_max_iter,
                                   int32_t regulation, double regulation_ratio, double rigid12[12], double* scale);

/*
 * ---- ORB-only extrinsic bundle adjustment [SURVEY.md 8(f) row 4] ----
 * One 7-vector vertex x = [rotation vector of R_cl, t_cl, scale] (CalibVertex, Optimizer.cc:40-63: additive update) and
 * N unary reprojection edges (calibEdge, Optimizer.cc:65-205): X_c0 = s * Xw; X_l0 = T_cl^-1 X_c0; X_li = T_lw X_l0;
 * X_ci = T_cl X_li; e = obs - project(X_ci). Per edge: information invSigma2 * I, Huber(sqrt(5.991)).
 * The device evaluates every edge (residual, Jacobian by forward-mode duals exactly as g2o's auto-diff does, robust
 * weight) and reduces the normal equations; the host runs g2o's Levenberg-Marquardt and the reference's four
 * optimise / classify rounds (Optimizer.cc:1511-1556 = 1698-1743). csrc/iba_ba.hip.
 */
typedef struct iba_ba_handle iba_ba_handle;
typedef struct iba_ba_desc {
    int64_t n_edges;
    int32_t n_frames;
    const double* frame_Tlw6;   /* [F*6] e->Tlw_quat: rotation vector and translation (Optimizer.cc:1457-1462 / 1627-1632) */
    const double* frame_intr;   /* [F*4] fx, fy, cx, cy */
    const int32_t* edge_frame;  /* [N] */
    const double* edge_Xw;      /* [N*3] e->Xw (the MapPoint in the reference camera frame, CV_32F values widened) */
    const double* edge_obs;     /* [N*2] kpUn.pt */
    const double* edge_info;    /* [N] invSigma2 = mvInvLevelSigma2[octave] */
    const int32_t* edge_slot;   /* [N] vnIndexEdgeMono: MapPoint slot inside its keyframe — the reference indexes its
                                   outlier flags with it, so flags alias across keyframes (reproduced) */
} iba_ba_desc;
typedef struct iba_ba_result {
    double x[7];
    int32_t n_inliers;          /* nInitialCorrespondences - nBad */
    int32_t n_edges;
    int32_t lm_iterations;      /* over the four rounds */
    int32_t evaluations;        /* device evaluations of the edge set */
    double chi2[4];             /* robustified chi2 of the active edges at the end of each round */
    int32_t n_bad[4];
} iba_ba_result;
iba_status iba_ba_create(const iba_ba_desc* desc, int device, iba_ba_handle** out);
void iba_ba_destroy(iba_ba_handle* h);
const char* iba_ba_last_error(const iba_ba_handle* h);
/* One linearisation at x: H (49, row-major, symmetric), b (7) and the robustified chi2 summed over the edges with
 * active[i] != 0 (NULL = all); robust = 0 drops the Huber kernel (round 4 of the reference). chi2_edges (NULL or [N])
 * receives e^T Omega e of EVERY edge. Sign convention of g2o: solve (H + lambda I) dx = b, x += dx. */
iba_status iba_ba_eval(iba_ba_handle* h, const double* x, const uint8_t* active, int32_t robust, double* H, double* b,
                       double* chi2_robust, double* chi2_edges);
/* OptimizeExtrinsicGlobal / OptimizeExtrinsicLocal schedule on the edge list (which of the two it is depends only on how
 * Xw and Tlw6 were built): 4 x (reset to x0, 10 LM iterations, classify at chi2 > 5.991), kernel off after round 3. */
iba_status iba_ba_optimize(iba_ba_handle* h, const double* x0, iba_ba_result* res);

/* Edge list of the ORB-only extrinsic BA straight from the dataset directory; LiDAR poses as ba_calib.cpp:43-44 passes them.
 * global = 1: OptimizeExtrinsicGlobal constants (Optimizer.cc:1611-1676); global = 0: OptimizeExtrinsicLocal (:1437-1501: MapPoints
 * in the frame of the oldest of the 20 best covisible keyframes, LiDAR pose relative to it). */
typedef struct iba_ba_dataset iba_ba_dataset;
iba_status iba_dataset_load_ba(const iba_dataset_paths* paths, int32_t global, iba_ba_dataset** out);
const iba_ba_desc* iba_ba_dataset_desc(const iba_ba_dataset* d);
void iba_ba_dataset_free(iba_ba_dataset* d);

#ifdef __cplusplus
}
#endif
#endif /* IBA_MI355X_H */
