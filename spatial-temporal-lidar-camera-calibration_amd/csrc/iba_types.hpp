// Shared host/device PODs of the MI355X IBA evaluation path. Device layout decisions:
//  * scans live in HBM as three float32 SoA arrays per frame in kd-tree leaf order (12 B/pt,
//    lossless: KITTI .bin is float32, io_tools.h:170-187), padded to a multiple of 4 with NaN so
//    every lane issues 16-byte loads;
//  * everything per frame hangs off one FrameHdr (offsets into flat device arrays);
//  * everything that depends on the candidate x hangs off one Cand block computed on the host
//    (Sim3Exp uses sin/cos/pow once per evaluation, g2o_tools.h:105-140).
#pragma once
#include <cstdint>

namespace iba {

#ifndef IBA_CHUNK
#define IBA_CHUNK 64
#endif
constexpr int kChunk = IBA_CHUNK;           // points per culling chunk (~3 kd leaves): static AABB, frustum-tested per candidate
constexpr int kCovisWord = 30;       // covisible KFs whose match bits sit beside the two flag bits of the 32-bit flag word (kp_fl) ...
constexpr int kMaxCovis = 62;        // ... and with a second word per keypoint (kp_fl2: slots 30..61, read from global memory only by frames that have them; r04) the limit per frame
constexpr int kPartialStride = 64;   // doubles per candidate in the partial-sum block
constexpr int kMaxChain = 512;       // most candidates one launch chain takes (= IBA_MAX_CHAIN): sizes the per-candidate tables in kernel arguments (one byte each)
#ifndef IBA_GRID_CELL
#define IBA_GRID_CELL 2
#endif
constexpr int kGridCell = IBA_GRID_CELL;   // cell of the 1-bit reject bitmap (px): 2 px = 15 KB of LDS at 1241x376, ~4 % of a scan queued (4 px: 4 KB, 11 %)
#ifndef IBA_COARSE_SHIFT
#define IBA_COARSE_SHIFT (IBA_GRID_CELL == 1 ? 4 : (IBA_GRID_CELL == 2 ? 3 : 2))
#endif
constexpr int kCoarseShift = IBA_COARSE_SHIFT;   // CSR cells of 16 px by default: coarse = fine >> kCoarseShift
#ifndef IBA_THREADS
#define IBA_THREADS 512
#endif
// frame-kernel block size. 512 threads = 8 waves at <= 128 VGPRs and ~50 KB of LDS: TWO blocks share a CU, and while
// one sits in a latency-bound phase (exact association, kd rounds, finalize) the other keeps the SIMDs busy. Measured on
// the C2 shape: 1024 threads + scan staged in LDS (one block per CU) 2.06 ms, 512 threads + scan read through L2 1.57 ms.
constexpr int kThreads = IBA_THREADS;
#ifndef IBA_LEAF_TARGET
#define IBA_LEAF_TARGET 24
#endif
#ifndef IBA_MAX_TREE_DEPTH
#define IBA_MAX_TREE_DEPTH 11
#endif
// the node table (8 B x 2^D) lives in LDS; beyond depth 11 (16 KB) it would push a block over half of the LDS and cost
// the second resident block, so larger scans get larger leaves instead (60 k points: 30 per leaf, 120 k: 59)
constexpr int kMaxTreeDepth = IBA_MAX_TREE_DEPTH;
constexpr int kLeafTarget = IBA_LEAF_TARGET;      // max points per kd-tree leaf

// partial-sum block layout (all doubles; counters < 2^53 carried exactly)
enum Partial {
    P_SUM_3D2D = 0, P_SUM_3D3D, P_HE_SUM, P_HE_CNT, P_CNT_3D2D, P_VALID_3D2D, P_CNT_3D3D, P_VALID_3D3D,
    P_VALID_PL, P_VALID_PT, P_FRAMES, P_NCORR,
    P_H0 = 12,            // 28 upper-triangular entries of H (row-major i<=j)
    P_B0 = 40,            // 7 entries of b
    P_CHI2 = 47, P_COST = 48, P_NF_3D2D = 49, P_NF_P2PL = 50, P_NF_P2PT = 51, P_NRES = 52,
    P_FRAMES_N = 53, P_NCORR_N = 54
};

struct FrameHdr {
    // scan (tree order)
    uint64_t pt_base;     // offset (elements) of this frame in xs/ys/zs/perm/inv_perm/plane arrays
    uint32_t P, Ppad;     // points, padded to x4
    uint64_t box_base;    // offset (chunks) into chunk_box[]: one AABB per kChunk consecutive tree positions
    uint32_t depth;       // kd-tree depth D: 2^D leaves, leaf j = [j*P>>D, (j+1)*P>>D)
    uint32_t node_base;   // offset into nodes[] (2^D - 1 entries)
    // keypoints
    uint64_t kp_base;     // offset into kp arrays
    uint32_t K;
    // keypoint grid
    uint32_t gw, gh;      // cells
    uint32_t gwc, ghc;    // coarse (16 px) cells
    uint64_t coarse_base; // offset into coarse_start[] (gwc*ghc+1 entries)
    uint64_t bitmap_base; // offset into bitmap[] (ceil(gw*gh/32) words)
    // covisibility
    uint32_t slot_base, n_slots;   // slots [slot_base, slot_base+n_slots)
    uint64_t match_base;  // offset into match_uv[] : n_slots * K float2, slot-major
    // camera
    double fx, fy, cx, cy, W, H;
    double Tcw[12];       // widened CV_32F pose, row-major 3x4
    double Tc_next[12];   // widened CV_32F product, unscaled
    double Tl_next[12];
    int32_t he_valid;     // global frame index < F-1
    int32_t global_frame;
    uint64_t mpk_base;    // offset into mpk[]: the keypoints of this frame that own a MapPoint (internal ids, ascending)
    uint32_t n_mpk, n_fk;  // n_fk: entries of the frame's flagged-keypoint list (r05)
    uint64_t fk_base;     // offset into fkp[]: (keypoint id, flag word) of every keypoint that owns a MapPoint or has a covisible match, ascending ids (even offset: read two at a time)
};

struct SlotHdr {
    double rel[12];       // widened CV_32F product, translation unscaled
};

// x-independent local plane at a scan point (kNN(max_pts) clipped to d^2 < r^2 around the point):
// what ComputeAlignmentDist (iba_global.cpp:125-147), ComputeLocalNeighbor (pointcloud.h:733-760) and
// ComputeLocalNormalSingleThre (pointcloud.h:699-717) all derive from.
struct PlaneRec {
    double nx, ny, nz;    // unit eigenvector of the smallest eigenvalue
    double reg_sum;       // sum_i |(p_i - c) . n|   (divide by k-1 at the use site)
    double far_d2;        // squared distance of the farthest kept neighbour
    int32_t k;            // kept neighbours (including the point itself)
    int32_t pad;
};

// per-candidate constants
struct Cand {
    double R[9], t[3], s;     // Sim3Exp(x): LiDAR -> camera
    double Ri[9], ti[3];      // Eigen Isometry inverse: R^T, -(R^T t)
    float s32; float pad0;    // (float)s for the CV_32F MapPoint product (iba_global.cpp:232)
    // Jacobian path: d/dx_k of Sim3Exp(x) (k = 0..5) and of SE3Exp(-x[0:6])
    double dR[3][9];          // dR/d omega_k
    double dt[6][3];          // dt/dx_k
    double Rlc[9], tlc[3];    // SE3Exp(-x[0:6])  (IBACalib2.hpp:573-577)
    double dRlc[3][9];
    double dtlc[6][3];
};

struct DevParams {
    double gate2;                 // max_pixel_dist^2
    double grid_margin;           // max_pixel_dist + 0.01  (exact f64 lookups)
    double bitmap_margin;         // max_pixel_dist + 0.45  (f32 pre-cull bitmap)
    int32_t num_min_corr_cost;
    double corr_3d_2d_threshold, corr_3d_3d_threshold;
    int32_t norm_max_pts, norm_min_pts;
    double norm_radius2, norm_reg_threshold, min_diff_dist2;
    int32_t use_plane, use_3d3d;  // use_3d3d = err_weight[1] > 1e-10
    int32_t num_min_corr;
    double max_3d_dist2, neigh_radius2;
    int32_t neigh_max_pts, neigh_min_pts;
    double local_min_diff_dist2, local_norm_reg_threshold;
    double robust_kernel_delta, robust_kernel_3ddelta;
    int32_t plane_cache;
    int32_t p2pix;                // iba_params.factor_3d2d_kind == 1: the 3d-2d residual is IBATestEdge (the matched scan point reprojected directly) instead of IBA_PlaneFactor
};

// frozen residual block of the Jacobian path (what BuildProblem hands to Ceres)
struct FactorRec {
    int32_t kind;         // 0 IBA_PlaneFactor, 1 Point2Plane, 2 Point2Point
    int32_t frame, kp, nconv;
    double fx, fy, cx, cy, u0, v0;
    double p0[3], n0[3];              // kind 0
    double mp[3], q[3], n[3];         // kind 1/2
    float u1[kMaxCovis], v1[kMaxCovis];
    uint32_t slot[kMaxCovis];         // global slot ids (relative pose lookup)
};

}  // namespace iba
