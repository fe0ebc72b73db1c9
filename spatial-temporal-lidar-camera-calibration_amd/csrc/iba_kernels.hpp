// CDNA4 (gfx950) kernels of the IBA cross-modality evaluation path, part 1 (wave64 / 160 KB LDS): device structs, the closed-form
// pieces (SE3Log, symmetric 3x3 eigenvector), the plane fits, projection and keypoint-grid helpers, and the kernels around the
// evaluation chain of iba_split_kernels.hpp:
//   iba_plane_kernel    x-independent local planes of every scan point (the memo of plane_cache = 1): kNN(<= 64, d^2 < r^2) +
//                       covariance + closed-form eigenvector = ComputeAlignmentDist / ComputeLocalNeighbor /
//                       ComputeLocalNormalSingleThre (iba_global.cpp:125-147, pointcloud.h:699-760); four fits per wave,
//                       then one lane per fit (fit_list_rows / fit_finish_lane, shared with iba_fit_kernel)
//   iba_he_kernel       K7: hand-eye term of every (candidate, frame) (iba_global.cpp:264-276)
//   iba_factor_kernel   residual blocks of a work list -> robust normal equations (IBACalib2.hpp:152-184, 570-625);
//                       iba_factor_mfma_kernel: the same sums on v_mfma_f64_16x16x4_f64 (optional, slower)
//   iba_residual_kernel per-row residuals and Jacobians of the frozen problem
//   iba_reduce_kernel   fixed-order sum of per-frame records (bitwise reproducible)
//
// All arithmetic that decides an index or a gate is IEEE double in the reference's expression order;
// the library is compiled with -ffp-contract=off so no mul+add is fused on either side.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <type_traits>

#include "iba_build.hpp"
#include "iba_types.hpp"

namespace iba {

// ---- the factor kernel's gathers as ONE cache line per keypoint and ONE per scan point (round 6) ----
// iba_factor_kernel is bound by the L2's request rate (~100 G 64-byte requests/s: profiles/r06*): a plane-factor block gathered from seven arrays
// (scan point, plane record, pixel, flag word, three match rows), a 3d-3d block from three. The same values side by side:
struct KpRec {     // 64 B per keypoint, x-independent, built at iba_create
    double cxz, cyz;          // ((u - cx) / fx, (v - cy) / fy): IBA_PlaneFactor's ray (IBACalib2.hpp:165-166)
    float m0u, m0v, m1u, m1v; // the keypoint's matches in the covisible slots 0, 1
    float m2u, m2v;           // ... and 2 (NaN = none; a keyframe with fewer slots: its last slot again, as load_match_pre reads them)
    uint32_t fl, pad0;        // the flag word (kp_fl)
    float mpx, mpy, mpz, pad1;// the MapPoint (kp_mp)
};
struct ScanRec {   // 64 B per scan point (tree order): the point and the unit normal of its memoised local plane; rebuilt with the plane memo
    float x, y, z; uint32_t pad0;
    double nx, ny, nz;
    double pad1[3];
};
static_assert(sizeof(KpRec) == 64 && sizeof(ScanRec) == 64, "one cache line each");

struct DevProblem {
    const FrameHdr* frames;
    const SlotHdr* slots;
    const float* xs; const float* ys; const float* zs;
    const uint32_t* perm;      // tree position -> original point index
    const uint32_t* inv_perm;  // original point index -> tree position
    const float4* pts4;        // the scan again as (x, y, z, original index bits) per tree position: divergent gathers take one 16 B load
    const float* chunk_box;    // [chunk][8]: min xyz, -, max xyz, - of kChunk consecutive tree positions (NaN padding ignored)
    const TreeNode* nodes;
    const float2* kp_uv;
    const float4* kp_mp;       // MapPoint world position (x,y,z); w = 1*(owns a MapPoint) + 2*(matched in >= 1 covisible KF) (the full flag word is kp_fl)
    const uint32_t* kp_fl;     // the same flag word (w) as an integer, for kernels that keep it in LDS
    const uint32_t* coarse_start; const float4* crec; const uint32_t* bitmap;   // keypoint grid (see iba_build.hpp)
    const float2* match_uv;    // [slot][K] matched covisible keypoint, NaN = no match
    const PlaneRec* plane_cost;   // x-independent plane records (norm_radius / norm_max_pts)
    const PlaneRec* plane_local;  // (neigh_radius / neigh_max_pts)
    const uint8_t* plane_ok;      // what the association asks of plane_local, per scan point: bit 0 ComputeLocalNeighbor is valid, bit 1 the plane passes (iba_verdict_kernel; nullptr when the planes are refitted)
    int32_t n_frames;
    int64_t n_kp_total;
    // plane_cache = 0: per-candidate plane records refitted inside every evaluation (slot 0 = frozen problem)
    PlaneRec* scratch_cost; PlaneRec* scratch_local;
    int64_t n_pt_total;
    int32_t scratch_slot_base;
    const uint32_t* mpk;       // per frame: the keypoints that own a MapPoint (FrameHdr::mpk_base, n_mpk)
    const uint32_t* kp_fl2;    // match bits of the covisible slots 30..61 per keypoint (nullptr: no frame has more than 30 slots)
    uint32_t* diag;            // diagnostic counters: [0] association blocks that took a speed-only fallback (a full queue or list: every scan point again),
                               // [1] iba_assoc2_kernel blocks whose note list of possible winners overflowed (every pair beyond the register window again)
    uint32_t max_k;            // largest keypoint count of a frame: row pitch of the per-(frame, keypoint) tables
    const uint2* fkp;          // per frame (FrameHdr::fk_base, n_fk): (keypoint id, flag word) of the keypoints that can own a term — a MapPoint and/or a covisible match —
                               // in ascending id order: the association tail walks these (~40 % of the keypoints), not every keypoint (r05)
    const KpRec* kp_rec;       // [n_kp_total] (r06) ...
    const ScanRec* scan_rec;   // ... [n_pt_total], valid while the planes are memoised (nullptr otherwise: the factor kernel gathers from the separate arrays)
};

struct LdsLayout {   // byte offsets into dynamic LDS, computed on the host from max P/K/D over frames
    uint32_t scan_stride;   // floats per coordinate array (0 = scan not staged in LDS)
    uint32_t off_best_d2, off_best_idx, off_nodes, off_bitmap, off_cstart, off_red, off_vis, vis_words, off_cand, cand_cap, total;
    uint32_t off_pair, pair_cap;   // iba_assoc_kernel only: (point, keypoint) pairs, 16 B each
    uint32_t off_kuv, off_kfl;     // iba_assoc_kernel only: (u, v) and flag word of every keypoint
    uint32_t rel_slots;            // covisible slots the relative-pose slab behind the reduction slab is sized for (the handle's busiest frame)
};
struct KArgs { DevProblem dp; DevParams prm; LdsLayout lay; };   // the frame kernel's parameter blocks, first kernel argument

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kWaves = kThreads / 64;
constexpr int kRedSlots = 16;   // doubles per wave in the reduction slab

__device__ __forceinline__ unsigned long long d2bits(double d) { return (unsigned long long)__double_as_longlong(d); }

// ---- SE3Log on device: g2o::SE3Quat(R,t).log() restated (see oracle/oracle_math.hpp) ----
__host__ __device__ inline void dev_se3log(const double* R, const double* t, double* out) {
    double q[4];
    double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) {
        double s = sqrt(tr + 1.0); q[3] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 4]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[i] = 0.5 * s; s = 0.5 / s;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * s; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * s; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    }
    if (q[3] < 0) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    const double x = q[0] / n, y = q[1] / n, z = q[2] / n, w = q[3] / n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    double r[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx, txz - twy, tyz + twx, 1 - (txx + tyy)};
    const double d = 0.5 * (r[0] + r[4] + r[8] - 1);
    const double dRx = r[7] - r[5], dRy = r[2] - r[6], dRz = r[3] - r[1];
    double ox, oy, oz, c2;
    if (fabs(d) > 0.99999) {
        ox = 0.5 * dRx; oy = 0.5 * dRy; oz = 0.5 * dRz; c2 = 1. / 12.;
    } else {
        const double theta = acos(d);
        const double k = theta / (2 * sqrt(1 - d * d));
        ox = k * dRx; oy = k * dRy; oz = k * dRz;
        c2 = (1 - theta / (2 * tan(theta / 2))) / (theta * theta);
    }
    const double Om[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    double Vi[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            const double o2 = Om[a * 3 + 0] * Om[0 * 3 + b] + Om[a * 3 + 1] * Om[1 * 3 + b] + Om[a * 3 + 2] * Om[2 * 3 + b];
            Vi[a * 3 + b] = ((a == b ? 1.0 : 0.0) - 0.5 * Om[a * 3 + b]) + c2 * o2;
        }
    out[0] = ox; out[1] = oy; out[2] = oz;
    for (int a = 0; a < 3; ++a) out[3 + a] = Vi[a * 3 + 0] * t[0] + Vi[a * 3 + 1] * t[1] + Vi[a * 3 + 2] * t[2];
}

// ---- closed-form symmetric 3x3 eigen solver: pointcloud.h:194-288, 378-463 restated for device ----
__device__ inline void dev_evec0(const double* A, double ev, double* o) {
    const double r0[3] = {A[0] - ev, A[1], A[2]}, r1[3] = {A[1], A[4] - ev, A[5]}, r2[3] = {A[2], A[5], A[8] - ev};
    const double a[3] = {r0[1] * r1[2] - r0[2] * r1[1], r0[2] * r1[0] - r0[0] * r1[2], r0[0] * r1[1] - r0[1] * r1[0]};
    const double b[3] = {r0[1] * r2[2] - r0[2] * r2[1], r0[2] * r2[0] - r0[0] * r2[2], r0[0] * r2[1] - r0[1] * r2[0]};
    const double c[3] = {r1[1] * r2[2] - r1[2] * r2[1], r1[2] * r2[0] - r1[0] * r2[2], r1[0] * r2[1] - r1[1] * r2[0]};
    const double d0 = a[0] * a[0] + a[1] * a[1] + a[2] * a[2], d1 = b[0] * b[0] + b[1] * b[1] + b[2] * b[2], d2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    double dmax = d0; int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { imax = 2; }
    if (imax == 0) { const double s = sqrt(d0); o[0] = a[0] / s; o[1] = a[1] / s; o[2] = a[2] / s; }
    else if (imax == 1) { const double s = sqrt(d1); o[0] = b[0] / s; o[1] = b[1] / s; o[2] = b[2] / s; }
    else { const double s = sqrt(d2); o[0] = c[0] / s; o[1] = c[1] / s; o[2] = c[2] / s; }
}
__device__ inline void dev_evec1(const double* A, const double* e0, double ev1, double* o) {
    double U[3], V[3];
    if (fabs(e0[0]) > fabs(e0[1])) { const double il = 1 / sqrt(e0[0] * e0[0] + e0[2] * e0[2]); U[0] = -e0[2] * il; U[1] = 0; U[2] = e0[0] * il; }
    else { const double il = 1 / sqrt(e0[1] * e0[1] + e0[2] * e0[2]); U[0] = 0; U[1] = e0[2] * il; U[2] = -e0[1] * il; }
    V[0] = e0[1] * U[2] - e0[2] * U[1]; V[1] = e0[2] * U[0] - e0[0] * U[2]; V[2] = e0[0] * U[1] - e0[1] * U[0];
    const double AU[3] = {A[0] * U[0] + A[1] * U[1] + A[2] * U[2], A[1] * U[0] + A[4] * U[1] + A[5] * U[2], A[2] * U[0] + A[5] * U[1] + A[8] * U[2]};
    const double AV[3] = {A[0] * V[0] + A[1] * V[1] + A[2] * V[2], A[1] * V[0] + A[4] * V[1] + A[5] * V[2], A[2] * V[0] + A[5] * V[1] + A[8] * V[2]};
    double m00 = U[0] * AU[0] + U[1] * AU[1] + U[2] * AU[2] - ev1;
    double m01 = U[0] * AV[0] + U[1] * AV[1] + U[2] * AV[2];
    double m11 = V[0] * AV[0] + V[1] * AV[1] + V[2] * AV[2] - ev1;
    const double a00 = fabs(m00), a01 = fabs(m01), a11 = fabs(m11);
    if (a00 >= a11) {
        if (fmax(a00, a01) > 0) {
            if (a00 >= a01) { m01 /= m00; m00 = 1 / sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / sqrt(1 + m00 * m00); m00 *= m01; }
            o[0] = m01 * U[0] - m00 * V[0]; o[1] = m01 * U[1] - m00 * V[1]; o[2] = m01 * U[2] - m00 * V[2];
        } else { o[0] = U[0]; o[1] = U[1]; o[2] = U[2]; }
    } else {
        if (fmax(a11, a01) > 0) {
            if (a11 >= a01) { m01 /= m11; m11 = 1 / sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / sqrt(1 + m11 * m11); m11 *= m01; }
            o[0] = m11 * U[0] - m01 * V[0]; o[1] = m11 * U[1] - m01 * V[1]; o[2] = m11 * U[2] - m01 * V[2];
        } else { o[0] = U[0]; o[1] = U[1]; o[2] = U[2]; }
    }
}
// eigenvector of the smallest eigenvalue of symmetric `cov` (row-major 9), then Eigen-style normalize()
__device__ inline void dev_smallest_evec(const double* cov, double* nrm) {
    double A[9];
    double mc = cov[0];
    for (int i = 1; i < 9; ++i) mc = cov[i] > mc ? cov[i] : mc;
    double v[3] = {0, 0, 0};
    if (mc != 0) {
        for (int i = 0; i < 9; ++i) A[i] = cov[i] / mc;
        const double nn = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (nn > 0) {
            const double q = (A[0] + A[4] + A[8]) / 3;
            const double b00 = A[0] - q, b11 = A[4] - q, b22 = A[8] - q;
            const double p = sqrt((b00 * b00 + b11 * b11 + b22 * b22 + nn * 2) / 6);
            const double c00 = b11 * b22 - A[5] * A[5];
            const double c01 = A[1] * b22 - A[5] * A[2];
            const double c02 = A[1] * A[5] - b11 * A[2];
            const double det = (b00 * c00 - A[1] * c01 + A[2] * c02) / (p * p * p);
            double hd = det * 0.5;
            hd = fmin(fmax(hd, -1.0), 1.0);
            const double angle = acos(hd) / 3.0;
            const double beta2 = cos(angle) * 2;
            const double beta0 = cos(angle + 2.09439510239319549) * 2;
            const double beta1 = -(beta0 + beta2);
            const double e0 = q + p * beta0, e1 = q + p * beta1, e2 = q + p * beta2;
            if (hd >= 0) {
                double v2[3]; dev_evec0(A, e2, v2);
                if (e2 < e0 && e2 < e1) { v[0] = v2[0]; v[1] = v2[1]; v[2] = v2[2]; }
                else {
                    double v1[3]; dev_evec1(A, v2, e1, v1);
                    if (e1 < e0 && e1 < e2) { v[0] = v1[0]; v[1] = v1[1]; v[2] = v1[2]; }
                    else { v[0] = v1[1] * v2[2] - v1[2] * v2[1]; v[1] = v1[2] * v2[0] - v1[0] * v2[2]; v[2] = v1[0] * v2[1] - v1[1] * v2[0]; }
                }
            } else {
                double v0[3]; dev_evec0(A, e0, v0);
                if (e0 < e1 && e0 < e2) { v[0] = v0[0]; v[1] = v0[1]; v[2] = v0[2]; }
                else {
                    double v1[3]; dev_evec1(A, v0, e1, v1);
                    if (e1 < e0 && e1 < e2) { v[0] = v1[0]; v[1] = v1[1]; v[2] = v1[2]; }
                    else { v[0] = v0[1] * v1[2] - v0[2] * v1[1]; v[1] = v0[2] * v1[0] - v0[0] * v1[2]; v[2] = v0[0] * v1[1] - v0[1] * v1[0]; }
                }
            }
        } else {
            if (cov[0] < cov[4] && cov[0] < cov[8]) v[0] = 1;
            else if (cov[4] < cov[0] && cov[4] < cov[8]) v[1] = 1;
            else v[2] = 1;
        }
    }
    const double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (z > 0) { const double n = sqrt(z); nrm[0] = v[0] / n; nrm[1] = v[1] / n; nrm[2] = v[2] / n; }
    else { nrm[0] = v[0]; nrm[1] = v[1]; nrm[2] = v[2]; }
}

// ---- plane fits, 64 per wave, in two phases ----
// Phase 1 (fit_list_rows): the sorted neighbour lists, FOUR at a time, one DPP row (16 lanes) per list. The list of a row
// lives SLOTS entries per lane (entry i = SLOTS * lane + slot; SLOTS = 2 serves max_pts <= 32, SLOTS = 4 up to 64). The
// traversal state is per row. A leaf is tested 16 points per row and pass; candidate L of every row is broadcast with
// row_newbcast:L and inserted IN PLACE: new[i] = min(old[i], max(old[i - 1], candidate)) (row_shr:1 brings old[i - 1] across
// lanes) — no insertion index, no ballot. A candidate that does not qualify is inserted as +inf, which moves nothing; the
// pruning bound (the max_pts-th entry) is refreshed once per pass, which is safe: a candidate admitted under a stale bound
// sorts behind max_pts entries and is never looked at. Same list, entry for entry, as plane_fit_wave (equal distances keep
// their visiting order, as nanoflann's KNNResultSet does).
// Phase 2 (fit_finish_lane): ONE LANE PER FIT. Covariance in list order (ComputeCovariance, pointcloud.h:126-158), the
// closed-form eigenvector and the regularity sum are serial per fit anyway; with a lane each nothing is computed 16 times.
// All 64 lanes must be active; one wave per workgroup (the __syncthreads order the wave's own LDS traffic).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {   // lanes without a source lane read 0
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// v_min_f64 / v_max_f64 on values known not to be NaN (fmin / fmax would first canonicalise what came through a DPP move)
__device__ __forceinline__ double f64_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double f64_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
constexpr int kDppRowShr1 = 0x111, kDppRowBcast0 = 0x150;   // row_shr:1; row_newbcast:0 (+L: lane L of each row to the row)

template <int L, int SLOTS>
__device__ __forceinline__ void fit_insert(double d2q, uint32_t base, double (&ed)[SLOTS], uint32_t (&ep)[SLOTS]) {
    const double cd = dpp_f64<kDppRowBcast0 + L>(d2q);
    const uint32_t cp = base + (uint32_t)L;
    const double pd = dpp_f64<kDppRowShr1>(ed[SLOTS - 1]);   // lane 0 of a row: 0 <= any squared distance, so entry 0 takes the candidate itself
    const uint32_t pp = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)ep[SLOTS - 1], kDppRowShr1, 0xf, 0xf, true);
    double nd[SLOTS]; uint32_t np[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const double prevd = s ? ed[s ? s - 1 : 0] : pd; const uint32_t prevp = s ? ep[s ? s - 1 : 0] : pp;
        const bool keep = ed[s] <= cd, pk = prevd <= cd;   // equal distances: the earlier visit stays in front
        nd[s] = f64_min(ed[s], f64_max(prevd, cd));        // = keep ? ed[s] : (pk ? cd : prevd) for a sorted list
        np[s] = keep ? ep[s] : (pk ? cp : prevp);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) { ed[s] = nd[s]; ep[s] = np[s]; }
}

constexpr int kFitPathMax = ((kMaxTreeDepth + 3) / 4) * 4;   // >= the builder's depth cap (kMaxTreeDepth = 11)
constexpr int kFitListStride(int slots) { return 16 * slots + 1; }   // words per list in LDS (odd: lane-per-fit reads are conflict-free)
template <int SLOTS> struct FitLds {
    uint32_t list[64][kFitListStride(SLOTS)];   // tree positions of the kept neighbours, nearest first
    double far_d2[64];
    int32_t count[64];
    double first_d2[4][16]; uint32_t first_pos[4][16];   // the four rows' first pass, sorted (see fit_list_rows)
};

// lists of the four scan points `cpos` (one per row; kNone: idle row) -> lds.list[fit], count[fit], far_d2[fit]; fit = fit0 + row
template <int SLOTS>
__device__ __forceinline__ void fit_list_rows(const float4* __restrict__ p4, const TreeNode* __restrict__ nodes, uint32_t P, uint32_t D, uint32_t cpos, double r2, int max_pts,
                                     FitLds<SLOTS>& lds, int fit0) {
    const int lane = threadIdx.x & 63, gl = lane & 15, row_sh = lane & 48;
    const bool act = cpos != kNone;
    double qx = 0, qy = 0, qz = 0;
    float qxf = 0.f, qyf = 0.f, qzf = 0.f;
    if (act) { const float4 c = p4[cpos]; qxf = c.x; qyf = c.y; qzf = c.z; qx = (double)c.x; qy = (double)c.y; qz = (double)c.z; }
    double ed[SLOTS]; uint32_t ep[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) { ed[s] = INFINITY; ep[s] = kNone; }
    double bound = r2;
    const uint32_t first_leaf = (1u << D) - 1u;
    const int kth_lane = row_sh + (max_pts - 1) / SLOTS, kth_slot = (max_pts - 1) % SLOTS;
    // Traversal state of the row: the current root-to-leaf path. pl[L] = a float LOWER bound of the squared distance from the
    // query to the splitting plane of level L, side bit L = the child taken there, done bit L = its other child needs no
    // visit any more. Backtracking picks the deepest level whose far side may still hold a point below the pruning bound —
    // compares on registers, no node is re-read on the way up. (A lower bound can only add a leaf none of whose points
    // qualifies; the leaves are still visited in the exact order, so equal distances keep theirs.)
    float pl[kFitPathMax];
#pragma unroll
    for (int L = 0; L < kFitPathMax; ++L) pl[L] = INFINITY;
    uint32_t side = 0u, done = 0u, node = 0u;
    int go = act ? 0 : -1;   // level to (re)start the descent at; -1: the row has finished
    bool first = true;       // wave-uniform: no candidate has been looked at yet
    while (__ballot(go >= 0) != 0ull) {
        uint32_t lo = 0, hi = 0;
        if (go >= 0) {
            uint32_t n1 = 1u;   // heap index + 1 of the node the descent continues from
            if (go > 0) {       // enter the far child of the ancestor at level go - 1
                const int t = go - 1;
                const uint32_t anc1 = (node + 1u) >> (D - (uint32_t)t);
                done |= 1u << t; side ^= 1u << t;
                n1 = (anc1 << 1) | ((side >> t) & 1u);
                const uint32_t keep = (1u << go) - 1u;
                side &= keep; done &= keep;
            }
#pragma unroll
            for (int L = 0; L < kFitPathMax; ++L) {
                if (L >= (int)D) break;
                if (L >= go) {
                    const TreeNode n = nodes[n1 - 1u];
                    const float df = (n.dim == 0 ? qxf : (n.dim == 1 ? qyf : qzf)) - n.split;   // the query is a scan point: float32-exact, so the sign is the exact one
                    const uint32_t r = df >= 0.0f ? 1u : 0u;
                    pl[L] = df * df * 0.9999996f;   // <= the exact squared difference (two roundings of 2^-24 each)
                    side |= r << L;
                    n1 = (n1 << 1) | r;
                }
            }
            node = n1 - 1u;
            const uint32_t j = node - first_leaf;
            lo = (uint32_t)(((uint64_t)j * P) >> D); hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
        }
        for (uint32_t base = lo; __ballot(base < hi) != 0ull; base += 16u) {
            const uint32_t i = base + (uint32_t)gl;
            double d2 = INFINITY;
            if (i < hi) {
                const float4 v = p4[i];
                const double dx = qx - (double)v.x, dy = qy - (double)v.y, dz = qz - (double)v.z;
                d2 = (dx * dx + dy * dy) + dz * dz;
            }
            const bool qual = d2 < bound;
            const unsigned long long m = __ballot(qual);
            if (m == 0ull) continue;
            const double d2q = qual ? d2 : INFINITY;
            if (first) {
                // the first pass of every row meets an empty list: rank its 16 candidates by (distance, lane) — 15 rotations within the
                // row — and drop them into place through LDS, instead of 16 insertions
                first = false;
                uint32_t rank = 0u;
#define IBA_FIT_RANK(KK) { const double od = dpp_f64<0x120 + KK>(d2q); const int ol = __builtin_amdgcn_update_dpp(0, gl, 0x120 + KK, 0xf, 0xf, true); /* row_ror:KK */ \
                        rank += (od < d2q || (od == d2q && ol < gl)) ? 1u : 0u; }
                IBA_FIT_RANK(1) IBA_FIT_RANK(2) IBA_FIT_RANK(3) IBA_FIT_RANK(4) IBA_FIT_RANK(5) IBA_FIT_RANK(6) IBA_FIT_RANK(7) IBA_FIT_RANK(8)
                IBA_FIT_RANK(9) IBA_FIT_RANK(10) IBA_FIT_RANK(11) IBA_FIT_RANK(12) IBA_FIT_RANK(13) IBA_FIT_RANK(14) IBA_FIT_RANK(15)
#undef IBA_FIT_RANK
                const int row = lane >> 4;
                lds.first_d2[row][rank] = d2q; lds.first_pos[row][rank] = qual ? i : kNone;
                __syncthreads();
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    const int e = SLOTS * gl + s;
                    ed[s] = e < 16 ? lds.first_d2[row][e & 15] : INFINITY; ep[s] = e < 16 ? lds.first_pos[row][e & 15] : kNone;
                }
                __syncthreads();
            } else {
            const uint32_t m16 = (uint32_t)(m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffu;   // lanes-of-a-row with a candidate in ANY row
#define IBA_FIT_STEP(L) if (m16 & (1u << L)) fit_insert<L, SLOTS>(d2q, base, ed, ep);
            IBA_FIT_STEP(0) IBA_FIT_STEP(1) IBA_FIT_STEP(2) IBA_FIT_STEP(3) IBA_FIT_STEP(4) IBA_FIT_STEP(5) IBA_FIT_STEP(6) IBA_FIT_STEP(7)
            IBA_FIT_STEP(8) IBA_FIT_STEP(9) IBA_FIT_STEP(10) IBA_FIT_STEP(11) IBA_FIT_STEP(12) IBA_FIT_STEP(13) IBA_FIT_STEP(14) IBA_FIT_STEP(15)
#undef IBA_FIT_STEP
            }
            double kv = ed[0];
#pragma unroll
            for (int s = 1; s < SLOTS; ++s) kv = kth_slot == s ? ed[s] : kv;
            bound = fmin(r2, __shfl(kv, kth_lane));
        }
        if (go >= 0) {
            const float bf = (float)bound * 1.0000002f;   // >= bound
            uint32_t cnd = 0u;
#pragma unroll
            for (int L = 0; L < kFitPathMax; ++L) cnd |= (pl[L] < bf ? 1u : 0u) << L;
            cnd &= ~done & ((1u << D) - 1u);
            done |= ~cnd;            // a level that fails now fails for good: the bound only shrinks
            go = cnd ? 32 - __clz((int)cnd) : -1;   // deepest candidate level + 1
        }
    }
    // kept neighbours of the row: finite entries below max_pts
    int count = 0;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const unsigned long long m = __ballot(SLOTS * gl + s < max_pts && ed[s] < INFINITY);
        count += __popc((uint32_t)(m >> row_sh) & 0xffffu);
    }
    const int fit = fit0 + (lane >> 4);
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int i = SLOTS * gl + s;
        if (i < count) {
            lds.list[fit][i] = ep[s];
            if (i == count - 1) lds.far_d2[fit] = ed[s];
        }
    }
    if (gl == 0) lds.count[fit] = count;
}

// the plane of one fit from its neighbour list (this lane's own fit; any subset of lanes)
__device__ __forceinline__ PlaneRec fit_finish_lane(const float4* __restrict__ p4, uint32_t cpos, const uint32_t* list, int count, double far_d2) {
    PlaneRec rec;
    rec.k = count; rec.pad = 0;
    rec.far_d2 = count > 0 ? far_d2 : 0.0;
    const float4 cq = p4[cpos];
    const double qx = (double)cq.x, qy = (double)cq.y, qz = (double)cq.z;
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < count; ++j) {
        const float4 v = p4[list[j]];
        const double x = (double)v.x, y = (double)v.y, z = (double)v.z;
        c[0] += x; c[1] += y; c[2] += z;
        c[3] += x * x; c[4] += x * y; c[5] += x * z; c[6] += y * y; c[7] += y * z; c[8] += z * z;
    }
    const double inv_n = (double)count;
    for (int i = 0; i < 9; ++i) c[i] /= inv_n;
    double cov[9];
    cov[0] = c[3] - c[0] * c[0]; cov[4] = c[6] - c[1] * c[1]; cov[8] = c[8] - c[2] * c[2];
    cov[1] = cov[3] = c[4] - c[0] * c[1]; cov[2] = cov[6] = c[5] - c[0] * c[2]; cov[5] = cov[7] = c[7] - c[1] * c[2];
    double nrm[3]; dev_smallest_evec(cov, nrm);
    double reg = 0.0;
    for (int j = 0; j < count; ++j) {
        const float4 v = p4[list[j]];
        const double ax = (double)v.x - qx, ay = (double)v.y - qy, az = (double)v.z - qz;
        reg += fabs(ax * nrm[0] + ay * nrm[1] + az * nrm[2]);   // |(p_j - c) . n|
    }
    rec.nx = nrm[0]; rec.ny = nrm[1]; rec.nz = nrm[2]; rec.reg_sum = reg;
    return rec;
}

// grid: (ceil(maxP / 64), n_frames) workgroups of one wave: 64 consecutive scan points (tree order)
#ifndef IBA_PLANE_WAVES
#define IBA_PLANE_WAVES 4
#endif
template <int SLOTS>
__global__ __launch_bounds__(64) void iba_plane_kernel(DevProblem dp, double r2, int max_pts, PlaneRec* out) {
    __shared__ FitLds<SLOTS> s_fit;
    const FrameHdr& h = dp.frames[blockIdx.y];
    const uint32_t pos0 = blockIdx.x * 64u;
    if (pos0 >= h.P) return;
    const float4* p4 = dp.pts4 + h.pt_base;
    const int lane = threadIdx.x;
    for (int r = 0; r < 16; ++r) {
        const uint32_t pos = pos0 + 4u * (uint32_t)r + (uint32_t)(lane >> 4);
        if (pos0 + 4u * (uint32_t)r >= h.P) break;
        fit_list_rows<SLOTS>(p4, dp.nodes + h.node_base, h.P, h.depth, pos < h.P ? pos : kNone, r2, max_pts, s_fit, 4 * r);
    }
    __syncthreads();
    const uint32_t pos = pos0 + (uint32_t)lane;
    if (pos < h.P) out[h.pt_base + pos] = fit_finish_lane(p4, pos, s_fit.list[lane], s_fit.count[lane], s_fit.far_d2[lane]);
}

// debug (iba_debug_knn): the sorted neighbour lists iba_plane_kernel builds around the scan points of ONE frame, dumped instead of fitted:
// per tree position the kept neighbours nearest first — original point index and exact squared distance, the expression of the list
// builder itself — and their number. Same grid as iba_plane_kernel for that frame, same fit_list_rows. The parity tests lay these
// lists beside the reference's own nanoflann kNN(30) (tests/golden/knn_nanoflann_v150.npz: iba_global.cpp:125-133).
template <int SLOTS>
__global__ __launch_bounds__(64) void iba_knn_dump_kernel(DevProblem dp, int frame, double r2, int max_pts, uint32_t* __restrict__ out_idx, double* __restrict__ out_d2, int32_t* __restrict__ out_cnt) {
    __shared__ FitLds<SLOTS> s_fit;
    const FrameHdr& h = dp.frames[frame];
    const uint32_t pos0 = blockIdx.x * 64u;
    if (pos0 >= h.P) return;
    const float4* p4 = dp.pts4 + h.pt_base;
    const uint32_t* perm = dp.perm + h.pt_base;
    const int lane = threadIdx.x;
    for (int r = 0; r < 16; ++r) {
        const uint32_t pos = pos0 + 4u * (uint32_t)r + (uint32_t)(lane >> 4);
        if (pos0 + 4u * (uint32_t)r >= h.P) break;
        fit_list_rows<SLOTS>(p4, dp.nodes + h.node_base, h.P, h.depth, pos < h.P ? pos : kNone, r2, max_pts, s_fit, 4 * r);
    }
    __syncthreads();
    const uint32_t pos = pos0 + (uint32_t)lane;
    if (pos >= h.P) return;
    const int count = s_fit.count[lane];
    out_cnt[pos] = count;
    const float4 cq = p4[pos];
    const double qx = (double)cq.x, qy = (double)cq.y, qz = (double)cq.z;
    for (int j = 0; j < count; ++j) {
        const uint32_t pj = s_fit.list[lane][j];
        const float4 v = p4[pj];
        const double dx = qx - (double)v.x, dy = qy - (double)v.y, dz = qz - (double)v.z;
        out_idx[(size_t)pos * max_pts + j] = perm[pj];
        out_d2[(size_t)pos * max_pts + j] = (dx * dx + dy * dy) + dz * dz;
    }
}

// ---- wave64 sum on the VALU (DPP row shifts + row broadcasts, no LDS traffic); total lands in lane 63 ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long b) {
    int lo = (int)(unsigned int)b, hi = (int)(unsigned int)(b >> 32);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)(unsigned int)hi << 32) | (unsigned long long)(unsigned int)lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) { return __longlong_as_double((long long)dpp_u64<CTRL, ROW_MASK>((unsigned long long)__double_as_longlong(x))); }
__device__ __forceinline__ double wave_sum_f64(double x) {   // fixed association order => bitwise reproducible
    x += dpp_f64<0x111, 0xf>(x); x += dpp_f64<0x112, 0xf>(x); x += dpp_f64<0x114, 0xf>(x); x += dpp_f64<0x118, 0xf>(x);   // row_shr 1,2,4,8
    x += dpp_f64<0x142, 0xa>(x);   // row_bcast:15 -> rows 1,3
    x += dpp_f64<0x143, 0xc>(x);   // row_bcast:31 -> rows 2,3
    return x;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t x) {   // inclusive prefix over the lanes of the wave (the total in lane 63)
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false); x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, false); x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, false);
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xa, 0xf, false); x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xc, 0xf, false);
    return x;
}
__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long x) {
    x += dpp_u64<0x111, 0xf>(x); x += dpp_u64<0x112, 0xf>(x); x += dpp_u64<0x114, 0xf>(x); x += dpp_u64<0x118, 0xf>(x);
    x += dpp_u64<0x142, 0xa>(x); x += dpp_u64<0x143, 0xc>(x);
    return x;
}
// ---- fixed-order block reduction of NV doubles per thread; the totals are returned in v[] on every thread ----
template <int NV>
__device__ inline void block_reduce(double* v, double* s_red /* (kWaves+1)*kRedSlots doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        double x = v[i];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
        v[i] = x;
    }
    __syncthreads();
    if (lane == 0)
        for (int i = 0; i < NV; ++i) s_red[wave * kRedSlots + i] = v[i];
    __syncthreads();
    if (threadIdx.x < NV) {
        double x = 0;
        for (int w = 0; w < kWaves; ++w) x += s_red[w * kRedSlots + threadIdx.x];
        s_red[kWaves * kRedSlots + threadIdx.x] = x;
    }
    __syncthreads();
    for (int i = 0; i < NV; ++i) v[i] = s_red[kWaves * kRedSlots + i];
    __syncthreads();
}

struct FrameCtx {   // wave-uniform per-block context
    const float* xs; const float* ys; const float* zs;   // LDS or HBM
    const float4* p4;                                    // HBM (x, y, z, index bits) per tree position; nullptr when the scan is staged in LDS
    const TreeNode* nodes;                               // LDS
    const uint32_t* bitmap;                              // LDS
    unsigned long long* best_d2; uint32_t* best_idx;     // LDS
    const uint16_t* cstart; int gwc;     // coarse CSR starts (LDS)
    const float4* crec;                  // HBM, frame-relative: (u, v, id bits, 0) sorted by coarse cell
    const uint32_t* perm;                                // HBM, frame-relative
    int gw, gh;
    float margin;
    double gate2;
    double fx, cx, cy, W, H;
    double R[9], t[3];
};

// ---- two IEEE quotients by ONE denominator (round 5) ----
// A pinhole projection divides twice by the same depth. The compiler's f64 division (LLVM AMDGPU LowerFDIV64) is, per quotient,
//   s0 = div_scale(den), s1 = div_scale(num); r = rcp(s0); two Newton steps on r (4 fma); q0 = s1 r; e = fma(-s0, q0, s1);
//   q = div_fmas(e, r, q0); div_fixup
// i.e. 11 instructions, one of them (v_rcp_f64) quarter rate, of which everything up to the refined reciprocal depends on the denominator
// alone. div2 computes that reciprocal once and the last three steps per numerator — the same operations on the same values, so the same,
// correctly rounded, quotients — WITHOUT the scaling steps, which is exact only while no intermediate leaves the normal range:
//   2^-100 <= |den| < 2^100 (checked first: the reciprocal and its refinement are then plain normal arithmetic), and
//   |quotient| >= 2^-700 (checked afterwards: then |num| = |q den| >= 2^-801, and the remainder e ~ 2^-53 num is not denormal);
//   upwards nothing can go wrong that the true quotient does not share (q0 overflows only if the quotient itself is beyond 2^1023).
// Anything else — zero, denormal, huge, NaN — takes the compiler's division. tests/test_gpu_division.py (through the debug entry
// iba_debug_div2_selftest) compares the two bit for bit over random and edge-case operands.
__device__ __forceinline__ void div2(const double n0, const double n1, const double den, double& q0, double& q1) {
    const uint32_t hi = (uint32_t)__double2hiint(den) & 0x7fffffffu;
    bool ok = (hi - 0x39b00000u) < (0x46300000u - 0x39b00000u);   // exponent field in [1023 - 100, 1023 + 100)
    if (ok) {
        const double r0 = __builtin_amdgcn_rcp(den);
        const double e0 = __builtin_fma(-den, r0, 1.0);
        const double r1 = __builtin_fma(r0, e0, r0);
        const double e1 = __builtin_fma(-den, r1, 1.0);
        const double r = __builtin_fma(r1, e1, r1);
        const double a0 = n0 * r, a1 = n1 * r;
        q0 = __builtin_fma(__builtin_fma(-den, a0, n0), r, a0);
        q1 = __builtin_fma(__builtin_fma(-den, a1, n1), r, a1);
        ok = fabs(q0) >= 0x1p-700 && fabs(q1) >= 0x1p-700;   // (NaN: not ok)
    }
    if (!ok) { q0 = n0 / den; q1 = n1 / den; }
}

// debug (iba_debug_div2_selftest): div2 beside the compiler's division on operands from the host; fast[i] = the denominator passed div2's first test
__global__ __launch_bounds__(256) void iba_div2_selftest_kernel(const double* __restrict__ n0, const double* __restrict__ n1, const double* __restrict__ den, long long n,
                                                                double* __restrict__ q0, double* __restrict__ q1, double* __restrict__ r0, double* __restrict__ r1, unsigned long long* __restrict__ n_fast) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a, b;
    div2(n0[i], n1[i], den[i], a, b);
    q0[i] = a; q1[i] = b;
    r0[i] = n0[i] / den[i]; r1[i] = n1[i] / den[i];
    const uint32_t hi = (uint32_t)__double2hiint(den[i]) & 0x7fffffffu;
    if ((hi - 0x39b00000u) < (0x46300000u - 0x39b00000u) && fabs(a) >= 0x1p-700 && fabs(b) >= 0x1p-700) atomicAdd(n_fast, 1ull);
}

// K1+K2: Tcl*p (pointcloud.h:82-86), pinhole projection and FOV cull (iba_global.cpp:68-81) for one scan point
__device__ __forceinline__ bool project_uv(const FrameCtx& c, float xf, float yf, float zf, double& u, double& v) {
    const double x = (double)xf, y = (double)yf, z = (double)zf;
    const double pcx = ((c.R[0] * x + c.R[1] * y) + c.R[2] * z) + c.t[0];
    const double pcy = ((c.R[3] * x + c.R[4] * y) + c.R[5] * z) + c.t[1];
    const double pcz = ((c.R[6] * x + c.R[7] * y) + c.R[8] * z) + c.t[2];
    if (!(pcz > 0)) return false;
    div2(c.fx * pcx + c.cx * pcz, c.fx * pcy + c.cy * pcz, pcz, u, v);   // (K p) / z; fx for both on purpose: iba_global.cpp:73
    return 0 <= u && u < c.W && 0 <= v && v < c.H;
}
// one scan point by tree position, where the lanes of a wave ask for unrelated positions: one 16 B gather instead of three
// 4 B gathers (the texture path handles one lane's address per clock either way)
template <bool AOS>
__device__ __forceinline__ void load_pt(const FrameCtx& c, uint32_t pos, float& x, float& y, float& z) {
    if (AOS) { const float4 v = c.p4[pos]; x = v.x; y = v.y; z = v.z; }
    else { x = c.xs[pos]; y = c.ys[pos]; z = c.zs[pos]; }
}
template <bool AOS>
__device__ __forceinline__ bool project_pos(const FrameCtx& c, uint32_t pos, double& u, double& v) {
    float x, y, z;
    load_pt<AOS>(c, pos, x, y, z);
    return project_uv(c, x, y, z, u, v);
}
// one LDS bit: can any keypoint be within max_pixel_dist of this pixel?
__device__ __forceinline__ bool near_keypoint(const FrameCtx& c, double u, double v) {
    const uint32_t cell = (uint32_t)grid_cell((float)v, c.gh) * (uint32_t)c.gw + (uint32_t)grid_cell((float)u, c.gw);
    return (c.bitmap[cell >> 5] >> (cell & 31)) & 1u;
}
// K3: exact 1-NN of every keypoint among the projected points, inverted: the projected point visits the
// keypoints of the <= 2x2 grid cells around it. PASS 1: ds_min_u64 on the keypoint's best d^2.
// PASS 2: resolve exact ties by the lowest original point index.
template <int PASS>
// returns (PASS 1) how many keypoints this point was the FIRST to reach: ds_min_u64 hands back the keypoint's previous best, and exactly one
// caller per keypoint sees the initial ~0 — the sum over a block is corrset.size() (r05: the tail no longer walks every keypoint to count them)
__device__ __forceinline__ uint32_t grid_match(const FrameCtx& c, double u, double v, uint32_t pos) {
    const float uf = (float)u, vf = (float)v;
    const int x0 = grid_cell(uf - c.margin, c.gw) >> kCoarseShift, x1 = grid_cell(uf + c.margin, c.gw) >> kCoarseShift;
    const int y0 = grid_cell(vf - c.margin, c.gh) >> kCoarseShift, y1 = grid_cell(vf + c.margin, c.gh) >> kCoarseShift;
    uint32_t hit = 0u;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t e0 = c.cstart[yy * c.gwc + x0], e1 = c.cstart[yy * c.gwc + x1 + 1];
        for (uint32_t e = e0; e < e1; ++e) {
            const float4 rec = c.crec[e];
            if (fabsf(rec.x - uf) > c.margin || fabsf(rec.y - vf) > c.margin) continue;   // cheap f32 reject (margin has 0.01 px slack)
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

constexpr int kPathMax = ((kMaxTreeDepth + 3) / 4) * 4;   // levels of the search path whose plane bounds are kept in registers (the builder caps D at kMaxTreeDepth = 11)
static_assert(kPathMax % 4 == 0 && kPathMax >= kMaxTreeDepth, "register path must cover the depth cap");
// a closer point, or an equally close one with a lower original index, replaces the best so far
__device__ __forceinline__ void nn_merge(double& best, uint32_t& bpos, double od, uint32_t op, const uint32_t* __restrict__ perm_g) {
    if (od < best) { best = od; bpos = op; }
    else if (od == best && op != kNone && op != bpos) { if (bpos == kNone || perm_g[op] < perm_g[bpos]) bpos = op; }
}
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// K7: hand-eye consistency term of every (candidate, frame) pair (iba_global.cpp:264-276); one lane each.
// The frame kernel adds he[b][f] only for frames that pass the corrset test.
// Two lanes per (candidate, frame): the even lane takes log(T_cl Tl), the odd one log(Tc T_cl) — the two halves of the term are
// independent chains of ~1000 dependent f64 instructions each, and a small batch has more lanes than work.
__device__ __forceinline__ double he_term(const FrameHdr& h, const Cand& cd, const int which, const bool live) {
    double l[6] = {0, 0, 0, 0, 0, 0};
    if (live && h.he_valid) {
        double CR[9], Ct[3];
        const double* Tl = h.Tl_next; const double* Tc = h.Tc_next;
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc)
                CR[r * 3 + cc] = which == 0 ? (cd.R[r * 3 + 0] * Tl[0 * 4 + cc] + cd.R[r * 3 + 1] * Tl[1 * 4 + cc]) + cd.R[r * 3 + 2] * Tl[2 * 4 + cc]
                                            : (Tc[r * 4 + 0] * cd.R[0 * 3 + cc] + Tc[r * 4 + 1] * cd.R[1 * 3 + cc]) + Tc[r * 4 + 2] * cd.R[2 * 3 + cc];
            Ct[r] = which == 0 ? ((cd.R[r * 3 + 0] * Tl[3] + cd.R[r * 3 + 1] * Tl[7]) + cd.R[r * 3 + 2] * Tl[11]) + cd.t[r]
                               : ((Tc[r * 4 + 0] * cd.t[0] + Tc[r * 4 + 1] * cd.t[1]) + Tc[r * 4 + 2] * cd.t[2]) + Tc[r * 4 + 3] * cd.s;
        }
        dev_se3log(CR, Ct, l);
    }
    double ss = 0;
    for (int k = 0; k < 6; ++k) {   // the even lane holds l1, its neighbour l2
        const double other = __shfl_xor(l[k], 1);
        ss += (l[k] - other) * (l[k] - other);
    }
    return sqrt(ss);
}
__global__ __launch_bounds__(64) void iba_he_kernel(DevProblem dp, const Cand* __restrict__ cands, int B, double* __restrict__ he) {
    const int i = blockIdx.x * 32 + (threadIdx.x >> 1);
    const bool live = i < B * dp.n_frames;
    const int ii = live ? i : 0;
    const double v = he_term(dp.frames[ii % dp.n_frames], cands[ii / dp.n_frames], threadIdx.x & 1, live);
    if (live && !(threadIdx.x & 1)) he[i] = v;
}
// the head of an evaluation in one launch: blocks [0, n_fetch) copy the candidate block (iba_fetch_kernel), the others compute
// K7 from the candidates where they lie in pinned host memory (13 doubles per lane, two distinct candidates per wave at most)
__global__ __launch_bounds__(64) void iba_fetch_he_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16, uint32_t n_fetch,
                                                          DevProblem dp, const Cand* __restrict__ cands_host, int B, double* __restrict__ he) {
    if (blockIdx.x < n_fetch) {
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < min(n16, (blockIdx.x + 1u) * 256u); i += 64u) dst[i] = src[i];
        return;
    }
    const int i = (int)(blockIdx.x - n_fetch) * 32 + (int)(threadIdx.x >> 1);
    const bool live = i < B * dp.n_frames;
    const int ii = live ? i : 0;
    const double v = he_term(dp.frames[ii % dp.n_frames], cands_host[ii / dp.n_frames], threadIdx.x & 1, live);
    if (live && !(threadIdx.x & 1)) he[i] = v;
}

// grid: 8 * ceil(n_frames/8) * B blocks of kThreads. Block i runs on XCD i%8 (round-robin dispatch), so
// all candidates of one frame share that XCD's L2 copy of the scan.
// The frame kernel reads its first argument through __builtin_amdgcn_kernarg_segment_ptr() at offset 0. This probe,
// launched once per handle with the same argument shape, confirms that the explicit arguments do start there.
__global__ void iba_kernarg_probe_kernel(KArgs ka_by_value, int32_t* ok) {
    typedef __attribute__((address_space(4))) const KArgs KArgsC;
    KArgsC* ka = (KArgsC*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    *ok = (ka->lay.total == ka_by_value.lay.total && ka->lay.cand_cap == ka_by_value.lay.cand_cap && ka->dp.n_frames == ka_by_value.dp.n_frames &&
           ka->dp.frames == ka_by_value.dp.frames && ka->prm.gate2 == ka_by_value.prm.gate2 && ka->prm.plane_cache == ka_by_value.prm.plane_cache) ? 1 : 0;
}

// ---- Jacobian path: residual blocks of the frozen association, evaluated at candidate x ----
// IBA_PlaneFactor (IBACalib2.hpp:152-184), Point2Plane/Point2Point_Factor (:570-584, 611-625), Huber
// IRLS weights as Ceres' Corrector applies them (rho'' <= 0), accumulated as the upper triangle of
// H = sum w J^T J, b = sum w J^T r. Derivatives are analytic: the chain rule through the same
// expressions the reference's Jets differentiate, with dR/dx, dt/dx from the host duals (Cand).
// Round 6: the chain is written with EXPLICIT fused multiply-adds (fma()). The library is still compiled with -ffp-contract=off, so
// exactly the products written as fma() fuse and nothing else: the arithmetic is a fixed sequence of IEEE operations that the oracle's
// kernel-order mirror (oracle/iba_oracle.cpp, plane_block_kernel_order) repeats with std::fma bit for bit. A fused chain rounds once
// where the unfused one rounds twice: per entry it is as close to the exact value as the reference's own double arithmetic or closer
// (measured against a long-double evaluation: tests/test_gpu_conditioning.py), and it is a third fewer f64 instructions
// in a kernel that is bound by f64 issue. (Rounds 2-5 kept the chain unfused and offered the compiler's contraction as a build switch,
// -DIBA_JAC_FMA: faster, but which products fuse was then the compiler's choice and could not be mirrored on the CPU.)
// What no build can remove: a plane factor whose viewing ray lies almost in its plane — Z0 = num / den with a cancelling den,
// residuals of 10^4 px, Jacobian entries of 10^9 — carries a relative uncertainty of ~1e-10 in the reference's own double
// arithmetic; one such block shifts entries of b by up to 1e-7 of themselves.
#define IBA_JAC_CONTRACT _Pragma("clang fp contract(off)")
// the accumulation H += (w J_i) J_j of terms that are already formed may fuse its multiply-add: that changes the sums by parts in
// 1e-16 of a term, not by the cancellation the Jacobian chain is sensitive to (measured: per-entry agreement unchanged)
#define IBA_ACC_CONTRACT _Pragma("clang fp contract(fast)")
struct NAcc {
    double H[28], b[7], chi2, cost, nf2d, nfpl, nfpt, nres;
    __device__ __forceinline__ void count(int f2d, int fpl, int fpt, int rows) { nf2d += (double)f2d; nfpl += (double)fpl; nfpt += (double)fpt; nres += (double)rows; }
};
// the same sums with the four block / row counters as integers in three registers instead of four doubles in eight (iba_factor2_kernel holds its
// accumulators for a whole range of a candidate's list and needs the registers; a lane evaluates < 65536 blocks of a kind: factor_waves() sees to it)
struct NAccP {
    double H[28], b[7], chi2, cost;
    uint32_t c2d_pl, cpt, rows;
    __device__ __forceinline__ void count(int f2d, int fpl, int fpt, int nr) { c2d_pl += (uint32_t)f2d | ((uint32_t)fpl << 16); cpt += (uint32_t)fpt; rows += (uint32_t)nr; }
};

__device__ __forceinline__ double fdot3(double a0, double b0, double a1, double b1, double a2, double b2) { return __builtin_fma(a2, b2, __builtin_fma(a1, b1, a0 * b0)); }
__device__ __forceinline__ double fdot3c(double a0, double b0, double a1, double b1, double a2, double b2, double c) { return __builtin_fma(a2, b2, __builtin_fma(a1, b1, __builtin_fma(a0, b0, c))); }

__device__ __forceinline__ void huber_w(double a, double s, double& rho0, double& w) {
    const double bb = a * a;
    if (s > bb) { const double r = sqrt(s); rho0 = 2.0 * a * r - bb; w = fmax(2.2250738585072014e-308, a / r); }
    else { rho0 = s; w = 1.0; }
}
// H (upper, row-major i<=j) index
__device__ __forceinline__ int hidx(int i, int j) { return i * 7 - (i * (i - 1)) / 2 + (j - i); }

struct Cam4 { double fx, fy, cx, cy; };

// IBA_PlaneFactor core: calls slot(ru, rv, gu, gv, hu, hv) per matched covisible KF, where the two residual rows are
// (ru, rv) and their Jacobian rows are [gu * z6, hu], [gv * z6, hv] (z6 = dZ0/dx[0:6], last column d/ds).
// Inputs: R, t and their derivatives (dR[k] for k < 3, dt[k] for k < 6) — pointers, so that a kernel may keep them where it likes (scalar
// registers, LDS) —, the camera of the keypoint's keyframe, (Cxz, Cyz) = ((u0 - cx) / fx, (v0 - cy) / fy) (the host divides once per keypoint:
// DevProblem::kp_c; the same IEEE quotient a kernel would form), the match words of the keypoint (m0: slots 0..29, m1: slots 30..61, MANY
// only) and its row of the match table, and rel_of(slot, ts) -> the 12 doubles [R_i | t_i] of a covisible slot, ts = s t_i (IBACalib2.hpp:175).
// The keypoint's matches in its first kMatchPre covisible slots, loaded UNCONDITIONALLY (a slot without a match holds NaN and its bit is clear): the
// loads need neither the flag word nor each other — a kernel issues them with its other gathers instead of one dependent round trip per matched slot
// (r06; the slots beyond, if a keyframe has them, are still fetched as the loop reaches them)
constexpr int kMatchPre = 3;
struct MatchPre { float u0, v0, u1, v1, u2, v2; };   // (plain floats: a struct of float2 was kept in scratch memory by the selects below)
__device__ __forceinline__ MatchPre load_match_pre(const float2* mrow, size_t K, uint32_t n_slots) {
    const float2 z = make_float2(0.f, 0.f);
    const float2 a = 0u < n_slots ? mrow[0] : z, b = 1u < n_slots ? mrow[K] : z, c = 2u < n_slots ? mrow[2 * K] : z;
    return MatchPre{a.x, a.y, b.x, b.y, c.x, c.y};
}
template <bool MANY, class RelFn, class SlotFn>
__device__ __forceinline__ int plane_core(const double* R, const double* t, const double (*dR)[9], const double (*dt)[3], const Cam4& cam, double Cxz, double Cyz,
                                          const double* p0, const double* n0, double* z6, uint32_t m0, uint32_t m1, const float2* mrow, size_t K, const MatchPre mp, RelFn rel_of, SlotFn slot) {
    IBA_JAC_CONTRACT
    double p0c[3], n0c[3];
    for (int r = 0; r < 3; ++r) {
        p0c[r] = fdot3c(R[r * 3], p0[0], R[r * 3 + 1], p0[1], R[r * 3 + 2], p0[2], t[r]);
        n0c[r] = fdot3(R[r * 3], n0[0], R[r * 3 + 1], n0[1], R[r * 3 + 2], n0[2]);
    }
    const double num = fdot3(n0c[0], p0c[0], n0c[1], p0c[1], n0c[2], p0c[2]);
    const double den = __builtin_fma(Cyz, n0c[1], __builtin_fma(Cxz, n0c[0], n0c[2]));
    const double iden = 1.0 / den;
    const double Z0 = num * iden;   // (one division for the depth and its derivatives)
    // a ROLLED loop over the three rotation parameters: 12 derivative constants are live at a time, not 45 (a kernel that keeps them in LDS would
    // otherwise hold them all in vector registers beside its 41 sums); the result lands in its register through selects
    double zr0 = 0, zr1 = 0, zr2 = 0;
#pragma unroll 1
    for (int kk = 0; kk < 3; ++kk) {
        const double* A = dR[kk]; const double* tk = dt[kk];
        double dn[3], dq[3];
        for (int r = 0; r < 3; ++r) {
            dn[r] = fdot3(A[r * 3], n0[0], A[r * 3 + 1], n0[1], A[r * 3 + 2], n0[2]);
            dq[r] = fdot3c(A[r * 3], p0[0], A[r * 3 + 1], p0[1], A[r * 3 + 2], p0[2], tk[r]);
        }
        const double dnum = fdot3c(n0c[0], dq[0], n0c[1], dq[1], n0c[2], dq[2], fdot3(dn[0], p0c[0], dn[1], p0c[1], dn[2], p0c[2]));
        const double dden = __builtin_fma(Cyz, dn[1], __builtin_fma(Cxz, dn[0], dn[2]));
        const double z = __builtin_fma(-Z0, dden, dnum) * iden;
        zr0 = kk == 0 ? z : zr0; zr1 = kk == 1 ? z : zr1; zr2 = kk == 2 ? z : zr2;
    }
    z6[0] = zr0; z6[1] = zr1; z6[2] = zr2;
    for (int kk = 3; kk < 6; ++kk) z6[kk] = fdot3(n0c[0], dt[kk][0], n0c[1], dt[kk][1], n0c[2], dt[kk][2]) * iden;   // (the plane normal does not move with the translation)
    const double P0x = Cxz * Z0, P0y = Cyz * Z0, P0z = Z0;
    int nconv = 0;
    // only the covisible slots whose match bit is set, in slot order; the next match is in flight during the arithmetic of the current one.
    // One pass per flag word: the slots 0..29, then — frames with more covisible keyframes only — the slots 30..61
    const int n_words = (MANY && m1 != 0u) ? 2 : 1;
#pragma unroll 1
    for (int wi = 0; wi < n_words; ++wi) {   // ONE copy of the loop body for both words (two inlined copies cost the kernel 3 %: code size)
        uint32_t mask = (!MANY || wi == 0) ? m0 : m1;
        const uint32_t base = (!MANY || wi == 0) ? 0u : (uint32_t)kCovisWord;
        while (mask) {
            const uint32_t sl = base + (uint32_t)__ffs((int)mask) - 1u;
            mask &= mask - 1u;
            float2 m = make_float2(sl == 0u ? mp.u0 : (sl == 1u ? mp.u1 : mp.u2), sl == 0u ? mp.v0 : (sl == 1u ? mp.v1 : mp.v2));
            if (sl >= (uint32_t)kMatchPre) m = mrow[(size_t)sl * K];
            double ts[3];
            const double* rel = rel_of(sl, ts);
            const double P1x = fdot3c(rel[0], P0x, rel[1], P0y, rel[2], P0z, ts[0]);
            const double P1y = fdot3c(rel[4], P0x, rel[5], P0y, rel[6], P0z, ts[1]);
            const double P1z = fdot3c(rel[8], P0x, rel[9], P0y, rel[10], P0z, ts[2]);
            const double iz = 1.0 / P1z, xz = P1x * iz, yz = P1y * iz;   // (one reciprocal for the residual and its derivatives)
            const double ru = __builtin_fma(cam.fx, xz, cam.cx) - (double)m.x;
            const double rv = __builtin_fma(cam.fy, yz, cam.cy) - (double)m.y;
            const double ax = __builtin_fma(rel[1], Cyz, __builtin_fma(rel[0], Cxz, rel[2])), ay = __builtin_fma(rel[5], Cyz, __builtin_fma(rel[4], Cxz, rel[6])),
                         az = __builtin_fma(rel[9], Cyz, __builtin_fma(rel[8], Cxz, rel[10]));
            const double fxiz = cam.fx * iz, fyiz = cam.fy * iz;
            const double gu = fxiz * __builtin_fma(-xz, az, ax), gv = fyiz * __builtin_fma(-yz, az, ay);
            const double hu = fxiz * __builtin_fma(-xz, rel[11], rel[3]), hv = fyiz * __builtin_fma(-yz, rel[11], rel[7]);
            slot(ru, rv, gu, gv, hu, hv);
            ++nconv;
        }
    }
    return nconv;
}

// the sums of one IBA_PlaneFactor block into the lane's accumulators
template <bool MANY, class RelFn, class Acc>
__device__ __forceinline__ void plane_accum(const double* R, const double* t, const double (*dR)[9], const double (*dt)[3], const Cam4& cam, double Cxz, double Cyz, const double* p0, const double* n0,
                                            uint32_t m0, uint32_t m1, const float2* mrow, size_t K, const MatchPre mp, RelFn rel_of, double delta, Acc& A) {
    IBA_JAC_CONTRACT
    double z6[6];
    double ssq = 0, G = 0, GH = 0, HH = 0, Gr = 0, Hr = 0;
    const int nconv = plane_core<MANY>(R, t, dR, dt, cam, Cxz, Cyz, p0, n0, z6, m0, m1, mrow, K, mp, rel_of, [&](double ru, double rv, double gu, double gv, double hu, double hv) {
IBA_ACC_CONTRACT
        ssq += ru * ru + rv * rv;
        G += gu * gu + gv * gv; GH += gu * hu + gv * hv; HH += hu * hu + hv * hv;
        Gr += gu * ru + gv * rv; Hr += hu * ru + hv * rv;
    });
    if (nconv == 0) return;
    double rho0, w; huber_w(delta, ssq, rho0, w);
    A.cost += 0.5 * rho0; A.chi2 += ssq; A.count(1, 0, 0, 2 * nconv);
    {
        IBA_ACC_CONTRACT
        for (int i = 0; i < 6; ++i) {
            const double wz = w * z6[i], wzG = wz * G;
            for (int j = i; j < 6; ++j) A.H[hidx(i, j)] += wzG * z6[j];
            A.H[hidx(i, 6)] += wz * GH;
            A.b[i] += wz * Gr;
        }
        A.H[27] += w * HH; A.b[6] += w * Hr;
    }
}

// IBATestEdge core (IBACalib.hpp:40-58; factor_3d2d_kind = 1): the matched scan point p0 reprojected DIRECTLY into every covisible keyframe
// that matches the keypoint — p0c = R p0 + t (:49), p1c = R_i p0c + s t_i (:48, :50), (fx p1x / p1z + cx, fy p1y / p1z + cy) - (u1, v1) (:52-55).
// One 2-row edge per matched slot, in slot order: calls edge(ru, rv, Ju[7], Jv[7]). Analytic chain rule through the same expressions the
// reference's auto-diff differentiates: d p0c / dx_k = dt_k (+ dR_k p0 for k < 3), d p1c / dx_k = R_i d p0c / dx_k, d p1c / ds = t_i.
template <bool MANY, class RelFn, class EdgeFn>
__device__ __forceinline__ int edge_core(const double* R, const double* t, const double (*dR)[9], const double (*dt)[3], const Cam4& cam, const double* p0,
                                         uint32_t m0, uint32_t m1, const float2* mrow, size_t K, const MatchPre mp, RelFn rel_of, EdgeFn edge) {
    IBA_JAC_CONTRACT
    double p0c[3], dq[6][3];
    for (int r = 0; r < 3; ++r) {
        p0c[r] = fdot3c(R[r * 3], p0[0], R[r * 3 + 1], p0[1], R[r * 3 + 2], p0[2], t[r]);
        for (int kk = 0; kk < 3; ++kk) dq[kk][r] = fdot3c(dR[kk][r * 3], p0[0], dR[kk][r * 3 + 1], p0[1], dR[kk][r * 3 + 2], p0[2], dt[kk][r]);
        for (int kk = 3; kk < 6; ++kk) dq[kk][r] = dt[kk][r];
    }
    int nconv = 0;
    const int n_words = (MANY && m1 != 0u) ? 2 : 1;
#pragma unroll 1
    for (int wi = 0; wi < n_words; ++wi) {
        uint32_t mask = (!MANY || wi == 0) ? m0 : m1;
        const uint32_t base = (!MANY || wi == 0) ? 0u : (uint32_t)kCovisWord;
        while (mask) {
            const uint32_t sl = base + (uint32_t)__ffs((int)mask) - 1u;
            mask &= mask - 1u;
            float2 m = make_float2(sl == 0u ? mp.u0 : (sl == 1u ? mp.u1 : mp.u2), sl == 0u ? mp.v0 : (sl == 1u ? mp.v1 : mp.v2));
            if (sl >= (uint32_t)kMatchPre) m = mrow[(size_t)sl * K];
            double ts[3];
            const double* rel = rel_of(sl, ts);   // ts = t_i * _s (IBACalib.hpp:48)
            const double P1x = fdot3c(rel[0], p0c[0], rel[1], p0c[1], rel[2], p0c[2], ts[0]);
            const double P1y = fdot3c(rel[4], p0c[0], rel[5], p0c[1], rel[6], p0c[2], ts[1]);
            const double P1z = fdot3c(rel[8], p0c[0], rel[9], p0c[1], rel[10], p0c[2], ts[2]);
            const double iz = 1.0 / P1z, xz = P1x * iz, yz = P1y * iz;   // (one reciprocal: see plane_core)
            const double ru = __builtin_fma(cam.fx, xz, cam.cx) - (double)m.x;
            const double rv = __builtin_fma(cam.fy, yz, cam.cy) - (double)m.y;
            const double fxiz = cam.fx * iz, fyiz = cam.fy * iz;
            double Ju[7], Jv[7];
            for (int kk = 0; kk < 6; ++kk) {
                const double qx = fdot3(rel[0], dq[kk][0], rel[1], dq[kk][1], rel[2], dq[kk][2]);
                const double qy = fdot3(rel[4], dq[kk][0], rel[5], dq[kk][1], rel[6], dq[kk][2]);
                const double qz = fdot3(rel[8], dq[kk][0], rel[9], dq[kk][1], rel[10], dq[kk][2]);
                Ju[kk] = fxiz * __builtin_fma(-xz, qz, qx); Jv[kk] = fyiz * __builtin_fma(-yz, qz, qy);
            }
            Ju[6] = fxiz * __builtin_fma(-xz, rel[11], rel[3]); Jv[6] = fyiz * __builtin_fma(-yz, rel[11], rel[7]);
            edge(ru, rv, Ju, Jv);
            ++nconv;
        }
    }
    return nconv;
}
// every edge is a residual block of its own: its own Huber weight (g2o: RobustKernelHuber per edge; delta = robust_kernel_delta)
template <bool MANY, class RelFn, class Acc>
__device__ __forceinline__ void edge_accum(const double* R, const double* t, const double (*dR)[9], const double (*dt)[3], const Cam4& cam, const double* p0,
                                           uint32_t m0, uint32_t m1, const float2* mrow, size_t K, const MatchPre mp, RelFn rel_of, double delta, Acc& A) {
    IBA_JAC_CONTRACT
    edge_core<MANY>(R, t, dR, dt, cam, p0, m0, m1, mrow, K, mp, rel_of, [&](double ru, double rv, const double* Ju, const double* Jv) {
        const double ssq = ru * ru + rv * rv;
        double rho0, w; huber_w(delta, ssq, rho0, w);
        A.cost += 0.5 * rho0; A.chi2 += ssq; A.count(1, 0, 0, 2);
        {
            IBA_ACC_CONTRACT
            for (int i = 0; i < 7; ++i) {
                const double wu = w * Ju[i], wv = w * Jv[i];
                for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wu * Ju[j] + wv * Jv[j];
                A.b[i] += wu * ru + wv * rv;
            }
        }
    });
}

// M = Rlc (s m) + tlc and dM/dx for the 3d-3d factors (IBACalib2.hpp:570-584, 611-625); Tcw: the keyframe's pose (3 x 4, row-major)
__device__ __forceinline__ void p2x_core(const double* Rlc, const double* tlc, const double (*dRlc)[9], const double (*dtlc)[3], double s, const double* Tcw, const float4 mp, double* M, double dM[7][3]) {
    IBA_JAC_CONTRACT
    const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
    const double m[3] = {fdot3c(Tcw[0], w0, Tcw[1], w1, Tcw[2], w2, Tcw[3]), fdot3c(Tcw[4], w0, Tcw[5], w1, Tcw[6], w2, Tcw[7]), fdot3c(Tcw[8], w0, Tcw[9], w1, Tcw[10], w2, Tcw[11])};
    const double sm[3] = {m[0] * s, m[1] * s, m[2] * s};
    for (int r = 0; r < 3; ++r) {
        M[r] = fdot3c(Rlc[r * 3], sm[0], Rlc[r * 3 + 1], sm[1], Rlc[r * 3 + 2], sm[2], tlc[r]);
        for (int kk = 0; kk < 3; ++kk) dM[kk][r] = fdot3c(dRlc[kk][r * 3], sm[0], dRlc[kk][r * 3 + 1], sm[1], dRlc[kk][r * 3 + 2], sm[2], dtlc[kk][r]);
        for (int kk = 3; kk < 6; ++kk) dM[kk][r] = dtlc[kk][r];
        dM[6][r] = fdot3(Rlc[r * 3], m[0], Rlc[r * 3 + 1], m[1], Rlc[r * 3 + 2], m[2]);
    }
}

__device__ __forceinline__ void p2x_accum(const double* Rlc, const double* tlc, const double (*dRlc)[9], const double (*dtlc)[3], double s, const double* Tcw, double delta3, const float4 mp, const double* Q, const double* n, bool is_plane, NAcc& A) {
    IBA_JAC_CONTRACT
    double M[3], dM[7][3];
    p2x_core(Rlc, tlc, dRlc, dtlc, s, Tcw, mp, M, dM);
    const double e[3] = {M[0] - Q[0], M[1] - Q[1], M[2] - Q[2]};
    if (is_plane) {
        const double r = fdot3(e[0], n[0], e[1], n[1], e[2], n[2]);
        double J[7];
        for (int kk = 0; kk < 7; ++kk) J[kk] = fdot3(dM[kk][0], n[0], dM[kk][1], n[1], dM[kk][2], n[2]);
        double rho0, w; huber_w(delta3, r * r, rho0, w);
        A.cost += 0.5 * rho0; A.chi2 += r * r; A.count(0, 1, 0, 1);
        { IBA_ACC_CONTRACT for (int i = 0; i < 7; ++i) { const double wj = w * J[i]; for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wj * J[j]; A.b[i] += wj * r; } }
    } else {
        const double ssq = fdot3(e[0], e[0], e[1], e[1], e[2], e[2]);
        double rho0, w; huber_w(delta3, ssq, rho0, w);
        A.cost += 0.5 * rho0; A.chi2 += ssq; A.count(0, 0, 1, 3);
        { IBA_ACC_CONTRACT for (int r = 0; r < 3; ++r)
            for (int i = 0; i < 7; ++i) { const double wj = w * dM[i][r]; for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wj * dM[j][r]; A.b[i] += wj * e[r]; } }
    }
}

// The 3d-3d blocks again for the factor kernels, with the derivative constants consumed in ROLLED loops (12 resp. 15 live at a time, see plane_core):
// the same fused chains as p2x_core / p2x_accum, term for term — the same bits.
__device__ __forceinline__ void p2x_head(const double* Rlc, const double* tlc, double s, const double* Tcw, const float4 mp, const double* Q, double* m, double* sm, double* e) {
    IBA_JAC_CONTRACT
    const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
    m[0] = fdot3c(Tcw[0], w0, Tcw[1], w1, Tcw[2], w2, Tcw[3]); m[1] = fdot3c(Tcw[4], w0, Tcw[5], w1, Tcw[6], w2, Tcw[7]); m[2] = fdot3c(Tcw[8], w0, Tcw[9], w1, Tcw[10], w2, Tcw[11]);
    for (int r = 0; r < 3; ++r) sm[r] = m[r] * s;
    for (int r = 0; r < 3; ++r) e[r] = fdot3c(Rlc[r * 3], sm[0], Rlc[r * 3 + 1], sm[1], Rlc[r * 3 + 2], sm[2], tlc[r]) - Q[r];
}
// Point2Plane_Factor (IBACalib2.hpp:611-625): one row J = n^T dM/dx
template <class Acc>
__device__ __forceinline__ void p2pl_accum(const double* Rlc, const double* tlc, const double (*dRlc)[9], const double (*dtlc)[3], double s, const double* Tcw, double delta3, const float4 mp, const double* Q, const double* n, Acc& A) {
    IBA_JAC_CONTRACT
    double m[3], sm[3], e[3];
    p2x_head(Rlc, tlc, s, Tcw, mp, Q, m, sm, e);
    const double r = fdot3(e[0], n[0], e[1], n[1], e[2], n[2]);
    double J[7];
    double j0 = 0, j1 = 0, j2 = 0;
#pragma unroll 1
    for (int kk = 0; kk < 3; ++kk) {
        const double* D = dRlc[kk]; const double* tk = dtlc[kk];
        double dm[3];
        for (int rr = 0; rr < 3; ++rr) dm[rr] = fdot3c(D[rr * 3], sm[0], D[rr * 3 + 1], sm[1], D[rr * 3 + 2], sm[2], tk[rr]);
        const double jj = fdot3(dm[0], n[0], dm[1], n[1], dm[2], n[2]);
        j0 = kk == 0 ? jj : j0; j1 = kk == 1 ? jj : j1; j2 = kk == 2 ? jj : j2;
    }
    J[0] = j0; J[1] = j1; J[2] = j2;
    for (int kk = 3; kk < 6; ++kk) J[kk] = fdot3(dtlc[kk][0], n[0], dtlc[kk][1], n[1], dtlc[kk][2], n[2]);
    {
        double d6[3];
        for (int rr = 0; rr < 3; ++rr) d6[rr] = fdot3(Rlc[rr * 3], m[0], Rlc[rr * 3 + 1], m[1], Rlc[rr * 3 + 2], m[2]);
        J[6] = fdot3(d6[0], n[0], d6[1], n[1], d6[2], n[2]);
    }
    double rho0, w; huber_w(delta3, r * r, rho0, w);
    A.cost += 0.5 * rho0; A.chi2 += r * r; A.count(0, 1, 0, 1);
    { IBA_ACC_CONTRACT for (int i = 0; i < 7; ++i) { const double wj = w * J[i]; for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wj * J[j]; A.b[i] += wj * r; } }
}
// Point2Point_Factor (IBACalib2.hpp:570-584): three rows, row rr = d M_rr / dx
template <class Acc>
__device__ __forceinline__ void p2pt_accum(const double* Rlc, const double* tlc, const double (*dRlc)[9], const double (*dtlc)[3], double s, const double* Tcw, double delta3, const float4 mp, const double* Q, Acc& A) {
    IBA_JAC_CONTRACT
    double m[3], sm[3], e[3];
    p2x_head(Rlc, tlc, s, Tcw, mp, Q, m, sm, e);
    const double ssq = fdot3(e[0], e[0], e[1], e[1], e[2], e[2]);
    double rho0, w; huber_w(delta3, ssq, rho0, w);
    A.cost += 0.5 * rho0; A.chi2 += ssq; A.count(0, 0, 1, 3);
#pragma unroll 1
    for (int rr = 0; rr < 3; ++rr) {
        double v[7];
        for (int kk = 0; kk < 3; ++kk) v[kk] = fdot3c(dRlc[kk][rr * 3], sm[0], dRlc[kk][rr * 3 + 1], sm[1], dRlc[kk][rr * 3 + 2], sm[2], dtlc[kk][rr]);
        for (int kk = 3; kk < 6; ++kk) v[kk] = dtlc[kk][rr];
        v[6] = fdot3(Rlc[rr * 3], m[0], Rlc[rr * 3 + 1], m[1], Rlc[rr * 3 + 2], m[2]);
        const double er = rr == 0 ? e[0] : (rr == 1 ? e[1] : e[2]);
        { IBA_ACC_CONTRACT for (int i = 0; i < 7; ++i) { const double wj = w * v[i]; for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wj * v[j]; A.b[i] += wj * er; } }
    }
}

// ---- the same cores behind the signatures of rounds 1-5 (a keyframe header, a Cand and the keypoint's pixel): iba_residual_kernel,
// iba_factor_mfma_kernel and the one-wave-per-(keyframe, candidate) iba_factor_kernel evaluate the SAME arithmetic as iba_factor2_kernel ----
template <bool MANY = true, class SlotFn>
__device__ __forceinline__ int plane_factor_core(const Cand& c, const FrameHdr& h, const DevProblem& dp, uint32_t k, uint32_t K,
                                                 double u0, double v0, const double* p0, const double* n0, double* z6, SlotFn slot, const double* rel_lds = nullptr) {
    const Cam4 cam{h.fx, h.fy, h.cx, h.cy};
    const double Cxz = (u0 - h.cx) / h.fx, Cyz = (v0 - h.cy) / h.fy;
    const uint32_t m0 = dp.kp_fl[h.kp_base + k] >> 2, m1 = (MANY && h.n_slots > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + k] : 0u;
    const double s = c.s;
    const MatchPre mp = load_match_pre(dp.match_uv + h.match_base + k, (size_t)K, h.n_slots);
    return plane_core<MANY>(c.R, c.t, c.dR, c.dt, cam, Cxz, Cyz, p0, n0, z6, m0, m1, dp.match_uv + h.match_base + k, (size_t)K, mp,
                            [&](uint32_t sl, double* ts) { const double* rel = rel_lds ? rel_lds + sl * 12u : dp.slots[h.slot_base + sl].rel; ts[0] = rel[3] * s; ts[1] = rel[7] * s; ts[2] = rel[11] * s; return rel; }, slot);
}
template <bool MANY>
__device__ inline void plane_factor_accum(const Cand& c, const FrameHdr& h, const DevProblem& dp, const DevParams& prm, uint32_t k, uint32_t K,
                                          double u0, double v0, const double* p0, const double* n0, NAcc& A, const double* rel_lds = nullptr) {
    const Cam4 cam{h.fx, h.fy, h.cx, h.cy};
    const double Cxz = (u0 - h.cx) / h.fx, Cyz = (v0 - h.cy) / h.fy;
    const uint32_t m0 = dp.kp_fl[h.kp_base + k] >> 2, m1 = (MANY && h.n_slots > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + k] : 0u;
    const double s = c.s;
    const MatchPre mp = load_match_pre(dp.match_uv + h.match_base + k, (size_t)K, h.n_slots);
    plane_accum<MANY>(c.R, c.t, c.dR, c.dt, cam, Cxz, Cyz, p0, n0, m0, m1, dp.match_uv + h.match_base + k, (size_t)K, mp,
                      [&](uint32_t sl, double* ts) { const double* rel = rel_lds ? rel_lds + sl * 12u : dp.slots[h.slot_base + sl].rel; ts[0] = rel[3] * s; ts[1] = rel[7] * s; ts[2] = rel[11] * s; return rel; }, prm.robust_kernel_delta, A);
}
template <bool MANY = true, class EdgeFn>
__device__ __forceinline__ int test_edge_core(const Cand& c, const FrameHdr& h, const DevProblem& dp, uint32_t k, uint32_t K, const double* p0, EdgeFn edge, const double* rel_lds = nullptr) {
    const Cam4 cam{h.fx, h.fy, h.cx, h.cy};
    const uint32_t m0 = dp.kp_fl[h.kp_base + k] >> 2, m1 = (MANY && h.n_slots > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + k] : 0u;
    const double s = c.s;
    const MatchPre mp = load_match_pre(dp.match_uv + h.match_base + k, (size_t)K, h.n_slots);
    return edge_core<MANY>(c.R, c.t, c.dR, c.dt, cam, p0, m0, m1, dp.match_uv + h.match_base + k, (size_t)K, mp,
                           [&](uint32_t sl, double* ts) { const double* rel = rel_lds ? rel_lds + sl * 12u : dp.slots[h.slot_base + sl].rel; ts[0] = rel[3] * s; ts[1] = rel[7] * s; ts[2] = rel[11] * s; return rel; }, edge);
}
template <bool MANY>
__device__ inline void test_edge_accum(const Cand& c, const FrameHdr& h, const DevProblem& dp, const DevParams& prm, uint32_t k, uint32_t K, const double* p0, NAcc& A, const double* rel_lds = nullptr) {
    const Cam4 cam{h.fx, h.fy, h.cx, h.cy};
    const uint32_t m0 = dp.kp_fl[h.kp_base + k] >> 2, m1 = (MANY && h.n_slots > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + k] : 0u;
    const double s = c.s;
    const MatchPre mp = load_match_pre(dp.match_uv + h.match_base + k, (size_t)K, h.n_slots);
    edge_accum<MANY>(c.R, c.t, c.dR, c.dt, cam, p0, m0, m1, dp.match_uv + h.match_base + k, (size_t)K, mp,
                     [&](uint32_t sl, double* ts) { const double* rel = rel_lds ? rel_lds + sl * 12u : dp.slots[h.slot_base + sl].rel; ts[0] = rel[3] * s; ts[1] = rel[7] * s; ts[2] = rel[11] * s; return rel; }, prm.robust_kernel_delta, A);
}
__device__ __forceinline__ void p2x_core(const Cand& c, const FrameHdr& h, const float4 mp, double* M, double dM[7][3]) { p2x_core(c.Rlc, c.tlc, c.dRlc, c.dtlc, c.s, h.Tcw, mp, M, dM); }
__device__ inline void p2x_factor_accum(const Cand& c, const FrameHdr& h, const DevParams& prm, const float4 mp, const double* Q, const double* n, bool is_plane, NAcc& A) {
    p2x_accum(c.Rlc, c.tlc, c.dRlc, c.dtlc, c.s, h.Tcw, prm.robust_kernel_3ddelta, mp, Q, n, is_plane, A);
}


// (point, memoised local-plane normal) records of every scan point: one thread each
__global__ __launch_bounds__(256) void iba_scanrec_kernel(const float4* __restrict__ pts4, const PlaneRec* __restrict__ planes_local, ScanRec* __restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 p = pts4[i];
    const PlaneRec r = planes_local[i];
    ScanRec o;
    o.x = p.x; o.y = p.y; o.z = p.z; o.pad0 = 0u; o.nx = r.nx; o.ny = r.ny; o.nz = r.nz; o.pad1[0] = o.pad1[1] = o.pad1[2] = 0.0;
    out[i] = o;
}

#ifndef IBA_FACTOR_WAVES
#define IBA_FACTOR_WAVES 2
#endif
#ifndef IBA_FACTOR_MFMA_WAVES
#define IBA_FACTOR_MFMA_WAVES 3   /* waves per SIMD the MFMA factor kernel is compiled for (<= 168 VGPRs: no spills; 4 spills 132 B per lane and is slower) */
#endif
#ifndef IBA_FACTOR_THREADS
#define IBA_FACTOR_THREADS 64   /* ~264 blocks per (candidate, frame): 64-thread blocks waste the least of their last pass (256: 0.34 ms, 64: 0.24 ms) */
#endif
constexpr int kFactorThreads = IBA_FACTOR_THREADS;
// grid: (n_frames, B), kFactorThreads threads (one wave). Works through the dense residual-block list the association pass left for this
// (candidate, frame): every lane owns a keypoint that has at least one block. list row = (per_cand ? b : 0).
// record (b, rec_base + frame) of `partials` receives this block's sums.
template <bool MANY, bool P2PIX = false, bool REC = false>   // P2PIX: the 3d-2d blocks are IBATestEdge edges (factor_3d2d_kind = 1); REC: the gathers come from the one-line records (DevProblem::kp_rec, scan_rec)
__global__ __launch_bounds__(kFactorThreads) __attribute__((amdgpu_waves_per_eu(IBA_FACTOR_WAVES, IBA_FACTOR_WAVES))) void iba_factor_kernel(DevProblem dp, DevParams prm, const Cand* __restrict__ cands, const uint4* __restrict__ flist,
                                                                    const uint32_t* __restrict__ fcount, int flist_stride, int per_cand,
                                                                    double* __restrict__ partials, int nrec, int rec_base, int B) {
    __shared__ double s_part[kFactorThreads / 64][48];
    __shared__ double s_tr[kFactorThreads / 64][21][65];   // [sum][lane], rows padded against bank conflicts
    extern __shared__ __align__(16) double s_rel[];        // relative poses of the frame's covisible slots: 12 doubles each, sized by the launch for the handle's largest slot count (r04: 62 slots as a static array cost every block 3 KB and the kernel 2 %)
    // block -> (keyframe, candidate). B = 0: a (keyframe, candidate) grid — block i runs on XCD i % 8, so with a keyframe count that is a multiple
    // of 8 a keyframe's candidates share one XCD's L2 and a candidate's keyframes follow each other (its list rows are contiguous). B > 0 (r05, any
    // other keyframe count: a rank's shard of 25): the association kernels' mapping — keyframe f on XCD f % 8 by construction, its candidates side by
    // side (a 25-keyframe shard, 64 candidates: 0.146 -> 0.136 ms per call; at 200 keyframes the plain grid is 4 us better and stays).
    int f = (int)blockIdx.x, b = (int)blockIdx.y;
    if (B > 0) {
        const int per_xcd = (dp.n_frames + 7) / 8;
        f = (int)(blockIdx.x & 7u) + 8 * (int)((blockIdx.x >> 3) / (uint32_t)B); b = (int)((blockIdx.x >> 3) % (uint32_t)B);
        if (f >= dp.n_frames || (int)((blockIdx.x >> 3) / (uint32_t)B) >= per_xcd) return;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameHdr& h = dp.frames[f];
    const Cand& c = cands[b];
    for (uint32_t i = tid; i < h.n_slots * 12u; i += kFactorThreads) s_rel[i] = dp.slots[h.slot_base + i / 12u].rel[i % 12u];
    __syncthreads();
    const size_t row = (size_t)(per_cand ? b : 0) * dp.n_frames + f;
    const uint4* fl = flist + row * (size_t)flist_stride;
    const uint32_t n = fcount[row];
    NAcc A;
    for (int i = 0; i < 28; ++i) A.H[i] = 0;
    for (int i = 0; i < 7; ++i) A.b[i] = 0;
    A.chi2 = A.cost = A.nf2d = A.nfpl = A.nfpt = A.nres = 0;
    const float4* p4 = dp.pts4 + h.pt_base;
    const PlaneRec* planes = prm.plane_cache ? dp.plane_local + h.pt_base
                                             : dp.scratch_local + (size_t)(per_cand ? dp.scratch_slot_base + b : 0) * (size_t)dp.n_pt_total + h.pt_base;
    // A list entry carries up to two residual blocks — the IBA_PlaneFactor of its keypoint (.y) and the 3d-3d factor of its
    // MapPoint (.z) — and about half of the entries have each: a lane per ENTRY left half of the wave idle in either body.
    // The entries are therefore split into two dense queues as they are read (64 at a time, one ballot each), and each body
    // runs on full waves of its own kind: 64 plane factors, then 64 3d-3d factors, whenever a queue holds that many, the
    // remainders at the end. Which lane adds which block is fixed by the list alone: the sums stay bitwise reproducible.
    constexpr uint32_t kQ = 128u;   // ring capacity: at most 63 waiting + 64 new
    __shared__ uint2 s_qa[kFactorThreads / 64][kQ], s_qb[kFactorThreads / 64][kQ];   // (keypoint, scan point) of a plane factor / (keypoint, point | kind) of a 3d-3d factor
    uint2* qa = s_qa[wave]; uint2* qb = s_qb[wave];
    uint32_t ha = 0u, ta = 0u, hb = 0u, tb = 0u;   // ring heads / tails (wave-uniform)
    // Round 6: a body is two halves — its GATHERS (unconditional: a lane without a block reads element 0 of every table, so that no load is merged
    // with a default behind a branch and waited for on the spot) and its ARITHMETIC; the gathers of the plane-factor batch and of the 3d-3d batch
    // that are due go out together. The hand-over between the lanes of the one wave is a wavefront-scope fence: the LDS executes a wave's
    // instructions in order, and a workgroup-scope release would wait for every load in flight (vmcnt(0)) — the next round's entries among them.
    struct PIn { bool on; uint32_t k, m0, m1; float px, py, pz; double n0x, n0y, n0z, u0, v0; MatchPre mp; };
    struct QIn { bool on, pl; float qx, qy, qz, mx, my, mz; double nx, ny, nz; };
    const float2* mtab = dp.match_uv + h.match_base;
    const uint32_t n_sl = h.n_slots, Kf = h.K;
    auto plane_load = [&](uint32_t cnt) -> PIn {
        PIn in;
        in.on = (uint32_t)lane < cnt;
        uint2 q = make_uint2(0u, 0u);
        if (in.on) q = qa[(ha + (uint32_t)lane) & (kQ - 1u)];
        in.k = q.x;
        if (REC) {   // one line of the keypoint's record, one of the scan point's
            const KpRec* kr = dp.kp_rec + h.kp_base + q.x;
            const ScanRec* sr = dp.scan_rec + h.pt_base + q.y;
            const double2 cz = *(const double2*)&kr->cxz;
            const float4 m01 = *(const float4*)&kr->m0u;
            const uint4 m2f = *(const uint4*)&kr->m2u;
            const float4 pt = *(const float4*)&sr->x;
            const double2 nxy = *(const double2*)&sr->nx;
            const double nz = sr->nz;
            in.mp = MatchPre{m01.x, m01.y, m01.z, m01.w, __uint_as_float(m2f.x), __uint_as_float(m2f.y)};
            in.px = pt.x; in.py = pt.y; in.pz = pt.z;
            in.m0 = m2f.z >> 2; in.m1 = (MANY && n_sl > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + q.x] : 0u;
            in.n0x = nxy.x; in.n0y = nxy.y; in.n0z = nz; in.u0 = cz.x; in.v0 = cz.y;   // (u0, v0 carry Cxz, Cyz here)
            ha += min(cnt, 64u);
            return in;
        }
        const bool has_m = n_sl > 0u;
        {
            const float2* mrow = mtab + (has_m ? q.x : 0u);
            const size_t s1 = has_m ? (size_t)min(1u, n_sl - 1u) * Kf : 0, s2 = has_m ? (size_t)min(2u, n_sl - 1u) * Kf : 0;
            const float2 a = has_m ? mrow[0] : dp.match_uv[0], bq = has_m ? mrow[s1] : dp.match_uv[0], cq = has_m ? mrow[s2] : dp.match_uv[0];
            in.mp = MatchPre{a.x, a.y, bq.x, bq.y, cq.x, cq.y};
        }
        const float4 pt = p4[q.y];
        in.px = pt.x; in.py = pt.y; in.pz = pt.z;
        in.m0 = dp.kp_fl[h.kp_base + q.x] >> 2; in.m1 = (MANY && n_sl > (uint32_t)kCovisWord) ? dp.kp_fl2[h.kp_base + q.x] : 0u;
        if (!P2PIX) {
            const PlaneRec& rec = planes[q.y];
            in.n0x = rec.nx; in.n0y = rec.ny; in.n0z = rec.nz;
            const float2 uv = dp.kp_uv[h.kp_base + q.x];
            in.u0 = (double)uv.x; in.v0 = (double)uv.y;
        } else { in.n0x = in.n0y = in.n0z = in.u0 = in.v0 = 0.0; }
        ha += min(cnt, 64u);
        return in;
    };
    const double cs = c.s;
    auto rel_of = [&](uint32_t sl, double* ts) { const double* rel = s_rel + sl * 12u; ts[0] = rel[3] * cs; ts[1] = rel[7] * cs; ts[2] = rel[11] * cs; return (const double*)rel; };
    const Cam4 cam{h.fx, h.fy, h.cx, h.cy};
    auto plane_compute = [&](const PIn& in) {
        if (in.on) {
            const double p0[3] = {(double)in.px, (double)in.py, (double)in.pz};
            if (P2PIX) edge_accum<MANY>(c.R, c.t, c.dR, c.dt, cam, p0, in.m0, in.m1, mtab + in.k, (size_t)Kf, in.mp, rel_of, prm.robust_kernel_delta, A);
            else {
                const double n0[3] = {in.n0x, in.n0y, in.n0z};
                const double Cxz = REC ? in.u0 : (in.u0 - h.cx) / h.fx, Cyz = REC ? in.v0 : (in.v0 - h.cy) / h.fy;   // (the record holds the quotient the host formed: the same IEEE division)
                plane_accum<MANY>(c.R, c.t, c.dR, c.dt, cam, Cxz, Cyz, p0, n0, in.m0, in.m1, mtab + in.k, (size_t)Kf, in.mp, rel_of, prm.robust_kernel_delta, A);
            }
        }
    };
    auto p2x_load = [&](uint32_t cnt) -> QIn {
        QIn in;
        in.on = (uint32_t)lane < cnt;
        uint2 q = make_uint2(0u, 0u);
        if (in.on) q = qb[(hb + (uint32_t)lane) & (kQ - 1u)];
        const uint32_t pos3 = q.y & 0x7FFFFFFFu;
        in.pl = (q.y >> 31) != 0u;
        if (REC) {
            const KpRec* kr = dp.kp_rec + h.kp_base + q.x;
            const ScanRec* sr = dp.scan_rec + h.pt_base + pos3;
            const float4 mp3 = *(const float4*)&kr->mpx, pt3 = *(const float4*)&sr->x;
            const double2 nxy = *(const double2*)&sr->nx;
            const double nz = sr->nz;
            in.qx = pt3.x; in.qy = pt3.y; in.qz = pt3.z; in.mx = mp3.x; in.my = mp3.y; in.mz = mp3.z; in.nx = nxy.x; in.ny = nxy.y; in.nz = nz;
            hb += min(cnt, 64u);
            return in;
        }
        const float4 pt3 = p4[pos3], mp3 = dp.kp_mp[h.kp_base + q.x];
        const PlaneRec& r3 = planes[pos3];
        in.qx = pt3.x; in.qy = pt3.y; in.qz = pt3.z; in.mx = mp3.x; in.my = mp3.y; in.mz = mp3.z; in.nx = r3.nx; in.ny = r3.ny; in.nz = r3.nz;
        hb += min(cnt, 64u);
        return in;
    };
    auto p2x_compute = [&](const QIn& in) {
        if (in.on) {
            const double Q[3] = {(double)in.qx, (double)in.qy, (double)in.qz}, nn[3] = {in.nx, in.ny, in.nz};
            const float4 mp3 = make_float4(in.mx, in.my, in.mz, 0.f);
            if (in.pl) p2pl_accum(c.Rlc, c.tlc, c.dRlc, c.dtlc, cs, h.Tcw, prm.robust_kernel_3ddelta, mp3, Q, nn, A);
            else p2pt_accum(c.Rlc, c.tlc, c.dRlc, c.dtlc, cs, h.Tcw, prm.robust_kernel_3ddelta, mp3, Q, A);
        }
    };
#if IBA_FACTOR_WAVES >= 3
    auto both = [&](uint32_t ca, uint32_t cb) { { const QIn ib = p2x_load(cb); p2x_compute(ib); } { const PIn ia = plane_load(ca); plane_compute(ia); } };   // (three waves per SIMD: the two bodies' inputs are not held at once)
#else
    auto both = [&](uint32_t ca, uint32_t cb) { const PIn ia = plane_load(ca); const QIn ib = p2x_load(cb); p2x_compute(ib); plane_compute(ia); };
#endif
    auto lds_order = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); };
    // (three of an entry's four words, loaded unconditionally — lanes past the end read the last entry again: see iba_factor2_kernel.hpp)
    auto entry_at = [&](uint32_t i) -> uint4 {
        typedef uint32_t u3v __attribute__((ext_vector_type(3)));
        const u3v v = *(const u3v*)(fl + min(i, n - 1u));
        return make_uint4(v.x, v.y, v.z, 0u);
    };
    if (n > 0u) {
    uint4 e_n = entry_at((uint32_t)tid);
    for (uint32_t i0 = 0; i0 < n; i0 += kFactorThreads) {
        const uint32_t i = i0 + (uint32_t)tid;
        const uint4 e = e_n;
        e_n = entry_at(i + kFactorThreads);   // the next entries are in flight during this round's arithmetic
        const bool hp = i < n && e.y != kNone, h3 = i < n && e.z != kNone;
        const unsigned long long bp = __ballot(hp), b3 = __ballot(h3), lt = (1ull << lane) - 1ull;
        if (hp) qa[(ta + (uint32_t)__popcll(bp & lt)) & (kQ - 1u)] = make_uint2(e.x, e.y);
        if (h3) qb[(tb + (uint32_t)__popcll(b3 & lt)) & (kQ - 1u)] = make_uint2(e.x, e.z);
        ta += (uint32_t)__popcll(bp); tb += (uint32_t)__popcll(b3);
        lds_order();
        if (ta - ha >= 64u || tb - hb >= 64u) both(ta - ha >= 64u ? 64u : 0u, tb - hb >= 64u ? 64u : 0u);   // (wave-uniform)
        lds_order();   // the slots just read may be rewritten by the next round
    }
    }
    if (ta - ha || tb - hb) both(ta - ha, tb - hb);
    // fixed-order reduction through LDS: every lane parks its 41 sums (two halves of <= 21 through an 11 KB transposing
    // buffer), then lane v adds the 64 lanes' values of sum v in lane order. (The DPP butterfly this replaces cost 12 moves
    // and 6 adds per 64-bit sum: 1.5 k instructions per block, a quarter of the kernel.)
    double* v = (double*)&A;   // 41 contiguous doubles
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int base = half * 21, cnt = half ? 20 : 21;
#pragma unroll
        for (int q = 0; q < 21; ++q) if (q < cnt) s_tr[wave][q][lane] = v[base + q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < cnt) {
            double x = 0;
#pragma unroll 16
            for (int j = 0; j < 64; ++j) x += s_tr[wave][lane][j];
            s_part[wave][base + lane] = x;
        }
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    if (tid < kPartialStride) {
        const int i = tid;
        double out = 0;
        int src = -1;   // NAcc slot -> partial slot
        if (i >= P_H0 && i < P_H0 + 28) src = i - P_H0;
        else if (i >= P_B0 && i < P_B0 + 7) src = 28 + (i - P_B0);
        else if (i == P_CHI2) src = 35; else if (i == P_COST) src = 36; else if (i == P_NF_3D2D) src = 37;
        else if (i == P_NF_P2PL) src = 38; else if (i == P_NF_P2PT) src = 39; else if (i == P_NRES) src = 40;
        if (src >= 0) { out = s_part[0][src]; for (int w = 1; w < kFactorThreads / 64; ++w) out += s_part[w][src]; }
        partials[((size_t)b * nrec + rec_base + f) * kPartialStride + i] = out;
    }
}

// ---- the same sums on the matrix cores: H = sum_rows u v^T on v_mfma_f64_16x16x4_f64 (IBA_FACTOR_MFMA=1; NOT the default) ----
// Measured on MI355X (C2 shape, 64 candidates, profiles/r02_factor_mfma.md): 0.22 ms at 3 waves per SIMD against 0.16 ms for
// the all-VALU kernel above. gfx950 runs v_mfma_f64_16x16x4_f64 at 64 cycles = 16 FMA / clock / SIMD, the rate of v_fma_f64
// itself, and a wave's VALU stream beside it slows to half (tools/ubench/mfma_f64.hip); only 2 x 8 x 8 of the 16 x 16 tile are
// useful here, and the Jacobians that produce the rows (95 % of the kernel's VALU work) stay on the VALU. The kernel is kept
// as the measured answer to "MFMA for the J^T J block" and is covered by the parity tests.
// Every residual block is a few rank-1 terms u v^T of the 8 x 8 matrix [H | b] (row / column 7 = the residual column):
//   IBA_PlaneFactor  J = [g z6^T | h] (2 NConv rows, g, h in R^{2 NConv}):  u1 = [z6, 0, 0], v1 = w [G z6, GH, Gr];
//                    u2 = e6, v2 = w [GH z6, HH, Hr]   with G = g.g, GH = g.h, HH = h.h, Gr = g.r, Hr = h.r
//   Point2Plane      u = [J, 0], v = w [J, r];          Point2Point: three such rows (the columns of dM, e)
// A lane computes its block's terms with the VALU exactly as iba_factor_kernel does, parks them in LDS (z6 and five scalars
// for a plane factor, up to three 8-double rows and the weight for a 3d-3d block), and the wave then feeds them to the MFMA
// four k-slots x two 8-column halves at a time: lane l supplies A[m = l & 15][k = l >> 4] and B[k][n = l & 15]; columns
// 0..7 and 8..15 carry DIFFERENT rows, so the two diagonal 8 x 8 blocks of the 16 x 16 accumulator are two partial sums of
// [H | b] (the off-diagonal blocks are ignored) and one instruction retires eight rows. The 36 sums live in 8 accumulator
// registers instead of 41 VGPR pairs; the scalars (cost, chi2, counts) stay on the VALU.
typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int kFactorPl = 12;    // doubles parked per lane: z6[6], wG, wGH, wGr, wHH, wHr of its plane factor, weight of its 3d-3d block
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(IBA_FACTOR_MFMA_WAVES, IBA_FACTOR_MFMA_WAVES))) void iba_factor_mfma_kernel(
    DevProblem dp, DevParams prm, const Cand* __restrict__ cands, const uint4* __restrict__ flist, const uint32_t* __restrict__ fcount, int flist_stride, int per_cand,
    double* __restrict__ partials, int nrec, int rec_base) {
    __shared__ __align__(16) double s_pl[64][kFactorPl];   // 6 KB + 4 KB: sixteen one-wave blocks per CU
    __shared__ __align__(16) double s_r3[64][8];           // one row [u (7) | residual] of every lane's 3d-3d block at a time
    double* s_out = &s_pl[0][0];                           // the finished record (after the last pass)
    const int f = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x;
    const FrameHdr& h = dp.frames[f];
    const Cand& c = cands[b];
    const size_t row = (size_t)(per_cand ? b : 0) * dp.n_frames + f;
    const uint4* fl = flist + row * (size_t)flist_stride;
    const uint32_t n = fcount[row];
    const float4* p4 = dp.pts4 + h.pt_base;
    const PlaneRec* planes = prm.plane_cache ? dp.plane_local + h.pt_base
                                             : dp.scratch_local + (size_t)(per_cand ? dp.scratch_slot_base + b : 0) * (size_t)dp.n_pt_total + h.pt_base;
    d4_t acc = {0.0, 0.0, 0.0, 0.0};
    double cost = 0, chi2 = 0, nf2d = 0, nfpl = 0, nfpt = 0, nres = 0, h66 = 0, b6 = 0;
    const int kslot = lane >> 4, half = (lane >> 3) & 1, ci = lane & 7;   // this lane's place in the MFMA operands
    auto lds_sync = [&]() {   // LDS hand-over between the lanes of the one wave of this block
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + (uint32_t)lane;
        uint4 e = make_uint4(0u, kNone, kNone, 0u);
        if (i < n) e = fl[i];
        const uint32_t k = e.x;
        bool has_pl = false, is_pt = false;
        double* P = s_pl[lane];
        {   // IBA_PlaneFactor (IBACalib2.hpp:152-184)
            IBA_JAC_CONTRACT
            double z6[6] = {0, 0, 0, 0, 0, 0};
            double ssq = 0, G = 0, GH = 0, HH = 0, Gr = 0, Hr = 0, w = 0;
            if (e.y != kNone) {
                const PlaneRec rec = planes[e.y];
                const float4 pt = p4[e.y];
                const double p0[3] = {(double)pt.x, (double)pt.y, (double)pt.z}, n0[3] = {rec.nx, rec.ny, rec.nz};
                const float2 uv = dp.kp_uv[h.kp_base + k];
                const int nconv = plane_factor_core(c, h, dp, k, h.K, (double)uv.x, (double)uv.y, p0, n0, z6, [&](double ru, double rv, double gu, double gv, double hu, double hv) {
IBA_JAC_CONTRACT
                    ssq += ru * ru + rv * rv;
                    G += gu * gu + gv * gv; GH += gu * hu + gv * hv; HH += hu * hu + hv * hv;
                    Gr += gu * ru + gv * rv; Hr += hu * ru + hv * rv;
                });
                if (nconv > 0) {
                    double rho0; huber_w(prm.robust_kernel_delta, ssq, rho0, w);
                    cost += 0.5 * rho0; chi2 += ssq; nf2d += 1.0; nres += 2.0 * nconv;
                    h66 += w * HH; b6 += w * Hr;
                    has_pl = true;
                }
            }
            if (!has_pl) { w = 0; for (int q = 0; q < 6; ++q) z6[q] = 0; }
            *(double2*)(P + 0) = make_double2(z6[0], z6[1]); *(double2*)(P + 2) = make_double2(z6[2], z6[3]); *(double2*)(P + 4) = make_double2(z6[4], z6[5]);
            *(double2*)(P + 6) = make_double2(w * G, w * GH); P[8] = w * Gr;
        }
        {   // Point2Plane / Point2Point (IBACalib2.hpp:570-584, 611-625): first row [u | residual] and the weight
            IBA_JAC_CONTRACT
            double w = 0;
            double* Q = s_r3[lane];
            if (e.z != kNone) {
                const uint32_t pos = e.z & 0x7FFFFFFFu; const bool is_plane = (e.z >> 31) != 0;
                const PlaneRec rec = planes[pos];
                const float4 pt = p4[pos];
                double M[3], dM[7][3];
                p2x_core(c, h, dp.kp_mp[h.kp_base + k], M, dM);
                const double ev[3] = {M[0] - (double)pt.x, M[1] - (double)pt.y, M[2] - (double)pt.z};
                if (is_plane) {
                    const double r = fdot3(ev[0], rec.nx, ev[1], rec.ny, ev[2], rec.nz);
                    double J[8];
                    for (int kk = 0; kk < 7; ++kk) J[kk] = fdot3(dM[kk][0], rec.nx, dM[kk][1], rec.ny, dM[kk][2], rec.nz);
                    J[7] = r;
                    double rho0; huber_w(prm.robust_kernel_3ddelta, r * r, rho0, w);
                    cost += 0.5 * rho0; chi2 += r * r; nfpl += 1.0; nres += 1.0;
#pragma unroll
                    for (int q = 0; q < 8; q += 2) *(double2*)(Q + q) = make_double2(J[q], J[q + 1]);
                } else {
                    const double ssq = fdot3(ev[0], ev[0], ev[1], ev[1], ev[2], ev[2]);
                    double rho0; huber_w(prm.robust_kernel_3ddelta, ssq, rho0, w);
                    cost += 0.5 * rho0; chi2 += ssq; nfpt += 1.0; nres += 3.0;
                    is_pt = true;
                    *(double2*)(Q + 0) = make_double2(dM[0][0], dM[1][0]); *(double2*)(Q + 2) = make_double2(dM[2][0], dM[3][0]);
                    *(double2*)(Q + 4) = make_double2(dM[4][0], dM[5][0]); *(double2*)(Q + 6) = make_double2(dM[6][0], ev[0]);
                }
            } else {
#pragma unroll
                for (int q = 0; q < 8; q += 2) *(double2*)(Q + q) = make_double2(0.0, 0.0);
            }
            P[11] = w;
        }
        const unsigned long long balT = __ballot(is_pt);
        lds_sync();
        // All operands of a group of instructions are fetched before the first of them issues (the LDS latency is paid once per
        // group, not once per instruction); empty slots hold zeros, so there is nothing to skip.
        // plane factors, one row each: eight source lanes per instruction. (The second rank-1 term of a plane factor, e6 v2^T,
        // only contributes H[6][6] and b[6] beyond what symmetry gives: those two sums stay on the VALU.)
        {
            double a[8], bb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double* S = s_pl[8 * j + 2 * kslot + half];
                const double zi = S[ci < 6 ? ci : 5], sc = S[ci < 6 ? 6 : (ci == 6 ? 7 : 8)];   // b = wG z_i | wGH | wGr
                a[j] = ci < 6 ? zi : 0.0;
                bb[j] = ci < 6 ? sc * zi : sc;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], bb[j], acc, 0, 0, 0);
        }
        // first row of every 3d-3d block: eight source lanes per instruction
        {
            double a[8], bb[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int src = 8 * j + 2 * kslot + half;
                const double ui = s_r3[src][ci], w = s_pl[src][11];
                a[j] = ci < 7 ? ui : 0.0;
                bb[j] = w * ui;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j], bb[j], acc, 0, 0, 0);
        }
        // rows 2 and 3 of the point-to-point blocks take the same slots, one row at a time
        if (balT) {
            for (int r = 1; r < 3; ++r) {
                __builtin_amdgcn_wave_barrier();
                if (is_pt) {   // rare: the block's Jacobian again rather than 48 registers held across the common path
                    IBA_JAC_CONTRACT
                    const float4 pt = p4[e.z & 0x7FFFFFFFu];
                    double M[3], dM[7][3];
                    p2x_core(c, h, dp.kp_mp[h.kp_base + k], M, dM);
                    const double er = r == 1 ? M[1] - (double)pt.y : M[2] - (double)pt.z;
                    double* Q = s_r3[lane];
                    *(double2*)(Q + 0) = make_double2(dM[0][r], dM[1][r]); *(double2*)(Q + 2) = make_double2(dM[2][r], dM[3][r]);
                    *(double2*)(Q + 4) = make_double2(dM[4][r], dM[5][r]); *(double2*)(Q + 6) = make_double2(dM[6][r], er);
                }
                lds_sync();
                for (int j = 0; j < 8; ++j) {
                    if (((balT >> (8 * j)) & 0xFFull) == 0ull) continue;
                    const int src = 8 * j + 2 * kslot + half;
                    const bool on = (balT >> src) & 1ull;   // the other lanes' slots still hold their first rows
                    const double ui = on ? s_r3[src][ci] : 0.0, w = s_pl[src][11];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ci < 7 ? ui : 0.0, w * ui, acc, 0, 0, 0);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();   // the slots are rewritten by the next pass
    }
    // [H | b] = diagonal block 0 + diagonal block 1 of the accumulator: lane l holds D[row = (l >> 4) + 4 reg][col = l & 15]
    s_out[lane] = 0.0;
    __builtin_amdgcn_wave_barrier();
    {
        // rows 8..15 / columns 8..15 live 8 lanes up (same row group, col + 8) in registers 2, 3
        const double up0 = __shfl_down(acc[2], 8), up1 = __shfl_down(acc[3], 8);
        if ((lane & 15) < 8) {
            const int col = lane & 15;
            const double v[2] = {acc[0] + up0, acc[1] + up1};
            for (int q = 0; q < 2; ++q) {
                const int r = (lane >> 4) + 4 * q;
                if (r <= col && col < 7) s_out[P_H0 + hidx(r, col)] = v[q];
                else if (col == 7 && r < 7) s_out[P_B0 + r] = v[q];
            }
        }
    }
    {
        const double t0 = wave_sum_f64(chi2), t1 = wave_sum_f64(cost), t2 = wave_sum_f64(nf2d), t3 = wave_sum_f64(nfpl), t4 = wave_sum_f64(nfpt), t5 = wave_sum_f64(nres);
        const double t6 = wave_sum_f64(h66), t7 = wave_sum_f64(b6);
        lds_sync();
        if (lane == 63) {
            s_out[P_CHI2] = t0; s_out[P_COST] = t1; s_out[P_NF_3D2D] = t2; s_out[P_NF_P2PL] = t3; s_out[P_NF_P2PT] = t4; s_out[P_NRES] = t5;
            s_out[P_H0 + hidx(6, 6)] += t6; s_out[P_B0 + 6] += t7;   // the plane factors' share of H[6][6], b[6]
        }
    }
    lds_sync();
    partials[((size_t)b * nrec + rec_base + f) * kPartialStride + lane] = s_out[lane];
}

// Per-residual output of the frozen problem (for Ceres/g2o adaptors and tests): one lane per keypoint, rows at
// row_off[kp] (int64, -1 = keypoint owns no block), plane-factor rows first, then the 3d-3d rows.
__global__ __launch_bounds__(64) void iba_residual_kernel(DevProblem dp, DevParams prm, const Cand* __restrict__ cands, const uint2* __restrict__ assoc,
                                                          const long long* __restrict__ row_off, double* __restrict__ r_out, double* __restrict__ J_out) {
    const int f = blockIdx.y;
    const FrameHdr& h = dp.frames[f];
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= h.K) return;
    const uint2 a = assoc[h.kp_base + k];
    long long row = row_off[h.kp_base + k];
    if (row < 0) return;
    const Cand& c = cands[0];
    const float* xs = dp.xs + h.pt_base; const float* ys = dp.ys + h.pt_base; const float* zs = dp.zs + h.pt_base;
    const PlaneRec* planes = prm.plane_cache ? dp.plane_local + h.pt_base : dp.scratch_local + h.pt_base;   // slot 0 = frozen problem
    if (a.x != kNone && prm.p2pix) {   // IBATestEdge rows: one 2-row edge per matched covisible keyframe
        const double p0[3] = {(double)xs[a.x], (double)ys[a.x], (double)zs[a.x]};
        test_edge_core(c, h, dp, k, h.K, p0, [&](double ru, double rv, const double* Ju, const double* Jv) {
            r_out[row] = ru; r_out[row + 1] = rv;
            for (int i = 0; i < 7; ++i) { J_out[row * 7 + i] = Ju[i]; J_out[(row + 1) * 7 + i] = Jv[i]; }
            row += 2;
        });
    } else
    if (a.x != kNone) {
        const PlaneRec rec = planes[a.x];
        const double p0[3] = {(double)xs[a.x], (double)ys[a.x], (double)zs[a.x]}, n0[3] = {rec.nx, rec.ny, rec.nz};
        const float2 uv = dp.kp_uv[h.kp_base + k];
        double z6[6];
        // two sweeps: z6 is only known after the core has run, so rows are written on the second one
        plane_factor_core(c, h, dp, k, h.K, (double)uv.x, (double)uv.y, p0, n0, z6, [&](double, double, double, double, double, double) {});
        plane_factor_core(c, h, dp, k, h.K, (double)uv.x, (double)uv.y, p0, n0, z6, [&](double ru, double rv, double gu, double gv, double hu, double hv) {
            r_out[row] = ru; r_out[row + 1] = rv;
            for (int i = 0; i < 6; ++i) { J_out[row * 7 + i] = gu * z6[i]; J_out[(row + 1) * 7 + i] = gv * z6[i]; }
            J_out[row * 7 + 6] = hu; J_out[(row + 1) * 7 + 6] = hv;
            row += 2;
        });
    }
    if (a.y != kNone) {
        const uint32_t pos = a.y & 0x7FFFFFFFu; const bool is_plane = (a.y >> 31) != 0;
        const PlaneRec rec = planes[pos];
        double M[3], dM[7][3];
        p2x_core(c, h, dp.kp_mp[h.kp_base + k], M, dM);
        const double e[3] = {M[0] - (double)xs[pos], M[1] - (double)ys[pos], M[2] - (double)zs[pos]};
        if (is_plane) {
            r_out[row] = fdot3(e[0], rec.nx, e[1], rec.ny, e[2], rec.nz);   // (the factor kernels' own expressions: p2x_accum)
            for (int i = 0; i < 7; ++i) J_out[row * 7 + i] = fdot3(dM[i][0], rec.nx, dM[i][1], rec.ny, dM[i][2], rec.nz);
        } else {
            for (int rr = 0; rr < 3; ++rr) { r_out[row + rr] = e[rr]; for (int i = 0; i < 7; ++i) J_out[(row + rr) * 7 + i] = dM[i][rr]; }
        }
    }
}

// sums the per-frame records of each candidate in a fixed order. grid: B blocks of kReduceThreads threads:
// 16 frame groups x 64 slots, loads of a group issued ahead of its (ordered) adds, then an ordered sum of the groups
constexpr int kReduceThreads = 1024;
__global__ __launch_bounds__(kReduceThreads) void iba_reduce_kernel(const double* __restrict__ frame_partials, int nf, double* __restrict__ out) {
    constexpr int NG = kReduceThreads / kPartialStride;
    __shared__ double s[NG][kPartialStride];
    const int b = blockIdx.x, i = threadIdx.x & 63, g = threadIdx.x >> 6;
    const double* src = frame_partials + (size_t)b * nf * kPartialStride;
    const int per = (nf + NG - 1) / NG, f0 = g * per, f1 = min(nf, f0 + per);
    double x = 0;
    int f = f0;
    for (; f + 4 <= f1; f += 4) {
        const double v0 = src[(size_t)f * kPartialStride + i], v1 = src[(size_t)(f + 1) * kPartialStride + i];
        const double v2 = src[(size_t)(f + 2) * kPartialStride + i], v3 = src[(size_t)(f + 3) * kPartialStride + i];
        x = (((x + v0) + v1) + v2) + v3;
    }
    for (; f < f1; ++f) x += src[(size_t)f * kPartialStride + i];
    s[g][i] = x;
    __syncthreads();
    if (g == 0) {
        double t = s[0][i];
#pragma unroll
        for (int q = 1; q < NG; ++q) t += s[q][i];
        out[(size_t)b * kPartialStride + i] = t;
    }
}

// copies n16 16-byte words (pinned host memory -> device): the candidate block of an evaluation
__global__ void iba_fetch_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t n16) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// the derivative half of B candidates (16-byte words [w0, w1) of every Cand): pinned host memory -> device, beside the kernels that are
// still reading the value half
__global__ void iba_fetch_jets_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, uint32_t B, uint32_t w0, uint32_t w1, uint32_t wcand) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x, per = w1 - w0;
    if (i >= B * per) return;
    const uint32_t w = (i / per) * wcand + w0 + i % per;
    dst[w] = src[w];
}

// writes two host-known values into their slots of B partial blocks (frozen-problem counts)
__global__ void iba_set_slots_kernel(double* __restrict__ partials, int B, int slot_a, double va, int slot_b, double vb) {
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);   // (one thread per candidate: a chain takes up to 512 since round 5)
    if (b < B) { partials[(size_t)b * kPartialStride + slot_a] = va; partials[(size_t)b * kPartialStride + slot_b] = vb; }
}

}  // namespace iba
