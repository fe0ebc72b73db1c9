// CDNA4 (gfx950) kernels of the IBA cross-modality evaluation path. Written for wave64 / 160 KB LDS.
//
// iba_frame_kernel: ONE workgroup (kThreads = 512: 8 waves, two workgroups per CU) evaluates ONE keyframe for ONE
// candidate extrinsic, start to finish. The scan's kd-tree nodes, the keypoint reject-bitmap, the coarse keypoint CSR,
// the per-keypoint 1-NN slots, the candidate queue and the work lists live in LDS; the scan itself (float32 SoA in kd
// leaf order) is read through the XCD's L2, which all candidates of a frame share (see iba_types.hpp / DESIGN.md 5):
//   phase 0.5 conservative frustum test of static 64-point chunk boxes -> compacted list of visible chunks
//   phase 1a  stream the visible chunks (16 B/lane), project in float32, test one bit of the dilated reject bitmap,
//             queue the ~10 % that may match
//   phase 1b  K1+K2+K3 exact: Tcl*p, pinhole projection, FOV test in f64, lookup in the static keypoint grid,
//             ds_min_u64 on the keypoint's best d^2 (replaces TransformPointCloud + the per-evaluation KDTree2D
//             rebuild + 1-NN queries, pointcloud.h:82-86, iba_global.cpp:55-96)
//   phase 2   exact tie resolution (lowest original point index) for the few points that hit
//   phase 3   corrset size test (iba_global.cpp:203, iba_local.cpp:192)
//   phase 4   K6: covisible reprojection residuals (iba_global.cpp:291-328);
//             K4+K5: MapPoint -> LiDAR frame, exact 1-NN by a resumable float32-conservative walk of the LDS-resident
//             tree with exact leaf scans, local plane (memoised per scan point, or refitted per evaluation)
//             (iba_global.cpp:223-252, 111-156; iba_local.cpp:207-300); K7: hand-eye term (iba_global.cpp:264-276)
//   phase 5   K8: fixed-order wave/block reduction -> one partial record per (candidate, frame); in the fused mode also
//             the dense residual-block list iba_factor_kernel consumes
// iba_reduce_kernel sums the records over frames in a fixed order (bitwise reproducible).
// iba_plane_kernel: wave-per-query kNN(<=32)+covariance+closed-form eigen = the x-independent part of
// ComputeAlignmentDist / ComputeLocalNeighbor / ComputeLocalNormalSingleThre.
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
    const float4* kp_mp;       // MapPoint world position (x,y,z); w = 1*(owns a MapPoint) + 2*(matched in >= 1 covisible KF)
    const uint32_t* kp_fl;     // the same flag word (w) as an integer, for kernels that keep it in LDS
    const uint32_t* coarse_start; const float4* crec; const uint32_t* bitmap;   // keypoint grid (see iba_build.hpp)
    const float2* match_uv;    // [slot][K] matched covisible keypoint, NaN = no match
    const PlaneRec* plane_cost;   // x-independent plane records (norm_radius / norm_max_pts)
    const PlaneRec* plane_local;  // (neigh_radius / neigh_max_pts)
    int32_t n_frames;
    int64_t n_kp_total;
    // plane_cache = 0: per-candidate plane records refitted inside every evaluation (slot 0 = frozen problem)
    PlaneRec* scratch_cost; PlaneRec* scratch_local;
    int64_t n_pt_total;
    int32_t scratch_slot_base;
};

struct LdsLayout {   // byte offsets into dynamic LDS, computed on the host from max P/K/D over frames
    uint32_t scan_stride;   // floats per coordinate array (0 = scan not staged in LDS)
    uint32_t off_best_d2, off_best_idx, off_nodes, off_bitmap, off_cstart, off_red, off_vis, vis_words, off_cand, cand_cap, total;
    uint32_t off_pair, pair_cap;   // iba_assoc_kernel only: (point, keypoint) pairs, 16 B each
    uint32_t off_kuv, off_kfl;     // iba_assoc_kernel only: (u, v) and flag word of every keypoint
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

// ---- stackless exact 1-NN in the implicit balanced kd-tree (one query per lane) ----
__device__ __forceinline__ void nn_search(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                                          const TreeNode* __restrict__ nodes, const uint32_t* __restrict__ perm_g, uint32_t P, uint32_t D,
                                          double qx, double qy, double qz, double& best, uint32_t& bpos) {
    best = INFINITY; bpos = kNone;
    uint32_t node = 0, depth = 0;
    const uint32_t first_leaf = (1u << D) - 1u;
    for (;;) {
        {
            while (depth < D) {
                const TreeNode n = nodes[node];
                const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
                node = 2u * node + 1u + ((qd - (double)n.split) >= 0.0 ? 1u : 0u);
                ++depth;
            }
            const uint32_t j = node - first_leaf;
            const uint32_t lo = (uint32_t)(((uint64_t)j * P) >> D), hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
            for (uint32_t i = lo; i < hi; ++i) {
                const double dx = qx - (double)xs[i], dy = qy - (double)ys[i], dz = qz - (double)zs[i];
                const double d2 = (dx * dx + dy * dy) + dz * dz;
                if (d2 < best) { best = d2; bpos = i; }
                else if (d2 == best && bpos != kNone) { if (perm_g[i] < perm_g[bpos]) bpos = i; }
            }
        }
        // climb in a tight loop until a far child can still hold a closer (or tying) point; lanes of a wave then
        // re-converge once per leaf visit instead of once per tree level
        bool go = false;
        while (depth > 0) {
            const uint32_t parent = (node - 1u) >> 1;
            const bool was_right = (node & 1u) == 0u;
            const TreeNode n = nodes[parent];
            const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
            const double diff = qd - (double)n.split;
            const bool near_right = diff >= 0.0;
            if (was_right == near_right && diff * diff <= best) { node = 2u * parent + 1u + (near_right ? 0u : 1u); go = true; break; }
            node = parent; --depth;
        }
        if (!go) break;
    }
}

// value of lane j for a wave-uniform j (two v_readlane), and the value of lane - 1 (DPP wave_shr:1; lane 0 gets 0)
__device__ __forceinline__ double lane_bcast(double v, int j) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, j), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), j);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ double wave_shr1(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x138, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- wave-cooperative kNN(max_pts <= 32, d^2 < r2) + plane fit around scan point `cpos` ----
// All 64 lanes must be active; traversal state is wave-uniform; leaf points are tested one per lane and
// inserted into the sorted list held one entry per lane (lane i = i-th nearest).
__device__ inline PlaneRec plane_fit_wave(const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                                          const TreeNode* __restrict__ nodes, uint32_t P, uint32_t D, uint32_t cpos, double r2, int max_pts) {
    const int lane = threadIdx.x & 63;
    const double qx = (double)xs[cpos], qy = (double)ys[cpos], qz = (double)zs[cpos];
    double my_d = INFINITY; uint32_t my_pos = kNone;
    int count = 0; double bound = r2;
    uint32_t node = 0, depth = 0;
    const uint32_t first_leaf = (1u << D) - 1u;
    for (;;) {
        {
            while (depth < D) {
                const TreeNode n = nodes[node];
                const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
                node = 2u * node + 1u + ((qd - (double)n.split) >= 0.0 ? 1u : 0u);
                ++depth;
            }
            const uint32_t j = node - first_leaf;
            const uint32_t lo = (uint32_t)(((uint64_t)j * P) >> D), hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
            for (uint32_t base = lo; base < hi; base += 64) {
                const uint32_t i = base + lane;
                double d2 = INFINITY;
                if (i < hi) {
                    const double dx = qx - (double)xs[i], dy = qy - (double)ys[i], dz = qz - (double)zs[i];
                    d2 = (dx * dx + dy * dy) + dz * dz;
                }
                unsigned long long mask = __ballot(d2 < bound);
                while (mask) {   // l, ins, count are wave-uniform: broadcasts are v_readlane, the shift is one DPP wave_shr — no LDS round trips
                    const int l = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const double cd = lane_bcast(d2, l);
                    if (cd < bound) {
                        const int ins = __popcll(__ballot(lane < count && my_d <= cd));
                        const double up_d = wave_shr1(my_d); const uint32_t up_p = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)my_pos, 0x138, 0xf, 0xf, false);
                        if (lane > ins) { my_d = up_d; my_pos = up_p; }
                        else if (lane == ins) { my_d = cd; my_pos = base + l; }
                        if (count < max_pts) ++count;
                        if (count == max_pts) bound = fmin(r2, lane_bcast(my_d, max_pts - 1));
                    }
                }
            }
        }
        bool go = false;
        while (depth > 0) {
            const uint32_t parent = (node - 1u) >> 1;
            const bool was_right = (node & 1u) == 0u;
            const TreeNode n = nodes[parent];
            const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
            const double diff = qd - (double)n.split;
            const bool near_right = diff >= 0.0;
            if (was_right == near_right && diff * diff < bound) { node = 2u * parent + 1u + (near_right ? 0u : 1u); go = true; break; }
            node = parent; --depth;
        }
        if (!go) break;
    }
    PlaneRec rec;
    rec.k = count; rec.pad = 0;
    rec.far_d2 = count > 0 ? lane_bcast(my_d, count - 1) : 0.0;
    // ComputeCovariance: one-pass raw moments in list order (pointcloud.h:126-158). Lane j gathers list entry j and forms
    // its nine terms (all gathers in flight together); the sums then run over j in list order on broadcast values, so
    // every addition happens in the reference's order.
    double mine[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double mx = 0, my = 0, mz = 0;
    if (lane < count) {
        mx = (double)xs[my_pos]; my = (double)ys[my_pos]; mz = (double)zs[my_pos];
        mine[0] = mx; mine[1] = my; mine[2] = mz;
        mine[3] = mx * mx; mine[4] = mx * my; mine[5] = mx * mz; mine[6] = my * my; mine[7] = my * mz; mine[8] = mz * mz;
    }
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < count; ++j) {
#pragma unroll
        for (int q = 0; q < 9; ++q) c[q] += lane_bcast(mine[q], j);
    }
    const double inv_n = (double)count;
    for (int i = 0; i < 9; ++i) c[i] /= inv_n;
    double cov[9];
    cov[0] = c[3] - c[0] * c[0]; cov[4] = c[6] - c[1] * c[1]; cov[8] = c[8] - c[2] * c[2];
    cov[1] = cov[3] = c[4] - c[0] * c[1]; cov[2] = cov[6] = c[5] - c[0] * c[2]; cov[5] = cov[7] = c[7] - c[1] * c[2];
    double nrm[3]; dev_smallest_evec(cov, nrm);
    double reg = 0;
    {
        const double ax = mx - qx, ay = my - qy, az = mz - qz;
        const double term = fabs(ax * nrm[0] + ay * nrm[1] + az * nrm[2]);   // lane j: |(p_j - c) . n|
        for (int j = 0; j < count; ++j) reg += lane_bcast(term, j);
    }
    rec.nx = nrm[0]; rec.ny = nrm[1]; rec.nz = nrm[2]; rec.reg_sum = reg;
    return rec;
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

constexpr int kFitPathMax = 12;   // >= the builder's depth cap (kMaxTreeDepth = 11)
constexpr int kFitListStride(int slots) { return 16 * slots + 1; }   // words per list in LDS (odd: lane-per-fit reads are conflict-free)
template <int SLOTS> struct FitLds {
    uint32_t list[64][kFitListStride(SLOTS)];   // tree positions of the kept neighbours, nearest first
    double far_d2[64];
    int32_t count[64];
};

// lists of the four scan points `cpos` (one per row; kNone: idle row) -> lds.list[fit], count[fit], far_d2[fit]; fit = fit0 + row
template <int SLOTS>
__device__ __forceinline__ void fit_list_rows(const float4* __restrict__ p4, const TreeNode* __restrict__ nodes, uint32_t P, uint32_t D, uint32_t cpos, double r2, int max_pts,
                                     FitLds<SLOTS>& lds, int fit0) {
    const int lane = threadIdx.x & 63, gl = lane & 15, row_sh = lane & 48;
    const bool act = cpos != kNone;
    double qx = 0, qy = 0, qz = 0;
    if (act) { const float4 c = p4[cpos]; qx = (double)c.x; qy = (double)c.y; qz = (double)c.z; }
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
                    const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
                    const double diff = qd - (double)n.split;
                    const uint32_t r = diff >= 0.0 ? 1u : 0u;
                    pl[L] = (float)(diff * diff) * 0.9999998f;
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
            const uint32_t m16 = (uint32_t)(m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffu;   // lanes-of-a-row with a candidate in ANY row
#define IBA_FIT_STEP(L) if (m16 & (1u << L)) fit_insert<L, SLOTS>(d2q, base, ed, ep);
            IBA_FIT_STEP(0) IBA_FIT_STEP(1) IBA_FIT_STEP(2) IBA_FIT_STEP(3) IBA_FIT_STEP(4) IBA_FIT_STEP(5) IBA_FIT_STEP(6) IBA_FIT_STEP(7)
            IBA_FIT_STEP(8) IBA_FIT_STEP(9) IBA_FIT_STEP(10) IBA_FIT_STEP(11) IBA_FIT_STEP(12) IBA_FIT_STEP(13) IBA_FIT_STEP(14) IBA_FIT_STEP(15)
#undef IBA_FIT_STEP
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

// K1+K2: Tcl*p (pointcloud.h:82-86), pinhole projection and FOV cull (iba_global.cpp:68-81) for one scan point
__device__ __forceinline__ bool project_uv(const FrameCtx& c, float xf, float yf, float zf, double& u, double& v) {
    const double x = (double)xf, y = (double)yf, z = (double)zf;
    const double pcx = ((c.R[0] * x + c.R[1] * y) + c.R[2] * z) + c.t[0];
    const double pcy = ((c.R[3] * x + c.R[4] * y) + c.R[5] * z) + c.t[1];
    const double pcz = ((c.R[6] * x + c.R[7] * y) + c.R[8] * z) + c.t[2];
    if (!(pcz > 0)) return false;
    u = (c.fx * pcx + c.cx * pcz) / pcz;
    v = (c.fx * pcy + c.cy * pcz) / pcz;   // fx on purpose: iba_global.cpp:73
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
__device__ __forceinline__ bool grid_match(const FrameCtx& c, double u, double v, uint32_t pos) {
    const float uf = (float)u, vf = (float)v;
    const int x0 = grid_cell(uf - c.margin, c.gw) >> kCoarseShift, x1 = grid_cell(uf + c.margin, c.gw) >> kCoarseShift;
    const int y0 = grid_cell(vf - c.margin, c.gh) >> kCoarseShift, y1 = grid_cell(vf + c.margin, c.gh) >> kCoarseShift;
    bool hit = false;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t e0 = c.cstart[yy * c.gwc + x0], e1 = c.cstart[yy * c.gwc + x1 + 1];
        for (uint32_t e = e0; e < e1; ++e) {
            const float4 rec = c.crec[e];
            if (fabsf(rec.x - uf) > c.margin || fabsf(rec.y - vf) > c.margin) continue;   // cheap f32 reject (margin has 0.01 px slack)
            const double du = (double)rec.x - u, dv = (double)rec.y - v;
            const double d2 = du * du + dv * dv;
            if (d2 <= c.gate2) {
                const uint32_t k = __float_as_uint(rec.z);
                if (PASS == 1) { atomicMin(&c.best_d2[k], d2bits(d2)); hit = true; }
                else if (c.best_d2[k] == d2bits(d2)) atomicMin(&c.best_idx[k], c.perm[pos]);
            }
        }
    }
    return hit;
}

// PASS-1 variant that also remembers up to two (keypoint, d^2) hits of this point (returned BY VALUE so they stay
// in registers), so that the tie pass needs no second grid walk. n counts all hits (a third is counted, not stored).
struct Hits { uint32_t k0, k1; unsigned long long d0, d1; int n; };
__device__ __forceinline__ Hits grid_match_rec(const FrameCtx& c, double u, double v) {
    const float uf = (float)u, vf = (float)v;
    const int x0 = grid_cell(uf - c.margin, c.gw) >> kCoarseShift, x1 = grid_cell(uf + c.margin, c.gw) >> kCoarseShift;
    const int y0 = grid_cell(vf - c.margin, c.gh) >> kCoarseShift, y1 = grid_cell(vf + c.margin, c.gh) >> kCoarseShift;
    Hits hh; hh.k0 = hh.k1 = 0u; hh.d0 = hh.d1 = 0ull; hh.n = 0;
    for (int yy = y0; yy <= y1; ++yy) {
        const uint32_t e0 = c.cstart[yy * c.gwc + x0], e1 = c.cstart[yy * c.gwc + x1 + 1];
        for (uint32_t e = e0; e < e1; ++e) {
            const float4 rec = c.crec[e];
            if (fabsf(rec.x - uf) > c.margin || fabsf(rec.y - vf) > c.margin) continue;
            const double du = (double)rec.x - u, dv = (double)rec.y - v;
            const double d2 = du * du + dv * dv;
            if (d2 <= c.gate2) {
                const uint32_t k = __float_as_uint(rec.z);
                atomicMin(&c.best_d2[k], d2bits(d2));
                const bool first = hh.n == 0, second = hh.n == 1;
                hh.k0 = first ? k : hh.k0; hh.d0 = first ? d2bits(d2) : hh.d0;
                hh.k1 = second ? k : hh.k1; hh.d1 = second ? d2bits(d2) : hh.d1;
                ++hh.n;
            }
        }
    }
    return hh;
}

// ---- exact 1-NN with G lanes per query (G = 1,2,4,8): the G lanes walk the tree in lockstep and split each leaf ----
#ifdef IBA_STAMPS
__device__ unsigned long long g_dbg[64];
#endif
// ---- exact 1-NN with G lanes per query (G = 1,2,4,8; run-time so that ONE copy of the traversal exists) ----
// The G lanes walk the tree in lockstep and split each leaf. The descent keeps the visited path in registers
// (split value per level, 2-bit dim, side and done bits), so backtracking — otherwise a chain of dependent LDS
// reads, one per tree level — is pure VALU: the deepest unfinished level whose split plane is within the current
// best distance is found with an unrolled scan, its far child is entered, and only the levels below are fetched.
constexpr int kPathMax = 12;   // levels whose plane bounds are kept in registers (the builder caps D at kMaxTreeDepth = 11); deeper trees use the generic LDS-walking branch
static_assert(kPathMax % 4 == 0 && kPathMax >= kMaxTreeDepth, "register path must cover the depth cap");
#ifndef IBA_NN_TO_END
#define IBA_NN_TO_END 0
#endif
#ifndef IBA_HIT_SLOTS
#define IBA_HIT_SLOTS 3
#endif
#ifndef IBA_LEAF_BATCH
#define IBA_LEAF_BATCH 2   /* points of a leaf scan whose loads are in flight together */
#endif
#ifndef IBA_RELOAD_MASK
#define IBA_RELOAD_MASK 0xff   /* phase boundaries (IBA_STAMP indices) at which the kernarg pointer is laundered */
#endif
#ifndef IBA_FRAME_WAVES
#define IBA_FRAME_WAVES 4   /* waves per SIMD the frame kernel is compiled for (register budget 512 / this) */
#endif
#ifndef IBA_NN_RESUME_GMAX
#define IBA_NN_RESUME_GMAX 32
#endif
#ifndef IBA_NN_FRESH_GMAX
#define IBA_NN_FRESH_GMAX 8
#endif
#ifndef IBA_NN_G_SLACK
#define IBA_NN_G_SLACK 0
#endif
#ifndef IBA_NN_END_AT
#define IBA_NN_END_AT 128
#endif
constexpr int kNNEndAt = IBA_NN_END_AT;       // resume rounds run to the end once this few queries are left
constexpr bool kNNToEnd = IBA_NN_TO_END;   // experiment switch: 1 = every query runs to completion in round 0
__device__ __forceinline__ void nn_search_group(int G, const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs,
                                                const TreeNode* __restrict__ nodes, const uint32_t* __restrict__ perm_g, uint32_t P, uint32_t D,
                                                double qx, double qy, double qz, double& best, uint32_t& bpos) {
    const uint32_t sub = threadIdx.x & (uint32_t)(G - 1);
    best = INFINITY; bpos = kNone;
    const uint32_t first_leaf = (1u << D) - 1u;
    auto scan_leaf = [&](uint32_t node) {
        const uint32_t j = node - first_leaf;
        const uint32_t lo = (uint32_t)(((uint64_t)j * P) >> D), hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
#pragma unroll 2
        for (uint32_t i = lo + sub; i < hi; i += G) {
            const double dx = qx - (double)xs[i], dy = qy - (double)ys[i], dz = qz - (double)zs[i];
            const double d2 = (dx * dx + dy * dy) + dz * dz;
            if (d2 < best) { best = d2; bpos = i; }
            else if (d2 == best && bpos != kNone) { if (perm_g[i] < perm_g[bpos]) bpos = i; }
        }
        for (int off = 1; off < G; off <<= 1) {
            const double od = __shfl_xor(best, off); const uint32_t op = __shfl_xor(bpos, off);
            if (od < best) { best = od; bpos = op; }
            else if (od == best && op != kNone && op != bpos) { if (bpos == kNone || perm_g[op] < perm_g[bpos]) bpos = op; }
        }
    };
    if (D > (uint32_t)kPathMax) {   // generic branch: ancestors are re-read from the node array on the way up
        uint32_t node = 0, depth = 0;
        for (;;) {
            while (depth < D) {
                const TreeNode n = nodes[node];
                const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
                node = 2u * node + 1u + ((qd - (double)n.split) >= 0.0 ? 1u : 0u);
                ++depth;
            }
            scan_leaf(node);
            bool go = false;
            while (depth > 0) {
                const uint32_t parent = (node - 1u) >> 1;
                const bool was_right = (node & 1u) == 0u;
                const TreeNode n = nodes[parent];
                const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
                const double diff = qd - (double)n.split;
                const bool near_right = diff >= 0.0;
                if (was_right == near_right && diff * diff <= best) { node = 2u * parent + 1u + (near_right ? 0u : 1u); go = true; break; }
                node = parent; --depth;
            }
            if (!go) break;
        }
        return;
    }
    // per level: float LOWER bound of (q[dim] - split)^2 (rounded down). Backtracking first builds, with one f32 compare
    // per level and no memory access, the set of levels whose plane may still be within the best distance; only the
    // deepest such level is then confirmed exactly (one node re-read) before its far child is entered.
    float pd2[kPathMax];
#pragma unroll
    for (int L = 0; L < kPathMax; ++L) pd2[L] = INFINITY;
    uint32_t side = 0u, done = 0u;   // per level: 1 = currently in the right child; 1 = far child handled / out of reach
    uint32_t node = 0u; int start = 0;
    for (;;) {
#pragma unroll
        for (int L = 0; L < kPathMax; ++L) {
            if (L >= start && L < (int)D) {
                const TreeNode n = nodes[node];
                const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
                const double diff = qd - (double)n.split;
                const uint32_t r = diff >= 0.0 ? 1u : 0u;
                pd2[L] = __double2float_rd(diff * diff);
                side = (side & ~(1u << L)) | (r << L);
                done &= ~(1u << L);
                node = 2u * node + 1u + r;
            }
        }
        scan_leaf(node);
        int go = -1;
        for (;;) {
            const float bestf = __double2float_ru(best);
            uint32_t cand = 0u;
#pragma unroll
            for (int L = 0; L < kPathMax; ++L) cand |= (pd2[L] <= bestf ? 1u : 0u) << L;
            cand &= ~done & ((1u << D) - 1u);
            done |= ~cand;                               // best only shrinks: a level out of reach stays out of reach
            if (cand == 0u) break;
            const int L = 31 - __clz((int)cand);         // deepest candidate level
            const uint32_t anc = ((node + 1u) >> (D - (uint32_t)L)) - 1u;   // ancestor of the current leaf at level L
            const TreeNode n = nodes[anc];
            const double qd = n.dim == 0 ? qx : (n.dim == 1 ? qy : qz);
            const double diff = qd - (double)n.split;
            done |= 1u << L;
            if (diff * diff <= best) { go = L; side ^= 1u << L; node = 2u * anc + 1u + ((side >> L) & 1u); break; }
        }
        if (go < 0) break;
        start = go + 1;
    }
}

// exchange with the partner lane of a butterfly step on the VALU (DPP), no LDS round trip:
// quad_perm [1,0,3,2] / [2,3,0,1], then row_half_mirror / row_mirror (after the quad steps every lane of a quad holds
// the same value, so mirroring pairs the two halves). Only the 32-lane step needs a cross-row move (ds_swizzle).
__device__ __forceinline__ void nn_merge(double& best, uint32_t& bpos, double od, uint32_t op, const uint32_t* __restrict__ perm_g) {
    if (od < best) { best = od; bpos = op; }
    else if (od == best && op != kNone && op != bpos) { if (bpos == kNone || perm_g[op] < perm_g[bpos]) bpos = op; }
}
__device__ __forceinline__ void nn_group_reduce(int G, double& best, uint32_t& bpos, const uint32_t* __restrict__ perm_g) {
    if (G >= 2) nn_merge(best, bpos, dpp_f64<0xB1, 0xf>(best), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bpos, 0xB1, 0xf, 0xf, false), perm_g);
    if (G >= 4) nn_merge(best, bpos, dpp_f64<0x4E, 0xf>(best), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bpos, 0x4E, 0xf, 0xf, false), perm_g);
    if (G >= 8) nn_merge(best, bpos, dpp_f64<0x141, 0xf>(best), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bpos, 0x141, 0xf, 0xf, false), perm_g);
    if (G >= 16) nn_merge(best, bpos, dpp_f64<0x140, 0xf>(best), (uint32_t)__builtin_amdgcn_update_dpp(0, (int)bpos, 0x140, 0xf, 0xf, false), perm_g);
    if (G >= 32) nn_merge(best, bpos, __shfl_xor(best, 16), __shfl_xor(bpos, 16), perm_g);
}

// v_min / v_max without the canonicalising v_max the IEEE-exact fminf / fmaxf expansion adds in front (a NaN operand
// returns the other one, like minnum / maxnum)
__device__ __forceinline__ float vmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// ---- two exact 1-NN searches in ONE traversal (fused mode), resumable leaf by leaf ----
// The association-path query qa and the cost-path query qc of the same MapPoint differ by ~1e-7 relative (different
// float/double islands in the reference), so they visit the same leaves. Pruning is the union of what either query
// still needs: per level the smaller of the two plane distances (rounded down) against the larger of the two current
// bests (rounded up), confirmed exactly per query. Results are identical to two separate searches.
//
// 72 % of the queries are finished after their first leaf, a few need 5..24: run to completion per lane group, a wave
// idles on its slowest query for 3/4 of the search time. So the traversal is cut into steps of ONE leaf visit. A step
// returns whether a far subtree is still within reach; the whole traversal state that has to survive is 64 bits
// (current leaf, done mask, level to enter) + the running bests: everything else (the per-level plane distances, the
// side bits) is a function of the current leaf's ancestors, which are addressable without walking
// (ancestor at level L of heap node n: ((n+1) >> (D-L)) - 1). Between steps the unfinished queries are compacted
// and re-spread over the block with more lanes each (see the fused branch of iba_frame_kernel).
struct DualNN {
    double bestA, bestC;        // running best d^2 (INFINITY at the start)
    uint32_t bposA, bposC;      // tree position of the best point
    uint32_t leaf, done;        // leaf index of the last visited leaf; per level: far side handled / out of reach
    int go;                     // level whose far child comes next (valid when a step returned true)
#ifdef IBA_STAMPS_FINE
    uint32_t visits;
#endif
};
template <int WHICH, bool AOS>   // WHICH bit 0: association-path query a is present, bit 1: cost-path query c (single-query modes compile the other half away); AOS: leaves are read from p4
__device__ __forceinline__ bool nn_dual_step(int G, const float* __restrict__ xs, const float* __restrict__ ys, const float* __restrict__ zs, const float4* __restrict__ p4,
                                             const TreeNode* __restrict__ nodes, const uint32_t* __restrict__ perm_g, uint32_t P, uint32_t D,
                                             bool actA, double ax, double ay, double az, bool actC, double cx, double cy, double cz,
                                             DualNN& st, bool fresh, bool to_end) {
    if (D > (uint32_t)kPathMax) {   // very deep trees: two plain searches, always to the end
        st.bestA = INFINITY; st.bposA = kNone; st.bestC = INFINITY; st.bposC = kNone;
        if ((WHICH & 1) && actA) nn_search_group(G, xs, ys, zs, nodes, perm_g, P, D, ax, ay, az, st.bestA, st.bposA);
        if ((WHICH & 2) && actC) nn_search_group(G, xs, ys, zs, nodes, perm_g, P, D, cx, cy, cz, st.bestC, st.bposC);
        return false;
    }
    const uint32_t sub = threadIdx.x & (uint32_t)(G - 1);
    const uint32_t first_leaf = (1u << D) - 1u;
    // The tree walk is float32 and CONSERVATIVE; only the leaf scans are exact. One float query p steers the descent;
    // del >= |q - p| (per axis, both queries). For a node with split s, d = fl(p - s):
    //   exact |q - s| >= |p - s| - del >= |d| (1 - 2^-24) - del,
    // so lb = max(|d| - del', 0)^2 * (1 - 2^-19) with del' = del (1 + 2^-19) stays below the exact squared plane
    // distance of either query through all float roundings. A far side is skipped only if lb > RN_float(best), which
    // implies lb > best; anything else is visited, so the result equals the exhaustive exact search.
    if (!(WHICH & 1)) actA = false;
    if (!(WHICH & 2)) actC = false;
    const float p0 = (float)(actC ? cx : ax), p1 = (float)(actC ? cy : ay), p2 = (float)(actC ? cz : az);
    float del;
    {
        double m = 0.0;
        if (actA) m = fmax(fmax(fabs(ax - (double)p0), fabs(ay - (double)p1)), fabs(az - (double)p2));
        if (actC) m = fmax(m, fmax(fmax(fabs(cx - (double)p0), fabs(cy - (double)p1)), fabs(cz - (double)p2)));
        del = (float)m * 1.00001f + 1e-30f;
    }
    // (the factor is applied to a as c = 0.999999f, c^2 <= 1 - 2^-19: one fma instead of a subtraction and a multiplication)
    const float delc = del * 0.999999f;
    auto lower_bound = [&](float d) { const float a = fmaxf(fmaf(fabsf(d), 0.999999f, -delc), 0.f); return a * a; };
    // leaf filter, see the leaf scan below; the middle term of e_const covers a flushed sqrt of a denormal u
    const float e_lin = 2.01f * del, e_const = 3.01f * del * del + 2.1e-19f * e_lin + 1e-37f;
    // contract: an inactive query comes in with NaN coordinates (the leaf scans run both queries unconditionally)
#ifdef IBA_STAMPS_FINE
    unsigned long long sg0 = __builtin_readcyclecounter(), sg1 = sg0, sg2 = sg0, sg3 = sg0, sg4 = sg0, sg5 = sg0;
    const int sgb = fresh ? 48 : 56;
#endif
    double bestA = st.bestA, bestC = st.bestC; uint32_t bposA = st.bposA, bposC = st.bposC;
    float pd2[kPathMax];
    uint32_t side = 0u, done = 0u, node = 0u; int start = 0, go = -1;
    if (fresh) {
#pragma unroll
        for (int L = 0; L < kPathMax; ++L) pd2[L] = INFINITY;
    } else {   // rebuild the path registers from the ancestors of the last leaf: independent LDS reads, issued together
        node = first_leaf + st.leaf; done = st.done; go = st.go;
#pragma unroll
        for (int L = 0; L < kPathMax; ++L) pd2[L] = INFINITY;
#pragma unroll
        for (int H = 0; H < kPathMax; H += 4) {   // four levels per batch of reads
            if (H < (int)D) {
                TreeNode nn[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int L = H + q; nn[q] = nodes[((node + 1u) >> (D - (uint32_t)(L < (int)D ? L : (int)D - 1))) - 1u]; }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int L = H + q;
                    const float lb = lower_bound((nn[q].dim == 0 ? p0 : (nn[q].dim == 1 ? p1 : p2)) - nn[q].split);
                    pd2[L] = L < (int)D ? lb : INFINITY;
                    if (L < (int)D) side |= (((node + 1u) >> (D - (uint32_t)L - 1u)) & 1u) << L;
                }
            }
        }
    }
    for (;;) {
#ifdef IBA_STAMPS_FINE
        sg1 = __builtin_readcyclecounter();
#endif
        if (go >= 0) {   // enter the far child at level go
            const uint32_t anc = ((node + 1u) >> (D - (uint32_t)go)) - 1u;
            done |= 1u << go; side ^= 1u << go;
            node = 2u * anc + 1u + ((side >> go) & 1u);
            start = go + 1;
        }
        {   // levels start .. D-1: their side / done bits are rewritten; n1 = node + 1 (children 2 n1 and 2 n1 + 1)
            const uint32_t keep = (1u << start) - 1u;
            side &= keep; done &= keep;
            uint32_t n1 = node + 1u;
#pragma unroll
            for (int L = 0; L < kPathMax; ++L) {
                if (L >= (int)D) break;   // uniform: one scalar branch ends the unrolled chain
                if (L >= start) {
                    const TreeNode n = nodes[n1 - 1u];
                    const float d = (n.dim == 0 ? p0 : (n.dim == 1 ? p1 : p2)) - n.split;
                    // right child iff the sign bit of d is clear (d = -0 goes left where d >= 0 would go right: the bound
                    // of that plane is 0 either way, so both children are visited and the result does not depend on it)
                    const uint32_t r = (~__float_as_uint(d)) >> 31;
                    pd2[L] = lower_bound(d);
                    side |= r << L;
                    n1 = (n1 << 1) | r;
                }
            }
            node = n1 - 1u;
        }
        {
#ifdef IBA_STAMPS_FINE
            sg2 = __builtin_readcyclecounter();
#endif
            const uint32_t j = node - first_leaf;
            const uint32_t lo = (uint32_t)(((uint64_t)j * P) >> D), hi = (uint32_t)(((uint64_t)(j + 1) * P) >> D);
            // The leaf is scanned in FLOAT first and confirmed exactly afterwards. With d_k = fl(p_k - x_k) every exact
            // axis difference of either query lies within del + 2^-24 |d_k| of d_k, so for u = fl(sum d_k^2) and
            // s1 = sum |d_k| <= sqrt(3 u):  |exact d^2 - u| <= 2 del s1 + 3 del^2 + 3.2e-7 u <= E(u) with
            //   E(u) = 1.001 e_lin sqrt(3 u) + 1.5e-6 u + e_const      (e_lin = 2.01 del, e_const = 3.01 del^2; the f64
            // rounding of the exact value is 1e-16, far inside the slack). g(u) = u - E(u) is a lower bound of a point's
            // exact d^2 and is increasing for u >= 4 e_lin^2. The scan keeps the two smallest u and the index of the
            // smallest. If g(m2) > m1 + E(m1) (all but ~1e-4 of the visits) only the arg-min can hold the leaf's exact
            // minimum and only that point is evaluated in double for the two queries — or none at all when g(m1) is
            // already above both running bests; otherwise every point with g(u) <= m1 + E(m1) is evaluated. Either way
            // the running bests see the same exact values and the same tie rule (lowest original index) as an
            // all-double scan. An inactive query has NaN coordinates: every compare is false, its index stays kNone.
            float m1 = INFINITY, m2 = INFINITY; uint32_t mi = kNone;
            auto point = [&](uint32_t i, float& x, float& y, float& z) {
                if (AOS) { const float4 v = p4[i]; x = v.x; y = v.y; z = v.z; }
                else { x = xs[i]; y = ys[i]; z = zs[i]; }
            };
            // (raw v_sqrt_f32 is good to 1 ulp, covered by the 1.001 factor; a denormal argument may come back as 0, covered by e_const)
            auto err_of = [&](float u) { return fmaf(1.001f * e_lin, __builtin_amdgcn_sqrtf(3.f * u), fmaf(1.5e-6f, u, e_const)); };
            float thi = INFINITY;   // m1 + E(m1): nothing above it can be the leaf's minimum
            bool single = false, skip = false;
#ifndef IBA_LEAF_EXACT_ONLY
            // the scan is read through L2: the loads of kLeafBatch points are issued together before any is consumed
            // (a step beyond the leaf re-reads its last point and is discarded)
            constexpr int kLeafBatch = IBA_LEAF_BATCH;
            for (uint32_t i0 = lo + sub; i0 < hi; i0 += (uint32_t)(kLeafBatch * G)) {
                float X[kLeafBatch], Y[kLeafBatch], Z[kLeafBatch];
#pragma unroll
                for (int u = 0; u < kLeafBatch; ++u) {
                    const uint32_t iu = i0 + (uint32_t)(u * G), ic = iu < hi ? iu : hi - 1u;
                    if (AOS) { const float4 v = p4[ic]; X[u] = v.x; Y[u] = v.y; Z[u] = v.z; }
                    else { X[u] = xs[ic]; Y[u] = ys[ic]; Z[u] = zs[ic]; }
                }
#pragma unroll
                for (int u = 0; u < kLeafBatch; ++u) {
                    const uint32_t i = i0 + (uint32_t)(u * G);
                    const float dx = p0 - X[u], dy = p1 - Y[u], dz = p2 - Z[u];
                    float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                    uu = i < hi ? uu : INFINITY;
                    mi = uu < m1 ? i : mi;
                    m2 = __builtin_amdgcn_fmed3f(m1, m2, uu);   // second smallest of {m1 <= m2, uu}
                    m1 = vmin(m1, uu);
                }
            }
#define IBA_DPPF(v, ctrl) __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), ctrl, 0xf, 0xf, false))
#define IBA_DPPU(v, ctrl) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, 0xf, 0xf, false)
#define IBA_LEAF_MERGE(o1e, o2e, oie) do { const float o1 = (o1e), o2 = (o2e); const uint32_t oi = (oie); \
                m2 = vmin3(vmax(m1, o1), m2, o2); mi = o1 < m1 ? oi : mi; m1 = vmin(m1, o1); } while (0)
            if (G >= 2) IBA_LEAF_MERGE(IBA_DPPF(m1, 0xB1), IBA_DPPF(m2, 0xB1), IBA_DPPU(mi, 0xB1));
            if (G >= 4) IBA_LEAF_MERGE(IBA_DPPF(m1, 0x4E), IBA_DPPF(m2, 0x4E), IBA_DPPU(mi, 0x4E));
            if (G >= 8) IBA_LEAF_MERGE(IBA_DPPF(m1, 0x141), IBA_DPPF(m2, 0x141), IBA_DPPU(mi, 0x141));
            if (G >= 16) IBA_LEAF_MERGE(IBA_DPPF(m1, 0x140), IBA_DPPF(m2, 0x140), IBA_DPPU(mi, 0x140));
            if (G >= 32) IBA_LEAF_MERGE(__shfl_xor(m1, 16), __shfl_xor(m2, 16), __shfl_xor(mi, 16));
#undef IBA_LEAF_MERGE
#undef IBA_DPPF
#undef IBA_DPPU
            // m1, m2 are the same in every lane of the group after the butterfly; mi may differ only when two lanes hold
            // equal m1, and then m2 == m1: not single.
            {
                const float mono = 4.f * e_lin * e_lin;   // g is increasing from here on
                thi = m1 + err_of(m1);
                single = m2 >= mono && m2 - err_of(m2) > thi;
                // an upper bound of the larger running best in float: round to nearest, then one part in 2^23 up
                const float bnear = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
                const float bmax = fmaf(fabsf(bnear), 1.2e-7f, bnear);
                skip = m1 >= mono && m1 - err_of(m1) > bmax;   // no point of this leaf can reach either running best
#ifdef IBA_LEAF_FORCE_SINGLE   /* fault injection for tests/test_gpu_edge_cases.py: trust the float arg-min blindly */
                single = true;
#endif
            }
#else
            mi = 0u;
#endif
            // exact confirmation: the one candidate, or (rarely) every point of this lane's slice that can reach below
            // thi — then the lanes of the group hold different candidates and the bests are reduced over the group
            if (mi != kNone && !skip) {
                uint32_t i = single ? mi : lo + sub;
                while (single || i < hi) {
                    float xf, yf, zf;
                    point(i, xf, yf, zf);
                    bool take = single;
                    if (!single) {
                        const float dx = p0 - xf, dy = p1 - yf, dz = p2 - zf;
                        const float uu = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
                        take = uu - err_of(uu) <= thi;
                    }
                    if (take) {
                        const double x = (double)xf, y = (double)yf, z = (double)zf;
                        if (WHICH & 1) {
                            const double dx = ax - x, dy = ay - y, dz = az - z;
                            nn_merge(bestA, bposA, (dx * dx + dy * dy) + dz * dz, i, perm_g);
                        }
                        if (WHICH & 2) {
                            const double dx = cx - x, dy = cy - y, dz = cz - z;
                            nn_merge(bestC, bposC, (dx * dx + dy * dy) + dz * dz, i, perm_g);
                        }
                    }
                    if (single) break;
                    i += (uint32_t)G;
                }
#ifdef IBA_STAMPS_FINE
                sg3 = __builtin_readcyclecounter();
#endif
                if (!single) {
                    if (WHICH & 1) nn_group_reduce(G, bestA, bposA, perm_g);
                    if (WHICH & 2) nn_group_reduce(G, bestC, bposC, perm_g);
                }
            }
#ifdef IBA_STAMPS_FINE
            sg4 = __builtin_readcyclecounter();
#endif
        }
        {   // deepest level whose far side may still be within reach of either query
            const float bestf = (float)fmax(actA ? bestA : -INFINITY, actC ? bestC : -INFINITY);
            uint32_t cand = 0u;
#pragma unroll
            for (int L = 0; L < kPathMax; ++L) cand |= (pd2[L] <= bestf ? 1u : 0u) << L;
            cand &= ~done & ((1u << D) - 1u);
            done |= ~cand;                               // the bests only shrink: out of reach stays out of reach
            go = cand ? 31 - __clz((int)cand) : -1;
        }
#ifdef IBA_STAMPS_FINE
        sg5 = __builtin_readcyclecounter();
        if (threadIdx.x == 0) { atomicAdd(&g_dbg[sgb + 1], sg2 - sg1); atomicAdd(&g_dbg[sgb + 2], sg3 - sg2); atomicAdd(&g_dbg[sgb + 3], sg4 - sg3); atomicAdd(&g_dbg[sgb + 4], sg5 - sg4); atomicAdd(&g_dbg[sgb + 5], 1ull); }
#endif
#ifdef IBA_STAMPS_FINE
        st.visits++;
#endif
        if (go < 0 || !to_end) break;
    }
#ifdef IBA_STAMPS_FINE
    if (threadIdx.x == 0) { atomicAdd(&g_dbg[sgb + 0], sg1 - sg0); atomicAdd(&g_dbg[sgb + 6], 1ull); atomicAdd(&g_dbg[sgb + 7], __builtin_readcyclecounter() - sg0); }
#endif
    st.bestA = bestA; st.bestC = bestC; st.bposA = bposA; st.bposC = bposC;
    st.leaf = node - first_leaf; st.done = done & 0xffffu; st.go = go;
    return go >= 0;
}

// diagnostic: the exact 1-NN search of the frame kernels on caller-supplied LiDAR-frame queries (iba_debug_nn). One lane
// group of G lanes per query, the search runs to its end; out_idx = ORIGINAL point index, out_d2 = exact squared distance.
__global__ __launch_bounds__(256) void iba_nn_probe_kernel(DevProblem dp, int frame, const double* __restrict__ q, int n, int G,
                                                           uint32_t* __restrict__ out_idx, double* __restrict__ out_d2) {
    extern __shared__ __align__(16) unsigned char smem[];
    TreeNode* s_nodes = (TreeNode*)smem;
    const FrameHdr& h = dp.frames[frame];
    const uint32_t P = h.P, D = h.depth;
    for (uint32_t i = threadIdx.x; i < (1u << D) - 1u; i += blockDim.x) s_nodes[i] = dp.nodes[h.node_base + i];
    __syncthreads();
    const int e = (int)((blockIdx.x * blockDim.x + threadIdx.x) / (uint32_t)G);
    const bool act = e < n && P > 0;
    double qx = NAN, qy = NAN, qz = NAN;
    if (act) { qx = q[3 * e]; qy = q[3 * e + 1]; qz = q[3 * e + 2]; }
    DualNN st; st.bestA = INFINITY; st.bestC = INFINITY; st.bposA = kNone; st.bposC = kNone; st.leaf = 0u; st.done = 0u; st.go = -1;
#ifdef IBA_STAMPS_FINE
    st.visits = 0u;
#endif
    const uint32_t* perm = dp.perm + h.pt_base;
    if (P > 0)   // whole lane groups are active or inactive together (n is padded to the group size by the launch)
        nn_dual_step<2, true>(G, dp.xs + h.pt_base, dp.ys + h.pt_base, dp.zs + h.pt_base, dp.pts4 + h.pt_base, s_nodes, perm, P, D,
                              false, NAN, NAN, NAN, act, qx, qy, qz, st, true, true);
    if (act && (threadIdx.x & (uint32_t)(G - 1)) == 0) { out_idx[e] = st.bposC != kNone ? perm[st.bposC] : kNone; out_d2[e] = st.bestC; }
}

enum FrameMode { MODE_COST = 0, MODE_CORR = 1, MODE_ASSOC = 2, MODE_BOTH = 3 };   // BOTH = BAError + BuildProblem association in one pass

#ifdef IBA_STAMPS   // diagnostic build only: per-phase shader-clock deltas of thread 0 into partial slots 56..63
#define IBA_STAMP(i) do { if (threadIdx.x == 0) { stamp_t[i] = __builtin_readcyclecounter(); } } while (0)
#elif defined(IBA_STOP_AFTER)   // diagnostic: cut the kernel short after a phase (results are garbage) to attribute time / instructions
#define IBA_STAMP(i) do { if ((i) == IBA_STOP_AFTER) return; } while (0)
#else
#define IBA_STAMP(i) do { } while (0)
#endif

// K7: hand-eye consistency term of every (candidate, frame) pair (iba_global.cpp:264-276); one lane each.
// The frame kernel adds he[b][f] only for frames that pass the corrset test.
__global__ __launch_bounds__(64) void iba_he_kernel(DevProblem dp, const Cand* __restrict__ cands, int B, double* __restrict__ he) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B * dp.n_frames) return;
    const int b = i / dp.n_frames, f = i % dp.n_frames;
    const FrameHdr& h = dp.frames[f];
    const Cand& cd = cands[b];
    double val = 0.0;
    if (h.he_valid) {
        double C1R[9], C1t[3], C2R[9], C2t[3];
        const double* Tl = h.Tl_next; const double* Tc = h.Tc_next;
        for (int r = 0; r < 3; ++r) {
            for (int cc = 0; cc < 3; ++cc) {
                C1R[r * 3 + cc] = (cd.R[r * 3 + 0] * Tl[0 * 4 + cc] + cd.R[r * 3 + 1] * Tl[1 * 4 + cc]) + cd.R[r * 3 + 2] * Tl[2 * 4 + cc];
                C2R[r * 3 + cc] = (Tc[r * 4 + 0] * cd.R[0 * 3 + cc] + Tc[r * 4 + 1] * cd.R[1 * 3 + cc]) + Tc[r * 4 + 2] * cd.R[2 * 3 + cc];
            }
            C1t[r] = ((cd.R[r * 3 + 0] * Tl[3] + cd.R[r * 3 + 1] * Tl[7]) + cd.R[r * 3 + 2] * Tl[11]) + cd.t[r];
            C2t[r] = ((Tc[r * 4 + 0] * cd.t[0] + Tc[r * 4 + 1] * cd.t[1]) + Tc[r * 4 + 2] * cd.t[2]) + Tc[r * 4 + 3] * cd.s;
        }
        double l1[6], l2[6];
        dev_se3log(C1R, C1t, l1); dev_se3log(C2R, C2t, l2);
        double ss = 0;
        for (int k = 0; k < 6; ++k) ss += (l1[k] - l2[k]) * (l1[k] - l2[k]);
        val = sqrt(ss);
    }
    he[i] = val;
}

// ordered (by keypoint id) append of the keypoints with `want` set to s_list; two barriers; n3 stays wave-uniform
__device__ __forceinline__ void ordered_append(bool want, uint32_t k, uint32_t& n3, uint32_t* s_list, uint32_t* s_wcnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(want);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(bal);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (int w = 0; w < kWaves; ++w) { const uint32_t cw = s_wcnt[w]; total += cw; if (w < wave) before += cw; }
    if (want) s_list[n3 + before + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull))] = k;
    n3 += total;
    __syncthreads();
}

// number of set flags in the block, known to every thread (one ballot per wave + 16 LDS words)
__device__ __forceinline__ uint32_t block_count(uint32_t my_count_wave_uniform, uint32_t* s_wcnt) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_wcnt[wave] = my_count_wave_uniform;
    __syncthreads();
    uint32_t total = 0;
    for (int w = 0; w < kWaves; ++w) total += s_wcnt[w];
    __syncthreads();
    return total;
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

template <int MODE, bool SCAN_LDS>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(IBA_FRAME_WAVES, IBA_FRAME_WAVES))) void iba_frame_kernel(KArgs ka_by_value, const Cand* __restrict__ cands, int B,
                                                             double* __restrict__ frame_partials, uint32_t* __restrict__ corr_out,
                                                             uint2* __restrict__ assoc_out, int nrec, const double* __restrict__ he,
                                                             uint4* __restrict__ flist, uint32_t* __restrict__ fcount, int flist_stride) {
    extern __shared__ __align__(16) unsigned char smem[];
    // The three parameter blocks (about 100 scalar registers' worth) are read from the kernarg segment — constant address
    // space, scalar loads — where they are used, and the pointer is passed through an empty asm at every phase boundary so
    // that the loads of a phase cannot be merged with, and kept alive since, those of the kernel entry. Left to itself
    // the compiler loads everything up front and spills 169 SGPRs into VGPR lanes (822 v_readlane / v_writelane in the
    // fused kernel); this way it is 99 and 352. (The same treatment of the frame header and the candidate constants, which
    // sit behind ordinary global pointers, made it worse: their fields are copied into locals at the top anyway.)
    typedef __attribute__((address_space(4))) const KArgs KArgsC;
    KArgsC* ka = (KArgsC*)__builtin_amdgcn_kernarg_segment_ptr();   // ka_by_value is the first argument: offset 0
    (void)ka_by_value;
#define dp (ka->dp)
#define prm (ka->prm)
#define lay (ka->lay)
#define IBA_RELOAD_AT(i) do { if ((IBA_RELOAD_MASK >> (i)) & 1) asm volatile("" : "+s"(ka)); } while (0)
    typedef typename std::conditional<SCAN_LDS, uint16_t, uint32_t>::type CandT;   // LDS mode implies P < 65536
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nf = dp.n_frames;
    const int per_xcd = (nf + 7) / 8;
    const int xcd = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int f = xcd + 8 * (jj / B), b = jj % B;
    if (f >= nf || jj / B >= per_xcd) return;
    const FrameHdr& h = dp.frames[f];
    const Cand& cd = cands[b];
    double* part = frame_partials + ((size_t)b * nrec + f) * kPartialStride;

    float* s_xs = (float*)smem; float* s_ys = s_xs + lay.scan_stride; float* s_zs = s_ys + lay.scan_stride;
    unsigned long long* s_best_d2 = (unsigned long long*)(smem + lay.off_best_d2);
    uint32_t* s_best_idx = (uint32_t*)(smem + lay.off_best_idx);
    TreeNode* s_nodes = (TreeNode*)(smem + lay.off_nodes);
    uint32_t* s_bitmap = (uint32_t*)(smem + lay.off_bitmap);
    uint16_t* s_cstart = (uint16_t*)(smem + lay.off_cstart);
    double* s_red = (double*)(smem + lay.off_red);            // kWaves * 4 doubles / u64
    double* s_rel = s_red + kWaves * 4;                       // kMaxCovis * 12 doubles: relative poses of the covisible KFs
    uint32_t* s_wcnt = (uint32_t*)(s_rel + kMaxCovis * 12);
    uint32_t* s_misc = s_wcnt + kWaves;                       // [0] candidate count, [1] overflow flag
    CandT* s_cand = (CandT*)(smem + lay.off_cand);
    uint32_t* s_list = (uint32_t*)(smem + lay.off_best_d2);   // aliases best_d2 after phase 2

    const uint32_t P = h.P, Ppad = h.Ppad, K = h.K, D = h.depth;
    const float* gxs = dp.xs + h.pt_base; const float* gys = dp.ys + h.pt_base; const float* gzs = dp.zs + h.pt_base;

#ifdef IBA_STAMPS
    unsigned long long stamp_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    IBA_STAMP(0); IBA_RELOAD_AT(0);
    // ---- phase 0: LDS init ----
    // the first (usually only) element per thread of each static table is fetched before anything is stored, so the
    // four global-load latencies overlap instead of queueing behind each other's LDS stores
    const uint32_t nnodes = (1u << D) - 1u;
    const uint32_t nbw = (h.gw * h.gh + 31u) >> 5;
    const uint32_t ncs = h.gwc * h.ghc + 1u;
    const uint32_t ut = (uint32_t)tid;
    TreeNode nv = TreeNode{0.f, 0u}; uint32_t bv = 0u, cv0 = 0u, cv1 = 0u; double rv = 0.0;
    if (ut < nnodes) nv = dp.nodes[h.node_base + ut];
    if (ut < nbw) bv = dp.bitmap[h.bitmap_base + ut];
    if (ut < ncs) cv0 = dp.coarse_start[h.coarse_base + ut];
    if (ut + kThreads < ncs) cv1 = dp.coarse_start[h.coarse_base + ut + kThreads];
    if (ut < h.n_slots * 12u) rv = dp.slots[h.slot_base + ut / 12].rel[ut % 12];
    for (uint32_t i = tid; i < K; i += kThreads) { s_best_d2[i] = ~0ull; s_best_idx[i] = kNone; }
    if (ut < nnodes) s_nodes[ut] = nv;
    if (ut < nbw) s_bitmap[ut] = bv;
    if (ut < ncs) s_cstart[ut] = (uint16_t)cv0;
    if (ut + kThreads < ncs) s_cstart[ut + kThreads] = (uint16_t)cv1;
    for (uint32_t i = ut + kThreads; i < nnodes; i += kThreads) s_nodes[i] = dp.nodes[h.node_base + i];
    for (uint32_t i = ut + kThreads; i < nbw; i += kThreads) s_bitmap[i] = dp.bitmap[h.bitmap_base + i];
    for (uint32_t i = ut + 2u * kThreads; i < ncs; i += kThreads) s_cstart[i] = (uint16_t)dp.coarse_start[h.coarse_base + i];
    if (ut < h.n_slots * 12u) s_rel[ut] = rv;
    if (tid < 4) s_misc[tid] = 0u;
    __syncthreads();

    FrameCtx c;
    c.xs = SCAN_LDS ? s_xs : gxs; c.ys = SCAN_LDS ? s_ys : gys; c.zs = SCAN_LDS ? s_zs : gzs;
    c.nodes = s_nodes; c.bitmap = s_bitmap; c.best_d2 = s_best_d2; c.best_idx = s_best_idx;
    c.cstart = s_cstart; c.gwc = (int)h.gwc; c.crec = dp.crec + h.kp_base;
    c.perm = dp.perm + h.pt_base;
    c.p4 = SCAN_LDS ? nullptr : dp.pts4 + h.pt_base;
    c.gw = (int)h.gw; c.gh = (int)h.gh; c.margin = (float)prm.grid_margin; c.gate2 = prm.gate2;
    c.fx = h.fx; c.cx = h.cx; c.cy = h.cy; c.W = h.W; c.H = h.H;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.R[i] = cd.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) c.t[i] = cd.t[i];

    IBA_STAMP(1); IBA_RELOAD_AT(1);
    // ---- phase 1a: stream the scan once (16 B/lane), keep it in LDS, and PRE-CULL in float32:
    //      a point can only matter if it may project within ~1.5 px of a keypoint. The f32 projection errs by
    //      < 0.3 px for depth > 0.1 m; the reject bitmap is dilated by max_pixel_dist + 0.45 px, so a clear bit
    //      proves that the exact (f64) test could not produce a match. Everything that survives (~6 %) — and
    //      everything the f32 test cannot decide (|depth| <= 0.1 m) — is queued for the exact f64 path below.
    const uint32_t cand_cap = lay.cand_cap;
    // ---- phase 0.5: conservative frustum test of the static per-chunk boxes. A LiDAR sweeps 180-360 degrees, the camera
    //      sees ~80 x 30: most chunks of kChunk consecutive tree positions (a few neighbouring kd leaves) cannot project
    //      into the image for this candidate and are neither loaded nor projected. The five half-spaces (left, right, top,
    //      bottom with an 8 px margin, depth > -0.2 m) are the pre-cull's own acceptance region widened far beyond its
    //      float error, moved into the LiDAR frame (n = R^T a, d = a . t); a box is dropped only if its farthest corner
    //      violates one of them by more than 1e-3 relative.
    uint32_t* s_vis = (uint32_t*)(smem + lay.off_vis);
    {
        const uint32_t nchunks = (P + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;
        const float m = 8.0f;
        const float A[5][3] = {{(float)c.fx, 0.f, (float)c.cx + m}, {-(float)c.fx, 0.f, (float)c.W + m - (float)c.cx},
                               {0.f, (float)c.fx, (float)c.cy + m}, {0.f, -(float)c.fx, (float)c.H + m - (float)c.cy}, {0.f, 0.f, 1.f}};
        float N[5][3], Dd[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int j = 0; j < 3; ++j) N[q][j] = (float)c.R[0 * 3 + j] * A[q][0] + (float)c.R[1 * 3 + j] * A[q][1] + (float)c.R[2 * 3 + j] * A[q][2];
            Dd[q] = (float)c.t[0] * A[q][0] + (float)c.t[1] * A[q][1] + (float)c.t[2] * A[q][2] + (q == 4 ? 0.2f : 0.f);
        }
        const float* boxes = dp.chunk_box + 8 * h.box_base;
        for (uint32_t ch0 = 0; ch0 < nchunks; ch0 += kThreads) {
            const uint32_t ch = ch0 + (uint32_t)tid;
            bool vis = false;
            if (ch < nchunks) {
                const float* b = boxes + 8 * (size_t)ch;
                const float lo3[3] = {b[0], b[1], b[2]}, hi3[3] = {b[4], b[5], b[6]};
                vis = true;
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    float smax = Dd[q], mag = fabsf(Dd[q]);
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        smax += N[q][j] >= 0.f ? N[q][j] * hi3[j] : N[q][j] * lo3[j];
                        mag += fabsf(N[q][j]) * fmaxf(fabsf(lo3[j]), fabsf(hi3[j]));
                    }
                    vis = vis && !(smax < -1e-3f * mag - 1e-6f);   // NaN boxes (empty chunk) compare false -> kept, harmless
                }
            }
            const unsigned long long bal = __ballot(vis);
#ifdef IBA_STAMPS
            if (lane == 0) { atomicAdd(&g_dbg[3], (unsigned long long)__popcll(bal)); if (wave == 0 && ch0 == 0) atomicAdd(&g_dbg[4], (unsigned long long)nchunks); }
#endif
            if (lane == 0) { s_vis[(ch0 >> 5) + 2u * (uint32_t)wave] = (uint32_t)bal; s_vis[(ch0 >> 5) + 2u * (uint32_t)wave + 1u] = (uint32_t)(bal >> 32); }
        }
    }
    __syncthreads();
    // visible chunks, compacted in order into a u16 list right behind the bit words (wave 0: one bit word per lane and pass)
    const uint32_t nchunks_all = (P + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;
    const uint32_t nvis_words = (nchunks_all + 31u) >> 5;
    uint16_t* s_vlist = (uint16_t*)(s_vis + lay.vis_words);
    if (!SCAN_LDS) {
        if (wave == 0) {
            uint32_t run = 0;
            for (uint32_t w0 = 0; w0 < nvis_words; w0 += 64u) {
                const uint32_t wi = w0 + (uint32_t)lane;
                uint32_t bits = wi < nvis_words ? s_vis[wi] : 0u;
                if (wi == nvis_words - 1u && (nchunks_all & 31u)) bits &= (1u << (nchunks_all & 31u)) - 1u;   // stale bits beyond the last chunk
                const uint32_t cntb = (uint32_t)__popc(bits);
                uint32_t incl = cntb;   // inclusive prefix over the wave
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) { const uint32_t o = __shfl_up(incl, off); if (lane >= off) incl += o; }
                uint32_t at = run + incl - cntb;
                while (bits) { const uint32_t bpos = (uint32_t)__ffs((int)bits) - 1u; bits &= bits - 1u; s_vlist[at++] = (uint16_t)(wi * 32u + bpos); }
                run += __shfl(incl, 63);
            }
            if (lane == 0) s_misc[1 + 2] = run;   // s_misc[3]: number of visible chunks (the NN rounds re-zero their counters later)
        }
        __syncthreads();
    }
    const uint32_t n_vis = SCAN_LDS ? 0u : s_misc[3];
    {
        const float r0 = (float)c.R[0], r1 = (float)c.R[1], r2 = (float)c.R[2], r3 = (float)c.R[3], r4 = (float)c.R[4], r5 = (float)c.R[5],
                    r6 = (float)c.R[6], r7 = (float)c.R[7], r8 = (float)c.R[8], t0 = (float)c.t[0], t1 = (float)c.t[1], t2 = (float)c.t[2];
        const float fxf = (float)c.fx, cxf = (float)c.cx, cyf = (float)c.cy, Wf = (float)c.W + 1.0f, Hf = (float)c.H + 1.0f;
        // SCAN_LDS: every point, in tree order (it has to be staged). Otherwise: the 4-point groups of the visible chunks only
        // (16 groups per chunk), so the number of dependent load rounds shrinks with the culling.
        const uint32_t n_groups = SCAN_LDS ? (Ppad + 3u) / 4u : n_vis * (uint32_t)(kChunk / 4);
        const uint32_t n_iter = (n_groups + kThreads - 1u) / kThreads;
        const float qn = __builtin_nanf("");
        const float4 nan4 = make_float4(qn, qn, qn, qn);
        auto group_base = [&](uint32_t g) -> uint32_t {   // first tree position of group g, or kNone
            if (g >= n_groups) return kNone;
            const uint32_t b4 = SCAN_LDS ? g * 4u : (uint32_t)s_vlist[g / (uint32_t)(kChunk / 4)] * (uint32_t)kChunk + (g % (uint32_t)(kChunk / 4)) * 4u;
            return b4 < Ppad ? b4 : kNone;
        };
        float4 X = nan4, Y = nan4, Z = nan4;   // software pipeline: the next 16-byte loads are in flight while 4 points are tested
        uint32_t base = group_base((uint32_t)tid);
        if (base != kNone) { X = *(const float4*)(gxs + base); Y = *(const float4*)(gys + base); Z = *(const float4*)(gzs + base); }
        for (uint32_t it = 0; it < n_iter; ++it) {
            const uint32_t nbase = group_base((uint32_t)tid + (it + 1u) * kThreads);
            float4 Xn = nan4, Yn = nan4, Zn = nan4;
            if (nbase != kNone) { Xn = *(const float4*)(gxs + nbase); Yn = *(const float4*)(gys + nbase); Zn = *(const float4*)(gzs + nbase); }
            if (SCAN_LDS && base != kNone) { *(float4*)(s_xs + base) = X; *(float4*)(s_ys + base) = Y; *(float4*)(s_zs + base) = Z; }
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
                    if (uf > -1.0f && uf < Wf && vf > -1.0f && vf < Hf) {
                        const uint32_t cell = (uint32_t)grid_cell(vf, c.gh) * (uint32_t)c.gw + (uint32_t)grid_cell(uf, c.gw);
                        pass[j] = (s_bitmap[cell >> 5] >> (cell & 31)) & 1u;
                    }
                } else if (zc > -0.1f) pass[j] = true;   // undecidable in f32 (NaN padding fails both tests): exact path decides
            }
            // one LDS atomic per wave per iteration reserves queue slots for all four points
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
                    if (off[j] < cand_cap) s_cand[off[j]] = (CandT)(base + j);   // pass[j] implies base != kNone (NaN coordinates never pass)
                    else {   // queue full: exact path inline, full rescan in phase 2 (speed only)
                        double u, v;
                        if (project_uv(c, px[j], py[j], pz[j], u, v)) grid_match<1>(c, u, v, base + j);
                        s_misc[1] = 1u;
                    }
                }
            }
            X = Xn; Y = Yn; Z = Zn; base = nbase;
        }
    }
    __syncthreads();
    IBA_STAMP(7); IBA_RELOAD_AT(7);
    // ---- phase 1b: exact f64 projection + FOV test of the queued points, keypoint grid lookup,
    //      ds_min_u64 on the keypoint's best d^2 ----
    const uint32_t ncand = min(s_misc[0], cand_cap);
#ifdef IBA_STAMPS
    if (tid == 0) { atomicAdd(&g_dbg[1], (unsigned long long)s_misc[0]); atomicAdd(&g_dbg[2], 1ull); }
#endif
    const bool overflow = s_misc[1] != 0u;
    // the first kHitSlots queue entries of each lane keep their (<= 2) hits in registers, so the tie pass needs no second
    // grid walk; later entries (queue longer than kHitSlots blocks) take the re-walk path
    constexpr int kHitSlots = IBA_HIT_SLOTS;   // ~400 queued points per block at the C2 shape (2-px reject bitmap)
    Hits hh[kHitSlots];
    bool redo = false;
#pragma unroll
    for (int q = 0; q < kHitSlots; ++q) {
        hh[q].n = 0; hh[q].k0 = hh[q].k1 = 0u; hh[q].d0 = hh[q].d1 = 0ull;
        if ((uint32_t)tid + (uint32_t)q * kThreads < ncand) {
            const uint32_t pos = (uint32_t)s_cand[tid + q * kThreads];
            double u, v;
            if (project_pos<!SCAN_LDS>(c, pos, u, v)) hh[q] = grid_match_rec(c, u, v);
        }
        redo |= hh[q].n > 2;
    }
    for (uint32_t i = (uint32_t)tid + (uint32_t)kHitSlots * kThreads; i < ncand; i += kThreads) {
        const uint32_t pos = (uint32_t)s_cand[i];
        double u, v;
        if (project_pos<!SCAN_LDS>(c, pos, u, v)) redo |= grid_match<1>(c, u, v, pos);
    }
    __syncthreads();
    IBA_STAMP(2); IBA_RELOAD_AT(2);
    // ---- phase 2: the winner of each keypoint records its original index; exact ties -> lowest index ----
#pragma unroll
    for (int q = 0; q < kHitSlots; ++q) {
        if (hh[q].n > 0) {
            const bool w0 = s_best_d2[hh[q].k0] == hh[q].d0;
            const bool w1 = hh[q].n > 1 && s_best_d2[hh[q].k1] == hh[q].d1;
            if (w0 || w1) {
                const uint32_t orig = c.perm[(uint32_t)s_cand[tid + q * kThreads]];
                if (w0) atomicMin(&s_best_idx[hh[q].k0], orig);
                if (w1) atomicMin(&s_best_idx[hh[q].k1], orig);
            }
        }
    }
    if (redo) {   // rare: a point within reach of > 2 keypoints, or a queue longer than two blocks: walk the grid again
        for (uint32_t i = tid; i < ncand; i += kThreads) {
            const uint32_t pos = (uint32_t)s_cand[i];
            double u, v;
            if (project_pos<!SCAN_LDS>(c, pos, u, v)) grid_match<2>(c, u, v, pos);
        }
    }
    if (overflow) {
        for (uint32_t pos = tid; pos < P; pos += kThreads) {
            double u, v;
            if (project_pos<!SCAN_LDS>(c, pos, u, v)) grid_match<2>(c, u, v, pos);
        }
    }
    __syncthreads();

    if (MODE == MODE_CORR) {   // dense dump: corr_out[kp_base + k] = original point index or kNone
        for (uint32_t k = tid; k < K; k += kThreads) corr_out[h.kp_base + k] = s_best_idx[k];
        return;
    }

    IBA_STAMP(3); IBA_RELOAD_AT(3);
    const float4* kp_mp = dp.kp_mp + h.kp_base;
    const uint32_t* inv_perm = dp.inv_perm + h.pt_base;
    const uint32_t Kceil = (K + kThreads - 1) / kThreads * kThreads;
    const double s = cd.s;
    uint32_t n3 = 0;   // wave-uniform running length of s_list

    // ---- phase 3: corrset.size() ----
    uint32_t n_corr;
    {
        uint32_t wc = 0;
        for (uint32_t k = tid; k < Kceil; k += kThreads) wc += (uint32_t)__popcll(__ballot(k < K && s_best_idx[k] != kNone));
        n_corr = block_count(wc, s_wcnt);
    }

    // local-plane records: memoised per scan point (plane_cache = 1) or refitted for exactly the points this
    // evaluation needs, as the reference does (plane_cache = 0), into this candidate's private scratch
    const bool cached = prm.plane_cache != 0;
    const size_t scr_off = (size_t)(dp.scratch_slot_base + b) * (size_t)dp.n_pt_total + h.pt_base;
    const PlaneRec* planes_cost = cached ? dp.plane_cost + h.pt_base : dp.scratch_cost + scr_off;
    const PlaneRec* planes_local = cached ? dp.plane_local + h.pt_base : dp.scratch_local + scr_off;
    uint32_t* s_bpos = s_list + K;   // NN results of the 3d-3d work list (second half of the aliased region)
    auto fit_points = [&](const uint32_t* list, uint32_t count, double r2, int max_pts, PlaneRec* dst, const uint32_t* same_as = nullptr) {
        for (uint32_t i = (uint32_t)wave; i < count; i += kWaves) {   // one wave per point: kNN + covariance + eigen
            const uint32_t pos = list[i];
            if (pos == kNone || (same_as && same_as[i] == pos)) continue;   // same_as: this item's record already exists elsewhere
            const PlaneRec rec = plane_fit_wave(c.xs, c.ys, c.zs, s_nodes, P, D, pos, r2, max_pts);
            if (lane == 0) dst[pos] = rec;
        }
        __syncthreads();
    };

    // ---- exact 1-NN of a work list in rounds (nn_dual_step). Round 0: every item descends and scans its first leaf —
    // uniform work, and 72 % of the queries end there. The unfinished ones park 80 B of state in LDS (the keypoint grid's and
    // the candidate queue's storage, dead by now) and are re-spread over the block with more lanes each; a round visits
    // one more leaf per query until few enough are left to run to the end. Slot order is irrelevant (results go to
    // per-item arrays), so the compaction is one LDS atomic per unfinished query.
    //   load(i, actA, actC, ax, ay, az, qx, qy, qz): which queries item i has, and their coordinates
    //   store(i, actA, actC, st): the finished search of item i
    struct HardState { double a[3], c[3], bestA, bestC; uint32_t item_go, leaf_done, bposA, bposC; };   // item | go << 24 | actA << 30 | actC << 31
    auto nn_rounds = [&](auto which_tag, uint32_t n_list, auto&& load, auto&& store) {
        constexpr int WHICH = decltype(which_tag)::value;
        HardState* s_hard = (HardState*)(smem + lay.off_bitmap);
        const uint32_t cap_hard = min((lay.total - lay.off_bitmap) / (uint32_t)sizeof(HardState), (uint32_t)kThreads);   // <= one pass per round
        uint32_t n_items = n_list; bool fresh = true; int round = 0;
        for (;;) {
            int G;
            if (fresh) {
                // lanes per query of the first round: as many as fit (a second pass for a few items over a power-of-two
                // boundary was measured and is slower than half the lanes for everybody)
                const uint32_t slack = (uint32_t)IBA_NN_G_SLACK;
                G = (IBA_NN_FRESH_GMAX >= 8 && n_items * 8u <= (uint32_t)kThreads + 8u * slack) ? 8 : ((IBA_NN_FRESH_GMAX >= 4 && n_items * 4u <= (uint32_t)kThreads + 4u * slack) ? 4 : ((IBA_NN_FRESH_GMAX >= 2 && n_items * 2u <= (uint32_t)kThreads + 2u * slack) ? 2 : 1));
            }
            else { G = IBA_NN_RESUME_GMAX; while ((uint32_t)G * n_items > (uint32_t)kThreads) G >>= 1; }
            const uint32_t per_it = (uint32_t)kThreads / (uint32_t)G;
            uint32_t* cnt = s_misc + 2 + (round & 1);
            if (tid == 0) *cnt = 0u;
            __syncthreads();
            for (uint32_t base = 0; base < n_items; base += per_it) {
                const uint32_t e = base + (uint32_t)tid / (uint32_t)G;
                uint32_t i = e; bool actA = false, actC = false;
                double ax = NAN, ay = NAN, az = NAN, qx = NAN, qy = NAN, qz = NAN;   // an absent query has NaN coordinates (nn_dual_step)
                DualNN st; st.bestA = INFINITY; st.bestC = INFINITY; st.bposA = kNone; st.bposC = kNone; st.leaf = 0u; st.done = 0u; st.go = -1;
                if (e < n_items) {
                    if (fresh) {
                        load(i, actA, actC, ax, ay, az, qx, qy, qz);
                        if (!actA) { ax = NAN; ay = NAN; az = NAN; }
                        if (!actC) { qx = NAN; qy = NAN; qz = NAN; }
                    } else {
                        const HardState hs = s_hard[e];
                        i = hs.item_go & 0xffffffu; st.go = (int)((hs.item_go >> 24) & 31u); actA = (hs.item_go >> 30) & 1u; actC = hs.item_go >> 31;
                        st.leaf = hs.leaf_done & 0xffffu; st.done = hs.leaf_done >> 16; st.bposA = hs.bposA; st.bposC = hs.bposC;
                        ax = hs.a[0]; ay = hs.a[1]; az = hs.a[2]; qx = hs.c[0]; qy = hs.c[1]; qz = hs.c[2];
                        st.bestA = hs.bestA; st.bestC = hs.bestC;   // parked with the state: no gather + recomputation on the way back in
                    }
                }
                __syncthreads();   // every state of this pass is in registers: its slot may be overwritten
                if (e < n_items) {
                    if (actA || actC) {
                        bool fr = fresh, to_end = kNNToEnd || (!fresh && n_items <= (uint32_t)kNNEndAt);
                        for (;;) {
                            const bool more = nn_dual_step<WHICH, !SCAN_LDS>(G, c.xs, c.ys, c.zs, c.p4, s_nodes, c.perm, P, D, actA, ax, ay, az, actC, qx, qy, qz, st, fr, to_end);
                            if (!more) break;
                            uint32_t slot = 0u;
                            if ((tid & (G - 1)) == 0) slot = atomicAdd(cnt, 1u);
                            slot = __shfl(slot, lane & ~(G - 1));
                            if (slot < cap_hard) {   // park: one more leaf next round
                                if ((tid & (G - 1)) == 0) {
                                    HardState hs; hs.a[0] = ax; hs.a[1] = ay; hs.a[2] = az; hs.c[0] = qx; hs.c[1] = qy; hs.c[2] = qz;
                                    hs.item_go = i | ((uint32_t)st.go << 24) | ((actA ? 1u : 0u) << 30) | ((actC ? 1u : 0u) << 31);
                                    hs.leaf_done = st.leaf | (st.done << 16); hs.bposA = st.bposA; hs.bposC = st.bposC; hs.bestA = st.bestA; hs.bestC = st.bestC;
                                    s_hard[slot] = hs;
                                }
                                i = kNone;
                                break;
                            }
                            fr = false; to_end = true;   // no room to park it: finish right here
                        }
                    }
                    if (i != kNone && (tid & (G - 1)) == 0) store(i, actA, actC, st);
                }
                __syncthreads();
            }
            n_items = min(*cnt, cap_hard);
#ifdef IBA_STOP_NN_ROUND
            if (round == IBA_STOP_NN_ROUND) return;   // diagnostic (fused mode): time attribution
#endif
            if (n_items == 0u) break;
            fresh = false; ++round;
        }
    };

    if (MODE == MODE_BOTH) {
        // ================= fused BAError + BuildProblem association: one work list, one kd traversal per MapPoint =================
        // (the dense per-keypoint association rows are not written in this mode: the factor kernel is fed by the block list)
        uint4* fl = flist + ((size_t)b * nf + f) * (size_t)flist_stride;
        const bool usedA = !((int)n_corr < prm.num_min_corr);        // iba_local.cpp:192
        const bool usedC = !((int)n_corr < prm.num_min_corr_cost);   // iba_global.cpp:203
        uint32_t* s_nnC = s_list + K;        // per list item: cost-path NN (aliases the 2nd half of best_d2)
        uint32_t* s_nnA = s_list + 2 * K;    // per list item: association-path NN / flags (aliases best_idx: only after its last reader)
        // work list: keypoints with a correspondence that own a MapPoint and/or a covisible match; entry = k | w << 16
        for (uint32_t k = tid; k < Kceil; k += kThreads) {
            const bool valid = k < K && s_best_idx[k] != kNone;
            const int w = k < K ? (int)kp_mp[k].w : 0;
            const bool want = (usedA || usedC) && valid && w != 0;
            ordered_append(want, k | ((uint32_t)w << 16), n3, s_list, s_wcnt);   // bits 16,17: flags; 18..27: covisible-slot mask
        }
        // matched scan point of every item (needed by the residual loop, the plane pass and, in refit mode, the fits)
        for (uint32_t i = tid; i < n3; i += kThreads) s_nnC[i] = inv_perm[s_best_idx[s_list[i] & 0xffffu]];
        __syncthreads();   // last read of s_best_idx: its storage now carries s_nnA
        if (!cached && usedA) {   // ComputeLocalNeighbor at the matched point of every association-eligible item
            for (uint32_t i = tid; i < n3; i += kThreads) s_nnA[i] = (((s_list[i] >> 16) & 3u) == 3u) ? s_nnC[i] : kNone;
            __syncthreads();
            fit_points(s_nnA, n3, prm.neigh_radius2, prm.neigh_max_pts, dp.scratch_local + scr_off);
        }
        double sum2d = 0.0, sum3d = 0.0;
        uint32_t c2 = 0, v2 = 0, c3 = 0, v3 = 0, vpl = 0, vpt = 0;
        // association: local plane at the matched point (iba_local.cpp:207-231)
        for (uint32_t i = tid; i < n3; i += kThreads) {
            const uint32_t k = s_list[i] & 0xffffu;
            uint32_t flagA = kNone;
            if (usedA && ((s_list[i] >> 16) & 3u) == 3u) {
                const uint32_t pos = s_nnC[i];
                const PlaneRec rec = planes_local[pos];
                uint2 a = make_uint2(kNone, kNone);
                const bool neigh_ok = !(rec.k < prm.neigh_min_pts || rec.far_d2 < prm.local_min_diff_dist2);
                if (neigh_ok && rec.reg_sum / (double)(rec.k - 1) < prm.local_norm_reg_threshold) a.x = pos;
                if (neigh_ok) flagA = 0u;
                fl[i] = make_uint4(k, a.x, kNone, 0u);   // .z (the 3d-3d block) follows once the neighbour is known
            } else fl[i] = make_uint4(k, kNone, kNone, 0u);
            s_nnA[i] = flagA;
        }
        // K6: 3d-2d covisible reprojection residuals (iba_global.cpp:291-328): only the slots whose match bit is set
        if (usedC) {
            for (uint32_t i = tid; i < n3; i += kThreads) {
                uint32_t mask = (s_list[i] >> 18) & 0x3ffu;
                if (!mask) continue;
                const uint32_t k = s_list[i] & 0xffffu, pos = s_nnC[i];
                float xf_, yf_, zf_; load_pt<!SCAN_LDS>(c, pos, xf_, yf_, zf_);
                const double x = (double)xf_, y = (double)yf_, z = (double)zf_;
                const double p0x = ((c.R[0] * x + c.R[1] * y) + c.R[2] * z) + c.t[0];
                const double p0y = ((c.R[3] * x + c.R[4] * y) + c.R[5] * z) + c.t[1];
                const double p0z = ((c.R[6] * x + c.R[7] * y) + c.R[8] * z) + c.t[2];
                const float2* mrow = dp.match_uv + h.match_base + k;
                float2 mm = mrow[(size_t)(__ffs((int)mask) - 1) * K];
                while (mask) {
                    const uint32_t sl = (uint32_t)__ffs((int)mask) - 1u;
                    mask &= mask - 1u;
                    const float2 cur = mm;
                    if (mask) mm = mrow[(size_t)(__ffs((int)mask) - 1) * K];   // next match is in flight during the arithmetic
                    const double* rel = s_rel + sl * 12;
                    const double p1x = ((rel[0] * p0x + rel[1] * p0y) + rel[2] * p0z) + rel[3] * s;
                    const double p1y = ((rel[4] * p0x + rel[5] * p0y) + rel[6] * p0z) + rel[7] * s;
                    const double p1z = ((rel[8] * p0x + rel[9] * p0y) + rel[10] * p0z) + rel[11] * s;
                    const double ou = h.fx * p1x / p1z + h.cx;
                    const double ov = h.fy * p1y / p1z + h.cy;
                    if (!(ou >= 0 && ou < h.W && ov >= 0 && ov < h.H)) continue;
                    const double eu = ou - (double)cur.x, ev = ov - (double)cur.y;
                    const double dist = sqrt(eu * eu + ev * ev);
                    if (dist < prm.corr_3d_2d_threshold) { sum2d += dist; ++v2; }
                    ++c2;
                }
            }
        }
        __syncthreads();
        IBA_STAMP(4); IBA_RELOAD_AT(4);
        // the two MapPoint -> LiDAR-frame queries (iba_local.cpp:238-239,282 and iba_global.cpp:231-234)
        auto queries = [&](uint32_t k, double& ax, double& ay, double& az, double& qx, double& qy, double& qz) {
            // the ~56 scalars of the two transforms are re-read from constant memory at each call site instead of being
            // carried (spilled to VGPR lanes) across the 1-NN phase: see the note on the kernarg pointer at the top
            typedef __attribute__((address_space(4))) const FrameHdr FrameHdrC;
            typedef __attribute__((address_space(4))) const Cand CandC;
            FrameHdrC* hq = (FrameHdrC*)&h; CandC* cq = (CandC*)&cd;
            asm volatile("" : "+s"(hq), "+s"(cq));
#define h (*hq)
#define cd (*cq)
            const double s = cd.s;
            const double ts0 = h.Tcw[3] * s, ts1 = h.Tcw[7] * s, ts2 = h.Tcw[11] * s;
            const float4 mp = kp_mp[k];
            const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
            const double mx = ((h.Tcw[0] * w0 + h.Tcw[1] * w1) + h.Tcw[2] * w2) + h.Tcw[3];
            const double my = ((h.Tcw[4] * w0 + h.Tcw[5] * w1) + h.Tcw[6] * w2) + h.Tcw[7];
            const double mz = ((h.Tcw[8] * w0 + h.Tcw[9] * w1) + h.Tcw[10] * w2) + h.Tcw[11];
            const double sx = mx * s, sy = my * s, sz = mz * s;
            ax = ((cd.Ri[0] * sx + cd.Ri[1] * sy) + cd.Ri[2] * sz) + cd.ti[0];
            ay = ((cd.Ri[3] * sx + cd.Ri[4] * sy) + cd.Ri[5] * sz) + cd.ti[1];
            az = ((cd.Ri[6] * sx + cd.Ri[7] * sy) + cd.Ri[8] * sz) + cd.ti[2];
            const float m0 = mp.x * cd.s32, m1 = mp.y * cd.s32, m2 = mp.z * cd.s32;   // CV_32F product (:232)
            const double a0 = (double)m0, a1 = (double)m1, a2 = (double)m2;
            const double cx_ = ((h.Tcw[0] * a0 + h.Tcw[1] * a1) + h.Tcw[2] * a2) + ts0;
            const double cy_ = ((h.Tcw[4] * a0 + h.Tcw[5] * a1) + h.Tcw[6] * a2) + ts1;
            const double cz_ = ((h.Tcw[8] * a0 + h.Tcw[9] * a1) + h.Tcw[10] * a2) + ts2;
            qx = ((cd.Ri[0] * cx_ + cd.Ri[1] * cy_) + cd.Ri[2] * cz_) + cd.ti[0];
            qy = ((cd.Ri[3] * cx_ + cd.Ri[4] * cy_) + cd.Ri[5] * cz_) + cd.ti[1];
            qz = ((cd.Ri[6] * cx_ + cd.Ri[7] * cy_) + cd.Ri[8] * cz_) + cd.ti[2];
#undef h
#undef cd
        };
        nn_rounds(std::integral_constant<int, 3>(), n3,
            [&](uint32_t i, bool& actA, bool& actC, double& ax, double& ay, double& az, double& qx, double& qy, double& qz) {
                actA = s_nnA[i] != kNone;
                actC = usedC && prm.use_3d3d && ((s_list[i] >> 16) & 1u);
                if (actA || actC) queries(s_list[i] & 0xffffu, ax, ay, az, qx, qy, qz);
            },
            [&](uint32_t i, bool actA, bool actC, const DualNN& st) {
                s_nnC[i] = actC ? st.bposC : kNone;
                s_nnA[i] = (actA && !(st.bestA > prm.max_3d_dist2)) ? st.bposA : kNone;   // the association keeps its neighbour only within max_3d_dist (iba_local.cpp:289)
            });
        __syncthreads();
        IBA_STAMP(5); IBA_RELOAD_AT(5);
        if (!cached) {
            fit_points(s_nnA, n3, prm.neigh_radius2, prm.neigh_max_pts, dp.scratch_local + scr_off);
            if (prm.use_plane) {
                // The cost path's neighbour is almost always the association's (the two queries differ by 1e-7): with the
                // same radius and cap — the reference's yaml values — the record just fitted is the record wanted.
                const bool same = prm.norm_radius2 == prm.neigh_radius2 && prm.norm_max_pts == prm.neigh_max_pts;
                fit_points(s_nnC, n3, prm.norm_radius2, prm.norm_max_pts, dp.scratch_cost + scr_off, same ? s_nnA : nullptr);
                if (same) {
                    const PlaneRec* lc = dp.scratch_local + scr_off; PlaneRec* cc = dp.scratch_cost + scr_off;
                    for (uint32_t i = tid; i < n3; i += kThreads) { const uint32_t pc = s_nnC[i]; if (pc != kNone && pc == s_nnA[i]) cc[pc] = lc[pc]; }
                    __syncthreads();
                }
            }
        }
        IBA_STAMP(6); IBA_RELOAD_AT(6);
        for (uint32_t i = tid; i < n3; i += kThreads) {
            const uint32_t k = s_list[i] & 0xffffu;
            // association: kind of the 3d-3d block (pointcloud.h:699-717), dense block list for the factor kernel
            const uint32_t bA = s_nnA[i];
            uint32_t ay_ = kNone;
            if (bA != kNone) {
                const PlaneRec r2 = planes_local[bA];
                const bool state = !(r2.k < prm.neigh_min_pts || r2.far_d2 < prm.local_min_diff_dist2) &&
                                   (r2.reg_sum / (double)(r2.k - 1) < prm.local_norm_reg_threshold);
                ay_ = bA | (state ? 0x80000000u : 0u);
            }
            if (bA != kNone) fl[i].z = ay_;
            // cost: point-to-plane / point-to-point distance (iba_global.cpp:111-156, 241-249)
            const uint32_t bC = s_nnC[i];
            if (bC != kNone) {
                double ax, ay, az, qx, qy, qz; queries(k, ax, ay, az, qx, qy, qz);
                float xf_, yf_, zf_; load_pt<!SCAN_LDS>(c, bC, xf_, yf_, zf_);
                const double ex = (double)xf_ - qx, ey = (double)yf_ - qy, ez = (double)zf_ - qz;
                double dist = sqrt((ex * ex + ey * ey) + ez * ez);
                bool is_plane = false;
                if (prm.use_plane) {
                    const PlaneRec rec = planes_cost[bC];
                    if (!(rec.far_d2 < prm.min_diff_dist2) && !(rec.k < prm.norm_min_pts) &&
                        !(rec.reg_sum / (double)(rec.k - 1) > prm.norm_reg_threshold)) {
                        dist = fabs(ex * rec.nx + ey * rec.ny + ez * rec.nz);
                        is_plane = true;
                    }
                }
                if (dist < prm.corr_3d_3d_threshold) { sum3d += dist; ++v3; if (is_plane) ++vpl; else ++vpt; }
                ++c3;
            }
        }
        if (tid == 0) fcount[(size_t)b * nf + f] = usedA ? n3 : 0u;
        // K8: reduction -> record (cost slots only if this frame counts for BAError)
        {
            const double w2d = wave_sum_f64(sum2d), w3d = wave_sum_f64(sum3d);
            const unsigned long long wa = wave_sum_u64((unsigned long long)c2 | ((unsigned long long)v2 << 21) | ((unsigned long long)c3 << 42));
            const unsigned long long wb = wave_sum_u64((unsigned long long)v3 | ((unsigned long long)vpl << 21) | ((unsigned long long)vpt << 42));
            unsigned long long* s_redu = (unsigned long long*)s_red;
            __syncthreads();
            if (lane == 63) { s_red[wave * 4 + 0] = w2d; s_red[wave * 4 + 1] = w3d; s_redu[wave * 4 + 2] = wa; s_redu[wave * 4 + 3] = wb; }
            __syncthreads();
            if (tid < kPartialStride) {
                double out = 0.0;
                if (usedC) {
                    if (tid == P_SUM_3D2D || tid == P_SUM_3D3D) { for (int w = 0; w < kWaves; ++w) out += s_red[w * 4 + (tid == P_SUM_3D2D ? 0 : 1)]; }
                    else if (tid >= P_CNT_3D2D && tid <= P_VALID_PT) {
                        unsigned long long a = 0, bb = 0;
                        for (int w = 0; w < kWaves; ++w) { a += s_redu[w * 4 + 2]; bb += s_redu[w * 4 + 3]; }
                        const unsigned long long msk = (1ull << 21) - 1ull;
                        const unsigned long long vals[6] = {a & msk, (a >> 21) & msk, (a >> 42) & msk, bb & msk, (bb >> 21) & msk, (bb >> 42) & msk};
                        out = (double)vals[tid - P_CNT_3D2D];
                        if (!prm.use_3d3d && (tid == P_CNT_3D3D || tid == P_VALID_3D3D)) out = 1.0;   // iba_global.cpp:214-220
                    }
                    else if (tid == P_FRAMES) out = 1.0;
                    else if (tid == P_NCORR) out = (double)n_corr;
                    else if (tid == P_HE_SUM) out = h.he_valid ? he[(size_t)b * nf + f] : 0.0;
                    else if (tid == P_HE_CNT) out = h.he_valid ? 1.0 : 0.0;
                }
                if (tid == P_FRAMES_N) out = usedA ? 1.0 : 0.0;
                else if (tid == P_NCORR_N) out = usedA ? (double)n_corr : 0.0;
                part[tid] = out;
            }
        }
#ifdef IBA_STAMPS
        __syncthreads();
        if (tid == 0) { const unsigned long long te = __builtin_readcyclecounter(); for (int i = 0; i < 7; ++i) part[56 + i] = (double)((i < 6 ? stamp_t[i + 1] : te) - stamp_t[i]); part[63] = (double)(stamp_t[7] - stamp_t[1]); }
#endif
        return;
    }

    bool used_assoc = false;
    if (MODE == MODE_ASSOC || MODE == MODE_BOTH) {
        // ---- BuildProblem association (iba_local.cpp:145-323): which residual blocks exist at this x ----
        uint2* arow = assoc_out + (size_t)b * dp.n_kp_total + h.kp_base;
        const bool used = !((int)n_corr < prm.num_min_corr);   // iba_local.cpp:192
        used_assoc = used;
        if (MODE == MODE_ASSOC && tid < kPartialStride) part[tid] = (used && tid == P_FRAMES_N) ? 1.0 : ((used && tid == P_NCORR_N) ? (double)n_corr : 0.0);
        const PlaneRec* planes = planes_local;
        // dense list of the keypoints that can own residual blocks: a correspondence, a MapPoint (iba_local.cpp:213)
        // AND a covisible match (:259-260). Everything else gets an empty association row right here.
        for (uint32_t k = tid; k < Kceil; k += kThreads) {
            const bool want = used && k < K && s_best_idx[k] != kNone && (((int)kp_mp[k].w) & 3) == 3;
            if (k < K && !want) arow[k] = make_uint2(kNone, kNone);
            ordered_append(want, k, n3, s_list, s_wcnt);
        }
        if (!cached) {   // ComputeLocalNeighbor at the matched scan point of every listed keypoint (iba_local.cpp:207)
            for (uint32_t i = tid; i < n3; i += kThreads) s_bpos[i] = inv_perm[s_best_idx[s_list[i]]];
            __syncthreads();
            fit_points(s_bpos, n3, prm.neigh_radius2, prm.neigh_max_pts, dp.scratch_local + scr_off);
        }
        for (uint32_t i = tid; i < n3; i += kThreads) {
            const uint32_t k = s_list[i];
            const uint32_t pos = inv_perm[s_best_idx[k]];
            const PlaneRec rec = planes[pos];
            uint2 a = make_uint2(kNone, kNone);
            // ComputeLocalNeighbor validity (pointcloud.h:752)
            const bool neigh_ok = !(rec.k < prm.neigh_min_pts || rec.far_d2 < prm.local_min_diff_dist2);
            if (neigh_ok && rec.reg_sum / (double)(rec.k - 1) < prm.local_norm_reg_threshold) a.x = pos;   // bvalid_plane (:231)
            arow[k] = a;
            s_bpos[i] = neigh_ok ? 0u : kNone;   // kNone: no 3d-3d block either (the `continue` at :209-211)
        }
        __syncthreads();
        // MapPoint -> LiDAR frame (iba_local.cpp:238-239, 282), 1-NN, local plane at the NN (pointcloud.h:699-717)
        auto q_assoc = [&](uint32_t k, double& qx, double& qy, double& qz) {
            const float4 mp = kp_mp[k];
            const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
            const double mx = ((h.Tcw[0] * w0 + h.Tcw[1] * w1) + h.Tcw[2] * w2) + h.Tcw[3];
            const double my = ((h.Tcw[4] * w0 + h.Tcw[5] * w1) + h.Tcw[6] * w2) + h.Tcw[7];
            const double mz = ((h.Tcw[8] * w0 + h.Tcw[9] * w1) + h.Tcw[10] * w2) + h.Tcw[11];
            const double sx = mx * s, sy = my * s, sz = mz * s;
            qx = ((cd.Ri[0] * sx + cd.Ri[1] * sy) + cd.Ri[2] * sz) + cd.ti[0];
            qy = ((cd.Ri[3] * sx + cd.Ri[4] * sy) + cd.Ri[5] * sz) + cd.ti[1];
            qz = ((cd.Ri[6] * sx + cd.Ri[7] * sy) + cd.Ri[8] * sz) + cd.ti[2];
        };
        nn_rounds(std::integral_constant<int, 1>(), n3,
            [&](uint32_t i, bool& actA, bool& actC, double& ax, double& ay, double& az, double& qx, double& qy, double& qz) {
                actA = s_bpos[i] != kNone; actC = false;
                if (actA) q_assoc(s_list[i], ax, ay, az);
            },
            [&](uint32_t i, bool actA, bool actC, const DualNN& st) { s_bpos[i] = (actA && !(st.bestA > prm.max_3d_dist2)) ? st.bposA : kNone; });   // :289
        __syncthreads();
        if (!cached) fit_points(s_bpos, n3, prm.neigh_radius2, prm.neigh_max_pts, dp.scratch_local + scr_off);
        for (uint32_t i = tid; i < n3; i += kThreads) {
            const uint32_t bpos = s_bpos[i];
            if (bpos == kNone) continue;
            const PlaneRec r2 = planes[bpos];
            const bool state = !(r2.k < prm.neigh_min_pts || r2.far_d2 < prm.local_min_diff_dist2) &&
                               (r2.reg_sum / (double)(r2.k - 1) < prm.local_norm_reg_threshold);
            arow[s_list[i]].y = bpos | (state ? 0x80000000u : 0u);
        }
        // dense residual-block list of this (candidate, frame) for the factor kernel: {keypoint, plane point, 3d-3d point|kind}
        {
            uint4* fl = flist + ((size_t)b * nf + f) * (size_t)flist_stride;
            for (uint32_t i = tid; i < n3; i += kThreads) {
                const uint32_t k = s_list[i];
                const uint2 a = arow[k];   // both halves were written by this same thread
                fl[i] = make_uint4(k, a.x, a.y, 0u);
            }
            if (tid == 0) fcount[(size_t)b * nf + f] = n3;
        }
        if (MODE == MODE_ASSOC) return;
        __syncthreads();   // s_list is reused by the cost path below
        n3 = 0;
    }

    if ((int)n_corr < prm.num_min_corr_cost) {   // iba_global.cpp:203: frame skipped entirely
        if (tid < kPartialStride) part[tid] = (used_assoc && tid == P_FRAMES_N) ? 1.0 : ((used_assoc && tid == P_NCORR_N) ? (double)n_corr : 0.0);
        return;
    }

    IBA_STAMP(4); IBA_RELOAD_AT(4);
    // ---- phase 4: ONE dense work list (keypoints with a correspondence that own a MapPoint and/or a covisible match:
    //      ~1 keypoint in 8), built in keypoint order with the only barriers of the phase; then all the arithmetic
    //      runs on full waves instead of dragging every wave through code most of its lanes skip ----
    for (uint32_t k = tid; k < Kceil; k += kThreads) {
        const bool want = k < K && s_best_idx[k] != kNone && kp_mp[k].w != 0.0f;
        ordered_append(want, k, n3, s_list, s_wcnt);
    }
    double sum2d = 0.0, sum3d = 0.0;
    uint32_t c2 = 0, v2 = 0, c3 = 0, v3 = 0, vpl = 0, vpt = 0;
    // K6: 3d-2d covisible reprojection residuals (iba_global.cpp:291-328)
    for (uint32_t i = tid; i < n3; i += kThreads) {
        const uint32_t k = s_list[i];
        float2 m[4];
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) m[sl] = (uint32_t)sl < h.n_slots ? dp.match_uv[h.match_base + (size_t)sl * K + k] : make_float2(__builtin_nanf(""), 0.f);
        const uint32_t pos = inv_perm[s_best_idx[k]];   // issued together with the match loads
        float xf_, yf_, zf_; load_pt<!SCAN_LDS>(c, pos, xf_, yf_, zf_);
        const double x = (double)xf_, y = (double)yf_, z = (double)zf_;
        const double p0x = ((c.R[0] * x + c.R[1] * y) + c.R[2] * z) + c.t[0];
        const double p0y = ((c.R[3] * x + c.R[4] * y) + c.R[5] * z) + c.t[1];
        const double p0z = ((c.R[6] * x + c.R[7] * y) + c.R[8] * z) + c.t[2];
        for (uint32_t sl = 0; sl < h.n_slots; ++sl) {
            const float2 mm = sl < 4 ? m[sl & 3] : dp.match_uv[h.match_base + (size_t)sl * K + k];
            if (mm.x != mm.x) continue;   // NaN: keypoint not in GetUordMatchedKptIds(pKFConv)
            const double* rel = s_rel + sl * 12;
            const double p1x = ((rel[0] * p0x + rel[1] * p0y) + rel[2] * p0z) + rel[3] * s;
            const double p1y = ((rel[4] * p0x + rel[5] * p0y) + rel[6] * p0z) + rel[7] * s;
            const double p1z = ((rel[8] * p0x + rel[9] * p0y) + rel[10] * p0z) + rel[11] * s;
            const double ou = h.fx * p1x / p1z + h.cx;
            const double ov = h.fy * p1y / p1z + h.cy;
            if (!(ou >= 0 && ou < h.W && ov >= 0 && ov < h.H)) continue;
            const double eu = ou - (double)mm.x, ev = ov - (double)mm.y;
            const double dist = sqrt(eu * eu + ev * ev);
            if (dist < prm.corr_3d_2d_threshold) { sum2d += dist; ++v2; }
            ++c2;
        }
    }

#ifdef IBA_STAMPS
    __syncthreads();
#endif
    IBA_STAMP(5); IBA_RELOAD_AT(5);
    // K4+K5: 3d-3d (MapPoint -> LiDAR frame, 1-NN with G lanes per query, memoised local plane)
#ifdef IBA_STAMPS
    const unsigned long long tw0 = __builtin_readcyclecounter();
#endif
    if (prm.use_3d3d) {
        auto q_cost = [&](uint32_t k, double& qx, double& qy, double& qz) {
            // transform constants re-read from constant memory at each call site (see `queries` in the fused branch)
            typedef __attribute__((address_space(4))) const FrameHdr FrameHdrC;
            typedef __attribute__((address_space(4))) const Cand CandC;
            FrameHdrC* hq = (FrameHdrC*)&h; CandC* cq = (CandC*)&cd;
            asm volatile("" : "+s"(hq), "+s"(cq));
#define h (*hq)
#define cd (*cq)
            const double s = cd.s;
            const double ts0 = h.Tcw[3] * s, ts1 = h.Tcw[7] * s, ts2 = h.Tcw[11] * s;   // TcwRS translation *= scale (:208)
            const float4 mp = kp_mp[k];
            const float m0 = mp.x * cd.s32, m1 = mp.y * cd.s32, m2 = mp.z * cd.s32;   // CV_32F product (:232)
            const double a0 = (double)m0, a1 = (double)m1, a2 = (double)m2;
            const double cx_ = ((h.Tcw[0] * a0 + h.Tcw[1] * a1) + h.Tcw[2] * a2) + ts0;
            const double cy_ = ((h.Tcw[4] * a0 + h.Tcw[5] * a1) + h.Tcw[6] * a2) + ts1;
            const double cz_ = ((h.Tcw[8] * a0 + h.Tcw[9] * a1) + h.Tcw[10] * a2) + ts2;
            qx = ((cd.Ri[0] * cx_ + cd.Ri[1] * cy_) + cd.Ri[2] * cz_) + cd.ti[0];
            qy = ((cd.Ri[3] * cx_ + cd.Ri[4] * cy_) + cd.Ri[5] * cz_) + cd.ti[1];
            qz = ((cd.Ri[6] * cx_ + cd.Ri[7] * cy_) + cd.Ri[8] * cz_) + cd.ti[2];
#undef h
#undef cd
        };
        nn_rounds(std::integral_constant<int, 2>(), n3,
            [&](uint32_t i, bool& actA, bool& actC, double& ax, double& ay, double& az, double& qx, double& qy, double& qz) {
                const uint32_t k = s_list[i];
                actA = false; actC = (((int)kp_mp[k].w) & 1) != 0;   // a covisible match without a MapPoint has no 3d-3d term
                if (actC) q_cost(k, qx, qy, qz);
            },
            [&](uint32_t i, bool actA, bool actC, const DualNN& st) { s_bpos[i] = actC ? st.bposC : kNone; });
        __syncthreads();
        if (!cached && prm.use_plane) fit_points(s_bpos, n3, prm.norm_radius2, prm.norm_max_pts, dp.scratch_cost + scr_off);   // iba_global.cpp:125-147
        for (uint32_t i = tid; i < n3; i += kThreads) {
            const uint32_t bpos = s_bpos[i];
            if (bpos == kNone) continue;
            double qx, qy, qz; q_cost(s_list[i], qx, qy, qz);
            float xf_, yf_, zf_; load_pt<!SCAN_LDS>(c, bpos, xf_, yf_, zf_);
            const double ax = (double)xf_ - qx, ay = (double)yf_ - qy, az = (double)zf_ - qz;
            double dist = sqrt((ax * ax + ay * ay) + az * az);   // (nn_pt - query_pt).norm()  (:122)
            bool is_plane = false;
            if (prm.use_plane) {
                const PlaneRec rec = planes_cost[bpos];
                if (!(rec.far_d2 < prm.min_diff_dist2) && !(rec.k < prm.norm_min_pts) &&
                    !(rec.reg_sum / (double)(rec.k - 1) > prm.norm_reg_threshold)) {
                    dist = fabs(ax * rec.nx + ay * rec.ny + az * rec.nz);
                    is_plane = true;
                }
            }
            if (dist < prm.corr_3d_3d_threshold) { sum3d += dist; ++v3; if (is_plane) ++vpl; else ++vpt; }
            ++c3;
        }
    }

#ifdef IBA_STAMPS
    __syncthreads();
#endif
    IBA_STAMP(6); IBA_RELOAD_AT(6);
    // ---- phase 5 (K8): two double sums + six exact integer counters packed 21 bits each; DPP wave sums, then the
    //      16 waves in fixed order. No atomics => bitwise reproducible.
    {
        const double w2d = wave_sum_f64(sum2d), w3d = wave_sum_f64(sum3d);
        const unsigned long long wa = wave_sum_u64((unsigned long long)c2 | ((unsigned long long)v2 << 21) | ((unsigned long long)c3 << 42));
        const unsigned long long wb = wave_sum_u64((unsigned long long)v3 | ((unsigned long long)vpl << 21) | ((unsigned long long)vpt << 42));
        unsigned long long* s_redu = (unsigned long long*)s_red;
        if (lane == 63) { s_red[wave * 4 + 0] = w2d; s_red[wave * 4 + 1] = w3d; s_redu[wave * 4 + 2] = wa; s_redu[wave * 4 + 3] = wb; }
        __syncthreads();
        if (tid < kPartialStride) {
            double out = 0.0;
            if (tid == P_SUM_3D2D || tid == P_SUM_3D3D) { for (int w = 0; w < kWaves; ++w) out += s_red[w * 4 + (tid == P_SUM_3D2D ? 0 : 1)]; }
            else if (tid >= P_CNT_3D2D && tid <= P_VALID_PT) {
                unsigned long long a = 0, bb = 0;
                for (int w = 0; w < kWaves; ++w) { a += s_redu[w * 4 + 2]; bb += s_redu[w * 4 + 3]; }
                const unsigned long long msk = (1ull << 21) - 1ull;
                const unsigned long long vals[6] = {a & msk, (a >> 21) & msk, (a >> 42) & msk, bb & msk, (bb >> 21) & msk, (bb >> 42) & msk};
                out = (double)vals[tid - P_CNT_3D2D];
                if (!prm.use_3d3d && (tid == P_CNT_3D3D || tid == P_VALID_3D3D)) out = 1.0;   // iba_global.cpp:214-220
            }
            else if (tid == P_FRAMES) out = 1.0;
            else if (tid == P_NCORR) out = (double)n_corr;
            else if (tid == P_HE_SUM) out = h.he_valid ? he[(size_t)b * nf + f] : 0.0;   // K7 (iba_he_kernel)
            else if (tid == P_HE_CNT) out = h.he_valid ? 1.0 : 0.0;
            else if (tid == P_FRAMES_N) out = used_assoc ? 1.0 : 0.0;
            else if (tid == P_NCORR_N) out = used_assoc ? (double)n_corr : 0.0;
            part[tid] = out;
        }
    }
#ifdef IBA_STAMPS
    __syncthreads();
    if (tid == 0) { const unsigned long long te = __builtin_readcyclecounter(); for (int i = 0; i < 7; ++i) part[56 + i] = (double)((i < 6 ? stamp_t[i + 1] : te) - stamp_t[i]); part[63] = (double)(stamp_t[7] - stamp_t[1]); }
#endif
}
#undef dp
#undef prm
#undef lay
#undef IBA_RELOAD_AT

// ---- Jacobian path: residual blocks of the frozen association, evaluated at candidate x ----
// IBA_PlaneFactor (IBACalib2.hpp:152-184), Point2Plane/Point2Point_Factor (:570-584, 611-625), Huber
// IRLS weights as Ceres' Corrector applies them (rho'' <= 0), accumulated as the upper triangle of
// H = sum w J^T J, b = sum w J^T r. Derivatives are analytic: the chain rule through the same
// expressions the reference's Jets differentiate, with dR/dx, dt/dx from the host duals (Cand).
// The Jacobian path is compiled WITHOUT FMA contraction, like the cost path: measured on the C2 / C3 scenes
// (tools/entry_parity_c2.py, r02), every entry of H and b above 1e-6 of the largest then agrees with the oracle's forward-mode
// duals to 6e-14 .. 2e-13 of ITSELF (cost 2e-12); with contraction the same figures are 5e-11 .. 1.2e-10 — inside BASELINE.md's
// 1e-10 gate only just, for 0.05 ms per 64 candidates (0.17 -> 0.22 ms). -DIBA_JAC_FMA builds the contracted variant.
// (What neither build can remove: a plane factor whose viewing ray lies almost in its plane — Z0 = num / den with a
// cancelling den, residuals of 10^4 px, Jacobian entries of 10^9 — carries a relative uncertainty of ~1e-10 in the
// reference's own double arithmetic; one such block shifts entries of b by up to 1e-7 of themselves in either build.)
#ifdef IBA_JAC_FMA
#define IBA_JAC_CONTRACT _Pragma("clang fp contract(fast)")
#else
#define IBA_JAC_CONTRACT _Pragma("clang fp contract(off)")
#endif
struct NAcc { double H[28], b[7], chi2, cost, nf2d, nfpl, nfpt, nres; };

__device__ __forceinline__ void huber_w(double a, double s, double& rho0, double& w) {
    const double bb = a * a;
    if (s > bb) { const double r = sqrt(s); rho0 = 2.0 * a * r - bb; w = fmax(2.2250738585072014e-308, a / r); }
    else { rho0 = s; w = 1.0; }
}
// H (upper, row-major i<=j) index
__device__ __forceinline__ int hidx(int i, int j) { return i * 7 - (i * (i - 1)) / 2 + (j - i); }

// IBA_PlaneFactor core: calls slot(ru, rv, gu, gv, hu, hv) per matched covisible KF, where the two residual rows are
// (ru, rv) and their Jacobian rows are [gu * z6, hu], [gv * z6, hv] (z6 = dZ0/dx[0:6], last column d/ds).
template <class SlotFn>
__device__ __forceinline__ int plane_factor_core(const Cand& c, const FrameHdr& h, const DevProblem& dp, uint32_t k, uint32_t K,
                                                 double u0, double v0, const double* p0, const double* n0, double* z6, SlotFn slot) {
    IBA_JAC_CONTRACT   // Jacobian path: H, b carry a relative budget, not bit parity
    double p0c[3], n0c[3];
    for (int r = 0; r < 3; ++r) {
        p0c[r] = ((c.R[r * 3] * p0[0] + c.R[r * 3 + 1] * p0[1]) + c.R[r * 3 + 2] * p0[2]) + c.t[r];
        n0c[r] = (c.R[r * 3] * n0[0] + c.R[r * 3 + 1] * n0[1]) + c.R[r * 3 + 2] * n0[2];
    }
    const double Cxz = (u0 - h.cx) / h.fx, Cyz = (v0 - h.cy) / h.fy;
    const double num = (n0c[0] * p0c[0] + n0c[1] * p0c[1]) + n0c[2] * p0c[2];
    const double den = (Cxz * n0c[0] + Cyz * n0c[1]) + n0c[2];
    const double Z0 = num / den;
    const double iden = 1.0 / den;   // derivative rows only: 1 ulp-level differences are inside the 1e-10 budget of H, b
    for (int kk = 0; kk < 6; ++kk) {
        double dpv[3], dnv[3] = {0, 0, 0};
        for (int r = 0; r < 3; ++r) {
            dpv[r] = c.dt[kk][r];
            if (kk < 3) {
                dpv[r] += (c.dR[kk][r * 3] * p0[0] + c.dR[kk][r * 3 + 1] * p0[1]) + c.dR[kk][r * 3 + 2] * p0[2];
                dnv[r] = (c.dR[kk][r * 3] * n0[0] + c.dR[kk][r * 3 + 1] * n0[1]) + c.dR[kk][r * 3 + 2] * n0[2];
            }
        }
        const double dnum = ((dnv[0] * p0c[0] + dnv[1] * p0c[1]) + dnv[2] * p0c[2]) + ((n0c[0] * dpv[0] + n0c[1] * dpv[1]) + n0c[2] * dpv[2]);
        const double dden = (Cxz * dnv[0] + Cyz * dnv[1]) + dnv[2];
        z6[kk] = (dnum - Z0 * dden) * iden;
    }
    const double P0x = Cxz * Z0, P0y = Cyz * Z0, P0z = Z0;
    int nconv = 0;
    // only the covisible slots whose match bit is set in the keypoint flags (kp_mp.w >> 2), in slot order; the next
    // match is in flight during the arithmetic of the current one
    uint32_t mask = ((uint32_t)(int)dp.kp_mp[h.kp_base + k].w >> 2) & ((1u << kMaxCovis) - 1u);
    const float2* mrow = dp.match_uv + h.match_base + k;
    float2 mnext = mask ? mrow[(size_t)(__ffs((int)mask) - 1) * K] : make_float2(0.f, 0.f);
    while (mask) {
        const uint32_t sl = (uint32_t)__ffs((int)mask) - 1u;
        mask &= mask - 1u;
        const float2 m = mnext;
        if (mask) mnext = mrow[(size_t)(__ffs((int)mask) - 1) * K];
        const double* rel = dp.slots[h.slot_base + sl].rel;
        const double tx = rel[3] * c.s, ty = rel[7] * c.s, tz = rel[11] * c.s;   // _t *= _s (IBACalib2.hpp:175)
        const double P1x = ((rel[0] * P0x + rel[1] * P0y) + rel[2] * P0z) + tx;
        const double P1y = ((rel[4] * P0x + rel[5] * P0y) + rel[6] * P0z) + ty;
        const double P1z = ((rel[8] * P0x + rel[9] * P0y) + rel[10] * P0z) + tz;
        const double ru = (h.fx * P1x / P1z + h.cx) - (double)m.x;
        const double rv = (h.fy * P1y / P1z + h.cy) - (double)m.y;
        const double ax = (rel[0] * Cxz + rel[1] * Cyz) + rel[2], ay = (rel[4] * Cxz + rel[5] * Cyz) + rel[6], az = (rel[8] * Cxz + rel[9] * Cyz) + rel[10];
        const double iz = 1.0 / P1z, xz = P1x * iz, yz = P1y * iz;
        const double gu = h.fx * iz * (ax - xz * az), gv = h.fy * iz * (ay - yz * az);
        const double hu = h.fx * iz * (rel[3] - xz * rel[11]), hv = h.fy * iz * (rel[7] - yz * rel[11]);
        slot(ru, rv, gu, gv, hu, hv);
        ++nconv;
    }
    return nconv;
}

__device__ inline void plane_factor_accum(const Cand& c, const FrameHdr& h, const DevProblem& dp, const DevParams& prm, uint32_t k, uint32_t K,
                                          double u0, double v0, const double* p0, const double* n0, NAcc& A) {
    IBA_JAC_CONTRACT   // Jacobian path: H, b carry a relative budget, not bit parity
    double z6[6];
    double ssq = 0, G = 0, GH = 0, HH = 0, Gr = 0, Hr = 0;
    const int nconv = plane_factor_core(c, h, dp, k, K, u0, v0, p0, n0, z6, [&](double ru, double rv, double gu, double gv, double hu, double hv) {
IBA_JAC_CONTRACT
        ssq += ru * ru + rv * rv;
        G += gu * gu + gv * gv; GH += gu * hu + gv * hv; HH += hu * hu + hv * hv;
        Gr += gu * ru + gv * rv; Hr += hu * ru + hv * rv;
    });
    if (nconv == 0) return;
    double rho0, w; huber_w(prm.robust_kernel_delta, ssq, rho0, w);
    A.cost += 0.5 * rho0; A.chi2 += ssq; A.nf2d += 1.0; A.nres += 2.0 * nconv;
    for (int i = 0; i < 6; ++i) {
        const double wz = w * z6[i];
        for (int j = i; j < 6; ++j) A.H[hidx(i, j)] += wz * G * z6[j];
        A.H[hidx(i, 6)] += wz * GH;
        A.b[i] += wz * Gr;
    }
    A.H[27] += w * HH; A.b[6] += w * Hr;
}

// M = Rlc (s m) + tlc and dM/dx for the 3d-3d factors (IBACalib2.hpp:570-584, 611-625)
__device__ __forceinline__ void p2x_core(const Cand& c, const FrameHdr& h, const float4 mp, double* M, double dM[7][3]) {
    IBA_JAC_CONTRACT   // Jacobian path: H, b carry a relative budget, not bit parity
    const double w0 = (double)mp.x, w1 = (double)mp.y, w2 = (double)mp.z;
    const double m[3] = {((h.Tcw[0] * w0 + h.Tcw[1] * w1) + h.Tcw[2] * w2) + h.Tcw[3], ((h.Tcw[4] * w0 + h.Tcw[5] * w1) + h.Tcw[6] * w2) + h.Tcw[7],
                         ((h.Tcw[8] * w0 + h.Tcw[9] * w1) + h.Tcw[10] * w2) + h.Tcw[11]};
    const double sm[3] = {m[0] * c.s, m[1] * c.s, m[2] * c.s};
    for (int r = 0; r < 3; ++r) {
        M[r] = ((c.Rlc[r * 3] * sm[0] + c.Rlc[r * 3 + 1] * sm[1]) + c.Rlc[r * 3 + 2] * sm[2]) + c.tlc[r];
        for (int kk = 0; kk < 6; ++kk) {
            double v = c.dtlc[kk][r];
            if (kk < 3) v += (c.dRlc[kk][r * 3] * sm[0] + c.dRlc[kk][r * 3 + 1] * sm[1]) + c.dRlc[kk][r * 3 + 2] * sm[2];
            dM[kk][r] = v;
        }
        dM[6][r] = (c.Rlc[r * 3] * m[0] + c.Rlc[r * 3 + 1] * m[1]) + c.Rlc[r * 3 + 2] * m[2];
    }
}

__device__ inline void p2x_factor_accum(const Cand& c, const FrameHdr& h, const DevParams& prm, const float4 mp, const double* Q, const double* n, bool is_plane, NAcc& A) {
    IBA_JAC_CONTRACT   // Jacobian path: H, b carry a relative budget, not bit parity
    double M[3], dM[7][3];
    p2x_core(c, h, mp, M, dM);
    const double e[3] = {M[0] - Q[0], M[1] - Q[1], M[2] - Q[2]};
    if (is_plane) {
        const double r = (e[0] * n[0] + e[1] * n[1]) + e[2] * n[2];
        double J[7];
        for (int kk = 0; kk < 7; ++kk) J[kk] = (dM[kk][0] * n[0] + dM[kk][1] * n[1]) + dM[kk][2] * n[2];
        double rho0, w; huber_w(prm.robust_kernel_3ddelta, r * r, rho0, w);
        A.cost += 0.5 * rho0; A.chi2 += r * r; A.nfpl += 1.0; A.nres += 1.0;
        for (int i = 0; i < 7; ++i) { const double wj = w * J[i]; for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wj * J[j]; A.b[i] += wj * r; }
    } else {
        const double ssq = (e[0] * e[0] + e[1] * e[1]) + e[2] * e[2];
        double rho0, w; huber_w(prm.robust_kernel_3ddelta, ssq, rho0, w);
        A.cost += 0.5 * rho0; A.chi2 += ssq; A.nfpt += 1.0; A.nres += 3.0;
        for (int r = 0; r < 3; ++r)
            for (int i = 0; i < 7; ++i) { const double wj = w * dM[i][r]; for (int j = i; j < 7; ++j) A.H[hidx(i, j)] += wj * dM[j][r]; A.b[i] += wj * e[r]; }
    }
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
__global__ __launch_bounds__(kFactorThreads) __attribute__((amdgpu_waves_per_eu(IBA_FACTOR_WAVES, IBA_FACTOR_WAVES))) void iba_factor_kernel(DevProblem dp, DevParams prm, const Cand* __restrict__ cands, const uint4* __restrict__ flist,
                                                                    const uint32_t* __restrict__ fcount, int flist_stride, int per_cand,
                                                                    double* __restrict__ partials, int nrec, int rec_base) {
    __shared__ double s_part[kFactorThreads / 64][48];
    __shared__ double s_tr[kFactorThreads / 64][21][65];   // [sum][lane], rows padded against bank conflicts
    const int f = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const FrameHdr& h = dp.frames[f];
    const Cand& c = cands[b];
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
    for (uint32_t i = tid; i < n; i += kFactorThreads) {
        const uint4 e = fl[i];
        const uint32_t k = e.x;
        if (e.y != kNone) {
            const PlaneRec rec = planes[e.y];
            const float4 pt = p4[e.y];
            const double p0[3] = {(double)pt.x, (double)pt.y, (double)pt.z}, n0[3] = {rec.nx, rec.ny, rec.nz};
            const float2 uv = dp.kp_uv[h.kp_base + k];
            plane_factor_accum(c, h, dp, prm, k, h.K, (double)uv.x, (double)uv.y, p0, n0, A);
        }
        if (e.z != kNone) {
            const uint32_t pos = e.z & 0x7FFFFFFFu; const bool is_plane = (e.z >> 31) != 0;
            const PlaneRec rec = planes[pos];
            const float4 pt = p4[pos];
            const double Q[3] = {(double)pt.x, (double)pt.y, (double)pt.z}, nn[3] = {rec.nx, rec.ny, rec.nz};
            p2x_factor_accum(c, h, prm, dp.kp_mp[h.kp_base + k], Q, nn, is_plane, A);
        }
    }
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
                    const double r = (ev[0] * rec.nx + ev[1] * rec.ny) + ev[2] * rec.nz;
                    double J[8];
                    for (int kk = 0; kk < 7; ++kk) J[kk] = (dM[kk][0] * rec.nx + dM[kk][1] * rec.ny) + dM[kk][2] * rec.nz;
                    J[7] = r;
                    double rho0; huber_w(prm.robust_kernel_3ddelta, r * r, rho0, w);
                    cost += 0.5 * rho0; chi2 += r * r; nfpl += 1.0; nres += 1.0;
#pragma unroll
                    for (int q = 0; q < 8; q += 2) *(double2*)(Q + q) = make_double2(J[q], J[q + 1]);
                } else {
                    const double ssq = (ev[0] * ev[0] + ev[1] * ev[1]) + ev[2] * ev[2];
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
            r_out[row] = (e[0] * rec.nx + e[1] * rec.ny) + e[2] * rec.nz;
            for (int i = 0; i < 7; ++i) J_out[row * 7 + i] = (dM[i][0] * rec.nx + dM[i][1] * rec.ny) + dM[i][2] * rec.nz;
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

// writes two host-known values into their slots of B partial blocks (frozen-problem counts)
__global__ void iba_set_slots_kernel(double* __restrict__ partials, int B, int slot_a, double va, int slot_b, double vb) {
    const int b = threadIdx.x;
    if (b < B) { partials[(size_t)b * kPartialStride + slot_a] = va; partials[(size_t)b * kPartialStride + slot_b] = vb; }
}

}  // namespace iba
