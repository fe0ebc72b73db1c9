// TEST INFRASTRUCTURE ONLY — CPU oracle: a restatement of the reference's IBA evaluation path.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
// and only as the checker / the timed CPU baseline — never from the product path.
//
// Follows, step for step:
//   FindProjectCorrespondences   src/examples/iba_global.cpp:55-96  (= iba_local.cpp:17-58)
//   ComputeAlignmentDist         src/examples/iba_global.cpp:111-156
//   BAError                      src/examples/iba_global.cpp:169-344 (= iba_func.cpp:179-354)
//   BALoss::eval_x packing       src/examples/iba_global.cpp:386-392
//   BuildProblem                 src/examples/iba_local.cpp:145-323
//   ComputeLocalNeighbor / ComputeLocalNormalSingleThre   include/pointcloud.h:733-760, 651-666, 699-717
//   IBA_PlaneFactor / Point2Point_Factor / Point2Plane_Factor  include/IBACalib2.hpp:152-184, 570-584, 611-625
//   IBAPlaneEdge (g2o twin, 20-d zero padded)                  include/IBACalib.hpp:103-140
//   ceres::HuberLoss + Corrector (third-party; iba_local.cpp:263, 291)
// including the per-evaluation 2-D kd-tree rebuild (iba_global.cpp:84).
//
// PARITY PIN STATUS (see oracle_math.hpp): kd-tree/kNN pinned to the reference's nanoflann; the
// closed-form math pinned by known-answer tests; the remainder "parity unpinned" because the
// reference has no tests and its third-party dependencies (Eigen, OpenCV, g2o, Ceres) are absent.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <unordered_map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/iba_mi355x.h"
#include "oracle_kdtree.hpp"
#include "oracle_math.hpp"

using namespace oracle;

namespace {

using KDTree2D = KDTree<2>;
using KDTree3D = KDTree<3>;

struct Covis {
    int frame;
    Iso3 rel;                           // widened CV_32F product, translation unscaled
    std::unordered_map<int, int> kptmap;  // GetUordMatchedKptIds
};

struct Frame {
    std::vector<double> pts;  // VecVector3d (P x 3 doubles), LiDAR frame
    std::unique_ptr<KDTree3D> tree;
    double fx, fy, cx, cy, W, H;
    std::vector<float> kp_uv;      // K x 2
    std::vector<uint8_t> has_mp;   // K
    std::vector<float> mp_w;       // K x 3
    Iso3 Tcw;                      // widened float pose
    std::vector<Covis> covis;
    Iso3 Tc_next;                  // widened, unscaled
    Iso3 Tl_next;
    size_t P() const { return pts.size() / 3; }
    size_t K() const { return kp_uv.size() / 2; }
};

struct Factor {
    int kind;  // 0 IBA_PlaneFactor, 1 Point2Plane_Factor, 2 Point2Point_Factor, 3 IBATestEdge (iba_params.factor_3d2d_kind = 1: one covisible keyframe, p0 = the matched scan point)
    int frame, kp;
    double fx, fy, cx, cy, u0, v0;
    std::vector<double> u1, v1;
    std::vector<M3d> R;
    std::vector<V3d> t;
    V3d p0, n0;          // plane factor
    V3d MapPoint, Q, n;  // 3d-3d factors
    int rows() const { return (kind == 0 || kind == 3) ? 2 * (int)u1.size() : (kind == 1 ? 1 : 3); }
};

struct Oracle {
    std::vector<Frame> frames;
    int leaf2d = 10, leaf3d = 30;
    std::vector<Factor> factors;  // frozen problem
    int bp_frames_used = 0, bp_n_corr = 0;
    int bp_f_begin = 0, bp_f_end = -1;   // TEST AID: BuildProblem over a frame range only (what one rank of a sharded run owns); -1 = all
};

Iso3 iso_from_f32(const float* m) {
    Iso3 T;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.R(r, c) = (double)m[r * 4 + c]; T.t[r] = (double)m[r * 4 + 3]; }
    return T;
}
Iso3 iso_from_f64(const double* m) {
    Iso3 T;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.R(r, c) = m[r * 4 + c]; T.t[r] = m[r * 4 + 3]; }
    return T;
}

using CorrSet = std::vector<std::pair<uint32_t, uint32_t>>;

// iba_global.cpp:55-96. `PC` = scan already transformed into the camera frame.
void FindProjectCorrespondences(const std::vector<double>& PC, const Frame& kf, int leaf_size, double max_corr_dist, CorrSet& corrset) {
    const size_t K = kf.K();
    std::vector<double> vKeyUn(2 * K);
    for (size_t i = 0; i < K; ++i) { vKeyUn[2 * i] = (double)kf.kp_uv[2 * i]; vKeyUn[2 * i + 1] = (double)kf.kp_uv[2 * i + 1]; }
    const double fx = kf.fx, cx = kf.cx, cy = kf.cy;
    const double H = kf.H, W = kf.W;
    std::vector<double> ProjectPC; std::vector<uint32_t> ProjectIndex;
    const size_t n = PC.size() / 3;
    for (size_t i = 0; i < n; ++i) {
        const double px = PC[3 * i], py = PC[3 * i + 1], pz = PC[3 * i + 2];
        if (pz > 0) {
            double u = (fx * px + cx * pz) / pz;
            double v = (fx * py + cy * pz) / pz;  // fx, not fy: iba_global.cpp:73
            if (0 <= u && u < W && 0 <= v && v < H) { ProjectPC.push_back(u); ProjectPC.push_back(v); ProjectIndex.push_back((uint32_t)i); }
        }
    }
    if (ProjectIndex.empty()) return;
    KDTree2D kdtree(ProjectPC.data(), ProjectIndex.size(), (size_t)leaf_size);
    for (uint32_t i = 0; i < K; ++i) {
        uint32_t index; double sq_dist = std::numeric_limits<double>::max();
        KNNResultSet rs(1); rs.init(&index, &sq_dist);
        kdtree.findNeighbors(rs, &vKeyUn[2 * i]);
        if (rs.size() > 0 && sq_dist <= max_corr_dist * max_corr_dist) corrset.emplace_back(i, ProjectIndex[index]);
    }
}

struct PlaneFit { size_t k; double far_d2; V3d normal; double reg_err; std::vector<uint32_t> idx; };
// Shared by ComputeAlignmentDist (:125-147), ComputeLocalNeighbor (pointcloud.h:733-760) and
// ComputeLocalNormalSingleThre (pointcloud.h:699-717): kNN(max_pts) around `center`, clip to d^2 < r^2.
void knn_clip(const Frame& f, const double* center, int max_pts, double radius, std::vector<uint32_t>& indices, std::vector<double>& sq_dist, size_t& k) {
    indices.assign(max_pts, 0); sq_dist.assign(max_pts, 0.0);
    KNNResultSet rs((size_t)max_pts); rs.init(indices.data(), sq_dist.data());
    f.tree->findNeighbors(rs, center);
    k = rs.size();
    k = std::distance(sq_dist.begin(), std::lower_bound(sq_dist.begin(), sq_dist.begin() + k, radius * radius));
    indices.resize(k); sq_dist.resize(k);
}
void normal_and_reg(const Frame& f, const std::vector<uint32_t>& indices, const V3d& ref_pt, V3d& normal, double& reg_err) {
    M3d cov = ComputeCovariance(f.pts.data(), indices.data(), indices.size());
    double ev[3]; normal = normalized(FastEigen3x3_EV(cov, ev));
    reg_err = 0;
    for (uint32_t idx : indices) {
        V3d p{f.pts[3 * idx], f.pts[3 * idx + 1], f.pts[3 * idx + 2]};
        reg_err += std::abs(dot(p - ref_pt, normal));
    }
}

// iba_global.cpp:111-156
void ComputeAlignmentDist(const Frame& f, const V3d& query_pt, const iba_params& p, bool& is_plane, double& dist) {
    uint32_t nn_idx; double nn_sq;
    KNNResultSet nn(1); nn.init(&nn_idx, &nn_sq);
    const double q[3] = {query_pt.x, query_pt.y, query_pt.z};
    f.tree->findNeighbors(nn, q);
    const V3d nn_pt{f.pts[3 * nn_idx], f.pts[3 * nn_idx + 1], f.pts[3 * nn_idx + 2]};
    double pt2pt_dist = norm(nn_pt - query_pt);
    is_plane = false; dist = pt2pt_dist;
    if (!p.use_plane) return;
    std::vector<uint32_t> indices; std::vector<double> sq_dist; size_t k;
    const double c[3] = {nn_pt.x, nn_pt.y, nn_pt.z};
    knn_clip(f, c, p.norm_max_pts, p.norm_radius, indices, sq_dist, k);
    if (sq_dist[k - 1] < p.min_diff_dist * p.min_diff_dist) return;
    if (k < (size_t)p.norm_min_pts) return;
    V3d normal; double reg_err; normal_and_reg(f, indices, nn_pt, normal, reg_err);
    if (reg_err / (k - 1) > p.norm_reg_threshold) return;
    is_plane = true; dist = std::abs(dot(nn_pt - query_pt, normal));
}

void transform_cloud(const Frame& f, const Iso3& T, std::vector<double>& PC) {  // pointcloud.h:82-86
    const size_t n = f.P(); PC.resize(3 * n);
    for (size_t i = 0; i < n; ++i) {
        V3d q = apply(T, V3d{f.pts[3 * i], f.pts[3 * i + 1], f.pts[3 * i + 2]});
        PC[3 * i] = q.x; PC[3 * i + 1] = q.y; PC[3 * i + 2] = q.z;
    }
}

// iba_global.cpp:169-344
void BAError(const Oracle& O, const iba_params& prm, const double* xvec, bool multiprocessing, iba_cost_out& out, int f_begin = 0, int f_end = -1, double* raw = nullptr) {
    double corr_3d_2d_err = 0, corr_3d_3d_err = 0, Cval = 0, Ccnt = 0;
    int cnt_3d_2d = 0, valid_cnt_3d_2d = 0, valid_pl_3d_3d = 0, valid_pt_3d_3d = 0, cnt_3d_3d = 0, valid_cnt_3d_3d = 0;
    int frames_used = 0, n_corr = 0;
    M3d rotation; V3d translation; double scale;
    Sim3Exp<double>(xvec, rotation, translation, scale);
    Iso3 Tcl{rotation, translation};
    const Iso3 Tlc = inverse(Tcl);
    const int F = (int)O.frames.size();
    (void)multiprocessing;
    const int Fb = f_begin, Fe = f_end < 0 ? F : f_end;
#pragma omp parallel for if (multiprocessing)
    for (int Fi = Fb; Fi < Fe; ++Fi) {
        const Frame& kf = O.frames[Fi];
        std::vector<double> PC;
        transform_cloud(kf, Tcl, PC);
        CorrSet corrset;
        FindProjectCorrespondences(PC, kf, O.leaf2d, prm.max_pixel_dist, corrset);
        if ((int)corrset.size() < prm.num_min_corr_cost) continue;  // hard-coded 30 at :203
        Iso3 TcwRS = kf.Tcw;
        TcwRS.t = {TcwRS.t.x * scale, TcwRS.t.y * scale, TcwRS.t.z * scale};
#pragma omp critical
        { frames_used++; n_corr += (int)corrset.size(); }
        if (prm.err_weight[1] <= 1e-10) {
#pragma omp critical
            { corr_3d_3d_err = 0; cnt_3d_3d++; valid_cnt_3d_3d++; }
        } else {
            const float s32 = (float)scale;  // cv::Mat(CV_32F) * double: OpenCV convertTo scales in float
            for (auto const& [point2d_idx, point3d_idx] : corrset) {
                (void)point3d_idx;
                if (!kf.has_mp[point2d_idx]) continue;
                const float* pw = &kf.mp_w[3 * point2d_idx];
                const float m0 = pw[0] * s32, m1 = pw[1] * s32, m2 = pw[2] * s32;
                V3d MapRefPose{(double)m0, (double)m1, (double)m2};
                MapRefPose = apply(TcwRS, MapRefPose);
                MapRefPose = apply(Tlc, MapRefPose);
                bool is_planefit; double dist;
                ComputeAlignmentDist(kf, MapRefPose, prm, is_planefit, dist);
#pragma omp critical
                {
                    if (dist < prm.corr_3d_3d_threshold) {
                        corr_3d_3d_err += dist; valid_cnt_3d_3d++;
                        if (is_planefit) valid_pl_3d_3d++; else valid_pt_3d_3d++;
                    }
                    cnt_3d_3d++;
                }
            }
        }
        if (Fi < F - 1) {
            Iso3 Tc = kf.Tc_next;
            Tc.t = {Tc.t.x * scale, Tc.t.y * scale, Tc.t.z * scale};
            const Iso3 C1 = compose(Tcl, kf.Tl_next);
            const Iso3 C2 = compose(Tc, Tcl);
            double l1[6], l2[6]; SE3Log(C1.R, C1.t, l1); SE3Log(C2.R, C2.t, l2);
            double ss = 0; for (int i = 0; i < 6; ++i) ss += (l1[i] - l2[i]) * (l1[i] - l2[i]);
#pragma omp critical
            { Cval += std::sqrt(ss); Ccnt++; }
        }
        std::vector<Iso3> relCVPoseList; relCVPoseList.reserve(kf.covis.size());
        for (auto const& cv : kf.covis) {
            Iso3 rel = cv.rel; rel.t = {rel.t.x * scale, rel.t.y * scale, rel.t.z * scale};
            relCVPoseList.push_back(rel);
        }
        for (auto& corr : corrset) {
            const int point2d_idx = corr.first, point3d_idx = corr.second;
            V3d p0{PC[3 * point3d_idx], PC[3 * point3d_idx + 1], PC[3 * point3d_idx + 2]};
            const double fx = kf.fx, fy = kf.fy, cx = kf.cx, cy = kf.cy, H = kf.H, W = kf.W;
            for (size_t ci = 0; ci < kf.covis.size(); ++ci) {
                auto it = kf.covis[ci].kptmap.find(point2d_idx);
                if (it == kf.covis[ci].kptmap.end()) continue;
                const int covis_idx = it->second;
                const Frame& ckf = O.frames[kf.covis[ci].frame];
                double u1 = ckf.kp_uv[2 * covis_idx], v1 = ckf.kp_uv[2 * covis_idx + 1];
                V3d p1 = apply(relCVPoseList[ci], p0);
                double obs_u1 = fx * p1.x / p1.z + cx;
                double obs_v1 = fy * p1.y / p1.z + cy;
                if (!(obs_u1 >= 0 && obs_u1 < W && obs_v1 >= 0 && obs_v1 < H)) continue;
                double err = (obs_u1 - u1) * (obs_u1 - u1) + (obs_v1 - v1) * (obs_v1 - v1);
                double dist = std::sqrt(err);
#pragma omp critical
                {
                    if (dist < prm.corr_3d_2d_threshold) { corr_3d_2d_err += dist; valid_cnt_3d_2d++; }
                    cnt_3d_2d++;
                }
            }
        }
    }
    if (raw) {   // un-normalised sums in the order of the product's partial block (for the sharding tests)
        raw[0] = corr_3d_2d_err; raw[1] = corr_3d_3d_err; raw[2] = Cval; raw[3] = Ccnt; raw[4] = cnt_3d_2d; raw[5] = valid_cnt_3d_2d; raw[6] = cnt_3d_3d;
        raw[7] = valid_cnt_3d_3d; raw[8] = valid_pl_3d_3d; raw[9] = valid_pt_3d_3d; raw[10] = frames_used; raw[11] = n_corr;
    }
    if (valid_cnt_3d_2d == 0 && prm.err_weight[0] > 1e-10) corr_3d_2d_err = std::numeric_limits<double>::max();
    else corr_3d_2d_err /= valid_cnt_3d_2d;
    if (valid_cnt_3d_3d == 0 && prm.err_weight[1] > 1e-10) corr_3d_3d_err = std::numeric_limits<double>::max();
    else corr_3d_3d_err /= valid_cnt_3d_3d;
    Cval /= Ccnt;
    out.f1 = corr_3d_2d_err; out.f2 = corr_3d_3d_err; out.C = Cval;
    out.valid_cnt_3d_2d = valid_cnt_3d_2d; out.cnt_3d_2d = cnt_3d_2d; out.cnt_3d_3d = cnt_3d_3d; out.valid_cnt_3d_3d = valid_cnt_3d_3d;
    out.valid_pl_3d_3d = valid_pl_3d_3d; out.valid_pt_3d_3d = valid_pt_3d_3d; out.frames_used = frames_used; out.n_corr = n_corr;
}

// ------------------------------------------------------------------------------------------------
// Residual functors (templated on double / Dual<7>)
// ------------------------------------------------------------------------------------------------
template <class T> V3<T> castv(const V3d& v) { return {T(v.x), T(v.y), T(v.z)}; }
template <class T> M3<T> castm(const M3d& m) { M3<T> r; for (int i = 0; i < 9; ++i) r.m[i] = T(m.m[i]); return r; }

template <class T>
void eval_plane_factor(const Factor& f, const T* x, T* error) {  // IBACalib2.hpp:152-184
    M3<T> Rcl; V3<T> tcl; T s;
    Sim3Exp<T>(x, Rcl, tcl, s);
    T fx(f.fx), fy(f.fy), cx(f.cx), cy(f.cy), u0(f.u0), v0(f.v0);
    V3<T> p0 = castv<T>(f.p0), n0 = castv<T>(f.n0);
    V3<T> p0c = mul(Rcl, p0) + tcl;
    V3<T> n0c = mul(Rcl, n0);
    T Cxz = (u0 - cx) / fx;
    T Cyz = (v0 - cy) / fy;
    T Z0 = dot(n0c, p0c) / (Cxz * n0c.x + Cyz * n0c.y + n0c.z);
    T X0 = Cxz * Z0, Y0 = Cyz * Z0;
    V3<T> P0{X0, Y0, Z0};
    for (size_t i = 0; i < f.u1.size(); ++i) {
        M3<T> R = castm<T>(f.R[i]);
        V3<T> t = castv<T>(f.t[i]);
        t = t * s;
        T u1(f.u1[i]), v1(f.v1[i]);
        V3<T> P1 = mul(R, P0) + t;
        T u1_obs = fx * P1.x / P1.z + cx;
        T v1_obs = fy * P1.y / P1.z + cy;
        error[2 * i] = u1_obs - u1;
        error[2 * i + 1] = v1_obs - v1;
    }
}
// IBATestEdge::operator() (IBACalib.hpp:40-58): the direct point-to-pixel term. _p0c = _Rcl _p0 + _tcl (:49), _p1c = _R _p0c + _t with
// _t = t * _s (:48, :50), projection with fx, fy (:52-53), error = observation - keypoint (:54-55). u0, v0 are members the functor never reads.
template <class T>
void eval_test_edge(const Factor& f, const T* x, T* error) {
    M3<T> Rcl; V3<T> tcl; T s;
    Sim3Exp<T>(x, Rcl, tcl, s);
    T fx(f.fx), fy(f.fy), cx(f.cx), cy(f.cy);
    V3<T> p0 = castv<T>(f.p0);
    V3<T> p0c = mul(Rcl, p0) + tcl;
    for (size_t i = 0; i < f.u1.size(); ++i) {   // (one covisible keyframe per edge: size 1)
        M3<T> R = castm<T>(f.R[i]);
        V3<T> t = castv<T>(f.t[i]);
        t = t * s;
        T u1(f.u1[i]), v1(f.v1[i]);
        V3<T> p1c = mul(R, p0c) + t;
        T u1_obs = fx * p1c.x / p1c.z + cx;
        T v1_obs = fy * p1c.y / p1c.z + cy;
        error[2 * i] = u1_obs - u1;
        error[2 * i + 1] = v1_obs - v1;
    }
}
template <class T>
void eval_p2x_factor(const Factor& f, const T* x, T* error) {  // IBACalib2.hpp:570-584, 611-625
    T inv[6] = {-x[0], -x[1], -x[2], -x[3], -x[4], -x[5]};
    M3<T> Rlc; V3<T> tlc; T s = x[6];
    SE3Exp<T>(inv, Rlc, tlc);
    V3<T> M = mul(Rlc, castv<T>(f.MapPoint) * s) + tlc;
    V3<T> Q = castv<T>(f.Q);
    if (f.kind == 2) { error[0] = M.x - Q.x; error[1] = M.y - Q.y; error[2] = M.z - Q.z; }
    else { error[0] = dot(M - Q, castv<T>(f.n)); }
}
template <class T>
void eval_any_factor(const Factor& f, const T* x, T* error) {
    if (f.kind == 0) eval_plane_factor<T>(f, x, error); else if (f.kind == 3) eval_test_edge<T>(f, x, error); else eval_p2x_factor<T>(f, x, error);
}
void eval_factor(const Factor& f, const double* x, double* r, double* J /*rows x 7, may be null*/) {
    using D7 = Dual<7>;
    const int rows = f.rows();
    if (!J) {
        eval_any_factor<double>(f, x, r);
        return;
    }
    D7 xd[7]; for (int i = 0; i < 7; ++i) xd[i] = D7(x[i], i);
    std::vector<D7> e(rows);
    eval_any_factor<D7>(f, xd, e.data());
    for (int i = 0; i < rows; ++i) { r[i] = e[i].a; for (int c = 0; c < 7; ++c) J[i * 7 + c] = e[i].v[c]; }
}

// iba_local.cpp:145-323 (association only; the factors are appended in frame order, the reference's
// insertion order under `omp critical` is nondeterministic)
void BuildProblem(Oracle& O, const iba_params& prm, const double* params, bool multithread = false) {
    O.factors.clear(); O.bp_frames_used = 0; O.bp_n_corr = 0;
    std::vector<std::vector<Factor>> per_frame(O.frames.size());
    int frames_used = 0, n_corr = 0;
    const double max_3d_dist2 = prm.max_3d_dist * prm.max_3d_dist;
    M3d init_rotation; V3d init_translation; double init_scale;
    Sim3Exp<double>(params, init_rotation, init_translation, init_scale);
    Iso3 initSE3{init_rotation, init_translation};
    const Iso3 initSE3inv = inverse(initSE3);
    (void)multithread;
#pragma omp parallel for schedule(static) if (multithread)   // iba_local.cpp:162
    for (size_t Fi = 0; Fi < O.frames.size(); ++Fi) {
        if ((int)Fi < O.bp_f_begin || (O.bp_f_end >= 0 && (int)Fi >= O.bp_f_end)) continue;
        const Frame& kf = O.frames[Fi];
        std::vector<Factor>& out_factors = per_frame[Fi];
        std::vector<double> points; transform_cloud(kf, initSE3, points);
        CorrSet pt2d3d_map;
        FindProjectCorrespondences(points, kf, O.leaf2d, prm.max_pixel_dist, pt2d3d_map);
        if ((int)pt2d3d_map.size() < prm.num_min_corr) continue;
#pragma omp critical
        { frames_used++; n_corr += (int)pt2d3d_map.size(); }
        for (size_t ci = 0; ci < pt2d3d_map.size(); ++ci) {
            const uint32_t point2d_idx = pt2d3d_map[ci].first, point3d_idx = pt2d3d_map[ci].second;
            const double* c = &kf.pts[3 * (size_t)point3d_idx];
            const bool test_edges = prm.factor_3d2d_kind == 1;
            if (test_edges) {
                // IBATestEdge (IBACalib.hpp:14-71) over the edge set of BAError's 3d-2d loop (iba_global.cpp:291-328): this correspondence x every
                // covisible keyframe that matches its keypoint, the scan point itself as _p0 — before any of BuildProblem's own tests
                for (auto const& cv : kf.covis) {
                    auto it = cv.kptmap.find((int)point2d_idx);
                    if (it == cv.kptmap.end()) continue;
                    const Frame& ckf = O.frames[cv.frame];
                    Factor te; te.kind = 3; te.frame = (int)Fi; te.kp = (int)point2d_idx;
                    te.fx = kf.fx; te.fy = kf.fy; te.cx = kf.cx; te.cy = kf.cy; te.u0 = kf.kp_uv[2 * point2d_idx]; te.v0 = kf.kp_uv[2 * point2d_idx + 1];
                    te.p0 = V3d{c[0], c[1], c[2]}; te.n0 = {0, 0, 0}; te.MapPoint = {0, 0, 0}; te.Q = {0, 0, 0}; te.n = {0, 0, 0};
                    te.u1.push_back(ckf.kp_uv[2 * it->second]); te.v1.push_back(ckf.kp_uv[2 * it->second + 1]);
                    te.R.push_back(cv.rel.R); te.t.push_back(cv.rel.t);
                    out_factors.push_back(te);
                }
                if (!(prm.err_weight[1] > 1e-10)) continue;   // no 3d-3d blocks (BAError's switch, iba_global.cpp:214-220)
            }
            // ComputeLocalNeighbor (pointcloud.h:733-760)
            std::vector<uint32_t> neigh_idx; std::vector<double> sq_dist; size_t k;
            knn_clip(kf, c, prm.neigh_max_pts, prm.neigh_radius, neigh_idx, sq_dist, k);
            if ((int)k < prm.neigh_min_pts || sq_dist[k - 1] < prm.local_min_diff_dist * prm.local_min_diff_dist) continue;
            if (!kf.has_mp[point2d_idx]) continue;
            const V3d nn_pt{c[0], c[1], c[2]};
            V3d normal; double reg_err; normal_and_reg(kf, neigh_idx, nn_pt, normal, reg_err);
            reg_err /= neigh_idx.size() - 1;
            const bool bvalid_plane = reg_err < prm.local_norm_reg_threshold;
            const double u0 = kf.kp_uv[2 * point2d_idx], v0 = kf.kp_uv[2 * point2d_idx + 1];
            V3d MapPoint{(double)kf.mp_w[3 * point2d_idx], (double)kf.mp_w[3 * point2d_idx + 1], (double)kf.mp_w[3 * point2d_idx + 2]};
            MapPoint = apply(kf.Tcw, MapPoint);
            Factor pf; pf.kind = 0; pf.frame = (int)Fi; pf.kp = (int)point2d_idx;
            pf.fx = kf.fx; pf.fy = kf.fy; pf.cx = kf.cx; pf.cy = kf.cy; pf.u0 = u0; pf.v0 = v0; pf.p0 = nn_pt; pf.n0 = normal;
            for (auto const& cv : kf.covis) {
                auto it = cv.kptmap.find((int)point2d_idx);
                if (it == cv.kptmap.end()) continue;
                const Frame& ckf = O.frames[cv.frame];
                pf.u1.push_back(ckf.kp_uv[2 * it->second]); pf.v1.push_back(ckf.kp_uv[2 * it->second + 1]);
                pf.R.push_back(cv.rel.R); pf.t.push_back(cv.rel.t);  // translation unscaled: iba_local.cpp:184-188
            }
            if (pf.u1.empty()) continue;
            if (bvalid_plane && !test_edges) out_factors.push_back(pf);
            V3d MapPointInLidar = apply(initSE3inv, MapPoint * init_scale);
            uint32_t mp_idx; double mp_sq;
            KNNResultSet rs(1); rs.init(&mp_idx, &mp_sq);
            const double q[3] = {MapPointInLidar.x, MapPointInLidar.y, MapPointInLidar.z};
            kf.tree->findNeighbors(rs, q);
            if (mp_sq > max_3d_dist2) continue;
            const V3d NN{kf.pts[3 * (size_t)mp_idx], kf.pts[3 * (size_t)mp_idx + 1], kf.pts[3 * (size_t)mp_idx + 2]};
            // ComputeLocalNormalSingleThre (pointcloud.h:699-717 -> 651-666)
            std::vector<uint32_t> idx2; std::vector<double> sq2; size_t k2;
            const double c2[3] = {NN.x, NN.y, NN.z};
            knn_clip(kf, c2, prm.neigh_max_pts, prm.neigh_radius, idx2, sq2, k2);
            V3d n2{0, 0, 1}; bool state = false;
            if (!((int)k2 < prm.neigh_min_pts || sq2[k2 - 1] < prm.local_min_diff_dist * prm.local_min_diff_dist)) {
                double re; normal_and_reg(kf, idx2, NN, n2, re);
                re /= idx2.size() - 1;
                state = re < prm.local_norm_reg_threshold;
            }
            Factor g; g.kind = state ? 1 : 2; g.frame = (int)Fi; g.kp = (int)point2d_idx;
            g.fx = g.fy = g.cx = g.cy = g.u0 = g.v0 = 0; g.MapPoint = MapPoint; g.Q = NN; g.n = n2; g.p0 = {0, 0, 0}; g.n0 = {0, 0, 0};
            out_factors.push_back(g);
        }
    }
    O.bp_frames_used = frames_used; O.bp_n_corr = n_corr;
    for (auto& v : per_frame) for (auto& f : v) O.factors.push_back(std::move(f));   // frame order (the reference's order under omp critical is nondeterministic)
}

// ceres::HuberLoss::Evaluate + Corrector with rho'' <= 0 (third-party, restated)
inline void huber(double a, double s, double& rho0, double& rho1) {
    const double b = a * a;
    if (s > b) { const double r = std::sqrt(s); rho0 = 2.0 * a * r - b; rho1 = std::max(std::numeric_limits<double>::min(), a / r); }
    else { rho0 = s; rho1 = 1.0; }
}

// TEST AID (no reference counterpart): with g_exact_sums the per-block contributions rho1 J^T J, rho1 J^T r, rho — the same
// doubles as otherwise — are ADDED in long double (x87 80-bit), i.e. the sums are taken (nearly) exactly instead of in the
// order-dependent double arithmetic of a sequential loop. Used to separate "the device differs from the reference's terms"
// from "two double summations of 10^5 cancelling terms differ from each other" (tests/test_gpu_golden_and_shapes.py).
// Mode 2 (round 6) goes one step further: the ROWS themselves are evaluated in long double (DualL<7>: the same expressions as the Dual<7> rows, 64-bit
// mantissa), the Huber weight and the products in long double too — "the value the reference's formulas have", against which BOTH double evaluations
// (this file's duals and the device's fused chain) are measured: tests/test_gpu_golden_and_shapes.py holds the device to the truth, not to this
// file's own rounding.
static int g_exact_sums = 0;

void EvalFactors(const Oracle& O, const iba_params& prm, const double* x, iba_normal_out& out, bool multithread = false) {
    std::memset(&out, 0, sizeof(out));
    const long n = (long)O.factors.size();
    (void)multithread;
    if (g_exact_sums == 2) {
        using L7 = DualL<7>;
        long double LH[49] = {0}, Lb[7] = {0}, Lcost = 0, Lchi2 = 0;
        L7 xl[7]; for (int i = 0; i < 7; ++i) xl[i] = L7(x[i], i);
        std::vector<L7> e(64);
        for (long fi = 0; fi < n; ++fi) {
            const Factor& f = O.factors[fi];
            const int rows = f.rows();
            if (rows > 64) continue;
            eval_any_factor<L7>(f, xl, e.data());
            long double s = 0; for (int i = 0; i < rows; ++i) s += e[i].a * e[i].a;
            const long double a = (f.kind == 0 || f.kind == 3) ? prm.robust_kernel_delta : prm.robust_kernel_3ddelta, bb = a * a;
            long double rho0 = s, rho1 = 1.0L;
            if (s > bb) { const long double rr = sqrtl(s); rho0 = 2.0L * a * rr - bb; rho1 = a / rr; }
            Lcost += 0.5L * rho0; Lchi2 += s;
            for (int i = 0; i < rows; ++i)
                for (int q = 0; q < 7; ++q) {
                    Lb[q] += rho1 * e[i].v[q] * e[i].a;
                    for (int c = 0; c < 7; ++c) LH[q * 7 + c] += rho1 * e[i].v[q] * e[i].v[c];
                }
            out.n_residuals += rows;
            if (f.kind == 0 || f.kind == 3) out.n_factor_3d2d++; else if (f.kind == 1) out.n_factor_p2pl++; else out.n_factor_p2pt++;
        }
        for (int i = 0; i < 49; ++i) out.H[i] = (double)LH[i];
        for (int i = 0; i < 7; ++i) out.b[i] = (double)Lb[i];
        out.cost = (double)Lcost; out.chi2 = (double)Lchi2;
        out.frames_used = O.bp_frames_used; out.n_corr = O.bp_n_corr;
        return;
    }
    if (g_exact_sums) {
        long double LH[49] = {0}, Lb[7] = {0}, Lcost = 0, Lchi2 = 0;
        double r[64], J[64 * 7];
        for (long fi = 0; fi < n; ++fi) {
            const Factor& f = O.factors[fi];
            const int rows = f.rows();
            if (rows > 64) continue;
            eval_factor(f, x, r, J);
            double s = 0; for (int i = 0; i < rows; ++i) s += r[i] * r[i];
            double rho0, rho1; huber((f.kind == 0 || f.kind == 3) ? prm.robust_kernel_delta : prm.robust_kernel_3ddelta, s, rho0, rho1);
            Lcost += 0.5 * rho0; Lchi2 += s;
            for (int i = 0; i < rows; ++i)
                for (int a = 0; a < 7; ++a) {
                    Lb[a] += rho1 * J[i * 7 + a] * r[i];
                    for (int c = 0; c < 7; ++c) LH[a * 7 + c] += rho1 * J[i * 7 + a] * J[i * 7 + c];
                }
            out.n_residuals += rows;
            if (f.kind == 0 || f.kind == 3) out.n_factor_3d2d++; else if (f.kind == 1) out.n_factor_p2pl++; else out.n_factor_p2pt++;
        }
        for (int i = 0; i < 49; ++i) out.H[i] = (double)LH[i];
        for (int i = 0; i < 7; ++i) out.b[i] = (double)Lb[i];
        out.cost = (double)Lcost; out.chi2 = (double)Lchi2;
        out.frames_used = O.bp_frames_used; out.n_corr = O.bp_n_corr;
        return;
    }
#pragma omp parallel if (multithread)
    {
        iba_normal_out loc; std::memset(&loc, 0, sizeof(loc));
        double r[64], J[64 * 7];
#pragma omp for schedule(static)
        for (long fi = 0; fi < n; ++fi) {
            const Factor& f = O.factors[fi];
            const int rows = f.rows();
            if (rows > 64) continue;
            eval_factor(f, x, r, J);
            double s = 0; for (int i = 0; i < rows; ++i) s += r[i] * r[i];
            double rho0, rho1; huber((f.kind == 0 || f.kind == 3) ? prm.robust_kernel_delta : prm.robust_kernel_3ddelta, s, rho0, rho1);
            loc.cost += 0.5 * rho0; loc.chi2 += s;
            for (int i = 0; i < rows; ++i)
                for (int a = 0; a < 7; ++a) {
                    loc.b[a] += rho1 * J[i * 7 + a] * r[i];
                    for (int c = 0; c < 7; ++c) loc.H[a * 7 + c] += rho1 * J[i * 7 + a] * J[i * 7 + c];
                }
            loc.n_residuals += rows;
            if (f.kind == 0 || f.kind == 3) loc.n_factor_3d2d++; else if (f.kind == 1) loc.n_factor_p2pl++; else loc.n_factor_p2pt++;
        }
#pragma omp critical
        {
            for (int i = 0; i < 49; ++i) out.H[i] += loc.H[i];
            for (int i = 0; i < 7; ++i) out.b[i] += loc.b[i];
            out.cost += loc.cost; out.chi2 += loc.chi2; out.n_residuals += loc.n_residuals;
            out.n_factor_3d2d += loc.n_factor_3d2d; out.n_factor_p2pl += loc.n_factor_p2pl; out.n_factor_p2pt += loc.n_factor_p2pt;
        }
    }
    out.frames_used = O.bp_frames_used; out.n_corr = O.bp_n_corr;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C interface (ctypes from tests / bench cpu_baseline)
// ------------------------------------------------------------------------------------------------
extern "C" {

void* oracle_create(const iba_problem_desc* d, int leaf2d, int leaf3d) {
    auto* O = new Oracle; O->leaf2d = leaf2d; O->leaf3d = leaf3d;
    const int F = d->n_frames;
    O->frames.resize(F);
    for (int f = 0; f < F; ++f) {
        Frame& fr = O->frames[f];
        const uint64_t p0 = d->pt_offset[f], p1 = d->pt_offset[f + 1];
        fr.pts.resize(3 * (p1 - p0));
        for (uint64_t i = 0; i < 3 * (p1 - p0); ++i) fr.pts[i] = (double)d->pts_xyz[3 * p0 + i];
        const double* in = d->intrinsics + 6 * f;
        fr.fx = in[0]; fr.fy = in[1]; fr.cx = in[2]; fr.cy = in[3]; fr.W = in[4]; fr.H = in[5];
        const uint64_t k0 = d->kp_offset[f], k1 = d->kp_offset[f + 1];
        fr.kp_uv.assign(d->kp_uv + 2 * k0, d->kp_uv + 2 * k1);
        fr.has_mp.assign(d->kp_has_mappoint + k0, d->kp_has_mappoint + k1);
        fr.mp_w.assign(d->kp_mappoint_w + 3 * k0, d->kp_mappoint_w + 3 * k1);
        fr.Tcw = iso_from_f32(d->Tcw + 12 * f);
        fr.Tc_next = iso_from_f32(d->Tc_next + 12 * f);
        fr.Tl_next = iso_from_f64(d->Tl_next + 12 * f);
        for (uint64_t s = d->covis_offset[f]; s < d->covis_offset[f + 1]; ++s) {
            Covis cv; cv.frame = d->covis_frame[s]; cv.rel = iso_from_f32(d->covis_relpose + 12 * s);
            for (uint64_t m = d->match_offset[s]; m < d->match_offset[s + 1]; ++m) cv.kptmap[d->match_kp_ref[m]] = d->match_kp_covis[m];
            fr.covis.push_back(std::move(cv));
        }
    }
#pragma omp parallel for schedule(static)
    for (int f = 0; f < F; ++f) {  // iba_global.cpp:361-367
        Frame& fr = O->frames[f];
        fr.tree.reset(new KDTree3D(fr.pts.data(), fr.P(), (size_t)leaf3d));
    }
    return O;
}
void oracle_destroy(void* h) { delete (Oracle*)h; }

int oracle_eval_cost(void* h, const iba_params* p, const double* x, int B, iba_cost_out* out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int b = 0; b < B; ++b) BAError(*(Oracle*)h, *p, x + 7 * b, nthreads > 1, out[b]);
    return 0;
}
int oracle_eval_cost_raw(void* h, const iba_params* p, const double* x, int f_begin, int f_end, double* raw12) {
    iba_cost_out tmp; BAError(*(Oracle*)h, *p, x, false, tmp, f_begin, f_end, raw12); return 0;
}
int oracle_eval_bbo(void* h, const iba_params* p, const double* x, int B, double he_threshold, double valid_rate, iba_bbo* out, int nthreads) {
    for (int b = 0; b < B; ++b) {  // iba_global.cpp:385-392
        iba_cost_out c; oracle_eval_cost(h, p, x + 7 * b, 1, &c, nthreads);
        out[b].f = c.f1 * p->err_weight[0] + c.f2 * p->err_weight[1];
        out[b].c1 = c.C - he_threshold; out[b].c2 = -c.C - he_threshold;
        out[b].c3 = valid_rate - static_cast<double>(c.valid_cnt_3d_2d) / (c.cnt_3d_2d + 1);
    }
    return 0;
}
int oracle_get_correspondences(void* h, const iba_params* p, const double* x, int frame, uint32_t* kp, uint32_t* pt, int cap, int* n) {
    Oracle& O = *(Oracle*)h;
    M3d R; V3d t; double s; Sim3Exp<double>(x, R, t, s);
    std::vector<double> PC; transform_cloud(O.frames[frame], Iso3{R, t}, PC);
    CorrSet cs; FindProjectCorrespondences(PC, O.frames[frame], O.leaf2d, p->max_pixel_dist, cs);
    *n = (int)cs.size();
    for (int i = 0; i < (int)cs.size() && i < cap; ++i) { kp[i] = cs[i].first; pt[i] = cs[i].second; }
    return 0;
}
int oracle_build_problem(void* h, const iba_params* p, const double* x) { BuildProblem(*(Oracle*)h, *p, x); return 0; }
int oracle_eval_factors(void* h, const iba_params* p, const double* x, int B, iba_normal_out* out) {
    for (int b = 0; b < B; ++b) EvalFactors(*(Oracle*)h, *p, x + 7 * b, out[b]);
    return 0;
}
void oracle_set_exact_sums(int on) { g_exact_sums = on; }   // 0 off, 1 long-double sums of the double rows, 2 long-double rows and sums
void oracle_set_frame_range(void* h, int f_begin, int f_end) { ((Oracle*)h)->bp_f_begin = f_begin; ((Oracle*)h)->bp_f_end = f_end; }
int oracle_eval_normal(void* h, const iba_params* p, const double* x, int B, iba_normal_out* out, int nthreads) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int b = 0; b < B; ++b) { BuildProblem(*(Oracle*)h, *p, x + 7 * b, nthreads > 1); EvalFactors(*(Oracle*)h, *p, x + 7 * b, out[b], nthreads > 1); }
    return 0;
}
int oracle_eval_residuals(void* h, const double* x, double* r, double* J, int32_t* block_id, int32_t* block_kind, int32_t* block_frame_kp, int64_t* n_rows) {
    Oracle& O = *(Oracle*)h;
    int64_t rows = 0; for (auto const& f : O.factors) rows += f.rows();
    *n_rows = rows;
    if (!r) return 0;
    int64_t at = 0; int32_t bid = 0;
    for (auto const& f : O.factors) {
        eval_factor(f, x, r + at, J ? J + 7 * at : nullptr);
        for (int i = 0; i < f.rows(); ++i) {
            if (block_id) block_id[at + i] = bid;
            if (block_kind) block_kind[at + i] = f.kind;
            if (block_frame_kp) { block_frame_kp[2 * (at + i)] = f.frame; block_frame_kp[2 * (at + i) + 1] = f.kp; }
        }
        at += f.rows(); ++bid;
    }
    return 0;
}
// IBAPlaneEdge (g2o twin): same math, error zero-padded to 20 (IBACalib.hpp:133-137). block = index of a
// kind-0 factor of the frozen problem.
int oracle_eval_plane_edge20(void* h, int64_t factor_index, const double* x, double* err20, double* J20x7) {
    Oracle& O = *(Oracle*)h;
    if (factor_index < 0 || factor_index >= (int64_t)O.factors.size()) return 1;
    const Factor& f = O.factors[factor_index];
    if (f.kind != 0 || f.u1.size() > 10) return 1;
    std::memset(err20, 0, 20 * sizeof(double)); if (J20x7) std::memset(J20x7, 0, 140 * sizeof(double));
    eval_factor(f, x, err20, J20x7);
    return 0;
}
// How well-conditioned each residual block of the frozen problem is at x, as a number in (0, 1]: for an IBA_PlaneFactor the two
// quotients of IBACalib2.hpp:163-183 are differences of products divided by a sum that may cancel —
//   Z0 = (n_c . p_c) / den, den = Cxz n_cx + Cyz n_cy + n_cz   -> |den| / (|Cxz n_cx| + |Cyz n_cy| + |n_cz|)   (the viewing ray lies in the plane: 0)
//   u = fx P1x / P1z + cx                                       -> |P1z| / (|R20 X0| + |R21 Y0| + |R22 Z0| + |s t_z|) per covisible keyframe
// the smallest of them; 1 for the 3d-3d blocks (no quotient). Two double evaluations of such a block (the device's chain rule, the
// Dual<7> arithmetic here) may differ by about eps / cond relative to the block's scale: the parity tests use it to tell
// conditioning from defects (tools/soak_parity.py, tests/test_gpu_conditioning.py).
int oracle_block_conditioning(void* h, const double* x, double* cond, int64_t* n_blocks) {
    Oracle& O = *(Oracle*)h;
    *n_blocks = (int64_t)O.factors.size();
    if (!cond) return 0;
    M3d Rcl; V3d tcl; double s;
    Sim3Exp<double>(x, Rcl, tcl, s);
    int64_t at = 0;
    for (auto const& f : O.factors) {
        double c = 1.0;
        if (f.kind == 0) {
            const V3d p0c = mul(Rcl, f.p0) + tcl, n0c = mul(Rcl, f.n0);
            const double Cxz = (f.u0 - f.cx) / f.fx, Cyz = (f.v0 - f.cy) / f.fy;
            const double den = Cxz * n0c.x + Cyz * n0c.y + n0c.z, mag = std::fabs(Cxz * n0c.x) + std::fabs(Cyz * n0c.y) + std::fabs(n0c.z);
            c = mag > 0 ? std::fabs(den) / mag : 0.0;
            const double Z0 = dot(n0c, p0c) / den, X0 = Cxz * Z0, Y0 = Cyz * Z0;
            for (size_t i = 0; i < f.u1.size(); ++i) {
                const M3d& R = f.R[i];
                const double tz = f.t[i].z * s;
                const double z = R.m[6] * X0 + R.m[7] * Y0 + R.m[8] * Z0 + tz, zm = std::fabs(R.m[6] * X0) + std::fabs(R.m[7] * Y0) + std::fabs(R.m[8] * Z0) + std::fabs(tz);
                c = std::min(c, zm > 0 ? std::fabs(z) / zm : 0.0);
            }
            if (!(c == c)) c = 0.0;
        }
        cond[at++] = c;
    }
    return 0;
}
// The FORWARD ERROR of this file's own double evaluation of every residual block of the frozen problem at x, measured against the
// same formulas in long double (DualL<7>, 64-bit mantissa): max over the block's rows of |r, J (double) - r, J (long double)|
// divided by the block's scale max(|J|, |r|, 1). Two correct double evaluations of a block (the device's analytic chain rule and the
// Dual<7> arithmetic here) cannot be expected to agree better than a small multiple of this: the parity tests bound the device's
// deviation by it (tests/parity_explain.py), instead of by a tolerance picked to make them pass.
// The block's SENSITIVITY to its inputs' last bits: how far its long-double rows move, relative to the block's scale, when x is moved by
// one unit in the last place per component (four sign patterns, the largest). The device derives R, t and their derivatives from x
// with its own (Jet<6>) arithmetic, this file with Dual<7>: each is the exact result for an x a few ulps away, so two correct
// evaluations of an ill-conditioned block may differ by a small multiple of this however accurate each of them is internally —
// the forward error of ONE evaluation (above) is a single draw of the same quantity and can come out small by luck.
static double block_sensitivity(const Factor& f, const double* x) {
    using L7 = DualL<7>;
    const int rows = f.rows();
    std::vector<L7> e0(rows), e1(rows);
    L7 xl[7]; for (int i = 0; i < 7; ++i) xl[i] = L7(x[i], i);
    eval_any_factor<L7>(f, xl, e0.data());
    long double scale = 1.0L;
    for (int i = 0; i < rows; ++i) { scale = std::max(scale, fabsl(e0[i].a)); for (int c = 0; c < 7; ++c) scale = std::max(scale, fabsl(e0[i].v[c])); }
    long double worst = 0.0L;
    for (int pat = 0; pat < 4; ++pat) {
        for (int i = 0; i < 7; ++i) {
            const int sgn = ((pat == 0) || (pat == 1 && (i & 1)) || (pat == 2 && (i % 3 == 0)) || (pat == 3 && i < 3)) ? 1 : -1;
            xl[i] = L7(x[i], i); xl[i].a = (long double)x[i] * (1.0L + (long double)sgn * 0x1p-52L);
        }
        eval_any_factor<L7>(f, xl, e1.data());
        for (int i = 0; i < rows; ++i) {
            worst = std::max(worst, fabsl(e1[i].a - e0[i].a));
            for (int c = 0; c < 7; ++c) worst = std::max(worst, fabsl(e1[i].v[c] - e0[i].v[c]));
        }
    }
    const long double sres = worst / scale;
    return (sres == sres) ? (double)sres : 1.0;
}
int oracle_block_sensitivity(void* h, const double* x, double* sens, int64_t* n_blocks) {
    Oracle& O = *(Oracle*)h;
    *n_blocks = (int64_t)O.factors.size();
    if (!sens) return 0;
    int64_t at = 0;
    for (auto const& f : O.factors) sens[at++] = block_sensitivity(f, x);
    return 0;
}
int oracle_block_forward_error(void* h, const double* x, double* err, int64_t* n_blocks) {
    Oracle& O = *(Oracle*)h;
    *n_blocks = (int64_t)O.factors.size();
    if (!err) return 0;
    using D7 = Dual<7>; using L7 = DualL<7>;
    D7 xd[7]; L7 xl[7];
    for (int i = 0; i < 7; ++i) { xd[i] = D7(x[i], i); xl[i] = L7(x[i], i); }
    int64_t at = 0;
    for (auto const& f : O.factors) {
        const int rows = f.rows();
        std::vector<D7> ed(rows); std::vector<L7> el(rows);
        eval_any_factor<D7>(f, xd, ed.data()); eval_any_factor<L7>(f, xl, el.data());
        long double scale = 1.0L, dev = 0.0L;
        for (int i = 0; i < rows; ++i) {
            scale = std::max(scale, fabsl(el[i].a)); dev = std::max(dev, fabsl((long double)ed[i].a - el[i].a));
            for (int c = 0; c < 7; ++c) { scale = std::max(scale, fabsl(el[i].v[c])); dev = std::max(dev, fabsl((long double)ed[i].v[c] - el[i].v[c])); }
        }
        const long double e = dev / scale;
        err[at++] = (e == e) ? (double)e : 1.0;   // NaN (a degenerate block): "no accuracy at all"
    }
    return 0;
}
// The analytic chain rule of an IBA_PlaneFactor block in the device kernel's operation order (plane_factor_core, csrc/iba_kernels.hpp),
// in double, from explicit inputs: R, t and their derivatives (dR[k] for k < 3, dt[k] for k < 6), the scale s, the plane normal n0.
// variant: how the cancelling factor (ax - xz az) is formed: 0 as the kernel does, 1 from the exact identity (ax tz - az tx) / P1z.
// out: rows x 8 doubles (r, J[7]).
static inline double kfdot3(double a0, double b0, double a1, double b1, double a2, double b2) { return std::fma(a2, b2, std::fma(a1, b1, a0 * b0)); }
static inline double kfdot3c(double a0, double b0, double a1, double b1, double a2, double b2, double c) { return std::fma(a2, b2, std::fma(a1, b1, std::fma(a0, b0, c))); }
static void plane_block_kernel_order(const Factor& f, const double* R, const double* t, const double (*dR)[9], const double (*dt)[3], double s, const double* n0, int variant, double* out) {
    // round 6: the kernel's chain is a fixed sequence of IEEE operations with EXPLICIT fused multiply-adds (csrc/iba_kernels.hpp, plane_core:
    // fdot3 / fdot3c); std::fma is the same correctly rounded operation, so this function reproduces the device's rows bit for bit
    const double p0[3] = {f.p0.x, f.p0.y, f.p0.z};
    double p0c[3], n0c[3];
    for (int r = 0; r < 3; ++r) { p0c[r] = kfdot3c(R[r * 3], p0[0], R[r * 3 + 1], p0[1], R[r * 3 + 2], p0[2], t[r]); n0c[r] = kfdot3(R[r * 3], n0[0], R[r * 3 + 1], n0[1], R[r * 3 + 2], n0[2]); }
    const double Cxz = (f.u0 - f.cx) / f.fx, Cyz = (f.v0 - f.cy) / f.fy;
    const double num = kfdot3(n0c[0], p0c[0], n0c[1], p0c[1], n0c[2], p0c[2]);
    const double den = std::fma(Cyz, n0c[1], std::fma(Cxz, n0c[0], n0c[2]));
    const double iden = 1.0 / den, Z0 = num * iden;
    double z6[6];
    for (int kk = 0; kk < 3; ++kk) {
        double dn[3], dq[3];
        for (int r = 0; r < 3; ++r) {
            dn[r] = kfdot3(dR[kk][r * 3], n0[0], dR[kk][r * 3 + 1], n0[1], dR[kk][r * 3 + 2], n0[2]);
            dq[r] = kfdot3c(dR[kk][r * 3], p0[0], dR[kk][r * 3 + 1], p0[1], dR[kk][r * 3 + 2], p0[2], dt[kk][r]);
        }
        const double dnum = kfdot3c(n0c[0], dq[0], n0c[1], dq[1], n0c[2], dq[2], kfdot3(dn[0], p0c[0], dn[1], p0c[1], dn[2], p0c[2]));
        const double dden = std::fma(Cyz, dn[1], std::fma(Cxz, dn[0], dn[2]));
        z6[kk] = std::fma(-Z0, dden, dnum) * iden;
    }
    for (int kk = 3; kk < 6; ++kk) z6[kk] = kfdot3(n0c[0], dt[kk][0], n0c[1], dt[kk][1], n0c[2], dt[kk][2]) * iden;
    const double P0x = Cxz * Z0, P0y = Cyz * Z0, P0z = Z0;
    for (size_t i = 0; i < f.u1.size(); ++i) {
        const double* rel = f.R[i].m;
        const double t3[3] = {f.t[i].x, f.t[i].y, f.t[i].z};
        const double tx = t3[0] * s, ty = t3[1] * s, tz = t3[2] * s;
        const double P1x = kfdot3c(rel[0], P0x, rel[1], P0y, rel[2], P0z, tx), P1y = kfdot3c(rel[3], P0x, rel[4], P0y, rel[5], P0z, ty), P1z = kfdot3c(rel[6], P0x, rel[7], P0y, rel[8], P0z, tz);
        const double iz = 1.0 / P1z, xz = P1x * iz, yz = P1y * iz;   // (the kernel forms the residual from the same reciprocal as its derivatives)
        const double ru = std::fma(f.fx, xz, f.cx) - f.u1[i], rv = std::fma(f.fy, yz, f.cy) - f.v1[i];
        const double ax = std::fma(rel[1], Cyz, std::fma(rel[0], Cxz, rel[2])), ay = std::fma(rel[4], Cyz, std::fma(rel[3], Cxz, rel[5])), az = std::fma(rel[7], Cyz, std::fma(rel[6], Cxz, rel[8]));
        double cu, cv;
        if (variant == 0) { cu = std::fma(-xz, az, ax); cv = std::fma(-yz, az, ay); }
        else { cu = (ax * tz - az * tx) * iz; cv = (ay * tz - az * ty) * iz; }   // P1 = Z0 a + s t  =>  ax - (P1x / P1z) az = (ax tz - az tx) / P1z exactly
        const double fxiz = f.fx * iz, fyiz = f.fy * iz;
        const double gu = fxiz * cu, gv = fyiz * cv;
        const double hu = fxiz * std::fma(-xz, t3[2], t3[0]), hv = fyiz * std::fma(-yz, t3[2], t3[1]);
        double* o0 = out + 8 * (2 * i); double* o1 = out + 8 * (2 * i + 1);
        o0[0] = ru; o1[0] = rv;
        for (int k = 0; k < 6; ++k) { o0[1 + k] = gu * z6[k]; o1[1 + k] = gv * z6[k]; }
        o0[7] = hu; o1[7] = hv;
    }
}
struct CandInputs { double R[9], t[3], dR[6][9], dt[6][3]; };
static CandInputs cand_inputs_of(const double* x) {   // R, t and their derivatives from the duals of Sim3Exp (the device's host math computes them with Jet<6>)
    using D7 = Dual<7>;
    D7 xd[7]; for (int i = 0; i < 7; ++i) xd[i] = D7(x[i], i);
    M3<D7> Rd; V3<D7> td; D7 sd;
    Sim3Exp<D7>(xd, Rd, td, sd);
    CandInputs c;
    for (int i = 0; i < 9; ++i) { c.R[i] = Rd.m[i].a; for (int k = 0; k < 6; ++k) c.dR[k][i] = Rd.m[i].v[k]; }
    for (int i = 0; i < 3; ++i) { c.t[i] = td[i].a; for (int k = 0; k < 6; ++k) c.dt[k][i] = td[i].v[k]; }
    return c;
}
// One IBA_PlaneFactor block of the frozen problem evaluated THREE ways at x (investigation tool for the parity tests): out[0] = the
// Dual<7> arithmetic of this file (double), out[1] = the same in long double rounded to double, out[2] = the kernel's operation order
// in double (plane_block_kernel_order). Each out[i] holds rows x 8 doubles: r, J[7].
int oracle_block_three_ways(void* h, int64_t block, const double* x, int variant, double* out0, double* out1, double* out2) {
    Oracle& O = *(Oracle*)h;
    if (block < 0 || block >= (int64_t)O.factors.size() || O.factors[block].kind != 0) return 1;
    const Factor& f = O.factors[block];
    using D7 = Dual<7>; using L7 = DualL<7>;
    D7 xd[7]; L7 xl[7];
    for (int i = 0; i < 7; ++i) { xd[i] = D7(x[i], i); xl[i] = L7(x[i], i); }
    const int rows = f.rows();
    std::vector<D7> ed(rows); std::vector<L7> el(rows);
    eval_plane_factor<D7>(f, xd, ed.data()); eval_plane_factor<L7>(f, xl, el.data());
    for (int i = 0; i < rows; ++i) { out0[8 * i] = ed[i].a; out1[8 * i] = (double)el[i].a; for (int c = 0; c < 7; ++c) { out0[8 * i + 1 + c] = ed[i].v[c]; out1[8 * i + 1 + c] = (double)el[i].v[c]; } }
    const CandInputs c = cand_inputs_of(x);
    const double n0[3] = {f.n0.x, f.n0.y, f.n0.z};
    plane_block_kernel_order(f, c.R, c.t, c.dR, c.dt, x[6], n0, variant, out2);
    return 0;
}
// A plane-factor block in the kernel's operation order from inputs GIVEN by the caller (the device's own: iba_debug_cand, iba_debug_plane):
// cand58 = R[9], t[3], dR[3][9], dt[6][3], s. out: rows x 8 (r, J[7]). The device's rows must equal this bit for bit.
int oracle_block_kernel_order(void* h, int64_t block, const double* cand58, const double n0[3], double* out) {
    Oracle& O = *(Oracle*)h;
    if (block < 0 || block >= (int64_t)O.factors.size() || O.factors[block].kind != 0) return 1;
    const Factor& f = O.factors[block];
    double dR[6][9] = {{0}}, dt[6][3];
    std::memcpy(dR, cand58 + 12, 216); std::memcpy(dt, cand58 + 39, 144);
    plane_block_kernel_order(f, cand58, cand58 + 9, dR, dt, cand58[57], n0, 0, out);
    return 0;
}
// INPUT SENSITIVITY of a plane-factor block: the device and this file each derive R, t and their derivatives from x with their own
// dual arithmetic (Jet<6> / Dual<7>: tools/device_vs_simulation.py finds the two bit-identical on some candidates and a last bit apart
// on others) and each fit the plane with their own libm; both then run the SAME formulas (device == plane_block_kernel_order bit for
// bit when the inputs are equal). How far apart two such evaluations may be is therefore a property of the block: the largest change
// of its rows, relative to the block's scale, when every derived input (entries of R, t, dR, dt, n0) moves by one unit in the last
// place, signs drawn at random, eight draws. A block whose viewing ray lies almost in its plane amplifies exactly these last bits.
static double plane_block_input_sensitivity(const Factor& f, const double* x) {
    if (f.kind != 0) return 0.0;
    const int rows = f.rows();
    const CandInputs c0 = cand_inputs_of(x);
    const double n0[3] = {f.n0.x, f.n0.y, f.n0.z};
    std::vector<double> base((size_t)rows * 8), pert((size_t)rows * 8);
    plane_block_kernel_order(f, c0.R, c0.t, c0.dR, c0.dt, x[6], n0, 0, base.data());
    double scale = 1.0;
    for (double v : base) scale = std::max(scale, std::fabs(v));
    uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)rows;
    auto flip = [&](double v) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return v * (1.0 + ((st >> 33) & 1ull ? 0x1p-52 : -0x1p-52)); };
    double worst = 0.0;
    for (int draw = 0; draw < 8; ++draw) {
        CandInputs c = c0; double n[3];
        for (double& v : c.R) v = flip(v);
        for (double& v : c.t) v = flip(v);
        for (int k = 0; k < 6; ++k) { for (double& v : c.dR[k]) v = flip(v); for (double& v : c.dt[k]) v = flip(v); }
        for (int i = 0; i < 3; ++i) n[i] = flip(n0[i]);
        plane_block_kernel_order(f, c.R, c.t, c.dR, c.dt, x[6], n, 0, pert.data());
        for (size_t i = 0; i < base.size(); ++i) { const double d = std::fabs(pert[i] - base[i]); if (d > worst || d != d) worst = (d == d) ? d : scale; }
    }
    return worst / scale;
}
int oracle_block_input_sensitivity(void* h, const double* x, double* sens, int64_t* n_blocks) {
    Oracle& O = *(Oracle*)h;
    *n_blocks = (int64_t)O.factors.size();
    if (!sens) return 0;
    int64_t at = 0;
    for (auto const& f : O.factors) sens[at++] = plane_block_input_sensitivity(f, x);
    return 0;
}
// The plane normal a residual block of the frozen problem carries (kind 0: n0 of IBA_PlaneFactor, kind 1: n of Point2Plane_Factor), and
// the block's rows with ANOTHER normal substituted: the device fits its planes with its own libm (acos / cos of the closed-form
// eigen-solver), so its normal differs from this file's in the last bits, and an ill-conditioned block amplifies exactly that input
// difference. With the device's normal substituted here the two evaluations see the same inputs (tests/parity_explain.py).
int oracle_block_normal(void* h, int64_t block, double n[3], double point[3], int32_t* frame) {   // + the scan point the plane was fitted at
    Oracle& O = *(Oracle*)h;
    if (block < 0 || block >= (int64_t)O.factors.size()) return 1;
    const Factor& f = O.factors[block];
    if (f.kind == 2 || f.kind == 3) return 1;   // (no normal in a point-to-point block or an IBATestEdge)
    const V3d& v = f.kind == 0 ? f.n0 : f.n;
    const V3d& q = f.kind == 0 ? f.p0 : f.Q;
    n[0] = v.x; n[1] = v.y; n[2] = v.z;
    point[0] = q.x; point[1] = q.y; point[2] = q.z;
    *frame = f.frame;
    return 0;
}
int oracle_block_rows_with_normal(void* h, int64_t block, const double* x, const double n[3], double* r, double* J, double* fwd_err) {
    Oracle& O = *(Oracle*)h;
    if (block < 0 || block >= (int64_t)O.factors.size()) return 1;
    Factor f = O.factors[block];
    if (f.kind == 2 || f.kind == 3) return 1;   // (no normal in a point-to-point block or an IBATestEdge)
    (f.kind == 0 ? f.n0 : f.n) = V3d{n[0], n[1], n[2]};
    eval_factor(f, x, r, J);
    if (fwd_err) {   // the forward error of THIS evaluation (double against long double), relative to the block's scale
        using L7 = DualL<7>;
        L7 xl[7]; for (int i = 0; i < 7; ++i) xl[i] = L7(x[i], i);
        const int rows = f.rows();
        std::vector<L7> el(rows);
        eval_any_factor<L7>(f, xl, el.data());
        long double scale = 1.0L, dev = 0.0L;
        for (int i = 0; i < rows; ++i) {
            scale = std::max(scale, fabsl(el[i].a)); dev = std::max(dev, fabsl((long double)r[i] - el[i].a));
            for (int c = 0; c < 7; ++c) { scale = std::max(scale, fabsl(el[i].v[c])); dev = std::max(dev, fabsl((long double)J[i * 7 + c] - el[i].v[c])); }
        }
        *fwd_err = std::max(std::max((double)(dev / scale), block_sensitivity(f, x)), plane_block_input_sensitivity(f, x));   // the yardstick of this evaluation: its forward error or the block's sensitivities, the largest
    }
    return 0;
}
// x-independent local plane record at one scan point (for the GPU plane-cache parity test)
int oracle_plane_at(void* h, int frame, uint32_t pt_idx, double radius, int max_pts, int32_t* k_out, double* far_d2, double* normal3, double* reg_err_sum) {
    Oracle& O = *(Oracle*)h; const Frame& f = O.frames[frame];
    std::vector<uint32_t> idx; std::vector<double> sq; size_t k;
    const double* c = &f.pts[3 * (size_t)pt_idx];
    knn_clip(f, c, max_pts, radius, idx, sq, k);
    *k_out = (int32_t)k; *far_d2 = k ? sq[k - 1] : 0.0;
    V3d n; double re; normal_and_reg(f, idx, V3d{c[0], c[1], c[2]}, n, re);
    normal3[0] = n.x; normal3[1] = n.y; normal3[2] = n.z; *reg_err_sum = re;
    return 0;
}
// How far the plane normal at a scan point moves when the entries of its covariance matrix move by their own measured accuracy (the
// double evaluation against the same formula in long double, at least one unit in the last place; signs drawn at random, eight
// draws; the largest component difference, sign-aligned). The closed-form eigen-solver (FastEigen3x3_EV:
// acos / cos of a ratio of nearly equal quantities) amplifies the last bits of what it is given — and of its own libm calls, which
// the device evaluates with another libm: two correct evaluations of it may differ by a small multiple of this
// (tests/parity_explain.py bounds the device-vs-oracle normal difference by it instead of by a fixed tolerance).
int oracle_plane_normal_sensitivity(void* h, int frame, uint32_t pt_idx, double radius, int max_pts, double* sens) {
    Oracle& O = *(Oracle*)h; const Frame& f = O.frames[frame];
    std::vector<uint32_t> idx; std::vector<double> sq; size_t k;
    const double* c = &f.pts[3 * (size_t)pt_idx];
    knn_clip(f, c, max_pts, radius, idx, sq, k);
    const M3d cov = ComputeCovariance(f.pts.data(), idx.data(), idx.size());
    // the accuracy of the covariance itself: one-pass raw moments E[x x^T] - E[x] E[x]^T cancel (a neighbourhood 30 m from the sensor with
    // a spread of decimetres loses four to five digits), and two summation orders of the same neighbours — the device sums them in
    // its list order — differ by about that much: the same formula in long double is the measure
    long double cl[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t j : idx) {
        const long double x = f.pts[3 * (size_t)j], y = f.pts[3 * (size_t)j + 1], z = f.pts[3 * (size_t)j + 2];
        cl[0] += x; cl[1] += y; cl[2] += z; cl[3] += x * x; cl[4] += x * y; cl[5] += x * z; cl[6] += y * y; cl[7] += y * z; cl[8] += z * z;
    }
    const long double nn = idx.empty() ? 1.0L : (long double)idx.size();
    for (long double& v : cl) v /= nn;
    const long double covl[6] = {cl[3] - cl[0] * cl[0], cl[6] - cl[1] * cl[1], cl[8] - cl[2] * cl[2], cl[4] - cl[0] * cl[1], cl[5] - cl[0] * cl[2], cl[7] - cl[1] * cl[2]};
    const double covd[6] = {cov(0, 0), cov(1, 1), cov(2, 2), cov(0, 1), cov(0, 2), cov(1, 2)};
    double err[6];
    // (the measured difference is ONE draw and can come out small by luck — seed 90431, r04: 2.8e-13 against a device normal 8.4e-11 away;
    //  no double evaluation of E[ab] - E[a] E[b] is better than a few units in the last place of the two terms it subtracts)
    const long double raw[6][2] = {{cl[3], cl[0] * cl[0]}, {cl[6], cl[1] * cl[1]}, {cl[8], cl[2] * cl[2]}, {cl[4], cl[0] * cl[1]}, {cl[5], cl[0] * cl[2]}, {cl[7], cl[1] * cl[2]}};
    for (int i = 0; i < 6; ++i) err[i] = std::max((double)fabsl((long double)covd[i] - covl[i]), 8.0 * 0x1p-52 * (double)(fabsl(raw[i][0]) + fabsl(raw[i][1]))) + std::fabs(covd[i]) * 0x1p-52;
    double ev[3];
    const V3d n0 = normalized(FastEigen3x3_EV(cov, ev));
    uint64_t st = 0x2545F4914F6CDD1Dull ^ ((uint64_t)pt_idx << 20) ^ (uint64_t)frame;
    int at = 0;
    auto flip = [&](double v) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; const double e = err[at++ % 6]; return v + ((st >> 33) & 1ull ? e : -e); };
    double worst = 0.0;
    for (int draw = 0; draw < 8; ++draw) {
        M3d cp = cov;
        at = 0;
        cp(0, 0) = flip(cov(0, 0)); cp(1, 1) = flip(cov(1, 1)); cp(2, 2) = flip(cov(2, 2));
        cp(0, 1) = flip(cov(0, 1)); cp(1, 0) = cp(0, 1); cp(0, 2) = flip(cov(0, 2)); cp(2, 0) = cp(0, 2); cp(1, 2) = flip(cov(1, 2)); cp(2, 1) = cp(1, 2);
        V3d n = normalized(FastEigen3x3_EV(cp, ev));
        if (dot(n, n0) < 0) n = V3d{-n.x, -n.y, -n.z};
        const double d = std::max(std::max(std::fabs(n.x - n0.x), std::fabs(n.y - n0.y)), std::fabs(n.z - n0.z));
        worst = (d == d) ? std::max(worst, d) : 1.0;
    }
    *sens = worst;
    return 0;
}
int oracle_num_factors(void* h) { return (int)((Oracle*)h)->factors.size(); }

// ---- unit-level exports for the known-answer tests ----
void oracle_sim3exp(const double* x, double* R9, double* t3, double* s) { M3d R; V3d t; Sim3Exp<double>(x, R, t, *s); std::memcpy(R9, R.m, 72); t3[0] = t.x; t3[1] = t.y; t3[2] = t.z; }
void oracle_se3exp(const double* x, double* R9, double* t3) { M3d R; V3d t; SE3Exp<double>(x, R, t); std::memcpy(R9, R.m, 72); t3[0] = t.x; t3[1] = t.y; t3[2] = t.z; }
void oracle_sim3exp_jet(const double* x, double* R9, double* t3, double* dR /*9x7*/, double* dt /*3x7*/) {
    using D7 = Dual<7>; D7 xd[7]; for (int i = 0; i < 7; ++i) xd[i] = D7(x[i], i);
    M3<D7> R; V3<D7> t; D7 s; Sim3Exp<D7>(xd, R, t, s);
    for (int i = 0; i < 9; ++i) { R9[i] = R.m[i].a; for (int c = 0; c < 7; ++c) dR[i * 7 + c] = R.m[i].v[c]; }
    for (int i = 0; i < 3; ++i) { t3[i] = t[i].a; for (int c = 0; c < 7; ++c) dt[i * 7 + c] = t[i].v[c]; }
}
void oracle_se3log(const double* R9, const double* t3, double* out6) { M3d R; std::memcpy(R.m, R9, 72); SE3Log(R, V3d{t3[0], t3[1], t3[2]}, out6); }
void oracle_covariance(const double* pts, const uint32_t* idx, uint64_t n, double* cov9) { M3d c = ComputeCovariance(pts, idx, n); std::memcpy(cov9, c.m, 72); }
void oracle_fast_eigen(const double* cov9, double* evec3, double* evals3) { M3d c; std::memcpy(c.m, cov9, 72); V3d v = FastEigen3x3_EV(c, evals3); evec3[0] = v.x; evec3[1] = v.y; evec3[2] = v.z; }
void oracle_huber(double a, double s, double* rho0, double* rho1) { huber(a, s, *rho0, *rho1); }
int oracle_knn(int dim, const double* pts, uint64_t n, int leaf, const double* queries, uint64_t nq, int k, uint32_t* out_idx, double* out_d2, int32_t* out_cnt) {
    auto run = [&](auto& tree, int D) {
        for (uint64_t q = 0; q < nq; ++q) {
            std::vector<uint32_t> indices(k); std::vector<double> sq(k, std::numeric_limits<double>::max());
            KNNResultSet rs((size_t)k); rs.init(indices.data(), sq.data());
            tree.findNeighbors(rs, queries + q * D);
            const int cnt = (int)rs.size(); out_cnt[q] = cnt;
            for (int j = 0; j < k; ++j) { out_idx[q * k + j] = j < cnt ? indices[j] : 0xFFFFFFFFu; out_d2[q * k + j] = j < cnt ? sq[j] : -1.0; }
        }
    };
    if (dim == 2) { KDTree<2> t(pts, n, (size_t)leaf); run(t, 2); return 0; }
    if (dim == 3) { KDTree<3> t(pts, n, (size_t)leaf); run(t, 3); return 0; }
    return 1;
}
// GeoCalib.h:18-33 computeCorrespondence restated on this file's kd tree (the restatement of the reference's nanoflann, pinned bit for bit against it:
// tests/test_oracle_kdtree.py): max_leaf 15 (:23), 1-NN (:25-28), kept when  sq_dist <= maxDistance  — squared against un-squared, as written (:29).
// nanoflann's knnSearch leaves sq_dist at the caller's initial 1000 and returns 0 only for an empty tree; the initial value is not a search radius.
int oracle_geo_correspondences(const double* src, uint64_t n_src, const double* tgt, uint64_t n_tgt, double maxDistance, uint32_t* out_src, uint32_t* out_tgt, int64_t* n_out) {
    *n_out = 0;
    if (n_tgt == 0) return 0;
    KDTree<3> tree(tgt, n_tgt, (size_t)15);
    int64_t n = 0;
    for (uint64_t i = 0; i < n_src; ++i) {
        uint32_t idx = 0; double sq = 1000;
        KNNResultSet rs((size_t)1); rs.init(&idx, &sq);
        tree.findNeighbors(rs, src + 3 * i);
        if (rs.size() > 0 && sq <= maxDistance) { out_src[n] = (uint32_t)i; out_tgt[n] = idx; ++n; }
    }
    *n_out = n;
    return 0;
}
int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
