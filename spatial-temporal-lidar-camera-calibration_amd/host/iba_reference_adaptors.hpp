// Reference-side adaptors: what a maintainer of gitouni/Spatial-Temporal-LiDAR-camera-Calibration adds to
// route the hot path through libiba_mi355x.so while keeping the existing call surface.
// This header needs the reference's own dependencies (Eigen, OpenCV, ORB_SLAM2 KeyFrame, optionally Ceres
// and g2o); it is NOT compiled in this repository's build (none of them exist in the image) — the product
// is the C-ABI library. Everything here is glue: packing + forwarding.
//
//   iba::PackedProblem / iba::pack()   KeyFrame*/PointClouds/vTwl -> iba_problem_desc (done ONCE)
//   iba::BAError(xvec, PointClouds, KdTrees, vTwl, KFIdMap, KeyFrames, iba_params, multiprocessing, verborse)
//                                      the reference's EXACT parameter list (iba_global.cpp:169-173; g2o::Vector7 form iba_func.cpp:179-183):
//                                      packs once — cache keyed on the addresses of PointClouds / KeyFrames — and forwards to
//                                      iba_eval_cost, so a call site (iba_global.cpp:372, 385; iba_func.cpp:463) compiles unchanged
//                                      once the reference's own BAError body is removed and this header is included
//   iba::BAError(xvec, Evaluator&)     the same on an evaluator the caller owns
//   iba::IbaAggregateCostFunction      ceres::SizedCostFunction<8, 7> replacing the blocks BuildProblem() adds (iba_local.cpp:263-308)
//   iba::IbaAggregateEdge              g2o::BaseUnaryEdge<8, ..., VertexSim3> (same vertex type as IBACalib.hpp:74)
// The two solver adaptors only forward to iba_eval_whitened (the math is behind the C-ABI and tested there).
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <tuple>
#include <unordered_map>
#include <utility>
#include <vector>

#include <Eigen/Dense>
#include <opencv2/core.hpp>

#include "iba_mi355x.h"
#include "orb_slam/include/KeyFrame.h"
#include "orb_slam/include/MapPoint.h"

namespace iba {

typedef std::vector<Eigen::Vector3d> VecVector3d;   // pointcloud.h:13

struct PackedProblem {
    std::vector<uint64_t> pt_offset, kp_offset, covis_offset, match_offset;
    std::vector<float> pts_xyz, kp_uv, kp_mappoint_w, Tcw, covis_relpose, Tc_next;
    std::vector<double> intrinsics, Tl_next;
    std::vector<uint8_t> kp_has_mappoint;
    std::vector<int32_t> covis_frame, match_kp_ref, match_kp_covis;
    iba_problem_desc desc() const {
        iba_problem_desc d{};
        d.n_frames = (int32_t)pt_offset.size() - 1;
        d.pt_offset = pt_offset.data(); d.pts_xyz = pts_xyz.data(); d.intrinsics = intrinsics.data();
        d.kp_offset = kp_offset.data(); d.kp_uv = kp_uv.data(); d.kp_has_mappoint = kp_has_mappoint.data(); d.kp_mappoint_w = kp_mappoint_w.data();
        d.Tcw = Tcw.data(); d.covis_offset = covis_offset.data(); d.covis_frame = covis_frame.data(); d.covis_relpose = covis_relpose.data();
        d.match_offset = match_offset.data(); d.match_kp_ref = match_kp_ref.data(); d.match_kp_covis = match_kp_covis.data();
        d.Tc_next = Tc_next.data(); d.Tl_next = Tl_next.data();
        return d;
    }
};

inline void push34(std::vector<float>& v, const cv::Mat& T) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) v.push_back(T.at<float>(r, c)); }

// Mirrors what BAError reads per frame (iba_global.cpp:194-289). KeyFrames sorted by mnId (:507).
inline PackedProblem pack(const std::vector<VecVector3d>& PointClouds, const std::vector<Eigen::Isometry3d>& vTwl,
                          const std::vector<ORB_SLAM2::KeyFrame*>& KeyFrames, int num_best_covis, int min_covis_weight) {
    PackedProblem P;
    const size_t F = KeyFrames.size();
    std::unordered_map<const ORB_SLAM2::KeyFrame*, int> index;
    for (size_t f = 0; f < F; ++f) index[KeyFrames[f]] = (int)f;
    P.pt_offset.push_back(0); P.kp_offset.push_back(0); P.covis_offset.push_back(0); P.match_offset.push_back(0);
    for (size_t f = 0; f < F; ++f) {
        const ORB_SLAM2::KeyFrame* kf = KeyFrames[f];
        for (auto const& p : PointClouds[f]) { P.pts_xyz.push_back((float)p.x()); P.pts_xyz.push_back((float)p.y()); P.pts_xyz.push_back((float)p.z()); }  // exact: read from float32 .bin
        P.pt_offset.push_back(P.pts_xyz.size() / 3);
        for (double v : {(double)kf->fx, (double)kf->fy, (double)kf->cx, (double)kf->cy, (double)kf->mnMaxX, (double)kf->mnMaxY}) P.intrinsics.push_back(v);
        const size_t K = kf->mvKeysUn.size();
        const size_t k0 = P.kp_uv.size() / 2;
        for (auto const& k : kf->mvKeysUn) { P.kp_uv.push_back(k.pt.x); P.kp_uv.push_back(k.pt.y); }
        P.kp_has_mappoint.resize(k0 + K, 0); P.kp_mappoint_w.resize(3 * (k0 + K), 0.f);
        for (auto const& [mpt, kp] : kf->mmapMpt2Kpt) {          // iba_global.cpp:210-213
            const cv::Mat Pw = mpt->GetWorldPos();
            P.kp_has_mappoint[k0 + kp] = 1;
            for (int i = 0; i < 3; ++i) P.kp_mappoint_w[3 * (k0 + kp) + i] = Pw.at<float>(i);
        }
        P.kp_offset.push_back(k0 + K);
        push34(P.Tcw, kf->GetPoseSafe());
        const cv::Mat InvRef = kf->GetPoseInverseSafe();
        auto covis = num_best_covis > 0 ? kf->GetBestCovisibilityKeyFramesSafe(num_best_covis) : kf->GetCovisiblesByWeightSafe(min_covis_weight);
        for (auto* ck : covis) {
            P.covis_frame.push_back(index.at(ck));
            push34(P.covis_relpose, ck->GetPose() * InvRef);      // CV_32F product, unscaled (iba_global.cpp:280)
            for (auto const& [kr, kc] : kf->GetUordMatchedKptIds(ck)) { P.match_kp_ref.push_back(kr); P.match_kp_covis.push_back(kc); }
            P.match_offset.push_back(P.match_kp_ref.size());
        }
        P.covis_offset.push_back(P.covis_frame.size());
        if (f + 1 < F) {                                           // iba_global.cpp:264-269
            push34(P.Tc_next, KeyFrames[f + 1]->GetPose() * InvRef);
            const Eigen::Isometry3d Tl = vTwl[f + 1].inverse() * vTwl[f];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) P.Tl_next.push_back(Tl.matrix()(r, c));
        } else {
            for (int i = 0; i < 12; ++i) { P.Tc_next.push_back(i % 5 == 0 ? 1.f : 0.f); P.Tl_next.push_back(i % 5 == 0 ? 1. : 0.); }
        }
    }
    return P;
}

class Evaluator {   // owns the handle, like BALoss owns kdtree_list (iba_global.cpp:404)
public:
    Evaluator(const PackedProblem& P, const iba_params& prm, int device = 0) {
        iba_problem_desc d = P.desc();
        if (iba_create(&d, &prm, device, 0, d.n_frames, &h_) != IBA_OK) throw std::runtime_error(iba_last_error(nullptr));
    }
    ~Evaluator() { iba_destroy(h_); }
    iba_handle* get() const { return h_; }
private:
    iba_handle* h_ = nullptr;
};

// Same return tuple as BAError (iba_global.cpp:343). `multiprocessing`/`verborse` have no meaning here.
inline std::tuple<double, double, double, int, int> BAError(const double* xvec, Evaluator& ev) {
    iba_cost_out o;
    if (iba_eval_cost(ev.get(), xvec, 1, &o) != IBA_OK) throw std::runtime_error(iba_last_error(ev.get()));
    return {o.f1, o.f2, o.C, o.valid_cnt_3d_2d, o.cnt_3d_2d};
}

// IBAGlobalParams (a class local to iba_global.cpp:26-52 / iba_func.cpp, hence the template) -> iba_params
template <class GlobalParams>
inline iba_params to_iba_params(const GlobalParams& g) {
    iba_params p;
    iba_default_params(&p);
    p.max_pixel_dist = g.max_pixel_dist; p.corr_3d_2d_threshold = g.corr_3d_2d_threshold; p.corr_3d_3d_threshold = g.corr_3d_3d_threshold;
    p.norm_max_pts = g.norm_max_pts; p.norm_min_pts = g.norm_min_pts; p.norm_radius = g.norm_radius; p.norm_reg_threshold = g.norm_reg_threshold;
    p.min_diff_dist = g.min_diff_dist; p.use_plane = g.use_plane ? 1 : 0;
    if (g.err_weight.size() >= 2) { p.err_weight[0] = g.err_weight[0]; p.err_weight[1] = g.err_weight[1]; }
    return p;
}

// The reference's own parameter list (iba_global.cpp:169-173). KdTrees and KFIdMap are accepted and ignored (the 3-D indices live
// on the device; KFIdMap is unused in the reference's body too), multiprocessing / verborse have no meaning here. The problem is
// packed and uploaded on the FIRST call and whenever the caller's containers or parameters change: the cache key is the addresses
// of the two vectors' storage, their sizes and the parameter values (BALoss holds them for the whole run, iba_global.cpp:346-404).
// Thread-compatible like the reference's serial NOMAD loop (:385): one evaluation at a time.
template <class KdTrees, class GlobalParams>
inline std::tuple<double, double, double, int, int> BAError(const double* xvec, const std::vector<VecVector3d>& PointClouds, const KdTrees& /*KdTrees*/,
                                                            const std::vector<Eigen::Isometry3d>& vTwl, const std::unordered_map<int, int>& /*KFIdMap*/,
                                                            const std::vector<ORB_SLAM2::KeyFrame*>& KeyFrames, const GlobalParams& gp /* the reference names it iba_params */,
                                                            const bool& /*multiprocessing*/ = false, const bool& /*verborse*/ = false) {
    struct Cached { const void* pc; const void* kf; size_t n_pc, n_kf; ::iba_params prm; int covis, weight; std::unique_ptr<PackedProblem> packed; std::unique_ptr<Evaluator> ev; };
    static Cached cache{nullptr, nullptr, 0, 0, ::iba_params{}, 0, 0, nullptr, nullptr};
    const ::iba_params prm = to_iba_params(gp);
    auto same_params = [](const ::iba_params& a, const ::iba_params& b) {
        return a.max_pixel_dist == b.max_pixel_dist && a.corr_3d_2d_threshold == b.corr_3d_2d_threshold && a.corr_3d_3d_threshold == b.corr_3d_3d_threshold && a.norm_max_pts == b.norm_max_pts &&
               a.norm_min_pts == b.norm_min_pts && a.norm_radius == b.norm_radius && a.norm_reg_threshold == b.norm_reg_threshold && a.min_diff_dist == b.min_diff_dist &&
               a.use_plane == b.use_plane && a.err_weight[0] == b.err_weight[0] && a.err_weight[1] == b.err_weight[1];
    };
    const bool same_data = cache.ev && cache.pc == (const void*)PointClouds.data() && cache.kf == (const void*)KeyFrames.data() && cache.n_pc == PointClouds.size() &&
                           cache.n_kf == KeyFrames.size() && cache.covis == gp.num_best_covis && cache.weight == gp.min_covis_weight;
    if (!same_data) {
        cache.ev.reset();
        cache.packed.reset(new PackedProblem(pack(PointClouds, vTwl, KeyFrames, gp.num_best_covis, gp.min_covis_weight)));
        cache.ev.reset(new Evaluator(*cache.packed, prm));
        cache.pc = PointClouds.data(); cache.kf = KeyFrames.data(); cache.n_pc = PointClouds.size(); cache.n_kf = KeyFrames.size();
        cache.covis = gp.num_best_covis; cache.weight = gp.min_covis_weight; cache.prm = prm;
    } else if (!same_params(cache.prm, prm)) {
        if (iba_set_params(cache.ev->get(), &prm) != IBA_OK) throw std::runtime_error(iba_last_error(cache.ev->get()));
        cache.prm = prm;
    }
    return BAError(xvec, *cache.ev);
}
// ... and with the 7-vector as an Eigen object (g2o::Vector7: iba_func.cpp:179-183)
template <class Vec7, class KdTrees, class GlobalParams, class = decltype(std::declval<const Vec7&>().data())>
inline std::tuple<double, double, double, int, int> BAError(const Vec7& xvec, const std::vector<VecVector3d>& PointClouds, const KdTrees& trees,
                                                            const std::vector<Eigen::Isometry3d>& vTwl, const std::unordered_map<int, int>& ids,
                                                            const std::vector<ORB_SLAM2::KeyFrame*>& KeyFrames, const GlobalParams& gp,
                                                            const bool& multiprocessing = false, const bool& verborse = false) {
    return BAError(static_cast<const double*>(xvec.data()), PointClouds, trees, vTwl, ids, KeyFrames, gp, multiprocessing, verborse);
}

}  // namespace iba

#if __has_include(<ceres/ceres.h>)
#include <ceres/ceres.h>
namespace iba {
// One cost function standing in for all residual blocks BuildProblem() would add (iba_local.cpp:263-308): the 8 rows
// iba_eval_whitened returns — J^T J = H, J^T r = b, |r|^2 = 2 cost, robust weights folded in on the device — so Ceres'
// Gauss-Newton model and its step acceptance both see the whole problem. No ceres::LossFunction on this block.
class IbaAggregateCostFunction : public ceres::SizedCostFunction<8, 7> {
public:
    explicit IbaAggregateCostFunction(Evaluator& ev) : ev_(ev) {}
    bool Evaluate(double const* const* x, double* residuals, double** jacobians) const override {
        double J[56];   // 8 x 7 row-major, Ceres' own layout
        if (iba_eval_whitened(ev_.get(), x[0], residuals, jacobians && jacobians[0] ? jacobians[0] : J) != IBA_OK) return false;   // association frozen by iba_build_problem
        return true;
    }
private:
    Evaluator& ev_;
};
}  // namespace iba
#endif

#if __has_include(<g2o/core/base_unary_edge.h>)
#include <g2o/core/base_unary_edge.h>
#include "g2o_tools.h"   // VertexSim3 (g2o_tools.h:13-30): additive 7-vector
namespace iba {
// Unary edge on the reference's own VertexSim3 (the shape of IBAPlaneEdge, IBACalib.hpp:74-155): 8-d error = r,
// Jacobian = J of iba_eval_whitened, information = identity: chi2 = |r|^2 = 2 cost.
class IbaAggregateEdge : public g2o::BaseUnaryEdge<8, Eigen::Matrix<double, 8, 1>, VertexSim3> {
public:
    explicit IbaAggregateEdge(Evaluator& ev) : ev_(ev) { setInformation(Eigen::Matrix<double, 8, 8>::Identity()); }
    void computeError() override { eval(); for (int i = 0; i < 8; ++i) _error[i] = r_[i]; }
    void linearizeOplus() override { eval(); _jacobianOplusXi = Eigen::Map<const Eigen::Matrix<double, 8, 7, Eigen::RowMajor>>(J_); }
    bool read(std::istream&) override { return false; }
    bool write(std::ostream&) const override { return false; }
private:
    void eval() {
        const VertexSim3* v = static_cast<const VertexSim3*>(_vertices[0]);
        if (iba_eval_whitened(ev_.get(), v->estimate().data(), r_, J_) != IBA_OK) throw std::runtime_error(iba_last_error(ev_.get()));
    }
    Evaluator& ev_;
    double r_[8], J_[56];
};
}  // namespace iba
#endif
