// A translation unit shaped like the reference's call sites of BAError (syntax check only, see README.md): IBAGlobalParams as the
// class local to iba_global.cpp:26-52, the containers main() holds, the three calls (iba_global.cpp:372, 385; iba_func.cpp:463).
#include <memory>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "iba_reference_adaptors.hpp"

class IBAGlobalParams {
public:
    double max_pixel_dist = 1.5; int kdtree2d_max_leaf_size = 10, kdtree3d_max_leaf_size = 30, num_best_covis = 1, min_covis_weight = 150;
    double corr_3d_2d_threshold = 40., corr_3d_3d_threshold = 5., he_threshold = 0.05; int norm_max_pts = 30, norm_min_pts = 5;
    double norm_radius = 0.6, norm_reg_threshold = 0.04, min_diff_dist = 0.01; std::vector<double> err_weight, lb, ub;
    int PointCloudskip = 1; bool PointCloudOnlyPositiveX = false; int max_bbeval = 200; double valid_rate = 0.5; bool verborse = true, use_plane = true;
};
struct KDTree3D;   // nanoflann::KDTreeVectorOfVectorsAdaptor<...> in the reference: only ever passed through
namespace g2o { typedef Eigen::VecN<double, 7> Vector7; }
using iba::BAError;
using iba::VecVector3d;

double call_sites(const std::vector<VecVector3d>& PointClouds, const std::vector<std::unique_ptr<KDTree3D>>& kdtree_list, const std::vector<Eigen::Isometry3d>& PointCloudPoses,
                  const std::unordered_map<int, int>& KFIdMap, const std::vector<ORB_SLAM2::KeyFrame*>& KeyFrames, const IBAGlobalParams& iba_params, const double* x, const g2o::Vector7& sim3log) {
    double f1, f2, C; int valid_cnt, cnt;
    std::tie(f1, f2, C, valid_cnt, cnt) = BAError(x, PointClouds, kdtree_list, PointCloudPoses, KFIdMap, KeyFrames, iba_params, false, true);   // iba_global.cpp:372
    std::tie(f1, f2, C, valid_cnt, cnt) = BAError(x, PointClouds, kdtree_list, PointCloudPoses, KFIdMap, KeyFrames, iba_params);                // :385 (defaults)
    std::tie(f1, f2, C, valid_cnt, cnt) = BAError(sim3log, PointClouds, kdtree_list, PointCloudPoses, KFIdMap, KeyFrames, iba_params, true, false);   // iba_func.cpp:463
    iba::PackedProblem P = iba::pack(PointClouds, PointCloudPoses, KeyFrames, iba_params.num_best_covis, iba_params.min_covis_weight);
    iba::Evaluator ev(P, iba::to_iba_params(iba_params));
    std::tie(f1, f2, C, valid_cnt, cnt) = BAError(x, ev);
    return f1 + f2 + C + valid_cnt + cnt;
}
