// TEST INFRASTRUCTURE ONLY. Thin C shim around the REFERENCE's own vendored nanoflann v1.5.0,
// compiled from the headers where they lie under /root/reference/include (never copied into this
// repo). Output goes to oracle/_ref/libref_nanoflann.so (git-ignored, travels with gpurun).
// Used to (1) validate oracle_kdtree.hpp and (2) generate tests/golden/knn_*.npz.
// Mirrors the typedefs at include/pointcloud.h:19-21:
//   KDTreeVectorOfVectorsAdaptor<VecVector{2,3}d, double, {2,3}, metric_L2_Simple, uint32_t>
#include <array>
#include <cstdint>
#include <limits>
#include <memory>
#include <vector>

#include "nanoflann.hpp"
#include "KDTreeVectorOfVectorsAdaptor.h"

namespace {
template <int DIM>
int knn_impl(const double* pts, uint64_t n, int leaf, const double* queries, uint64_t nq, int k, uint32_t* out_idx,
             double* out_d2, int32_t* out_cnt) {
    using Vec = std::vector<std::array<double, DIM>>;
    using Tree = nanoflann::KDTreeVectorOfVectorsAdaptor<Vec, double, DIM, nanoflann::metric_L2_Simple, uint32_t>;
    Vec data(n);
    for (uint64_t i = 0; i < n; ++i)
        for (int d = 0; d < DIM; ++d) data[i][d] = pts[i * DIM + d];
    if (n == 0) {
        for (uint64_t q = 0; q < nq; ++q) out_cnt[q] = 0;
        return 0;
    }
    Tree tree(DIM, data, leaf);
    for (uint64_t q = 0; q < nq; ++q) {
        // same call pattern as iba_global.cpp:87-92 / :124-129
        std::vector<uint32_t> indices(k);
        std::vector<double> sq_dist(k, std::numeric_limits<double>::max());
        nanoflann::KNNResultSet<double, uint32_t> resultSet(k);
        resultSet.init(indices.data(), sq_dist.data());
        tree.index->findNeighbors(resultSet, queries + q * DIM, nanoflann::SearchParameters());
        const int cnt = (int)resultSet.size();
        out_cnt[q] = cnt;
        for (int j = 0; j < k; ++j) {
            out_idx[q * k + j] = j < cnt ? indices[j] : 0xFFFFFFFFu;
            out_d2[q * k + j] = j < cnt ? sq_dist[j] : -1.0;
        }
    }
    return 0;
}
}  // namespace

extern "C" int ref_nanoflann_knn(int dim, const double* pts, uint64_t n, int leaf, const double* queries, uint64_t nq,
                                 int k, uint32_t* out_idx, double* out_d2, int32_t* out_cnt) {
    if (dim == 2) return knn_impl<2>(pts, n, leaf, queries, nq, k, out_idx, out_d2, out_cnt);
    if (dim == 3) return knn_impl<3>(pts, n, leaf, queries, nq, k, out_idx, out_d2, out_cnt);
    return 1;
}

extern "C" int ref_nanoflann_version(void) { return NANOFLANN_VERSION; }

// GeoCalib.h:18-33 computeCorrespondence on the reference's nanoflann: the same index type (KDTreeSingleIndexAdaptor, L2_Simple_Adaptor<double, .>,
// 3, std::uint32_t), max_leaf 15 (:23), knnSearch with num_res = 1 (:25-28) and the test  sq_dist[0] <= maxDistance  as written (:29). The reference
// hands nanoflann a std::vector<Eigen::Vector3d>, which is no nanoflann dataset (no kdtree_get_pt: the header compiles nowhere); the
// dataset adaptor below is the minimal one nanoflann's own documentation prescribes — everything that decides a result is the reference's code.
namespace {
struct GeoCloud {
    const double* p; size_t n;
    inline size_t kdtree_get_point_count() const { return n; }
    inline double kdtree_get_pt(const size_t idx, const size_t dim) const { return p[3 * idx + dim]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};
}  // namespace
extern "C" int ref_geo_correspondences(const double* src, uint64_t n_src, const double* tgt, uint64_t n_tgt, double maxDistance,
                                       uint32_t* out_src, uint32_t* out_tgt, int64_t* n_out) {
    typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<double, GeoCloud>, GeoCloud, 3, std::uint32_t> KDTreeType;
    *n_out = 0;
    if (n_tgt == 0) return 0;
    GeoCloud cloud{tgt, (size_t)n_tgt};
    std::unique_ptr<KDTreeType> kdtree(new KDTreeType(3, cloud, {15}));   // max_leaf_size = 15
    int64_t n = 0;
    for (std::uint32_t i = 0; i < (std::uint32_t)n_src; ++i) {
        std::uint32_t num_res = 1;
        std::vector<std::uint32_t> query_index(num_res);
        std::vector<double> sq_dist(num_res, 1000);
        num_res = kdtree->knnSearch(src + 3 * (size_t)i, num_res, query_index.data(), sq_dist.data());
        if (num_res > 0 && sq_dist[0] <= maxDistance) { out_src[n] = i; out_tgt[n] = query_index[0]; ++n; }
    }
    *n_out = n;
    return 0;
}
