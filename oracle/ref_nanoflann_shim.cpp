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
