// Host-side, once per problem: the static per-scan 3-D index and the per-frame keypoint grid.
//
// The reference builds one nanoflann KDTree3D per scan in the BALoss constructor
// (iba_global.cpp:361-367) and a fresh KDTree2D over the projected points in EVERY evaluation
// (iba_global.cpp:84). Here:
//  * 3-D: an implicit, perfectly balanced kd-tree. Depth D is fixed per scan, leaf j owns the
//    contiguous point range [j*P>>D, (j+1)*P>>D) of the leaf-ordered SoA arrays, inner node (d,k)
//    splits at rank ((2k+1)*P)>>(d+1). Nothing but (split value, split dim) per inner node is stored,
//    so the whole node array of a 10k-point scan is 4 KB and lives in LDS during traversal.
//  * 2-D: the roles are swapped. Keypoints are static, so THEY are binned once into a uniform grid
//    (cell >= 2*(max_pixel_dist+margin)); projected points look keypoints up. No per-evaluation build.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <vector>

#include "iba_types.hpp"

namespace iba {

struct TreeNode { float split; uint32_t dim; };

inline uint32_t tree_depth_for(uint32_t P) {
    uint32_t D = 0;
    while ((P >> D) > (uint32_t)kLeafTarget && D < (uint32_t)kMaxTreeDepth) ++D;   // leaf size = ceil(P / 2^D) <= kLeafTarget up to the depth cap
    return D;
}

// Reorders a scan into leaf order. in: xyz AoS float (P x 3). out: idx[P] = original index at tree position.
inline void build_tree(const float* xyz, uint32_t P, uint32_t D, std::vector<uint32_t>& idx, std::vector<TreeNode>& nodes) {
    idx.resize(P);
    std::iota(idx.begin(), idx.end(), 0u);
    nodes.assign(((size_t)1 << D) - 1, TreeNode{0.f, 0u});
    if (D == 0 || P == 0) return;
    struct Item { uint32_t d, k; };
    std::vector<Item> stack;
    stack.push_back({0, 0});
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        const uint64_t lo = ((uint64_t)it.k * P) >> it.d, hi = ((uint64_t)(it.k + 1) * P) >> it.d;
        const uint64_t mid = ((uint64_t)(2 * it.k + 1) * P) >> (it.d + 1);
        float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (uint64_t i = lo; i < hi; ++i)
            for (int a = 0; a < 3; ++a) { const float v = xyz[3 * (size_t)idx[i] + a]; mn[a] = std::min(mn[a], v); mx[a] = std::max(mx[a], v); }
        uint32_t dim = 0; float ext = mx[0] - mn[0];
        for (uint32_t a = 1; a < 3; ++a) if (mx[a] - mn[a] > ext) { ext = mx[a] - mn[a]; dim = a; }
        auto cmp = [&](uint32_t a, uint32_t b) {
            const float va = xyz[3 * (size_t)a + dim], vb = xyz[3 * (size_t)b + dim];
            return va < vb || (va == vb && a < b);
        };
        float split = 0.f;
        if (mid > lo && mid < hi) {
            std::nth_element(idx.begin() + lo, idx.begin() + mid, idx.begin() + hi, cmp);
            split = xyz[3 * (size_t)idx[mid] + dim];   // left <= split <= right
        } else if (hi > lo) {
            split = (mid <= lo) ? mn[dim] : mx[dim];
        }
        const size_t heap = (((size_t)1 << it.d) - 1) + it.k;
        nodes[heap] = TreeNode{split, dim};
        if (it.d + 1 < D) { stack.push_back({it.d + 1, 2 * it.k}); stack.push_back({it.d + 1, 2 * it.k + 1}); }
    }
    // deterministic order inside each leaf: ascending original index
    for (uint64_t j = 0; j < ((uint64_t)1 << D); ++j) {
        const uint64_t lo = (j * P) >> D, hi = ((j + 1) * P) >> D;
        std::sort(idx.begin() + lo, idx.begin() + hi);
    }
}

struct KpGrid {
    uint32_t gw = 0, gh = 0;
    std::vector<uint32_t> bitmap;       // 1 bit per 4-px cell: some keypoint within `margin` (L-inf) of the cell
    // coarse CSR (16-px cells = 4x4 fine cells): small enough (~4 KB as u16) to live in LDS during the lookups
    uint32_t gwc = 0, ghc = 0;
    std::vector<uint32_t> coarse_start; // gwc*ghc + 1
    std::vector<float> crec;            // K x 4: u, v, bit pattern of the keypoint id, 0 — sorted by (coarse cell, id)
};

// cell coordinate of a pixel coordinate: one cell of padding on the low side, clamped
__host__ __device__ inline int grid_cell(float u, int n) {
    int c = (int)floorf(u * (1.0f / (float)kGridCell)) + 1;
    return c < 0 ? 0 : (c >= n ? n - 1 : c);
}

inline void build_kp_grid(const float* uv, uint32_t K, double W, double H, double margin, KpGrid& g) {
    g.gw = (uint32_t)std::ceil(W / kGridCell) + 3;
    g.gh = (uint32_t)std::ceil(H / kGridCell) + 3;
    const size_t nc = (size_t)g.gw * g.gh;
    g.gwc = (g.gw + (1u << kCoarseShift) - 1u) >> kCoarseShift; g.ghc = (g.gh + (1u << kCoarseShift) - 1u) >> kCoarseShift;
    const size_t ncc = (size_t)g.gwc * g.ghc;
    g.coarse_start.assign(ncc + 1, 0);
    std::vector<uint32_t> cell_of(K);
    for (uint32_t k = 0; k < K; ++k) {
        const int cxi = grid_cell(uv[2 * k], (int)g.gw) >> kCoarseShift, cyi = grid_cell(uv[2 * k + 1], (int)g.gh) >> kCoarseShift;
        cell_of[k] = (uint32_t)cyi * g.gwc + (uint32_t)cxi;
        g.coarse_start[cell_of[k] + 1]++;
    }
    for (size_t c = 0; c < ncc; ++c) g.coarse_start[c + 1] += g.coarse_start[c];
    g.crec.assign(4 * (size_t)K, 0.f);
    std::vector<uint32_t> fill(g.coarse_start.begin(), g.coarse_start.end() - 1);
    for (uint32_t k = 0; k < K; ++k) {   // ascending k => ids sorted inside each cell
        const uint32_t e = fill[cell_of[k]]++;
        float idf; std::memcpy(&idf, &k, 4);
        g.crec[4 * (size_t)e] = uv[2 * k]; g.crec[4 * (size_t)e + 1] = uv[2 * k + 1]; g.crec[4 * (size_t)e + 2] = idf;
    }
    g.bitmap.assign((nc + 31) / 32, 0u);
    const float m = (float)margin;
    for (uint32_t k = 0; k < K; ++k) {
        const int x0 = grid_cell(uv[2 * k] - m, (int)g.gw), x1 = grid_cell(uv[2 * k] + m, (int)g.gw);
        const int y0 = grid_cell(uv[2 * k + 1] - m, (int)g.gh), y1 = grid_cell(uv[2 * k + 1] + m, (int)g.gh);
        for (int y = y0; y <= y1; ++y)
            for (int x = x0; x <= x1; ++x) { const size_t c = (size_t)y * g.gw + x; g.bitmap[c >> 5] |= 1u << (c & 31); }
    }
}

}  // namespace iba
