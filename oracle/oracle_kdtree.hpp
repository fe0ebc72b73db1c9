// TEST INFRASTRUCTURE ONLY — restatement of the exact kd-tree the reference uses through
// nanoflann v1.5.0 (include/nanoflann.hpp:63) + KDTreeVectorOfVectorsAdaptor.h:55-145.
// Follows: divideTree :1039-1096, middleSplit_ :1209-1259, planeSplit :1270-1312,
// computeMinMax, computeBoundingBox, buildIndex :1544-1566, findNeighbors :1587-1608,
// computeInitialDistances :1314-1338, searchLevel :1735-1811, KNNResultSet :164-237
// (no NANOFLANN_FIRST_MATCH => the earlier-visited element wins exact ties),
// L2_Simple_Adaptor :512-542 (squared L2, accumulated dimension by dimension).
// Pinned against the real nanoflann compiled from /root/reference (oracle/_ref) by
// tests/test_oracle_kdtree.py and the fixtures in tests/golden/.
#pragma once
#include <array>
#include <cstddef>
#include <cstdint>
#include <limits>
#include <vector>

namespace oracle {

struct KNNResultSet {  // nanoflann.hpp:164-237
    uint32_t* indices; double* dists; size_t capacity; size_t count;
    explicit KNNResultSet(size_t cap) : indices(nullptr), dists(nullptr), capacity(cap), count(0) {}
    void init(uint32_t* i, double* d) { indices = i; dists = d; count = 0; if (capacity) dists[capacity - 1] = (std::numeric_limits<double>::max)(); }
    size_t size() const { return count; }
    bool full() const { return count == capacity; }
    bool addPoint(double dist, uint32_t index) {
        size_t i;
        for (i = count; i > 0; --i) {
            if (dists[i - 1] > dist) {
                if (i < capacity) { dists[i] = dists[i - 1]; indices[i] = indices[i - 1]; }
            } else break;
        }
        if (i < capacity) { dists[i] = dist; indices[i] = index; }
        if (count < capacity) count++;
        return true;
    }
    double worstDist() const { return dists[capacity - 1]; }
};

template <int DIM>
class KDTree {
public:
    struct Interval { double low, high; };
    using BoundingBox = std::array<Interval, DIM>;
    struct Node {
        int32_t child1 = -1, child2 = -1;
        size_t left = 0, right = 0;          // leaf
        int divfeat = 0; double divlow = 0, divhigh = 0;  // inner
    };

    // pts: N x DIM doubles (row-major), borrowed (nanoflann keeps a const-ref to the dataset)
    KDTree(const double* pts, size_t n, size_t leaf_max_size) : pts_(pts), size_(n), leaf_max_size_(leaf_max_size) { buildIndex(); }

    size_t size() const { return size_; }

    bool findNeighbors(KNNResultSet& result, const double* vec) const {
        if (size_ == 0) return false;
        const float epsError = 1 + 0.0f;  // SearchParameters().eps == 0
        std::array<double, DIM> dists; dists.fill(0.0);
        double dist = computeInitialDistances(vec, dists);
        searchLevel(result, vec, root_, dist, dists, epsError);
        return result.full();
    }

    // exposed for tests
    const std::vector<uint32_t>& vAcc() const { return vAcc_; }
    size_t num_nodes() const { return nodes_.size(); }

private:
    const double* pts_; size_t size_; size_t leaf_max_size_;
    std::vector<uint32_t> vAcc_; std::vector<Node> nodes_; int32_t root_ = -1; BoundingBox root_bbox_;

    double get(uint32_t idx, int d) const { return pts_[(size_t)idx * DIM + d]; }

    void buildIndex() {
        vAcc_.resize(size_);
        for (size_t i = 0; i < size_; ++i) vAcc_[i] = (uint32_t)i;
        nodes_.clear(); root_ = -1;
        if (size_ == 0) return;
        for (int i = 0; i < DIM; ++i) root_bbox_[i].low = root_bbox_[i].high = get(vAcc_[0], i);
        for (size_t k = 1; k < size_; ++k)
            for (int i = 0; i < DIM; ++i) {
                const double val = get(vAcc_[k], i);
                if (val < root_bbox_[i].low) root_bbox_[i].low = val;
                if (val > root_bbox_[i].high) root_bbox_[i].high = val;
            }
        nodes_.reserve(2 * size_ / (leaf_max_size_ ? leaf_max_size_ : 1) + 16);
        root_ = divideTree(0, size_, root_bbox_);
    }

    void computeMinMax(size_t ind, size_t count, int element, double& min_elem, double& max_elem) const {
        min_elem = get(vAcc_[ind], element); max_elem = min_elem;
        for (size_t i = 1; i < count; ++i) {
            double val = get(vAcc_[ind + i], element);
            if (val < min_elem) min_elem = val;
            if (val > max_elem) max_elem = val;
        }
    }

    int32_t divideTree(size_t left, size_t right, BoundingBox& bbox) {
        const int32_t id = (int32_t)nodes_.size();
        nodes_.emplace_back();
        if ((right - left) <= leaf_max_size_) {
            nodes_[id].child1 = nodes_[id].child2 = -1; nodes_[id].left = left; nodes_[id].right = right;
            for (int i = 0; i < DIM; ++i) { bbox[i].low = get(vAcc_[left], i); bbox[i].high = get(vAcc_[left], i); }
            for (size_t k = left + 1; k < right; ++k)
                for (int i = 0; i < DIM; ++i) {
                    const double val = get(vAcc_[k], i);
                    if (bbox[i].low > val) bbox[i].low = val;
                    if (bbox[i].high < val) bbox[i].high = val;
                }
        } else {
            size_t idx; int cutfeat; double cutval;
            middleSplit_(left, right - left, idx, cutfeat, cutval, bbox);
            nodes_[id].divfeat = cutfeat;
            BoundingBox left_bbox(bbox); left_bbox[cutfeat].high = cutval;
            int32_t c1 = divideTree(left, left + idx, left_bbox);
            BoundingBox right_bbox(bbox); right_bbox[cutfeat].low = cutval;
            int32_t c2 = divideTree(left + idx, right, right_bbox);
            nodes_[id].child1 = c1; nodes_[id].child2 = c2;
            nodes_[id].divlow = left_bbox[cutfeat].high; nodes_[id].divhigh = right_bbox[cutfeat].low;
            for (int i = 0; i < DIM; ++i) {
                bbox[i].low = std::min(left_bbox[i].low, right_bbox[i].low);
                bbox[i].high = std::max(left_bbox[i].high, right_bbox[i].high);
            }
        }
        return id;
    }

    void middleSplit_(size_t ind, size_t count, size_t& index, int& cutfeat, double& cutval, const BoundingBox& bbox) {
        const double EPS = 0.00001;
        double max_span = bbox[0].high - bbox[0].low;
        for (int i = 1; i < DIM; ++i) { double span = bbox[i].high - bbox[i].low; if (span > max_span) max_span = span; }
        double max_spread = -1; cutfeat = 0;
        for (int i = 0; i < DIM; ++i) {
            double span = bbox[i].high - bbox[i].low;
            if (span > (1 - EPS) * max_span) {
                double min_elem, max_elem; computeMinMax(ind, count, i, min_elem, max_elem);
                double spread = max_elem - min_elem;
                if (spread > max_spread) { cutfeat = i; max_spread = spread; }
            }
        }
        double split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
        double min_elem, max_elem; computeMinMax(ind, count, cutfeat, min_elem, max_elem);
        if (split_val < min_elem) cutval = min_elem;
        else if (split_val > max_elem) cutval = max_elem;
        else cutval = split_val;
        size_t lim1, lim2; planeSplit(ind, count, cutfeat, cutval, lim1, lim2);
        if (lim1 > count / 2) index = lim1;
        else if (lim2 < count / 2) index = lim2;
        else index = count / 2;
    }

    void planeSplit(size_t ind, size_t count, int cutfeat, const double& cutval, size_t& lim1, size_t& lim2) {
        size_t left = 0, right = count - 1;
        for (;;) {
            while (left <= right && get(vAcc_[ind + left], cutfeat) < cutval) ++left;
            while (right && left <= right && get(vAcc_[ind + right], cutfeat) >= cutval) --right;
            if (left > right || !right) break;
            std::swap(vAcc_[ind + left], vAcc_[ind + right]); ++left; --right;
        }
        lim1 = left; right = count - 1;
        for (;;) {
            while (left <= right && get(vAcc_[ind + left], cutfeat) <= cutval) ++left;
            while (right && left <= right && get(vAcc_[ind + right], cutfeat) > cutval) --right;
            if (left > right || !right) break;
            std::swap(vAcc_[ind + left], vAcc_[ind + right]); ++left; --right;
        }
        lim2 = left;
    }

    double computeInitialDistances(const double* vec, std::array<double, DIM>& dists) const {
        double dist = 0;
        for (int i = 0; i < DIM; ++i) {
            if (vec[i] < root_bbox_[i].low) { dists[i] = (vec[i] - root_bbox_[i].low) * (vec[i] - root_bbox_[i].low); dist += dists[i]; }
            if (vec[i] > root_bbox_[i].high) { dists[i] = (vec[i] - root_bbox_[i].high) * (vec[i] - root_bbox_[i].high); dist += dists[i]; }
        }
        return dist;
    }

    double evalMetric(const double* a, uint32_t b_idx) const {  // L2_Simple_Adaptor::evalMetric
        double result = 0;
        for (int i = 0; i < DIM; ++i) { const double diff = a[i] - get(b_idx, i); result += diff * diff; }
        return result;
    }

    bool searchLevel(KNNResultSet& result_set, const double* vec, int32_t node_id, double mindist, std::array<double, DIM>& dists, const float epsError) const {
        const Node& node = nodes_[node_id];
        if (node.child1 < 0 && node.child2 < 0) {
            double worst_dist = result_set.worstDist();
            for (size_t i = node.left; i < node.right; ++i) {
                const uint32_t accessor = vAcc_[i];
                double dist = evalMetric(vec, accessor);
                if (dist < worst_dist) { if (!result_set.addPoint(dist, vAcc_[i])) return false; }
            }
            return true;
        }
        int idx = node.divfeat; double val = vec[idx];
        double diff1 = val - node.divlow, diff2 = val - node.divhigh;
        int32_t bestChild, otherChild; double cut_dist;
        if ((diff1 + diff2) < 0) { bestChild = node.child1; otherChild = node.child2; cut_dist = (val - node.divhigh) * (val - node.divhigh); }
        else { bestChild = node.child2; otherChild = node.child1; cut_dist = (val - node.divlow) * (val - node.divlow); }
        if (!searchLevel(result_set, vec, bestChild, mindist, dists, epsError)) return false;
        double dst = dists[idx];
        mindist = mindist + cut_dist - dst;
        dists[idx] = cut_dist;
        if (mindist * epsError <= result_set.worstDist()) {
            if (!searchLevel(result_set, vec, otherChild, mindist, dists, epsError)) return false;
        }
        dists[idx] = dst;
        return true;
    }
};

}  // namespace oracle
