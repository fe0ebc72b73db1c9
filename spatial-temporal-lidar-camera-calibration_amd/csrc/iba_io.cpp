// On-disk formats of the reference pipeline -> flat iba_problem_desc. Host only, no GPU,
// no OpenCV / ORB-SLAM2 / yaml-cpp: the few constructs cv::FileStorage emits are parsed directly.
//
// What is restated here (reference file:line):
//  * readPointCloud, .bin branch                         io_tools.h:142-196
//  * listdir (regular files, sorted by name)             kitti_tools.h:48-62
//  * ReadPoseList                                        kitti_tools.h:66-87
//  * readSim3 / writeSim3                                kitti_tools.h:96-158
//  * FrameId.yml                                         System.cc:597-609
//  * KeyFrames/NNNNNN.yml keys                           KeyFrame.cc:31-80 (reader), 209-252 (writer)
//  * Map.yml / MapPoint nodes                            Map.cc:162-170, 213-231; MapPoint.cc:435-476
//  * restore logic (ids -> objects, sort by mnId)        System.cc:612-694, KeyFrame.cc:104-131, 133-168
//  * what main() derives from them                       iba_global.cpp:464-505, iba_local.cpp:379-406
//  * what BAError / BuildProblem read off the KeyFrames  iba_global.cpp:205-213, 253-289; iba_local.cpp:165-190
//  * KeyFrame::SetPose (float Twc), GetMatchedKptIds     KeyFrame.cc:271-285, 527-538
//  * covisible selection                                 KeyFrame.cc:417-439
//
// cv::Mat CV_32F products (relative poses, iba_global.cpp:267, 280) follow OpenCV's small-matrix gemm path: float
// accumulation, k ascending, no fused multiply-add (this file is built with -ffp-contract=off). OpenCV is not in the
// build image, so bitwise parity of those products with a real OpenCV build is UNPINNED; oracle/formats.py restates
// the same arithmetic independently and the tests compare the two.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../include/iba_mi355x.h"

namespace {

thread_local std::string g_io_err;
iba_status io_fail(iba_status s, const std::string& m) { g_io_err = m; return s; }

bool read_file(const std::string& path, std::string& out) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(n > 0 ? (size_t)n : 0);
    const size_t got = n > 0 ? std::fread(&out[0], 1, (size_t)n, f) : 0;
    std::fclose(f);
    return got == out.size();
}

// ---- cv::FileStorage YAML subset: block mappings by indentation, flow sequences (possibly wrapped / nested),
// block sequences of flow sequences, !!opencv-matrix nodes ----
struct YLine { size_t b, e; int indent; };
class CvYaml {
   public:
    std::string text;
    std::vector<YLine> lines;
    bool load(const std::string& path) {
        if (!read_file(path, text)) return false;
        size_t i = 0;
        const size_t n = text.size();
        while (i < n) {
            size_t j = i;
            while (j < n && text[j] != '\n') ++j;
            size_t e = j;
            if (e > i && text[e - 1] == '\r') --e;
            size_t b = i;
            int ind = 0;
            while (b < e && text[b] == ' ') { ++b; ++ind; }
            const bool skip = (b == e) || text[b] == '#' || text[b] == '%' || (e - b >= 3 && text.compare(b, 3, "---") == 0 && ind == 0) ||
                              (e - b >= 3 && text.compare(b, 3, "...") == 0 && ind == 0);
            if (!skip) lines.push_back(YLine{b, e, ind});
            i = j + 1;
        }
        return true;
    }
    int size() const { return (int)lines.size(); }
    // first line after `line` whose indent is <= that of `line` (end of its block)
    int block_end(int line) const {
        int k = line + 1;
        while (k < size() && lines[k].indent > lines[line].indent) ++k;
        return k;
    }
    bool key_is(int line, const char* key) const {
        const YLine& L = lines[line];
        const size_t kl = std::strlen(key);
        return L.e - L.b > kl && text.compare(L.b, kl, key) == 0 && text[L.b + kl] == ':';
    }
    int find(int lo, int hi, int indent, const char* key) const {
        for (int k = lo; k < hi; ++k)
            if (lines[k].indent == indent && key_is(k, key)) return k;
        return -1;
    }
    // every numeric token of the value of the entry at `line`: the rest of the line after "key:" plus all following
    // lines that are indented deeper (wrapped flow sequences, block sequences of flow sequences)
    bool numbers(int line, std::vector<double>& out) const {
        const YLine& L = lines[line];
        size_t p = L.b;
        while (p < L.e && text[p] != ':') ++p;
        if (p == L.e) return false;
        ++p;
        if (!scan(p, L.e, out)) return false;
        const int end = block_end(line);
        for (int k = line + 1; k < end; ++k)
            if (!scan(lines[k].b, lines[k].e, out)) return false;
        return true;
    }
    bool scalar(int line, double& v) const {
        std::vector<double> t;
        if (!numbers(line, t) || t.size() != 1) return false;
        v = t[0];
        return true;
    }
    // !!opencv-matrix node at `line`
    bool matrix(int line, int& rows, int& cols, std::vector<double>& data) const {
        const int end = block_end(line);
        if (end == line + 1) return false;
        const int ind = lines[line + 1].indent;
        const int r = find(line + 1, end, ind, "rows"), c = find(line + 1, end, ind, "cols"), d = find(line + 1, end, ind, "data");
        double rv, cv;
        if (r < 0 || c < 0 || d < 0 || !scalar(r, rv) || !scalar(c, cv)) return false;
        rows = (int)rv; cols = (int)cv;
        data.clear();
        return numbers(d, data) && (long long)data.size() == (long long)rows * cols;
    }

   private:
    bool scan(size_t p, size_t e, std::vector<double>& out) const {
        while (p < e) {
            const char ch = text[p];
            if (ch == ' ' || ch == '\t' || ch == '[' || ch == ']' || ch == ',' || ch == '{' || ch == '}') { ++p; continue; }
            if (ch == '-' && (p + 1 == e || text[p + 1] == ' ')) { ++p; continue; }   // block sequence marker
            if (ch == '#') break;
            size_t q = p;
            while (q < e && text[q] != ' ' && text[q] != ',' && text[q] != ']' && text[q] != '[' && text[q] != '}') ++q;
            const std::string tok = text.substr(p, q - p);
            double v;
            if (tok == ".Inf" || tok == ".inf" || tok == "+.Inf") v = std::numeric_limits<double>::infinity();
            else if (tok == "-.Inf" || tok == "-.inf") v = -std::numeric_limits<double>::infinity();
            else if (tok == ".Nan" || tok == ".NaN" || tok == ".nan") v = std::numeric_limits<double>::quiet_NaN();
            else {
                char* endp = nullptr;
                v = std::strtod(tok.c_str(), &endp);
                if (endp == tok.c_str() || *endp != '\0') return false;   // not a number (string / tag): caller asked for the wrong node
            }
            out.push_back(v);
            p = q;
        }
        return true;
    }
};

bool list_regular_files(const std::string& dir, std::vector<std::string>& names) {   // kitti_tools.h:48-62
    names.clear();
    DIR* d = opendir(dir.c_str());
    if (!d) return false;
    while (dirent* ent = readdir(d)) {
        bool reg = ent->d_type == DT_REG;
        if (ent->d_type == DT_UNKNOWN) {
            struct stat st;
            reg = stat((dir + ent->d_name).c_str(), &st) == 0 && S_ISREG(st.st_mode);
        }
        if (reg) names.push_back(ent->d_name);
    }
    closedir(d);
    std::sort(names.begin(), names.end());
    return true;
}

std::string with_slash(const char* p) {   // checkpath, kitti_tools.h:19-22
    std::string s(p ? p : "");
    if (!s.empty() && s.back() != '/') s += '/';
    return s;
}

bool suffix_is(const std::string& name, const char* suf) {
    const size_t dot = name.find_last_of('.');
    return dot != std::string::npos && name.compare(dot + 1, std::string::npos, suf) == 0;
}

iba_status read_kitti_bin_impl(const std::string& file, int skip, bool only_positive_x, std::vector<float>& xyz) {
    std::string raw;
    if (!read_file(file, raw)) return io_fail(IBA_ERR_IO, "file " + file + " cannot open");   // io_tools.h:158-160
    if (skip < 1) return io_fail(IBA_ERR_INVALID_ARG, "pointcloud skip must be >= 1");
    const size_t n = raw.size() / 16;   // XYZI float32 (io_tools.h:166)
    // `for (i = 0; i <= num_points - skip; i += skip)` on size_t: with fewer points than `skip` the reference wraps
    // around and reads past the end; there is nothing sensible to reproduce
    if (n < (size_t)skip) return io_fail(IBA_ERR_UNSUPPORTED, "point cloud " + file + " has fewer points than `skip`");
    const size_t count = (n - (size_t)skip) / (size_t)skip + 1;   // iterations; each one consumes the NEXT record
    xyz.clear();
    xyz.reserve(3 * count);
    for (size_t i = 0; i < count; ++i) {
        float rec[4];
        std::memcpy(rec, raw.data() + 16 * i, 16);
        if (only_positive_x && rec[0] <= 0) continue;   // io_tools.h:174-180
        xyz.push_back(rec[0]); xyz.push_back(rec[1]); xyz.push_back(rec[2]);
    }
    return IBA_OK;
}

iba_status read_numbers(const std::string& file, std::vector<double>& v) {
    std::string raw;
    if (!read_file(file, raw)) return io_fail(IBA_ERR_IO, "Cannot open file: " + file);
    const char* p = raw.c_str();
    const char* end = p + raw.size();
    v.clear();
    while (p < end) {
        while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) ++p;
        if (p >= end) break;
        char* q = nullptr;
        const double x = std::strtod(p, &q);
        if (q == p) break;   // operator>> stops at the first non-number, so do we
        v.push_back(x);
        p = q;
    }
    return IBA_OK;
}

// ---- float matrices as cv::Mat CV_32F would compute them ----
inline void mul44f(const float* A, const float* B, float* C) {   // C = A * B, 4x4 row-major
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const float t = ((A[i * 4 + 0] * B[0 * 4 + j] + A[i * 4 + 1] * B[1 * 4 + j]) + A[i * 4 + 2] * B[2 * 4 + j]) + A[i * 4 + 3] * B[3 * 4 + j];
            C[i * 4 + j] = t;
        }
}
inline void set_pose_inverse(const float* Tcw, float* Twc) {   // KeyFrame::SetPose, KeyFrame.cc:271-282
    for (int i = 0; i < 16; ++i) Twc[i] = (i % 5 == 0) ? 1.f : 0.f;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Twc[r * 4 + c] = Tcw[c * 4 + r];   // Rwc = Rcw.t()
    for (int r = 0; r < 3; ++r) {                                     // Ow = -Rwc * tcw
        const float t = (Twc[r * 4 + 0] * Tcw[0 * 4 + 3] + Twc[r * 4 + 1] * Tcw[1 * 4 + 3]) + Twc[r * 4 + 2] * Tcw[2 * 4 + 3];
        Twc[r * 4 + 3] = -t;
    }
}

struct KeyFrameInfo {
    int mnId = -1, mnFrameId = -1;
    float fx = 0, fy = 0, cx = 0, cy = 0;
    int maxX = 0, maxY = 0;
    std::vector<float> uv;                    // 2 per keypoint (mvKeysUn[i].pt)
    std::vector<int> octave;                  // mvKeysUn[i].octave
    std::vector<float> inv_level_sigma2;      // mvInvLevelSigma2 (optional in the file: only the BA edges need it)
    float Tcw[16], Twc[16];
    std::vector<int> conn_ids, weights;       // mvpOrderedConnectedKeyFramesId, mvOrderedWeights
    std::unordered_map<int, int> mpt2kpt_id;  // KeyFrameConstInfo::mmapMpt2KptId (first insertion wins, KeyFrame.cc:76-79)
    std::vector<int> mpt_ids;                 // mvpMapPointsId in file order
    std::map<int, int> mpt2kpt;               // restored KeyFrame::mmapMpt2Kpt keyed by MapPoint id (KeyFrame.cc:120-129)
    std::vector<int> covis;                   // indices (into the sorted keyframe list) of the covisible KFs BAError uses
    std::string file, err;
};

bool load_keyframe(const std::string& path, KeyFrameInfo& kf) {
    kf.file = path;
    CvYaml y;
    if (!y.load(path)) { kf.err = "cannot open " + path; return false; }
    auto need = [&](const char* key) {
        const int k = y.find(0, y.size(), 0, key);
        if (k < 0) kf.err = path + ": key '" + key + "' missing";
        return k;
    };
    auto get_scalar = [&](const char* key, double& v) {
        const int k = need(key);
        if (k < 0) return false;
        if (!y.scalar(k, v)) { kf.err = path + ": key '" + key + "' is not a scalar"; return false; }
        return true;
    };
    auto get_ints = [&](const char* key, std::vector<int>& out) {
        const int k = need(key);
        if (k < 0) return false;
        std::vector<double> t;
        if (!y.numbers(k, t)) { kf.err = path + ": key '" + key + "' is not a numeric sequence"; return false; }
        out.resize(t.size());
        for (size_t i = 0; i < t.size(); ++i) out[i] = (int)t[i];
        return true;
    };
    double v;
    if (!get_scalar("mnId", v)) return false; kf.mnId = (int)v;
    if (!get_scalar("mnFrameId", v)) return false; kf.mnFrameId = (int)v;
    if (!get_scalar("fx", v)) return false; kf.fx = (float)v;
    if (!get_scalar("fy", v)) return false; kf.fy = (float)v;
    if (!get_scalar("cx", v)) return false; kf.cx = (float)v;
    if (!get_scalar("cy", v)) return false; kf.cy = (float)v;
    if (!get_scalar("mnMaxX", v)) return false; kf.maxX = (int)v;
    if (!get_scalar("mnMaxY", v)) return false; kf.maxY = (int)v;
    {   // vector<cv::KeyPoint>: 7 numbers per keypoint (pt.x, pt.y, size, angle, response, octave, class_id), either as
        // one flat flow sequence (OpenCV 3) or as a block sequence of 7-element flow sequences (OpenCV 4)
        const int k = need("mvKeysUn");
        if (k < 0) return false;
        std::vector<double> t;
        if (!y.numbers(k, t) || t.size() % 7 != 0) { kf.err = path + ": mvKeysUn is not a sequence of 7-number keypoints"; return false; }
        const size_t K = t.size() / 7;
        kf.uv.resize(2 * K);
        kf.octave.resize(K);
        for (size_t i = 0; i < K; ++i) { kf.uv[2 * i] = (float)t[7 * i]; kf.uv[2 * i + 1] = (float)t[7 * i + 1]; kf.octave[i] = (int)t[7 * i + 5]; }
        const int ks = y.find(0, y.size(), 0, "mvInvLevelSigma2");
        std::vector<double> sg;
        if (ks >= 0 && y.numbers(ks, sg)) { kf.inv_level_sigma2.resize(sg.size()); for (size_t i = 0; i < sg.size(); ++i) kf.inv_level_sigma2[i] = (float)sg[i]; }
    }
    {
        const int k = need("Pose");
        if (k < 0) return false;
        int rows, cols;
        std::vector<double> d;
        if (!y.matrix(k, rows, cols, d) || rows != 4 || cols != 4) { kf.err = path + ": Pose is not a 4x4 opencv-matrix"; return false; }
        for (int i = 0; i < 16; ++i) kf.Tcw[i] = (float)d[i];
        set_pose_inverse(kf.Tcw, kf.Twc);
    }
    std::vector<int> kpt_ids;
    if (!get_ints("mvpMapPointsId", kf.mpt_ids) || !get_ints("mvpCorrKeyPointsId", kpt_ids)) return false;
    if (kf.mpt_ids.size() != kpt_ids.size()) { kf.err = path + ": mvpMapPointsId and mvpCorrKeyPointsId differ in length"; return false; }   // KeyFrame.cc:76
    for (size_t i = 0; i < kf.mpt_ids.size(); ++i) kf.mpt2kpt_id.insert(std::make_pair(kf.mpt_ids[i], kpt_ids[i]));
    if (!get_ints("mvpOrderedConnectedKeyFramesId", kf.conn_ids) || !get_ints("mvOrderedWeights", kf.weights)) return false;
    return true;
}

}  // namespace

struct iba_dataset {
    iba_problem_desc desc;
    std::vector<uint64_t> pt_offset, kp_offset, covis_offset, match_offset;
    std::vector<float> pts_xyz, kp_uv, kp_mappoint_w, Tcw, covis_relpose, Tc_next;
    std::vector<double> intrinsics, Tl_next;
    std::vector<uint8_t> kp_has_mappoint;
    std::vector<int32_t> covis_frame, match_kp_ref, match_kp_covis, mn_id, mn_frame_id;
};

extern "C" {

const char* iba_io_last_error(void) { return g_io_err.c_str(); }
void iba_io_free(void* p) { std::free(p); }

// the numbers of one top-level entry of a cv::FileStorage YAML file (the dialect of KeyFrames/NNNNNN.yml, Map.yml and of ORB-SLAM2's
// settings files such as config/orb_ori/KITTI00-02.yaml): a scalar, a flow sequence, or the data of an !!opencv-matrix node
iba_status iba_read_cv_yaml_numbers(const char* file, const char* key, double* out, int32_t cap, int32_t* n_out) {
    if (!file || !key || !n_out) return io_fail(IBA_ERR_INVALID_ARG, "null argument");
    CvYaml y;
    if (!y.load(file)) return io_fail(IBA_ERR_IO, std::string("Cannot open file: ") + file);
    const int line = y.find(0, y.size(), 0, key);
    if (line < 0) return io_fail(IBA_ERR_IO, std::string(file) + ": no top-level entry '" + key + "'");
    std::vector<double> v;
    int rows = 0, cols = 0;
    if (!y.numbers(line, v) && !y.matrix(line, rows, cols, v)) return io_fail(IBA_ERR_IO, std::string(file) + ": entry '" + key + "' is not numeric");
    if (rows > 0) { std::vector<double> m; if (y.matrix(line, rows, cols, m)) v = m; }
    *n_out = (int32_t)v.size();
    for (int32_t i = 0; i < (int32_t)v.size() && i < cap && out; ++i) out[i] = v[(size_t)i];
    return IBA_OK;
}

iba_status iba_read_kitti_bin(const char* file, int32_t skip, int32_t only_positive_x, float** xyz, int64_t* n_points) {
    if (!file || !xyz || !n_points) return io_fail(IBA_ERR_INVALID_ARG, "null argument");
    std::vector<float> v;
    const iba_status s = read_kitti_bin_impl(file, skip, only_positive_x != 0, v);
    if (s != IBA_OK) return s;
    *n_points = (int64_t)(v.size() / 3);
    *xyz = (float*)std::malloc(std::max<size_t>(v.size(), 1) * sizeof(float));
    if (!*xyz) return io_fail(IBA_ERR_IO, "out of memory");
    if (!v.empty()) std::memcpy(*xyz, v.data(), v.size() * sizeof(float));   // (memcpy from a null pointer is undefined even for 0 bytes: found by the UBSan build)
    return IBA_OK;
}

iba_status iba_read_pose_list(const char* file, double** poses12, int64_t* n_poses) {
    if (!file || !poses12 || !n_poses) return io_fail(IBA_ERR_INVALID_ARG, "null argument");
    std::vector<double> v;
    const iba_status s = read_numbers(file, v);
    if (s != IBA_OK) return s;
    const size_t n = v.size() / 12;
    *n_poses = (int64_t)n;
    *poses12 = (double*)std::malloc(std::max<size_t>(12 * n, 1) * sizeof(double));
    if (!*poses12) return io_fail(IBA_ERR_IO, "out of memory");
    if (n) std::memcpy(*poses12, v.data(), 12 * n * sizeof(double));
    return IBA_OK;
}

iba_status iba_read_sim3(const char* file, double rigid12[12], double* scale) {   // kitti_tools.h:146-158
    if (!file || !rigid12 || !scale) return io_fail(IBA_ERR_INVALID_ARG, "null argument");
    std::vector<double> v;
    const iba_status s = read_numbers(file, v);
    if (s != IBA_OK) return s;
    // the reference starts from identity / scale 1 and overwrites whatever the stream yields
    const double ident[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int i = 0; i < 12; ++i) rigid12[i] = (size_t)i < v.size() ? v[i] : ident[i];
    *scale = v.size() > 12 ? v[12] : 1.0;
    return IBA_OK;
}

iba_status iba_write_sim3(const char* file, const double rigid12[12], double scale) {   // kitti_tools.h:96-107
    if (!file || !rigid12) return io_fail(IBA_ERR_INVALID_ARG, "null argument");
    FILE* f = std::fopen(file, "w");
    if (!f) return io_fail(IBA_ERR_IO, std::string("cannot write ") + file);
    for (int i = 0; i < 12; ++i) std::fprintf(f, "%.17g ", rigid12[i]);   // precision(max_digits10), default float format
    std::fprintf(f, "%.17g", scale);
    std::fclose(f);
    return IBA_OK;
}

const iba_problem_desc* iba_dataset_desc(const iba_dataset* d) { return d ? &d->desc : nullptr; }
void iba_dataset_free(iba_dataset* d) { delete d; }
iba_status iba_dataset_frame_ids(const iba_dataset* d, int32_t frame, int32_t* mn_id, int32_t* mn_frame_id) {
    if (!d || frame < 0 || frame >= d->desc.n_frames) return io_fail(IBA_ERR_INVALID_ARG, "frame out of range");
    if (mn_id) *mn_id = d->mn_id[frame];
    if (mn_frame_id) *mn_frame_id = d->mn_frame_id[frame];
    return IBA_OK;
}

iba_status iba_dataset_load(const iba_dataset_paths* P, iba_dataset** out) {
    if (!P || !out || !P->frame_id_file || !P->lidar_pose_file || !P->pointcloud_dir || !P->keyframe_dir || !P->map_file)
        return io_fail(IBA_ERR_INVALID_ARG, "null path");
    *out = nullptr;
    // ---- FrameId.yml (iba_global.cpp:464-465) ----
    std::vector<int> vKFId, vKFFrameId;
    {
        CvYaml y;
        if (!y.load(P->frame_id_file)) return io_fail(IBA_ERR_IO, std::string("cannot open ") + P->frame_id_file);
        const int a = y.find(0, y.size(), 0, "mnId"), b = y.find(0, y.size(), 0, "mnFrameId");
        std::vector<double> ta, tb;
        if (b < 0 || !y.numbers(b, tb)) return io_fail(IBA_ERR_IO, std::string(P->frame_id_file) + ": 'mnFrameId' missing");
        if (a >= 0) y.numbers(a, ta);
        for (double v : tb) vKFFrameId.push_back((int)v);
        for (double v : ta) vKFId.push_back((int)v);
    }
    const int F = (int)vKFFrameId.size();
    if (F == 0) return io_fail(IBA_ERR_IO, "FrameId.yml lists no keyframes");
    // ---- LiDAR poses of the keyframes (iba_global.cpp:467-479) ----
    std::vector<double> raw;
    {
        const iba_status s = read_numbers(P->lidar_pose_file, raw);
        if (s != IBA_OK) return s;
    }
    const int n_raw = (int)(raw.size() / 12);
    std::vector<double> Twl((size_t)F * 12);
    {
        auto pose = [&](int id) { return raw.data() + 12 * (size_t)id; };
        for (int f = 0; f < F; ++f)
            if (vKFFrameId[f] < 0 || vKFFrameId[f] >= n_raw) return io_fail(IBA_ERR_IO, "FrameId.yml refers to a frame beyond the LiDAR pose list");
        if (vKFFrameId[0] == 0) {
            for (int f = 0; f < F; ++f) std::memcpy(&Twl[12 * (size_t)f], pose(vKFFrameId[f]), 12 * sizeof(double));
        } else {   // refPose = raw[first].inverse(); refPose * raw[id]
            const double* R0 = pose(vKFFrameId[0]);
            double Ri[9], ti[3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ri[r * 3 + c] = R0[c * 4 + r];
            for (int r = 0; r < 3; ++r) ti[r] = -((Ri[r * 3] * R0[3] + Ri[r * 3 + 1] * R0[7]) + Ri[r * 3 + 2] * R0[11]);
            for (int f = 0; f < F; ++f) {
                const double* T = pose(vKFFrameId[f]);
                double* o = &Twl[12 * (size_t)f];
                for (int r = 0; r < 3; ++r) {
                    for (int c = 0; c < 3; ++c) o[r * 4 + c] = (Ri[r * 3] * T[c] + Ri[r * 3 + 1] * T[4 + c]) + Ri[r * 3 + 2] * T[8 + c];
                    o[r * 4 + 3] = ((Ri[r * 3] * T[3] + Ri[r * 3 + 1] * T[7]) + Ri[r * 3 + 2] * T[11]) + ti[r];
                }
            }
        }
    }
    // ---- Map.yml: MapPoint id -> world position (Map.cc:162-170, MapPoint.cc:440-442) ----
    std::unordered_map<int, std::array<float, 3>> map_points;
    {
        CvYaml y;
        if (!y.load(P->map_file)) return io_fail(IBA_ERR_IO, std::string("cannot open ") + P->map_file);
        const int top = y.find(0, y.size(), 0, "mspMapPoints");
        if (top < 0) return io_fail(IBA_ERR_IO, std::string(P->map_file) + ": 'mspMapPoints' missing");
        const int end = y.block_end(top);
        if (end > top + 1) {
            const int ind = y.lines[top + 1].indent;
            for (int k = top + 1; k < end;) {
                if (y.lines[k].indent != ind) { ++k; continue; }
                const int ke = y.block_end(k);
                if (ke > k + 1) {
                    const int cind = y.lines[k + 1].indent;
                    const int a = y.find(k + 1, ke, cind, "mnId"), b = y.find(k + 1, ke, cind, "mWorldPos");
                    double idv;
                    int rows, cols;
                    std::vector<double> d;
                    if (a < 0 || b < 0 || !y.scalar(a, idv) || !y.matrix(b, rows, cols, d) || rows * cols != 3)
                        return io_fail(IBA_ERR_IO, std::string(P->map_file) + ": malformed MapPoint node");
                    map_points[(int)idv] = {(float)d[0], (float)d[1], (float)d[2]};
                }
                k = ke;
            }
        }
    }
    // ---- KeyFrames/*.yml (System.cc:626-643), sorted by mnId (KeyFrame::lId) ----
    const std::string kf_dir = with_slash(P->keyframe_dir);
    std::vector<std::string> names;
    if (!list_regular_files(kf_dir, names)) return io_fail(IBA_ERR_IO, "Cannot open directory: " + kf_dir);
    std::vector<std::string> info_files;
    for (const std::string& n : names)
        if ((suffix_is(n, "yml") || suffix_is(n, "yaml")) && n != "FrameId.yml") info_files.push_back(kf_dir + n);
    if ((int)info_files.size() != F)
        return io_fail(IBA_ERR_IO, "FrameId.yml lists " + std::to_string(F) + " keyframes but " + kf_dir + " holds " + std::to_string(info_files.size()));
    std::vector<KeyFrameInfo> kfs(F);
    {
        std::atomic<int> next(0);
        const int nt = std::max(1, std::min<int>((int)std::thread::hardware_concurrency(), F));
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t) th.emplace_back([&]() { for (int i; (i = next.fetch_add(1)) < F;) load_keyframe(info_files[i], kfs[i]); });
        for (auto& t : th) t.join();
        for (const KeyFrameInfo& k : kfs)
            if (!k.err.empty()) return io_fail(IBA_ERR_IO, k.err);
    }
    std::sort(kfs.begin(), kfs.end(), [](const KeyFrameInfo& a, const KeyFrameInfo& b) { return a.mnId < b.mnId; });
    std::unordered_map<int, int> KFIdMap;   // mnId -> index (iba_global.cpp:502-505)
    for (int f = 0; f < F; ++f) KFIdMap[kfs[f].mnId] = f;
    for (int f = 0; f < F; ++f) {
        KeyFrameInfo& kf = kfs[f];
        const int K = (int)(kf.uv.size() / 2);
        // restored mmapMpt2Kpt: only MapPoints that exist in the map (KeyFrame.cc:120-129)
        for (int id : kf.mpt_ids) {
            if (!map_points.count(id)) continue;   // "[Warning] Unconnected Map Points"
            const int kp = kf.mpt2kpt_id.at(id);
            if (kp < 0 || kp >= K) return io_fail(IBA_ERR_IO, kf.file + ": keypoint index of a MapPoint is out of range");
            kf.mpt2kpt[id] = kp;
        }
        // covisible keyframes BAError iterates: restored ordered list (ids that exist, KeyFrame.cc:152-159), then the
        // first N, or those with weight >= w (upper_bound with a > b on the stored weights, KeyFrame.cc:426-439)
        std::vector<int> conn;
        for (int id : kf.conn_ids) { auto it = KFIdMap.find(id); if (it != KFIdMap.end()) conn.push_back(it->second); }
        size_t n;
        if (P->num_best_covis > 0) n = std::min<size_t>(conn.size(), (size_t)P->num_best_covis);
        else {
            size_t cnt = 0;
            while (cnt < kf.weights.size() && !(P->min_covis_weight > kf.weights[cnt])) ++cnt;
            n = (conn.empty() || cnt == kf.weights.size()) ? 0 : std::min(cnt, conn.size());   // it == end() -> empty
        }
        kf.covis.assign(conn.begin(), conn.begin() + n);
    }
    // ---- pack ----
    iba_dataset* D = new iba_dataset();
    D->mn_id.resize(F); D->mn_frame_id.resize(F);
    for (int f = 0; f < F; ++f) { D->mn_id[f] = kfs[f].mnId; D->mn_frame_id[f] = vKFFrameId[f]; }
    // point clouds: file index = frame id (iba_global.cpp:494, iba_local.cpp:394)
    const std::string pc_dir = with_slash(P->pointcloud_dir);
    std::vector<std::string> pc_files;
    if (!list_regular_files(pc_dir, pc_files)) { delete D; return io_fail(IBA_ERR_IO, "Cannot open directory: " + pc_dir); }
    std::vector<std::vector<float>> clouds(F);
    {
        for (int f = 0; f < F; ++f)
            if (vKFFrameId[f] < 0 || vKFFrameId[f] >= (int)pc_files.size()) { delete D; return io_fail(IBA_ERR_IO, "FrameId.yml refers to a frame beyond the point-cloud files"); }
        std::atomic<int> next(0);
        std::vector<iba_status> st(F, IBA_OK);
        std::vector<std::string> msg(F);
        const int nt = std::max(1, std::min<int>((int)std::thread::hardware_concurrency(), F));
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&]() {
                for (int i; (i = next.fetch_add(1)) < F;) {
                    st[i] = read_kitti_bin_impl(pc_dir + pc_files[vKFFrameId[i]], P->pointcloud_skip, P->only_positive_x != 0, clouds[i]);
                    if (st[i] != IBA_OK) msg[i] = g_io_err;
                }
            });
        for (auto& t : th) t.join();
        for (int f = 0; f < F; ++f)
            if (st[f] != IBA_OK) { const iba_status s = st[f]; const std::string m = msg[f]; delete D; return io_fail(s, m); }
    }
    D->pt_offset.assign(1, 0); D->kp_offset.assign(1, 0); D->covis_offset.assign(1, 0); D->match_offset.assign(1, 0);
    for (int f = 0; f < F; ++f) {
        const KeyFrameInfo& kf = kfs[f];
        const int K = (int)(kf.uv.size() / 2);
        D->pts_xyz.insert(D->pts_xyz.end(), clouds[f].begin(), clouds[f].end());
        D->pt_offset.push_back(D->pts_xyz.size() / 3);
        std::vector<float>().swap(clouds[f]);
        const double in[6] = {(double)kf.fx, (double)kf.fy, (double)kf.cx, (double)kf.cy, (double)kf.maxX, (double)kf.maxY};
        D->intrinsics.insert(D->intrinsics.end(), in, in + 6);
        D->kp_uv.insert(D->kp_uv.end(), kf.uv.begin(), kf.uv.end());
        // keypoint -> MapPoint: BAError inverts mmapMpt2Kpt (iba_global.cpp:210-213); two MapPoints on one keypoint are
        // resolved by the reference in pointer-hash order, here by the LOWEST MapPoint id (std::map order, first wins)
        std::vector<uint8_t> has(K, 0);
        std::vector<float> mpw(3 * (size_t)K, 0.f);
        for (auto const& pr : kf.mpt2kpt) {
            if (has[pr.second]) continue;
            has[pr.second] = 1;
            const auto& w = map_points.at(pr.first);
            mpw[3 * (size_t)pr.second] = w[0]; mpw[3 * (size_t)pr.second + 1] = w[1]; mpw[3 * (size_t)pr.second + 2] = w[2];
        }
        D->kp_has_mappoint.insert(D->kp_has_mappoint.end(), has.begin(), has.end());
        D->kp_mappoint_w.insert(D->kp_mappoint_w.end(), mpw.begin(), mpw.end());
        D->kp_offset.push_back(D->kp_offset.back() + (uint64_t)K);
        D->Tcw.insert(D->Tcw.end(), kf.Tcw, kf.Tcw + 12);
        for (int g : kf.covis) {
            const KeyFrameInfo& kg = kfs[g];
            float rel[16];
            mul44f(kg.Tcw, kf.Twc, rel);   // pKFConv->GetPose() * InvRefCVPose (iba_global.cpp:280)
            D->covis_frame.push_back(g);
            D->covis_relpose.insert(D->covis_relpose.end(), rel, rel + 12);
            // GetMatchedKptIds (KeyFrame.cc:527-538): keypoints of the two KFs that observe the same MapPoint; one entry
            // per reference keypoint (lowest MapPoint id wins), ordered by reference keypoint
            std::map<int, int> m;
            for (auto const& pr : kf.mpt2kpt) {
                auto it = kg.mpt2kpt.find(pr.first);
                if (it != kg.mpt2kpt.end()) m.insert(std::make_pair(pr.second, it->second));
            }
            for (auto const& pr : m) { D->match_kp_ref.push_back(pr.first); D->match_kp_covis.push_back(pr.second); }
            D->match_offset.push_back(D->match_kp_ref.size());
        }
        D->covis_offset.push_back(D->covis_frame.size());
        // hand-eye pair (iba_global.cpp:264-270): Tc = Tcw_{f+1} * Twc_f (float), Tl = Twl_{f+1}^-1 * Twl_f (double)
        float tc[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
        double tl[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        if (f < F - 1) {
            mul44f(kfs[f + 1].Tcw, kf.Twc, tc);
            const double* A = &Twl[12 * (size_t)(f + 1)];
            const double* B = &Twl[12 * (size_t)f];
            double Ri[9], ti[3];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ri[r * 3 + c] = A[c * 4 + r];
            for (int r = 0; r < 3; ++r) ti[r] = -((Ri[r * 3] * A[3] + Ri[r * 3 + 1] * A[7]) + Ri[r * 3 + 2] * A[11]);
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) tl[r * 4 + c] = (Ri[r * 3] * B[c] + Ri[r * 3 + 1] * B[4 + c]) + Ri[r * 3 + 2] * B[8 + c];
                tl[r * 4 + 3] = ((Ri[r * 3] * B[3] + Ri[r * 3 + 1] * B[7]) + Ri[r * 3 + 2] * B[11]) + ti[r];
            }
        }
        D->Tc_next.insert(D->Tc_next.end(), tc, tc + 12);
        D->Tl_next.insert(D->Tl_next.end(), tl, tl + 12);
    }
    iba_problem_desc& d = D->desc;
    std::memset(&d, 0, sizeof(d));
    d.n_frames = F;
    d.pt_offset = D->pt_offset.data(); d.pts_xyz = D->pts_xyz.data(); d.intrinsics = D->intrinsics.data();
    d.kp_offset = D->kp_offset.data(); d.kp_uv = D->kp_uv.data(); d.kp_has_mappoint = D->kp_has_mappoint.data();
    d.kp_mappoint_w = D->kp_mappoint_w.data(); d.Tcw = D->Tcw.data();
    d.covis_offset = D->covis_offset.data(); d.covis_frame = D->covis_frame.data(); d.covis_relpose = D->covis_relpose.data();
    d.match_offset = D->match_offset.data(); d.match_kp_ref = D->match_kp_ref.data(); d.match_kp_covis = D->match_kp_covis.data();
    d.Tc_next = D->Tc_next.data(); d.Tl_next = D->Tl_next.data();
    *out = D;
    return IBA_OK;
}

}  // extern "C"

// ---- edge list of the ORB-only extrinsic BA, OptimizeExtrinsicGlobal variant (Optimizer.cc:1566-1676; ba_calib.cpp:40-45, 71) ----
struct iba_ba_dataset {
    iba_ba_desc desc;
    std::vector<double> frame_Tlw6, frame_intr, edge_Xw, edge_obs, edge_info;
    std::vector<int32_t> edge_frame, edge_slot;
};

namespace {
void rotvec_from_matrix(const double* R, double* out) {   // Eigen::AngleAxisd(R): quaternion route, angle * axis
    double q[4];
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) { double s = std::sqrt(tr + 1.0); q[3] = 0.5 * s; s = 0.5 / s; q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s; }
    else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[i] = 0.5 * s; s = 0.5 / s;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * s; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * s; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    }
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n < 2.220446049250313e-16) n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    if (n != 0.0) {
        double w = q[3];
        if (w < 0) { n = -n; w = -w; }
        const double angle = 2.0 * std::atan2(std::fabs(n), w);
        for (int i = 0; i < 3; ++i) out[i] = angle * (q[i] / n);
    } else out[0] = out[1] = out[2] = 0.0;
}
}  // namespace

extern "C" {

const iba_ba_desc* iba_ba_dataset_desc(const iba_ba_dataset* d) { return d ? &d->desc : nullptr; }
void iba_ba_dataset_free(iba_ba_dataset* d) { delete d; }

iba_status iba_dataset_load_ba(const iba_dataset_paths* P, int32_t global, iba_ba_dataset** out) {
    if (!P || !out || !P->frame_id_file || !P->lidar_pose_file || !P->keyframe_dir || !P->map_file) return io_fail(IBA_ERR_INVALID_ARG, "null path");
    *out = nullptr;
    std::vector<int> vKFFrameId;
    {
        CvYaml y;
        if (!y.load(P->frame_id_file)) return io_fail(IBA_ERR_IO, std::string("cannot open ") + P->frame_id_file);
        const int b = y.find(0, y.size(), 0, "mnFrameId");
        std::vector<double> tb;
        if (b < 0 || !y.numbers(b, tb)) return io_fail(IBA_ERR_IO, std::string(P->frame_id_file) + ": 'mnFrameId' missing");
        for (double v : tb) vKFFrameId.push_back((int)v);
    }
    const int F = (int)vKFFrameId.size();
    if (F == 0) return io_fail(IBA_ERR_IO, "FrameId.yml lists no keyframes");
    std::vector<double> raw;
    { const iba_status s = read_numbers(P->lidar_pose_file, raw); if (s != IBA_OK) return s; }
    const int n_raw = (int)(raw.size() / 12);
    for (int f = 0; f < F; ++f)
        if (vKFFrameId[f] < 0 || vKFFrameId[f] >= n_raw) return io_fail(IBA_ERR_IO, "FrameId.yml refers to a frame beyond the LiDAR pose list");
    std::unordered_map<int, std::array<float, 3>> map_points;
    {
        CvYaml y;
        if (!y.load(P->map_file)) return io_fail(IBA_ERR_IO, std::string("cannot open ") + P->map_file);
        const int top = y.find(0, y.size(), 0, "mspMapPoints");
        if (top < 0) return io_fail(IBA_ERR_IO, std::string(P->map_file) + ": 'mspMapPoints' missing");
        const int end = y.block_end(top);
        if (end > top + 1) {
            const int ind = y.lines[top + 1].indent;
            for (int k = top + 1; k < end;) {
                if (y.lines[k].indent != ind) { ++k; continue; }
                const int ke = y.block_end(k);
                if (ke > k + 1) {
                    const int cind = y.lines[k + 1].indent;
                    const int a = y.find(k + 1, ke, cind, "mnId"), b = y.find(k + 1, ke, cind, "mWorldPos");
                    double idv; int rows, cols; std::vector<double> dd;
                    if (a < 0 || b < 0 || !y.scalar(a, idv) || !y.matrix(b, rows, cols, dd) || rows * cols != 3) return io_fail(IBA_ERR_IO, std::string(P->map_file) + ": malformed MapPoint node");
                    map_points[(int)idv] = {(float)dd[0], (float)dd[1], (float)dd[2]};
                }
                k = ke;
            }
        }
    }
    const std::string kf_dir = with_slash(P->keyframe_dir);
    std::vector<std::string> names, info_files;
    if (!list_regular_files(kf_dir, names)) return io_fail(IBA_ERR_IO, "Cannot open directory: " + kf_dir);
    for (const std::string& n : names)
        if ((suffix_is(n, "yml") || suffix_is(n, "yaml")) && n != "FrameId.yml") info_files.push_back(kf_dir + n);
    if ((int)info_files.size() != F) return io_fail(IBA_ERR_IO, "FrameId.yml and the keyframe directory disagree on the number of keyframes");
    std::vector<KeyFrameInfo> kfs(F);
    for (int i = 0; i < F; ++i) if (!load_keyframe(info_files[i], kfs[i])) return io_fail(IBA_ERR_IO, kfs[i].err);
    std::sort(kfs.begin(), kfs.end(), [](const KeyFrameInfo& a, const KeyFrameInfo& b) { return a.mnId < b.mnId; });   // System.cc:511-512
    iba_ba_dataset* D = new iba_ba_dataset();
    std::unordered_map<int, int> KFIdMap;
    for (int f = 0; f < F; ++f) KFIdMap[kfs[f].mnId] = f;
    // vTwl as ba_calib.cpp:43-44 builds it: raw LiDAR pose of each keyframe's frame id, in keyframe order
    auto twl_of = [&](int index) { return raw.data() + 12 * (size_t)vKFFrameId[index]; };
    for (int f = 0; f < F; ++f) {
        const KeyFrameInfo& kf = kfs[f];
        const double* Twl = twl_of(f);
        const float* T0 = kfs[0].Tcw;   // Global: Tc0w = allKF[0]->GetPose() (Optimizer.cc:1585)
        double T6[6];
        if (global) {
            const double R[9] = {Twl[0], Twl[1], Twl[2], Twl[4], Twl[5], Twl[6], Twl[8], Twl[9], Twl[10]};
            rotvec_from_matrix(R, T6);                                  // Rlw = Twl.rotation(): the reference's naming (:1627-1632)
            T6[3] = Twl[3]; T6[4] = Twl[7]; T6[5] = Twl[11];
        } else {
            // Local (Optimizer.cc:1441-1462): the oldest of the 20 best covisible keyframes is the reference frame;
            // Told_l = vTwl[nKFID]^-1 * Twl with nKFID = its mnId used AS AN INDEX into vTwl (reproduced), Tl_old = Told_l^-1
            std::vector<int> con;
            for (int id : kf.conn_ids) { auto it = KFIdMap.find(id); if (it != KFIdMap.end()) con.push_back(it->second); }
            if (con.size() > 20) con.resize(20);
            if (con.empty()) { delete D; return io_fail(IBA_ERR_UNSUPPORTED, kf.file + ": keyframe without covisible keyframes (the reference dereferences ConKFS[0])"); }
            int oldest = con[0];
            for (int g : con) if (kfs[g].mnId < kfs[oldest].mnId) oldest = g;   // sort by KeyFrame::lId, take [0]
            T0 = kfs[oldest].Tcw;                                                // T_old_w
            const int nKFID = kfs[oldest].mnId;
            if (nKFID < 0 || nKFID >= F) { delete D; return io_fail(IBA_ERR_UNSUPPORTED, kf.file + ": mnId of the oldest covisible keyframe is not a valid index into the pose list (the reference indexes vTwl with it)"); }
            const double* A = twl_of(nKFID);   // Twold
            double Ri[9], ti[3], M[12];
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ri[r * 3 + c] = A[c * 4 + r];
            for (int r = 0; r < 3; ++r) ti[r] = -((Ri[r * 3] * A[3] + Ri[r * 3 + 1] * A[7]) + Ri[r * 3 + 2] * A[11]);
            for (int r = 0; r < 3; ++r) {      // Told_l = Twold^-1 * Twl
                for (int c = 0; c < 3; ++c) M[r * 4 + c] = (Ri[r * 3] * Twl[c] + Ri[r * 3 + 1] * Twl[4 + c]) + Ri[r * 3 + 2] * Twl[8 + c];
                M[r * 4 + 3] = ((Ri[r * 3] * Twl[3] + Ri[r * 3 + 1] * Twl[7]) + Ri[r * 3 + 2] * Twl[11]) + ti[r];
            }
            double Rinv[9];                    // Told_linv
            for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rinv[r * 3 + c] = M[c * 4 + r];
            rotvec_from_matrix(Rinv, T6);
            for (int r = 0; r < 3; ++r) T6[3 + r] = -((Rinv[r * 3] * M[3] + Rinv[r * 3 + 1] * M[7]) + Rinv[r * 3 + 2] * M[11]);
        }
        D->frame_Tlw6.insert(D->frame_Tlw6.end(), T6, T6 + 6);
        const double in[4] = {(double)kf.fx, (double)kf.fy, (double)kf.cx, (double)kf.cy};
        D->frame_intr.insert(D->frame_intr.end(), in, in + 4);
        const int K = (int)(kf.uv.size() / 2);
        int slot = 0;   // index in the restored mvpMapPoints (KeyFrame.cc:120-129): MapPoints that exist in the map, file order
        for (int id : kf.mpt_ids) {
            auto it = map_points.find(id);
            if (it == map_points.end()) continue;
            const int kp = kf.mpt2kpt_id.at(id);
            if (kp < 0 || kp >= K) { delete D; return io_fail(IBA_ERR_IO, kf.file + ": keypoint index of a MapPoint is out of range"); }
            const int oc = kf.octave[kp];
            if (oc < 0 || oc >= (int)kf.inv_level_sigma2.size()) { delete D; return io_fail(IBA_ERR_IO, kf.file + ": keypoint octave outside mvInvLevelSigma2"); }
            const float* Xw = it->second.data();
            for (int r = 0; r < 3; ++r) {   // T.R * Xw + T.t in CV_32F, T = Tc0w (Global, :1661-1665) or T_old_w (Local, :1493-1497)
                const float t = (T0[r * 4 + 0] * Xw[0] + T0[r * 4 + 1] * Xw[1]) + T0[r * 4 + 2] * Xw[2];
                D->edge_Xw.push_back((double)(t + T0[r * 4 + 3]));
            }
            D->edge_obs.push_back((double)kf.uv[2 * kp]); D->edge_obs.push_back((double)kf.uv[2 * kp + 1]);
            D->edge_info.push_back((double)kf.inv_level_sigma2[oc]);
            D->edge_frame.push_back(f);
            D->edge_slot.push_back(slot++);
        }
    }
    iba_ba_desc& d = D->desc;
    d.n_edges = (int64_t)D->edge_frame.size(); d.n_frames = F;
    d.frame_Tlw6 = D->frame_Tlw6.data(); d.frame_intr = D->frame_intr.data(); d.edge_frame = D->edge_frame.data();
    d.edge_Xw = D->edge_Xw.data(); d.edge_obs = D->edge_obs.data(); d.edge_info = D->edge_info.data(); d.edge_slot = D->edge_slot.data();
    *out = D;
    return IBA_OK;
}

}  // extern "C"
