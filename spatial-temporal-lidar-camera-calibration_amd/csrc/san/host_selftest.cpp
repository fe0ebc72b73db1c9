// CPU-tier self-test of the HOST-ONLY sources (iba_hostonly.cpp, iba_io.cpp, iba_handeye.cpp) under
// -fsanitize=address,undefined (`make -C csrc san`, run by tests/test_sanitizers_cpu.py). Functional parity of these files is
// asserted elsewhere (tests/test_formats_cpu.py, test_handeye_cpu.py, test_mads_cpu.py against their oracles); here the same
// code paths run with the sanitizers on, plus what a file parser must survive: truncated and garbled inputs.
//   usage: host_selftest [dataset_dir]     dataset_dir = a directory in the reference's on-disk formats (FrameId.yml,
//   lidar_poses.txt, velodyne/, KeyFrames/, Map.yml — the layout tests/test_formats_cpu.py writes); optional.
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

#include "../../../include/iba_mi355x.h"
#include "../../../include/iba_mi355x_debug.h"

static int failures = 0;
#define CHECK(c) do { if (!(c)) { std::fprintf(stderr, "host_selftest: %s failed at line %d\n", #c, __LINE__); ++failures; } } while (0)

static std::string slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::string(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>()); }
static void spit(const std::string& p, const std::string& s) { std::ofstream f(p, std::ios::binary); f.write(s.data(), (std::streamsize)s.size()); }

static void mads_and_whitening() {
    for (int problem = 0; problem < 4; ++problem) {
        double x0[7] = {0, 0, 0, 0, 0, 0, 9.0};
        iba_mads_options o;
        CHECK(iba_default_mads_options(x0, &o) == IBA_OK);
        o.max_bb_eval = 1500;
        iba_mads_result r;
        std::vector<double> tr(8 * 1600);
        int32_t n = 0;
        CHECK(iba_mads_selftest_trace(problem, x0, &o, &r, tr.data(), 1600, &n) == IBA_OK);
        CHECK(n > 0 && n <= 1500 + 64 && r.evaluations == n);
        CHECK(iba_mads_selftest_trace(problem, x0, &o, &r, tr.data(), 10, &n) == IBA_OK);   // a trace buffer smaller than the run
    }
    iba_normal_out nrm;
    std::memset(&nrm, 0, sizeof(nrm));
    for (int i = 0; i < 7; ++i) { nrm.H[i * 7 + i] = 2.0 + i; nrm.b[i] = 0.1 * i; }
    nrm.H[1] = nrm.H[7] = 0.3; nrm.cost = 5.0;
    double r8[8], J[56];
    CHECK(iba_whiten_normal(&nrm, r8, J) == IBA_OK);
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j < 7; ++j) { double a = 0; for (int k = 0; k < 8; ++k) a += J[k * 7 + i] * J[k * 7 + j]; CHECK(std::fabs(a - nrm.H[i * 7 + j]) < 1e-12); }
    std::memset(&nrm, 0, sizeof(nrm));   // rank-deficient: zero rows, no division by zero
    CHECK(iba_whiten_normal(&nrm, r8, J) == IBA_OK);
    iba_params p; CHECK(iba_default_params(&p) == IBA_OK);
    std::vector<double> part((size_t)3 * iba_partial_stride(), 0.0);
    iba_cost_out co[3]; iba_normal_out no[3];
    CHECK(iba_finalize_cost(&p, part.data(), 3, co) == IBA_OK && iba_finalize_normal(&p, part.data(), 3, no) == IBA_OK);
}

static void handeye() {
    // camera motions = X (scaled) LiDAR motions X^-1 for a known X: the closed form and both refinements run to the end
    const int n = 40;
    std::vector<double> Ta(12 * n), Tb(12 * n);
    auto rot = [](double ax, double ay, double az, double* R) {
        const double th = std::sqrt(ax * ax + ay * ay + az * az);
        const double k = th > 1e-12 ? std::sin(th) / th : 1.0, c = th > 1e-12 ? (1 - std::cos(th)) / (th * th) : 0.5;
        const double K[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) { double kk = 0; for (int m = 0; m < 3; ++m) kk += K[i * 3 + m] * K[m * 3 + j]; R[i * 3 + j] = (i == j) + k * K[i * 3 + j] + c * kk; }
    };
    double RX[9]; rot(0.3, -1.1, 0.9, RX);
    const double tX[3] = {0.05, -0.08, -0.27}, s = 0.2;
    for (int i = 0; i < n; ++i) {
        double RB[9]; rot(0.02 * std::sin(i), 0.03 * std::cos(0.7 * i), 0.05 * std::sin(0.3 * i), RB);
        const double tB[3] = {1.0 + 0.1 * std::sin(i), 0.05 * std::cos(i), 0.02 * i / n};
        // A = X B X^-1 with translation / s
        double RA[9], tmp[9];
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double a = 0; for (int m = 0; m < 3; ++m) a += RX[r * 3 + m] * RB[m * 3 + c]; tmp[r * 3 + c] = a; }
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { double a = 0; for (int m = 0; m < 3; ++m) a += tmp[r * 3 + m] * RX[c * 3 + m]; RA[r * 3 + c] = a; }
        double tA[3];
        for (int r = 0; r < 3; ++r) { double a = tX[r]; for (int m = 0; m < 3; ++m) a += RX[r * 3 + m] * tB[m]; for (int m = 0; m < 3; ++m) a -= RA[r * 3 + m] * tX[m]; tA[r] = a / s; }
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) { Ta[12 * i + 4 * r + c] = RA[r * 3 + c]; Tb[12 * i + 4 * r + c] = RB[r * 3 + c]; } Ta[12 * i + 4 * r + 3] = tA[r]; Tb[12 * i + 4 * r + 3] = tB[r]; }
    }
    double rigid[12], scale = 0;
    CHECK(iba_handeye(Ta.data(), Tb.data(), n, rigid, &scale) == IBA_OK);
    CHECK(std::fabs(scale - s) < 1e-6);
    double r2[12], s2 = 0;
    CHECK(iba_handeye_robust(Ta.data(), Tb.data(), n, rigid, scale, 0.1, 1, 0.005, 10, r2, &s2) == IBA_OK);
    CHECK(iba_handeye_lineprocess(Ta.data(), Tb.data(), n, rigid, scale, 10, 64.0, 1.4, 0.1, 20, 1, 0.005, r2, &s2) == IBA_OK);
    std::vector<double> mot(12 * (n - 1));
    CHECK(iba_pose_to_motion(Ta.data(), n, mot.data()) == IBA_OK);
    double x[7], back[12], sb = 0;
    CHECK(iba_handeye(Ta.data(), Tb.data(), 1, rigid, &scale) != IBA_OK || true);   // a degenerate input must fail or succeed cleanly
    (void)x; (void)back; (void)sb;
}

static void small_files(const std::string& dir) {
    // KITTI .bin: XYZI float32 records, incl. a truncated tail
    std::vector<float> pts;
    for (int i = 0; i < 101; ++i) { pts.push_back(i % 2 ? 1.0f * i : -1.0f * i); pts.push_back(0.5f * i); pts.push_back(-0.1f * i); pts.push_back(0.3f); }
    const std::string bin = dir + "/a.bin";
    spit(bin, std::string((const char*)pts.data(), pts.size() * 4));
    for (int skip : {1, 2, 7, 200})
        for (int pos : {0, 1}) {
            float* xyz = nullptr; int64_t np = 0;
            const iba_status st = iba_read_kitti_bin(bin.c_str(), skip, pos, &xyz, &np);
            CHECK(skip > 101 ? st != IBA_OK : st == IBA_OK);   // skip > number of points: the reference's unsigned `num_points - skip` wraps; refused here
            if (st == IBA_OK) { CHECK(np >= 0 && np <= 101); iba_io_free(xyz); }
        }
    spit(bin, std::string((const char*)pts.data(), pts.size() * 4 - 7));   // ends inside a record
    { float* xyz = nullptr; int64_t np = 0; const iba_status st = iba_read_kitti_bin(bin.c_str(), 1, 0, &xyz, &np); if (st == IBA_OK) { CHECK(np <= 100); iba_io_free(xyz); } }
    { float* xyz = nullptr; int64_t np = 0; CHECK(iba_read_kitti_bin((dir + "/missing.bin").c_str(), 1, 0, &xyz, &np) == IBA_ERR_IO); CHECK(std::strlen(iba_io_last_error()) > 0); }
    // pose list with a ragged last line, sim3 round trip, garbage
    const std::string pl = dir + "/poses.txt";
    spit(pl, "1 0 0 0.5 0 1 0 0 0 0 1 0\n1 0 0 1.5 0 1 0 0 0 0 1 0\n1 0 0\n");
    { double* p12 = nullptr; int64_t n = 0; CHECK(iba_read_pose_list(pl.c_str(), &p12, &n) == IBA_OK); CHECK(n == 2); iba_io_free(p12); }
    spit(pl, "not numbers at all\n");
    { double* p12 = nullptr; int64_t n = 0; const iba_status st = iba_read_pose_list(pl.c_str(), &p12, &n); if (st == IBA_OK) { CHECK(n == 0); iba_io_free(p12); } }
    const double rigid[12] = {1, 0, 0, 0.1, 0, 1, 0, -0.2, 0, 0, 1, 0.3};
    const std::string sf = dir + "/sim3.txt";
    CHECK(iba_write_sim3(sf.c_str(), rigid, 0.25) == IBA_OK);
    double back[12], sc = 0;
    CHECK(iba_read_sim3(sf.c_str(), back, &sc) == IBA_OK && sc == 0.25 && back[3] == 0.1);
    spit(sf, "1 2 3");
    CHECK(iba_read_sim3(sf.c_str(), back, &sc) == IBA_OK && sc == 1.0 && back[0] == 1.0 && back[3] == 0.0);   // the reference starts from identity / scale 1 and keeps what the stream does not yield (kitti_tools.h:146-158)
}

static void dataset(const std::string& root, const std::string& scratch) {
    iba_dataset_paths p;
    std::memset(&p, 0, sizeof(p));
    const std::string fid = root + "/FrameId.yml", lo = root + "/lidar_poses.txt", pc = root + "/velodyne", kf = root + "/KeyFrames", mp = root + "/Map.yml";
    p.frame_id_file = fid.c_str(); p.lidar_pose_file = lo.c_str(); p.pointcloud_dir = pc.c_str(); p.keyframe_dir = kf.c_str(); p.map_file = mp.c_str();
    p.pointcloud_skip = 1; p.only_positive_x = 0; p.num_best_covis = 3; p.min_covis_weight = 100;
    iba_dataset* d = nullptr;
    CHECK(iba_dataset_load(&p, &d) == IBA_OK);
    if (d) {
        const iba_problem_desc* desc = iba_dataset_desc(d);
        CHECK(desc && desc->n_frames > 0);
        int32_t a = 0, b = 0;
        CHECK(iba_dataset_frame_ids(d, 0, &a, &b) == IBA_OK);
        CHECK(iba_dataset_frame_ids(d, desc->n_frames, &a, &b) != IBA_OK);
        iba_dataset_free(d);
    }
    for (int global = 0; global < 2; ++global) { iba_ba_dataset* bd = nullptr; if (iba_dataset_load_ba(&p, global, &bd) == IBA_OK) { CHECK(iba_ba_dataset_desc(bd) != nullptr); iba_ba_dataset_free(bd); } else { std::fprintf(stderr, "iba_dataset_load_ba(global=%d): %s\n", global, iba_io_last_error()); CHECK(false); } }
    p.num_best_covis = -1; p.min_covis_weight = 1;   // covisibility by weight
    d = nullptr; if (iba_dataset_load(&p, &d) == IBA_OK) iba_dataset_free(d); else CHECK(false);
    p.num_best_covis = 3;
    // every yml of the dataset cut at a sweep of byte offsets, and with bytes flipped: the loader must fail or load, cleanly
    const std::string victims[3] = {fid, mp, kf + "/000000.yml"};
    for (const std::string& v : victims) {
        const std::string whole = slurp(v);
        if (whole.empty()) continue;
        const std::string tmp = scratch + "/victim.yml";
        for (int cut = 0; cut < 24; ++cut) {
            std::string s = whole.substr(0, whole.size() * (size_t)cut / 24);
            spit(tmp, s);
            iba_dataset_paths q = p;
            std::string kfdir = kf;
            if (v == fid) q.frame_id_file = tmp.c_str(); else if (v == mp) q.map_file = tmp.c_str();
            else {   // a keyframe directory with the cut file in place of keyframe 0 and links to the others
                kfdir = scratch + "/kf_cut"; (void)!system(("rm -rf '" + kfdir + "' && mkdir -p '" + kfdir + "' && for f in '" + kf + "'/*.yml; do ln -s \"$f\" '" + kfdir + "/'; done && rm -f '" + kfdir + "/000000.yml' && cp '" + tmp + "' '" + kfdir + "/000000.yml'").c_str());
                q.keyframe_dir = kfdir.c_str();
            }
            iba_dataset* dd = nullptr;
            const iba_status st = iba_dataset_load(&q, &dd);
            if (st == IBA_OK) iba_dataset_free(dd); else CHECK(std::strlen(iba_io_last_error()) > 0);
        }
        unsigned lcg = 12345u;
        for (int trial = 0; trial < 16 && v != kf + "/000000.yml"; ++trial) {
            std::string s = whole;
            for (int k = 0; k < 8; ++k) { lcg = lcg * 1664525u + 1013904223u; s[(size_t)(lcg >> 8) % s.size()] = (char)(" :[]-,.0a\n"[(lcg >> 3) % 10]); }
            spit(tmp, s);
            iba_dataset_paths q = p;
            if (v == fid) q.frame_id_file = tmp.c_str(); else q.map_file = tmp.c_str();
            iba_dataset* dd = nullptr;
            if (iba_dataset_load(&q, &dd) == IBA_OK) iba_dataset_free(dd);
        }
    }
}

int main(int argc, char** argv) {
    char tmpl[] = "/tmp/iba_san_XXXXXX";
    const char* scratch = mkdtemp(tmpl);
    if (!scratch) { std::perror("mkdtemp"); return 2; }
    mads_and_whitening();
    handeye();
    small_files(scratch);
    if (argc > 1) dataset(argv[1], scratch);
    (void)!system((std::string("rm -rf '") + scratch + "'").c_str());
    std::printf("host_selftest: %d failure(s)%s\n", failures, argc > 1 ? " (with dataset)" : "");
    return failures ? 1 : 0;
}
