// iba_func on the MI355X path: the batch evaluator of the reference (src/examples/iba_func.cpp:23-38, 454-471) written
// against the C-ABI only (include/iba_mi355x.h). Reads a text file of candidate 7-vectors [omega, upsilon, s], evaluates
// BAError for each against a dataset directory in the reference's on-disk formats, and writes one line
// "f1 f2 C valid_rate" per candidate with setprecision(precision), lines separated by '\n' (none after the last) — the
// format iba_func writes (:457, :466-468).
//
//   iba_func <FrameId.yml> <lidar_poses.txt> <velodyne_dir> <KeyFrames_dir> <Map.yml> <sim3_list.txt> <out.txt> [precision=6] [device=0]
//   iba_func --config <config.yml> [device=0]     the reference's own call: ONE argument, its yaml-cpp config file (iba_func.cpp:356-406:
//                                                 io.BaseDir/VOIdFile/LOFile/PointCloudDir/init_sim3/res_file/precision, orb.KeyFrameDir/MapFile,
//                                                 runtime.*), read by iba_run_config_* (csrc/iba_config.cpp)
//
// Differences from the reference, on purpose: candidates are evaluated 64 per launch; a trailing newline in the list does
// not produce the junk record the reference's `while (ifs.peek() != EOF)` loop appends (its 7 extractions fail and leave
// the record uninitialised). Without --config the parameters are config/calib/00/iba_calib_global.yml's (iba_default_params + the yml's overrides).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
#include <vector>

#include "iba_mi355x.h"

int main(int argc, char** argv) {
    const bool from_config = argc >= 3 && std::string(argv[1]) == "--config";
    if (!from_config && argc < 8) { std::fprintf(stderr, "usage: %s FrameId.yml lidar_poses velodyne_dir KeyFrames_dir Map.yml sim3_list out [precision] [device]\n       %s --config config.yml [device]\n", argv[0], argv[0]); return 2; }
    int precision = 6, device = 0;
    iba_dataset_paths paths;
    iba_params prm;
    std::string list_file, out_file;
    iba_run_config* cfg = nullptr;
    if (from_config) {
        if (iba_run_config_load(argv[2], &cfg) != IBA_OK || iba_run_config_paths(cfg, 0, &paths) != IBA_OK || iba_run_config_params(cfg, 0, &prm) != IBA_OK) {
            std::fprintf(stderr, "config: %s\n", iba_run_config_last_error()); return 1;
        }
        const char* in = iba_run_config_get(cfg, "io.init_sim3"); const char* res = iba_run_config_get(cfg, "io.res_file"); const char* pr = iba_run_config_get(cfg, "io.precision");
        if (!in || !res || !pr) { std::fprintf(stderr, "config: io.init_sim3 / io.res_file / io.precision missing (iba_func.cpp:401-403)\n"); return 1; }
        list_file = in; out_file = res; precision = std::atoi(pr);   // (used as given, not joined with BaseDir: iba_func.cpp:401-402)
        device = argc > 3 ? std::atoi(argv[3]) : 0;
    } else {
        precision = argc > 8 ? std::atoi(argv[8]) : 6; device = argc > 9 ? std::atoi(argv[9]) : 0;
        paths = iba_dataset_paths{argv[1], argv[2], argv[3], argv[4], argv[5], /*skip*/ 1, /*only_positive_x*/ 0, /*num_best_covis*/ 3, /*min_covis_weight*/ 100};
        iba_default_params(&prm);
        prm.corr_3d_3d_threshold = 10.0; prm.norm_reg_threshold = 0.02; prm.min_diff_dist = 0.2;   // iba_calib_global.yml:26-34
        list_file = argv[6]; out_file = argv[7];
    }
    iba_dataset* ds = nullptr;
    if (iba_dataset_load(&paths, &ds) != IBA_OK) { std::fprintf(stderr, "iba_dataset_load: %s\n", iba_io_last_error()); return 1; }
    iba_handle* h = nullptr;
    const iba_problem_desc* desc = iba_dataset_desc(ds);
    if (iba_create(desc, &prm, device, 0, desc->n_frames, &h) != IBA_OK) { std::fprintf(stderr, "iba_create: %s\n", iba_last_error(nullptr)); return 1; }
    std::vector<double> xs;   // ReadSim3List: whitespace-separated numbers, 7 per record
    {
        std::ifstream ifs(list_file);
        if (!ifs) { std::fprintf(stderr, "Cannot open file: %s\n", list_file.c_str()); return 1; }
        double v;
        while (ifs >> v) xs.push_back(v);
        xs.resize(xs.size() / 7 * 7);
    }
    const size_t n = xs.size() / 7;
    std::ofstream ofs(out_file);
    ofs << std::setprecision(precision);
    std::vector<iba_cost_out> out(IBA_MAX_BATCH);
    for (size_t i0 = 0; i0 < n; i0 += IBA_MAX_BATCH) {
        const int B = (int)std::min<size_t>(IBA_MAX_BATCH, n - i0);
        if (iba_eval_cost(h, xs.data() + 7 * i0, B, out.data()) != IBA_OK) { std::fprintf(stderr, "iba_eval_cost: %s\n", iba_last_error(h)); return 1; }
        for (int b = 0; b < B; ++b) {
            const iba_cost_out& o = out[b];
            const double valid_rate = static_cast<double>(o.valid_cnt_3d_2d) / o.cnt_3d_2d;   // iba_func.cpp:465
            ofs << o.f1 << " " << o.f2 << " " << o.C << " " << valid_rate;
            if (i0 + b != n - 1) ofs << "\n";
            std::printf("%zu | %zu f1: %lf, f2: %lf, C: %lf, valid: %lf %%\n", i0 + b + 1, n, o.f1, o.f2, o.C, valid_rate * 100.);
        }
    }
    ofs.close();
    iba_destroy(h);
    iba_dataset_free(ds);
    iba_run_config_free(cfg);
    return 0;
}
