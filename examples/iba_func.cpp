// iba_func on the MI355X path: the batch evaluator of the reference (src/examples/iba_func.cpp:23-38, 454-471) written
// against the C-ABI only (include/iba_mi355x.h). Reads a text file of candidate 7-vectors [omega, upsilon, s], evaluates
// BAError for each against a dataset directory in the reference's on-disk formats, and writes one line
// "f1 f2 C valid_rate" per candidate with setprecision(precision), lines separated by '\n' (none after the last) — the
// format iba_func writes (:457, :466-468).
//
//   iba_func <FrameId.yml> <lidar_poses.txt> <velodyne_dir> <KeyFrames_dir> <Map.yml> <sim3_list.txt> <out.txt> [precision=6] [device=0]
//
// Differences from the reference, on purpose: candidates are evaluated 64 per launch; a trailing newline in the list does
// not produce the junk record the reference's `while (ifs.peek() != EOF)` loop appends (its 7 extractions fail and leave
// the record uninitialised). Parameters are config/calib/00/iba_calib_global.yml's (iba_default_params + the yml's overrides).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <vector>

#include "iba_mi355x.h"

int main(int argc, char** argv) {
    if (argc < 8) { std::fprintf(stderr, "usage: %s FrameId.yml lidar_poses velodyne_dir KeyFrames_dir Map.yml sim3_list out [precision] [device]\n", argv[0]); return 2; }
    const int precision = argc > 8 ? std::atoi(argv[8]) : 6, device = argc > 9 ? std::atoi(argv[9]) : 0;
    iba_dataset_paths paths = {argv[1], argv[2], argv[3], argv[4], argv[5], /*skip*/ 1, /*only_positive_x*/ 0, /*num_best_covis*/ 3, /*min_covis_weight*/ 100};
    iba_dataset* ds = nullptr;
    if (iba_dataset_load(&paths, &ds) != IBA_OK) { std::fprintf(stderr, "iba_dataset_load: %s\n", iba_io_last_error()); return 1; }
    iba_params prm;
    iba_default_params(&prm);
    prm.corr_3d_3d_threshold = 10.0; prm.norm_reg_threshold = 0.02; prm.min_diff_dist = 0.2;   // iba_calib_global.yml:26-34
    iba_handle* h = nullptr;
    const iba_problem_desc* desc = iba_dataset_desc(ds);
    if (iba_create(desc, &prm, device, 0, desc->n_frames, &h) != IBA_OK) { std::fprintf(stderr, "iba_create: %s\n", iba_last_error(nullptr)); return 1; }
    std::vector<double> xs;   // ReadSim3List: whitespace-separated numbers, 7 per record
    {
        std::ifstream ifs(argv[6]);
        if (!ifs) { std::fprintf(stderr, "Cannot open file: %s\n", argv[6]); return 1; }
        double v;
        while (ifs >> v) xs.push_back(v);
        xs.resize(xs.size() / 7 * 7);
    }
    const size_t n = xs.size() / 7;
    std::ofstream ofs(argv[7]);
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
    return 0;
}
