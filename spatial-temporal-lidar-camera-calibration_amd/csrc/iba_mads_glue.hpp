// Glue between the C structs of the global-stage caller (iba_mads_options / iba_mads_result, include/iba_mi355x.h) and the
// host-side driver (iba_mads.hpp); shared by the device entry points (iba_capi.hip, iba_group.hip) and the host-only ones.
#pragma once
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../include/iba_mi355x.h"
#include "iba_mads.hpp"

namespace iba {
inline void to_mads(const iba_mads_options* opt, MadsOptions& o) {
    o.max_bb_eval = opt->max_bb_eval; o.min_mesh = opt->min_mesh; o.seed = opt->seed;
    o.bases_per_poll = std::max(1, std::min(4, opt->bases_per_poll)); o.speculative = opt->speculative != 0; o.max_batch = IBA_MAX_BATCH; o.vns_max_idle = std::max(0, opt->vns_max_idle);
    for (int i = 0; i < 7; ++i) { o.lb[i] = opt->lb[i]; o.ub[i] = opt->ub[i]; o.init_frame[i] = opt->init_frame[i]; }
}
inline void from_mads(const MadsResult& r, iba_mads_result* res) {
    std::memcpy(res->x, r.best.x, sizeof(res->x));
    res->f = r.best.f; res->c1 = r.best.c[0]; res->c2 = r.best.c[1]; res->c3 = r.best.c[2];
    res->feasible = r.feasible; res->evaluations = r.evaluations; res->iterations = r.iterations; res->batches = r.batches;
    res->cache_hits = r.cache_hits; res->restarts = r.restarts; res->stop_reason = r.stop_reason;
}
inline bool mads_options_ok(const iba_mads_options* opt) {
    if (!opt || opt->max_bb_eval < 1 || !(opt->min_mesh > 0)) return false;
    for (int i = 0; i < 7; ++i) if (!(opt->lb[i] <= opt->ub[i]) || !(opt->init_frame[i] > 0)) return false;
    return true;
}
// copies the recorded evaluations (8 doubles each: x, f) into the caller's buffer; *n = how many there were
inline void hand_over_trace(const std::vector<double>& tr, double* trace, int32_t cap, int32_t* n) {
    const int32_t have = (int32_t)(tr.size() / 8);
    if (n) *n = have;
    if (trace && cap > 0) std::memcpy(trace, tr.data(), sizeof(double) * 8 * (size_t)std::min(have, cap));
}
}  // namespace iba
