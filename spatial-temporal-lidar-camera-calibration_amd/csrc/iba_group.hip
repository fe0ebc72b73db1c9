// Multi-GPU evaluation inside ONE process: the keyframes of a problem sharded over the GPUs of a node, one iba_handle per
// device, one RCCL communicator per device (ncclCommInitAll). The reference's only parallel strategy is the frame loop
// (`#pragma omp parallel for` over keyframes with critical-section sums: iba_global.cpp:193, 239, 318; iba_func.cpp:203;
// iba_local.cpp:162): here every device evaluates its frames into a partial block of B x 64 doubles, ONE
// ncclAllReduce(sum, f64) over xGMI adds the blocks in place on every device, and device 0's copy is finalised on the host
// (iba_finalize_*). No point, keypoint or tree ever crosses a link; the message is 32 KB at B = 64, latency-bound.
// Built on the public single-device entry points only (iba_eval_*_partial, iba_build_problem, iba_finalize_*).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/iba_mi355x.h"
#include "iba_lm.hpp"
#include "iba_mads.hpp"

using namespace iba;

struct iba_group {
    int n = 0;
    std::vector<int> dev;
    std::vector<iba_handle*> h;
    std::vector<ncclComm_t> comm;
    std::vector<hipStream_t> st;
    std::vector<double*> d_part;          // per device: IBA_MAX_BATCH x stride doubles
    std::vector<int32_t> f_begin, f_end;
    double* h_part = nullptr;             // pinned
    iba_params params{};
    int stride = 64;
    std::string err;
};

namespace {
thread_local std::string g_group_create_error;

iba_status gfail(iba_group* g, iba_status s, const std::string& m) { if (g) g->err = m; else g_group_create_error = m; return s; }

#define G_HIP(g, expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return gfail(g, IBA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); } while (0)
#define G_NCCL(g, expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return gfail(g, IBA_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(_r)); } while (0)
#define G_IBA(g, i, expr) do { iba_status _s = (expr); if (_s != IBA_OK) return gfail(g, _s, std::string("device ") + std::to_string((g)->dev[i]) + ": " + iba_last_error((g)->h[i])); } while (0)

// contiguous frame ranges balanced by scan points: the cut before rank r is the first frame boundary at or beyond r / n of
// the points (the rule of shard_frames in the Python plumbing, so that both launch styles partition a problem identically)
void shard(const iba_problem_desc* d, int n, std::vector<int32_t>& b, std::vector<int32_t>& e) {
    const int F = d->n_frames;
    const double tot = F > 0 ? (double)(d->pt_offset[F] - d->pt_offset[0]) : 0.0;
    std::vector<int32_t> cuts(n + 1, 0);
    for (int r = 0; r <= n; ++r) {
        const double target = tot * r / n;
        int i = 0;
        while (i < F && (double)(d->pt_offset[i] - d->pt_offset[0]) < target) ++i;
        cuts[r] = i;
    }
    cuts[0] = 0; cuts[n] = F;
    for (int r = 1; r <= n; ++r) cuts[r] = std::max(cuts[r], cuts[r - 1]);
    b.assign(cuts.begin(), cuts.begin() + n); e.assign(cuts.begin() + 1, cuts.end());
}

// sum of the partial blocks over the devices, in place on every device: ONE collective per evaluation
iba_status allreduce(iba_group* g, int B) {
    G_NCCL(g, ncclGroupStart());
    for (int i = 0; i < g->n; ++i) {
        ncclResult_t r = ncclAllReduce(g->d_part[i], g->d_part[i], (size_t)B * g->stride, ncclDouble, ncclSum, g->comm[i], g->st[i]);
        if (r != ncclSuccess) { ncclGroupEnd(); return gfail(g, IBA_ERR_HIP, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
    }
    G_NCCL(g, ncclGroupEnd());
    return IBA_OK;
}

iba_status fetch(iba_group* g, int B) {   // device 0's summed block -> pinned host memory; every stream drained
    G_HIP(g, hipSetDevice(g->dev[0]));
    G_HIP(g, hipMemcpyAsync(g->h_part, g->d_part[0], sizeof(double) * (size_t)B * g->stride, hipMemcpyDeviceToHost, g->st[0]));
    for (int i = 0; i < g->n; ++i) { G_HIP(g, hipSetDevice(g->dev[i])); G_HIP(g, hipStreamSynchronize(g->st[i])); }
    return IBA_OK;
}
}  // namespace

extern "C" {

const char* iba_group_last_error(const iba_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }
int32_t iba_group_size(const iba_group* g) { return g ? g->n : 0; }

iba_status iba_group_frame_range(const iba_group* g, int32_t rank, int32_t* frame_begin, int32_t* frame_end) {
    if (!g || rank < 0 || rank >= g->n || !frame_begin || !frame_end) return IBA_ERR_INVALID_ARG;
    *frame_begin = g->f_begin[rank]; *frame_end = g->f_end[rank];
    return IBA_OK;
}

void iba_group_destroy(iba_group* g) {
    if (!g) return;
    for (int i = 0; i < g->n; ++i) {
        (void)hipSetDevice(g->dev[i]);
        if (i < (int)g->st.size() && g->st[i]) (void)hipStreamSynchronize(g->st[i]);
        if (i < (int)g->comm.size() && g->comm[i]) (void)ncclCommDestroy(g->comm[i]);
        if (i < (int)g->h.size() && g->h[i]) iba_destroy(g->h[i]);
        if (i < (int)g->d_part.size() && g->d_part[i]) (void)hipFree(g->d_part[i]);
        if (i < (int)g->st.size() && g->st[i]) (void)hipStreamDestroy(g->st[i]);
    }
    if (g->h_part) (void)hipHostFree(g->h_part);
    delete g;
}

iba_status iba_group_create(const iba_problem_desc* desc, const iba_params* params, const int32_t* devices, int32_t n_devices, iba_group** out) {
    g_group_create_error.clear();
    if (!desc || !params || !devices || !out || n_devices < 1) return gfail(nullptr, IBA_ERR_INVALID_ARG, "bad arguments");
    *out = nullptr;
    iba_group* g = new iba_group;
    g->n = n_devices; g->params = *params; g->stride = iba_partial_stride();
    g->dev.assign(devices, devices + n_devices);
    g->h.assign(n_devices, nullptr); g->comm.assign(n_devices, nullptr); g->st.assign(n_devices, nullptr); g->d_part.assign(n_devices, nullptr);
    shard(desc, n_devices, g->f_begin, g->f_end);
    auto bail = [&](iba_status s, const std::string& m) { g_group_create_error = m; iba_group_destroy(g); return s; };
    for (int i = 0; i < n_devices; ++i) {
        const iba_status s = iba_create(desc, params, devices[i], g->f_begin[i], g->f_end[i], &g->h[i]);
        if (s != IBA_OK) return bail(s, std::string("iba_create on device ") + std::to_string(devices[i]) + ": " + iba_last_error(nullptr));
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->st[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipMalloc((void**)&g->d_part[i], sizeof(double) * (size_t)IBA_MAX_BATCH * g->stride);
        if (e != hipSuccess) return bail(IBA_ERR_HIP, std::string("stream / buffer on device ") + std::to_string(devices[i]) + ": " + hipGetErrorString(e));
    }
    if (hipHostMalloc((void**)&g->h_part, sizeof(double) * (size_t)IBA_MAX_BATCH * g->stride) != hipSuccess) return bail(IBA_ERR_HIP, "hipHostMalloc");
    const ncclResult_t r = ncclCommInitAll(g->comm.data(), n_devices, g->dev.data());
    if (r != ncclSuccess) return bail(IBA_ERR_HIP, std::string("ncclCommInitAll: ") + ncclGetErrorString(r));
    *out = g;
    return IBA_OK;
}

iba_status iba_group_set_params(iba_group* g, const iba_params* p) {
    if (!g || !p) return IBA_ERR_INVALID_ARG;
    for (int i = 0; i < g->n; ++i) G_IBA(g, i, iba_set_params(g->h[i], p));
    g->params = *p;
    return IBA_OK;
}

iba_status iba_group_eval_cost(iba_group* g, const double* x, int32_t B, iba_cost_out* out) {
    if (!g || !x || !out || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    for (int b0 = 0; b0 < B; b0 += IBA_MAX_BATCH) {   // larger batches run as consecutive chunks
        const int Bc = std::min(IBA_MAX_BATCH, B - b0);
        for (int i = 0; i < g->n; ++i) G_IBA(g, i, iba_eval_cost_partial(g->h[i], x + 7 * b0, Bc, g->d_part[i], g->st[i]));
        iba_status s = allreduce(g, Bc); if (s != IBA_OK) return s;
        s = fetch(g, Bc); if (s != IBA_OK) return s;
        s = iba_finalize_cost(&g->params, g->h_part, Bc, out + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_eval_bbo(iba_group* g, const double* x, int32_t B, double he_threshold, double valid_rate, iba_bbo* out) {
    if (!g || !out || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    std::vector<iba_cost_out> c((size_t)B);
    const iba_status s = iba_group_eval_cost(g, x, B, c.data()); if (s != IBA_OK) return s;
    for (int b = 0; b < B; ++b) {   // iba_global.cpp:386-392
        out[b].f = c[b].f1 * g->params.err_weight[0] + c[b].f2 * g->params.err_weight[1];
        out[b].c1 = c[b].C - he_threshold; out[b].c2 = -c[b].C - he_threshold;
        out[b].c3 = valid_rate - static_cast<double>(c[b].valid_cnt_3d_2d) / (c[b].cnt_3d_2d + 1);
    }
    return IBA_OK;
}

iba_status iba_group_eval_full(iba_group* g, const double* x, int32_t B, iba_cost_out* cost, iba_normal_out* normal) {
    if (!g || !x || !cost || !normal || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    for (int b0 = 0; b0 < B; b0 += IBA_MAX_BATCH) {
        const int Bc = std::min(IBA_MAX_BATCH, B - b0);
        for (int i = 0; i < g->n; ++i) G_IBA(g, i, iba_eval_full_partial(g->h[i], x + 7 * b0, Bc, g->d_part[i], g->st[i]));
        iba_status s = allreduce(g, Bc); if (s != IBA_OK) return s;
        s = fetch(g, Bc); if (s != IBA_OK) return s;
        s = iba_finalize_cost(&g->params, g->h_part, Bc, cost + b0); if (s != IBA_OK) return s;
        s = iba_finalize_normal(&g->params, g->h_part, Bc, normal + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_eval_normal(iba_group* g, const double* x, int32_t B, iba_normal_out* normal) {
    if (!g || !x || !normal || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    for (int b0 = 0; b0 < B; b0 += IBA_MAX_BATCH) {
        const int Bc = std::min(IBA_MAX_BATCH, B - b0);
        for (int i = 0; i < g->n; ++i) G_IBA(g, i, iba_eval_normal_partial(g->h[i], x + 7 * b0, Bc, g->d_part[i], g->st[i]));
        iba_status s = allreduce(g, Bc); if (s != IBA_OK) return s;
        s = fetch(g, Bc); if (s != IBA_OK) return s;
        s = iba_finalize_normal(&g->params, g->h_part, Bc, normal + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_build_problem(iba_group* g, const double* x_assoc) {
    if (!g || !x_assoc) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    for (int i = 0; i < g->n; ++i) G_IBA(g, i, iba_build_problem(g->h[i], x_assoc));
    return IBA_OK;
}

iba_status iba_group_eval_factors(iba_group* g, const double* x, int32_t B, iba_normal_out* normal) {
    if (!g || !x || !normal || B < 1) return gfail(g, IBA_ERR_INVALID_ARG, "bad arguments");
    for (int b0 = 0; b0 < B; b0 += IBA_MAX_BATCH) {
        const int Bc = std::min(IBA_MAX_BATCH, B - b0);
        for (int i = 0; i < g->n; ++i) G_IBA(g, i, iba_eval_factors_partial(g->h[i], x + 7 * b0, Bc, g->d_part[i], g->st[i]));
        iba_status s = allreduce(g, Bc); if (s != IBA_OK) return s;
        s = fetch(g, Bc); if (s != IBA_OK) return s;
        s = iba_finalize_normal(&g->params, g->h_part, Bc, normal + b0); if (s != IBA_OK) return s;
    }
    return IBA_OK;
}

iba_status iba_group_calibrate_lm(iba_group* g, const double* x0, const iba_lm_options* opt, iba_lm_result* res) {
    if (!g || !x0 || !res) return gfail(g, IBA_ERR_INVALID_ARG, "null argument");
    LmOptions o;
    if (opt) {
        o.max_outer_iterations = opt->max_outer_iterations; o.max_inner_iterations = opt->max_inner_iterations; o.min_diff = opt->min_diff;
        o.function_tolerance = opt->function_tolerance; o.gradient_tolerance = opt->gradient_tolerance; o.parameter_tolerance = opt->parameter_tolerance;
        o.initial_trust_region_radius = opt->initial_trust_region_radius;
    }
    iba_status st = IBA_OK;
    LmResult r;
    const bool ok = calibrate_lm(x0, o,
        [&](const double* x) { st = iba_group_build_problem(g, x); return st == IBA_OK; },
        [&](const double* x, double* H, double* gr, double& cost) {
            iba_normal_out n; st = iba_group_eval_factors(g, x, 1, &n);
            if (st != IBA_OK) return false;
            std::memcpy(H, n.H, sizeof(n.H)); std::memcpy(gr, n.b, sizeof(n.b)); cost = n.cost;
            return true;
        }, r);
    if (!ok) return st == IBA_OK ? IBA_ERR_STATE : st;
    std::memcpy(res->x, r.x, sizeof(r.x));
    res->outer_iterations = r.outer_iterations; res->inner_iterations = r.inner_iterations; res->evaluations = r.evaluations; res->converged = r.converged;
    res->initial_cost = r.initial_cost; res->final_cost = r.final_cost;
    return IBA_OK;
}

iba_status iba_group_calibrate_mads(iba_group* g, const double* x0, const iba_mads_options* opt, iba_mads_result* res) {
    if (!g || !x0 || !res) return gfail(g, IBA_ERR_INVALID_ARG, "null argument");
    iba_mads_options dflt;
    if (!opt) { iba_default_mads_options(x0, &dflt); opt = &dflt; }
    if (opt->max_bb_eval < 1 || !(opt->min_mesh > 0)) return gfail(g, IBA_ERR_INVALID_ARG, "bad MADS options");
    for (int i = 0; i < 7; ++i) if (!(opt->lb[i] <= opt->ub[i]) || !(opt->init_frame[i] > 0)) return gfail(g, IBA_ERR_INVALID_ARG, "bad MADS options (bounds, frame sizes)");
    MadsOptions o;
    o.max_bb_eval = opt->max_bb_eval; o.min_mesh = opt->min_mesh; o.seed = opt->seed;
    o.bases_per_poll = std::max(1, std::min(4, opt->bases_per_poll)); o.speculative = opt->speculative != 0; o.max_batch = IBA_MAX_BATCH; o.vns_max_idle = std::max(0, opt->vns_max_idle);
    for (int i = 0; i < 7; ++i) { o.lb[i] = opt->lb[i]; o.ub[i] = opt->ub[i]; o.init_frame[i] = opt->init_frame[i]; }
    iba_status st = IBA_OK;
    MadsResult r;
    const bool ok = mads_minimize(x0, o, [&](const double* X, int B, MadsPoint* out) {
        iba_bbo bbo[IBA_MAX_BATCH];
        st = iba_group_eval_bbo(g, X, B, opt->he_threshold, opt->valid_rate, bbo);
        if (st != IBA_OK) return false;
        for (int b = 0; b < B; ++b) { out[b].f = bbo[b].f; out[b].c[0] = bbo[b].c1; out[b].c[1] = bbo[b].c2; out[b].c[2] = bbo[b].c3; }
        return true;
    }, r);
    if (!ok) return st == IBA_OK ? IBA_ERR_STATE : st;
    std::memcpy(res->x, r.best.x, sizeof(res->x));
    res->f = r.best.f; res->c1 = r.best.c[0]; res->c2 = r.best.c[1]; res->c3 = r.best.c[2];
    res->feasible = r.feasible; res->evaluations = r.evaluations; res->iterations = r.iterations; res->batches = r.batches;
    res->cache_hits = r.cache_hits; res->restarts = r.restarts; res->stop_reason = r.stop_reason;
    return IBA_OK;
}

// For a caller that runs one process per GPU and owns the communicator (MPI / torchrun style): the one collective of the
// path on the caller's ncclComm_t, so that nothing but this header is needed on the caller's side.
iba_status iba_comm_allreduce(void* nccl_comm, void* d_partials, int32_t B, void* stream) {
    if (!nccl_comm || !d_partials || B < 1) return IBA_ERR_INVALID_ARG;
    const ncclResult_t r = ncclAllReduce(d_partials, d_partials, (size_t)B * (size_t)iba_partial_stride(), ncclDouble, ncclSum, (ncclComm_t)nccl_comm, (hipStream_t)stream);
    return r == ncclSuccess ? IBA_OK : IBA_ERR_HIP;
}

}  // extern "C"
