// HOST-ONLY entry points of the C-ABI (include/iba_mi355x.h): defaults, the host finalisation of summed partial blocks, the
// whitening of normal equations, the MADS driver on analytic black boxes. No HIP in this file: it is part of libiba_mi355x.so and
// is ALSO compiled by plain g++ with -fsanitize=address,undefined / thread for the CPU tier (csrc/san/, `make -C csrc san`).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/iba_mi355x.h"
#include "../../include/iba_mi355x_debug.h"
#include "iba_lm.hpp"
#include "iba_mads.hpp"
#include "iba_host_math.hpp"
#include "iba_mads_glue.hpp"
#include "iba_pair_plan.hpp"
#include "iba_types.hpp"

using namespace iba;

extern "C" {

// debug (host only): what the library derives from a candidate x before any kernel runs — Sim3Exp(x) and its forward-mode derivatives,
// exactly the numbers the factor kernel reads (csrc/iba_host_math.hpp). out66 = R[9], t[3], dR[3][9], dt[6][3], s. The parity tests hand
// them to the oracle's restatement of the kernel's formulas: the device must then equal that CPU evaluation bit for bit.
iba_status iba_debug_cand(const double x[7], double out66[58]) {
    if (!x || !out66) return IBA_ERR_INVALID_ARG;
    static thread_local Cand c;
    make_cand(x, c);
    std::memcpy(out66, c.R, 72); std::memcpy(out66 + 9, c.t, 24); std::memcpy(out66 + 12, c.dR, 216); std::memcpy(out66 + 39, c.dt, 144); out66[57] = c.s;
    return IBA_OK;
}

iba_status iba_default_create_options(iba_create_options* o) {
    if (!o) return IBA_ERR_INVALID_ARG;
    std::memset(o, 0, sizeof(*o));
    o->struct_size = (int32_t)sizeof(*o);
    o->common_pairs = 1; o->common_max_px = 20.0; o->max_pair_groups = 4; o->pair_memo = 1; o->pair_memo_max_batch = 40; o->pair_inflation = 1.25;
    o->anchored_lists = 1; o->anchor_reach = 0.06; o->side_stream = 1; o->spin_wait = 1; o->factor_mfma = 0; o->pair_list_capacity = 0;
    o->max_chain_batch = IBA_MAX_CHAIN; o->chain_fold = 1;
    return IBA_OK;
}

iba_status iba_default_params(iba_params* p) {
    if (!p) return IBA_ERR_INVALID_ARG;
    std::memset(p, 0, sizeof(*p));
    p->max_pixel_dist = 1.5; p->num_min_corr_cost = 30; p->corr_3d_2d_threshold = 40.; p->corr_3d_3d_threshold = 5.;
    p->norm_max_pts = 30; p->norm_min_pts = 5; p->norm_radius = 0.6; p->norm_reg_threshold = 0.04; p->min_diff_dist = 0.01;
    p->err_weight[0] = 1.0; p->err_weight[1] = 1.0; p->use_plane = 1;
    p->num_min_corr = 30; p->max_3d_dist = 1.0; p->neigh_radius = 0.6; p->neigh_max_pts = 30; p->neigh_min_pts = 5;
    p->local_min_diff_dist = 0.2; p->local_norm_reg_threshold = 0.001; p->robust_kernel_delta = 2.98; p->robust_kernel_3ddelta = 1.0;
    p->plane_cache = 1;
    p->factor_3d2d_kind = 0;
    return IBA_OK;
}


int32_t iba_partial_stride(void) { return kPartialStride; }

iba_status iba_finalize_cost(const iba_params* p, const double* part, int32_t B, iba_cost_out* out) {
    if (!p || !part || !out || B < 1) return IBA_ERR_INVALID_ARG;
    for (int b = 0; b < B; ++b) {
        const double* q = part + (size_t)b * kPartialStride; iba_cost_out& o = out[b];
        o.valid_cnt_3d_2d = (int32_t)q[P_VALID_3D2D]; o.cnt_3d_2d = (int32_t)q[P_CNT_3D2D];
        o.cnt_3d_3d = (int32_t)q[P_CNT_3D3D]; o.valid_cnt_3d_3d = (int32_t)q[P_VALID_3D3D];
        o.valid_pl_3d_3d = (int32_t)q[P_VALID_PL]; o.valid_pt_3d_3d = (int32_t)q[P_VALID_PT];
        o.frames_used = (int32_t)q[P_FRAMES]; o.n_corr = (int32_t)q[P_NCORR];
        // iba_global.cpp:330-338
        if (o.valid_cnt_3d_2d == 0 && p->err_weight[0] > 1e-10) o.f1 = std::numeric_limits<double>::max();
        else o.f1 = q[P_SUM_3D2D] / (double)o.valid_cnt_3d_2d;
        if (o.valid_cnt_3d_3d == 0 && p->err_weight[1] > 1e-10) o.f2 = std::numeric_limits<double>::max();
        else o.f2 = (p->err_weight[1] <= 1e-10 ? 0.0 : q[P_SUM_3D3D]) / (double)o.valid_cnt_3d_3d;
        o.C = q[P_HE_SUM] / q[P_HE_CNT];
    }
    return IBA_OK;
}


// ---- Jacobian path ----
iba_status iba_finalize_normal(const iba_params* p, const double* part, int32_t B, iba_normal_out* out) {
    if (!p || !part || !out || B < 1) return IBA_ERR_INVALID_ARG;
    for (int b = 0; b < B; ++b) {
        const double* q = part + (size_t)b * kPartialStride; iba_normal_out& o = out[b];
        int at = 0;
        for (int i = 0; i < 7; ++i)
            for (int j = i; j < 7; ++j) { o.H[i * 7 + j] = q[P_H0 + at]; o.H[j * 7 + i] = q[P_H0 + at]; ++at; }
        for (int i = 0; i < 7; ++i) o.b[i] = q[P_B0 + i];
        o.cost = q[P_COST]; o.chi2 = q[P_CHI2];
        o.n_factor_3d2d = (int32_t)q[P_NF_3D2D]; o.n_factor_p2pl = (int32_t)q[P_NF_P2PL]; o.n_factor_p2pt = (int32_t)q[P_NF_P2PT];
        o.n_residuals = (int32_t)q[P_NRES]; o.frames_used = (int32_t)q[P_FRAMES_N]; o.n_corr = (int32_t)q[P_NCORR_N];
    }
    return IBA_OK;
}


// ---- one 8-row residual block standing for the whole frozen problem (Ceres / g2o adaptors) ----
// [J | r] = upper Cholesky factor of M = [[H, b], [b^T, 2 cost]]: J^T J = H, J^T r = b, |r|^2 = 2 cost. M is positive
// semi-definite: it is a sum over residual blocks of w [J_k | r_k]^T [J_k | r_k] plus (rho_k - w_k s_k) >= 0 on the last
// diagonal entry (Huber: rho(s) - rho'(s) s = a (sqrt(s) - a) > 0 beyond the kink). A pivot that is not positive relative to
// its column (rank-deficient H: no factor constrains that direction) leaves a zero row.
iba_status iba_whiten_normal(const iba_normal_out* n, double r[8], double J[56]) {
    if (!n || !r || !J) return IBA_ERR_INVALID_ARG;
    double M[64];
    for (int i = 0; i < 7; ++i) { for (int j = 0; j < 7; ++j) M[i * 8 + j] = n->H[i * 7 + j]; M[i * 8 + 7] = n->b[i]; M[7 * 8 + i] = n->b[i]; }
    M[63] = 2.0 * n->cost;
    double R[64]; std::memset(R, 0, sizeof(R));   // upper triangular, R^T R = M
    for (int i = 0; i < 8; ++i) {
        double d = M[i * 8 + i];
        for (int k = 0; k < i; ++k) d -= R[k * 8 + i] * R[k * 8 + i];
        if (!(d > 1e-14 * std::fabs(M[i * 8 + i])) || !(d > 0)) continue;   // zero row
        const double rii = std::sqrt(d);
        R[i * 8 + i] = rii;
        for (int j = i + 1; j < 8; ++j) {
            double v = M[i * 8 + j];
            for (int k = 0; k < i; ++k) v -= R[k * 8 + i] * R[k * 8 + j];
            R[i * 8 + j] = v / rii;
        }
    }
    for (int i = 0; i < 8; ++i) { for (int j = 0; j < 7; ++j) J[i * 7 + j] = R[i * 8 + j]; r[i] = R[i * 8 + 7]; }
    return IBA_OK;
}


iba_status iba_default_lm_options(iba_lm_options* o) {
    if (!o) return IBA_ERR_INVALID_ARG;
    const LmOptions d;
    o->max_outer_iterations = d.max_outer_iterations; o->max_inner_iterations = d.max_inner_iterations; o->min_diff = d.min_diff;
    o->function_tolerance = d.function_tolerance; o->gradient_tolerance = d.gradient_tolerance; o->parameter_tolerance = d.parameter_tolerance;
    o->initial_trust_region_radius = d.initial_trust_region_radius;
    return IBA_OK;
}


// ---- global stage caller (csrc/iba_mads.hpp) ----
iba_status iba_default_mads_options(const double* x0, iba_mads_options* o) {
    if (!x0 || !o) return IBA_ERR_INVALID_ARG;
    static const double lb[7] = {-0.1, -0.1, -0.1, -0.3, -0.3, -0.3, -1.0}, ub[7] = {0.1, 0.1, 0.1, 0.3, 0.3, 0.3, 1.0};   // iba_calib_global.yml:39-40
    o->max_bb_eval = 5000;
    for (int i = 0; i < 7; ++i) { o->lb[i] = x0[i] + lb[i]; o->ub[i] = x0[i] + ub[i]; o->init_frame[i] = 0.5; }
    o->min_mesh = 1e-6; o->he_threshold = 0.094; o->valid_rate = 0.95; o->seed = 0; o->bases_per_poll = 2; o->speculative = 1; o->vns_max_idle = 6;
    return IBA_OK;
}

iba_status iba_mads_selftest_trace(int32_t problem, const double* x0, const iba_mads_options* opt, iba_mads_result* res, double* trace, int32_t cap, int32_t* n_trace) {
    if (!x0 || !res || !mads_options_ok(opt) || problem < 0 || problem > 3) return IBA_ERR_INVALID_ARG;
    MadsOptions o; to_mads(opt, o);
    std::vector<double> tr;
    if (trace || n_trace) o.trace = &tr;
    static const double a[7] = {0.3, -0.2, 0.1, 0.25, -0.15, 0.05, 9.5};
    MadsResult r;
    mads_minimize(x0, o, [&](const double* X, int B, MadsPoint* out) {
        for (int b = 0; b < B; ++b) {
            const double* x = X + 7 * b;
            double f = 0, c0 = -1, c1 = -1, c2 = -1;
            if (problem == 0) { for (int i = 0; i < 7; ++i) f += (1.0 + i) * (x[i] - a[i]) * (x[i] - a[i]); }
            else if (problem == 1) { f = (x[0] - 1.0) * (x[0] - 1.0); for (int i = 1; i < 7; ++i) f += (x[i] - a[i]) * (x[i] - a[i]); c0 = x[0] - 0.5; }
            else if (problem == 3) {   // many narrow local basins (period 0.08) under a shallow bowl: the global one is at a
                const double two_pi = 6.283185307179586;
                for (int i = 0; i < 7; ++i) { const double d = x[i] - a[i] - 0.0123 * (i + 1); f += 2.0 * d * d + 0.3 * (1.0 - std::cos(two_pi * d / 0.08)); }
            }
            else {
                double m = 0, s1 = 0;
                for (int i = 0; i < 7; ++i) { const double d = std::fabs(x[i] - a[i]); m = std::max(m, d); s1 += d; }
                f = m + 0.1 * s1; c0 = 0.2 - x[1]; c1 = x[3] + x[4] - 0.05;   // optimum on both constraint boundaries
            }
            out[b].f = f; out[b].c[0] = c0; out[b].c[1] = c1; out[b].c[2] = c2;
        }
        return true;
    }, r);
    from_mads(r, res);
    hand_over_trace(tr, trace, cap, n_trace);
    return IBA_OK;
}
iba_status iba_mads_selftest(int32_t problem, const double* x0, const iba_mads_options* opt, iba_mads_result* res) {
    return iba_mads_selftest_trace(problem, x0, opt, res, nullptr, 0, nullptr);
}



// The planner of the shared pair searches on B candidates x (host only): how plan_pairs would group them on a handle whose largest
// focal length is max_fx — group index per candidate, nominal projection spread per group, number of groups (1: the whole batch
// shares one search; 0: wide everywhere, every candidate searches for itself).
iba_status iba_debug_plan_groups(const double* x, int32_t B, double max_fx, double max_px, int32_t max_groups, int32_t* group_of, double* group_px, int32_t* n_groups) {
    if (!x || B < 1 || B > kMaxChain || !n_groups) return IBA_ERR_INVALID_ARG;
    static thread_local Cand hc[kMaxChain];
    static thread_local GroupPick gp[kMaxPairGroups];
    for (int b = 0; b < B; ++b) make_cand_values(x + 7 * b, hc[b]);
    gp[0].n = B; for (int b = 0; b < B; ++b) gp[0].idx[b] = b;
    int ng = 0;
    if (pick_group(max_fx, hc, gp[0])) ng = gp[0].px <= max_px ? 1 : cluster_batch(max_fx, hc, B, max_px, max_groups, gp);
    *n_groups = ng;
    for (int g = 0; g < ng; ++g) {
        if (group_px) group_px[g] = gp[g].px;
        if (group_of) for (int j = 0; j < gp[g].n; ++j) group_of[gp[g].idx[j]] = g;
    }
    return IBA_OK;
}

}  // extern "C"
