// Host-side planner of the shared 2d-3d pair searches (see iba_pairs_kernel in iba_split_kernels.hpp): the spread of a group of
// candidates around a reference, the nominal projection spread that decides whether a group may share one search, and the
// clustering of a wide batch into tight groups. HOST ONLY (no HIP): used by iba_capi.hip (plan_pairs) and, through
// iba_debug_plan_groups, by the CPU tests and tools.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

#include "../../include/iba_mi355x.h"
#include "iba_types.hpp"

namespace iba {

constexpr int kMaxPairGroups = 4;
struct GroupRef { double R[9], t[3], rho[9], tau[3]; };   // reference candidate of a group and the entrywise bound of the group's motion around it

// The reference candidate of a batch and the spread of the batch around it (see iba_pairs_kernel): reference = the candidate
// nearest the batch mean, rho_ij = max_b |R_b R_0^T - I|_ij, tau_i = max_b |t_b - R_b R_0^T t_0|_i, both inflated for their own
// rounding. Returns false when the batch is too wide for common pairs to pay (nominal projection spread above max_px at a
// point 12 m out, 10 m deep), in which case every candidate searches for itself (iba_assoc_kernel).
// rho_ij = max_b |R_b R_0^T - I|_ij, tau_i = max_b |t_b - R_b R_0^T t_0|_i of a batch around a reference (R_0, t_0), inflated for their own
// rounding; rel (optional): the candidates' own (M_b, a_b) as floats. false: a NaN / absurd candidate, no bound.
// (idx: the members of the group among hc[], nullptr = hc[0 .. B))
inline bool batch_spread(const Cand* hc_all, int B, const double* R0, const double* t0, double* rho, double* tau, float (*rel)[12], const int* idx = nullptr) {
    for (int i = 0; i < 9; ++i) rho[i] = 0;
    for (int i = 0; i < 3; ++i) tau[i] = 0;
    for (int b = 0; b < B; ++b) {
        const Cand& c = hc_all[idx ? idx[b] : b];
        double A[9];
        for (int r = 0; r < 3; ++r)
            for (int q = 0; q < 3; ++q) A[r * 3 + q] = (c.R[r * 3] * R0[q * 3] + c.R[r * 3 + 1] * R0[q * 3 + 1]) + c.R[r * 3 + 2] * R0[q * 3 + 2];   // R_b R_0^T
        for (int r = 0; r < 3; ++r) {
            const double a = c.t[r] - ((A[r * 3] * t0[0] + A[r * 3 + 1] * t0[1]) + A[r * 3 + 2] * t0[2]);
            if (!(std::fabs(a) <= 1e30)) return false;
            tau[r] = std::max(tau[r], std::fabs(a));
            if (rel) rel[b][9 + r] = (float)a;
            for (int q = 0; q < 3; ++q) { const double m = A[r * 3 + q] - (r == q ? 1.0 : 0.0), e = std::fabs(m); if (!(e <= 4.0)) return false; rho[r * 3 + q] = std::max(rho[r * 3 + q], e); if (rel) rel[b][r * 3 + q] = (float)m; }
        }
    }
    for (int i = 0; i < 9; ++i) rho[i] = rho[i] * (1.0 + 1e-9) + 1e-15;
    for (int i = 0; i < 3; ++i) tau[i] = tau[i] * (1.0 + 1e-9) + 1e-15;
    return true;
}

constexpr int kOwnBoundMax = 64;     // a group of at most this many candidates may hand its members' own motions to the pair search (one lane each; PairsPlan::rel)
struct GroupPick { int n = 0; int idx[kMaxChain]; int ref = 0; GroupRef gr; float rel[kOwnBoundMax][12]; double px = 0; bool ok = false; };
inline double nominal_px_of(double max_fx, const double* rho, const double* tau) {
    double rho_row = 0, tau_max = 0;
    for (int r = 0; r < 3; ++r) { rho_row = std::max(rho_row, rho[r * 3] + rho[r * 3 + 1] + rho[r * 3 + 2]); tau_max = std::max(tau_max, tau[r]); }
    return max_fx * (rho_row * 12.0 + tau_max) * 1.8 / 10.0;
}
// reference (the member nearest the group's mean) and spread of a group of candidates
inline bool pick_group(double max_fx, const Cand* hc, GroupPick& g) {
    double mean[12] = {0};
    for (int j = 0; j < g.n; ++j) { const Cand& c = hc[g.idx[j]]; for (int i = 0; i < 9; ++i) mean[i] += c.R[i]; for (int i = 0; i < 3; ++i) mean[9 + i] += c.t[i]; }
    for (double& m : mean) m /= (double)g.n;
    double best = INFINITY; g.ref = g.idx[0];
    for (int j = 0; j < g.n; ++j) {
        const Cand& c = hc[g.idx[j]];
        double d = 0;
        for (int i = 0; i < 9; ++i) d = std::max(d, 12.0 * std::fabs(c.R[i] - mean[i]));
        for (int i = 0; i < 3; ++i) d = std::max(d, std::fabs(c.t[i] - mean[9 + i]));
        if (d < best) { best = d; g.ref = g.idx[j]; }
    }
    if (!(best < INFINITY)) return false;   // a NaN candidate: no bound
    std::memcpy(g.gr.R, hc[g.ref].R, sizeof(g.gr.R)); std::memcpy(g.gr.t, hc[g.ref].t, sizeof(g.gr.t));
    if (!batch_spread(hc, g.n, g.gr.R, g.gr.t, g.gr.rho, g.gr.tau, g.n <= kOwnBoundMax ? g.rel : nullptr, g.idx)) return false;
    g.px = nominal_px_of(max_fx, g.gr.rho, g.gr.tau);
    return true;
}
inline double cand_px(double max_fx, const Cand& a, const Cand& r) {   // nominal projection distance of candidate a from a reference r
    double rho[9], tau[3];
    if (!batch_spread(&a, 1, r.R, r.t, rho, tau, nullptr)) return INFINITY;
    return nominal_px_of(max_fx, rho, tau);
}

// Greedy clustering of a batch whose whole spread exceeds max_px: farthest-point seeds (the first is the candidate nearest the
// batch mean), every candidate joins its nearest seed; accepted as soon as EVERY group's own nominal spread is at most max_px, given
// up beyond max_groups groups. gp[0] must hold the whole batch picked (pick_group). Returns the number of groups, 0 when the batch
// is wide everywhere.
inline int cluster_batch(double max_fx, const Cand* hc, int B, double max_px, int max_groups, GroupPick* gp) {
    if (max_groups < 2 || B < 2) return 0;
    max_groups = std::min(max_groups, kMaxPairGroups);
    int seeds[kMaxPairGroups]; seeds[0] = gp[0].ref;
    static thread_local double dist[kMaxPairGroups][kMaxChain];
    for (int b = 0; b < B; ++b) dist[0][b] = cand_px(max_fx, hc[b], hc[seeds[0]]);
    for (int ng = 1;;) {
        // the candidate farthest from its nearest seed becomes the next seed
        int far = -1; double fd = -1;
        for (int b = 0; b < B; ++b) { double d = INFINITY; for (int g = 0; g < ng; ++g) d = std::min(d, dist[g][b]); if (!(d <= fd)) { fd = d; far = b; } }
        if (far < 0 || !(fd < INFINITY) || ng >= max_groups) return 0;
        seeds[ng] = far;
        for (int b = 0; b < B; ++b) dist[ng][b] = cand_px(max_fx, hc[b], hc[far]);
        ++ng;
        for (int g = 0; g < ng; ++g) gp[g].n = 0;
        for (int b = 0; b < B; ++b) { int bg = 0; for (int g = 1; g < ng; ++g) if (dist[g][b] < dist[bg][b]) bg = g; gp[bg].idx[gp[bg].n++] = b; }
        bool ok = true;
        for (int g = 0; g < ng && ok; ++g) ok = gp[g].n > 0 && pick_group(max_fx, hc, gp[g]) && gp[g].px <= max_px;
        if (ok) return ng;
    }
}

}  // namespace iba
