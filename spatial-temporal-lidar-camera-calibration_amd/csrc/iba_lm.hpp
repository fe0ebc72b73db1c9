// Host-side caller of the Jacobian path: the outer re-association loop of iba_local (iba_local.cpp:434-460)
// with a Ceres-style Levenberg-Marquardt inner solve on the 7x7 normal equations the device returns.
// Ceres itself is a third-party dependency of the reference (absent here); this follows its published
// trust-region LM (LevenbergMarquardtStrategy + TrustRegionMinimizer): Jacobi column scaling,
// (H + diag(H)/radius) dx = -g, step acceptance by relative decrease, radius update
// radius /= max(1/3, 1 - (2 rho - 1)^3), and its three convergence tests in the minimizer's own order (gradient tolerance
// at the top of an iteration; parameter and function tolerance after the candidate's evaluation and before acceptance).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>

namespace iba {

struct LmOptions {
    int max_outer_iterations = 30;      // max_iba_iter
    int max_inner_iterations = 30;      // options.max_num_iterations (iba_local.cpp:437)
    double min_diff = 1e-6;             // iba_min_diff: allClose(last, cur) on the 7-vector (iba_local.cpp:454)
    double function_tolerance = 1e-6, gradient_tolerance = 1e-10, parameter_tolerance = 1e-8;   // Ceres defaults
    double initial_trust_region_radius = 1e4, max_trust_region_radius = 1e16, min_trust_region_radius = 1e-32;
    double min_relative_decrease = 1e-3, min_lm_diagonal = 1e-6, max_lm_diagonal = 1e32;
};
struct LmResult {
    double x[7];
    int outer_iterations = 0, inner_iterations = 0, evaluations = 0, converged = 0;
    double initial_cost = 0, final_cost = 0;
};

// solves A x = b for symmetric positive definite 7x7 A (row-major); returns false if not SPD
inline bool chol_solve7(const double* A, const double* b, double* x) {
    double L[49]; std::memset(L, 0, sizeof(L));
    for (int i = 0; i < 7; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = A[i * 7 + j];
            for (int k = 0; k < j; ++k) s -= L[i * 7 + k] * L[j * 7 + k];
            if (i == j) { if (!(s > 0)) return false; L[i * 7 + i] = std::sqrt(s); }
            else L[i * 7 + j] = s / L[j * 7 + j];
        }
    double y[7];
    for (int i = 0; i < 7; ++i) { double s = b[i]; for (int k = 0; k < i; ++k) s -= L[i * 7 + k] * y[k]; y[i] = s / L[i * 7 + i]; }
    for (int i = 6; i >= 0; --i) { double s = y[i]; for (int k = i + 1; k < 7; ++k) s -= L[k * 7 + i] * x[k]; x[i] = s / L[i * 7 + i]; }
    return true;
}

// Build(x): freeze the association at x. Eval(x, H, g, cost): residual blocks of the frozen association at x.
template <class Build, class Eval>
inline bool calibrate_lm(const double* x0, const LmOptions& o, Build build, Eval eval, LmResult& r) {
    double x[7]; std::memcpy(x, x0, sizeof(x));
    double last[7]; std::memcpy(last, x0, sizeof(last));
    r = LmResult();
    for (int outer = 0; outer < o.max_outer_iterations; ++outer) {
        if (!build(x)) return false;
        double H[49], g[7], cost;
        if (!eval(x, H, g, cost)) return false;
        ++r.evaluations;
        if (outer == 0) r.initial_cost = cost;
        double radius = o.initial_trust_region_radius, decrease_factor = 2.0;
        double scale[7];   // Jacobi scaling, fixed at the first Jacobian of the solve (Ceres)
        for (int i = 0; i < 7; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(std::max(H[i * 7 + i], 0.0)));
        for (int it = 0; it < o.max_inner_iterations; ++it) {
            ++r.inner_iterations;
            double gmax = 0; for (int i = 0; i < 7; ++i) gmax = std::max(gmax, std::fabs(g[i]));
            if (gmax <= o.gradient_tolerance) break;
            double Hs[49], gs[7];
            for (int i = 0; i < 7; ++i) { gs[i] = scale[i] * g[i]; for (int j = 0; j < 7; ++j) Hs[i * 7 + j] = scale[i] * H[i * 7 + j] * scale[j]; }
            double A[49]; std::memcpy(A, Hs, sizeof(A));
            for (int i = 0; i < 7; ++i) A[i * 7 + i] += std::min(std::max(Hs[i * 7 + i], o.min_lm_diagonal), o.max_lm_diagonal) / radius;
            double ngs[7], ds[7]; for (int i = 0; i < 7; ++i) ngs[i] = -gs[i];
            bool ok = chol_solve7(A, ngs, ds);
            double model = 0;
            if (ok) { for (int i = 0; i < 7; ++i) { double hd = 0; for (int j = 0; j < 7; ++j) hd += Hs[i * 7 + j] * ds[j]; model -= ds[i] * (gs[i] + 0.5 * hd); } }
            if (!ok || !(model > 0)) { radius = std::max(o.min_trust_region_radius, radius / decrease_factor); decrease_factor *= 2; if (radius <= o.min_trust_region_radius) break; continue; }
            double xn[7], step2 = 0, xn2 = 0;
            for (int i = 0; i < 7; ++i) { const double d = scale[i] * ds[i]; xn[i] = x[i] + d; step2 += d * d; xn2 += x[i] * x[i]; }
            // TrustRegionMinimizer::Minimize order: evaluate the candidate, then the parameter- and the function-tolerance
            // tests against the CURRENT point's cost — a run that stops on either keeps x (the candidate is not committed,
            // and the stop can come on a step that would have been rejected) — and only then accept or reject.
            double Hn[49], gn[7], cn;
            if (!eval(xn, Hn, gn, cn)) return false;
            ++r.evaluations;
            if (std::sqrt(step2) <= o.parameter_tolerance * (std::sqrt(xn2) + o.parameter_tolerance)) break;
            if (std::fabs(cost - cn) <= o.function_tolerance * cost) break;
            const double rho = (cost - cn) / model;
            if (rho > o.min_relative_decrease) {
                std::memcpy(x, xn, sizeof(x)); std::memcpy(H, Hn, sizeof(H)); std::memcpy(g, gn, sizeof(g));
                cost = cn;
                const double t = 2.0 * rho - 1.0;
                radius = std::min(o.max_trust_region_radius, radius / std::max(1.0 / 3.0, 1.0 - t * t * t)); decrease_factor = 2.0;
            } else {
                radius = std::max(o.min_trust_region_radius, radius / decrease_factor); decrease_factor *= 2;
                if (radius <= o.min_trust_region_radius) break;
            }
        }
        r.final_cost = cost; r.outer_iterations = outer + 1;
        bool close = true;   // allClose (IBACalib2.hpp:9-18)
        for (int i = 0; i < 7; ++i) if (std::fabs(last[i] - x[i]) > o.min_diff) close = false;
        if (close) { r.converged = 1; break; }
        std::memcpy(last, x, sizeof(last));
    }
    std::memcpy(r.x, x, sizeof(x));
    return true;
}

}  // namespace iba
