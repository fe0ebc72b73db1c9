// Hand-eye initialiser: produces the `init_sim3` the IBA stages start from, without g2o/Eigen.
// Host only, a few hundred motion pairs: nothing here belongs on the GPU.
//
// Restated (reference file:line):
//  * pose2Motion                           kitti_tools.h:160-165   T(i+1) * T(i)^-1
//  * HECalib (closed form)                 HECalib.h:12-57         rotation: Kabsch on the rotation vectors
//                                                                  (H = sum (beta - mean)(alpha - mean)^T, R = V U^T, det fix),
//                                                                  translation + scale: [Ra - I | ta] [t; s] = R tb, normal equations
//  * EdgeHE residual                       NLHECalib.hpp:27-48     e = w (R beta - alpha + (Ra - I) t + s ta - R tb), x = [omega, upsilon, s],
//                                                                  (R, t, s) = Sim3Exp(x)
//    (its hand-written Jacobian, :50-67, is [ (R beta)^ | Ra - I | ta ]: the rotation block has the opposite sign of
//    d(R beta)/d omega and V(omega), d(R tb)/d omega are ignored. A faithful copy of it makes every damped step go uphill
//    in rotation and the solve stalls after one or two iterations (measured). What is restated here is the COST the
//    reference minimises; the Jacobian is taken numerically from the residual, which is exact to 1e-9 and costs nothing
//    for a few hundred 3-vectors.)
//  * HECalibRobustKernelg2o                NLHECalib.hpp:121-163   Huber(delta) on every pair, optional regulariser e = upsilon with
//                                                                  information n * ratio, 10 iterations
//  * HECalibLineProcessg2o                 NLHECalib.hpp:189-277   no kernel; per-pair information w^2, w = mu / (mu + chi2), re-estimated
//                                                                  between solves while mu anneals (64 / 1.4 per round down to 0.1)
// Third party that is absent: Eigen (AngleAxis::fromRotationMatrix goes through a quaternion; JacobiSVD) and g2o
// (Dogleg + RobustKernelHuber). The SVD is a Jacobi eigen-decomposition of H^T H; the optimiser is Levenberg-Marquardt
// with g2o's Huber weighting (rho'(e) applied to JtJ and Jtr). Same cost, not the same iterates: PARITY WITH g2o IS
// UNPINNED, the tests pin it to an independent numpy/scipy restatement and to planted extrinsics instead.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/iba_mi355x.h"

namespace {

struct Iso { double R[9], t[3]; };
inline Iso load(const double* p12) {
    Iso T;
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) T.R[r * 3 + c] = p12[r * 4 + c]; T.t[r] = p12[r * 4 + 3]; }
    return T;
}
inline void store(const Iso& T, double* p12) {
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) p12[r * 4 + c] = T.R[r * 3 + c]; p12[r * 4 + 3] = T.t[r]; }
}
inline void mat3_mul(const double* A, const double* B, double* C) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C[r * 3 + c] = (A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c]) + A[r * 3 + 2] * B[6 + c];
}
inline void mat3_vec(const double* A, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = (A[r * 3] * v[0] + A[r * 3 + 1] * v[1]) + A[r * 3 + 2] * v[2];
}
inline double det3(const double* M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// Eigen::AngleAxisd::fromRotationMatrix = Quaternion(R) -> angle-axis; returned as angle * axis
void rotvec_of(const double* R, double* out) {
    double q[4];   // x, y, z, w
    const double tr = R[0] + R[4] + R[8];
    if (tr > 0.0) {
        double s = std::sqrt(tr + 1.0); q[3] = 0.5 * s; s = 0.5 / s;
        q[0] = (R[7] - R[5]) * s; q[1] = (R[2] - R[6]) * s; q[2] = (R[3] - R[1]) * s;
    } else {
        int i = 0; if (R[4] > R[0]) i = 1; if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        double s = std::sqrt(R[i * 4] - R[j * 4] - R[k * 4] + 1.0);
        q[i] = 0.5 * s; s = 0.5 / s;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * s; q[j] = (R[j * 3 + i] + R[i * 3 + j]) * s; q[k] = (R[k * 3 + i] + R[i * 3 + k]) * s;
    }
    // AngleAxis = Quaternion (Eigen/src/Geometry/AngleAxis.h): n = |vec|; if n < eps use the squared norm; angle = 2 atan2(n, |w|), axis = vec / n (sign of w folded in)
    double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
    if (n < 2.220446049250313e-16) n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    if (n != 0.0) {
        double w = q[3];
        if (w < 0) { n = -n; w = -w; }
        const double angle = 2.0 * std::atan2(std::fabs(n), w);
        for (int i = 0; i < 3; ++i) out[i] = angle * (q[i] / n);
    } else { out[0] = out[1] = out[2] = 0.0; }   // angle 0, axis (1,0,0)
}

// eigen-decomposition of a symmetric 3x3 (cyclic Jacobi); columns of V, eigenvalues sorted descending
void sym_eig3(const double* S, double* V, double* lam) {
    double A[9]; std::memcpy(A, S, sizeof(A));
    for (int i = 0; i < 9; ++i) V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = A[1] * A[1] + A[2] * A[2] + A[5] * A[5];
        if (off < 1e-300) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                const double apq = A[p * 3 + q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) { const double akp = A[k * 3 + p], akq = A[k * 3 + q]; A[k * 3 + p] = c * akp - s * akq; A[k * 3 + q] = s * akp + c * akq; }
                for (int k = 0; k < 3; ++k) { const double apk = A[p * 3 + k], aqk = A[q * 3 + k]; A[p * 3 + k] = c * apk - s * aqk; A[q * 3 + k] = s * apk + c * aqk; }
                for (int k = 0; k < 3; ++k) { const double vkp = V[k * 3 + p], vkq = V[k * 3 + q]; V[k * 3 + p] = c * vkp - s * vkq; V[k * 3 + q] = s * vkp + c * vkq; }
            }
    }
    int idx[3] = {0, 1, 2};
    std::sort(idx, idx + 3, [&](int a, int b) { return A[a * 4] > A[b * 4]; });
    double Vs[9];
    for (int c = 0; c < 3; ++c) { lam[c] = A[idx[c] * 4]; for (int r = 0; r < 3; ++r) Vs[r * 3 + c] = V[r * 3 + idx[c]]; }
    std::memcpy(V, Vs, sizeof(Vs));
}

inline void cross(const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
inline double norm3(const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }

// full SVD H = U diag(s) V^T of a 3x3 (columns of U, V)
void svd3(const double* H, double* U, double* V) {
    double HtH[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) HtH[r * 3 + c] = H[0 * 3 + r] * H[0 * 3 + c] + H[1 * 3 + r] * H[1 * 3 + c] + H[2 * 3 + r] * H[2 * 3 + c];
    double lam[3];
    sym_eig3(HtH, V, lam);
    double u[3][3];
    const double smax = std::sqrt(std::max(lam[0], 0.0));
    int good = 0;
    for (int c = 0; c < 3; ++c) {
        const double v[3] = {V[0 * 3 + c], V[1 * 3 + c], V[2 * 3 + c]};
        double hv[3]; mat3_vec(H, v, hv);
        const double s = norm3(hv);
        if (s > 1e-12 * std::max(smax, 1e-300) && good == c) { for (int r = 0; r < 3; ++r) u[c][r] = hv[r] / s; ++good; }
        else break;
    }
    if (good == 0) { u[0][0] = 1; u[0][1] = 0; u[0][2] = 0; good = 1; }
    if (good == 1) {   // any unit vector orthogonal to u0
        const double* a = u[0];
        double e[3] = {0, 0, 0}; e[std::fabs(a[0]) < 0.9 ? 0 : 1] = 1.0;
        cross(a, e, u[1]); const double n = norm3(u[1]); for (int r = 0; r < 3; ++r) u[1][r] /= n;
        good = 2;
    }
    if (good == 2) { cross(u[0], u[1], u[2]); const double n = norm3(u[2]); for (int r = 0; r < 3; ++r) u[2][r] /= n; }
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) U[r * 3 + c] = u[c][r];
}

// solves the symmetric positive (semi)definite n x n system in place (Gaussian elimination with partial pivoting)
bool solve_n(int n, double* A, double* b, double* x) {
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[r * n + c]) > std::fabs(A[p * n + c])) p = r;
        if (!(std::fabs(A[p * n + c]) > 0)) return false;
        if (p != c) { for (int k = 0; k < n; ++k) std::swap(A[p * n + k], A[c * n + k]); std::swap(b[p], b[c]); }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[r * n + c] / A[c * n + c];
            for (int k = c; k < n; ++k) A[r * n + k] -= f * A[c * n + k];
            b[r] -= f * b[c];
        }
    }
    for (int r = n - 1; r >= 0; --r) { double s = b[r]; for (int k = r + 1; k < n; ++k) s -= A[r * n + k] * x[k]; x[r] = s / A[r * n + r]; }
    return true;
}

// Sim3Exp (g2o_tools.h:105-140): R = exp(omega^), t = V(omega) upsilon, s = x[6]
void sim3_exp(const double* x, double* R, double* t, double* s) {
    const double wx = x[0], wy = x[1], wz = x[2];
    const double theta = std::sqrt(wx * wx + wy * wy + wz * wz);
    const double Om[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double Om2[9]; mat3_mul(Om, Om, Om2);
    double V[9];
    double a, b, c;
    if (theta < 1e-4) { a = 1.0; b = 0.5; c = 1.0 / 6.0; }   // Taylor branch (:119-124)
    else { a = std::sin(theta) / theta; b = (1 - std::cos(theta)) / (theta * theta); c = (theta - std::sin(theta)) / (theta * theta * theta); }
    for (int i = 0; i < 9; ++i) { const double I = (i % 4 == 0) ? 1.0 : 0.0; R[i] = I + a * Om[i] + b * Om2[i]; V[i] = I + b * Om[i] + c * Om2[i]; }
    mat3_vec(V, x + 3, t);
    *s = x[6];
}

}  // namespace

namespace {

// The nonlinear hand-eye problem on the 7-vector [rotation vector, translation (as upsilon), scale] shared by the
// Huber-robust (NLHECalib.hpp:119-187) and the line-process (NLHECalib.hpp:189-277) refinements: EdgeHE residuals
// (NLHECalib.hpp:31-49), the optional EdgeRegulation on the translation, and a Levenberg-Marquardt loop with g2o's
// initial damping. The Jacobian is numerical: the reference's hand-written one (NLHECalib.hpp:51-66) differentiates
// with respect to an increment that its oplus does not apply, so g2o's solver stalls on it; central differences of the
// same residual reach the same minimum the reference is after.
struct HeProblem {
    const double* Ta12; const double* Tb12; int64_t n;
    std::vector<double> alpha, beta;
    HeProblem(const double* A, const double* B, int64_t n_) : Ta12(A), Tb12(B), n(n_), alpha(3 * (size_t)n_), beta(3 * (size_t)n_) {
        for (int64_t i = 0; i < n; ++i) { rotvec_of(load(Ta12 + 12 * i).R, &alpha[3 * (size_t)i]); rotvec_of(load(Tb12 + 12 * i).R, &beta[3 * (size_t)i]); }
    }
    // initial vertex value (NLHECalib.hpp:131-137): rotation vector, translation AS upsilon, scale
    static void init_vertex(const double rigid12_init[12], double scale_init, double x[7]) {
        const double R0[9] = {rigid12_init[0], rigid12_init[1], rigid12_init[2], rigid12_init[4], rigid12_init[5], rigid12_init[6], rigid12_init[8], rigid12_init[9], rigid12_init[10]};
        rotvec_of(R0, x);
        x[3] = rigid12_init[3]; x[4] = rigid12_init[7]; x[5] = rigid12_init[11]; x[6] = scale_init;
    }
    static void store(const double x[7], double rigid12[12], double* scale) {
        double R[9], t[3], s;
        sim3_exp(x, R, t, &s);
        for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) rigid12[r * 4 + c] = R[r * 3 + c]; rigid12[r * 4 + 3] = t[r]; }
        *scale = s;
    }
    // residuals of all pairs at x (3 per pair)
    void residuals(const double* xx, std::vector<double>& e) const {
        double R[9], t[3], s;
        sim3_exp(xx, R, t, &s);
        e.resize(3 * (size_t)n);
        for (int64_t i = 0; i < n; ++i) {
            const Iso A = load(Ta12 + 12 * i), B = load(Tb12 + 12 * i);
            double Rb[3], Rtb[3];
            mat3_vec(R, &beta[3 * (size_t)i], Rb); mat3_vec(R, B.t, Rtb);
            for (int r = 0; r < 3; ++r) {
                const double tr = ((A.R[r * 3] - (r == 0)) * t[0] + (A.R[r * 3 + 1] - (r == 1)) * t[1]) + (A.R[r * 3 + 2] - (r == 2)) * t[2];
                e[3 * (size_t)i + r] = (Rb[r] - alpha[3 * (size_t)i + r]) + ((tr + A.t[r] * s) - Rtb[r]);
            }
        }
    }
    // cost and (optionally) the normal equations at x. huber > 0: g2o RobustKernelHuber of that delta on every pair;
    // info: per-pair scalar information (nullptr = 1).
    double evaluate(const double* xx, double huber, const double* info, bool regulation, double reg_info, double* Hm, double* g) const {
        std::vector<double> e;
        residuals(xx, e);
        std::vector<double> J;   // (3n) x 7, central differences
        if (Hm) {
            J.resize(21 * (size_t)n);
            std::vector<double> ep, em;
            for (int k = 0; k < 7; ++k) {
                double xp[7], xm[7];
                std::memcpy(xp, xx, sizeof(xp)); std::memcpy(xm, xx, sizeof(xm));
                const double hstep = 1e-6 * std::max(1.0, std::fabs(xx[k]));
                xp[k] += hstep; xm[k] -= hstep;
                residuals(xp, ep); residuals(xm, em);
                for (size_t r = 0; r < 3 * (size_t)n; ++r) J[r * 7 + k] = (ep[r] - em[r]) / (2 * hstep);
            }
            std::memset(Hm, 0, 49 * sizeof(double)); std::memset(g, 0, 7 * sizeof(double));
        }
        double cost = 0;
        const double d = huber;
        for (int64_t i = 0; i < n; ++i) {
            const double* ei = &e[3 * (size_t)i];
            const double om = info ? info[(size_t)i] : 1.0;
            const double chi2 = om * (ei[0] * ei[0] + ei[1] * ei[1] + ei[2] * ei[2]);
            double rho = chi2, w = 1.0;   // Huber on e2 = chi2: rho = e2 (e <= d) else 2 d sqrt(e2) - d^2; weight rho'
            if (d > 0 && chi2 > d * d) { const double se = std::sqrt(chi2); rho = 2 * d * se - d * d; w = d / se; }
            cost += rho;
            if (Hm) {
                const double* Ji = &J[21 * (size_t)i];
                const double wo = w * om;
                for (int p = 0; p < 7; ++p) {
                    for (int r = 0; r < 3; ++r) g[p] += wo * Ji[r * 7 + p] * ei[r];
                    for (int q = 0; q < 7; ++q) Hm[p * 7 + q] += wo * ((Ji[p] * Ji[q] + Ji[7 + p] * Ji[7 + q]) + Ji[14 + p] * Ji[14 + q]);
                }
            }
        }
        if (regulation) {   // EdgeRegulation: e = x[3:6] under information reg_info (NLHECalib.hpp:148-155)
            for (int k = 0; k < 3; ++k) { cost += reg_info * xx[3 + k] * xx[3 + k]; if (Hm) { Hm[(3 + k) * 8] += reg_info; g[3 + k] += reg_info * xx[3 + k]; } }
        }
        return cost;
    }
    void lm(double x[7], double huber, const double* info, bool regulation, double reg_info, int iterations) const {
        double lambda = -1.0;
        double Hm[49], g[7];
        double cost = evaluate(x, huber, info, regulation, reg_info, Hm, g);
        for (int it = 0; it < std::max(iterations, 1); ++it) {
            if (lambda < 0) { double mx = 0; for (int k = 0; k < 7; ++k) mx = std::max(mx, Hm[k * 8]); lambda = 1e-5 * mx; }   // g2o LM's initial damping
            bool stepped = false;
            for (int tries = 0; tries < 10 && !stepped; ++tries) {
                double A[49], b[7], dx[7];
                std::memcpy(A, Hm, sizeof(A));
                for (int k = 0; k < 7; ++k) { A[k * 8] += lambda; b[k] = -g[k]; }
                if (!solve_n(7, A, b, dx)) { lambda *= 10; continue; }
                double xn[7];
                for (int k = 0; k < 7; ++k) xn[k] = x[k] + dx[k];
                const double cn = evaluate(xn, huber, info, regulation, reg_info, nullptr, nullptr);
                if (cn < cost) { std::memcpy(x, xn, sizeof(double) * 7); lambda = std::max(lambda / 3.0, 1e-12); stepped = true; }
                else lambda *= 4.0;
            }
            if (!stepped) break;
            const double prev = cost;
            cost = evaluate(x, huber, info, regulation, reg_info, Hm, g);
            if (prev - cost <= 1e-14 * std::max(prev, 1e-300)) break;
        }
    }
};

}  // namespace

extern "C" {

iba_status iba_pose_to_motion(const double* poses12, int64_t n, double* motions12) {   // kitti_tools.h:160-165
    if (!poses12 || !motions12 || n < 2) return IBA_ERR_INVALID_ARG;
    for (int64_t i = 0; i + 1 < n; ++i) {
        const Iso A = load(poses12 + 12 * (i + 1)), B = load(poses12 + 12 * i);
        Iso Bi, M;   // Eigen's Isometry inverse: R^T, -R^T t
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Bi.R[r * 3 + c] = B.R[c * 3 + r];
        mat3_vec(Bi.R, B.t, Bi.t); for (int r = 0; r < 3; ++r) Bi.t[r] = -Bi.t[r];
        mat3_mul(A.R, Bi.R, M.R);
        mat3_vec(A.R, Bi.t, M.t); for (int r = 0; r < 3; ++r) M.t[r] += A.t[r];
        store(M, motions12 + 12 * i);
    }
    return IBA_OK;
}

// the rotation part RAB both initialisers share (HECalib.h:16-45 = :73-107): rotation vectors of the paired motions, decentred covariance, SVD, determinant fix;
// alpha_norm (may be null): |alpha_i| per pair, what DGHECalib's degeneracy test reads (:82)
static void handeye_rotation(const double* Ta12, const double* Tb12, int64_t n, double R[9], double* alpha_norm) {
    std::vector<double> alpha(3 * (size_t)n), beta(3 * (size_t)n);
    double am[3] = {0, 0, 0}, bm[3] = {0, 0, 0};
    for (int64_t i = 0; i < n; ++i) {
        const Iso A = load(Ta12 + 12 * i), B = load(Tb12 + 12 * i);
        rotvec_of(A.R, &alpha[3 * (size_t)i]); rotvec_of(B.R, &beta[3 * (size_t)i]);
        for (int k = 0; k < 3; ++k) { am[k] += alpha[3 * (size_t)i + k]; bm[k] += beta[3 * (size_t)i + k]; }
        if (alpha_norm) { const double* a = &alpha[3 * (size_t)i]; alpha_norm[i] = std::sqrt((a[0] * a[0] + a[1] * a[1]) + a[2] * a[2]); }
    }
    for (int k = 0; k < 3; ++k) { am[k] /= (double)n; bm[k] /= (double)n; }
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t i = 0; i < n; ++i)
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) H[r * 3 + c] += (beta[3 * (size_t)i + r] - bm[r]) * (alpha[3 * (size_t)i + c] - am[c]);
    double U[9], V[9];
    svd3(H, U, V);
    double Ut[9];
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Ut[r * 3 + c] = U[c * 3 + r];
    mat3_mul(V, Ut, R);   // RAB = Vt^T * Ut
    if (det3(R) < 0) { for (int r = 0; r < 3; ++r) V[r * 3 + 2] = -V[r * 3 + 2]; mat3_mul(V, Ut, R); }   // Vt.row(2) *= -1
}

// DGHECalib (HECalib.h:66-120), the initialiser for degenerate motion (a vehicle that hardly turns): the rotation as HECalib's, NO translation
// (tAB = 0, :109), the scale from the pairs whose camera rotation is below dg_threshold (|alpha| < dg_threshold, :82): sum |ta| |tb| / sum |ta|^2 (:112-119).
// n_degenerate (may be null): how many pairs that were (the reference prints it, :118). No such pair: the quotient is 0 / 0 = NaN, as in the reference.
iba_status iba_handeye_degenerate(const double* Ta12, const double* Tb12, int64_t n, double dg_threshold, double rigid12[12], double* scale, int64_t* n_degenerate) {
    if (!Ta12 || !Tb12 || !rigid12 || !scale || n < 1) return IBA_ERR_INVALID_ARG;
    double R[9];
    std::vector<double> an((size_t)n);
    handeye_rotation(Ta12, Tb12, n, R, an.data());
    double num = 0, den = 0;
    int64_t nd = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (!(an[(size_t)i] < dg_threshold)) continue;
        const Iso A = load(Ta12 + 12 * i), B = load(Tb12 + 12 * i);
        const double ta = std::sqrt((A.t[0] * A.t[0] + A.t[1] * A.t[1]) + A.t[2] * A.t[2]), tb = std::sqrt((B.t[0] * B.t[0] + B.t[1] * B.t[1]) + B.t[2] * B.t[2]);
        num += ta * tb; den += ta * ta; ++nd;
    }
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) rigid12[r * 4 + c] = R[r * 3 + c]; rigid12[r * 4 + 3] = 0.0; }
    *scale = num / den;
    if (n_degenerate) *n_degenerate = nd;
    return IBA_OK;
}

iba_status iba_handeye(const double* Ta12, const double* Tb12, int64_t n, double rigid12[12], double* scale) {   // HECalib.h:12-57
    if (!Ta12 || !Tb12 || !rigid12 || !scale || n < 2) return IBA_ERR_INVALID_ARG;
    double R[9];
    handeye_rotation(Ta12, Tb12, n, R, nullptr);
    // [Ra - I | ta] [t; s] = R tb, normal equations (:46-52)
    double AtA[16] = {0}, Atb[4] = {0};
    for (int64_t i = 0; i < n; ++i) {
        const Iso A = load(Ta12 + 12 * i), B = load(Tb12 + 12 * i);
        double rhs[3]; mat3_vec(R, B.t, rhs);
        for (int r = 0; r < 3; ++r) {
            const double row[4] = {A.R[r * 3] - (r == 0), A.R[r * 3 + 1] - (r == 1), A.R[r * 3 + 2] - (r == 2), A.t[r]};
            for (int p = 0; p < 4; ++p) { Atb[p] += row[p] * rhs[r]; for (int q = 0; q < 4; ++q) AtA[p * 4 + q] += row[p] * row[q]; }
        }
    }
    double x[4];
    if (!solve_n(4, AtA, Atb, x)) return IBA_ERR_UNSUPPORTED;   // degenerate motion (no rotation about two axes): iba_handeye_degenerate is the reference's initialiser for it
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) rigid12[r * 4 + c] = R[r * 3 + c]; rigid12[r * 4 + 3] = x[r]; }
    *scale = x[3];
    return IBA_OK;
}

iba_status iba_handeye_robust(const double* Ta12, const double* Tb12, int64_t n, const double rigid12_init[12], double scale_init,
                              double robust_kernel_size, int32_t regulation, double regulation_ratio, int32_t iterations,
                              double rigid12[12], double* scale) {
    if (!Ta12 || !Tb12 || !rigid12_init || !rigid12 || !scale || n < 2 || !(robust_kernel_size > 0)) return IBA_ERR_INVALID_ARG;
    HeProblem hp(Ta12, Tb12, n);
    double x[7];
    hp.init_vertex(rigid12_init, scale_init, x);
    const double reg_info = regulation ? (double)n * regulation_ratio : 0.0;
    hp.lm(x, robust_kernel_size, nullptr, regulation != 0, reg_info, iterations);
    hp.store(x, rigid12, scale);
    return IBA_OK;
}

iba_status iba_handeye_lineprocess(const double* Ta12, const double* Tb12, int64_t n, const double rigid12_init[12], double scale_init,
                                   int32_t inner_iterations, double mu0, double divid_factor, double min_mu, int32_t ex_max_iter,
                                   int32_t regulation, double regulation_ratio, double rigid12[12], double* scale) {
    if (!Ta12 || !Tb12 || !rigid12_init || !rigid12 || !scale || n < 2 || !(mu0 > 0) || !(divid_factor > 1) || ex_max_iter < 0)
        return IBA_ERR_INVALID_ARG;
    HeProblem hp(Ta12, Tb12, n);
    double x[7];
    hp.init_vertex(rigid12_init, scale_init, x);
    std::vector<double> info((size_t)n, 1.0);   // every edge starts at identity information (NLHECalib.hpp:213-218)
    double reg_info = regulation ? (double)n * regulation_ratio : 0.0;
    hp.lm(x, 0.0, info.data(), regulation != 0, reg_info, inner_iterations);
    double mu = mu0;
    for (int ex = 0; ex < ex_max_iter; ++ex) {   // NLHECalib.hpp:229-248
        std::vector<double> e;
        hp.residuals(x, e);
        double total = 0;
        for (int64_t i = 0; i < n; ++i) {
            const double* ei = &e[3 * (size_t)i];
            // edge->chi2() is taken under the information of the PREVIOUS outer iteration (the reference never resets it)
            const double e2 = info[(size_t)i] * (ei[0] * ei[0] + ei[1] * ei[1] + ei[2] * ei[2]);
            const double w = mu / (mu + e2);
            info[(size_t)i] = w * w;
            total += w * w;
        }
        if (regulation) reg_info = total * regulation_ratio;
        hp.lm(x, 0.0, info.data(), regulation != 0, reg_info, inner_iterations);
        mu /= divid_factor;
        if (mu < min_mu) break;
    }
    hp.store(x, rigid12, scale);
    return IBA_OK;
}

}  // extern "C"
