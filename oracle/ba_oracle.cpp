// TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the ORB-only extrinsic BA edge of the reference:
// calibEdge::operator() (Optimizer.cc:65-205) evaluated with forward-mode duals, which is what
// G2O_MAKE_AUTO_AD_FUNCTIONS does with ceres::Jet (third party, absent). Robust weighting follows g2o's
// BaseUnaryEdge::constructQuadraticForm with RobustKernelHuber: rho' scales Omega; H += J^T (rho' Omega) J,
// b -= J^T (rho' Omega) e. PARITY UNPINNED against g2o itself (absent); pinned by finite differences and by the planted
// scenes in tests/.
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/iba_mi355x.h"
#include "oracle_math.hpp"

namespace {
using oracle::Dual;
typedef Dual<7> J7;

template <class T> struct Vec3 { T v[3]; };
template <class T> inline Vec3<T> cross(const Vec3<T>& a, const Vec3<T>& b) { return {{a.v[1] * b.v[2] - a.v[2] * b.v[1], a.v[2] * b.v[0] - a.v[0] * b.v[2], a.v[0] * b.v[1] - a.v[1] * b.v[0]}}; }
template <class T> inline T dot(const Vec3<T>& a, const Vec3<T>& b) { return a.v[0] * b.v[0] + a.v[1] * b.v[1] + a.v[2] * b.v[2]; }
template <class T> inline T norm(const Vec3<T>& a) { using oracle::sqrt; using std::sqrt; return sqrt(dot(a, a)); }

// the angle-axis rotation block that appears four times in calibEdge::operator()
template <class T>
inline Vec3<T> rotate(const Vec3<T>& w, const Vec3<T>& p) {
    using oracle::cos; using oracle::sin; using std::cos; using std::sin;
    const T theta = norm(w);
    if (oracle::scalar_of(theta) > 0.0) {
        const Vec3<T> v = {{w.v[0] / theta, w.v[1] / theta, w.v[2] / theta}};
        const T cth = cos(theta), sth = sin(theta);
        const Vec3<T> vxp = cross(v, p);
        const T vdp = dot(v, p);
        const T omc = T(1.0) - cth;
        return {{p.v[0] * cth + vxp.v[0] * sth + v.v[0] * vdp * omc, p.v[1] * cth + vxp.v[1] * sth + v.v[1] * vdp * omc, p.v[2] * cth + vxp.v[2] * sth + v.v[2] * vdp * omc}};
    }
    const Vec3<T> wxp = cross(w, p);
    return {{p.v[0] + wxp.v[0], p.v[1] + wxp.v[1], p.v[2] + wxp.v[2]}};
}

template <class T>
void calib_edge(const T* calib, const double* Xw, const double* Tlw6, const double* intr, const double* obs, T* err) {
    const T scale = calib[6];
    const Vec3<T> Xc0 = {{scale * T(Xw[0]), scale * T(Xw[1]), scale * T(Xw[2])}};                 // :87
    const Vec3<T> wlc = {{-calib[0], -calib[1], -calib[2]}};                                      // :90-93
    const Vec3<T> tneg = {{-calib[3], -calib[4], -calib[5]}};
    const Vec3<T> tlc = rotate(wlc, tneg);                                                         // :95-112
    Vec3<T> Xl0 = rotate(wlc, Xc0);                                                                // :116-133
    for (int i = 0; i < 3; ++i) Xl0.v[i] = Xl0.v[i] + tlc.v[i];
    const Vec3<T> wlw = {{T(Tlw6[0]), T(Tlw6[1]), T(Tlw6[2])}};
    Vec3<T> Xli = rotate(wlw, Xl0);                                                                // :136-155
    for (int i = 0; i < 3; ++i) Xli.v[i] = Xli.v[i] + T(Tlw6[3 + i]);
    const Vec3<T> wcl = {{calib[0], calib[1], calib[2]}};
    Vec3<T> Xci = rotate(wcl, Xli);                                                                // :158-177
    for (int i = 0; i < 3; ++i) Xci.v[i] = Xci.v[i] + calib[3 + i];
    const T pu = T(intr[0]) * Xci.v[0] / Xci.v[2] + T(intr[2]);                                    // :188-189
    const T pv = T(intr[1]) * Xci.v[1] / Xci.v[2] + T(intr[3]);
    err[0] = T(obs[0]) - pu;                                                                       // :192
    err[1] = T(obs[1]) - pv;
}
}  // namespace

extern "C" {

// residual (2) and Jacobian (2x7, row-major) of one edge
void oracle_ba_edge(const double* x, const double* Xw, const double* Tlw6, const double* intr, const double* obs, double* e, double* J) {
    J7 c[7], err[2];
    for (int i = 0; i < 7; ++i) c[i] = J7(x[i], i);
    calib_edge<J7>(c, Xw, Tlw6, intr, obs, err);
    for (int r = 0; r < 2; ++r) { e[r] = err[r].a; for (int k = 0; k < 7; ++k) J[r * 7 + k] = err[r].v[k]; }
}

// same contract as iba_ba_eval
void oracle_ba_eval(const iba_ba_desc* d, const double* x, const uint8_t* active, int robust, double huber_delta, double* H, double* b,
                    double* chi2_robust, double* chi2_edges) {
    std::memset(H, 0, 49 * sizeof(double)); std::memset(b, 0, 7 * sizeof(double));
    double sum = 0;
    for (int64_t i = 0; i < d->n_edges; ++i) {
        const int f = d->edge_frame[i];
        double e[2], J[14];
        oracle_ba_edge(x, d->edge_Xw + 3 * i, d->frame_Tlw6 + 6 * f, d->frame_intr + 4 * f, d->edge_obs + 2 * i, e, J);
        const double info = d->edge_info[i];
        const double chi2 = info * (e[0] * e[0] + e[1] * e[1]);
        if (chi2_edges) chi2_edges[i] = chi2;
        if (active && !active[i]) continue;
        double rho0 = chi2, rho1 = 1.0;
        if (robust) {
            const double dsqr = huber_delta * huber_delta;
            if (chi2 > dsqr) { const double sq = std::sqrt(chi2); rho0 = 2 * sq * huber_delta - dsqr; rho1 = huber_delta / sq; }
        }
        sum += rho0;
        const double w = rho1 * info;
        for (int p = 0; p < 7; ++p) {
            b[p] -= w * (J[p] * e[0] + J[7 + p] * e[1]);
            for (int q = 0; q < 7; ++q) H[p * 7 + q] += w * (J[p] * J[q] + J[7 + p] * J[7 + q]);
        }
    }
    *chi2_robust = sum;
}

}  // extern "C"
