// Host-side (once per candidate) Lie-group math of the IBA path: Sim3Exp / SE3Exp and their
// derivatives with respect to x = [omega, upsilon, s], evaluated with forward-mode duals exactly as
// the reference's autodiff functors would (g2o_tools.h:105-140, 149-183; IBACalib2.hpp:152-160,
// 570-577). The expression order mirrors the reference so R, t are bit-identical to a CPU run.
#pragma once
#include <cmath>

#include "iba_types.hpp"

namespace iba {

template <int N>
struct Jet {
    double a; double v[N];
    Jet() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
    Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }
    static Jet seed(double s, int k) { Jet j(s); j.v[k] = 1.0; return j; }
};
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Jet<N> operator-(const Jet<N>& f) { Jet<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
    Jet<N> h; const double gi = 1.0 / g.a, fg = f.a * gi; h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi;
    return h;
}
template <int N> inline Jet<N> jsqrt(const Jet<N>& f) { Jet<N> h; const double t = std::sqrt(f.a), ti = 1.0 / (2.0 * t); h.a = t; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * ti; return h; }
template <int N> inline Jet<N> jcos(const Jet<N>& f) { Jet<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
template <int N> inline Jet<N> jsin(const Jet<N>& f) { Jet<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Jet<N> jpow(const Jet<N>& f, double p) { Jet<N> h; h.a = std::pow(f.a, p); const double t = p * std::pow(f.a, p - 1.0); for (int i = 0; i < N; ++i) h.v[i] = t * f.v[i]; return h; }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jcos(double x) { return std::cos(x); }
inline double jsin(double x) { return std::sin(x); }
inline double jpow(double x, double p) { return std::pow(x, p); }
inline double jval(double x) { return x; }
template <int N> inline double jval(const Jet<N>& x) { return x.a; }

// R (row-major 9), t (3) = exp of the first six entries of `u` (g2o_tools.h:105-140 / 149-183)
template <class T>
inline void se3_exp(const T* u, T* R, T* t) {
    const T wx = u[0], wy = u[1], wz = u[2];
    const T theta = jsqrt(wx * wx + wy * wy + wz * wz);
    T Om[9] = {T(0.0), -wz, wy, wz, T(0.0), -wx, -wy, wx, T(0.0)};
    T Om2[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Om2[r * 3 + c] = Om[r * 3 + 0] * Om[0 * 3 + c] + Om[r * 3 + 1] * Om[1 * 3 + c] + Om[r * 3 + 2] * Om[2 * 3 + c];
    T V[9];
    if (jval(theta) < 1e-4) {
        for (int i = 0; i < 9; ++i) {
            const T I = T((i % 4 == 0) ? 1.0 : 0.0);
            R[i] = (I + Om[i]) + T(0.5) * Om2[i];
            V[i] = (I + T(0.5) * Om[i]) + (T(1.) / T(6.)) * Om2[i];
        }
    } else {
        const T costh = jcos(theta), sinth = jsin(theta);
        const T invth2 = jpow(theta, -2.0), invth3 = jpow(theta, -3.0);
        const T ka = sinth / theta, kb = (T(1.) - costh) * invth2, kc = (theta - sinth) * invth3;
        for (int i = 0; i < 9; ++i) {
            const T I = T((i % 4 == 0) ? 1.0 : 0.0);
            R[i] = (I + ka * Om[i]) + kb * Om2[i];
            V[i] = (I + kb * Om[i]) + kc * Om2[i];
        }
    }
    for (int r = 0; r < 3; ++r) t[r] = V[r * 3 + 0] * u[3] + V[r * 3 + 1] * u[4] + V[r * 3 + 2] * u[5];
}

// the candidate's VALUES: Sim3Exp(x) and its inverse — what the association, search and sum kernels read (bit-identical to the
// reference's CPU evaluation: same libm, same expression order)
inline void make_cand_values(const double* x, Cand& c) {
    se3_exp<double>(x, c.R, c.t);
    c.s = x[6];
    c.s32 = (float)x[6];
    c.pad0 = 0.f;
    // Eigen::Transform<Isometry>::inverse(): linear^T, -(linear^T * translation)
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) c.Ri[r * 3 + k] = c.R[k * 3 + r];
    for (int r = 0; r < 3; ++r) c.ti[r] = -(c.Ri[r * 3 + 0] * c.t[0] + c.Ri[r * 3 + 1] * c.t[1] + c.Ri[r * 3 + 2] * c.t[2]);
}
// the candidate's DERIVATIVES (forward-mode duals of Sim3Exp(x) and of SE3Exp(-x[0:6])): what the factor kernel alone reads —
// five sixths of the host time of a candidate (two exponentials on Jet<6>). Computed while the GPU is already at work on the values.
inline void make_cand_jets(const double* x, Cand& c) {
    using J6 = Jet<6>;
    J6 xd[6], Rd[9], td[3];
    for (int k = 0; k < 6; ++k) xd[k] = J6::seed(x[k], k);
    se3_exp<J6>(xd, Rd, td);
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 9; ++i) c.dR[k][i] = Rd[i].v[k];
    for (int k = 0; k < 6; ++k)
        for (int i = 0; i < 3; ++i) c.dt[k][i] = td[i].v[k];
    // SE3Exp(-x[0:6]) and its derivative with respect to x (IBACalib2.hpp:573-577, 614-619)
    J6 nx[6];
    for (int k = 0; k < 6; ++k) nx[k] = -xd[k];
    se3_exp<J6>(nx, Rd, td);
    for (int i = 0; i < 9; ++i) c.Rlc[i] = Rd[i].a;
    for (int i = 0; i < 3; ++i) c.tlc[i] = td[i].a;
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 9; ++i) c.dRlc[k][i] = Rd[i].v[k];
    for (int k = 0; k < 6; ++k)
        for (int i = 0; i < 3; ++i) c.dtlc[k][i] = td[i].v[k];
}
inline void make_cand(const double* x, Cand& c) { make_cand_values(x, c); make_cand_jets(x, c); }

}  // namespace iba
