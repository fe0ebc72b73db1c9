// TEST INFRASTRUCTURE ONLY — CPU restatement (oracle) of the reference's math helpers.
// Nothing under oracle/ may be linked, imported or executed by the product path.
// PARITY PIN STATUS: the reference has no tests/golden vectors for this path.
// Pins used instead: (1) kd-tree/kNN semantics checked against the reference's vendored nanoflann
// compiled from /root/reference/include (oracle/_ref, fixtures in tests/golden/), (2) known-answer
// tests against scipy/numpy for the closed-form math. Everything that depends on absent third-party
// code (Eigen product order, g2o::SE3Quat::log, Ceres Jet/Huber, OpenCV CV_32F products) is a
// restatement of the published algorithm => those parts are "parity unpinned".
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

namespace oracle {

// ---------------------------------------------------------------------------------------------
// Forward-mode dual numbers mirroring ceres::Jet<double,N> arithmetic (third-party, absent from
// /root/reference; call sites IBACalib2.hpp:207-211, 593-596, 636-639). Formulas follow the
// published ceres/jet.h definitions.
// ---------------------------------------------------------------------------------------------
template <int N>
struct Dual {
    double a;
    double v[N];
    Dual() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
    Dual(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT implicit on purpose
    Dual(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1.0; }
};
template <int N> inline Dual<N> operator+(const Dual<N>& f, const Dual<N>& g) { Dual<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline Dual<N> operator-(const Dual<N>& f, const Dual<N>& g) { Dual<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline Dual<N> operator-(const Dual<N>& f) { Dual<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline Dual<N> operator*(const Dual<N>& f, const Dual<N>& g) { Dual<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline Dual<N> operator/(const Dual<N>& f, const Dual<N>& g) {
    Dual<N> h; const double gi = 1.0 / g.a; const double fg = f.a * gi; h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi;
    return h; }
template <int N> inline Dual<N>& operator+=(Dual<N>& f, const Dual<N>& g) { f = f + g; return f; }
template <int N> inline Dual<N>& operator*=(Dual<N>& f, const Dual<N>& g) { f = f * g; return f; }
template <int N> inline bool operator<(const Dual<N>& f, const Dual<N>& g) { return f.a < g.a; }
template <int N> inline Dual<N> sqrt(const Dual<N>& f) { Dual<N> h; const double t = std::sqrt(f.a); const double ti = 1.0 / (2.0 * t); h.a = t; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * ti; return h; }
template <int N> inline Dual<N> cos(const Dual<N>& f) { Dual<N> h; h.a = std::cos(f.a); const double s = -std::sin(f.a); for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
template <int N> inline Dual<N> sin(const Dual<N>& f) { Dual<N> h; h.a = std::sin(f.a); const double c = std::cos(f.a); for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline Dual<N> pow(const Dual<N>& f, double p) { Dual<N> h; h.a = std::pow(f.a, p); const double t = p * std::pow(f.a, p - 1.0); for (int i = 0; i < N; ++i) h.v[i] = t * f.v[i]; return h; }
inline double sqrt(double x) { return std::sqrt(x); }
inline double cos(double x) { return std::cos(x); }
inline double sin(double x) { return std::sin(x); }
inline double pow(double x, double p) { return std::pow(x, p); }
// The same dual numbers in `long double` (x87 80-bit: 64-bit mantissa): NOT a restatement of anything in the reference — a
// higher-precision evaluation of the same formulas, used to MEASURE the forward error of the double evaluation of a residual block
// (oracle_block_forward_error: how far two correct double evaluations of an ill-conditioned block may be from each other).
template <int N>
struct DualL {
    long double a;
    long double v[N];
    DualL() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
    DualL(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT
    DualL(long double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT
    DualL(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1.0L; }
};
template <int N> inline DualL<N> operator+(const DualL<N>& f, const DualL<N>& g) { DualL<N> h; h.a = f.a + g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] + g.v[i]; return h; }
template <int N> inline DualL<N> operator-(const DualL<N>& f, const DualL<N>& g) { DualL<N> h; h.a = f.a - g.a; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] - g.v[i]; return h; }
template <int N> inline DualL<N> operator-(const DualL<N>& f) { DualL<N> h; h.a = -f.a; for (int i = 0; i < N; ++i) h.v[i] = -f.v[i]; return h; }
template <int N> inline DualL<N> operator*(const DualL<N>& f, const DualL<N>& g) { DualL<N> h; h.a = f.a * g.a; for (int i = 0; i < N; ++i) h.v[i] = f.a * g.v[i] + f.v[i] * g.a; return h; }
template <int N> inline DualL<N> operator/(const DualL<N>& f, const DualL<N>& g) {
    DualL<N> h; const long double gi = 1.0L / g.a; const long double fg = f.a * gi; h.a = fg;
    for (int i = 0; i < N; ++i) h.v[i] = (f.v[i] - fg * g.v[i]) * gi;
    return h; }
template <int N> inline bool operator<(const DualL<N>& f, const DualL<N>& g) { return f.a < g.a; }
template <int N> inline DualL<N> sqrt(const DualL<N>& f) { DualL<N> h; const long double t = sqrtl(f.a); const long double ti = 1.0L / (2.0L * t); h.a = t; for (int i = 0; i < N; ++i) h.v[i] = f.v[i] * ti; return h; }
template <int N> inline DualL<N> cos(const DualL<N>& f) { DualL<N> h; h.a = cosl(f.a); const long double s = -sinl(f.a); for (int i = 0; i < N; ++i) h.v[i] = s * f.v[i]; return h; }
template <int N> inline DualL<N> sin(const DualL<N>& f) { DualL<N> h; h.a = sinl(f.a); const long double c = cosl(f.a); for (int i = 0; i < N; ++i) h.v[i] = c * f.v[i]; return h; }
template <int N> inline DualL<N> pow(const DualL<N>& f, double p) { DualL<N> h; h.a = powl(f.a, (long double)p); const long double t = (long double)p * powl(f.a, (long double)p - 1.0L); for (int i = 0; i < N; ++i) h.v[i] = t * f.v[i]; return h; }
template <class T> inline double scalar_of(const T& x) { return x.a; }
template <> inline double scalar_of<double>(const double& x) { return x; }

// ---------------------------------------------------------------------------------------------
// Tiny fixed-size linear algebra (stands in for Eigen::Matrix3d / Vector3d; products are evaluated
// coefficient-wise, left to right — Eigen's lazy 3x3 product order is not observable here).
// ---------------------------------------------------------------------------------------------
template <class T> struct V3 { T x, y, z; T& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); } const T& operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); } };
template <class T> struct M3 { T m[9]; T& operator()(int r, int c) { return m[r * 3 + c]; } const T& operator()(int r, int c) const { return m[r * 3 + c]; } };
using V3d = V3<double>;
using M3d = M3<double>;

template <class T> inline V3<T> operator+(const V3<T>& a, const V3<T>& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class T> inline V3<T> operator-(const V3<T>& a, const V3<T>& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class T> inline V3<T> operator*(const V3<T>& a, const T& s) { return {a.x * s, a.y * s, a.z * s}; }
template <class T> inline T dot(const V3<T>& a, const V3<T>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class T> inline V3<T> cross(const V3<T>& a, const V3<T>& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
template <class T> inline V3<T> mul(const M3<T>& A, const V3<T>& v) {
    return {A(0, 0) * v.x + A(0, 1) * v.y + A(0, 2) * v.z, A(1, 0) * v.x + A(1, 1) * v.y + A(1, 2) * v.z,
            A(2, 0) * v.x + A(2, 1) * v.y + A(2, 2) * v.z}; }
template <class T> inline M3<T> mul(const M3<T>& A, const M3<T>& B) {
    M3<T> C; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) C(r, c) = A(r, 0) * B(0, c) + A(r, 1) * B(1, c) + A(r, 2) * B(2, c); return C; }
template <class T> inline M3<T> transpose(const M3<T>& A) { M3<T> B; for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) B(r, c) = A(c, r); return B; }
inline double norm(const V3d& a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }

// Rigid transform = Eigen::Isometry3d. apply(): linear*v + translation (pointcloud.h:82-86).
struct Iso3 { M3d R; V3d t; };
inline V3d apply(const Iso3& T, const V3d& p) { V3d q = mul(T.R, p); return {q.x + T.t.x, q.y + T.t.y, q.z + T.t.z}; }
// Eigen Transform<Isometry>::inverse(): R^T, -(R^T * t)
inline Iso3 inverse(const Iso3& T) { Iso3 I; I.R = transpose(T.R); V3d q = mul(I.R, T.t); I.t = {-q.x, -q.y, -q.z}; return I; }
inline Iso3 compose(const Iso3& A, const Iso3& B) { Iso3 C; C.R = mul(A.R, B.R); V3d q = mul(A.R, B.t); C.t = {q.x + A.t.x, q.y + A.t.y, q.z + A.t.z}; return C; }

// ---------------------------------------------------------------------------------------------
// skew / Sim3Exp / SE3Exp — g2o_tools.h:58-69, 105-140, 149-183
// ---------------------------------------------------------------------------------------------
template <class T> inline M3<T> skew(const V3<T>& v) {
    M3<T> m; for (int i = 0; i < 9; ++i) m.m[i] = T(0.0);
    m(0, 1) = -v.z; m(0, 2) = v.y; m(1, 2) = -v.x; m(1, 0) = v.z; m(2, 0) = -v.y; m(2, 1) = v.x; return m; }

template <class T>
inline void SE3ExpImpl(const T* update, M3<T>& R, V3<T>& t) {
    V3<T> omega{update[0], update[1], update[2]};
    V3<T> upsilon{update[3], update[4], update[5]};
    T theta = sqrt(omega.x * omega.x + omega.y * omega.y + omega.z * omega.z);
    M3<T> Omega = skew<T>(omega);
    M3<T> Omega2 = mul(Omega, Omega);
    M3<T> V;
    if (theta < T(1e-4)) {  // g2o_tools.h:119-124 Taylor branch
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            T I = T(r == c ? 1.0 : 0.0);
            R(r, c) = (I + Omega(r, c)) + T(0.5) * Omega2(r, c);
            V(r, c) = (I + T(0.5) * Omega(r, c)) + (T(1.) / T(6.)) * Omega2(r, c);
        }
    } else {                // g2o_tools.h:125-137
        T costh = cos(theta), sinth = sin(theta);
        T invth2 = pow(theta, -2.0), invth3 = pow(theta, -3.0);
        T ka = sinth / theta, kb = (T(1.) - costh) * invth2, kc = (theta - sinth) * invth3;
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            T I = T(r == c ? 1.0 : 0.0);
            R(r, c) = (I + ka * Omega(r, c)) + kb * Omega2(r, c);
            V(r, c) = (I + kb * Omega(r, c)) + kc * Omega2(r, c);
        }
    }
    t = mul(V, upsilon);
}
template <class T> inline void Sim3Exp(const T* update, M3<T>& R, V3<T>& t, T& s) { SE3ExpImpl(update, R, t); s = update[6]; }
template <class T> inline void SE3Exp(const T* update, M3<T>& R, V3<T>& t) { SE3ExpImpl(update, R, t); }

// ---------------------------------------------------------------------------------------------
// SE3Log — g2o_tools.h:78-82 -> g2o::SE3Quat(R,t).log()  (third-party g2o tag 20230223_git, absent;
// restated from g2o/types/slam3d/se3quat.h + Eigen's Quaternion(Matrix3) / toRotationMatrix()).
// ---------------------------------------------------------------------------------------------
inline void quat_from_R(const M3d& mat, double q[4] /*x,y,z,w*/) {
    double t = mat(0, 0) + mat(1, 1) + mat(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0); q[3] = 0.5 * t; t = 0.5 / t;
        q[0] = (mat(2, 1) - mat(1, 2)) * t; q[1] = (mat(0, 2) - mat(2, 0)) * t; q[2] = (mat(1, 0) - mat(0, 1)) * t;
    } else {
        int i = 0; if (mat(1, 1) > mat(0, 0)) i = 1; if (mat(2, 2) > mat(i, i)) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(mat(i, i) - mat(j, j) - mat(k, k) + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        q[3] = (mat(k, j) - mat(j, k)) * t; q[j] = (mat(j, i) + mat(i, j)) * t; q[k] = (mat(k, i) + mat(i, k)) * t;
    }
}
inline M3d R_from_quat(const double q[4]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    M3d r; r(0, 0) = 1 - (tyy + tzz); r(0, 1) = txy - twz; r(0, 2) = txz + twy; r(1, 0) = txy + twz; r(1, 1) = 1 - (txx + tzz);
    r(1, 2) = tyz - twx; r(2, 0) = txz - twy; r(2, 1) = tyz + twx; r(2, 2) = 1 - (txx + tyy); return r;
}
inline void SE3Log(const M3d& Rin, const V3d& t, double out[6]) {
    double q[4]; quat_from_R(Rin, q);
    if (q[3] < 0) { for (int i = 0; i < 4; ++i) q[i] = -q[i]; }           // SE3Quat::normalizeRotation
    const double n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; ++i) q[i] /= n;
    const M3d R = R_from_quat(q);
    const double d = 0.5 * (R(0, 0) + R(1, 1) + R(2, 2) - 1);
    V3d dR{R(2, 1) - R(1, 2), R(0, 2) - R(2, 0), R(1, 0) - R(0, 1)};
    V3d omega; M3d Vinv;
    if (std::abs(d) > 0.99999) {
        omega = {0.5 * dR.x, 0.5 * dR.y, 0.5 * dR.z};
        M3d Om = skew<double>(omega); M3d Om2 = mul(Om, Om);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Vinv(r, c) = ((r == c ? 1.0 : 0.0) - 0.5 * Om(r, c)) + (1. / 12.) * Om2(r, c);
    } else {
        const double theta = std::acos(d);
        const double k = theta / (2 * std::sqrt(1 - d * d));
        omega = {k * dR.x, k * dR.y, k * dR.z};
        M3d Om = skew<double>(omega); M3d Om2 = mul(Om, Om);
        const double c2 = (1 - theta / (2 * std::tan(theta / 2))) / (theta * theta);
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Vinv(r, c) = ((r == c ? 1.0 : 0.0) - 0.5 * Om(r, c)) + c2 * Om2(r, c);
    }
    V3d up = mul(Vinv, t);
    out[0] = omega.x; out[1] = omega.y; out[2] = omega.z; out[3] = up.x; out[4] = up.y; out[5] = up.z;
}

// ---------------------------------------------------------------------------------------------
// ComputeCovariance (one-pass raw moments) — pointcloud.h:126-158
// ---------------------------------------------------------------------------------------------
inline M3d ComputeCovariance(const double* pts /*AoS*/, const uint32_t* indices, size_t n) {
    M3d cov;
    if (n == 0) { for (int i = 0; i < 9; ++i) cov.m[i] = (i % 4 == 0) ? 1.0 : 0.0; return cov; }
    double c[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (size_t j = 0; j < n; ++j) {
        const double* p = pts + 3 * (size_t)indices[j];
        c[0] += p[0]; c[1] += p[1]; c[2] += p[2];
        c[3] += p[0] * p[0]; c[4] += p[0] * p[1]; c[5] += p[0] * p[2];
        c[6] += p[1] * p[1]; c[7] += p[1] * p[2]; c[8] += p[2] * p[2];
    }
    for (int i = 0; i < 9; ++i) c[i] /= (double)n;
    cov(0, 0) = c[3] - c[0] * c[0]; cov(1, 1) = c[6] - c[1] * c[1]; cov(2, 2) = c[8] - c[2] * c[2];
    cov(0, 1) = c[4] - c[0] * c[1]; cov(1, 0) = cov(0, 1);
    cov(0, 2) = c[5] - c[0] * c[2]; cov(2, 0) = cov(0, 2);
    cov(1, 2) = c[7] - c[1] * c[2]; cov(2, 1) = cov(1, 2);
    return cov;
}

// ---------------------------------------------------------------------------------------------
// ComputeEigenvector0/1, FastEigen3x3_EV — pointcloud.h:194-288, 378-463
// ---------------------------------------------------------------------------------------------
inline V3d ComputeEigenvector0(const M3d& A, double eval0) {
    V3d row0{A(0, 0) - eval0, A(0, 1), A(0, 2)};
    V3d row1{A(0, 1), A(1, 1) - eval0, A(1, 2)};
    V3d row2{A(0, 2), A(1, 2), A(2, 2) - eval0};
    V3d r0xr1 = cross(row0, row1), r0xr2 = cross(row0, row2), r1xr2 = cross(row1, row2);
    double d0 = dot(r0xr1, r0xr1), d1 = dot(r0xr2, r0xr2), d2 = dot(r1xr2, r1xr2);
    double dmax = d0; int imax = 0;
    if (d1 > dmax) { dmax = d1; imax = 1; }
    if (d2 > dmax) { imax = 2; }
    if (imax == 0) { double s = std::sqrt(d0); return {r0xr1.x / s, r0xr1.y / s, r0xr1.z / s}; }
    else if (imax == 1) { double s = std::sqrt(d1); return {r0xr2.x / s, r0xr2.y / s, r0xr2.z / s}; }
    else { double s = std::sqrt(d2); return {r1xr2.x / s, r1xr2.y / s, r1xr2.z / s}; }
}
inline V3d ComputeEigenvector1(const M3d& A, const V3d& evec0, double eval1) {
    V3d U, V;
    if (std::abs(evec0.x) > std::abs(evec0.y)) {
        double inv_length = 1 / std::sqrt(evec0.x * evec0.x + evec0.z * evec0.z);
        U = {-evec0.z * inv_length, 0, evec0.x * inv_length};
    } else {
        double inv_length = 1 / std::sqrt(evec0.y * evec0.y + evec0.z * evec0.z);
        U = {0, evec0.z * inv_length, -evec0.y * inv_length};
    }
    V = cross(evec0, U);
    V3d AU{A(0, 0) * U.x + A(0, 1) * U.y + A(0, 2) * U.z, A(0, 1) * U.x + A(1, 1) * U.y + A(1, 2) * U.z, A(0, 2) * U.x + A(1, 2) * U.y + A(2, 2) * U.z};
    V3d AV{A(0, 0) * V.x + A(0, 1) * V.y + A(0, 2) * V.z, A(0, 1) * V.x + A(1, 1) * V.y + A(1, 2) * V.z, A(0, 2) * V.x + A(1, 2) * V.y + A(2, 2) * V.z};
    double m00 = U.x * AU.x + U.y * AU.y + U.z * AU.z - eval1;
    double m01 = U.x * AV.x + U.y * AV.y + U.z * AV.z;
    double m11 = V.x * AV.x + V.y * AV.y + V.z * AV.z - eval1;
    double absM00 = std::abs(m00), absM01 = std::abs(m01), absM11 = std::abs(m11);
    double max_abs_comp;
    if (absM00 >= absM11) {
        max_abs_comp = std::max(absM00, absM01);
        if (max_abs_comp > 0) {
            if (absM00 >= absM01) { m01 /= m00; m00 = 1 / std::sqrt(1 + m01 * m01); m01 *= m00; }
            else { m00 /= m01; m01 = 1 / std::sqrt(1 + m00 * m00); m00 *= m01; }
            return {m01 * U.x - m00 * V.x, m01 * U.y - m00 * V.y, m01 * U.z - m00 * V.z};
        } else return U;
    } else {
        max_abs_comp = std::max(absM11, absM01);
        if (max_abs_comp > 0) {
            if (absM11 >= absM01) { m01 /= m11; m11 = 1 / std::sqrt(1 + m01 * m01); m01 *= m11; }
            else { m11 /= m01; m01 = 1 / std::sqrt(1 + m11 * m11); m11 *= m01; }
            return {m11 * U.x - m01 * V.x, m11 * U.y - m01 * V.y, m11 * U.z - m01 * V.z};
        } else return U;
    }
}
// returns eigenvector of the smallest eigenvalue; evals = eigenvalues of the SCALED matrix.
inline V3d FastEigen3x3_EV(const M3d& covariance, double evals[3]) {
    M3d A = covariance;
    evals[0] = evals[1] = evals[2] = 0;
    double max_coeff = A.m[0];
    for (int i = 1; i < 9; ++i) if (A.m[i] > max_coeff) max_coeff = A.m[i];  // signed maxCoeff (:386)
    if (max_coeff == 0) return {0, 0, 0};
    for (int i = 0; i < 9; ++i) A.m[i] /= max_coeff;
    double nrm = A(0, 1) * A(0, 1) + A(0, 2) * A(0, 2) + A(1, 2) * A(1, 2);
    if (nrm > 0) {
        double q = (A(0, 0) + A(1, 1) + A(2, 2)) / 3;
        double b00 = A(0, 0) - q, b11 = A(1, 1) - q, b22 = A(2, 2) - q;
        double p = std::sqrt((b00 * b00 + b11 * b11 + b22 * b22 + nrm * 2) / 6);
        double c00 = b11 * b22 - A(1, 2) * A(1, 2);
        double c01 = A(0, 1) * b22 - A(1, 2) * A(0, 2);
        double c02 = A(0, 1) * A(1, 2) - b11 * A(0, 2);
        double det = (b00 * c00 - A(0, 1) * c01 + A(0, 2) * c02) / (p * p * p);
        double half_det = det * 0.5;
        half_det = std::min(std::max(half_det, -1.0), 1.0);
        double angle = std::acos(half_det) / (double)3;
        double const two_thirds_pi = 2.09439510239319549;
        double beta2 = std::cos(angle) * 2;
        double beta0 = std::cos(angle + two_thirds_pi) * 2;
        double beta1 = -(beta0 + beta2);
        evals[0] = q + p * beta0; evals[1] = q + p * beta1; evals[2] = q + p * beta2;
        if (half_det >= 0) {
            V3d evec2 = ComputeEigenvector0(A, evals[2]);
            if (evals[2] < evals[0] && evals[2] < evals[1]) return evec2;
            V3d evec1 = ComputeEigenvector1(A, evec2, evals[1]);
            if (evals[1] < evals[0] && evals[1] < evals[2]) return evec1;
            return cross(evec1, evec2);
        } else {
            V3d evec0 = ComputeEigenvector0(A, evals[0]);
            if (evals[0] < evals[1] && evals[0] < evals[2]) return evec0;
            V3d evec1 = ComputeEigenvector1(A, evec0, evals[1]);
            if (evals[1] < evals[0] && evals[1] < evals[2]) return evec1;
            return cross(evec0, evec1);
        }
    } else {
        for (int i = 0; i < 9; ++i) A.m[i] *= max_coeff;
        evals[0] = evals[1] = evals[2] = 0;
        if (A(0, 0) < A(1, 1) && A(0, 0) < A(2, 2)) return {1, 0, 0};
        else if (A(1, 1) < A(0, 0) && A(1, 1) < A(2, 2)) return {0, 1, 0};
        else return {0, 0, 1};
    }
}
inline V3d normalized(const V3d& v) {  // Eigen normalize(): divides by norm() when squaredNorm > 0
    double z = v.x * v.x + v.y * v.y + v.z * v.z;
    if (z > 0) { double n = std::sqrt(z); return {v.x / n, v.y / n, v.z / n}; }
    return v;
}

}  // namespace oracle
