// SplineBuilders.h -- TSpline3_red coefficients from knots alone, ROOT-free (set-up time).
//
// The reference builds a TSpline3_red from a ROOT TSpline3 (Splines/SplineStructs.h:311-652):
// ROOT supplies the knots and, for kTSpline3, the cubic coefficients; the other interpolation
// schemes (kLinear, kMonotonic, kAkima, kKochanekBartels) are MaCh3's own arithmetic on the knots.
// Here the ROOT part is replaced by RootTSpline3Coefficients(), the Forsythe-Malcolm-Moler `spline`
// routine that TSpline3 implements (no end-condition options).  Arithmetic types follow the
// reference's float_t = float build operation by operation, so the coefficients are bit-identical
// to what SetFunc would store given the same ROOT spline (tests/test_spline_builders.py).
#pragma once

#include <array>
#include <cmath>
#include <vector>

#include "Core.h"
#include "SplineStructs.h"

namespace M3 {

/// b, c, d of ROOT's TSpline3 through (x, y), in double (FMM `spline`; third-party algorithm)
inline void RootTSpline3Coefficients(const std::vector<double>& x, const std::vector<double>& y, std::vector<double>& b,
                                     std::vector<double>& c, std::vector<double>& d) {
  const int n = int(x.size());
  b.assign(n, 0.0); c.assign(n, 0.0); d.assign(n, 0.0);
  if (n < 2) return;
  const int last = n - 1;
  if (n == 2) {
    b[0] = b[1] = (y[1] - y[0]) / (x[1] - x[0]);
    return;
  }
  // tridiagonal system: b = diagonal, d = off-diagonal, c = right-hand side
  d[0] = x[1] - x[0];
  c[1] = (y[1] - y[0]) / d[0];
  for (int i = 1; i < last; ++i) {
    d[i] = x[i + 1] - x[i];
    b[i] = 2. * (d[i - 1] + d[i]);
    c[i + 1] = (y[i + 1] - y[i]) / d[i];
    c[i] = c[i + 1] - c[i];
  }
  // end conditions: third derivatives from the cubics through the first / last four points
  b[0] = -d[0];
  b[last] = -d[n - 2];
  c[0] = c[last] = 0.;
  if (n != 3) {
    c[0] = c[2] / (x[3] - x[1]) - c[1] / (x[2] - x[0]);
    c[last] = c[n - 2] / (x[last] - x[n - 3]) - c[n - 3] / (x[n - 2] - x[n - 4]);
    c[0] = c[0] * d[0] * d[0] / (x[3] - x[0]);
    c[last] = -c[last] * d[n - 2] * d[n - 2] / (x[last] - x[n - 4]);
  }
  for (int i = 1; i < n; ++i) {  // forward elimination
    const double t = d[i - 1] / b[i - 1];
    b[i] = b[i] - t * d[i - 1];
    c[i] = c[i] - t * c[i - 1];
  }
  c[last] = c[last] / b[last];  // back substitution
  for (int i = n - 2; i >= 0; --i) c[i] = (c[i] - d[i] * c[i + 1]) / b[i];
  b[last] = (y[last] - y[n - 2]) / d[n - 2] + d[n - 2] * (c[n - 2] + 2. * c[last]);
  for (int i = 0; i < last; ++i) {
    b[i] = (y[i + 1] - y[i]) / d[i] - d[i] * (c[i + 1] + 2. * c[i]);
    d[i] = (c[i + 1] - c[i]) / d[i];
    c[i] = 3. * c[i];
  }
  c[last] = 3. * c[last];
  d[last] = d[n - 2];
}

namespace detail {
using F = M3::float_t;
struct Knots {
  std::vector<F> x, y;
  std::vector<std::array<F, 3>> p;  // {b, c, d}
  explicit Knots(const std::vector<double>& xs, const std::vector<double>& ys) : x(xs.size()), y(ys.size()), p(xs.size(), {0.f, 0.f, 0.f}) {
    for (size_t i = 0; i < xs.size(); ++i) { x[i] = F(xs[i]); y[i] = F(ys[i]); }
  }
};

// segments the input ROOT spline holds as straight lines stay straight (SplineStructs.h:434-444, 552-562)
inline void KeepLinearSegments(Knots& k, const std::vector<double>& rb, const std::vector<double>& rc, const std::vector<double>& rd) {
  for (size_t i = 0; i + 1 < k.x.size(); ++i)
    if (rc[i] == 0.0 && rd[i] == 0.0) k.p[i] = {F(rb[i]), F(0.0), F(0.0)};
}

inline void Linear(Knots& k, const std::vector<double>& xs, const std::vector<double>& ys) {  // :351-373
  const size_t n = xs.size();
  for (size_t i = 0; i < n; ++i) {
    const double slope = (i + 1 == n) ? double(k.p[i - 1][0]) : (ys[i + 1] - ys[i]) / (xs[i + 1] - xs[i]);
    k.p[i] = {F(slope), F(0), F(0)};
  }
}

inline void Akima(Knots& k) {  // :376-432
  const int n = int(k.x.size());
  const std::vector<F>& X = k.x;
  const std::vector<F>& Y = k.y;
  // secant slopes m[-2 .. n] (stored shifted by 2); two extrapolated slopes at either end
  std::vector<F> m(n + 3);
  auto sec = [&](int a, int b) { return (Y[b] - Y[a]) / (X[b] - X[a]); };
  m[0] = F(3.0 * (Y[1] - Y[0]) / (X[1] - X[0]) - 2.0 * (Y[2] - Y[1]) / (X[2] - X[1]));
  m[1] = F(2.0 * (Y[1] - Y[0]) / (X[1] - X[0]) - (Y[2] - Y[1]) / (X[2] - X[1]));
  for (int i = 0; i < n - 1; ++i) m[i + 2] = sec(i, i + 1);
  m[n + 1] = F(2.0 * (Y[n - 1] - Y[n - 2]) / (X[n - 1] - X[n - 2]) - (Y[n - 2] - Y[n - 3]) / (X[n - 2] - X[n - 3]));
  m[n + 2] = F(3.0 * (Y[n - 1] - Y[n - 2]) / (X[n - 1] - X[n - 2]) - 2.0 * (Y[n - 2] - Y[n - 3]) / (X[n - 2] - X[n - 3]));
  // knot slopes: Akima's weighted mean of the neighbouring secants
  std::vector<F> s(n);
  for (int j = 0; j < n; ++j) {
    const int i = j + 2;
    const F wl = std::abs(m[i + 1] - m[i]), wr = std::abs(m[i - 1] - m[i - 2]);
    s[j] = (wl + wr != 0.0) ? (wl * m[i - 1] + wr * m[i]) / (wl + wr) : m[i];
  }
  for (int i = 0; i < n - 1; ++i) {
    const F h = X[i + 1] - X[i];
    const F c = F(3.0 * (Y[i + 1] - Y[i]) / (X[i + 1] - X[i]) - 2.0 * s[i] - s[i + 1]) / h;
    const F d = F((s[i + 1] + s[i]) - 2.0 * (Y[i + 1] - Y[i]) / (X[i + 1] - X[i])) / (h * h);
    k.p[i] = {s[i], c, d};
  }
}

inline void Monotonic(Knots& k) {  // :450-550 (Fritsch-Carlson limiter on each segment)
  const int n = int(k.x.size());
  const std::vector<F>& X = k.x;
  const std::vector<F>& Y = k.y;
  if (n == 2) {
    k.p[0] = {(Y[1] - Y[0]) / (X[1] - X[0]), 0.f, 0.f};
    return;
  }
  std::vector<F> sec(n - 1), tan(n, 0.f);
  for (int i = 0; i < n - 1; ++i) sec[i] = (Y[i + 1] - Y[i]) / (X[i + 1] - X[i]);
  tan[0] = sec[0];
  tan[n - 1] = sec[n - 2];
  for (int i = 1; i < n - 1; ++i) {
    const bool same_sign = (sec[i - 1] >= 0.0 && sec[i] >= 0.0) | (sec[i - 1] < 0.0 && sec[i] < 0.0);
    if (same_sign) tan[i] = F((sec[i - 1] + sec[i]) / 2.0);
  }
  for (int i = 0; i < n - 1; ++i) {
    if (sec[i] == 0.0) {
      tan[i] = tan[i + 1] = 0.f;
      continue;
    }
    const F alpha = tan[i] / sec[i], beta = tan[i + 1] / sec[i];
    if (alpha < 0.0) tan[i] = 0.f;
    if (beta < 0.0) tan[i + 1] = 0.f;
    if (alpha * alpha + beta * beta > 9.0) {
      const F tau = F(3.0 / std::sqrt(alpha * alpha + beta * beta));
      tan[i] = tau * alpha * sec[i];
      tan[i + 1] = tau * beta * sec[i];
    }
  }
  for (int i = 0; i < n - 1; ++i) {
    const F h = X[i + 1] - X[i];
    const F b = tan[i] * h;
    const F c = F(3.0 * (Y[i + 1] - Y[i]) - 2.0 * h * tan[i] - h * tan[i + 1]);
    const F d = F(2.0 * (Y[i] - Y[i + 1]) + h * (tan[i] + tan[i + 1]));
    k.p[i] = {b / h, c / (h * h), d / (h * h * h)};
  }
}

inline void KochanekBartels(Knots& k) {  // :568-643 with tension = continuity = bias = 0 (Catmull-Rom)
  const int n = int(k.x.size());
  const std::vector<F>& X = k.x;
  const std::vector<F>& Y = k.y;
  constexpr F T = 0.0, C = 0.0, B = 0.0;
  std::vector<F> tan(n, 0.f);
  for (int i = 1; i < n - 1; ++i) {
    const F d0 = (Y[i] - Y[i - 1]) / (X[i] - X[i - 1]);
    const F d1 = (Y[i + 1] - Y[i]) / (X[i + 1] - X[i]);
    const F in = (1.0 - T) * (1.0 + B) * (1.0 + C) * 0.5 * d0;
    const F ou = (1.0 - T) * (1.0 - B) * (1.0 - C) * 0.5 * d1;
    tan[i] = in + ou;
  }
  tan[0] = (Y[1] - Y[0]) / (X[1] - X[0]);
  tan[n - 1] = (Y[n - 1] - Y[n - 2]) / (X[n - 1] - X[n - 2]);
  for (int i = 0; i < n - 1; ++i) {
    const F h = X[i + 1] - X[i], dy = Y[i + 1] - Y[i], m0 = tan[i], m1 = tan[i + 1];
    k.p[i] = {m0, (3 * dy / (h * h)) - (2 * m0 + m1) / h, (m0 + m1 - 2 * dy / h) / (h * h)};
  }
}
}  // namespace detail
}  // namespace M3

/// A TSpline3_red through (x, y) with the requested interpolation (what SetFunc would leave in the
/// object, ref: Splines/SplineStructs.h:311-652).  The last knot carries {0,0,0}: it is never
/// evaluated (SplineBase.cpp:97).
inline TSpline3_red* MakeTSpline3_red(const std::vector<double>& x, const std::vector<double>& y, SplineInterpolation InterPolation = kTSpline3) {
  if (x.size() != y.size() || x.size() < 2) throw MaCh3Exception(__FILE__, __LINE__, "MakeTSpline3_red: need >= 2 knots");
  if ((InterPolation == kAkima || InterPolation == kKochanekBartels) && x.size() < 3)
    throw MaCh3Exception(__FILE__, __LINE__, "MakeTSpline3_red: this interpolation needs >= 3 knots");
  std::vector<double> rb, rc, rd;
  M3::RootTSpline3Coefficients(x, y, rb, rc, rd);
  M3::detail::Knots k(x, y);
  switch (InterPolation) {
    case kTSpline3:
      for (size_t i = 0; i < x.size(); ++i) k.p[i] = {M3::float_t(rb[i]), M3::float_t(rc[i]), M3::float_t(rd[i])};
      break;
    case kLinear:
    case kLinearFunc: M3::detail::Linear(k, x, y); break;
    case kAkima: M3::detail::Akima(k); M3::detail::KeepLinearSegments(k, rb, rc, rd); break;
    case kMonotonic: M3::detail::Monotonic(k); if (x.size() > 2) M3::detail::KeepLinearSegments(k, rb, rc, rd); break;
    case kKochanekBartels: M3::detail::KochanekBartels(k); break;
    default: throw MaCh3Exception(__FILE__, __LINE__, "Unsupported interpolation type");
  }
  std::vector<M3::float_t> flat(x.size() * 3);
  for (size_t i = 0; i < x.size(); ++i)
    for (int j = 0; j < 3; ++j) flat[3 * i + j] = k.p[i][size_t(j)];
  return new TSpline3_red(k.x.data(), k.y.data(), M3::int_t(x.size()), flat.data());
}
