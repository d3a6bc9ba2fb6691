/*
 * m3_oracle.cpp -- TEST INFRASTRUCTURE ONLY (see m3_oracle.h).
 *
 * ROOT-free CPU restatement of the reference's per-step likelihood path, in the
 * reference's data layout and operation order.  "ref:" comments give the
 * /root/reference file:line each block follows.  PARITY UNPINNED (no golden
 * vectors exist in the reference tree); pinned by authored known-answer tests.
 *
 * Built twice by oracle/Makefile:
 *   libm3oracle_serial.so  no OpenMP: `omp simd reduction` pragmas are inert, so
 *                          float products run strictly left to right -- the
 *                          bit-exact reference for per-event weights.
 *   libm3oracle_omp.so     -fopenmp -DMULTITHREAD: the reference's pragmas active;
 *                          this is the timed CPU baseline ("port").
 * float_t is float: SMonolith only works in the _LOW_MEMORY_STRUCTS_ build
 * (ref: Samples/SampleHandlerFD.cpp:1244-1254).
 */
#include "m3_oracle.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <new>
#include <vector>

#ifdef MULTITHREAD
#include <omp.h>
#endif

namespace {

typedef m3o_float_t m3_float_t;  // ref: Manager/Core.h:27-39 (float with _LOW_MEMORY_STRUCTS_, else double)
// ref: Manager/Core.h:44-51 M3::fmaf_t
static inline float m3_fmaf_t(float x, float y, float z) { return std::fmaf(x, y, z); }
static inline double m3_fmaf_t(double x, double y, double z) { return std::fma(x, y, z); }
typedef short m3_int_t;

const double kLowMCBound = M3O_LOW_MC_BOUND;
const m3_float_t kUnity = 1.0f;  // ref: Manager/Core.h:56-72

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

/* ------------------------------------------------------------------------- */
/* Test statistics                                                           */
/* ------------------------------------------------------------------------- */

// ref: Samples/SampleHandlerBase.cpp:17-31
double m3o_poisson_llh(double data, double mc) {
  if (data == 0) return mc;
  if (mc < kLowMCBound) {
    if (data > kLowMCBound) return (kLowMCBound - data + data * std::log(data / kLowMCBound));
    else if (data >= mc) return 0.;
  }
  return (mc - data + data * std::log(data / mc));
}

// ref: Samples/SampleHandlerBase.cpp:35-192
double m3o_test_stat_llh(int kind, double data, double mc, double w2) {
  switch (kind) {
    case M3O_kBarlowBeeston: {  // ref: :46-87
      double newmc = mc;
      if (mc < kLowMCBound) {
        if (data > kLowMCBound) newmc = kLowMCBound;
        else if (data >= mc) return 0.;
      }
      const double fractional = std::sqrt(w2) / newmc;
      const double fractional2 = fractional * fractional;
      const double temp = newmc * fractional2 - 1;
      const double temp2 = temp * temp + 4 * data * fractional2;
      if (temp2 < 0) return std::numeric_limits<double>::quiet_NaN();  // reference throws
      const double beta = (-1 * temp + sqrt(temp2)) / 2.;
      double stat = mc * beta;
      if (data > 0) {
        newmc *= beta;
        stat = newmc - data + data * std::log(data / newmc);
      }
      double penalty = 0;
      if (fractional > 0) penalty = (beta - 1) * (beta - 1) / (2 * fractional2);
      return stat + penalty;
    }
    case M3O_kDembinskiAbdelmotteleb: {  // ref: :90-125
      if (w2 == 0) return m3o_poisson_llh(data, mc);
      double newmc = mc;
      if (mc < kLowMCBound) {
        if (data > kLowMCBound) newmc = kLowMCBound;
        else if (data >= mc) return 0.;
      }
      const double k = newmc * newmc / w2;
      const double beta = (data + k) / (newmc + k);
      newmc *= beta;
      const double penalty = k * beta - k + k * std::log(k / (k * beta));
      double stat = newmc;
      if (data > 0) stat = newmc - data + data * std::log(data / newmc);
      return stat + penalty;
    }
    case M3O_kIceCube: {  // ref: :133-159 (long double + lgammal)
      if (w2 == 0) return m3o_poisson_llh(data, mc);
      const long double b = mc / w2;
      const long double a = mc * b + 1;
      const double stat = double(-1 * (a * logl(b) + lgammal(data + a) - lgammal(data + 1) -
                                       ((data + a) * log1pl(b)) - lgammal(a)));
      if (mc <= data) {
        if (data <= kLowMCBound) return 0.;
        const double poisson = m3o_poisson_llh(data, kLowMCBound);
        if (stat > poisson) return poisson;
      }
      return stat;
    }
    case M3O_kPearson: {  // ref: :162-176
      if (data == 0) return mc / 2.;
      if (mc < kLowMCBound) {
        if (data > kLowMCBound) return (data - kLowMCBound) * (data - kLowMCBound) / (2. * kLowMCBound);
        else if (data >= mc) return 0.;
      }
      return (data - mc) * (data - mc) / (2 * mc);
    }
    case M3O_kPoisson:  // ref: :178-183
      return m3o_poisson_llh(data, mc);
    default:
      return std::numeric_limits<double>::quiet_NaN();  // reference throws
  }
}

/* ------------------------------------------------------------------------- */
/* Reduced response functions (double-precision per-spline cross-check)       */
/* ------------------------------------------------------------------------- */

// ref: Splines/SplineStructs.h:673-703 (TSpline3_red::FindX)
int m3o_tspline3_find_x(const float* xpos, int npoints, double x) {
  int segment = 0;
  int kHigh = npoints - 1;
  if (x <= xpos[0]) {
    segment = 0;
  } else if (x >= xpos[npoints - 1]) {
    segment = kHigh;
  } else {
    int kHalf = 0;
    while (kHigh - segment > 1) {
      kHalf = (segment + kHigh) / 2;
      if (x > xpos[kHalf]) segment = kHalf;
      else kHigh = kHalf;
    }
  }
  if (segment >= npoints - 1 && npoints > 1) segment = npoints - 2;
  return segment;
}

// ref: Splines/SplineStructs.h:706-717 (TSpline3_red::Eval)
double m3o_tspline3_eval(const float* xpos, const float* yresp, const float* par, int npoints,
                         double var) {
  const int segment = m3o_tspline3_find_x(xpos, npoints, var);
  const m3_float_t x = xpos[segment], y = yresp[segment];
  const m3_float_t b = par[3 * segment + 0], c = par[3 * segment + 1], d = par[3 * segment + 2];
  const double dx = var - x;
  return y + dx * (b + dx * (c + d * dx));
}

// ref: Splines/SplineStructs.h:192-193 (TF1_red::Eval)
double m3o_tf1_eval(const float* par, double var) { return par[1] + par[0] * var; }

/* ------------------------------------------------------------------------- */
/* TSpline3_red coefficient builders (set-up time)                            */
/* ------------------------------------------------------------------------- */

// ROOT's TSpline3 (no end-condition options) is the `spline` routine of Forsythe, Malcolm & Moler,
// "Computer Methods for Mathematical Computations" (1977): end conditions from the cubics through
// the first / last four points.  ROOT is not part of the reference tree (third-party, parity
// unpinned); this restates the published algorithm in double, as ROOT runs it.
void m3o_root_tspline3_coeffs(int n, const double* x, const double* y, double* b, double* c, double* d) {
  if (n < 2) { if (n == 1) { b[0] = c[0] = d[0] = 0; } return; }
  const int nm1 = n - 1;
  if (n < 3) {
    b[0] = (y[1] - y[0]) / (x[1] - x[0]);
    c[0] = 0.; d[0] = 0.;
    b[1] = b[0]; c[1] = 0.; d[1] = 0.;
    return;
  }
  d[0] = x[1] - x[0];
  c[1] = (y[1] - y[0]) / d[0];
  for (int i = 1; i < nm1; ++i) {
    d[i] = x[i + 1] - x[i];
    b[i] = 2. * (d[i - 1] + d[i]);
    c[i + 1] = (y[i + 1] - y[i]) / d[i];
    c[i] = c[i + 1] - c[i];
  }
  b[0] = -d[0];
  b[nm1] = -d[n - 2];
  c[0] = 0.;
  c[nm1] = 0.;
  if (n != 3) {
    c[0] = c[2] / (x[3] - x[1]) - c[1] / (x[2] - x[0]);
    c[nm1] = c[n - 2] / (x[nm1] - x[n - 3]) - c[n - 3] / (x[n - 2] - x[n - 4]);
    c[0] = c[0] * d[0] * d[0] / (x[3] - x[0]);
    c[nm1] = -c[nm1] * d[n - 2] * d[n - 2] / (x[nm1] - x[n - 4]);
  }
  for (int i = 1; i < n; ++i) {
    const double t = d[i - 1] / b[i - 1];
    b[i] = b[i] - t * d[i - 1];
    c[i] = c[i] - t * c[i - 1];
  }
  c[nm1] = c[nm1] / b[nm1];
  for (int ib = 1; ib < n; ++ib) {
    const int i = n - 1 - ib;
    c[i] = (c[i] - d[i] * c[i + 1]) / b[i];
  }
  b[nm1] = (y[nm1] - y[n - 2]) / d[n - 2] + d[n - 2] * (c[n - 2] + 2. * c[nm1]);
  for (int i = 0; i < nm1; ++i) {
    b[i] = (y[i + 1] - y[i]) / d[i] - d[i] * (c[i + 1] + 2. * c[i]);
    d[i] = (c[i + 1] - c[i]) / d[i];
    c[i] = 3. * c[i];
  }
  c[nm1] = 3. * c[nm1];
  d[nm1] = d[n - 2];
}

// ref: Splines/SplineStructs.h:311-652 (TSpline3_red::SetFunc).  `spline` = the ROOT TSpline3 the
// reference is handed: knots x,y (double) and ROOT's own b,c,d (root_bcd[n][3]).  Output per knot
// {x, y, b, c, d} as M3::float_t.  interpolation: 0 kTSpline3, 1 kLinear, 2 kMonotonic, 3 kAkima,
// 4 kKochanekBartels, 5 kLinearFunc.  Returns 0, or -1 for an unsupported type (reference throws).
int m3o_build_tspline3_red(int InterPolation, int nPoints, const double* sx, const double* sy, const double* root_bcd,
                           float* out /*[nPoints][5]*/) {
  std::vector<m3_float_t> XPos(nPoints), YResp(nPoints);
  std::vector<std::array<m3_float_t, 3>> Par(nPoints, {0.f, 0.f, 0.f});
  if (InterPolation == 0) {  // kTSpline3 :331-345
    for (int i = 0; i < nPoints; ++i) {
      XPos[i] = m3_float_t(sx[i]);
      YResp[i] = m3_float_t(sy[i]);
      Par[i][0] = m3_float_t(root_bcd[3 * i]);
      Par[i][1] = m3_float_t(root_bcd[3 * i + 1]);
      Par[i][2] = m3_float_t(root_bcd[3 * i + 2]);
    }
  } else if (InterPolation == 1 || InterPolation == 5) {  // kLinear / kLinearFunc :351-373
    for (int k = 0; k < nPoints; ++k) {
      const double x1 = sx[k], y1 = sy[k];
      double tempb = 0;
      if (k == nPoints - 1) {
        tempb = Par[k - 1][0];
      } else {
        const double x2 = sx[k + 1], y2 = sy[k + 1];
        tempb = (y2 - y1) / (x2 - x1);
      }
      XPos[k] = m3_float_t(x1);
      YResp[k] = m3_float_t(y1);
      Par[k][0] = m3_float_t(tempb);
      Par[k][1] = m3_float_t(0);
      Par[k][2] = m3_float_t(0);
    }
  } else if (InterPolation == 3) {  // kAkima :376-447
    for (int i = 0; i < nPoints; ++i) {
      XPos[i] = m3_float_t(sx[i]);
      YResp[i] = m3_float_t(sy[i]);
    }
    std::vector<m3_float_t> mvals(nPoints + 3), svals(nPoints + 1);
    for (int i = -2; i <= nPoints; ++i) {
      if (i == -2) {
        mvals[i + 2] = m3_float_t(3.0 * (YResp[1] - YResp[0]) / (XPos[1] - XPos[0]) - 2.0 * (YResp[2] - YResp[1]) / (XPos[2] - XPos[1]));
      } else if (i == -1) {
        mvals[i + 2] = m3_float_t(2.0 * (YResp[1] - YResp[0]) / (XPos[1] - XPos[0]) - (YResp[2] - YResp[1]) / (XPos[2] - XPos[1]));
      } else if (i == nPoints) {
        mvals[i + 2] = m3_float_t(3.0 * (YResp[nPoints - 1] - YResp[nPoints - 2]) / (XPos[nPoints - 1] - XPos[nPoints - 2]) -
                                  2.0 * (YResp[nPoints - 2] - YResp[nPoints - 3]) / (XPos[nPoints - 2] - XPos[nPoints - 3]));
      } else if (i == nPoints - 1) {
        mvals[i + 2] = m3_float_t(2.0 * (YResp[nPoints - 1] - YResp[nPoints - 2]) / (XPos[nPoints - 1] - XPos[nPoints - 2]) -
                                  (YResp[nPoints - 2] - YResp[nPoints - 3]) / (XPos[nPoints - 2] - XPos[nPoints - 3]));
      } else {
        mvals[i + 2] = (YResp[i + 1] - YResp[i]) / (XPos[i + 1] - XPos[i]);
      }
    }
    // the reference's loop runs i = 2 .. nPoints+2 and reads mvals[i+1] one past its nPoints+3
    // entries on the last pass (svals[nPoints] of that pass is only used for the last knot's
    // unused coefficients); the out-of-bounds element is taken as 0 here.
    auto mv = [&](int k) -> m3_float_t { return k < nPoints + 3 ? mvals[k] : m3_float_t(0); };
    for (int i = 2; i <= nPoints + 2; i++) {
      if (i - 2 > nPoints) break;
      if (std::abs(mv(i + 1) - mv(i)) + std::abs(mv(i - 1) - mv(i - 2)) != 0.0) {
        svals[i - 2] = (std::abs(mv(i + 1) - mv(i)) * mv(i - 1) + std::abs(mv(i - 1) - mv(i - 2)) * mv(i)) /
                       (std::abs(mv(i + 1) - mv(i)) + std::abs(mv(i - 1) - mv(i - 2)));
      } else {
        svals[i - 2] = mv(i);
      }
    }
    for (int i = 0; i < nPoints - 1; i++) {  // the reference also runs i = nPoints-1 (reads past the knots); that quad is never evaluated
      m3_float_t b, c, d;
      b = svals[i];
      c = m3_float_t(3.0 * (YResp[i + 1] - YResp[i]) / (XPos[i + 1] - XPos[i]) - 2.0 * svals[i] - svals[i + 1]) / (XPos[i + 1] - XPos[i]);
      d = m3_float_t((svals[i + 1] + svals[i]) - 2.0 * (YResp[i + 1] - YResp[i]) / (XPos[i + 1] - XPos[i])) /
          ((XPos[i + 1] - XPos[i]) * (XPos[i + 1] - XPos[i]));
      Par[i][0] = b; Par[i][1] = c; Par[i][2] = d;
    }
    for (int i = 0; i < nPoints - 1; i++) {  // :434-444 keep segments the input spline has as straight lines
      const double b = root_bcd[3 * i], c = root_bcd[3 * i + 1], d = root_bcd[3 * i + 2];
      if (c == 0.0 && d == 0.0) {
        Par[i][0] = m3_float_t(b); Par[i][1] = m3_float_t(0.0); Par[i][2] = m3_float_t(0.0);
      }
    }
  } else if (InterPolation == 2) {  // kMonotonic :450-566
    std::vector<m3_float_t> Secants(nPoints > 1 ? nPoints - 1 : 1), Tangents(nPoints);
    for (int i = 0; i < nPoints; ++i) {
      XPos[i] = m3_float_t(sx[i]);
      YResp[i] = m3_float_t(sy[i]);
      Tangents[i] = 0.0;
    }
    if (nPoints == 2) {
      Par[0][0] = (YResp[1] - YResp[0]) / (XPos[1] - XPos[0]);
      Par[0][1] = 0.0; Par[0][2] = 0.0; Par[1][1] = 0.0; Par[1][2] = 0.0;
    } else {
      for (int i = 0; i < nPoints - 1; ++i) Secants[i] = (YResp[i + 1] - YResp[i]) / (XPos[i + 1] - XPos[i]);
      Tangents[0] = Secants[0];
      Tangents[nPoints - 1] = Secants[nPoints - 2];
      m3_float_t alpha, beta;
      for (int i = 1; i < nPoints - 1; ++i) {
        if ((Secants[i - 1] >= 0.0 && Secants[i] >= 0.0) | (Secants[i - 1] < 0.0 && Secants[i] < 0.0)) {
          Tangents[i] = m3_float_t((Secants[i - 1] + Secants[i]) / 2.0);
        }
      }
      for (int i = 0; i < nPoints - 1; ++i) {
        if (Secants[i] == 0.0) {
          Tangents[i] = 0.0;
          Tangents[i + 1] = 0.0;
        } else {
          alpha = Tangents[i] / Secants[i];
          beta = Tangents[i + 1] / Secants[i];
          if (alpha < 0.0) Tangents[i] = 0.0;
          if (beta < 0.0) Tangents[i + 1] = 0.0;
          if (alpha * alpha + beta * beta > 9.0) {
            m3_float_t tau = m3_float_t(3.0 / std::sqrt(alpha * alpha + beta * beta));
            Tangents[i] = tau * alpha * Secants[i];
            Tangents[i + 1] = tau * beta * Secants[i];
          }
        }
      }
      m3_float_t dx;
      for (int i = 0; i < nPoints - 1; i++) {
        m3_float_t b, c, d;
        dx = XPos[i + 1] - XPos[i];
        b = Tangents[i] * dx;
        c = m3_float_t(3.0 * (YResp[i + 1] - YResp[i]) - 2.0 * dx * Tangents[i] - dx * Tangents[i + 1]);
        d = m3_float_t(2.0 * (YResp[i] - YResp[i + 1]) + dx * (Tangents[i] + Tangents[i + 1]));
        Par[i][0] = b / dx;
        Par[i][1] = c / (dx * dx);
        Par[i][2] = d / (dx * dx * dx);
      }
      Par[nPoints - 1][0] = 0.0; Par[nPoints - 1][1] = 0.0; Par[nPoints - 1][2] = 0.0;
      for (int i = 0; i < nPoints - 1; i++) {  // :552-562
        const double b = root_bcd[3 * i], c = root_bcd[3 * i + 1], d = root_bcd[3 * i + 2];
        if (c == 0.0 && d == 0.0) {
          Par[i][0] = m3_float_t(b); Par[i][1] = 0.0; Par[i][2] = 0.0;
        }
      }
    }
  } else if (InterPolation == 4) {  // kKochanekBartels :568-643, T = C = B = 0 (Catmull-Rom)
    std::vector<m3_float_t> Tangents(nPoints);
    for (int i = 0; i < nPoints; ++i) {
      XPos[i] = m3_float_t(sx[i]);
      YResp[i] = m3_float_t(sy[i]);
      Tangents[i] = 0.0;
    }
    constexpr m3_float_t T = 0.0, Cc = 0.0, B = 0.0;
    for (int i = 1; i < nPoints - 1; ++i) {
      m3_float_t d0 = (YResp[i] - YResp[i - 1]) / (XPos[i] - XPos[i - 1]);
      m3_float_t d1 = (YResp[i + 1] - YResp[i]) / (XPos[i + 1] - XPos[i]);
      m3_float_t term1 = (1.0 - T) * (1.0 + B) * (1.0 + Cc) * 0.5 * d0;
      m3_float_t term2 = (1.0 - T) * (1.0 - B) * (1.0 - Cc) * 0.5 * d1;
      Tangents[i] = term1 + term2;
    }
    Tangents[0] = (YResp[1] - YResp[0]) / (XPos[1] - XPos[0]);
    Tangents[nPoints - 1] = (YResp[nPoints - 1] - YResp[nPoints - 2]) / (XPos[nPoints - 1] - XPos[nPoints - 2]);
    for (int i = 0; i < nPoints - 1; ++i) {
      m3_float_t dx = XPos[i + 1] - XPos[i];
      m3_float_t dy = YResp[i + 1] - YResp[i];
      m3_float_t m0 = Tangents[i];
      m3_float_t m1 = Tangents[i + 1];
      Par[i][0] = m0;
      Par[i][1] = (3 * dy / (dx * dx)) - (2 * m0 + m1) / dx;
      Par[i][2] = (m0 + m1 - 2 * dy / dx) / (dx * dx);
    }
    Par[nPoints - 1][0] = 0.0; Par[nPoints - 1][1] = 0.0; Par[nPoints - 1][2] = 0.0;
  } else {
    return -1;
  }
  for (int i = 0; i < nPoints; ++i) {
    out[5 * i] = XPos[i]; out[5 * i + 1] = YResp[i];
    out[5 * i + 2] = Par[i][0]; out[5 * i + 3] = Par[i][1]; out[5 * i + 4] = Par[i][2];
  }
  return 0;
}

/* ------------------------------------------------------------------------- */
/* Spline monolith                                                            */
/* ------------------------------------------------------------------------- */

// ref: Splines/SplineStructs.h:21-44
struct FastSplineInfo_o {
  m3_int_t nPts = -999;
  std::vector<m3_float_t> xPts;
  m3_int_t CurrSegment = 0;
  const double* splineParsPointer = nullptr;
};

// Arrays that are written for the first time inside an OpenMP loop with the static schedule of the loops that later
// walk them, so that on a multi-socket host every page lands on the memory node of the thread that will use it
// (first touch).  The reference allocates on one thread; this only matters for the TIMED baseline, which should not be
// handicapped by a placement the GPU side does not suffer from (bench.py: cpu_baseline).  Serial build: plain arrays.
template <class T>
struct HomedArray {
  T* p = nullptr;
  uint64_t n = 0;
  HomedArray() = default;
  HomedArray(const HomedArray&) = delete;
  HomedArray& operator=(const HomedArray&) = delete;
  ~HomedArray() { release(); }
  void release() {
    if (!p) return;
#ifdef MULTITHREAD
#pragma omp parallel for schedule(static)
#endif
    for (uint64_t i = 0; i < n; ++i) p[i].~T();
    std::free(p);
    p = nullptr;
    n = 0;
  }
  template <class F>
  void build(uint64_t count, F&& make) {  // make(i) -> T, called on the thread that owns element i
    release();
    n = count;
    p = static_cast<T*>(std::malloc(sizeof(T) * (count ? count : 1)));
#ifdef MULTITHREAD
#pragma omp parallel for schedule(static)
#endif
    for (uint64_t i = 0; i < count; ++i) new (&p[i]) T(make(i));
  }
  void assign(uint64_t count, const T& v) { build(count, [&](uint64_t) { return v; }); }
  void copy_from(const T* src, uint64_t count) { build(count, [&](uint64_t i) { return src[i]; }); }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](uint64_t i) { return p[i]; }
  const T& operator[](uint64_t i) const { return p[i]; }
  uint64_t size() const { return n; }
};

struct m3o_monolith {
  uint64_t NEvents = 0;
  short nParams = 0;
  short max_knots = 0;
  uint64_t NSplines_valid = 0;
  uint64_t NTF1_valid = 0;
  // borrowed reference-layout arrays (ref: Splines/SplineCommon.h:30-50, SplineMonolith.h:106-137)
  const float* coeff_x = nullptr;
  const float* coeff_many = nullptr;
  const unsigned int* nKnots_arr = nullptr;
  const int16_t* paramNo_arr = nullptr;
  const float* coeff_TF1_many = nullptr;
  const int16_t* paramNo_TF1_arr = nullptr;
  const unsigned int* nParamPerEvent = nullptr;
  const unsigned int* nParamPerEvent_tf1 = nullptr;
  // owned
  std::vector<FastSplineInfo_o> SplineInfoArray;
  std::vector<short> SplineSegments;
  std::vector<float> ParamValues;
  std::vector<short> CurrSegState;  // mirror of SplineInfoArray[i].CurrSegment for inspection
  HomedArray<float> cpu_weights_spline_var;
  HomedArray<float> cpu_weights_tf1_var;
  HomedArray<float> cpu_total_weights;
  // m3o_monolith_localise: owned, first-touched copies of the borrowed arrays
  HomedArray<float> own_coeff_many, own_coeff_TF1_many;
  HomedArray<unsigned int> own_nKnots_arr, own_nParamPerEvent, own_nParamPerEvent_tf1;
  HomedArray<int16_t> own_paramNo_arr, own_paramNo_TF1_arr;
};

m3o_monolith* m3o_monolith_create(uint64_t n_events, int n_params, int max_knots,
                                  const float* coeff_x, const int16_t* npts, uint64_t n_splines,
                                  const float* coeff_many, const unsigned int* nknots_arr,
                                  const int16_t* param_no, uint64_t n_tf1, const float* coeff_tf1,
                                  const int16_t* param_no_tf1, const unsigned int* n_param_per_event,
                                  const unsigned int* n_param_per_event_tf1) {
  m3o_monolith* m = new m3o_monolith();
  m->NEvents = n_events;
  m->nParams = short(n_params);
  m->max_knots = short(max_knots);
  m->NSplines_valid = n_splines;
  m->NTF1_valid = n_tf1;
  m->coeff_x = coeff_x;
  m->coeff_many = coeff_many;
  m->nKnots_arr = nknots_arr;
  m->paramNo_arr = param_no;
  m->coeff_TF1_many = coeff_tf1;
  m->paramNo_TF1_arr = param_no_tf1;
  m->nParamPerEvent = n_param_per_event;
  m->nParamPerEvent_tf1 = n_param_per_event_tf1;
  // ref: SplineMonolith.cpp:387-401 -- xPts come from the first spline seen for the
  // parameter; here they are read back from coeff_x, which the reference fills with
  // the same knots (SplineMonolith.cpp:162-164).
  m->SplineInfoArray.resize(n_params);
  for (int i = 0; i < n_params; ++i) {
    FastSplineInfo_o& info = m->SplineInfoArray[i];
    if (npts[i] > 0) {
      info.nPts = npts[i];
      info.xPts.assign(coeff_x + size_t(i) * max_knots, coeff_x + size_t(i) * max_knots + npts[i]);
    }
  }
  // ref: SplineMonolith.cpp:85-93
  m->SplineSegments.assign(n_params, 0);
  m->ParamValues.assign(n_params, -999.f);
  m->CurrSegState.assign(n_params, 0);
  // ref: SplineMonolith.cpp:240-244
  m->cpu_total_weights.assign(n_events, 0.f);
  m->cpu_weights_spline_var.assign(n_splines, 0.f);
  m->cpu_weights_tf1_var.assign(n_tf1, 0.f);
  return m;
}

void m3o_monolith_destroy(m3o_monolith* m) { delete m; }

// Re-home the borrowed reference-layout arrays: copies them, inside OpenMP loops, into arrays the oracle owns (see
// HomedArray).  total_nknots = knots in coeff_many.  The values, the layout and every computation stay the same.
void m3o_monolith_localise(m3o_monolith* m, uint64_t total_nknots) {
  m->own_coeff_many.copy_from(m->coeff_many, 4 * total_nknots);
  m->coeff_many = m->own_coeff_many.data();
  m->own_nKnots_arr.copy_from(m->nKnots_arr, m->NSplines_valid);
  m->nKnots_arr = m->own_nKnots_arr.data();
  m->own_paramNo_arr.copy_from(m->paramNo_arr, m->NSplines_valid);
  m->paramNo_arr = m->own_paramNo_arr.data();
  m->own_coeff_TF1_many.copy_from(m->coeff_TF1_many, 2 * m->NTF1_valid);
  m->coeff_TF1_many = m->own_coeff_TF1_many.data();
  m->own_paramNo_TF1_arr.copy_from(m->paramNo_TF1_arr, m->NTF1_valid);
  m->paramNo_TF1_arr = m->own_paramNo_TF1_arr.data();
  m->own_nParamPerEvent.copy_from(m->nParamPerEvent, 2 * m->NEvents);
  m->nParamPerEvent = m->own_nParamPerEvent.data();
  m->own_nParamPerEvent_tf1.copy_from(m->nParamPerEvent_tf1, 2 * m->NEvents);
  m->nParamPerEvent_tf1 = m->own_nParamPerEvent_tf1.data();
}

// ref: Splines/SplineMonolith.h:44-46
void m3o_monolith_set_spline_pointers(m3o_monolith* m, const double* const* ptrs) {
  for (m3_int_t i = 0; i < m->nParams; ++i) m->SplineInfoArray[i].splineParsPointer = ptrs[i];
}

// ref: Splines/SplineBase.cpp:44-109
void m3o_monolith_find_spline_segment(m3o_monolith* m) {
  for (m3_int_t i = 0; i < m->nParams; ++i) {
    FastSplineInfo_o& info = m->SplineInfoArray[i];
    const m3_int_t nPoints = info.nPts;
    const std::vector<m3_float_t>& xArray = info.xPts;

    const float xvar = float(*info.splineParsPointer);  // ref: :54 double -> float
    m->ParamValues[i] = xvar;

    if (xArray.size() == 0) continue;  // ref: :60

    m3_int_t segment = 0;
    m3_int_t kHigh = m3_int_t(nPoints - 1);
    const m3_int_t PreviousSegment = info.CurrSegment;

    if (xvar <= xArray[0]) {
      segment = 0;
    } else if (xvar >= xArray[nPoints - 1]) {
      segment = kHigh;
    } else if (xArray[PreviousSegment + 1] > xvar && xvar >= xArray[PreviousSegment]) {  // ref: :76
      segment = PreviousSegment;
    } else {
      m3_int_t kHalf = 0;
      while (kHigh - segment > 1) {  // ref: :84-94
        kHalf = m3_int_t((segment + kHigh) / 2);
        if (xvar > xArray[kHalf]) segment = kHalf;
        else kHigh = kHalf;
      }
    }
    if (segment >= nPoints - 1 && nPoints > 1) segment = m3_int_t(nPoints - 2);  // ref: :97

    info.CurrSegment = segment;
    m->SplineSegments[i] = short(info.CurrSegment);
    m->CurrSegState[i] = segment;
  }
}

void m3o_monolith_set_vals_segments(m3o_monolith* m, const float* vals, const int16_t* segments) {
  for (int i = 0; i < m->nParams; ++i) {
    m->ParamValues[i] = vals[i];
    m->SplineSegments[i] = segments[i];
  }
}

// ref: Splines/SplineMonolith.cpp:727-788
void m3o_monolith_calc_spline_weights(m3o_monolith* m) {
  const uint64_t NSplines_valid = m->NSplines_valid;
  const uint64_t NTF1_valid = m->NTF1_valid;
  const short* SplineSegments = m->SplineSegments.data();
  const float* ParamValues = m->ParamValues.data();
  const short _max_knots = m->max_knots;
  float* cpu_weights_spline_var = m->cpu_weights_spline_var.data();
  float* cpu_weights_tf1_var = m->cpu_weights_tf1_var.data();
#ifdef MULTITHREAD
#pragma omp parallel
  {
#endif
#ifdef MULTITHREAD
#pragma omp for simd nowait
#endif
    for (uint64_t splineNum = 0; splineNum < NSplines_valid; ++splineNum) {
      const short Param = m->paramNo_arr[splineNum];
      const short segment = SplineSegments[Param];
      // the reference narrows this index to short (ref: :747); widened here so P*K > 32767 works
      const int segment_X = int(Param) * _max_knots + segment;
      // ref: :750 (unsigned int arithmetic, as in the reference)
      const unsigned int CurrentKnotPos = m->nKnots_arr[splineNum] * 4 + segment * 4;
      const float fY = m->coeff_many[CurrentKnotPos];
      const float fB = m->coeff_many[CurrentKnotPos + 1];
      const float fC = m->coeff_many[CurrentKnotPos + 2];
      const float fD = m->coeff_many[CurrentKnotPos + 3];
      const float dx = ParamValues[Param] - m->coeff_x[segment_X];
      cpu_weights_spline_var[splineNum] = fmaf(dx, fmaf(dx, fmaf(dx, fD, fC), fB), fY);  // ref: :762
    }
#ifdef MULTITHREAD
#pragma omp for simd
#endif
    for (uint64_t tf1Num = 0; tf1Num < NTF1_valid; ++tf1Num) {
      const float x = ParamValues[m->paramNo_TF1_arr[tf1Num]];
      const uint64_t TF1_Index = tf1Num * 2;  // _nTF1Coeff_ = 2 (ref: Splines/SplineCommon.h:15)
      const float a = m->coeff_TF1_many[TF1_Index];
      const float b = m->coeff_TF1_many[TF1_Index + 1];
      cpu_weights_tf1_var[tf1Num] = fmaf(a, x, b);  // ref: :779
    }
#ifdef MULTITHREAD
  }
#endif
}

// ref: Splines/SplineMonolith.cpp:792-830
void m3o_monolith_calc_total_event_weight(m3o_monolith* m) {
  const uint64_t NEvents = m->NEvents;
  const float* cpu_weights_spline_var = m->cpu_weights_spline_var.data();
  const float* cpu_weights_tf1_var = m->cpu_weights_tf1_var.data();
  float* cpu_total_weights = m->cpu_total_weights.data();
#ifdef MULTITHREAD
#pragma omp parallel for
#endif
  for (uint64_t EventNum = 0; EventNum < NEvents; ++EventNum) {
    float totalWeight = 1.0f;
    const uint64_t Offset = 2 * EventNum;
    const uint64_t startIndex = m->nParamPerEvent[Offset + 1];
    const uint64_t numParams = m->nParamPerEvent[Offset];
#ifdef MULTITHREAD
#pragma omp simd reduction(* : totalWeight)
#endif
    for (uint64_t id = 0; id < numParams; ++id) totalWeight *= cpu_weights_spline_var[startIndex + id];

    const uint64_t startIndex_tf1 = m->nParamPerEvent_tf1[Offset + 1];
    const uint64_t numParams_tf1 = m->nParamPerEvent_tf1[Offset];
#ifdef MULTITHREAD
#pragma omp simd reduction(* : totalWeight)
#endif
    for (uint64_t id = 0; id < numParams_tf1; ++id) totalWeight *= cpu_weights_tf1_var[startIndex_tf1 + id];

    cpu_total_weights[EventNum] = totalWeight;
  }
}

// ref: Splines/SplineMonolith.cpp:712-723
void m3o_monolith_evaluate(m3o_monolith* m) {
  m3o_monolith_find_spline_segment(m);
  m3o_monolith_calc_spline_weights(m);
  m3o_monolith_calc_total_event_weight(m);
}

const float* m3o_monolith_total_weights(const m3o_monolith* m) { return m->cpu_total_weights.data(); }
const float* m3o_monolith_spline_weights(const m3o_monolith* m) { return m->cpu_weights_spline_var.data(); }
const float* m3o_monolith_tf1_weights(const m3o_monolith* m) { return m->cpu_weights_tf1_var.data(); }
const int16_t* m3o_monolith_segments(const m3o_monolith* m) { return m->SplineSegments.data(); }
const float* m3o_monolith_param_values(const m3o_monolith* m) { return m->ParamValues.data(); }
int16_t* m3o_monolith_curr_segment_state(m3o_monolith* m) {
  // expose FastSplineInfo.CurrSegment so tests can seed/inspect the shortcut state (ref: SplineBase.cpp:73-78)
  for (int i = 0; i < m->nParams; ++i) m->CurrSegState[i] = m->SplineInfoArray[i].CurrSegment;
  return m->CurrSegState.data();
}

/* ------------------------------------------------------------------------- */
/* Binned spline handler                                                      */
/* ------------------------------------------------------------------------- */

// ref: Splines/BinnedSplineHandler.h:11-165 (the per-step members only)
struct m3o_binned {
  short nParams = 0;
  int MonolithSize = 0;
  const int* coeffindexvec = nullptr;             // [MonolithSize] first knot of the bin-spline
  const int* uniquesplinevec_Monolith = nullptr;  // [MonolithSize] unique spline parameter
  int nUnique = 0;
  const int* uniquecoeffindices = nullptr;        // [nUnique] the non-flat bin-splines
  const m3_float_t* xcoeff_arr = nullptr;         // [CoeffIndex]
  const m3_float_t* manycoeff_arr = nullptr;      // [CoeffIndex * 4]
  std::vector<FastSplineInfo_o> SplineInfoArray;
  std::vector<short> SplineSegments;
  std::vector<float> ParamValues;
  std::vector<m3_float_t> weightvec_Monolith;
};

int m3o_float_t_size(void) { return int(sizeof(m3o_float_t)); }

m3o_binned* m3o_binned_create(int n_params, int max_knots, const m3o_float_t* xpts, const int16_t* npts, int monolith_size,
                              const int* coeffindexvec, const int* uniquesplinevec, int n_unique,
                              const int* uniquecoeffindices, const m3o_float_t* xcoeff, const m3o_float_t* manycoeff) {
  m3o_binned* b = new m3o_binned();
  b->nParams = short(n_params);
  b->MonolithSize = monolith_size;
  b->coeffindexvec = coeffindexvec;
  b->uniquesplinevec_Monolith = uniquesplinevec;
  b->nUnique = n_unique;
  b->uniquecoeffindices = uniquecoeffindices;
  b->xcoeff_arr = xcoeff;
  b->manycoeff_arr = manycoeff;
  b->SplineInfoArray.resize(n_params);
  for (int i = 0; i < n_params; ++i)
    if (npts[i] > 0) {
      b->SplineInfoArray[i].nPts = npts[i];
      b->SplineInfoArray[i].xPts.assign(xpts + size_t(i) * max_knots, xpts + size_t(i) * max_knots + npts[i]);
    }
  b->SplineSegments.assign(n_params, 0);
  b->ParamValues.assign(n_params, -999.f);
  b->weightvec_Monolith.assign(monolith_size, 1.0f);  // ref: BinnedSplineHandler.cpp:236 (flat splines stay 1)
  return b;
}
void m3o_binned_destroy(m3o_binned* b) { delete b; }
void m3o_binned_set_spline_pointers(m3o_binned* b, const double* const* ptrs) {
  for (int i = 0; i < b->nParams; ++i) b->SplineInfoArray[i].splineParsPointer = ptrs[i];
}

// ref: Splines/BinnedSplineHandler.cpp:295-341 (Evaluate = FindSplineSegment + CalcSplineWeights)
void m3o_binned_evaluate(m3o_binned* b) {
  // FindSplineSegment: the SplineBase routine over this handler's unique parameters
  for (m3_int_t i = 0; i < b->nParams; ++i) {
    FastSplineInfo_o& info = b->SplineInfoArray[i];
    const m3_int_t nPoints = info.nPts;
    const std::vector<m3_float_t>& xArray = info.xPts;
    const float xvar = float(*info.splineParsPointer);
    b->ParamValues[i] = xvar;
    if (xArray.size() == 0) continue;
    m3_int_t segment = 0;
    m3_int_t kHigh = m3_int_t(nPoints - 1);
    const m3_int_t PreviousSegment = info.CurrSegment;
    if (xvar <= xArray[0]) segment = 0;
    else if (xvar >= xArray[nPoints - 1]) segment = kHigh;
    else if (xArray[PreviousSegment + 1] > xvar && xvar >= xArray[PreviousSegment]) segment = PreviousSegment;
    else {
      while (kHigh - segment > 1) {
        const m3_int_t kHalf = m3_int_t((segment + kHigh) / 2);
        if (xvar > xArray[kHalf]) segment = kHalf;
        else kHigh = kHalf;
      }
    }
    if (segment >= nPoints - 1 && nPoints > 1) segment = m3_int_t(nPoints - 2);
    info.CurrSegment = segment;
    b->SplineSegments[i] = short(segment);
  }
  const size_t n = size_t(b->nUnique);
#ifdef MULTITHREAD
#pragma omp parallel for simd
#endif
  for (size_t iCoeff = 0; iCoeff < n; ++iCoeff) {
    const int iSpline = b->uniquecoeffindices[iCoeff];
    const short uniqueIndex = short(b->uniquesplinevec_Monolith[iSpline]);
    const short currentsegment = short(b->SplineSegments[uniqueIndex]);
    const int segCoeff = b->coeffindexvec[iSpline] + currentsegment;
    const int coeffOffset = segCoeff * 4;
    const m3_float_t y = b->manycoeff_arr[coeffOffset + 0];
    const m3_float_t bb = b->manycoeff_arr[coeffOffset + 1];
    const m3_float_t c = b->manycoeff_arr[coeffOffset + 2];
    const m3_float_t d = b->manycoeff_arr[coeffOffset + 3];
    const m3_float_t xvar = m3_float_t(*b->SplineInfoArray[uniqueIndex].splineParsPointer);  // ref: :327
    const m3_float_t dx = xvar - b->xcoeff_arr[segCoeff];
    m3_float_t weight = m3_fmaf_t(dx, m3_fmaf_t(dx, m3_fmaf_t(dx, d, c), bb), y);  // M3::fmaf_t
    if (weight < 0) weight = 0.;  // ref: :337
    b->weightvec_Monolith[iSpline] = weight;
  }
}
const m3o_float_t* m3o_binned_weights(const m3o_binned* b) { return b->weightvec_Monolith.data(); }
const int16_t* m3o_binned_segments(const m3o_binned* b) { return b->SplineSegments.data(); }

/* ------------------------------------------------------------------------- */
/* Binning                                                                    */
/* ------------------------------------------------------------------------- */

// ref: Samples/SampleStructs.h:172-184
struct BinShiftLookup_o {
  double lower_binedge, lower_lower_binedge, upper_binedge, upper_upper_binedge;
};

// ref: Samples/SampleStructs.h:186-220 -- note the (lo, hi] test
struct BinInfo_o {
  std::vector<std::array<double, 2>> Extent;
  bool IsEventInside(const std::vector<const double*>& KinVars) const {
    bool inside = true;
    const size_t N = KinVars.size();
    for (size_t i = 0; i < N; ++i) {
      const double Var = *KinVars[i];
      const bool in_bin = (Var > Extent[i][0]) & (Var <= Extent[i][1]);
      inside &= in_bin;
    }
    return inside;
  }
};

// ref: Samples/SampleStructs.h:232-675
struct SampleBinningInfo_o {
  std::vector<std::vector<double>> BinEdges;
  std::vector<int> AxisNBins;
  int nBins = -999;
  int GlobalOffset = -999;
  std::vector<std::vector<BinShiftLookup_o>> BinLookup;
  std::vector<int> Strides;
  bool Uniform = true;
  std::vector<BinInfo_o> Bins;
  std::vector<std::vector<int>> BinGridMapping;

  // ref: :648-659
  void InitialiseStrides(int Dimension) {
    Strides.resize(Dimension);
    int stride = 1;
    for (int i = 0; i < Dimension; ++i) {
      Strides[i] = stride;
      stride *= AxisNBins[i];
    }
  }
  // ref: :615-646
  void InitialiseLookUpSingleDimension(std::vector<BinShiftLookup_o>& Bin_Lookup,
                                       const std::vector<double>& Bin_Edges, int TotBins) {
    Bin_Lookup.resize(TotBins);
    for (int bin_i = 0; bin_i < TotBins; bin_i++) {
      double low_lower_edge = -999999.123456;  // M3::_DEFAULT_RETURN_VAL_
      const double low_edge = Bin_Edges[bin_i];
      const double upper_edge = Bin_Edges[bin_i + 1];
      double upper_upper_edge = -999999.123456;
      low_lower_edge = (bin_i == 0) ? Bin_Edges[0] : Bin_Edges[bin_i - 1];
      if (bin_i + 2 < TotBins) upper_upper_edge = Bin_Edges[bin_i + 2];
      else if (bin_i + 1 < TotBins) upper_upper_edge = Bin_Edges[bin_i + 1];
      Bin_Lookup[bin_i].lower_binedge = low_edge;
      Bin_Lookup[bin_i].upper_binedge = upper_edge;
      Bin_Lookup[bin_i].lower_lower_binedge = low_lower_edge;
      Bin_Lookup[bin_i].upper_upper_binedge = upper_upper_edge;
    }
  }
  // ref: :663-670
  void InitialiseBinMigrationLookUp(int Dimension) {
    BinLookup.resize(Dimension);
    for (int i = 0; i < Dimension; ++i) InitialiseLookUpSingleDimension(BinLookup[i], BinEdges[i], AxisNBins[i]);
    InitialiseStrides(Dimension);
  }
  // ref: :250-287
  bool InitUniform(const std::vector<std::vector<double>>& InputEdges) {
    BinEdges = InputEdges;
    Uniform = true;
    AxisNBins.resize(BinEdges.size());
    nBins = 1;
    for (size_t iDim = 0; iDim < BinEdges.size(); iDim++) {
      const auto& Edges = BinEdges[iDim];
      if (!std::is_sorted(Edges.begin(), Edges.end())) return false;
      if (Edges.size() == 0) return false;
      AxisNBins[iDim] = int(Edges.size()) - 1;
      nBins *= AxisNBins[iDim];
    }
    GlobalOffset = 0;
    InitialiseBinMigrationLookUp(int(BinEdges.size()));
    return true;
  }
  // ref: :397-462
  void InitialiseGridMapping() {
    const size_t Dim = BinEdges.size();
    int NGridBins = 1;
    for (size_t d = 0; d < Dim; ++d) NGridBins *= AxisNBins[d];
    BinGridMapping.assign(NGridBins, std::vector<int>());
    std::vector<int> MultiIndex(Dim, 0);
    std::function<void(size_t, int)> scan = [&](size_t d, int LinearIndex) {
      if (d == Dim) {
        for (size_t iBin = 0; iBin < Bins.size(); ++iBin) {
          bool overlap = true;
          for (size_t i = 0; i < Dim; ++i) {
            const double a_lo = Bins[iBin].Extent[i][0], a_hi = Bins[iBin].Extent[i][1];
            const double b_lo = BinEdges[i][MultiIndex[i]], b_hi = BinEdges[i][MultiIndex[i] + 1];
            if (!(a_hi > b_lo && a_lo < b_hi)) { overlap = false; break; }
          }
          if (overlap) BinGridMapping[LinearIndex].push_back(int(iBin));
        }
        return;
      }
      for (int i = 0; i < AxisNBins[d]; ++i) {
        MultiIndex[d] = i;
        int NewLinearIndex = LinearIndex;
        if (d > 0) {
          int stride = 1;
          for (size_t s = 0; s < d; ++s) stride *= AxisNBins[s];
          NewLinearIndex += i * stride;
        } else {
          NewLinearIndex = i;
        }
        scan(d + 1, NewLinearIndex);
      }
    };
    scan(0, 0);
  }
  // ref: :465-530 (overlap / gap validation :290-394 is setup-time diagnostics, omitted)
  bool InitNonUniform(int nbins, int ndim, const double* extents) {
    Uniform = false;
    nBins = 0;
    if (ndim <= 1 || nbins <= 0) return false;  // ref: :471-474
    Bins.resize(nbins);
    for (int iBin = 0; iBin < nbins; ++iBin) {
      BinInfo_o NewBin;
      for (int d = 0; d < ndim; ++d)
        NewBin.Extent.push_back({extents[(size_t(iBin) * ndim + d) * 2], extents[(size_t(iBin) * ndim + d) * 2 + 1]});
      Bins[iBin] = std::move(NewBin);
      nBins++;
    }
    const int BinsPerDimension = 10;  // ref: :499
    BinEdges.resize(ndim);
    AxisNBins.resize(ndim);
    for (int iDim = 0; iDim < ndim; iDim++) {
      double MinVal = std::numeric_limits<double>::max();
      double MaxVal = std::numeric_limits<double>::lowest();
      for (const auto& bin : Bins) {
        MinVal = std::min(MinVal, bin.Extent[iDim][0]);
        MaxVal = std::max(MaxVal, bin.Extent[iDim][1]);
      }
      BinEdges[iDim].resize(BinsPerDimension + 1);
      const double BinWidth = (MaxVal - MinVal) / double(BinsPerDimension);
      for (int iEdge = 0; iEdge <= BinsPerDimension; iEdge++) BinEdges[iDim][iEdge] = MinVal + double(iEdge) * BinWidth;
      AxisNBins[iDim] = BinsPerDimension;
    }
    GlobalOffset = 0;
    InitialiseBinMigrationLookUp(ndim);
    InitialiseGridMapping();
    return true;
  }
  // ref: :577-613
  int FindBin(double KinVar, int NomBin, int N_Bins, const std::vector<double>& Bin_Edges,
              const std::vector<BinShiftLookup_o>& Bin_Lookup) const {
    if (KinVar < Bin_Edges[0] || KinVar >= Bin_Edges[N_Bins]) return M3O_UNDER_OVERFLOW_BIN;
    if (NomBin > M3O_UNDER_OVERFLOW_BIN) {
      const BinShiftLookup_o& Bin = Bin_Lookup[NomBin];
      const double lower = Bin.lower_binedge, upper = Bin.upper_binedge;
      const double lower_lower = Bin.lower_lower_binedge, upper_upper = Bin.upper_upper_binedge;
      if (KinVar < upper && KinVar >= lower) return NomBin;
      if (KinVar < lower && KinVar >= lower_lower) return NomBin - 1;
      if (KinVar < upper_upper && KinVar >= upper) return NomBin + 1;
    }
    return int(std::distance(Bin_Edges.begin(), std::upper_bound(Bin_Edges.begin(), Bin_Edges.end(), KinVar)) - 1);
  }
  int FindBin(int Dimension, double Var, int NomBin) const {  // ref: :559-561
    return FindBin(Var, NomBin, AxisNBins[Dimension], BinEdges[Dimension], BinLookup[Dimension]);
  }
};

struct m3o_binning {
  std::vector<SampleBinningInfo_o> SampleBinning;
  int TotalNumberOfBins = 0;
  // ref: Samples/BinningHandler.cpp:341-355
  void SetGlobalBinNumbers() {
    int GlobalOffsetCounter = 0;
    for (size_t iSample = 0; iSample < SampleBinning.size(); iSample++) {
      SampleBinning[iSample].GlobalOffset = GlobalOffsetCounter;
      GlobalOffsetCounter += SampleBinning[iSample].nBins;
    }
    TotalNumberOfBins = GlobalOffsetCounter;
  }
  // ref: Samples/BinningHandler.cpp:257-291
  int FindGlobalBin(int NomSample, const std::vector<const double*>& KinVar, const std::vector<int>& NomBin) const {
    const int Dim = int(KinVar.size());
    const SampleBinningInfo_o& Binning = SampleBinning[NomSample];
    int GlobalBin = 0;
    for (int i = 0; i < Dim; ++i) {
      const double Var = *KinVar[i];
      const int Bin = Binning.FindBin(i, Var, NomBin[i]);
      if (Bin < 0) return M3O_UNDER_OVERFLOW_BIN;
      GlobalBin += Bin * Binning.Strides[i];
    }
    if (Binning.Uniform) {
      GlobalBin += Binning.GlobalOffset;
      return GlobalBin;
    } else {
      const auto& BinMapping = Binning.BinGridMapping[GlobalBin];
      for (size_t iBin = 0; iBin < BinMapping.size(); iBin++) {
        const int BinNumber = BinMapping[iBin];
        if (Binning.Bins[BinNumber].IsEventInside(KinVar)) return BinNumber + Binning.GlobalOffset;
      }
      return M3O_UNDER_OVERFLOW_BIN;
    }
  }
  // ref: Samples/BinningHandler.cpp:294-307
  int FindNominalBin(int iSample, int iDim, double Var) const {
    const auto& edges = SampleBinning[iSample].BinEdges[iDim];
    if (Var < edges.front() || Var >= edges.back()) return M3O_UNDER_OVERFLOW_BIN;
    return int(std::distance(edges.begin(), std::upper_bound(edges.begin(), edges.end(), Var)) - 1);
  }
};

m3o_binning* m3o_binning_create(void) { return new m3o_binning(); }
void m3o_binning_destroy(m3o_binning* b) { delete b; }

int m3o_binning_add_uniform(m3o_binning* b, int ndim, const int* n_edges, const double* edges) {
  std::vector<std::vector<double>> in(ndim);
  size_t off = 0;
  for (int d = 0; d < ndim; ++d) {
    in[d].assign(edges + off, edges + off + n_edges[d]);
    off += n_edges[d];
  }
  SampleBinningInfo_o s;
  if (!s.InitUniform(in)) return -1;
  b->SampleBinning.emplace_back(std::move(s));
  b->SetGlobalBinNumbers();
  return int(b->SampleBinning.size()) - 1;
}

int m3o_binning_add_nonuniform(m3o_binning* b, int nbins, int ndim, const double* extents) {
  SampleBinningInfo_o s;
  if (!s.InitNonUniform(nbins, ndim, extents)) return -1;
  b->SampleBinning.emplace_back(std::move(s));
  b->SetGlobalBinNumbers();
  return int(b->SampleBinning.size()) - 1;
}

int m3o_binning_n_samples(const m3o_binning* b) { return int(b->SampleBinning.size()); }
int m3o_binning_n_bins(const m3o_binning* b) { return b->TotalNumberOfBins; }
int m3o_binning_sample_n_bins(const m3o_binning* b, int s) { return b->SampleBinning[s].nBins; }
int m3o_binning_sample_start(const m3o_binning* b, int s) { return b->SampleBinning[s].GlobalOffset; }
int m3o_binning_sample_end(const m3o_binning* b, int s) {  // ref: BinningHandler.cpp:329-337
  if (s == int(b->SampleBinning.size()) - 1) return b->TotalNumberOfBins;
  return b->SampleBinning[s + 1].GlobalOffset;
}
int m3o_binning_n_dim(const m3o_binning* b, int s) { return int(b->SampleBinning[s].BinEdges.size()); }
int m3o_binning_find_nominal_bin(const m3o_binning* b, int s, int dim, double var) { return b->FindNominalBin(s, dim, var); }
int m3o_binning_find_bin(const m3o_binning* b, int s, int dim, double var, int nom_bin) {
  return b->SampleBinning[s].FindBin(dim, var, nom_bin);
}
int m3o_binning_find_global_bin(const m3o_binning* b, int s, const double* kinvar, const int* nom_bin) {
  const int ndim = int(b->SampleBinning[s].BinEdges.size());
  std::vector<const double*> kv(ndim);
  std::vector<int> nb(ndim);
  for (int d = 0; d < ndim; ++d) { kv[d] = &kinvar[d]; nb[d] = nom_bin[d]; }
  return b->FindGlobalBin(s, kv, nb);
}

/* ------------------------------------------------------------------------- */
/* Sample handler                                                             */
/* ------------------------------------------------------------------------- */

// ref: Samples/FarDetectorCoreInfoStruct.h:82-126 (fields used on the hot path only)
struct EventInfo_o {
  std::vector<const double*> norm_pointers;
  std::vector<const m3_float_t*> total_weight_pointers;
  std::vector<const double*> KinVar;
  std::vector<int> NomBin;
  int NominalSample = -999;
};

// closed-form stand-ins for the experiment's functional-parameter callbacks
// (ref: Samples/SampleHandlerFD.cpp:545-564 ApplyShifts; funcParsGrid holds std::function objects)
struct FuncShift_o {
  int sample;  // -1: every sample
  int dim;
  int kind;    // 1: x *= (1 + theta)   2: x += theta
  const double* valuePtr;
};

// ref: Samples/SampleStructs.h KinematicCut {ParamToCutOnIt, LowerBound, UpperBound}
struct KinematicCut_o {
  int sample;  // -1: every sample
  int column;
  double LowerBound, UpperBound;
};

struct m3o_sample {
  std::vector<KinematicCut_o> Selection;
  HomedArray<double> kin_nominal, kin_shifted;  // KinVar pointers alias kin_shifted
  int kin_stride = 1;
  std::vector<FuncShift_o> shifts;
  const m3o_binning* Binning = nullptr;
  m3o_monolith* SplineHandler = nullptr;
  m3o_binned* BinnedHandler = nullptr;
  HomedArray<EventInfo_o> MCSamples;
  std::vector<double> SampleHandlerFD_array, SampleHandlerFD_array_w2, SampleHandlerFD_data;
  int fTestStatistic = M3O_kPoisson;
  bool FirstTimeW2 = true;  // ref: SampleHandlerFD.cpp:28-30
  bool UpdateW2 = false;    // ref: SampleHandlerFD.cpp:62-65
  double t_spline = 0, t_fill = 0;
};

m3o_sample* m3o_sample_create(const m3o_binning* b, uint64_t n_events, const int32_t* sample_id,
                              int kin_stride, const double* kin, const uint64_t* norm_off,
                              const int16_t* norm_idx, const double* par_base, const int32_t* osc_idx,
                              const m3o_float_t* osc_table, m3o_monolith* mono) {
#ifdef M3O_FLOAT_T_DOUBLE
  if (mono) return nullptr;  // SMonolith's float weights cannot sit behind const M3::float_t* in this build
#endif
  m3o_sample* s = new m3o_sample();
  s->Binning = b;
  s->SplineHandler = mono;
  s->kin_stride = kin_stride;
  s->kin_nominal.copy_from(kin, n_events * uint64_t(kin_stride));
  s->kin_shifted.copy_from(kin, n_events * uint64_t(kin_stride));
  // every event is set up by the thread that will fill it (its pointer vectors live on that thread's memory node)
  s->MCSamples.build(n_events, [&](uint64_t e) {
    EventInfo_o ev;
    ev.NominalSample = sample_id ? sample_id[e] : 0;
    const int ndim = int(b->SampleBinning[ev.NominalSample].BinEdges.size());
    ev.KinVar.resize(ndim);
    ev.NomBin.resize(ndim);
    for (int d = 0; d < ndim; ++d) {
      ev.KinVar[d] = &s->kin_shifted[e * kin_stride + d];
      // ref: SampleHandlerFD.cpp:757-887 FindNominalBinAndEdges -> BinningHandler::FindNominalBin
      ev.NomBin[d] = b->FindNominalBin(ev.NominalSample, d, *ev.KinVar[d]);
    }
    // ref: SampleHandlerFD.cpp:637-663 SetupNormParameters
    if (norm_off)
      for (uint64_t k = norm_off[e]; k < norm_off[e + 1]; ++k) ev.norm_pointers.push_back(&par_base[norm_idx[k]]);
    // ref: SampleHandlerFD.cpp:1108-1122 osc pointer first (SetupOscParameters runs before SetupSplines)
    if (osc_idx && osc_idx[e] >= 0) ev.total_weight_pointers.push_back(&osc_table[osc_idx[e]]);
    // ref: SampleHandlerFD.cpp:1244-1249 then the monolith's total weight
#ifndef M3O_FLOAT_T_DOUBLE
    if (mono) ev.total_weight_pointers.push_back(&mono->cpu_total_weights[e]);
#endif
    return ev;
  });
  const int nb = b->TotalNumberOfBins;
  s->SampleHandlerFD_array.assign(nb, 0.);
  s->SampleHandlerFD_array_w2.assign(nb, 0.);
  s->SampleHandlerFD_data.assign(nb, 0.);
  (void)kUnity;
  return s;
}

void m3o_sample_destroy(m3o_sample* s) { delete s; }

void m3o_sample_set_test_statistic(m3o_sample* s, int kind, int update_w2) {
  s->fTestStatistic = kind;
  s->UpdateW2 = update_w2 != 0;
}

// ref: Samples/SampleHandlerFD.cpp:1027-1044
void m3o_sample_add_data(m3o_sample* s, const double* data) {
  s->SampleHandlerFD_data.assign(data, data + s->Binning->TotalNumberOfBins);
}

// ref: Samples/SampleHandlerFD.cpp:568-594
static m3_float_t CalcWeightTotal(const EventInfo_o* MCEvent) {
  m3_float_t TotalWeight = 1.0;
  const int nNorms = int(MCEvent->norm_pointers.size());
#ifdef MULTITHREAD
#pragma omp simd reduction(* : TotalWeight)
#endif
  for (int iParam = 0; iParam < nNorms; ++iParam) TotalWeight *= static_cast<m3_float_t>(*(MCEvent->norm_pointers[iParam]));
  const int TotalWeights = int(MCEvent->total_weight_pointers.size());
#ifdef MULTITHREAD
#pragma omp simd reduction(* : TotalWeight)
#endif
  for (int iWeight = 0; iWeight < TotalWeights; ++iWeight) TotalWeight *= *(MCEvent->total_weight_pointers[iWeight]);
  return TotalWeight;
}

// ref: Samples/SampleHandlerFD.cpp:545-564
static void ApplyShifts(m3o_sample* s, uint64_t iEvent) {
  const size_t nShifts = s->shifts.size();
  if (nShifts == 0) return;  // :550
  EventInfo_o& ev = s->MCSamples[iEvent];
  // ResetShifts: back to the nominal values
  for (int d = 0; d < s->kin_stride; ++d) s->kin_shifted[iEvent * s->kin_stride + d] = s->kin_nominal[iEvent * s->kin_stride + d];
  for (size_t i = 0; i < nShifts; ++i) {
    const FuncShift_o& fp = s->shifts[i];
    if (fp.dim >= s->kin_stride || (fp.sample >= 0 && fp.sample != ev.NominalSample)) continue;
    double& x = s->kin_shifted[iEvent * s->kin_stride + fp.dim];
    const double th = *fp.valuePtr;
    const int kind = fp.kind & 0xFF, aux = fp.kind >> 8;  // kind 3: the callback adds theta times another (nominal) variable
    x = kind == 1 ? x * (1.0 + th) : (kind == 2 ? x + th : x + th * s->kin_nominal[iEvent * s->kin_stride + aux]);
  }
}

// ref: Samples/SampleHandlerFD.cpp:281-294
static bool IsEventSelected(const m3o_sample* s, int iSample, uint64_t iEvent) {
  for (const KinematicCut_o& Cut : s->Selection) {
    if (Cut.sample >= 0 && Cut.sample != iSample) continue;
    if (Cut.column >= s->kin_stride) continue;
    const double Val = s->kin_shifted[iEvent * s->kin_stride + Cut.column];  // ReturnKinematicParameter
    if ((Val < Cut.LowerBound) || (Val >= Cut.UpperBound)) return false;
  }
  return true;
}

void m3o_sample_add_cut(m3o_sample* s, int sample, int column, double lower, double upper) {
  s->Selection.push_back({sample, column, lower, upper});
}

void m3o_sample_add_shift(m3o_sample* s, int sample, int dim, int kind, const double* value_ptr) {
  s->shifts.push_back({sample, dim, kind, value_ptr});
}

// ref: Samples/SampleHandlerFD.cpp:454-463
static void ResetHistograms(m3o_sample* s) {
  const int nBins = s->Binning->TotalNumberOfBins;
  std::fill_n(s->SampleHandlerFD_array.begin(), nBins, 0.0);
  if (s->FirstTimeW2) std::fill_n(s->SampleHandlerFD_array_w2.begin(), nBins, 0.0);
}

// ref: Samples/SampleHandlerFD.cpp:352-383 (serial) / :390-448 (FillArray_MP)
// ApplyShifts (no functional parameters -> early return :550), IsEventSelected (no cuts)
// and CalcWeightFunc (default no-op, SampleHandlerFD.h:289) are empty in the synthetic configs.
static void FillArray(m3o_sample* s) {
  const uint64_t NumberOfEvents = s->MCSamples.size();
  double* MC_Array_for_reduction = s->SampleHandlerFD_array.data();
  double* W2_array_for_reduction = s->SampleHandlerFD_array_w2.data();
  const bool FirstTimeW2 = s->FirstTimeW2;
#ifdef MULTITHREAD
  const int TotalBins = s->Binning->TotalNumberOfBins;
#pragma omp parallel for reduction(+ : MC_Array_for_reduction[:TotalBins], W2_array_for_reduction[:TotalBins])
#endif
  for (uint64_t iEvent = 0; iEvent < NumberOfEvents; ++iEvent) {
    ApplyShifts(s, iEvent);  // ref: :419, before the selection and the weight
    const EventInfo_o* MCEvent = &s->MCSamples[iEvent];
    if (!IsEventSelected(s, MCEvent->NominalSample, iEvent)) continue;  // ref: :423
    const m3_float_t totalweight = CalcWeightTotal(MCEvent);
    if (totalweight <= 0.) continue;  // ref: :432
    const int GlobalBin = s->Binning->FindGlobalBin(MCEvent->NominalSample, MCEvent->KinVar, MCEvent->NomBin);
    if (GlobalBin > M3O_UNDER_OVERFLOW_BIN) {  // ref: :443-446
      MC_Array_for_reduction[GlobalBin] += totalweight;
      if (FirstTimeW2) W2_array_for_reduction[GlobalBin] += totalweight * totalweight;  // float product, then widened
    }
  }
}

// ref: Samples/SampleHandlerFD.cpp:1196-1243 (SetSplinePointers, BinnedSplineHandler branch): the
// event's bin-spline weights are appended to total_weight_pointers, after the osc pointer
void m3o_sample_set_binned(m3o_sample* s, m3o_binned* b, const uint64_t* w_off, const int32_t* w_idx) {
  s->BinnedHandler = b;
  for (size_t e = 0; e < s->MCSamples.size(); ++e)
    for (uint64_t k = w_off[e]; k < w_off[e + 1]; ++k)
      s->MCSamples[e].total_weight_pointers.push_back(&b->weightvec_Monolith[w_idx[k]]);
}

// ref: Samples/SampleHandlerFD.cpp:196 AddAdditionalWeightPointers (experiment hook): further pointers appended to
// total_weight_pointers, behind oscillation and splines; what they point at may be rewritten between steps (CalcWeightFunc,
// SampleHandlerFD.h:289).  idx == NULL: event e points at base[e]; else at base[idx[e]], or gets no pointer when idx[e] < 0.
void m3o_sample_add_weight_pointers(m3o_sample* s, const m3o_float_t* base, const int32_t* idx) {
  for (size_t e = 0; e < s->MCSamples.size(); ++e) {
    if (!idx) s->MCSamples[e].total_weight_pointers.push_back(&base[e]);
    else if (idx[e] >= 0) s->MCSamples[e].total_weight_pointers.push_back(&base[idx[e]]);
  }
}

// ref: Samples/SampleHandlerFD.cpp:316-343
void m3o_sample_reweight(m3o_sample* s) {
  ResetHistograms(s);
  // Oscillator->Evaluate(): probabilities are pre-staged in osc_table (BASELINE configs)
  const double t0 = now_s();
  if (s->SplineHandler) m3o_monolith_evaluate(s->SplineHandler);
  if (s->BinnedHandler) m3o_binned_evaluate(s->BinnedHandler);
  const double t1 = now_s();
  FillArray(s);
  const double t2 = now_s();
  s->t_spline = t1 - t0;
  s->t_fill = t2 - t1;
  if (!s->UpdateW2) s->FirstTimeW2 = false;  // ref: :342
}

// ref: Samples/SampleHandlerFD.cpp:1284-1300
double m3o_sample_get_likelihood(const m3o_sample* s) {
  double negLogL = 0.;
  const int nb = s->Binning->TotalNumberOfBins;
#ifdef MULTITHREAD
#pragma omp parallel for reduction(+ : negLogL)
#endif
  for (int idx = 0; idx < nb; ++idx)
    negLogL += m3o_test_stat_llh(s->fTestStatistic, s->SampleHandlerFD_data[idx], s->SampleHandlerFD_array[idx],
                                 s->SampleHandlerFD_array_w2[idx]);
  return negLogL;
}

// ref: Samples/SampleHandlerFD.cpp:1262-1281
double m3o_sample_get_sample_likelihood(const m3o_sample* s, int isample) {
  const int Start = m3o_binning_sample_start(s->Binning, isample);
  const int End = m3o_binning_sample_end(s->Binning, isample);
  double negLogL = 0.;
#ifdef MULTITHREAD
#pragma omp parallel for reduction(+ : negLogL)
#endif
  for (int idx = Start; idx < End; ++idx)
    negLogL += m3o_test_stat_llh(s->fTestStatistic, s->SampleHandlerFD_data[idx], s->SampleHandlerFD_array[idx],
                                 s->SampleHandlerFD_array_w2[idx]);
  return negLogL;
}

const double* m3o_sample_mc(const m3o_sample* s) { return s->SampleHandlerFD_array.data(); }
const double* m3o_sample_w2(const m3o_sample* s) { return s->SampleHandlerFD_array_w2.data(); }
const double* m3o_sample_data(const m3o_sample* s) { return s->SampleHandlerFD_data.data(); }

void m3o_sample_event_weights(const m3o_sample* s, m3o_float_t* out) {
  for (size_t e = 0; e < s->MCSamples.size(); ++e) out[e] = CalcWeightTotal(&s->MCSamples[e]);
}
void m3o_sample_event_bins(const m3o_sample* s, int32_t* out) {
  for (size_t e = 0; e < s->MCSamples.size(); ++e) {
    ApplyShifts(const_cast<m3o_sample*>(s), e);
    if (!IsEventSelected(s, s->MCSamples[e].NominalSample, e)) { out[e] = M3O_UNDER_OVERFLOW_BIN; continue; }
    const EventInfo_o& ev = s->MCSamples[e];
    out[e] = s->Binning->FindGlobalBin(ev.NominalSample, ev.KinVar, ev.NomBin);
  }
}
void m3o_sample_last_timing(const m3o_sample* s, double* t_spline, double* t_fill) {
  *t_spline = s->t_spline;
  *t_fill = s->t_fill;
}

int m3o_is_multithread(void) {
#ifdef MULTITHREAD
  return 1;
#else
  return 0;
#endif
}
void m3o_set_num_threads(int n) {
#ifdef MULTITHREAD
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int m3o_max_threads(void) {
#ifdef MULTITHREAD
  return omp_get_max_threads();
#else
  return 1;
#endif
}
