// SplineStructs.h -- reduced response functions, ROOT-free.
//
// Same class names and accessors as the reference's Splines/SplineStructs.h so that code
// that builds a MasterSpline (std::vector<std::vector<TResponseFunction_red*>>) for SMonolith
// compiles unchanged:
//   TResponseFunction_red (:131-145), TF1_red (:148-262), TSpline3_red (:265-308, 673-734),
//   FastSplineInfo (:21-44), enum RespFuncType / SplineInterpolation (Parameters/ParameterStructs.h).
// What is NOT here: the constructors taking ROOT's TSpline3* / TF1* (ROOT is not a
// dependency of this engine).  Knots + coefficients come in through the (X, Y, N, P)
// constructor the reference already has (:285-308), or are built from knots alone by
// SplineBuilders.h.
#pragma once

#include <vector>

#include "Core.h"

/// Parameters/ParameterStructs.h: which reduced type a spline parameter uses
enum RespFuncType { kTSpline3_red, kTF1_red, kRespFuncTypes };

/// Parameters/ParameterStructs.h: interpolation schemes for TSpline3_red coefficients
enum SplineInterpolation { kTSpline3, kLinear, kMonotonic, kAkima, kKochanekBartels, kLinearFunc, kSplineInterpolations };

/// ref: Splines/SplineStructs.h:21-44
struct FastSplineInfo {
  M3::int_t nPts = -999;
  std::vector<M3::float_t> xPts;
  M3::int_t CurrSegment = 0;
  const double* splineParsPointer = nullptr;
};

/// ref: Splines/SplineStructs.h:131-145
class TResponseFunction_red {
 public:
  virtual ~TResponseFunction_red() {}
  virtual double Eval(const double var) = 0;
  virtual void Print() = 0;
  virtual M3::int_t GetNp() = 0;
};

/// ref: Splines/SplineStructs.h:148-262.  Linear response Par[1] + Par[0]*x.
class TF1_red : public TResponseFunction_red {
 public:
  TF1_red() {}
  /// deep copy of nSize parameters (the reference's version of this constructor writes through
  /// an unallocated pointer, :166-171; this one allocates)
  TF1_red(M3::int_t nSize, const M3::float_t* Array) : Par(Array, Array + nSize) {}
  void SetSize(M3::int_t nSpline) { Par.assign(nSpline, 0.f); }
  int GetSize() { return int(Par.size()); }
  void SetParameter(M3::int_t Parameter, M3::float_t Value) { Par.at(Parameter) = Value; }
  double GetParameter(M3::int_t Parameter) {
    if (Parameter < 0 || Parameter >= M3::int_t(Par.size())) throw MaCh3Exception(__FILE__, __LINE__, "TF1_red: parameter index out of range");
    return Par[Parameter];
  }
  double Eval(const double var) override { return Par[1] + Par[0] * var; }  // :192-193
  M3::int_t GetNp() override { return M3::int_t(Par.size()); }
  void Print() override {
    std::printf("TF1_red: length %zu\n", Par.size());
    for (size_t i = 0; i < Par.size(); ++i) std::printf("  coeff %zu = %g\n", i, double(Par[i]));
  }

 private:
  std::vector<M3::float_t> Par;
};

/// ref: Splines/SplineStructs.h:265-308, 673-734.  {x, y, b, c, d} per knot.
class TSpline3_red : public TResponseFunction_red {
 public:
  TSpline3_red() {}
  /// knots X[N], responses Y[N] and per-knot {b, c, d} = P[j][0..2] (ref: :285-308)
  TSpline3_red(const M3::float_t* X, const M3::float_t* Y, M3::int_t N, M3::float_t** P) {
    XPos.assign(X, X + N);
    YResp.assign(Y, Y + N);
    Par.resize(size_t(N) * 3);
    for (int j = 0; j < N; ++j)
      for (int k = 0; k < 3; ++k) Par[size_t(j) * 3 + k] = P[j][k];
  }
  /// same, with the coefficients as one flat [N][3] array
  TSpline3_red(const M3::float_t* X, const M3::float_t* Y, M3::int_t N, const M3::float_t* P_flat) {
    XPos.assign(X, X + N);
    YResp.assign(Y, Y + N);
    Par.assign(P_flat, P_flat + size_t(N) * 3);
  }

  /// segment search in double, as TSpline3::FindX (ref: :673-703)
  int FindX(double x) {
    const int nPoints = int(XPos.size());
    int segment = 0, kHigh = nPoints - 1;
    if (x <= XPos[0]) {
      segment = 0;
    } else if (x >= XPos[nPoints - 1]) {
      segment = kHigh;
    } else {
      while (kHigh - segment > 1) {
        const int kHalf = (segment + kHigh) / 2;
        if (x > XPos[kHalf]) segment = kHalf;
        else kHigh = kHalf;
      }
    }
    if (segment >= nPoints - 1 && nPoints > 1) segment = nPoints - 2;
    return segment;
  }
  double Eval(double var) override {  // ref: :706-717
    const int s = FindX(var);
    const double dx = var - XPos[s];
    return YResp[s] + dx * (Par[3 * s] + dx * (Par[3 * s + 1] + Par[3 * s + 2] * dx));
  }
  M3::int_t GetNp() override { return M3::int_t(XPos.size()); }
  void GetKnot(int i, M3::float_t& xtmp, M3::float_t& ytmp) { xtmp = XPos[i]; ytmp = YResp[i]; }
  void GetCoeff(int segment, M3::float_t& x, M3::float_t& y, M3::float_t& b, M3::float_t& c, M3::float_t& d) {
    x = XPos[segment]; y = YResp[segment];
    b = Par[3 * segment]; c = Par[3 * segment + 1]; d = Par[3 * segment + 2];
  }
  void Print() override {
    std::printf("TSpline3_red: %zu knots\n", XPos.size());
    for (size_t i = 0; i < XPos.size(); ++i)
      std::printf("  %zu: x=%g y=%g b=%g c=%g d=%g\n", i, double(XPos[i]), double(YResp[i]), double(Par[3 * i]), double(Par[3 * i + 1]), double(Par[3 * i + 2]));
  }

 private:
  std::vector<M3::float_t> XPos, YResp, Par;
};
