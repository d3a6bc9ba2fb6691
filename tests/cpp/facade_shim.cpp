// C entry points over the header-only facade, for the python tests (no GPU needed).
#include <cstring>

#include "../../include/mach3_amd/SampleHandlerFD.h"
#include "../../include/mach3_amd/SplineBuilders.h"

extern "C" {

// MakeTSpline3_red -> out[n][5] = {x, y, b, c, d}; returns 0, or -1 if the facade threw
int facade_build_spline(int interpolation, int n, const double* x, const double* y, float* out) {
  try {
    std::vector<double> xs(x, x + n), ys(y, y + n);
    TSpline3_red* s = MakeTSpline3_red(xs, ys, SplineInterpolation(interpolation));
    for (int i = 0; i < n; ++i) {
      M3::float_t xx, yy, b, c, d;
      s->GetCoeff(i, xx, yy, b, c, d);
      out[5 * i] = xx; out[5 * i + 1] = yy; out[5 * i + 2] = b; out[5 * i + 3] = c; out[5 * i + 4] = d;
    }
    delete s;
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

void facade_root_coeffs(int n, const double* x, const double* y, double* b, double* c, double* d) {
  std::vector<double> xs(x, x + n), ys(y, y + n), vb, vc, vd;
  M3::RootTSpline3Coefficients(xs, ys, vb, vc, vd);
  std::memcpy(b, vb.data(), sizeof(double) * n);
  std::memcpy(c, vc.data(), sizeof(double) * n);
  std::memcpy(d, vd.data(), sizeof(double) * n);
}

double facade_spline_eval(int n, const float* x, const float* y, const float* bcd, double var) {
  TSpline3_red s(x, y, M3::int_t(n), bcd);
  return s.Eval(var);
}

// BinningHandler of the facade: global bin of one event, for integer parity with the oracle
static int find_bin(const BinningHandler& b, int sample, const double* kin, const int* nom_bin) {
  std::vector<const double*> kv;
  std::vector<int> nb;
  for (int d = 0; d < b.GetNDim(sample); ++d) {
    kv.push_back(&kin[d]);
    nb.push_back(nom_bin ? nom_bin[d] : b.FindNominalBin(sample, d, kin[d]));
  }
  return b.FindGlobalBin(sample, kv, nb);
}

int facade_find_global_bin(int n_samples, const int* n_dim, const int* n_edges, const double* edges, int sample, const double* kin,
                           const int* nom_bin) {
  BinningHandler b;
  int k = 0;
  size_t off = 0;
  for (int s = 0; s < n_samples; ++s) {
    std::vector<std::vector<double>> ed;
    for (int d = 0; d < n_dim[s]; ++d, ++k) {
      ed.emplace_back(edges + off, edges + off + n_edges[k]);
      off += size_t(n_edges[k]);
    }
    b.AddUniformSample(ed);
  }
  return find_bin(b, sample, kin, nom_bin);
}

// non-uniform sample: boxes[n_boxes][n_dim][2]
int facade_find_nonuniform_bin(int n_boxes, int n_dim, const double* boxes, const double* kin, const int* nom_bin) {
  BinningHandler b;
  const std::vector<std::array<double, 2>> one_box(static_cast<size_t>(n_dim));
  std::vector<std::vector<std::array<double, 2>>> bx(static_cast<size_t>(n_boxes), one_box);
  for (int i = 0; i < n_boxes; ++i)
    for (int d = 0; d < n_dim; ++d) bx[size_t(i)][size_t(d)] = {boxes[(size_t(i) * n_dim + d) * 2], boxes[(size_t(i) * n_dim + d) * 2 + 1]};
  const int sample = b.AddNonUniformSample(bx);
  return find_bin(b, sample, kin, nom_bin);
}
}
