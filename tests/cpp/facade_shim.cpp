// C entry points over the header-only facade, for the python tests (no GPU needed).
#include <cstring>

#include "../../include/mach3_amd/SampleHandlerFD.h"
#include "../../include/mach3_amd/BinnedSplineHandler.h"
#include "../../include/mach3_amd/Fitters.h"
#include "../../include/mach3_amd/SplineBuilders.h"
#include "../../include/mach3_amd/Sharding.h"

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

// ---- Fitters.h on a toy target: correlated Gaussian prior x Gaussian "sample" likelihood.
// kind 0: MR2T2, 1: DelayedMR2T2, 2: MR2T2 + adaptive proposal.  out = {acceptance, delayed accepts,
// out-of-bounds count, mean[3], var[3], cov01}.
namespace {
struct ToySample : public SampleHandlerBase {
  const double* par;  // reads the proposed values, like a real sample
  double mu[3] = {0.5, -0.2, 0.1}, sig[3] = {0.3, 0.5, 0.2};
  int n_reweight = 0;
  explicit ToySample(const double* p) : par(p) { nSamples = 1; }
  std::string GetName() const override { return "Toy"; }
  std::string GetSampleTitle(const int) const override { return "Toy"; }
  void Reweight() override { ++n_reweight; }
  double GetLikelihood() const override {
    double l = 0;
    for (int i = 0; i < 3; ++i) l += 0.5 * (par[i] - mu[i]) * (par[i] - mu[i]) / (sig[i] * sig[i]);
    return l;
  }
  double GetSampleLikelihood(const int) const override { return GetLikelihood(); }
  void CleanMemoryBeforeFit() override {}
  void PrintRates(const bool) override {}
};
}  // namespace

// ParameterHandlerBase step-path semantics that need no sample: returns 0, or the number of the failing check
int facade_parameter_handler_checks(unsigned long long seed) {
  try {
    const double pi = 3.14159265358979323846;
    const std::vector<double> prefit = {0.0, 1.0, 0.5};
    const std::vector<double> cov = {1.0, 0.3, 0.0, 0.3, 1.0, 0.0, 0.0, 0.0, 4.0};
    ParameterHandlerBase ph("toy", prefit, cov, {-10, -10, -pi}, {10, 10, pi}, {}, seed);
    // 1: a fixed parameter never moves, the others do, and it contributes no random number to their step
    ph.SetFixParameter(1);
    if (!ph.IsParameterFixed(1) || ph.IsParameterFixed(0)) return 1;
    for (int s = 0; s < 200; ++s) {
      ph.ProposeStep();
      if (ph.GetParProp(1) != 1.0) return 1;
      ph.AcceptStep();
    }
    if (ph.GetParCurr(0) == 0.0 || ph.GetParCurr(2) == 0.5) return 1;
    ph.SetFreeParameter(1);
    ph.ProposeStep();
    if (ph.GetParProp(1) == 1.0) return 1;
    // 2: circular bounds keep a cyclic parameter inside [-pi, pi] whatever the step (ref: CircularParBounds)
    ph.EnableCircularBounds(2, -pi, pi);
    ph.SetStepScale(3.0);
    for (int s = 0; s < 2000; ++s) {
      ph.ProposeStep();
      if (ph.GetParProp(2) > pi || ph.GetParProp(2) < -pi) return 2;
      ph.AcceptStep();
    }
    ph.SetParCurrProp(2, 3.0);
    // the wrap itself: 3.0 + 0.5 -> -pi + (3.5 - pi)
    {
      ParameterHandlerBase q("q", {3.5}, {1.0}, {-pi}, {pi}, {}, 1);
      q.CircularParBounds(0, -pi, pi);
      if (std::fabs(q.GetParProp(0) - (-pi + (3.5 - pi))) > 1e-15) return 2;
      q.SetParCurrProp(0, -4.0);
      q.CircularParBounds(0, -pi, pi);
      if (std::fabs(q.GetParProp(0) - (pi - (4.0 - pi))) > 1e-15) return 2;
    }
    // 3: circular range outside the physical bounds is refused
    try {
      ph.EnableCircularBounds(0, -20, 20);
      return 3;
    } catch (const MaCh3Exception&) {
    }
    // 4: the flip mirrors about the point in about half of the proposals (ref: FlipParameterValue)
    {
      ParameterHandlerBase f("f", {2.0}, {1e-12}, {-10}, {10}, {}, seed + 5);
      f.EnableFlip(0, 0.0);
      int flipped = 0;
      for (int s = 0; s < 4000; ++s) {
        f.SetParCurrProp(0, 2.0);
        f.ProposeStep();
        flipped += f.GetParProp(0) < 0;
      }
      if (flipped < 1800 || flipped > 2200) return 4;
    }
    // 5: flat prior drops the parameter from the penalty; out of bounds -> _LARGE_LOGL_ per parameter
    ph.SetParCurrProp(0, 1.0); ph.SetParCurrProp(1, 1.0); ph.SetParCurrProp(2, 0.5);
    const double with0 = ph.CalcLikelihood();   // d = (1,0,0): 0.5 * Vinv00 = 0.5 / (1 - 0.09)
    if (std::fabs(with0 - 0.5 / 0.91) > 1e-12) return 5;
    ph.SetFlatPrior(0, true);
    if (ph.CalcLikelihood() != 0.0) return 5;
    ph.SetParCurrProp(1, 11.0);
    if (ph.GetLikelihood() != M3::_LARGE_LOGL_) return 5;
    // 6: RandomConfiguration stays inside the bounds and leaves fixed parameters alone
    ph.SetParCurrProp(1, 1.0);
    ph.SetFixParameter(0);
    const double before = ph.GetParCurr(0);
    for (int s = 0; s < 100; ++s) {
      ph.RandomConfiguration();
      if (ph.GetParCurr(0) != before || ph.CheckBounds() != 0 || ph.GetParCurr(1) != ph.GetParProp(1)) return 6;
    }
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

// 1 when registering the same sub-sample title twice is refused (Fitters/FitterBase.cpp:262-279), else 0
// The parameter handler's prior and correlated throw on the device (ParameterHandlerBase::AttachDevice -> m3hip_par_*)
// against the host loops of the same class, on a random SPD covariance of n parameters: two handlers with the same seed
// walk the same chain, one on the host, one attached to device 0.  out[0] = worst relative difference of the prior term,
// out[1] = worst difference of a proposed value in units of the step's largest move, out[2] = steps compared.
// Half way a new throw matrix arrives while a device throw is in flight: it must be dropped and recomputed.  Needs a GPU.
int facade_par_device_check(int n, unsigned long long seed, int n_steps, double* out) {
  m3hip_handle h = nullptr;
  try {
    if (m3hip_create(0, &h) != M3HIP_OK) return 100;
    std::mt19937_64 g(seed);
    std::normal_distribution<double> N(0.0, 1.0);
    auto spd = [&](double ridge) {
      std::vector<double> a(size_t(n) * n), c(size_t(n) * n, 0.0);
      for (auto& v : a) v = N(g);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = 0.0;
          for (int k = 0; k < n; ++k) s += a[size_t(i) * n + k] * a[size_t(j) * n + k];
          c[size_t(i) * n + j] = c[size_t(j) * n + i] = s / n + (i == j ? ridge : 0.0);
        }
      return c;
    };
    const std::vector<double> cov = spd(0.5), cov2 = spd(0.25);
    std::vector<double> prefit(static_cast<size_t>(n), 0.0), lo(static_cast<size_t>(n), -1e3), up(static_cast<size_t>(n), 1e3);
    for (auto& v : prefit) v = N(g);
    std::vector<bool> flat(static_cast<size_t>(n), false);
    for (int i = 0; i < n; i += 37) flat[size_t(i)] = true;  // some flat priors: they drop out of the prior term
    ParameterHandlerBase host("host", prefit, cov, lo, up, flat, seed), dev("dev", prefit, cov, lo, up, flat, seed);
    host.SetFixParameter(3 % n);
    dev.SetFixParameter(3 % n);
    dev.AttachDevice(h);
    if (!dev.OnDevice() || host.OnDevice()) return 1;
    double worst_prior = 0.0, worst_prop = 0.0;
    for (int s = 0; s < n_steps; ++s) {
      host.ProposeStep();
      dev.ProposeStep();  // from the second step on with the throw the device computed
      double move = 0.0, diff = 0.0;
      for (int i = 0; i < n; ++i) {
        move = std::max(move, std::fabs(host.GetParProp(i) - host.GetParCurr(i)));
        diff = std::max(diff, std::fabs(host.GetParProp(i) - dev.GetParProp(i)));
      }
      if (!(move > 0.0)) return 2;
      worst_prop = std::max(worst_prop, diff / move);
      if (dev.GetParProp(3 % n) != dev.GetParCurr(3 % n)) return 3;  // fixed parameters stay
      const double p_host = host.GetLikelihood();
      dev.LaunchDeviceStep();
      if (s == n_steps / 2) {  // a new throw matrix while a throw is in flight
        host.SetThrowMatrix(cov2);
        dev.SetThrowMatrix(cov2);
      }
      const double p_dev = dev.CollectDeviceStep();
      worst_prior = std::max(worst_prior, std::fabs(p_dev - p_host) / std::max(std::fabs(p_host), 1e-300));
      host.AcceptStep();
      dev.AcceptStep();
    }
    out[0] = worst_prior;
    out[1] = worst_prop;
    out[2] = n_steps;
    m3hip_destroy(h);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_par_device_check: %s\n", e.what());
    if (h) m3hip_destroy(h);
    return 99;
  }
}

// SampleHandlerFD::CalcNormsBins for one event against a list of parameters given as flat arrays: per parameter index, the four
// match lists as CSR (off[4 * n_par + 1], values), one optional bounded variable {column, n_ranges, ranges...} per parameter as CSR of
// doubles (kin_off[n_par + 1], kin: {column, lo0, hi0, lo1, hi1, ...}; empty = no bounds).  Returns the number of indices written.
int facade_calc_norms_bins(int n_par, const int* index, const int* off, const int* values, const int* kin_off, const double* kin, int mode, int target,
                           int nupdg, int nupdg_unosc, int is_nc, int n_kin, const double* kinvars, short* out) {
  try {
    std::vector<NormParameter> pars(static_cast<size_t>(n_par));
    for (int p = 0; p < n_par; ++p) {
      NormParameter& np = pars[size_t(p)];
      np.name = "p" + std::to_string(p);
      np.index = index[p];
      std::vector<int>* lists[4] = {&np.modes, &np.pdgs, &np.preoscpdgs, &np.targets};
      for (int l = 0; l < 4; ++l) lists[l]->assign(values + off[4 * p + l], values + off[4 * p + l + 1]);
      const int k0 = kin_off[p], k1 = kin_off[p + 1];
      if (k1 > k0) {
        np.hasKinBounds = true;
        np.KinematicColumn.push_back(int(kin[k0]));
        std::vector<std::vector<double>> ranges;
        for (int k = k0 + 1; k + 1 < k1; k += 2) ranges.push_back({kin[k], kin[k + 1]});
        np.Selection.push_back(ranges);
      }
    }
    EventTruth ev;
    ev.mode = mode; ev.target = target; ev.nupdg = nupdg; ev.nupdgUnosc = nupdg_unosc; ev.isNC = is_nc != 0;
    const auto bins = SampleHandlerFD::CalcNormsBins(pars, ev, std::vector<double>(kinvars, kinvars + n_kin));
    for (size_t i = 0; i < bins.size(); ++i) out[i] = bins[i];
    return int(bins.size());
  } catch (const std::exception&) {
    return -1;
  }
}

// BinnedSplineIndex::GetEventSplines for n_ev events: axes as (n_edges[3], edges), per systematic its modes (CSR mode_off / modes) and
// spline indices (CSR id_off / ids, mode-major over the bins); out_off[n_ev + 1] / out_idx = the events' CSR rows
int facade_event_splines(const int* n_edges, const double* edges, int n_syst, const int* mode_off, const int* modes, const int* id_off, const int* ids,
                         int n_flat, const char* isflat, int n_ev, const int* ev_mode, const double* ev_vars, int* out_off, int* out_idx, int cap) {
  try {
    const double* e = edges;
    std::vector<double> ax[3];
    for (int a = 0; a < 3; ++a) { ax[a].assign(e, e + n_edges[a]); e += n_edges[a]; }
    BinnedSplineIndex bi(ax[0], ax[1], ax[2]);
    for (int s = 0; s < n_syst; ++s)
      bi.AddSystematic(std::vector<int>(modes + mode_off[s], modes + mode_off[s + 1]), std::vector<int32_t>(ids + id_off[s], ids + id_off[s + 1]));
    std::vector<char> flat(isflat, isflat + n_flat);
    int n = 0;
    out_off[0] = 0;
    for (int i = 0; i < n_ev; ++i) {
      const auto r = bi.GetEventSplines(ev_mode[i], ev_vars[3 * i], ev_vars[3 * i + 1], ev_vars[3 * i + 2], &flat);
      for (int32_t id : r) {
        if (n >= cap) return -2;
        out_idx[n++] = id;
      }
      out_off[i + 1] = n;
    }
    return n;
  } catch (const std::exception&) {
    return -1;
  }
}

// M3::ShardRanges (Sharding.h) for the comparison with mach3_amd/dist.py:shard_ranges: cuts[2 * n_ranks] = {first, last} per rank
int facade_shard_ranges(unsigned long long n_events, int n_ranks, const double* weights, unsigned long long* cuts) {
  try {
    std::vector<double> w;
    if (weights) w.assign(weights, weights + n_events);
    const auto r = M3::ShardRanges(n_events, n_ranks, w);
    for (int i = 0; i < n_ranks; ++i) {
      cuts[2 * i] = r[size_t(i)].first;
      cuts[2 * i + 1] = r[size_t(i)].second;
    }
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

// Two ParameterHandlerBase objects attached to ONE engine (MCMCBase loops over all systematics objects): each owns a slot, so
// with equal or different numbers of parameters both get their own prior and throw.  out[0] = worst relative difference of a
// prior against the handler's own host loop, out[1] = steps walked.
int facade_two_handlers_one_engine(int na, int nb, unsigned long long seed, int n_steps, double* out) {
  m3hip_handle h = nullptr;
  try {
    if (m3hip_create(0, &h) != M3HIP_OK) return 100;
    auto make = [&](const char* name, int n, double ridge, unsigned long long sd) {
      std::mt19937_64 g(sd ^ 0x9E3779B97F4A7C15ull);  // the handler is a function of its seed: a second call builds its twin
      std::normal_distribution<double> N(0.0, 1.0);
      std::vector<double> a(size_t(n) * n), c(size_t(n) * n, 0.0);
      for (auto& v : a) v = N(g);
      for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
          double s = 0.0;
          for (int k = 0; k < n; ++k) s += a[size_t(i) * n + k] * a[size_t(j) * n + k];
          c[size_t(i) * n + j] = c[size_t(j) * n + i] = s / n + (i == j ? ridge : 0.0);
        }
      std::vector<double> prefit(static_cast<size_t>(n), 0.0), lo(static_cast<size_t>(n), -1e3), up(static_cast<size_t>(n), 1e3);
      for (auto& v : prefit) v = N(g);
      return ParameterHandlerBase(name, prefit, c, lo, up, {}, sd);
    };
    ParameterHandlerBase A = make("xsec", na, 0.5, seed + 1), B = make("detector", nb, 0.3, seed + 2);
    // host-only twins: same state, same random streams (built again from the same seeds: a handler cannot be copied, it may own a device slot)
    ParameterHandlerBase Ah = make("xsec", na, 0.5, seed + 1), Bh = make("detector", nb, 0.3, seed + 2);
    A.AttachDevice(h);
    B.AttachDevice(h);
    double worst = 0.0;
    for (int s = 0; s < n_steps; ++s) {
      A.ProposeStep(); B.ProposeStep(); Ah.ProposeStep(); Bh.ProposeStep();
      for (int i = 0; i < na; ++i)
        if (std::fabs(A.GetParProp(i) - Ah.GetParProp(i)) > 1e-9 * (1.0 + std::fabs(Ah.GetParProp(i)))) return 2;
      for (int i = 0; i < nb; ++i)
        if (std::fabs(B.GetParProp(i) - Bh.GetParProp(i)) > 1e-9 * (1.0 + std::fabs(Bh.GetParProp(i)))) return 3;
      A.LaunchDeviceStep();
      B.LaunchDeviceStep();  // both in flight before either is collected (MCMCBase::EvaluateProposal's order)
      const double pa = A.CollectDeviceStep(), pb = B.CollectDeviceStep();
      const double ha = Ah.GetLikelihood(), hb = Bh.GetLikelihood();
      worst = std::max(worst, std::fabs(pa - ha) / std::max(std::fabs(ha), 1e-300));
      worst = std::max(worst, std::fabs(pb - hb) / std::max(std::fabs(hb), 1e-300));
      A.AcceptStep(); B.AcceptStep(); Ah.AcceptStep(); Bh.AcceptStep();
    }
    A.DetachDevice();
    B.DetachDevice();
    out[0] = worst;
    out[1] = n_steps;
    m3hip_destroy(h);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_two_handlers_one_engine: %s\n", e.what());
    if (h) m3hip_destroy(h);
    return 99;
  }
}

// SetThrowMatrix's panel Cholesky on a random SPD matrix of n parameters: out[0] = max |L L^T - A| / max |A|, out[1] = 1 if the
// factor is bit-identical with one thread and with `threads` (the build has OpenMP), out[2] = 1 if a singular matrix was made
// decomposable by a diagonal shift (M3::MakeMatrixPosDef's behaviour)
int facade_cholesky_check(int n, unsigned long long seed, int threads, double* out) {
  try {
    std::mt19937_64 g(seed);
    std::normal_distribution<double> N(0.0, 1.0);
    std::vector<double> a(size_t(n) * n), c(size_t(n) * n, 0.0);
    for (auto& v : a) v = N(g);
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += a[size_t(i) * n + k] * a[size_t(j) * n + k];
        c[size_t(i) * n + j] = c[size_t(j) * n + i] = s / n + (i == j ? 0.1 : 0.0);
      }
    std::vector<double> prefit(static_cast<size_t>(n), 0.0), lo(static_cast<size_t>(n), -1e3), up(static_cast<size_t>(n), 1e3);
    ParameterHandlerBase p1("one", prefit, c, lo, up, {}, 1), pt("team", prefit, c, lo, up, {}, 1);
    p1.SetAdaptionThreads(1);
    pt.SetAdaptionThreads(threads);
    p1.SetThrowMatrix(c);
    pt.SetThrowMatrix(c);
    const std::vector<double>&L1 = p1.GetThrowMatrixCholesky(), &Lt = pt.GetThrowMatrixCholesky();
    out[1] = std::memcmp(L1.data(), Lt.data(), sizeof(double) * L1.size()) == 0 ? 1.0 : 0.0;
    double worst = 0.0, amax = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = 0.0;
        for (int k = 0; k <= j; ++k) s += Lt[size_t(i) * n + k] * Lt[size_t(j) * n + k];
        worst = std::max(worst, std::fabs(s - c[size_t(i) * n + j]));
        amax = std::max(amax, std::fabs(c[size_t(i) * n + j]));
      }
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j)
        if (Lt[size_t(i) * n + j] != 0.0) return 2;  // upper triangle stays empty
    out[0] = worst / amax;
    // rank-deficient: two identical rows/columns
    std::vector<double> sing(c);
    for (int k = 0; k < n; ++k) sing[size_t(1) * n + k] = sing[size_t(k) * n + 1] = c[size_t(0) * n + k];
    sing[size_t(1) * n + 1] = c[0];
    sing[size_t(0) * n + 1] = sing[size_t(1) * n + 0] = c[0];
    out[2] = 0.0;
    try {
      pt.SetThrowMatrix(sing);
      out[2] = 1.0;
    } catch (const MaCh3Exception&) {
    }
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_cholesky_check: %s\n", e.what());
    return 99;
  }
}

int facade_duplicate_title_refused() {
  const std::vector<double> prefit = {0.0, 0.0, 0.0};
  const std::vector<double> cov = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  ParameterHandlerBase ph("toy", prefit, cov, {-5, -5, -5}, {5, 5, 5}, {}, 1);
  ToySample a(ph.RetPointer(0)), b(ph.RetPointer(0));
  MR2T2 f(1);
  f.AddSampleHandler(&a);
  try {
    f.AddSampleHandler(&b);
  } catch (const MaCh3Exception& e) {
    return std::string(e.what()).find("Duplicate sample title") != std::string::npos ? 1 : 0;
  }
  return 0;
}

// kind + 10: the same chain with the parameter handler attached to device 0 (ParameterHandlerBase::AttachDevice): the prior and
// the correlated throw of every step come from the m3hip_par_* kernels through MCMCBase::EvaluateProposal (needs a GPU)
int facade_toy_chain(int kind, int n_steps, unsigned long long seed, double* out) {
  m3hip_handle dev = nullptr;
  try {
    const std::vector<double> prefit = {0.0, 0.0, 0.0};
    const std::vector<double> cov = {1.0, 0.6, 0.0, 0.6, 1.0, 0.0, 0.0, 0.0, 0.25};
    ParameterHandlerBase ph("toy", prefit, cov, {-5, -5, -5}, {5, 5, 0.6}, {}, seed + 1);
    if (kind >= 10) {
      kind -= 10;
      if (m3hip_create(0, &dev) != M3HIP_OK) return -2;
      ph.AttachDevice(dev);
    }
    ToySample sample(ph.RetPointer(0));
    MR2T2 plain(seed);
    DelayedMR2T2 delayed(seed, 1.0, 0.3, 2, 0.1);
    MCMCBase* f = kind == 1 ? static_cast<MCMCBase*>(&delayed) : static_cast<MCMCBase*>(&plain);
    if (kind == 3) plain.SetAnnealTemp(double(n_steps) / 8);  // hot at the start, cooling as the chain goes on
    if (kind == 2) {
      ph.SetThrowMatrix({0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01});  // a poor starting proposal that adaption has to fix
      ph.InitialiseAdaption(200, n_steps, 2000, 500);
    }
    f->AddSampleHandler(&sample);
    f->AddSystObj(&ph);
    f->RunMCMC(n_steps);
    const std::vector<double>& st = f->GetStepStore();
    const int burn = n_steps / 5, n = n_steps - burn;
    double m[3] = {0, 0, 0}, v[3] = {0, 0, 0}, c01 = 0;
    for (int s = burn; s < n_steps; ++s)
      for (int i = 0; i < 3; ++i) m[i] += st[size_t(s) * 3 + i];
    for (int i = 0; i < 3; ++i) m[i] /= n;
    for (int s = burn; s < n_steps; ++s) {
      for (int i = 0; i < 3; ++i) v[i] += (st[size_t(s) * 3 + i] - m[i]) * (st[size_t(s) * 3 + i] - m[i]);
      c01 += (st[size_t(s) * 3] - m[0]) * (st[size_t(s) * 3 + 1] - m[1]);
    }
    out[0] = double(f->GetAccCount()) / n_steps;
    out[1] = kind == 1 ? double(delayed.GetDelayedAcceptCount()) : 0.0;
    out[2] = double(f->GetOutOfBoundsCount());
    for (int i = 0; i < 3; ++i) { out[3 + i] = m[i]; out[6 + i] = v[i] / n; }
    out[9] = c01 / n;
    out[10] = double(sample.n_reweight);
    if (dev) m3hip_destroy(dev);
    return 0;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "facade_toy_chain: %s\n", e.what());
    if (dev) m3hip_destroy(dev);
    return -1;
  }
}

// SMonolith::EventOrderByParameterList on a MasterSpline described by kind[e][p]: 0 none, 1 TSpline3 with 3 knots,
// 2 TSpline3 with ONE knot (flat: dropped by PrepareForGPU, but still part of the event's mode signature), 3 TF1
int facade_event_order(int n_events, int n_params, const int* kind, const int* is_tf1, unsigned int* order_out) {
  try {
    std::vector<RespFuncType> types;
    for (int p = 0; p < n_params; ++p) types.push_back(is_tf1[p] ? kTF1_red : kTSpline3_red);
    std::vector<std::vector<TResponseFunction_red*>> ms(static_cast<size_t>(n_events), std::vector<TResponseFunction_red*>(static_cast<size_t>(n_params), nullptr));
    float X[3] = {-1.f, 0.f, 1.f}, Y[3] = {0.9f, 1.f, 1.1f}, Pf[9] = {0};
    float tf[2] = {0.1f, 1.f};
    for (int e = 0; e < n_events; ++e)
      for (int p = 0; p < n_params; ++p) {
        const int k = kind[size_t(e) * n_params + p];
        if (k == 1) ms[size_t(e)][size_t(p)] = new TSpline3_red(X, Y, 3, Pf);
        else if (k == 2) ms[size_t(e)][size_t(p)] = new TSpline3_red(X, Y, 1, Pf);
        else if (k == 3) ms[size_t(e)][size_t(p)] = new TF1_red(2, tf);
      }
    const std::vector<uint32_t> order = SMonolith::EventOrderByParameterList(ms);
    for (size_t i = 0; i < order.size(); ++i) order_out[i] = order[i];
    for (auto& ev : ms)
      for (auto* rf : ev) delete rf;
    return int(order.size());
  } catch (const std::exception&) {
    return -1;
  }
}
// SampleHandlerFD::EventOrderByOscillationBin
int facade_osc_order(int n_events, const int* osc_idx, unsigned int* order_out) {
  try {
    const std::vector<uint32_t> order = SampleHandlerFD::EventOrderByOscillationBin(std::vector<int32_t>(osc_idx, osc_idx + n_events));
    for (size_t i = 0; i < order.size(); ++i) order_out[i] = order[i];
    return int(order.size());
  } catch (const std::exception&) {
    return -1;
  }
}
}
