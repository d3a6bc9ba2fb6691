// mr2t2_sharded.cpp -- ONE sample over several GPUs from the side a MaCh3 maintainer would call it: N processes, one per GPU,
// each builds the facade objects (SMonolith, SampleHandlerFD) over ITS block of events (M3::ShardRanges), joins the exchange
// (SampleHandlerFD::InitComm over RCCL, or CreateExchange / OpenExchange: the one-shot peer-store form), and then runs the SAME
// seeded Metropolis chain (Fitters/MR2T2.cpp:25-115) as every other rank.  Reweight() ends in one exchange of the bin vector,
// so every rank sees the whole sample's likelihood, takes the same accept / reject decisions and proposes the same next step:
// no further communication, no master rank.  The reference has no counterpart (one device: Manager/gpuUtils.cu:71).
//
// The two small blobs the ranks must hand each other (RCCL's 128-byte unique id from rank 0, or every rank's 64-byte IPC
// handle) travel through files in a directory here; MPI_Bcast / MPI_Allgather do the same job in an MPI build.
//
//   usage: mr2t2_sharded <rank> <n_ranks> <dir> [exchange=oneshot|rccl|none] [n_events=200000] [n_params=64] [n_steps=300] [device=-1] [order_key=0]
//          order_key 1: the events' oscillation bins go to the engine as its order key (SMonolithGPU::next_event_order_key)
//          device -1: rank % visible devices (one process per GPU; ranks that share a card: the one-shot exchange only)
//   every rank prints one line  `RESULT rank=.. events=[a,b) steps=.. accepted=.. acc=<0/1 per step> logl=<%.17g> mc0=<%.17g> ...`
//   and writes it to <dir>/result_<rank>.txt; with n_ranks = 1 and exchange = none it is the one-process chain to compare with.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>

#include "../include/mach3_amd/Fitters.h"
#include "../include/mach3_amd/SampleHandlerFD.h"
#include "../include/mach3_amd/Sharding.h"
#include "../mach3_amd/synth/m3synth.h"

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// file rendez-vous: a blob appears under its final name only when it is complete
void publish(const std::string& dir, const std::string& name, const void* p, size_t n) {
  const std::string f = dir + "/" + name, tmp = f + ".tmp";
  {
    std::ofstream o(tmp, std::ios::binary);
    o.write(static_cast<const char*>(p), std::streamsize(n));
    if (!o) throw MaCh3Exception(__FILE__, __LINE__, "cannot write " + tmp);
  }
  if (std::rename(tmp.c_str(), f.c_str()) != 0) throw MaCh3Exception(__FILE__, __LINE__, "cannot rename " + tmp);
}
void fetch(const std::string& dir, const std::string& name, void* p, size_t n, double limit_s = 120.0) {
  const std::string f = dir + "/" + name;
  const double t0 = now_s();
  for (;;) {
    std::ifstream i(f, std::ios::binary);
    if (i && i.read(static_cast<char*>(p), std::streamsize(n))) return;
    if (now_s() - t0 > limit_s) throw MaCh3Exception(__FILE__, __LINE__, "rendez-vous: " + f + " did not appear");
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 4) {
    std::fprintf(stderr, "usage: %s <rank> <n_ranks> <dir> [oneshot|rccl|none] [n_events] [n_params] [n_steps] [device]\n", argv[0]);
    return 2;
  }
  const int rank = std::atoi(argv[1]), n_ranks = std::atoi(argv[2]);
  const std::string dir = argv[3];
  const std::string exchange = argc > 4 ? argv[4] : "oneshot";
  const uint64_t E = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 200000;
  const int P = argc > 6 ? std::atoi(argv[6]) : 64;
  const int NSteps = argc > 7 ? std::atoi(argv[7]) : 300;
  int device = argc > 8 ? std::atoi(argv[8]) : -1;
  const bool order_key = argc > 9 && std::atoi(argv[9]) != 0;  // the engine stores the events in oscillation-bin order (invisible here)
  if (rank < 0 || rank >= n_ranks || (n_ranks > 1 && exchange == "none")) {
    std::fprintf(stderr, "bad rank / n_ranks / exchange\n");
    return 2;
  }

  m3synth_config cfg{};  // a ragged mixed sample: every event draws its parameters independently -> bricks
  cfg.seed = 20260402;
  cfg.n_params = P;
  cfg.n_knots = 7;
  cfg.mixed = 1;
  cfg.n_norm = 16;
  cfg.max_norm_per_event = 3;
  cfg.n_samples = 2;
  cfg.n_bins = 100;
  cfg.osc_n = 4000;
  cfg.occupancy = 0.25;
  cfg.bin_lo = 0.0;
  cfg.bin_hi = 5.0;
  cfg.nc_fraction = 0.1;
  const int nPar = P + cfg.n_norm;

  try {
    if (device < 0) {
      int n_dev = 0;
      if (m3hip_device_count(&n_dev) != M3HIP_OK || n_dev < 1) throw MaCh3Exception(__FILE__, __LINE__, "no HIP device");
      device = rank % n_dev;
    }
    // ---- this rank's events: contiguous, on 1024-event boundaries, balanced by the responses the events carry (every rank
    // counts the whole sample in blocks of 1024 events -- a pure function of the event number, so all ranks get the same cuts)
    std::vector<double> weight(E);
    for (uint64_t b0 = 0; b0 < E; b0 += 1024) {
      const uint64_t b1 = std::min(E, b0 + 1024);
      uint64_t s = 0, t = 0;
      m3synth_count(&cfg, b0, b1, &s, &t);
      for (uint64_t e = b0; e < b1; ++e) weight[e] = double(s + t) / double(b1 - b0);
    }
    const auto cuts = M3::ShardRanges(E, n_ranks, weight);
    const uint64_t e0 = cuts[size_t(rank)].first, e1 = cuts[size_t(rank)].second, n = e1 - e0;
    std::vector<double>().swap(weight);

    // ---- the shard as response-function objects -> SMonolith (the reference's constructor, Splines/SplineMonolith.h:18-26)
    uint64_t S = 0, T = 0;
    m3synth_count(&cfg, e0, e1, &S, &T);
    std::vector<float> coeff_many(S * 7 * 4), coeff_tf1(T * 2), coeff_x(size_t(P) * 7);
    std::vector<unsigned> nknots(S), npe(2 * n), npe_t(2 * n);
    std::vector<int16_t> pno(S), pno_t(T), npts(P);
    m3synth_knots(&cfg, coeff_x.data(), npts.data());
    m3synth_monolith(&cfg, e0, e1, coeff_many.data(), nknots.data(), pno.data(), coeff_tf1.data(), pno_t.data(), npe.data(), npe_t.data());
    std::vector<RespFuncType> SplineType(P);
    for (int p = 0; p < P; ++p) SplineType[p] = (p & 1) ? kTF1_red : kTSpline3_red;
    std::vector<std::vector<TResponseFunction_red*>> MasterSpline(n, std::vector<TResponseFunction_red*>(P, nullptr));
    for (uint64_t e = 0; e < n; ++e) {
      for (unsigned i = 0; i < npe[2 * e]; ++i) {
        const unsigned s = npe[2 * e + 1] + i;
        const float* c = &coeff_many[size_t(nknots[s]) * 4];
        float X[7], Y[7], Pf[21];
        for (int k = 0; k < 7; ++k) {
          X[k] = coeff_x[size_t(pno[s]) * 7 + k];
          Y[k] = c[4 * k];
          Pf[3 * k] = c[4 * k + 1]; Pf[3 * k + 1] = c[4 * k + 2]; Pf[3 * k + 2] = c[4 * k + 3];
        }
        MasterSpline[e][pno[s]] = new TSpline3_red(X, Y, 7, Pf);
      }
      for (unsigned i = 0; i < npe_t[2 * e]; ++i) {
        const unsigned t = npe_t[2 * e + 1] + i;
        MasterSpline[e][pno_t[t]] = new TF1_red(2, &coeff_tf1[2 * t]);
      }
    }
    std::vector<float>().swap(coeff_many);

    std::vector<double> nominal(nPar), lo(nPar), hi(nPar), cov(size_t(nPar) * nPar, 0.0);
    m3synth_params(&cfg, 0, nominal.data());
    for (int i = 0; i < nPar; ++i) {
      const double sigma = i < P ? 1.0 : 0.1;
      lo[i] = i < P ? -5.0 : 0.0;
      hi[i] = i < P ? 5.0 : 2.0;
      cov[size_t(i) * nPar + i] = sigma * sigma;
    }
    ParameterHandlerBase ph("synthetic_systematics", nominal, cov, lo, hi, {}, 4242);  // the same seed on every rank

    if (order_key) {  // the events' oscillation bins, handed over before the monolith they order is built
      std::vector<int32_t> sid(n), osc(n);
      std::vector<double> erec(n);
      std::vector<uint64_t> noff(n + 1);
      std::vector<int16_t> nidx(n * cfg.max_norm_per_event + 1);
      m3synth_events(&cfg, e0, e1, sid.data(), erec.data(), osc.data(), noff.data(), nidx.data());
      SMonolithGPU::next_event_order_key() = osc;
    }
    auto splines = std::make_unique<SMonolith>(MasterSpline, SplineType, false, "", device);
    std::vector<const double*> ptrs(P);
    for (int p = 0; p < P; ++p) ptrs[p] = ph.RetPointer(p);
    splines->setSplinePointers(ptrs);
    auto bins = std::make_unique<BinningHandler>();
    std::vector<double> edges(cfg.n_bins + 1);
    m3synth_bin_edges(&cfg, edges.data());
    for (int s = 0; s < cfg.n_samples; ++s) bins->AddUniformSample({edges});
    SampleHandlerFD sample("ShardedFD", std::move(splines), std::move(bins), ph.RetPointer(0), nPar);
    sample.SetTestStatistic(kBarlowBeeston);
    {
      std::vector<int32_t> sid(n), osc(n);
      std::vector<double> erec(n);
      std::vector<uint64_t> noff(n + 1);
      std::vector<int16_t> nidx(n * cfg.max_norm_per_event + 1);
      m3synth_events(&cfg, e0, e1, sid.data(), erec.data(), osc.data(), noff.data(), nidx.data());
      for (uint64_t e = 0; e < n; ++e)
        sample.AddEvent(unsigned(e), sid[e], {erec[e]}, std::vector<int16_t>(nidx.begin() + noff[e], nidx.begin() + noff[e + 1]), osc[e]);
      sample.FinaliseEvents();
      std::vector<float> tbl(cfg.osc_n);
      m3synth_osc_table(&cfg, tbl.data());
      sample.SetOscillationWeights(tbl);
    }

    // ---- join the exchange
    if (exchange == "rccl") {
      std::array<char, M3HIP_UNIQUE_ID_BYTES> id{};
      if (rank == 0) {
        id = SampleHandlerFD::GetCommUniqueId();
        publish(dir, "rccl_unique_id", id.data(), id.size());
      }
      fetch(dir, "rccl_unique_id", id.data(), id.size());
      sample.InitComm(id.data(), rank, n_ranks, /*Validate=*/true);
    } else if (exchange == "oneshot") {
      const auto mine = sample.CreateExchange(rank, n_ranks);
      publish(dir, "xch_handle_" + std::to_string(rank), mine.data(), mine.size());
      std::vector<std::array<char, M3HIP_XCH_HANDLE_BYTES>> all(static_cast<size_t>(n_ranks));
      for (int r = 0; r < n_ranks; ++r) fetch(dir, "xch_handle_" + std::to_string(r), all[size_t(r)].data(), M3HIP_XCH_HANDLE_BYTES);
      sample.OpenExchange(all, /*TimeoutSeconds=*/20.0);  // the ranks of a test box start seconds apart
    }

    // ---- Asimov data at nominal (Fitters/MaCh3Factory.h:142-153): after the exchange every rank holds the WHOLE sample's bins
    sample.Reweight();
    for (int s = 0; s < cfg.n_samples; ++s) sample.AddData(s, sample.GetMCArray(s));
    if (std::fabs(sample.GetLikelihood()) > 1e-9) throw MaCh3Exception(__FILE__, __LINE__, "Asimov likelihood is not zero");

    // ---- the same seeded chain on every rank
    MR2T2 fitter(1234);
    fitter.AddSampleHandler(&sample);
    fitter.AddSystObj(&ph);
    ph.SetStepScale(0.4 / std::sqrt(double(nPar)));
    const double t0 = now_s();
    fitter.RunMCMC(NSteps);
    const double el = now_s() - t0;

    // the prediction at the chain's last CURRENT point (one more collective Reweight, the same on every rank)
    for (int i = 0; i < nPar; ++i) ph.SetParCurrProp(i, ph.GetParCurr(i));
    sample.Reweight();
    const std::vector<double> mc = sample.GetMCArray();
    double mc_sum = 0.0;
    for (double v : mc) mc_sum += v;

    std::string acc;
    for (char a : fitter.GetAcceptStore()) acc.push_back(a ? '1' : '0');
    char line[512];
    m3hip_info info{};
    m3hip_get_info(sample.GetSplineHandler()->Handle(), &info);
    std::snprintf(line, sizeof(line), "RESULT rank=%d ranks=%d exchange=%s events=[%llu,%llu) steps=%d accepted=%d oob=%d logl=%.17g mc0=%.17g mcsum=%.17g steps_per_s=%.0f keyed_chunks=%llu osc_lines=%.1f",
                  rank, n_ranks, exchange.c_str(), (unsigned long long)e0, (unsigned long long)e1, fitter.GetStepCount(), fitter.GetAccCount(),
                  fitter.GetOutOfBoundsCount(), fitter.GetLogLCurr(), mc[0], mc_sum, NSteps / el, (unsigned long long)info.n_keyed_chunks, info.osc_lines_per_kevent);
    const std::string out = std::string(line) + " acc=" + acc + "\n";
    std::fputs(out.c_str(), stdout);
    {
      std::ofstream o(dir + "/bins_" + std::to_string(rank) + ".bin", std::ios::binary);
      o.write(reinterpret_cast<const char*>(mc.data()), std::streamsize(sizeof(double) * mc.size()));
    }
    publish(dir, "result_" + std::to_string(rank) + ".txt", out.data(), out.size());
    if (n_ranks > 1) {
      // leave together: a rank that destroys its exchange buffer while a peer still pushes into it would fault the peer
      for (int r = 0; r < n_ranks; ++r) {
        char c = 0;
        fetch(dir, "result_" + std::to_string(r) + ".txt", &c, 1);
      }
      sample.CloseComm();
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rank %d FAILED: %s\n", rank, e.what());
    return 1;
  }
  std::printf("OK\n");
  return 0;
}
