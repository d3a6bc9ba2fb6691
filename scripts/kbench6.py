"""Events of several interaction modes (each with its own list of 96 of 128 TSpline3 parameters) in random order:
engine-side grouping (default) vs none (M3HIP_GROUP_EVENTS=0), and the same MC pre-sorted by mode."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import capi
from mach3_amd.engine import Engine, SMonolith, SampleHandlerFD
E = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
P, K, L = 128, 7, 96
rng = np.random.default_rng(1)
modes = [np.sort(rng.choice(P, L, replace=False)).astype(np.int16) for _ in range(4)]
def build(which):
    pn = np.concatenate([modes[m] for m in which]).astype(np.int16)          # ascending within an event
    S = len(pn)
    ch = dict(n_events=len(which), n_splines=S, n_tf1=0, total_nknots=S * K,
              coeff_many=(1 + 0.01 * rng.standard_normal(S * K * 4)).astype(np.float32),
              nknots_arr=(np.arange(S, dtype=np.uint64) * K).astype(np.uint32), param_no=pn,
              coeff_tf1=np.zeros(0, np.float32), param_no_tf1=np.zeros(0, np.int16),
              npe=np.stack([np.full(len(which), L, np.uint32), (np.arange(len(which), dtype=np.uint64) * L).astype(np.uint32)], axis=1).reshape(-1),
              npe_tf1=np.zeros(2 * len(which), np.uint32))
    return ch
cx = np.tile(np.arange(-3.0, 4.0, dtype=np.float32), P)
npts = np.full(P, K, np.int16)
which_rand = rng.integers(0, 4, E)
for label, which, env in (("random order, engine grouping", which_rand, None), ("random order, no grouping", which_rand, "0"), ("sorted by mode", np.sort(which_rand), None)):
    if env is None: os.environ.pop("M3HIP_GROUP_EVENTS", None)
    else: os.environ["M3HIP_GROUP_EVENTS"] = env
    eng = Engine(0)
    chunks = [build(which[i:i + 131072]) for i in range(0, E, 131072)]
    mono = SMonolith(eng, P, K, cx, npts, chunks)
    par = np.zeros(P)
    mono.setSplinePointers(par)
    sh = SampleHandlerFD(eng, 1, np.array([0, 200], np.int32), par, capi.kPoisson)
    sh.SetupEvents(0, rng.integers(0, 200, E).astype(np.int32), None, None, None)
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    info = eng.info()
    eng.timing_enable(True)
    t = time.perf_counter()
    while time.perf_counter() - t < 1.0: par[:] = rng.normal(0, 0.3, P); sh.Step()
    eng.timing_reset()
    t = time.perf_counter()
    for i in range(100): par[:] = rng.normal(0, 0.3, P); sh.Step()
    us = (time.perf_counter() - t) / 100 * 1e6
    f, _ = eng.timing_read(0)
    print(f"{label}: tiles {info.n_tiles_uniform}u ({info.n_tiles_padded} padded) / {info.n_tiles_brick}b, grouped chunks {info.n_grouped_chunks}: "
          f"kernel {f*1e3:.1f} us ({info.alg_bytes_fused/f/1e6:.0f} GB/s algorithmic), step {us:.1f} us", flush=True)
    eng.close()
