"""The step of a fit that MOVES oscillation parameters: the weight table is replaced before every reweight (the reference refills it
inside every Reweight, Samples/SampleHandlerFD.cpp:323 -> OscillationHandler::Evaluate).  Fused-kernel time and step time of one
configuration (or a shard of it) with the table
  resting          set once (the pre-staged case BASELINE's configs describe; after a few steps the weights ride in the packed words)
  moving, host     m3hip_set_osc_table every step (pinned staging + asynchronous H2D copy of the table)
  moving, device   m3hip_set_osc_table_device every step (the table where a GPU calculator left it: no copy)
usage: python scripts/osc_moves.py <config> <events> [sorted] [spl]
  sorted: the MC registered in oscillation-bin order (SynthSpec.osc_sorted).  M3HIP_LIB names another build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from mach3_amd import synth
from mach3_amd.engine import build_from_synth

cfg, ev = sys.argv[1], int(sys.argv[2])
kw = {}
if "sorted" in sys.argv[3:]:
    kw["osc_sorted"] = True
if "spl" in sys.argv[3:]:
    kw["mixed"] = False
if os.environ.get("M3_GEXP_OSC_N"):
    kw["osc_n"] = int(os.environ["M3_GEXP_OSC_N"])
spec = synth.CONFIGS[cfg].scaled(ev, **kw)
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps = [synth.params(spec, i + 1) for i in range(16)]
base = synth.osc_table(spec)
K = 8
tables = [np.ascontiguousarray(np.roll(base, 977 * k)) for k in range(K)]  # same values, every entry different from lap to lap
dev = [torch.from_numpy(t).cuda() for t in tables]
torch.cuda.synchronize()
info = eng.info()
tag = f"[{os.path.basename(os.environ.get('M3HIP_LIB', 'libm3hip.so'))}{' OSC_COLUMN=' + os.environ['M3HIP_OSC_COLUMN'] if os.environ.get('M3HIP_OSC_COLUMN') else ''}]"
print(f"{tag} {spec.name}{' osc-sorted' if spec.osc_sorted else ''}: E={ev} osc table {spec.osc_n} entries ({4 * spec.osc_n / 1024:.0f} KB), G={info.brick_groups}, "
      f"bricks {info.n_tiles_brick}, alg {info.alg_bytes_fused / 1e9:.3f} GB", flush=True)


def run(case):
    def step(i):
        if case == "host":
            sh.SetOscillationWeights(tables[i % K])
        elif case == "device":
            sh.SetOscillationWeightsDevice(dev[i % K].data_ptr(), spec.osc_n)
        par[:] = laps[i % 16]
        return sh.Step()
    if case == "resting":
        sh.SetOscillationWeights(tables[0])
    eng.timing_enable(True)
    for i in range(24):
        step(i)
    res = []
    for rep in range(3):
        eng.timing_reset()
        t = time.perf_counter()
        for i in range(48):
            step(i)
        st = (time.perf_counter() - t) / 48 * 1e6
        k, n = eng.timing_read(0)
        res.append((k * 1e3, st))
    k = min(r[0] for r in res)
    print(f"  {case:8s} kernel {' / '.join(f'{r[0]:.1f}' for r in res)} us -> {info.alg_bytes_fused / k / 1e3 / 8000:.3f} of 8 TB/s; "
          f"step {' / '.join(f'{r[1]:.1f}' for r in res)} us", flush=True)
    return res


for case in (sys.argv[sys.argv.index("cases") + 1].split(",") if "cases" in sys.argv else ("resting", "host", "device", "resting")):
    run(case)
# the three forms give the same likelihood for the same table and parameters
par[:] = laps[3]
sh.SetOscillationWeights(tables[5]); a = sh.Step()
sh.SetOscillationWeightsDevice(dev[5].data_ptr(), spec.osc_n); b = sh.Step()
for _ in range(4):
    c = sh.Step()  # ... and once it rests
print(f"  LLH host {a!r} device {b!r} resting {c!r}  equal: {a == b == c}", flush=True)
