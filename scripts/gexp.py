"""A/B helper: fused-kernel time of one configuration (or a shard of it) for the library named by M3HIP_LIB and the brick
size named by M3HIP_BRICK_GROUPS.  usage: python scripts/gexp.py <config> <events> [spl]   (spl: TSpline3 only)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
cfg, ev = sys.argv[1], int(sys.argv[2])
kw = {"mixed": False} if len(sys.argv) > 3 and sys.argv[3] == "spl" else {}
if os.environ.get("M3_GEXP_OSC_N"):  # size of the pre-staged oscillation table (config 5: 204 000 entries; the others 4 000)
    kw["osc_n"] = int(os.environ["M3_GEXP_OSC_N"])
spec = synth.CONFIGS[cfg].scaled(ev, **kw)
engine = None
if os.environ.get("M3_GEXP_REPLICAS"):  # flush replicas of the bin vector (1 .. 64) instead of the engine's choice
    from mach3_amd.engine import Engine
    engine = Engine(0)
    engine._chk(engine.lib.m3hip_debug_set_hist_replicas(engine.h, int(os.environ["M3_GEXP_REPLICAS"])))
eng, mono, sh, par = build_from_synth(spec, engine=engine)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps = [synth.params(spec, i + 1) for i in range(16)]
eng.timing_enable(True)
for i in range(20):
    par[:] = laps[i % 16]; sh.Step()
res = []
for rep in range(3):
    eng.timing_reset()
    t = time.perf_counter()
    for i in range(40):
        par[:] = laps[i % 16]; sh.Step()
    step = (time.perf_counter() - t) / 40 * 1e6
    k, _ = eng.timing_read(0)
    res.append((k * 1e3, step))
info = eng.info()
k = min(r[0] for r in res)
print(f"[{os.path.basename(os.environ.get('M3HIP_LIB', 'libm3hip.so'))}] G={info.brick_groups} {cfg}{' spl' if kw else ''} E={ev}: bricks {info.n_tiles_brick} "
      f"kernel {' / '.join(f'{r[0]:.1f}' for r in res)} us -> {info.alg_bytes_fused / k / 1e3 / 8000:.3f} of 8 TB/s, step {min(r[1] for r in res):.1f} us", flush=True)
