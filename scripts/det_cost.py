"""Cost of the deterministic fill (m3hip_set_deterministic) on configs[1] and on a configs[2] shard: step time with and without."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
for cfg, ev in (("c2", 1_000_000), ("c3", 2_000_000)):
    spec = synth.CONFIGS[cfg].scaled(ev)
    eng, mono, sh, par = build_from_synth(spec)
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    laps = [synth.params(spec, i + 1) for i in range(16)]
    for det in (0, 1, 0, 1):
        eng.set_deterministic(bool(det))
        for i in range(30): par[:] = laps[i % 16]; sh.Step()
        t0 = time.perf_counter()
        n = 200
        for i in range(n): par[:] = laps[i % 16]; sh.Step()
        dt = (time.perf_counter() - t0) / n
        print(f"{cfg} E={ev} deterministic={det}: {dt*1e6:.1f} us/step", flush=True)
    eng.close()
