import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
for cfg, ev, kw in (("c3", 1000000, {}), ("c5", 1000000, {}), ("c3", 1000000, dict(mixed=False)),):
    spec = synth.CONFIGS[cfg].scaled(ev, **kw)
    eng, mono, sh, par = build_from_synth(spec)
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    laps=[synth.params(spec,i+1) for i in range(16)]
    info=eng.info()
    eng.timing_enable(True)
    for i in range(20): par[:]=laps[i%16]; sh.Step()
    eng.timing_reset()
    for i in range(30): par[:]=laps[i%16]; sh.Step()
    f,_=eng.timing_read(0)
    for i in range(3): par[:]=laps[i]; mono.Evaluate(); mono.SynchroniseMemTransfer()
    eng.timing_reset()
    for i in range(30): par[:]=laps[i%16]; mono.Evaluate(); mono.SynchroniseMemTransfer()
    e,_=eng.timing_read(1)
    print(cfg, kw, "S=%d T=%d bricks=%d" % (info.n_splines, info.n_tf1, info.n_tiles_brick), f"fused {f*1e3:.1f} us ({info.alg_bytes_fused/f/1e6:.0f} GB/s)   eval-only {e*1e3:.1f} us ({info.alg_bytes_eval/e/1e6:.0f} GB/s)")
    eng.close()
