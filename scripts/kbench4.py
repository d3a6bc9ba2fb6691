"""Brick kernel knob sweep on a C3-shaped shard (ragged occupancy 1/16, mixed TSpline3/TF1)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
cfgs = os.environ.get("CFGS", "c3,c5").split(",")
for cfg, kw in [(c, {}) for c in cfgs]:
    spec = synth.CONFIGS[cfg].scaled(ev, **kw)
    eng, mono, sh, par = build_from_synth(spec)
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    laps=[synth.params(spec,i+1) for i in range(16)]
    info=eng.info()
    eng.timing_enable(True)
    for unroll, pad in ((8, 0), (16, 0)):
        eng.lib.m3hip_debug_set_knobs(eng.h, unroll, 1, pad)
        for i in range(20): par[:]=laps[i%16]; sh.Step()
        eng.timing_reset()
        for i in range(40): par[:]=laps[i%16]; sh.Step()
        f,_=eng.timing_read(0)
        print(cfg, "E=%d S=%d T=%d bricks=%d" % (ev, info.n_splines, info.n_tf1, info.n_tiles_brick), f"knob {unroll} lds_pad {pad}: fused {f*1e3:.1f} us ({info.alg_bytes_fused/f/1e6:.0f} GB/s algorithmic)", flush=True)
    eng.close()
