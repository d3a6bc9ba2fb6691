"""Interleaved A/B of reweight-kernel knobs in one process (cdna guide rule 24)."""
import sys, os, time, itertools
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
spec = synth.CONFIGS["c2"].scaled(int(os.environ.get("EV","1000000")))
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps=[synth.params(spec,i+1) for i in range(16)]
info=eng.info()
variants=[(8,0,0),(16,0,0),(8,1,0),(16,1,0),(8,0,20480),(8,0,40960),(16,1,20480),(16,1,40960),(16,0,40960),(8,1,40960),(16,1,65536)]
res={v:[] for v in variants}
eng.timing_enable(True)
for rnd in range(3):
    for v in variants:
        eng.lib.m3hip_debug_set_knobs(eng.h,*v)
        for i in range(3): par[:]=laps[i]; sh.Step()
        eng.timing_reset()
        for i in range(20): par[:]=laps[i%16]; sh.Step()
        ms,n=eng.timing_read(0); res[v].append(ms)
for v in variants:
    m=np.median(res[v]); print(f"unroll={v[0]:2d} nt={v[1]} lds_pad={v[2]:6d}: median {m*1e3:7.1f} us  min {min(res[v])*1e3:7.1f} us  -> {info.alg_bytes_fused/m/1e6:7.0f} GB/s")
