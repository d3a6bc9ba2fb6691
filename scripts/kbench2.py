import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
spec = synth.CONFIGS["c2"].scaled(int(os.environ.get("EV","1000000")))
from mach3_amd.engine import Engine
eng=Engine(0)
eng.lib.m3hip_debug_set_hist_replicas(eng.h, int(os.environ.get('HIST_REP','16')))
eng, mono, sh, par = build_from_synth(spec, engine=eng)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps=[synth.params(spec,i+1) for i in range(16)]
info=eng.info()
eng.timing_enable(True)
for v in [(8,0,0),(8,1,0),(16,1,65536)]:
    eng.lib.m3hip_debug_set_knobs(eng.h,*v)
    for i in range(3): par[:]=laps[i]; sh.Step()
    eng.timing_reset()
    import time
    t0=time.perf_counter()
    for i in range(50): par[:]=laps[i%16]; sh.Step()
    wall=(time.perf_counter()-t0)/50
    f,_=eng.timing_read(0)
    for i in range(3): par[:]=laps[i]; mono.Evaluate(); mono.SynchroniseMemTransfer()
    eng.timing_reset()
    for i in range(20): par[:]=laps[i%16]; mono.Evaluate(); mono.SynchroniseMemTransfer()
    e,_=eng.timing_read(1)
    print(v, f"step {wall*1e6:.1f} us  fused {f*1e3:.1f} us ({info.alg_bytes_fused/f/1e6:.0f} GB/s)   eval-only {e*1e3:.1f} us ({info.alg_bytes_eval/e/1e6:.0f} GB/s)")
