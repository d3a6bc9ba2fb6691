import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
spec = synth.CONFIGS["c2"].scaled(20000)
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); sh.AddData(sh.GetMCArray())
if "--sleep" in sys.argv: time.sleep(1.0)
laps=[synth.params(spec,i+1) for i in range(8)]
ts=[]
t_start=time.perf_counter()
for i in range(600):
    t0=time.perf_counter(); par[:]=laps[i%8]; sh.Step(); ts.append(time.perf_counter()-t0)
ts=np.array(ts)*1e6
print("first 10:", ts[:10].round(0))
for a in range(0,600,50): print(a, "median us", np.median(ts[a:a+50]).round(1), "max", ts[a:a+50].max().round(0), "elapsed ms", round(ts[:a].sum()/1e3,1))
