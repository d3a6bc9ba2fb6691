import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
if "--torch" in sys.argv:
    import torch; torch.cuda.set_device(0)
from mach3_amd import synth, capi
from mach3_amd.engine import build_from_synth
spec = synth.CONFIGS["c2"].scaled(int(os.environ.get("EV","200000")))
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps=[synth.params(spec,i+1) for i in range(8)]
acc=np.zeros(6); out=np.zeros(6)
import ctypes as C
eng.lib.m3hip_debug_device_span(eng.h, 1, None); dev=[]
if "--timing" in sys.argv: eng.timing_enable(True)
for i in range(10): par[:]=laps[i%8]; sh.Step()
t0=time.perf_counter(); n=100
for i in range(n):
    par[:]=laps[i%8]; sh.Step()
    eng.lib.m3hip_debug_host_times(eng.h, out.ctypes.data); acc+=out
    d=(C.c_double*5)(); eng.lib.m3hip_debug_device_span(eng.h, 0, C.cast(d, C.POINTER(C.c_double))); dev.append(list(d))
el=time.perf_counter()-t0
print("ms/step", 1e3*el/n, "phases us", (acc/n*1e6).round(1), "device span us median", np.median(np.array(dev)[:,0]).round(1))
dv=np.array(dev); off=np.min(dv[:,4]-dv[:,2])  # host-dev clock offset from the fastest completion
sub=dv[:,1]+off-dv[:,3]; vis=dv[:,4]-(dv[:,2]+off)
print("  launch->k_prep start us: median %.1f max %.1f | k_llh end->flag seen us: median %.1f max %.1f" % (np.median(sub), sub.max(), np.median(vis), vis.max()))
