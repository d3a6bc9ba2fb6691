"""Step latency of one libm3hip build (M3HIP_LIB=...): run several builds in ONE gpurun call, boxes differ."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np, ctypes as C
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
cfg = sys.argv[2] if len(sys.argv) > 2 else "c2"
spec = synth.CONFIGS[cfg].scaled(ev)
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps=[synth.params(spec,i+1) for i in range(16)]
t=time.perf_counter(); i=0
while time.perf_counter()-t<1.5: par[:]=laps[i%16]; sh.Step(); i+=1
for rep in range(3):
    t=time.perf_counter()
    for i in range(500): par[:]=laps[i%16]; sh.Step()
    us=(time.perf_counter()-t)/500*1e6
    print(os.path.basename(os.environ.get("M3HIP_LIB","libm3hip.so")), f"{cfg} E={ev} step {us:.1f} us", flush=True)
eng.lib.m3hip_debug_device_span(eng.h, 1, None); dev=[]
for i in range(200):
    par[:]=laps[i%16]; sh.Step()
    d=(C.c_double*5)(); eng.lib.m3hip_debug_device_span(eng.h, 0, C.cast(d, C.POINTER(C.c_double))); dev.append(d[0])
print("   device span k_prep start -> k_llh end: median %.1f us" % np.median(dev))
