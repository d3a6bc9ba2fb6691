import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
if "--torch" in sys.argv:
    import torch; torch.cuda.set_device(0)
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
spec = synth.CONFIGS["c2"].scaled(100000, n_params=32)
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); mc0 = sh.GetMCArray(); sh.AddData(mc0)
par[:] = synth.params(spec, 3); l0 = sh.Step()
uid = eng.comm_unique_id(); print("uid", len(uid))
eng.comm_init(uid, 0, 1)
l1 = sh.Step(); print("llh without/with comm:", l0, l1, "equal", l0 == l1)
t0=time.perf_counter()
for i in range(100): sh.Step()
print("ms/step with 1-rank comm", (time.perf_counter()-t0)*10)
import ctypes
print("maps librccl:", [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l][:1], [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][:1])
