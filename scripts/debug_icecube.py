import sys, os, math
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np
import oracle_binding as ob
from mach3_amd import synth, capi
from mach3_amd.engine import build_from_synth, SampleHandlerFD
spec = synth.CONFIGS["c2"].scaled(3000, n_params=37)
eng, mono, sh, par = build_from_synth(spec, chunk_events=2048)
lib = ob.load("serial")
# one sample per bin -> per-sample likelihood == per-bin statistic
sh = SampleHandlerFD(eng, 200, np.arange(201, dtype=np.int32), par, capi.kIceCube)
sh.Reweight(); data = sh.GetMCArray(); sh.AddData(data)
par[:] = synth.params(spec, 7); sh.Reweight()
got = sh.GetLikelihood(); mc, w2 = sh.GetMCArray(), sh.GetW2Array()
per = sh._per_sample.copy()
o = np.array([lib.m3o_test_stat_llh(2, data[i], mc[i], w2[i]) for i in range(200)])
d = np.abs(per-o); idx = np.argsort(-d)[:8]
print("total", got, o.sum(), per.sum())
for i in idx: print(i, data[i], mc[i], w2[i], "gpu", per[i], "oracle", o[i])
