import sys, os
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np
import oracle_binding as ob
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
spec = synth.CONFIGS["c2"].scaled(3000, n_params=37)
eng, mono, sh, par = build_from_synth(spec, chunk_events=2048)
omono, osamp, opar, ch, ev = ob.build_oracle_from_synth(spec, 0, spec.n_events)
sh.RecordEventWeights(True)
sh.Reweight(); osamp.reweight()
gw = sh.GetEventWeights(0, spec.n_events); ow = osamp.event_weights()
gs = sh.GetSplineWeights(0, spec.n_events); os_ = omono.total_weights
print("event w equal:", np.array_equal(gw.view(np.uint32), ow.view(np.uint32)), "spline w equal:", np.array_equal(gs.view(np.uint32), os_.view(np.uint32)))
bad = np.nonzero(gw != ow)[0]
print("n bad", len(bad), bad[:10])
for e in bad[:5]:
    print(e, gw[e], ow[e], gs[e], os_[e], ev["osc_idx"][e], ev["norm_off"][e+1]-ev["norm_off"][e], ev["bin_idx"][e])
mc = sh.GetMCArray(); print(np.abs(mc-osamp.mc).max(), mc.sum(), osamp.mc.sum())
# recompute hist from gpu event weights on host
h = np.zeros(spec.total_bins); 
sel = (gw>0)&(ev["bin_idx"]>=0)
np.add.at(h, ev["bin_idx"][sel], gw[sel].astype(np.float64))
print("host hist from gpu weights vs gpu hist", np.abs(h-mc).max(), "vs oracle", np.abs(h-osamp.mc).max())
