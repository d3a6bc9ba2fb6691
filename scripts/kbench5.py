"""Nearly uniform ragged sample (config-2 shape with a fraction of each event's responses dropped, as the
reference drops flat splines): identity-padded uniform tiles (default) vs bricks (M3HIP_PAD_MIN_FILL=2)."""
import sys, os, time
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for occ in (0.9, 0.75):
    for pad in ("0.78", "2"):
        os.environ["M3HIP_PAD_MIN_FILL"] = pad
        spec = synth.CONFIGS["c2"].scaled(ev, occupancy=occ)
        eng, mono, sh, par = build_from_synth(spec)
        sh.Reweight(); sh.AddData(sh.GetMCArray())
        laps=[synth.params(spec,i+1) for i in range(16)]
        info=eng.info()
        eng.timing_enable(True)
        for i in range(20): par[:]=laps[i%16]; sh.Step()
        eng.timing_reset()
        t=time.perf_counter()
        for i in range(100): par[:]=laps[i%16]; sh.Step()
        us=(time.perf_counter()-t)/100*1e6
        f,_=eng.timing_read(0)
        print(f"occupancy {occ} pad_min_fill {pad}: tiles {info.n_tiles_uniform}u ({info.n_tiles_padded} padded) / {info.n_tiles_brick}b, S={info.n_splines} (+{info.pad_responses} identity): "
              f"kernel {f*1e3:.1f} us ({info.alg_bytes_fused/f/1e6:.0f} GB/s algorithmic), step {us:.1f} us", flush=True)
        eng.close()
