"""Brick kernel time against responses per brick: the config-3 shape at several occupancies (same 1024-event bricks, more
or fewer stages each) -- separates what a brick costs per response from what it costs per brick."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
ev = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
for occ in (1 / 32, 1 / 16, 1 / 8, 1 / 4):
    spec = synth.CONFIGS["c3"].scaled(ev, occupancy=occ)
    eng, mono, sh, par = build_from_synth(spec)
    sh.Reweight(); sh.AddData(sh.GetMCArray())
    laps = [synth.params(spec, i + 1) for i in range(16)]
    info = eng.info()
    eng.timing_enable(True)
    for i in range(20): par[:] = laps[i % 16]; sh.Step()
    eng.timing_reset()
    for i in range(40): par[:] = laps[i % 16]; sh.Step()
    f, _ = eng.timing_read(0)
    nb = info.n_tiles_brick
    print(f"occupancy 1/{round(1/occ)}: {nb} bricks, {(info.n_splines + info.n_tf1) / max(nb,1):.0f} responses per brick, fused {f*1e3:.1f} us, "
          f"{info.alg_bytes_fused/f/1e6:.0f} GB/s algorithmic ({info.alg_bytes_fused/f/1e6/8000:.3f}), {f*1e3/max(nb,1)*512:.1f} us per brick slot", flush=True)
    eng.close()
