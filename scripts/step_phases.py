"""Where a step's time goes on the host side: the phases m3hip_debug_host_times reports and the device span of a step, for one
configuration, with the kernel-timing events off / on every launch / on every 2nd launch (bench.py's setting) and with 16 or 64
parameter vectors in rotation.  usage: python scripts/step_phases.py <config> <events>"""
import os, sys, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
cfg, ev = sys.argv[1], int(sys.argv[2])
spec = synth.CONFIGS[cfg].scaled(ev)
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps = [synth.params(spec, i + 1) for i in range(64)]
ph = (C.c_double * 6)()
span = (C.c_double * 5)()
eng._chk(eng.lib.m3hip_debug_device_span(eng.h, 1, span))
for i in range(30):
    par[:] = laps[i % 16]; sh.Step()
for every, nl in ((0, 16), (1, 16), (2, 16), (2, 64), (0, 64)):
    eng.timing_enable(every); eng.timing_reset()
    acc = np.zeros(6); sp = 0.0
    t = time.perf_counter()
    for i in range(60):
        par[:] = laps[i % nl]; sh.Step()
        eng.lib.m3hip_debug_host_times(eng.h, ph); acc += np.array(list(ph))
        eng.lib.m3hip_debug_device_span(eng.h, 1, span); sp += span[0]
    step = (time.perf_counter() - t) / 60 * 1e6
    k, n = eng.timing_read(0)
    print(f"{cfg} E={ev} events every {every}, {nl} vectors: step {step:.1f} us, device span {sp / 60:.1f} us, kernel {k * 1e3:.1f} us ({n} timed); "
          f"host phases (us): " + " ".join(f"{x / 60 * 1e6:.1f}" for x in acc), flush=True)
