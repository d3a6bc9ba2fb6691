"""Soak of the one-shot exchange on one card: N processes (one rank each), a sharded ragged sample, many steps through m3hip_xch_*;
every rank's likelihood of every step goes to a file and rank 0 compares them at the end -- a lost or stale slot, a parity slip or
a time-out shows as a differing likelihood or an M3HIP_ECOMM.  usage: python scripts/xch_soak.py [ranks=2] [steps=20000] [events=20000]"""
import os, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from mach3_amd import synth, dist as m3dist
from mach3_amd.engine import build_from_synth
rank, world, tmp, steps, events = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
spec = synth.CONFIGS["c3"].scaled(events, n_params=64)
eng, mono, sh, par = build_from_synth(spec, device_id=0, event_range=m3dist.shard_ranges(spec.n_events, world)[rank])
def publish(name, blob):
    f = os.path.join(tmp, name); open(f + ".tmp", "wb").write(blob); os.rename(f + ".tmp", f)
def fetch(name):
    f, t0 = os.path.join(tmp, name), time.time()
    while not os.path.exists(f):
        time.sleep(0.02); assert time.time() - t0 < 120
    return open(f, "rb").read()
publish(f"h{rank}", eng.xch_create(rank, world))
eng.xch_open([fetch(f"h{r}") for r in range(world)])
eng.xch_set_timeout(20.0)
sh.Reweight(); sh.AddData(sh.GetMCArray())
laps = [synth.params(spec, 1 + i) for i in range(64)]
out = np.zeros(steps)
t0 = time.perf_counter()
for i in range(steps):
    par[:] = laps[(i * 7 + i // 64) %% 64]
    out[i] = sh.Step()
el = time.perf_counter() - t0
np.save(os.path.join(tmp, f"llh{rank}.npy"), out)
publish(f"done{rank}", b"x")
for r in range(world): fetch(f"done{r}")
print(f"rank {rank}: {steps} steps, {1e6 * el / steps:.1f} us/step", flush=True)
eng.close()
''' % ROOT

ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
events = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
import numpy as np
with tempfile.TemporaryDirectory() as tmp:
    w = os.path.join(tmp, "w.py"); open(w, "w").write(WORKER)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, w, str(r), str(ranks), tmp, str(steps), str(events)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(ranks)]
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            for q in procs: q.kill()
            print("TIMED OUT"); sys.exit(1)
    if any(p.returncode for p in procs):
        print("\n".join(o[-1500:] for o in outs)); sys.exit(1)
    llh = [np.load(os.path.join(tmp, f"llh{r}.npy")) for r in range(ranks)]
    same = all(np.array_equal(llh[0], l) for l in llh[1:])
    print("".join(o.splitlines()[-1] + "\n" for o in outs), end="")
    print(f"one-shot exchange soak: {ranks} ranks x {steps} steps on one card: likelihoods {'bit-identical on all ranks' if same else 'DIFFER'}; "
          f"{len(np.unique(llh[0]))} distinct values, all finite: {bool(np.isfinite(llh[0]).all())}")
    sys.exit(0 if same else 1)
