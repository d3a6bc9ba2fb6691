"""The in-library exchange (m3hip_comm_init: k_fold_hist -> ONE ncclAllReduce of [mc | w2] -> k_unpark_w2) on the GPU.

One rank: the whole code path, including the w2 parking while w2 accumulates and the host-side validator.
Two ranks: two processes, each with its own engine and event shard, on the ONE card of the test box.  RCCL may refuse
two ranks on one device (NCCL's "duplicate GPU" check); the test then skips with RCCL's message -- the multi-card run is
the driver's -- but where it is allowed this is a real two-rank all-reduce through the library."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from mach3_amd import capi, synth
from mach3_amd.engine import build_from_synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n_samples,n_bins", [(1, 200), (4, 200), (3, 2500)])
def test_one_rank_exchange_with_w2_and_validator(fresh_engine, n_samples, n_bins):
    """w2 travels in the same all-reduce as mc while it accumulates (first Reweight, and every one with UpdateW2);
    bins, w2 and likelihood must come out as without a communicator, for few and for many bins (1 or 32 replicas)."""
    spec = synth.CONFIGS["c3"].scaled(6000, n_params=48, n_samples=n_samples, n_bins=n_bins)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    sh.Reweight()
    mc0, w20 = sh.GetMCArray().copy(), sh.GetW2Array().copy()
    sh.AddData(mc0)
    par[:] = synth.params(spec, 3)
    ref = sh.Step()
    mc_ref, w2_ref = sh.GetMCArray().copy(), sh.GetW2Array().copy()
    assert np.array_equal(w2_ref, w20)                       # frozen at the first Reweight (SampleHandlerFD.cpp:342)

    eng2, mono2, sh2, par2 = build_from_synth(spec, device_id=0)
    eng2.comm_init(eng2.comm_unique_id(), 0, 1)
    eng2.comm_set_validate(True)
    sh2.Reweight()                                           # w2 accumulates: [mc | w2] in one message
    assert np.allclose(sh2.GetMCArray(), mc0, rtol=1e-12) and np.allclose(sh2.GetW2Array(), w20, rtol=1e-12)
    sh2.AddData(mc0)
    par2[:] = par
    got = sh2.Step()
    assert got == pytest.approx(ref, rel=1e-12)
    assert np.allclose(sh2.GetMCArray(), mc_ref, rtol=1e-12) and np.allclose(sh2.GetW2Array(), w2_ref, rtol=1e-12)
    rel, n = eng2.comm_last_validation()
    assert n >= 2 and rel <= 1e-15
    eng2.close()


_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
from mach3_amd import synth, dist as m3dist
from mach3_amd.engine import build_from_synth
rank, world, tmp = int(sys.argv[1]), 2, sys.argv[2]
spec = synth.CONFIGS["c3"].scaled(7000, n_params=64)
e0, e1 = m3dist.shard_ranges(spec.n_events, world)[rank]
eng, mono, sh, par = build_from_synth(spec, device_id=0, event_range=(e0, e1))
uid_file = os.path.join(tmp, "uid")
if rank == 0:
    uid = eng.comm_unique_id()
    open(uid_file + ".tmp", "wb").write(uid); os.rename(uid_file + ".tmp", uid_file)
else:
    t0 = time.time()
    while not os.path.exists(uid_file):
        time.sleep(0.05)
        assert time.time() - t0 < 60
    uid = open(uid_file, "rb").read()
try:
    eng.comm_init(uid, rank, world)
except Exception as e:
    open(os.path.join(tmp, f"refused{{rank}}"), "w").write(str(e))
    sys.exit(0)
eng.comm_set_validate(True)
out = []
sh.Reweight(); data = sh.GetMCArray().copy(); sh.AddData(data)
for lap in (1, 2, 3):
    par[:] = synth.params(spec, lap)
    llh = sh.Step()
    out.append((llh, sh.GetMCArray().copy(), sh.GetW2Array().copy()))
rel, n = eng.comm_last_validation()
np.savez(os.path.join(tmp, f"rank{{rank}}.npz"), llh=[o[0] for o in out], mc=np.stack([o[1] for o in out]), w2=np.stack([o[2] for o in out]),
         data=data, rel=rel, n=n)
eng.close()
"""


def test_two_ranks_all_reduce_matches_unsharded():
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(_WORKER.format(root=ROOT))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, str(r), tmp], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in (0, 1)]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=240)[0].decode())
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail("two-rank RCCL workers timed out")
        refused = [f for f in os.listdir(tmp) if f.startswith("refused")]
        if refused:
            pytest.skip("RCCL refuses two ranks on one device: " + open(os.path.join(tmp, refused[0])).read()[:200])
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        r0, r1 = (np.load(os.path.join(tmp, f"rank{r}.npz")) for r in (0, 1))
        # every rank holds the same all-reduced vectors and the same likelihood
        assert np.array_equal(r0["mc"], r1["mc"]) and np.array_equal(r0["w2"], r1["w2"]) and np.array_equal(r0["llh"], r1["llh"])
        assert r0["rel"] <= 1e-15 and r0["n"] >= 4
        # ... equal to one engine over the whole sample
        spec = synth.CONFIGS["c3"].scaled(7000, n_params=64)
        eng, mono, sh, par = build_from_synth(spec, device_id=0)
        sh.Reweight()
        assert np.allclose(sh.GetMCArray(), r0["data"], rtol=1e-12)
        sh.AddData(r0["data"])
        for i, lap in enumerate((1, 2, 3)):
            par[:] = synth.params(spec, lap)
            llh = sh.Step()
            assert np.allclose(sh.GetMCArray(), r0["mc"][i], rtol=1e-12) and np.allclose(sh.GetW2Array(), r0["w2"][i], rtol=1e-12)
            assert llh == pytest.approx(float(r0["llh"][i]), rel=1e-9)
        eng.close()
