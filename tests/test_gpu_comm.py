"""The in-library exchanges on the GPU: m3hip_comm_init (k_fold_hist -> ONE ncclAllReduce of [mc | w2] -> k_unpark_w2) and the
one-shot form m3hip_xch_* (k_fold_hist -> every rank stores its vector into every rank's buffer -> rank-ordered sum).

One rank: the whole RCCL code path, including the w2 parking while w2 accumulates and the host-side validator.
Two ranks: two processes, each with its own engine and event shard, on the ONE card of the test box.  RCCL refuses two
ranks on one device (NCCL's "duplicate GPU" check): that test skips with RCCL's message -- the multi-card run is the
driver's.  The one-shot exchange needs only IPC-mapped memory, which two processes on one card have: there the engine's
N = 2 logic (w2 in the same message while it accumulates, the first_w2 transition, the replica fold, the parities) runs
on a device."""
import ctypes as C
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
ranks, world, tmp, mode, update_w2, n_events = [int(r) for r in sys.argv[1].split(",")], int(sys.argv[5]), sys.argv[2], sys.argv[3], sys.argv[4] == "1", int(sys.argv[6])
spec = synth.CONFIGS["c3"].scaled(n_events, n_params=64)
cuts = m3dist.shard_ranges(spec.n_events, world)
# one engine per rank this process drives (several: "one process, several engines"; the exchange wires such peers up directly)
local = []
for rank in ranks:
    eng, mono, sh, par = build_from_synth(spec, device_id=0, event_range=cuts[rank])
    sh.SetTestStatistic(spec.test_statistic, update_w2)
    local.append((rank, eng, sh, par))

def publish(name, blob):
    f = os.path.join(tmp, name)
    open(f + ".tmp", "wb").write(blob); os.rename(f + ".tmp", f)

def fetch(name):
    f, t0 = os.path.join(tmp, name), time.time()
    while not os.path.exists(f):
        time.sleep(0.05)
        assert time.time() - t0 < 90
    return open(f, "rb").read()

try:
    if mode == "rccl":
        rank, eng, sh, par = local[0]
        if rank == 0:
            publish("uid", eng.comm_unique_id())
        eng.comm_init(fetch("uid"), rank, world)
        eng.comm_set_validate(True)
    else:  # one-shot exchange: the 64-byte IPC handles travel through files here (any transport will do)
        for rank, eng, sh, par in local:
            publish(f"handle{{rank}}", eng.xch_create(rank, world))
        handles = [fetch(f"handle{{r}}") for r in range(world)]
        for rank, eng, sh, par in local:
            eng.xch_open(handles)
            eng.xch_set_timeout(20.0)       # the processes of a test box start seconds apart
except Exception as e:
    open(os.path.join(tmp, f"refused{{ranks[0]}}"), "w").write(str(e))
    sys.exit(0)

def step_all():
    # a step is a collective: every rank of this process launches before any of them waits
    for rank, eng, sh, par in local:
        sh.Reweight()
    return [sh.GetLikelihood() for rank, eng, sh, par in local]

out = {{rank: [] for rank in ranks}}
step_all()
first = {{}}
for rank, eng, sh, par in local:
    first[rank] = (sh.GetMCArray().copy(), sh.GetW2Array().copy())
    sh.AddData(first[rank][0])
for lap in (1, 2, 3, 4, 5):
    for rank, eng, sh, par in local:
        par[:] = synth.params(spec, lap)
    llhs = step_all()
    for (rank, eng, sh, par), llh in zip(local, llhs):
        out[rank].append((llh, sh.GetMCArray().copy(), sh.GetW2Array().copy()))
for rank, eng, sh, par in local:
    rel, n = eng.comm_last_validation() if mode == "rccl" else (0.0, 99)
    o = out[rank]
    np.savez(os.path.join(tmp, f"rank{{rank}}.npz"), llh=[x[0] for x in o], mc=np.stack([x[1] for x in o]), w2=np.stack([x[2] for x in o]),
             data=first[rank][0], w2_first=first[rank][1], rel=rel, n=n)
# leave together: nobody frees a buffer a peer may still be pushing into
publish(f"done{{ranks[0]}}", b"x")
for r in range(0, world, len(ranks)):
    fetch(f"done{{r}}")
for rank, eng, sh, par in local:
    eng.close()
"""


def _two_ranks(mode, update_w2, world=2, per_process=1, n_events=7000):
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "worker.py")
        open(script, "w").write(_WORKER.format(root=ROOT))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        groups = [",".join(str(r) for r in range(g, g + per_process)) for g in range(0, world, per_process)]
        procs = [subprocess.Popen([sys.executable, script, g, tmp, mode, "1" if update_w2 else "0", str(world), str(n_events)], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for g in groups]
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=300)[0].decode())
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                pytest.fail(f"{world}-rank {mode} workers timed out")
        refused = [f for f in os.listdir(tmp) if f.startswith("refused")]
        if refused:
            why = open(os.path.join(tmp, refused[0])).read()[:300]
            if mode == "rccl":
                pytest.skip("RCCL refuses two ranks on one device: " + why)
            pytest.fail("the one-shot exchange could not be set up: " + why)
        assert all(p.returncode == 0 for p in procs), "\n".join(outs)
        r0 = np.load(os.path.join(tmp, "rank0.npz"))
        # every rank holds the same exchanged vectors and the same likelihood (the one-shot sum is rank-ordered: bit-identical)
        for r in range(1, world):
            r1 = np.load(os.path.join(tmp, f"rank{r}.npz"))
            assert np.array_equal(r0["mc"], r1["mc"]) and np.array_equal(r0["w2"], r1["w2"]) and np.array_equal(r0["llh"], r1["llh"])
            assert np.array_equal(r0["data"], r1["data"]) and np.array_equal(r0["w2_first"], r1["w2_first"])
        assert r0["rel"] <= 1e-15 and r0["n"] >= 4
        # ... equal to one engine over the whole sample
        spec = synth.CONFIGS["c3"].scaled(n_events, n_params=64)
        eng, mono, sh, par = build_from_synth(spec, device_id=0)
        sh.SetTestStatistic(spec.test_statistic, update_w2)
        sh.Reweight()
        assert np.allclose(sh.GetMCArray(), r0["data"], rtol=1e-12) and np.allclose(sh.GetW2Array(), r0["w2_first"], rtol=1e-12)
        sh.AddData(r0["data"])
        for i, lap in enumerate((1, 2, 3, 4, 5)):
            par[:] = synth.params(spec, lap)
            llh = sh.Step()
            assert np.allclose(sh.GetMCArray(), r0["mc"][i], rtol=1e-12) and np.allclose(sh.GetW2Array(), r0["w2"][i], rtol=1e-12)
            assert llh == pytest.approx(float(r0["llh"][i]), rel=1e-9)
            if not update_w2:
                assert np.array_equal(r0["w2"][i], r0["w2_first"])   # frozen after the first Reweight, on every rank
        eng.close()


def test_two_ranks_all_reduce_matches_unsharded():
    _two_ranks("rccl", False)


@pytest.mark.parametrize("update_w2", [False, True])
def test_two_ranks_one_shot_exchange_matches_unsharded(update_w2):
    """two processes on one card through m3hip_xch_*: w2 travels with mc while it accumulates (first Reweight, or every one
    with UpdateW2), then the message shrinks to mc alone"""
    _two_ranks("oneshot", update_w2)


def test_four_ranks_one_shot_exchange_matches_unsharded():
    """four processes on one card: slot [parity][rank] of every rank's buffer, four push blocks per step, a pull over four
    slots -- the indexing two ranks cannot get wrong"""
    _two_ranks("oneshot", False, world=4)


@pytest.mark.parametrize("update_w2", [False, True])
def test_eight_ranks_one_shot_exchange_matches_unsharded(update_w2):
    """N = 8 -- what the driver's node runs -- rehearsed on one card: eight slots per parity in every rank's buffer, eight push
    blocks and a pull over eight slots per step, both parities, the first_w2 transition (the message shrinks from [mc | w2] to
    mc), with and without UpdateW2.  The pool allows six processes on a card, so the eight ranks are four processes driving
    two engines each: ranks of one process are recognised by their handles and wired up directly (m3hip_xch_open), ranks of
    different processes through IPC -- both kinds of peer in one exchange."""
    _two_ranks("oneshot", update_w2, world=8, per_process=2, n_events=8 * 2048 + 500)


def test_two_ranks_in_one_process_one_shot_exchange():
    """one process driving two engines (ranks 0 and 1 of two): no IPC mapping at all, the peers' buffers as they are"""
    _two_ranks("oneshot", False, world=2, per_process=2)


def test_two_engines_of_one_process_on_two_devices_one_shot_exchange():
    """one process driving one engine on each of TWO devices: the peers are found by their handles and wired up without IPC, so the
    pushing device must be given peer access to the other's buffer by hand (hipDeviceEnablePeerAccess; the IPC route gets it from
    hipIpcMemLazyEnablePeerAccess) -- without it the first push is a memory access fault.  Skipped on a box with one device."""
    from mach3_amd import dist as m3dist
    n_dev = C.c_int(0)
    capi.load_m3hip().m3hip_device_count(C.byref(n_dev))
    if n_dev.value < 2:
        pytest.skip("needs two devices in one process")
    spec = synth.CONFIGS["c3"].scaled(7000, n_params=64)
    cuts = m3dist.shard_ranges(spec.n_events, 2)
    local = []
    for rank in (0, 1):
        eng, mono, sh, par = build_from_synth(spec, device_id=rank, event_range=cuts[rank])
        local.append((eng, sh, par))
    handles = [eng.xch_create(rank, 2) for rank, (eng, sh, par) in enumerate(local)]
    for eng, sh, par in local:
        eng.xch_open(handles)
        eng.xch_set_timeout(20.0)
    ref_eng, ref_mono, ref_sh, ref_par = build_from_synth(spec, device_id=0)
    for lap in range(4):
        for eng, sh, par in local:
            par[:] = synth.params(spec, lap)
            sh.Reweight()  # a step is a collective: both launch before either waits
        llh = [sh.GetLikelihood() for eng, sh, par in local]
        ref_par[:] = synth.params(spec, lap)
        want = ref_sh.Step()
        if lap == 0:
            for eng, sh, par in local:
                sh.AddData(sh.GetMCArray())
            ref_sh.AddData(ref_sh.GetMCArray())
            continue
        assert llh[0] == llh[1] and llh[0] == pytest.approx(want, rel=1e-9)
        assert np.array_equal(local[0][1].GetMCArray(), local[1][1].GetMCArray())
        assert np.allclose(local[0][1].GetMCArray(), ref_sh.GetMCArray(), rtol=1e-12)
    for eng, sh, par in local:
        eng.close()
    ref_eng.close()


_SILENT_PEER = r"""
import os, sys, time
sys.path.insert(0, {root!r})
from mach3_amd import synth
from mach3_amd.engine import build_from_synth
tmp = sys.argv[1]
spec = synth.CONFIGS["c3"].scaled(3000, n_params=32)
eng, mono, sh, par = build_from_synth(spec, device_id=0)
f = os.path.join(tmp, "handle1")
open(f + ".tmp", "wb").write(eng.xch_create(1, 2)); os.rename(f + ".tmp", f)
t0 = time.time()
while not os.path.exists(os.path.join(tmp, "done")) and time.time() - t0 < 90:
    time.sleep(0.1)            # owns a buffer, never pushes
eng.close()
"""


def test_one_shot_exchange_reports_a_missing_peer(fresh_engine):
    """a rank whose peer never pushes gets M3HIP_ECOMM after the time-out instead of a hung device"""
    import time
    spec = synth.CONFIGS["c3"].scaled(3000, n_params=32)
    eng, mono, sh, par = build_from_synth(spec, engine=fresh_engine)
    with tempfile.TemporaryDirectory() as tmp:
        script = os.path.join(tmp, "peer.py")
        open(script, "w").write(_SILENT_PEER.format(root=ROOT))
        peer = subprocess.Popen([sys.executable, script, tmp], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                                stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        try:
            h0 = eng.xch_create(0, 2)
            t0 = time.time()
            while not os.path.exists(os.path.join(tmp, "handle1")):
                time.sleep(0.05)
                assert time.time() - t0 < 80 and peer.poll() is None, peer.stdout.read().decode() if peer.poll() is not None else "peer too slow"
            eng.xch_open([h0, open(os.path.join(tmp, "handle1"), "rb").read()])
            eng.xch_set_timeout(0.5)
            t1 = time.time()
            with pytest.raises(capi.M3HipError, match="did not arrive within 0.5"):
                sh.Step()
            assert time.time() - t1 < 1.9, "the configured time-out, not the 2 s default"
            # the Reweight -> GetMCArray flow must not hand out this rank's partial bins either
            sh.Reweight()
            with pytest.raises(capi.M3HipError, match="did not arrive"):
                sh.GetMCArray()
            eng.xch_destroy()
        finally:
            open(os.path.join(tmp, "done"), "w").write("x")
            peer.wait(timeout=60)
    sh.Reweight()                   # back to the one-rank path
    assert np.isfinite(sh.GetLikelihood())


@pytest.mark.parametrize("n_ranks", [2, 4])
def test_bench_ranks_on_one_card_with_check(n_ranks):
    """bench.py's N > 1 path end to end on the one card of the test box at a reduced size: ONE workload for the whole --gpus
    series (configs[2], the N = 1 line's: `value` over N is a strong-scaling curve), and from four ranks on configs[3] -- four
    samples, 800 bins -- under `extra` with its own step time, roofline and work rate.  torch.distributed over gloo for the
    rendez-vous, the engine's one-shot exchange for the bins (RCCL refuses several ranks on one device), events sharded by
    responses with the cuts counted on rank 0 and broadcast, a time-bounded spin-up whose step count rank 0 decides for
    everybody (every step is a collective), `--check`: the sharded likelihood of three laps equals the one-rank likelihood
    of the same sample (1e-9 relative).  Timing means nothing with the ranks on one GPU; the logic does."""
    import json
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, M3_BENCH_BACKEND="gloo", M3_BENCH_EXCHANGE="oneshot", HSA_ENABLE_IPC_MODE_LEGACY="0")
    n_events = 150_000 * n_ranks
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--events", str(n_events),
                        "--steps", "10", "--warmup", "2", "--spinup-seconds", "0.5", "--check", "--strict"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == n_ranks and "one-shot" in line["config"]["comm"]
    assert line["config"]["ranks_seen"] == n_ranks
    assert "configs[2]" in line["config"]["workload"] and line["config"]["total_events"] == n_events      # the same workload for every N
    assert line["check"]["ok"] and line["check"]["max_rel_diff"] <= 1e-9
    assert line["config"]["events_per_gpu"] < n_events and line["roofline"]["avg_kernel_ms_by_rank"]["min"] > 0
    assert "extra_failed" not in line
    if n_ranks >= 4:
        c4 = line["extra"]["configs[3]"]
        assert "configs[3]" in c4["workload"] and "4 samples" in c4["workload"] and "800 bins" in c4["workload"]
        assert c4["ms_per_step"] > 0 and c4["event_steps_per_s"] > 0 and c4["roofline"]["avg_kernel_ms"] > 0 and c4["roofline"]["frac"] > 0
        assert c4["events_per_gpu"] < 4 * n_events and "one-shot" in c4["comm"]
    else:
        assert "configs[3]" not in line.get("extra", {})


def test_bench_line_survives_a_stop_inside_an_extra_leg():
    """From four ranks on bench.py runs configs[3] under `extra` AFTER the headline numbers are in hand.  A driver whose time-out
    fires inside that leg (eight ranks building 4e7 events on a cold node) must still find the headline: rank 0 prints the line
    (flushed, `"final": false`) before the collective legs and the complete line at the end.  Here the leg stalls and the whole
    process group is stopped from outside as soon as the first line has arrived."""
    import json
    import signal
    import socket
    import threading
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, M3_BENCH_BACKEND="gloo", M3_BENCH_EXCHANGE="oneshot", HSA_ENABLE_IPC_MODE_LEGACY="0", M3_BENCH_TEST_STALL_LEG_S="300")
    p = subprocess.Popen([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--events", "400000",
                          "--steps", "6", "--warmup", "2", "--spinup-seconds", "0.3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                         start_new_session=True)
    got = []
    t = threading.Thread(target=lambda: got.append(p.stdout.readline()), daemon=True)
    t.start()
    t.join(400)
    # stop exactly what this test started: the launcher and its descendants (the ranks may sit in sessions of their own)
    def descendants(root):
        kids = {}
        for d in os.listdir("/proc"):
            if d.isdigit():
                try:
                    with open(f"/proc/{d}/stat") as f:
                        ppid = int(f.read().rsplit(")", 1)[1].split()[1])
                    kids.setdefault(ppid, []).append(int(d))
                except (OSError, ValueError, IndexError):
                    pass
        out, todo = [], [root]
        while todo:
            for k in kids.get(todo.pop(), []):
                out.append(k)
                todo.append(k)
        return out
    mine = descendants(p.pid) + [p.pid]
    for pid in mine:
        try:
            os.kill(pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
    p.wait(30)
    import time
    for _ in range(600):  # ... and wait until they are gone, and with them their hold on the GPU, before the next test starts processes of its own
        if not any(os.path.exists(f"/proc/{pid}") for pid in mine if pid != p.pid):
            break
        time.sleep(0.1)
    time.sleep(1.0)
    assert got and got[0].strip(), "no line before the leg"
    line = json.loads(got[0])
    assert line["final"] is False and line["n_gpus"] == 4 and line["value"] > 0 and line["config"]["ranks_seen"] == 4
    assert line["roofline"]["frac"] > 0 and "configs[2]" in line["config"]["workload"] and line["config"]["headline"] == "configs[2]"


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher around it (WORLD_SIZE unset): the process starts the two ranks itself as
    fresh children (it never touches the GPU), relays rank 0's line, and the line says two ranks took part -- the first
    multi-GPU contact must not silently measure one GPU."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(M3_BENCH_BACKEND="gloo", M3_BENCH_EXCHANGE="oneshot", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--events", "200000", "--steps", "6", "--warmup", "2",
                        "--spinup-seconds", "0.3", "--check"], env=env, capture_output=True, text=True, timeout=500)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE line on stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["ranks_seen"] == 2
    assert line["check"]["ok"]
