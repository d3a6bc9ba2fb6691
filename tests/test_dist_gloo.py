"""N > 1 path on CPUs: world_size-2 gloo.  Each rank takes its event shard, fills its partial
bin vector (the oracle stands in for the device here), one all-reduce, then every rank
evaluates the same likelihood -- which must equal the unsharded oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_events, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import oracle_binding as ob
    from mach3_amd import capi, dist as m3dist, synth

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = synth.CONFIGS["c3"].scaled(n_events, n_params=64)
    e0, e1 = m3dist.shard_ranges(n_events, world)[rank]
    # bench.py's cuts: rank 0 counts the responses once and broadcasts; must equal what a rank would compute by itself
    import importlib.util
    spec_ = importlib.util.spec_from_file_location("m3_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(bench)
    big = synth.CONFIGS["c3"].scaled(300_000)
    cut_bcast, cut_own = bench.shard_of(big, rank, world, dist), bench.shard_of(big, rank, world)
    mono, samp, par, ch, ev = ob.build_oracle_from_synth(spec, e0, e1)
    lib = ob.load("serial")
    out = []
    data = None
    for lap in (0, 1, 2):
        par[:] = synth.params(spec, lap)
        samp.reweight()
        mc, w2 = samp.mc, samp.w2
        m3dist.allreduce_bins(mc, w2)
        if lap == 0:
            data = mc.copy()
        llh = sum(lib.m3o_test_stat_llh(capi.kBarlowBeeston, data[i], mc[i], w2[i]) for i in range(len(mc)))
        out.append((llh, mc.sum()))
    q.put((rank, (e0, e1), out, tuple(cut_bcast), tuple(cut_own)))
    dist.destroy_process_group()


def test_two_rank_event_sharding_matches_unsharded():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    from mach3_amd import capi, synth

    n_events, world = 5000, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_events, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards tile the sample on 1024-event boundaries
    assert res[0][1] == (0, 3072) and res[1][1] == (3072, 5000)
    # bench.py's broadcast cuts: what each rank would have computed alone, contiguous, covering the sample
    assert all(r[3] == r[4] for r in res)
    assert res[0][3][0] == 0 and res[0][3][1] == res[1][3][0] and res[1][3][1] == 300_000 and res[0][3][1] % 1024 == 0
    # unsharded reference
    spec = synth.CONFIGS["c3"].scaled(n_events, n_params=64)
    mono, samp, par, ch, ev = ob.build_oracle_from_synth(spec, 0, n_events)
    samp.set_test_statistic(capi.kBarlowBeeston)
    exp = []
    for lap in (0, 1, 2):
        par[:] = synth.params(spec, lap)
        samp.reweight()
        if lap == 0:
            samp.add_data(samp.mc)
        exp.append((samp.get_likelihood(), samp.mc.sum()))
    for rank in range(world):
        for (llh, tot), (ellh, etot) in zip(res[rank][2], exp):
            assert tot == pytest.approx(etot, rel=1e-12)
            assert abs(llh - ellh) <= 1e-9 * max(abs(ellh), 1.0)
    assert res[0][2] == res[1][2]  # every rank holds the identical all-reduced result


def test_shard_ranges():
    from mach3_amd.dist import shard_ranges

    r = shard_ranges(10_000_000, 8)
    assert r[0][0] == 0 and r[-1][1] == 10_000_000
    assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
    assert all((a[1] - a[0]) % 1024 == 0 for a in r[:-1])
    assert max(b - a for a, b in r) - min(b - a for a, b in r) <= 8 * 1024
    # weighted: the heavy second half gets more ranks
    w = np.ones(81920)
    w[40960:] = 3.0
    rw = shard_ranges(81920, 4, w)
    loads = [w[a:b].sum() for a, b in rw]
    assert max(loads) / min(loads) < 1.1 and rw[-1][1] == 81920 and rw[0][1] == 40960
    assert shard_ranges(100, 4)[0] == (0, 100)  # fewer events than one group: rank 0 holds them all


def test_bench_shards_cover_the_sample_and_balance_the_responses():
    """bench.py's split of a configuration over ranks (what every rank computes for itself, from the generator's counts):
    contiguous, on 1024-event boundaries, the same on every rank, responses balanced to a per cent"""
    import importlib.util

    spec_ = importlib.util.spec_from_file_location("m3_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(bench)
    from mach3_amd import synth

    spec = synth.CONFIGS["c3"].scaled(600_000)
    for n in (2, 4, 8):
        r = [bench.shard_of(spec, k, n) for k in range(n)]
        assert r == [bench.shard_of(spec, k, n) for k in range(n)]
        assert r[0][0] == 0 and r[-1][1] == spec.n_events and all(a[1] == b[0] for a, b in zip(r, r[1:]))
        assert all(a[1] % 1024 == 0 for a in r[:-1])
        loads = [sum(synth.count(spec, a, b)) for a, b in r]
        assert max(loads) / min(loads) < 1.02, loads
    assert bench.shard_of(spec, 0, 1) == (0, spec.n_events)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """under a launcher, `--gpus N` with another WORLD_SIZE is an error (exit code 2), not a warning and a line for the wrong N;
    decided before anything touches a GPU, so it runs here"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 2 and "WORLD_SIZE=4" in r.stderr and r.stdout.strip() == ""
