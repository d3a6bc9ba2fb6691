"""Host-side re-layout (reference flat arrays -> segment-major tiles): pure data movement,
checked on the CPU against the source arrays through the m3hip_debug_layout probe."""
import os

import numpy as np
import pytest

from helpers import debug_layout, make_monolith, natural_cubic
from mach3_amd import synth

TE = 256
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_uniform_tiles(n_params, K, npts, ch, lay):
    """every uniform tile against the source arrays; an event that lacks a slot's parameter (identity-padded
    tiles) must hold the identity response there: {1,0,0,0} / {0,1}"""
    spl, tf1, meta, desc = lay["spl"], lay["tf1"], lay["meta"], lay["desc"]
    cm = ch["coeff_many"].reshape(-1, 4)
    ct = ch["coeff_tf1"].reshape(-1, 2)
    npe = ch["npe"].reshape(-1, 2)
    npt = ch["npe_tf1"].reshape(-1, 2)
    ident4, ident2 = np.array([1, 0, 0, 0], np.float32), np.array([0, 1], np.float32)
    n_pad = 0
    for t, d in enumerate(desc):
        spl_off, tf1_off, meta_off, js, jt, nev, e0, nplanes = [int(v) for v in d]
        assert nev == min(TE, ch["n_events"] - e0)
        m = meta[meta_off: meta_off + 2 * js + jt]
        # per event: parameter -> index of its response
        own_s = [{int(ch["param_no"][s]): s for s in range(int(npe[e, 1]), int(npe[e, 1] + npe[e, 0]))} for e in range(e0, e0 + nev)]
        own_t = [{int(ch["param_no_tf1"][s]): s for s in range(int(npt[e, 1]), int(npt[e, 1] + npt[e, 0]))} for e in range(e0, e0 + nev)]
        plane = 0
        seen = [0] * nev
        for j in range(js):
            p = int(m[2 * j])
            assert int(m[2 * j + 1]) == plane
            nseg = int(npts[p]) - 1
            for le in range(nev):
                s_idx = own_s[le].get(p)
                for k in range(nseg):
                    got = spl[spl_off + (plane + k) * TE + le]
                    if s_idx is None:
                        assert np.array_equal(got, ident4)
                    else:
                        exp = cm[int(ch["nknots_arr"][s_idx]) + k]
                        assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))
                seen[le] += s_idx is not None
                n_pad += s_idx is None
            for k in range(nseg):
                pad = spl[spl_off + (plane + k) * TE + np.arange(nev, TE)]
                assert (pad == ident4).all()
            plane += nseg
        assert plane == nplanes
        assert seen == [len(o) for o in own_s]           # every response of every event sits in some slot
        seen = [0] * nev
        for j in range(jt):
            p = int(m[2 * js + j])
            for le in range(nev):
                s_idx = own_t[le].get(p)
                got = tf1[tf1_off + j * TE + le]
                if s_idx is None:
                    assert np.array_equal(got, ident2)
                else:
                    assert np.array_equal(got.view(np.uint32), ct[s_idx].view(np.uint32))
                seen[le] += s_idx is not None
                n_pad += s_idx is None
        assert seen == [len(o) for o in own_t]
    return n_pad


@pytest.mark.parametrize("n_events,n_params", [(256, 8), (700, 5), (1, 3), (1024 + 17, 16)])
def test_dense_layout_roundtrip(n_events, n_params):
    spec = synth.CONFIGS["c2"].scaled(n_events, n_params=n_params)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, n_events)
    rc, err, lay = debug_layout(n_params, spec.n_knots, npts, ch)
    assert rc == 0, err
    assert lay["n_tiles"] == (n_events + TE - 1) // TE
    assert lay["spl"].shape[0] == lay["n_tiles"] * n_params * (spec.n_knots - 1) * TE
    check_uniform_tiles(n_params, spec.n_knots, npts, ch, lay)


def test_mixed_uniform_layout_with_different_knot_counts():
    # parameters with 3, 7 and 5 knots + two TF1 parameters; every event carries all of them
    xs = {0: np.array([-1.0, 0, 1]), 2: np.arange(-3.0, 4.0), 3: np.array([-2.0, -1, 0, 1, 2])}
    rng = np.random.default_rng(1)
    events = []
    for _ in range(300):
        ev = []
        for p in range(5):
            if p in xs:
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
            else:
                ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    cx, npts, ch = make_monolith(5, 7, xs, events)
    rc, err, lay = debug_layout(5, 7, npts, ch)
    assert rc == 0, err
    assert int(lay["desc"][0][7]) == 2 + 6 + 4
    check_uniform_tiles(5, 7, npts, ch, lay)


def test_layout_rejects_inconsistent_input():
    spec = synth.CONFIGS["c2"].scaled(300, n_params=4)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, 300)
    bad = dict(ch)
    bad["param_no"] = ch["param_no"].copy()
    bad["param_no"][5] = 9                      # parameter out of range
    rc, err, _ = debug_layout(4, 7, npts, bad)
    assert rc != 0 and "paramNo" in err
    bad = dict(ch)
    bad["npe"] = ch["npe"].copy()
    bad["npe"][-2] += 1                         # last event's run exceeds n_splines
    rc, err, _ = debug_layout(4, 7, npts, bad)
    assert rc != 0 and "exceeds" in err
    bad = dict(ch)
    bad["total_nknots"] = ch["total_nknots"] - 3  # knots exceed the coefficient array
    rc, err, _ = debug_layout(4, 7, npts, bad)
    assert rc != 0 and "total_nknots" in err
    npts2 = npts.copy()
    npts2[1] = -999                             # a spline on a parameter that has no knots
    rc, err, _ = debug_layout(4, 7, npts2, ch)
    assert rc != 0 and "knot count" in err


STAGE_EVENT_MAX, STAGE_CAP, STAGE_MAX_RUNS, HDR0, TF1_SKIP, PAIR_WORDS = 15, 5120, 512, 8, 255, 80


def brick_event_lists(npts, ch, lay, b):
    """Replay a brick's metadata stream on the CPU with coefficient quads instead of weights, the way
    k_reweight_brick walks it: returns, per event of the brick, the ordered list of (param, kind,
    coefficients) the kernel would multiply.  TSpline3 stages: phase A scatters response r to its slot of
    the compact array, phase B lets every lane read its event's slots (they lie next to each other); TF1: the wave-block's jagged rows, lane per event."""
    spl, tf1, words, tdict = lay["spl"], lay["tf1"], lay["words"], lay["tf1_dict"]
    spl_off, tf1_off, w_off, n_f4, n_f2, nev, e0, nw = [int(v) for v in lay["bdesc"][b]]
    assert w_off % 32 == 0 and nw % 32 == 0            # every brick's stream starts on a 128-byte line
    w = words[w_off: w_off + nw]
    out = [[] for _ in range(nev)]
    n_stages, off_tf1, G = int(w[0]), int(w[1]), int(w[3])
    assert G in (1, 2, 4) and nev <= 1024 * G
    n_blocks = 16 * G
    used = np.zeros(max(1, n_f4), bool)
    pads_total, entries_total, aligned_runs = 0, 0, True
    for s in range(n_stages):
        h0, off_runs, off_pairs, off_resp = [int(v) for v in w[HDR0 + 4 * s: HDR0 + 4 * s + 4]]
        n_resp, n_runs, rows = h0 & 0xFFFF, (h0 >> 16) & 0x3FF, h0 >> 26
        assert 0 < n_resp <= STAGE_CAP and 0 < n_runs <= STAGE_MAX_RUNS and off_runs % 2 == 0
        runs = w[off_runs: off_runs + 2 * n_runs].reshape(-1, 2)
        pairs = w[off_pairs: off_pairs + PAIR_WORDS]
        grp = (pairs & 0xFFFF).astype(np.int64)
        woff = (pairs[:n_blocks] >> 16).astype(np.int64)
        assert not (pairs[n_blocks:] >> 16).any()
        assert (grp[(n_resp + 63) // 64:] == n_runs).all()
        cb = w[off_pairs + PAIR_WORDS: off_pairs + PAIR_WORDS + 128 * G]
        cnt = np.zeros(1024 * G, np.int64)                # the way a lane fetches its G nibbles
        if G == 1:
            by = cb.view(np.uint8)
            cnt[:512], cnt[512:] = by & 15, by >> 4
        elif G == 2:
            by = cb.view(np.uint8)
            cnt[:1024], cnt[1024:] = by & 15, by >> 4
        else:
            hw = cb.view(np.uint16)
            for h in range(4):
                cnt[1024 * h: 1024 * (h + 1)] = (hw >> (4 * h)) & 15
        assert not cnt[nev:].any() and cnt.max() == rows <= STAGE_EVENT_MAX
        resp = np.zeros(n_resp, np.uint16)                # per-lane records
        nw, lanes = (n_resp + 1023) // 1024, min(1024, n_resp)     # record [l] = nw uint16: the lane's words next to each other
        n16 = nw * lanes
        assert off_resp % 32 == 0                                   # a wave's 64 records are whole 128-byte lines
        rec16 = w[off_resp: off_resp + (n16 + 1) // 2].view(np.uint16)
        assert not rec16[n16:].any()
        rec = rec16[:n16].reshape(lanes, nw)
        for u in range(nw):
            n_u = max(0, min(lanes, n_resp - 1024 * u))
            resp[1024 * u: 1024 * u + n_u] = rec[:n_u, u]
            assert not rec[n_u:, u].any()
        ell = {}
        run = -1
        n_pad = 0
        for r in range(n_resp):
            if r % 64 == 0:
                run = int(grp[r // 64])                  # the kernel: run of the group's first response ...
            wd = int(resp[r])
            slot, last_run = (wd & 0x7FFC) >> 2, wd & 1
            assert not wd & 0x8002
            base = int(np.int32(runs[run][0]))
            stride, p = int(runs[run][1]) & 0xFFFF, int(runs[run][1]) >> 16
            assert stride % 8 == 0
            nseg = int(npts[p]) - 1
            idx = [base + k * stride + r for k in range(nseg)]
            assert all(0 <= i < n_f4 for i in idx) and not used[idx].any()
            used[idx] = True
            if slot >= STAGE_CAP:                        # padding of a run's last line (runs padded to whole lines in the list):
                assert slot < STAGE_CAP + 16 and (r + 1) % 8 != 1      # dropped into the array's slack; never first in a line
                n_pad += 1
                run += last_run
                continue
            assert slot < n_resp
            assert slot not in ell                       # one slot per response
            ell[slot] = (p, "spl", np.stack([spl[spl_off + i] for i in idx]))
            run += last_run                              # ... + the runs that ended before this response
        assert run == n_runs                             # the last response closes the last run
        assert cnt.sum() == n_resp - n_pad
        pads_total += n_pad
        # phase B: every lane reads its event's slots: first slot of the wave-block + exclusive prefix sum of the counts
        # over the block's lanes (a wave scan in the kernel) + j
        for blk_i in range(n_blocks):
            lanes = np.arange(64 * blk_i, 64 * blk_i + 64)
            excl = np.cumsum(cnt[lanes]) - cnt[lanes]
            for i, le in enumerate(lanes):
                for j in range(int(cnt[le])):
                    out[le].append(ell.pop(int(woff[blk_i]) + int(excl[i]) + j))
        assert not ell                                   # every scattered weight is consumed by its event
        for e in range(nev):
            ps = [x[0] for x in out[e] if x[1] == "spl"]
            assert ps == sorted(ps)
        # planes start on 128-byte lines (start_r of a run = index of its first response, recovered from the words)
        ends = np.flatnonzero(resp & 1) + 1
        first_r = np.concatenate([[0], ends[:-1]])
        for rn in range(n_runs):
            assert (int(np.int32(runs[rn][0])) + int(first_r[rn])) % 8 == 0
        entries_total += n_resp
        aligned_runs = aligned_runs and not (first_r % 8).any()
    # runs padded to whole lines in the lists (so that a group of 64 entries is eight whole lines): all of a brick's or none,
    # and only where that costs at most 8 % more entries
    if pads_total:
        assert aligned_runs
        assert os.environ.get("M3HIP_PAD_RUNS") == "1" or pads_total * 100 <= 8 * (entries_total - pads_total)
    # TF1: jagged rows per wave-block of 64 lane-slots; slot k = 1024 h + l works on event 1024 h + sigma[k] of the brick
    n_slots = 1024 * G
    if int(w[5]):                                       # no event with more than 63 codes: the counts ride in the sigma words
        hdr = w[off_tf1: off_tf1 + 32 * G + n_slots // 2]
        s16 = hdr[32 * G:].view(np.uint16).astype(np.int64)
        cnts, sigma = s16 >> 10, s16 & 1023
    else:
        hdr = w[off_tf1: off_tf1 + 32 * G + n_slots // 4 + n_slots // 2]
        cnts = hdr[32 * G: 32 * G + n_slots // 4].view(np.uint8)
        sigma = hdr[32 * G + n_slots // 4:].view(np.uint16).astype(np.int64)
        assert sigma.max() < 1024 and (cnts.max() > 63 or G == 2)   # (bricks of 2048 events keep the byte array)
    ev_of = (np.arange(n_slots) // 1024) * 1024 + sigma
    for h in range(G):                                  # a permutation of every slab, existing events first, longest first
        sl = slice(1024 * h, 1024 * (h + 1))
        assert sorted(sigma[sl]) == list(range(1024))
        key = [(int(cnts[k]) + 1 if ev_of[k] < nev else 0) for k in range(1024 * h, 1024 * (h + 1))]
        if int(w[4]):                                   # sorted by code count ...
            assert key == sorted(key, reverse=True)
        else:                                           # ... or left in their own order (no event beyond the first batch of 16 rows)
            assert list(sigma[sl]) == list(range(1024)) and max(key) <= 17
    assert all(int(cnts[k]) == 0 for k in range(n_slots) if ev_of[k] >= nev)
    slots = 0
    for wv in range(n_blocks):
        cbase, dbase = int(hdr[wv]), int(hdr[16 * G + wv])
        assert cbase % 16 == 0 and dbase % 32 == 0 and (tf1_off + cbase) % 16 == 0   # a wave-block's rows and code dwords start on 128-byte lines
        assert cbase - slots < 16                                                    # (behind at most a line's worth of identity slots)
        for i in range(slots, cbase):
            assert np.array_equal(tf1[tf1_off + i], np.array([0, 1], np.float32))
        slots = cbase
        lanes = [k for k in range(64 * wv, 64 * wv + 64) if ev_of[k] < nev]
        kd = {k: -1 for k in lanes}
        rows = max([int(cnts[k]) for k in lanes], default=0)
        for j in range(rows):
            act = [k for k in lanes if int(cnts[k]) > j]
            if j % 4 == 0:
                actq = act
                codes = {k: int(w[dbase + i]) for i, k in enumerate(actq)}
                dbase += len(actq)
            for i, k in enumerate(act):
                cd = (codes[k] >> (8 * (j % 4))) & 0xFF
                ab = tf1[tf1_off + cbase + i]
                if cd == TF1_SKIP:
                    kd[k] += 254
                    assert np.array_equal(ab, np.array([0, 1], np.float32))
                else:
                    assert cd > 0
                    kd[k] += cd
                    out[int(ev_of[k])].append((int(tdict[kd[k]]), "tf1", ab))
            cbase += len(act)
            slots += len(act)
    assert -(-slots // 16) * 16 == int(w[2]) == n_f2
    return e0, out


def check_bricks(npts, ch, lay):
    cm = ch["coeff_many"].reshape(-1, 4)
    ct = ch["coeff_tf1"].reshape(-1, 2)
    npe = ch["npe"].reshape(-1, 2)
    npt = ch["npe_tf1"].reshape(-1, 2)
    for b in range(lay["n_bricks"]):
        e0, lists = brick_event_lists(npts, ch, lay, b)
        for le, got in enumerate(lists):
            e = e0 + le
            exp = []
            for i in range(int(npe[e, 0])):
                s = int(npe[e, 1]) + i
                p = int(ch["param_no"][s])
                k0 = int(ch["nknots_arr"][s])
                exp.append((p, "spl", cm[k0: k0 + int(npts[p]) - 1]))
            for i in range(int(npt[e, 0])):
                t = int(npt[e, 1]) + i
                exp.append((int(ch["param_no_tf1"][t]), "tf1", ct[t]))
            assert len(got) == len(exp), (e, len(got), len(exp))
            for g, x in zip(got, exp):
                assert g[0] == x[0] and g[1] == x[1]
                assert np.array_equal(np.asarray(g[2]).view(np.uint32), np.asarray(x[2]).view(np.uint32))


@pytest.mark.parametrize("n_events,cfg", [(700, "c3"), (2048 + 300, "c3"), (1500, "c5")])
def test_ragged_brick_layout(n_events, cfg):
    """Sparse random occupancy (configs 3-5): every event's product list, replayed from the brick
    stream, equals its reference-layout run: same parameters, same order, same coefficient bits."""
    spec = synth.CONFIGS[cfg].scaled(n_events)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, n_events)
    rc, err, lay = debug_layout(spec.n_params, spec.n_knots, npts, ch)
    assert rc == 0, err
    assert lay["n_bricks"] == (n_events + 1023) // 1024 and lay["n_tiles"] == 0
    check_bricks(npts, ch, lay)


@pytest.mark.parametrize("pad", [0, 1])
def test_runs_padded_to_whole_lines_or_not(pad, monkeypatch):
    """The planner pads a brick's runs to whole 128-byte lines in the stages' lists where that is cheap (M3HIP_PAD_RUNS: 0
    never, 1 always, default by cost): both forms replay to the same event lists, and the padded one has every run start on
    a multiple of eight entries."""
    monkeypatch.setenv("M3HIP_PAD_RUNS", str(pad))
    spec = synth.CONFIGS["c3"].scaled(1024 + 500)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, spec.n_events)
    rc, err, lay = debug_layout(spec.n_params, spec.n_knots, npts, ch)
    assert rc == 0, err
    check_bricks(npts, ch, lay)
    w = lay["words"]
    w0 = int(lay["bdesc"][0][2])
    entries = sum(int(w[w0 + HDR0 + 4 * s]) & 0xFFFF for s in range(int(w[w0])))
    real = int(ch["npe"][:2 * 1024:2].sum())            # TSpline3 responses of the first brick's 1024 events
    assert (entries > real) if pad else (entries == real)
    assert entries <= 1.08 * real


def test_more_runs_than_a_stage_holds():
    """2 600 parameters in 1 400 sparse events: a brick sees more than 512 distinct spline parameters, so its
    TSpline3 stages are cut by run count (the resolved run table of a stage lives in LDS), not by responses."""
    rng = np.random.default_rng(5)
    X3, X5 = np.array([-1.0, 0, 1]), np.array([-2.0, -1, 0, 1, 2])
    P = 2600
    xs = {p: (X3, X5)[p % 2] for p in range(P) if p % 5 != 4}
    events = []
    for _ in range(1400):
        ev = []
        for p in range(P):
            if rng.random() < 0.006:
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))) if p in xs
                          else (p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    cx, npts, ch = make_monolith(P, 7, xs, events)
    rc, err, lay = debug_layout(P, 7, npts, ch)
    assert rc == 0, err
    assert lay["n_bricks"] == 2 and lay["n_tiles"] == 0
    w = lay["words"]
    w0 = int(lay["bdesc"][0][2])
    stages = [int(w[w0 + HDR0 + 4 * s]) & 0x3FFFFFF for s in range(int(w[w0]))]      # n_resp | n_runs << 16 (| rows << 26)
    assert len(stages) >= 3 and all(h >> 16 <= STAGE_MAX_RUNS for h in stages) and sum(h >> 16 for h in stages) > 1024
    n_runs, n_resp = sum(h >> 16 for h in stages), sum(h & 0xFFFF for h in stages)
    assert len(stages) == -(-n_runs // STAGE_MAX_RUNS) > -(-n_resp // STAGE_CAP)   # the run count made the cuts, not the responses
    assert max(h >> 16 for h in stages) - min(h >> 16 for h in stages) < 64        # ... and the stages came out even
    check_bricks(npts, ch, lay)


def test_mixed_uniform_and_ragged_groups():
    """A chunk whose first 1024 events are dense and whose tail is ragged: uniform tiles and a
    brick side by side, stages split at the response cap."""
    P = 24
    X7 = np.arange(-3.0, 4.0)
    rng = np.random.default_rng(7)
    quad = lambda: natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))
    events = []
    for e in range(1024):
        events.append([(p, "spl", quad()) if p % 3 else (p, "tf1", (0.1 * rng.standard_normal(), 1.0)) for p in range(P)])
    for e in range(900):
        events.append([(p, "spl", quad()) if p % 3 else (p, "tf1", (0.1 * rng.standard_normal(), 1.0))
                       for p in range(P) if rng.random() < 0.5])   # too sparse for identity padding: a brick
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P) if p % 3}, events)
    rc, err, lay = debug_layout(P, 7, npts, ch)
    assert rc == 0, err
    assert lay["n_tiles"] == 4 and lay["n_bricks"] == 1
    check_uniform_tiles(P, 7, npts, ch, lay)
    check_bricks(npts, ch, lay)
    w = lay["words"]
    assert int(w[0]) == 2 and int(w[2]) > 3000  # ~7200 TSpline3 responses in two stages (equal halves), ~3600 TF1 slots


def test_ragged_requires_ascending_parameters():
    X7 = np.arange(-3.0, 4.0)
    c = natural_cubic(X7, 1 + 0.1 * X7)
    events = [[(0, "spl", c), (1, "spl", c)], [(1, "spl", c), (0, "spl", c)]]  # second event descending
    cx, npts, ch = make_monolith(2, 7, {0: X7, 1: X7}, events)
    rc, err, _ = debug_layout(2, 7, npts, ch)
    assert rc != 0 and "ascending" in err


def test_deep_brick_many_stages():
    """~390 000 responses in a sparse 1024-event group: a brick of ~90 stages (round 2's header block refused it at 62);
    the refusal beyond M3_BRICK_MAX_STAGES = 250 is exercised by tests/cpp/layout_sanitize.cpp (1.4 M responses)."""
    rng = np.random.default_rng(9)
    P, X3 = 640, np.array([-1.0, 0, 1])
    c = natural_cubic(X3, [0.9, 1.0, 1.2])
    events = [[(p, "spl", c) for p in range(P) if rng.random() < 0.6] for _ in range(1024)]
    assert sum(len(e) for e in events) > 62 * STAGE_CAP
    cx, npts, ch = make_monolith(P, 3, {p: X3 for p in range(P)}, events)
    rc, err, lay = debug_layout(P, 3, npts, ch)
    assert rc == 0, err
    assert lay["n_bricks"] == 1 and 62 < int(lay["words"][int(lay["bdesc"][0][2])]) <= 250


@pytest.mark.parametrize("groups,n_events,cfg", [(2, 2048 + 1024 + 300, "c3"), (4, 4096 + 1500, "c5"), (4, 4096, "c3")])
def test_bricks_of_several_event_groups(monkeypatch, groups, n_events, cfg):
    """G = 2 / 4: bricks of 2048 / 4096 events (lane l owns events l, l + 1024, ...), a short brick at the end;
    replayed event by event like the 1024-event ones."""
    monkeypatch.setenv("M3HIP_BRICK_GROUPS", str(groups))
    spec = synth.CONFIGS[cfg].scaled(n_events)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, n_events)
    rc, err, lay = debug_layout(spec.n_params, spec.n_knots, npts, ch)
    assert rc == 0, err
    assert lay["n_bricks"] == -(-n_events // (1024 * groups)) and lay["n_tiles"] == 0
    assert [int(d[5]) for d in lay["bdesc"]] == [min(1024 * groups, n_events - 1024 * groups * b) for b in range(lay["n_bricks"])]
    check_bricks(npts, ch, lay)


@pytest.mark.parametrize("groups,n_events,short_from,short_to,sizes", [
    (1, 2048 + 3072 + 1024 + 300, 2048, 2048 + 3072, [1024, 1024, 768, 768, 768, 768, 1024, 300]),
    (2, 2048 + 6144 + 1000, 2048, 2048 + 6144, [2048, 1536, 1536, 1536, 1536, 1000]),
    (1, 3072 + 600, 0, 3072, [768, 768, 768, 768, 600]),
])
def test_short_bricks_near_the_end_of_a_launch(monkeypatch, groups, n_events, short_from, short_to, sizes):
    """Between two given events the planner cuts bricks at three quarters of their size, four where three full ones would be (the
    engine asks for that when it knows the sample's size, so that every compute unit gets the same number of bricks:
    m3hip_upload.hip ShortBrickPlan); no brick spans the stretch's ends, and the short bricks replay event by event like the full ones."""
    monkeypatch.setenv("M3HIP_BRICK_GROUPS", str(groups))
    monkeypatch.setenv("M3HIP_DEBUG_SHORT_FROM", str(short_from))
    monkeypatch.setenv("M3HIP_DEBUG_SHORT_TO", str(short_to))
    spec = synth.CONFIGS["c3"].scaled(n_events)
    cx, npts = synth.knots(spec)
    ch = synth.monolith_chunk(spec, 0, n_events)
    rc, err, lay = debug_layout(spec.n_params, spec.n_knots, npts, ch)
    assert rc == 0, err
    assert [int(d[5]) for d in lay["bdesc"]] == sizes and lay["n_tiles"] == 0
    assert [int(d[6]) for d in lay["bdesc"]] == list(np.cumsum([0] + sizes[:-1]))
    check_bricks(npts, ch, lay)


@pytest.mark.parametrize("events,chunk,groups,bricks", [
    (5_000_000, 131072, 1, 5120),           # half of config 3 / config 4's share of one GPU in eight: 4 883 bricks = 19.07 per unit -> 20
    (5_000_000, 5_000_192, 1, 5120),        # ... in one call
    (2_500_000, 131072, 1, 2560),           # a quarter: 9.54 per unit -> 10
    (10_000_000, 131072, 1, 9766),          # config 3: 38.15 per unit, 2.2 % of the launch to gain: measured level, left alone
    (12_500_000, 131072, 4, 3052),          # one GPU's share of config 5: 11.92 per unit, 0.7 % to gain: left alone
    (1_250_000, 131072, 1, 1221),           # 4.8 rounds: too few (measured 2 % worse with short bricks)
    (5120 * 1024, 131072, 1, 5120),         # a multiple already
    (100_000_000, 131072, 1, 97657),        # 381 bricks per unit
    (2100 * 4096, 129 * 1024, 4, 2304),     # chunks that are no multiple of the brick (129 x 1024 events at G = 4): every chunk ends in a
                                            # 1024-event brick of its own (66 chunks x 33 = 2 148 bricks), the stretch starts on chunk-local boundaries
    (5150 * 2048, 131 * 1024, 2, 5376),     # ... at G = 2 (78 chunks x 66 bricks + 41 = 5 189)
])
def test_brick_count_is_a_multiple_of_the_compute_units(events, chunk, groups, bricks):
    """the rule itself, walked chunk by chunk on the host (ShortBrickPlan through m3hip_debug_short_brick_plan)"""
    import ctypes as C
    from mach3_amd import capi
    lib = capi.load_m3hip()
    nb, ns = C.c_uint64(0), C.c_uint64(0)
    assert lib.m3hip_debug_short_brick_plan(events, chunk, groups, 256, C.byref(nb), C.byref(ns)) == 0
    assert nb.value == bricks
    B = 1024 * groups
    full = events // chunk * -(-chunk // B) + -(-(events % chunk) // B) if chunk < events else -(-events // B)   # bricks are cut chunk by chunk
    assert ns.value == 4 * (bricks - full)              # a conversion of three full bricks into four short ones gains one
    if bricks != full:
        assert bricks % 256 == 0


def test_layout_under_sanitizers(tmp_path):
    """the host re-layout under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool):
    exact-size input arrays and stores, random dense / mixed / ragged / empty chunks, the stage-limit refusal"""
    import subprocess
    exe = tmp_path / "layout_sanitize"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-pthread", "-o", str(exe),
                        os.path.join(ROOT, "tests", "cpp", "layout_sanitize.cpp"), os.path.join(ROOT, "mach3_amd", "csrc", "m3hip_layout.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout + r.stderr
    assert "ERROR" not in r.stderr and "runtime error" not in r.stderr
    assert "stages exceed" in r.stdout and r.stdout.count("bricks") >= 5


def test_nearly_uniform_groups_become_identity_padded_tiles(monkeypatch):
    """Events of one interaction mode share their list of systematics, but the reference drops an event's flat
    (one-knot) splines from the monolith (SplineMonolith.cpp:160-207), so the lists differ by a few entries.
    Such groups are laid out as uniform tiles over the union list, with the identity response {1,0,0,0} / {0,1}
    where an event lacks a parameter: its float product is unchanged bit for bit.  Sparse random lists stay bricks."""
    rng = np.random.default_rng(17)
    X7, X5 = np.arange(-3.0, 4.0), np.array([-2.0, -1, 0, 1, 2])
    P = 30
    xs = {p: (X7 if p % 2 else X5) for p in range(P) if p % 5}
    events = []
    for e in range(1024 + 700):
        ev = []
        for p in range(P):
            if rng.random() < 0.12:
                continue                                    # this event's response to p is flat: dropped
            if p in xs:
                ev.append((p, "spl", natural_cubic(xs[p], 1 + 0.1 * rng.standard_normal(len(xs[p])))))
            else:
                ev.append((p, "tf1", (0.05 * rng.standard_normal(), 1.0)))
        events.append(ev)
    cx, npts, ch = make_monolith(P, 7, xs, events)
    monkeypatch.setenv("M3HIP_PAD_MIN_FILL", "0.8")    # the default (0.88) is right at this sample's fill
    rc, err, lay = debug_layout(P, 7, npts, ch)
    assert rc == 0, err
    assert lay["n_bricks"] == 0 and lay["n_tiles"] == 7
    n_pad = check_uniform_tiles(P, 7, npts, ch, lay)
    n_real = ch["n_splines"] + ch["n_tf1"]
    assert 0.08 < n_pad / (n_pad + n_real) < 0.16
    # with the padding switched off the same chunk is two bricks
    monkeypatch.setenv("M3HIP_PAD_MIN_FILL", "2")
    rc, err, lay = debug_layout(P, 7, npts, ch)
    assert rc == 0 and lay["n_bricks"] == 2 and lay["n_tiles"] == 0
    monkeypatch.setenv("M3HIP_PAD_MIN_FILL", "0.8")
    rc, err, lay = debug_layout(P, 7, npts, ch)
    assert rc == 0 and lay["n_bricks"] == 0


def test_mode_blocks_keep_their_uniform_tiles():
    """Blocks of events that share a list (interaction modes) whose boundaries do not fall on multiples of 1024:
    only the 256-event tiles that straddle a boundary (or hold sparse events) become bricks; each run of such
    tiles, cut at 1024 events, is one brick."""
    rng = np.random.default_rng(23)
    X7 = np.arange(-3.0, 4.0)
    P = 10
    quad = lambda: natural_cubic(X7, 1 + 0.1 * rng.standard_normal(7))
    list_a, list_b = [0, 1, 2, 3, 4], [3, 4, 5, 6, 7, 8]
    events = [[(p, "spl", quad()) for p in list_a] for _ in range(1000)]            # mode A: tiles 0-2 uniform, tile 3 mixed
    events += [[(p, "spl", quad()) for p in list_b] for _ in range(1300)]           # mode B: tiles 4-7 uniform, tile 8 mixed
    events += [[(p, "spl", quad()) for p in range(P) if rng.random() < 0.3] for _ in range(1500)]  # sparse tail
    cx, npts, ch = make_monolith(P, 7, {p: X7 for p in range(P)}, events)
    rc, err, lay = debug_layout(P, 7, npts, ch)
    assert rc == 0, err
    first = sorted(int(d[6]) for d in lay["desc"])
    assert first == [0, 256, 512, 1024, 1280, 1536, 1792]                            # the seven uniform tiles
    bricks = sorted((int(d[6]), int(d[5])) for d in lay["bdesc"])
    assert bricks == [(768, 256), (2048, 1024), (3072, 728)]                         # (first event, events)
    check_uniform_tiles(P, 7, npts, ch, lay)
    check_bricks(npts, ch, lay)
