"""Byte inventory of the brick format, on the CPU: the distinct 128-B lines one step of k_reweight_brick touches per brick,
stream by stream, from the engine's own host-side layout (m3hip_debug_layout) of a shard of a BASELINE configuration.
What the FETCH_SIZE counter reads beyond this sum is lines fetched more than once.
usage: python scripts/brick_inventory.py [config] [events]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

from helpers import debug_layout
from mach3_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
ev = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
spec = synth.CONFIGS[cfg].scaled(ev)
cx, npts = synth.knots(spec)
ch = synth.monolith_chunk(spec, 0, ev)
rc, err, lay = debug_layout(spec.n_params, spec.n_knots, npts, ch)
assert rc == 0, err
words = lay["words"]
HDR0 = 8
par = synth.params(spec, 1)[: spec.n_params].astype(np.float32)
xk = cx.reshape(spec.n_params, spec.n_knots)
seg = np.array([min(max(int(np.searchsorted(xk[p, : npts[p]], par[p], side="left")) - 1, 0), int(npts[p]) - 2) if npts[p] > 1 else 0 for p in range(spec.n_params)])
tot = dict(quads_used=0, quad_lines=0, tf1_used=0, tf1_lines=0, words=0, resp_words=0, run_tab=0, pair_tab=0, stage_counts=0, hdr=0, tf1_hdr=0, tf1_sigma=0, tf1_codes=0, runs=0, stages=0)
nb = lay["n_bricks"]
for b in range(nb):
    spl_off, tf1_off, w_off, n_f4, n_f2, nev, e0, nw = [int(v) for v in lay["bdesc"][b]]
    w = words[w_off: w_off + nw]
    n_stages, off_tf1, G = int(w[0]), int(w[1]), int(w[3])
    lines = set()
    for s in range(n_stages):
        h0, off_runs, off_grp, off_resp = [int(v) for v in w[HDR0 + 4 * s: HDR0 + 4 * s + 4]]
        n_resp, n_runs = h0 & 0xFFFF, (h0 >> 16) & 0x3FF
        runs = w[off_runs: off_runs + 2 * n_runs].reshape(-1, 2)
        resp = np.zeros(n_resp, np.uint16)   # per-lane records (m3hip_layout.h: stage_record_words)
        nrec, lanes = (n_resp + 1023) // 1024, min(1024, n_resp)     # record [l] = nrec uint16: the lane's words next to each other
        n16 = nrec * lanes
        rec16 = w[off_resp: off_resp + (n16 + 1) // 2].view(np.uint16)
        rec = rec16[:n16].reshape(lanes, nrec)
        for u in range(nrec):
            n_u = max(0, min(lanes, n_resp - 1024 * u))
            resp[1024 * u: 1024 * u + n_u] = rec[:n_u, u]
        ends = np.flatnonzero(resp & 1) + 1
        first_r = np.concatenate([[0], ends[:-1]])
        for rn in range(n_runs):
            base, stride, p = int(np.int32(runs[rn][0])), int(runs[rn][1]) & 0xFFFF, int(runs[rn][1]) >> 16
            q0 = spl_off + base + seg[p] * stride + int(first_r[rn])
            n = int(ends[rn] - first_r[rn])  # entries of the run: its responses + (where the planner pads) the rest of its last line
            tot["quads_used"] += 16 * int(((resp[first_r[rn]: ends[rn]] >> 2) < 5120).sum())
            for L in range((16 * q0) // 128, (16 * (q0 + n) - 1) // 128 + 1):
                lines.add(L)
        tot["resp_words"] += 4 * ((n16 + 1) // 2)
        tot["run_tab"] += 8 * n_runs
        tot["pair_tab"] += 4 * (5120 // 64)
        tot["stage_counts"] += 512 * G
        tot["runs"] += n_runs
        tot["stages"] += 1
    tot["quad_lines"] += 128 * len(lines)
    tot["tf1_used"] += 8 * n_f2
    tot["tf1_lines"] += 128 * ((8 * (tf1_off + n_f2) + 127) // 128 - (8 * tf1_off) // 128)
    tot["words"] += 4 * nw
    tot["hdr"] += 4 * (HDR0 + 4 * n_stages)
    cnt_words = 0 if int(w[5]) else 256 * G             # the count bytes: absent where the counts ride in the sigma words
    tot["tf1_hdr"] += 4 * (32 * G + cnt_words)          # wave-block bases + one count byte per lane-slot
    tot["tf1_sigma"] += 2 * 1024 * G
    tot["tf1_codes"] += 4 * (nw - off_tf1 - 32 * G - cnt_words - 512 * G)   # upper bound: to the end of the (128-B padded) stream
S, T = ch["n_splines"], ch["n_tf1"]
alg = 16 * S + 8 * T + 8 * ev
sample_side = ev * 8   # the packed word {bin, osc, <= 3 norm indices}; the three arrays it replaces were 4 + 4 + 2 * slots bytes
print(f"{cfg} E={ev}: {nb} bricks, S={S} T={T}; per brick (KB): algorithmic {alg / nb / 1e3:.1f}")
for k in ("quads_used", "quad_lines", "tf1_used", "tf1_lines", "words", "resp_words", "run_tab", "pair_tab", "stage_counts", "hdr", "tf1_hdr", "tf1_sigma", "tf1_codes"):
    print(f"  {k:12s} {tot[k] / nb / 1e3:8.2f}")
print(f"  sample side (packed word)       {sample_side / nb / 1e3:8.2f}")
total = tot["quad_lines"] + tot["tf1_lines"] + tot["words"] + sample_side
print(f"  distinct lines in total {total / nb / 1e3:.1f} KB per brick = {total / alg:.3f} x algorithmic; runs per brick {tot['runs'] / nb:.0f}, stages {tot['stages'] / nb:.2f}, "
      f"responses per run {S / max(tot['runs'], 1):.1f}")
