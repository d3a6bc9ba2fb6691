"""Where a brick's workgroup spends its time: phase stamps of the diagnostic build (`make stamps`,
build/libm3hip_stamps.so, loaded through M3HIP_LIB).  Prints the mean / median time of every phase over all
bricks of one launch.  Shares of a diagnostic build, not a timing of the product.
usage: M3HIP_LIB=build/libm3hip_stamps.so python scripts/brick_stamps.py [events] [config]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from mach3_amd import synth
from mach3_amd.engine import build_from_synth

ev = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
cfg = sys.argv[2] if len(sys.argv) > 2 else "c3"
spec = synth.CONFIGS[cfg].scaled(ev)
eng, mono, sh, par = build_from_synth(spec)
sh.Reweight()
sh.AddData(sh.GetMCArray())
for i in range(10):
    par[:] = synth.params(spec, i + 1)
    sh.Step()
eng.sync()
nb = int(eng.info().n_tiles_brick)
n = min(nb, 16384)
buf = np.zeros(n * 16, np.uint64)
rc = eng.lib.m3hip_debug_read_stamps(C.c_void_p(buf.ctypes.data), C.c_uint64(buf.size))
assert rc == 0, rc
t = buf.reshape(n, 16).astype(np.int64)
G = int(eng.info().brick_groups)
names = {1: "prologue: first round trip, tables of stage 0, barrier", 2: "(stamp 2: unused)", 3: "stage 0 (waits for the first quads)", 4: "stage 1", 5: "stage 2",
         6: "stage 3", 8: "last stage", 9: "TF1 rows", 10: "event tail (norm, osc, bin, LDS fill) + barrier", 11: "flush of the bin vector"}
tot = (t[:, 11] - t[:, 0]) / 100.0
if G > 1:
    d13 = (t[:, 13] - t[:, 8]) / 100.0
    ok = (t[:, 13] >= t[:, 8]) & (t[:, 13] <= t[:, 11])
    print(f"  G = {G}: slab 0 (TF1 rows + tail) after the last stage: mean {d13[ok].mean():.2f} us")
print(f"{cfg} E={ev}: {nb} bricks; brick lifetime mean {tot.mean():.1f} us, median {np.median(tot):.1f} us")
# a brick writes only the stamps of the stages it has (3 + s for its first four stages except the last one, 8 after the last);
# slots it did not write hold what an earlier launch left there: a stamp counts when it lies between its predecessor and the end
acc = {k: [] for k in names}
for r in t:
    prev = r[0]
    for k in sorted(names):
        if prev <= r[k] <= r[11]:
            acc[k].append((r[k] - prev) / 100.0)
            prev = r[k]
for k in sorted(names):
    d = np.array(acc[k])
    if d.size and k != 2:
        print(f"  {names[k]:55s} mean {d.mean():6.2f} us  median {np.median(d):6.2f} us  ({d.size:5d} bricks; {100 * d.sum() / tot.sum():4.1f} % of all brick time)")
first, last = t[:, 0].min(), t[:, 11].max()
print(f"  launch span {((last - first) / 100.0):.1f} us; bricks in flight on average {tot.sum() / ((last - first) / 100.0):.0f}")
# occupancy over the launch and the gaps between successive workgroups of one CU (slot 12: HW_ID | XCC_ID << 32)
hw = buf.reshape(n, 16)[:, 12]
cu = ((hw >> 32) & 0xF).astype(np.int64) * 4096 + ((hw >> 8) & 0xFF).astype(np.int64)   # xcc, (se, sh, cu)
t0, t1 = t[:, 0], t[:, 11]
span = last - first
edges = np.linspace(first, last, 21)
occ = [(np.clip(np.minimum(t1, b) - np.maximum(t0, a), 0, None).sum() / (b - a)) for a, b in zip(edges[:-1], edges[1:])]
print("  bricks in flight per twentieth of the launch: " + " ".join(f"{o:.0f}" for o in occ))
ncu = len(np.unique(cu))
idle2, idle1, gaps = 0, 0, []
for c in np.unique(cu):
    m = cu == c
    ev_t = np.concatenate([t0[m], t1[m]])
    ev_d = np.concatenate([np.ones(m.sum(), np.int64), -np.ones(m.sum(), np.int64)])
    o = np.lexsort((ev_d, ev_t))
    ev_t, ev_d = ev_t[o], ev_d[o]
    lvl = np.cumsum(ev_d)
    # time at level < 2 between the CU's first start and its last start (the tail after the last start is counted separately)
    last_start = t0[m].max()
    for i in range(len(ev_t) - 1):
        a, b = ev_t[i], min(ev_t[i + 1], last_start)
        if b > a and lvl[i] < 2:
            (gaps.append((b - a) / 100.0))
            idle1 += (b - a) * (2 - lvl[i])
    idle2 += (2 * (last - last_start) - np.clip(t1[m] - last_start, 0, None).sum())
# phase of a brick's start inside the lifetime of the brick that shares its CU at that moment (0 / 1: the two run in step,
# 0.5: one streams while the other is in its prologue or its tail)
ph = []
for c in np.unique(cu):
    m = np.where(cu == c)[0]
    o = m[np.argsort(t0[m])]
    for i, b in enumerate(o):
        for b2 in o[max(0, i - 3):i]:
            if t0[b2] < t0[b] < t1[b2]:
                ph.append((t0[b] - t0[b2]) / (t1[b2] - t0[b2]))
ph = np.array(ph)
hist_ph, _ = np.histogram(ph, bins=10, range=(0, 1))
print("  phase of a brick's start inside its CU neighbour's lifetime, tenths: " + " ".join(f"{100 * h / max(1, ph.size):.0f}%" for h in hist_ph))
print(f"  {ncu} CUs seen; slot-time idle between a CU's workgroups {100 * idle1 / (2 * ncu * span):.1f} % of all slot time "
      f"(gap mean {np.mean(gaps) if gaps else 0:.2f} us, median {np.median(gaps) if gaps else 0:.2f} us, {len(gaps)} gaps); "
      f"idle behind a CU's last start (tail) {100 * idle2 / (2 * ncu * span):.1f} %; ramp (first start after launch start) "
      f"{100 * sum((t0[cu == c].min() - first) * 2 for c in np.unique(cu)) / (2 * ncu * span):.1f} %")
eng.close()
