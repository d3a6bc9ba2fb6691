"""Per-phase launch time of the re-binning pass from a rocprofv3 kernel trace of scripts/f4_ragged.py: its first 200 launches of the pass
have the shift parameter drawn afresh every step (about half of the events change bin), the last 200 walk it like a chain's.
usage: python scripts/f4_phases.py <dir with *kernel_trace.csv>"""
import csv, glob, statistics, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted((r for r in csv.DictReader(open(f)) if "k_rebin" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
name = rows[0]["Kernel_Name"][:60] if rows else "?"
print(f"{name}: {len(d)} launches; shift drawn afresh (first 200): avg {sum(d[:200]) / 200:.1f} us, median {statistics.median(d[:200]):.1f}; "
      f"chain-like walk (the rest): avg {sum(d[200:]) / max(1, len(d[200:])):.1f} us, median {statistics.median(d[200:]):.1f}")
