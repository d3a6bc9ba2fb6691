"""f2 driver for the profiler: the parameter handler's prior + correlated throw on the device (k_par_stage / k_par_rows / k_par_finish)
at BASELINE configs[4]'s 1 040 parameters, launch + collect in a loop.  usage: python scripts/par_rows_loop.py [n=1040] [steps=2000]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from mach3_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1040
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
rng = np.random.default_rng(3)
a = rng.standard_normal((n, n))
cov = a @ a.T / n + 0.3 * np.eye(n)
eng = Engine(0)
sl = eng.par_create()
eng.par_upload(sl, 0, np.linalg.inv(cov))
eng.par_upload(sl, 1, np.linalg.cholesky(cov))
d, z = rng.standard_normal(n), rng.standard_normal(n)
for i in range(200):
    eng.par_launch(sl, d, z); eng.par_result(sl, n)
t = time.perf_counter()
for i in range(steps):
    eng.par_launch(sl, d, z); eng.par_result(sl, n)
el = (time.perf_counter() - t) / steps * 1e6
print(f"f2: prior + correlated throw on the device at n = {n}: {el:.1f} us per launch + collect; matrices 2 x {n * (n + 1) // 2 * 8 / 1e6:.2f} MB (lower triangles)", flush=True)
eng.close()
