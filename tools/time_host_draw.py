"""Time of the bit-exact RandomState draw of pr_rho on the host by thread count (development aid)."""
import numpy as np, time, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vimure_amd import _hostlib
print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
shape = (4, 2000, 2000, 2)
cov = np.ones(shape[:3], np.uint8)
out = np.empty(shape)
out[:] = 0
for rep in range(2):
    for thr in (1, 8, 16, 32):
        p = np.random.RandomState(5); t = time.perf_counter(); a = _hostlib.draw_pr_rho(p, shape, 0.0, cov, out=out, threads=thr); dt = time.perf_counter() - t
        print(thr, round(dt, 4), flush=True)
