#!/usr/bin/env python3
"""Host-resident loop of bench.py without torch (page-locked pool from the HIP runtime): ms per step for
device inputs, host inputs with depth in flight, and host inputs with depth + 1 (one staged).  Run under
`rocprofv3 --kernel-trace --memory-copy-trace` for the timeline.   usage: host_loop_probe.py [steps] [B] [n]"""
import collections, importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
tp = importlib.import_module("teaser-plusplus_amd")
from util import HipBuffers
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
modes = sys.argv[4].split(",") if len(sys.argv) > 4 else ["dev", "host0", "host1"]
P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                       rotation_max_iterations=100, rotation_cost_threshold=0.005)
s = tp.RobustRegistrationSolver(P)
D = int(os.environ.get("DEPTH", "2"))
s.set_pipeline_depth(D)
mem = HipBuffers()
pool_h, pool_d = [], []
for k in range(8):
    src = np.empty((B * n, 3)); dst = np.empty((B * n, 3))
    for b in range(B):
        pr = tp.synth_problem(1000 + k * B + b, n, float(os.environ.get("RHO", "0.95")), 0.01)
        src[b * n:(b + 1) * n] = pr["src"].T; dst[b * n:(b + 1) * n] = pr["dst"].T
    pool_h.append((mem.pinned(src), mem.pinned(dst))); pool_d.append((mem.device(src), mem.device(dst)))
off = np.arange(B, dtype=np.int64) * n; sz = np.full(B, n, dtype=np.int32)
LOG = []
def loop(pool, host, extra, count):
    tickets = collections.deque()
    for k in range(count):
        if len(tickets) == D + extra:
            t = tickets.popleft(); t0 = time.perf_counter(); s.wait(t); LOG.append(("W%d" % t, t0, time.perf_counter()))
        a, b = pool[k % len(pool)]
        t0 = time.perf_counter(); t = s.submit_batch(a, b, off, sz, host=host); LOG.append(("S%d" % t, t0, time.perf_counter()))
        tickets.append(t)
    while tickets:
        t = tickets.popleft(); t0 = time.perf_counter(); s.wait(t); LOG.append(("W%d" % t, t0, time.perf_counter()))
for name in modes:
    pool, host, extra = {"dev": (pool_d, False, 0), "host0": (pool_h, True, 0), "host1": (pool_h, True, 1)}[name]
    loop(pool, host, extra, 6)
    t0 = time.perf_counter(); loop(pool, host, extra, steps); dt = time.perf_counter() - t0
    print("%s: %.4f ms/step" % (name, 1e3 * dt / steps))
    base = LOG[-40][1]
    print(" ".join("%s[%.0f+%.0f]" % (nm, 1e6 * (a - base), 1e6 * (b - a)) for nm, a, b in LOG[-40:-10]))
    LOG.clear()
