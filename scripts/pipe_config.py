#!/usr/bin/env python3
"""A synthetic workload through the asynchronous batch API, nothing else in the process: ms per step at a given depth.
usage: pipe_config.py <n> <outlier_ratio> <batch> <depth> <steps> [distinct_batches]
(config 3: 50000 0.99 1 3 24; config 4: 5000 0.9 128 2 24).  One JSON line."""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")


def main():
    import torch

    n, rho, B, depth, steps = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    nbuf = int(sys.argv[6]) if len(sys.argv) > 6 else 4
    dev = torch.device("cuda", 0)
    P = tp.RobustRegistrationSolver.Params(noise_bound=0.01, cbar2=1.0, estimate_scaling=False, rotation_gnc_factor=1.4,
                                           rotation_max_iterations=100, rotation_cost_threshold=0.005)
    solver = tp.RobustRegistrationSolver(P, device=-1)
    solver.set_pipeline_depth(depth)
    bufs = []
    for k in range(nbuf):
        ss, dd = [], []
        for b in range(B):
            pr = tp.synth_problem(1234 + 977 * k + b, n, rho, 0.01)
            ss.append(pr["src"].T)
            dd.append(pr["dst"].T)
        bufs.append((torch.from_numpy(np.ascontiguousarray(np.concatenate(ss))).to(dev),
                     torch.from_numpy(np.ascontiguousarray(np.concatenate(dd))).to(dev)))
    offsets = np.arange(B, dtype=np.int64) * n
    sizes = np.full(B, n, dtype=np.int32)

    def loop(count):
        tickets, last = [], None
        for k in range(count):
            if len(tickets) == depth:
                last = solver.wait(tickets.pop(0))
            s_t, d_t = bufs[k % nbuf]
            tickets.append(solver.submit_batch(s_t.data_ptr(), d_t.data_ptr(), offsets, sizes))
        while tickets:
            last = solver.wait(tickets.pop(0))
        return last

    loop(depth * nbuf)
    times = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        last = loop(steps)
        torch.cuda.synchronize()
        times.append(1e3 * (time.perf_counter() - t0) / steps)
    print(json.dumps(dict(n=n, rho=rho, batch=B, depth=depth, steps=steps, ms_per_step=round(float(np.median(times)), 4),
                          repeats=[round(t, 4) for t in times], clique0=int(last[0].clique_size),
                          finisher=os.environ.get("TEASER_HIP_FINISHER", "1"),
                          hwq=os.environ.get("GPU_MAX_HW_QUEUES", "default"))), flush=True)
    del solver  # (TEASER_HIP_HOST_TRACE prints when the handle is destroyed)


if __name__ == "__main__":
    main()
