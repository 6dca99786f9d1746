#!/usr/bin/env python3
"""estimate_scaling = true (the K7 scale stage, registration.cc:410-425): wall time per solve for
  (a) batches of small problems (one sorting workgroup each; TEASER_SCALE_BATCH=0 runs them one after the other),
  (b) single large problems (device-wide radix sort).  GPU only.  One JSON line per configuration.
  (c) `bench`: the three problems of bench.py's `configs.scale` line (N = 10 k, 95 % outliers, noise 0.013), each solved
      synchronously with every stage timed, then the same three through the two-lane pipeline.
usage: profile_scale.py [small|large|all|bench]"""
import importlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tp = importlib.import_module("teaser-plusplus_amd")


def run(n, batch, reps=3, seed=77):
    P = tp.RobustRegistrationSolver.Params(noise_bound=0.02, cbar2=1.0, estimate_scaling=True,
                                           rotation_gnc_factor=1.4, rotation_max_iterations=100,
                                           rotation_cost_threshold=0.005)
    s = tp.RobustRegistrationSolver(P)
    probs = [tp.synth_problem(seed + b, n, 0.8, 0.01) for b in range(batch)]
    srcs, dsts = [p["src"] for p in probs], [p["dst"] * 1.5 for p in probs]
    sols = s.solve_batch(srcs, dsts)
    s.set_profiling(True)
    walls, tim = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        sols = s.solve_batch(srcs, dsts)
        walls.append(time.perf_counter() - t0)
        tim.append(s.get_profile()["tim_graph_ms"])
    print(json.dumps(dict(n=n, batch=batch, scale_batch=os.environ.get("TEASER_SCALE_BATCH", "1"),
                          wall_ms=1e3 * float(np.median(walls)), scale_plus_graph_ms=float(np.median(tim)),
                          scale0=float(sols[0].scale), valid=int(sum(bool(o.valid) for o in sols)))), flush=True)


def run_bench_problems(reps=3):
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    class A:
        pass

    solver, wl = bench.scale_workload(tp, A(), 0)
    dev = torch.device("cuda", 0)
    bufs = [(torch.from_numpy(s).to(dev), torch.from_numpy(d).to(dev)) for s, d in wl["pool"]]
    keep = ("tim_aux_ms", "tim_graph_ms", "heuristic_ms", "peel_ms", "colour_ms", "exact_ms", "rotation_ms",
            "translation_ms", "total_ms")
    def sync_pass(label):
        for i, (s_t, d_t) in enumerate(bufs):
            solver.set_profiling(0)
            out = solver.solve_batch_device(s_t.data_ptr(), d_t.data_ptr(), wl["offsets"], wl["sizes"])
            rows, walls = [], []
            solver.set_profiling(1)
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = solver.solve_batch_device(s_t.data_ptr(), d_t.data_ptr(), wl["offsets"], wl["sizes"])
                walls.append(1e3 * (time.perf_counter() - t0))
                pf = solver.get_profile()
                rows.append([float(pf[k]) for k in keep])
            solver.set_profiling(0)
            o = out[0]
            print(json.dumps(dict(when=label, problem=i, wall_ms_profiled=[round(w, 3) for w in walls],
                                  stages_per_rep=[dict(zip(keep, [round(v, 3) for v in r])) for r in rows],
                                  scale=float(o.scale), clique=int(o.clique_size), edges=int(o.num_edges),
                                  valid=int(o.valid), exact_run=int(o.clique_exact_run),
                                  heuristic=int(o.heuristic_size))), flush=True)

    def pipelined(host):
        # the pipelined loop bench.py times (depth 2; host inputs: one more batch staged)
        pinned = [(tp.PinnedArray(s), tp.PinnedArray(d)) for s, d in wl["pool"]] if host else None
        cap = 3 if host else 2
        for steps in (6, 12):
            tickets = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                if len(tickets) == cap:
                    solver.wait(tickets.pop(0))
                s_t, d_t = (pinned if host else bufs)[k % len(bufs)]
                tickets.append(solver.submit_batch(s_t.data_ptr(), d_t.data_ptr(), wl["offsets"], wl["sizes"], host=host))
            while tickets:
                solver.wait(tickets.pop(0))
            print(json.dumps(dict(pipelined_steps=steps, host=host,
                                  ms_per_step=round(1e3 * (time.perf_counter() - t0) / steps, 3))), flush=True)

    sync_pass("fresh handle")
    solver.set_pipeline_depth(2)
    pipelined(False)
    sync_pass("after the device-resident pipelined loop")
    pipelined(True)
    sync_pass("after the host-resident pipelined loop")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "bench":
        run_bench_problems()
    if what in ("small", "all"):
        run(100, 64)
        run(200, 256)
        run(500, 128)
        run(724, 64)
    if what in ("large", "all"):
        run(2000, 1)
        run(10000, 1)
