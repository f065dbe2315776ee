#!/usr/bin/env python3
"""Interleaved A/B sweep of tuning knobs (env vars read at lbm_create) in ONE process:
    python scripts/sweep.py --grid 8192x8192 --steps 100 --rounds 4 "LBM_TUNE_MAXBLOCKS=4096" "LBM_TUNE_MAXBLOCKS=16384,LBM_TUNE_SKEW=0" ...
Each config is created, warmed, timed `rounds` times in round-robin order; prints min / median."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpilattice_boltzmann_amd as lbm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="8192x8192")
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("configs", nargs="+")
a = ap.parse_args()
nx, ny = (int(v) for v in a.grid.split("x"))
p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
obst = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
KNOBS = ["LBM_TUNE_MAXBLOCKS", "LBM_TUNE_SKEW", "LBM_TUNE_NARROW_MAX", "LBM_TUNE_TILE_MAX", "LBM_TUNE_TILE_GEOM", "LBM_TUNE_MULTI_K", "LBM_TUNE_MULTI_REMAP", "LBM_TUNE_MACRO_K"]
res = {c: [] for c in a.configs}
for r in range(a.rounds):
    for cfg in a.configs:
        for k in KNOBS:
            os.environ.pop(k, None)
        flags = a.flags
        for kv in cfg.split(","):
            if "=" in kv:
                k, v = kv.split("=")
                if k == "FLAGS":
                    flags = int(v)
                else:
                    os.environ[k] = v
        sim = lbm.Simulation(p, obst, flags=flags)
        sim.run(10)
        sim.run(a.steps)
        ms, n = sim.partition.last_run_kernel_ms()
        res[cfg].append(ms / a.steps * 1e3)      # per lattice step (a launch may cover several steps)
        sim.close()
for cfg, v in res.items():
    print(f"{cfg:60s} min {min(v):8.1f}  med {statistics.median(v):8.1f}  max {max(v):8.1f} us/step   " + " ".join(f"{x:.0f}" for x in v), flush=True)
