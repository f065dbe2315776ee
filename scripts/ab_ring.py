#!/usr/bin/env python3
"""Same-process A/B of run-boundary settings on a 1-rank peer-to-peer ring (box-to-box spread exceeds what a
per-run overhead of 10-20 us is worth): one ring, short runs, settings alternated round-robin, wall time of
`run(steps)` + device synchronise per run as bench.py times it.

    python scripts/ab_ring.py --grid 8192x1024 --steps 20 --rounds 40 LBM_SPIN_WAIT_US=0 LBM_SPIN_WAIT_US=4000

Each positional argument is one setting: comma-separated KEY=VALUE pairs put into the environment before the run
(only knobs the library reads per call take effect: LBM_SPIN_WAIT_US).  With --per-context every setting gets its OWN ring,
created with the setting in the environment (knobs read at lbm_create: LBM_TUNE_MULTI_GEOM, LBM_TUNE_MACRO_K ...); a setting "-"
is the default environment; --single runs the grid as one periodic launch per macro-step instead of a ring."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import mpilattice_boltzmann_amd as lbm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="8192x1024")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--exchange", default="p2p")
ap.add_argument("--per-context", action="store_true")
ap.add_argument("--single", action="store_true")
ap.add_argument("settings", nargs="+")
a = ap.parse_args()
nx, ny = (int(v) for v in a.grid.split("x"))
p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
obst = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
def apply(setting):
    keys = []
    if setting != "-":
        for kv in setting.split(","):
            k, v = kv.split("=")
            os.environ[k] = v
            keys.append(k)
    return keys


def make():
    if a.single:
        return lbm.Simulation(p, obst)
    return lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange=a.exchange, strict=True)


sims = {}
if a.per_context:
    for s in a.settings:
        keys = apply(s)
        sims[s] = make()
        for k in keys:
            os.environ.pop(k, None)
        sims[s].run(a.steps)
else:
    one = make()
    one.run(a.steps)
    sims = {s: one for s in a.settings}
res = {s: [] for s in a.settings}
for r in range(a.rounds):
    for s in a.settings:
        if not a.per_context:
            apply(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sims[s].run(a.steps)
        torch.cuda.synchronize()
        res[s].append((time.perf_counter() - t0) / a.steps * 1e6)
for s in a.settings:
    v = res[s][2:]
    print(f"{s:40s} min {min(v):7.2f}  med {statistics.median(v):7.2f}  max {max(v):7.2f} us/step", flush=True)
for sim in set(sims.values()):
    sim.close()
