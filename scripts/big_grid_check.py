#!/usr/bin/env python3
"""Ad-hoc maximum-size check (not part of the suite: ~45 GB of device state per context):
a grid whose plane exceeds 2 GiB, so that the multi kernel's 32-bit byte offsets use bit 31.
lbm_multi_kernel<3> (7 steps = 3 + 4) and <4> (7 = 4 + 3) against the one-step kernel on the same deck, bit for bit.

    python scripts/big_grid_check.py [--grid 16384x36864]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpilattice_boltzmann_amd as lbm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="16384x36864")
ap.add_argument("--steps", type=int, default=7)
a = ap.parse_args()
nx, ny = (int(v) for v in a.grid.split("x"))
p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
t = time.time()
obst = np.zeros((ny, nx), np.int32)
rng = np.random.default_rng(7)
idx = rng.integers(0, nx * ny, size=nx * ny // 200)
obst.reshape(-1)[idx] = 1
obst[0, :] = obst[-1, :] = 1
obst[:, 0] = obst[:, -1] = 1
print(f"deck {nx}x{ny}: {nx * ny / 2**30:.2f} Gi cells, plane {nx * ny * 4 / 2**30:.2f} GiB ({time.time() - t:.0f} s)", flush=True)

out = {}
for name, k in (("multi", "3"), ("multi4", "4"), ("one-step", "0")):
    os.environ["LBM_TUNE_MULTI_K"] = k
    s = lbm.Simulation(p, obst)
    d = s.partition.describe()
    t = time.time()
    av = s.run(a.steps)
    print(f"{name}: kernel {d['kernel']}  {a.steps} steps in {time.time() - t:.2f} s  av[-1] = {av[-1]:.9e}", flush=True)
    out[name] = (s.local_cells().view(np.uint32), av)
    s.close()
same = True
step = 1 << 24
a1 = out["one-step"][0].reshape(-1)
for which in ("multi", "multi4"):
    a0 = out[which][0].reshape(-1)
    for i in range(0, a0.size, step):
        if not np.array_equal(a0[i:i + step], a1[i:i + step]):
            same = False
            print(which, "cells differ in chunk", i // step, flush=True)
            break
    print(which, "cells bit-identical:", same, " av rel diff:", float(np.max(np.abs(out[which][1] - out["one-step"][1]) / out["one-step"][1])))
sys.exit(0 if same else 1)
