#!/usr/bin/env python3
"""Randomised cross-check of the kernel family on one GPU: every case runs the same deck through the
library's own choice (lbm_multi_kernel / lbm_tile_kernel, random K / geometry) and through the one-step
kernel (LBM_TUNE_MULTI_K=0, LBM_TUNE_TILE_MAX=0), and the final populations must agree bit for bit.
No oracle involved (the one-step kernel is pinned to it by the test suite): thousands of cells x
hundreds of shapes in a minute.

    python scripts/fuzz_kernels.py [--cases 300] [--seed 1]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpilattice_boltzmann_amd as lbm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=300)
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()
rng = np.random.default_rng(a.seed)
KNOBS = ["LBM_TUNE_MULTI_K", "LBM_TUNE_TILE_MAX", "LBM_TUNE_TILE_GEOM", "LBM_TUNE_MACRO_K"]
bad = 0
for case in range(a.cases):
    kind = rng.choice(["multi", "tile", "ring"])
    if kind == "tile":
        T = int(rng.choice([8, 16]))
        nx, ny = T * int(rng.integers(1, 20)), T * int(rng.integers(1, 20))
        if ny < 3:
            ny = T * 2
        env = {"LBM_TUNE_TILE_MAX": str(1 << 30), "LBM_TUNE_TILE_GEOM": str(T * 10 + int(rng.choice([4, 8]))), "LBM_TUNE_MULTI_K": "0"}
    else:
        nx = 2 * int(rng.integers(64, 400)) if rng.random() < 0.7 else 64 * int(rng.integers(2, 12))
        ny = int(rng.integers(32, 300))
        K = int(rng.integers(1, 5))
        env = {"LBM_TUNE_TILE_MAX": "0", "LBM_TUNE_MULTI_K": str(K), "LBM_TUNE_MACRO_K": str(max(K, 2) if kind == "ring" else K)}
    steps = int(rng.integers(1, 40))
    dens = float(rng.choice([0.0, 0.002, 0.05, 0.3]))
    p = lbm.Params(nx, ny, steps, 4, float(rng.choice([0.1, 1.0])), float(rng.choice([0.005, 0.05, 0.5])), float(rng.choice([0.7, 1.3, 1.85, 1.97])))
    obst = (rng.random((ny, nx)) < dens).astype(np.int32)
    if rng.random() < 0.5:
        obst[0, :] = obst[-1, :] = 1
    if rng.random() < 0.2:
        obst[ny - 2, :] = 1
    if obst.all():
        obst[1, 1] = 0
    res = []
    for variant in ("fast", "one-step"):
        for k in KNOBS:
            os.environ.pop(k, None)
        flags, kw = 0, {}
        if variant == "fast":
            os.environ.update(env)
            if kind == "ring":
                flags, kw = lbm._capi.FLAG_FORCE_HALO, {"exchange": "rccl"}
        else:
            os.environ.update({"LBM_TUNE_MULTI_K": "0", "LBM_TUNE_TILE_MAX": "0"})
        s = lbm.Simulation(p, obst, flags=flags, **kw)
        desc = s.partition.describe()["kernel"] if variant == "fast" else None
        av = np.concatenate([s.run(steps // 2), s.run(steps - steps // 2)]) if steps > 1 else s.run(steps)
        res.append((s.local_cells().view(np.uint32).copy(), av, desc))
        s.close()
    same = np.array_equal(res[0][0], res[1][0])
    avd = float(np.max(np.abs(res[0][1] - res[1][1]) / np.maximum(np.abs(res[1][1]), 1e-30)))
    if not same or avd > 1e-6:
        bad += 1
        print(f"MISMATCH case {case}: {kind} {nx}x{ny} steps {steps} env {env} dens {dens} kernel {res[0][2]} cells_same={same} av_rel={avd:.2e}", flush=True)
    elif case % 25 == 0:
        print(f"case {case}: {kind} {nx}x{ny} steps {steps} {res[0][2]} ok", flush=True)
print(f"{a.cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
