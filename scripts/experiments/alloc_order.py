#!/usr/bin/env python3
"""Does WHERE the two grids live move the step time?  (Same-process A/B runs showed the first context of a process
≈ 3 % faster than the second, whichever library it belonged to.)  Creates contexts of the 8192 x 8192 deck one after the
other — some behind a dummy allocation that shifts their addresses — prints the grid addresses (LBM_DEBUG_ADDR=1) and
times each context in turn, several rounds.

    python scripts/experiments/alloc_order.py [--grid 8192x8192] [--steps 60]"""
import argparse
import os
import sys

os.environ["LBM_DEBUG_ADDR"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import mpilattice_boltzmann_amd as lbm  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="8192x8192")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--rounds", type=int, default=3)
a = ap.parse_args()
nx, ny = (int(v) for v in a.grid.split("x"))
p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
obst = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
free = lbm.count_free_cells(obst)


def timed(part):
    part.run(a.steps)
    ms, n = part.last_run_kernel_ms()
    return ms / a.steps * 1e3


ctxs, pads = [], []
for label, pad_bytes in (("A (first)", 0), ("B", 0), ("C behind a 1 MiB pad", 1 << 20), ("D behind a 96 MiB pad", 96 << 20), ("E behind a 1 GiB pad", 1 << 30)):
    if pad_bytes:
        pads.append(torch.empty(pad_bytes, dtype=torch.uint8, device="cuda"))
    sys.stderr.write(f"{label}: ")
    ctxs.append((label, lbm.Partition(p, free, obst)))
for _, part in ctxs:
    part.run(12)
for r in range(a.rounds):
    print("round", r, "  ".join(f"{label}: {timed(part):.1f}" for label, part in ctxs), flush=True)
label0, part0 = ctxs.pop(0)
part0.close()
sys.stderr.write("F (after A was freed): ")
ctxs.append(("F (after A was freed)", lbm.Partition(p, free, obst)))
ctxs[-1][1].run(12)
for r in range(a.rounds):
    print("round", r, "  ".join(f"{label}: {timed(part):.1f}" for label, part in ctxs), flush=True)
