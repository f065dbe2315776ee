#!/usr/bin/env python3
"""Per-step timing of the step path in its different forms on one GPU (tuning aid, not the bench).

    python scripts/measure.py --grid 8192x8192 --steps 200 [--mode single|ring|ring-rccl|ring-torch] [--flags N]

`ring` / `ring-rccl` run the row-partitioned code path (interior + edge launches, peer-to-peer or RCCL exchange on a side
stream) on a 1-rank ring, which is what one rank of an N-GPU run executes per step for a grid of
this size: e.g. --grid 8192x1024 is one rank's share of the 8192x8192 deck on 8 GPUs."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mpilattice_boltzmann_amd as lbm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="8192x8192", help="NXxNY")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="single", choices=["single", "ring", "ring-rccl", "ring-torch"],
                    help="ring = 1-rank ring over the peer-to-peer loop, ring-rccl over the RCCL loop, ring-torch over torch.distributed")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--repeat", type=int, default=3)
    a = ap.parse_args()
    nx, ny = (int(v) for v in a.grid.split("x"))
    p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
    obst = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
    flags = a.flags
    kw = {}
    if a.mode != "single":
        flags |= lbm._capi.FLAG_FORCE_HALO
    if a.mode == "ring-torch":
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="file:///tmp/measure_rdv_%d" % os.getpid(), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
        kw = dict(distributed=True, exchange="torch")
    elif a.mode != "single":
        kw = dict(exchange="rccl" if a.mode == "ring-rccl" else "p2p", strict=True)
    sim = lbm.Simulation(p, obst, flags=flags, **kw)
    sim.run(a.warmup)
    best = None
    for _ in range(a.repeat):
        t = time.perf_counter()
        sim.run(a.steps)
        dt = time.perf_counter() - t
        best = dt if best is None else min(best, dt)
    kms, launches = sim.partition.last_run_kernel_ms()
    print(f"{a.grid} mode={a.mode} flags={flags}: {best / a.steps * 1e6:9.2f} us/step wall  "
          f"{kms / a.steps * 1e3:9.2f} us/step device  {nx * ny * a.steps / best / 1e6:10.1f} MLUPS", flush=True)
    sim.close()


if __name__ == "__main__":
    main()
