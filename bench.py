#!/usr/bin/env python3
"""bench.py — MLUPS of the D2Q9-BGK timestep path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 8192x8192]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json config 5, the one the metric's targets are quoted on): the synthetic
8192x8192 deck of SURVEY.md §8(d) — params 8192, 8192, <steps>, 10, 0.1, 0.005, 1.85; walls on the
four edges plus interior cells blocked i.i.d. with p = 0.005 from splitmix64(seed 42); initial
state = the reference's uniform equilibrium.  One "step" = one lattice timestep of the WHOLE grid
(accelerate_flow + fused propagate/rebound/collision/av_velocity, d2q9-bgk.c:345-367).  With N > 1
ranks the SAME grid is row-partitioned (d2q9-bgk.c:834-862) over the GPUs — strong scaling — with
a one-row halo exchange per step over RCCL and one all-reduce of the per-step sums at the end
(d2q9-bgk.c:396).  The timed region is the reference's (d2q9-bgk.c:278-398): step loop + av_vels
reduction, inputs resident in HBM, no file I/O.

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL = 108.0          # 18 reads + 9 writes of fp32 (BASELINE.json north_star)
PHYS_BYTES_PER_CELL = 72.125         # 9 reads + 9 writes + 1 mask bit actually moved by the pull kernel
HBM_PEAK_GBS = 8000.0                # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="8192x8192", help="NXxNY of the synthetic deck (default: the BASELINE config)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ring", action="store_true",
                    help="N=1 only: run the row-partitioned code path on a 1-rank ring (the rank exchanges with itself over RCCL)")
    ap.add_argument("--exchange", default="rccl", choices=["rccl", "torch"],
                    help="N>1 halo exchange: native RCCL loop (liblbm_d2q9_rccl.so) or torch.distributed P2P ops")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of the 1-core port sample")
    return ap.parse_args()


def host_cores() -> int:
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU
    box gives one GPU's share of the host, not all the cores /proc/cpuinfo lists)."""
    n = len(os.sched_getaffinity(0))
    quota = None
    try:                                               # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = int(q) / int(period)
    except Exception:
        try:                                           # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except Exception:
            pass
    if quota is not None:
        return max(1, min(n, int(quota)))
    return 16 if n > 64 else n                         # no visible quota on a many-core host: one GPU's share is 16 cores


def reference_binary_baseline(lbm, nx: int, ny: int) -> dict | None:
    """The UNMODIFIED reference (oracle/_ref/d2q9-bgk_ref, built from /root/reference by `make -C oracle
    ref` in the build container) timed on this box: one MPI rank = its serial loop.  It can only run a
    whole deck and always writes final_state.dat through fprintf, so the sample is the same synthetic
    recipe at 1/16 of the cells (2048x2048 for the 8192x8192 workload), 40 steps; the figure is the
    reference's own "Elapsed time" line (loop only, d2q9-bgk.c:278-398)."""
    import shutil
    import subprocess
    import tempfile
    ref = os.path.join(ROOT, "oracle", "_ref", "d2q9-bgk_ref")
    if not os.path.exists(ref):
        return None
    snx, sny, steps = max(64, nx // 4), max(64, ny // 4), 40
    tmp = tempfile.mkdtemp(prefix="lbm_ref_")
    try:
        pp, op = lbm.write_synthetic_deck(tmp, "sample", lbm.Params(snx, sny, steps, 10, 0.1, 0.005, 1.85), 0.005, 42, True)
        r = subprocess.run([ref, pp, op], cwd=tmp, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            return None
        elapsed = float([l for l in r.stdout.splitlines() if l.startswith("Elapsed time")][0].split()[2])
        return {"value": snx * sny * steps / elapsed / 1e6, "unit": "MLUPS", "cores": 1, "kind": "reference",
                "sample": f"unmodified d2q9-bgk.c (gcc -std=c99 -O3, MPICH, 1 rank) on the same synthetic recipe at "
                          f"{snx}x{sny}, {steps} steps, its own 'Elapsed time' = {elapsed:.3f} s"}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(lbm, params, obstacles, target_s: float) -> dict:
    """CPU figures measured on this box's host cores beside the GPU number:
      * the reference binary itself on a bounded sample (kind "reference") when oracle/_ref is present;
      * the oracle (CPU restatement of d2q9-bgk.c's path, digest-pinned to that binary) on a bounded
        sample of the SAME deck: a few steps on one core (the reference's serial loop), then the
        row-parallel form on all usable cores.  It is the headline (kind "port") when the binary is absent."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    cells = params.nx * params.ny
    steps1 = max(2, int(target_s * 85e6 / cells))            # ~85-100 MLUPS per core expected
    t = time.perf_counter()
    oracle_lib.run_fast(params, obstacles, steps1, 1)
    dt1 = time.perf_counter() - t
    ncores = host_cores()
    stepsn = max(4, int(steps1 * ncores * 0.4))
    t = time.perf_counter()
    oracle_lib.run_fast(params, obstacles, stepsn, ncores)
    dtn = time.perf_counter() - t
    model = ""
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        pass
    port = {"value": cells * steps1 / dt1 / 1e6, "unit": "MLUPS", "cores": 1, "kind": "port",
            "sample": f"{steps1} steps of the same {params.nx}x{params.ny} deck (init + loop, {dt1:.1f} s), gcc -std=c99 -O3"}
    allc = {"value": cells * stepsn / dtn / 1e6, "unit": "MLUPS", "cores": ncores, "kind": "port",
            "sample": f"{stepsn} steps of the same deck, row-parallel OpenMP, {dtn:.1f} s"}
    out = reference_binary_baseline(lbm, params.nx, params.ny) or dict(port)
    out["port_1core"] = port
    out["port_all_cores"] = allc
    out["cpu_model"] = model
    return out


# The contract is ONE JSON line on stdout.  Native libraries write there too (RCCL prints a version
# banner at communicator creation on some boxes), so everything else goes to stderr: fd 1 is pointed at
# fd 2 for the whole run and the line is written to the saved descriptor.
REAL_STDOUT = 1


def quiet_stdout() -> None:
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def main() -> None:
    args = parse_args()
    quiet_stdout()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "LBM_FORCE_DEVICE" in os.environ:          # testing aid: several ranks on one device (if the communicator allows it)
        local_rank = int(os.environ["LBM_FORCE_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    import mpilattice_boltzmann_amd as lbm
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if rank == 0:
        lbm.build()                                       # no-op when lib/ is current; other ranks wait below
    if dist is not None:
        dist.barrier()
    lbm.load_library()
    nx, ny = (int(v) for v in args.workload.lower().split("x"))
    params = lbm.Params(nx, ny, args.steps, 10, 0.1, 0.005, 1.85)
    obstacles = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
    flags = lbm._capi.FLAG_FORCE_HALO if args.ring else 0
    sim = lbm.Simulation(params, obstacles, device=local_rank, flags=flags, distributed=world > 1, exchange=args.exchange)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sim.run(args.warmup)                                  # untimed
    sync_all()
    t0 = time.perf_counter()
    av = sim.run(args.steps)                              # EXACTLY K steps (+ the av_vels reduction, as the reference times it)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert av.shape == (args.steps,) and np.all(np.isfinite(av)) and np.all(av > 0)

    kernel_ms, launches = sim.partition.last_run_kernel_ms()
    desc = sim.partition.describe()
    macro_k = sim.partition.macro_steps
    sim.close()

    if rank == 0:
        cells = nx * ny
        mlups = cells * args.steps / elapsed / 1e6
        # dominant kernel: the fused step kernel; average launch duration from HIP events on its stream.
        # A launch of lbm_multi_kernel<K> advances its cells by K steps, so the algorithmic bytes of a
        # launch are 108 B x cells x steps-per-launch (the convention counts traffic per cell-STEP).
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        if world == 1 and not args.ring:
            cells_per_launch = desc["cells_per_launch"]
            steps_per_launch = args.steps / max(launches, 1)
        else:                                             # interior + edge launch per (macro-)step on this rank
            cells_per_launch = desc["cells_per_launch"] / 2.0
            steps_per_launch = args.steps / max(launches / 2.0, 1)
        achieved = ALGO_BYTES_PER_CELL * cells_per_launch * steps_per_launch / avg_launch_s / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            # PMC passes are separate runs (rocprofv3 --pmc), so the figure is read from the committed
            # summary; it is only quoted when it was measured for this workload AND this kernel
            if t.get("workload") == f"{nx}x{ny}" and t.get("kernel") == desc["kernel"] and world == 1 and not args.ring:
                traffic = t.get("hbm_bytes_per_launch")
        out = {
            "metric": "MLUPS", "value": mlups, "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {nx}x{ny} D2Q9-BGK deck (walls + p=0.005 random obstacles, splitmix64 seed 42), "
                                   f"density 0.1 accel 0.005 omega 1.85", "nx": nx, "ny": ny,
                       "partitioning": ("single GPU" if not args.ring else "1-rank ring (self exchange over RCCL)") if world == 1 else f"{world} row blocks, one {max(macro_k, 1)}-row halo exchange per {max(macro_k, 1)} steps (RCCL send/recv, {args.exchange} loop), one all-reduce after the loop"},
            "pct_hbm_roofline": 100.0 * mlups / world / (HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_CELL / 1e6),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_GBps": None if traffic is None else traffic / avg_launch_s / 1e9,      # physical HBM rate
                         "traffic_frac": None if traffic is None else traffic / avg_launch_s / 1e9 / HBM_PEAK_GBS,
                         "kernel": desc["kernel"],
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": launches, "steps_per_launch": steps_per_launch,
                         "algorithmic_bytes_per_cell_step": ALGO_BYTES_PER_CELL,
                         "note": "frac > 1 is possible: the 108 B/cell-step convention assumes one pass over HBM per step; "
                                 "lbm_multi_kernel advances K steps per pass (traffic = measured HBM bytes per launch)"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lbm, params, obstacles, args.cpu_seconds)
        os.write(REAL_STDOUT, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
