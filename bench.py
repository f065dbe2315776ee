#!/usr/bin/env python3
"""bench.py — MLUPS of the D2Q9-BGK timestep path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 8192x8192]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no rank environment starts the N rank processes itself (a child
`python -m torch.distributed.run ... bench.py --gpus N ...`, before this process touches the GPU) and relays
the ONE JSON line; under torch.distributed.run it is a rank.  The N > 1 loops are tried in the order
p2p, rccl, torch until one completes AND passes the bit-exact check against a single-GPU run; the line says
which one ran (`config.loop`, `config.rccl_nranks`, `config.macro_k`) and what was tried (`attempts`).

Workload (BASELINE.json config 5, the one the metric's targets are quoted on): the synthetic
8192x8192 deck of SURVEY.md §8(d) — params 8192, 8192, <steps>, 10, 0.1, 0.005, 1.85; walls on the
four edges plus interior cells blocked i.i.d. with p = 0.005 from splitmix64(seed 42); initial
state = the reference's uniform equilibrium.  One "step" = one lattice timestep of the WHOLE grid
(accelerate_flow + fused propagate/rebound/collision/av_velocity, d2q9-bgk.c:345-367).  With N > 1
ranks the SAME grid is row-partitioned (d2q9-bgk.c:834-862) over the GPUs — strong scaling — with
a K-row halo exchange per K steps (direct peer-to-peer stores over xGMI, or RCCL send/recv) and one
reduction of the per-step sums at the end (d2q9-bgk.c:396).  The timed region is the reference's (d2q9-bgk.c:278-398): step loop + av_vels
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="8192x8192", help="NXxNY of the synthetic deck (default: the BASELINE config)")
    ap.add_argument("--reps", type=int, default=0,
                    help="the timed region (EXACTLY --steps steps between two barriers) is repeated this many times and the "
                         "MEDIAN is reported: a single short region carries the first launches' ramp.  Default: 5, or 9 when "
                         "--steps <= 50 (a 7 ms region is still ramping through its first three repetitions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ring", action="store_true",
                    help="N=1 only: run the row-partitioned code path on a 1-rank ring (the rank exchanges with itself)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "rccl", "torch"],
                    help="N>1 (or --ring) halo exchange: direct peer-to-peer stores, native RCCL loop, or torch.distributed P2P "
                         "ops; auto = try them in that order")
    ap.add_argument("--step-allreduce", action="store_true",
                    help="RCCL loop: one all-reduce per (macro-)step instead of one after the loop (north_star wording; measured mode)")
    ap.add_argument("--no-variants", action="store_true", help="N=1: skip the extra timing of the same deck with LBM_FLAG_FAST_AVVELS")
    ap.add_argument("--no-verify", action="store_true", help="N>1: skip the bit-exact check against a single-GPU run of the same deck")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of the 1-core port sample")
    ap.add_argument("--launch-timeout", type=float, default=900.0, help="self-launch: seconds before a set of rank processes is given up")
    ap.add_argument("--dry-launch", action="store_true",
                    help="self-launch test: the rank processes only rendezvous (gloo), report their ranks and exit; no GPU is touched")
    args = ap.parse_args(argv)
    if args.reps <= 0:
        args.reps = 9 if args.steps <= 50 else 5
    return args


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


def emit(obj: dict) -> None:
    os.write(REAL_STDOUT, (json.dumps(obj) + "\n").encode())


def free_port() -> int:
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N rank processes as a CHILD
    `python -m torch.distributed.run` (the reference's `mpirun -np N`, mpi_submit:63) and relay its one JSON
    line.  Nothing here imports torch or touches the GPU; a child that fails, hangs past --launch-timeout or
    fails its bit-exact check is followed by the next exchange mode, and the line records every attempt."""
    import signal
    import subprocess
    # auto: the ranks themselves try p2p, rccl, torch in turn (each attempt checked against a single-GPU run);
    # this level only steps in when a whole set of rank processes crashed or hung, and then pins the mode
    modes = [args.exchange] if args.exchange != "auto" else ["auto", "rccl", "torch"]
    if args.dry_launch:
        modes = modes[:1]
    attempts = []
    clean, skip = [], False                     # argv without any --exchange option
    for a in argv:
        if skip:
            skip = False
        elif a == "--exchange":
            skip = True
        elif not a.startswith("--exchange="):
            clean.append(a)
    for mode in modes:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *clean, "--exchange", mode]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL and hipIpc* across rank processes need it
        t0 = time.time()
        proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, start_new_session=True)
        try:
            out, _ = proc.communicate(timeout=args.launch_timeout)
            rc = proc.returncode
        except subprocess.TimeoutExpired:
            try:
                os.killpg(proc.pid, signal.SIGKILL)                # the process group this call started, nothing else
            except ProcessLookupError:
                pass
            out, _ = proc.communicate()
            rc = -9
        line = None
        for raw in (out or b"").decode(errors="replace").splitlines():
            raw = raw.strip()
            if raw.startswith("{") and raw.endswith("}"):
                try:
                    line = json.loads(raw)
                except ValueError:
                    pass
        attempt = {"exchange": mode, "returncode": rc, "seconds": round(time.time() - t0, 1)}
        good = rc == 0 and isinstance(line, dict) and ("value" in line or line.get("dry_launch"))
        if not good:
            attempt["error"] = "timed out" if rc == -9 else (line or {}).get("error", "no result line")
        attempts.append(attempt)
        if good:
            line["launch_attempts"] = attempts
            emit(line)
            return 0
        sys.stderr.write(f"bench.py: {args.gpus}-rank run with exchange={mode} failed ({attempt['error']}); trying the next mode\n")
    sys.stderr.write(f"bench.py: no exchange mode completed: {json.dumps(attempts)}\n")
    return 1


def dry_rank() -> None:
    """--dry-launch: rendezvous over gloo, rank 0 reports who showed up.  No GPU, no library."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, {"rank": int(os.environ["RANK"]), "local_rank": int(os.environ["LOCAL_RANK"]),
                                  "world": int(os.environ["WORLD_SIZE"]), "pid": os.getpid()})
    if dist.get_rank() == 0:
        emit({"dry_launch": True, "n_gpus": dist.get_world_size(), "ranks": seen})
    dist.barrier()
    dist.destroy_process_group()


def load_roofline(kernel: str, workload: str) -> dict | None:
    """Per-launch HBM bytes and VALU cycles of the dominant kernel from the PMC passes of this commit
    (profiles/r02/roofline.json, written by scripts/make_roofline.py from the rocprofv3 CSVs beside it)."""
    path = os.path.join(ROOT, "profiles", "r02", "roofline.json")
    if not os.path.exists(path):
        return None
    try:
        r = json.load(open(path))
    except ValueError:
        return None
    return r if r.get("kernel") == kernel and r.get("workload") == workload else None


def main() -> int:
    argv = sys.argv[1:]
    args = parse_args(argv)
    quiet_stdout()
    in_rank = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not in_rank and (args.gpus > 1 or args.dry_launch):
        return self_launch(args, argv)
    if args.dry_launch:
        dry_rank()
        return 0
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "LBM_FORCE_DEVICE" in os.environ:          # testing aid: several ranks on one device
        local_rank = int(os.environ["LBM_FORCE_DEVICE"])
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} does not match --gpus {args.gpus}")

    import torch
    import mpilattice_boltzmann_amd as lbm
    torch.cuda.set_device(local_rank)
    dist = None
    backend = os.environ.get("LBM_DIST_BACKEND", "nccl")   # "gloo": ranks may share a GPU (p2p exchange only; testing aid)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    if rank == 0:
        lbm.build()                                       # no-op when lib/ is current; other ranks wait below
    if dist is not None:
        dist.barrier()
    lbm.load_library()
    nx, ny = (int(v) for v in args.workload.lower().split("x"))
    params = lbm.Params(nx, ny, args.steps, 10, 0.1, 0.005, 1.85)
    obstacles = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
    flags = lbm._capi.FLAG_FORCE_HALO if args.ring else 0
    partitioned = world > 1 or args.ring

    def fail(message: str) -> int:
        """A rank-symmetric failure: one JSON line with the reason (the self-launcher reads it), exit 1."""
        if rank == 0:
            emit({"error": message, "n_gpus": world, "exchange": args.exchange})
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return 1

    def all_ok(flag: bool) -> bool:
        if dist is None:
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def check_against_single_gpu(sim, av_last, steps_done):
        """The partitioned run against ONE GPU doing the whole grid, bit for bit: 64-bit digest of this rank's
        rows (lbm_state_checksum) and the last run's av_vels — the multi-GPU parity test, run where the GPUs are."""
        whole = lbm.Simulation(params, obstacles, device=local_rank)
        av_ref = whole.run(steps_done)[-len(av_last):] if len(av_last) else np.zeros(0, np.float32)
        y0, y1 = sim.partition.y0, sim.partition.y0 + sim.partition.ny_local
        same = sim.partition.checksum() == whole.partition.checksum(y0, y1)
        whole.close()
        av_err = float(np.max(np.abs(av_last.astype(np.float64) - av_ref.astype(np.float64)) / av_ref.astype(np.float64))) if len(av_last) else 0.0
        return bool(same and av_err < 1e-6), same, av_err

    # Which loop runs.  strict: what is asked for runs or the attempt fails — never a silent fall-back inside
    # Simulation, which would be reported as the faster loop at a fraction of its speed.  With --exchange auto
    # the loops are tried in turn HERE, visibly: each must set up on every rank, complete the warm-up and
    # reproduce a single-GPU run of the same deck bit for bit before it is timed.
    verify_on = partitioned and not args.no_verify
    if not partitioned:
        modes = ["auto"]
    elif args.exchange != "auto":
        modes = [args.exchange]
    else:
        modes = ["rccl"] if args.step_allreduce else ["p2p", "rccl", "torch"]
    attempts, sim = [], None
    for mode in modes:
        note, ok = None, False
        try:
            sim = lbm.Simulation(params, obstacles, device=local_rank, flags=flags, distributed=world > 1, exchange=mode,
                                 strict=True, step_allreduce=args.step_allreduce)
            ok = True
        except lbm.LbmError as e:                         # raised on every rank together
            sim, note = None, f"set-up: {e}"
        if ok:
            try:
                av_w = sim.run(args.warmup)               # untimed
                if verify_on:
                    good, same, av_err = check_against_single_gpu(sim, av_w, args.warmup)
                    if not good:
                        ok, note = False, f"parity after the warm-up: state digest equal {same}, av_vels rel err {av_err:.2e}"
            except lbm.LbmError as e:
                ok, note = False, f"warm-up: {e}"
        ok = all_ok(ok)
        attempts.append({"exchange": mode, "ok": ok, **({"error": note or "failed on another rank"} if not ok else {})})
        if ok:
            break
        if sim is not None:
            sim.close()
            sim = None
        if rank == 0:
            sys.stderr.write(f"bench.py: exchange={mode} not usable ({attempts[-1]['error']}); trying the next one\n")
    if sim is None:
        return fail("no exchange mode completed: " + json.dumps(attempts))

    try:
        times, av = [], None
        for _ in range(max(1, args.reps)):
            sync_all()                                    # barrier + device synchronise: every rank starts together
            t0 = time.perf_counter()
            av = sim.run(args.steps)                      # EXACTLY K steps (+ the av_vels reduction, as the reference times it)
            torch.cuda.synchronize()                      # this rank's device work is complete (the reduction made it wait for
            times.append(time.perf_counter() - t0)        # every rank's sums); the MAX over ranks below is the job's time.  The
            # barrier that closes the bracket is the next repetition's sync_all / the all-reduce of the times: an NCCL barrier
            # inside the region would add its own ~0.1 ms to a 1 ms region of steps.
            kernel_ms, launches = sim.partition.last_run_kernel_ms()
    except lbm.LbmError as e:
        return fail(f"run: {e}")
    if dist is not None:                                  # per repetition: the slowest rank's time
        t = torch.tensor(times, dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        times = [float(v) for v in t.cpu()]
    elapsed = float(np.median(times))
    assert av.shape == (args.steps,) and np.all(np.isfinite(av)) and np.all(av > 0)

    desc = sim.partition.describe()
    what = sim.describe()
    steps_done = args.warmup + args.steps * max(1, args.reps)

    # N = 1: the same deck with LBM_FLAG_FAST_AVVELS (float sum|u| terms; default off) — both figures side by side
    variants = None
    if world == 1 and not args.ring and not args.no_variants:
        alt = lbm.Simulation(params, obstacles, device=local_rank, flags=flags | lbm._capi.FLAG_FAST_AVVELS)
        if "fast av_vels" in alt.partition.describe()["kernel"]:
            alt.run(args.warmup)
            talt, av_alt = [], None
            for _ in range(max(1, args.reps)):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                av_alt = alt.run(args.steps)
                torch.cuda.synchronize()
                talt.append(time.perf_counter() - t0)
            med = float(np.median(talt))
            variants = {"fast_av_vels": {"value": nx * ny * args.steps / med / 1e6, "unit": "MLUPS", "ms_per_step": med / args.steps * 1e3,
                                         "av_vels_max_rel_diff_to_default": float(np.max(np.abs(av_alt.astype(np.float64) - av) / av)),
                                         "note": "LBM_FLAG_FAST_AVVELS: each cell's sum|u| term in float instead of double; populations "
                                                 "identical bit for bit; NOT the headline (default off).  Measured in a second context of this process: where a "
                                                 "context's grids land moves its step time by 3-5 % either way (DESIGN.md 4.2), so compare with the "
                                                 "same-process A/B in profiles/r02/ab_fast_avvels_8192.txt rather than with `value`"}}
        alt.close()

    verify = None
    if verify_on:
        good, same, av_err = check_against_single_gpu(sim, av, steps_done)
        good = all_ok(good)
        verify = {"ok": good, "what": f"every rank's rows bit-identical (64-bit state digest) to a single-GPU run of the whole grid on the same "
                                      f"device, after the warm-up ({args.warmup} steps) and after all {steps_done} steps; av_vels of the last "
                                      f"repetition within 1e-6", "av_vels_max_rel": av_err}
        if not good:
            sim.close()
            return fail(f"parity: partitioned run ({what['loop']} loop) differs from the single-GPU run after {steps_done} steps "
                        f"(state digest equal: {same}, av_vels rel err {av_err:.2e})")
    sim.close()

    if rank == 0:
        cells = nx * ny
        mlups = cells * args.steps / elapsed / 1e6
        # dominant kernel: the fused step kernel; average launch duration from HIP events on its stream (last repetition).
        # A launch of lbm_multi_kernel<K> advances its cells by K steps, so the algorithmic bytes of a
        # launch are 108 B x cells x steps-per-launch (the convention counts traffic per cell-STEP).
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        two_launches = partitioned and (what["p2p"] is None or "edge stream" in what["p2p"])
        if not two_launches:
            cells_per_launch = desc["cells_per_launch"]
            steps_per_launch = args.steps / max(launches, 1)
        else:                                             # interior + edge launch per (macro-)step on this rank
            cells_per_launch = desc["cells_per_launch"] / 2.0
            steps_per_launch = args.steps / max(launches / 2.0, 1)
        algo_gbs = ALGO_BYTES_PER_CELL * cells_per_launch * steps_per_launch / avg_launch_s / 1e9
        # The binding limits, from PMC passes of this commit (profiles/r02/): physical HBM bytes and VALU busy
        # cycles per launch, each against this run's launch time.  `frac` is the larger of the two: a fraction
        # of something the chip can actually deliver.  The 108-B convention figure (which assumes one pass over
        # HBM per step, while this kernel makes one per K steps) is reported beside it, not as `frac`.
        pmc = load_roofline(desc["kernel"], f"{nx}x{ny}") if (world == 1 and not args.ring) else None
        # partitioned runs of the same kernel: the per-launch counters scale with the cells a launch advances (the PMC
        # passes profile one process on the whole 8192x8192 grid; a rank's launches run the same code on fewer tiles)
        scaled = None
        if pmc is None and partitioned:
            whole = load_roofline(desc["kernel"], "8192x8192")
            if whole is not None:
                scale = cells_per_launch * steps_per_launch / (8192.0 * 8192.0 * whole["steps_per_launch"])
                scaled = dict(whole, hbm_bytes_per_launch=whole["hbm_bytes_per_launch"] * scale)
                pmc = scaled
        roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "kernel": desc["kernel"], "avg_launch_ms": avg_launch_s * 1e3, "launches": launches, "steps_per_launch": steps_per_launch,
                "convention_108B": {"GBps": algo_gbs, "ratio_to_peak": algo_gbs / HBM_PEAK_GBS, "bytes_per_cell_step": ALGO_BYTES_PER_CELL,
                                    "note": "north_star's accounting (18 reads + 9 writes per cell per step); exceeds 1 because "
                                            "lbm_multi_kernel makes one pass over HBM per K steps"}}
        if pmc is not None:
            hbm_gbs = pmc["hbm_bytes_per_launch"] / avg_launch_s / 1e9
            frac_hbm = hbm_gbs / HBM_PEAK_GBS
            # VALU: busy cycles over available cycles, both counted in ONE profiled pass (4 x SQ_ACTIVE_INST_VALU over
            # GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): a ratio of cycles, so it does not depend on the clock the chip
            # holds (profiled passes run ~8 % slower than this un-profiled run, at a lower clock)
            frac_valu = pmc["frac_valu_profiled"]
            simds = pmc.get("simds", 1024)
            limits = {"hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": frac_hbm},
                      "valu": {"achieved": frac_valu * simds, "peak": float(simds), "unit": "busy SIMDs", "frac": frac_valu}}
            bound = "hbm" if frac_hbm >= frac_valu else "valu"
            roof.update(limits[bound])
            roof.update({"bound": bound, "traffic": pmc["hbm_bytes_per_launch"], "frac_hbm_physical": frac_hbm, "frac_valu": frac_valu,
                         "limits": limits, "lds_bank_conflict_frac": pmc.get("lds_bank_conflict_frac"),
                         "scaled_from_single_gpu_pmc": scaled is not None,
                         "pmc_source": "profiles/r02/roofline.json (scripts/make_roofline.py over the rocprofv3 --pmc CSVs beside it)",
                         "note": "two limits, each a fraction of something the chip can deliver; achieved/peak/frac are those of the larger "
                                 "(`bound`).  hbm: physical bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes) over THIS "
                                 "run's launch time, against 8 TB/s.  valu: 4 x SQ_ACTIVE_INST_VALU busy cycles per launch over the cycles of the "
                                 "same profiled pass (GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs — a cycle ratio, independent of the clock held"})
        else:
            roof.update({"achieved": algo_gbs, "frac": None,
                         "note": "no PMC summary for this kernel/workload in profiles/r02/roofline.json: only the 108-B convention figure"})
        exchange_txt = {"p2p": "direct peer-to-peer stores into the neighbours' ghost rows (xGMI), flags + one-wave wait kernels, "
                               "all-gather + local sum after the loop",
                        "rccl": "RCCL send/recv on a side stream, " + ("one all-reduce per macro-step" if what["step_allreduce"] else "one all-reduce after the loop"),
                        "torch": "torch.distributed P2P ops, one all-reduce after the loop", "single": ""}[what["loop"]]
        k = max(what["macro_k"], 1)
        if world == 1:
            part_txt = "single GPU" if not args.ring else f"1-rank ring (self exchange), {k}-row halo exchange per {k} steps: {exchange_txt}"
        else:
            part_txt = f"{world} row blocks (d2q9-bgk.c:834-862), one {k}-row halo exchange per {k} steps: {exchange_txt}"
        out = {
            "metric": "MLUPS", "value": mlups, "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {nx}x{ny} D2Q9-BGK deck (walls + p=0.005 random obstacles, splitmix64 seed 42), "
                                   f"density 0.1 accel 0.005 omega 1.85", "nx": nx, "ny": ny, "partitioning": part_txt,
                       "loop": what["loop"], "macro_k": what["macro_k"], "rccl_nranks": what["rccl_nranks"], "p2p": what["p2p"],
                       "step_allreduce": what["step_allreduce"], "kernel": desc["kernel"]},
            "timing": {"reps": len(times), "statistic": "median over reps of (max over ranks of the time of EXACTLY `steps` steps): every rank starts behind a barrier + "
                                    "device synchronise and stops its clock when its own device work, which ends with the global reduction, is complete",
                       "ms_per_rep": [t * 1e3 for t in times]},
            "pct_hbm_roofline": 100.0 * mlups / world / (HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_CELL / 1e6),
            "roofline": roof,
        }
        if variants is not None:
            out["variants"] = variants
        if verify is not None:
            out["parity_check"] = verify
        if partitioned:
            out["exchange_attempts"] = attempts
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lbm, params, obstacles, args.cpu_seconds)
        emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
