#!/usr/bin/env python3
"""bench.py — MLUPS of the D2Q9-BGK timestep path on MI355X, with roofline and CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 8192x8192]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no rank environment starts the N rank processes itself (a child
`python -m torch.distributed.run ... bench.py --gpus N ...`, before this process touches the GPU) and relays
the ONE JSON line; under torch.distributed.run it is a rank.

Workload (BASELINE.json config 5, the one the metric's targets are quoted on): the synthetic
8192x8192 deck of SURVEY.md §8(d) — params 8192, 8192, <steps>, 10, 0.1, 0.005, 1.85; walls on the
four edges plus interior cells blocked i.i.d. with p = 0.005 from splitmix64(seed 42); initial
state = the reference's uniform equilibrium.  One "step" = one lattice timestep of the WHOLE grid
(accelerate_flow + fused propagate/rebound/collision/av_velocity, d2q9-bgk.c:345-367).  With N > 1
ranks the SAME grid is row-partitioned (d2q9-bgk.c:834-862) over the GPUs — strong scaling — with
a k-row halo exchange per k steps (direct peer-to-peer stores over xGMI, or RCCL send/recv) and one
reduction of the per-step sums at the end (d2q9-bgk.c:396).  The timed region is the reference's
(d2q9-bgk.c:278-398): step loop + av_vels reduction, inputs resident in HBM, no file I/O.

What one N > 1 invocation harvests (each part checked bit for bit against a single-GPU run of the same deck
on the rank's own device before it is timed, and again after):
  value / ms_per_step   the headline: loops tried in the order p2p, rccl, torch until one sets up on every
                        rank, completes the warm-up and passes the check (`exchange_attempts` says what was tried)
  phases                where a p2p run's time goes (HIP events per rank: set-up, steps, reduction, push kernels)
  variants.rccl, variants.rccl_step_allreduce
                        the same deck over the RCCL loop, and with north_star's one all-reduce per (macro-)step
  secondary.input_1024x1024
                        BASELINE.json config 4: the shipped 1024x1024 deck, all 20 000 steps, on the same N ranks
                        (p2p and rccl), Reynolds line and av_vels compared with the reference binary's
At N = 1 the line also carries `sustained` (the headline deck for 3 x >= 2000 steps back to back: what the socket power cap
leaves) and `secondary.shipped_decks` (BASELINE.json configs 2 - 3: the reference's four input decks through lbm_run, each
with seconds, us per step, MLUPS, final_state.dat digest == the reference binary's, av_vels under check.py's rule).
Everything after the headline runs under a time budget (--budget-s): a part that does not fit is recorded as
skipped, and a watchdog prints the line as far as it got if anything hangs — the driver's run is killed at
600 s, so the line must be out well before.  The control plane (handles, flags, barriers) is a gloo group:
its collectives time out with an exception instead of aborting the process, and the peer-to-peer loop then
does not depend on RCCL at all.

Prints ONE JSON line on rank 0 (see DESIGN.md §Measurement for every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CELL = 108.0          # 18 reads + 9 writes of fp32 (BASELINE.json north_star)
PHYS_BYTES_PER_CELL = 72.125         # 9 reads + 9 writes + 1 mask bit actually moved by a one-step pull kernel
HBM_PEAK_GBS = 8000.0                # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
PROFILE_ROUND = "r04"                # profiles/<round>/roofline.json: the PMC passes the roofline object rests on
DECKS = os.path.join(ROOT, "tests", "golden", "decks")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="8192x8192", help="NXxNY of the synthetic deck (default: the BASELINE config)")
    ap.add_argument("--reps", type=int, default=0,
                    help="the timed region (EXACTLY --steps steps between two barriers) is repeated this many times and the "
                         "MEDIAN is reported: a single short region carries the first launches' ramp.  Default: 5, or 21 when "
                         "--steps <= 50: regions of 1-7 ms separated by host gaps keep speeding up for ten or more repetitions "
                         "(clocks, caches: 1.44, 1.14, 1.17, 1.17, 1.16, 1.16, 1.13, 1.12, 1.10 ms for nine 20-step runs of an 8-GPU "
                         "rank's share, 1.06 ms after sixty), and the median of nine sat in the middle of that ramp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ring", action="store_true",
                    help="N=1 only: run the row-partitioned code path on a 1-rank ring (the rank exchanges with itself)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "p2p", "rccl", "torch"],
                    help="N>1 (or --ring) halo exchange: direct peer-to-peer stores, native RCCL loop, or torch.distributed P2P "
                         "ops; auto = try them in that order")
    ap.add_argument("--step-allreduce", action="store_true",
                    help="RCCL loop: one all-reduce per (macro-)step instead of one after the loop (north_star wording; measured mode)")
    ap.add_argument("--no-variants", action="store_true",
                    help="skip the extra timings: N=1: the other two forms of the sum|u| terms (LBM_FLAG_FAST_AVVELS, LBM_FLAG_EXACT_AVVELS); partitioned runs: the RCCL loop and its per-step all-reduce mode")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures — N=1: the four shipped decks through lbm_run with their acceptance checks (BASELINE.json configs 2-3) "
                         "and the sustained 3 x 2000-step line; partitioned runs: the shipped 1024x1024 deck on the same ranks (config 4)")
    ap.add_argument("--rank-grid", default="",
                    help="PXxPY: headline over the tile (2-D) decomposition on PX x PY = N ranks instead of the reference's row blocks (peer-to-peer "
                         "loop only; with --ring: 1x1, a rank that is its own neighbour in every direction); auto: lbm_choose_rank_grid decides "
                         "(row blocks for the square decks, column blocks for grids much wider than tall)")
    ap.add_argument("--column-block", action="store_true",
                    help="with --ring --rank-grid 1x1: the ring stands for a block of a PX x 1 tiling (a column block: no ghost rows, its rows wrap inside "
                         "the launch, an exchange is the column push alone) instead of a block of any tiling (ghost rows, row push onto itself)")
    ap.add_argument("--no-power", action="store_true", help="do not sample the card's socket power / shader clock (hwmon files) during the headline")
    ap.add_argument("--no-phases", action="store_true", help="skip the profiled extra repetition behind `phases` / the per-launch roofline timing")
    ap.add_argument("--secondary-steps", type=int, default=0, help="steps of the 1024x1024 deck (default: its own 20 000)")
    ap.add_argument("--no-verify", action="store_true", help="N>1: skip the bit-exact check against a single-GPU run of the same deck")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target CPU time of the 1-core port sample")
    ap.add_argument("--budget-s", type=float, default=0.0,
                    help="seconds this invocation may take before the watchdog prints the line as far as it got (default 420; the "
                         "driver kills a run at 600).  Optional parts that do not fit are recorded as skipped")
    ap.add_argument("--launch-timeout", type=float, default=150.0,
                    help="self-launch: seconds before a set of rank processes is given up (three modes at most: 450 s in the worst case)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="self-launch test: the rank processes only rendezvous (gloo), report their ranks and exit; no GPU is touched")
    args = ap.parse_args(argv)
    if args.reps <= 0:
        args.reps = 21 if args.steps <= 50 else 5
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
# fd 2 for the whole run and the line is written to the saved descriptor — once: by the main thread when
# it is done, or by the watchdog with what had been banked when the budget ran out.
REAL_STDOUT = 1
_EMIT_LOCK = threading.Lock()
_EMITTED = False
_BANK: dict | None = None            # rank 0: the line as far as it is complete (the headline at least)
_STAGE = "start"                     # what the rank is doing (the watchdog names it)


def quiet_stdout() -> None:
    global REAL_STDOUT
    sys.stdout.flush()
    REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def emit(obj: dict) -> bool:
    """Write the line unless one has been written already.  True if this call wrote it."""
    global _EMITTED
    with _EMIT_LOCK:
        if _EMITTED:
            return False
        os.write(REAL_STDOUT, (json.dumps(obj) + "\n").encode())
        _EMITTED = True
        return True


def bank(obj: dict) -> None:
    global _BANK
    with _EMIT_LOCK:
        _BANK = json.loads(json.dumps(obj))          # a copy: later stages keep filling the original


def stage(name: str) -> None:
    global _STAGE
    _STAGE = name


def start_watchdog(rank: int, world: int, budget_s: float, t_start: float) -> None:
    """A rank that is still running when the budget ends stops itself: rank 0 prints what it has banked (the
    headline, if that was reached, with a note of what was cut short) or an error line, every rank leaves with
    os._exit — the only exit that works from under a hung collective or a kernel that never returns."""
    def watch():
        while time.time() - t_start < budget_s:
            time.sleep(0.5)
        if _EMITTED:
            return
        code = 0
        if rank == 0:
            banked = _BANK
            if banked is not None:
                banked["truncated"] = f"budget of {budget_s:.0f} s ran out during: {_STAGE}"
                emit(banked)
            else:
                emit({"error": f"budget of {budget_s:.0f} s ran out during: {_STAGE}", "n_gpus": world})
                code = 1
        else:
            time.sleep(1.0)                          # let rank 0 write first: its exit ends the launcher's wait
        sys.stderr.write(f"bench.py: rank {rank}: budget of {budget_s:.0f} s ran out during: {_STAGE}\n")
        sys.stderr.flush()
        os._exit(code)
    threading.Thread(target=watch, daemon=True).start()


def free_port() -> int:
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_modes(args) -> list[str]:
    """Exchange modes the self-launcher tries, one set of rank processes each.  auto: the ranks themselves try
    p2p, rccl, torch in turn; this level only steps in when a whole set of rank processes crashed or hung, and then
    pins the mode.  The worst case is len(modes) x --launch-timeout seconds (tests pin: < 500 s with the defaults)."""
    if args.dry_launch:
        return [args.exchange]
    return [args.exchange] if args.exchange != "auto" else ["auto", "rccl", "torch"]


def self_launch(args, argv) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N rank processes as a CHILD
    `python -m torch.distributed.run` (the reference's `mpirun -np N`, mpi_submit:63) and relay its one JSON
    line.  Nothing here imports torch or touches the GPU; a child that fails, hangs past --launch-timeout or
    fails its bit-exact check is followed by the next exchange mode, and the line records every attempt."""
    import signal
    import subprocess
    attempts = []
    clean, skip = [], False                     # argv without any --exchange / --budget-s option
    for a in argv:
        if skip:
            skip = False
        elif a in ("--exchange", "--budget-s"):
            skip = True
        elif not a.startswith("--exchange=") and not a.startswith("--budget-s="):
            clean.append(a)
    # the ranks' own budget ends before this level gives up on them, so that a slow run still returns its headline
    budget = args.budget_s if args.budget_s > 0 else max(30.0, args.launch_timeout - 25.0)
    for mode in launch_modes(args):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__), *clean, "--exchange", mode,
               "--budget-s", str(min(budget, max(30.0, args.launch_timeout - 25.0)))]
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
        # a line with a value counts even when the launcher had to be killed afterwards (a rank stuck in tear-down)
        good = isinstance(line, dict) and ("value" in line or (rc == 0 and line.get("dry_launch")))
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


def load_roofline() -> dict | None:
    """Per-launch HBM bytes and VALU busy cycles of the step kernels from the PMC passes of this round
    (profiles/<round>/roofline.json, written by scripts/make_roofline.py from the rocprofv3 CSVs beside it)."""
    path = os.path.join(ROOT, "profiles", PROFILE_ROUND, "roofline.json")
    try:
        return json.load(open(path))
    except (OSError, ValueError):
        return None


class PowerSampler:
    """Socket power and shader clock of ONE card while a region runs, from the card's hwmon files (readable by an ordinary
    user; no GPU call).  Why it is in the line: lbm_multi_kernel<4> runs AT the socket power limit (1380 of 1400 W, shader
    clock ~2.2 of 2.4 GHz — scripts/power_trace.py), so neither the HBM nor the VALU fraction is the
    binding limit: energy per cell-step is.  `pci` = "dddd:bb:dd.f" of the card (None: the card whose clock moved most)."""

    FILES = {"power_uW": ("power1_input", "power1_average"), "cap_uW": ("power1_cap",), "sclk_Hz": ("freq1_input",)}

    def __init__(self, pci: str | None, root: str = "/sys/class/drm", period_s: float = 0.002):
        import glob
        self.period_s, self.samples, self._stop, self._thread = period_s, {}, threading.Event(), None
        self.cards = {}
        seen = set()
        for dev in sorted(glob.glob(os.path.join(root, "card*", "device"))):
            real = os.path.realpath(dev)
            hw = sorted(glob.glob(os.path.join(dev, "hwmon", "hwmon*")))
            if real in seen or not hw or (pci is not None and os.path.basename(real).lower() != pci.lower()):
                continue
            seen.add(real)
            files = {k: next((os.path.join(hw[0], n) for n in names if os.path.exists(os.path.join(hw[0], n))), None)
                     for k, names in self.FILES.items()}
            if files["power_uW"] or files["sclk_Hz"]:
                self.cards[os.path.basename(real)] = files
        self.pci = pci

    @staticmethod
    def _read(path):
        try:
            with open(path) as f:
                return float(f.read().strip())
        except (OSError, ValueError, TypeError):
            return None

    def _poll(self):
        while not self._stop.is_set():
            for name, files in self.cards.items():
                self.samples.setdefault(name, []).append({k: self._read(p) for k, p in files.items()})
            self._stop.wait(self.period_s)

    def start(self):
        if self.cards:
            self._thread = threading.Thread(target=self._poll, daemon=True)
            self._thread.start()
        return self

    def stop(self) -> dict | None:
        """Statistics of the samples taken while the shader clock was at least half its maximum over the region."""
        if self._thread is None:
            return None
        self._stop.set()
        self._thread.join()

        def swing(rows):
            v = [r["sclk_Hz"] for r in rows if r.get("sclk_Hz") is not None]
            return (max(v) - min(v)) if v else 0.0

        if not self.samples:
            return None
        name = max(self.samples, key=lambda n: swing(self.samples[n]))
        rows = self.samples[name]
        clocks = [r["sclk_Hz"] for r in rows if r.get("sclk_Hz") is not None]
        top = max(clocks) if clocks else 0.0
        busy = [r for r in rows if (r.get("sclk_Hz") or 0.0) >= 0.5 * top] if top > 0 else rows

        def med(key, scale):
            v = sorted(r[key] * scale for r in busy if r.get(key) is not None)
            return v[len(v) // 2] if v else None

        def top_of(key, scale):
            v = [r[key] * scale for r in busy if r.get(key) is not None]
            return max(v) if v else None

        out = {"card": name, "chosen_by": "pci address of the device" if self.pci else "largest clock swing among the visible cards",
               "samples": len(rows), "samples_under_load": len(busy),
               "socket_w_median": med("power_uW", 1e-6), "socket_w_max": top_of("power_uW", 1e-6), "cap_w": med("cap_uW", 1e-6),
               "sclk_mhz_median": med("sclk_Hz", 1e-6), "sclk_mhz_max": top_of("sclk_Hz", 1e-6)}
        if out["socket_w_median"] and out["cap_w"]:
            out["frac_of_cap"] = out["socket_w_median"] / out["cap_w"]
        out["note"] = ("hwmon power1_input / freq1_input of the card, polled every ~2 ms over the headline's settling and timed repetitions; "
                       "statistics over the samples with the shader clock at >= half its maximum in the region")
        return out


def device_pci_address(index: int) -> str | None:
    try:
        import torch
        p = torch.cuda.get_device_properties(index)
        return f"{int(p.pci_domain_id):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}.0"
    except Exception:                                     # noqa: BLE001 — telemetry only: never in the way of the measurement
        return None


def roofline_object(kernel: str, nx: int, ny: int, cells_per_launch: float, launch_profile, avg_launch_s: float, launches: int,
                    steps: int, pmc: dict | None, scale: float = 1.0, kernel_span_s: float | None = None) -> dict:
    """The `roofline` object of the line.  launch_profile: [(steps advanced, us)] per step-kernel launch of a profiled
    repetition of this very run (HIP events on the kernels' stream), or None: then only the whole-run average is known.
    `scale`: a rank's launches advance `scale` x the cells the PMC passes profiled (partitioned runs of the same kernel).

      achieved / peak / frac / traffic   the DOMINANT kernel (the instantiation that advances most of the steps):
                                         PHYSICAL HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate
                                         passes, profiles/<round>/) over that kernel's average launch duration in THIS run
      by_section_8d                      SURVEY.md §8(d)'s own accounting: 108 B x cells x steps of the launch over the same
                                         duration — exceeds 1 because lbm_multi_kernel<K> makes one HBM pass per K steps
      run_mix                            every instantiation the run launched (a 20-step run is 4 x K=3 + 2 x K=4): launches,
                                         live duration, profiled bytes; frac_hbm_physical_run = all bytes over all kernel time
      limits.valu                        VALU busy share of a PROFILED pass (a constant of the commit, not of this run)
      limits.power                       added by main(): socket power / shader clock sampled live during the headline (PowerSampler)"""
    by_k: dict[int, list[float]] = {}
    time_scale = 1.0
    if launch_profile:
        # The events around every launch of the profiled repetition stretch it by a per cent or two.  The TIMED repetitions
        # carry only two events (first launch .. last launch, `kernel_span_s`): the profiled durations are scaled so that they
        # add up to that span — the profiled repetition supplies the SPLIT between instantiations, the timed region the time.
        # (On the 64 x 23 / 768-lane launch the bracketed repetition of a 20-step region runs 20 - 25 % longer than a timed one: an event
        # between two launches keeps the second from being set up behind the first.  The factor is reported: `launch_time_scale`.)
        total_us = sum(float(us) for _, us in launch_profile)
        raw_scale = kernel_span_s * 1e6 / total_us if (kernel_span_s and total_us > 0) else None
        # accepted band 0.70 .. 1.05: the bracketed repetition can only be LONGER than a timed one, by at most the
        # 20 - 25 % seen on 20-step regions; a factor outside the band is not applied and the line says so (`launch_time_scale_rejected`)
        if raw_scale is not None and 0.70 <= raw_scale <= 1.05:
            time_scale = raw_scale
        for k, us in launch_profile:
            by_k.setdefault(int(k), []).append(float(us) * time_scale)
        unscaled_by_k: dict[int, list[float]] = {}
        for k, us in launch_profile:
            unscaled_by_k.setdefault(int(k), []).append(float(us))
    kernels = (pmc or {}).get("kernels", {})
    family = kernel.split("<")[0]
    roof = {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None, "kernel": kernel,
            "launches": launches, "avg_launch_ms_whole_run": avg_launch_s * 1e3}
    if not by_k:
        # no per-launch timing (partitioned run): the whole run's average launch, the run's average steps per launch
        spl = steps / max(launches, 1)
        algo = ALGO_BYTES_PER_CELL * cells_per_launch * spl / avg_launch_s / 1e9
        roof["by_section_8d"] = {"achieved": algo, "frac": algo / HBM_PEAK_GBS, "unit": "GB/s", "steps_per_launch": spl,
                                 "note": "108 B x cells x steps per launch / average launch time; exceeds 1: one HBM pass per K steps"}
        entry = kernels.get(kernel)
        if entry:
            hbm = entry["hbm_bytes_per_launch"] * scale * (spl / entry["steps_per_launch"])
            roof.update({"achieved": hbm / avg_launch_s / 1e9, "frac": hbm / avg_launch_s / 1e9 / HBM_PEAK_GBS, "traffic": hbm,
                         "scaled_from_single_gpu_pmc": True,
                         "note": "a rank's launches run the profiled kernel on fewer tiles: its per-launch bytes scaled by the cells and "
                                 "steps a launch advances here, over this run's average launch time"})
        return roof
    dominant = max(by_k, key=lambda k: k * len(by_k[k]))
    mix, bytes_run, time_run = {}, 0.0, 0.0
    for k, durs in sorted(by_k.items()):
        name = f"{family}<{k}>" if "multi" in family or "tile" in family else kernel
        t = sum(durs) / len(durs) * 1e-6
        entry = kernels.get(name)
        m = {"kernel": name, "launches": len(durs), "avg_launch_ms": t * 1e3, "min_launch_ms": min(durs) * 1e-3,
             "avg_launch_ms_bracketed_unscaled": sum(unscaled_by_k[k]) / len(unscaled_by_k[k]) * 1e-3,
             "by_section_8d_frac": ALGO_BYTES_PER_CELL * cells_per_launch * k / t / 1e9 / HBM_PEAK_GBS}
        if entry:
            hbm = entry["hbm_bytes_per_launch"] * scale
            m.update({"hbm_bytes_per_launch": hbm, "hbm_GBps": hbm / t / 1e9, "frac_hbm_physical": hbm / t / 1e9 / HBM_PEAK_GBS,
                      "frac_hbm_physical_bracketed_unscaled": hbm / (m["avg_launch_ms_bracketed_unscaled"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "bytes_per_cell_step": hbm / (cells_per_launch * k)})
            bytes_run += hbm * len(durs)
            time_run += t * len(durs)
        mix[f"K{k}"] = m
    d = mix[f"K{dominant}"]
    roof.update({"kernel": d["kernel"], "avg_launch_ms": d["avg_launch_ms"], "steps_per_launch": dominant, "run_mix": mix,
                 "launch_time_scale": time_scale,
                 **({"launch_time_scale_rejected": raw_scale} if (raw_scale is not None and time_scale == 1.0 and abs(raw_scale - 1.0) > 1e-12) else {}),
                 "by_section_8d": {"achieved": d["by_section_8d_frac"] * HBM_PEAK_GBS, "frac": d["by_section_8d_frac"], "unit": "GB/s",
                                   "bytes_per_launch": ALGO_BYTES_PER_CELL * cells_per_launch * dominant,
                                   "note": "SURVEY.md §8(d): 108 B (18 reads + 9 writes) x cells x steps of the launch / its average "
                                           "duration / 8 TB/s; exceeds 1: lbm_multi_kernel<K> makes one HBM pass per K steps"}})
    if "hbm_bytes_per_launch" in d:
        roof.update({"achieved": d["hbm_GBps"], "frac": d["frac_hbm_physical"], "traffic": d["hbm_bytes_per_launch"],
                     "frac_hbm_physical": d["frac_hbm_physical"],
                     "frac_hbm_physical_run": (bytes_run / time_run / 1e9 / HBM_PEAK_GBS) if time_run > 0 else None,
                     "pmc_source": f"profiles/{PROFILE_ROUND}/roofline.json (scripts/make_roofline.py over the rocprofv3 --pmc CSVs beside it)",
                     "note": "achieved = PHYSICAL HBM bytes per launch of the dominant kernel (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate "
                             "passes) / its average launch duration in this run (HIP events on the kernels' stream: the split between "
                             "instantiations from a repetition with events around every launch, scaled by launch_time_scale to the "
                             "first-launch..last-launch span of the timed repetitions); frac = achieved / 8 TB/s; traffic = those bytes"})
        entry = kernels.get(d["kernel"], {})
        if "frac_valu_profiled" in entry:
            roof["limits"] = {"hbm": {"achieved": d["hbm_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": d["frac_hbm_physical"]},
                              "valu": {"frac": entry["frac_valu_profiled"], "kind": "profiled-pass constant",
                                       "note": "4 x SQ_ACTIVE_INST_VALU busy cycles over the cycles of the SAME profiled pass (GRBM_GUI_ACTIVE / 8 "
                                               "XCDs x 1024 SIMDs): a property of the commit's kernel, it does not move with this run's timing"}}
            roof["lds_bank_conflict_frac"] = entry.get("lds_bank_conflict_frac")
    else:
        roof["note"] = f"no PMC summary for {d['kernel']} in profiles/{PROFILE_ROUND}/roofline.json: only the §8(d) figure"
    return roof


def main() -> int:
    argv = sys.argv[1:]
    args = parse_args(argv)
    t_start = time.time()
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
    budget = args.budget_s if args.budget_s > 0 else 420.0
    start_watchdog(rank, world, budget, t_start)

    def left() -> float:
        return budget - (time.time() - t_start)

    stage("import torch")
    import datetime
    import torch
    import mpilattice_boltzmann_amd as lbm
    torch.cuda.set_device(local_rank)
    # a device-side wait of the peer-to-peer loop gives up after this long (a failed or missing peer, or ranks entering a
    # run this far apart): 10 s keeps a dead transport from eating the budget the fall-backs need
    os.environ.setdefault("LBM_P2P_TIMEOUT_MS", "10000")
    dist = None
    # Control plane: gloo (LBM_DIST_BACKEND=nccl selects torch's NCCL group instead).  Handles, agreement flags and
    # barriers are host objects; gloo's collectives time out with an exception where an NCCL collective that never
    # completes takes the process down, and with gloo the peer-to-peer loop needs no RCCL anywhere.  The RCCL loop
    # brings its own communicator (ncclCommInitRank on an id passed over this group); the torch loop, the last resort,
    # then stages its halo rows through the host.
    backend = os.environ.get("LBM_DIST_BACKEND", "gloo")
    if world > 1:
        stage("process group")
        import torch.distributed as dist
        ctl_timeout = datetime.timedelta(seconds=max(30.0, min(120.0, budget / 3)))
        if backend != "nccl":
            # one node: the loopback interface always resolves (the container's hostname may not); if gloo cannot be brought
            # up on it, or at all, the control plane falls back to torch's NCCL group rather than ending the run here
            had_ifname = "GLOO_SOCKET_IFNAME" in os.environ
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            try:
                dist.init_process_group(backend, timeout=ctl_timeout)
            except Exception as e:      # noqa: BLE001 - every rank fails the same way (same environment) or the rendezvous times out for all
                sys.stderr.write(f"bench.py: rank {rank}: {backend} control group failed ({e}); trying nccl\n")
                if not had_ifname:
                    os.environ.pop("GLOO_SOCKET_IFNAME", None)
                backend = "nccl"
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=ctl_timeout)
    on_host = backend != "nccl"
    stage("build")
    if rank == 0:
        lbm.build()                                       # no-op when lib/ is current; other ranks wait below
    if dist is not None:
        dist.barrier()
    lbm.load_library()
    nx, ny = (int(v) for v in args.workload.lower().split("x"))
    params = lbm.Params(nx, ny, args.steps, 10, 0.1, 0.005, 1.85)
    stage("synthetic deck")
    obstacles = lbm.synthetic_obstacles(nx, ny, 0.005, 42, True)
    flags = lbm._capi.FLAG_FORCE_HALO if args.ring else 0
    partitioned = world > 1 or args.ring
    head_grid = None
    if args.rank_grid == "auto":                      # lbm_choose_rank_grid: row blocks (None) or the tile grid with the least redundant work
        head_grid = lbm.choose_rank_grid(params, world, flags) if partitioned else None
    elif args.rank_grid:
        try:
            head_grid = tuple(int(v) for v in args.rank_grid.lower().split("x"))
            assert len(head_grid) == 2 and head_grid[0] * head_grid[1] == world and partitioned
        except (ValueError, AssertionError):
            if rank == 0:
                emit({"error": f"--rank-grid {args.rank_grid}: expected PXxPY with PX * PY = {world} ranks (and --ring on one GPU)", "n_gpus": world})
            return 2

    if head_grid == (1, 1) and not args.column_block:
        os.environ["LBM_TUNE_TILE_GHOST_ROWS"] = "1"       # read by lbm_tile_layout_of: the one rank is its own south / north neighbour through the row push

    def default_grid(n: int):
        """The rank grid of the `p2p_tiles` variant: as square as n allows, the longer side along x (8 -> 4 x 2)."""
        py = max(d for d in range(1, int(n ** 0.5) + 1) if n % d == 0)
        return (n // py, py)

    def fail(message: str) -> int:
        """A rank-symmetric failure: one JSON line with the reason (the self-launcher reads it), exit 1."""
        if rank == 0:
            emit({"error": message, "n_gpus": world, "exchange": args.exchange})
        if dist is not None:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception:           # noqa: BLE001 - a rank is gone: nothing to meet
                pass
        return 1

    def agree(flag: bool) -> bool:
        """True iff true on every rank."""
        if dist is None:
            return flag
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cpu" if on_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(int(t.item()))

    def min_over_ranks(v: float) -> float:
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device="cpu" if on_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item())

    def max_over_ranks(values: list[float]) -> list[float]:
        if dist is None:
            return values
        t = torch.tensor(values, dtype=torch.float64, device="cpu" if on_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.cpu()]

    def gather(obj) -> list:
        if dist is None:
            return [obj]
        box = [None] * world
        dist.all_gather_object(box, obj)
        return box

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- the single-GPU reference every partitioned run is checked against, computed once per (deck, step count) ----
    ref_cache: dict = {}

    def single_gpu_reference(p, obst, steps_done: int, y0: int, y1: int, n_av: int):
        """(digest of rows [y0, y1), last n_av av_vels) of ONE GPU doing the whole grid for steps_done steps."""
        key = (p.nx, p.ny, steps_done, y0, y1)
        if key not in ref_cache:
            whole = lbm.Simulation(p, obst, device=local_rank)
            av = whole.run(steps_done)
            ref_cache[key] = (whole.partition.checksum(y0, y1), av)
            whole.close()
        digest, av = ref_cache[key]
        return digest, av[len(av) - n_av:] if n_av else np.zeros(0, np.float32)

    def check_against_single_gpu(sim, p, obst, av_last, steps_done):
        """The partitioned run against ONE GPU doing the whole grid, bit for bit: 64-bit digest of this rank's
        rows (lbm_state_checksum) and the last run's av_vels — the multi-GPU parity test, run where the GPUs are.
        Never raises (a failure on one rank must not leave the others alone in the next collective)."""
        try:
            if sim.rank_grid is not None:                 # tiles: the ranks' digests add up to the whole grid's (lbm_state_checksum is additive)
                mine = None
                try:
                    mine = sim.partition.checksum()
                except lbm.LbmError as e:
                    sys.stderr.write(f"bench.py: rank {rank}: state digest: {e}\n")
                every = gather(mine)                      # (every rank takes part, whatever happened to its own digest)
                digest, av_ref = single_gpu_reference(p, obst, steps_done, 0, p.ny, len(av_last))
                same = all(d is not None for d in every) and sum(every) % (1 << 64) == digest
            else:
                y0, y1 = sim.partition.y0, sim.partition.y0 + sim.partition.ny_local
                digest, av_ref = single_gpu_reference(p, obst, steps_done, y0, y1, len(av_last))
                same = sim.partition.checksum() == digest
            av_err = float(np.max(np.abs(av_last.astype(np.float64) - av_ref.astype(np.float64)) / av_ref.astype(np.float64))) if len(av_last) else 0.0
            return bool(same and av_err < 1e-6), same, av_err
        except (lbm.LbmError, ValueError, FloatingPointError) as e:
            sys.stderr.write(f"bench.py: rank {rank}: parity check could not run: {e}\n")
            return False, False, float("nan")

    verify_on = partitioned and not args.no_verify

    def set_up(p, obst, mode: str, step_allreduce: bool, warmup: int, fl: int, grid=None):
        """One loop on every rank: create, warm up, check.  Returns (sim or None, note) — rank-symmetric."""
        note, ok, sim = None, False, None
        try:
            sim = lbm.Simulation(p, obst, device=local_rank, flags=fl, distributed=world > 1, exchange=mode, strict=True,
                                 step_allreduce=step_allreduce, rank_grid=grid)
            ok = True
        except lbm.LbmError as e:                         # raised on every rank together
            sim, note = None, f"set-up: {e}"
        if ok:
            try:
                av_w = sim.run(warmup)                    # untimed
                if verify_on:
                    good, same, av_err = check_against_single_gpu(sim, p, obst, av_w, warmup)
                    if not good:
                        ok, note = False, f"parity after the warm-up: state digest equal {same}, av_vels rel err {av_err:.2e}"
            except lbm.LbmError as e:
                ok, note = False, f"warm-up: {e}"
        ok = agree(ok)
        if not ok and sim is not None:
            sim.close()
            sim = None
        return sim, (None if ok else (note or "failed on another rank"))

    settle_log: dict = {}
    kernel_spans: dict = {}

    def timed(sim, steps: int, reps: int, what: str = "headline"):
        """reps x (barrier + device sync; EXACTLY `steps` steps + the reduction; device sync) -> per-rep seconds, MAX over
        ranks; last av_vels.  The barrier that closes one bracket is the next repetition's: an NCCL barrier inside the
        region would add its own ~0.1 ms to a 1 ms region of steps.  A rank whose run fails keeps meeting the others at
        the barriers of the remaining repetitions, then every rank raises together.

        Short regions are preceded by UNTIMED repetitions of the same region until ~60 ms of back-to-back runs have passed
        (`timing.settle_reps`): the part needs that long to settle — the same for a 6 ms region on one GPU (7.7, 6.9, 6.6,
        6.4, 6.4 ... ms) and a 1 ms region of an 8-GPU rank's share (1.33, 1.05, 1.10, 1.14, 1.12, 1.08, 1.06, 1.04 ... 0.98 ms
        after twenty) — and W = 5 warm-up steps are 0.3 - 1.6 ms.  The count is agreed on by the ranks from the first run."""
        times, av, err = [], None, None
        spans = kernel_spans.setdefault(what, [])         # (device ms first..last step kernel, launches) of every repetition, this rank
        del spans[:]

        def one():
            nonlocal av, err
            sync_all()
            if err is not None:
                return 0.0
            t0 = time.perf_counter()
            try:
                av = sim.run(steps)
            except lbm.LbmError as e:
                err = str(e)
            torch.cuda.synchronize()                      # this rank's device work is complete: the reduction made it wait
            dt = time.perf_counter() - t0                 # for every rank's sums
            try:
                spans.append(sim.partition.last_run_kernel_ms())
            except lbm.LbmError:
                spans.append((0.0, 0))
            return dt

        settle = 0
        if steps <= 50 and reps > 1:
            first = max_over_ranks([one()])[0]
            settle = 1 + (0 if first <= 0 else int(min(63, max(0, round(0.06 / first)))))
            for _ in range(settle - 1):
                one()
        settle_log[what] = settle
        for _ in range(max(1, reps)):
            times.append(one())
        if not agree(err is None):
            raise lbm.LbmError(err or "the run failed on another rank")
        return max_over_ranks(times), av

    # ---- headline ----------------------------------------------------------------------------------------------
    if not partitioned:
        modes = ["auto"]
    elif head_grid is not None:
        modes = ["p2p"]
    elif args.exchange != "auto":
        modes = [args.exchange]
    else:
        modes = ["rccl"] if args.step_allreduce else ["p2p", "rccl", "torch"]
    # ranks that share a device (testing aid) cannot form an RCCL communicator: said up front, not found out by a hang
    devices = gather((os.uname().nodename, local_rank))
    shared_gpu = len(set(devices)) < len(devices)
    attempts, sim = [], None
    for mode in modes:
        stage(f"headline: {mode} loop set-up and warm-up")
        if mode == "rccl" and shared_gpu:
            attempts.append({"exchange": mode, "ok": False, "error": "not usable: ranks share a GPU (RCCL refuses duplicate devices)"})
            continue
        sim, note = set_up(params, obstacles, mode, args.step_allreduce, args.warmup, flags, head_grid)
        attempts.append({"exchange": mode, "ok": sim is not None, **({"error": note} if sim is None else {})})
        if sim is not None:
            break
        if rank == 0:
            sys.stderr.write(f"bench.py: exchange={mode} not usable ({note}); trying the next one\n")
    if sim is None:
        return fail("no exchange mode completed: " + json.dumps(attempts))

    stage("headline: timed repetitions")
    sampler = PowerSampler(device_pci_address(local_rank)).start() if rank == 0 and not args.no_power else None
    try:
        times, av = timed(sim, args.steps, args.reps)
        # the kernel span of the MEDIAN repetition (the one `value` is), not of the last one
        reps_spans = kernel_spans["headline"][-len(times):]
        kernel_ms, launches = reps_spans[int(np.argsort(times)[len(times) // 2])]
    except lbm.LbmError as e:
        return fail(f"run: {e}")
    finally:
        power = sampler.stop() if sampler is not None else None
    elapsed = float(np.median(times))
    assert av.shape == (args.steps,) and np.all(np.isfinite(av)) and np.all(av > 0)
    desc = sim.partition.describe()
    what = sim.describe()
    sim_layout = dict(sim.layout)
    steps_done = args.warmup + args.steps * (max(1, args.reps) + settle_log.get("headline", 0))

    # one more repetition with HIP events around every launch: the per-kernel launch durations behind `roofline`
    # (single GPU) or where a rank's run goes (`phases`, peer-to-peer loop).  Not timed: the events perturb the schedule.
    launch_profile, phases = None, None
    if not args.no_phases:
        stage("profiled repetition")
        mine, perr = None, None
        try:
            if not partitioned:
                sim.partition.set_profile(True)
                sim.run(args.steps)
                launch_profile = sim.partition.launch_profile()
                sim.partition.set_profile(False)
                steps_done += args.steps
            elif what["loop"] == "p2p":
                sim._p2p.set_profile(True)
                sync_all()
                av = sim.run(args.steps)
                mine = sim._p2p.phases()
                sim._p2p.set_profile(False)
                steps_done += args.steps
        except lbm.LbmError as e:
            perr = str(e)
        if not agree(perr is None):                       # every rank leaves together
            return fail(f"profiled run: {perr or 'failed on another rank'}")
        if mine is not None:
            every = gather(mine)
            names = list(mine)
            phases = {"unit": "us (macro_steps: a count)", "steps": args.steps,
                      "max_over_ranks": {n: max(r[n] for r in every) for n in names},
                      "mean_over_ranks": {n: sum(r[n] for r in every) / len(every) for n in names},
                      "per_rank": every,
                      "note": "one extra repetition with HIP timing events around every launch (lbm_p2p_set_profile; the events stretch "
                              "the run by ~10 %): setup = run start -> first step kernel; macro_step_steady = interior launch to interior "
                              "launch without the first and last macro-step; push_* = the push kernel INCLUDING its wait for both "
                              "neighbours' rows; reduce = last step kernel -> global sums in host memory; host_overhead = wall time of "
                              "the call - device span"}

    verify = None
    if verify_on:
        stage("headline: parity check")
        good, same, av_err = check_against_single_gpu(sim, params, obstacles, av, steps_done)
        good = agree(good)
        verify = {"ok": good, "what": f"every rank's rows bit-identical (64-bit state digest) to a single-GPU run of the whole grid on the same "
                                      f"device, after the warm-up ({args.warmup} steps) and after all {steps_done} steps; av_vels of the last "
                                      f"repetition within 1e-6", "av_vels_max_rel": av_err}
        if not good:
            sim.close()
            return fail(f"parity: partitioned run ({what['loop']} loop) differs from the single-GPU run after {steps_done} steps "
                        f"(state digest equal: {same}, av_vels rel err {av_err:.2e})")
    sim.close()

    out = None
    if rank == 0:
        cells = nx * ny
        mlups = cells * args.steps / elapsed / 1e6
        # a macro-step of the edge-stream schedule is two concurrent launches (interior + edge tiles): counted as ONE
        # launch over all of the rank's cells, so that bytes and time refer to the same thing
        two_launches = partitioned and (what["p2p"] is None or "edge stream" in what["p2p"])
        n_launch = max(launches // 2 if two_launches else launches, 1)
        lay = sim_layout
        if partitioned and lay.get("macro_k", 0) > 0:    # K-step loops: the launches the run's groups hold (the first launch of a group is
            groups = lbm.plan_groups(lay["macro_k"], lay["ghost"], lay["group"], args.steps)      # an interior + an edge launch: one launch here)
            n_launch = max(sum(len(g) for g in groups), 1)
        avg_launch_s = kernel_ms / 1e3 / n_launch
        cells_per_launch = float(desc["cells_per_launch"])
        pmc = load_roofline()
        scale = 1.0
        if partitioned and pmc is not None:
            pnx, pny = (int(v) for v in pmc.get("workload", "8192x8192").split("x"))
            scale = cells_per_launch / float(pnx * pny)
        elif pmc is not None and pmc.get("workload") != f"{nx}x{ny}":
            pmc = None
        roof = roofline_object(desc["kernel"], nx, ny, cells_per_launch, launch_profile, avg_launch_s, n_launch, args.steps, pmc, scale,
                               kernel_span_s=kernel_ms / 1e3)
        if power is not None:
            roof.setdefault("limits", {})["power"] = power
        exchange_txt = {"p2p": "direct peer-to-peer stores into the neighbours' ghost rows (xGMI), flags + one-wave wait kernels, "
                               "all-gather + local sum after the loop",
                        "rccl": "RCCL send/recv on a side stream, " + ("one all-reduce per macro-step" if what["step_allreduce"] else "one all-reduce after the loop"),
                        "torch": "torch.distributed P2P ops, one all-reduce after the loop", "single": ""}[what["loop"]]
        k = max(what["macro_k"], 1)
        how = (f"launches of K = {k} steps, one halo exchange per group of up to {sim_layout.get('group', 1)} launches ({sim_layout.get('ghost', 0)} ghost rows: the "
               f"first launch of a group advances the ghost rows the later ones read)" if what["macro_k"] else "one halo exchange per step")
        if world == 1:
            part_txt = "single GPU" if not args.ring else f"1-rank ring (self exchange), {how}: {exchange_txt}"
        else:
            part_txt = f"{world} row blocks (d2q9-bgk.c:834-862), {how}: {exchange_txt}"
        if head_grid is not None:
            part_txt = (f"tile (2-D) decomposition, {head_grid[0]} x {head_grid[1]} ranks: blocks of {sim_layout.get('nx_local')} x {sim_layout.get('ny_local')} cells with "
                        f"{sim_layout.get('ghost_x')} ghost columns and {sim_layout.get('ghost')} ghost rows per side; per exchange the columns travel west / east, "
                        f"then whole storage rows south / north; {how}: {exchange_txt}")
        out = {
            "metric": "MLUPS", "value": mlups, "unit": "MLUPS", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"synthetic {nx}x{ny} D2Q9-BGK deck (walls + p=0.005 random obstacles, splitmix64 seed 42), "
                                   f"density 0.1 accel 0.005 omega 1.85", "nx": nx, "ny": ny, "partitioning": part_txt,
                       "loop": what["loop"], "macro_k": what["macro_k"], "ghost_rows": sim_layout.get("ghost", 0), "launches_per_exchange": sim_layout.get("group", 1),
                       "rank_grid": list(head_grid) if head_grid is not None else None,
                       "rccl_nranks": what["rccl_nranks"], "p2p": what["p2p"],
                       "step_allreduce": what["step_allreduce"], "kernel": desc["kernel"], "control_plane": backend if world > 1 else None},
            "timing": {"reps": len(times), "settle_reps": settle_log.get("headline", 0),
                       "statistic": "median over reps of (max over ranks of the time of EXACTLY `steps` steps): every rank starts behind a barrier + "
                                    "device synchronise and stops its clock when its own device work, which ends with the global reduction, is complete; "
                                    "regions of <= 50 steps are preceded by `settle_reps` untimed repetitions of the same region (~60 ms of back-to-back runs: "
                                    "the part's clocks and caches take that long to settle whatever the region length)",
                       "ms_per_rep": [t * 1e3 for t in times]},
            "pct_hbm_roofline": 100.0 * mlups / world / (HBM_PEAK_GBS * 1e9 / ALGO_BYTES_PER_CELL / 1e6),
            "roofline": roof,
        }
        if phases is not None:
            out["phases"] = phases
        if verify is not None:
            out["parity_check"] = verify
        if partitioned:
            out["exchange_attempts"] = attempts
        bank(out)                                          # from here on the watchdog has a line to print

    # ---- N = 1: the same deck with the other two forms of the sum|u| terms (kernels/common.h finish_pair_lo) -----------
    if world == 1 and not args.ring and not args.no_variants:
        variants = {}
        # av_vels of the three forms are compared from the SAME state: the deck's first `steps` steps in fresh contexts (the
        # headline's context has advanced by its own number of settling repetitions)
        base = lbm.Simulation(params, obstacles, device=local_rank, flags=flags)
        av_first = base.run(args.steps).astype(np.float64)
        base.close()
        for vname, vflag, marker, vnote in (
                ("fast_av_vels", lbm._capi.FLAG_FAST_AVVELS, "fast av_vels",
                 "LBM_FLAG_FAST_AVVELS: each cell's sum|u| term in plain float"),
                ("exact_av_vels", lbm._capi.FLAG_EXACT_AVVELS, "double-precision av_vels terms",
                 "LBM_FLAG_EXACT_AVVELS: each term in double precision, correctly rounded (rounds 1-2's form), instead of the default's "
                 "compensated float sums (relative error ~2^-44 per term)")):
            stage(f"variant: {vname}")
            alt = lbm.Simulation(params, obstacles, device=local_rank, flags=flags | vflag)
            if marker in alt.partition.describe()["kernel"]:
                av_alt = alt.run(args.steps).astype(np.float64)
                talt, _ = timed(alt, args.steps, args.reps, what=vname)
                med = float(np.median(talt))
                variants[vname] = {"value": nx * ny * args.steps / med / 1e6, "unit": "MLUPS", "ms_per_step": med / args.steps * 1e3,
                                   "av_vels_max_rel_diff_to_default": float(np.max(np.abs(av_alt - av_first) / av_first)),
                                   "av_vels_values_that_differ": int(np.count_nonzero(av_alt != av_first)),
                                   "note": vnote + "; populations identical bit for bit; NOT the headline.  Measured in another context of this "
                                           "process: where a context's grids land moves its step time by 3-5 % either way (DESIGN.md 4.2), so "
                                           "compare with a same-process A/B (scripts/ab_libs.py) rather than with `value`"}
            alt.close()
        if variants:
            out["variants"] = variants
            bank(out)

    def optional_part(name: str, need_s: float, body):
        """Run `body` if every rank has `need_s` seconds of budget left; record what happened otherwise.  Rank-symmetric."""
        if min_over_ranks(left()) < need_s:
            return {"skipped": f"budget: fewer than {need_s:.0f} s left"}
        stage(name)
        t0 = time.time()
        try:
            res = body()
        except lbm.LbmError as e:
            res = {"error": str(e)}
        if isinstance(res, dict):
            res["seconds"] = round(time.time() - t0, 1)
        return res

    # ---- N = 1: what the power cap leaves when the deck runs on and on, and BASELINE.json configs 2 - 3 (the shipped decks) ------
    def sustained_run():
        """The headline deck for >= 2000 steps x 3, back to back (a region of seconds, not milliseconds: the socket sits at its power
        limit and the shader clock settles ~5 % below the short regions'; also long enough for an outside sampler to see the GPU busy)."""
        n = max(2000, args.steps)
        s2 = lbm.Simulation(lbm.Params(nx, ny, n, 10, 0.1, 0.005, 1.85), obstacles, device=local_rank, flags=flags)
        s2.run(args.warmup)
        sampler2 = PowerSampler(device_pci_address(local_rank)).start() if not args.no_power else None
        try:
            t2, _ = timed(s2, n, 3, what="sustained")
        finally:
            pw = sampler2.stop() if sampler2 is not None else None
        s2.close()
        worst = max(t2)
        return {"steps_per_rep": n, "reps": 3, "ms_per_rep": [t * 1e3 for t in t2], "value": nx * ny * n / worst / 1e6, "unit": "MLUPS",
                "ms_per_step": worst / n * 1e3, "statistic": "the SLOWEST of three back-to-back repetitions (the part warms up over them)",
                **({"power": pw} if pw else {})}

    def shipped_decks():
        """BASELINE.json configs 2 and 3 (and the other two shipped decks): the reference's own input files through lbm_run on this GPU, whole
        runs, timed over the reference's window (d2q9-bgk.c:278-398), then its acceptance check — final_state.dat byte for byte against
        the reference binary's (sha256 in tests/golden/digests.json), the Reynolds line as a string, av_vels against the SHIPPED golden
        under check/check.py's rule (1 %), and against the reference binary's float run at the sampled steps."""
        import hashlib
        import shutil
        import tempfile
        dig = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))
        res = {}
        for name in ("128x128", "128x256", "256x256", "1024x1024"):
            d = dig[name]
            p2 = lbm.read_params(os.path.join(DECKS, d["params"]))
            o2, _ = lbm.read_obstacles(os.path.join(DECKS, d["obstacles"]), p2.nx, p2.ny)
            s3 = lbm.Simulation(p2, o2, device=local_rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            av3 = s3.run(p2.max_iters)
            torch.cuda.synchronize()
            sec = time.perf_counter() - t0
            obs = s3.gather_observables()
            tmp = tempfile.mkdtemp(prefix="lbm_deck_")
            try:
                s3.write_values(av3, tmp, observables=obs)
                h = hashlib.sha256()
                with open(os.path.join(tmp, "final_state.dat"), "rb") as fh:
                    for blk in iter(lambda: fh.read(1 << 20), b""):
                        h.update(blk)
                golden = lbm.checker.load_av_vels(os.path.join(ROOT, "tests", "golden", "check", f"{name}.av_vels.dat.gz"))
                rep = lbm.checker.diff_values(golden, av3)
                idx = np.asarray(d["av_sample_steps"])
                res[name] = {"steps": p2.max_iters, "seconds": sec, "us_per_step": sec / p2.max_iters * 1e6,
                             "value": p2.nx * p2.ny * p2.max_iters / sec / 1e6, "unit": "MLUPS", "kernel": s3.partition.describe()["kernel"],
                             "final_state_sha256_equals_reference": h.hexdigest() == d["final_state_sha256"],
                             "reynolds_line_equals_reference": ("Reynolds number:\t\t%.12E" % s3.reynolds(observables=obs)) == d["reynolds_line"],
                             "av_vels_vs_shipped_golden_max_pct": abs(rep.max_diff_pcnt), "check_py_passes": not rep.failed(1.0),
                             "av_vels_max_rel_to_reference_samples": float(np.max(np.abs(av3[idx] - np.asarray(d["av_sample_values"])) / np.asarray(d["av_sample_values"]))),
                             "reference_1core_s_in_build_container": d.get("ref_elapsed_s")}
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
                s3.close()
        return res

    if world == 1 and not args.ring and not args.no_secondary:
        out["sustained"] = optional_part("sustained: headline deck, 3 x >= 2000 steps", 30.0, sustained_run)
        bank(out)
        out["secondary"] = {"shipped_decks": optional_part("secondary: the four shipped decks", 40.0, shipped_decks)}
        bank(out)

    # ---- partitioned runs: the RCCL loop, and north_star's per-step all-reduce, on the same deck -----------------------
    def variant(mode: str, step_allreduce: bool, env: dict | None = None, grid=None):
        def body():
            saved = {k: os.environ.get(k) for k in (env or {})}
            os.environ.update(env or {})                    # knobs read at lbm_create; the same on every rank
            try:
                return body_inner()
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v

        def body_inner():
            if mode == "rccl" and shared_gpu:
                return {"error": "not usable: ranks share a GPU (RCCL refuses duplicate devices)"}
            vs, note = set_up(params, obstacles, mode, step_allreduce, args.warmup, flags, grid)
            if vs is None:
                return {"error": note}
            vt, vav = timed(vs, args.steps, args.reps, what=f"variant {mode} {step_allreduce}")
            done = args.warmup + args.steps * (max(1, args.reps) + settle_log.get(f"variant {mode} {step_allreduce}", 0))
            res = {"value": nx * ny * args.steps / float(np.median(vt)) / 1e6, "unit": "MLUPS", "ms_per_step": float(np.median(vt)) / args.steps * 1e3,
                   "ms_per_rep": [t * 1e3 for t in vt], **{k2: v for k2, v in vs.describe().items() if k2 in ("loop", "macro_k", "rccl_nranks", "step_allreduce")},
                   "ghost_rows": vs.layout.get("ghost"), "launches_per_exchange": vs.layout.get("group")}
            if grid is not None:
                res.update(rank_grid=list(grid), block=[vs.layout.get("nx_local"), vs.layout.get("ny_local")], ghost_columns=vs.layout.get("ghost_x"), p2p=vs.describe()["p2p"])
            if verify_on:
                good, same, av_err = check_against_single_gpu(vs, params, obstacles, vav, done)
                res["parity_ok"] = agree(good)
            vs.close()
            return res
        return body

    if partitioned and not args.no_variants:
        variants = {}
        # p2p_exchange_every_launch: rounds 1-3's loop (K ghost rows, an exchange before every launch) beside round 4's one exchange per
        # group of launches — on real links the difference is what the deeper halo is worth
        for name, mode, sar, venv in (("p2p", "p2p", False, None), ("p2p_exchange_every_launch", "p2p", False, {"LBM_TUNE_MACRO_GHOST": "0"}),
                                      ("rccl", "rccl", False, None), ("rccl_step_allreduce", "rccl", True, None)):
            if venv is None and (what["loop"], what["step_allreduce"]) == (mode, sar):
                continue                                   # that is the headline
            if venv is not None and (what["loop"] != mode or not what["macro_k"]):
                continue                                   # (only beside a K-step headline of the same loop)
            variants[name] = optional_part(f"variant: {name}", 45.0, variant(mode, sar, venv))
            if rank == 0:
                out["variants"] = variants
                bank(out)
        # the same deck over the tile (2-D) decomposition, as square a rank grid as the ranks allow: what the
        # smaller halo and the four neighbours are worth on real links, beside the headline's row blocks
        if world > 1 and head_grid is None and what["loop"] == "p2p":
            variants["p2p_tiles"] = optional_part("variant: p2p_tiles", 45.0, variant("p2p", False, None, default_grid(world)))
            if rank == 0:
                out["variants"] = variants
                bank(out)

    # ---- BASELINE.json config 4: the shipped 1024x1024 deck on the same ranks ------------------------------------------
    def deck_on_ranks(name: str, modes, steps_override: int = 0):
        digests = json.load(open(os.path.join(ROOT, "tests", "golden", "digests.json")))[name]
        p4 = lbm.read_params(os.path.join(DECKS, digests["params"]))
        o4, _ = lbm.read_obstacles(os.path.join(DECKS, digests["obstacles"]), p4.nx, p4.ny)
        n4 = steps_override if steps_override > 0 else p4.max_iters
        res = {"deck": f"input_{name}.params + obstacles_{name}.dat (tests/golden/decks: the reference's own files)", "steps": n4}
        if name == "1024x1024":
            res.update(reference_published_s=5.90364, reference_published_note="d2q9-bgk_best.out:8-12, 64 MPI ranks on 4 x 16 Xeon E5-2670 cores, 20 000 steps")
        for mode in modes:
            if mode == "rccl" and shared_gpu:
                res[mode] = {"error": "not usable: ranks share a GPU (RCCL refuses duplicate devices)"}
                continue
            if mode == "p2p_tiles" and world == 1:
                continue
            s4, note = set_up(p4, o4, "p2p" if mode == "p2p_tiles" else mode, False, 0, flags, default_grid(world) if mode == "p2p_tiles" else None)
            if s4 is None:
                res[mode] = {"error": note}
                continue
            t4, av4 = timed(s4, n4, 1, what="secondary")
            r4 = {"seconds": t4[0], "us_per_step": t4[0] / n4 * 1e6, "value": p4.nx * p4.ny * n4 / t4[0] / 1e6, "unit": "MLUPS",
                  **{k2: v for k2, v in s4.describe().items() if k2 in ("loop", "macro_k", "p2p", "rccl_nranks")}}
            if verify_on:
                good, same, av_err = check_against_single_gpu(s4, p4, o4, av4, n4)
                r4["parity_ok"] = agree(good)
            if n4 == p4.max_iters:                         # the whole deck: the reference binary's own Reynolds line and av_vels
                re_line = s4.reynolds()
                if rank == 0:
                    r4["reynolds_line_equals_reference"] = ("Reynolds number:\t\t%.12E" % re_line) == digests["reynolds_line"]
                    idx = np.asarray(digests["av_sample_steps"])
                    r4["av_vels_max_rel_to_reference_samples"] = float(np.max(np.abs(av4[idx] - np.asarray(digests["av_sample_values"])) /
                                                                               np.asarray(digests["av_sample_values"])))
            s4.close()
            res[mode] = r4
        return res

    if partitioned and not args.no_secondary:
        sec = optional_part("secondary: input_1024x1024", 40.0, lambda: deck_on_ranks("1024x1024", ("p2p", "rccl", "p2p_tiles"), args.secondary_steps))
        if rank == 0:
            out["secondary"] = {"input_1024x1024": sec}
            bank(out)
        # the three small shipped decks on the same ranks, whole runs (north_star: "MLUPS on the provided grids ... at 1, 2, 4 and 8 GPUs"): launch-
        # bound on one GPU, thinner than K-step mode allows from 4 or 8 ranks on (the one-step loop) — numbers for the record, each checked
        if world > 1 and not args.ring:
            for name in ("256x256", "128x256", "128x128"):
                small = optional_part(f"secondary: input_{name}", 30.0, lambda name=name: deck_on_ranks(name, ("p2p",)))
                if rank == 0:
                    out["secondary"][f"input_{name}"] = small
                    bank(out)

    if rank == 0:
        if world == 1 and not args.ring and not args.no_cpu_baseline:
            stage("cpu baseline")
            out["cpu_baseline"] = cpu_baseline(lbm, params, obstacles, args.cpu_seconds)
        out["wall_s"] = round(time.time() - t_start, 1)
        emit(out)
    stage("tear-down")
    if dist is not None:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception:               # noqa: BLE001 - the line is out; a rank that is gone changes nothing
            pass
    return 0


if __name__ == "__main__":
    sys.exit(main())
