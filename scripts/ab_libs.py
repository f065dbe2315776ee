#!/usr/bin/env python3
"""A/B of different BUILDS of liblbm_d2q9.so in one process on one GPU (box-to-box spread is ~5 %,
larger than most kernel changes):

    python scripts/ab_libs.py --grid 8192x8192 --steps 60 --rounds 4 [--env LBM_TUNE_MULTI_K=3] a.so b.so ...

Each library is dlopen'ed privately, gets its own context on the same synthetic deck, and the runs are
interleaved round-robin; prints device time per step (HIP events inside the library): min / median.
Keep variant builds under mpilattice-boltzmann_amd/lib/variants/ (git-ignored, travels with gpurun)."""
import argparse
import ctypes as C
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import mpilattice_boltzmann_amd as lbm  # noqa: E402  (loads torch's HIP runtime first)
from mpilattice_boltzmann_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="8192x8192")
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--env", action="append", default=[], help="KEY=VALUE set before every lbm_create")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
nx, ny = (int(v) for v in a.grid.split("x"))
for kv in a.env:
    k, v = kv.split("=")
    os.environ[k] = v
p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
obst = np.ascontiguousarray(lbm.synthetic_obstacles(nx, ny, 0.005, 42, True), dtype=np.int32)
free_cells = int(obst.size - obst.sum())
cp = _capi.CParams(p.nx, p.ny, p.max_iters, p.reynolds_dim, p.density, p.accel, p.omega)

ctxs = []
for n_spec, spec in enumerate(a.libs):    # "path.so", "path.so:FLAGS" (lbm_create flags, e.g. 64 = fast av_vels) or "path.so:FLAGS:KEY=VAL,KEY=VAL"
    parts = spec.split(":")               # (environment for that entry's lbm_create only; empty FLAGS = --flags)
    path, fl = parts[0], parts[1] if len(parts) > 1 else ""
    flags = int(fl) if fl else a.flags
    own_env = dict(kv.split("=") for kv in parts[2].split(",")) if len(parts) > 2 and parts[2] else {}
    saved = {k: os.environ.get(k) for k in own_env}
    os.environ.update(own_env)
    spec = f"{n_spec}:{spec}"             # the same library may appear more than once (allocation order matters by ~3 %)
    lib = C.CDLL(os.path.abspath(path))
    for name in ("lbm_create", "lbm_run", "lbm_last_run_kernel_ms", "lbm_destroy", "lbm_last_error"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = _capi._SIGNATURES[name]
    ctx = C.c_void_p()
    if lib.lbm_create(C.byref(ctx), C.byref(cp), free_cells, obst.ctypes.data_as(C.POINTER(C.c_int)), 0, ny, 0, flags):
        raise SystemExit(f"{path}: {lib.lbm_last_error().decode()}")
    ctxs.append((spec, lib, ctx))
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v

av = (C.c_float * a.steps)()
res = {path: [] for path, _, _ in ctxs}
digest, first_av = {}, {}
for path, lib, ctx in ctxs:
    lib.lbm_run(ctx, 12, av)
    digest[path] = bytes(av)[:48]
    first_av[path] = np.array(av[:12], dtype=np.float32)
for r in range(a.rounds):
    for path, lib, ctx in ctxs:
        if lib.lbm_run(ctx, a.steps, av):
            raise SystemExit(f"{path}: {lib.lbm_last_error().decode()}")
        ms, n = C.c_double(), C.c_int()
        lib.lbm_last_run_kernel_ms(ctx, C.byref(ms), C.byref(n))
        res[path].append(ms.value / a.steps * 1e3)
ref = digest[ctxs[0][0]]
for path, lib, ctx in ctxs:
    v = res[path]
    nd = 1 if min(v) >= 20 else 3
    same = "av==first" if digest[path] == ref else "av DIFFERS from first"
    if digest[path] != ref:      # av_vels are floats: how many of the 12 differ, and by how much
        a0, a1 = first_av[ctxs[0][0]].astype(np.float64), first_av[path].astype(np.float64)
        same += f" ({int((a0 != a1).sum())}/12 values, max rel {np.max(np.abs(a1 - a0) / np.abs(a0)):.1e})"
    print(f"{path.split(':', 1)[0]:>2s} {os.path.basename(path.split(':', 1)[1]):50s} min {min(v):8.{nd}f}  med {statistics.median(v):8.{nd}f}  max {max(v):8.{nd}f} us/step   {same}", flush=True)
    lib.lbm_destroy(ctx)
