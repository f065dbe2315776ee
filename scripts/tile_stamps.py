#!/usr/bin/env python3
"""Where the time of one lbm_tile_kernel launch goes (diagnostic build only):

    bash scripts/build_variant.sh tile_stamps -DLBM_TILE_STAMPS=1
    python scripts/tile_stamps.py [--grid 256x256] [--steps 64]

Block 1, lane 0 stamps the shader clock (s_memtime) at the start, after the load phase, after every sub-step and
after the epilogue of the LAST launch; the product build contains no stamp."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import mpilattice_boltzmann_amd as lbm  # noqa: E402
from mpilattice_boltzmann_amd import _capi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--grid", default="256x256")
ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--lib", default=os.path.join(_capi.PKG, "lib", "variants", "tile_stamps.so"))
a = ap.parse_args()
nx, ny = (int(v) for v in a.grid.split("x"))
lib = C.CDLL(os.path.abspath(a.lib))
for name in ("lbm_create", "lbm_run", "lbm_destroy", "lbm_last_error", "lbm_describe"):
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = _capi._SIGNATURES[name]
lib.lbm_debug_tile_stamps.restype, lib.lbm_debug_tile_stamps.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
p = lbm.Params(nx, ny, a.steps, 10, 0.1, 0.005, 1.85)
obst = np.ascontiguousarray(lbm.synthetic_obstacles(nx, ny, 0.005, 42, True), dtype=np.int32)
cp = _capi.CParams(p.nx, p.ny, p.max_iters, p.reynolds_dim, p.density, p.accel, p.omega)
ctx = C.c_void_p()
assert lib.lbm_create(C.byref(ctx), C.byref(cp), int(obst.size - obst.sum()), obst.ctypes.data_as(C.POINTER(C.c_int)), 0, ny, 0, 0) == 0
name = C.create_string_buffer(128)
lib.lbm_describe(ctx, name, 128, None, None)
av = (C.c_float * a.steps)()
for _ in range(3):
    assert lib.lbm_run(ctx, a.steps, av) == 0
st = (C.c_ulonglong * 16)()
assert lib.lbm_debug_tile_stamps(st) == 0
t = [int(v) for v in st]
if "multi" in name.value.decode():      # lbm_multi_kernel<K>: a block in the middle of the launch, its CU shared with two others
    labels = ["sub-step 1 (global loads, relax, LDS writes)", "barrier"] + [f"sub-step {i} (in LDS)" for i in range(2, 9)]
    k = int(name.value.decode().split("<")[1].split(",")[0].split(">")[0])
    labels = labels[:1 + k] + ["epilogue (sums)"]
    labels[k] = f"sub-step {k} (LDS -> global stores)" if k > 1 else labels[k]
else:
    labels = ["load"] + [f"sub-step {i}" for i in range(1, 9)] + ["epilogue"]
print(name.value.decode(), a.grid)
prev = t[0]
for i, lab in enumerate(labels, start=1):
    if t[i] <= prev:
        continue
    print(f"  {lab:46s} {t[i] - prev:7d} cycles")
    prev = t[i]
print(f"  {'total':46s} {prev - t[0]:7d} cycles (shader clock)")
lib.lbm_destroy(ctx)
