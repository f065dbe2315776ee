"""ctypes wrapper of oracle/liboracle_d2q9.so — the CPU restatement of the reference's path.

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, as the checker / reported baseline.  Builds the library with `make -C oracle` when missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_d2q9.so")
CLI_PATH = os.path.join(ORACLE_DIR, "d2q9_oracle")
REF_BIN = os.path.join(ORACLE_DIR, "_ref", "d2q9-bgk_ref")
Q = 9


class OracleParams(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("max_iters", C.c_int), ("reynolds_dim", C.c_int),
                ("density", C.c_float), ("accel", C.c_float), ("omega", C.c_float)]


_P = C.POINTER
_lib = None


def build() -> None:
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, capture_output=True)


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("d2q9_oracle.c", "d2q9_oracle.h")]
        if not os.path.exists(LIB_PATH) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs):
            build()
        L = C.CDLL(LIB_PATH)
        fp, ip, dp = _P(C.c_float), _P(C.c_int), _P(C.c_double)
        pp = _P(OracleParams)
        L.oracle_read_params.argtypes = [C.c_char_p, pp, C.c_char_p, C.c_size_t]
        L.oracle_read_obstacles.argtypes = [C.c_char_p, C.c_int, C.c_int, ip, ip, C.c_char_p, C.c_size_t]
        L.oracle_decompose.argtypes = [C.c_int, C.c_int, ip, ip]
        L.oracle_decompose.restype = None
        L.oracle_init_cells.argtypes = [pp, fp, C.c_int]
        L.oracle_init_cells.restype = None
        L.oracle_accelerate_row.argtypes = [pp, fp, ip]
        L.oracle_accelerate_row.restype = None
        L.oracle_timestep_rows.argtypes = [pp, fp, fp, ip, C.c_int, C.c_int, dp]
        L.oracle_timestep_rows.restype = C.c_float
        L.oracle_av_velocity_sum.argtypes = [pp, fp, ip, C.c_int]
        L.oracle_av_velocity_sum.restype = C.c_float
        L.oracle_reynolds.argtypes = [pp, C.c_float]
        L.oracle_reynolds.restype = C.c_float
        L.oracle_run.argtypes = [pp, ip, C.c_int, C.c_int, C.c_int, fp, fp, dp]
        L.oracle_run_fast.argtypes = [pp, ip, C.c_int, C.c_int, C.c_int, fp, fp]
        L.oracle_write_final_state.argtypes = [C.c_char_p, pp, fp, ip, C.c_int, C.c_int, C.c_int]
        L.oracle_write_av_vels.argtypes = [C.c_char_p, fp, C.c_int]
        _lib = L
    return _lib


def cparams(p) -> OracleParams:
    """p: anything with nx, ny, max_iters, reynolds_dim, density, accel, omega."""
    return OracleParams(p.nx, p.ny, p.max_iters, p.reynolds_dim, p.density, p.accel, p.omega)


def _f(a):
    return a.ctypes.data_as(_P(C.c_float))


def _i(a):
    return a.ctypes.data_as(_P(C.c_int))


def _d(a):
    return a.ctypes.data_as(_P(C.c_double))


def read_params(path: str) -> OracleParams:
    p = OracleParams()
    err = C.create_string_buffer(1200)
    if lib().oracle_read_params(os.fsencode(path), C.byref(p), err, 1200):
        raise RuntimeError(err.value.decode())
    return p


def read_obstacles(path: str, nx: int, ny: int):
    obst = np.zeros((ny, nx), np.int32)
    free = C.c_int(0)
    err = C.create_string_buffer(1200)
    if lib().oracle_read_obstacles(os.fsencode(path), nx, ny, _i(obst), C.byref(free), err, 1200):
        raise RuntimeError(err.value.decode())
    return obst, free.value


def decompose(ny: int, size: int):
    a, b = (C.c_int * size)(), (C.c_int * size)()
    lib().oracle_decompose(ny, size, a, b)
    return list(a), list(b)


def run(p, obstacles: np.ndarray, n_steps: int, nthreads: int = 1, exact: bool = True):
    """Whole run (d2q9-bgk.c:315-396, one rank).  Returns cells (ny,nx,9) f32, av_vels f32 in the
    reference's summation order, av_exact f64 (double-accumulated yardstick) or None."""
    cp = cparams(p)
    obstacles = np.ascontiguousarray(obstacles, np.int32)
    free = int(obstacles.size - np.count_nonzero(obstacles))
    cells = np.empty((p.ny, p.nx, Q), np.float32)
    av = np.zeros(max(n_steps, 1), np.float32)
    ex = np.zeros(max(n_steps, 1), np.float64) if exact else None
    rc = lib().oracle_run(C.byref(cp), _i(obstacles), free, n_steps, nthreads, _f(cells), _f(av), _d(ex) if exact else None)
    assert rc == 0
    return cells, av[:n_steps], (ex[:n_steps] if exact else None)


def run_fast(p, obstacles: np.ndarray, n_steps: int, nthreads: int = 1):
    """The timed CPU-baseline form (per-thread float accumulators)."""
    cp = cparams(p)
    obstacles = np.ascontiguousarray(obstacles, np.int32)
    free = int(obstacles.size - np.count_nonzero(obstacles))
    cells = np.empty((p.ny, p.nx, Q), np.float32)
    av = np.zeros(max(n_steps, 1), np.float32)
    rc = lib().oracle_run_fast(C.byref(cp), _i(obstacles), free, n_steps, nthreads, _f(cells), _f(av))
    assert rc == 0
    return cells, av[:n_steps]


def av_velocity_sum(p, cells: np.ndarray, obstacles: np.ndarray) -> float:
    cp = cparams(p)
    cells = np.ascontiguousarray(cells, np.float32)
    obstacles = np.ascontiguousarray(obstacles, np.int32)
    return float(lib().oracle_av_velocity_sum(C.byref(cp), _f(cells), _i(obstacles), obstacles.shape[0]))


def reynolds(p, av_velocity: float) -> float:
    cp = cparams(p)
    return float(lib().oracle_reynolds(C.byref(cp), C.c_float(av_velocity)))


def write_final_state(path: str, p, cells: np.ndarray, obstacles: np.ndarray, displ: int = 0, append: bool = False) -> None:
    cp = cparams(p)
    cells = np.ascontiguousarray(cells, np.float32)
    obstacles = np.ascontiguousarray(obstacles, np.int32)
    assert lib().oracle_write_final_state(os.fsencode(path), C.byref(cp), _f(cells), _i(obstacles), obstacles.shape[0], displ, int(append)) == 0


def write_av_vels(path: str, av: np.ndarray) -> None:
    av = np.ascontiguousarray(av, np.float32)
    assert lib().oracle_write_av_vels(os.fsencode(path), _f(av), av.size) == 0


class OraclePartition:
    """One rank of the reference (halo'd AoS rows, d2q9-bgk.c:865-877) stepped by the restatement,
    exposing the same split-phase interface as the HIP Partition so the distributed host logic can
    be exercised on CPU (gloo) with this as the stand-in device.  Halo messages here are whole AoS
    rows, as the reference sends them (:295-313)."""

    def __init__(self, p, free_cells: int, obstacles_rows: np.ndarray, y0: int, is_last: bool):
        import torch
        self.p, self.cp = p, cparams(p)
        self.nyl, self.nx, self.y0, self.is_last = obstacles_rows.shape[0], p.nx, y0, is_last
        self.cells = np.zeros((self.nyl + 2, self.nx, Q), np.float32)
        self.tmp = np.zeros_like(self.cells)
        self.obst = np.zeros((self.nyl + 2, self.nx), np.int32)
        self.obst[1:-1] = obstacles_rows
        lib().oracle_init_cells(C.byref(self.cp), _f(self.cells[1:]), self.nyl)
        n = self.nx * Q
        self._send = (torch.zeros(n), torch.zeros(n))
        self._recv = (torch.zeros(n), torch.zeros(n))
        self.sums = []

    def halo_send(self, d):
        return self._send[d]

    def halo_recv(self, d):
        return self._recv[d]

    def _fill_send(self):
        import torch
        self._send[0].copy_(torch.from_numpy(self.cells[1].reshape(-1)))          # first owned row -> south (`top`)
        self._send[1].copy_(torch.from_numpy(self.cells[self.nyl].reshape(-1)))   # last owned row -> north (`bottom`)

    def step_prepare(self, n_steps, stream=None):
        self.sums = []
        self._fill_send()

    def step_interior(self, stream=None):
        if self.is_last:   # accelerate_flow on local row ny_local-1 (:345-348, :449)
            lib().oracle_accelerate_row(C.byref(self.cp), _f(self.cells[self.nyl - 1]), _i(self.obst[self.nyl - 1]))
        self._terms = np.zeros((self.nyl + 2, self.nx), np.float64)
        if self.nyl > 2:
            lib().oracle_timestep_rows(C.byref(self.cp), _f(self.cells), _f(self.tmp), _i(self.obst), 2, self.nyl, _d(self._terms))

    def step_boundary(self, stream=None):
        self.cells[0] = self._recv[0].numpy().reshape(self.nx, Q)                 # halo row 0 <- south neighbour's last row
        self.cells[self.nyl + 1] = self._recv[1].numpy().reshape(self.nx, Q)      # halo row n+1 <- north neighbour's first row
        lib().oracle_timestep_rows(C.byref(self.cp), _f(self.cells), _f(self.tmp), _i(self.obst), 1, 2, _d(self._terms))
        if self.nyl > 1:
            lib().oracle_timestep_rows(C.byref(self.cp), _f(self.cells), _f(self.tmp), _i(self.obst), self.nyl, self.nyl + 1, _d(self._terms))

    def step_finish(self, stream=None):
        self.sums.append(float(self._terms[1:-1].sum(axis=1).sum()))
        self.cells, self.tmp = self.tmp, self.cells
        self._fill_send()

    def step_collect(self, n_steps, stream=None):
        return np.asarray(self.sums[:n_steps], np.float64)

    def get_cells(self):
        return self.cells[1:-1].copy()


def run_from(p, obstacles: np.ndarray, cells0: np.ndarray, n_steps: int):
    """d2q9-bgk.c:315-394 for one rank starting from an arbitrary state (small grids: Python loop
    over steps, C restatement for the rows).  Returns cells, av_exact (float64 per step)."""
    cp = cparams(p)
    ny, nx = p.ny, p.nx
    free = int(obstacles.size - np.count_nonzero(obstacles))
    inv = np.float32(1.0) / np.float32(free)
    cells = np.zeros((ny + 2, nx, Q), np.float32)
    cells[1:-1] = cells0
    tmp = np.zeros_like(cells)
    obst = np.zeros((ny + 2, nx), np.int32)
    obst[1:-1] = obstacles
    terms = np.zeros((ny + 2, nx), np.float64)
    out = []
    for _ in range(n_steps):
        cells[ny + 1] = cells[1]          # self exchange (:245-247, :295-303)
        cells[0] = cells[ny]
        lib().oracle_accelerate_row(C.byref(cp), _f(cells[ny - 1]), _i(obst[ny - 1]))
        lib().oracle_timestep_rows(C.byref(cp), _f(cells), _f(tmp), _i(obst), 1, ny + 1, _d(terms))
        out.append(terms[1:-1].sum(axis=1).sum() * np.float64(inv))
        cells, tmp = tmp, cells
    return cells[1:-1].copy(), np.asarray(out)
