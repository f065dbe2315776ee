"""Host side of the timestep path above the C ABI — the Python mirror of the reference's `main`.

Names follow the reference (`d2q9-bgk.c`): a *partition* is what one MPI rank owns there (a block
of consecutive rows, `:834-862`); the step loop is `:315-394`; the end-of-run reduction is `:396`.

  Partition          one lbm_ctx: the HIP state of one partition on one GPU
  HaloExchange       the per-step neighbour exchange (`:295-313,326-327,364`) over torch.distributed
                     (backend "nccl" = RCCL over xGMI on MI355X; "gloo" in the CPU tests)
  run_partitioned    the step loop for one rank of a row-partitioned run, backend-agnostic
  Simulation         paramfile + obstaclefile in, av_vels / final state / Reynolds number out

PyTorch is used for plumbing only (device buffers handed to RCCL, streams, process groups).  All
arithmetic of the path happens in liblbm_d2q9.so; there is no CPU fallback in this package.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Protocol, Sequence

import numpy as np

from . import _capi
from ._capi import CParams, LbmError, check
from .decks import Params

SOUTH, NORTH = 0, 1   # dir 0: towards row y-1 (reference `top`, rank-1); dir 1: towards y+1 (`bottom`)


# ------------------------------------------------------------------------------------------------
# host-only helpers (no GPU): parsers, decomposition, epilogue
# ------------------------------------------------------------------------------------------------

def _cparams(p: Params) -> CParams:
    return CParams(p.nx, p.ny, p.max_iters, p.reynolds_dim, p.density, p.accel, p.omega)


def read_params(paramfile: str) -> Params:
    """`initialise()`'s parameter-file half (`d2q9-bgk.c:772-803`); LbmError carries the die() text."""
    lib = _capi.load_library()
    cp = CParams()
    check(lib.lbm_read_params(os.fsencode(paramfile), C.byref(cp)))
    # float fields are C floats; keep their exact values
    return Params(cp.nx, cp.ny, cp.max_iters, cp.reynolds_dim, float(cp.density), float(cp.accel), float(cp.omega))


def read_obstacles(obstaclefile: str, nx: int, ny: int) -> tuple[np.ndarray, int]:
    """`initialise()`'s obstacle-file half (`d2q9-bgk.c:917-953`) -> ((ny, nx) int32 map, free_cells)."""
    lib = _capi.load_library()
    obst = np.zeros((ny, nx), dtype=np.int32)
    free = C.c_int(0)
    check(lib.lbm_read_obstacles(os.fsencode(obstaclefile), nx, ny, _capi.as_int_ptr(obst), C.byref(free)))
    return obst, free.value


def count_free_cells(obstacles: np.ndarray) -> int:
    """`numOfFreeCells` (`d2q9-bgk.c:805,945-946`) for an in-memory map."""
    return int(obstacles.size - np.count_nonzero(obstacles))


def decompose(ny: int, size: int) -> tuple[list[int], list[int]]:
    """Row decomposition of `d2q9-bgk.c:834-862` -> (ny_local[size], displs[size])."""
    lib = _capi.load_library()
    nyl = (C.c_int * size)()
    dis = (C.c_int * size)()
    check(lib.lbm_decompose(ny, size, nyl, dis))
    return list(nyl), list(dis)


def plan_steps(K: int, n_steps: int, four_rows: bool = True) -> list[int]:
    """`lbm_plan_steps`: the launches (macro-steps) a run of n_steps is cut into — K at a time, 4s and 3s at the end."""
    lib = _capi.load_library()
    n = lib.lbm_plan_steps(K, 1 if four_rows else 0, n_steps, None, 0)
    if n < 0:
        raise LbmError(lib.lbm_last_error().decode(errors="replace"))
    out = (C.c_int * max(n, 1))()
    lib.lbm_plan_steps(K, 1 if four_rows else 0, n_steps, out, n)
    return [int(out[i]) for i in range(n)]


def rank_layout(params: Params, nranks: int, rank: int, flags: int = 0) -> dict:
    """`lbm_rank_layout`: rows of `rank` by the reference's rule (`d2q9-bgk.c:834-862`) plus the stepping
    mode of the WHOLE run, derived from global quantities only — the same answer on every rank."""
    lib = _capi.load_library()
    lay = _capi.CLayout()
    cp = _cparams(params)
    check(lib.lbm_rank_layout(C.byref(cp), nranks, rank, flags, C.byref(lay)))
    return {"y0": lay.y0, "ny_local": lay.ny_local, "macro_k": lay.macro_k, "ghost": lay.ghost, "group": lay.group}


def plan_groups(K: int, ghost: int, group_max: int, n_steps: int) -> list[list[int]]:
    """`lbm_plan_group` over a whole run: the launches between consecutive halo exchanges of a partitioned run, as lists of steps —
    e.g. K = 4 on 8 ghost rows, 20 steps: [[4, 4], [4, 4], [4]]."""
    lib = _capi.load_library()
    out, left = [], n_steps
    buf = (C.c_int * 8)()
    while left > 0:
        n = lib.lbm_plan_group(K, ghost, group_max, left, buf, 8)
        if n <= 0:
            raise LbmError(lib.lbm_last_error().decode(errors="replace") if n < 0 else "lbm_plan_group: empty group")
        out.append([int(buf[i]) for i in range(n)])
        left -= sum(out[-1])
    return out


def tile_layout(params: Params, px: int, py: int, rank: int, flags: int = 0) -> dict:
    """`lbm_tile_layout_of`: the block of `rank` = ry * px + rx in a px x py tile (2-D) decomposition — rows by the reference's
    rule over py, columns in whole x-pairs over px — and the K-step layout of the whole run (ghost rows, ghost columns)."""
    lib = _capi.load_library()
    lay = _capi.CTileLayout()
    cp = _cparams(params)
    check(lib.lbm_tile_layout_of(C.byref(cp), px, py, rank, flags, C.byref(lay)))
    return {name: int(getattr(lay, name)) for name, _ in _capi.CTileLayout._fields_}


def choose_rank_grid(params: Params, nranks: int, flags: int = 0) -> Optional[tuple[int, int]]:
    """`lbm_choose_rank_grid`: None for the reference's row blocks, or (px, py) for the tile decomposition — whichever leaves the ranks
    the fewest cells to recompute (grids much wider than tall come out as tiles)."""
    lib = _capi.load_library()
    cp = _cparams(params)
    px, py = C.c_int(0), C.c_int(0)
    check(lib.lbm_choose_rank_grid(C.byref(cp), nranks, flags, C.byref(px), C.byref(py)))
    return None if px.value == 1 else (px.value, py.value)


def obstacle_window(obstacles: np.ndarray, layout: dict) -> np.ndarray:
    """The part of the global map one rank needs: its owned rows plus `ghost` rows below and above,
    wrapping periodically — what the root hands each rank instead of the reference's `MPI_Scatterv` of
    the owned rows alone (`d2q9-bgk.c:968-970`).  A rank of the tile decomposition (`tile_layout`) gets
    its columns plus `ghost_x` on each side of those rows."""
    ny, nx = obstacles.shape
    gy = layout.get("ghost_y", layout["ghost"])          # (column blocks of the tile decomposition keep no ghost rows)
    rows = np.arange(layout["y0"] - gy, layout["y0"] + layout["ny_local"] + gy) % ny
    if "ghost_x" not in layout:
        return np.ascontiguousarray(obstacles[rows], dtype=np.int32)
    cols = np.arange(layout["x0"] - layout["ghost_x"], layout["x0"] + layout["nx_local"] + layout["ghost_x"]) % nx
    return np.ascontiguousarray(obstacles[np.ix_(rows, cols)], dtype=np.int32)


def av_velocity_host(params: Params, cells: np.ndarray, obstacles: np.ndarray) -> float:
    """`av_velocity()` without the reduce (`d2q9-bgk.c:716-751`): float tot_u over the given rows."""
    lib = _capi.load_library()
    cells = np.ascontiguousarray(cells, dtype=np.float32)
    obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
    cp = _cparams(params)
    return float(lib.lbm_av_velocity_host(C.byref(cp), _capi.as_float_ptr(cells), _capi.as_int_ptr(obstacles),
                                          obstacles.shape[0]))


def av_velocity_obs(params: Params, obs: np.ndarray, obstacles: np.ndarray) -> float:
    """The same value from device-computed observables (`lbm_get_observables`): u_x, u_y are the floats of
    `d2q9-bgk.c:732-746`; the double sqrt and the float accumulation (`:748`) happen here."""
    lib = _capi.load_library()
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
    cp = _cparams(params)
    return float(lib.lbm_av_velocity_obs(C.byref(cp), _capi.as_float_ptr(obs), _capi.as_int_ptr(obstacles), obstacles.shape[0]))


def reynolds(params: Params, av_velocity: float) -> float:
    """`calc_reynolds()` (`d2q9-bgk.c:1005-1007`)."""
    lib = _capi.load_library()
    cp = _cparams(params)
    return float(lib.lbm_reynolds(C.byref(cp), C.c_float(av_velocity)))


def write_final_state(path: str, params: Params, cells: np.ndarray, obstacles: np.ndarray, displ: int = 0,
                      append: bool = False) -> None:
    """`write_values()`'s final_state.dat part (`d2q9-bgk.c:1054-1120`)."""
    lib = _capi.load_library()
    cells = np.ascontiguousarray(cells, dtype=np.float32)
    obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
    cp = _cparams(params)
    check(lib.lbm_write_final_state(os.fsencode(path), C.byref(cp), _capi.as_float_ptr(cells),
                                    _capi.as_int_ptr(obstacles), obstacles.shape[0], displ, int(append)))


def write_final_state_obs(path: str, params: Params, obs: np.ndarray, obstacles: np.ndarray, displ: int = 0,
                          append: bool = False) -> None:
    """The same file from `lbm_get_observables` output (4 floats per cell computed on the device)."""
    lib = _capi.load_library()
    obs = np.ascontiguousarray(obs, dtype=np.float32)
    obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
    cp = _cparams(params)
    check(lib.lbm_write_final_state_obs(os.fsencode(path), C.byref(cp), _capi.as_float_ptr(obs),
                                        _capi.as_int_ptr(obstacles), obstacles.shape[0], displ, int(append)))


def write_av_vels(path: str, av_vels: np.ndarray) -> None:
    """`write_values()`'s av_vels.dat part (`d2q9-bgk.c:1127-1139`)."""
    lib = _capi.load_library()
    av = np.ascontiguousarray(av_vels, dtype=np.float32)
    check(lib.lbm_write_av_vels(os.fsencode(path), _capi.as_float_ptr(av), av.size))


# ------------------------------------------------------------------------------------------------
# one partition on one GPU
# ------------------------------------------------------------------------------------------------

class Partition:
    """Device state of rows [y0, y0+ny_local) — one `lbm_ctx` (include/lbm_d2q9.h)."""

    def __init__(self, params: Params, free_cells: int, obstacles_rows: np.ndarray, y0: int = 0,
                 device: int = 0, flags: int = 0, obstacles_global: Optional[np.ndarray] = None,
                 rank_of: Optional[tuple[int, int]] = None, tile_of: Optional[tuple[int, int, int]] = None):
        """obstacles_global: the whole (ny, nx) map; when given (`lbm_create_global`) an eligible
        row partition runs in K-step mode (`macro_steps` > 0) and obstacles_rows is ignored.
        rank_of = (rank, nranks): obstacles_rows is this rank's obstacle WINDOW (`obstacle_window`) and the
        context comes from `lbm_create_rank`, whose stepping mode is the same on every rank of the run.
        tile_of = (rank, px, py): a rank of the tile decomposition (`lbm_create_tile`); obstacles_rows is its
        window of `tile_layout` (rows and columns with their ghosts)."""
        self._lib = _capi.load_library()
        obst = np.ascontiguousarray(obstacles_rows, dtype=np.int32)
        self.x0, self.nx_local = 0, params.nx
        if tile_of is not None:
            rank, px, py = tile_of
            lay = tile_layout(params, px, py, rank, flags)
            if obst.shape != (lay["ny_local"] + 2 * lay["ghost_y"], lay["nx_local"] + 2 * lay["ghost_x"]):
                raise ValueError("obstacles_rows must be the rank's window: (ny_local + 2*ghost_y, nx_local + 2*ghost_x)")
            self.params, self.free_cells, self.device = params, free_cells, device
            self.y0, self.ny_local, self.x0, self.nx_local = lay["y0"], lay["ny_local"], lay["x0"], lay["nx_local"]
            self.free_cells_inv = np.float32(1.0) / np.float32(free_cells)
            self._ctx = C.c_void_p()
            cp = _cparams(params)
            check(self._lib.lbm_create_tile(C.byref(self._ctx), C.byref(cp), free_cells, _capi.as_int_ptr(obst), px, py, rank, device, flags))
            self._halo_tensors = None
            return
        if obst.ndim != 2 or obst.shape[1] != params.nx:
            raise ValueError("obstacles_rows must be (ny_local, nx)")
        self.params, self.free_cells, self.y0, self.ny_local, self.device = params, free_cells, y0, obst.shape[0], device
        self.free_cells_inv = np.float32(1.0) / np.float32(free_cells)          # d2q9-bgk.c:950
        self._ctx = C.c_void_p()
        cp = _cparams(params)
        if rank_of is not None:
            rank, nranks = rank_of
            lay = rank_layout(params, nranks, rank, flags)
            if obst.shape[0] != lay["ny_local"] + 2 * lay["ghost"]:
                raise ValueError("obstacles_rows must be the rank's window: ny_local + 2*ghost rows")
            self.y0, self.ny_local = lay["y0"], lay["ny_local"]
            check(self._lib.lbm_create_rank(C.byref(self._ctx), C.byref(cp), free_cells, _capi.as_int_ptr(obst), nranks, rank,
                                            device, flags))
        elif obstacles_global is not None:
            glob = np.ascontiguousarray(obstacles_global, dtype=np.int32)
            if glob.shape != (params.ny, params.nx):
                raise ValueError("obstacles_global must be (ny, nx)")
            check(self._lib.lbm_create_global(C.byref(self._ctx), C.byref(cp), free_cells, _capi.as_int_ptr(glob), y0,
                                              self.ny_local, device, flags))
        else:
            check(self._lib.lbm_create(C.byref(self._ctx), C.byref(cp), free_cells, _capi.as_int_ptr(obst), y0,
                                       self.ny_local, device, flags))
        self._halo_tensors = None

    @property
    def macro_steps(self) -> int:
        """K if the partition runs in K-step mode (lbm_macro_*), else 0."""
        return int(self._lib.lbm_macro_steps(self._ctx))

    @property
    def macro_next(self) -> int:
        """Steps of the macro-step about to be made (`lbm_macro_next_steps`): K, fewer at the end of a run, or 3s and
        4s where the partition keeps four ghost rows at K = 3.  0 when no run is in progress."""
        return int(self._lib.lbm_macro_next_steps(self._ctx))

    @property
    def macro_launches(self) -> int:
        """Launches of the macro-step about to be made (`lbm_macro_next_launches`): one exchange, then that many launches."""
        return int(self._lib.lbm_macro_next_launches(self._ctx))

    def macro_prepare(self, n_steps: int, stream=None) -> None:
        check(self._lib.lbm_macro_prepare(self._ctx, n_steps, self._stream_ptr(stream)))

    def macro_interior(self, stream=None) -> None:
        check(self._lib.lbm_macro_interior(self._ctx, self._stream_ptr(stream)))

    def macro_edge(self, stream=None) -> None:
        check(self._lib.lbm_macro_edge(self._ctx, self._stream_ptr(stream)))

    def macro_finish(self, stream=None) -> None:
        check(self._lib.lbm_macro_finish(self._ctx, self._stream_ptr(stream)))

    def macro_receive_from(self, src: "Partition", direction: int, stream=None) -> None:
        """In-process exchange: src's rows travelling in `direction` become this partition's ghost rows."""
        check(self._lib.lbm_macro_exchange_local(self._ctx, src._ctx, direction, self._stream_ptr(stream)))

    # -- lifetime --
    def close(self) -> None:
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.lbm_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- whole-domain run (ny_local == ny) --
    def run(self, n_steps: int) -> np.ndarray:
        """`d2q9-bgk.c:315-394` x n_steps on a self-contained domain; returns av_vels (float32)."""
        av = np.zeros(max(n_steps, 1), dtype=np.float32)
        check(self._lib.lbm_run(self._ctx, n_steps, _capi.as_float_ptr(av)))
        return av[:n_steps]

    # -- state --
    def tile_info(self) -> dict:
        """`lbm_tile_info`: the block this context owns and its K-step layout."""
        lay = _capi.CTileLayout()
        check(self._lib.lbm_tile_info(self._ctx, C.byref(lay)))
        return {name: int(getattr(lay, name)) for name, _ in _capi.CTileLayout._fields_}

    def get_cells(self) -> np.ndarray:
        cells = np.empty((self.ny_local, self.nx_local, _capi.NSPEEDS), dtype=np.float32)
        check(self._lib.lbm_get_cells(self._ctx, _capi.as_float_ptr(cells)))
        return cells

    def set_cells(self, cells: np.ndarray) -> None:
        cells = np.ascontiguousarray(cells, dtype=np.float32)
        if cells.shape != (self.ny_local, self.nx_local, _capi.NSPEEDS):
            raise ValueError("cells must be (ny_local, nx_local, 9)")
        check(self._lib.lbm_set_cells(self._ctx, _capi.as_float_ptr(cells)))

    def get_observables(self) -> np.ndarray:
        """(ny_local, nx, 4) float32: u_x, u_y, u, pressure per cell, computed on the device (`d2q9-bgk.c:1084-1111`)."""
        obs = np.empty((self.ny_local, self.nx_local, 4), dtype=np.float32)
        check(self._lib.lbm_get_observables(self._ctx, _capi.as_float_ptr(obs)))
        return obs

    def checksum(self, y_begin: Optional[int] = None, y_end: Optional[int] = None) -> int:
        """`lbm_state_checksum` of the global rows [y_begin, y_end) (default: all owned rows): additive over
        disjoint row ranges, so the ranks' digests sum (mod 2**64) to the whole grid's."""
        y_begin = self.y0 if y_begin is None else y_begin
        y_end = self.y0 + self.ny_local if y_end is None else y_end
        d = C.c_ulonglong(0)
        check(self._lib.lbm_state_checksum(self._ctx, y_begin, y_end, C.byref(d)))
        return int(d.value)

    def av_velocity_sum(self) -> float:
        tot = C.c_double(0.0)
        check(self._lib.lbm_av_velocity_sum(self._ctx, C.byref(tot)))
        return tot.value

    def last_run_kernel_ms(self) -> tuple[float, int]:
        """(device ms from first to last step kernel of the last run, number of step-kernel launches)."""
        ms, n = C.c_double(0.0), C.c_int(0)
        check(self._lib.lbm_last_run_kernel_ms(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_profile(self, on: bool) -> None:
        """`lbm_set_profile`: timing events around every step-kernel launch of the following `run`s."""
        check(self._lib.lbm_set_profile(self._ctx, 1 if on else 0))

    def launch_profile(self) -> list[tuple[int, float]]:
        """[(steps advanced, duration in us)] per step-kernel launch of the last profiled `run`, in order."""
        n = C.c_int(0)
        check(self._lib.lbm_launch_profile(self._ctx, 0, None, None, C.byref(n)))
        steps, us = (C.c_int * max(n.value, 1))(), (C.c_double * max(n.value, 1))()
        check(self._lib.lbm_launch_profile(self._ctx, n.value, steps, us, C.byref(n)))
        return [(int(steps[i]), float(us[i])) for i in range(n.value)]

    def describe(self) -> dict:
        name = C.create_string_buffer(256)
        cells, nbytes = C.c_longlong(0), C.c_longlong(0)
        check(self._lib.lbm_describe(self._ctx, name, 256, C.byref(cells), C.byref(nbytes)))
        return {"kernel": name.value.decode(), "cells_per_launch": cells.value, "state_bytes": nbytes.value}

    # -- split-phase stepping (row-partitioned runs) --
    @property
    def halo_floats(self) -> int:
        return int(self._lib.lbm_halo_floats(self._ctx))

    def bind_halo_tensors(self, torch_device) -> None:
        """Allocate the four halo messages as torch tensors (so the communicator can use them) and
        hand their device pointers to the library."""
        import torch
        n = self.halo_floats
        t = [torch.zeros(n, dtype=torch.float32, device=torch_device) for _ in range(4)]
        check(self._lib.lbm_bind_halo_buffers(self._ctx, *(C.c_void_p(x.data_ptr()) for x in t)))
        self._halo_tensors = {"send": (t[0], t[1]), "recv": (t[2], t[3])}

    def halo_send(self, direction: int):
        return self._halo_tensors["send"][direction]

    def halo_recv(self, direction: int):
        return self._halo_tensors["recv"][direction]

    @staticmethod
    def _stream_ptr(stream) -> C.c_void_p:
        """stream: None = the context's own stream; otherwise a hipStream_t handle as an int
        (e.g. torch.cuda.Stream().cuda_stream).  The HIP null stream (handle 0) cannot be named
        through the ABI — use an explicit stream when kernels must be ordered with other work."""
        if stream is None:
            return C.c_void_p(0)
        if not isinstance(stream, int) or stream == 0:
            raise ValueError("stream must be None or a non-null hipStream_t handle (create a torch.cuda.Stream)")
        return C.c_void_p(stream)

    def step_prepare(self, n_steps: int, stream=None) -> None:
        check(self._lib.lbm_step_prepare(self._ctx, n_steps, self._stream_ptr(stream)))

    def step_interior(self, stream=None) -> None:
        check(self._lib.lbm_step_interior(self._ctx, self._stream_ptr(stream)))

    def step_boundary(self, stream=None) -> None:
        check(self._lib.lbm_step_boundary(self._ctx, self._stream_ptr(stream)))

    def step_finish(self, stream=None) -> None:
        check(self._lib.lbm_step_finish(self._ctx, self._stream_ptr(stream)))

    def step_collect(self, n_steps: int, stream=None) -> np.ndarray:
        out = np.zeros(max(n_steps, 1), dtype=np.float64)
        check(self._lib.lbm_step_collect(self._ctx, self._stream_ptr(stream), _capi.as_double_ptr(out), n_steps))
        return out[:n_steps]


class PartitionBackend(Protocol):
    """What run_partitioned needs from one rank's state (Partition implements it on the GPU)."""

    def halo_send(self, direction: int): ...
    def halo_recv(self, direction: int): ...
    def step_prepare(self, n_steps: int, stream=None) -> None: ...
    def step_interior(self, stream=None) -> None: ...
    def step_boundary(self, stream=None) -> None: ...
    def step_finish(self, stream=None) -> None: ...
    def step_collect(self, n_steps: int, stream=None) -> np.ndarray: ...


# ------------------------------------------------------------------------------------------------
# neighbour exchange + partitioned step loop
# ------------------------------------------------------------------------------------------------

class HaloExchange:
    """Periodic ring of ranks (`d2q9-bgk.c:244-247`): each step every rank sends its two edge rows'
    outgoing populations and receives its two halo messages (`:295-313,326-327`), non-blocking, as
    one batch (a single RCCL group on the nccl backend, over the direct xGMI link per neighbour)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.south = (self.rank - 1) % self.size      # reference `top`    (:245-246)
        self.north = (self.rank + 1) % self.size      # reference `bottom` (:247)
        # gloo cannot move device memory: device halo buffers are then staged through host copies
        # (testing aid — lets several ranks share one GPU, which RCCL refuses)
        self.host_staged = dist.get_backend(group) == "gloo"
        self._staging = None

    def _global(self, group_rank: int) -> int:
        return self._dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def start(self, part: PartitionBackend):
        """Post the four transfers (`MPI_Startall`, `:327`).  Send order [south, north] pairs with
        receive order [north, south] exactly as the reference's request arrays (`:295-303`), which
        is what keeps the two messages apart when both neighbours are the same rank (size 2)."""
        d = self._dist
        send_s, send_n = part.halo_send(SOUTH), part.halo_send(NORTH)
        recv_s, recv_n = part.halo_recv(SOUTH), part.halo_recv(NORTH)
        self._staging = None
        if self.host_staged and send_s.is_cuda:
            self._staging = (recv_s, recv_n, recv_s.cpu(), recv_n.cpu())
            send_s, send_n = send_s.cpu(), send_n.cpu()          # synchronising device-to-host copies
            recv_s, recv_n = self._staging[2], self._staging[3]
        ops = [
            d.P2POp(d.isend, send_s, self._global(self.south), self.group, tag=0),
            d.P2POp(d.isend, send_n, self._global(self.north), self.group, tag=1),
            d.P2POp(d.irecv, recv_n, self._global(self.north), self.group, tag=0),
            d.P2POp(d.irecv, recv_s, self._global(self.south), self.group, tag=1),
        ]
        return d.batch_isend_irecv(ops)

    def wait(self, requests) -> None:
        """`MPI_Waitall` (`:364`).  On nccl this only makes the current stream wait; the host goes on."""
        for r in requests:
            r.wait()
        if self._staging is not None:
            dev_s, dev_n, host_s, host_n = self._staging
            dev_s.copy_(host_s)
            dev_n.copy_(host_n)
            self._staging = None

    def allreduce_sum(self, values: np.ndarray, torch_device) -> np.ndarray:
        """The end-of-run `MPI_Reduce(..., MPI_SUM, ...)` (`:396`), here as an all-reduce."""
        import torch
        t = torch.from_numpy(np.ascontiguousarray(values))
        if not self.host_staged:
            t = t.to(torch_device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()


class RcclRing:
    """The native step loop of liblbm_d2q9_rccl.so for one rank: RCCL send/recv on a side HIP stream
    overlapped with the interior kernel, one all-reduce at the end (`d2q9-bgk.c:295-313,326-327,364,396`).
    torch.distributed is used once, to hand rank 0's 128-byte RCCL id to every rank."""

    def __init__(self, partition: "Partition", group=None, *, rank: Optional[int] = None, size: Optional[int] = None,
                 step_allreduce: bool = False):
        self.partition = partition
        self._comm = C.c_void_p()
        if size is None:
            import torch
            import torch.distributed as dist
            rank, size = dist.get_rank(group), dist.get_world_size(group)
            # Two collectives follow (the id broadcast and ncclCommInitRank).  They may only be entered when
            # EVERY rank can enter them, or the ranks that can would block in them for ever: agree first.
            lib, ident, err = None, C.create_string_buffer(_capi.COMM_ID_BYTES), None
            try:
                lib = _capi.load_rccl_library()
                if rank == 0:
                    check(lib.lbm_comm_unique_id(ident))
            except (LbmError, RuntimeError, OSError) as e:       # library missing on this node, RCCL error
                err = e
            on_host = dist.get_backend(group) == "gloo"
            ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32,
                              device="cpu" if on_host else torch.device("cuda", partition.device))
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
            if int(ok.item()) != 1:
                raise LbmError(f"native RCCL loop unavailable on at least one rank ({err or 'this rank is fine'})")
            self._lib = lib
            box = [ident.raw if rank == 0 else None]
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            ident = C.create_string_buffer(box[0], _capi.COMM_ID_BYTES)
        else:                                   # single process (size must be 1): exchange with itself
            self._lib = _capi.load_rccl_library()
            ident = C.create_string_buffer(_capi.COMM_ID_BYTES)
            check(self._lib.lbm_comm_unique_id(ident))
        self.rank, self.size = rank, size
        check(self._lib.lbm_comm_create(C.byref(self._comm), partition._ctx, ident, size, rank))
        if step_allreduce:
            check(self._lib.lbm_comm_set_step_allreduce(self._comm, 1))
        self.step_allreduce = step_allreduce

    @property
    def nranks(self) -> int:
        """Ranks of the communicator as RCCL reports it (`ncclCommCount`)."""
        return int(self._lib.lbm_comm_nranks(self._comm))

    def run(self, n_steps: int) -> np.ndarray:
        """Global per-step tot_u (float64, n_steps), identical on every rank."""
        out = np.zeros(max(n_steps, 1), dtype=np.float64)
        check(self._lib.lbm_comm_run(self._comm, n_steps, _capi.as_double_ptr(out)))
        return out[:n_steps]

    def close(self) -> None:
        if getattr(self, "_comm", None) is not None and self._comm:
            self._lib.lbm_comm_destroy(self._comm)
            self._comm = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class P2PRing:
    """The native step loop with direct peer-to-peer halo stores (include/lbm_d2q9_p2p.h) for one rank:
    every K steps a kernel stores this rank's edge rows straight into the neighbours' ghost rows over
    xGMI and raises an epoch flag there; no communication library on the path
    (`d2q9-bgk.c:295-313,326-327,364,396`).  Set-up is `create` -> exchange of handles -> `connect`:
      * across processes, `P2PRing(partition, group)` all-gathers the handles over torch.distributed;
      * inside one process, `P2PRing.local_ring(partitions)` passes them in memory (ranks then run on one
        host thread each, `run_all`)."""

    def __init__(self, partition: "Partition", group=None, *, rank: Optional[int] = None, size: Optional[int] = None,
                 connect: bool = True):
        self._lib = _capi.load_library()
        self.partition = partition
        self._t = C.c_void_p()
        if size is None:
            import torch.distributed as dist
            rank, size = dist.get_rank(group), dist.get_world_size(group)
        self.rank, self.size = rank, size
        self._group = group
        self._connected_over_group = False
        # every rank reaches the all-gather below whether or not its own set-up worked: a failure is
        # carried in place of the handle and raised on all ranks together
        blob, err = None, None
        try:
            check(self._lib.lbm_p2p_create(C.byref(self._t), partition._ctx, size, rank))
            buf = C.create_string_buffer(_capi.P2P_HANDLE_BYTES)
            check(self._lib.lbm_p2p_handle(self._t, buf))
            blob = buf.raw
        except LbmError as e:
            err = str(e)
        self.handle = blob
        self._err = err
        if connect:
            import torch.distributed as dist
            box = [None] * size
            dist.all_gather_object(box, blob if err is None else ("error", err), group=group)
            bad = [(r, b[1]) for r, b in enumerate(box) if isinstance(b, tuple)]
            if bad:
                self.close()
                raise LbmError("peer-to-peer set-up failed on rank(s) " + "; ".join(f"{r}: {m}" for r, m in bad))
            # Mapping the peers (hipIpcOpenMemHandle, hipDeviceEnablePeerAccess) can fail on ONE rank only.  The
            # ranks therefore agree on the outcome with a second all-gather — which is also the meeting point
            # that keeps anybody from pushing rows into a neighbour that has not mapped its peers yet — and on
            # any failure every rank unmaps, meets again (memory another process has mapped must not be freed
            # before that process has unmapped it) and raises the same error.
            err = None
            try:
                self.connect(box)
            except LbmError as e:
                err = str(e)
            outcome = [None] * size
            dist.all_gather_object(outcome, err, group=group)
            self._connected_over_group = size > 1
            bad = [(r, m) for r, m in enumerate(outcome) if m is not None]
            if bad:
                self.close()
                raise LbmError("peer-to-peer connect failed on rank(s) " + "; ".join(f"{r}: {m}" for r, m in bad))
        elif err is not None:
            raise LbmError(err)

    def connect(self, handles: Sequence[bytes]) -> None:
        check(self._lib.lbm_p2p_connect(self._t, b"".join(handles)))

    @classmethod
    def local_ring(cls, partitions: Sequence["Partition"]) -> list["P2PRing"]:
        """Rings for several partitions of ONE process (same or different GPUs), connected in memory."""
        n = len(partitions)
        rings = [cls(p, rank=r, size=n, connect=False) for r, p in enumerate(partitions)]
        handles = [r.handle for r in rings]
        for r in rings:
            r.connect(handles)
        return rings

    @staticmethod
    def run_all(rings: Sequence["P2PRing"], n_steps: int) -> list[np.ndarray]:
        """`run` of every ring of a process at once, one host thread per rank (each call blocks until its
        rank's device work is done, and the ranks wait for one another's rows)."""
        import threading
        out: list = [None] * len(rings)

        def work(i):
            try:
                out[i] = rings[i].run(n_steps)
            except Exception as e:      # noqa: BLE001 - re-raised below on the calling thread
                out[i] = e
        threads = [threading.Thread(target=work, args=(i,)) for i in range(len(rings))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        errors = [(i, o) for i, o in enumerate(out) if isinstance(o, Exception)]
        if errors:
            # every rank's own words: the ranks that only waited for a failed one say "did not arrive in time", the failed one says why
            first = next((o for _, o in errors if "did not arrive" not in str(o)), errors[0][1])
            raise type(first)("; ".join(f"rank {i}: {o}" for i, o in errors)) from first
        return out

    def describe(self) -> str:
        buf = C.create_string_buffer(256)
        check(self._lib.lbm_p2p_describe(self._t, buf, 256))
        return buf.value.decode()

    def run(self, n_steps: int) -> np.ndarray:
        """Global per-step tot_u (float64, n_steps), bitwise identical on every rank."""
        out = np.zeros(max(n_steps, 1), dtype=np.float64)
        check(self._lib.lbm_p2p_run(self._t, n_steps, _capi.as_double_ptr(out)))
        return out[:n_steps]

    def set_profile(self, on: bool) -> None:
        """`lbm_p2p_set_profile`: HIP timing events around the launches of the following `run`s."""
        check(self._lib.lbm_p2p_set_profile(self._t, 1 if on else 0))

    def phases(self) -> dict:
        """Where the last profiled `run` spent its time (`lbm_p2p_phases`): {name: microseconds} (macro_steps: a count)."""
        v = (C.c_double * _capi.P2P_PHASES)()
        check(self._lib.lbm_p2p_phases(self._t, v))
        out, i = {}, 0
        while True:
            name = self._lib.lbm_p2p_phase_name(i)
            if not name:
                return out
            out[name.decode()] = float(v[i])
            i += 1

    def close(self) -> None:
        """Unmap the peers, meet the other ranks (memory another process has mapped must not be freed before that
        process has unmapped it), then free this rank's window.  The caller frees the partition afterwards."""
        if getattr(self, "_t", None) is None or not self._t:
            return
        self._lib.lbm_p2p_disconnect(self._t)
        if self._connected_over_group:
            try:
                import torch.distributed as dist
                if dist.is_initialized():
                    dist.barrier(group=self._group)
            except Exception:       # noqa: BLE001 - the group is gone (interpreter shutdown, failed peer): nothing to meet
                pass
        self._lib.lbm_p2p_destroy(self._t)
        self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_partitioned(part: PartitionBackend, exchange: HaloExchange, n_steps: int, free_cells_inv: np.float32,
                    torch_device="cpu", stream=None) -> np.ndarray:
    """One rank's share of `d2q9-bgk.c:315-396`.  Returns the global av_vels (float32, n_steps)."""
    part.step_prepare(n_steps, stream)
    for _ in range(n_steps):
        requests = exchange.start(part)          # :326-327  exchange starts ...
        part.step_interior(stream)               # :350      ... and overlaps the rows that need no halo
        exchange.wait(requests)                  # :364
        part.step_boundary(stream)               # :365-366
        part.step_finish(stream)                 # :376-378
    local = part.step_collect(n_steps, stream)   # per-step tot_u of this partition (double)
    total = exchange.allreduce_sum(local, torch_device)                      # :396
    return (total * np.float64(free_cells_inv)).astype(np.float32)           # :367


# ------------------------------------------------------------------------------------------------
# the CLI contract as an object
# ------------------------------------------------------------------------------------------------

EXCHANGES = ("auto", "p2p", "rccl", "torch")


class Simulation:
    """paramfile + obstaclefile -> av_vels, final state, Reynolds number (the reference's `main`).

    With a torch.distributed process group of size > 1 (one process per GPU), each rank owns the rows
    `rank_layout()` gives it and `run()` performs the halo exchange — `loop` tells which way:
      "p2p"    native loop, direct peer-to-peer stores into the neighbours' ghost rows (K-step mode) or halo slots (one-step mode)
      "rccl"   native loop of liblbm_d2q9_rccl.so, RCCL send/recv (K-step or one-step mode)
      "torch"  one-step loop driven from Python over torch.distributed P2P ops
      "single" the whole grid on one GPU, one `lbm_run`.
    `exchange="auto"` = p2p.  With strict=False a native loop
    that cannot be set up on every rank falls back to the next one (with a warning); with strict=True that
    is an error — what ran is always `loop`, never the request."""

    def __init__(self, params: Params, obstacles: Optional[np.ndarray], *, device: int = 0, flags: int = 0,
                 distributed: bool = False, group=None, exchange: str = "auto", strict: bool = False,
                 free_cells: Optional[int] = None, step_allreduce: bool = False, rank_grid: Optional[tuple[int, int]] = None):
        """obstacles: the whole (ny, nx) map — or, in a distributed run, None on every rank but 0: rank 0 then
        hands each rank its window of rows (the reference's `MPI_Scatterv`, `d2q9-bgk.c:968-970`) and the
        free-cell count (`MPI_Bcast`, `:966`).
        rank_grid = (px, py): the tile (2-D) decomposition over px x py = size ranks (`tile_layout`) instead of the
        reference's row blocks — peer-to-peer loop only; (1, 1) is a one-rank ring that exchanges with itself;
        "auto": `choose_rank_grid` decides between row blocks and tiles."""
        if exchange not in EXCHANGES:
            raise ValueError(f"exchange must be one of {EXCHANGES}")
        self.params = params
        self.device = device
        self.exchange: Optional[HaloExchange] = None
        self.rank, self.size = 0, 1
        self._group = group
        if distributed:
            self.exchange = HaloExchange(group)
            self.rank, self.size = self.exchange.rank, self.exchange.size
        if obstacles is not None:
            obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
            if obstacles.shape != (params.ny, params.nx):
                raise ValueError("obstacles must be (ny, nx)")
        elif not distributed or self.rank == 0:
            raise ValueError("obstacles may only be None on ranks other than 0 of a distributed run")
        self.obstacles = obstacles          # whole map where this rank has it (rank 0 always), else None
        self.rank_grid = None
        if isinstance(rank_grid, str):
            if rank_grid != "auto":
                raise ValueError("rank_grid is (px, py), None or \"auto\"")
            rank_grid = choose_rank_grid(params, self.size, flags) if exchange in ("auto", "p2p") and not step_allreduce else None
        if rank_grid is not None:
            px, py = int(rank_grid[0]), int(rank_grid[1])
            if px < 1 or py < 1 or px * py != self.size:
                raise ValueError(f"rank_grid {rank_grid} does not match {self.size} rank(s)")
            if exchange not in ("auto", "p2p") or step_allreduce:
                raise ValueError("the tile decomposition is stepped by the peer-to-peer loop only")
            self.rank_grid = (px, py)
        self._partitioned = self.size > 1 or bool(flags & _capi.FLAG_FORCE_HALO) or self.rank_grid is not None
        self._flags = flags
        self._torch_device = None
        self._stream = None
        self._ring: Optional[RcclRing] = None
        self._p2p: Optional[P2PRing] = None
        self.step_allreduce = step_allreduce
        if not self._partitioned:
            self.free_cells = count_free_cells(obstacles) if free_cells is None else free_cells
            self.free_cells_inv = np.float32(1.0) / np.float32(self.free_cells)
            self.layout = {"y0": 0, "ny_local": params.ny, "macro_k": 0, "ghost": 0, "group": 1}
            self.ny_local, self.displs = [params.ny], [0]
            self.y0, self.nyl = 0, params.ny
            self.partition = Partition(params, self.free_cells, obstacles, 0, device, flags)
            self._window = obstacles
            self.loop = "single"
            return
        self.ny_local, self.displs = decompose(params.ny, self.size)
        want = exchange
        if want == "auto":
            want = "p2p"
        if step_allreduce and want != "rccl":
            if exchange in ("p2p", "torch"):
                raise ValueError("step_allreduce is a mode of the RCCL loop")
            want = "rccl"
        # try the loops in order; what could not be set up on EVERY rank is skipped by all ranks together
        order = {"p2p": ["p2p", "rccl", "torch"], "rccl": ["rccl", "torch"], "torch": ["torch"]}[want]
        if self.rank_grid is not None:
            order = ["p2p"]
        last_err = None
        for loop in order:
            one_step = _capi.FLAG_ONE_STEP if loop == "torch" else 0
            self._make_partition(flags | one_step, free_cells)
            try:
                if loop == "p2p":
                    self._p2p = P2PRing(self.partition, group) if distributed else P2PRing.local_ring([self.partition])[0]
                elif loop == "rccl":
                    self._ring = (RcclRing(self.partition, group, step_allreduce=step_allreduce) if distributed
                                  else RcclRing(self.partition, rank=0, size=1, step_allreduce=step_allreduce))
                else:
                    if not distributed:
                        raise LbmError("a forced-halo run outside torch.distributed needs a native loop (p2p or rccl)")
                    self._setup_torch_loop(device)
                self.loop = loop
                return
            except LbmError as e:             # raised on every rank together (see RcclRing / P2PRing)
                last_err = str(e)
                if strict:
                    break
                if self.rank == 0:
                    import warnings
                    warnings.warn(f"{loop} loop unavailable ({e}); falling back")
        self.partition.close()
        raise LbmError(f"no step loop could be set up for exchange={exchange!r}: {last_err}")

    # -- set-up helpers --
    def _make_partition(self, flags: int, free_cells: Optional[int]) -> None:
        """(Re)create this rank's partition for `flags`: layout from global quantities, obstacle window from
        the local map, or scattered by rank 0 when only rank 0 holds the map."""
        if getattr(self, "partition", None) is not None:
            if self._partition_flags == flags:
                return
            self.partition.close()
        self._partition_flags = flags
        layout_of = ((lambda r: tile_layout(self.params, self.rank_grid[0], self.rank_grid[1], r, flags)) if self.rank_grid is not None
                     else (lambda r: rank_layout(self.params, self.size, r, flags)))
        lay = layout_of(self.rank)
        self.layout = lay
        self.y0, self.nyl = lay["y0"], lay["ny_local"]
        if self.exchange is not None and self.size > 1:
            import torch.distributed as dist
            group = self._group
            have = [None] * self.size
            dist.all_gather_object(have, self.obstacles is not None, group=group)
            if not all(have):
                # scatter of obstacle rows (:968-970) + broadcast of the free-cell count (:966) from rank 0
                box, wins = [None], None
                if self.rank == 0:
                    free = count_free_cells(self.obstacles) if free_cells is None else free_cells
                    wins = [(obstacle_window(self.obstacles, layout_of(r)), free) for r in range(self.size)]
                dist.scatter_object_list(box, wins, src=self.exchange._global(0), group=group)
                self._finish_partition(box[0][0], box[0][1], flags)
                return
        free = count_free_cells(self.obstacles) if free_cells is None else free_cells
        self._finish_partition(obstacle_window(self.obstacles, lay), free, flags)

    def _finish_partition(self, window: np.ndarray, free_cells: int, flags: int) -> None:
        self.free_cells = free_cells
        self.free_cells_inv = np.float32(1.0) / np.float32(free_cells)
        self._window = window
        if self.rank_grid is not None:
            self.partition = Partition(self.params, free_cells, window, device=self.device, flags=flags, tile_of=(self.rank, *self.rank_grid))
            return
        self.partition = Partition(self.params, free_cells, window, device=self.device, flags=flags, rank_of=(self.rank, self.size))

    def _setup_torch_loop(self, device: int) -> None:
        import torch
        self._torch_device = torch.device("cuda", device)
        # one explicit stream carries the step kernels; RCCL orders its own stream against it
        # at batch_isend_irecv() (start) and at wait(), so the exchange overlaps step_interior
        self._stream = torch.cuda.Stream(self._torch_device)
        with torch.cuda.stream(self._stream):
            self.partition.bind_halo_tensors(self._torch_device)
        self._stream.synchronize()

    @classmethod
    def from_files(cls, paramfile: str, obstaclefile: str, **kw) -> "Simulation":
        params = read_params(paramfile)
        obstacles, _ = read_obstacles(obstaclefile, params.nx, params.ny)
        return cls(params, obstacles, **kw)

    def describe(self) -> dict:
        """What actually runs: loop, K, ranks as the transport itself reports them."""
        d = {"loop": self.loop, "macro_k": self.partition.macro_steps, "ranks": self.size, "rccl_nranks": None, "p2p": None,
             "step_allreduce": bool(self._ring is not None and self._ring.step_allreduce)}
        if self._ring is not None:
            d["rccl_nranks"] = self._ring.nranks
        if self._p2p is not None:
            d["p2p"] = self._p2p.describe()
        return d

    def run(self, n_steps: Optional[int] = None) -> np.ndarray:
        """The timed region of the reference (`d2q9-bgk.c:278-398`): step loop + av_vels reduction."""
        n = self.params.max_iters if n_steps is None else n_steps
        if not self._partitioned:
            return self.partition.run(n)
        if self._p2p is not None or self._ring is not None:
            tot = (self._p2p or self._ring).run(n)
            return (tot * np.float64(self.free_cells_inv)).astype(np.float32)          # :367
        import torch
        with torch.cuda.stream(self._stream):
            av = run_partitioned(self.partition, self.exchange, n, self.free_cells_inv, self._torch_device,
                                 self._stream.cuda_stream)
        self._stream.synchronize()
        return av

    def local_cells(self) -> np.ndarray:
        return self.partition.get_cells()

    def _gather_rows(self, local: np.ndarray) -> Optional[np.ndarray]:
        """Rank 0 gets the ranks' row blocks concatenated in rank order (None elsewhere) — the order in which
        the reference's ranks append to final_state.dat (`d2q9-bgk.c:1049-1057`)."""
        if self.size == 1:
            return local
        import torch
        import torch.distributed as dist
        group = self.exchange.group
        on_host = dist.get_backend(group) == "gloo"
        dev = torch.device("cpu") if on_host else torch.device("cuda", self.device)
        mine = torch.from_numpy(local).to(dev)
        if self.rank_grid is not None:           # tile decomposition: rank 0 places every rank's block
            if self.rank != 0:
                dist.send(mine, dst=self.exchange._global(0), group=group)
                return None
            whole = np.empty((self.params.ny, self.params.nx) + tuple(local.shape[2:]), dtype=local.dtype)
            for r in range(self.size):
                lay = tile_layout(self.params, self.rank_grid[0], self.rank_grid[1], r, self._partition_flags)
                block = mine
                if r > 0:
                    block = torch.empty((lay["ny_local"], lay["nx_local"]) + tuple(local.shape[2:]), dtype=mine.dtype, device=dev)
                    dist.recv(block, src=self.exchange._global(r), group=group)
                whole[lay["y0"]:lay["y0"] + lay["ny_local"], lay["x0"]:lay["x0"] + lay["nx_local"]] = block.cpu().numpy()
            return whole
        if self.rank == 0:
            parts = [torch.empty((n,) + tuple(local.shape[1:]), dtype=mine.dtype, device=dev) for n in self.ny_local]
            parts[0] = mine
            for r in range(1, self.size):
                dist.recv(parts[r], src=self.exchange._global(r), group=group)
            return torch.cat(parts, dim=0).cpu().numpy()
        dist.send(mine, dst=self.exchange._global(0), group=group)
        return None

    def gather_cells(self) -> Optional[np.ndarray]:
        """Whole-grid AoS cells on rank 0 (None elsewhere)."""
        return self._gather_rows(self.local_cells())

    def gather_observables(self) -> Optional[np.ndarray]:
        """Whole-grid (u_x, u_y, u, pressure) on rank 0 (None elsewhere): 4 floats per cell computed on the
        device instead of the 9 populations (`d2q9-bgk.c:1084-1111`)."""
        return self._gather_rows(self.partition.get_observables())

    def _whole_map(self) -> np.ndarray:
        if self.obstacles is None:
            raise LbmError("the whole obstacle map lives on rank 0 only")
        return self.obstacles

    def reynolds(self, cells: Optional[np.ndarray] = None, *, observables: Optional[np.ndarray] = None) -> Optional[float]:
        """`calc_reynolds` on the whole grid, reference order (`d2q9-bgk.c:707-757,1002-1008`).  Rank 0
        returns the value; the other ranks of a distributed run take part in the gather and return None."""
        if cells is None and observables is None:
            observables = self.gather_observables()
        if self.rank != 0:
            return None
        if observables is not None:
            tot_u = np.float32(av_velocity_obs(self.params, observables, self._whole_map()))
        else:
            tot_u = np.float32(av_velocity_host(self.params, cells, self._whole_map()))
        return reynolds(self.params, float(tot_u * self.free_cells_inv))

    def write_values(self, av_vels: np.ndarray, directory: str = ".", cells: Optional[np.ndarray] = None, *,
                     observables: Optional[np.ndarray] = None) -> None:
        """`write_values()` (`d2q9-bgk.c:1034-1143`): rank 0 writes both files; the other ranks only take part
        in the gather (when neither cells nor observables are passed in)."""
        if cells is None and observables is None:
            observables = self.gather_observables()
        if self.rank != 0:
            return
        path = os.path.join(directory, "final_state.dat")
        if observables is not None:
            write_final_state_obs(path, self.params, observables, self._whole_map())
        else:
            write_final_state(path, self.params, cells, self._whole_map())
        write_av_vels(os.path.join(directory, "av_vels.dat"), av_vels)

    def close(self) -> None:
        if self._ring is not None:
            self._ring.close()
            self._ring = None
        if self._p2p is not None:
            self._p2p.close()
            self._p2p = None
        if getattr(self, "partition", None) is not None:
            self.partition.close()
