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


def av_velocity_host(params: Params, cells: np.ndarray, obstacles: np.ndarray) -> float:
    """`av_velocity()` without the reduce (`d2q9-bgk.c:716-751`): float tot_u over the given rows."""
    lib = _capi.load_library()
    cells = np.ascontiguousarray(cells, dtype=np.float32)
    obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
    cp = _cparams(params)
    return float(lib.lbm_av_velocity_host(C.byref(cp), _capi.as_float_ptr(cells), _capi.as_int_ptr(obstacles),
                                          obstacles.shape[0]))


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
                 device: int = 0, flags: int = 0, obstacles_global: Optional[np.ndarray] = None):
        """obstacles_global: the whole (ny, nx) map; when given (`lbm_create_global`) an eligible
        row partition runs in K-step mode (`macro_steps` > 0) and obstacles_rows is ignored."""
        self._lib = _capi.load_library()
        obst = np.ascontiguousarray(obstacles_rows, dtype=np.int32)
        if obst.ndim != 2 or obst.shape[1] != params.nx:
            raise ValueError("obstacles_rows must be (ny_local, nx)")
        self.params, self.free_cells, self.y0, self.ny_local, self.device = params, free_cells, y0, obst.shape[0], device
        self.free_cells_inv = np.float32(1.0) / np.float32(free_cells)          # d2q9-bgk.c:950
        self._ctx = C.c_void_p()
        cp = _cparams(params)
        if obstacles_global is not None:
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
    def get_cells(self) -> np.ndarray:
        cells = np.empty((self.ny_local, self.params.nx, _capi.NSPEEDS), dtype=np.float32)
        check(self._lib.lbm_get_cells(self._ctx, _capi.as_float_ptr(cells)))
        return cells

    def set_cells(self, cells: np.ndarray) -> None:
        cells = np.ascontiguousarray(cells, dtype=np.float32)
        if cells.shape != (self.ny_local, self.params.nx, _capi.NSPEEDS):
            raise ValueError("cells must be (ny_local, nx, 9)")
        check(self._lib.lbm_set_cells(self._ctx, _capi.as_float_ptr(cells)))

    def av_velocity_sum(self) -> float:
        tot = C.c_double(0.0)
        check(self._lib.lbm_av_velocity_sum(self._ctx, C.byref(tot)))
        return tot.value

    def last_run_kernel_ms(self) -> tuple[float, int]:
        """(device ms from first to last step kernel of the last run, number of step-kernel launches)."""
        ms, n = C.c_double(0.0), C.c_int(0)
        check(self._lib.lbm_last_run_kernel_ms(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

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

    def __init__(self, partition: "Partition", group=None, *, rank: Optional[int] = None, size: Optional[int] = None):
        self._lib = _capi.load_rccl_library()
        self.partition = partition
        ident = C.create_string_buffer(_capi.COMM_ID_BYTES)
        if size is None:
            import torch.distributed as dist
            rank, size = dist.get_rank(group), dist.get_world_size(group)
            box = [None]
            if rank == 0:
                check(self._lib.lbm_comm_unique_id(ident))
                box[0] = ident.raw
            src = dist.get_global_rank(group, 0) if group is not None else 0
            dist.broadcast_object_list(box, src=src, group=group)
            ident = C.create_string_buffer(box[0], _capi.COMM_ID_BYTES)
        else:                                   # single process (size must be 1): exchange with itself
            check(self._lib.lbm_comm_unique_id(ident))
        self.rank, self.size = rank, size
        self._comm = C.c_void_p()
        check(self._lib.lbm_comm_create(C.byref(self._comm), partition._ctx, ident, size, rank))

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

class Simulation:
    """paramfile + obstaclefile -> av_vels, final state, Reynolds number (the reference's `main`).

    With a torch.distributed process group of size > 1 (one process per GPU), each rank owns the
    rows `decompose()` gives it and `run()` performs the halo exchange over the group — by default in
    the native RCCL loop, K steps per exchange where the partition is eligible (`lbm_create_global`,
    `Partition.macro_steps`); otherwise the whole grid lives on one GPU and `run()` is one `lbm_run`."""

    def __init__(self, params: Params, obstacles: np.ndarray, *, device: int = 0, flags: int = 0,
                 distributed: bool = False, group=None, exchange: str = "rccl"):
        """exchange: how a distributed run moves its halos — "rccl" = the native loop of
        liblbm_d2q9_rccl.so (default), "torch" = torch.distributed P2P ops from Python."""
        if exchange not in ("rccl", "torch"):
            raise ValueError("exchange must be 'rccl' or 'torch'")
        obstacles = np.ascontiguousarray(obstacles, dtype=np.int32)
        if obstacles.shape != (params.ny, params.nx):
            raise ValueError("obstacles must be (ny, nx)")
        self.params, self.obstacles = params, obstacles
        self.free_cells = count_free_cells(obstacles)
        self.free_cells_inv = np.float32(1.0) / np.float32(self.free_cells)
        self.device = device
        self.exchange: Optional[HaloExchange] = None
        self.rank, self.size = 0, 1
        if distributed:
            self.exchange = HaloExchange(group)
            self.rank, self.size = self.exchange.rank, self.exchange.size
        self.ny_local, self.displs = decompose(params.ny, self.size)
        y0, nyl = self.displs[self.rank], self.ny_local[self.rank]
        self.y0, self.nyl = y0, nyl
        # the native RCCL loop runs eligible partitions in K-step mode (needs the whole obstacle map for
        # the ghost rows); the torch loop drives the one-step split-phase calls
        one_step = _capi.FLAG_ONE_STEP if exchange == "torch" else 0
        self.partition = Partition(params, self.free_cells, obstacles[y0:y0 + nyl], y0, device, flags | one_step,
                                   obstacles_global=obstacles)
        self._torch_device = None
        self._stream = None
        self._ring: Optional[RcclRing] = None
        # a forced-halo whole-grid partition is a 1-rank ring that exchanges with itself (:245-247)
        self._partitioned = self.size > 1 or bool(flags & _capi.FLAG_FORCE_HALO)
        if self._partitioned and not distributed:
            if exchange != "rccl":
                raise ValueError("a forced-halo run outside torch.distributed needs exchange='rccl'")
            self._ring = RcclRing(self.partition, rank=0, size=1)
        elif self._partitioned and exchange == "rccl":
            self._ring = self._ring_or_none(group)
            if self._ring is None:      # every rank agreed: the native loop is unavailable, use the torch loop
                self.partition.close()
                self.partition = Partition(params, self.free_cells, obstacles[y0:y0 + nyl], y0, device,
                                           flags | _capi.FLAG_ONE_STEP, obstacles_global=obstacles)
                self._setup_torch_loop(device)
        elif self._partitioned:
            self._setup_torch_loop(device)

    def _ring_or_none(self, group) -> Optional[RcclRing]:
        """RcclRing, or None on EVERY rank if liblbm_d2q9_rccl.so or its communicator failed on any."""
        import torch
        import torch.distributed as dist
        ring, err = None, None
        try:
            ring = RcclRing(self.partition, group)
        except (LbmError, RuntimeError, OSError) as e:       # missing library, RCCL error
            err = e
        ok = torch.tensor([0 if ring is None else 1], dtype=torch.int32,
                          device="cpu" if dist.get_backend(group) == "gloo" else torch.device("cuda", self.device))
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 1:
            return ring
        if ring is not None:
            ring.close()
        if self.rank == 0:
            import warnings
            warnings.warn(f"native RCCL loop unavailable ({err}); falling back to the torch.distributed loop")
        return None

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

    def run(self, n_steps: Optional[int] = None) -> np.ndarray:
        """The timed region of the reference (`d2q9-bgk.c:278-398`): step loop + av_vels reduction."""
        n = self.params.max_iters if n_steps is None else n_steps
        if not self._partitioned:
            return self.partition.run(n)
        if self._ring is not None:
            tot = self._ring.run(n)
            return (tot * np.float64(self.free_cells_inv)).astype(np.float32)          # :367
        import torch
        with torch.cuda.stream(self._stream):
            av = run_partitioned(self.partition, self.exchange, n, self.free_cells_inv, self._torch_device,
                                 self._stream.cuda_stream)
        self._stream.synchronize()
        return av

    def local_cells(self) -> np.ndarray:
        return self.partition.get_cells()

    def gather_cells(self) -> Optional[np.ndarray]:
        """Whole-grid AoS cells on rank 0 (None elsewhere) — what `write_values` serialises rank by rank."""
        local = self.local_cells()
        if self.size == 1:
            return local
        import torch
        import torch.distributed as dist
        if self._torch_device is None:
            self._torch_device = torch.device("cuda", self.device)
        if dist.get_backend(self.exchange.group) == "gloo":
            self._gather_device = torch.device("cpu")
        else:
            self._gather_device = self._torch_device
        mine = torch.from_numpy(local).to(self._gather_device)
        if self.rank == 0:
            parts = [torch.empty((n, self.params.nx, _capi.NSPEEDS), dtype=torch.float32, device=self._gather_device)
                     for n in self.ny_local]
            parts[0] = mine
            for r in range(1, self.size):
                dist.recv(parts[r], src=self.exchange._global(r), group=self.exchange.group)
            return torch.cat(parts, dim=0).cpu().numpy()
        dist.send(mine, dst=self.exchange._global(0), group=self.exchange.group)
        return None

    def reynolds(self, cells: Optional[np.ndarray] = None) -> float:
        """`calc_reynolds` on whole-grid cells (rank 0), reference order (`d2q9-bgk.c:707-757,1002-1008`)."""
        cells = self.gather_cells() if cells is None else cells
        tot_u = np.float32(av_velocity_host(self.params, cells, self.obstacles))
        return reynolds(self.params, float(tot_u * self.free_cells_inv))

    def write_values(self, av_vels: np.ndarray, directory: str = ".", cells: Optional[np.ndarray] = None) -> None:
        cells = self.gather_cells() if cells is None else cells
        if self.rank == 0:
            write_final_state(os.path.join(directory, "final_state.dat"), self.params, cells, self.obstacles)
            write_av_vels(os.path.join(directory, "av_vels.dat"), av_vels)

    def close(self) -> None:
        if self._ring is not None:
            self._ring.close()
        self.partition.close()
