"""CPU suite, part 3: the row-partitioned host logic (decomposition, neighbour ranks, exchange
order, overlap sequence, end-of-run reduction) over torch.distributed with the gloo backend and
world sizes 2 and 3.  The device is stood in for by tests/oracle_lib.OraclePartition (the C
restatement stepping one rank's halo'd rows), injected through the PartitionBackend protocol — the
product's host code (HaloExchange, run_partitioned, decompose) is what is under test."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, size, init_file, case, steps, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mpilattice_boltzmann_amd as lbm
    import oracle_lib
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=size)
    p = lbm.Params(*case["params"])
    obst = lbm.synthetic_obstacles(p.nx, p.ny, case["p"], case["seed"], case["walls"])
    free = lbm.count_free_cells(obst)
    ny_local, displs = lbm.decompose(p.ny, size)
    ex = lbm.HaloExchange()
    assert (ex.south, ex.north) == ((rank - 1) % size, (rank + 1) % size)
    part = oracle_lib.OraclePartition(p, free, obst[displs[rank]:displs[rank] + ny_local[rank]], displs[rank],
                                      is_last=(rank == size - 1))
    av = lbm.run_partitioned(part, ex, steps, np.float32(1.0) / np.float32(free))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), cells=part.get_cells(), av=av, y0=displs[rank])
    dist.barrier()
    dist.destroy_process_group()


CASES = [
    dict(params=(32, 24, 40, 4, 0.1, 0.01, 1.7), p=0.08, seed=5, walls=False),
    dict(params=(16, 7, 25, 4, 0.1, 0.005, 1.2), p=0.15, seed=6, walls=True),     # uneven split, 3 + 4 rows
]


@pytest.mark.parametrize("size", [2, 3])
@pytest.mark.parametrize("case", CASES)
def test_partitioned_run_over_gloo_equals_single_rank(lbm, oracle, size, case):
    steps = case["params"][2]
    with tempfile.TemporaryDirectory() as tmp:
        init_file = os.path.join(tmp, "rendezvous")
        mp.spawn(_worker, args=(size, init_file, case, steps, tmp), nprocs=size, join=True)
        ranks = [np.load(os.path.join(tmp, f"rank{r}.npz")) for r in range(size)]
    p = lbm.Params(*case["params"])
    obst = lbm.synthetic_obstacles(p.nx, p.ny, case["p"], case["seed"], case["walls"])
    ref_cells, ref_av, ref_exact = oracle.run(p, obst, steps)
    cells = np.concatenate([r["cells"] for r in ranks], axis=0)
    assert [int(r["y0"]) for r in ranks] == lbm.decompose(p.ny, size)[1]
    assert np.array_equal(cells.view(np.uint32), ref_cells.view(np.uint32))          # decomposition-invariant state
    for r in ranks:                                                                   # all-reduced: same on every rank
        assert np.array_equal(r["av"], ranks[0]["av"])
    assert np.allclose(ranks[0]["av"].astype(np.float64), ref_exact, rtol=1e-6)
    assert np.allclose(ranks[0]["av"], ref_av, rtol=2e-4)


class _FakeP2PLib:
    """Stands in for liblbm_d2q9.so's lbm_p2p_* entry points (no GPU here): create / handle succeed on every rank,
    connect fails on the rank named in `fail_on` ONLY — what hipIpcOpenMemHandle or hipDeviceEnablePeerAccess failing
    for one device pair looks like from the host side."""

    def __init__(self, rank, fail_on):
        self.rank, self.fail_on, self.calls = rank, fail_on, []

    def lbm_p2p_create(self, out, *a):
        out._obj.value = 1                                # a non-null lbm_p2p*: close() has something to free
        return 0

    def lbm_p2p_handle(self, t, buf):
        return 0

    def lbm_p2p_connect(self, t, blobs):
        self.calls.append("connect")
        return 1 if self.rank == self.fail_on else 0

    def lbm_p2p_disconnect(self, t):
        self.calls.append("disconnect")
        return 0

    def lbm_p2p_destroy(self, t):
        self.calls.append("destroy")
        return 0

    def lbm_last_error(self):
        return b"hipIpcOpenMemHandle: invalid device pointer"


def _connect_worker(rank, size, init_file, fail_on, out_dir):
    sys.path.insert(0, ROOT)
    import ctypes as C
    import torch.distributed as dist
    import mpilattice_boltzmann_amd as lbm
    from mpilattice_boltzmann_amd import _capi, host
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=size)
    fake = _FakeP2PLib(rank, fail_on)
    _capi.load_library = lambda: fake                     # host.P2PRing and check() resolve the library through _capi

    class Part:                                           # only ._ctx is touched before connect
        _ctx = C.c_void_p(1)
    what = "connected"
    try:
        ring = lbm.P2PRing(Part(), None)
        ring._t = C.c_void_p(0)                           # nothing for __del__ to do
    except lbm.LbmError as e:
        what = str(e)
    # the ranks must still be in step with one another: a collective right after the constructor pairs up
    flag = [None] * size
    dist.all_gather_object(flag, rank)
    with open(os.path.join(out_dir, f"rank{rank}.txt"), "w") as fh:
        fh.write(what + "\n" + ",".join(fake.calls) + "\n" + ",".join(str(v) for v in flag))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("size,fail_on", [(2, 1), (3, 0), (3, -1)])
def test_p2p_connect_failure_on_one_rank_is_raised_on_every_rank(lbm, size, fail_on):
    """lbm_p2p_connect failing on ONE rank used to raise there alone while the others sat in a barrier —
    mismatched collectives from then on.  Now the outcome is agreed on: every rank raises the same LbmError, unmaps
    and frees, and the process group is still in step afterwards."""
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_connect_worker, args=(size, os.path.join(tmp, "rendezvous"), fail_on, tmp), nprocs=size, join=True)
        got = [open(os.path.join(tmp, f"rank{r}.txt")).read().split("\n") for r in range(size)]
    for r, (what, calls, flag) in enumerate(got):
        assert flag == ",".join(str(v) for v in range(size))
        if fail_on < 0:
            assert what == "connected" and calls == "connect"
        else:
            assert what.startswith(f"peer-to-peer connect failed on rank(s) {fail_on}: ") and "hipIpcOpenMemHandle" in what
            assert calls == "connect,disconnect,destroy"


def _tile_worker(rank, size, init_file, grid, nx, ny, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import mpilattice_boltzmann_amd as lbm
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=size)
    p = lbm.Params(nx, ny, 10, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.05, 11, True)
    # the host logic of Simulation for a tile run, with the device object left out: rank 0 alone holds the map and scatters the windows
    # (d2q9-bgk.c:966-970 for blocks instead of rows), every rank's block is gathered into its place on rank 0
    sim = object.__new__(lbm.Simulation)
    sim.params, sim.device, sim._group, sim.rank_grid = p, 0, None, grid
    sim.exchange = lbm.HaloExchange()
    sim.rank, sim.size = sim.exchange.rank, sim.exchange.size
    sim.obstacles = obst if rank == 0 else None
    got = {}
    sim._finish_partition = lambda window, free, flags: got.update(window=window, free=free)
    sim._make_partition(0, None)
    lay = lbm.tile_layout(p, grid[0], grid[1], rank)
    assert sim.layout == lay and (sim.y0, sim.nyl) == (lay["y0"], lay["ny_local"])
    assert got["free"] == lbm.count_free_cells(obst) and np.array_equal(got["window"], lbm.obstacle_window(obst, lay))
    index = np.arange(ny * nx * 3, dtype=np.float32).reshape(ny, nx, 3)
    mine = np.ascontiguousarray(index[lay["y0"]:lay["y0"] + lay["ny_local"], lay["x0"]:lay["x0"] + lay["nx_local"]])
    whole = sim._gather_rows(mine)
    assert (whole is None) == (rank != 0)
    if rank == 0:
        assert np.array_equal(whole, index)
        open(os.path.join(out_dir, "ok"), "w").write("ok")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("grid,nx,ny", [((2, 2), 512, 256), ((2, 1), 644, 128), ((3, 1), 1290, 64)])
def test_tile_scatter_and_gather_over_gloo(lbm, grid, nx, ny):
    """The 2-D (tile) decomposition's host side over a gloo group (world sizes 4, 2, 3): rank 0 hands every rank the window of rows AND
    columns it needs, blocks come back into their places — uneven column blocks included."""
    size = grid[0] * grid[1]
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_tile_worker, args=(size, os.path.join(tmp, "rendezvous"), grid, nx, ny, tmp), nprocs=size, join=True)
        assert os.path.exists(os.path.join(tmp, "ok"))
