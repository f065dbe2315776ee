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
