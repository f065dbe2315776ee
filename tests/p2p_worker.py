"""One rank of a multi-process partitioned run (test helper, launched by torch.distributed.run from
tests/test_gpu_parity.py).  The ranks form a gloo group and run a native loop — by default the peer-to-peer one of
include/lbm_d2q9_p2p.h, mapping one another's grids with hipIpcOpenMemHandle.  On a one-GPU box all ranks share
device 0: every code path of an N-GPU run except the xGMI wire.  With LBM_WORKER_DEVICE=local_rank every rank takes
its own GPU (the tests that switch themselves on when the box has several): then the stores, flags and RCCL
messages cross real links, and the RCCL loop (which refuses two ranks on one device) can run with nranks > 1.
Several cases per launch; rank 0 compares the gathered state with the oracle, bit for bit.

    python -m torch.distributed.run --nproc-per-node N ... tests/p2p_worker.py '<json list of cases>'
case = {"nx", "ny", "K" (0 = library default), "schedule" ("edge" | "serial" | ""), "runs": [steps, ...], "p", "seed", "walls",
        "scatter" (only rank 0 holds the obstacle map), "exchange" ("p2p" | "rccl"), "step_allreduce",
        "ghost", "group" (LBM_TUNE_MACRO_GHOST / _GROUP: ghost rows kept and most launches per halo exchange; default 2 K rows, two launches),
        "grid" ([px, py]: the tile (2-D) decomposition over px x py = N ranks instead of row blocks; peer-to-peer loop)}"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> int:
    import torch
    import torch.distributed as dist
    import mpilattice_boltzmann_amd as lbm
    import oracle_lib
    cases = json.loads(sys.argv[1])
    own_gpu = os.environ.get("LBM_WORKER_DEVICE") == "local_rank"
    device = int(os.environ["LOCAL_RANK"]) if own_gpu else 0
    torch.cuda.set_device(device)
    dist.init_process_group("gloo")
    rank, size = dist.get_rank(), dist.get_world_size()
    print(f"RANK {rank} UP", flush=True)          # the group exists: whatever happens from here on is a verdict, never retried (conftest.py)
    bad = 0
    for i, c in enumerate(cases):
        exchange = c.get("exchange", "p2p")
        for key, val in (("LBM_TUNE_MACRO_K", c.get("K", 0)), ("LBM_P2P_SCHEDULE", c.get("schedule", "")),
                         ("LBM_RCCL_SCHEDULE", c.get("schedule", "") if exchange == "rccl" else ""),
                         ("LBM_TUNE_MACRO_GHOST", c.get("ghost", "")), ("LBM_TUNE_MACRO_GROUP", c.get("group", ""))):
            if val:
                os.environ[key] = str(val)
            else:
                os.environ.pop(key, None)
        total = sum(c["runs"])
        p = lbm.Params(c["nx"], c["ny"], total, 4, 0.1, 0.01, 1.7)
        obst = lbm.synthetic_obstacles(p.nx, p.ny, c.get("p", 0.03), c.get("seed", 5), c.get("walls", False))
        mine = obst if (rank == 0 or not c.get("scatter")) else None
        sim = lbm.Simulation(p, mine, device=device, distributed=True, exchange=exchange, strict=True,
                             step_allreduce=bool(c.get("step_allreduce")), rank_grid=tuple(c["grid"]) if c.get("grid") else None)
        assert sim.loop == exchange, sim.describe()
        if c.get("grid"):
            assert f"tiles {c['grid'][0]} x {c['grid'][1]}" in sim.describe()["p2p"], sim.describe()
        if exchange == "p2p":
            assert "ipc" in sim.describe()["p2p"], sim.describe()
            assert ("one-step" in sim.describe()["p2p"]) == (sim.partition.macro_steps == 0)
        else:
            assert sim.describe()["rccl_nranks"] == size and sim.describe()["step_allreduce"] == bool(c.get("step_allreduce")), sim.describe()
        if c.get("K"):
            assert sim.partition.macro_steps == c["K"]
        av = np.concatenate([sim.run(n) for n in c["runs"]])
        everyone = [None] * size
        dist.all_gather_object(everyone, av.tobytes())
        same_av = all(b == everyone[0] for b in everyone)          # the reduction is bitwise the same on every rank
        digests = [None] * size
        dist.all_gather_object(digests, sim.partition.checksum())
        cells = sim.gather_cells()
        sim.close()
        ok = True
        if rank == 0:
            ref_cells, _, ref_exact = oracle_lib.run(p, obst, total, nthreads=4)
            ok = bool(np.array_equal(cells.view(np.uint32), ref_cells.view(np.uint32)))
            err = float(np.max(np.abs(av.astype(np.float64) - ref_exact) / ref_exact))
            ok = ok and err < 1e-6 and same_av
            # additive digest: the ranks' digests sum to the digest of the whole grid on one context
            whole = lbm.Simulation(p, obst, device=device)
            whole.run(total)
            ok = ok and (sum(digests) % (1 << 64)) == whole.partition.checksum()
            whole.close()
            print(f"CASE {i} {'ok' if ok else 'FAILED'} ranks={size} {c} av_err={err:.2e} same_av={same_av}", flush=True)
        flag = [ok]
        dist.broadcast_object_list(flag, src=0)
        bad += 0 if flag[0] else 1
    dist.barrier()
    dist.destroy_process_group()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
