"""Several ranks of one peer-to-peer run as contexts of ONE process, one host thread per rank (test helper, run
as a fresh process by tests/test_gpu_parity.py with GPU_MAX_HW_QUEUES raised: ranks that share a device need a
hardware queue each — see lbm_p2p_connect).  argv: nx ny size K schedule"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> int:
    nx, ny, size, K = (int(v) for v in sys.argv[1:5])
    os.environ["LBM_TUNE_MACRO_K"] = str(K)
    os.environ["LBM_P2P_SCHEDULE"] = sys.argv[5]
    import mpilattice_boltzmann_amd as lbm
    import oracle_lib
    steps = 31
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx * 5 + ny, False)
    free = lbm.count_free_cells(obst)
    lays = [lbm.rank_layout(p, size, r) for r in range(size)]
    assert all(l["macro_k"] == K for l in lays), lays
    parts = [lbm.Partition(p, free, lbm.obstacle_window(obst, lays[r]), rank_of=(r, size)) for r in range(size)]
    rings = lbm.P2PRing.local_ring(parts)
    assert all("serial" in r.describe() and "in-process" in r.describe() for r in rings), rings[0].describe()
    a = lbm.P2PRing.run_all(rings, 20)
    b = lbm.P2PRing.run_all(rings, 11)
    for r in range(1, size):                                    # the reduction is bitwise the same on every rank
        assert np.array_equal(a[r], a[0]) and np.array_equal(b[r], b[0])
    cells = np.concatenate([q.get_cells() for q in parts], axis=0)
    for ring in rings:
        ring.close()
    for q in parts:
        q.close()
    ref_cells, _, ref_exact = oracle_lib.run(p, obst, steps, nthreads=4)
    assert np.array_equal(cells.view(np.uint32), ref_cells.view(np.uint32))
    av = np.concatenate([a[0], b[0]]) * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < 1e-12
    print("IN-PROCESS RING ok")
    return 0


if __name__ == "__main__":
    sys.exit(main())
