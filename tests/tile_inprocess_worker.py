"""The ranks of a px x py TILE (2-D) decomposition as contexts of ONE process, one host thread per rank (test helper, run as a
fresh process by tests/test_gpu_parity.py with GPU_MAX_HW_QUEUES raised, as tests/p2p_inprocess_worker.py).  Every rank keeps ghost
rows and ghost columns; per exchange the columns travel west / east first, then whole storage rows south / north.
argv: nx ny px py K ghost group runs(comma separated) [walls] [flags=<lbm_create flags>] [sched=edge|serial] [yghost]
(yghost: LBM_TUNE_TILE_GHOST_ROWS=1 — column blocks (py = 1) keep ghost rows and push rows onto themselves instead of wrapping in the launch;
ranks of one process on one device run the serial schedule whatever is asked for; a 1 x 1 grid takes the edge-stream schedule when asked)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main() -> int:
    nx, ny, px, py, K = (int(v) for v in sys.argv[1:6])
    ghost, group = sys.argv[6], sys.argv[7]
    runs = [int(v) for v in sys.argv[8].split(",")]
    walls = "walls" in sys.argv[9:]
    flags = next((int(a.split("=")[1]) for a in sys.argv[9:] if a.startswith("flags=")), 0)
    os.environ.pop("LBM_TUNE_TILE_GHOST_ROWS", None)
    if "yghost" in sys.argv[9:]:
        os.environ["LBM_TUNE_TILE_GHOST_ROWS"] = "1"
    sched = next((a.split("=")[1] for a in sys.argv[9:] if a.startswith("sched=")), "")
    os.environ.pop("LBM_P2P_SCHEDULE", None)
    if sched:
        os.environ["LBM_P2P_SCHEDULE"] = sched
    if K:
        os.environ["LBM_TUNE_MACRO_K"] = str(K)
    if ghost != "-":
        os.environ["LBM_TUNE_MACRO_GHOST"] = ghost
    if group != "-":
        os.environ["LBM_TUNE_MACRO_GROUP"] = group
    import mpilattice_boltzmann_amd as lbm
    import oracle_lib
    steps = sum(runs)
    size = px * py
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx * 5 + ny, walls)
    free = lbm.count_free_cells(obst)
    lays = [lbm.tile_layout(p, px, py, r, flags) for r in range(size)]
    assert sum(l["nx_local"] * l["ny_local"] for l in lays) == nx * ny
    parts = [lbm.Partition(p, free, lbm.obstacle_window(obst, lays[r]), flags=flags, tile_of=(r, px, py)) for r in range(size)]
    if flags & lbm._capi.FLAG_FAST_AVVELS:
        assert "fast av_vels" in parts[0].describe()["kernel"]
    if flags & lbm._capi.FLAG_EXACT_AVVELS:
        assert "double-precision" in parts[0].describe()["kernel"]
    assert all(q.tile_info() == lays[r] for r, q in enumerate(parts)), (parts[0].tile_info(), lays[0])
    rings = lbm.P2PRing.local_ring(parts)
    d = rings[0].describe()
    assert f"tiles {px} x {py}" in d and ("edge stream" if (sched == "edge" and size == 1) else "serial") in d, d
    assert ("rows wrap in the launch" in d) == (py == 1 and "yghost" not in sys.argv[9:]), d
    assert lays[0]["ghost_y"] == (0 if "rows wrap" in d else lays[0]["ghost"])
    out = [lbm.P2PRing.run_all(rings, n) for n in runs]
    for o in out:
        for r in range(1, size):                                # the reduction is bitwise the same on every rank
            assert np.array_equal(o[r], o[0])
    cells = np.empty((ny, nx, 9), dtype=np.float32)
    obs = np.empty((ny, nx, 4), dtype=np.float32)
    digest, tot_u = 0, 0.0
    for q, l in zip(parts, lays):
        ys, xs = slice(l["y0"], l["y0"] + l["ny_local"]), slice(l["x0"], l["x0"] + l["nx_local"])
        cells[ys, xs] = q.get_cells()
        obs[ys, xs] = q.get_observables()
        digest = (digest + q.checksum()) % (1 << 64)
        tot_u += q.av_velocity_sum()
    ref_cells, _, ref_exact = oracle_lib.run(p, obst, steps, nthreads=4)
    assert np.array_equal(cells.view(np.uint32), ref_cells.view(np.uint32)), "populations differ from the oracle"
    av = np.concatenate([o[0] for o in out]) * np.float64(np.float32(1.0) / np.float32(free))
    tol = 2e-6 if flags & lbm._capi.FLAG_FAST_AVVELS else 1e-12          # (float terms: an ulp of the float av_vels is)
    assert np.max(np.abs(av - ref_exact) / ref_exact) < tol, np.max(np.abs(av - ref_exact) / ref_exact)
    # the same state on one context: digests add up, observables and the velocity sum agree
    whole = lbm.Partition(p, free, obst)
    whole.set_cells(ref_cells)
    assert digest == whole.checksum()
    assert np.array_equal(obs.view(np.uint32), whole.get_observables().view(np.uint32))
    assert abs(tot_u - whole.av_velocity_sum()) <= 1e-9 * abs(tot_u)
    whole.close()
    # set_cells of a tile: its block only; a further run from that state matches the oracle's continuation
    more = 9
    for q, l in zip(parts, lays):
        q.set_cells(ref_cells[l["y0"]:l["y0"] + l["ny_local"], l["x0"]:l["x0"] + l["nx_local"]])
    lbm.P2PRing.run_all(rings, more)
    p2 = lbm.Params(nx, ny, steps + more, 4, 0.1, 0.01, 1.7)
    ref2, _, _ = oracle_lib.run(p2, obst, steps + more, nthreads=4)
    for q, l in zip(parts, lays):
        got = q.get_cells()
        assert np.array_equal(got.view(np.uint32), ref2[l["y0"]:l["y0"] + l["ny_local"], l["x0"]:l["x0"] + l["nx_local"]].view(np.uint32))
    for ring in rings:
        ring.close()
    for q in parts:
        q.close()
    print(f"TILES ok {px}x{py} {d}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
