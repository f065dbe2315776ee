"""GPU suite: the HIP path, called through the C ABI, against the oracle and the golden vectors.

Bar: populations BIT-EXACT (the kernels keep the reference's operation order, unfused, with
correctly rounded 1/x and sqrt), hence final_state.dat byte-identical to the reference binary's
(digests.json).  av_vels: the per-cell terms are the reference's, summed as a double tree instead
of a serial float accumulator, so they are compared (a) with the oracle's double-accumulated
yardstick at 1e-6 relative and (b) with the reference-order values at the tolerance the float
accumulator itself allows, and (c) with the shipped goldens under check.py's 1 % rule."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, deck_paths

pytestmark = pytest.mark.gpu

SMALL = ["tiny_8x3", "open_64x48", "rand_64x48", "walls_40x24", "dense_32x32", "strongaccel_32x16",
         "accelrow_blocked_32x16", "column_24x20", "wide_256x8", "tall_8x256", "synth_512x512_t100",
         "128x256_t2000", "256x256_t1000", "1024x1024_t200"]
STRESS_KINDS = ["open", "dense", "accel_row_blocked", "strong_accel", "omega_low", "omega_high"]
AV_EXACT_RTOL = 1e-6


def free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def load_case(lbm, digests, name):
    ppath, opath = deck_paths(name, digests)
    p = lbm.read_params(ppath)
    obst, free = lbm.read_obstacles(opath, p.nx, p.ny)
    return p, obst, free


@pytest.fixture(params=["vector", "narrow", "auto"])
def kernel_form(request, monkeypatch):
    """Which step kernel lbm_run uses on the small test grids: "vector" = 4 cells per lane on every
    size; "narrow" = the library's choice among the one-step kernels (one cell per lane up to 64 K
    cells); "auto" = the library's full choice (up to 8 steps per launch with lbm_tile_kernel on
    periodic grids whose edges are multiples of 16 and that hold <= 256 K cells)."""
    if request.param == "vector":
        monkeypatch.setenv("LBM_TUNE_NARROW_MAX", "0")
    if request.param in ("vector", "narrow"):
        monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
        monkeypatch.setenv("LBM_TUNE_MULTI_K", "0")
        monkeypatch.setenv("LBM_TUNE_MACRO_K", "0")
    return request.param


@pytest.mark.parametrize("name", SMALL)
def test_state_bit_exact_and_files_match_reference_digests(lbm, oracle, digests, tmp_path, kernel_form, name):
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst)
    av = sim.run()
    cells = sim.local_cells()
    re = sim.reynolds(cells)
    sim.write_values(av, str(tmp_path), cells)
    sim.close()
    ref_cells, ref_av, ref_exact = oracle.run(p, obst, p.max_iters, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]     # = the reference binary's file
    assert "Reynolds number:\t\t%.12E" % re == digests[name]["reynolds_line"]
    assert av.shape == ref_av.shape
    assert np.max(np.abs(av.astype(np.float64) - ref_exact) / ref_exact) < AV_EXACT_RTOL
    assert np.allclose(av, ref_av, rtol=5e-3 if p.nx * p.ny > 100000 else 2e-4)


@pytest.mark.parametrize("name", ["128x128", "128x256", "256x256", "1024x1024"])
def test_cli_on_shipped_decks(lbm, digests, tmp_path, name):
    """The drop-in CLI on the four shipped decks (BASELINE.json configs): stdout contract, files in
    cwd, final_state.dat byte-identical to the reference binary's, av_vels inside check.py's 1 %."""
    ppath, opath = deck_paths(name, digests)
    r = subprocess.run([lbm.CLI_PATH, ppath, opath], cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0] == "==done=="                                                   # d2q9-bgk.c:411-415
    assert out[1] == digests[name]["reynolds_line"]
    assert out[2].startswith("Elapsed time:\t\t\t") and out[2].endswith(" (s)")
    assert out[3].startswith("Elapsed user CPU time:\t\t") and out[4].startswith("Elapsed system CPU time:\t")
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    ck = lbm.checker
    ref_fs = os.path.join(GOLDEN, "check", f"{name}.final_state.dat.gz")
    has_fs = os.path.exists(ref_fs)            # 256x256 / 1024x1024 final-state goldens are absent upstream
    rep = ck.check_files(os.path.join(GOLDEN, "check", f"{name}.av_vels.dat.gz"), ref_fs if has_fs else None,
                         str(tmp_path / "av_vels.dat"), str(tmp_path / "final_state.dat") if has_fs else None)
    assert rep.ok, rep.message
    assert abs(rep.av_vels.max_diff_pcnt) < 0.3
    # reference-binary av_vels samples (float accumulator) vs ours (double tree)
    av = ck.load_av_vels(str(tmp_path / "av_vels.dat"))
    steps = np.asarray(digests[name]["av_sample_steps"])
    ref = np.asarray(digests[name]["av_sample_values"])
    assert av.size == digests[name]["steps"]
    assert np.allclose(av[steps], ref, rtol=4e-3 if name == "1024x1024" else 5e-4)


def test_repeated_runs_equal_one_run(lbm, digests, kernel_form):
    p, obst, _ = load_case(lbm, digests, "rand_64x48")
    a = lbm.Simulation(p, obst)
    av_a = a.run(25)
    b = lbm.Simulation(p, obst)
    av_b = np.concatenate([b.run(10), b.run(1), b.run(14)])
    assert np.array_equal(bits(a.local_cells()), bits(b.local_cells()))
    assert np.array_equal(av_a, av_b)
    assert b.run(0).size == 0
    a.close(); b.close()


def test_runs_are_deterministic(lbm, digests):
    p, obst, _ = load_case(lbm, digests, "synth_512x512_t100")
    res = []
    for _ in range(2):
        s = lbm.Simulation(p, obst)
        res.append((s.run(60), s.local_cells()))
        s.close()
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(bits(res[0][1]), bits(res[1][1]))


@pytest.mark.parametrize("flags", [1, 2])
def test_store_policy_does_not_change_results(lbm, oracle, digests, flags):
    p, obst, _ = load_case(lbm, digests, "walls_40x24")
    s = lbm.Simulation(p, obst, flags=flags)
    s.run(50)
    ref_cells, _, _ = oracle.run(p, obst, 50)
    assert np.array_equal(bits(s.local_cells()), bits(ref_cells))
    s.close()


def test_set_cells_random_state(lbm, oracle, kernel_form):
    rng = np.random.default_rng(11)
    p = lbm.Params(48, 20, 8, 4, 0.1, 0.02, 1.6)
    obst = lbm.synthetic_obstacles(48, 20, 0.08, 9, False)
    cells0 = (rng.random((20, 48, 9), dtype=np.float32) * 0.02 + 0.004).astype(np.float32)
    part = lbm.Partition(p, lbm.count_free_cells(obst), obst)
    part.set_cells(cells0)
    assert np.array_equal(bits(part.get_cells()), bits(cells0))
    av = part.run(8)
    ref_cells, ref_av = oracle.run_from(p, obst, cells0, 8)
    assert np.array_equal(bits(part.get_cells()), bits(ref_cells))
    assert np.max(np.abs(av - ref_av) / ref_av) < AV_EXACT_RTOL
    part.close()


def test_short_reciprocal_equals_the_division_on_every_float(tmp_path):
    """relax_core's 1.0f / density (d2q9-bgk.c:561) is v_rcp_f32 + one Newton step wherever the result is a normal number and
    the compiler's IEEE division elsewhere (kernels/exact_math.h recip_exact, included by the enumeration program).  That the two agree is not an argument but a
    count: all 2^32 bit patterns, on this GPU, every run of the suite (about a second)."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "experiments", "recip_exhaustive.hip")
    exe = str(tmp_path / "recip_exhaustive")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", src, "-o", exe], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 of them differ" in r.stdout, r.stdout
    # ... and the shipped functions themselves (kernels/exact_math.h, the header common.h includes), scalar and packed
    assert "shipped recip_exact (kernels/exact_math.h), all 2^32 bit patterns: scalar form 0 differ, packed form 0 differ" in r.stdout, r.stdout


def test_short_double_sqrt_is_correctly_rounded_on_every_float(tmp_path):
    """sqrt((double)u_sq) of d2q9-bgk.c:667 is v_rsq_f64 + two Newton corrections (kernels/exact_math.h sqrt_of_float, included by the enumeration program), four
    instructions fewer than the compiler's sequence.  Enumerated against the correctly rounded square root on all
    2 139 095 041 non-negative floats, every run of the suite."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "experiments", "sqrt_exhaustive.hip")
    exe = str(tmp_path / "sqrt_exhaustive")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", src, "-o", exe], check=True, capture_output=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if "SHIPPED kernels/exact_math.h sqrt_of_float" in l or "<- the shipped form" in l]
    assert len(line) == 2 and all(l.split()[-1] == "0" for l in line), r.stdout


@pytest.mark.parametrize("form", ["vector", "tile", "multi"])
@pytest.mark.parametrize("steps", [1, 3])
def test_cells_outside_the_short_reciprocal(lbm, oracle, monkeypatch, form, steps):
    """Densities the short reciprocal does not cover — zero, denormal, above 2^126 — send their wave down the division:
    a 3x3 block of populations 2^123 (density 9 * 2^123 > 2^126: reciprocal denormal, momentum exactly zero, so the centre stays finite), a block of
    zeros (1/0) and a block of denormals (1/x overflows).  Bits equal the oracle's wherever it holds a number; where it
    holds a NaN the device holds one too."""
    if form == "vector":
        monkeypatch.setenv("LBM_TUNE_NARROW_MAX", "0")
    if form != "tile":
        monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    if form != "multi":
        monkeypatch.setenv("LBM_TUNE_MULTI_K", "0")
    nx, ny = 64, 32
    rng = np.random.default_rng(5)
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.02, 1.6)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, 3, False)
    cells0 = (rng.random((ny, nx, 9), dtype=np.float32) * 0.02 + 0.004).astype(np.float32)
    cells0[10:13, 10:13, :] = np.float32(2.0 ** 123)
    cells0[19:22, 39:42, :] = np.float32(0.0)
    cells0[25:28, 20:23, :] = np.float32(1e-40)
    obst[9:14, 9:14] = 0
    obst[18:23, 38:43] = 0
    obst[24:29, 19:24] = 0
    part = lbm.Partition(p, lbm.count_free_cells(obst), obst)
    part.set_cells(cells0)
    with np.errstate(all="ignore"):
        part.run(steps)
        got = part.get_cells()
        ref_cells, _ = oracle.run_from(p, obst, cells0, steps)
    if form == "multi":
        assert "lbm_multi_kernel" in part.describe()["kernel"]
    if form == "tile":
        assert "lbm_tile_kernel" in part.describe()["kernel"]
    nan_ref, nan_got = np.isnan(ref_cells), np.isnan(got)
    assert nan_ref.any() and not nan_ref.all()
    assert np.array_equal(nan_ref, nan_got)
    assert np.array_equal(bits(got)[~nan_got], bits(ref_cells)[~nan_ref])
    if steps == 1:          # the centre of the 2^123 block went through the division with a denormal reciprocal and stayed finite
        assert np.all(np.isfinite(got[11, 11])) and np.all(np.abs(got[11, 11]) > 1e35)
    part.close()


@pytest.mark.parametrize("nx,ny", [(4, 3), (8, 3), (4, 64), (2048, 3), (12, 7), (36, 5), (1028, 6)])
def test_odd_shapes(lbm, oracle, kernel_form, nx, ny):
    """nx only needs to be a multiple of 4 here (the reference silently needs 8, d2q9-bgk.c:453,520);
    rows of any count >= 3; tiles that straddle rows."""
    p = lbm.Params(nx, ny, 30, 4, 0.1, 0.01, 1.4)
    obst = lbm.synthetic_obstacles(nx, ny, 0.1, nx * 31 + ny, False)
    if obst.all():
        obst[0, 0] = 0
    s = lbm.Simulation(p, obst)
    av = s.run(30)
    ref_cells, _, ref_exact = oracle.run(p, obst, 30)
    assert np.array_equal(bits(s.local_cells()), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < AV_EXACT_RTOL
    s.close()


def test_device_av_velocity_matches_host(lbm, digests):
    p, obst, free = load_case(lbm, digests, "rand_64x48")
    s = lbm.Simulation(p, obst)
    s.run(200)
    cells = s.local_cells()
    host = lbm.av_velocity_host(p, cells, obst)
    dev = s.partition.av_velocity_sum()
    assert abs(dev - host) / host < 1e-5
    s.close()


def _ring_exchange(parts):
    """The halo exchange of d2q9-bgk.c:295-313 between partitions that live in one process:
    southward messages land in the southern neighbour's north halo and vice versa."""
    n = len(parts)
    for r, part in enumerate(parts):
        parts[(r - 1) % n].halo_recv(lbm_NORTH).copy_(part.halo_send(lbm_SOUTH))
        parts[(r + 1) % n].halo_recv(lbm_SOUTH).copy_(part.halo_send(lbm_NORTH))


lbm_SOUTH, lbm_NORTH = 0, 1


@pytest.mark.parametrize("case,size", [("rand_64x48", 2), ("rand_64x48", 3), ("rand_64x48", 5), ("walls_40x24", 8),
                                       ("tall_8x256", 64), ("wide_256x8", 2), ("synth_512x512_t100", 8)])
def test_row_partitioned_stepping_equals_single_partition(lbm, oracle, digests, kernel_form, case, size):
    """Split-phase C ABI (interior / boundary / halo buffers) with `size` partitions of one grid on
    one GPU, exchanged by device copies: must be bit-identical to the single-partition run, and the
    summed per-step tot_u must match."""
    import torch
    p, obst, free = load_case(lbm, digests, case)
    steps = min(p.max_iters, 60)
    ny_local, displs = lbm.decompose(p.ny, size)
    dev = torch.device("cuda", 0)
    parts = []
    for r in range(size):
        part = lbm.Partition(p, free, obst[displs[r]:displs[r] + ny_local[r]], displs[r])
        part.bind_halo_tensors(dev)
        parts.append(part)
    torch.cuda.synchronize()
    tstream = torch.cuda.Stream(dev)         # every kernel and every halo copy on ONE explicit stream
    stream = tstream.cuda_stream
    with torch.cuda.stream(tstream):
        for part in parts:
            part.step_prepare(steps, stream)
        for _ in range(steps):
            _ring_exchange(parts)
            for part in parts:
                part.step_interior(stream)
                part.step_boundary(stream)
                part.step_finish(stream)
        sums = sum(part.step_collect(steps, stream) for part in parts)
    tstream.synchronize()
    cells = np.concatenate([part.get_cells() for part in parts], axis=0)
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    av = sums * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < 1e-12
    for part in parts:
        part.close()


def test_full_size_8192_properties_and_short_parity(lbm, oracle):
    """BASELINE.json config 5 (synthetic 8192x8192, p=0.005, seed 42, walls).  Size-independent
    properties: mass is conserved by stream + bounce-back + BGK + accelerate (all exchange mass
    between populations only), obstacle interiors never change, av_vels is positive and grows
    from rest; plus a short bit-exact comparison with the (multi-threaded) oracle."""
    n = 8192
    p = lbm.Params(n, n, 50, 10, 0.1, 0.005, 1.85)
    obst = lbm.synthetic_obstacles(n, n, 0.005, 42, True)
    s = lbm.Simulation(p, obst)
    av = s.run(50)
    cells = s.local_cells()
    s.close()
    mass = cells.sum(dtype=np.float64)
    mass0 = np.float64(n) * n * (np.float64(np.float32(0.1) * np.float32(4.0) / np.float32(9.0))
                                 + 4 * np.float64(np.float32(0.1) / np.float32(9.0))
                                 + 4 * np.float64(np.float32(0.1) / np.float32(36.0)))
    assert abs(mass - mass0) / mass0 < 1e-6
    assert np.all(av > 0) and np.all(np.diff(av) > 0)
    ref_cells, _, ref_exact = oracle.run(p, obst, 50, nthreads=min(os.cpu_count() or 8, 16))
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL
    del cells
    # the same deck as a K-step row partition on a 1-rank ring (what each rank of an N-GPU run executes), both native loops;
    # the state digest of the ring equals the digest of the plain run (lbm_state_checksum: 8 bytes instead of 2.4 GB)
    digest = None
    for exchange in ("p2p", "rccl"):
        ring = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange=exchange, strict=True)
        assert ring.partition.macro_steps >= 2 and ring.loop == exchange
        av_ring = np.concatenate([ring.run(31), ring.run(19)])
        assert np.max(np.abs(av_ring - ref_exact) / ref_exact) < AV_EXACT_RTOL
        if digest is None:
            assert np.array_equal(bits(ring.local_cells()), bits(ref_cells))
            digest = ring.partition.checksum()
            half = ring.partition.checksum(0, n // 2) + ring.partition.checksum(n // 2, n)
            assert half % (1 << 64) == digest                      # additive over disjoint row ranges
        else:
            assert ring.partition.checksum() == digest
        ring.close()
    # ... and as one rank of the tile (2-D) decomposition at full size: a column block (no ghost rows: the launches wrap in y, only columns
    # travel) and a block of any tiling (ghost rows and columns; rows pushed onto the rank itself, corners in two hops), under both schedules
    for ghost_rows, schedule in (("", ""), ("1", "serial"), ("1", "edge")):           # (ranks of >= 2^25 cells take the edge-stream schedule by default)
        os.environ.pop("LBM_TUNE_TILE_GHOST_ROWS", None)
        os.environ.pop("LBM_P2P_SCHEDULE", None)
        if ghost_rows:
            os.environ["LBM_TUNE_TILE_GHOST_ROWS"] = ghost_rows
        if schedule:
            os.environ["LBM_P2P_SCHEDULE"] = schedule
        try:
            tile = lbm.Simulation(p, obst, exchange="p2p", strict=True, rank_grid=(1, 1))
            what = tile.describe()["p2p"]
            assert ("rows wrap in the launch" in what) == (not ghost_rows) and ("edge stream" in what) == (schedule != "serial"), what
            av_tile = np.concatenate([tile.run(31), tile.run(19)])
            assert np.max(np.abs(av_tile - ref_exact) / ref_exact) < AV_EXACT_RTOL
            assert tile.partition.checksum() == digest, what
            tile.close()
        finally:
            os.environ.pop("LBM_TUNE_TILE_GHOST_ROWS", None)
            os.environ.pop("LBM_P2P_SCHEDULE", None)


def test_mid_size_longer_run(lbm, oracle):
    """2048x2048 synthetic deck, 300 steps: lbm_multi_kernel on a grid well past the caches."""
    n = 2048
    p = lbm.Params(n, n, 300, 10, 0.1, 0.005, 1.85)
    obst = lbm.synthetic_obstacles(n, n, 0.005, 42, True)
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"].startswith("lbm_multi_kernel")
    av = s.run(300)
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, 300, nthreads=min(os.cpu_count() or 8, 16))
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("case", ["rand_64x48", "walls_40x24", "synth_512x512_t100"])
def test_rccl_ring_of_one_rank(lbm, oracle, digests, kernel_form, case):
    """The native RCCL step loop (liblbm_d2q9_rccl.so) on a 1-rank ring: the rank sends its edge rows
    to itself through RCCL on the side stream, exactly the reference's 1-rank behaviour
    (d2q9-bgk.c:245-247).  Exercises communicator set-up, the event ordering between the exchange
    and the interior / boundary kernels, and the end-of-run collect."""
    p, obst, free = load_case(lbm, digests, case)
    steps = min(p.max_iters, 80)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="rccl")
    av = np.concatenate([sim.run(steps // 2), sim.run(steps - steps // 2)])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("exchange", ["rccl", "torch"])
def test_distributed_world_of_one_over_nccl(lbm, oracle, digests, tmp_path, exchange):
    """torch.distributed with the nccl (= RCCL) backend, world size 1: both exchange back ends run
    their real code path (unique-id broadcast / batch_isend_irecv to self, stream ordering)."""
    import torch
    import torch.distributed as dist
    p, obst, free = load_case(lbm, digests, "rand_64x48")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, distributed=True, exchange=exchange)
        av = sim.run(60)
        cells = sim.gather_cells()
        sim.close()
    finally:
        dist.destroy_process_group()
    ref_cells, _, ref_exact = oracle.run(p, obst, 60)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


def test_graph_replay_equals_direct_launches(lbm, digests, kernel_form):
    """With LBM_FLAG_GRAPH lbm_run replays 64-step hipGraphs; by default it launches every step.
    Both must give the same bits, for step counts around the block size and across calls."""
    p, obst, _ = load_case(lbm, digests, "rand_64x48")
    for steps in (1, 2, 64, 65, 66, 129, 130, 300):
        a = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_GRAPH)
        b = lbm.Simulation(p, obst)
        av_a = np.concatenate([a.run(steps), a.run(131)])
        av_b = np.concatenate([b.run(steps), b.run(131)])
        assert np.array_equal(av_a, av_b), steps
        assert np.array_equal(bits(a.local_cells()), bits(b.local_cells())), steps
        a.close(); b.close()


def test_experiment_build_passes_its_suite(lbm):
    """The forms that measured slower and are no longer in liblbm_d2q9.so — lbm_sweep_kernel (LBM_TUNE_SWEEP) and the LDS-staged one-step
    kernel (LBM_FLAG_KERNEL_LDS) — live on behind -DLBM_EXPERIMENTS=1: this builds that variant of the library
    (scripts/build_variant.sh experiments; a copy built in the build container travels with the tree and is reused when it is newer than
    the sources) and runs their parity tests (tests/experiments_suite.py, against the oracle bit for bit) ONCE, in one process of its own.
    The shipped library refuses the flag instead of ignoring it."""
    import sys
    from conftest import ROOT
    p = lbm.Params(64, 32, 4, 4, 0.1, 0.01, 1.7)
    with pytest.raises(lbm.LbmError, match="LBM_EXPERIMENTS"):
        lbm.Partition(p, 64 * 32, np.zeros((32, 64), np.int32), flags=lbm._capi.FLAG_KERNEL_LDS)
    variant = lbm.build(experiments=True)["lib_experiments"]          # reused when newer than the sources (built by __graft_entry__.build())
    env = dict(os.environ, LBM_LIBRARY=variant)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "experiments_suite.py"), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, (r.stdout[-3000:], r.stderr[-2000:])


@pytest.mark.parametrize("nx,ny", [(1, 4), (2, 3), (5, 3), (7, 9), (18, 6), (30, 11), (1023, 4), (257, 33)])
def test_row_lengths_not_multiple_of_four(lbm, oracle, nx, ny):
    """Any nx works (one-cell-per-lane form); the reference silently needs nx % 8 == 0 (d2q9-bgk.c:453,520)."""
    p = lbm.Params(nx, ny, 25, 4, 0.1, 0.01, 1.6)
    obst = lbm.synthetic_obstacles(nx, ny, 0.1, nx * 131 + ny, False)
    if obst.all():
        obst[0, 0] = 0
    s = lbm.Simulation(p, obst)
    assert "narrow" in s.partition.describe()["kernel"]
    av = s.run(25)
    ref_cells, _, ref_exact = oracle.run(p, obst, 25)
    assert np.array_equal(bits(s.local_cells()), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < AV_EXACT_RTOL
    s.close()


@pytest.mark.parametrize("name", ["rand_64x48", "walls_40x24", "accelrow_blocked_32x16", "strongaccel_32x16", "synth_512x512_t100"])
def test_narrow_kernel_on_regular_decks(lbm, oracle, digests, monkeypatch, name):
    monkeypatch.setenv("LBM_TUNE_NARROW_MAX", str(1 << 30))
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MULTI_K", "0")
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "0")
    p, obst, free = load_case(lbm, digests, name)
    steps = min(p.max_iters, 150)
    s = lbm.Simulation(p, obst)
    assert "narrow" in s.partition.describe()["kernel"]
    av = s.run(steps)
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL
    # and in a ring (interior / boundary launches, halo messages written by scalar stores)
    ring = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="rccl")
    av2 = ring.run(steps)
    assert np.array_equal(bits(ring.local_cells()), bits(ref_cells))
    assert np.max(np.abs(av2 - ref_exact) / ref_exact) < AV_EXACT_RTOL
    ring.close()


def test_python_cli_under_torchrun_world_of_one(lbm, digests, tmp_path):
    """d2q9_bgk.py (the N-GPU front end with the reference's CLI contract) launched the way the
    N > 1 case is, with one rank: same stdout lines and byte-identical final_state.dat."""
    import sys
    from conftest import ROOT
    name = "128x256_t2000"
    ppath, opath = deck_paths(name, digests)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "d2q9_bgk.py"), ppath, opath]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    i = out.index("==done==")
    assert out[i + 1] == digests[name]["reynolds_line"]
    assert out[i + 2].startswith("Elapsed time:\t\t\t")
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    av = lbm.checker.load_av_vels(str(tmp_path / "av_vels.dat"))
    assert np.allclose(av[np.asarray(digests[name]["av_sample_steps"])], digests[name]["av_sample_values"], rtol=5e-4)


@pytest.mark.parametrize("nx,ny,steps", [(16, 16, 37), (32, 16, 8), (16, 48, 9), (64, 32, 1), (128, 128, 100), (512, 256, 23)])
@pytest.mark.parametrize("geom", ["168", "164", "88", "84"])
def test_tile_kernel_shapes_and_step_counts(lbm, oracle, monkeypatch, geom, nx, ny, steps):
    """lbm_tile_kernel: grids smaller than the 32x32 region (the region then wraps around the domain
    more than once), step counts that are not multiples of 8, repeated calls."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", str(1 << 30))
    monkeypatch.setenv("LBM_TUNE_TILE_GEOM", geom)     # T*10 + H: owned tile edge, ghost ring / steps per launch
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.02, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.07, nx * 7 + ny, False)
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"].startswith("lbm_tile_kernel<")
    av = np.concatenate([s.run(steps), s.run(5), s.run(8)])
    ref_cells, _, ref_exact = oracle.run(p, obst, steps + 13)
    assert np.array_equal(bits(s.local_cells()), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL
    s.close()


@pytest.mark.parametrize("ranks,name,exchange", [(2, "128x256_t2000", "torch"), (3, "256x256_t1000", "torch"),
                                                 (2, "256x256_t1000", "p2p"), (3, "1024x1024_t200", "p2p"),
                                                 # the four shipped decks in full (BASELINE.json configs) as multi-rank runs
                                                 (2, "128x128", "p2p"), (3, "128x256", "p2p"), (4, "256x256", "p2p"), (4, "1024x1024", "p2p")])
def test_several_ranks_share_the_gpu_over_gloo(lbm, digests, tmp_path, ranks, name, exchange):
    """The multi-process row-partitioned run end to end — torchrun, one process per rank, each with
    its own HIP partition, neighbour exchange, final reduction, rank-0 gather and output — on ONE GPU.
    RCCL refuses ranks that share a device, so the process group is gloo and the halos travel either
    staged through the host ("torch": HaloExchange.host_staged, one-step loop) or by the native
    peer-to-peer loop ("p2p": every rank maps its neighbours' grids with hipIpcOpenMemHandle and stores
    its edge rows into their ghost rows — the N-GPU code path except for the xGMI wire).  Only rank 0
    parses the files; the obstacle rows are scattered."""
    import sys
    from conftest import ROOT
    ppath, opath = deck_paths(name, digests)
    env = dict(os.environ, LBM_FORCE_DEVICE="0", LBM_DIST_BACKEND="gloo", LBM_EXCHANGE=exchange, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "d2q9_bgk.py"), ppath, opath]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert out[out.index("==done==") + 1] == digests[name]["reynolds_line"]
    assert f"{exchange} loop" in [l for l in out if l.startswith("MLUPS")][0]
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    av = lbm.checker.load_av_vels(str(tmp_path / "av_vels.dat"))
    assert np.allclose(av[np.asarray(digests[name]["av_sample_steps"])], digests[name]["av_sample_values"], rtol=5e-4)


@pytest.mark.parametrize("ranks,grid,name", [(4, "2x2", "1024x1024"), (2, "2x1", "256x256")])
def test_rank_processes_run_a_tile_decomposition_of_the_shipped_decks(lbm, digests, tmp_path, ranks, grid, name):
    """LBM_RANK_GRID under torch.distributed.run: the shipped decks IN FULL (20 000 / 80 000 steps) over the tile (2-D) decomposition, one
    process per rank sharing this GPU over IPC — rank 0 parses, scatters windows of rows and columns, gathers blocks — and the files the
    reference binary writes come out: final_state.dat byte for byte, the Reynolds line as a string."""
    import sys
    from conftest import ROOT
    ppath, opath = deck_paths(name, digests)
    env = dict(os.environ, LBM_FORCE_DEVICE="0", LBM_DIST_BACKEND="gloo", LBM_RANK_GRID=grid, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LBM_EXCHANGE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "d2q9_bgk.py"), ppath, opath]
    r = subprocess.run(cmd, cwd=tmp_path, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [l for l in r.stdout.splitlines() if l.strip()]
    assert out[out.index("==done==") + 1] == digests[name]["reynolds_line"]
    assert f"p2p loop, {grid[0]} x {grid[2]} tiles" in [l for l in out if l.startswith("MLUPS")][0]
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    av = lbm.checker.load_av_vels(str(tmp_path / "av_vels.dat"))
    assert np.allclose(av[np.asarray(digests[name]["av_sample_steps"])], digests[name]["av_sample_values"], rtol=5e-4)


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("name,steps", [("synth_512x512_t100", 100), ("synth_512x512_t100", 37), ("1024x1024_t200", 200),
                                        ("128x128", 41), ("rand_64x48", 103)])
def test_k_steps_per_pass_kernel(lbm, oracle, digests, monkeypatch, name, steps, K):
    """lbm_multi_kernel<K>: a 64x16 tile advanced by up to K steps per launch (ring recomputed
    redundantly, intermediate states in LDS, in place); step counts that are not multiples of K end
    with a shorter launch.  Must be bit-identical to the one-step path."""
    monkeypatch.setenv("LBM_TUNE_MULTI_K", str(K))
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    p, obst, free = load_case(lbm, digests, name)
    assert p.nx % 64 == 0 and p.ny % 16 == 0
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"] == f"lbm_multi_kernel<{K}>"
    av = np.concatenate([s.run(steps), s.run(3)])
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps + 3, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("steps", [1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14])
def test_run_lengths_that_four_does_not_divide(lbm, oracle, digests, monkeypatch, steps):
    """lbm_run at K = 4 (64 x 13 tiles; the choice from 1 M cells up) splits a step count 4 does not divide into 4s and 3s
    (n = 4a + 3, 4a + 6, 4a + 9): launches of different tile heights alternate inside one run, each folding the other's sums."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MULTI_K", "4")
    p, obst, free = load_case(lbm, digests, "synth_512x512_t100")
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"] == "lbm_multi_kernel<4>"
    av = np.concatenate([s.run(steps), s.run(steps)])
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, 2 * steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert av.shape == (2 * steps,) and np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("geom,nx,ny,steps", [(None, 1346, 811, 11), (None, 1024, 1030, 9), ("2", 322, 140, 14), ("2", 128, 57, 10), ("0", 1154, 930, 7),
                                              ("1", 1346, 811, 6)])
def test_multi_kernel_geometries_on_ragged_grids(lbm, oracle, monkeypatch, geom, nx, ny, steps):
    """The three launch geometries of lbm_multi_kernel<4> (kernels/multi.h: standard 64 x 13 / 512 lanes, narrow 32 x 13, tall 64 x 23 /
    768 lanes — the library's choice from 2^20 cells up) on grids that no tile size divides: last tile column and row stick out of
    the grid, step counts that 4 does not divide mix in the K = 3 launch (64 x 16), the run is cut in two.  Against the oracle."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MULTI_K", "4")
    if geom is not None:
        monkeypatch.setenv("LBM_TUNE_MULTI_GEOM", geom)
    p = lbm.Params(nx, ny, steps, 10, 0.1, 0.005, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.02, nx * 7 + ny, True)
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"] == "lbm_multi_kernel<4>"
    av = np.concatenate([s.run(steps - 4), s.run(4)])
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < AV_EXACT_RTOL


@pytest.mark.parametrize("steps", [1, 2, 4, 5, 7, 8, 10, 11, 13])
@pytest.mark.parametrize("tail4", ["1", "0"])
def test_run_lengths_that_three_does_not_divide(lbm, oracle, digests, monkeypatch, steps, tail4):
    """lbm_run at K = 3 splits a step count into 3s and 4s where that avoids a 1- or 2-step launch (n = 3a + 4 or
    3a + 8; LBM_TUNE_MULTI_TAIL4=0: the plain min(3, left) split).  Either way the state after n steps and the n
    av_vels are the one-step path's."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MULTI_TAIL4", tail4)
    p, obst, free = load_case(lbm, digests, "synth_512x512_t100")
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"] == "lbm_multi_kernel<3>"
    av = np.concatenate([s.run(steps), s.run(steps)])
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, 2 * steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert av.shape == (2 * steps,) and np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


def _k_step_partitions_in_process(lbm, parts, steps, K):
    """Several K-step partitions of one grid on one GPU, ghost rows exchanged by device copies
    (lbm_macro_exchange_local) in the order of the native loop; returns the summed per-step tot_u."""
    import torch
    size = len(parts)
    tstream = torch.cuda.Stream(torch.device("cuda", 0))
    st = tstream.cuda_stream
    with torch.cuda.stream(tstream):
        for part in parts:
            part.macro_prepare(steps, st)
        done = 0
        while done < steps:
            for r, part in enumerate(parts):
                part.macro_receive_from(parts[(r - 1) % size], lbm.NORTH, st)    # southern neighbour's top rows
                part.macro_receive_from(parts[(r + 1) % size], lbm.SOUTH, st)    # northern neighbour's bottom rows
            for part in parts:
                part.macro_interior(st)
                part.macro_edge(st)
            k = parts[0].macro_next                                              # steps until the next exchange; the same on every partition
            assert 1 <= k <= 32 and all(part.macro_next == k for part in parts)
            for part in parts:
                part.macro_finish(st)
            done += k
        sums = sum(part.step_collect(steps, st) for part in parts)
    tstream.synchronize()
    return sums


def _stress_deck(lbm, kind, nx, ny):
    """Decks that take the rare paths of lbm_multi_kernel: no obstacle anywhere (the bounce-back branch
    is never entered), every second cell blocked (always entered, also on the accelerate row), row
    ny-2 walled off (accelerate_flow finds no free cell), an acceleration so strong that the
    positivity test of d2q9-bgk.c:461-463 fails for most cells, relaxation at both ends of the range."""
    omega, accel, density = 1.85, 0.005, 0.1
    obst = np.zeros((ny, nx), np.int32)
    if kind == "open":
        pass
    elif kind == "dense":
        obst = lbm.synthetic_obstacles(nx, ny, 0.5, 3, False)
    elif kind == "accel_row_blocked":
        obst = lbm.synthetic_obstacles(nx, ny, 0.01, 5, True)
        obst[ny - 2, :] = 1
    elif kind == "strong_accel":
        obst = lbm.synthetic_obstacles(nx, ny, 0.02, 9, False)
        accel = 0.9
    elif kind == "omega_low":
        obst = lbm.synthetic_obstacles(nx, ny, 0.02, 11, True)
        omega, density = 0.6, 1.0
    elif kind == "omega_high":
        obst = lbm.synthetic_obstacles(nx, ny, 0.02, 13, True)
        omega, accel = 1.99, 0.01
    if obst.all():
        obst[1, 1] = 0
    return lbm.Params(nx, ny, 40, 8, density, accel, omega), obst


STRESS = STRESS_KINDS


@pytest.mark.parametrize("K", [2, 3, 4])
@pytest.mark.parametrize("tile", [64, 32])
@pytest.mark.parametrize("kind", STRESS)
def test_multi_kernel_rare_paths(lbm, oracle, monkeypatch, kind, K, tile):
    """Whole periodic grid (edge tiles only at 256x64, inner tiles too at 448x112) through
    lbm_multi_kernel<K>: uniform branches for accelerate_flow / bounce-back / inner tiles, pair flags;
    both tile widths (64 x 16, and the 32 x 16 form small partitions get)."""
    monkeypatch.setenv("LBM_TUNE_MULTI_K", str(K))
    monkeypatch.setenv("LBM_TUNE_MULTI_TILE", str(tile))
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    for nx, ny in ((256, 64), (448, 112)):
        p, obst = _stress_deck(lbm, kind, nx, ny)
        s = lbm.Simulation(p, obst)
        assert s.partition.describe()["kernel"] == f"lbm_multi_kernel<{K}>"
        av = np.concatenate([s.run(29), s.run(11)])
        cells = s.local_cells()
        s.close()
        ref_cells, _, ref_exact = oracle.run(p, obst, 40, nthreads=4)
        assert np.array_equal(bits(cells), bits(ref_cells)), (kind, nx, ny)
        assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < AV_EXACT_RTOL


@pytest.mark.parametrize("tile", [64, 32])
@pytest.mark.parametrize("kind", STRESS)
def test_k_step_partitions_rare_paths(lbm, oracle, monkeypatch, kind, tile):
    """The same decks as K-step row partitions exchanged in-process (3 partitions of 448x112: the
    accelerate row lies in one partition's owned rows and in its neighbour's ghost rows)."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "3")
    monkeypatch.setenv("LBM_TUNE_MULTI_TILE", str(tile))
    p, obst = _stress_deck(lbm, kind, 448, 112)
    ny_local, displs = lbm.decompose(p.ny, 3)
    free = int(obst.size - obst.sum())
    parts = [lbm.Partition(p, free, obst[displs[r]:displs[r] + ny_local[r]], displs[r], obstacles_global=obst) for r in range(3)]
    assert all(q.macro_steps == 3 for q in parts)
    sums = _k_step_partitions_in_process(lbm, parts, 40, 3)
    cells = np.concatenate([q.get_cells() for q in parts], axis=0)
    for q in parts:
        q.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, 40, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells)), kind
    av = sums * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < 1e-12


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("name,steps", [("synth_512x512_t100", 61), ("128x128", 50), ("1024x1024_t200", 30)])
def test_k_step_mode_on_a_ring_of_one(lbm, oracle, digests, monkeypatch, name, steps, K):
    """K-step mode of a row-partitioned run (lbm_create_global -> K ghost rows, lbm_macro_*): the native
    RCCL loop on a 1-rank ring, the rank's ghost rows filled from its own edge rows over RCCL."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="rccl")
    assert sim.partition.macro_steps == K
    av = np.concatenate([sim.run(steps), sim.run(5)])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps + 5, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("name,size,K", [("synth_512x512_t100", 2, 2), ("synth_512x512_t100", 4, 3), ("synth_512x512_t100", 8, 4),
                                         ("synth_512x512_t100", 16, 2), ("1024x1024_t200", 8, 4), ("128x128", 4, 1)])
def test_k_step_mode_with_several_partitions(lbm, oracle, digests, monkeypatch, name, size, K):
    """Several K-step partitions of one grid on one GPU, ghost rows exchanged by device copies
    (lbm_macro_exchange_local) in the order of the native loop: must equal the single-partition run."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    p, obst, free = load_case(lbm, digests, name)
    steps = min(p.max_iters, 45)
    ny_local, displs = lbm.decompose(p.ny, size)
    parts = [lbm.Partition(p, free, obst[displs[r]:displs[r] + ny_local[r]], displs[r], obstacles_global=obst) for r in range(size)]
    assert all(part.macro_steps == K for part in parts)
    sums = _k_step_partitions_in_process(lbm, parts, steps, K)
    cells = np.concatenate([part.get_cells() for part in parts], axis=0)
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    av = sums * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < 1e-12
    for part in parts:
        part.close()


@pytest.mark.parametrize("K", [2, 3, 4])
@pytest.mark.parametrize("nx,ny,tile", [(128, 32, 64), (130, 34, 64), (200, 150, 64), (1000, 1000, 64), (1026, 258, 64), (190, 47, 64),
                                        (130, 34, 32), (200, 150, 32), (1000, 1000, 32), (190, 47, 32)])
def test_k_steps_kernel_on_grids_that_tiles_do_not_divide(lbm, oracle, monkeypatch, nx, ny, K, tile):
    """Even nx >= 128, ny >= 32: the last tile column / row of lbm_multi_kernel sticks out of the grid;
    the overhanging cells are periodic images that are computed but neither stored nor summed."""
    monkeypatch.setenv("LBM_TUNE_MULTI_K", str(K))
    monkeypatch.setenv("LBM_TUNE_MULTI_TILE", str(tile))
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    steps = 40 if nx * ny > 500000 else 95
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx * 3 + ny, False)
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"] == f"lbm_multi_kernel<{K}>"
    av = np.concatenate([s.run(steps), s.run(2)])
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps + 2, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("nx,ny,size,K", [(256, 200, 3, 2), (256, 200, 3, 3), (130, 100, 2, 4), (1000, 1000, 8, 3),
                                          (192, 99, 2, 3), (128, 260, 8, 4), (512, 70, 2, 2), (256, 131, 4, 3)])
def test_k_step_partitions_with_rows_the_tile_does_not_divide(lbm, oracle, monkeypatch, nx, ny, size, K):
    """Row partitions with arbitrary row counts (the reference's decomposition gives 125 rows per rank
    for a 1000-row grid on 8 ranks): the last tile row sticks out past the ghost rows and the top edge
    launch covers two tile rows when the last one is thinner than K."""
    import torch
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    steps = 31
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx * 5 + ny, False)
    free = lbm.count_free_cells(obst)
    ny_local, displs = lbm.decompose(ny, size)
    parts = [lbm.Partition(p, free, obst[displs[r]:displs[r] + ny_local[r]], displs[r], obstacles_global=obst) for r in range(size)]
    assert all(part.macro_steps == K for part in parts), [part.macro_steps for part in parts]
    tstream = torch.cuda.Stream(torch.device("cuda", 0))
    st = tstream.cuda_stream
    with torch.cuda.stream(tstream):
        for part in parts:
            part.macro_prepare(steps, st)
        done = 0
        while done < steps:
            for r, part in enumerate(parts):
                part.macro_receive_from(parts[(r - 1) % size], lbm.NORTH, st)
                part.macro_receive_from(parts[(r + 1) % size], lbm.SOUTH, st)
            for part in parts:
                part.macro_interior(st)
                part.macro_edge(st)
            k = parts[0].macro_next                                              # steps until the next exchange; the same on every partition
            assert 1 <= k <= 32 and all(part.macro_next == k for part in parts)
            for part in parts:
                part.macro_finish(st)
            done += k
        sums = sum(part.step_collect(steps, st) for part in parts)
    tstream.synchronize()
    cells = np.concatenate([part.get_cells() for part in parts], axis=0)
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    av = sums * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < 1e-12
    for part in parts:
        part.close()


# (ghost rows, most launches per exchange): rounds 1-3's loop (K rows, an exchange before every launch); the default (2 K rows, two
# launches); deeper groups; fewer ghost rows than two launches make steps (single launches on spare rows); a cap below what the rows allow
GROUPINGS = [("0", ""), ("", ""), ("12", ""), ("16", ""), ("7", ""), ("16", "2"), ("8", "1"), ("24", ""), ("32", "")]     # (24 / 32 rows: six / eight launches per exchange, the smallest ranks' default)


@pytest.mark.parametrize("ghost,group", GROUPINGS)
@pytest.mark.parametrize("nx,ny,K,exchange,schedule", [(192, 99, 4, "p2p", "edge"), (130, 100, 3, "p2p", "serial"), (256, 131, 4, "rccl", "edge"),
                                                      (512, 70, 2, "p2p", "edge"), (1024, 300, 4, "p2p", "edge"), (128, 64, 4, "rccl", "serial")])
def test_groups_of_launches_per_exchange_on_a_ring_of_one(lbm, oracle, monkeypatch, nx, ny, K, exchange, schedule, ghost, group):
    """One halo exchange per GROUP of launches (d2q9-bgk.c:326-328,364 once per up to `ghost` steps): the first launch of a group also
    advances the ghost rows the later launches read — rows that belong to the neighbour (here: the rank itself, around the ring) and
    must not enter this rank's per-step sums.  Both native loops, both schedules, row counts no tile height divides, repeated runs
    whose step counts leave groups of every length, against the oracle bit for bit."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    monkeypatch.setenv("LBM_P2P_SCHEDULE", schedule)
    monkeypatch.setenv("LBM_RCCL_SCHEDULE", schedule)
    if ghost:
        monkeypatch.setenv("LBM_TUNE_MACRO_GHOST", ghost)
    if group:
        monkeypatch.setenv("LBM_TUNE_MACRO_GROUP", group)
    runs = [20, 11, 1, 29]
    p = lbm.Params(nx, ny, sum(runs), 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx * 5 + ny, False)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange=exchange, strict=True)
    lay = sim.layout
    two = 8 if K == 3 else 2 * K                         # the default: by the rank's size (lbm_kernels.hip macro_ghost_for)
    by_size = two if nx * ny >= 1 << 21 else max(16 // K * K, two) if ny >= 128 else two if ny >= 64 else (4 if K == 3 else K)
    if nx <= 2048 and nx * ny <= 1 << 19 and ny >= 128:         # the smallest ranks: 24 rows from 128 rows per rank, 32 from 256
        by_size = max(by_size, (32 if ny >= 256 else 24) // K * K)
    want_ghost = max(min(int(ghost), 32), K) if ghost else by_size
    assert sim.loop == exchange and (lay["macro_k"], lay["ghost"]) == (K, want_ghost) and lay["group"] == (int(group) if group else min(max(want_ghost // K, 1), 8))
    av = np.concatenate([sim.run(n) for n in runs])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, sum(runs), nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("ghost,group", GROUPINGS)
@pytest.mark.parametrize("nx,ny,size,K", [(256, 200, 3, 4), (1000, 1000, 8, 4), (192, 99, 2, 3), (128, 260, 8, 4), (2048, 1100, 2, 4)])
def test_groups_of_launches_per_exchange_with_several_partitions(lbm, oracle, monkeypatch, nx, ny, size, K, ghost, group):
    """The same through the split-phase entry points (exchange of ALL ghost rows by device copies; lbm_macro_interior / _edge, then
    lbm_macro_finish, which makes the later launches of the group): uneven partitions, the tall geometry (2048 x 1100 on two ranks),
    per-step sums of each rank over ITS rows only — they must add up to the oracle's to 1e-12."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    if ghost:
        monkeypatch.setenv("LBM_TUNE_MACRO_GHOST", ghost)
    if group:
        monkeypatch.setenv("LBM_TUNE_MACRO_GROUP", group)
    steps = 31
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx * 5 + ny, False)
    free = lbm.count_free_cells(obst)
    lays = [lbm.rank_layout(p, size, r) for r in range(size)]
    parts = [lbm.Partition(p, free, lbm.obstacle_window(obst, lays[r]), rank_of=(r, size)) for r in range(size)]
    assert all(part.macro_steps == K for part in parts)
    sums = _k_step_partitions_in_process(lbm, parts, steps, K)
    cells = np.concatenate([part.get_cells() for part in parts], axis=0)
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    av = sums * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < 1e-12
    for part in parts:
        part.close()


def test_randomised_decks_against_the_oracle(lbm, oracle, monkeypatch):
    """Random shapes (odd nx too), parameters and obstacle densities through the library's default kernel
    choice, against the oracle: the anchor that scripts/fuzz_kernels.py's GPU-vs-GPU cross-check hangs on."""
    rng = np.random.default_rng(2026)
    for case in range(60):
        big = case % 6 == 0
        nx = int(rng.integers(128, 700)) & ~1 if big else int(rng.integers(1, 200))
        ny = int(rng.integers(32, 200)) if big else int(rng.integers(3, 90))
        steps = int(rng.integers(1, 30))
        p = lbm.Params(nx, ny, steps, 4, float(rng.choice([0.1, 1.0])), float(rng.choice([0.005, 0.05, 0.5])),
                       float(rng.choice([0.7, 1.3, 1.85, 1.97])))
        obst = (rng.random((ny, nx)) < float(rng.choice([0.0, 0.01, 0.2]))).astype(np.int32)
        if rng.random() < 0.3:
            obst[ny - 2, :] = 1
        if obst.all():
            obst[0, 0] = 0
        s = lbm.Simulation(p, obst)
        av = s.run(steps)
        cells = s.local_cells()
        kernel = s.partition.describe()["kernel"]
        s.close()
        ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
        assert np.array_equal(bits(cells), bits(ref_cells)), (case, nx, ny, steps, kernel)
        assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < AV_EXACT_RTOL, (case, nx, ny, steps, kernel)


def test_randomised_kernel_cross_check(lbm):
    """scripts/fuzz_kernels.py with a fixed seed: random shapes, decks, K and tile geometries through the
    multi / tile kernels, the 1-rank K-step ring and in-process K-step / one-step partitions, each against
    the one-step kernel on the same deck, bit for bit."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_kernels", os.path.join(ROOT, "scripts", "fuzz_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(["--cases", "80", "--seed", "11"]) == 0


@pytest.mark.parametrize("K", [2, 3, 4])
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
@pytest.mark.parametrize("nx,ny", [(206, 142), (130, 37), (650, 62)])
def test_k_step_ring_message_sizes(lbm, oracle, monkeypatch, nx, ny, K, exchange):
    """K*nx floats per plane and direction travel in the packed halo message; K = 3 with nx = 206 / 130 /
    650 makes that count 2 mod 4 (a float4 copy loop once dropped the last two floats of such a message
    and overran into the first owned row).  Found by scripts/fuzz_kernels.py."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    p = lbm.Params(nx, ny, 20, 4, 0.1, 0.05, 1.85)
    obst = lbm.synthetic_obstacles(nx, ny, 0.05, nx + ny, False)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange=exchange, strict=True)
    assert sim.partition.macro_steps == K
    av = np.concatenate([sim.run(2), sim.run(18)])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, 20, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
@pytest.mark.parametrize("nx,ny", [(200, 150), (130, 37), (1000, 125)])
def test_k_step_ring_of_one_with_odd_rows(lbm, oracle, nx, ny, exchange):
    p = lbm.Params(nx, ny, 40, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.03, nx + ny, False)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange=exchange, strict=True)
    assert sim.partition.macro_steps > 0
    av = sim.run(40)
    ref_cells, _, ref_exact = oracle.run(p, obst, 40, nthreads=4)
    assert np.array_equal(bits(sim.local_cells()), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL
    sim.close()


def test_fallback_to_the_torch_loop_when_the_native_loop_is_unavailable(lbm, oracle, digests, tmp_path, monkeypatch):
    """If liblbm_d2q9_rccl.so cannot be used on any rank, every rank switches to the torch.distributed
    loop (agreed by an all-reduce) instead of failing."""
    import torch
    import torch.distributed as dist
    p, obst, free = load_case(lbm, digests, "rand_64x48")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdv", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    monkeypatch.setattr(lbm._capi, "LIB_RCCL_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(lbm._capi, "_lib_rccl", None)
    try:
        with pytest.warns(UserWarning, match="falling back"):
            sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, distributed=True, exchange="rccl")
        assert sim._ring is None and sim.partition.macro_steps == 0
        av = sim.run(40)
        cells = sim.gather_cells()
        sim.close()
    finally:
        dist.destroy_process_group()
    ref_cells, _, ref_exact = oracle.run(p, obst, 40)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("exchange", ["p2p", "rccl"])
def test_full_1024_deck_as_a_k_step_ring_matches_the_reference_file(lbm, digests, tmp_path, exchange):
    """All 20 000 steps of the shipped 1024x1024 deck through the row-partitioned code path (K-step
    mode, native peer-to-peer / RCCL loop, 1-rank ring): final_state.dat must still be the reference binary's file."""
    name = "1024x1024"
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange=exchange, strict=True)
    assert sim.partition.macro_steps > 0
    av = sim.run()
    cells = sim.local_cells()
    assert "Reynolds number:\t\t%.12E" % sim.reynolds(cells) == digests[name]["reynolds_line"]
    sim.write_values(av, str(tmp_path), cells)
    sim.close()
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    steps = np.asarray(digests[name]["av_sample_steps"])
    assert np.allclose(av[steps], digests[name]["av_sample_values"], rtol=4e-3)


# ---- peer-to-peer halo transport (include/lbm_d2q9_p2p.h) ---------------------------------------------

@pytest.mark.parametrize("name", ["tiny_8x3", "rand_64x48", "walls_40x24", "wide_256x8", "tall_8x256", "column_24x20", "128x128"])
def test_p2p_ring_of_one_in_one_step_mode(lbm, oracle, digests, monkeypatch, name, kernel_form):
    """The peer-to-peer loop for runs that are not eligible for K-step mode (here: forced off, or shapes that never
    are — 3 rows, 8-cell rows, odd sizes): the boundary launch stores its outgoing populations straight into the
    neighbour's window (the rank's own, on a 1-rank ring), one kernel per step raises and awaits the flags."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "0")
    p, obst, free = load_case(lbm, digests, name)
    steps = min(p.max_iters, 70)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="p2p", strict=True)
    assert sim.loop == "p2p" and sim.partition.macro_steps == 0 and "one-step" in sim.describe()["p2p"]
    av = np.concatenate([sim.run(steps - 7), sim.run(7)])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("K", [1, 2, 3, 4])
@pytest.mark.parametrize("schedule", ["edge", "serial"])
@pytest.mark.parametrize("name,steps", [("synth_512x512_t100", 61), ("128x128", 50), ("1024x1024_t200", 31)])
def test_p2p_ring_of_one(lbm, oracle, digests, monkeypatch, name, steps, K, schedule):
    """The peer-to-peer loop on a 1-rank ring: the rank pushes its edge rows into its OWN ghost rows and
    raises its own flags — push kernel, flag protocol, wait kernels, both schedules, repeated runs, step
    counts K does not divide, the all-gather reduction with one contributor."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    monkeypatch.setenv("LBM_P2P_SCHEDULE", schedule)
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="p2p", strict=True)
    assert sim.loop == "p2p" and sim.partition.macro_steps == K and schedule.split()[0] in sim.describe()["p2p"]
    av = np.concatenate([sim.run(steps), sim.run(5), sim.run(1)])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps + 6, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


P2P_CASES = {
    2: [dict(nx=130, ny=100, K=4, schedule="edge", runs=[20, 11]), dict(nx=192, ny=99, K=3, schedule="serial", runs=[7, 24]),
        dict(nx=512, ny=70, K=2, schedule="edge", runs=[31], scatter=True), dict(nx=1024, ny=1024, K=0, schedule="", runs=[13, 2], walls=True),
        dict(nx=256, ny=64, K=1, schedule="serial", runs=[9]),
        # rounds 1-3's loop (an exchange before every launch), and four launches per exchange on 16 ghost rows
        dict(nx=130, ny=100, K=4, schedule="edge", runs=[20, 11], ghost="0"), dict(nx=192, ny=99, K=4, schedule="edge", runs=[37, 20], ghost="16"),
        # tile (2-D) decomposition: two column blocks (each rank is its own south / north neighbour and the other's west AND east one)
        dict(nx=512, ny=128, K=0, schedule="", runs=[20, 11], grid=[2, 1], walls=True),
        dict(nx=1024, ny=256, K=0, schedule="edge", runs=[20, 11], grid=[2, 1]), dict(nx=1024, ny=256, K=4, schedule="edge", runs=[9, 8], grid=[1, 2], ghost="0"),
        dict(nx=1024, ny=512, K=0, schedule="", runs=[20, 11], grid=[1, 2], scatter=True), dict(nx=1024, ny=512, K=0, schedule="edge", runs=[9, 12], grid=[1, 2])],
    3: [dict(nx=256, ny=200, K=2, schedule="edge", runs=[20, 11]), dict(nx=256, ny=200, K=4, schedule="edge", runs=[20, 21], ghost="12"), dict(nx=256, ny=200, K=3, schedule="serial", runs=[31], scatter=True),
        dict(nx=1000, ny=400, K=0, schedule="edge", runs=[5, 5, 5], walls=True),
        dict(nx=772, ny=96, K=0, schedule="", runs=[13, 8], grid=[3, 1], scatter=True)],        # column blocks of 258, 258, 256
    4: [dict(nx=256, ny=131, K=3, schedule="edge", runs=[31]), dict(nx=128, ny=260, K=4, schedule="serial", runs=[17, 14]),
        dict(nx=128, ny=260, K=4, schedule="edge", runs=[17, 14], ghost="16", group="3"),
        dict(nx=2048, ny=4100, K=0, schedule="", runs=[7], scatter=True, p=0.005),
        # one-step mode: 12 / 11-row ranks, an odd row length, a 3-row last rank (d2q9-bgk.c:848-849)
        dict(nx=64, ny=48, K=0, schedule="", runs=[25, 6]), dict(nx=37, ny=45, K=0, schedule="", runs=[19], scatter=True),
        dict(nx=16, ny=9, K=0, schedule="", runs=[30]),
        # tiles: 2 x 2 (corners through two hops) and four column blocks; K = 3 on 2 x 2
        dict(nx=512, ny=256, K=0, schedule="", runs=[20, 11], grid=[2, 2], scatter=True), dict(nx=1024, ny=64, K=0, schedule="", runs=[9, 8], grid=[4, 1]),
        dict(nx=384, ny=200, K=3, schedule="", runs=[31], grid=[2, 2], walls=True),
        dict(nx=1024, ny=512, K=0, schedule="edge", runs=[20, 11], grid=[2, 2], scatter=True)],
}


@pytest.mark.parametrize("ranks", [2, 3, 4])
def test_p2p_ranks_in_separate_processes_share_the_gpu(lbm, ranks):
    """The peer-to-peer loop as an N-GPU run executes it — one process per rank, each mapping its neighbours'
    grids with hipIpcOpenMemHandle — with the ranks sharing this box's one GPU (tests/p2p_worker.py).  Uneven row
    counts (the reference's decomposition gives 34, 33, 33 rows for ny = 100 on 3 ranks), neighbours with
    different plane strides, south == north (2 ranks), tile rows that stick out, both schedules, repeated runs
    with step counts K does not divide, the obstacle scatter, the rank-order reduction, the additive digest."""
    import json
    import sys
    from conftest import ROOT
    from conftest import run_rank_processes
    # 20 s: the bound of every wait of the loop (a rank's oracle comparison takes rank 0 out of step for a second or two; the ranks
    # meet at a gloo collective before every run).  Loopback gloo: the box's hostname need not resolve.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LBM_P2P_TIMEOUT_MS="20000", GLOO_SOCKET_IFNAME="lo")

    def cmd(port):
        return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
                "--master-port", str(port), os.path.join(ROOT, "tests", "p2p_worker.py"), json.dumps(P2P_CASES[ranks])]
    # complete output of every attempt: gpurun_out/test_artifacts/p2p_shared_gpu_<ranks>.attemptN.log.  A second attempt only when the
    # rank processes never formed their group (conftest.is_rendezvous_failure) — never after an LbmError / time-out, an assert or a fault.
    r = run_rank_processes(cmd, env, f"p2p_shared_gpu_{ranks}")
    lines = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
    assert r.returncode == 0 and len(lines) == len(P2P_CASES[ranks]) and all(" ok " in l for l in lines), (r.stdout[-2000:], r.stderr[-3000:])


# Cases for a box with SEVERAL GPUs (one rank per device, over real links): both native loops, north_star's per-step
# all-reduce, ny % size != 0, step counts K does not divide (3s and 4s at K = 3), one-step mode, repeated runs, the scatter.
MULTI_GPU_CASES = {
    2: [dict(nx=2048, ny=2050, K=0, schedule="", runs=[20, 11], walls=True, p=0.005), dict(nx=192, ny=99, K=3, schedule="serial", runs=[7, 24]),
        dict(nx=130, ny=100, K=4, schedule="edge", runs=[31], scatter=True),
        dict(nx=2048, ny=2050, K=0, schedule="", runs=[20, 11], walls=True, p=0.005, exchange="rccl"),
        dict(nx=192, ny=99, K=3, schedule="serial", runs=[7, 24], exchange="rccl"),
        dict(nx=512, ny=70, K=2, schedule="edge", runs=[31], scatter=True, exchange="rccl", step_allreduce=True),
        dict(nx=1024, ny=1024, K=0, schedule="", runs=[13, 2], walls=True, exchange="rccl", step_allreduce=True),
        dict(nx=64, ny=48, K=0, schedule="", runs=[25, 6]), dict(nx=64, ny=48, K=0, schedule="", runs=[25, 6], exchange="rccl"),
        # four launches per exchange on 16 ghost rows (ranks of 200 rows), over both loops and both schedules; three on 12
        dict(nx=256, ny=400, K=0, schedule="", runs=[37, 20]), dict(nx=256, ny=400, K=0, schedule="edge", runs=[37, 20], exchange="rccl"),
        dict(nx=1024, ny=300, K=4, schedule="edge", runs=[21, 20], ghost="12", scatter=True),
        # tile (2-D) decomposition over a real link: two column blocks (columns west / east across the link, rows onto the rank itself)
        dict(nx=2048, ny=1100, K=0, schedule="", runs=[20, 11], grid=[2, 1], walls=True, p=0.005), dict(nx=484, ny=78, K=1, schedule="", runs=[9, 10], grid=[2, 1], ghost="4", group="2"),
        # ... and two ROW blocks of tile ranks (ghost rows and ghost columns; rows and corners across the link, columns onto the rank itself), both schedules
        dict(nx=1024, ny=512, K=0, schedule="", runs=[20, 11], grid=[1, 2], scatter=True), dict(nx=1024, ny=512, K=0, schedule="edge", runs=[9, 12], grid=[1, 2])],
    3: [dict(nx=256, ny=200, K=3, schedule="edge", runs=[20, 11]), dict(nx=1000, ny=400, K=0, schedule="edge", runs=[5, 5, 5], walls=True),
        dict(nx=772, ny=96, K=0, schedule="", runs=[13, 8], grid=[3, 1], scatter=True),
        dict(nx=256, ny=200, K=3, schedule="", runs=[20, 11], exchange="rccl", scatter=True),
        dict(nx=1000, ny=400, K=0, schedule="", runs=[16], walls=True, exchange="rccl", step_allreduce=True),
        dict(nx=37, ny=45, K=0, schedule="", runs=[19], scatter=True), dict(nx=37, ny=45, K=0, schedule="", runs=[19], exchange="rccl", step_allreduce=True)],
}


@pytest.mark.parametrize("ranks", [2, 3])
def test_native_loops_with_one_gpu_per_rank(lbm, ranks):
    """Switches itself on when the box has at least `ranks` GPUs: the peer-to-peer loop and the RCCL loop (with and
    without one all-reduce per macro-step) as fresh rank processes, ONE DEVICE EACH — stores, flags and RCCL messages
    over real links, ncclCommInitRank / ncclSend / ncclRecv / ncclAllReduce with nranks > 1 — against the oracle bit
    for bit (tests/p2p_worker.py).  On the one-GPU box the same worker runs with all ranks on device 0
    (test_p2p_ranks_in_separate_processes_share_the_gpu); RCCL refuses that, hence this test."""
    import json
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < ranks:
        pytest.skip(f"needs {ranks} GPUs, this box has {torch.cuda.device_count()}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", LBM_P2P_TIMEOUT_MS="20000", LBM_WORKER_DEVICE="local_rank", GLOO_SOCKET_IFNAME="lo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "tests", "p2p_worker.py"), json.dumps(MULTI_GPU_CASES[ranks])]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    lines = [l for l in r.stdout.splitlines() if l.startswith("CASE")]
    assert r.returncode == 0 and len(lines) == len(MULTI_GPU_CASES[ranks]) and all(" ok " in l for l in lines), (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_on_two_gpus_harvests_every_part(lbm):
    """Switches itself on with >= 2 GPUs: the driver's own N = 2 invocation on a smaller deck — headline over p2p, phases,
    both RCCL variants and the shipped 1024x1024 deck on two real devices, every part parity-checked."""
    import json
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--workload", "4096x4096",
                        "--secondary-steps", "2000"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][-1])
    assert out["config"]["loop"] == "p2p" and out["parity_check"]["ok"] is True and "truncated" not in out
    assert out["variants"]["rccl"]["parity_ok"] is True and out["variants"]["rccl"]["rccl_nranks"] == 2
    assert out["variants"]["rccl_step_allreduce"]["parity_ok"] is True and out["variants"]["rccl_step_allreduce"]["step_allreduce"] is True
    sec = out["secondary"]["input_1024x1024"]
    assert sec["p2p"]["parity_ok"] is True and sec["rccl"]["parity_ok"] is True
    assert out["phases"]["max_over_ranks"]["macro_steps"] == 3 and out["phases"]["max_over_ranks"]["launches"] == 5 and len(out["phases"]["per_rank"]) == 2


@pytest.mark.parametrize("nx,ny,size,K,schedule", [(256, 200, 3, 3, "serial"), (130, 100, 2, 4, "serial"), (192, 99, 2, 3, "edge"),
                                                   # >= 2^20 cells per rank: several tall-geometry contexts (79 KB frames) in one process
                                                   (2048, 1100, 2, 4, "serial")])
def test_p2p_partitions_in_one_process(lbm, nx, ny, size, K, schedule):
    """Several ranks of one run as contexts of ONE process (one host thread per rank, as a single-process
    multi-GPU host drives them), connected through plain pointers instead of IPC handles
    (tests/p2p_inprocess_worker.py).  Ranks of one process that share a DEVICE run the serial schedule whatever
    was asked for and need a hardware queue each (a wait kernel must never sit in front of the push it waits
    for), hence the fresh process with GPU_MAX_HW_QUEUES raised, as the C shim does for LBM_GPUS."""
    import sys
    from conftest import ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", LBM_P2P_TIMEOUT_MS="10000")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "p2p_inprocess_worker.py"), str(nx), str(ny), str(size), str(K), schedule],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "IN-PROCESS RING ok" in r.stdout, r.stderr[-3000:]


TILE_CASES = [   # nx ny px py K ghost group runs [walls]
    "512 256 1 1 4 - - 20,11 yghost",     # one rank: its own neighbour in all four directions (and on both diagonals)
    "512 256 2 1 4 - - 20,11 yghost",     # column blocks that keep ghost rows (each rank its own south / north neighbour through the row push)
    "512 256 1 1 4 - - 20,11",            # column blocks proper (py = 1): no ghost rows, the rows wrap inside the launch, only columns travel
    "512 256 2 1 4 - - 20,11",
    "520 100 4 1 3 - - 19,7 walls",       # ... K = 3, rows the tile height does not divide
    "1028 203 3 1 4 7 - 25,3",            # ... uneven column blocks, an odd row count, one launch per exchange
    "512 256 1 2 4 - - 20,11",            # row blocks with ghost columns that wrap onto the rank itself
    "512 256 2 2 4 - - 20,11 walls",
    "768 384 3 2 4 - - 33",
    "1024 512 4 2 3 - - 19,7",            # K = 3: 15 ghost rows, 16 ghost columns, five launches per exchange
    "640 300 2 3 4 7 - 25",               # 7 ghost rows / 8 ghost columns: one launch per exchange, the two grids as the double buffer
    "644 300 2 3 4 12 2 25,3",            # uneven column blocks (322 = 161 pairs each ... 644 / 2), groups capped at two launches
    "1290 200 3 2 4 0 - 18",              # 430-column blocks: storage rows of 438 floats (8-byte pushes), an exchange before every launch
    "2048 1100 2 1 4 - - 17",             # >= 2^20 cells per rank: the tall geometry
    # uneven column blocks (162, 162, 160 and 344, 342, 342 columns: storage rows of different widths on the ranks of one row of the rank
    # grid) — the per-step sums of every rank must land in the same slot of every window; and a second run longer than the first
    "484 78 3 1 1 4 2 9,9",
    "1028 200 3 1 4 - - 10,11",
    "590 267 2 3 1 16 - 18,19",           # K = 1: eight one-step launches per exchange
    # the edge-stream schedule of a tile rank: the rectangle of tiles inside the rim beside the exchange, the rim behind it
    "512 256 1 1 4 - - 20,11 sched=edge yghost",
    "512 256 1 1 4 0 - 20,11 walls sched=edge yghost",   # one launch per exchange
    "580 300 1 1 3 - - 31 sched=edge yghost",
    "2048 1100 1 1 4 - - 17,8 sched=edge yghost",
    "2048 1100 1 1 4 - - 17,8 sched=edge",               # a column block: the rim is two tile columns
    "448 256 4 1 4 - - 13,9",             # blocks narrower than two tiles: 112 owned columns in storage rows of 144
    "512 256 2 2 4 - - 20,11 walls flags=64",     # the other two forms of the sum|u| terms in the tile launch form (LBM_FLAG_FAST_AVVELS / _EXACT_AVVELS)
    "768 384 3 2 4 - - 33 flags=128",
]


@pytest.mark.parametrize("case", TILE_CASES)
def test_tile_decomposition_in_one_process(lbm, case):
    """SURVEY.md section 8(f) row 3, second half: the 2-D (tile) decomposition the reference's report discusses and never built.  The px x py
    ranks as contexts of one process on this GPU (tests/tile_inprocess_worker.py): populations bit for bit against the oracle after
    several runs, the per-step sums, the additive digest, observables and the velocity sum against one context holding the whole
    grid, and a further run from a state written with lbm_set_cells."""
    import sys
    from conftest import ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16", LBM_P2P_TIMEOUT_MS="10000")
    for k in ("LBM_TUNE_MACRO_K", "LBM_TUNE_MACRO_GHOST", "LBM_TUNE_MACRO_GROUP"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tile_inprocess_worker.py"), *case.split()], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "TILES ok" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])


def test_tile_context_moves_and_digests_its_block_through_a_column_window(lbm, oracle, monkeypatch):
    """lbm_get_cells / _set_cells / _get_observables / _state_checksum / _av_velocity_sum of a tile rank see its ny_local x nx_local block
    inside storage rows that also hold ghost columns: random state in, the same state out; observables in row chunks (as blocks of more
    than 16 M cells are fetched) equal to one fetch and to a whole-grid context's; digests of row ranges add up."""
    p = lbm.Params(512, 256, 10, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(512, 256, 0.03, 3, False)
    free = lbm.count_free_cells(obst)
    lay = lbm.tile_layout(p, 2, 2, 3)
    part = lbm.Partition(p, free, lbm.obstacle_window(obst, lay), tile_of=(3, 2, 2))
    rng = np.random.default_rng(5)
    state = (rng.random((256, 512, 9), dtype=np.float32) * 0.02 + 0.004).astype(np.float32)
    ys, xs = slice(lay["y0"], lay["y0"] + lay["ny_local"]), slice(lay["x0"], lay["x0"] + lay["nx_local"])
    part.set_cells(state[ys, xs])
    assert np.array_equal(part.get_cells().view(np.uint32), state[ys, xs].view(np.uint32))
    whole = lbm.Partition(p, free, obst)
    whole.set_cells(state)
    obs = part.get_observables()
    assert np.array_equal(obs.view(np.uint32), whole.get_observables()[ys, xs].view(np.uint32))
    monkeypatch.setenv("LBM_TUNE_OBS_CHUNK_CELLS", str(37 * lay["nx_local"] + 5))      # 37 rows per fetch
    assert np.array_equal(part.get_observables().view(np.uint32), obs.view(np.uint32))
    monkeypatch.delenv("LBM_TUNE_OBS_CHUNK_CELLS")
    y0, y1 = lay["y0"], lay["y0"] + lay["ny_local"]
    mid = y0 + 51
    assert (part.checksum(y0, mid) + part.checksum(mid, y1)) % (1 << 64) == part.checksum() and part.checksum(mid, mid) == 0
    # the four blocks of the grid, each set from the same state, add up to the whole grid's digest
    total = part.checksum()
    for r in range(3):
        l = lbm.tile_layout(p, 2, 2, r)
        q = lbm.Partition(p, free, lbm.obstacle_window(obst, l), tile_of=(r, 2, 2))
        q.set_cells(state[l["y0"]:l["y0"] + l["ny_local"], l["x0"]:l["x0"] + l["nx_local"]])
        total = (total + q.checksum()) % (1 << 64)
        q.close()
    assert total == whole.checksum()
    want = sum(float(np.sqrt(np.float64(o[0] * o[0] + o[1] * o[1]))) for o in obs[obst[ys, xs] == 0])
    assert abs(part.av_velocity_sum() - want) <= 1e-9 * want
    with pytest.raises(lbm.LbmError, match="rows outside"):
        part.checksum(0, 10)
    part.close()
    whole.close()


def test_tile_ranks_are_refused_where_they_cannot_run(lbm):
    """A rank of the tile decomposition is stepped by the peer-to-peer loop only: the RCCL loop, the split-phase calls and lbm_run say so
    instead of stepping it wrongly; a transport of the wrong size is turned away at create."""
    import ctypes as C
    p = lbm.Params(512, 256, 10, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(512, 256, 0.03, 3, False)
    lay = lbm.tile_layout(p, 2, 1, 0)
    part = lbm.Partition(p, lbm.count_free_cells(obst), lbm.obstacle_window(obst, lay), tile_of=(0, 2, 1))
    assert part.get_cells().shape == (256, 256, 9) and part.macro_steps == 4
    with pytest.raises(lbm.LbmError, match="peer-to-peer loop"):
        part.macro_prepare(8)
    with pytest.raises(lbm.LbmError, match="not a self-contained domain"):
        part.run(4)
    with pytest.raises(lbm.LbmError, match="tile decomposition"):
        lbm.RcclRing(part, rank=0, size=1)
    with pytest.raises(lbm.LbmError, match="rank 0 of 2 tiles"):
        lbm.P2PRing(part, rank=0, size=3, connect=False)
    part.close()


def test_p2p_refuses_ranks_with_different_layouts(lbm, monkeypatch):
    """Ranks created with different K (what per-rank decisions gave for uneven partitions) are turned away at
    connect with an error — not left to hang in the first exchange."""
    p = lbm.Params(256, 128, 10, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(256, 128, 0.03, 3, False)
    free = lbm.count_free_cells(obst)
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "3")
    a = lbm.Partition(p, free, obst[:64], 0, obstacles_global=obst)
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "2")
    b = lbm.Partition(p, free, obst[64:], 64, obstacles_global=obst)
    assert (a.macro_steps, b.macro_steps) == (3, 2)
    with pytest.raises(lbm.LbmError, match="different layout"):
        lbm.P2PRing.local_ring([a, b])
    a.close(); b.close()


def test_p2p_missing_peer_is_an_error_not_a_hang(lbm, monkeypatch):
    """A rank whose neighbour never starts the run gives up after the time-out with an error."""
    monkeypatch.setenv("LBM_P2P_TIMEOUT_MS", "300")
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "3")
    p = lbm.Params(256, 128, 10, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(256, 128, 0.03, 3, False)
    free = lbm.count_free_cells(obst)
    parts = [lbm.Partition(p, free, lbm.obstacle_window(obst, lbm.rank_layout(p, 2, r)), rank_of=(r, 2)) for r in range(2)]
    rings = lbm.P2PRing.local_ring(parts)
    with pytest.raises(lbm.LbmError, match="did not arrive in time"):
        rings[0].run(6)                                         # rank 1 never runs
    for ring in rings:
        ring.close()
    for q in parts:
        q.close()


def test_bench_self_launch_two_ranks_on_one_gpu(lbm):
    """`python bench.py --gpus 2` as the driver calls it (no rank environment): the parent starts two rank
    processes, they share this box's one GPU (gloo group, peer-to-peer halos over hipIpc), check themselves
    bit for bit against a single-GPU run, and ONE JSON line comes back saying what ran."""
    import json
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LBM_FORCE_DEVICE="0", LBM_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5", "--reps", "3",
                        "--workload", "2048x2048", "--secondary-steps", "3000"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 20 and out["value"] > 0 and "truncated" not in out
    assert out["config"]["loop"] == "p2p" and out["config"]["macro_k"] == 4 and "ipc" in out["config"]["p2p"]
    assert out["parity_check"]["ok"] is True and out["exchange_attempts"][0] == {"exchange": "p2p", "ok": True}
    assert out["launch_attempts"][0]["returncode"] == 0
    # what one multi-rank invocation harvests beside the headline: phases of a profiled repetition,
    # the RCCL variants (here: recorded as not usable, the two ranks share this box's one GPU — not a crash), and
    # BASELINE.json config 4, the shipped 1024 x 1024 deck on the same ranks
    ph = out["phases"]
    assert len(ph["per_rank"]) == 2 and ph["max_over_ranks"]["macro_steps"] == 3 and ph["max_over_ranks"]["launches"] == 5   # 20 = (4 + 4) + (4 + 4) + 4: three exchanges
    for name in ("host_total", "setup", "steps", "reduce", "macro_step_avg", "interior_avg", "push_first", "push_avg", "host_overhead"):
        assert ph["max_over_ranks"][name] > 0, name
    for name in ("rccl", "rccl_step_allreduce"):
        assert "ranks share a GPU" in out["variants"][name]["error"]
    # rounds 1-3's loop (an exchange before every launch) beside the headline's one exchange per group of launches
    every = out["variants"]["p2p_exchange_every_launch"]
    assert every["parity_ok"] is True and (every["ghost_rows"], every["launches_per_exchange"]) == (4, 1) and every["value"] > 0
    assert (out["config"]["ghost_rows"], out["config"]["launches_per_exchange"]) == (8, 2)
    tiles = out["variants"]["p2p_tiles"]                            # the same deck over the tile decomposition (2 ranks: two column blocks)
    assert tiles["parity_ok"] is True and tiles["rank_grid"] == [2, 1] and tiles["block"] == [1024, 2048] and tiles["value"] > 0
    sec = out["secondary"]["input_1024x1024"]
    assert sec["steps"] == 3000 and sec["p2p"]["parity_ok"] is True and sec["p2p"]["value"] > 0 and "ranks share a GPU" in sec["rccl"]["error"]
    assert sec["p2p_tiles"]["parity_ok"] is True and "tiles 2 x 1" in sec["p2p_tiles"]["p2p"]
    for name in ("256x256", "128x256", "128x128"):                  # the small shipped decks on the same ranks, whole runs
        small = out["secondary"][f"input_{name}"]["p2p"]
        assert small["parity_ok"] is True and small["reynolds_line_equals_reference"] is True and small["value"] > 0


@pytest.mark.parametrize("gpus,name", [(2, "256x256_t1000"), (3, "1024x1024_t200"), (4, "128x256_t2000"), (4, "rand_64x48"), (3, "tall_8x256")])
def test_cli_drives_several_ranks_from_one_process(lbm, digests, tmp_path, gpus, name):
    """LBM_GPUS=N: the C shim as a single-process multi-GPU host — N ranks of the reference's decomposition, one
    host thread each, peer-to-peer halos.  Here all ranks sit on device 0 (LBM_DEVICES), the way a one-GPU box
    allows; same stdout contract, byte-identical final_state.dat, same Reynolds line as one rank."""
    ppath, opath = deck_paths(name, digests)
    env = dict(os.environ, LBM_GPUS=str(gpus), LBM_DEVICES=",".join(["0"] * gpus), LBM_P2P_TIMEOUT_MS="20000")
    r = subprocess.run([lbm.CLI_PATH, ppath, opath], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0] == "==done==" and out[1] == digests[name]["reynolds_line"]
    assert f"({gpus} GPUs, peer-to-peer halos)" in out[5]
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    av = lbm.checker.load_av_vels(str(tmp_path / "av_vels.dat"))
    assert np.allclose(av[np.asarray(digests[name]["av_sample_steps"])], digests[name]["av_sample_values"], rtol=5e-4)


@pytest.mark.parametrize("gpus,grid,name", [(4, "2x2", "1024x1024_t200"), (2, "2x1", "256x256_t1000"), (2, "1x2", "256x256_t1000"), (8, "4x2", "1024x1024_t200"),
                                            (8, "8x1", "1024x1024_t200")])     # eight column blocks of 128 columns: no ghost rows, the launches wrap in y
def test_cli_drives_a_tile_decomposition_from_one_process(lbm, digests, tmp_path, gpus, grid, name):
    """LBM_GPUS=N LBM_RANK_GRID=PXxPY: the same single-process host over the tile (2-D) decomposition — the shipped decks end in the
    reference binary's final_state.dat byte for byte, whichever way the grid is cut (4 x 2: BASELINE.json config 4's eight ranks)."""
    ppath, opath = deck_paths(name, digests)
    env = dict(os.environ, LBM_GPUS=str(gpus), LBM_RANK_GRID=grid, LBM_DEVICES=",".join(["0"] * gpus), LBM_P2P_TIMEOUT_MS="20000")
    r = subprocess.run([lbm.CLI_PATH, ppath, opath], cwd=tmp_path, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    out = r.stdout.splitlines()
    assert out[0] == "==done==" and out[1] == digests[name]["reynolds_line"]
    assert f"({gpus} GPUs, peer-to-peer halos, tile decomposition)" in out[5]
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    av = lbm.checker.load_av_vels(str(tmp_path / "av_vels.dat"))
    assert np.allclose(av[np.asarray(digests[name]["av_sample_steps"])], digests[name]["av_sample_values"], rtol=5e-4)
    bad = subprocess.run([lbm.CLI_PATH, ppath, opath], cwd=tmp_path, capture_output=True, text=True, timeout=60, env=dict(env, LBM_RANK_GRID="3x5"))
    assert bad.returncode != 0 and "LBM_RANK_GRID" in bad.stderr


def test_cli_chooses_tiles_for_a_grid_much_wider_than_tall(lbm, tmp_path):
    """LBM_RANK_GRID=auto (lbm_choose_rank_grid): a 8192 x 256 deck on 4 ranks — 64-row blocks of 512 K cells — runs as 4 x 1 tiles, the shipped
    square deck as row blocks; either way the files equal a one-rank run's byte for byte."""
    p = lbm.Params(8192, 256, 24, 10, 0.1, 0.005, 1.85)
    assert lbm.choose_rank_grid(p, 4) == (4, 1)
    ppath, opath = lbm.write_synthetic_deck(str(tmp_path), "8192x256", p, p=0.005, seed=7)
    outs = {}
    for tag, extra in (("one", {}), ("auto", dict(LBM_GPUS="4", LBM_DEVICES="0,0,0,0", LBM_RANK_GRID="auto", LBM_P2P_TIMEOUT_MS="20000"))):
        d = tmp_path / tag
        d.mkdir()
        env = {k: v for k, v in os.environ.items() if k not in ("LBM_GPUS", "LBM_DEVICES", "LBM_RANK_GRID")}
        env.update(extra)
        r = subprocess.run([lbm.CLI_PATH, ppath, opath], cwd=d, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        outs[tag] = (r.stdout.splitlines(), sha256(d / "final_state.dat"), lbm.checker.load_av_vels(str(d / "av_vels.dat")))
    assert "(4 GPUs, peer-to-peer halos, tile decomposition)" in outs["auto"][0][5]
    assert outs["one"][0][1] == outs["auto"][0][1] and outs["one"][1] == outs["auto"][1]
    assert np.allclose(outs["one"][2], outs["auto"][2], rtol=2e-7, atol=0)


@pytest.mark.parametrize("devices", ["0,0", "0,1"])
def test_cli_ranks_of_one_process_on_the_tall_geometry(lbm, tmp_path, devices):
    """The single-process multi-device host (LBM_GPUS=N, one context per rank in ONE process: the reference's `mpirun -np N`,
    mpi_submit:63) on a deck with >= 2^20 cells per rank, so that every context launches lbm_multi_kernel<4> on 64 x 23 tiles — 79 KB of
    dynamic LDS, a limit that is raised per DEVICE by lbm_create (a per-process flag once left the second device of such a host at
    the default).  "0,0": both ranks on this box's GPU.  "0,1": one device per rank — switches itself on with two GPUs.  Against a
    one-rank run of the same binary: same bytes in final_state.dat and av_vels.dat within the float's own rounding."""
    import torch
    if devices == "0,1" and torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    p = lbm.Params(2048, 1100, 44, 10, 0.1, 0.005, 1.85)
    ppath, opath = lbm.write_synthetic_deck(str(tmp_path), "2048x1100", p, p=0.005, seed=7)
    outs = {}
    for tag, extra in (("one", {}), ("two", dict(LBM_GPUS="2", LBM_DEVICES=devices, LBM_P2P_TIMEOUT_MS="20000"))):
        d = tmp_path / tag
        d.mkdir()
        env = {k: v for k, v in os.environ.items() if k not in ("LBM_GPUS", "LBM_DEVICES")}
        env.update(extra)
        r = subprocess.run([lbm.CLI_PATH, ppath, opath], cwd=d, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr
        outs[tag] = (r.stdout.splitlines(), sha256(d / "final_state.dat"), lbm.checker.load_av_vels(str(d / "av_vels.dat")))
    assert "(2 GPUs, peer-to-peer halos)" in outs["two"][0][5] and "(1 GPU)" in outs["one"][0][5]
    assert outs["one"][0][1] == outs["two"][0][1]                       # the Reynolds line
    assert outs["one"][1] == outs["two"][1]                             # final_state.dat, byte for byte
    assert np.allclose(outs["one"][2], outs["two"][2], rtol=2e-7, atol=0)


def test_observables_path_writes_the_reference_file(lbm, digests, tmp_path):
    """lbm_get_observables (4 floats per cell computed on the device) + lbm_write_final_state_obs /
    lbm_av_velocity_obs against the 9-population path: same bytes, same Reynolds line; also on a random state
    with a zero-density cell (NaN columns)."""
    name = "256x256_t1000"
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst)
    av = sim.run()
    obs = sim.gather_observables()
    assert "Reynolds number:\t\t%.12E" % sim.reynolds(observables=obs) == digests[name]["reynolds_line"]
    sim.write_values(av, str(tmp_path), observables=obs)
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    sim.close()
    rng = np.random.default_rng(3)
    q = lbm.Params(48, 20, 8, 4, 0.1, 0.02, 1.6)
    o2 = lbm.synthetic_obstacles(48, 20, 0.08, 9, False)
    o2[0, :4] = 0
    cells0 = (rng.random((20, 48, 9), dtype=np.float32) * 0.02 + 0.004).astype(np.float32)
    cells0[0, 0] = 0.0                                            # rho = 0: NaN u_x, u_y, u (the reference prints -NAN)
    cells0[0, 1] = [1e-45, 0, 0, 0, 0, 0, 0, 0, 0]
    cells0[0, 2] = [-1.0, 0.5, 0, 0, 0, 0, 0, 0, 0]
    part = lbm.Partition(q, lbm.count_free_cells(o2), o2)
    part.set_cells(cells0)
    a, b = str(tmp_path / "a.dat"), str(tmp_path / "b.dat")
    lbm.write_final_state(a, q, cells0, o2)
    lbm.write_final_state_obs(b, q, part.get_observables(), o2)
    part.close()
    assert open(a, "rb").read() == open(b, "rb").read()


def test_av_velocity_sum_on_a_k_step_partition_with_unaligned_ghost_rows(lbm, monkeypatch):
    """nx = 130, K = 3 puts the first owned cell at bit 390 of the obstacle bitfield (not a
    multiple of 32); lbm_av_velocity_sum must test the owned cells' own bits."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "3")
    p = lbm.Params(130, 40, 30, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(130, 40, 0.2, 17, False)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="p2p", strict=True)
    assert sim.partition.macro_steps == 3
    sim.run(30)
    host = lbm.av_velocity_host(p, sim.local_cells(), obst)
    dev = sim.partition.av_velocity_sum()
    sim.close()
    assert abs(dev - host) / host < 1e-5


@pytest.mark.parametrize("K", [0, 3])
def test_rccl_ring_with_one_all_reduce_per_step(lbm, oracle, digests, monkeypatch, K):
    """The measured mode of north_star's wording: every (macro-)step's totals folded at once and all-reduced
    on the compute stream, the next step behind it.  Same results as the hoisted reduction."""
    monkeypatch.setenv("LBM_TUNE_MACRO_K", str(K))
    p, obst, free = load_case(lbm, digests, "synth_512x512_t100")
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="rccl", strict=True, step_allreduce=True)
    assert sim.describe()["step_allreduce"] and sim.describe()["rccl_nranks"] == 1 and sim.partition.macro_steps == K
    av = np.concatenate([sim.run(41), sim.run(7)])
    cells = sim.local_cells()
    sim.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, 48, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


def test_rank_partitions_from_windows_only(lbm, oracle, monkeypatch):
    """lbm_create_rank: a rank is built from its obstacle WINDOW alone (owned rows + ghost rows); in-process
    exchange by device copies as in the other K-step partition tests."""
    nx, ny, size, steps = 256, 200, 3, 25
    p = lbm.Params(nx, ny, steps, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.05, 77, True)
    free = lbm.count_free_cells(obst)
    parts = [lbm.Partition(p, free, lbm.obstacle_window(obst, lbm.rank_layout(p, size, r)), rank_of=(r, size)) for r in range(size)]
    K = parts[0].macro_steps
    assert K == 4 and all(q.macro_steps == K for q in parts)
    sums = _k_step_partitions_in_process(lbm, parts, steps, K)
    cells = np.concatenate([q.get_cells() for q in parts], axis=0)
    for q in parts:
        q.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    av = sums * np.float64(np.float32(1.0) / np.float32(free))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < 1e-12


@pytest.mark.parametrize("name", ["128x128", "128x256", "256x256", "1024x1024"])
def test_fast_av_vels_flag_on_the_shipped_decks(lbm, digests, tmp_path, monkeypatch, name):
    """LBM_FLAG_FAST_AVVELS (float sum|u| terms in lbm_multi_kernel, default off): populations must not move by a
    bit (final_state.dat still the reference binary's file) and av_vels must stay inside check.py's 1 % against the
    shipped (double-precision) goldens on all four decks — observed: as close as the exact-term form."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")                     # the small decks through lbm_multi_kernel too
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FAST_AVVELS)
    assert sim.partition.describe()["kernel"] == ("lbm_multi_kernel<4, fast av_vels>" if name == "1024x1024" else "lbm_multi_kernel<3, fast av_vels>")
    av = sim.run()
    sim.write_values(av, str(tmp_path))
    sim.close()
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    ck = lbm.checker
    rep = ck.check_files(os.path.join(GOLDEN, "check", f"{name}.av_vels.dat.gz"), None, str(tmp_path / "av_vels.dat"), None)
    assert rep.ok and abs(rep.av_vels.max_diff_pcnt) < 0.3, rep.message
    steps = np.asarray(digests[name]["av_sample_steps"])
    assert np.allclose(av[steps], digests[name]["av_sample_values"], rtol=4e-3 if name == "1024x1024" else 5e-4)


@pytest.mark.parametrize("name,steps,kernel", [("synth_512x512_t100", 100, "multi"), ("1024x1024_t200", 200, "multi"), ("128x128", 203, "tile"),
                                               ("256x256_t1000", 77, "tile"), ("128x128", 50, "multi")])
def test_fast_av_vels_flag_against_the_oracle(lbm, oracle, digests, monkeypatch, name, steps, kernel):
    """The same flag against the oracle's exact per-step sums: 1e-6 relative (observed ~5e-8), state bit-exact;
    lbm_multi_kernel and lbm_tile_kernel (pair and one-cell-per-lane sub-steps)."""
    if kernel == "multi":
        monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    p, obst, free = load_case(lbm, digests, name)
    exact = lbm.Simulation(p, obst)
    fast = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FAST_AVVELS)
    assert f"lbm_{kernel}_kernel" in fast.partition.describe()["kernel"] and "fast av_vels" in fast.partition.describe()["kernel"]
    av_e, av_f = exact.run(steps), fast.run(steps)
    assert np.array_equal(bits(exact.local_cells()), bits(fast.local_cells()))
    exact.close()
    cells = fast.local_cells()
    fast.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av_f - ref_exact) / ref_exact) < AV_EXACT_RTOL
    assert np.max(np.abs(av_f.astype(np.float64) - av_e) / av_e) < 1e-6


@pytest.mark.parametrize("name,steps", [("synth_512x512_t100", 100), ("1024x1024_t200", 200), ("128x128", 50)])
def test_three_forms_of_the_av_vels_terms(lbm, oracle, digests, monkeypatch, name, steps):
    """lbm_multi_kernel's sum|u| terms (kernels/common.h finish_pair_lo): compensated float sums by default, double
    precision with LBM_FLAG_EXACT_AVVELS, plain float with LBM_FLAG_FAST_AVVELS.  Populations never move by a bit; the
    default's av_vels (floats) are the double-precision form's, value for value; as a 1-rank ring (tot_u in double) the two
    agree to 1e-12 with each other and with the oracle's exact per-step sums."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    p, obst, free = load_case(lbm, digests, name)
    sims = {"default": lbm.Simulation(p, obst), "exact": lbm.Simulation(p, obst, flags=lbm._capi.FLAG_EXACT_AVVELS),
            "fast": lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FAST_AVVELS)}
    names = {k: v.partition.describe()["kernel"] for k, v in sims.items()}
    assert names["default"].endswith(">") and "av_vels" not in names["default"], names
    assert "double-precision av_vels terms" in names["exact"] and "fast av_vels" in names["fast"], names
    av = {k: v.run(steps) for k, v in sims.items()}
    cells = {k: v.local_cells() for k, v in sims.items()}
    for v in sims.values():
        v.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    for k in sims:
        assert np.array_equal(bits(cells[k]), bits(ref_cells)), k
        assert np.max(np.abs(av[k].astype(np.float64) - ref_exact) / ref_exact) < AV_EXACT_RTOL, k
    assert np.array_equal(av["default"], av["exact"])
    tot = {}
    for k, flags in (("default", 0), ("exact", lbm._capi.FLAG_EXACT_AVVELS)):
        ring = lbm.Simulation(p, obst, flags=flags | lbm._capi.FLAG_FORCE_HALO, exchange="p2p", strict=True)
        assert ring.loop == "p2p"
        tot[k] = ring._p2p.run(steps) * np.float64(np.float32(1.0) / np.float32(free))
        assert np.array_equal(bits(ring.local_cells()), bits(ref_cells))
        ring.close()
    assert np.max(np.abs(tot["default"] - tot["exact"]) / tot["exact"]) < 1e-12
    assert np.max(np.abs(tot["default"] - ref_exact) / ref_exact) < 1e-12


def test_launch_profile_and_ring_phases(lbm, digests, monkeypatch):
    """The profiling entry points behind bench.py's `roofline` and `phases` (the reference's MPI_Pcontrol("mainloop") region,
    d2q9-bgk.c:275-277,404-406): lbm_set_profile / lbm_launch_profile name every step-kernel launch of lbm_run with the steps it
    advanced — 20 steps at K = 3 are 4 + 4 + 3 + 3 + 3 + 3 — and their durations add up to the run's kernel span;
    lbm_p2p_set_profile / lbm_p2p_phases account for a ring run's wall time.  Profiling must not change a single bit."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    p, obst, free = load_case(lbm, digests, "synth_512x512_t100")
    plain, prof = lbm.Simulation(p, obst), lbm.Simulation(p, obst)
    prof.partition.set_profile(True)
    av_a, av_b = plain.run(20), prof.run(20)
    launches = prof.partition.launch_profile()
    span_ms, n = prof.partition.last_run_kernel_ms()
    assert [k for k, _ in launches] == [4, 4, 3, 3, 3, 3] and n == 6
    assert all(us > 0 for _, us in launches) and 0.5 < sum(us for _, us in launches) / (span_ms * 1e3) < 1.5
    assert np.array_equal(av_a, av_b) and np.array_equal(bits(plain.local_cells()), bits(prof.local_cells()))
    prof.partition.set_profile(False)
    prof.run(7)
    assert prof.partition.launch_profile() == []                       # off again: nothing recorded
    plain.close(); prof.close()
    for schedule in ("edge", "serial"):
        monkeypatch.setenv("LBM_P2P_SCHEDULE", schedule)
        ring = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO, exchange="p2p", strict=True)
        with pytest.raises(lbm.LbmError, match="no profiled run"):
            ring._p2p.phases()
        ring._p2p.set_profile(True)
        av_r = ring.run(20)
        ph = ring._p2p.phases()
        assert np.max(np.abs(av_r - av_a) / av_a) < AV_EXACT_RTOL
        k = ring.partition.macro_steps
        groups = lbm.plan_groups(k, ring.layout["ghost"], ring.layout["group"], 20)      # 512 x 512, K = 4 on 16 ghost rows: (4 + 4 + 4 + 4) + 4
        assert (ph["macro_steps"], ph["launches"]) == (len(groups), sum(len(g) for g in groups)) and ph["launches"] == (6 if k == 3 else 5)
        assert ph["whole_avg"] > 0                                       # the second launch of a group: all tiles, nothing exchanged
        assert ph["host_total"] >= ph["device_span"] > 0 and abs(ph["setup"] + ph["steps"] + ph["reduce"] - ph["device_span"]) < 0.05 * ph["device_span"] + 5.0
        # (512 x 512 rows keep 32 ghost rows: the 20 steps are ONE group of five launches — a first push and no later one)
        assert ph["interior_avg"] > 0 and ph["push_first"] > 0 and (ph["push_avg"] > 0) == (len(groups) > 1) and (ph["edge_avg"] > 0) == (schedule == "edge")
        assert abs(ph["host_overhead"] - (ph["host_total"] - ph["device_span"])) < 1e-6
        ring.close()
