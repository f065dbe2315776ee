"""Parity suite of the EXPERIMENT build (hipcc -DLBM_EXPERIMENTS=1: scripts/build_variant.sh experiments): the forms that measured
slower and stay in the tree for the record — lbm_sweep_kernel (kernels/sweep.h, LBM_TUNE_SWEEP) and the LDS-staged one-step kernel
(LBM_FLAG_KERNEL_LDS) — against the oracle, bit for bit.  liblbm_d2q9.so as shipped does not carry them.  Not collected by a plain
`pytest tests`: tests/test_gpu_parity.py::test_experiment_build_passes_its_suite builds the variant and runs this file ONCE, in a
process of its own, with LBM_LIBRARY pointing at it."""
import os

import numpy as np
import pytest

from conftest import deck_paths
from test_gpu_parity import AV_EXACT_RTOL, STRESS_KINDS, _stress_deck, bits, load_case, sha256

pytestmark = pytest.mark.gpu


def test_this_is_the_experiment_build(lbm):
    assert os.environ.get("LBM_LIBRARY", "").endswith("experiments.so") and lbm._capi.LIB_PATH == os.environ["LBM_LIBRARY"]


@pytest.mark.parametrize("name", ["tiny_8x3", "rand_64x48", "walls_40x24", "wide_256x8", "tall_8x256",
                                  "synth_512x512_t100", "1024x1024_t200"])
def test_lds_staged_kernel_same_results(lbm, oracle, digests, monkeypatch, name):
    """LBM_FLAG_KERNEL_LDS: the LDS-tiled form of the step kernel (aligned loads, x+-1 neighbours and
    the obstacle bitfield through LDS) must give the same bits as the direct-load form."""
    monkeypatch.setenv("LBM_TUNE_NARROW_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MULTI_K", "0")
    p, obst, free = load_case(lbm, digests, name)
    steps = min(p.max_iters, 150)
    s = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_KERNEL_LDS)
    assert "lds" in s.partition.describe()["kernel"]
    av = s.run(steps)
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("nx,ny", [(4, 3), (12, 7), (36, 5), (1028, 6), (2048, 3)])
def test_lds_staged_kernel_odd_shapes(lbm, oracle, monkeypatch, nx, ny):
    monkeypatch.setenv("LBM_TUNE_NARROW_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MULTI_K", "0")
    p = lbm.Params(nx, ny, 30, 4, 0.1, 0.01, 1.4)
    obst = lbm.synthetic_obstacles(nx, ny, 0.1, nx * 31 + ny, False)
    if obst.all():
        obst[0, 0] = 0
    s = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_KERNEL_LDS | lbm._capi.FLAG_NT_STORES)
    s.run(30)
    ref_cells, _, _ = oracle.run(p, obst, 30)
    assert np.array_equal(bits(s.local_cells()), bits(ref_cells))
    s.close()


def test_lds_staged_kernel_in_a_ring(lbm, oracle, digests, monkeypatch):
    monkeypatch.setenv("LBM_TUNE_NARROW_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_MACRO_K", "0")
    p, obst, free = load_case(lbm, digests, "synth_512x512_t100")
    sim = lbm.Simulation(p, obst, flags=lbm._capi.FLAG_FORCE_HALO | lbm._capi.FLAG_KERNEL_LDS, exchange="rccl")
    sim.run(40)
    ref_cells, _, _ = oracle.run(p, obst, 40, nthreads=4)
    assert np.array_equal(bits(sim.local_cells()), bits(ref_cells))
    sim.close()



@pytest.mark.parametrize("mode", ["2", "1", "0"])
@pytest.mark.parametrize("R", [5, 4])
@pytest.mark.parametrize("nx,ny,steps,blocks", [(128, 64, 7, 4), (64, 96, 9, 3), (256, 200, 12, 8), (192, 77, 10, 6), (512, 512, 31, 512),
                                                 (1024, 333, 6, 48), (64, 64, 3, 1)])
def test_sweep_kernel_shapes_and_step_counts(lbm, oracle, monkeypatch, nx, ny, steps, blocks, R, mode):
    """lbm_sweep_kernel<R> (kernels/sweep.h: the 3-step launch as a streaming pipeline in y, strips of 64 columns, R rows
    per tick) against the oracle, bit for bit: one strip (both x wraps in one block), several strips, segments whose rows
    neither R nor the segment count divides, segments shorter than the pipeline is deep, step counts with 1-, 2- and 4-step
    tails (those launches are lbm_multi_kernel's), repeated runs, obstacles on the ring columns, the accelerate row."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_SWEEP", str(R))
    monkeypatch.setenv("LBM_TUNE_SWEEP_BLOCKS", str(blocks))
    monkeypatch.setenv("LBM_TUNE_SWEEP_MODE", mode)                # storage form of the pipeline: kernels/sweep.h SweepGeom
    p = lbm.Params(nx, ny, steps + 4, 4, 0.1, 0.01, 1.7)
    obst = lbm.synthetic_obstacles(nx, ny, 0.04, nx + 3 * ny + R, False)
    s = lbm.Simulation(p, obst)
    assert s.partition.describe()["kernel"] == f"lbm_sweep_kernel<{R}>"
    av = np.concatenate([s.run(steps), s.run(4)])
    cells = s.local_cells()
    s.close()
    ref_cells, _, ref_exact = oracle.run(p, obst, steps + 4, nthreads=4)
    assert np.array_equal(bits(cells), bits(ref_cells))
    assert np.max(np.abs(av - ref_exact) / ref_exact) < AV_EXACT_RTOL


@pytest.mark.parametrize("kind", STRESS_KINDS)
def test_sweep_kernel_rare_paths(lbm, oracle, monkeypatch, kind):
    """The stress decks (no obstacle anywhere, every second cell blocked, row ny-2 walled off, an acceleration that fails
    the positivity test, relaxation at both ends of the range) through lbm_sweep_kernel<5>, two segments."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_SWEEP", "5")
    monkeypatch.setenv("LBM_TUNE_SWEEP_BLOCKS", "8")
    for nx, ny in ((256, 64), (448, 112)):
        if nx % 64:
            continue
        p, obst = _stress_deck(lbm, kind, nx, ny)
        s = lbm.Simulation(p, obst)
        assert s.partition.describe()["kernel"] == "lbm_sweep_kernel<5>"
        av = np.concatenate([s.run(29), s.run(11)])
        cells = s.local_cells()
        s.close()
        ref_cells, _, ref_exact = oracle.run(p, obst, 40, nthreads=4)
        assert np.array_equal(bits(cells), bits(ref_cells)), (kind, nx, ny)
        assert np.max(np.abs(av - ref_exact) / np.maximum(ref_exact, 1e-30)) < AV_EXACT_RTOL


@pytest.mark.parametrize("name", ["128x128", "1024x1024"])
def test_sweep_kernel_on_shipped_decks(lbm, digests, tmp_path, monkeypatch, name):
    """Whole shipped decks (40 000 / 20 000 steps) through lbm_sweep_kernel<5>: final_state.dat is the reference binary's file."""
    monkeypatch.setenv("LBM_TUNE_TILE_MAX", "0")
    monkeypatch.setenv("LBM_TUNE_SWEEP", "5")
    p, obst, free = load_case(lbm, digests, name)
    sim = lbm.Simulation(p, obst)
    assert sim.partition.describe()["kernel"] == "lbm_sweep_kernel<5>"
    av = sim.run()
    sim.write_values(av, str(tmp_path))
    sim.close()
    assert sha256(tmp_path / "final_state.dat") == digests[name]["final_state_sha256"]
    steps = np.asarray(digests[name]["av_sample_steps"])
    assert np.allclose(av[steps], digests[name]["av_sample_values"], rtol=4e-3 if name == "1024x1024" else 5e-4)




def test_randomised_cross_check_with_the_experiment_forms(lbm):
    """scripts/fuzz_kernels.py on the experiment build: its "sweep" and "lds" cases are drawn only here."""
    import importlib.util
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("fuzz_kernels", os.path.join(ROOT, "scripts", "fuzz_kernels.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.EXPERIMENTS and mod.main(["--cases", "60", "--seed", "12"]) == 0
