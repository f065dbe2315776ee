"""CPU suite, part 1: the oracle (oracle/d2q9_oracle.c) against the golden vectors.

digests.json holds sha256 of final_state.dat / av_vels.dat produced by the UNMODIFIED reference
binary (oracle/_ref, see tests/golden/make_fixtures.py); the restatement must reproduce them byte
for byte.  small_cases.npz holds the reference binary's parsed outputs for the small decks; the
shipped double-precision goldens (check/*.dat.gz) are checked with the check.py criterion."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, deck_paths

# cases the oracle finishes in seconds on one core (the full 128x128 deck takes ~10 s)
FAST_CASES = ["tiny_8x3", "open_64x48", "rand_64x48", "walls_40x24", "dense_32x32", "strongaccel_32x16",
              "accelrow_blocked_32x16", "column_24x20", "wide_256x8", "tall_8x256", "synth_512x512_t100",
              "128x256_t2000", "256x256_t1000", "1024x1024_t200", "128x128"]


def sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def run_case(oracle, digests, name, nthreads=1):
    ppath, opath = deck_paths(name, digests)
    p = oracle.read_params(ppath)
    obst, free = oracle.read_obstacles(opath, p.nx, p.ny)
    cells, av, exact = oracle.run(p, obst, p.max_iters, nthreads=nthreads)
    return p, obst, free, cells, av, exact


@pytest.mark.parametrize("name", FAST_CASES)
def test_oracle_reproduces_reference_digests(oracle, digests, tmp_path, name):
    p, obst, free, cells, av, exact = run_case(oracle, digests, name)
    fs, avf = str(tmp_path / "final_state.dat"), str(tmp_path / "av_vels.dat")
    oracle.write_final_state(fs, p, cells, obst)
    oracle.write_av_vels(avf, av)
    assert sha256(fs) == digests[name]["final_state_sha256"]
    assert sha256(avf) == digests[name]["av_vels_sha256"]
    # Reynolds line (d2q9-bgk.c:408,412)
    tot = np.float32(oracle.av_velocity_sum(p, cells, obst))
    re = oracle.reynolds(p, float(tot * (np.float32(1.0) / np.float32(free))))
    assert "Reynolds number:\t\t%.12E" % re == digests[name]["reynolds_line"]
    # the double-accumulated yardstick stays close to the reference-order float sum on these sizes
    assert np.allclose(av, exact, rtol=5e-3 if p.nx * p.ny > 100000 else 2e-4)


@pytest.mark.parametrize("name", ["rand_64x48", "walls_40x24", "synth_512x512_t100"])
def test_oracle_thread_count_does_not_change_results(oracle, digests, name):
    _, _, _, c1, a1, e1 = run_case(oracle, digests, name, nthreads=1)
    _, _, _, c3, a3, e3 = run_case(oracle, digests, name, nthreads=3)
    assert np.array_equal(c1.view(np.uint32), c3.view(np.uint32))
    assert np.array_equal(a1, a3) and np.array_equal(e1, e3)


@pytest.mark.parametrize("name", ["rand_64x48", "dense_32x32"])
def test_oracle_fast_form_same_state(oracle, digests, name):
    p, obst, _, cells, av, _ = run_case(oracle, digests, name)
    c1, a1 = oracle.run_fast(p, obst, p.max_iters, 1)
    c4, a4 = oracle.run_fast(p, obst, p.max_iters, 4)
    assert np.array_equal(cells.view(np.uint32), c1.view(np.uint32))
    assert np.array_equal(cells.view(np.uint32), c4.view(np.uint32))
    assert np.array_equal(av, a1)                      # one thread = the reference's serial order
    assert np.allclose(av, a4, rtol=1e-5)              # only the float summation order moves


def test_oracle_matches_parsed_reference_outputs(oracle, digests):
    small = np.load(os.path.join(GOLDEN, "small_cases.npz"))
    names = sorted({k.split("__")[0] for k in small.files})
    assert len(names) >= 6
    for name in names:
        p, obst, free, cells, av, _ = run_case(oracle, digests, name)
        ref_fs, ref_av = small[f"{name}__final_state"], small[f"{name}__av_vels"]
        # text round trip: %.12E prints 13 significant digits of the float
        assert np.allclose(av.astype(np.float64), ref_av, rtol=1e-12, atol=0)
        rho = cells.sum(axis=2, dtype=np.float64).reshape(-1)
        pressure = np.where(obst.reshape(-1) != 0, np.float64(np.float32(p.density) * np.float32(1.0 / 3.0)), ref_fs[:, 5])
        assert np.array_equal(ref_fs[:, 6].astype(np.int32), obst.reshape(-1))
        assert np.allclose(rho[obst.reshape(-1) == 0] / 3.0, pressure[obst.reshape(-1) == 0], rtol=1e-6)


@pytest.mark.parametrize("name,av_pct,fs_pct", [("128x128", 0.0809, 0.0702)])
def test_oracle_passes_check_py_against_shipped_goldens(lbm, oracle, digests, tmp_path, name, av_pct, fs_pct):
    """The float reference sits at <=0.25 % of check.py's 1 % budget against the shipped double
    goldens; the restatement must land on the same numbers."""
    p, obst, _, cells, av, _ = run_case(oracle, digests, name)
    fs, avf = str(tmp_path / "final_state.dat"), str(tmp_path / "av_vels.dat")
    oracle.write_final_state(fs, p, cells, obst)
    oracle.write_av_vels(avf, av)
    rep = lbm.checker.check_files(os.path.join(GOLDEN, "check", f"{name}.av_vels.dat.gz"),
                                  os.path.join(GOLDEN, "check", f"{name}.final_state.dat.gz"), avf, fs)
    assert rep.ok, rep.message
    assert abs(abs(rep.av_vels.max_diff_pcnt) - av_pct) < 2e-3
    assert abs(abs(rep.final_state.max_diff_pcnt) - fs_pct) < 2e-3
    assert "Both tests passed!" in rep.message


def test_checker_semantics(lbm):
    ck = lbm.checker
    ref_av = np.array([1.0, 2.0, 3.0])
    coords = np.array([[0, 0, 0.5], [1, 0, 0.25]])
    ok = ck.check_arrays(ref_av, coords, ref_av * 1.005, coords)
    assert ok.ok and abs(ok.av_vels.max_diff_pcnt + 100 * 0.005 / 1.005) < 1e-9     # relative to SIM (check.py:87)
    assert not ck.check_arrays(ref_av, coords, ref_av * 1.02, coords).ok            # > 1 %
    assert ck.check_arrays(ref_av, coords, ref_av * 1.02, coords, tolerance=5.0).ok
    moved = coords.copy(); moved[1, 0] = 2
    assert ck.check_arrays(ref_av, coords, ref_av, moved).message == "Final state files coordinates were not the same"
    assert ck.check_arrays(ref_av, coords, ref_av[:2], coords).message == "Different number of steps in av_vels files"
    zero = ref_av.copy(); zero[1] = 0.0
    assert not ck.check_arrays(ref_av, coords, zero, coords).ok                     # x/0 -> inf -> fail (check.py:134)
    nan = ref_av.copy(); nan[0] = np.nan
    assert not ck.check_arrays(ref_av, coords, nan, coords).ok
    bad_p = coords.copy(); bad_p[0, 2] *= 1.5
    rep = ck.check_arrays(ref_av, coords, ref_av, bad_p)
    assert not rep.ok and "final state failed check" in rep.message and "av_vels failed check" not in rep.message
