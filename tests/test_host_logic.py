"""CPU suite, part 2: the host half of the C ABI (no GPU): exports, parsers with the reference's
error text, decomposition rule, writers, and that the device path refuses to run without a GPU
instead of falling back to anything."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol(lbm):
    header = open(os.path.join(ROOT, "include", "lbm_d2q9.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lbm_[a-z_0-9]+)\s*\(", header))
    assert len(declared) >= 25
    nm = subprocess.run(["nm", "-D", "--defined-only", lbm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (lbm_[a-z_0-9]+)", nm))
    assert declared <= exported, declared - exported
    assert declared == set(lbm.EXPORTS), declared ^ set(lbm.EXPORTS)
    lib = lbm.load_library()
    assert lib.lbm_abi_version() == 5 == lbm._capi.ABI_VERSION


def test_p2p_entry_points_are_exported_by_the_core_library(lbm):
    header = open(os.path.join(ROOT, "include", "lbm_d2q9_p2p.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lbm_p2p_[a-z_0-9]+)\s*\(", header))
    assert declared == {"lbm_p2p_create", "lbm_p2p_handle", "lbm_p2p_connect", "lbm_p2p_disconnect", "lbm_p2p_destroy", "lbm_p2p_run",
                        "lbm_p2p_describe", "lbm_p2p_set_profile", "lbm_p2p_phases", "lbm_p2p_phase_name"}
    nm = subprocess.run(["nm", "-D", "--defined-only", lbm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert declared <= set(re.findall(r" T (lbm_[a-z_0-9]+)", nm))
    assert declared == set(lbm.P2P_EXPORTS)
    # no communication library behind it: the core library does not link RCCL
    needed = subprocess.run(["readelf", "-d", lbm.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "rccl" not in needed.lower()


def test_headers_are_c99(lbm, tmp_path):
    """north_star asks for a thin C host: the three headers must compile as C (gcc -std=c99 -pedantic) and a C
    caller must link against the library and get an answer from the host-only half."""
    src = tmp_path / "caller.c"
    src.write_text('''
#include <stdio.h>
#include "lbm_d2q9.h"
#include "lbm_d2q9_p2p.h"
#include "lbm_d2q9_rccl.h"
int main(void)
{
  int ny_local[3], displs[3];
  lbm_params p = {1024, 190, 10, 10, 0.1f, 0.005f, 1.85f};
  lbm_layout lay;
  if (lbm_abi_version() != LBM_ABI_VERSION) return 2;
  if (lbm_decompose(10, 3, ny_local, displs)) return 3;
  if (lbm_rank_layout(&p, 6, 5, LBM_FLAG_DEFAULT, &lay)) return 4;
  printf("%d %d %d | %d %d %d\\n", ny_local[0], ny_local[1], ny_local[2], lay.y0, lay.ny_local, lay.macro_k);
  {                                     /* the tile (2-D) decomposition's host half, as INTEGRATION.md section 3c uses it */
    lbm_params wide = {16384, 512, 10, 10, 0.1f, 0.005f, 1.85f};
    lbm_tile_layout t;
    int px = 0, py = 0, nxl[8], xd[8];
    if (lbm_choose_rank_grid(&wide, 8, LBM_FLAG_DEFAULT, &px, &py)) return 5;
    if (lbm_tile_layout_of(&wide, px, py, 7, LBM_FLAG_DEFAULT, &t)) return 6;
    if (lbm_decompose_columns(wide.nx, px, nxl, xd)) return 7;
    printf("%d x %d | %d %d %d %d | %d %d %d\\n", px, py, t.x0, t.nx_local, t.y0, t.ny_local, t.ghost, t.ghost_x, t.ghost_y);
  }
  return 0;
}
''')
    exe = tmp_path / "caller"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-L",
                        os.path.dirname(lbm.LIB_PATH), "-llbm_d2q9", f"-Wl,-rpath,{os.path.dirname(lbm.LIB_PATH)}", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout == "4 3 3 | 159 31 0\n8 x 1 | 14336 2048 0 512 | 32 32 0\n", (r.returncode, r.stdout, r.stderr)


def test_rccl_library_exports_every_declared_symbol(lbm):
    header = open(os.path.join(ROOT, "include", "lbm_d2q9_rccl.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(lbm_comm_[a-z_0-9]+)\s*\(", header))
    assert declared == {"lbm_comm_unique_id", "lbm_comm_create", "lbm_comm_destroy", "lbm_comm_run", "lbm_comm_nranks",
                        "lbm_comm_set_step_allreduce"}
    nm = subprocess.run(["nm", "-D", "--defined-only", lbm.LIB_RCCL_PATH], capture_output=True, text=True, check=True).stdout
    assert declared <= set(re.findall(r" T (lbm_[a-z_0-9]+)", nm))
    assert declared == set(lbm.RCCL_EXPORTS)
    lbm.load_rccl_library()


def test_cli_binary_links_and_prints_usage(lbm):
    r = subprocess.run([lbm.CLI_PATH], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr == f"Usage: {lbm.CLI_PATH} <paramfile> <obstaclefile>\n"   # d2q9-bgk.c:1153-1157
    r = subprocess.run([lbm.CLI_PATH, "/nonexistent.params", "x"], capture_output=True, text=True)
    assert r.returncode == 1
    assert re.fullmatch(r"Error at line \d+ of file .*\ncould not open input parameter file: /nonexistent.params\n", r.stderr)


FIELDS = ["nx", "ny", "maxIters", "reynolds_dim", "density", "accel", "omega"]


@pytest.mark.parametrize("n_ok", range(7))
def test_read_params_error_names_the_missing_field(lbm, oracle, tmp_path, n_ok):
    good = ["128", "64", "10", "10", "0.1", "0.005", "1.85"]
    f = tmp_path / "p.params"
    f.write_text("\n".join(good[:n_ok] + ["oops"]) + "\n")
    with pytest.raises(lbm.LbmError, match=f"could not read param file: {FIELDS[n_ok]}$"):
        lbm.read_params(str(f))
    with pytest.raises(RuntimeError, match=f"could not read param file: {FIELDS[n_ok]}$"):
        oracle.read_params(str(f))


def test_read_params_values(lbm, tmp_path):
    f = tmp_path / "p.params"
    f.write_text("1024 1024\n20000\n10\n0.1\n0.01\n1.85\n")     # any whitespace separates tokens (fscanf)
    p = lbm.read_params(str(f))
    assert (p.nx, p.ny, p.max_iters, p.reynolds_dim) == (1024, 1024, 20000, 10)
    assert np.float32(p.density) == np.float32(0.1) and np.float32(p.omega) == np.float32(1.85)


@pytest.mark.parametrize("text,msg", [
    ("1 1\n", "expected 3 values per line in obstacle file"),
    ("8 0 1\n", "obstacle x-coord out of range"),
    ("-1 0 1\n", "obstacle x-coord out of range"),
    ("0 4 1\n", "obstacle y-coord out of range"),
    ("0 0 2\n", "obstacle blocked value should be 1"),
])
def test_read_obstacles_errors(lbm, oracle, tmp_path, text, msg):
    f = tmp_path / "o.dat"
    f.write_text("1 1 1\n" + text)
    with pytest.raises(lbm.LbmError, match=msg):
        lbm.read_obstacles(str(f), 8, 4)
    with pytest.raises(RuntimeError, match=msg):
        oracle.read_obstacles(str(f), 8, 4)
    with pytest.raises(lbm.LbmError, match="could not open input obstacles file"):
        lbm.read_obstacles(str(tmp_path / "missing.dat"), 8, 4)


def test_read_obstacles_duplicates_count_once(lbm, oracle, tmp_path):
    f = tmp_path / "o.dat"
    f.write_text("1 1 1\n2 3 1\n1 1 1\n7 0 1\n")                # duplicate line, as in the shipped decks
    obst, free = lbm.read_obstacles(str(f), 8, 4)
    assert free == 8 * 4 - 3 and obst.sum() == 3 and obst[1, 1] == obst[3, 2] == obst[0, 7] == 1
    o2, f2 = oracle.read_obstacles(str(f), 8, 4)
    assert np.array_equal(obst, o2) and free == f2
    # shipped deck: 512 lines, 508 distinct
    obst, free = lbm.read_obstacles(os.path.join(GOLDEN, "decks", "obstacles_128x128.dat"), 128, 128)
    assert obst.sum() == 508 and free == 128 * 128 - 508


def test_decompose_follows_the_reference_rule(lbm, oracle):
    for ny in [3, 8, 64, 127, 128, 129, 130, 191, 256, 1000, 1024, 8192]:
        for size in [1, 2, 3, 4, 7, 8, 16, 63, 64]:
            if ny // size < 1:
                continue
            a, b = lbm.decompose(ny, size)
            assert (a, b) == oracle.decompose(ny, size)
            assert sum(a) == ny and b[0] == 0 and all(b[i + 1] == b[i] + a[i] for i in range(size - 1))
            if ny // size >= 3 or size == 1 or ny >= 2 * size + 1:
                assert a[-1] >= 3 or size == 1      # accelerate row never on a sent row (d2q9-bgk.c:848-849)
    assert lbm.decompose(1024, 8) == ([128] * 8, [128 * i for i in range(8)])
    assert lbm.decompose(128, 64)[0][-2:] == [1, 3]


def test_rank_layout_is_one_decision_for_all_ranks(lbm):
    """K-step mode and K must not be decided per rank.  Uneven decompositions
    (ny = 190 on 6 ranks: 32,32,32,32,31,31; ny = 127 on 4: 32,32,32,31) and partitions straddling the
    K = 3 / K = 4 size threshold give every rank the same answer."""
    for nx, ny, size in [(1024, 190, 6), (1024, 127, 4), (8192, 8192, 8), (1024, 1024, 8), (2048, 2049, 2), (8192, 515, 2),
                         (130, 100, 3), (126, 400, 4), (8192, 8192, 1), (4096, 1000, 7), (1024, 1024, 16), (256, 200, 2), (1024, 1024, 4), (512, 1024, 2),
                         (2048, 512, 2), (4096, 512, 4)]:
        p = lbm.Params(nx, ny, 10, 10, 0.1, 0.005, 1.85)
        lays = [lbm.rank_layout(p, size, r) for r in range(size)]
        nyl, dis = lbm.decompose(ny, size)
        assert [l["ny_local"] for l in lays] == nyl and [l["y0"] for l in lays] == dis
        assert len({l["macro_k"] for l in lays}) == 1, lays
        # ghost rows and launches per halo exchange (round 4), one answer for all ranks: 2 K rows, two launches for ranks of >= 2 M cells
        # (the edge-stream schedule); below, as deep as the rows carry: 16 rows, four launches from 128 rows per rank, 8 from 64, else K
        big = nx * max(nyl) >= 1 << 21
        want = (0, 1) if lays[0]["macro_k"] == 0 else (8, 2) if big or 64 <= min(nyl) < 128 else (16, 4) if min(nyl) >= 128 else (4, 1)
        if want == (16, 4) and nx <= 2048 and nx * max(nyl) <= 1 << 19:        # the smallest ranks: 24 rows / six launches, 32 / eight from 256 rows per rank
            want = (32, 8) if min(nyl) >= 256 else (24, 6)
        assert all((l["ghost"], l["group"]) == want for l in lays), (nx, ny, size, lays)
        k = lays[0]["macro_k"]
        if size == 1:
            assert k == 0                                            # a whole periodic grid needs no ghost rows ...
            assert lbm.rank_layout(p, 1, 0, lbm._capi.FLAG_FORCE_HALO)["macro_k"] == 4   # ... a 1-rank ring does (K = 4 at every size since round 3)
        elif min(nyl) < 32 or nx % 2 or nx < 128 and nx % 64:
            assert k == 0                                            # one ineligible rank puts EVERY rank in one-step mode
        else:
            assert k == 4                                            # (round 2: 3 above 2 M cells per rank)
        assert all(l["macro_k"] == 0 for l in (lbm.rank_layout(p, size, r, lbm._capi.FLAG_ONE_STEP) for r in range(size)))
    p = lbm.Params(1024, 190, 10, 10, 0.1, 0.005, 1.85)
    assert [lbm.rank_layout(p, 6, r)["macro_k"] for r in range(6)] == [0] * 6       # ranks 4, 5 own 31 rows
    with pytest.raises(lbm.LbmError):
        lbm.rank_layout(p, 6, 6)


def test_obstacle_window_wraps_periodically(lbm):
    obst = np.arange(12 * 4, dtype=np.int32).reshape(12, 4)
    w = lbm.obstacle_window(obst, {"y0": 0, "ny_local": 5, "ghost": 2})
    assert w.shape == (9, 4) and np.array_equal(w[:, 0] // 4, [10, 11, 0, 1, 2, 3, 4, 5, 6])
    w = lbm.obstacle_window(obst, {"y0": 8, "ny_local": 4, "ghost": 3})
    assert np.array_equal(w[:, 0] // 4, [5, 6, 7, 8, 9, 10, 11, 0, 1, 2])
    assert np.array_equal(lbm.obstacle_window(obst, {"y0": 3, "ny_local": 2, "ghost": 0}), obst[3:5])


def test_observables_writer_and_reynolds_equal_the_cells_path(lbm, oracle, tmp_path):
    """lbm_write_final_state_obs / lbm_av_velocity_obs on (u_x, u_y, u, pressure) computed as the device
    kernel computes them must give the bytes / the float of the 9-population path."""
    rng = np.random.default_rng(7)
    p = lbm.Params(24, 10, 3, 7, 0.1, 0.005, 1.3)
    cells = (rng.random((10, 24, 9), dtype=np.float32) * 0.02 + 0.005).astype(np.float32)
    obst = (rng.random((10, 24)) < 0.2).astype(np.int32)
    f = cells
    rho = np.zeros(cells.shape[:2], np.float32)
    for k in range(9):
        rho = rho + f[..., k]
    ux = (f[..., 1] + f[..., 5] + f[..., 8] - (f[..., 3] + f[..., 6] + f[..., 7])) / rho
    uy = (f[..., 2] + f[..., 5] + f[..., 6] - (f[..., 4] + f[..., 7] + f[..., 8])) / rho
    u = np.sqrt((ux * ux + uy * uy).astype(np.float64)).astype(np.float32)
    obs = np.stack([ux, uy, u, rho * np.float32(1.0 / 3.0)], axis=-1).astype(np.float32)
    a, b = str(tmp_path / "a.dat"), str(tmp_path / "b.dat")
    lbm.write_final_state(a, p, cells, obst, displ=3)
    lbm.write_final_state_obs(b, p, obs, obst, displ=3)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert lbm.av_velocity_obs(p, obs, obst) == lbm.av_velocity_host(p, cells, obst) == oracle.av_velocity_sum(p, cells, obst)
    # more than one block of cells: the square roots are taken by several threads, the float accumulation stays serial
    big = lbm.Params(1500, 1100, 3, 7, 0.1, 0.005, 1.3)
    cb = (rng.random((1100, 1500, 9), dtype=np.float32) * 0.02 + 0.005).astype(np.float32)
    ob = (rng.random((1100, 1500)) < 0.1).astype(np.int32)
    rho = cb.sum(axis=-1, dtype=np.float32)          # any positive density will do for this comparison
    ux = ((cb[..., 1] + cb[..., 5] + cb[..., 8] - (cb[..., 3] + cb[..., 6] + cb[..., 7])) / rho).astype(np.float32)
    uy = ((cb[..., 2] + cb[..., 5] + cb[..., 6] - (cb[..., 4] + cb[..., 7] + cb[..., 8])) / rho).astype(np.float32)
    ob4 = np.stack([ux, uy, ux, ux], axis=-1).astype(np.float32)
    acc = np.float32(0.0)
    terms = np.sqrt((ux * ux + uy * uy).astype(np.float64))
    for v in terms[ob == 0]:
        acc = np.float32(np.float64(acc) + v)
    assert lbm.av_velocity_obs(big, ob4, ob) == float(acc)


def test_writers_and_epilogue_match_the_oracle(lbm, oracle, tmp_path):
    rng = np.random.default_rng(5)
    p = lbm.Params(16, 6, 3, 7, 0.1, 0.005, 1.3)
    cells = (rng.random((6, 16, 9), dtype=np.float32) * 0.02 + 0.005).astype(np.float32)
    obst = (rng.random((6, 16)) < 0.2).astype(np.int32)
    a, b = str(tmp_path / "a.dat"), str(tmp_path / "b.dat")
    lbm.write_final_state(a, p, cells, obst)
    oracle.write_final_state(b, p, cells, obst)
    assert open(a, "rb").read() == open(b, "rb").read()
    lbm.write_final_state(a, p, cells[:2], obst[:2], displ=0)
    lbm.write_final_state(a, p, cells[2:], obst[2:], displ=2, append=True)      # rank-by-rank append (d2q9-bgk.c:1054-1057)
    assert open(a, "rb").read() == open(b, "rb").read()
    av = rng.random(5, dtype=np.float32)
    lbm.write_av_vels(a, av)
    oracle.write_av_vels(b, av)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert open(a).readline() == "0:\t%.12E\n" % av[0]
    # special values, many rows (several formatting blocks and threads), every exponent width
    big = lbm.Params(64, 1200, 3, 7, 0.1, 0.005, 1.3)
    cb = (rng.random((1200, 64, 9), dtype=np.float32) * np.float32(10.0) ** rng.integers(-30, 20, (1200, 64, 1))).astype(np.float32)
    cb[0, 0] = 0.0                      # rho = 0 -> nan / inf columns
    cb[0, 1] = [1e-45, 0, 0, 0, 0, 0, 0, 0, 0]   # denormal
    cb[0, 2] = [-1.0, 0.5, 0, 0, 0, 0, 0, 0, 0]
    cb[1, 0, 1] = np.inf
    ob = (rng.random((1200, 64)) < 0.1).astype(np.int32)
    ob[0, :3] = 0
    lbm.write_final_state(a, big, cb, ob, displ=5)
    oracle.write_final_state(b, big, cb, ob, displ=5)
    assert open(a, "rb").read() == open(b, "rb").read()
    assert lbm.av_velocity_host(p, cells, obst) == oracle.av_velocity_sum(p, cells, obst)
    assert lbm.reynolds(p, 0.0123) == oracle.reynolds(p, 0.0123)


def test_device_path_fails_loudly_without_a_gpu(lbm):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = lbm.Params(16, 8, 1, 4, 0.1, 0.005, 1.0)
    with pytest.raises(lbm.LbmError, match="hip"):
        lbm.Partition(p, 128, np.zeros((8, 16), np.int32))


def test_create_validates_arguments(lbm):
    lib = lbm.load_library()
    ctx = C.c_void_p()
    obst = np.zeros((8, 16), np.int32)
    for kw, msg in [(dict(nx=0), "nx must be positive"), (dict(ny=2), "ny must be >= 3")]:
        p = dict(nx=16, ny=8); p.update(kw)
        cp = lbm._capi.CParams(p["nx"], p["ny"], 1, 4, 0.1, 0.005, 1.0)
        rc = lib.lbm_create(C.byref(ctx), C.byref(cp), 100, lbm._capi.as_int_ptr(obst), 0, p["ny"], 0, 0)
        assert rc != 0 and msg in lib.lbm_last_error().decode()


def test_synthetic_deck_is_reproducible(lbm, tmp_path):
    a = lbm.synthetic_obstacles(64, 32, 0.05, 42, True)
    b = lbm.synthetic_obstacles(64, 32, 0.05, 42, True)
    assert np.array_equal(a, b) and a[0].all() and a[-1].all() and a[:, 0].all() and a[:, -1].all()
    assert int(lbm.decks.splitmix64(42, 1)[0]) == 0xBDD732262FEB6E95   # published splitmix64 test vector
    inner = lbm.synthetic_obstacles(512, 512, 0.005, 42, False)
    assert 0.003 < inner.mean() < 0.007
    pp, op = lbm.write_synthetic_deck(str(tmp_path), "t", lbm.Params(64, 32, 5, 4, 0.1, 0.005, 1.85), 0.05, 42)
    obst, free = lbm.read_obstacles(op, 64, 32)
    assert np.array_equal(obst, a) and free == 64 * 32 - a.sum()
    assert lbm.read_params(pp).max_iters == 5


def test_front_ends_parse_arguments_without_a_gpu():
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "d2q9_bgk.py")], capture_output=True, text=True)
    assert r.returncode == 1 and r.stderr.startswith("Usage: ") and r.stderr.rstrip().endswith("<paramfile> <obstaclefile>")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "d2q9_bgk.py"), "/nonexistent.params", "x"], capture_output=True, text=True)
    assert r.returncode == 1                                           # die() format of d2q9-bgk.c:1145-1151
    assert re.fullmatch(r"Error at line \d+ of file d2q9_bgk\.py:\ncould not open input parameter file: /nonexistent.params\n", r.stderr)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--gpus" in r.stdout and "--steps" in r.stdout and "--warmup" in r.stdout
    r = subprocess.run([sys.executable, os.path.join(ROOT, "check", "check.py"), "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "--ref-av-vels-file" in r.stdout


@pytest.mark.parametrize("n", [2, 3])
def test_bench_starts_its_own_rank_processes(n):
    """`python bench.py --gpus N` outside torch.distributed.run (how the driver calls it) must start N fresh
    rank processes itself, before anything touches the GPU, and relay ONE JSON line.  --dry-launch makes the
    ranks rendezvous over gloo and report themselves instead of running the GPU path."""
    import json
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1", "--dry-launch"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["dry_launch"] is True and out["n_gpus"] == n
    assert sorted(d["rank"] for d in out["ranks"]) == list(range(n)) and all(d["world"] == n for d in out["ranks"])
    assert len({d["pid"] for d in out["ranks"]}) == n and os.getpid() not in {d["pid"] for d in out["ranks"]}
    assert out["launch_attempts"][0]["returncode"] == 0


@pytest.mark.parametrize("name", ["recip_exhaustive.hip", "sqrt_exhaustive.hip", "grid_barrier.hip", "xcd_handoff.hip"])
def test_enumeration_programs_compile_for_gfx950(tmp_path, name):
    """The exhaustive checks behind the short reciprocal and the short double sqrt run on the GPU box (test_gpu_parity.py
    compiles them there); here only that they cross-compile, so a typo cannot turn those GPU tests red."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "scripts", "experiments", name)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", src, "-o", str(tmp_path / "a.out")],
                   check=True, capture_output=True, timeout=600)


def test_roofline_json_recomputes_from_the_committed_counter_files():
    """bench.py's two roofline fractions come from profiles/r02/roofline.json; that file must follow from the rocprofv3
    --pmc CSVs committed beside it and from nothing else (the formulas of scripts/make_roofline.py, restated here)."""
    import csv
    import glob
    import json
    d = os.path.join(ROOT, "profiles", "r02")
    roof = json.load(open(os.path.join(d, "roofline.json")))
    kernel = roof["kernel_full_name"]
    mean, dur = {}, {}
    for f in glob.glob(os.path.join(d, "pmc_*.csv")):
        acc = {}
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"] == kernel:
                v, t = acc.setdefault(r["Counter_Name"], ([], []))
                v.append(float(r["Counter_Value"]))
                t.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for name, (v, t) in acc.items():
            mean[name], dur[name] = sum(v) / len(v), sum(t) / len(t) * 1e-9
    for name in ("FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"):
        assert name in mean, name
    hbm = 2.0 * 1024.0 * mean["FETCH_SIZE"] + 1024.0 * mean["WRITE_SIZE"]
    assert abs(hbm / roof["hbm_bytes_per_launch"] - 1.0) < 1e-9
    assert hbm >= 2 * roof["minimum_read_bytes_per_launch"] * 0.99            # every value read once and written once at least
    clock = mean["GRBM_GUI_ACTIVE"] / 8 / dur["GRBM_GUI_ACTIVE"]
    frac_valu = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / (dur["SQ_ACTIVE_INST_VALU"] * clock * 1024)
    assert abs(frac_valu / roof["frac_valu_profiled"] - 1.0) < 1e-9
    frac_hbm = hbm / (0.5 * (dur["FETCH_SIZE"] + dur["WRITE_SIZE"])) / 8.0e12
    assert abs(frac_hbm / roof["frac_hbm_physical_profiled"] - 1.0) < 1e-9
    assert 0.0 < frac_valu < 1.0 and 0.0 < frac_hbm < 1.0                      # fractions of something the chip delivers


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_fallback_chain_and_budget_fit_the_drivers_limit():
    """the driver kills a bench run at 600 s.  Self-launched, the three-mode worst case (every set of rank
    processes hanging until --launch-timeout) must sum below 500 s, the ranks' own budget must end before their launcher
    gives up on them, and a rank started directly by the driver (no launcher) prints what it has at 420 s."""
    bench = _bench_module()
    a = bench.parse_args(["--gpus", "8", "--steps", "20", "--warmup", "5"])
    modes = bench.launch_modes(a)
    assert modes == ["auto", "rccl", "torch"]
    assert len(modes) * a.launch_timeout < 500.0
    assert a.budget_s == 0.0                                        # default: 420 s in a rank, launch_timeout - 25 under the launcher
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "budget = args.budget_s if args.budget_s > 0 else 420.0" in src and 'os.environ.setdefault("LBM_P2P_TIMEOUT_MS", "10000")' in src
    assert bench.parse_args(["--exchange", "p2p"]).exchange == "p2p" and bench.launch_modes(bench.parse_args(["--exchange", "p2p"])) == ["p2p"]


def test_bench_watchdog_prints_the_banked_line_when_the_budget_runs_out(tmp_path):
    """A rank that hangs after its headline (a collective that never returns, a kernel that never ends) must not take the
    line with it: at the end of the budget rank 0 prints what was banked, marked `truncated`, and leaves with exit code 0;
    with nothing banked it prints an error line and leaves with 1.  Exactly one line either way."""
    import json
    import sys
    prog = ("import sys, time, importlib.util\n"
            f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
            "b.quiet_stdout()\n"
            "b.start_watchdog(0, 2, 1.0, time.time())\n"
            "if sys.argv[1] == 'banked':\n"
            "    b.bank({'metric': 'MLUPS', 'value': 1.0})\n"
            "b.stage('variant: rccl')\n"
            "print('noise on stdout')\n"
            "time.sleep(30)\n")
    for what, code in (("banked", 0), ("nothing", 1)):
        r = subprocess.run([sys.executable, "-c", prog, what], capture_output=True, text=True, timeout=60)
        assert r.returncode == code, r.stderr
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        if what == "banked":
            assert out["value"] == 1.0 and out["truncated"] == "budget of 1 s ran out during: variant: rccl"
        else:
            assert "ran out during: variant: rccl" in out["error"] and "value" not in out


def test_bench_power_sampler_reads_the_cards_hwmon_files(tmp_path):
    """`roofline.limits.power` of the line: bench.py's PowerSampler polls a card's hwmon files (socket power, power cap, shader
    clock) during the headline and keeps the samples taken under load.  On a fake /sys/class/drm tree: the card is found by
    its PCI address (or, without one, by the largest clock swing), idle samples are left out, and a tree without hwmon files
    gives None instead of an error."""
    import time as _time
    bench = _bench_module()
    root = tmp_path / "drm"
    cards = {}
    for n, pci in ((0, "0000:05:00.0"), (1, "0000:5a:00.0")):
        real = tmp_path / "pci" / pci
        hw = real / "hwmon" / "hwmon3"
        hw.mkdir(parents=True)
        (root / f"card{n}").mkdir(parents=True)
        os.symlink(real, root / f"card{n}" / "device")
        (hw / "power1_input").write_text("240000000\n")
        (hw / "power1_cap").write_text("1400000000\n")
        (hw / "freq1_input").write_text("95000000\n")
        cards[pci] = hw
    assert bench.PowerSampler("0000:ff:00.0", root=str(root)).start().stop() is None          # no such card: no thread, no result
    for pci_arg in ("0000:5A:00.0", None):
        sm = bench.PowerSampler(pci_arg, root=str(root), period_s=0.001).start()
        _time.sleep(0.03)                                                                    # idle samples
        (cards["0000:5a:00.0"] / "freq1_input").write_text("2228000000\n")
        (cards["0000:5a:00.0"] / "power1_input").write_text("1378000000\n")
        _time.sleep(0.05)
        out = sm.stop()
        (cards["0000:5a:00.0"] / "freq1_input").write_text("95000000\n")
        (cards["0000:5a:00.0"] / "power1_input").write_text("240000000\n")
        assert out["card"] == "0000:5a:00.0" and 0 < out["samples_under_load"] < out["samples"]
        assert out["socket_w_median"] == 1378.0 and out["cap_w"] == 1400.0 and out["sclk_mhz_median"] == 2228.0
        assert abs(out["frac_of_cap"] - 1378 / 1400) < 1e-12
    assert bench.PowerSampler(None, root=str(tmp_path / "nothing")).start().stop() is None


def test_bench_roofline_object_weights_the_launch_mix():
    """ADVICE r02 / VERDICT r02 item 4: a 20-step run is 4 x K=3 + 2 x K=4 launches (or 2 x K=4 + 4 x K=3): every
    instantiation's bytes are divided by ITS OWN live duration, `frac` is the dominant kernel's physical HBM fraction,
    the §8(d) figure (108 B x cells x steps / time) is stated beside it, and the VALU share is labelled a constant."""
    bench = _bench_module()
    cells = 8192 * 8192
    pmc = {"workload": "8192x8192", "kernels": {
        "lbm_multi_kernel<3>": {"steps_per_launch": 3, "hbm_bytes_per_launch": 5.4e9, "frac_valu_profiled": 0.8, "lds_bank_conflict_frac": 0.9},
        "lbm_multi_kernel<4>": {"steps_per_launch": 4, "hbm_bytes_per_launch": 5.9e9, "frac_valu_profiled": 0.85}}}
    prof = [(4, 1500.0), (4, 1480.0), (3, 1010.0), (3, 1000.0), (3, 990.0), (3, 1000.0)]
    r = bench.roofline_object("lbm_multi_kernel<3>", 8192, 8192, float(cells), prof, 1.2e-3, 6, 20, pmc)
    assert r["kernel"] == "lbm_multi_kernel<3>" and r["steps_per_launch"] == 3 and r["bound"] == "hbm" and r["peak"] == 8000.0
    assert abs(r["avg_launch_ms"] - 1.0) < 1e-9 and abs(r["achieved"] - 5400.0) < 1e-6 and abs(r["frac"] - 0.675) < 1e-9 and r["traffic"] == 5.4e9
    assert abs(r["by_section_8d"]["frac"] - 108.0 * cells * 3 / 1.0e-3 / 8.0e12) < 1e-9 and r["by_section_8d"]["frac"] > 2.0
    k4 = r["run_mix"]["K4"]
    assert k4["launches"] == 2 and abs(k4["frac_hbm_physical"] - 5.9e9 / 1.49e-3 / 8.0e12) < 1e-9
    total = (4 * 5.4e9 + 2 * 5.9e9) / (4 * 1.0e-3 + 2 * 1.49e-3) / 8.0e12
    assert abs(r["frac_hbm_physical_run"] - total) < 1e-9
    assert r["limits"]["valu"] == {"frac": 0.8, "kind": "profiled-pass constant", "note": r["limits"]["valu"]["note"]}
    # a partitioned run has no per-launch timing: the whole-run average, bytes scaled by cells and steps per launch
    r2 = bench.roofline_object("lbm_multi_kernel<3>", 8192, 8192, cells / 8.0, None, 150e-6, 7, 20, pmc, scale=1.0 / 8.0)
    assert r2["scaled_from_single_gpu_pmc"] is True and abs(r2["traffic"] - 5.4e9 / 8.0 * (20.0 / 7.0) / 3.0) < 1.0
    assert abs(r2["by_section_8d"]["frac"] - 108.0 * cells / 8.0 * (20.0 / 7.0) / 150e-6 / 8.0e12) < 1e-9


def _pmc_means(directory, workload="8192"):
    """{kernel full name: ({counter: mean per dispatch}, {counter: mean dispatch seconds})} from the committed rocprofv3 --pmc CSVs."""
    import csv
    import glob
    acc = {}
    for f in glob.glob(os.path.join(directory, f"pmc_*_{workload}.csv")):
        for r in csv.DictReader(open(f)):
            if "lbm_multi_kernel" not in r["Kernel_Name"]:
                continue
            v, t = acc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], ([], []))
            v.append(float(r["Counter_Value"]))
            t.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return {k: ({c: sum(v) / len(v) for c, (v, t) in d.items()}, {c: sum(t) / len(t) * 1e-9 for c, (v, t) in d.items()}) for k, d in acc.items()}


def test_saved_driver_style_line_recomputes_from_the_committed_counter_files():
    """every roofline.* fraction of a saved driver-style bench line (profiles/r04/bench_n1_driver_style.json:
    --steps 20 --warmup 5 — five lbm_multi_kernel<4> launches since K = 4 became lbm_run's choice; four K = 3 and two K = 4
    before) follows, within 2 %, from the rocprofv3 --pmc CSVs committed beside it and the launch durations the line itself
    states — bytes of each instantiation over ITS OWN duration, the run's fraction weighted by the launches it made, the
    §8(d) figure, the VALU share."""
    import csv
    import json
    d = os.path.join(ROOT, "profiles", "r04")
    line = json.load(open(os.path.join(d, "bench_n1_driver_style.json")))
    roof = line["roofline"]
    assert line["steps"] == 20 and line["warmup"] == 5 and line["n_gpus"] == 1 and line["config"]["nx"] == 8192 == line["config"]["ny"]
    pmc = _pmc_means(d)
    cells = 8192 * 8192
    mix = roof["run_mix"]
    assert sum(int(k[1:]) * m["launches"] for k, m in mix.items()) == 20            # the launches of the profiled repetition add up to the steps
    full = {int(k[1:]): [n for n in pmc if f"lbm_multi_kernel<{k[1:]}, 2, " in n][0] for k in mix}
    bytes_run = time_run = 0.0
    for k, name in full.items():
        mean, dur = pmc[name]
        hbm = 2.0 * 1024.0 * mean["FETCH_SIZE"] + 1024.0 * mean["WRITE_SIZE"]          # gfx950: FETCH_SIZE reports half of a coalesced read stream
        assert hbm >= 2 * 36 * cells * 0.99                                             # every value read once and written once at least
        m = mix[f"K{k}"]
        t = m["avg_launch_ms"] * 1e-3
        assert abs(m["hbm_bytes_per_launch"] / hbm - 1.0) < 0.02
        assert abs(m["frac_hbm_physical"] / (hbm / t / 8.0e12) - 1.0) < 0.02
        assert abs(m["by_section_8d_frac"] / (108.0 * cells * k / t / 8.0e12) - 1.0) < 0.02
        bytes_run += hbm * m["launches"]
        time_run += t * m["launches"]
    dom = max(full, key=lambda k: k * mix[f"K{k}"]["launches"])                         # the instantiation that advances most of the steps
    assert roof["kernel"] == f"lbm_multi_kernel<{dom}>" and roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] / mix[f"K{dom}"]["frac_hbm_physical"] - 1.0) < 1e-9 and abs(roof["achieved"] / roof["peak"] / roof["frac"] - 1.0) < 1e-9
    assert abs(roof["traffic"] / mix[f"K{dom}"]["hbm_bytes_per_launch"] - 1.0) < 1e-9
    assert abs(roof["frac_hbm_physical_run"] / (bytes_run / time_run / 8.0e12) - 1.0) < 0.02
    assert abs(roof["by_section_8d"]["frac"] / mix[f"K{dom}"]["by_section_8d_frac"] - 1.0) < 1e-9 and roof["by_section_8d"]["frac"] > 1.0
    mean, dur = pmc[full[dom]]
    clock = mean["GRBM_GUI_ACTIVE"] / 8 / dur["GRBM_GUI_ACTIVE"]
    frac_valu = 4.0 * mean["SQ_ACTIVE_INST_VALU"] / (dur["SQ_ACTIVE_INST_VALU"] * clock * 1024)
    assert abs(roof["limits"]["valu"]["frac"] / frac_valu - 1.0) < 0.02 and roof["limits"]["valu"]["kind"] == "profiled-pass constant"
    assert 0.0 < roof["frac"] < 1.0 and 0.0 < frac_valu < 1.0                           # fractions of something the chip delivers
    # the live launch durations agree with the rocprofv3 --kernel-trace --stats summary of the same command (the 200-step
    # line of the same session against the trace's average for the dominant kernel)
    stats = {r["Name"]: r for r in csv.DictReader(open(os.path.join(d, "kernel_stats_bench_8192.csv")))}
    long_line = json.load(open(os.path.join(d, "bench_n1.json")))
    avg_ns = float(stats[full[dom]]["AverageNs"])
    assert long_line["roofline"]["kernel"] == roof["kernel"]
    assert abs(long_line["roofline"]["avg_launch_ms"] * 1e6 / avg_ns - 1.0) < 0.10
    assert int(stats[full[dom]]["Calls"]) >= 20
    # the scaled launch durations come with the bracketed repetition's own, and the factor between them stays inside the band it may take
    assert 0.70 <= roof["launch_time_scale"] <= 1.05 and "launch_time_scale_rejected" not in roof
    assert abs(mix[f"K{dom}"]["avg_launch_ms_bracketed_unscaled"] * roof["launch_time_scale"] / mix[f"K{dom}"]["avg_launch_ms"] - 1.0) < 1e-9
    # the N = 1 line carries BASELINE.json configs 2 - 3 and a sustained figure
    decks = long_line["secondary"]["shipped_decks"]
    assert set(decks) >= {"128x128", "128x256", "256x256", "1024x1024"}
    for name in ("128x128", "128x256", "256x256", "1024x1024"):        # (optional_part adds its own "seconds" beside them)
        r = decks[name]
        assert r["final_state_sha256_equals_reference"] is True and r["reynolds_line_equals_reference"] is True and r["check_py_passes"] is True, name
        assert r["av_vels_vs_shipped_golden_max_pct"] < 1.0 and r["seconds"] > 0 and abs(r["value"] * r["seconds"] * 1e6 / (r["steps"] * int(name.split("x")[0]) * int(name.split("x")[1])) - 1.0) < 1e-9
    sus = long_line["sustained"]
    assert sus["steps_per_rep"] >= 2000 and sus["reps"] == 3 and 0.9 < sus["value"] / long_line["value"] <= 1.02
    # ... and BASELINE.json config 3's "rocprof HBM GB/s vs roofline" at 1024 x 1024: its own PMC passes (Infinity-Cache resident: 72 MiB of state)
    r1024 = json.load(open(os.path.join(d, "roofline_1024x1024.json")))
    mean, dur = _pmc_means(d, "1024")[r1024["kernel_full_name"]]
    hbm = 2.0 * 1024.0 * mean["FETCH_SIZE"] + 1024.0 * mean["WRITE_SIZE"]
    assert r1024["workload"] == "1024x1024" and r1024["kernel"] == "lbm_multi_kernel<4>" and abs(r1024["hbm_bytes_per_launch"] / hbm - 1.0) < 1e-6
    assert hbm >= 2 * 36 * 1024 * 1024 * 0.97 and r1024["hbm_bytes_per_cell_step"] < 21.0


def test_step_plans_end_in_fours_and_threes(lbm):
    """lbm_plan_steps — the one rule lbm_run, the split-phase macro-steps and the peer-to-peer loop cut a run by: K at a time,
    and where four rows are there (whole grids; partitions with four ghost rows) a count K does not divide ends in 4s and 3s
    instead of a 1- or 2-step launch (1050 / 1290 us per 3- / 4-step launch of the 8192 x 8192 deck against 870 / 1000 for a
    1- / 2-step one).  Every plan adds up; short launches only where no mix of 3s and 4s exists (n = 1, 2, 5)."""
    assert lbm.plan_steps(4, 20) == [4, 4, 4, 4, 4]
    assert lbm.plan_steps(4, 21) == [3, 3, 3, 4, 4, 4]                # the odd launches come first
    assert lbm.plan_steps(4, 22) == [3, 3, 4, 4, 4, 4]
    assert lbm.plan_steps(4, 23) == [3, 4, 4, 4, 4, 4]
    assert lbm.plan_steps(3, 20) == [4, 4, 3, 3, 3, 3]
    assert lbm.plan_steps(3, 20, four_rows=False) == [3, 3, 3, 3, 3, 3, 2]
    assert lbm.plan_steps(4, 6, four_rows=False) == [4, 2]
    assert lbm.plan_steps(2, 7) == [2, 2, 2, 1] and lbm.plan_steps(4, 0) == []
    for K in (3, 4):
        for n in range(1, 200):
            plan = lbm.plan_steps(K, n)
            assert sum(plan) == n and all(1 <= k <= 4 for k in plan)
            if n not in (1, 2, 5):
                assert min(plan) >= 3, (K, n, plan)
            plain = lbm.plan_steps(K, n, four_rows=False)
            assert sum(plain) == n and all(k == K for k in plain[:-1]) and 1 <= plain[-1] <= K
    with pytest.raises(lbm.LbmError):
        lbm.plan_steps(5, 10)


def test_only_a_failed_rendezvous_is_ever_run_again():
    """conftest.is_rendezvous_failure — the one condition under which a test may start its rank processes a second time: they never
    formed their group.  A bounded wait of the peer-to-peer loop that ran out (the visible form of a hang), an LbmError of any kind,
    an assert in the worker, a GPU fault, a rank killed by a signal, a parity FAILED line: each is a verdict and stays one."""
    from conftest import is_rendezvous_failure as again
    gloo = "RuntimeError: [../third_party/gloo/gloo/transport/tcp/pair.cc:144] Gloo connectFullMesh failed with Connection refused"
    summary = "Root Cause (first observed failure):\n[0]:\n  rank      : 1 (local_rank: 1)\n  exitcode  : {code} (pid: 77)\n"
    assert again(1, "", gloo + "\n" + summary.format(code=1))
    assert again(1, "", "torch.distributed.DistNetworkError: The client socket has timed out after 600s\n")
    assert again(1, "", "Other Failures:\n  exitcode  : -15 (pid: 76)\n" + gloo + "\n" + summary.format(code=1))    # the agent's SIGTERMs do not count
    assert not again(0, "", gloo)                                                                    # it passed
    assert not again(1, "RANK 0 UP\n", gloo)                                                         # the group had formed
    assert not again(1, "", "mpilattice-boltzmann_amd._capi.LbmError: lbm_p2p_run: rank 1: a peer's data did not arrive in time [code 1]\n" + gloo)
    assert not again(1, "", "lbm_p2p_run: rank 0: a neighbour's grids are out of step with this rank's\n" + gloo)
    assert not again(1, "CASE 0 FAILED ranks=2\n", gloo)
    assert not again(1, "", "AssertionError: {'loop': 'rccl'}\n" + gloo)
    assert not again(1, "", "Memory access fault by GPU node-2\n" + gloo)
    assert not again(1, "", gloo + "\n" + summary.format(code=-11))                                  # the first rank to fail died of a signal
    assert not again(1, "", "something else went wrong\n" + summary.format(code=1))                  # no rendezvous text at all


def test_groups_of_launches_between_exchanges(lbm, monkeypatch):
    """lbm_plan_group — what a partitioned run does between two halo exchanges (d2q9-bgk.c:326-328,364): the launches of
    lbm_plan_steps for as long as their steps add up to at most the ghost rows (the first launch of a group advances the ghost
    rows the later ones read).  K = 4 on 8 rows: two launches per exchange; rounds 1-3's loop is the case ghost = K."""
    assert lbm.plan_groups(4, 8, 2, 20) == [[4, 4], [4, 4], [4]]
    assert lbm.plan_groups(4, 8, 2, 21) == [[3, 3], [3, 4], [4, 4]]
    assert lbm.plan_groups(4, 4, 1, 10) == [[3], [3], [4]] and lbm.plan_groups(4, 8, 1, 10) == [[3], [3], [4]]
    assert lbm.plan_groups(4, 7, 2, 16) == [[4], [4], [4], [4]]                      # 4 + 4 > 7
    assert lbm.plan_groups(4, 7, 2, 14) == [[3, 3], [4], [4]]
    assert lbm.plan_groups(3, 8, 2, 20) == [[4, 4], [3, 3], [3, 3]]
    assert lbm.plan_groups(4, 16, 4, 50) == [[3, 3, 4, 4], [4, 4, 4, 4], [4, 4, 4, 4], [4]]
    assert lbm.plan_groups(2, 4, 2, 7) == [[2, 2], [2, 1]] and lbm.plan_groups(1, 2, 2, 3) == [[1, 1], [1]]
    for K in (1, 2, 3, 4):
        for ghost in range(K, 17):
            for group in (1, 2, 3, 8):
                for n in (1, 2, 5, 19, 20, 21, 22, 23, 64, 101):
                    groups = lbm.plan_groups(K, ghost, group, n)
                    flat = [k for g in groups for k in g]
                    assert flat == lbm.plan_steps(K, n, four_rows=ghost >= 4)                      # the same launches, only grouped
                    assert all(1 <= len(g) <= group and sum(g) <= ghost for g in groups)
                    # greedy: a group ends only where the next launch would not fit (or the cap is reached, or the run ends)
                    for g, nxt in zip(groups, groups[1:]):
                        assert len(g) == group or sum(g) + nxt[0] > ghost
    with pytest.raises(lbm.LbmError):
        lbm.plan_groups(4, 3, 1, 10)                                                                # fewer ghost rows than a launch makes steps
    # the layout's answer follows the environment knobs, identically on every rank
    p = lbm.Params(8192, 8192, 10, 10, 0.1, 0.005, 1.85)
    monkeypatch.setenv("LBM_TUNE_MACRO_GHOST", "12")
    assert [(l["ghost"], l["group"]) for l in (lbm.rank_layout(p, 8, r) for r in range(8))] == [(12, 3)] * 8
    monkeypatch.setenv("LBM_TUNE_MACRO_GROUP", "2")
    assert lbm.rank_layout(p, 8, 3)["group"] == 2
    monkeypatch.setenv("LBM_TUNE_MACRO_GHOST", "0")                                                 # rounds 1-3: K rows, one launch per exchange
    monkeypatch.delenv("LBM_TUNE_MACRO_GROUP")
    assert (lbm.rank_layout(p, 8, 3)["ghost"], lbm.rank_layout(p, 8, 3)["group"]) == (4, 1)
    monkeypatch.setenv("LBM_TUNE_MACRO_GHOST", "99")
    assert lbm.rank_layout(p, 8, 3)["ghost"] == 32                                                  # kMaxGhost


def test_tile_layout_is_one_decision_for_all_ranks(lbm, monkeypatch):
    """The tile (2-D) decomposition (the reference's report discusses it, d2q9-bgk.c:834-862 splits rows only):
    rows by the reference's rule over py, columns in whole x-pairs over px; the blocks tile the grid exactly; K, ghost rows, ghost
    columns (ghost rows rounded up to even) and launches per exchange are the same on every rank; a grid that would leave a rank
    outside K-step mode is an error that names the row decomposition."""
    import ctypes as C
    lib = lbm._capi.load_library()
    for nx, px in [(1024, 4), (1290, 3), (644, 2), (8192, 8), (130, 1), (10, 5)]:
        n, d = (C.c_int * px)(), (C.c_int * px)()
        assert lib.lbm_decompose_columns(nx, px, n, d) == 0
        n, d = list(n), list(d)
        assert sum(n) == nx and all(v % 2 == 0 and v > 0 for v in n) and max(n) - min(n) <= 2 and d == [sum(n[:i]) for i in range(px)]
    assert lib.lbm_decompose_columns(129, 2, (C.c_int * 2)(), (C.c_int * 2)()) != 0           # odd nx: no whole pairs
    assert lib.lbm_decompose_columns(8, 5, (C.c_int * 5)(), (C.c_int * 5)()) != 0             # fewer pairs than ranks
    for nx, ny, px, py in [(8192, 8192, 4, 2), (8192, 8192, 2, 4), (1024, 1024, 4, 2), (1290, 200, 3, 2), (512, 256, 1, 1), (16384, 512, 8, 1), (768, 384, 3, 2)]:
        p = lbm.Params(nx, ny, 10, 10, 0.1, 0.005, 1.85)
        lays = [lbm.tile_layout(p, px, py, r) for r in range(px * py)]
        nyl, dis = lbm.decompose(ny, py)
        cover = np.zeros((ny, nx), dtype=np.int32)
        for r, l in enumerate(lays):
            assert (l["rx"], l["ry"]) == (r % px, r // px) and (l["px"], l["py"]) == (px, py)
            assert (l["ny_local"], l["y0"]) == (nyl[l["ry"]], dis[l["ry"]])
            cover[l["y0"]:l["y0"] + l["ny_local"], l["x0"]:l["x0"] + l["nx_local"]] += 1
        assert np.all(cover == 1)
        assert len({(l["macro_k"], l["ghost"], l["ghost_x"], l["group"]) for l in lays}) == 1
        l = lays[0]
        assert l["macro_k"] == 4 and l["ghost_x"] == (l["ghost"] + 1) // 2 * 2 and l["ghost"] <= 32
        # as a row partition of the same cells: 8 ghost rows for ranks of >= 2 M cells, else as deep as the rows carry — and 32 ghost COLUMNS
        # (eight launches per exchange) for column blocks below 2 M cells, which keep no ghost rows at all
        big = max(x["nx_local"] for x in lays) * max(nyl) >= 1 << 21
        if py == 1 and not big and min(x["nx_local"] for x in lays) >= 256:
            assert (l["ghost"], l["ghost_x"], l["ghost_y"], l["group"]) == (32, 32, 0, 8)
        else:
            assert l["ghost"] == (8 if big or 64 <= min(nyl) < 128 else 16 if min(nyl) >= 128 else 4) and l["ghost_y"] == (0 if py == 1 else l["ghost"])
    monkeypatch.setenv("LBM_TUNE_MACRO_GHOST", "7")
    assert {(l["ghost"], l["ghost_x"], l["group"]) for l in (lbm.tile_layout(lbm.Params(640, 300, 1, 1, 0.1, 0.005, 1.85), 2, 3, r) for r in range(6))} == {(7, 8, 1)}
    monkeypatch.delenv("LBM_TUNE_MACRO_GHOST")
    for nx, ny, px, py in [(128, 128, 2, 2), (1024, 60, 2, 2), (1022, 512, 4, 1), (1024, 1024, 3, 2)]:      # 80-float rows; 30-row ranks; odd pairs... all fine but:
        p = lbm.Params(nx, ny, 10, 10, 0.1, 0.005, 1.85)
        if (nx, ny, px, py) in [(1022, 512, 4, 1), (1024, 1024, 3, 2)]:
            assert lbm.tile_layout(p, px, py, 0)["macro_k"] == 4                              # uneven column blocks are fine
            continue
        with pytest.raises(lbm.LbmError, match="row decomposition"):
            lbm.tile_layout(p, px, py, 0)
    with pytest.raises(lbm.LbmError, match="row decomposition"):
        lbm.tile_layout(lbm.Params(512, 256, 10, 10, 0.1, 0.005, 1.85), 2, 2, 0, lbm._capi.FLAG_ONE_STEP)
    with pytest.raises(lbm.LbmError):
        lbm.tile_layout(lbm.Params(512, 256, 10, 10, 0.1, 0.005, 1.85), 2, 2, 4)


def test_rank_grid_choice_follows_the_measurements(lbm):
    """lbm_choose_rank_grid: the reference's row blocks (d2q9-bgk.c:834-862) wherever they measured faster — every BASELINE.json config, and small
    decks on many ranks — and tiles where they did: wide column blocks in place of row blocks of 128 rows or fewer, the grids much wider than
    tall that SURVEY.md section 8(f) row 3 names (DESIGN.md 6.5 lists the measured pair behind each line); the same answer on every rank."""
    P = lambda nx, ny: lbm.Params(nx, ny, 10, 10, 0.1, 0.005, 1.85)
    for nx, ny, n in [(8192, 8192, 8), (8192, 8192, 4), (8192, 8192, 2), (8192, 8192, 1),      # 8192 x 1024 rows 43.8 us/step against 46.0 as 1024 x 8192 column blocks
                      (1024, 1024, 8), (1024, 1024, 4), (1024, 1024, 2),                       # 1024 x 128 rows 3.28 against 4.05 as 128 x 1024; 1024 x 256 3.99 / 4.93
                      (1024, 1024, 16),                                                        # 1024 x 64 rows 3.25 against 3.51 as 512 x 128 tiles
                      (128, 128, 4), (256, 256, 2), (128, 256, 2)]:                            # the small shipped decks: no tiling is eligible
        assert lbm.choose_rank_grid(P(nx, ny), n) is None, (nx, ny, n)
    assert lbm.choose_rank_grid(P(16384, 512), 8) == (8, 1)          # 64-row blocks 13.2 us/step against 8.7 as 2048 x 512 column blocks
    assert lbm.choose_rank_grid(P(32768, 256), 8) == (8, 1)          # 32-row blocks: 18.3 against 9.4
    assert lbm.choose_rank_grid(P(65536, 128), 8) == (8, 1)          # 16-row blocks (one-step loop): 21.8 against 9.5
    assert lbm.choose_rank_grid(P(2048, 512), 4) == (4, 1)           # 2048 x 128 rows 4.36 against 4.01 as 512 x 512
    assert lbm.choose_rank_grid(P(2048, 512), 8) == (8, 1)           # 2048 x 64 rows 4.12 against 3.31 as 256 x 512
    assert lbm.choose_rank_grid(P(4096, 512), 4) == (4, 1)           # 4096 x 128 rows 6.15 against 5.37 as 1024 x 512
    assert lbm.choose_rank_grid(P(8192, 256), 4) == (4, 1)           # 8192 x 64 rows 7.45 against 5.64 as 2048 x 256
    px, py = lbm.choose_rank_grid(P(4096, 4096), 64)                 # 64-row blocks; 64-column blocks are not eligible: tiles with ghost rows
    assert px * py == 64 and px > 1 and py > 1 and lbm.tile_layout(P(4096, 4096), px, py, 0)["ny_local"] >= 128
    assert lbm.choose_rank_grid(P(16384, 512), 8, lbm._capi.FLAG_ONE_STEP) is None       # no K-step mode, no tiles
    with pytest.raises(lbm.LbmError):
        lbm.choose_rank_grid(P(16384, 512), 0)


def test_rank_grid_choice_is_always_a_runnable_decomposition(lbm):
    """Property over random grids and rank counts: whatever lbm_choose_rank_grid answers can be created — the row decomposition always can
    (lbm_rank_layout), a tile grid is px x py = nranks with every rank's layout defined, the blocks tiling the grid exactly and all ranks
    agreeing on K, the ghost depths and the launches per exchange; the answer does not depend on which rank asks (no rank argument at all)."""
    rng = np.random.default_rng(11)
    seen_tiles = 0
    for _ in range(300):
        nx = 2 * int(rng.integers(8, 20000)) if rng.random() < 0.8 else int(rng.integers(1, 5000))
        ny = int(rng.integers(3, 6000))
        n = int(rng.choice([1, 2, 3, 4, 6, 8, 12, 16, 32, 64]))
        if ny < n:
            continue
        p = lbm.Params(nx, ny, 10, 10, 0.1, 0.005, 1.85)
        grid = lbm.choose_rank_grid(p, n)
        assert grid == lbm.choose_rank_grid(p, n)
        if grid is None:
            lays = [lbm.rank_layout(p, n, r) for r in range(n)]
            assert sum(l["ny_local"] for l in lays) == ny
            continue
        seen_tiles += 1
        px, py = grid
        assert px > 1 and px * py == n
        lays = [lbm.tile_layout(p, px, py, r) for r in range(n)]
        assert sum(l["nx_local"] * l["ny_local"] for l in lays) == nx * ny
        assert len({(l["macro_k"], l["ghost"], l["ghost_x"], l["ghost_y"], l["group"]) for l in lays}) == 1 and lays[0]["macro_k"] > 0
        assert all(l["nx_local"] % 2 == 0 and l["nx_local"] >= l["ghost_x"] for l in lays)
        assert lays[0]["ghost_y"] == (0 if py == 1 else lays[0]["ghost"])
    assert seen_tiles > 20


def test_tile_obstacle_window_wraps_in_both_directions(lbm):
    p = lbm.Params(512, 256, 10, 10, 0.1, 0.005, 1.85)
    obst = np.arange(256 * 512, dtype=np.int32).reshape(256, 512)
    for r in range(4):
        l = lbm.tile_layout(p, 2, 2, r)
        w = lbm.obstacle_window(obst, l)
        assert w.shape == (l["ny_local"] + 2 * l["ghost"], l["nx_local"] + 2 * l["ghost_x"])
        for (i, j) in [(0, 0), (l["ghost"], l["ghost_x"]), (w.shape[0] - 1, w.shape[1] - 1), (3, w.shape[1] - 2)]:
            assert w[i, j] == obst[(l["y0"] - l["ghost"] + i) % 256, (l["x0"] - l["ghost_x"] + j) % 512]


def test_shipped_library_carries_no_experiment_kernels(lbm):
    """lbm_sweep_kernel and lbm_step_kernel_lds measured slower than what runs by default and are compiled only with -DLBM_EXPERIMENTS=1
    (scripts/build_variant.sh experiments; their parity tests: tests/experiments_suite.py).  The shipped code object holds neither."""
    blob = open(lbm.LIB_PATH, "rb").read()
    assert b"lbm_multi_kernel" in blob and b"lbm_tile_kernel" in blob and b"lbm_step_kernel" in blob
    assert b"lbm_sweep_kernel" not in blob and b"lbm_step_kernel_lds" not in blob
