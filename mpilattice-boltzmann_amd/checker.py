"""Python-3 restatement of the reference's acceptance check (`check/check.py`, Python 2 only).

Semantics followed line by line (`check/check.py:62-147`):
  * av_vels: column 1 of `av_vels.dat`; final state: columns 0, 1, 5 (x, y, pressure) of
    `final_state.dat` (`:65-66`) — the velocity columns are not checked;
  * fail if the (x, y) columns differ anywhere (`:75-77`) or the step counts differ (`:80-82`);
  * diff = ref - sim; diff_pcnt = 100 * diff / (ref - diff) (i.e. relative to sim) (`:86-87`);
  * the reported entry is argmax |diff_pcnt| (`:89`); a check fails when that percentage is not
    finite or exceeds the tolerance, default 1 % (`:26-31,134-135`).
Usable as a library (arrays in, report out) and as a CLI with the reference's flags.
"""
from __future__ import annotations

import argparse
import gzip
import sys
from dataclasses import dataclass

import numpy as np


def _open(path: str):
    return gzip.open(path, "rt") if path.endswith(".gz") else open(path, "r")


def load_av_vels(path: str) -> np.ndarray:
    with _open(path) as fh:
        return np.loadtxt(fh, usecols=[1], ndmin=1)                  # check.py:65


def load_final_state(path: str) -> np.ndarray:
    with _open(path) as fh:
        return np.loadtxt(fh, usecols=[0, 1, 5], ndmin=2)            # check.py:66


@dataclass
class Diff:
    """`get_diff_values` (`check/check.py:84-101`)."""

    max_diff_step: int
    max_diff: float
    max_diff_pcnt: float
    sim_val: float
    ref_val: float
    total: float

    def failed(self, tolerance: float) -> bool:                      # check.py:134-135
        return (not np.isfinite(self.max_diff_pcnt)) or abs(self.max_diff_pcnt) > tolerance


def diff_values(ref_vals: np.ndarray, sim_vals: np.ndarray) -> Diff:
    ref_vals = np.asarray(ref_vals, dtype=np.float64)
    sim_vals = np.asarray(sim_vals, dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        diff = ref_vals - sim_vals                                   # check.py:86
        diff_pcnt = 100.0 * (diff / (ref_vals - diff))               # check.py:87
    # np.argmax treats NaN as the maximum in Python 2-era numpy as well: a NaN anywhere is reported
    step = int(np.argmax(np.abs(diff_pcnt)))                         # check.py:89
    return Diff(step, float(diff[step]), float(diff_pcnt[step]), float(sim_vals[step]), float(ref_vals[step]),
                float(np.sum(np.abs(diff))))


@dataclass
class Report:
    ok: bool
    message: str
    av_vels: Diff | None = None
    final_state: Diff | None = None


def check_arrays(ref_av: np.ndarray, ref_final: np.ndarray | None, sim_av: np.ndarray,
                 sim_final: np.ndarray | None, tolerance: float = 1.0) -> Report:
    """ref_final / sim_final are (n, 3) arrays of x, y, pressure; pass None for both to check
    av_vels only (two of the reference's shipped final-state goldens are absent)."""
    lines = []
    fs = None
    if ref_final is not None:
        if ref_final.shape != sim_final.shape or np.any(ref_final[:, 0:2] != sim_final[:, 0:2]):   # check.py:75-77
            return Report(False, "Final state files coordinates were not the same")
    if ref_av.size != sim_av.size:                                                                # check.py:80-82
        return Report(False, "Different number of steps in av_vels files")
    av = diff_values(ref_av, sim_av)
    lines += [f"Total difference in av_vels : {av.total:.12E}",                                   # check.py:104-108
              f"Biggest difference (at step {av.max_diff_step:d}) : {av.max_diff:.12E}",
              f"  {av.sim_val:.12E} vs. {av.ref_val:.12E} = {av.max_diff_pcnt:.2g}%", ""]
    failed = av.failed(tolerance)
    if ref_final is not None:
        fs = diff_values(ref_final[:, 2], sim_final[:, 2])                                        # check.py:114
        jj, ii = int(sim_final[fs.max_diff_step, 0]), int(sim_final[fs.max_diff_step, 1])         # check.py:122-125
        lines += [f"Total difference in final_state : {fs.total:.12E}",
                  f"Biggest difference (at coord ({jj:d},{ii:d})) : {fs.max_diff:.12E}",
                  f"  {fs.sim_val:.12E} vs. {fs.ref_val:.12E} = {fs.max_diff_pcnt:.2g}%", ""]
        if fs.failed(tolerance):
            lines.append("final state failed check")                                              # check.py:137-138
            failed = True
    if av.failed(tolerance):
        lines.append("av_vels failed check")                                                      # check.py:139-140
    if not failed:
        lines.append("Both tests passed!" if ref_final is not None else "av_vels test passed!")   # check.py:146
    return Report(not failed, "\n".join(lines), av, fs)


def check_files(ref_av_vels_file: str, ref_final_state_file: str | None, av_vels_file: str,
                final_state_file: str | None, tolerance: float = 1.0) -> Report:
    ref_av, sim_av = load_av_vels(ref_av_vels_file), load_av_vels(av_vels_file)
    ref_fs = sim_fs = None
    if ref_final_state_file is not None:
        ref_fs, sim_fs = load_final_state(ref_final_state_file), load_final_state(final_state_file)
    return check_arrays(ref_av, ref_fs, sim_av, sim_fs, tolerance)


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(description="Testing script for HPC LBM coursework (Python 3 restatement)")
    ap.add_argument("--tolerance", type=float, default=1.0, help="Percentage tolerance to match against reference results")
    ap.add_argument("--ref-av-vels-file", required=True)
    ap.add_argument("--ref-final-state-file", default=None, help="omit to check av_vels only")
    ap.add_argument("--av-vels-file", required=True)
    ap.add_argument("--final-state-file", default=None)
    a = ap.parse_args(argv)
    rep = check_files(a.ref_av_vels_file, a.ref_final_state_file, a.av_vels_file, a.final_state_file, a.tolerance)
    print(rep.message)
    return 0 if rep.ok else 1                                                                     # check.py:143-147


if __name__ == "__main__":
    sys.exit(main())
