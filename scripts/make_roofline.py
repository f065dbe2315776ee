#!/usr/bin/env python3
"""profiles/<tag>/roofline.json from the rocprofv3 --pmc passes of scripts/gpu_session.sh's `round` step.

    python scripts/make_roofline.py <tag> gpurun_out/pmc_<tag>_* [--workload 8192x8192]

Every directory holds one `--pmc` pass (rocprofv3 `*counter_collection.csv`).  For EVERY step-kernel instantiation
the passes dispatched (a 20-step run of the 8192 x 8192 deck is 4 x lbm_multi_kernel<3> + 2 x lbm_multi_kernel<4>) the
per-dispatch values of every counter are averaged — `kernels[<short name>]` — and the two limits a kernel can be held
against are derived, both recomputable from the CSVs copied next to the JSON, nothing else:

  HBM    bytes per launch = 2 x 1024 x FETCH_SIZE + 1024 x WRITE_SIZE   (KB counters; gfx950 reports half of a
         coalesced read stream: MI355X_MICROARCH.md §HBM; lower-bound check: every source value is read at
         least once per launch = 36 B x cells)
         frac_hbm_physical = bytes / launch duration / 8.0 TB/s
  VALU   busy cycles per launch = 4 x SQ_ACTIVE_INST_VALU  (the counter is in quad-cycles, summed over SIMDs)
         frac_valu = busy cycles / (launch duration x clock x 1024 SIMDs),
         clock = GRBM_GUI_ACTIVE / 8 XCDs / launch duration of the same pass
  (`valu_issue_estimate`: SQ_INSTS_VALU x 4 cycles over the same denominator — what the instruction count
  alone predicts when every instruction is a 4-cycle packed one.)

bench.py divides the per-launch numerators by ITS OWN launch time (HIP events, un-profiled clock)."""
import argparse
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK = 8.0e12
SIMDS = 1024
XCDS = 8


def read_pass(d):
    """{kernel name: {counter: (values per dispatch, durations ns)}}, source file — every step-kernel instantiation of the pass."""
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        return {}, None
    out = {}
    for r in csv.DictReader(open(files[0])):
        if not re.search(r"lbm_(multi|step|tile)_kernel", r["Kernel_Name"]):
            continue
        vals, durs = out.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], ([], []))
        vals.append(float(r["Counter_Value"]))
        durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out, files[0]


def short_name(full):
    m = re.search(r"lbm_multi_kernel<(\d+)", full)
    if m:
        return f"lbm_multi_kernel<{m.group(1)}>", int(m.group(1))
    m = re.search(r"(lbm_\w+_kernel\w*)", full)
    return (m.group(1) if m else full.split("(")[0]), 1


def derive(mean, dur_ns, nx, ny, steps):
    """The per-launch quantities of one kernel from its counter means (formulas of the module docstring)."""
    out = {"steps_per_launch": steps, "minimum_read_bytes_per_launch": 36 * nx * ny, "algorithmic_bytes_per_launch_108B": 108 * nx * ny * steps}
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        rd, wr = 2.0 * 1024.0 * mean["FETCH_SIZE"], 1024.0 * mean["WRITE_SIZE"]
        out.update({"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
                    "hbm_bytes_per_cell_step": (rd + wr) / (nx * ny * steps),
                    "frac_hbm_physical_profiled": (rd + wr) / (0.5 * (dur_ns["FETCH_SIZE"] + dur_ns["WRITE_SIZE"]) * 1e-9) / HBM_PEAK})
    if "GRBM_GUI_ACTIVE" in mean:
        out["clock_hz"] = mean["GRBM_GUI_ACTIVE"] / XCDS / (dur_ns["GRBM_GUI_ACTIVE"] * 1e-9)
    clock = out.get("clock_hz", 2.4e9)
    if "SQ_ACTIVE_INST_VALU" in mean:
        busy = 4.0 * mean["SQ_ACTIVE_INST_VALU"]
        out["valu_busy_cycles_per_launch"] = busy
        out["frac_valu_profiled"] = busy / (dur_ns["SQ_ACTIVE_INST_VALU"] * 1e-9 * clock * SIMDS)
    if "SQ_INSTS_VALU" in mean:
        out["valu_insts_per_launch"] = mean["SQ_INSTS_VALU"]
        out["valu_issue_estimate_profiled"] = 4.0 * mean["SQ_INSTS_VALU"] / (dur_ns["SQ_INSTS_VALU"] * 1e-9 * clock * SIMDS)
    if "SQ_LDS_BANK_CONFLICT" in mean and mean.get("SQ_ACTIVE_INST_LDS"):
        out["lds_bank_conflict_frac"] = mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_ACTIVE_INST_LDS"]
    if "SQ_LDS_BANK_CONFLICT" in mean and mean.get("SQ_LDS_IDX_ACTIVE"):
        out["lds_bank_conflict_of_idx_active"] = mean["SQ_LDS_BANK_CONFLICT"] / mean["SQ_LDS_IDX_ACTIVE"]
    if "SQ_WAIT_ANY" in mean and mean.get("SQ_WAVE_CYCLES"):
        out["wait_any_of_wave_cycles"] = mean["SQ_WAIT_ANY"] / mean["SQ_WAVE_CYCLES"]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--workload", default="8192x8192")
    ap.add_argument("--suffix", default="", help="roofline<suffix>.json (a second workload of the same round, e.g. _1024x1024)")
    a = ap.parse_args()
    nx, ny = (int(v) for v in a.workload.split("x"))
    dst = os.path.join(ROOT, "profiles", a.tag)
    os.makedirs(dst, exist_ok=True)
    mean, dur_ns, sources, calls = {}, {}, {}, {}            # per kernel (full name)
    for d in a.dirs:
        per_kernel, f = read_pass(d)
        if not per_kernel:
            continue
        names = sorted({c for k in per_kernel.values() for c in k})
        copy = os.path.join(dst, "pmc_" + "_".join(names).lower()[:80] + f"_{nx}.csv")
        shutil.copyfile(f, copy)
        for kernel, counters in per_kernel.items():
            for cname, (vals, durs) in counters.items():
                mean.setdefault(kernel, {})[cname] = sum(vals) / len(vals)
                dur_ns.setdefault(kernel, {})[cname] = sum(durs) / len(durs)
                calls[kernel] = max(calls.get(kernel, 0), len(vals))
                sources.setdefault(kernel, {})[cname] = {"file": os.path.relpath(copy, ROOT), "dispatches": len(vals), "min": min(vals),
                                                         "max": max(vals), "mean_dispatch_ns": dur_ns[kernel][cname]}
    kernels = {}
    for kernel in mean:
        short, steps = short_name(kernel)
        kernels[short] = dict(derive(mean[kernel], dur_ns[kernel], nx, ny, steps), kernel_full_name=kernel, dispatches_profiled=calls[kernel],
                              counters_mean_per_launch=mean[kernel], source=sources[kernel])
    # the dominant kernel (advances the most steps over the profiled passes) is repeated at the top level
    dom = max(kernels, key=lambda k: kernels[k]["steps_per_launch"] * kernels[k]["dispatches_profiled"])
    out = {"workload": a.workload, "round": a.tag, "simds": SIMDS, "kernel": dom, "kernels": kernels}
    out.update({k: v for k, v in kernels[dom].items() if k not in ("counters_mean_per_launch", "source")})
    json.dump(out, open(os.path.join(dst, f"roofline{a.suffix}.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("source", "counters_mean_per_launch")} for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
