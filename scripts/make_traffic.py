#!/usr/bin/env python3
"""profiles/traffic.json from the two rocprofv3 --pmc passes of scripts/gpu_round.sh:

    python scripts/make_traffic.py <tag> gpurun_out/pmc_fetch_<tag> gpurun_out/pmc_write_<tag> [--workload 8192x8192]

Takes the dispatches of the dominant step kernel (largest grid), applies the gfx950 corrections of
MI355X_MICROARCH.md (FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE reports half of a coalesced read
stream) and writes bytes per launch.  Also copies the two counter CSVs to profiles/<tag>/."""
import argparse
import csv
import glob
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("tag")
ap.add_argument("fetch_dir")
ap.add_argument("write_dir")
ap.add_argument("--workload", default="8192x8192")
a = ap.parse_args()
nx, ny = (int(v) for v in a.workload.split("x"))


def collect(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and
            re.search(r"lbm_(multi|step|tile)_kernel", r["Kernel_Name"])]
    count = {}
    for r in rows:
        count[r["Kernel_Name"]] = count.get(r["Kernel_Name"], 0) + 1
    name = max(count, key=count.get)                               # the full-K launches of the timed run
    vals = [float(r["Counter_Value"]) for r in rows if r["Kernel_Name"] == name]
    return f, name, vals


ff, kname, fetch = collect(a.fetch_dir, "FETCH_SIZE")
wf, kname2, write = collect(a.write_dir, "WRITE_SIZE")
assert kname == kname2
m = re.search(r"lbm_multi_kernel<(\d+)", kname)
steps = int(m.group(1)) if m else 1
short = f"lbm_multi_kernel<{steps}>" if m else kname.split("(")[0]
rd = 2.0 * 1024.0 * sum(fetch) / len(fetch)
wr = 1024.0 * sum(write) / len(write)
out = {
    "workload": a.workload, "kernel": short, "steps_per_launch": steps, "round": a.tag,
    "counters": {"FETCH_SIZE": {"launches": len(fetch), "mean_KB": sum(fetch) / len(fetch), "min_KB": min(fetch), "max_KB": max(fetch)},
                 "WRITE_SIZE": {"launches": len(write), "mean_KB": sum(write) / len(write), "min_KB": min(write), "max_KB": max(write)}},
    "correction": "HBM read bytes = 2 x FETCH_SIZE x 1024 (gfx950 reports half of a coalesced read stream, MI355X_MICROARCH.md "
                  "section HBM; check: every source value must be read at least once per launch = 36 B x cells). "
                  "Write bytes = WRITE_SIZE x 1024. Separate --pmc passes.",
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "hbm_bytes_per_cell_step": (rd + wr) / (nx * ny * steps),
    "minimum_read_bytes_per_launch": 36 * nx * ny,
    "algorithmic_bytes_per_launch": 108 * nx * ny * steps,
}
dst = os.path.join(ROOT, "profiles", a.tag)
os.makedirs(dst, exist_ok=True)
shutil.copyfile(ff, os.path.join(dst, f"pmc_fetch_size_{nx}.csv"))
shutil.copyfile(wf, os.path.join(dst, f"pmc_write_size_{nx}.csv"))
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("kernel", "hbm_read_bytes_per_launch", "hbm_write_bytes_per_launch", "hbm_bytes_per_cell_step")}))
