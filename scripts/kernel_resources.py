#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of liblbm_d2q9.so's device code, from
`hipcc -Rpass-analysis=kernel-resource-usage` (compile-only: runs without a GPU).

    python scripts/kernel_resources.py [--out profiles/r02/kernel_resources.txt]

Exit status 1 if any kernel uses scratch (spills or runtime-indexed private arrays)."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mpilattice-boltzmann_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return out.stdout.splitlines() if out.returncode == 0 else names


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, "-c", os.path.join(CSRC, "lbm_kernels.hip"), "-o", os.path.join(tmp, "k.o"),
               "-Rpass-analysis=kernel-resource-usage"]
        r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        return r.returncode
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +(\S[^:]*): (.*?) \[-Rpass", line) or re.search(r"remark: +(\S[^:]*): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        elif cur is not None:
            cur[key] = val
    names = demangle([r_["name"] for r_ in rows])
    lines = [f"{'kernel':70s} {'VGPRs':>5s} {'AGPRs':>5s} {'SGPRs':>5s} {'scratch B/lane':>14s} {'waves/SIMD':>10s} {'LDS B':>7s}"]
    bad = 0
    for row, name in zip(rows, names):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*\)$", "", name).replace("void ", "")
        scratch = int(row.get("ScratchSize [bytes/lane]", "0"))
        bad += scratch > 0
        lines.append(f"{name:70s} {row.get('VGPRs', '?'):>5s} {row.get('AGPRs', '?'):>5s} {row.get('TotalSGPRs', '?'):>5s} "
                     f"{scratch:14d} {row.get('Occupancy [waves/SIMD]', '?'):>10s} {row.get('LDS Size [bytes/block]', '?'):>7s}")
    text = "\n".join(lines) + "\n"
    sys.stdout.write(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as fh:
            fh.write("# hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Rpass-analysis=kernel-resource-usage (dynamic LDS is not included)\n" + text)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
