#!/usr/bin/env python3
"""Is a launch held back by the power limit?  Polls the GPU's hwmon / sysfs files (socket power, shader clock, power cap)
every few ms while a CHILD process runs the workload (this process never touches the GPU), and prints the statistics of
the samples taken while the shader clock is up.

    python scripts/power_trace.py [--label X] -- python scripts/ab_libs.py --grid 8192x8192 --steps 200 --rounds 10 lib.so
"""
import argparse
import glob
import os
import statistics
import subprocess
import sys
import threading
import time


def read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def find_cards():
    """Every amdgpu card with hwmon files (a box shows all the host's GPUs and their partitions; the one the child runs on
    is picked afterwards: the one whose shader clock moved most)."""
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if read(os.path.join(dev, "vendor")) == "0x1002" and glob.glob(os.path.join(dev, "hwmon/hwmon*")):
            real = os.path.realpath(dev)
            if real not in [r for r, _ in out]:
                out.append((real, dev))
    return [d for _, d in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--label", default="")
    ap.add_argument("--period-ms", type=float, default=5.0)
    ap.add_argument("cmd", nargs=argparse.REMAINDER)
    a = ap.parse_args()
    cmd = a.cmd[1:] if a.cmd and a.cmd[0] == "--" else a.cmd
    cards = find_cards()
    if not cards:
        raise SystemExit("no amdgpu card with hwmon files under /sys/class/drm")
    per_card = {}
    for dev in cards:
        hw = glob.glob(os.path.join(dev, "hwmon/hwmon*"))[0]
        files = {
            "power_uW": [os.path.join(hw, "power1_input"), os.path.join(hw, "power1_average")],
            "cap_uW": [os.path.join(hw, "power1_cap")],
            "sclk_Hz": [os.path.join(hw, "freq1_input")],
            "temp_mC": [os.path.join(hw, "temp2_input"), os.path.join(hw, "temp1_input")],
        }
        per_card[dev] = {k: next((p for p in v if read(p) is not None), None) for k, v in files.items()}
    print(f"[power_trace] polling {len(cards)} card(s)", flush=True)
    all_samples, stop = {dev: [] for dev in cards}, threading.Event()

    def poll():
        while not stop.is_set():
            for dev, avail in per_card.items():
                row = {}
                for k, p in avail.items():
                    v = read(p) if p else None
                    row[k] = float(v) if v not in (None, "") else None
                row["t"] = time.time()
                all_samples[dev].append(row)
            time.sleep(a.period_ms / 1e3)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    time.sleep(0.3)
    rc = subprocess.call(cmd)
    stop.set()
    th.join()

    def swing(dev):
        v = [r["sclk_Hz"] for r in all_samples[dev] if r.get("sclk_Hz") is not None]
        return (max(v) - min(v)) if v else 0.0

    dev = max(cards, key=swing)
    samples = all_samples[dev]
    top = max((r["sclk_Hz"] or 0.0) for r in samples)
    print(f"[power_trace] card with the largest clock swing: {dev} ({os.path.basename(os.path.realpath(dev))})")
    idle = samples[:20]
    busy = [r for r in samples if (r["sclk_Hz"] or 0.0) >= 0.5 * top] or samples

    def stat(rows, key, scale):
        v = [r[key] * scale for r in rows if r.get(key) is not None]
        if not v:
            return "n/a"
        v.sort()
        return f"min {v[0]:.0f} med {statistics.median(v):.0f} p90 {v[int(0.9 * (len(v) - 1))]:.0f} max {v[-1]:.0f}"

    print(f"[power_trace] {a.label} samples {len(samples)} (shader clock >= half its maximum: {len(busy)})")
    print(f"  idle   power W: {stat(idle, 'power_uW', 1e-6)}   sclk MHz: {stat(idle, 'sclk_Hz', 1e-6)}")
    print(f"  loaded power W: {stat(busy, 'power_uW', 1e-6)}   cap W: {stat(busy, 'cap_uW', 1e-6)}")
    print(f"  loaded sclk MHz: {stat(busy, 'sclk_Hz', 1e-6)}   temp C: {stat(busy, 'temp_mC', 1e-3)}")
    sys.exit(rc)


if __name__ == "__main__":
    main()
