#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; never combined with
tracing of other domains) of `bench.py --pmc-calibrate` into HBM bytes per launch per kernel.

    python tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv WORKLOAD_KEY > profiles/rNN_pmc_traffic.json

Calibration: bench.py --pmc-calibrate ends with one torch.sign over a 1 GiB tensor (the only "sign_kernel" dispatch);
its counter values are mapped to exactly 2^30 bytes read / written, which
absorbs the counter's unit (KiB) and the gfx950 wide-load correction (/opt/skills/guides/MI355X_MICROARCH.md, HBM section:
FETCH_SIZE reports 1/2 of the bytes of 16 B/lane streaming reads).  Kernels with a different access width inherit the
streaming calibration -- treat their absolute numbers as estimates, ratios between variants are unaffected."""
import csv
import json
import sys
from collections import defaultdict


def load(path, counter):
    rows = []
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            rows.append((r["Kernel_Name"], float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0)))
    return rows


def summarise(rows):
    cal = None
    for name, v, grid in rows:
        if "sign_kernel" in name:
            cal = v   # bench.py --pmc-calibrate: torch.sign over 2^28 floats
    agg = defaultdict(lambda: [0, 0.0])
    for name, v, _ in rows:
        a = agg[name]
        a[0] += 1
        a[1] += v
    return cal, agg


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def main():
    fpath, wpath, key = sys.argv[1], sys.argv[2], sys.argv[3]
    fcal, fagg = summarise(load(fpath, "FETCH_SIZE"))
    wcal, wagg = summarise(load(wpath, "WRITE_SIZE"))
    if not fcal or not wcal:
        raise SystemExit("calibration copy not found in the PMC output")
    fscale, wscale = float(1 << 30) / fcal, float(1 << 30) / wcal
    kernels = []
    for name, (n, tot) in fagg.items():
        wn, wtot = wagg.get(name, (0, 0.0))
        rd = tot / n * fscale
        wr = (wtot / wn * wscale) if wn else 0.0
        kernels.append({"name": short(name), "launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                        "hbm_bytes_per_launch": rd + wr})
    kernels.sort(key=lambda k: -k["hbm_bytes_per_launch"] * k["launches"])
    json.dump({"workload_key": key, "calibration": {"fetch_raw_per_GiB": fcal, "write_raw_per_GiB": wcal,
                                                    "bytes_per_fetch_unit": fscale, "bytes_per_write_unit": wscale},
               "kernels": [k for k in kernels if "elementwise" not in k["name"]][:40]}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
