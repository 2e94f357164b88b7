#!/usr/bin/env python3
"""Summarise separate rocprofv3 --pmc passes (never combined with other trace domains) over tools/probe_counters.py
into per-kernel HBM bytes per launch and MFMA utilisation.

    python tools/pmc_summary.py --key WORKLOAD_KEY --fetch F_counter_collection.csv --write W_counter_collection.csv \
        [--sq SQ_counter_collection.csv --sq-trace SQ_kernel_trace.csv] > profiles/rNN_pmc_traffic.json

* HBM bytes: FETCH_SIZE / WRITE_SIZE, calibrated IN THE SAME PASS on the probe's torch.sign over a 1 GiB tensor
  (2^30 bytes read, 2^30 written): that absorbs the counter unit and the gfx950 wide-load correction
  (/opt/skills/guides/MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports 1/2 of the bytes of 16 B/lane streaming
  reads).  Kernels with another access width inherit the streaming calibration: treat absolutes as estimates.
* MFMA utilisation: SQ_VALU_MFMA_BUSY_CYCLES (cycles the matrix pipes were busy, summed over the chip's 1024 SIMDs)
  / (1024 x GRBM_GUI_ACTIVE cycles of the dispatch).  GRBM_GUI_ACTIVE is reported per XCD or summed over the 8 XCDs
  depending on the tool build; the ratio to the dispatch's wall time (kernel trace) tells which, and is recorded.
* kernel_src_sha: sha256 over csrc/*.hip, *.h at collection time; bench.py quotes a summary only if it matches."""
import argparse
import csv
import hashlib
import json
import os
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "revisit-anything_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    n = name.replace("void ", "", 1).strip()
    if n.startswith("(anonymous namespace)::"):   # kernels of an unnamed namespace: keep the identifier, not the empty prefix
        n = n[len("(anonymous namespace)::"):]
    return n.split("(")[0].strip()


WINDOW = False   # --window: keep only the dispatches BETWEEN the first and the last torch.sign marker (bench.py --pmc-calibrate)


def _rows(path):
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    key = next((c for c in ("Dispatch_Id", "Start_Timestamp") if rows and c in rows[0]), None)
    if key is None:   # (no order column: keep the file's order)
        key = "_order"
        for i, r in enumerate(rows):
            r[key] = i
    rows.sort(key=lambda r: int(r[key]))
    return rows, key


def _window(rows, key):
    """(rows inside the marker window, the marker rows).  The markers themselves calibrate the byte counters."""
    marks = sorted({int(r[key]) for r in rows if "sign_kernel" in r["Kernel_Name"]})
    if not WINDOW:
        return rows, [r for r in rows if "sign_kernel" in r["Kernel_Name"]]
    if len(marks) < 2:
        raise SystemExit("--window: need the two torch.sign marker dispatches of `bench.py --pmc-calibrate` in the pass")
    lo, hi = marks[0], marks[-1]
    return [r for r in rows if lo < int(r[key]) < hi], [r for r in rows if int(r[key]) in (lo, hi)]


def load_counters(path):
    """{kernel: {counter: [values per dispatch]}} (of the marker window with --window; the markers are always included)"""
    out = defaultdict(lambda: defaultdict(list))
    rows, key = _rows(path)
    inside, marks = _window(rows, key)
    for r in inside + ([] if not WINDOW else marks):
        out[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out


def load_durations(path):
    out = defaultdict(list)
    rows, key = _rows(path)
    for r in _window(rows, key)[0]:
        out[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


def launch_multiset(path):
    """{kernel: dispatches inside the marker window} of a kernel-trace or counter CSV (runtime copy / fill kernels left out)"""
    rows, key = _rows(path)
    out = defaultdict(int)
    seen = set()
    for r in _window(rows, key)[0]:
        if (r[key], r["Kernel_Name"]) in seen:      # (a counter CSV holds one row per counter and dispatch)
            continue
        seen.add((r[key], r["Kernel_Name"]))
        name = short(r["Kernel_Name"])
        if "__amd_rocclr" in name or "at::native" in name or "sign_kernel" in name:
            continue
        out[name] += 1
    return dict(out)


def avg(v):
    return sum(v) / len(v) if v else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--key", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--sq", default=None)
    ap.add_argument("--sq-trace", default=None)
    ap.add_argument("--lds", default=None, help="counter_collection.csv of the LDS pass (SQ_LDS_* / SQ_INSTS_LDS / SQ_ACTIVE_INST_LDS ...)")
    ap.add_argument("--window", action="store_true", help="the passes ran over `bench.py --pmc-calibrate`: summarise only the dispatches "
                    "between its two torch.sign markers = the TIMED steps (round 6)")
    ap.add_argument("--steps", type=int, default=1, help="--window: timed steps inside the window (launches_per_step = launches / steps)")
    ap.add_argument("--timeline-trace", default=None, help="--window: kernel_trace.csv of the same command WITHOUT counters (the step "
                    "timeline committed beside the summary); the per-kernel launch multiset of the two windows must be equal")
    a = ap.parse_args()
    global WINDOW
    WINDOW = a.window
    F, Wc = load_counters(a.fetch), load_counters(a.write)
    fcal = next((avg(v["FETCH_SIZE"]) for k, v in F.items() if "sign_kernel" in k and v.get("FETCH_SIZE")), None)
    wcal = next((avg(v["WRITE_SIZE"]) for k, v in Wc.items() if "sign_kernel" in k and v.get("WRITE_SIZE")), None)
    if not fcal or not wcal:
        raise SystemExit("calibration dispatch (torch.sign over 1 GiB) not found in the PMC output")
    fscale, wscale = float(1 << 30) / fcal, float(1 << 30) / wcal
    SQ = load_counters(a.sq) if a.sq else {}
    DUR = load_durations(a.sq_trace) if a.sq_trace else {}
    LDS = load_counters(a.lds) if a.lds else {}
    kernels = []
    for name, c in F.items():
        if "elementwise" in name or "sign_kernel" in name or not c.get("FETCH_SIZE"):
            continue
        n = len(c["FETCH_SIZE"])
        rd = avg(c["FETCH_SIZE"]) * fscale
        wv = Wc.get(name, {}).get("WRITE_SIZE")
        wr = avg(wv) * wscale if wv else 0.0
        k = {"name": name, "launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
             "hbm_bytes_per_launch": rd + wr}
        if a.window:
            k["launches_per_step"] = n / a.steps
            k["hbm_bytes_per_step"] = (rd + wr) * n / a.steps
        s = SQ.get(name)
        if s and s.get("SQ_VALU_MFMA_BUSY_CYCLES") and s.get("GRBM_GUI_ACTIVE"):
            busy, gui = avg(s["SQ_VALU_MFMA_BUSY_CYCLES"]), avg(s["GRBM_GUI_ACTIVE"])
            dur = avg(DUR.get(name, []))
            xcd = 1
            if dur and gui / dur > 4.0:       # > 4 "GHz": the counter is summed over the 8 XCDs
                xcd = 8
            k["mfma_busy_cycles"] = busy
            k["gui_active_cycles"] = gui / xcd
            k["gui_active_summed_over_xcds"] = xcd == 8
            k["mfma_util"] = busy / (1024.0 * gui / xcd) if gui else None
            if dur:
                k["avg_duration_ms_under_pmc"] = dur / 1e6
                k["effective_clock_ghz"] = gui / xcd / dur
            for extra in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY",
                          "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_INSTS_VALU_MFMA_MOPS_F16"):
                if s.get(extra):
                    k[extra.lower()] = avg(s[extra])
        ld = LDS.get(name)
        if ld:   # LDS pass: raw per-launch averages (summed over all SIMDs / XCDs) + two ratios that do not depend on units
            k["lds_pass"] = {c.lower(): avg(v) for c, v in ld.items() if v}
            act, conf = ld.get("SQ_LDS_IDX_ACTIVE"), ld.get("SQ_LDS_BANK_CONFLICT")
            if act and conf and avg(act):
                k["lds_bank_conflict_frac"] = avg(conf) / avg(act)      # conflict cycles / cycles the LDS index unit was active
            busy, lact = ld.get("SQ_BUSY_CYCLES"), ld.get("SQ_ACTIVE_INST_LDS")
            if busy and lact and avg(busy):
                k["lds_inst_active_per_busy_cycle"] = avg(lact) / avg(busy)
        kernels.append(k)
    kernels.sort(key=lambda k: -k["hbm_bytes_per_launch"] * k["launches"])
    extra = {}
    if a.window:
        ms_pmc = launch_multiset(a.fetch)
        extra["timed_steps_in_window"] = a.steps
        extra["launch_multiset_per_window"] = dict(sorted(ms_pmc.items()))
        extra["describe_stage_hbm_bytes_per_step"] = sum(
            k["hbm_bytes_per_step"] for k in kernels
            if any(t in k["name"] for t in ("incidence", "adjacency", "assign", "prep_kernel", "group_plan", "gram_norms", "token_norms",
                                            "gemm_f16x3", "project_aggregate", "normalize_rows", "aggregate_kernel")))
        if a.timeline_trace:
            ms_tl = launch_multiset(a.timeline_trace)
            extra["launch_multiset_equals_timeline"] = ms_tl == ms_pmc
            if ms_tl != ms_pmc:
                extra["launch_multiset_differences"] = {k_: [ms_pmc.get(k_, 0), ms_tl.get(k_, 0)] for k_ in sorted(set(ms_pmc) | set(ms_tl))
                                                        if ms_pmc.get(k_, 0) != ms_tl.get(k_, 0)}
    json.dump({"workload_key": a.key, "kernel_src_sha": kernel_src_sha(),
               "provenance": ("tools/gpu_round_artifacts.sh pmc: rocprofv3 --pmc passes over `bench.py --pmc-calibrate` itself "
                              "(--kernel-include-regex = the library's kernels + the marker); the dispatches of the TIMED steps, cut out "
                              "between the two torch.sign markers") if a.window else
                             "tools/gpu_round_artifacts.sh pmc: rocprofv3 --pmc passes over tools/probe_counters.py",
               **extra,
               "calibration": {"fetch_raw_per_GiB": fcal, "write_raw_per_GiB": wcal, "bytes_per_fetch_unit": fscale,
                               "bytes_per_write_unit": wscale},
               "kernels": kernels[:40]}, __import__("sys").stdout, indent=1)
    if extra.get("launch_multiset_equals_timeline") is False:
        print("pmc_summary: the counter passes' launch multiset differs from the timeline's: " + json.dumps(extra["launch_multiset_differences"]),
              file=__import__("sys").stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
