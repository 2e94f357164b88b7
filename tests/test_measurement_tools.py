"""The measurement plumbing (no GPU): tools/pmc_summary.py turns rocprofv3 counter CSVs into per-kernel HBM bytes / MFMA
utilisation with its in-pass calibration, and bench.py quotes such a summary only when it was collected from the
current kernel sources."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(path, rows):
    with open(path, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)


def test_pmc_summary_calibrates_and_derives_ratios(tmp_path):
    GiB = 1 << 30
    cal = "void at::native::vectorized_elementwise_kernel<4, at::native::sign_kernel_cuda(...)>"
    kern = "void knn_f16_filter_kernel<256, 256, 4, 2, 64, 3, 0, true, 0, 2>(unsigned short const*)"
    # FETCH_SIZE in "units" of 64 B that under-report wide loads by 2x: 1 GiB read shows as 2^30 / 128 units
    fetch = [{"Kernel_Name": cal, "Counter_Name": "FETCH_SIZE", "Counter_Value": GiB / 128},
             {"Kernel_Name": kern, "Counter_Name": "FETCH_SIZE", "Counter_Value": 3 * GiB / 128},
             {"Kernel_Name": kern, "Counter_Name": "FETCH_SIZE", "Counter_Value": 5 * GiB / 128}]
    write = [{"Kernel_Name": cal, "Counter_Name": "WRITE_SIZE", "Counter_Value": GiB / 64},
             {"Kernel_Name": kern, "Counter_Name": "WRITE_SIZE", "Counter_Value": GiB / 128}]
    sq = [{"Kernel_Name": kern, "Counter_Name": "SQ_VALU_MFMA_BUSY_CYCLES", "Counter_Value": 1024 * 600.0},
          {"Kernel_Name": kern, "Counter_Name": "GRBM_GUI_ACTIVE", "Counter_Value": 8 * 1000.0}]          # summed over 8 XCDs
    trace = [{"Kernel_Name": kern, "Start_Timestamp": 0, "End_Timestamp": 500}]                             # 500 ns: 8000/500 > 4 "GHz"
    lds = [{"Kernel_Name": kern, "Counter_Name": "SQ_LDS_BANK_CONFLICT", "Counter_Value": 5.0},
           {"Kernel_Name": kern, "Counter_Name": "SQ_LDS_IDX_ACTIVE", "Counter_Value": 1000.0},
           {"Kernel_Name": kern, "Counter_Name": "SQ_BUSY_CYCLES", "Counter_Value": 400.0},
           {"Kernel_Name": kern, "Counter_Name": "SQ_ACTIVE_INST_LDS", "Counter_Value": 100.0}]
    for name, rows in (("f", fetch), ("w", write), ("s", sq), ("t", trace), ("l", lds)):
        _write(tmp_path / f"{name}.csv", rows)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--key", "k1", "--fetch", str(tmp_path / "f.csv"),
                          "--write", str(tmp_path / "w.csv"), "--sq", str(tmp_path / "s.csv"), "--sq-trace", str(tmp_path / "t.csv"),
                          "--lds", str(tmp_path / "l.csv")], capture_output=True, text=True, check=True).stdout
    j = json.loads(out)
    assert j["workload_key"] == "k1" and len(j["kernel_src_sha"]) == 16
    (k,) = j["kernels"]
    assert k["name"].startswith("knn_f16_filter_kernel") and k["launches"] == 2
    assert abs(k["read_bytes_per_launch"] - 4 * GiB) < 1 and abs(k["write_bytes_per_launch"] - 0.5 * GiB) < 1   # calibrated averages
    assert k["gui_active_summed_over_xcds"] is True and abs(k["mfma_util"] - 0.6) < 1e-12
    assert abs(k["lds_bank_conflict_frac"] - 0.005) < 1e-12 and abs(k["lds_inst_active_per_busy_cycle"] - 0.25) < 1e-12


def test_bench_quotes_a_pmc_summary_only_for_the_current_kernel_sources(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench

    sha = bench.kernel_src_sha()                                  # of the real sources
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))             # profiles/ is looked up under ROOT
    monkeypatch.setattr(bench, "kernel_src_sha", lambda: sha)
    os.makedirs(tmp_path / "profiles")
    rec = {"workload_key": "wl", "kernel_src_sha": "0" * 16,
           "kernels": [{"name": "knn_f16_filter_kernel<256>", "hbm_bytes_per_launch": 7.0, "mfma_util": 0.5},
                       {"name": "_Z18token_norms_kernelILb0EEvPKf", "hbm_bytes_per_launch": 3.0, "mfma_util": 0.3}]}
    (tmp_path / "profiles" / "r09_pmc_traffic.json").write_text(json.dumps(rec))
    t, u, why = bench.pmc_counters("knn_f16_filter_kernel", "wl")
    assert t is None and u is None and "not quoted" in why                      # stale sources: refused
    rec["kernel_src_sha"] = sha
    (tmp_path / "profiles" / "r09_pmc_traffic.json").write_text(json.dumps(rec))
    assert bench.pmc_counters("knn_f16_filter_kernel", "wl")[:2] == (7.0, 0.5)
    assert bench.pmc_counters("token_norms_kernel", "wl")[:2] == (3.0, 0.3)      # mangled names are matched too
    assert bench.pmc_counters("knn_f16_filter_kernel", "other")[0] is None        # another workload: not this file


def test_pmc_summary_window_cuts_the_timed_steps_out_between_the_markers(tmp_path):
    """Round 6: the counter passes run over bench.py itself; `--window` keeps the dispatches between the two torch.sign markers
    (the timed steps), reports launches per step, and compares the launch multiset with the counter-free timeline's."""
    GiB = 1 << 30
    cal = "void at::native::vectorized_elementwise_kernel<4, at::native::sign_kernel_cuda(...)>"
    filt = "void knn_f16_filter_kernel<256, 256, 4, 2, 64, 3, 0, true, 0, 2>(unsigned short const*)"
    asg = "void assign_wide_kernel<2>(float const*)"
    cp = "__amd_rocclr_copyBuffer"

    def disp(i, name, ctr, val):
        return {"Dispatch_Id": i, "Kernel_Name": name, "Counter_Name": ctr, "Counter_Value": val, "Start_Timestamp": 10 * i, "End_Timestamp": 10 * i + 5}

    # DB build (outside the window): 3 filter launches that must NOT be averaged in; window: 2 steps x (1 assign + 3 filters)
    order = [asg, filt, filt, filt, cal, asg, filt, filt, filt, cp, asg, filt, filt, filt, cal, filt]
    fetch = [disp(i, n, "FETCH_SIZE", (GiB / 64 if n == cal else (8 * GiB / 64 if i < 4 or i == 15 else 2 * GiB / 64))) for i, n in enumerate(order) if n != cp]
    write = [disp(i, n, "WRITE_SIZE", (GiB / 64 if n == cal else GiB / 128)) for i, n in enumerate(order) if n != cp]
    trace = [{"Dispatch_Id": i, "Kernel_Name": n, "Start_Timestamp": 10 * i, "End_Timestamp": 10 * i + 5} for i, n in enumerate(order)]
    for name, rows in (("f", fetch), ("w", write), ("t", trace)):
        _write(tmp_path / f"{name}.csv", rows)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), "--window", "--steps", "2", "--key", "k2", "--fetch", str(tmp_path / "f.csv"),
           "--write", str(tmp_path / "w.csv"), "--timeline-trace", str(tmp_path / "t.csv")]
    j = json.loads(subprocess.run(cmd, capture_output=True, text=True, check=True).stdout)
    assert j["launch_multiset_equals_timeline"] is True and j["timed_steps_in_window"] == 2
    assert j["launch_multiset_per_window"] == {"assign_wide_kernel<2>": 2, "knn_f16_filter_kernel<256, 256, 4, 2, 64, 3, 0, true, 0, 2>": 6}
    k = {x["name"].split("<")[0]: x for x in j["kernels"]}
    assert k["knn_f16_filter_kernel"]["launches"] == 6 and k["knn_f16_filter_kernel"]["launches_per_step"] == 3
    assert abs(k["knn_f16_filter_kernel"]["read_bytes_per_launch"] - 2 * GiB) < 1        # the 8-GiB launches outside the window are not in it
    assert "timed steps" in j["provenance"].lower() or "TIMED" in j["provenance"]
    # a timeline with another launch count fails the tool (exit 3) after the summary is written
    trace[9]["Kernel_Name"] = asg        # (the runtime copy inside the window becomes a third assignment launch)
    _write(tmp_path / "t.csv", trace)
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 3 and json.loads(r.stdout)["launch_multiset_equals_timeline"] is False
