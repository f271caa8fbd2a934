"""The profile summariser behind profiles/*_summary.md (tools/summarize_profile.py), on a synthetic rocprofv3 output directory: the
UNION of overlapping launches, per-stream sums, and the roofline fraction that follows from the rendered steps and the
algorithmic bytes of the bench line printed under the profiler -- the arithmetic a reader of profiles/ relies on."""
import csv
import json
import os
import subprocess
import sys

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_union_of_intervals():
    from summarize_profile import union_ns
    assert union_ns([]) == 0
    assert union_ns([(0, 10), (5, 20), (30, 40)]) == 30                  # overlapping launches of two streams count once
    assert union_ns([(5, 20), (0, 10), (10, 12), (40, 41)]) == 21        # any order; nested and touching intervals
    assert union_ns([(0, 100), (10, 20), (30, 40)]) == 100


def test_summary_states_union_steps_and_fraction(tmp_path):
    src = tmp_path / "prof_x"
    (src / "kt" / "run").mkdir(parents=True)
    cols = ["Kind", "Agent_Id", "Queue_Id", "Stream_Id", "Thread_Id", "Dispatch_Id", "Kernel_Id", "Kernel_Name", "Correlation_Id",
            "Start_Timestamp", "End_Timestamp"]
    name = "void ptk::k_path_q<false, false, 144>(ptk::SegArgs, ptk::PathArgs)"
    rows = [("2", name, 1_000_000, 5_000_000), ("3", name, 3_000_000, 9_000_000),      # two streams, overlapping: union 8 ms, sum 10 ms
            ("2", "ptk::k_fold(ptk::FoldArgs)", 9_000_000, 9_500_000),
            ("0", "__amd_rocclr_copyBuffer", 0, 100)]
    with open(src / "kt" / "run" / "1_kernel_trace.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(cols)
        for k, (sid, kn, a, b) in enumerate(rows):
            w.writerow(["KERNEL_DISPATCH", "Agent 2", 1, sid, 1, k, 1, kn, k, a, b])
    line = {"steps": 10, "warmup": 4, "ms_per_step": 0.4, "config": {"workload": "synthetic"},
            "roofline": {"algorithmic_bytes_per_step": 8.0e8, "frac": 0.25, "kernel_events": None}}
    (src / "kt.log").write_text("noise\n" + json.dumps(line) + "\n")
    (src / "pmc_SQ_WAVES" / "run").mkdir(parents=True)
    with open(src / "pmc_SQ_WAVES" / "run" / "2_counter_collection.csv", "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_ALL)
        w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value"])
        for v in (6.0e8, 6.0e8):
            w.writerow([name, "SQ_INSTS_VALU", v])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "summarize_profile.py"), str(src), "x"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.load(open(tmp_path / "profiles" / "x_summary.json"))
    k = out["kernels"]["k_path_q"]
    assert k["launches"] == 2 and abs(k["total_ms"] - 10.0) < 1e-9 and abs(k["union_ms"] - 8.0) < 1e-9
    assert k["per_stream_ms"] == {"2": 4.0, "3": 6.0}
    assert abs(out["render_union_ms"]["trace_kernels"] - 8.0) < 1e-9 and abs(out["render_union_ms"]["trace_and_fold_kernels"] - 8.5) < 1e-9
    rp = out["roofline_from_profile"]
    assert rp["steps_profiled"] == 4 + 2 * 10
    assert abs(rp["frac_trace_kernels"] - 8.0e8 * 24 / 8.0e-3 / 8e12) < 1e-9          # 0.3
    assert abs(rp["valu_issue"]["frac_of_2_cycle_peak"] - 1.2e9 / 8.0e-3 / 1.2288e12) < 1e-9
    md = open(tmp_path / "profiles" / "x_summary.md").read()
    assert "24 rendered steps" in md and "union ms" in md and "frac 0.3000" in md
