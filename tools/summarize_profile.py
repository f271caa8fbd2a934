#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into profiles/<tag>_summary.json + .md.

Kernel-trace: per-kernel launch count, total/avg/min/max duration (ns).  PMC: per-kernel mean of
each counter per dispatch.  HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes
(FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 and both counters are in
KiB: /opt/skills/guides/MI355X_MICROARCH.md section HBM); dword-per-lane accesses are outside the
calibrated shapes, so both the raw and the corrected figure are reported.

The header of the .md states what a reader needs to reproduce bench.py's roofline block from this directory alone: the rendered
steps, the algorithmic bytes per step, the UNION of the trace kernels' [start, end] intervals (launches of the two streams overlap,
so the per-kernel sum of durations over-counts) and the fractions that follow -- VALU issue and measured HBM bandwidth first, the
survey-unit figure last."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def short(name):
    import re
    m = re.search(r"(k_bounce_defer|k_bounce_q|k_bounce_seg|k_bounce)<([a-z, ]+)>", name)
    if m:
        flags = [x.strip() == "true" for x in m.group(2).split(",")]
        if m.group(1) in ("k_bounce_defer", "k_bounce_q"):
            return "%s<%s%s>" % (m.group(1), "last" if flags[0] else "mid", ",gen" if flags[1] else "")
        tag = "%s<%s,%s" % (m.group(1), "lds" if flags[0] else "scalar", "last" if flags[1] else "mid")
        if len(flags) > 2:
            tag += ",cull" if flags[2] else ",brute"
        if len(flags) > 3 and flags[3]:
            tag += ",gen"
        return tag + ">"
    for k in ("k_path_q", "k_path_w", "k_fold", "k_generate", "k_flat", "k_display"):
        if k in name:
            return k
    return name[:60]


def union_ns(intervals):
    """total length of the union of [start, end] intervals"""
    tot, cur_a, cur_b = 0, None, None
    for a, b in sorted(intervals):
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        elif b > cur_b:
            cur_b = b
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def bench_line(src):
    """the JSON line bench.py printed under the kernel-trace pass (tools/profile.sh keeps it in kt.log)"""
    path = os.path.join(src, "kt.log")
    if os.path.exists(path):
        for ln in open(path, errors="replace"):
            if ln.startswith("{"):
                try:
                    return json.loads(ln)
                except ValueError:
                    pass
    return None


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = {"tag": tag, "kernels": {}, "pmc": {}}
    render = lambda name: any(k in name for k in ("k_bounce", "k_path", "k_generate"))          # noqa: E731  the trace + scatter + compact launches
    for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        if not any(render(r["Kernel_Name"]) for r in rows):
            continue                                                                          # (the trace of another process of the run)
        agg, spans, streams = defaultdict(list), defaultdict(list), defaultdict(lambda: defaultdict(int))
        for row in rows:
            k, a, b = short(row["Kernel_Name"]), int(row["Start_Timestamp"]), int(row["End_Timestamp"])
            agg[k].append(b - a)
            spans[k].append((a, b))
            streams[k][row.get("Stream_Id", "?")] += b - a
        for k, v in agg.items():
            # total_ms SUMS the launches -- launches of several streams overlap, so it exceeds the time the kernel kept the GPU busy;
            # union_ms is the length of the union of the launches' [start, end] intervals: that time
            out["kernels"][k] = {"launches": len(v), "total_ms": sum(v) / 1e6, "union_ms": union_ns(spans[k]) / 1e6,
                                 "per_stream_ms": {sid: t / 1e6 for sid, t in sorted(streams[k].items())},
                                 "avg_us": sum(v) / len(v) / 1e3, "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
        rk = [iv for k, v in spans.items() if k.startswith(("k_bounce", "k_path", "k_generate")) for iv in v]
        rkf = rk + [iv for iv in spans.get("k_fold", [])]
        out["render_union_ms"] = {"trace_kernels": union_ns(rk) / 1e6, "trace_and_fold_kernels": union_ns(rkf) / 1e6}
        # the launch sequence of one render iteration (generate + bounces), from the tail of the trace
        rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
        seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                int(r["Start_Timestamp"])) for r in rows if render(r["Kernel_Name"])]
        gens = [i for i, x in enumerate(seq) if x[0] == "k_generate"]
        if len(gens) >= 3:
            a, b = gens[-3], gens[-2]
            out["one_iteration"] = [{"kernel": k, "us": round(us, 2), "gap_before_us": round((seq[i][2] - (seq[i - 1][2] + seq[i - 1][1] * 1e3)) / 1e3, 2) if i > 0 else None}
                                    for i, (k, us, _) in enumerate(seq) if a <= i < b]
    # the roofline fraction from THIS profile: SURVEY.md 8(d)'s algorithmic bytes of the rendered steps (bench.py's own figure, from the
    # line it printed under the profiler) over the time the render kernels kept the GPU busy (union of their intervals)
    line = bench_line(src)
    if line and out.get("render_union_ms"):
        steps = int(line["warmup"]) + 2 * int(line["steps"])          # tools/profile.sh: W warm-up steps, one untimed and one timed K-step pass
        roof = line["roofline"]
        ab = float(roof["algorithmic_bytes_per_step"])
        u, uf = out["render_union_ms"]["trace_kernels"], out["render_union_ms"]["trace_and_fold_kernels"]
        out["roofline_from_profile"] = {
            "workload": line["config"]["workload"], "steps_profiled": steps, "algorithmic_bytes_per_step": ab,
            "frac_trace_kernels": ab * steps / (u * 1e-3) / 8e12, "frac_trace_and_fold_kernels": ab * steps / (uf * 1e-3) / 8e12,
            "ms_per_step_trace_kernels": u / steps, "ms_per_step_trace_and_fold_kernels": uf / steps,
            "bench_line_under_the_profiler": {"ms_per_step": line["ms_per_step"], "frac": roof["frac"], "steps": line["steps"],
                                              "kernel_only_frac": (roof.get("kernel_events") or {}).get("kernel_only_frac")},
            "units": "SURVEY.md 8(d) algorithmic bytes / 8 TB/s (a work rate in survey-byte units; the HBM bandwidth really achieved is the PMC figure below)"}
    for d in glob.glob(os.path.join(src, "pmc_*")):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            agg = defaultdict(lambda: defaultdict(list))
            for row in csv.DictReader(open(f)):
                agg[short(row["Kernel_Name"])][row["Counter_Name"]].append(float(row["Counter_Value"]))
            for k, cs in agg.items():
                for c, v in cs.items():
                    out["pmc"].setdefault(k, {})[c] = {"mean_per_dispatch": sum(v) / len(v), "dispatches": len(v)}
    for k, cs in out["pmc"].items():
        if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
            fs, ws = cs["FETCH_SIZE"]["mean_per_dispatch"], cs["WRITE_SIZE"]["mean_per_dispatch"]
            cs["hbm_bytes_per_launch_raw"] = (fs + ws) * 1024
            cs["hbm_bytes_per_launch_corrected"] = (2 * fs + ws) * 1024
    # what bounds the kernels, from the counters of this profile: VALU issue against the 2-cycle peak, HBM bytes against 8 TB/s
    rp = out.get("roofline_from_profile")
    if rp:
        steps, busy_s = rp["steps_profiled"], out["render_union_ms"]["trace_kernels"] * 1e-3
        valu = sum(cs["SQ_INSTS_VALU"]["mean_per_dispatch"] * cs["SQ_INSTS_VALU"]["dispatches"] for k, cs in out["pmc"].items()
                   if k.startswith(("k_bounce", "k_path")) and "SQ_INSTS_VALU" in cs)
        hbm = sum((2 * cs["FETCH_SIZE"]["mean_per_dispatch"] + cs["WRITE_SIZE"]["mean_per_dispatch"]) * 1024 * cs["FETCH_SIZE"]["dispatches"]
                  for k, cs in out["pmc"].items() if k.startswith(("k_bounce", "k_path")) and "FETCH_SIZE" in cs and "WRITE_SIZE" in cs)
        if valu:
            rp["valu_issue"] = {"wave_instructions_per_step": valu / steps, "frac_of_2_cycle_peak": valu / busy_s / 1.2288e12}
        if hbm:
            rp["hbm_measured"] = {"bytes_per_step": hbm / steps, "GB_s": hbm / busy_s / 1e9, "frac_of_peak": hbm / busy_s / 8e12,
                                  "ratio_to_algorithmic_bytes": hbm / steps / rp["algorithmic_bytes_per_step"]}
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
    with open("profiles/%s_summary.md" % tag, "w") as md:
        md.write("# rocprofv3 summary %s\n\n" % tag)
        if rp:
            md.write("Workload: %s.  **%d rendered steps** in this profile (tools/profile.sh: warm-up + one untimed + one timed pass).\n\n" % (rp["workload"], rp["steps_profiled"]))
            if "valu_issue" in rp:
                md.write("* **What bounds it -- VALU issue**: %.1f M wave64 vector instructions per step = **%.3f of the 2-cycle issue peak** (1024 SIMDs x 2.4 GHz / 2) over the time the trace kernels kept the GPU busy.\n"
                         % (rp["valu_issue"]["wave_instructions_per_step"] / 1e6, rp["valu_issue"]["frac_of_2_cycle_peak"]))
            if "hbm_measured" in rp:
                md.write("* **HBM really moved** (PMC, (2 x FETCH_SIZE + WRITE_SIZE) x 1024): %.1f MB per step = %.0f GB/s = **%.3f of the 8 TB/s peak**, %.2f x the algorithmic bytes.\n"
                         % (rp["hbm_measured"]["bytes_per_step"] / 1e6, rp["hbm_measured"]["GB_s"], rp["hbm_measured"]["frac_of_peak"], rp["hbm_measured"]["ratio_to_algorithmic_bytes"]))
            md.write("* Stated roofline, in SURVEY.md 8(d)'s algorithmic-byte units (NOT achieved HBM bandwidth): %.1f MB per step x %d steps / union of the trace kernels' intervals (%.3f ms) / 8 TB/s = **frac %.4f** "
                     "(%.4f with k_fold's intervals in the union: %.3f ms; %.4f / %.4f ms per step).\n"
                     % (rp["algorithmic_bytes_per_step"] / 1e6, rp["steps_profiled"], out["render_union_ms"]["trace_kernels"], rp["frac_trace_kernels"],
                        rp["frac_trace_and_fold_kernels"], out["render_union_ms"]["trace_and_fold_kernels"], rp["ms_per_step_trace_kernels"], rp["ms_per_step_trace_and_fold_kernels"]))
            bl = rp["bench_line_under_the_profiler"]
            md.write("* The bench line printed under the profiler (its %d-step timed pass, wall clock incl. launch gaps): %.4f ms/step, frac %.4f, kernel_only_frac %s.\n\n"
                     % (bl["steps"], bl["ms_per_step"], bl["frac"], bl["kernel_only_frac"]))
        md.write("`total ms` sums the launches (launches of the two streams overlap: it exceeds the busy time); `union ms` is the length of the union of the launches' [start, end] intervals.\n\n")
        md.write("| kernel | launches | avg us | min us | max us | total ms | union ms | per stream ms |\n|---|---|---|---|---|---|---|---|\n")
        for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
            md.write("| %s | %d | %.2f | %.2f | %.2f | %.3f | %.3f | %s |\n" % (k, v["launches"], v["avg_us"], v["min_us"], v["max_us"], v["total_ms"], v["union_ms"],
                                                                              ", ".join("%s: %.3f" % (sid, t) for sid, t in v["per_stream_ms"].items())))
        md.write("\n## PMC (mean per dispatch)\n\n")
        for k, cs in out["pmc"].items():
            md.write("### %s\n\n" % k)
            for c, v in sorted(cs.items()):
                md.write("- %s: %s\n" % (c, ("%.4g" % v["mean_per_dispatch"]) if isinstance(v, dict) else ("%.4g" % v)))
            md.write("\n")
    print(open("profiles/%s_summary.md" % tag).read())


if __name__ == "__main__":
    main()
