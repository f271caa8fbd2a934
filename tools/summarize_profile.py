#!/usr/bin/env python3
"""Summarise a tools/profile.sh output directory into profiles/<tag>_summary.json + .md.

Kernel-trace: per-kernel launch count, total/avg/min/max duration (ns).  PMC: per-kernel mean of
each counter per dispatch.  HBM traffic per launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 bytes
(FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950 and both counters are in
KiB: /opt/skills/guides/MI355X_MICROARCH.md section HBM); dword-per-lane accesses are outside the
calibrated shapes, so both the raw and the corrected figure are reported."""
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


def main():
    src, tag = sys.argv[1], sys.argv[2]
    out = {"tag": tag, "kernels": {}, "pmc": {}}
    for f in glob.glob(os.path.join(src, "kt", "**", "*kernel_trace.csv"), recursive=True):
        agg = defaultdict(list)
        for row in csv.DictReader(open(f)):
            agg[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for k, v in agg.items():
            out["kernels"][k] = {"launches": len(v), "total_ms": sum(v) / 1e6, "avg_us": sum(v) / len(v) / 1e3,
                                 "min_us": min(v) / 1e3, "max_us": max(v) / 1e3}
        # the launch sequence of one render iteration (generate + bounces), from the tail of the trace
        rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
        seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                int(r["Start_Timestamp"])) for r in rows if "k_bounce" in r["Kernel_Name"] or "k_path" in r["Kernel_Name"] or "k_generate" in r["Kernel_Name"]]
        gens = [i for i, x in enumerate(seq) if x[0] == "k_generate"]
        if len(gens) >= 3:
            a, b = gens[-3], gens[-2]
            out["one_iteration"] = [{"kernel": k, "us": round(us, 2), "gap_before_us": round((seq[i][2] - (seq[i - 1][2] + seq[i - 1][1] * 1e3)) / 1e3, 2) if i > 0 else None}
                                    for i, (k, us, _) in enumerate(seq) if a <= i < b]
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
    os.makedirs("profiles", exist_ok=True)
    json.dump(out, open("profiles/%s_summary.json" % tag, "w"), indent=1)
    with open("profiles/%s_summary.md" % tag, "w") as md:
        md.write("# rocprofv3 summary %s\n\n| kernel | launches | avg us | min us | max us | total ms |\n|---|---|---|---|---|---|\n" % tag)
        for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["total_ms"]):
            md.write("| %s | %d | %.2f | %.2f | %.2f | %.3f |\n" % (k, v["launches"], v["avg_us"], v["min_us"], v["max_us"], v["total_ms"]))
        md.write("\n## PMC (mean per dispatch)\n\n")
        for k, cs in out["pmc"].items():
            md.write("### %s\n\n" % k)
            for c, v in sorted(cs.items()):
                md.write("- %s: %s\n" % (c, ("%.4g" % v["mean_per_dispatch"]) if isinstance(v, dict) else ("%.4g" % v)))
            md.write("\n")
    print(open("profiles/%s_summary.md" % tag).read())


if __name__ == "__main__":
    main()
