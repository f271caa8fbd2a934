"""Timeline of the LAST pass in a rocprofv3 kernel trace of tools/shard_trace.py: per launch start (relative), duration,
gap to the previous launch on the same queue.  usage: python tools/trace_timeline.py <dir with *_kernel_trace.csv> [launches per pass]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows = [r for r in rows if "k_bounce" in r["Kernel_Name"] or "k_fold" in r["Kernel_Name"] or "k_path" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# passes are separated by idle gaps > 1 ms
groups, cur, last_end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if last_end is not None and s - last_end > 1_000_000:
        groups.append(cur); cur = []
    cur.append(r); last_end = max(last_end or 0, e)
groups.append(cur)
g = groups[-1]
t0 = min(int(r["Start_Timestamp"]) for r in g)
t1 = max(int(r["End_Timestamp"]) for r in g)
print("passes seen:", len(groups), "; last pass: %d launches, span %.1f us" % (len(g), (t1 - t0) / 1e3))
prev = {}
busy = 0
for r in g:
    q = r.get("Queue_Id", "0")
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    short = "fold" if "k_fold" in name else ("q<%s>" % name.split("k_bounce_q")[1][:24] if "k_bounce_q" in name else name[:30])
    gap = (s - prev[q]) / 1e3 if q in prev else 0.0
    print("queue %s  start %8.1f us  dur %7.1f us  gap %6.1f us  %s  grid %s" % (q, (s - t0) / 1e3, (e - s) / 1e3, gap, short, r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
    prev[q] = e
    busy += e - s
print("sum of durations %.1f us over span %.1f us" % (busy / 1e3, (t1 - t0) / 1e3))
