"""Sum PMC counters over the k_bounce_* / k_fold dispatches of the LAST n dispatches-per-pass in a rocprofv3 --pmc csv.
usage: python tools/pmc_sum.py <dir> <dispatches per pass>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
per = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(f)) if "k_bounce" in r["Kernel_Name"] or "k_fold" in r["Kernel_Name"]]
ids = sorted({int(r["Dispatch_Id"]) for r in rows})
last = set(ids[-per:])
tot = collections.Counter()
for r in rows:
    if int(r["Dispatch_Id"]) in last:
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
print({k: round(v) for k, v in tot.items()})
