"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel name: python tools/pmc_summary.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        cnt[k][row["Counter_Name"]] += 1
for k in sorted(agg, key=lambda k: -sum(agg[k].values()))[:8]:
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} total {v:.4g}  per-dispatch {v / cnt[k][c]:.4g}  (n={cnt[k][c]})")
