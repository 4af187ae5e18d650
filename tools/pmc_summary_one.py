"""Dev: average per-launch counter values of kernels matching a substring in a rocprofv3 --pmc output directory."""
import csv, glob, collections, sys
d, pat = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:32s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
