"""pmc_mean.py DIR SUBSTR: per-launch mean of every counter of the kernels whose name contains SUBSTR,
from the counter_collection csv files of rocprofv3 --pmc runs under DIR."""
import csv, sys, glob, collections
agg, cnt = collections.defaultdict(float), collections.defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(agg):
    print("  %-30s %14.0f   (%d launches)" % (k, agg[k] / cnt[k], cnt[k]))
