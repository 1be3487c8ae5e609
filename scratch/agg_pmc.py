import csv, sys, collections, glob
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:6]:
    n = max(v for (kk, _), v in cnt.items() if kk == k)
    print(k, "dispatches", n)
    wc = d.get("SQ_WAVE_CYCLES", 1)
    for c, v in sorted(d.items()):
        print("   %-28s %14.0f  per-wavecycle %.3f" % (c, v, v / wc))
