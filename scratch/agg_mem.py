"""Per-kernel memory-side counters from rocprofv3 --pmc passes: sums and per-dispatch means."""
import csv, sys, collections, glob, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(conv_kernel|wgrad_kernel)<[^>]*>", r["Kernel_Name"])
            k = m.group(0) if m else r["Kernel_Name"].split("(")[0][-50:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k][r["Counter_Name"]] += 1
for k, dd in sorted(agg.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0))[:8]:
    parts = []
    for c, v in sorted(dd.items()):
        n = cnt[k][c]
        parts.append(f"{c}={v:.4g} (n={n}, mean {v / n:.4g})")
    print(k, "|", "; ".join(parts))
