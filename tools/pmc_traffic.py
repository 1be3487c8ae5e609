"""profiles/<name>_pmc_traffic.json from rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, TCC hit/miss):
per-launch means per kernel; FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced
reads), both counters are in KB."""
import csv, sys, collections, glob, re, json
out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            m = re.search(r"(conv_kernel|wgrad_kernel|wgrad_walk_kernel)<([^>]*)>", name)
            if m:
                args = [a.strip() for a in m.group(2).split(",")]
                # this rocprofv3 prints the bf16 instantiations as "<bool _Accum, int, E, ...>": the
                # element type and the first integer argument (the stride, 1 in this workload) are lost
                if args[:3] == ["bool _Accum", "int", "E"]:
                    rest = args[3:]
                    if m.group(1) == "conv_kernel":
                        rest = rest[:4]                      # NB, RW, LTW, KS (workgroup shape arguments dropped)
                    args = ["bf16", "1"] + rest
                k = m.group(1) + "<" + ",".join(args) + ">"
            else:
                m2 = re.search(r"(conv_kernel|wgrad_kernel|wgrad_walk_kernel)I(DF16b|f)((?:Li\d+E)+)", name)
                if m2:
                    nums = re.findall(r"Li(\d+)E", m2.group(3))
                    k = m2.group(1) + "<" + ("bf16" if m2.group(2) == "DF16b" else "float") + "," + ",".join(nums) + ">"
                else:
                    m4 = re.search(r"rdb_kernel<(true|false)", name)
                    m3 = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", name)
                    # the fused dense-block kernel under bench.py's names for its two directions
                    k = ("rdb_bwd_kernel" if m4.group(1) == "true" else "rdb_fwd_kernel") if m4 else \
                        (m3.group(1) if m3 else name[:60])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
res = {}
for k, dd in agg.items():
    e = {}
    if "FETCH_SIZE" in dd: e["fetch_bytes"] = round(2 * 1000 * dd["FETCH_SIZE"] / cnt[k]["FETCH_SIZE"])
    if "WRITE_SIZE" in dd: e["write_bytes"] = round(1000 * dd["WRITE_SIZE"] / cnt[k]["WRITE_SIZE"])
    if "TCC_HIT_sum" in dd and "TCC_MISS_sum" in dd:
        e["l2_hit_rate"] = round(dd["TCC_HIT_sum"] / max(dd["TCC_HIT_sum"] + dd["TCC_MISS_sum"], 1), 3)
    e["launches_seen"] = max(cnt[k].values())
    res[k] = e
top = dict(sorted(res.items(), key=lambda kv: -kv[1].get("fetch_bytes", 0) * kv[1]["launches_seen"])[:10])
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum, one pass each, over "
           "`python3 bench.py --no-cpu-baseline --no-profile --extras 0 --steps 2 --warmup 1`; per-launch means; FETCH_SIZE (KB) x2 "
           "per the gfx950 correction, WRITE_SIZE (KB) as read", "kernels": top}, open(out, "w"), indent=1)
print(json.dumps(top, indent=1)[:1500])
