"""Per-queue view of one training step from a rocprofv3 --kernel-trace csv: busy time per queue, time
during which NO queue runs anything, and the longest such holes with the kernels either side."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
adam = [r for r in rows if "adam" in r["Kernel_Name"]]
# a step ends with the generator's Adam (the larger launch of the two)
big = sorted(adam, key=lambda r: r["e"] - r["s"], reverse=True)[: len(adam) // 2 + 1]
ends = sorted(r["e"] for r in big)
t0, t1 = ends[-3], ends[-2]
seg = [r for r in rows if t0 <= r["s"] < t1]
print("step window %.3f ms, %d kernels" % ((t1 - t0) / 1e6, len(seg)))
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
for q, v in byq.items():
    busy = sum(r["e"] - r["s"] for r in v)
    print("queue %s: %4d kernels, busy %.3f ms, first %.3f last %.3f" % (q, len(v), busy / 1e6, (v[0]["s"] - t0) / 1e6, (v[-1]["e"] - t0) / 1e6))
# union of busy intervals
iv = sorted((r["s"], r["e"], r) for r in seg)
holes, cur_e, last = [], t0, None
for s, e, r in iv:
    if s > cur_e:
        holes.append((s - cur_e, cur_e, last, r))
    if e > cur_e:
        cur_e, last = e, r
tot = sum(h[0] for h in holes)
print("nothing running: %.3f ms in %d holes (>5 us: %.3f ms)" % (tot / 1e6, len(holes), sum(h[0] for h in holes if h[0] > 5000) / 1e6))
for d, at, a, b in sorted(holes, reverse=True)[:12]:
    print("  %7.1f us at %.3f ms  after %-40s before %s" % (d / 1e3, (at - t0) / 1e6, (a["Kernel_Name"][-60:] if a else "-"), b["Kernel_Name"][-60:]))
