"""Per-step timeline from a rocprofv3 --kernel-trace csv: kernel time vs gaps on the main chain."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split by queue
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
print("queues:", {q: len(v) for q, v in byq.items()})
mainq = max(byq, key=lambda q: len(byq[q]))
m = byq[mainq]
# steps: find adam kernels
names = [r["Kernel_Name"] for r in m]
adam = [i for i, n in enumerate(names) if "adam" in n]
print("adam launches:", len(adam))
if len(adam) >= 3:
    a, b = adam[-3], adam[-2]
    seg = m[a + 1:b + 1]
    t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    gaps = [int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"]) for i in range(len(seg) - 1)]
    print("step span %.3f ms, %d kernels on main queue, kernel time %.3f ms, gaps %.3f ms (median gap %.2f us)" % (
        (t1 - t0) / 1e6, len(seg), dur / 1e6, sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3))
    # forward part: up to the loss kernel
    li = [i for i, r in enumerate(seg) if "pix_loss" in r["Kernel_Name"]]
    if li:
        f = seg[:li[0]]
        fd = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f)
        print("forward: span %.3f ms, kernels %d, kernel time %.3f ms" % (
            (int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e6, len(f), fd / 1e6))
        bk = seg[li[0]:]
        bd = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in bk)
        print("backward(main queue): span %.3f ms, kernels %d, kernel time %.3f ms" % (
            (int(bk[-1]["End_Timestamp"]) - int(bk[0]["Start_Timestamp"])) / 1e6, len(bk), bd / 1e6))
    agg = collections.defaultdict(lambda: [0, 0])
    for r in seg:
        k = r["Kernel_Name"][:60]
        agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
    for k, (d, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:8]:
        print("  %-60s %8.3f ms %5d  %.2f us" % (k, d / 1e6, n, d / n / 1e3))
    # side queue in the same window
    for q, v in byq.items():
        if q == mainq: continue
        sv = [r for r in v if t0 <= int(r["Start_Timestamp"]) <= t1]
        if sv:
            sd = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in sv)
            print("queue %s in this step: %d kernels, %.3f ms, from %.3f to %.3f ms" % (
                q, len(sv), sd / 1e6, (int(sv[0]["Start_Timestamp"]) - t0) / 1e6, (int(sv[-1]["End_Timestamp"]) - t0) / 1e6))
