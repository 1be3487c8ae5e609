"""Timeline of one training step from a rocprofv3 --kernel-trace csv: where the chain ends, what the
weight-gradient queue still runs after it, and how long until the next step's first convolution."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
byq = collections.defaultdict(list)
for r in rows:
    byq[r["Queue_Id"]].append(r)
mainq = max(byq, key=lambda q: len(byq[q]))
m = byq[mainq]
adam = [i for i, r in enumerate(m) if "adam" in r["Kernel_Name"]]
a, b = adam[-3], adam[-2]
t0 = m[a]["e"]
seg = m[a + 1:b + 1]
nxt = m[b + 1:b + 12]
def rel(t): return (t - t0) / 1e3
print("previous adam ends at 0; this step's main-queue kernels: %d" % len(seg))
for r in seg[:6]:
    print("  %9.1f %9.1f  %s" % (rel(r["s"]), rel(r["e"]), r["Kernel_Name"][:70]))
convs = [r for r in seg if "conv_kernel" in r["Kernel_Name"]]
li = [i for i, r in enumerate(seg) if "pix_loss" in r["Kernel_Name"]][0]
print("forward: first conv starts %.1f, loss kernel at %.1f" % (rel(convs[0]["s"]), rel(seg[li]["s"])))
lastconv = convs[-1]
print("last conv of the chain ends %.1f" % rel(lastconv["e"]))
print("main queue after the last conv:")
for r in seg[seg.index(lastconv) + 1:]:
    print("  %9.1f %9.1f  %s" % (rel(r["s"]), rel(r["e"]), r["Kernel_Name"][:70]))
print("next step begins:")
for r in nxt[:5]:
    print("  %9.1f %9.1f  %s" % (rel(r["s"]), rel(r["e"]), r["Kernel_Name"][:70]))
for q, v in byq.items():
    if q == mainq: continue
    sv = [r for r in v if m[a]["e"] <= r["s"] <= m[b]["e"]]
    print("queue %s: %d kernels" % (q, len(sv)))
    for r in sv:
        print("  %9.1f %9.1f  (%7.1f us) %s" % (rel(r["s"]), rel(r["e"]), (r["e"] - r["s"]) / 1e3, r["Kernel_Name"][:60]))
