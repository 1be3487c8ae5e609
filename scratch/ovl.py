"""Overlap analysis of a rocprofv3 kernel trace: conv durations inside / outside wgrad windows."""
import csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
lo, hi = adam[-2], adam[-1]
step = rows[lo + 1:hi + 1]
t0 = step[0][0]
print("step span %.3f ms, kernels %d" % ((step[-1][1] - t0) / 1e6, len(step)))
wg = [(s, e) for s, e, k, q in step if "wgrad_kernel" in k]
for s, e in wg:
    print("  wgrad %8.3f -> %8.3f ms  (%.3f)" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6))
def inside(s, e):
    return any(ws <= s and e <= we for ws, we in wg)
for tag in ("E, 1, 2, 5, 3>", "E, 2, 2, 5, 3>"):
    a = [e - s for s, e, k, q in step if tag in k and "conv" in k and inside(s, e)]
    b = [e - s for s, e, k, q in step if tag in k and "conv" in k and not inside(s, e)]
    print(tag, "inside wgrad: n=%d avg %.2f us | outside: n=%d avg %.2f us" % (
        len(a), sum(a) / max(len(a), 1) / 1e3, len(b), sum(b) / max(len(b), 1) / 1e3))
# busy union
ev = sorted([(s, 1) for s, e, k, q in step] + [(e, -1) for s, e, k, q in step])
def first(pred):
    for s, e, k, q in step:
        if pred(k):
            return (s - t0) / 1e6, (e - t0) / 1e6
    return None
def last(pred):
    r = None
    for s, e, k, q in step:
        if pred(k):
            r = ((s - t0) / 1e6, (e - t0) / 1e6)
    return r
print("pack      ", first(lambda k: "pack_kernel" in k))
print("loss (end of forward)", first(lambda k: "pix_loss" in k))
print("last conv ", last(lambda k: "conv_kernel" in k))
print("last wgrad", last(lambda k: "wgrad_kernel" in k))
print("unpack    ", last(lambda k: "unpack" in k))
print("adam      ", last(lambda k: "adam" in k))
tot = {}
for s, e, k, q in step:
    import re
    m = re.search(r"(conv_kernel|wgrad_kernel)<[^>]*>", k)
    kk = m.group(0) if m else k.split("(")[0][-40:]
    tot.setdefault(kk, [0, 0]); tot[kk][0] += e - s; tot[kk][1] += 1
for kk, (t, n) in sorted(tot.items(), key=lambda kv: -kv[1][0])[:8]:
    print(f"{t/1e6:8.3f} ms {n:5d} x {t/n/1e3:8.2f} us  {kk}")
