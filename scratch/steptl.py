"""Timeline of the last generator warm-up step in a rocprofv3 kernel trace: per stream busy time,
per kernel family time, and the largest idle gaps on the main chain."""
import csv, glob, sys, re
from collections import defaultdict
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id"))))
rows.sort()
def nm(k):
    m = re.search(r"(conv_kernel|wgrad_kernel)<[^>]*>", k)
    return m.group(0) if m else re.sub(r"^.*N_1\d\d?", "", k.split("(")[0])[:40]
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
lo, hi = adam[-2], adam[-1]
step = rows[lo + 1:hi + 1]
t0 = step[0][0]
print("step span %.3f ms, kernels %d" % ((step[-1][1] - t0) / 1e6, len(step)))
by_stream = defaultdict(list)
for s, e, k, st in step:
    by_stream[st].append((s, e, k))
for st, L in by_stream.items():
    fam = defaultdict(float)
    for s, e, k in L:
        fam[nm(k)] += (e - s) / 1e6
    print("stream", st, "n=%d busy %.3f ms first %.3f last %.3f" % (len(L), sum(e - s for s, e, _ in L) / 1e6, (L[0][0] - t0) / 1e6, (L[-1][1] - t0) / 1e6))
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:6]:
        print("     %-60s %.3f ms" % (k, v))
