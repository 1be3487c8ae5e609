"""Summarise a rocprofv3 kernel trace: busy time, inter-kernel gaps, per-kernel stats."""
import csv
import glob
import sys
from collections import defaultdict

path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = len(rows)
tail = rows[n // 2:]          # steady state: the last half of the run
busy = sum(e - s for s, e, _ in tail)
span = tail[-1][1] - tail[0][0]
gaps = [tail[i + 1][0] - tail[i][1] for i in range(len(tail) - 1)]
gaps_small = [g for g in gaps if g < 50_000]
print(f"kernels {len(tail)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  ({100*busy/span:.1f} %)")
print(f"gaps: mean {sum(gaps_small)/len(gaps_small)/1e3:.2f} us  median {sorted(gaps_small)[len(gaps_small)//2]/1e3:.2f} us "
      f" sum(<50us) {sum(gaps_small)/1e6:.2f} ms  n>50us {len(gaps)-len(gaps_small)}")
per = defaultdict(lambda: [0, 0])
for s, e, k in tail:
    k = k.split("(")[0][:70]
    per[k][0] += e - s
    per[k][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{t/1e6:9.3f} ms {c:7d} x {t/c/1e3:8.2f} us  {k}")
