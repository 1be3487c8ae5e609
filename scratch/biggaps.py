import csv, glob, sys, re
path = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
def nm(k):
    m = re.search(r"(conv_kernel|wgrad_kernel)<[^>]*>", k)
    return m.group(0) if m else re.sub(r"^.*N_1\d\d?", "", k.split("(")[0])[:40]
adam = [i for i, r in enumerate(rows) if "adam" in r[2]]
lo, hi = adam[-3], adam[-1]          # one full adversarial step = two adam launches (D, G)
step = rows[lo + 1:hi + 1]
t0 = step[0][0]
print("step span %.3f ms, kernels %d, busy %.3f ms" % ((step[-1][1] - t0) / 1e6, len(step), sum(e - s for s, e, _ in step) / 1e6))
gaps = sorted(((step[i + 1][0] - step[i][1], i) for i in range(len(step) - 1)), reverse=True)
for g, i in gaps[:25]:
    print("%8.1f us after %-42s before %-42s at %.3f ms" % (g / 1e3, nm(step[i][2]), nm(step[i + 1][2]), (step[i][1] - t0) / 1e6))
small = [g for g, _ in gaps if g < 20000]
print("gaps <20us: n=%d sum=%.3f ms" % (len(small), sum(small) / 1e6))
