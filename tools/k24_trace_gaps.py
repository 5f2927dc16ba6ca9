"""Summarise a rocprofv3 kernel trace of `bench.py --code 224`: busy time, gaps and per-pass durations of the last update."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"][:48], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
seg = [k for k in ks if "k24" in k[0] or "chainback" in k[0]]
idx = [i for i, k in enumerate(seg) if "chainback" in k[0]]
s = seg[idx[-2] + 1:idx[-1]]
busy = sum(e - st for _, st, e in s)
span = s[-1][2] - s[0][1]
gaps = [s[i + 1][1] - s[i][2] for i in range(len(s) - 1)]
print(len(s), "kernels; busy", busy / 1e3, "us; span", span / 1e3, "us; gaps", sum(gaps) / 1e3, "us")
big = [g for g in gaps if g > 5000]
print("gaps > 5 us:", len(big), "total", sum(big) / 1e3, "us; other gaps avg", (sum(gaps) - sum(big)) / max(1, len(gaps) - len(big)), "ns")
c = collections.Counter()
d = collections.Counter()
for n, st, e in s:
    if e - st > 4000:
        c[n] += e - st
        d[n] += 1
for n in c:
    print(n, d[n], round(c[n] / d[n] / 1e3, 2), "us")
short = [(n, e - st) for n, st, e in s if e - st < 4000 and "pass" in n]
print("cancelled passes:", len(short), "total", sum(x for _, x in short) / 1e3, "us")
