"""Aggregate a rocprofv3 kernel_trace.csv into per-kernel totals (whole run) and write a compact
text summary.  python tools/summarize_trace.py <trace.csv> <out.txt> [title]"""
import collections
import csv
import sys

trace, out = sys.argv[1], sys.argv[2]
title = sys.argv[3] if len(sys.argv) > 3 else trace
d = collections.defaultdict(lambda: [0, 0.0])
t0 = t1 = None
for r in csv.DictReader(open(trace)):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    t0 = s if t0 is None else min(t0, s)
    t1 = e if t1 is None else max(t1, e)
    v = d[r['Kernel_Name']]
    v[0] += 1
    v[1] += (e - s) / 1e3
tot = sum(v[1] for v in d.values())
cnt = sum(v[0] for v in d.values())
with open(out, 'w') as f:
    f.write(f"{title}\n{cnt} launches, {tot / 1e3:.2f} ms total kernel time, span {(t1 - t0) / 1e6:.1f} ms\n")
    for n, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:45]:
        f.write(f"{v[1] / 1e3:9.2f} ms {100 * v[1] / tot:5.1f}% n={v[0]:6d} avg={v[1] / v[0]:9.1f}us  {n[:110]}\n")
