import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"]); t1 = int(rows[-1]["End_Timestamp"])
pat = sys.argv[2]
hits = [r for r in rows if pat in r["Kernel_Name"]]
print(f"span {(t1-t0)/1e6:.0f} ms; {len(hits)} launches of *{pat}*")
if hits:
    ts = [(int(r['Start_Timestamp']) - t0) / 1e6 for r in hits]
    import collections
    buckets = collections.Counter(int(t // 500) * 500 for t in ts)
    for b in sorted(buckets): print(f"  t={b:6d}..{b+500} ms: {buckets[b]}")
