"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel name (+ grid)."""
import collections, csv, sys
path, counter = sys.argv[1], sys.argv[2]
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] != counter:
        continue
    d[(r["Kernel_Name"][:70], r["Grid_Size"])].append(float(r["Counter_Value"]))
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    if "modconv" in k[0] or "fba_" in k[0]:
        print(f"{counter} mean={sum(v)/len(v):14.1f} n={len(v):5d} grid={k[1]:>8s} {k[0]}")
