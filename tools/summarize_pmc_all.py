import collections, csv, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "modconv" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][20:50], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        d[(r["Kernel_Name"][20:50], r["Grid_Size"])]["_dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()})
