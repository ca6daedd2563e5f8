"""Averages per counter of the Winograd kernels in gpurun_out/w4pmc/*.csv (tools/w4_pmc.sh)."""
import collections, csv, glob, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/w4pmc", "*.csv"))):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "wino" in r["Kernel_Name"] and "weights" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:34], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (n, c), v in sorted(agg.items()):
        print(os.path.basename(f), n, f"{c:28s} n={len(v)} avg={sum(v) / len(v):.4g}")
