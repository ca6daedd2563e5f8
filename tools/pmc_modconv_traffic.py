"""Mean HBM-side bytes per launch of the fp32-MFMA convolution kernels (g2s::modconv_kernel, direct
implicit GEMM, and g2s::conv_bwd_kernel, its fused data- + weight-gradient form; g2s::wino_kernel,
Winograd F(2x2,3x3)) from two rocprofv3 --pmc passes of
tools/pmc_iter.py (FETCH_SIZE, WRITE_SIZE; units KiB; FETCH_SIZE doubled per the gfx950 correction
of MI355X_MICROARCH.md, confirmed on 4 B/lane and 16 B/lane streams by tools/pmc_calib.py).  Only
the launches after the marker kernel (the 20-step 7:7:6 cycle) are counted.
python tools/pmc_modconv_traffic.py fetch.csv write.csv out.json"""
import csv, json, sys

KERNELS = {"direct": ("modconv_kernel", "conv_bwd_kernel"), "winograd": ("wino_kernel", "wino4_kernel")}


def load(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    marks = [int(r["Dispatch_Id"]) for r in rows if "bitwise_not" in r["Kernel_Name"]]
    start = marks[0] if marks else -1
    return [r for r in rows if int(r["Dispatch_Id"]) > start]


def mean(rows, pat):
    vals = [float(r["Counter_Value"]) for r in rows if any(p in r["Kernel_Name"] for p in pat)]
    return (sum(vals) / len(vals) if vals else 0.0), len(vals)


fr, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
ALL = [p for pats in KERNELS.values() for p in pats]
f, nf = mean(fr, ALL)
w, nw = mean(wr, ALL)
per = {}
for name, pat in KERNELS.items():
    fk, nk = mean(fr, pat)
    wk, _ = mean(wr, pat)
    per[name] = {"launches": nk, "fetch_bytes_per_launch": 2 * fk * 1024, "write_bytes_per_launch": wk * 1024,
                 "traffic_bytes_per_launch": 2 * fk * 1024 + wk * 1024}
out = {"kernel": "g2s::modconv_kernel + g2s::conv_bwd_kernel + g2s::wino_kernel + g2s::wino4_kernel", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
       "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024, "per_kernel": per,
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1",
       "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 tools/pmc_iter.py  (one eager 20-step "
                  "7:7:6 cycle of face128_n8 after warm-up; separate passes per counter)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
