"""Mean HBM-side bytes per g2s::modconv_kernel launch from two rocprofv3 --pmc passes of
tools/pmc_iter.py (FETCH_SIZE, WRITE_SIZE; units KiB; FETCH_SIZE doubled per the gfx950 correction
of MI355X_MICROARCH.md, confirmed on 4 B/lane and 16 B/lane streams by tools/pmc_calib.py).  Only
the launches after the marker kernel (the 20-step 7:7:6 cycle) are counted.
python tools/pmc_modconv_traffic.py fetch.csv write.csv out.json"""
import csv, json, sys


def mean(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    marks = [int(r["Dispatch_Id"]) for r in rows if "bitwise_not" in r["Kernel_Name"]]
    start = marks[0] if marks else -1
    vals = [float(r["Counter_Value"]) for r in rows
            if "modconv_kernel" in r["Kernel_Name"] and int(r["Dispatch_Id"]) > start]
    return sum(vals) / len(vals), len(vals)


f, nf = mean(sys.argv[1], "FETCH_SIZE")
w, nw = mean(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "g2s::modconv_kernel", "launches_fetch_pass": nf, "launches_write_pass": nw,
       "fetch_bytes_per_launch": 2 * f * 1024, "write_bytes_per_launch": w * 1024,
       "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024,
       "correction": "FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B), WRITE_SIZE x1",
       "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 tools/pmc_iter.py  (one eager 20-step "
                  "7:7:6 cycle of face128_n8 after warm-up; separate passes per counter)"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out))
