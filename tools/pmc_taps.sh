#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM" "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TA_TCP_STATE_READ_sum" "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $C -d /tmp/pt$i -o p --output-format csv -- python3 $R/tools/pmc_taps.py > /tmp/pt$i.log 2>&1
  F=$(find /tmp/pt$i -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "modconv_kernel" in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(by)
names = ["k1(T=1)", "k2(T=4)", "k3(T=9)", "k4(T=16)"]
for gi in range(0, len(ids), 3):
    grp = ids[gi:gi + 3]
    agg = collections.OrderedDict()
    for d in grp:
        for c, v in by[d].items():
            agg[c] = agg.get(c, 0) + v / len(grp)
    print(names[gi // 3] if gi // 3 < 4 else gi, " ".join(f"{c}={v:.4g}" for c, v in agg.items()))
PY
done
