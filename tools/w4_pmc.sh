# PMC passes over one launch signature of the F(4x4) Winograd kernel (run on the GPU box): bash tools/w4_pmc.sh [f4|f2]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/w4pmc
mkdir -p $O
cd /tmp
k=${1:-f4}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM -d /tmp/p1_$k -o p --output-format csv -- python3 $R/tools/w4_one.py $k 8 256 256 64 > $O/p1_$k.log 2>&1
cp $(find /tmp/p1_$k -name "*counter_collection.csv" | head -1) $O/p1_$k.csv
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d /tmp/p3_$k -o p --output-format csv -- python3 $R/tools/w4_one.py $k 8 256 256 64 > $O/p3_$k.log 2>&1
cp $(find /tmp/p3_$k -name "*counter_collection.csv" | head -1) $O/p3_$k.csv
echo ok
