set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
cd /tmp
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f --output-format csv -- python3 $R/tools/pmc_iter.py > $O/pmc_fetch.log 2>&1
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python3 $R/tools/pmc_iter.py > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_modconv_traffic.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $O/modconv_pmc.json
