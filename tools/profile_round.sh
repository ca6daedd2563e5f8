#!/bin/bash
# Round-end evidence (run on the GPU box): kernel-trace stats of the default bench command, the two
# PMC passes behind roofline.traffic, per-step-kind breakdowns.  Outputs under gpurun_out/round/.
#   bash tools/profile_round.sh
set -e
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
cd /tmp
echo "[1] warm bench (MIOpen / library initialisation)"
python3 $R/bench.py --steps 20 --warmup 10 --no-cpu-baseline > $O/warm.json 2> $O/warm.err
echo "[2] kernel trace + stats of the default bench command"
G2S_BENCH_MARK=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python3 $R/tools/window_trace.py $(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1) 40 40 $O/conv_rocprof.json > $O/bench_timed_region.txt
echo "[3] PMC FETCH_SIZE"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmc_f -o f --output-format csv -- python3 $R/tools/pmc_iter.py > $O/pmc_fetch.log 2>&1
echo "[4] PMC WRITE_SIZE"
timeout -k 5 400 rocprofv3 --pmc WRITE_SIZE -d /tmp/pmc_w -o w --output-format csv -- python3 $R/tools/pmc_iter.py > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_modconv_traffic.py $(find /tmp/pmc_f -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_w -name "*counter_collection.csv" | head -1) $O/modconv_pmc.json
echo "[5] per-step-kind breakdowns"
bash $R/tools/profile_kinds.sh "1 2 3"
cp $R/gpurun_out/kinds/step*.txt $O/
echo "[6] micro-benchmarks"
python3 $R/tools/bench_raster.py > $O/raster_microbench.txt 2>&1
python3 $R/tools/conv_census.py > $O/conv_census.txt 2>&1 || true
python3 $R/tools/bench_wino.py 1 -256 > $O/wino_microbench.txt 2>&1 || true
python3 $R/bench.py --workload face256_fp16 --no-cpu-baseline > $O/bench_face256_fp16.json 2> $O/bench_face256_fp16.err || true
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
echo done
