#!/bin/bash
# Per-step-kind kernel breakdown of the graph-replayed timed region (run on the GPU box).
#   bash tools/profile_kinds.sh [kinds, default "1 2 3"]
set -e
export TMPDIR=/tmp G2S_BENCH_MARK=1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/kinds
cd /tmp
for K in ${1:-1 2 3}; do
  rocprofv3 --kernel-trace -d /tmp/prof_k$K -o k$K --output-format csv -- python3 $R/bench.py --only $K --steps 20 --warmup 10 --no-cpu-baseline > $R/gpurun_out/kinds/bench_k$K.json 2> $R/gpurun_out/kinds/bench_k$K.err
  F=$(find /tmp/prof_k$K -name "*kernel_trace.csv" | head -1)
  python3 $R/tools/window_trace.py $F 20 250 > $R/gpurun_out/kinds/step$K.txt
  echo "step $K done"
done
