# usage: bash tools/_w4abl.sh "13 45 77 109"   (on the CPU side: builds each variant, runs it on the GPU box)
for v in $1; do
  touch gan-2d-to-3d_amd/csrc/winograd4.hip
  G2S_HIPFLAGS_WINOGRAD4="$FLAGS -DW4DBG=$v" python gan-2d-to-3d_amd/build.py > /dev/null 2>&1
  echo "W4DBG=$v"
  /usr/local/graft/bin/gpurun --timeout 300 -- 'G2S_W4_SIGS=2,0 timeout -k 10 120 python tools/bench_wino4.py 0 2>&1 | grep "^(" | cut -c1-100' 2>&1 | grep "^("
done
touch gan-2d-to-3d_amd/csrc/winograd4.hip; python gan-2d-to-3d_amd/build.py > /dev/null 2>&1
