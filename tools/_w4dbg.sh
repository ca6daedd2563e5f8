for dbg in 0 1 13 16; do echo "dbg=$dbg"; G2S_W4_DBG=$dbg G2S_W4_SIGS=2,0,4 timeout -k 10 120 python tools/bench_wino4.py 0 2>&1 | grep "^(" ; done
