"""What the fp32 matrix pipe sustains (g2s_mfma_probe): back-to-back independent v_mfma_f32_32x32x2_f32 from registers,
1, 2, 4 or 8 waves per workgroup (8 = two per SIMD), one or several workgroups per CU.  python tools/bench_mfma_peak.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import gan2shape_amd  # noqa
from gan2shape_amd import lib

L = lib.load()
out = torch.zeros(4, device="cuda")
iters = 20000
print(f"{'blocks':>7s} {'waves':>6s} {'ms':>8s} {'TFLOP/s':>9s} {'of 157.3':>9s} {'cycles per MFMA and SIMD at 2.4 GHz':>38s}")
for blocks, waves in [(256, 4), (256, 8), (512, 4), (1024, 4), (256, 1), (256, 2), (2048, 4)]:
    for _ in range(2):
        lib.check(L.g2s_mfma_probe(lib.ptr(out), blocks, waves, iters, lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        lib.check(L.g2s_mfma_probe(lib.ptr(out), blocks, waves, iters, lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    flop = blocks * waves * iters * 8 * 4096.0
    tf = flop / ms / 1e9
    # per SIMD: waves on a SIMD = ceil(blocks / 256) * ceil(waves / 4) when everything is resident
    per_simd = max(1, -(-blocks // 256)) * max(1, -(-waves // 4)) * iters * 8
    print(f"{blocks:7d} {waves:6d} {ms:8.3f} {tf:9.1f} {tf / 157.3:9.3f} {ms * 1e-3 * 2.4e9 / per_simd:38.1f}", flush=True)

print("with 8 x 16-byte LDS operand reads per lane per 16 MFMAs (the Winograd inner loop's fragment traffic):")
for blocks, waves in [(256, 4), (256, 8)]:
    for agpr in (0, 1):
        for _ in range(2):
            lib.check(L.g2s_mfma_lds_probe(lib.ptr(out), blocks, waves, 5000, agpr, lib.stream()))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            lib.check(L.g2s_mfma_lds_probe(lib.ptr(out), blocks, waves, 5000, agpr, lib.stream()))
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        tf = blocks * waves * 5000 * 16 * 4096.0 / ms / 1e9
        print(f"{blocks:7d} {waves:6d}  accumulators in {'AccVGPRs ' if agpr else 'ArchVGPRs'}  {ms:8.3f} ms {tf:9.1f} TFLOP/s {tf / 157.3:7.3f}", flush=True)
