"""Summarise the timed region of a `G2S_BENCH_MARK=1 python bench.py ...` rocprofv3 kernel trace:
keeps the launches between the two marker kernels (bitwise_not), prints category totals per
iteration and the top kernels.  python tools/window_trace.py <kernel_trace.csv> <iters> [top] [summary.json]
The optional JSON holds the per-iteration time of the fp32-MFMA convolution kernels in the timed region:
bench.py reads the newest committed profiles/r*_conv_rocprof.json and prices the run's executed FLOP against
it (roofline.rocprof) beside its own HIP-event figure."""
import collections
import csv
import json
import sys

trace, iters = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
rows = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "bitwise_not" in r["Kernel_Name"]]
assert len(marks) >= 2, f"need two marker kernels, found {len(marks)}"
rows = rows[marks[0] + 1:marks[1]]
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6
cats = collections.OrderedDict([
    ("g2s modconv (MFMA implicit GEMM)", ("modconv_kernel", "conv_bwd_kernel")), ("g2s winograd F(2x2,3x3) (MFMA)", ("wino_kernel",)), ("g2s winograd F(4x4,3x3) (MFMA)", ("wino4_kernel",)), ("g2s upfirdn2d", ("upfirdn2d",)),
    ("g2s bias/act", ("fba_", "noise_bias")), ("g2s raster", ("raster_",)),
    ("g2s geometry/shading/lpips/rowops", ("g2s::",)),
    ("MIOpen conv (winograd/igemm/gemm)", ("miopenSp3", "igemm_", "Cijk_", "gemm", "Im2d2Col", "Col2Im", "naive_conv", "MIOpen", "conv")),
    ("layout transposes", ("batched_transpose", "transpose")),
    ("norm/pool/upsample/grid_sample", ("RowwiseMoments", "ComputeFused", "ComputeInternalGradients", "GroupNorm", "avg_pool", "max_pool", "upsample", "grid_sampler", "SubTensorOp", "OpTensor")),
    ("reductions", ("reduce_kernel",)), ("optimizer (foreach)", ("multi_tensor", "foreach")),
    ("copies/fills", ("copyBuffer", "fillBuffer", "FillFunctor", "copy_", "CatArray", "direct_copy")),
    ("elementwise (aten)", ("elementwise", "at::native")),
])
tot, cnt = collections.Counter(), collections.Counter()
per = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for c, pats in cats.items():
        if any(p in n for p in pats):
            break
    else:
        c = "other"
    tot[c] += d
    cnt[c] += 1
    per[n][0] += 1
    per[n][1] += d
T = sum(tot.values())
print(f"timed region: {len(rows)} launches over {span:.1f} ms = {span / iters:.3f} ms/iteration wall, "
      f"{T / iters / 1e3:.3f} ms kernel time and {len(rows) / iters:.0f} launches per iteration")
for c, v in tot.most_common():
    print(f"  {c:38s} {v / iters / 1e3:7.3f} ms  {100 * v / T:5.1f}%  {cnt[c] / iters:7.1f} launches/iter")
print("top kernels (per iteration):")
for n, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {v[1] / iters / 1e3:7.3f} ms {100 * v[1] / T:5.1f}% n={v[0] / iters:6.1f} avg={v[1] / v[0]:8.1f}us  {n[:120]}")
if len(sys.argv) > 4:
    direct, wino2, wino4 = (tot["g2s modconv (MFMA implicit GEMM)"], tot["g2s winograd F(2x2,3x3) (MFMA)"],
                            tot["g2s winograd F(4x4,3x3) (MFMA)"])
    wino = wino2 + wino4
    with open(sys.argv[4], "w") as f:
        json.dump({"what": "rocprofv3 --kernel-trace durations inside the graph-replayed timed region of bench.py (G2S_BENCH_MARK=1), per iteration",
                   "iterations": iters, "launches_per_iteration": len(rows) / iters, "kernel_ms_per_iteration": T / iters / 1e3,
                   "wall_ms_per_iteration": span / iters,
                   "conv_ms_per_iteration": {"direct": direct / iters / 1e3, "winograd": wino / iters / 1e3,
                                             "winograd_f2x2": wino2 / iters / 1e3, "winograd_f4x4": wino4 / iters / 1e3,
                                             "total": (direct + wino) / iters / 1e3},
                   "conv_launches_per_iteration": {"direct": cnt["g2s modconv (MFMA implicit GEMM)"] / iters,
                                                   "winograd": (cnt["g2s winograd F(2x2,3x3) (MFMA)"] + cnt["g2s winograd F(4x4,3x3) (MFMA)"]) / iters}}, f, indent=1)
