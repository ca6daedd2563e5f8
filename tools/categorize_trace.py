"""Per-iteration kernel time by category from a rocprofv3 kernel trace of `bench.py --only K`."""
import collections, csv, sys
trace, iters = sys.argv[1], float(sys.argv[2])
cats = collections.OrderedDict([
    ("g2s modconv", ("modconv_kernel",)), ("g2s upfirdn2d", ("upfirdn2d",)), ("g2s bias/act", ("fba_", "noise_bias")),
    ("g2s raster", ("raster_",)),
    ("MIOpen conv (winograd/igemm/gemm)", ("miopenSp3", "igemm_", "Cijk_", "gemm", "Im2d2Col", "Col2Im", "naive_conv", "MIOpen")),
    ("layout transposes", ("batched_transpose", "transpose")),
    ("norm/pool/upsample/grid_sample", ("RowwiseMoments", "ComputeFused", "ComputeInternalGradients", "GroupNorm", "avg_pool", "max_pool", "upsample", "grid_sampler", "SubTensorOp", "OpTensor")),
    ("reductions", ("reduce_kernel",)), ("optimizer (foreach)", ("multi_tensor", "foreach")),
    ("copies/fills", ("copyBuffer", "fillBuffer", "FillFunctor", "copy_", "CatArray", "direct_copy")),
    ("elementwise (aten)", ("elementwise", "at::native")),
])
tot = collections.Counter(); cnt = collections.Counter()
for r in csv.DictReader(open(trace)):
    n = r["Kernel_Name"]; d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for c, pats in cats.items():
        if any(p in n for p in pats):
            break
    else:
        c = "other"
    tot[c] += d; cnt[c] += 1
T = sum(tot.values())
print(f"per iteration: {T/iters/1e3:.2f} ms kernel time, {sum(cnt.values())/iters:.0f} launches")
for c, v in tot.most_common():
    print(f"  {c:36s} {v/iters/1e3:7.3f} ms  {100*v/T:5.1f}%  {cnt[c]/iters:7.1f} launches")
