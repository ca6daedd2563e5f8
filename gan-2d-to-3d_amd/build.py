"""Build libg2s.so (hand-written HIP kernels for gfx950 + the C ABI of include/g2s.h).

    python gan-2d-to-3d_amd/build.py [--force]

hipcc cross-compiles for gfx950 without a GPU.  Objects and the library stay in-tree
(gan-2d-to-3d_amd/lib/, git-ignored) so that they travel to the GPU box with the snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libg2s.so")
ARCH = "gfx950"

SOURCES = {
    # file: extra flags
    "api.hip": [],
    "raster.hip": ["-ffp-contract=off"],  # bit-parity with the oracle's unfused arithmetic
    "raster_rgb.hip": ["-ffp-contract=off"],
    "fused_bias_act.hip": [],
    "upfirdn2d.hip": [],
    "modconv.hip": [],
    "thinconv.hip": [],
    # packed f32 VALU (v_pk_*) beside MFMAs costs more than it saves (MI355X_MICROARCH.md): no SLP packing
    "winograd.hip": ["-fno-slp-vectorize"],
    "winograd4.hip": ["-fno-slp-vectorize"],
    "split_reduce.hip": [],
    "rowops.hip": [],
    "lpips.hip": [],
    "pool.hip": [],
    "losses.hip": [],
    "geometry.hip": ["-ffp-contract=off"],
    "grid_sample.hip": ["-ffp-contract=off"],
    "groupnorm.hip": [],
    "conv_wgrad.hip": [],
    "adam.hip": [],
    "probe.hip": [],
}
# The kernels take their descriptor struct by value.  Clang copies such a parameter into a private
# alloca and relies on InstCombine to fold the copy back onto the (constant) kernarg segment — a
# transform that gives up after 300 users of the alloca.  The convolution kernel, with ~17 inlined
# template bodies, is beyond that: the whole descriptor then lives in scratch, every field read back
# counts as divergent and each buffer load is wrapped in a waterfall loop (half the MFMA rate; the
# disassembly test in tests/test_host_cpu.py guards against it).  Raise the limit.
COMMON = ["--offload-arch=" + ARCH, "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-mllvm", "-instcombine-max-copied-from-constant-users=100000"]


def _newer(src, dst, deps):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(d) > t for d in [src] + deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    headers.append(os.path.join(HERE, "..", "include", "g2s.h"))
    objs, jobs = [], []
    for name, extra in SOURCES.items():
        src = os.path.join(CSRC, name)
        if not os.path.exists(src):
            continue
        obj = os.path.join(LIBDIR, name.replace(".hip", ".o"))
        if force or _newer(src, obj, headers):
            dev = os.environ.get("G2S_HIPFLAGS_" + name.split(".")[0].upper(), "").split()   # measurement builds
            jobs.append([hipcc] + COMMON + extra + dev + ["-c", src, "-o", obj])
        objs.append(obj)
    relink = force or bool(jobs)
    if jobs:
        # the translation units are independent: compile them side by side (the two MFMA convolution
        # kernels take minutes each, the rest seconds)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as pool:
            list(pool.map(run, jobs))
    if relink or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
