"""CPU sanitizer pass: the CPU-side C / C++ of the test infrastructure — the oracle
(oracle/g2s_oracle.c + raster_body.inc) and the host emulation of the product's tile rasterizer
(tests/raster_tile_emulation.cpp, which compiles the product header csrc/raster_core.h) — built with
AddressSanitizer + UndefinedBehaviorSanitizer (-fno-sanitize-recover: any finding aborts) and
driven through the raster / soup / backward / render_rgb / ops cases of the ordinary CPU tests.
CPU only: nothing of the kind runs on the GPU box's card."""
import os
import subprocess
import sys

import pytest

from oracle import capi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.timeout(600)
def test_oracle_and_tile_emulation_under_asan_ubsan():
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    so = capi.build_sanitized(os.path.join(HERE, "_build"))
    env = dict(os.environ, G2S_ORACLE_LIB=so, LD_PRELOAD=os.path.realpath(asan),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(HERE, "test_raster_core_cpu.py"), os.path.join(HERE, "test_oracle_raster.py"),
                        os.path.join(HERE, "test_oracle_ops.py")],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=580)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
