"""CPU oracle for the GAN2Shape hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package (gan-2d-to-3d_amd/) never imports it and has no CPU fallback.

  oracle.capi      ctypes bindings of oracle/libg2s_oracle.so (plain-C restatement: rasterizer,
                   fused_bias_act, upfirdn2d, modulated conv)
  oracle.geometry  numpy restatement of GAN2Shape/renderer/{renderer,utils}.py geometry
  oracle.losses    numpy restatement of GAN2Shape/losses.py

Pinning status:
  fused_bias_act, upfirdn2d, modconv, geometry, losses : pinned by golden vectors generated from
      the reference's own importable Python (tests/golden/make_golden.py).
  rasterizer (neural_renderer, external un-vendored CUDA package) : PARITY UNPINNED — analytic
      known-answer tests only.
"""
