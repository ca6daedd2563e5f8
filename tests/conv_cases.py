"""Shared helpers of the convolution parity tests (CPU and GPU).

`expected_modconv(x, w, s, demod, mode, transpose)` gives what g2s_modconv must return, computed by
the C oracle (oracle/g2s_oracle.c: the reference's ModulatedConv2d.forward,
stylegan2-pytorch/model.py:250-291, restated literally).  The oracle has no data-gradient entry
point; the transposed call is expressed through the same oracle function with the adjoint's
weights (channels swapped, taps flipped for the padded stride-1 geometry):

    PLAIN^T (g) = PLAIN (g, w'[i,o,ky,kx] = w[o,i,k-1-ky,k-1-kx])
    UP2^T   (g) = DOWN2 (g, w'[i,o,ky,kx] = w[o,i,ky,kx])
    DOWN2^T (g) = UP2   (g, w'[i,o,ky,kx] = w[o,i,ky,kx])

tests/test_oracle_ops.py checks these identities against torch's autograd on the CPU.
"""
import numpy as np

from oracle import capi

PLAIN, UP2, DOWN2 = 0, 1, 2


def adjoint_weights(w, mode):
    """Weights w' and mode' such that conv_mode(w)^T == conv_mode'(w')."""
    wt = np.ascontiguousarray(np.swapaxes(w, 0, 1))
    if mode == PLAIN:
        return np.ascontiguousarray(wt[:, :, ::-1, ::-1]), PLAIN
    return wt, (DOWN2 if mode == UP2 else UP2)


def expected_modconv(x, w, in_scale, out_scale, mode, transpose):
    """x [B,C,H,W]; w [Cout,Cin,k,k] (already scaled); in_scale [B,C] or None multiplies the input
    channels, out_scale [B,Cy] or None the output channels (what g2s_modconv computes)."""
    if transpose:
        w, mode = adjoint_weights(w, mode)
    B, C = x.shape[:2]
    s = np.ones((B, C), np.float32) if in_scale is None else in_scale
    y = capi.modconv(x, w, s, 1.0, False, mode)
    if out_scale is not None:
        y = y * out_scale[:, :, None, None]
    return y
