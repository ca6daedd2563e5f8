// conv_wgrad.hip — weight gradient of the small trained nets' convolutions (g2s_conv2d_wgrad,
// include/g2s.h) as an fp32-MFMA GEMM whose reduction runs over pixels:
//
//     dw[a][n = (g, ky, kx)] = sum_{K = (b, py, px)} A[b,a,py,px] * G[b,g,py*s+ky-p,px*s+kx-p]
//
// Replaces the weight-gradient half of torch's ConvolutionBackward for nn.Conv2d (A = grad_out,
// G = input) and nn.ConvTranspose2d (A = input, G = grad_out) of GAN2Shape/networks.py:23-244.
// Tiles: 64 (a) x 64 (n) outputs per workgroup, 32 pixels per K tile, 2x2 waves of one 32x32 MFMA
// tile each.  Both operands are read with lanes along the pixel index (contiguous in memory) and
// stored k-major in LDS, so the MFMA operand fetch is a conflict-free ds_read_b32.  The pixel
// reduction is split over blockIdx.y until the chip is full; partial sums meet in float atomics.
#include "conv_wgrad_core.h"

namespace g2s {

__global__ __launch_bounds__(WG_THREADS) void conv_wgrad_kernel(WgradParams p) {
    __shared__ float As[WG_BK][WG_BM + 1];
    __shared__ float Bs[WG_BK][WG_BN + 1];
    wgrad_block(p, As, Bs, blockIdx.x, blockIdx.y, blockIdx.z);
}

}  // namespace g2s

using namespace g2s;

static int wgrad_impl(const float *A, const float *G, float *dw, int B, int Ca, int Cg, int PH, int PW, int GH,
                      int GW, int k, int stride, int pad, int dw_is_zero, int groups, g2s_stream_t stream) {
    WgradParams p;
    int tiles, split;
    const int rc = wgrad_plan(A, G, dw, B, Ca, Cg, PH, PW, GH, GW, k, stride, pad, groups, p, tiles, split);
    if (rc != G2S_OK) return rc;
    hipStream_t st = as_stream(stream);
    if (p.atomic && !dw_is_zero &&
        hipMemsetAsync(dw, 0, (size_t)groups * Ca * p.N * sizeof(float), st) != hipSuccess)
        return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(dw) failed");
    conv_wgrad_kernel<<<dim3(tiles, split, groups), WG_THREADS, 0, st>>>(p);
    return check_launch("g2s_conv2d_wgrad");
}

extern "C" int g2s_conv2d_wgrad(const float *A, const float *G, float *dw, int B, int Ca, int Cg,
                                int PH, int PW, int GH, int GW, int k, int stride, int pad,
                                int dw_is_zero, g2s_stream_t stream) {
    return wgrad_impl(A, G, dw, B, Ca, Cg, PH, PW, GH, GW, k, stride, pad, dw_is_zero, 1, stream);
}

extern "C" int g2s_conv2d_wgrad_grouped(const float *A, const float *G, float *dw, int B, int Ca, int Cg,
                                        int PH, int PW, int GH, int GW, int k, int stride, int pad,
                                        int dw_is_zero, int groups, g2s_stream_t stream) {
    return wgrad_impl(A, G, dw, B, Ca, Cg, PH, PW, GH, GW, k, stride, pad, dw_is_zero, groups, stream);
}
