// lpips.hip — the per-layer tail of the LPIPS distance in one pass per direction.
//
// Reference: GAN2Shape/stylegan2/stylegan2-pytorch/lpips/networks_basic.py:64-92 (PNetLin.forward)
// with lpips/__init__.py:40-42 (normalize_tensor): per layer
//     u = f0 / (sqrt(sum_c f0^2) + eps),  v = f1 / (sqrt(sum_c f1^2) + eps)
//     out[n] += mean_hw sum_c w_c (u_c - v_c)^2
// which the reference runs as ~16 elementwise / reduce / 1x1-conv launches per layer (and twice as
// many in backward).  Here: 64 pixels x 4..16 channel slices per workgroup, two sweeps (norms, then the
// weighted squared difference); lanes run along pixels so every channel plane is read coalesced.
// Backward gives the gradient w.r.t. f0 only (f1 is the target branch, no gradient in GAN2Shape:
// model.py:159-160,275-276 mask the target with a detached mask of the input image).
#include "g2s_common.h"

namespace g2s {

constexpr float LPIPS_EPS = 1e-10f;

// Workgroup = 64 pixels (lanes: coalesced along a channel plane) x SL channel slices (waves).
// Channel sums are combined across the slices through LDS.  SL = 16 for the small late layers
// (few pixels, 512 channels) keeps enough loads in flight; SL = 4 for the big early layers.
template <int SL>
__device__ __forceinline__ float slice_sum(float v, float (*red)[64], int sl, int px) {
    __syncthreads();  // protect the previous use of `red`
    red[sl][px] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int i = 0; i < SL; i++) t += red[i][px];
    return t;
}

// grid (ceil(HW / 64), N), block 64 * SL
template <int SL>
__global__ __launch_bounds__(64 * SL) void lpips_layer_fwd(const float *__restrict__ f0,
                                                           const float *__restrict__ f1,
                                                           const float *__restrict__ w,
                                                           float *__restrict__ out, int C, int HW) {
    __shared__ float red[SL][64];
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int n = blockIdx.y;
    float total = 0.0f;
    // one 64-pixel group per workgroup (default), or every group of the image in turn when the grid is
    // (1, N) — deterministic mode: out[n] then receives one sum with a fixed order
    for (int p0 = blockIdx.x * 64; p0 < HW; p0 += gridDim.x * 64) {
        const int p = p0 + px;
        const bool ok = p < HW;
        const float *a = f0 + (size_t)n * C * HW + (ok ? p : 0), *b = f1 + (size_t)n * C * HW + (ok ? p : 0);
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 4
        for (int c = sl; c < C; c += SL) {
            const float x = a[(size_t)c * HW], y = b[(size_t)c * HW];
            s0 += x * x;
            s1 += y * y;
        }
        s0 = slice_sum<SL>(s0, red, sl, px);
        s1 = slice_sum<SL>(s1, red, sl, px);
        const float ia = 1.0f / (sqrtf(s0) + LPIPS_EPS), ib = 1.0f / (sqrtf(s1) + LPIPS_EPS);
        float d = 0.0f;
#pragma unroll 4
        for (int c = sl; c < C; c += SL) {
            const float df = a[(size_t)c * HW] * ia - b[(size_t)c * HW] * ib;
            d += w[c] * df * df;
        }
        d = slice_sum<SL>(ok ? d : 0.0f, red, sl, px);
        if (sl == 0) {  // one wave: sum over the 64 pixels of this group
            for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
            total += d / (float)HW;
        }
    }
    if (sl == 0 && px == 0) unsafeAtomicAdd(out + n, total);
}

// g0[n,c,p] = gout[n]/HW * ( 2 w_c diff_c / a  -  f0_c / (a^2 n0) * sum_k 2 w_k diff_k f0_k )
// g_in (optional): a gradient that reaches f0 from elsewhere (the next VGG slice, through its max pool),
// added here; gate: f0 is a ReLU output — the sum passes only where f0 > 0 (the ReLU's own backward),
// so that the trunk's backward needs neither an add nor a gate launch at a tap.
template <int SL>
__global__ __launch_bounds__(64 * SL) void lpips_layer_bwd(const float *__restrict__ f0,
                                                           const float *__restrict__ f1,
                                                           const float *__restrict__ w,
                                                           const float *__restrict__ gout,
                                                           const float *__restrict__ g_in, const int gate,
                                                           float *__restrict__ g0, int C, int HW) {
    __shared__ float red[SL][64];
    const int px = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int n = blockIdx.y, p = blockIdx.x * 64 + px;
    const bool ok = p < HW;
    const float *a = f0 + (size_t)n * C * HW + (ok ? p : 0), *b = f1 + (size_t)n * C * HW + (ok ? p : 0);
    float *g = g0 + (size_t)n * C * HW + p;
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll 4
    for (int c = sl; c < C; c += SL) {
        const float x = a[(size_t)c * HW], y = b[(size_t)c * HW];
        s0 += x * x;
        s1 += y * y;
    }
    s0 = slice_sum<SL>(s0, red, sl, px);
    s1 = slice_sum<SL>(s1, red, sl, px);
    const float n0 = sqrtf(s0);
    const float ia = 1.0f / (n0 + LPIPS_EPS), ib = 1.0f / (sqrtf(s1) + LPIPS_EPS);
    float t = 0.0f;
#pragma unroll 4
    for (int c = sl; c < C; c += SL) {
        const float x = a[(size_t)c * HW];
        t += 2.0f * w[c] * (x * ia - b[(size_t)c * HW] * ib) * x;
    }
    t = slice_sum<SL>(t, red, sl, px);
    const float go = gout[n] / (float)HW;
    const float k2 = (n0 > 0.0f) ? t * ia * ia / n0 : 0.0f;
    if (!ok) return;
    const float *gi = g_in ? g_in + (size_t)n * C * HW + p : nullptr;
#pragma unroll 4
    for (int c = sl; c < C; c += SL) {
        const float x = a[(size_t)c * HW];
        float v = go * (2.0f * w[c] * (x * ia - b[(size_t)c * HW] * ib) * ia - x * k2);
        if (gi) v += gi[(size_t)c * HW];
        if (gate && !(x > 0.0f)) v = 0.0f;
        g[(size_t)c * HW] = v;
    }
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_lpips_layer_fwd(const float *f0, const float *f1, const float *w, float *out,
                                   int N, int C, int HW, g2s_stream_t stream) {
    G2S_REQUIRE(f0 && f1 && w && out && N > 0 && C > 0 && HW > 0, "bad argument");
    G2S_REQUIRE(N <= 65535, "N too large for grid.y");
    const int groups = deterministic() ? 1 : cdiv(HW, 64);   // deterministic: one workgroup walks the whole image
    if (HW <= 1024) lpips_layer_fwd<16><<<dim3(groups, N), 1024, 0, as_stream(stream)>>>(f0, f1, w, out, C, HW);
    else lpips_layer_fwd<4><<<dim3(groups, N), 256, 0, as_stream(stream)>>>(f0, f1, w, out, C, HW);
    return check_launch("g2s_lpips_layer_fwd");
}

extern "C" int g2s_lpips_layer_bwd_ex(const float *f0, const float *f1, const float *w, const float *gout,
                                      const float *g_in, int relu_gate, float *g0, int N, int C, int HW,
                                      g2s_stream_t stream) {
    G2S_REQUIRE(f0 && f1 && w && gout && g0 && N > 0 && C > 0 && HW > 0, "bad argument");
    G2S_REQUIRE(N <= 65535, "N too large for grid.y");
    if (HW <= 1024)
        lpips_layer_bwd<16><<<dim3(cdiv(HW, 64), N), 1024, 0, as_stream(stream)>>>(f0, f1, w, gout, g_in, relu_gate, g0, C, HW);
    else
        lpips_layer_bwd<4><<<dim3(cdiv(HW, 64), N), 256, 0, as_stream(stream)>>>(f0, f1, w, gout, g_in, relu_gate, g0, C, HW);
    return check_launch("g2s_lpips_layer_bwd");
}

extern "C" int g2s_lpips_layer_bwd(const float *f0, const float *f1, const float *w,
                                   const float *gout, float *g0, int N, int C, int HW,
                                   g2s_stream_t stream) {
    return g2s_lpips_layer_bwd_ex(f0, f1, w, gout, nullptr, 0, g0, N, C, HW, stream);
}
