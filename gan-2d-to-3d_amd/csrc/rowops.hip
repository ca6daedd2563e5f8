// rowops.hip — row-wise fused reductions that remove the elementwise/reduce launch chains around the
// modulated convolution (HBM-bound: every operand is read exactly once).
//
//   g2s_rows_dot_scale   gradients of  y = demod * conv(W, s * x)  that are reductions over H*W:
//                          dot[r]  = (sum_i a[r,i] * b[r,i]) * (inv ? 1 / inv[r] : 1)
//                          out[r,i] = b[r,i] * s[r]                       (optional, may alias b)
//                        used as  gs = sum_hw x * gxs ; gx = gxs * s     and   gd = sum_hw gy * y / demod
//   g2s_demod_fwd/_bwd   demod[b,o] = rsqrt(sum_i wsq[o,i] * s[b,i]^2 + eps)
//                        (ModulatedConv2d.forward, stylegan2-pytorch/model.py:254-258, with
//                        weight = scale * W * style  =>  sum_{i,t} weight^2 = sum_i wsq[o,i] s[b,i]^2)
#include <algorithm>
#include "g2s_common.h"

namespace g2s {

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wavefront per row, 4 rows per workgroup.  Rows of any length: a scalar head brings the row
// to 16-byte alignment (all operands share the row offset and have 16-byte aligned bases), then
// float4 body, scalar tail.
__global__ __launch_bounds__(256) void rows_dot_scale(const float *__restrict__ a,
                                                      const float *__restrict__ b,
                                                      const float *__restrict__ s,
                                                      const float *__restrict__ inv,
                                                      float *__restrict__ out,
                                                      float *__restrict__ dot, int rows, int n,
                                                      int aligned_bases) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t off = (size_t)row * n;
    const float *ar = a ? a + off : nullptr;
    const float *br = b + off;
    float *orow = out ? out + off : nullptr;
    const float sc = s ? s[row] : 1.0f;
    float acc = 0.0f;
    int head = aligned_bases ? (int)((4 - (off & 3)) & 3) : n;  // unaligned bases: all scalar
    if (head > n) head = n;
    const int n4 = (n - head) >> 2, tail0 = head + 4 * n4;
    for (int i = lane; i < head; i += 64) {
        const float bv = br[i];
        if (ar) acc += ar[i] * bv;
        if (orow) orow[i] = bv * sc;
    }
    for (int i = lane; i < n4; i += 64) {
        const float4 bv = reinterpret_cast<const float4 *>(br + head)[i];
        if (ar) {
            const float4 av = reinterpret_cast<const float4 *>(ar + head)[i];
            acc += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
        }
        if (orow)
            reinterpret_cast<float4 *>(orow + head)[i] = make_float4(bv.x * sc, bv.y * sc, bv.z * sc, bv.w * sc);
    }
    for (int i = tail0 + lane; i < n; i += 64) {
        const float bv = br[i];
        if (ar) acc += ar[i] * bv;
        if (orow) orow[i] = bv * sc;
    }
    if (dot) {
        acc = wave_sum(acc);
        if (lane == 0) dot[row] = inv ? acc / inv[row] : acc;
    }
}

// one wavefront per (b, o)
__global__ __launch_bounds__(256) void demod_fwd(const float *__restrict__ wsq,
                                                 const float *__restrict__ s,
                                                 float *__restrict__ demod, int B, int Cin, int Cout,
                                                 float eps) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= B * Cout) return;
    const int b = idx / Cout, o = idx % Cout;
    const float *w = wsq + (size_t)o * Cin, *sb = s + (size_t)b * Cin;
    float acc = 0.0f;
    for (int i = lane; i < Cin; i += 64) acc += w[i] * sb[i] * sb[i];
    acc = wave_sum(acc);
    if (lane == 0) demod[idx] = 1.0f / sqrtf(acc + eps);
}

// gs[b,i] = -s[b,i] * sum_o gd[b,o] * demod[b,o]^3 * wsq[o,i]
// grid (ceil(Cin / 64), B); 4 waves split the o range, lanes run along i (coalesced wsq rows).
__global__ __launch_bounds__(256) void demod_bwd(const float *__restrict__ wsq,
                                                 const float *__restrict__ s,
                                                 const float *__restrict__ demod,
                                                 const float *__restrict__ gd,
                                                 const float *gs_add,   // may alias gs (in-place add)
                                                 float *gs, int B, int Cin, int Cout) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, b = blockIdx.y;
    float acc = 0.0f;
    if (i < Cin) {
#pragma unroll 8
        for (int o = wave; o < Cout; o += 4) {
            const float d = demod[b * Cout + o];
            acc += gd[b * Cout + o] * d * d * d * wsq[(size_t)o * Cin + i];
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < Cin) {
        const float g = -s[b * Cin + i] * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        gs[b * Cin + i] = gs_add ? gs_add[b * Cin + i] + g : g;
    }
}

// One pass over the activation x = gain * leaky_relu(yconv + noise_w * noise + bias) that sits between two layers of
// the frozen generator, in the BACKWARD pass (synthesis.py): x is the output of the producer layer's StyledConv tail
// and the input of its consumers — the next modulated convolution (g1 = gradient w.r.t. s1 * x from its data-gradient
// GEMM) and, behind every second layer, ToRGB (g2, s2).  Per row r = (b, c), n = H * W:
//     dot1[r] = sum_i x * g1          dot2[r] = sum_i x * g2          (the consumers' style gradients)
//     out[r,i] = (g1 * s1[r] + g2 * s2[r]) * gain * (x > 0 ? 1 : slope)   (gradient w.r.t. the producer's
//                                                                          pre-activation: FusedLeakyReLU's gate)
//     gdot[r] = sum_i out * yconv / demod[r],  yconv = (x > 0 ? x : x / slope) / gain - noise_w * noise[i] - bias[c]
//               (d loss / d demod of the producer; its convolution output is recovered from x: the leaky ReLU is
//               invertible, so the forward never stores the pre-activation)
// replacing four passes (rows_dot_scale x 2, the accumulation of the two consumers' gradients, the gate) by one.
// One wavefront per row; every operand is read once.
__global__ __launch_bounds__(256) void synth_rows(const float *__restrict__ x, const float *__restrict__ g1,
                                                  const float *__restrict__ s1, const float *__restrict__ g2,
                                                  const float *__restrict__ s2, const float *__restrict__ noise,
                                                  const float *__restrict__ noise_w, const float *__restrict__ bias,
                                                  const float *__restrict__ demod, float *__restrict__ out,
                                                  float *__restrict__ dot1, float *__restrict__ dot2,
                                                  float *__restrict__ gdot, int rows, int channels, int n, float slope,
                                                  float gain, int vec) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t off = (size_t)row * n;
    const float *xr = x + off, *g1r = g1 + off, *g2r = g2 ? g2 + off : nullptr;
    float *orow = out ? out + off : nullptr;
    const float a1 = s1[row], a2 = g2 ? s2[row] : 0.0f;
    const float nw = gdot ? noise_w[0] : 0.0f, bi = gdot ? bias[row % channels] : 0.0f;
    const float inv_gain = 1.0f / gain, inv_slope = 1.0f / slope;
    float d1 = 0.0f, d2 = 0.0f, dg = 0.0f;
    auto one = [&](float xv, float gv1, float gv2, float nz) {
        d1 += xv * gv1;
        d2 += xv * gv2;
        const float o = (gv1 * a1 + gv2 * a2) * (xv > 0.0f ? gain : gain * slope);
        if (gdot) dg += o * ((xv > 0.0f ? xv : xv * inv_slope) * inv_gain - nw * nz - bi);
        return o;
    };
    if (vec) {
        for (int i = lane; i < (n >> 2); i += 64) {
            const float4 xv = reinterpret_cast<const float4 *>(xr)[i];
            const float4 v1 = reinterpret_cast<const float4 *>(g1r)[i];
            const float4 v2 = g2r ? reinterpret_cast<const float4 *>(g2r)[i] : float4{0.f, 0.f, 0.f, 0.f};
            const float4 nz = gdot ? reinterpret_cast<const float4 *>(noise)[i] : float4{0.f, 0.f, 0.f, 0.f};
            float4 o;
            o.x = one(xv.x, v1.x, v2.x, nz.x);
            o.y = one(xv.y, v1.y, v2.y, nz.y);
            o.z = one(xv.z, v1.z, v2.z, nz.z);
            o.w = one(xv.w, v1.w, v2.w, nz.w);
            if (orow) reinterpret_cast<float4 *>(orow)[i] = o;
        }
    } else {
        for (int i = lane; i < n; i += 64) {
            const float o = one(xr[i], g1r[i], g2r ? g2r[i] : 0.0f, gdot ? noise[i] : 0.0f);
            if (orow) orow[i] = o;
        }
    }
    d1 = wave_sum(d1);
    d2 = wave_sum(d2);
    dg = wave_sum(dg);
    if (lane == 0) {
        dot1[row] = d1;
        if (dot2) dot2[row] = d2;
        if (gdot) gdot[row] = dg / demod[row];
    }
}

// out[c] = sum_{b, i} g[b, c, i]: the bias gradient of a convolution (nn.Conv2d(bias=True) of the offset encoder's
// residual blocks, GAN2Shape/networks.py:170-244) — torch's generic reduction over dims (0, 2, 3) takes 15-23 us on
// these few-KB tensors.  One workgroup per channel, fixed summation order.
__global__ __launch_bounds__(256) void channel_sum(const float *__restrict__ g, float *__restrict__ out, int B, int C,
                                                   int n) {
    __shared__ float red[4];
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.0f;
    const int total = B * n;
    for (int i = threadIdx.x; i < total; i += 256) acc += g[((size_t)(i / n) * C + c) * n + i % n];
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = red[0] + red[1] + red[2] + red[3];
}

// The demodulations of ALL styled layers of the frozen generator in one launch each way (synthesis.py): the
// styles of every layer are known before the first convolution, and every layer's two style-gradient inputs
// (the convolution path's sum x * g and d loss / d demod) are complete when the backward pass ends.
struct DemodMulti {
    const float *wsq[G2S_DEMOD_MAX_LAYERS], *s[G2S_DEMOD_MAX_LAYERS], *gd[G2S_DEMOD_MAX_LAYERS];
    float *demod[G2S_DEMOD_MAX_LAYERS], *gs[G2S_DEMOD_MAX_LAYERS];
    int Cin[G2S_DEMOD_MAX_LAYERS], Cout[G2S_DEMOD_MAX_LAYERS];
    int layers, B;
    float eps;
};

__global__ __launch_bounds__(256) void demod_fwd_multi(DemodMulti d) {
    const int l = blockIdx.y;
    const int Cin = d.Cin[l], Cout = d.Cout[l];
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= d.B * Cout) return;
    const int b = idx / Cout, o = idx % Cout;
    const float *w = d.wsq[l] + (size_t)o * Cin, *sb = d.s[l] + (size_t)b * Cin;
    float acc = 0.0f;
    for (int i = lane; i < Cin; i += 64) acc += w[i] * sb[i] * sb[i];
    acc = wave_sum(acc);
    if (lane == 0) d.demod[l][idx] = 1.0f / sqrtf(acc + d.eps);
}

// grid (ceil(max Cin / 64), B, layers): demod_bwd of every layer, added in place to gs (the convolution path's sum)
__global__ __launch_bounds__(256) void demod_bwd_multi(DemodMulti d) {
    __shared__ float red[4][64];
    const int l = blockIdx.z;
    const int Cin = d.Cin[l], Cout = d.Cout[l];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, b = blockIdx.y;
    if (blockIdx.x * 64 >= Cin) return;      // whole workgroup: no barrier is skipped by part of it
    const float *wsq = d.wsq[l], *demod = d.demod[l], *gd = d.gd[l];
    float acc = 0.0f;
    if (i < Cin) {
#pragma unroll 8
        for (int o = wave; o < Cout; o += 4) {
            const float dm = demod[b * Cout + o];
            acc += gd[b * Cout + o] * dm * dm * dm * wsq[(size_t)o * Cin + i];
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < Cin) {
        const float g = -d.s[l][b * Cin + i] * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        d.gs[l][b * Cin + i] += g;
    }
}

}  // namespace g2s

using namespace g2s;

static int demod_multi_fill(DemodMulti &d, const void *const *wsq, const void *const *s, const void *const *demod,
                            const void *const *gd, const void *const *gs, const int *Cin, const int *Cout, int layers,
                            int B, int &max_cin, int &max_cout) {
    G2S_REQUIRE(layers > 0 && layers <= G2S_DEMOD_MAX_LAYERS && B > 0, "1..%d layers, B positive", G2S_DEMOD_MAX_LAYERS);
    max_cin = max_cout = 0;
    for (int l = 0; l < layers; l++) {
        G2S_REQUIRE(wsq[l] && s[l] && demod[l] && Cin[l] > 0 && Cout[l] > 0, "layer %d: NULL pointer or empty", l);
        d.wsq[l] = (const float *)wsq[l];
        d.s[l] = (const float *)s[l];
        d.demod[l] = (float *)demod[l];
        d.gd[l] = gd ? (const float *)gd[l] : nullptr;
        d.gs[l] = gs ? (float *)gs[l] : nullptr;
        d.Cin[l] = Cin[l];
        d.Cout[l] = Cout[l];
        max_cin = std::max(max_cin, Cin[l]);
        max_cout = std::max(max_cout, Cout[l]);
    }
    d.layers = layers;
    d.B = B;
    return G2S_OK;
}

extern "C" int g2s_demod_fwd_multi(const void *const *wsq, const void *const *s, const void *const *demod, const int *Cin,
                                   const int *Cout, int layers, int B, float eps, g2s_stream_t stream) {
    DemodMulti d{};
    int mi, mo;
    const int rc = demod_multi_fill(d, wsq, s, demod, nullptr, nullptr, Cin, Cout, layers, B, mi, mo);
    if (rc != G2S_OK) return rc;
    d.eps = eps;
    demod_fwd_multi<<<dim3(cdiv((long)B * mo, 4), layers), 256, 0, as_stream(stream)>>>(d);
    return check_launch("g2s_demod_fwd_multi");
}

extern "C" int g2s_demod_bwd_multi(const void *const *wsq, const void *const *s, const void *const *demod,
                                   const void *const *gd, const void *const *gs, const int *Cin, const int *Cout,
                                   int layers, int B, g2s_stream_t stream) {
    G2S_REQUIRE(gd && gs, "gd, gs must not be NULL");
    DemodMulti d{};
    int mi, mo;
    const int rc = demod_multi_fill(d, wsq, s, demod, gd, gs, Cin, Cout, layers, B, mi, mo);
    if (rc != G2S_OK) return rc;
    for (int l = 0; l < layers; l++) G2S_REQUIRE(gd[l] && gs[l], "layer %d: gd / gs NULL", l);
    demod_bwd_multi<<<dim3(cdiv(mi, 64), B, layers), 256, 0, as_stream(stream)>>>(d);
    return check_launch("g2s_demod_bwd_multi");
}

extern "C" int g2s_channel_sum(const float *g, float *out, int B, int C, int n, g2s_stream_t stream) {
    G2S_REQUIRE(g && out && B > 0 && C > 0 && n > 0 && (long)B * n < (1l << 31), "bad argument");
    channel_sum<<<C, 256, 0, as_stream(stream)>>>(g, out, B, C, n);
    return check_launch("g2s_channel_sum");
}

extern "C" int g2s_synth_bwd_rows(const float *x, const float *g1, const float *s1, const float *g2, const float *s2,
                                  const float *noise, const float *noise_w, const float *bias, const float *demod,
                                  float *out, float *dot1, float *dot2, float *gdot, int rows, int channels, int n,
                                  float slope, float gain, g2s_stream_t stream) {
    G2S_REQUIRE(x && g1 && s1 && dot1 && rows > 0 && channels > 0 && n > 0, "x, g1, s1, dot1 must not be NULL; sizes positive");
    G2S_REQUIRE((g2 == nullptr) == (s2 == nullptr) && (g2 == nullptr) == (dot2 == nullptr), "g2, s2, dot2 come together");
    G2S_REQUIRE(!gdot || (noise && noise_w && bias && demod), "gdot needs noise, noise_w, bias, demod");
    G2S_REQUIRE(slope > 0.0f && gain > 0.0f, "slope and gain must be positive (the activation is inverted)");
    uintptr_t bits = (uintptr_t)x | (uintptr_t)g1 | (uintptr_t)g2 | (uintptr_t)out | (gdot ? (uintptr_t)noise : 0);
    const int vec = (n % 4 == 0) && (bits & 15) == 0;
    synth_rows<<<cdiv(rows, 4), 256, 0, as_stream(stream)>>>(x, g1, s1, g2, s2, noise, noise_w, bias, demod, out, dot1,
                                                               dot2, gdot, rows, channels, n, slope, gain, vec);
    return check_launch("g2s_synth_bwd_rows");
}

extern "C" int g2s_rows_dot_scale(const float *a, const float *b, const float *s, const float *inv,
                                  float *out, float *dot, int rows, int n, g2s_stream_t stream) {
    G2S_REQUIRE(b != nullptr && rows > 0 && n > 0, "b must not be NULL; rows, n positive");
    G2S_REQUIRE(out || dot, "nothing to compute");
    G2S_REQUIRE(!dot || a, "dot needs a");
    const int aligned = !((uintptr_t)b & 15) && (!a || !((uintptr_t)a & 15)) && (!out || !((uintptr_t)out & 15));
    hipStream_t st = as_stream(stream);
    rows_dot_scale<<<cdiv(rows, 4), 256, 0, st>>>(a, b, s, inv, out, dot, rows, n, aligned);
    return check_launch("g2s_rows_dot_scale");
}

extern "C" int g2s_demod_fwd(const float *wsq, const float *s, float *demod, int B, int Cin, int Cout,
                             float eps, g2s_stream_t stream) {
    G2S_REQUIRE(wsq && s && demod && B > 0 && Cin > 0 && Cout > 0, "bad argument");
    demod_fwd<<<cdiv((long)B * Cout, 4), 256, 0, as_stream(stream)>>>(wsq, s, demod, B, Cin, Cout, eps);
    return check_launch("g2s_demod_fwd");
}

extern "C" int g2s_demod_bwd(const float *wsq, const float *s, const float *demod, const float *gd,
                             float *gs, int B, int Cin, int Cout, g2s_stream_t stream) {
    G2S_REQUIRE(wsq && s && demod && gd && gs && B > 0 && Cin > 0 && Cout > 0, "bad argument");
    demod_bwd<<<dim3(cdiv(Cin, 64), B), 256, 0, as_stream(stream)>>>(wsq, s, demod, gd, nullptr, gs, B, Cin, Cout);
    return check_launch("g2s_demod_bwd");
}

extern "C" int g2s_demod_bwd_add(const float *wsq, const float *s, const float *demod, const float *gd,
                                 const float *gs_add, float *gs, int B, int Cin, int Cout, g2s_stream_t stream) {
    G2S_REQUIRE(wsq && s && demod && gd && gs && B > 0 && Cin > 0 && Cout > 0, "bad argument");
    demod_bwd<<<dim3(cdiv(Cin, 64), B), 256, 0, as_stream(stream)>>>(wsq, s, demod, gd, gs_add, gs, B, Cin, Cout);
    return check_launch("g2s_demod_bwd_add");
}
