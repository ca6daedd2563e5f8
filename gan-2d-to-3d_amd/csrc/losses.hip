// losses.hip — masked / weighted L1 between two feature tensors in one pass per direction:
//     num = sum_{b,c,p} |x[b,c,p] - y[b,c,p]| * w[b,p]        (w == NULL: weight 1)
// the numerator of PhotometricLoss and of every level of DiscriminatorLoss
// (GAN2Shape/losses.py:6-51: loss = (|a - b| * mask.expand_as).sum() / mask.expand_as.sum(), which the
// reference runs as sub, abs, expand, mul, two full-size sums and a division per level).  The
// denominator C * sum(w) needs only the small weight map and stays on the host side (torch).
// Backward: gx = sign(x - y) * w * coef[0]  (coef = incoming gradient / denominator, a device scalar).
#include <algorithm>
#include "g2s_common.h"

namespace g2s {

constexpr int WL1_THREADS = 256;

__device__ __forceinline__ float wl1_block_sum(float v, float *sm) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    return (threadIdx.x == 0) ? (sm[0] + sm[1]) + (sm[2] + sm[3]) : 0.0f;
}

// x, y [B, C, HW]; w [B, HW] or NULL.  grid (chunks of 4 * 256 * UNROLL floats per (b, c) plane..)
__global__ __launch_bounds__(WL1_THREADS) void wl1_fwd(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ w, float *__restrict__ num,
                                                       int C, int HW4, long total4) {
    __shared__ float sm[4];
    float acc = 0.0f;
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    for (long i = (long)blockIdx.x * WL1_THREADS + threadIdx.x; i < total4; i += (long)gridDim.x * WL1_THREADS) {
        const float4 a = x4[i], b = y4[i];
        float4 m = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (w) {
            const long plane = i / HW4;              // b * C + c
            m = w4[(plane / C) * HW4 + (i - plane * HW4)];
        }
        acc += (fabsf(a.x - b.x) * m.x + fabsf(a.y - b.y) * m.y) + (fabsf(a.z - b.z) * m.z + fabsf(a.w - b.w) * m.w);
    }
    const float s = wl1_block_sum(acc, sm);
    if (threadIdx.x == 0) unsafeAtomicAdd(num, s);
}

__global__ __launch_bounds__(WL1_THREADS) void wl1_bwd(const float *__restrict__ x, const float *__restrict__ y,
                                                       const float *__restrict__ w, const float *__restrict__ coef,
                                                       float *__restrict__ gx, int C, int HW4, long total4) {
    const float k = coef[0];
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    float4 *g4 = reinterpret_cast<float4 *>(gx);
    for (long i = (long)blockIdx.x * WL1_THREADS + threadIdx.x; i < total4; i += (long)gridDim.x * WL1_THREADS) {
        const float4 a = x4[i], b = y4[i];
        float4 m = make_float4(k, k, k, k);
        if (w) {
            const long plane = i / HW4;
            const float4 ww = w4[(plane / C) * HW4 + (i - plane * HW4)];
            m = make_float4(ww.x * k, ww.y * k, ww.z * k, ww.w * k);
        }
        auto sgn = [](float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); };   // torch: sign(0) = 0
        g4[i] = make_float4(sgn(a.x - b.x) * m.x, sgn(a.y - b.y) * m.y, sgn(a.z - b.z) * m.z, sgn(a.w - b.w) * m.w);
    }
}

// The same with the denominator in the pass: numden[0] += sum |x - y| * w, numden[1] += sum w (over b, c, p — the
// reference's mask.expand_as(err).sum(); w == NULL: the element count), so that the loss is one division away.
// THREADS = 256 in the default launch (many workgroups, float atomics); 1024 for the single workgroup of
// deterministic mode (one fixed order; four times the loads in flight of a 256-thread one).
template <int THREADS>
__global__ __launch_bounds__(THREADS) void wl1_fwd2(const float *__restrict__ x, const float *__restrict__ y,
                                                    const float *__restrict__ w, float *__restrict__ numden,
                                                    int C, int HW4, long total4) {
    __shared__ float sm[2][THREADS / 64];
    float acc = 0.0f, wsum = 0.0f;
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *w4 = reinterpret_cast<const float4 *>(w);
    for (long i = (long)blockIdx.x * THREADS + threadIdx.x; i < total4; i += (long)gridDim.x * THREADS) {
        const float4 a = x4[i], b = y4[i];
        float4 m = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        if (w) {
            const long plane = i / HW4;              // b * C + c
            m = w4[(plane / C) * HW4 + (i - plane * HW4)];
        }
        acc += (fabsf(a.x - b.x) * m.x + fabsf(a.y - b.y) * m.y) + (fabsf(a.z - b.z) * m.z + fabsf(a.w - b.w) * m.w);
        wsum += (m.x + m.y) + (m.z + m.w);
    }
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_down(acc, o, 64);
        wsum += __shfl_down(wsum, o, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) {
        sm[0][wave] = acc;
        sm[1][wave] = wsum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f, d = 0.0f;
        for (int k = 0; k < THREADS / 64; k++) {   // fixed order
            s += sm[0][k];
            d += sm[1][k];
        }
        unsafeAtomicAdd(numden, s);
        unsafeAtomicAdd(numden + 1, d);
    }
}

// gx = gadd + sign(x - y) * w * g[0] / den[0]   (gadd may be NULL)
__global__ __launch_bounds__(WL1_THREADS) void wl1_bwd2(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ w, const float *__restrict__ g,
                                                        const float *__restrict__ den, const float *__restrict__ gadd,
                                                        float *__restrict__ gx, int C, int HW4, long total4) {
    const float k = g[0] / den[0];
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *w4 = reinterpret_cast<const float4 *>(w), *a4 = reinterpret_cast<const float4 *>(gadd);
    float4 *g4 = reinterpret_cast<float4 *>(gx);
    for (long i = (long)blockIdx.x * WL1_THREADS + threadIdx.x; i < total4; i += (long)gridDim.x * WL1_THREADS) {
        const float4 a = x4[i], b = y4[i];
        float4 m = make_float4(k, k, k, k);
        if (w) {
            const long plane = i / HW4;
            const float4 ww = w4[(plane / C) * HW4 + (i - plane * HW4)];
            m = make_float4(ww.x * k, ww.y * k, ww.z * k, ww.w * k);
        }
        auto sgn = [](float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); };   // torch: sign(0) = 0
        float4 r = make_float4(sgn(a.x - b.x) * m.x, sgn(a.y - b.y) * m.y, sgn(a.z - b.z) * m.z, sgn(a.w - b.w) * m.w);
        if (gadd) {
            const float4 e = a4[i];
            r = make_float4(e.x + r.x, e.y + r.y, e.z + r.z, e.w + r.w);
        }
        g4[i] = r;
    }
}

// One launch for a level of the discriminator-feature loss's backward (losses._DFeatureL1):
//     gx      = (gadd + gadd2) * add_scale + sign(x - y) * w * g[0] / den[0]
//               (the residual join (a + b) / sqrt(2) of the block above, stylegan2-pytorch/model.py:693-697, with this
//                level's masked-L1 gradient; x == NULL: no L1 term; gadd / gadd2 may be NULL)
//     gx_gate = gx * gain * (gate_ref > 0 ? 1 : slope)      (FusedLeakyReLU's backward of the block's conv2, optional)
__global__ __launch_bounds__(WL1_THREADS) void wl1_bwd3(const float *__restrict__ x, const float *__restrict__ y,
                                                        const float *__restrict__ w, const float *__restrict__ g,
                                                        const float *__restrict__ den, const float *__restrict__ gadd,
                                                        const float *__restrict__ gadd2, float add_scale,
                                                        float *__restrict__ gx, const float *__restrict__ gate_ref,
                                                        float slope, float gain, float *__restrict__ gx_gate, int C,
                                                        int HW4, long total4) {
    const float k = x ? g[0] / den[0] : 0.0f;
    const float4 *x4 = reinterpret_cast<const float4 *>(x), *y4 = reinterpret_cast<const float4 *>(y);
    const float4 *w4 = reinterpret_cast<const float4 *>(w), *a4 = reinterpret_cast<const float4 *>(gadd);
    const float4 *b4 = reinterpret_cast<const float4 *>(gadd2), *r4 = reinterpret_cast<const float4 *>(gate_ref);
    float4 *g4 = reinterpret_cast<float4 *>(gx), *q4 = reinterpret_cast<float4 *>(gx_gate);
    for (long i = (long)blockIdx.x * WL1_THREADS + threadIdx.x; i < total4; i += (long)gridDim.x * WL1_THREADS) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        if (x) {
            const float4 a = x4[i], b = y4[i];
            float4 m = make_float4(k, k, k, k);
            if (w) {
                const long plane = i / HW4;
                const float4 ww = w4[(plane / C) * HW4 + (i - plane * HW4)];
                m = make_float4(ww.x * k, ww.y * k, ww.z * k, ww.w * k);
            }
            auto sgn = [](float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); };
            r = make_float4(sgn(a.x - b.x) * m.x, sgn(a.y - b.y) * m.y, sgn(a.z - b.z) * m.z, sgn(a.w - b.w) * m.w);
        }
        if (gadd) {
            float4 e = a4[i];
            if (gadd2) {
                const float4 f = b4[i];
                e = make_float4(e.x + f.x, e.y + f.y, e.z + f.z, e.w + f.w);
            }
            r = make_float4(e.x * add_scale + r.x, e.y * add_scale + r.y, e.z * add_scale + r.z, e.w * add_scale + r.w);
        }
        if (gx) g4[i] = r;
        if (gx_gate) {
            const float4 f = r4[i];
            const float lo = gain * slope;
            q4[i] = make_float4(r.x * (f.x > 0.0f ? gain : lo), r.y * (f.y > 0.0f ? gain : lo),
                                r.z * (f.z > 0.0f ? gain : lo), r.w * (f.w > 0.0f ? gain : lo));
        }
    }
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_weighted_l1_bwd3(const float *x, const float *y, const float *w, const float *g, const float *den,
                                    const float *gadd, const float *gadd2, float add_scale, float *gx,
                                    const float *gate_ref, float slope, float gain, float *gx_gate, int B, int C, int HW,
                                    g2s_stream_t stream) {
    G2S_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0, "sizes must be positive, HW a multiple of 4");
    G2S_REQUIRE(!x || (y && g && den), "the L1 term needs x, y, g and den");
    G2S_REQUIRE(x || gadd, "nothing to compute");
    G2S_REQUIRE(!gadd2 || gadd, "gadd2 needs gadd");
    G2S_REQUIRE(gx || gx_gate, "no output");
    G2S_REQUIRE((gx_gate == nullptr) == (gate_ref == nullptr), "gx_gate and gate_ref come together");
    const long total4 = (long)B * C * HW / 4;
    const int blocks = (int)std::min<long>(cdiv(total4, WL1_THREADS * 4), 4096);
    wl1_bwd3<<<blocks, WL1_THREADS, 0, as_stream(stream)>>>(x, y, w, g, den, gadd, gadd2, add_scale, gx, gate_ref, slope,
                                                            gain, gx_gate, C, HW / 4, total4);
    return check_launch("g2s_weighted_l1_bwd3");
}

static int wl1_check(const void *x, const void *y, int B, int C, int HW) {
    G2S_REQUIRE(x && y, "NULL pointer argument");
    G2S_REQUIRE(B > 0 && C > 0 && HW > 0 && HW % 4 == 0, "sizes must be positive, HW a multiple of 4");
    return G2S_OK;
}

extern "C" int g2s_weighted_l1_fwd(const float *x, const float *y, const float *w, float *num, int B, int C,
                                   int HW, g2s_stream_t stream) {
    int rc = wl1_check(x, y, B, C, HW);
    if (rc) return rc;
    G2S_REQUIRE(num, "num must not be NULL (a ZEROED device float: the sum is accumulated into it)");
    const long total4 = (long)B * C * HW / 4;
    // deterministic mode: one workgroup strides over everything, so the sum has one fixed order
    const int blocks = deterministic() ? 1 : (int)std::min<long>(cdiv(total4, WL1_THREADS * 4), 2048);
    wl1_fwd<<<blocks, WL1_THREADS, 0, as_stream(stream)>>>(x, y, w, num, C, HW / 4, total4);
    return check_launch("g2s_weighted_l1_fwd");
}

extern "C" int g2s_weighted_l1_bwd(const float *x, const float *y, const float *w, const float *coef, float *gx,
                                   int B, int C, int HW, g2s_stream_t stream) {
    int rc = wl1_check(x, y, B, C, HW);
    if (rc) return rc;
    G2S_REQUIRE(coef && gx, "NULL pointer argument");
    const long total4 = (long)B * C * HW / 4;
    const int blocks = (int)std::min<long>(cdiv(total4, WL1_THREADS * 4), 4096);
    wl1_bwd<<<blocks, WL1_THREADS, 0, as_stream(stream)>>>(x, y, w, coef, gx, C, HW / 4, total4);
    return check_launch("g2s_weighted_l1_bwd");
}

extern "C" int g2s_weighted_l1_fwd2(const float *x, const float *y, const float *w, float *numden, int B, int C,
                                    int HW, g2s_stream_t stream) {
    int rc = wl1_check(x, y, B, C, HW);
    if (rc) return rc;
    G2S_REQUIRE(numden, "numden must not be NULL (two ZEROED device floats: numerator, denominator)");
    const long total4 = (long)B * C * HW / 4;
    if (deterministic()) {   // one workgroup strides over everything: the sums have one fixed order
        wl1_fwd2<1024><<<1, 1024, 0, as_stream(stream)>>>(x, y, w, numden, C, HW / 4, total4);
    } else {
        const int blocks = (int)std::min<long>(cdiv(total4, WL1_THREADS * 4), 2048);
        wl1_fwd2<WL1_THREADS><<<blocks, WL1_THREADS, 0, as_stream(stream)>>>(x, y, w, numden, C, HW / 4, total4);
    }
    return check_launch("g2s_weighted_l1_fwd2");
}

extern "C" int g2s_weighted_l1_bwd2(const float *x, const float *y, const float *w, const float *g, const float *den,
                                    const float *gadd, float *gx, int B, int C, int HW, g2s_stream_t stream) {
    int rc = wl1_check(x, y, B, C, HW);
    if (rc) return rc;
    G2S_REQUIRE(g && den && gx, "NULL pointer argument");
    const long total4 = (long)B * C * HW / 4;
    const int blocks = (int)std::min<long>(cdiv(total4, WL1_THREADS * 4), 4096);
    wl1_bwd2<<<blocks, WL1_THREADS, 0, as_stream(stream)>>>(x, y, w, g, den, gadd, gx, C, HW / 4, total4);
    return check_launch("g2s_weighted_l1_bwd2");
}
