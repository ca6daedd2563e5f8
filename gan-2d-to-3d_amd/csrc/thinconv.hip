// thinconv.hip — 1x1 convolution from at most 4 input channels over MANY pixels: the discriminator's
// fromRGB (ConvLayer(3, C, 1): 1x1 + bias + leaky-ReLU, stylegan2-pytorch/model.py:709).  On an MFMA
// tile its K = 3 is padded to 16 and the launch costs 36 us at the workload's size; it is a streaming
// problem — read 3 planes, write C planes once (write bound): 22 us here.
//
//   y[b,m,p] = act(bias[m] + sum_{c<Cr} W(m,c) x[b,c,p])
//
// Lane = 4 consecutive pixels; the 4 waves of a workgroup share 64 pixel quads and split the output
// channels.  Only used for >= 64 K pixels.  (Measured and NOT adopted: the same scheme for 3x3 kernels —
// VGG conv1_1 3 -> 64: 28.5 vs 27.1 us on the MFMA tile — and for <= 4 OUTPUT channels — conv1_1's
// data-gradient 64 -> 3: 51.5 vs 54.7 us; ToRGB's data-gradient wins above 64^2 only.)
// fp32 FMA chains: results differ from the MFMA path by summation order only.
#include "thinconv.h"

namespace g2s {

constexpr int TC_THREADS = 256;

struct ThinDesc {
    const float *x, *w, *bias;
    float *y;
    int B, Cr, M, H, W;
    int act;
    float alpha, gain;
};

__global__ __launch_bounds__(TC_THREADS) void thin_in_kernel(ThinDesc d) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int HW4 = (d.H * d.W) >> 2;
    const int q = blockIdx.x * 64 + lane, b = blockIdx.y;
    if (q >= HW4) return;
    float4 xin[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        xin[c] = c < d.Cr ? reinterpret_cast<const float4 *>(d.x)[((size_t)b * d.Cr + c) * HW4 + q] : float4{0.f, 0.f, 0.f, 0.f};
    const int per = (d.M + 3) / 4, m0 = wave * per, m1 = min(d.M, m0 + per);
    float4 *yb = reinterpret_cast<float4 *>(d.y) + (size_t)b * d.M * HW4 + q;
    for (int m = m0; m < m1; m++) {
        const float bi = d.bias ? d.bias[m] : 0.0f;
        float4 r{bi, bi, bi, bi};
#pragma unroll
        for (int c = 0; c < 4; c++) {
            if (c < d.Cr) {
                const float wv = d.w[m * d.Cr + c];
                r.x += wv * xin[c].x;
                r.y += wv * xin[c].y;
                r.z += wv * xin[c].z;
                r.w += wv * xin[c].w;
            }
        }
        if (d.act) {
            r.x = (r.x > 0.f ? r.x : r.x * d.alpha) * d.gain;
            r.y = (r.y > 0.f ? r.y : r.y * d.alpha) * d.gain;
            r.z = (r.z > 0.f ? r.z : r.z * d.alpha) * d.gain;
            r.w = (r.w > 0.f ? r.w : r.w * d.alpha) * d.gain;
        }
        yb[(size_t)m * HW4] = r;
    }
}

bool thin_conv_eligible(int B, int Cr, int M, int H, int W, int k, int transpose) {
    return k == 1 && !transpose && Cr <= 4 && M >= 16 && (H * W) % 4 == 0 && (long)B * H * W >= (1l << 16) && B <= 65535;
}

int thin_conv_launch(const float *x, const float *w, const float *bias, float *y, int B, int Cr, int M, int H, int W,
                     int act, float alpha, float gain, g2s_stream_t stream) {
    G2S_REQUIRE(x && w && y, "x, w, y must not be NULL");
    G2S_REQUIRE((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) % 16 == 0,
                "x and y must be 16-byte aligned");
    ThinDesc d{};
    d.x = x;
    d.w = w;       // [M, Cr, 1, 1]
    d.bias = bias;
    d.y = y;
    d.B = B;
    d.Cr = Cr;
    d.M = M;
    d.H = H;
    d.W = W;
    d.act = act;
    d.alpha = alpha;
    d.gain = gain;
    thin_in_kernel<<<dim3(cdiv((long)H * W / 4, 64), B), TC_THREADS, 0, as_stream(stream)>>>(d);
    return check_launch("g2s conv (thin 1x1)");
}

}  // namespace g2s
