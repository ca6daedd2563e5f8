// split_reduce.hip — see split_reduce.h.  HBM bound: reads `slices` copies, writes one.
#include <algorithm>
#include "split_reduce.h"

namespace g2s {

__global__ __launch_bounds__(256) void split_reduce_kernel(const float *part, int slices, int64_t n, float *y,
                                                           const float *bias, int64_t hw, int channels, int act,
                                                           float alpha, float gain, const float *noise,
                                                           const float *noise_w) {
    const float nw = noise ? noise_w[0] : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    const bool vec = (n & 3) == 0 && ((reinterpret_cast<uintptr_t>(part) | reinterpret_cast<uintptr_t>(y)) & 15) == 0;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        const int cnt = (int)min((int64_t)4, n - i);
        if (vec) {
            float4 v[SPLIT_REDUCE_MAX];
#pragma unroll
            for (int s = 0; s < SPLIT_REDUCE_MAX; s++)
                if (s < slices) v[s] = *reinterpret_cast<const float4 *>(part + (int64_t)s * n + i);
#pragma unroll
            for (int s = 0; s < SPLIT_REDUCE_MAX; s++)
                if (s < slices) {
                    r[0] += v[s].x;
                    r[1] += v[s].y;
                    r[2] += v[s].z;
                    r[3] += v[s].w;
                }
        } else {
            for (int s = 0; s < slices; s++)
                for (int j = 0; j < cnt; j++) r[j] += part[(int64_t)s * n + i + j];
        }
        for (int j = 0; j < cnt; j++) {
            float v = r[j];
            if (bias) v += bias[((i + j) / hw) % channels];
            if (noise) v += nw * noise[(i + j) % hw];
            if (act) v = (v > 0.0f ? v : v * alpha) * gain;
            r[j] = v;
        }
        if (vec) *reinterpret_cast<float4 *>(y + i) = float4{r[0], r[1], r[2], r[3]};
        else
            for (int j = 0; j < cnt; j++) y[i + j] = r[j];
    }
}

int split_reduce_launch(const float *part, int slices, int64_t n, float *y, const float *bias, int64_t hw,
                        int channels, int act, float alpha, float gain, g2s_stream_t stream, const float *noise,
                        const float *noise_w) {
    G2S_REQUIRE(slices >= 1 && slices <= SPLIT_REDUCE_MAX, "split reduce: 1..%d slices", SPLIT_REDUCE_MAX);
    const int64_t quads = (n + 3) / 4;
    const int blocks = (int)std::min<int64_t>((quads + 255) / 256, 256 * 8);
    split_reduce_kernel<<<blocks, 256, 0, as_stream(stream)>>>(part, slices, n, y, bias, hw, channels, act, alpha, gain, noise, noise_w);
    return check_launch("split reduce");
}

}  // namespace g2s
