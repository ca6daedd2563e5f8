// pool.hip — nn.MaxPool2d(kernel_size=2, stride=2) of the VGG16 trunk (lpips/pretrained_networks.py:
// 97-135 via torchvision's vgg16.features), forward and backward, as two streaming kernels.
// torch's backward clears the whole input gradient (one fill per pool) and then scatters through
// saved int64 indices; here the backward re-derives the winner of each 2x2 window from the saved
// input — first maximum in row-major order, NaN wins, exactly torch's forward rule — and WRITES all
// four input gradients of the window: no fill, no index tensor, 16-byte accesses.
// Lane = 4 consecutive output pixels (8 input pixels of two rows).  H, W even, W % 8 == 0.
#include "g2s_common.h"

namespace g2s {

__device__ __forceinline__ bool beats(float v, float best) { return v > best || v != v; }

__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const float *x, float *y, long quads, int OH, int OW4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % OW4);
        const long r = i / OW4;                 // (plane, output row)
        const long plane = r / OH;
        const int oy = (int)(r - plane * OH);
        const int W = OW4 * 8;
        const float4 *top = reinterpret_cast<const float4 *>(x + (plane * 2 * OH + 2 * oy) * W + 8 * q);
        const float4 *bot = reinterpret_cast<const float4 *>(x + (plane * 2 * OH + 2 * oy + 1) * W + 8 * q);
        const float4 t0 = top[0], t1 = top[1], b0 = bot[0], b1 = bot[1];
        auto win = [](float a, float b, float c, float d) {
            float m = a;
            if (beats(b, m)) m = b;
            if (beats(c, m)) m = c;
            if (beats(d, m)) m = d;
            return m;
        };
        float4 o;
        o.x = win(t0.x, t0.y, b0.x, b0.y);
        o.y = win(t0.z, t0.w, b0.z, b0.w);
        o.z = win(t1.x, t1.y, b1.x, b1.y);
        o.w = win(t1.z, t1.w, b1.z, b1.w);
        reinterpret_cast<float4 *>(y)[i] = o;
    }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const float *x, const float *gy, float *gx, long quads,
                                                           int OH, int OW4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (long)gridDim.x * blockDim.x) {
        const int q = (int)(i % OW4);
        const long r = i / OW4;
        const long plane = r / OH;
        const int oy = (int)(r - plane * OH);
        const int W = OW4 * 8;
        const long o_top = (plane * 2 * OH + 2 * oy) * W + 8 * q, o_bot = o_top + W;
        const float4 t0 = *reinterpret_cast<const float4 *>(x + o_top), t1 = *reinterpret_cast<const float4 *>(x + o_top + 4);
        const float4 b0 = *reinterpret_cast<const float4 *>(x + o_bot), b1 = *reinterpret_cast<const float4 *>(x + o_bot + 4);
        const float4 g = reinterpret_cast<const float4 *>(gy)[i];
        // winner of a window (a b / c d): index of the first maximum in the order a, b, c, d
        auto route = [](float a, float b, float c, float d, float gv, float &ga, float &gb, float &gc, float &gd) {
            int k = 0;
            float m = a;
            if (beats(b, m)) { m = b; k = 1; }
            if (beats(c, m)) { m = c; k = 2; }
            if (beats(d, m)) { m = d; k = 3; }
            ga = k == 0 ? gv : 0.0f;
            gb = k == 1 ? gv : 0.0f;
            gc = k == 2 ? gv : 0.0f;
            gd = k == 3 ? gv : 0.0f;
        };
        float4 gt0, gt1, gb0, gb1;
        route(t0.x, t0.y, b0.x, b0.y, g.x, gt0.x, gt0.y, gb0.x, gb0.y);
        route(t0.z, t0.w, b0.z, b0.w, g.y, gt0.z, gt0.w, gb0.z, gb0.w);
        route(t1.x, t1.y, b1.x, b1.y, g.z, gt1.x, gt1.y, gb1.x, gb1.y);
        route(t1.z, t1.w, b1.z, b1.w, g.w, gt1.z, gt1.w, gb1.z, gb1.w);
        *reinterpret_cast<float4 *>(gx + o_top) = gt0;
        *reinterpret_cast<float4 *>(gx + o_top + 4) = gt1;
        *reinterpret_cast<float4 *>(gx + o_bot) = gb0;
        *reinterpret_cast<float4 *>(gx + o_bot + 4) = gb1;
    }
}

// The entry of the offset encoder's residual block (GAN2Shape/networks.py:170-194): both branches start
// from the same x — nn.ReLU() on the residual path, nn.AvgPool2d(2, 2) on the identity path.  One pass
// writes both; the backward joins the two incoming gradients in one pass as well
// (gx = ga * (x > 0) + gb / 4), where autograd runs threshold_backward, avg_pool2d_backward and an add.
// Lane = one 2x2 window.  H, W even.
__global__ __launch_bounds__(256) void res_split_fwd_kernel(const float *__restrict__ x, float *__restrict__ r,
                                                            float *__restrict__ p, long windows, int OH, int OW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= windows) return;
    const int ox = (int)(i % OW);
    const long row = i / OW;                     // (plane, output row)
    const long plane = row / OH;
    const int oy = (int)(row - plane * OH);
    const long base = (plane * 2 * OH + 2 * oy) * (2L * OW) + 2 * ox;
    const float2 t = *reinterpret_cast<const float2 *>(x + base), b = *reinterpret_cast<const float2 *>(x + base + 2 * OW);
    auto relu = [](float v) { return v > 0.0f ? v : (v != v ? v : 0.0f); };   // NaN stays NaN, like torch
    *reinterpret_cast<float2 *>(r + base) = make_float2(relu(t.x), relu(t.y));
    *reinterpret_cast<float2 *>(r + base + 2 * OW) = make_float2(relu(b.x), relu(b.y));
    p[i] = ((((0.0f + t.x) + t.y) + b.x) + b.y) / 4.0f;                        // ATen's accumulation order
}

__global__ __launch_bounds__(256) void res_split_bwd_kernel(const float *__restrict__ x, const float *__restrict__ ga,
                                                            const float *__restrict__ gb, float *__restrict__ gx,
                                                            long windows, int OH, int OW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= windows) return;
    const int ox = (int)(i % OW);
    const long row = i / OW;
    const long plane = row / OH;
    const int oy = (int)(row - plane * OH);
    const long base = (plane * 2 * OH + 2 * oy) * (2L * OW) + 2 * ox;
    const float2 t = *reinterpret_cast<const float2 *>(x + base), b = *reinterpret_cast<const float2 *>(x + base + 2 * OW);
    float2 at = make_float2(0.0f, 0.0f), ab = at;
    if (ga) {
        at = *reinterpret_cast<const float2 *>(ga + base);
        ab = *reinterpret_cast<const float2 *>(ga + base + 2 * OW);
    }
    const float q = gb ? gb[i] / 4.0f : 0.0f;
    auto gate = [](float g, float v) { return v > 0.0f ? g : 0.0f; };
    *reinterpret_cast<float2 *>(gx + base) = make_float2(gate(at.x, t.x) + q, gate(at.y, t.y) + q);
    *reinterpret_cast<float2 *>(gx + base + 2 * OW) = make_float2(gate(ab.x, b.x) + q, gate(ab.y, b.y) + q);
}

// The mask pyramid of the discriminator-feature loss (GAN2Shape/losses.py:24-30: the mask box-averaged down
// to every feature resolution): LEVELS successive avg_pool2d(., 2, 2) of a [planes, H, W] map in one launch.
// One lane owns a (2^LEVELS)^2 input block and reduces it level by level in registers, in ATen's order
// (((a + b) + c) + d) / 4 per window.
template <int LEVELS>
__global__ __launch_bounds__(64) void avg_pyramid_kernel(const float *__restrict__ x, float *__restrict__ o1,
                                                         float *__restrict__ o2, float *__restrict__ o3,
                                                         float *__restrict__ o4, long blocks, int H, int W) {
    constexpr int S = 1 << LEVELS;
    const long i = (long)blockIdx.x * 64 + threadIdx.x;
    if (i >= blocks) return;
    const int bw = W / S, bh = H / S;
    const int bx = (int)(i % bw);
    const long r = i / bw;
    const long plane = r / bh;
    const int by = (int)(r - plane * bh);
    float cur[S / 2][S / 2];
    const float *src = x + (plane * H + (long)by * S) * W + (long)bx * S;
#pragma unroll
    for (int yy = 0; yy < S / 2; yy++)
#pragma unroll
        for (int xx = 0; xx < S / 2; xx++) {
            const float2 t = *reinterpret_cast<const float2 *>(src + (2 * yy) * W + 2 * xx);
            const float2 b = *reinterpret_cast<const float2 *>(src + (2 * yy + 1) * W + 2 * xx);
            cur[yy][xx] = ((((0.0f + t.x) + t.y) + b.x) + b.y) / 4.0f;
        }
    float *outs[4] = {o1, o2, o3, o4};
    int side = S / 2;
#pragma unroll
    for (int l = 0; l < LEVELS; l++) {
        const int lw = W >> (l + 1), lh = H >> (l + 1);
        float *dst = outs[l] + (plane * lh + (long)by * side) * lw + (long)bx * side;
#pragma unroll
        for (int yy = 0; yy < S / 2; yy++)
#pragma unroll
            for (int xx = 0; xx < S / 2; xx++)
                if (yy < side && xx < side) dst[yy * lw + xx] = cur[yy][xx];
        if (l + 1 < LEVELS) {
#pragma unroll
            for (int yy = 0; yy < S / 4; yy++)
#pragma unroll
                for (int xx = 0; xx < S / 4; xx++)
                    if (yy < side / 2 && xx < side / 2)
                        cur[yy][xx] = ((((0.0f + cur[2 * yy][2 * xx]) + cur[2 * yy][2 * xx + 1]) + cur[2 * yy + 1][2 * xx]) +
                                       cur[2 * yy + 1][2 * xx + 1]) / 4.0f;
            side /= 2;
        }
    }
}

}  // namespace g2s

using namespace g2s;

static int pool_args(const void *a, const void *b, int64_t planes, int H, int W) {
    G2S_REQUIRE(a && b, "NULL pointer argument");
    G2S_REQUIRE(planes > 0 && H >= 2 && W >= 8 && H % 2 == 0 && W % 8 == 0, "H even, W a multiple of 8");
    G2S_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0, "16-byte alignment");
    return G2S_OK;
}

extern "C" int g2s_maxpool2x2_fwd(const float *x, float *y, int64_t planes, int H, int W, g2s_stream_t stream) {
    const int rc = pool_args(x, y, planes, H, W);
    if (rc != G2S_OK) return rc;
    const long quads = planes * (H / 2) * (W / 8);
    maxpool2_fwd_kernel<<<(unsigned)std::min<long>((quads + 255) / 256, 256 * 8), 256, 0, as_stream(stream)>>>(
        x, y, quads, H / 2, W / 8);
    return check_launch("g2s_maxpool2x2_fwd");
}

extern "C" int g2s_maxpool2x2_bwd(const float *x, const float *gy, float *gx, int64_t planes, int H, int W,
                                  g2s_stream_t stream) {
    const int rc = pool_args(x, gx, planes, H, W);
    if (rc != G2S_OK) return rc;
    G2S_REQUIRE(gy && (reinterpret_cast<uintptr_t>(gy) & 15) == 0, "gy NULL or misaligned");
    const long quads = planes * (H / 2) * (W / 8);
    maxpool2_bwd_kernel<<<(unsigned)std::min<long>((quads + 255) / 256, 256 * 8), 256, 0, as_stream(stream)>>>(
        x, gy, gx, quads, H / 2, W / 8);
    return check_launch("g2s_maxpool2x2_bwd");
}

static int res_split_check(int64_t planes, int H, int W) {
    G2S_REQUIRE(planes > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "planes, H, W positive; H, W even");
    G2S_REQUIRE(planes * H * W < (1LL << 40), "tensor too large");
    return G2S_OK;
}

extern "C" int g2s_res_split_fwd(const float *x, float *relu_out, float *pool_out, int64_t planes, int H, int W,
                                 g2s_stream_t stream) {
    G2S_REQUIRE(x && relu_out && pool_out, "NULL pointer argument");
    int rc = res_split_check(planes, H, W);
    if (rc) return rc;
    const long windows = (long)planes * (H / 2) * (W / 2);
    res_split_fwd_kernel<<<cdiv(windows, 256), 256, 0, as_stream(stream)>>>(x, relu_out, pool_out, windows, H / 2, W / 2);
    return check_launch("g2s_res_split_fwd");
}

extern "C" int g2s_res_split_bwd(const float *x, const float *g_relu, const float *g_pool, float *gx, int64_t planes,
                                 int H, int W, g2s_stream_t stream) {
    G2S_REQUIRE(x && gx && (g_relu || g_pool), "NULL pointer argument (one of g_relu / g_pool may be NULL)");
    int rc = res_split_check(planes, H, W);
    if (rc) return rc;
    const long windows = (long)planes * (H / 2) * (W / 2);
    res_split_bwd_kernel<<<cdiv(windows, 256), 256, 0, as_stream(stream)>>>(x, g_relu, g_pool, gx, windows, H / 2, W / 2);
    return check_launch("g2s_res_split_bwd");
}

extern "C" int g2s_avg_pyramid(const float *x, float *const *levels, int n_levels, int64_t planes, int H, int W,
                               g2s_stream_t stream) {
    G2S_REQUIRE(x && levels && planes > 0 && n_levels >= 1 && n_levels <= 4, "bad argument (1 <= n_levels <= 4)");
    const int S = 1 << n_levels;
    G2S_REQUIRE(H > 0 && W > 0 && H % S == 0 && W % S == 0, "H and W must be multiples of 2^n_levels");
    for (int l = 0; l < n_levels; l++) G2S_REQUIRE(levels[l], "NULL level pointer");
    float *o[4] = {levels[0], n_levels > 1 ? levels[1] : nullptr, n_levels > 2 ? levels[2] : nullptr,
                   n_levels > 3 ? levels[3] : nullptr};
    const long blocks = (long)planes * (H / S) * (W / S);
    hipStream_t st = as_stream(stream);
    const int grid = (int)cdiv(blocks, 64);
    switch (n_levels) {
        case 1: avg_pyramid_kernel<1><<<grid, 64, 0, st>>>(x, o[0], o[1], o[2], o[3], blocks, H, W); break;
        case 2: avg_pyramid_kernel<2><<<grid, 64, 0, st>>>(x, o[0], o[1], o[2], o[3], blocks, H, W); break;
        case 3: avg_pyramid_kernel<3><<<grid, 64, 0, st>>>(x, o[0], o[1], o[2], o[3], blocks, H, W); break;
        default: avg_pyramid_kernel<4><<<grid, 64, 0, st>>>(x, o[0], o[1], o[2], o[3], blocks, H, W); break;
    }
    return check_launch("g2s_avg_pyramid");
}
