// upfirdn2d.hip — zero-insert upsample -> pad/crop -> FIR -> downsample, for gfx950.
//
// Replaces upfirdn2d_op.upfirdn2d (GAN2Shape/stylegan2/stylegan2-pytorch/op/upfirdn2d.cpp:12-22,
// op/upfirdn2d_kernel.cu:107-207,209-369).  HBM-bound ((N_in + N_out) x 4 bytes): one workgroup
// stages the input footprint of a 32x32 output tile in LDS once (each input sample is read once
// from L2/HBM instead of up to 16 times) and each lane produces 4 outputs of one column, with the
// polyphase tap pattern resolved at compile time.
//
//   out[m, oy, ox] = sum_{ky,kx} U[oy*down + ky - pad_y0, ox*down + kx - pad_x0] * k[kh-1-ky, kw-1-kx]
//   U = x with (up-1) zeros inserted after every sample (out-of-range = 0).
#include "g2s_common.h"
#include <hip/hip_fp16.h>

namespace g2s {

struct UpfirParams {
    int major, in_h, in_w, out_h, out_w, kh, kw;
    int up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    // optional epilogue (g2s_upfirdn2d_nba): the StyledConv tail behind the blur of an up-sampling layer —
    // gain * leaky_relu(out + noise_w[0] * noise[oy, ox] + bias[m % channels], alpha); bias == NULL: none
    const float *bias, *noise, *noise_w;
    int channels;
    float alpha, gain;
};

template <typename T> __device__ __forceinline__ float upfir_finish(const UpfirParams &p, float acc, int m, int oy, int ox) {
    if (p.bias) {
        acc += p.bias[m % p.channels] + p.noise_w[0] * p.noise[oy * p.out_w + ox];
        acc = (acc > 0.0f ? acc : acc * p.alpha) * p.gain;
    }
    return acc;
}

__device__ __host__ __forceinline__ int floor_div(int a, int b) {
    int q = a / b;
    return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q;
}

template <typename T> __device__ __forceinline__ float ldf(const T *p);
template <> __device__ __forceinline__ float ldf<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float ldf<__half>(const __half *p) { return __half2float(*p); }
template <typename T> __device__ __forceinline__ void stf(T *p, float v);
template <> __device__ __forceinline__ void stf<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<__half>(__half *p, float v) { *p = __float2half(v); }

// Output tile TWX x TH per workgroup of 256 threads, block = (TW, 256 / TW): 32 x 32 for small maps;
// 64 x 16 / 128 x 16 for wider ones (whole rows: input rows are read as long contiguous runs);
// TWX = 160 with TW = 128 covers 129..160-wide outputs (the discriminator's 129 x 129 blur) in one
// tile per row band — lanes loop over x, only the first few take the second turn.
// UP, DOWN apply to both axes; KH x KW <= 4x4 compile-time taps.
template <typename T, int UP, int DOWN, int KH, int KW, int TW, int TH, int TWX = TW>
__global__ __launch_bounds__(256) void upfirdn2d_tiled(const T *__restrict__ x,
                                                       const float *__restrict__ k,
                                                       T *__restrict__ y, UpfirParams p) {
    constexpr int IN_H = ((TH - 1) * DOWN + KH - 1) / UP + 2;
    constexpr int IN_W = ((TWX - 1) * DOWN + KW - 1) / UP + 2;
    __shared__ float sx[IN_H][IN_W + 1];
    __shared__ float sk[KH][KW];  // flipped taps

    const int tid = threadIdx.y * TW + threadIdx.x;
    const int tiles_x = (p.out_w + TWX - 1) / TWX;
    const int tile_x0 = (blockIdx.x % tiles_x) * TWX;
    const int tile_y0 = (blockIdx.x / tiles_x) * TH;
    if (tid < KH * KW) {
        const int ky = tid / KW, kx = tid % KW;
        sk[ky][kx] = k[(KH - 1 - ky) * KW + (KW - 1 - kx)];
    }
    // first input sample that can contribute to this tile: smallest iy with iy*UP >= oy0*DOWN - pad
    const int in_y0 = floor_div(tile_y0 * DOWN - p.pad_y0 + UP - 1, UP);
    const int in_x0 = floor_div(tile_x0 * DOWN - p.pad_x0 + UP - 1, UP);

    // this thread's slots of the staged input footprint: plane-independent offsets (-1 = outside the
    // image -> 0); the next plane's samples are fetched into registers while the current one is filtered
    constexpr int NE = (IN_H * IN_W + 255) / 256;
    int soff[NE];
#pragma unroll
    for (int e = 0; e < NE; e++) {
        const int i = tid + e * 256;
        const int ry = i / IN_W, rx = i % IN_W;
        const int iy = in_y0 + ry, ix = in_x0 + rx;
        soff[e] = (i < IN_H * IN_W && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w) ? iy * p.in_w + ix : -1;
    }
    float pre[NE];
    auto fetch = [&](int m) {
        const T *xm = x + (size_t)m * p.in_h * p.in_w;
#pragma unroll
        for (int e = 0; e < NE; e++) pre[e] = soff[e] >= 0 ? ldf<T>(xm + soff[e]) : 0.0f;
    };
    if ((int)blockIdx.y < p.major) fetch(blockIdx.y);
    for (int m = blockIdx.y; m < p.major; m += gridDim.y) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; e++) {
            const int i = tid + e * 256;
            if (i < IN_H * IN_W) sx[i / IN_W][i % IN_W] = pre[e];
        }
        __syncthreads();
        if (m + (int)gridDim.y < p.major) fetch(m + gridDim.y);
        for (int ox = tile_x0 + threadIdx.x; ox < tile_x0 + TWX; ox += TW) {
        if constexpr (UP == 1) {
            // strips of 4 consecutive output rows per thread: the (3*DOWN + KH) x KW input window is
            // read from LDS once into registers (7 reads per output for the 4x4 blur instead of
            // 16 + 16 tap reads), taps live in registers
            constexpr int ROWS = 3 * DOWN + KH, STRIPS = TH / (4 * (256 / TW));
            static_assert(STRIPS >= 1 && STRIPS * 4 * (256 / TW) == TH, "tile shape");
            float kr[KH][KW];
#pragma unroll
            for (int a = 0; a < KH; a++)
#pragma unroll
                for (int b = 0; b < KW; b++) kr[a][b] = sk[a][b];
#pragma unroll
            for (int sp = 0; sp < STRIPS; sp++) {
            const int oy0 = tile_y0 + (threadIdx.y * STRIPS + sp) * 4;
            const int ry0 = oy0 * DOWN - p.pad_y0 - in_y0, rx0 = ox * DOWN - p.pad_x0 - in_x0;
            if (ox < p.out_w && oy0 < p.out_h) {
                float win[ROWS][KW];
#pragma unroll
                for (int j = 0; j < ROWS; j++)
#pragma unroll
                    for (int i = 0; i < KW; i++) win[j][i] = sx[ry0 + j][rx0 + i];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float acc = 0.0f;
#pragma unroll
                    for (int ky = 0; ky < KH; ky++)
#pragma unroll
                        for (int kx = 0; kx < KW; kx++) acc += win[r * DOWN + ky][kx] * kr[ky][kx];
                    if (oy0 + r < p.out_h)
                        stf<T>(y + ((size_t)m * p.out_h + oy0 + r) * p.out_w + ox, upfir_finish<T>(p, acc, m, oy0 + r, ox));
                }
            }
            }
        } else {
            static_assert(UP == 1 || (TW == 32 && TH == 32), "the up-sampling path uses 32 x 32 tiles");
#pragma unroll
        for (int r = 0; r < TH / 8; r++) {
            const int oy = tile_y0 + threadIdx.y + 8 * r;
            if (ox >= p.out_w || oy >= p.out_h) continue;
            const int by = oy * DOWN - p.pad_y0, bx = ox * DOWN - p.pad_x0;
            // first tap whose upsampled coordinate lands on a real sample
            const int ky0 = (UP == 1) ? 0 : ((UP - (by % UP + UP) % UP) % UP);
            const int kx0 = (UP == 1) ? 0 : ((UP - (bx % UP + UP) % UP) % UP);
            const int ry0 = floor_div(by + ky0, UP) - in_y0;
            const int rx0 = floor_div(bx + kx0, UP) - in_x0;
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < (KH + UP - 1) / UP; j++)
#pragma unroll
                for (int i = 0; i < (KW + UP - 1) / UP; i++) {
                    const int ky = ky0 + j * UP, kx = kx0 + i * UP;
                    if (ky < KH && kx < KW) acc += sx[ry0 + j][rx0 + i] * sk[ky][kx];
                }
            stf<T>(y + ((size_t)m * p.out_h + oy) * p.out_w + ox, upfir_finish<T>(p, acc, m, oy, ox));
        }
        }
        }
    }
}

// Generic fallback (any up/down per axis, any kernel size): one thread per output.
template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_generic(const T *__restrict__ x,
                                                         const float *__restrict__ k,
                                                         T *__restrict__ y, UpfirParams p) {
    const long total = (long)p.major * p.out_h * p.out_w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % p.out_w);
        const int oy = (int)((i / p.out_w) % p.out_h);
        const int m = (int)(i / ((long)p.out_w * p.out_h));
        const T *xm = x + (size_t)m * p.in_h * p.in_w;
        float acc = 0.0f;
        for (int ky = 0; ky < p.kh; ky++) {
            const int uy = oy * p.down_y + ky - p.pad_y0;
            if (uy < 0 || uy % p.up_y || uy / p.up_y >= p.in_h) continue;
            for (int kx = 0; kx < p.kw; kx++) {
                const int ux = ox * p.down_x + kx - p.pad_x0;
                if (ux < 0 || ux % p.up_x || ux / p.up_x >= p.in_w) continue;
                acc += ldf<T>(xm + (size_t)(uy / p.up_y) * p.in_w + ux / p.up_x) *
                       k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
            }
        }
        stf<T>(y + i, upfir_finish<T>(p, acc, m, oy, ox));
    }
}

template <typename T, int UP, int DOWN, int KH, int KW>
static void launch_tiled(const T *x, const float *k, T *y, const UpfirParams &p, hipStream_t st) {
    if (UP == 1 && p.out_w > 128 && p.out_w <= 160) {
        constexpr int TW = 128, TH = 16, TWX = 160;
        const int tiles = cdiv(p.out_w, TWX) * cdiv(p.out_h, TH);
        const int gy = std::min(p.major, std::max(1, 2048 / tiles));
        upfirdn2d_tiled<T, UP == 1 ? UP : 1, DOWN, KH, KW, TW, TH, TWX><<<dim3(tiles, gy), dim3(TW, 256 / TW), 0, st>>>(x, k, y, p);
    } else if (UP == 1 && p.out_w >= 96) {
        constexpr int TW = 128, TH = 16;
        const int tiles = cdiv(p.out_w, TW) * cdiv(p.out_h, TH);
        const int gy = std::min(p.major, std::max(1, 2048 / tiles));
        upfirdn2d_tiled<T, UP == 1 ? UP : 1, DOWN, KH, KW, TW, TH><<<dim3(tiles, gy), dim3(TW, 256 / TW), 0, st>>>(x, k, y, p);
    } else if (UP == 1 && p.out_w >= 48) {
        constexpr int TW = 64, TH = 16;
        const int tiles = cdiv(p.out_w, TW) * cdiv(p.out_h, TH);
        const int gy = std::min(p.major, std::max(1, 2048 / tiles));
        upfirdn2d_tiled<T, UP == 1 ? UP : 1, DOWN, KH, KW, TW, TH><<<dim3(tiles, gy), dim3(TW, 256 / TW), 0, st>>>(x, k, y, p);
    } else {
        constexpr int TW = 32, TH = 32;
        const int tiles = cdiv(p.out_w, TW) * cdiv(p.out_h, TH);
        const int gy = std::min(p.major, std::max(1, 2048 / tiles));
        upfirdn2d_tiled<T, UP, DOWN, KH, KW, TW, TH><<<dim3(tiles, gy), dim3(TW, 256 / TW), 0, st>>>(x, k, y, p);
    }
}

template <typename T>
static int dispatch(const T *x, const float *k, T *y, const UpfirParams &p, hipStream_t st) {
    const bool sq = p.up_x == p.up_y && p.down_x == p.down_y;
    const int up = p.up_x, down = p.down_x;
    if (sq && p.kh == 4 && p.kw == 4 && up == 1 && down == 1) launch_tiled<T, 1, 1, 4, 4>(x, k, y, p, st);
    else if (sq && p.kh == 4 && p.kw == 4 && up == 2 && down == 1) launch_tiled<T, 2, 1, 4, 4>(x, k, y, p, st);
    else if (sq && p.kh == 4 && p.kw == 4 && up == 1 && down == 2) launch_tiled<T, 1, 2, 4, 4>(x, k, y, p, st);
    else {
        const long total = (long)p.major * p.out_h * p.out_w;
        upfirdn2d_generic<T><<<std::min(cdiv(total, 256), 8192), 256, 0, st>>>(x, k, y, p);
    }
    return check_launch("g2s_upfirdn2d");
}

}  // namespace g2s

using namespace g2s;

static int upfirdn2d_launch(const void *x, const float *k, void *y, int major, int in_h, int in_w,
                            int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                            int pad_x1, int pad_y0, int pad_y1, int dtype, g2s_stream_t stream, int channels,
                            const float *bias, const float *noise, const float *noise_w, float alpha, float gain) {
    G2S_REQUIRE(x && k && y, "x, k, y must not be NULL");
    G2S_REQUIRE(major > 0 && in_h > 0 && in_w > 0 && kh > 0 && kw > 0, "sizes must be positive");
    G2S_REQUIRE(up_x > 0 && up_y > 0 && down_x > 0 && down_y > 0, "up/down must be positive");
    G2S_REQUIRE(dtype == G2S_F32 || dtype == G2S_F16, "dtype must be G2S_F32 or G2S_F16");
    UpfirParams p{};
    p.bias = bias;
    p.noise = noise;
    p.noise_w = noise_w;
    p.channels = channels;
    p.alpha = alpha;
    p.gain = gain;
    p.major = major;
    p.in_h = in_h;
    p.in_w = in_w;
    p.kh = kh;
    p.kw = kw;
    p.up_x = up_x;
    p.up_y = up_y;
    p.down_x = down_x;
    p.down_y = down_y;
    p.pad_x0 = pad_x0;
    p.pad_y0 = pad_y0;
    p.out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    p.out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    G2S_REQUIRE(p.out_h > 0 && p.out_w > 0, "empty output (%d x %d)", p.out_h, p.out_w);
    hipStream_t st = as_stream(stream);
    if (dtype == G2S_F32) return dispatch<float>((const float *)x, k, (float *)y, p, st);
    return dispatch<__half>((const __half *)x, k, (__half *)y, p, st);
}

extern "C" int g2s_upfirdn2d(const void *x, const float *k, void *y, int major, int in_h, int in_w,
                             int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                             int pad_x1, int pad_y0, int pad_y1, int dtype, g2s_stream_t stream) {
    return upfirdn2d_launch(x, k, y, major, in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0,
                            pad_y1, dtype, stream, 1, nullptr, nullptr, nullptr, 0.0f, 1.0f);
}

// upfirdn2d (float32) with the StyledConv tail of an up-sampling layer in its store (stylegan2-pytorch/
// model.py:264-275 Blur, then :349-355): y = gain * leaky_relu(upfirdn2d(x) + noise_w[0] * noise[oy, ox] +
// bias[m % channels], alpha); x [major = B * channels, in_h, in_w], noise [out_h, out_w].
extern "C" int g2s_upfirdn2d_nba(const float *x, const float *k, float *y, int major, int channels, int in_h, int in_w,
                                 int kh, int kw, int up, int down, int pad_x0, int pad_x1, int pad_y0, int pad_y1,
                                 const float *bias, const float *noise, const float *noise_w, float alpha, float gain,
                                 g2s_stream_t stream) {
    G2S_REQUIRE(bias && noise && noise_w && channels > 0 && major % channels == 0,
                "bias, noise, noise_w must not be NULL; major must be a multiple of channels");
    return upfirdn2d_launch(x, k, y, major, in_h, in_w, kh, kw, up, up, down, down, pad_x0, pad_x1, pad_y0, pad_y1,
                            G2S_F32, stream, channels, bias, noise, noise_w, alpha, gain);
}

