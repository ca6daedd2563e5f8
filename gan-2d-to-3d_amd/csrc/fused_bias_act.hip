// fused_bias_act.hip — bias + leaky-ReLU (+ noise) elementwise kernels for gfx950.
//
// Replaces fused.fused_bias_act (GAN2Shape/stylegan2/stylegan2-pytorch/op/fused_bias_act.cpp:11-20,
// op/fused_bias_act_kernel.cu:19-99).  HBM-bound: 2 x N x 4 bytes forward, 3 x N x 4 backward.
// 16 bytes per lane per access, grid-stride over at most 2048 workgroups.
#include "g2s_common.h"
#include <hip/hip_fp16.h>

namespace g2s {

enum { MODE_LIN = 0, MODE_ZERO = 1, MODE_LRELU = 2, MODE_LRELU_GRAD = 3 };

__device__ __forceinline__ float act_apply(int mode, float x, float ref, float alpha, float scale) {
    float y;
    switch (mode) {
    case MODE_ZERO: y = 0.0f; break;
    case MODE_LRELU: y = (x > 0.0f) ? x : x * alpha; break;
    case MODE_LRELU_GRAD: y = (ref > 0.0f) ? x : x * alpha; break;
    default: y = x; break;
    }
    return y * scale;
}

// Vector path: n % 4 == 0 and (no bias or step_b % 4 == 0): the 4 elements share one bias entry.
template <int MODE, bool BIAS>
__global__ __launch_bounds__(256) void fba_f32_vec4(const float4 *__restrict__ x,
                                                    const float *__restrict__ b,
                                                    const float4 *__restrict__ ref,
                                                    float4 *__restrict__ y, unsigned n4,
                                                    unsigned step_b4, unsigned size_b, float alpha,
                                                    float scale) {
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 v = x[i];
        if (BIAS) {
            const float bb = b[(i / step_b4) % size_b];
            v.x += bb;
            v.y += bb;
            v.z += bb;
            v.w += bb;
        }
        float4 r = make_float4(0, 0, 0, 0);
        if (MODE == MODE_LRELU_GRAD) r = ref[i];
        float4 o;
        o.x = act_apply(MODE, v.x, r.x, alpha, scale);
        o.y = act_apply(MODE, v.y, r.y, alpha, scale);
        o.z = act_apply(MODE, v.z, r.z, alpha, scale);
        o.w = act_apply(MODE, v.w, r.w, alpha, scale);
        y[i] = o;
    }
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<__half>(__half v) { return __half2float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ __half from_f<__half>(float v) { return __float2half(v); }

// Scalar path: any n / step_b, f32 or f16 storage (arithmetic in the storage type's value set,
// as the reference's scalar_t kernel does; here accumulated through float).
template <typename T>
__global__ __launch_bounds__(256) void fba_scalar(const T *__restrict__ x, const T *__restrict__ b,
                                                  const T *__restrict__ ref, T *__restrict__ y,
                                                  long n, long step_b, long size_b, int mode,
                                                  float alpha, float scale) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long)gridDim.x * blockDim.x) {
        float v = to_f<T>(x[i]);
        if (b) v = to_f<T>(from_f<T>(v + to_f<T>(b[(i / step_b) % size_b])));
        const float r = ref ? to_f<T>(ref[i]) : 0.0f;
        y[i] = from_f<T>(act_apply(mode, v, r, alpha, scale));
    }
}

// y[b,c,hw] = lrelu(x[b,c,hw] + nw * noise[hw] + bias[c]) * scale ; grid (ceil(HW4/256), B*C)
__global__ __launch_bounds__(256) void noise_bias_act_vec4(const float4 *__restrict__ x,
                                                           const float4 *__restrict__ noise,
                                                           const float *__restrict__ noise_w,
                                                           const float *__restrict__ bias,
                                                           float4 *__restrict__ y, int C, int HW4,
                                                           float alpha, float scale) {
    const int row = blockIdx.y;
    const float bb = bias ? bias[row % C] : 0.0f;
    const float nw = noise ? noise_w[0] : 0.0f;
    const float4 *xr = x + (size_t)row * HW4;
    float4 *yr = y + (size_t)row * HW4;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW4; i += gridDim.x * blockDim.x) {
        float4 v = xr[i];
        if (noise) {
            const float4 nz = noise[i];
            v.x += nw * nz.x;
            v.y += nw * nz.y;
            v.z += nw * nz.z;
            v.w += nw * nz.w;
        }
        float4 o;
        o.x = act_apply(MODE_LRELU, v.x + bb, 0.f, alpha, scale);
        o.y = act_apply(MODE_LRELU, v.y + bb, 0.f, alpha, scale);
        o.z = act_apply(MODE_LRELU, v.z + bb, 0.f, alpha, scale);
        o.w = act_apply(MODE_LRELU, v.w + bb, 0.f, alpha, scale);
        yr[i] = o;
    }
}

__global__ __launch_bounds__(256) void noise_bias_act_scalar(const float *__restrict__ x,
                                                             const float *__restrict__ noise,
                                                             const float *__restrict__ noise_w,
                                                             const float *__restrict__ bias,
                                                             float *__restrict__ y, int C, int HW,
                                                             float alpha, float scale) {
    const int row = blockIdx.y;
    const float bb = bias ? bias[row % C] : 0.0f;
    const float nw = noise ? noise_w[0] : 0.0f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < HW; i += gridDim.x * blockDim.x) {
        float v = x[(size_t)row * HW + i];
        if (noise) v += nw * noise[i];
        y[(size_t)row * HW + i] = act_apply(MODE_LRELU, v + bb, 0.f, alpha, scale);
    }
}

static int mode_of(int act, int grad) {
    switch (act * 10 + grad) {
    case 12: case 32: return MODE_ZERO;
    case 30: return MODE_LRELU;
    case 31: return MODE_LRELU_GRAD;
    default: return MODE_LIN;  // 10, 11 and the reference's `default:` label
    }
}

static bool aligned16(const void *p) { return ((uintptr_t)p & 15) == 0; }

template <int MODE>
static void launch_vec4(const float *x, const float *b, const float *ref, float *y, long n,
                        long step_b, long size_b, float alpha, float scale, hipStream_t st) {
    const unsigned n4 = (unsigned)(n / 4);
    const int grid = std::min(cdiv(n4, 256), 2048);
    if (b)
        fba_f32_vec4<MODE, true><<<grid, 256, 0, st>>>((const float4 *)x, b, (const float4 *)ref,
                                                      (float4 *)y, n4, (unsigned)(step_b / 4),
                                                      (unsigned)size_b, alpha, scale);
    else
        fba_f32_vec4<MODE, false><<<grid, 256, 0, st>>>((const float4 *)x, b, (const float4 *)ref,
                                                       (float4 *)y, n4, 1u, 1u, alpha, scale);
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_fused_bias_act(const void *x, const void *bias, const void *ref, void *y,
                                  int64_t n, int64_t step_b, int64_t size_b, int act, int grad,
                                  float alpha, float scale, int dtype, g2s_stream_t stream) {
    G2S_REQUIRE(n >= 0, "n must be >= 0");
    if (n == 0) return G2S_OK;
    G2S_REQUIRE(x && y, "x and y must not be NULL");
    G2S_REQUIRE(dtype == G2S_F32 || dtype == G2S_F16, "dtype must be G2S_F32 or G2S_F16");
    if (bias) G2S_REQUIRE(step_b > 0 && size_b > 0, "step_b and size_b must be positive with a bias");
    const int mode = mode_of(act, grad);
    if (mode == MODE_LRELU_GRAD) G2S_REQUIRE(ref != nullptr, "act=3, grad=1 needs ref");
    hipStream_t st = as_stream(stream);
    const bool vec = dtype == G2S_F32 && n % 4 == 0 && n / 4 < 0xffffffffll &&
                     (!bias || step_b % 4 == 0) && aligned16(x) && aligned16(y) &&
                     (mode != MODE_LRELU_GRAD || aligned16(ref));
    if (vec) {
        const float *xf = (const float *)x, *bf = (const float *)bias, *rf = (const float *)ref;
        float *yf = (float *)y;
        switch (mode) {
        case MODE_ZERO: launch_vec4<MODE_ZERO>(xf, bf, rf, yf, n, step_b, size_b, alpha, scale, st); break;
        case MODE_LRELU: launch_vec4<MODE_LRELU>(xf, bf, rf, yf, n, step_b, size_b, alpha, scale, st); break;
        case MODE_LRELU_GRAD: launch_vec4<MODE_LRELU_GRAD>(xf, bf, rf, yf, n, step_b, size_b, alpha, scale, st); break;
        default: launch_vec4<MODE_LIN>(xf, bf, rf, yf, n, step_b, size_b, alpha, scale, st); break;
        }
    } else {
        const int grid = std::min(cdiv(n, 256), 4096);
        const long sb = bias ? step_b : 1, zb = bias ? size_b : 1;
        if (dtype == G2S_F32)
            fba_scalar<float><<<grid, 256, 0, st>>>((const float *)x, (const float *)bias,
                                                    (const float *)ref, (float *)y, n, sb, zb, mode,
                                                    alpha, scale);
        else
            fba_scalar<__half><<<grid, 256, 0, st>>>((const __half *)x, (const __half *)bias,
                                                     (const __half *)ref, (__half *)y, n, sb, zb,
                                                     mode, alpha, scale);
    }
    return check_launch("g2s_fused_bias_act");
}

extern "C" int g2s_noise_bias_act(const float *x, const float *noise, const float *noise_w,
                                  const float *bias, float *y, int B, int C, int HW, float alpha,
                                  float scale, g2s_stream_t stream) {
    G2S_REQUIRE(x && y, "x and y must not be NULL");
    G2S_REQUIRE(B > 0 && C > 0 && HW > 0, "B, C, HW must be positive");
    G2S_REQUIRE((long)B * C <= 65535, "B*C too large for grid.y");
    if (noise) G2S_REQUIRE(noise_w != nullptr, "noise needs noise_w");
    hipStream_t st = as_stream(stream);
    if (HW % 4 == 0 && aligned16(x) && aligned16(y) && (!noise || aligned16(noise))) {
        const int HW4 = HW / 4;
        noise_bias_act_vec4<<<dim3(std::min(cdiv(HW4, 256), 64), B * C), 256, 0, st>>>(
            (const float4 *)x, (const float4 *)noise, noise_w, bias, (float4 *)y, C, HW4, alpha, scale);
    } else {
        noise_bias_act_scalar<<<dim3(std::min(cdiv(HW, 256), 64), B * C), 256, 0, st>>>(
            x, noise, noise_w, bias, y, C, HW, alpha, scale);
    }
    return check_launch("g2s_noise_bias_act");
}

// ------------------------------------------------------------------------------------------------
// y = (a + b + bias[c]) * scale: the residual joins of the frozen nets as one pass — ToRGB
// (stylegan2-pytorch/model.py:371-377: conv + bias, then + upsample(skip)) and the discriminator's
// ResBlock (model.py:693-697: (out + skip) / sqrt(2)).  b and bias may be NULL.
namespace g2s {

__global__ __launch_bounds__(256) void add_bias_scale_kernel(const float *a, const float *b, const float *bias,
                                                             float *y, int64_t n, int64_t hw, int C, float scale) {
    const bool vec = (n & 3) == 0 && (hw & 3) == 0 &&
                     ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (vec) {
            float4 v = *reinterpret_cast<const float4 *>(a + i);
            if (b) {
                const float4 w = *reinterpret_cast<const float4 *>(b + i);
                v.x += w.x;
                v.y += w.y;
                v.z += w.z;
                v.w += w.w;
            }
            const float bi = bias ? bias[(i / hw) % C] : 0.0f;   // hw % 4 == 0: one channel per lane
            *reinterpret_cast<float4 *>(y + i) = float4{(v.x + bi) * scale, (v.y + bi) * scale, (v.z + bi) * scale,
                                                        (v.w + bi) * scale};
        } else {
            for (int64_t j = i; j < i + 4 && j < n; j++)
                y[j] = (a[j] + (b ? b[j] : 0.0f) + (bias ? bias[(j / hw) % C] : 0.0f)) * scale;
        }
    }
}

// y = clamp(x, lo, hi) and its backward gx = g where lo <= x <= hi (torch's rule, bounds inclusive), one
// launch each; dir 0: forward (g unused), 1: backward.
__global__ void clamp_kernel(const float *__restrict__ x, const float *__restrict__ g, float *__restrict__ y,
                             int64_t n, float lo, float hi, int dir) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n && (((uintptr_t)(x + i) | (uintptr_t)(y + i) | (uintptr_t)(dir ? g + i : x + i)) & 15) == 0) {
            const float4 v = *reinterpret_cast<const float4 *>(x + i);
            float4 o;
            if (dir) {
                const float4 q = *reinterpret_cast<const float4 *>(g + i);
                o = float4{(v.x >= lo && v.x <= hi) ? q.x : 0.0f, (v.y >= lo && v.y <= hi) ? q.y : 0.0f,
                           (v.z >= lo && v.z <= hi) ? q.z : 0.0f, (v.w >= lo && v.w <= hi) ? q.w : 0.0f};
            } else {
                auto cl = [&](float t) { return t < lo ? lo : (t > hi ? hi : t); };   // NaN stays NaN, as in torch
                o = float4{cl(v.x), cl(v.y), cl(v.z), cl(v.w)};
            }
            *reinterpret_cast<float4 *>(y + i) = o;
        } else {
            for (int64_t j = i; j < i + 4 && j < n; j++)
                y[j] = dir ? ((x[j] >= lo && x[j] <= hi) ? g[j] : 0.0f) : (x[j] < lo ? lo : (x[j] > hi ? hi : x[j]));
        }
    }
}

}  // namespace g2s

extern "C" int g2s_clamp(const float *x, const float *g, float *y, int64_t n, float lo, float hi, int backward,
                         g2s_stream_t stream) {
    G2S_REQUIRE(x && y && n > 0 && (!backward || g) && lo <= hi, "bad argument");
    const int64_t quads = (n + 3) / 4;
    const int blocks = (int)std::min<int64_t>((quads + 255) / 256, 256 * 8);
    g2s::clamp_kernel<<<blocks, 256, 0, g2s::as_stream(stream)>>>(x, g, y, n, lo, hi, backward ? 1 : 0);
    return g2s::check_launch("g2s_clamp");
}

extern "C" int g2s_add_bias_scale(const float *a, const float *b, const float *bias, float *y, int64_t n, int64_t hw,
                                  int C, float scale, g2s_stream_t stream) {
    G2S_REQUIRE(a && y, "a and y must not be NULL");
    G2S_REQUIRE(n > 0 && hw > 0 && C > 0, "n, hw, C must be positive");
    const int64_t quads = (n + 3) / 4;
    const int blocks = (int)std::min<int64_t>((quads + 255) / 256, 256 * 8);
    g2s::add_bias_scale_kernel<<<blocks, 256, 0, g2s::as_stream(stream)>>>(a, b, bias, y, n, hw, C, scale);
    return g2s::check_launch("g2s_add_bias_scale");
}

