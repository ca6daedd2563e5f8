// grid_sample.hip — the reconstruction warp of the step: F.grid_sample(texture, grid, 'bilinear',
// align_corners=True, padding 'zeros') followed by .clamp(-1, 1) (GAN2Shape/model.py:150,270 after
// :147,:267 — the reference calls torch.nn.functional.grid_sample, ATen GridSampler), forward and
// backward as one launch each.
//
// Why the product carries its own: ATen's backward scatters the texture gradient with float atomics
// and has no deterministic variant (the only such op on the training path, tools/check_reproducibility.py).
// Here the scatter is float atomics by default and 64-bit fixed-point integer atomics in deterministic
// mode — integer adds commute, so the gradient is bit-identical from run to run.  The clamp rides in
// the same pass (its gate is re-derived from the recomputed sample in the backward).
//
// HBM-bound gather: one thread per output pixel, the C channels in a loop (C = 3 here); the four
// texel reads per channel hit L2 (neighbouring pixels share them).  Arithmetic follows ATen's
// (nw, ne, sw, se order, weights from the corner differences), so results agree with torch to the
// last bit when neither side contracts to FMA, else to one rounding.
#include "g2s_common.h"

namespace g2s {

struct GsDims { int C, IH, IW, H, W; };

struct Corner {
    int x0, y0;              // nw texel
    float wx0, wx1, wy0, wy1;  // (ix_se - ix), (ix - ix_nw), (iy_se - iy), (iy - iy_nw)
    bool in_x0, in_x1, in_y0, in_y1;
};

__device__ __forceinline__ Corner corner_of(float gx, float gy, const GsDims &d) {
    // align_corners=True: -1 -> texel 0, +1 -> texel size-1
    const float ix = ((gx + 1.0f) / 2.0f) * (float)(d.IW - 1), iy = ((gy + 1.0f) / 2.0f) * (float)(d.IH - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    Corner c;
    // a float outside the int range (or NaN) must not hit undefined conversion: such a pixel has no
    // in-range corner, as in ATen
    const bool sane = fx >= -2.0f && fx <= (float)d.IW && fy >= -2.0f && fy <= (float)d.IH;
    c.x0 = sane ? (int)fx : -2;
    c.y0 = sane ? (int)fy : -2;
    c.wx0 = (fx + 1.0f) - ix;
    c.wx1 = ix - fx;
    c.wy0 = (fy + 1.0f) - iy;
    c.wy1 = iy - fy;
    c.in_x0 = c.x0 >= 0 && c.x0 < d.IW;
    c.in_x1 = c.x0 + 1 >= 0 && c.x0 + 1 < d.IW;
    c.in_y0 = c.y0 >= 0 && c.y0 < d.IH;
    c.in_y1 = c.y0 + 1 >= 0 && c.y0 + 1 < d.IH;
    return c;
}

// grid (ceil(H*W / 256), B)
__global__ __launch_bounds__(256) void grid_sample_fwd(const float *__restrict__ x, const float *__restrict__ grid,
                                                       float *__restrict__ y, GsDims d, int clamp, float lo, float hi) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.H * d.W) return;
    const float *g = grid + ((size_t)b * d.H * d.W + p) * 2;
    const Corner c = corner_of(g[0], g[1], d);
    const float nw = c.wx0 * c.wy0, ne = c.wx1 * c.wy0, sw = c.wx0 * c.wy1, se = c.wx1 * c.wy1;
    const size_t plane = (size_t)d.IH * d.IW;
    const float *xb = x + (size_t)b * d.C * plane + (long)c.y0 * d.IW + c.x0;
    float *yb = y + (size_t)b * d.C * d.H * d.W + p;
    for (int ch = 0; ch < d.C; ch++, xb += plane, yb += (size_t)d.H * d.W) {
        float v = 0.0f;
        if (c.in_y0 && c.in_x0) v += xb[0] * nw;
        if (c.in_y0 && c.in_x1) v += xb[1] * ne;
        if (c.in_y1 && c.in_x0) v += xb[d.IW] * sw;
        if (c.in_y1 && c.in_x1) v += xb[d.IW + 1] * se;
        if (clamp) v = v < lo ? lo : (v > hi ? hi : v);   // NaN passes through, like torch.clamp
        *yb = v;
    }
}

constexpr float GS_FIX = 1099511627776.0f;  // 2^40: range +-8.4e6, resolution 9e-13

template <bool FIXED>
__device__ __forceinline__ void scatter(float *gx, long long *gfix, size_t idx, float v) {
    if (FIXED) atomicAdd(reinterpret_cast<unsigned long long *>(gfix + idx), (unsigned long long)__float2ll_rn(v * GS_FIX));
    else unsafeAtomicAdd(gx + idx, v);
}

// gx (or gfix) zero-filled by the launcher; ggrid written directly.  Either may be NULL.
template <bool FIXED>
__global__ __launch_bounds__(256) void grid_sample_bwd(const float *__restrict__ gy, const float *__restrict__ x,
                                                       const float *__restrict__ grid, float *__restrict__ gx,
                                                       long long *__restrict__ gfix, float *__restrict__ ggrid,
                                                       GsDims d, int clamp, float lo, float hi, int need_gx) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.H * d.W) return;
    const float *g = grid + ((size_t)b * d.H * d.W + p) * 2;
    const Corner c = corner_of(g[0], g[1], d);
    const float nw = c.wx0 * c.wy0, ne = c.wx1 * c.wy0, sw = c.wx0 * c.wy1, se = c.wx1 * c.wy1;
    const size_t plane = (size_t)d.IH * d.IW;
    const size_t base = (size_t)b * d.C * plane + (long)c.y0 * d.IW + c.x0;   // only dereferenced for in-range corners
    const float *gyb = gy + (size_t)b * d.C * d.H * d.W + p;
    float gix = 0.0f, giy = 0.0f;
    for (int ch = 0; ch < d.C; ch++, gyb += (size_t)d.H * d.W) {
        const size_t o = base + (size_t)ch * plane;
        const float v_nw = (c.in_y0 && c.in_x0) ? x[o] : 0.0f, v_ne = (c.in_y0 && c.in_x1) ? x[o + 1] : 0.0f;
        const float v_sw = (c.in_y1 && c.in_x0) ? x[o + d.IW] : 0.0f, v_se = (c.in_y1 && c.in_x1) ? x[o + d.IW + 1] : 0.0f;
        float go = *gyb;
        if (clamp) {   // the forward's own sum, so the gate sees the value that was clamped
            float v = 0.0f;
            if (c.in_y0 && c.in_x0) v += v_nw * nw;
            if (c.in_y0 && c.in_x1) v += v_ne * ne;
            if (c.in_y1 && c.in_x0) v += v_sw * sw;
            if (c.in_y1 && c.in_x1) v += v_se * se;
            if (!(v >= lo && v <= hi)) go = 0.0f;
        }
        if (need_gx) {
            if (c.in_y0 && c.in_x0) scatter<FIXED>(gx, gfix, o, nw * go);
            if (c.in_y0 && c.in_x1) scatter<FIXED>(gx, gfix, o + 1, ne * go);
            if (c.in_y1 && c.in_x0) scatter<FIXED>(gx, gfix, o + d.IW, sw * go);
            if (c.in_y1 && c.in_x1) scatter<FIXED>(gx, gfix, o + d.IW + 1, se * go);
        }
        gix -= v_nw * c.wy0 * go;  giy -= v_nw * c.wx0 * go;
        gix += v_ne * c.wy0 * go;  giy -= v_ne * c.wx1 * go;
        gix -= v_sw * c.wy1 * go;  giy += v_sw * c.wx0 * go;
        gix += v_se * c.wy1 * go;  giy += v_se * c.wx1 * go;
    }
    if (ggrid) {
        float *o = ggrid + ((size_t)b * d.H * d.W + p) * 2;
        o[0] = ((float)(d.IW - 1) / 2.0f) * gix;
        o[1] = ((float)(d.IH - 1) / 2.0f) * giy;
    }
}

__global__ __launch_bounds__(256) void fixed_to_float(const long long *__restrict__ src, float *__restrict__ dst, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = (float)((double)src[i] * (1.0 / 1099511627776.0));
}

static int gs_check(int B, int C, int IH, int IW, int H, int W) {
    G2S_REQUIRE(B > 0 && C > 0 && IH > 1 && IW > 1 && H > 0 && W > 0, "sizes must be positive (input side >= 2)");
    G2S_REQUIRE(B <= 65535 && (long)H * W < (1L << 30) && (long)IH * IW < (1L << 30), "image too large");
    return G2S_OK;
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_grid_sample_fwd(const float *x, const float *grid, float *y, int B, int C, int IH, int IW,
                                   int H, int W, int clamp, float lo, float hi, g2s_stream_t stream) {
    G2S_REQUIRE(x && grid && y, "NULL pointer argument");
    int rc = gs_check(B, C, IH, IW, H, W);
    if (rc) return rc;
    grid_sample_fwd<<<dim3(cdiv(H * W, 256), B), 256, 0, as_stream(stream)>>>(x, grid, y, GsDims{C, IH, IW, H, W},
                                                                             clamp, lo, hi);
    return check_launch("g2s_grid_sample_fwd");
}

extern "C" size_t g2s_grid_sample_bwd_workspace_bytes(int B, int C, int IH, int IW) {
    if (B <= 0 || C <= 0 || IH <= 0 || IW <= 0) return 0;
    return (size_t)B * C * IH * IW * sizeof(long long) + 256;
}

extern "C" int g2s_grid_sample_bwd(const float *gy, const float *x, const float *grid, float *gx, float *ggrid,
                                   int B, int C, int IH, int IW, int H, int W, int clamp, float lo, float hi,
                                   void *workspace, size_t workspace_bytes, g2s_stream_t stream) {
    G2S_REQUIRE(gy && x && grid && (gx || ggrid), "NULL pointer argument (one of gx / ggrid may be NULL)");
    int rc = gs_check(B, C, IH, IW, H, W);
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    const GsDims d{C, IH, IW, H, W};
    const size_t n = (size_t)B * C * IH * IW;
    const dim3 g(cdiv(H * W, 256), B);
    if (gx && deterministic()) {
        if (!workspace || workspace_bytes < g2s_grid_sample_bwd_workspace_bytes(B, C, IH, IW))
            return fail(G2S_ERR_WORKSPACE, "deterministic mode: the backward needs its fixed-point workspace "
                        "(g2s_grid_sample_bwd_workspace_bytes = %zu bytes, got %zu)",
                        g2s_grid_sample_bwd_workspace_bytes(B, C, IH, IW), workspace ? workspace_bytes : (size_t)0);
        long long *fix = reinterpret_cast<long long *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
        if (!precleared() && hipMemsetAsync(fix, 0, n * sizeof(long long), st) != hipSuccess)
            return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(workspace) failed");
        grid_sample_bwd<true><<<g, 256, 0, st>>>(gy, x, grid, gx, fix, ggrid, d, clamp, lo, hi, 1);
        fixed_to_float<<<cdiv((long)n, 256), 256, 0, st>>>(fix, gx, (long)n);
    } else {
        if (gx && !precleared() && hipMemsetAsync(gx, 0, n * sizeof(float), st) != hipSuccess)
            return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(gx) failed");
        grid_sample_bwd<false><<<g, 256, 0, st>>>(gy, x, grid, gx, nullptr, ggrid, d, clamp, lo, hi, gx != nullptr);
    }
    return check_launch("g2s_grid_sample_bwd");
}
