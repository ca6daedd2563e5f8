// raster_rgb.hip — texture path of the rasterizer: nr.Renderer.render_rgb as GAN2Shape's
// visualisation helpers use it (GAN2Shape/renderer/renderer.py:196,230,248,272,275: render_yaw,
// render_view, render_given_view(grid_sample=False) with texture cubes from
// renderer/utils.py:83-109).  Forward only (nothing in the reference differentiates through it).
//
// Second pass over the maps the depth rasterizer (raster.hip) saves — winning face id and clamped,
// renormalised barycentric weights per supersample: per sample the texture cube of the winning
// face is read trilinearly at perspective-corrected coordinates
//     t_k = clamp(w_k * (ts - 1) * D / z_k, 0, ts - 1 - eps),   D = 1 / sum_k (w_k / z_k)
// (the reversed fill_back copy of a face reads the cube with axes 0 and 2 swapped), background
// samples take the background colour, then the same vertical flip + 2x2 average as the depth map.
// Semantics follow the external neural_renderer package as recalled in SURVEY.md Appendix A
// (PARITY UNPINNED, like the depth path); the oracle restates them independently on the CPU.
#include "g2s_common.h"
#include "raster_core.h"

namespace g2s {

struct RgbParams {
    const float *verts;      // [B, N, 3] camera space (z is what the weights are corrected with)
    const int32_t *faces;    // [F, 3] or NULL (implicit regular grid)
    const int32_t *face_idx; // [B, is, is]
    const float *bary;       // [B, is, is, 3]
    const float *tex;        // [B, F, ts, ts, ts, C]
    float *out;              // [B, C, S, S]
    int B, N, F, S, is, ssaa, ts, C;
    float eps;
    float bg[4];
};

template <bool IMPLICIT>
__device__ __forceinline__ void sample_colour(const RgbParams &p, int b, int yi, int xi, float col[4]) {
    const size_t si = ((size_t)b * p.is + yi) * p.is + xi;
    const int fn = p.face_idx[si];
    if (fn < 0) {
        for (int c = 0; c < p.C; c++) col[c] = p.bg[c];
        return;
    }
    const int g = fn % p.F;
    const bool rev = fn >= p.F;
    int v[3];
    if (IMPLICIT) {
        implicit_face(g, p.S, v);
    } else {
        v[0] = p.faces[3 * g];
        v[1] = p.faces[3 * g + 1];
        v[2] = p.faces[3 * g + 2];
    }
    if (rev) {
        const int t = v[0];
        v[0] = v[2];
        v[2] = t;
    }
    float w[3], z[3];
    for (int k = 0; k < 3; k++) {
        w[k] = p.bary[3 * si + k];
        z[k] = p.verts[((size_t)b * p.N + v[k]) * 3 + 2];
    }
    const float depth = 1.0f / (w[0] / z[0] + w[1] / z[1] + w[2] / z[2]);
    const int ts = p.ts;
    float tf[3];
    int ti[3];
    for (int k = 0; k < 3; k++) {
        float t = w[k] * (float)(ts - 1) * (depth / z[k]);
        t = fmaxf(t, 0.0f);
        t = fminf(t, (float)(ts - 1) - p.eps);
        ti[k] = (int)t;
        tf[k] = t - (float)ti[k];
    }
    const float *tex = p.tex + ((size_t)b * p.F + g) * ts * ts * ts * p.C;
    for (int c = 0; c < p.C; c++) col[c] = 0.0f;
    for (int pn = 0; pn < 8; pn++) {
        float wt = 1.0f;
        int idx[3];
        for (int k = 0; k < 3; k++) {
            if (((pn >> k) & 1) == 0) {
                wt *= 1.0f - tf[k];
                idx[k] = ti[k];
            } else {
                wt *= tf[k];
                idx[k] = min(ti[k] + 1, ts - 1);   // weight 0 there when ts == 1
            }
        }
        // the reversed copy's cube is textures.permute(0, 1, 4, 3, 2, 5): axes 0 and 2 swapped
        const int isc = rev ? (idx[2] * ts + idx[1]) * ts + idx[0] : (idx[0] * ts + idx[1]) * ts + idx[2];
        for (int c = 0; c < p.C; c++) col[c] += wt * tex[(size_t)isc * p.C + c];
    }
}

template <bool IMPLICIT>
__global__ void raster_rgb_kernel(RgbParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.B * p.S * p.S) return;
    const int b = (int)(i / ((long)p.S * p.S));
    const int r = (int)((i / p.S) % p.S), c0 = (int)(i % p.S);
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int dy = 0; dy < p.ssaa; dy++)
        for (int dx = 0; dx < p.ssaa; dx++) {
            const int yi = p.is - 1 - (r * p.ssaa + dy);   // row of the (unflipped) raster
            float col[4];
            sample_colour<IMPLICIT>(p, b, yi, c0 * p.ssaa + dx, col);
            for (int c = 0; c < p.C; c++) sum[c] += col[c];
        }
    const float inv = 1.0f / (float)(p.ssaa * p.ssaa);
    for (int c = 0; c < p.C; c++) p.out[(((size_t)b * p.C + c) * p.S + r) * p.S + c0] = sum[c] * inv;
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_raster_rgb_fwd(const float *verts, const int32_t *faces, const int32_t *face_idx,
                                  const float *bary, const float *textures, int B, int n_verts, int n_faces,
                                  int S, int ssaa, int ts, int C, const float *background, float eps,
                                  float *rgb_out, g2s_stream_t stream) {
    G2S_REQUIRE(verts && face_idx && bary && textures && rgb_out && background, "NULL pointer argument");
    G2S_REQUIRE(B > 0 && n_verts > 0 && n_faces > 0 && S > 0, "sizes must be positive");
    G2S_REQUIRE(ssaa == 1 || ssaa == 2, "ssaa must be 1 or 2");
    G2S_REQUIRE(ts >= 1 && ts <= 8 && C >= 1 && C <= 4, "texture size 1..8, 1..4 channels");
    G2S_REQUIRE(faces || (n_verts == S * S && n_faces == 2 * (S - 1) * (S - 1)),
                "implicit topology needs S*S vertices and 2(S-1)^2 faces");
    RgbParams p{};
    p.verts = verts;
    p.faces = faces;
    p.face_idx = face_idx;
    p.bary = bary;
    p.tex = textures;
    p.out = rgb_out;
    p.B = B;
    p.N = n_verts;
    p.F = n_faces;
    p.S = S;
    p.is = S * ssaa;
    p.ssaa = ssaa;
    p.ts = ts;
    p.C = C;
    p.eps = eps;
    for (int c = 0; c < C; c++) p.bg[c] = background[c];
    const long n = (long)B * S * S;
    if (faces) raster_rgb_kernel<false><<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(p);
    else raster_rgb_kernel<true><<<cdiv(n, 256), 256, 0, as_stream(stream)>>>(p);
    return check_launch("g2s_raster_rgb_fwd");
}
