// raster.hip — differentiable depth rasterizer for gfx950 (MI355X), behind
// g2s_raster_depth_fwd / g2s_raster_depth_bwd (include/g2s.h).
//
// Replaces neural_renderer.Renderer.render_depth as used by GAN2Shape/renderer/renderer.py:116-125
// (semantics: SURVEY.md Appendix A).  The reference's kernel tests every one of the 64 516 faces at
// every one of the 65 536 raster samples (4.2e9 tests per image).  Here:
//
//   setup   one wavefront per 8x8-quad mesh block (implicit grid topology) or per 64 consecutive
//           faces (explicit topology): project vertices once, reduce the block's bounding box.
//   tiles   one wavefront per 8x8-sample tile, 4 tiles (16x16 samples) per workgroup, no barriers:
//             1. lanes test chunk boxes against the tile, wave-ballot the survivors;
//             2. per surviving chunk, lane = quad / face: front-face test + face box test + exact
//                (conservative) triangle-vs-tile edge test, ballot
//                + mbcnt compaction of the surviving oriented faces into a per-wave LDS list;
//             3. lane = sample: walk the list with LDS broadcast reads, three edge inequalities per
//                face, remember covering faces (up to 8 byte-indices in a 64-bit register);
//             4. lane = sample: evaluate only the remembered (sample, face) fragments — inverse
//                vertex matrix, clamped barycentrics, perspective-correct depth — keep the
//                lexicographic (depth, face id) minimum;
//             5. epilogue fused: vertical flip + 2x2 average pooling through DPP shuffles.
//           ~70 candidate faces per tile instead of 64 516.
//   bwd     one thread per raster sample: analytic gradient to the three projected vertices with
//           float atomics, then one thread per vertex for the projection backward.
//
// Compiled with -ffp-contract=off: results equal oracle/raster_body.inc bit for bit up to the
// association order of the 2x2 average.
#include "g2s_common.h"
#include "raster_core.h"
#include <limits.h>

namespace g2s {

constexpr int TILE = 8;      // samples per tile side (64 samples = one wavefront)
constexpr int CAP = 128;     // LDS face-list capacity per wavefront
constexpr int WAVES = 4;     // wavefronts (tiles) per workgroup
constexpr int MAXH = 8;      // covering faces remembered per sample before evaluation

struct RasterParams {
    const float *verts;
    const int32_t *faces;
    int B, N, F, S, is, ssaa, fill_back;
    float near_, far_;
    Cam cam;
    float4 *proj;     // [B, N]   (u_n, v_n, z, -)
    float4 *chunkbb;  // [B, nchunks] (xmin, ymin, xmax, ymax)
    int nchunks, nblk_side;
    float *depth_out;
    int32_t *face_idx;
    float *bary;
};

__device__ __forceinline__ float wave_min(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---------------------------------------------------------------------------------- setup
// Implicit topology: grid (nblk_side^2, B), 64 threads. Block (bi, bj) owns quads [8bi, 8bi+8) x
// [8bj, 8bj+8) and therefore the 9x9 vertices around them.
__global__ __launch_bounds__(64) void raster_setup_implicit(RasterParams p) {
    const int blk = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int bi = blk / p.nblk_side, bj = blk % p.nblk_side;
    float xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
    for (int t = lane; t < 81; t += 64) {
        const int i = bi * 8 + t / 9, j = bj * 8 + t % 9;
        if (i < p.S && j < p.S) {
            const int vid = i * p.S + j;
            const float *v = p.verts + ((size_t)b * p.N + vid) * 3;
            float u, w;
            project(v[0], v[1], v[2], p.cam, u, w);
            p.proj[(size_t)b * p.N + vid] = make_float4(u, w, v[2], 0.0f);
            xmin = fminf(xmin, u);
            xmax = fmaxf(xmax, u);
            ymin = fminf(ymin, w);
            ymax = fmaxf(ymax, w);
        }
    }
    xmin = wave_min(xmin);
    ymin = wave_min(ymin);
    xmax = wave_max(xmax);
    ymax = wave_max(ymax);
    if (lane == 0) p.chunkbb[(size_t)b * p.nchunks + blk] = make_float4(xmin, ymin, xmax, ymax);
}

// Explicit topology, step 1: one thread per vertex.
__global__ __launch_bounds__(256) void raster_project(RasterParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.B * p.N) return;
    const float *v = p.verts + i * 3;
    float u, w;
    project(v[0], v[1], v[2], p.cam, u, w);
    p.proj[i] = make_float4(u, w, v[2], 0.0f);
}

// Explicit topology, step 2: one wavefront per chunk of 64 consecutive faces.
__global__ __launch_bounds__(64) void raster_chunk_boxes(RasterParams p) {
    const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int f = c * 64 + lane;
    float xmin = INFINITY, ymin = INFINITY, xmax = -INFINITY, ymax = -INFINITY;
    if (f < p.F) {
        for (int k = 0; k < 3; k++) {
            const float4 q = p.proj[(size_t)b * p.N + p.faces[3 * f + k]];
            xmin = fminf(xmin, q.x);
            xmax = fmaxf(xmax, q.x);
            ymin = fminf(ymin, q.y);
            ymax = fmaxf(ymax, q.y);
        }
    }
    xmin = wave_min(xmin);
    ymin = wave_min(ymin);
    xmax = wave_max(xmax);
    ymax = wave_max(ymax);
    if (lane == 0) p.chunkbb[(size_t)b * p.nchunks + c] = make_float4(xmin, ymin, xmax, ymax);
}

// ---------------------------------------------------------------------------------- tiles
struct Lists {  // per-wave LDS face list, structure of float4 arrays (conflict-free lane writes)
    float4 e0[CAP], e1[CAP], e2[CAP], zf[CAP];
};

struct Sample {  // per-lane state
    int xi, yi;
    float xp, yp;
    bool valid;
    float best_zp;
    int best_fn;
    float w0, w1, w2;
    unsigned long long hits;
    int nh;
};

__device__ __forceinline__ F4 ld4(const float4 &v) { return F4{v.x, v.y, v.z, v.w}; }

// step 4: evaluate the remembered covering faces of every sample.
__device__ __forceinline__ void eval_hits(const Lists &L, Sample &s, const RasterParams &p) {
    for (int r = 0; r < MAXH; ++r) {
        const bool act = s.nh > r;
        if (!__ballot(act)) break;
        if (act) {
            const int e = (int)((s.hits >> (8 * r)) & 0xffull);
            const float4 a0 = L.e0[e], a1 = L.e1[e], a2 = L.e2[e], a3 = L.zf[e];
            float fi[9], w[3], zp;
            face_inverse(a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, p.is, fi);
            if (fragment(fi, a3.x, a3.y, a3.z, s.xi, s.yi, p.near_, p.far_, w, zp)) {
                const int fn = __float_as_int(a3.w);
                if (wins(zp, fn, s.best_zp, s.best_fn)) {
                    s.best_zp = zp;
                    s.best_fn = fn;
                    s.w0 = w[0];
                    s.w1 = w[1];
                    s.w2 = w[2];
                }
            }
        }
    }
    s.nh = 0;
    s.hits = 0ull;
}

// step 3: every sample walks the list (LDS broadcast reads).
__device__ __forceinline__ void walk_list(const Lists &L, int count, Sample &s,
                                          const RasterParams &p) {
    __builtin_amdgcn_wave_barrier();
    for (int e = 0; e < count; ++e) {
        const F4 a0 = ld4(L.e0[e]), a1 = ld4(L.e1[e]), a2 = ld4(L.e2[e]);
        const bool in = s.valid && covers(a0, a1, a2, s.xp, s.yp);
        if (in) {
            s.hits = (s.hits << 8) | (unsigned long long)e;
            s.nh++;
        }
        if (__ballot(s.nh >= MAXH)) eval_hits(L, s, p);
    }
    eval_hits(L, s, p);  // list indices die with the list
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void append(Lists &L, int &count, bool pass, const FaceRec &r, int lane,
                                       Sample &s, const RasterParams &p) {
    const unsigned long long m = __ballot(pass);
    if (!m) return;
    const int n = __popcll(m);
    if (count + n > CAP) {
        walk_list(L, count, s, p);
        count = 0;
    }
    const int idx = count + __popcll(m & ((1ull << lane) - 1ull));
    if (pass) {
        L.e0[idx] = make_float4(r.e0.x, r.e0.y, r.e0.z, r.e0.w);
        L.e1[idx] = make_float4(r.e1.x, r.e1.y, r.e1.z, r.e1.w);
        L.e2[idx] = make_float4(r.e2.x, r.e2.y, r.e2.z, r.e2.w);
        L.zf[idx] = make_float4(r.zf.x, r.zf.y, r.zf.z, r.zf.w);
    }
    count += n;
}

// One geometric triangle (p0, p1, p2) with id g: box test against the tile, then both orientations.
__device__ __forceinline__ void bin_triangle(Lists &L, int &count, bool valid, const float4 &p0,
                                             const float4 &p1, const float4 &p2, int g, int lane,
                                             float txlo, float tylo, float txhi, float tyhi,
                                             Sample &s, const RasterParams &p) {
    const float bxmin = fminf(p0.x, fminf(p1.x, p2.x)), bxmax = fmaxf(p0.x, fmaxf(p1.x, p2.x));
    const float bymin = fminf(p0.y, fminf(p1.y, p2.y)), bymax = fmaxf(p0.y, fmaxf(p1.y, p2.y));
    const bool ov = valid && bbox_overlaps(bxmin, bymin, bxmax, bymax, txlo, tylo, txhi, tyhi);
    {
        const bool pass = ov && !back_facing(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y) &&
                          tile_may_cover(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, txlo, tylo, txhi, tyhi);
        append(L, count, pass, make_rec(p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, g),
               lane, s, p);
    }
    if (p.fill_back) {  // reversed copy: vertex order (p2, p1, p0), id g + F
        const bool pass = ov && !back_facing(p2.x, p2.y, p1.x, p1.y, p0.x, p0.y) &&
                          tile_may_cover(p2.x, p2.y, p1.x, p1.y, p0.x, p0.y, txlo, tylo, txhi, tyhi);
        append(L, count, pass,
               make_rec(p2.x, p2.y, p2.z, p1.x, p1.y, p1.z, p0.x, p0.y, p0.z, g + p.F), lane, s, p);
    }
}

template <bool IMPLICIT>
__global__ __launch_bounds__(64 * WAVES) void raster_tiles(RasterParams p) {
    __shared__ Lists lists[WAVES];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles_side = (p.is + TILE - 1) / TILE;
    const int wg_side = (tiles_side + 1) / 2;
    const int tx = (blockIdx.x % wg_side) * 2 + (wave & 1);
    const int ty = (blockIdx.x / wg_side) * 2 + (wave >> 1);
    const int b = blockIdx.y;
    if (tx >= tiles_side || ty >= tiles_side) return;  // no barriers below: a wave may leave alone
    Lists &L = lists[wave];

    Sample s;
    s.xi = tx * TILE + (lane & 7);
    s.yi = ty * TILE + (lane >> 3);
    s.valid = s.xi < p.is && s.yi < p.is;
    s.xp = sample_centre(s.xi, p.is);
    s.yp = sample_centre(s.yi, p.is);
    s.best_zp = p.far_;
    s.best_fn = INT_MAX;
    s.w0 = s.w1 = s.w2 = 0.0f;
    s.hits = 0ull;
    s.nh = 0;

    const int x_last = min(tx * TILE + TILE - 1, p.is - 1), y_last = min(ty * TILE + TILE - 1, p.is - 1);
    const float txlo = sample_centre(tx * TILE, p.is), txhi = sample_centre(x_last, p.is);
    const float tylo = sample_centre(ty * TILE, p.is), tyhi = sample_centre(y_last, p.is);

    const float4 *bb = p.chunkbb + (size_t)b * p.nchunks;
    const float4 *proj = p.proj + (size_t)b * p.N;
    int count = 0;
    for (int c0 = 0; c0 < p.nchunks; c0 += 64) {
        const int c = c0 + lane;
        bool hit = false;
        if (c < p.nchunks) {
            const float4 q = bb[c];
            hit = bbox_overlaps(q.x, q.y, q.z, q.w, txlo, tylo, txhi, tyhi);
        }
        unsigned long long mask = __ballot(hit);
        while (mask) {
            const int cc = c0 + __builtin_ctzll(mask);
            mask &= mask - 1ull;
            if (IMPLICIT) {
                const int Sm1 = p.S - 1;
                const int qi = (cc / p.nblk_side) * 8 + (lane >> 3);
                const int qj = (cc % p.nblk_side) * 8 + (lane & 7);
                const bool valid = qi < Sm1 && qj < Sm1;
                float4 v00 = make_float4(0, 0, 0, 0), v10 = v00, v01 = v00, v11 = v00;
                if (valid) {
                    v00 = proj[qi * p.S + qj];
                    v10 = proj[(qi + 1) * p.S + qj];
                    v01 = proj[qi * p.S + qj + 1];
                    v11 = proj[(qi + 1) * p.S + qj + 1];
                }
                const int q = qi * Sm1 + qj;
                bin_triangle(L, count, valid, v00, v10, v01, q, lane, txlo, tylo, txhi, tyhi, s, p);
                bin_triangle(L, count, valid, v01, v10, v11, Sm1 * Sm1 + q, lane, txlo, tylo, txhi,
                             tyhi, s, p);
            } else {
                const int f = cc * 64 + lane;
                const bool valid = f < p.F;
                float4 p0 = make_float4(0, 0, 0, 0), p1 = p0, p2 = p0;
                if (valid) {
                    p0 = proj[p.faces[3 * f + 0]];
                    p1 = proj[p.faces[3 * f + 1]];
                    p2 = proj[p.faces[3 * f + 2]];
                }
                bin_triangle(L, count, valid, p0, p1, p2, f, lane, txlo, tylo, txhi, tyhi, s, p);
            }
        }
    }
    walk_list(L, count, s, p);

    // ---- epilogue: saved maps (unflipped raster), flip + average pooling
    const bool bg = s.best_fn == INT_MAX;
    const float d = bg ? p.far_ : s.best_zp;
    if (p.face_idx && s.valid) {
        const size_t si = ((size_t)b * p.is + s.yi) * p.is + s.xi;
        p.face_idx[si] = bg ? -1 : s.best_fn;
        p.bary[3 * si + 0] = s.w0;
        p.bary[3 * si + 1] = s.w1;
        p.bary[3 * si + 2] = s.w2;
    }
    if (p.ssaa == 2) {
        const float d_r = __shfl_down(d, 1), d_u = __shfl_down(d, 8), d_ur = __shfl_down(d, 9);
        if (s.valid && !(lane & 1) && !((lane >> 3) & 1)) {
            // flipped row 2r = raster row yi+1 (upper), flipped row 2r+1 = raster row yi
            const float sum = ((d_u + d_ur) + d) + d_r;
            p.depth_out[((size_t)b * p.S + (p.S - 1 - s.yi / 2)) * p.S + s.xi / 2] = sum / 4.0f;
        }
    } else if (s.valid) {
        p.depth_out[((size_t)b * p.S + (p.S - 1 - s.yi)) * p.S + s.xi] = d;
    }
}

// ---------------------------------------------------------------------------------- backward
struct BwdParams {
    const float *verts;
    const int32_t *faces;
    const float *grad_depth;
    const int32_t *face_idx;
    const float *bary;
    int B, N, F, S, is, ssaa;
    Cam cam;
    float *gacc;  // [B, N, 3]: first (g_u, g_v, g_z) of the projected vertices, then in place xyz
};

__global__ __launch_bounds__(256) void raster_bwd_samples(BwdParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.B * p.is * p.is) return;
    const int fn = p.face_idx[i];
    if (fn < 0) return;
    const int b = (int)(i / ((long)p.is * p.is));
    const int pn = (int)(i % ((long)p.is * p.is));
    const int yi = pn / p.is, xi = pn % p.is;
    const int fr = p.is - 1 - yi;
    const float g = p.grad_depth[((size_t)b * p.S + fr / p.ssaa) * p.S + xi / p.ssaa] /
                    (float)(p.ssaa * p.ssaa);
    if (g == 0.0f) return;  // clamped / masked pixels contribute exact zeros
    int v[3];
    const int gidx = fn % p.F;
    if (p.faces) {
        v[0] = p.faces[3 * gidx];
        v[1] = p.faces[3 * gidx + 1];
        v[2] = p.faces[3 * gidx + 2];
    } else {
        implicit_face(gidx, p.S, v);
    }
    if (fn >= p.F) {
        const int t = v[0];
        v[0] = v[2];
        v[2] = t;
    }
    float px[3], py[3], pz[3];
    for (int k = 0; k < 3; k++) {
        const float *q = p.verts + ((size_t)b * p.N + v[k]) * 3;
        project(q[0], q[1], q[2], p.cam, px[k], py[k]);
        pz[k] = q[2];
    }
    const float w[3] = {p.bary[3 * i], p.bary[3 * i + 1], p.bary[3 * i + 2]};
    float gx[3], gy[3], gz[3];
    fragment_backward(px, py, pz, w, p.is, g, gx, gy, gz);
    for (int k = 0; k < 3; k++) {
        float *dst = p.gacc + ((size_t)b * p.N + v[k]) * 3;
        unsafeAtomicAdd(dst + 0, gx[k]);
        unsafeAtomicAdd(dst + 1, gy[k]);
        unsafeAtomicAdd(dst + 2, gz[k]);
    }
}

__global__ __launch_bounds__(256) void raster_bwd_project(BwdParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.B * p.N) return;
    const float *q = p.verts + i * 3;
    float *g = p.gacc + i * 3;
    float gx, gy, gz;
    project_backward(q[0], q[1], q[2], p.cam, g[0], g[1], g[2], gx, gy, gz);
    g[0] = gx;
    g[1] = gy;
    g[2] = gz;
}

static int implicit_blocks(int S) { return (S - 1 + 7) / 8; }

static int make_cam(const float *K, float orig_size, Cam &c) {
    G2S_REQUIRE(K != nullptr, "K must be a host pointer to 9 floats");
    G2S_REQUIRE(K[6] == 0.0f && K[7] == 0.0f && K[8] == 1.0f, "K third row must be 0 0 1");
    c = Cam{K[0], K[1], K[2], K[3], K[4], K[5], orig_size};
    return G2S_OK;
}

static int check_shape(const int32_t *faces, int B, int N, int F, int S, int ssaa) {
    G2S_REQUIRE(B > 0 && N > 0 && F > 0 && S > 0, "B, n_verts, n_faces, S must be positive");
    G2S_REQUIRE(ssaa == 1 || ssaa == 2, "ssaa must be 1 or 2 (got %d)", ssaa);
    G2S_REQUIRE(S * ssaa <= 8 * 255, "raster side %d too large", S * ssaa);
    if (!faces) {
        G2S_REQUIRE(S >= 2 && N == S * S && F == 2 * (S - 1) * (S - 1),
                    "implicit topology needs n_verts == S*S and n_faces == 2*(S-1)^2 "
                    "(got n_verts=%d n_faces=%d S=%d)", N, F, S);
    }
    G2S_REQUIRE((long)F * 2 < INT_MAX, "too many faces");
    return G2S_OK;
}

}  // namespace g2s

using namespace g2s;

extern "C" size_t g2s_raster_workspace_bytes(int B, int n_verts, int n_faces, int S) {
    if (B <= 0 || n_verts <= 0 || n_faces <= 0 || S <= 0) return 0;
    const long nb = implicit_blocks(S);
    const long nchunks = (nb * nb > (n_faces + 63) / 64) ? nb * nb : (n_faces + 63) / 64;
    return (size_t)B * ((size_t)n_verts + (size_t)nchunks) * sizeof(float4) + 256;
}

extern "C" int g2s_raster_depth_fwd(const float *verts, const int32_t *faces, int B, int n_verts,
                                    int n_faces, int S, const float *K, float orig_size, int ssaa,
                                    int fill_back, float near_, float far_, float *depth_out,
                                    int32_t *face_idx_out, float *bary_out, void *workspace,
                                    size_t workspace_bytes, g2s_stream_t stream) {
    G2S_REQUIRE(verts && depth_out && workspace, "verts, depth_out and workspace must not be NULL");
    G2S_REQUIRE((face_idx_out == nullptr) == (bary_out == nullptr),
                "face_idx_out and bary_out must both be given or both be NULL");
    int rc = check_shape(faces, B, n_verts, n_faces, S, ssaa);
    if (rc) return rc;
    if (workspace_bytes < g2s_raster_workspace_bytes(B, n_verts, n_faces, S))
        return fail(G2S_ERR_WORKSPACE, "workspace too small: %zu < %zu", workspace_bytes,
                    g2s_raster_workspace_bytes(B, n_verts, n_faces, S));
    RasterParams p{};
    rc = make_cam(K, orig_size, p.cam);
    if (rc) return rc;
    p.verts = verts;
    p.faces = faces;
    p.B = B;
    p.N = n_verts;
    p.F = n_faces;
    p.S = S;
    p.ssaa = ssaa;
    p.is = S * ssaa;
    p.fill_back = fill_back ? 1 : 0;
    p.near_ = near_;
    p.far_ = far_;
    p.depth_out = depth_out;
    p.face_idx = face_idx_out;
    p.bary = bary_out;
    uintptr_t base = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    p.proj = reinterpret_cast<float4 *>(base);
    p.chunkbb = p.proj + (size_t)B * n_verts;
    hipStream_t st = as_stream(stream);
    const int tiles_side = (p.is + TILE - 1) / TILE;
    const int wg_side = (tiles_side + 1) / 2;
    if (!faces) {
        p.nblk_side = implicit_blocks(S);
        p.nchunks = p.nblk_side * p.nblk_side;
        raster_setup_implicit<<<dim3(p.nchunks, B), 64, 0, st>>>(p);
        raster_tiles<true><<<dim3(wg_side * wg_side, B), 64 * WAVES, 0, st>>>(p);
    } else {
        p.nblk_side = 0;
        p.nchunks = (n_faces + 63) / 64;
        raster_project<<<cdiv((long)B * n_verts, 256), 256, 0, st>>>(p);
        raster_chunk_boxes<<<dim3(p.nchunks, B), 64, 0, st>>>(p);
        raster_tiles<false><<<dim3(wg_side * wg_side, B), 64 * WAVES, 0, st>>>(p);
    }
    return check_launch("g2s_raster_depth_fwd");
}

extern "C" int g2s_raster_depth_bwd(const float *verts, const int32_t *faces,
                                    const float *grad_depth, const int32_t *face_idx,
                                    const float *bary, int B, int n_verts, int n_faces, int S,
                                    const float *K, float orig_size, int ssaa, float *grad_verts,
                                    g2s_stream_t stream) {
    G2S_REQUIRE(verts && grad_depth && face_idx && bary && grad_verts, "NULL pointer argument");
    int rc = check_shape(faces, B, n_verts, n_faces, S, ssaa);
    if (rc) return rc;
    BwdParams p{};
    rc = make_cam(K, orig_size, p.cam);
    if (rc) return rc;
    p.verts = verts;
    p.faces = faces;
    p.grad_depth = grad_depth;
    p.face_idx = face_idx;
    p.bary = bary;
    p.B = B;
    p.N = n_verts;
    p.F = n_faces;
    p.S = S;
    p.ssaa = ssaa;
    p.is = S * ssaa;
    p.gacc = grad_verts;
    hipStream_t st = as_stream(stream);
    if (hipMemsetAsync(grad_verts, 0, (size_t)B * n_verts * 3 * sizeof(float), st) != hipSuccess)
        return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(grad_verts) failed");
    raster_bwd_samples<<<cdiv((long)B * p.is * p.is, 256), 256, 0, st>>>(p);
    raster_bwd_project<<<cdiv((long)B * n_verts, 256), 256, 0, st>>>(p);
    return check_launch("g2s_raster_depth_bwd");
}
