// raster.hip — differentiable depth rasterizer for gfx950 (MI355X), behind
// g2s_raster_depth_fwd / g2s_raster_depth_bwd (include/g2s.h).
//
// Replaces neural_renderer.Renderer.render_depth as used by GAN2Shape/renderer/renderer.py:116-125
// (semantics: SURVEY.md Appendix A).  The reference's kernel tests every one of the 64 516 faces at
// every one of the 65 536 raster samples (4.2e9 tests per image).  Here:
//
//   setup   one wavefront per 8x8-quad mesh block (implicit grid topology) or per 64 consecutive
//           faces (explicit topology): project vertices once, reduce the block's bounding box.
//   tiles   one workgroup of 4 wavefronts per 8x8-sample tile (lane = sample); the surviving mesh
//           blocks are dealt round-robin to the 4 waves, each with its own LDS list and queue, and
//           the waves meet in one shared per-sample minimum:
//             1. lanes test chunk boxes against the tile, wave-ballot the survivors;
//             2. per surviving chunk, lane = quad / face: front-face test + face box test + exact
//                (conservative) triangle-vs-tile edge test, ballot
//                + mbcnt compaction of the surviving oriented faces into a per-wave LDS list;
//             3. lane = list entry: inverse vertex matrix once per listed face; then lane = sample:
//                walk the list with LDS broadcast reads, three edge inequalities per face, and
//                push covering (sample, face) pairs into an LDS ring (ballot + mbcnt);
//             4. lane = queued fragment, 64 at a time whatever sample they belong to: clamped
//                barycentrics, perspective-correct depth, near/far; the lexicographic
//                (depth, face id) minimum per sample is one ds_min_u64 on an order-preserving key;
//             5. epilogue fused: winner's weights recomputed once, vertical flip + 2x2 average
//                pooling through shuffles.
//           ~70 candidate faces per tile instead of 64 516.
//   bwd     lane = raster sample of an 8x8 tile: analytic gradient to the three projected vertices,
//           neighbouring samples of the same face merged by shuffles, then float atomics; one
//           thread per vertex for the projection backward.
//
// Compiled with -ffp-contract=off: results equal oracle/raster_body.inc bit for bit up to the
// association order of the 2x2 average.
#include "g2s_common.h"
#include "raster_core.h"
#include <limits.h>
#include <stdlib.h>

namespace g2s {

constexpr int TILE = 8;      // samples per tile side (64 samples = one wavefront)
constexpr int CAP = 64;      // LDS face-list capacity per wavefront
constexpr int QCAP = 128;    // fragment queue ring (drained at 64, refilled by at most 64)
constexpr int WAVES = 4;     // wavefronts per workgroup, all working on one tile

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
struct Lists {  // per-wave LDS state, structure of float4 arrays (conflict-free lane writes)
    float4 e0[CAP], e1[CAP], e2[CAP], zf[CAP];  // coverage edges; vertex depths + face id
    float4 fa[CAP], fb[CAP];                    // inverse vertex matrix entries 0-3, 4-7
    float fc[CAP];                              // inverse vertex matrix entry 8
    unsigned short queue[QCAP];                 // pending fragments: list entry << 6 | sample
};

struct Tile {  // per-wave state; count/head/qn are wave-uniform
    int tx, ty, lane;
    float xp, yp;  // this lane's sample centre
    bool valid;    // sample inside the raster
    int count;     // faces in the list
    int head, qn;  // fragment queue ring: first pending entry, number pending
};

__device__ __forceinline__ F4 ld4(const float4 &v) { return F4{v.x, v.y, v.z, v.w}; }

__device__ __forceinline__ int lane_prefix(unsigned long long m) {  // set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// Order-preserving float -> uint map, so that the lexicographic (depth, face id) minimum is one
// 64-bit integer minimum (ds_min_u64).
__device__ __forceinline__ unsigned depth_key(float z) {
    const unsigned u = __float_as_uint(z + 0.0f);  // -0 -> +0
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_depth(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// step 4: lane = one pending (sample, face) fragment: clamped barycentrics, perspective-correct
// depth, near/far rejection, then the packed minimum into the sample's slot.
__device__ __forceinline__ void eval_queue(Lists &L, unsigned long long *best, const Tile &t, int n,
                                           const RasterParams &p) {
    __builtin_amdgcn_wave_barrier();
    if (t.lane < n) {
        const unsigned pr = L.queue[(t.head + t.lane) & (QCAP - 1)];
        const int sl = pr & 63, e = pr >> 6;
        const float4 fa = L.fa[e], fb = L.fb[e], z = L.zf[e];
        const float fi[9] = {fa.x, fa.y, fa.z, fa.w, fb.x, fb.y, fb.z, fb.w, L.fc[e]};
        float w[3], zp;
        if (fragment(fi, z.x, z.y, z.z, t.tx * TILE + (sl & 7), t.ty * TILE + (sl >> 3), p.near_,
                     p.far_, w, zp) &&
            zp == zp) {
            const unsigned long long key =
                ((unsigned long long)depth_key(zp) << 32) | (unsigned)__float_as_int(z.w);
            atomicMin(&best[sl], key);
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// steps 3+4 for the current list: lane = list entry computes the inverse vertex matrix once per
// face, then lane = sample walks the list (LDS broadcast reads, three edge inequalities per face)
// and queues its covering faces; the queue is drained 64 fragments at a time.
__device__ __forceinline__ void walk_list(Lists &L, unsigned long long *best, Tile &t,
                                          const RasterParams &p) {
    __builtin_amdgcn_wave_barrier();
    const int count = t.count;
    if (count == 0) return;
    if (t.lane < count) {
        const float4 a0 = L.e0[t.lane], a1 = L.e1[t.lane], a2 = L.e2[t.lane];
        float fi[9];
        face_inverse(a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, p.is, fi);
        L.fa[t.lane] = make_float4(fi[0], fi[1], fi[2], fi[3]);
        L.fb[t.lane] = make_float4(fi[4], fi[5], fi[6], fi[7]);
        L.fc[t.lane] = fi[8];
    }
    __builtin_amdgcn_wave_barrier();
    F4 a0 = ld4(L.e0[0]), a1 = ld4(L.e1[0]), a2 = ld4(L.e2[0]);
    for (int e = 0; e < count; ++e) {
        const int en = e + 1 < CAP ? e + 1 : e;  // next entry (stale data past count is harmless)
        const F4 n0 = ld4(L.e0[en]), n1 = ld4(L.e1[en]), n2 = ld4(L.e2[en]);
        const bool in = t.valid && covers(a0, a1, a2, t.xp, t.yp);
        const unsigned long long m = __ballot(in);
        if (m) {
            if (in) L.queue[(t.head + t.qn + lane_prefix(m)) & (QCAP - 1)] = (unsigned short)((e << 6) | t.lane);
            t.qn += __popcll(m);
            if (t.qn >= 64) {
                eval_queue(L, best, t, 64, p);
                t.head = (t.head + 64) & (QCAP - 1);
                t.qn -= 64;
            }
        }
        a0 = n0;
        a1 = n1;
        a2 = n2;
    }
    if (t.qn) {  // list indices die with the list
        eval_queue(L, best, t, t.qn, p);
        t.head = (t.head + t.qn) & (QCAP - 1);
        t.qn = 0;
    }
    t.count = 0;
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void append(Lists &L, unsigned long long *best, Tile &t, bool pass,
                                       const FaceRec &r, const RasterParams &p) {
    const unsigned long long m = __ballot(pass);
    if (!m) return;
    const int n = __popcll(m);
    if (t.count + n > CAP) walk_list(L, best, t, p);
    const int idx = t.count + lane_prefix(m);
    if (pass) {
        L.e0[idx] = make_float4(r.e0.x, r.e0.y, r.e0.z, r.e0.w);
        L.e1[idx] = make_float4(r.e1.x, r.e1.y, r.e1.z, r.e1.w);
        L.e2[idx] = make_float4(r.e2.x, r.e2.y, r.e2.z, r.e2.w);
        L.zf[idx] = make_float4(r.zf.x, r.zf.y, r.zf.z, r.zf.w);
    }
    t.count += n;
}

// One geometric triangle (p0, p1, p2) with id g.  The face (p0, p1, p2) keeps id g; with fill_back
// its reversed copy (p2, p1, p0) has id g + F.  kernel_2 drops back-facing orientations, so of the
// two at most one survives unless both "return if backside" tests fail together (zero signed area
// in fp32) — that case takes the second, wave-uniformly rare, append.
__device__ __forceinline__ void bin_triangle(Lists &L, unsigned long long *best, Tile &t, bool valid,
                                             const float4 &p0, const float4 &p1, const float4 &p2,
                                             int g, float txlo, float tylo, float txhi, float tyhi,
                                             const RasterParams &p) {
    const float bxmin = fminf(p0.x, fminf(p1.x, p2.x)), bxmax = fmaxf(p0.x, fmaxf(p1.x, p2.x));
    const float bymin = fminf(p0.y, fminf(p1.y, p2.y)), bymax = fmaxf(p0.y, fmaxf(p1.y, p2.y));
    const bool ov = valid && bbox_overlaps(bxmin, bymin, bxmax, bymax, txlo, tylo, txhi, tyhi);
    const bool fwd = !back_facing(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y);
    const bool rev = p.fill_back && !back_facing(p2.x, p2.y, p1.x, p1.y, p0.x, p0.y);
    {
        const float4 a = fwd ? p0 : p2, c = fwd ? p2 : p0;
        const bool pass = ov && (fwd || rev) &&
                          tile_may_cover(a.x, a.y, p1.x, p1.y, c.x, c.y, txlo, tylo, txhi, tyhi);
        append(L, best, t, pass,
               make_rec(a.x, a.y, a.z, p1.x, p1.y, p1.z, c.x, c.y, c.z, fwd ? g : g + p.F), p);
    }
    const bool both = ov && fwd && rev;
    if (__ballot(both)) {
        const bool pass = both && tile_may_cover(p2.x, p2.y, p1.x, p1.y, p0.x, p0.y, txlo, tylo, txhi, tyhi);
        append(L, best, t, pass,
               make_rec(p2.x, p2.y, p2.z, p1.x, p1.y, p1.z, p0.x, p0.y, p0.z, g + p.F), p);
    }
}

// Vertices of this lane's quad (implicit: v00, v10, v01, v11) or face (explicit: p0, p1, p2) of chunk cc.
struct ChunkVerts {
    float4 v[4];
    bool valid;
    int id;
};

template <bool IMPLICIT>
__device__ __forceinline__ ChunkVerts load_chunk(const RasterParams &p, const float4 *proj, int cc,
                                                 int lane) {
    ChunkVerts c;
    c.v[0] = c.v[1] = c.v[2] = c.v[3] = make_float4(0, 0, 0, 0);
    if (IMPLICIT) {
        const int Sm1 = p.S - 1;
        const int qi = (cc / p.nblk_side) * 8 + (lane >> 3);
        const int qj = (cc % p.nblk_side) * 8 + (lane & 7);
        c.valid = qi < Sm1 && qj < Sm1;
        c.id = qi * Sm1 + qj;
        if (c.valid) {
            c.v[0] = proj[qi * p.S + qj];
            c.v[1] = proj[(qi + 1) * p.S + qj];
            c.v[2] = proj[qi * p.S + qj + 1];
            c.v[3] = proj[(qi + 1) * p.S + qj + 1];
        }
    } else {
        const int f = cc * 64 + lane;
        c.valid = f < p.F;
        c.id = f;
        if (c.valid) {
            c.v[0] = proj[p.faces[3 * f + 0]];
            c.v[1] = proj[p.faces[3 * f + 1]];
            c.v[2] = proj[p.faces[3 * f + 2]];
        }
    }
    return c;
}

// SPLIT = 4: the workgroup's 4 waves share one tile (few tiles in flight: shortens the longest
// serial chain); SPLIT = 1: one tile per wave, 2x2 tiles per workgroup (many tiles in flight: no
// duplicated per-wave prologue).
template <bool IMPLICIT, int SPLIT>
__global__ __launch_bounds__(64 * WAVES) void raster_tiles(RasterParams p) {
    __shared__ Lists lists[WAVES];
    __shared__ unsigned long long bests[WAVES / SPLIT][64];  // per sample: packed (depth key, face id) minimum
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles_side = (p.is + TILE - 1) / TILE;
    Tile t;
    t.lane = lane;
    int part;  // this wave's share of the surviving chunks: rank mod SPLIT
    if (SPLIT == WAVES) {
        // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous band
        // of tiles so that neighbouring tiles find the shared mesh blocks in the same L2
        const int ntiles = tiles_side * tiles_side;
        int tile = blockIdx.x;
        if ((ntiles & 7) == 0) tile = (tile & 7) * (ntiles >> 3) + (tile >> 3);
        t.tx = tile % tiles_side;
        t.ty = tile / tiles_side;
        part = wave;
    } else {
        const int wg_side = (tiles_side + 1) / 2;
        t.tx = (blockIdx.x % wg_side) * 2 + (wave & 1);
        t.ty = (blockIdx.x / wg_side) * 2 + (wave >> 1);
        part = 0;
        if (t.tx >= tiles_side || t.ty >= tiles_side) return;  // no workgroup barrier in this variant
    }
    const int b = blockIdx.y;
    Lists &L = lists[wave];
    unsigned long long *best = bests[SPLIT == WAVES ? 0 : wave];

    const int xi = t.tx * TILE + (lane & 7), yi = t.ty * TILE + (lane >> 3);
    t.valid = xi < p.is && yi < p.is;
    t.xp = sample_centre(xi, p.is);
    t.yp = sample_centre(yi, p.is);
    t.count = t.head = t.qn = 0;
    if (part == 0) best[lane] = ((unsigned long long)depth_key(p.far_) << 32) | (unsigned)INT_MAX;
    if (SPLIT == WAVES) __syncthreads();

    const int x_last = min(t.tx * TILE + TILE - 1, p.is - 1), y_last = min(t.ty * TILE + TILE - 1, p.is - 1);
    const float txlo = sample_centre(t.tx * TILE, p.is), txhi = sample_centre(x_last, p.is);
    const float tylo = sample_centre(t.ty * TILE, p.is), tyhi = sample_centre(y_last, p.is);

    const float4 *bb = p.chunkbb + (size_t)b * p.nchunks;
    const float4 *proj = p.proj + (size_t)b * p.N;
    int rank = 0;  // running index of surviving chunks: chunk k belongs to the wave with part == k mod SPLIT
    for (int c0 = 0; c0 < p.nchunks; c0 += 64) {
        const int c = c0 + lane;
        bool hit = false;
        if (c < p.nchunks) {
            const float4 q = bb[c];
            hit = bbox_overlaps(q.x, q.y, q.z, q.w, txlo, tylo, txhi, tyhi);
        }
        const unsigned long long all = __ballot(hit);
        // keep the bits whose rank (rank + number of lower set bits) is congruent to this wave
        unsigned long long mask = all;
        if (SPLIT > 1) {
            mask = 0ull;
            unsigned long long m = all;
            int k = rank;
            while (m) {
                const unsigned long long low = m & (0ull - m);
                if ((k & (SPLIT - 1)) == part) mask |= low;
                m ^= low;
                ++k;
            }
            rank = k;
        }
        if (!mask) continue;
        // the vertices of the next surviving chunk are requested before the current one is binned
        ChunkVerts cur = load_chunk<IMPLICIT>(p, proj, c0 + __builtin_ctzll(mask), lane);
        mask &= mask - 1ull;
        while (true) {
            const bool more = mask != 0ull;
            ChunkVerts nxt = cur;
            if (more) {
                nxt = load_chunk<IMPLICIT>(p, proj, c0 + __builtin_ctzll(mask), lane);
                mask &= mask - 1ull;
            }
            if (IMPLICIT) {
                const int Sm1 = p.S - 1;
                bin_triangle(L, best, t, cur.valid, cur.v[0], cur.v[1], cur.v[2], cur.id, txlo, tylo,
                             txhi, tyhi, p);
                bin_triangle(L, best, t, cur.valid, cur.v[2], cur.v[1], cur.v[3], Sm1 * Sm1 + cur.id,
                             txlo, tylo, txhi, tyhi, p);
            } else {
                bin_triangle(L, best, t, cur.valid, cur.v[0], cur.v[1], cur.v[2], cur.id, txlo, tylo,
                             txhi, tyhi, p);
            }
            if (!more) break;
            cur = nxt;
        }
    }
    walk_list(L, best, t, p);
    if (SPLIT == WAVES) {
        __syncthreads();
        if (part != 0) return;
    }

    // ---- epilogue: saved maps (unflipped raster), flip + average pooling
    const unsigned long long win = best[lane];
    const int best_fn = (int)(unsigned)(win & 0xffffffffull);
    const bool bg = best_fn == INT_MAX;
    const float d = bg ? p.far_ : key_depth((unsigned)(win >> 32));
    if (p.face_idx && t.valid) {
        // weights of the winning fragment: same arithmetic on the same inputs as when it was queued
        float w[3] = {0.0f, 0.0f, 0.0f};
        if (!bg) {
            int v[3];
            const int g = best_fn % p.F;
            if (IMPLICIT) {
                implicit_face(g, p.S, v);
            } else {
                v[0] = p.faces[3 * g];
                v[1] = p.faces[3 * g + 1];
                v[2] = p.faces[3 * g + 2];
            }
            if (best_fn >= p.F) {
                const int tmp = v[0];
                v[0] = v[2];
                v[2] = tmp;
            }
            const float4 q0 = proj[v[0]], q1 = proj[v[1]], q2 = proj[v[2]];
            float fi[9], zp;
            face_inverse(q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, p.is, fi);
            fragment(fi, q0.z, q1.z, q2.z, xi, yi, p.near_, p.far_, w, zp);
        }
        const size_t si = ((size_t)b * p.is + yi) * p.is + xi;
        p.face_idx[si] = bg ? -1 : best_fn;
        p.bary[3 * si + 0] = w[0];
        p.bary[3 * si + 1] = w[1];
        p.bary[3 * si + 2] = w[2];
    }
    if (p.ssaa == 2) {
        const float d_r = __shfl_down(d, 1), d_u = __shfl_down(d, 8), d_ur = __shfl_down(d, 9);
        if (t.valid && !(lane & 1) && !((lane >> 3) & 1)) {
            // flipped row 2r = raster row yi+1 (upper), flipped row 2r+1 = raster row yi
            const float sum = ((d_u + d_ur) + d) + d_r;
            p.depth_out[((size_t)b * p.S + (p.S - 1 - yi / 2)) * p.S + xi / 2] = sum / 4.0f;
        }
    } else if (t.valid) {
        p.depth_out[((size_t)b * p.S + (p.S - 1 - yi)) * p.S + xi] = d;
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
    long long *gfix;  // deterministic mode: the same sums as 2^-40 fixed point (integer adds commute)
};

// Deterministic mode accumulates in 64-bit fixed point (LDS and global integer atomics): the sums no
// longer depend on the order the waves arrive in.  Resolution 2^-40 ~ 9e-13, range +-8.4e6 per
// component; a non-finite contribution is not representable and is dropped.
constexpr float FIX_SCALE = 1099511627776.0f;  // 2^40
__device__ __forceinline__ long long to_fix(float v) { return __float2ll_rn(v * FIX_SCALE); }
__device__ __forceinline__ float from_fix(long long v) { return (float)((double)v * (1.0 / 1099511627776.0)); }
__device__ __forceinline__ void acc_add(float *dst, float v) { unsafeAtomicAdd(dst, v); }
__device__ __forceinline__ void acc_add(long long *dst, float v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)to_fix(v));
}
__device__ __forceinline__ void acc_add(long long *dst, long long v) {
    atomicAdd(reinterpret_cast<unsigned long long *>(dst), (unsigned long long)v);
}
__device__ __forceinline__ void lds_add(float *dst, float v) { atomicAdd(dst, v); }
__device__ __forceinline__ void lds_add(long long *dst, float v) { acc_add(dst, v); }

// One wave per 8x8-sample tile (4 tiles per workgroup).  The 2x2 super-samples of a pixel mostly hit
// the same face: before the atomics, x- and then y-neighbours with the same winning face merge
// their nine partial gradients through shuffles, so a face's vertices receive one atomic triple per
// merged group instead of one per sample.
template <typename ACC>   // float (default) or long long (deterministic mode, fixed point)
__global__ __launch_bounds__(256) void raster_bwd_samples(BwdParams p) {
    ACC *const gout = sizeof(ACC) == 8 ? reinterpret_cast<ACC *>(p.gfix) : reinterpret_cast<ACC *>(p.gacc);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tiles_side = (p.is + TILE - 1) / TILE;
    const int tile = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (tile >= tiles_side * tiles_side) return;  // whole wave
    const int xi = (tile % tiles_side) * TILE + (lane & 7), yi = (tile / tiles_side) * TILE + (lane >> 3);
    const bool inside = xi < p.is && yi < p.is;
    const long i = ((long)b * p.is + (inside ? yi : 0)) * p.is + (inside ? xi : 0);
    int fn = inside ? p.face_idx[i] : -1;
    float g = 0.0f;
    if (fn >= 0) {
        const int fr = p.is - 1 - yi;
        g = p.grad_depth[((size_t)b * p.S + fr / p.ssaa) * p.S + xi / p.ssaa] / (float)(p.ssaa * p.ssaa);
        if (g == 0.0f) fn = -1;  // clamped / masked pixels contribute exact zeros
    }
    int v[3] = {0, 0, 0};
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (fn >= 0) {
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
            acc[3 * k] = gx[k];
            acc[3 * k + 1] = gy[k];
            acc[3 * k + 2] = gz[k];
        }
    }
#pragma unroll
    for (int step = 0; step < 2; step++) {
        const int m = step ? 8 : 1;
        const int pfn = __shfl_xor(fn, m);
        const bool same = fn >= 0 && pfn == fn;
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const float o = __shfl_xor(acc[k], m);
            if (same) acc[k] += o;
        }
        if (same && (lane & m)) fn = -1;  // the upper lane of a merged pair retires
    }
    // per-wave LDS hash table keyed by vertex id: neighbouring faces of a tile share vertices, so
    // the tile's contributions are summed with LDS atomics and each distinct vertex reaches global
    // memory once (three float atomics) instead of once per merged sample group and corner
    constexpr int HS = 128;
    __shared__ int hkey[4][HS];
    __shared__ ACC hval[4][HS][3];
    for (int i2 = lane; i2 < HS; i2 += 64) {
        hkey[wave][i2] = -1;
        hval[wave][i2][0] = hval[wave][i2][1] = hval[wave][i2][2] = (ACC)0;
    }
    __builtin_amdgcn_wave_barrier();
    if (fn >= 0) {
        for (int k = 0; k < 3; k++) {
            int slot = (v[k] * 0x9E3779B1u) >> 25;  // top 7 bits: 0 .. HS-1
            bool done = false;
            for (int probe = 0; probe < 16 && !done; probe++) {
                const int old = atomicCAS(&hkey[wave][slot], -1, v[k]);
                if (old == -1 || old == v[k]) {
                    lds_add(&hval[wave][slot][0], acc[3 * k]);
                    lds_add(&hval[wave][slot][1], acc[3 * k + 1]);
                    lds_add(&hval[wave][slot][2], acc[3 * k + 2]);
                    done = true;
                }
                slot = (slot + 1) & (HS - 1);
            }
            if (!done) {  // crowded table (many distinct vertices in one tile): straight to memory
                ACC *dst = gout + ((size_t)b * p.N + v[k]) * 3;
                acc_add(dst + 0, acc[3 * k]);
                acc_add(dst + 1, acc[3 * k + 1]);
                acc_add(dst + 2, acc[3 * k + 2]);
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i2 = lane; i2 < HS; i2 += 64) {
        const int key = hkey[wave][i2];
        if (key >= 0) {
            ACC *dst = gout + ((size_t)b * p.N + key) * 3;
            acc_add(dst + 0, hval[wave][i2][0]);
            acc_add(dst + 1, hval[wave][i2][1]);
            acc_add(dst + 2, hval[wave][i2][2]);
        }
    }
}

__global__ __launch_bounds__(256) void raster_bwd_project(BwdParams p) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.B * p.N) return;
    const float *q = p.verts + i * 3;
    float *g = p.gacc + i * 3;
    float gx, gy, gz;
    if (p.gfix) {
        const long long *f = p.gfix + i * 3;
        project_backward(q[0], q[1], q[2], p.cam, from_fix(f[0]), from_fix(f[1]), from_fix(f[2]), gx, gy, gz);
    } else {
        project_backward(q[0], q[1], q[2], p.cam, g[0], g[1], g[2], gx, gy, gz);
    }
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

// g2s_raster_tune: per-thread override of the tile-to-wave mapping (tools/bench_raster.py).
static thread_local int g_force_waves = 0;

extern "C" int g2s_raster_tune(int waves_per_tile) {
    G2S_REQUIRE(waves_per_tile == 0 || waves_per_tile == 1 || waves_per_tile == 4,
                "waves_per_tile must be 0 (built-in choice), 1 or 4");
    g_force_waves = waves_per_tile;
    return G2S_OK;
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
    const int ntiles = tiles_side * tiles_side, wg_side = (tiles_side + 1) / 2;
    // few tiles: all 4 waves of a workgroup on one tile; many tiles: one tile per wave
    const bool split = g_force_waves ? g_force_waves == 4 : (long)ntiles * B <= 4096;
    const dim3 grid = split ? dim3(ntiles, B) : dim3(wg_side * wg_side, B);
    if (!faces) {
        p.nblk_side = implicit_blocks(S);
        p.nchunks = p.nblk_side * p.nblk_side;
        raster_setup_implicit<<<dim3(p.nchunks, B), 64, 0, st>>>(p);
        if (split) raster_tiles<true, WAVES><<<grid, 64 * WAVES, 0, st>>>(p);
        else raster_tiles<true, 1><<<grid, 64 * WAVES, 0, st>>>(p);
    } else {
        p.nblk_side = 0;
        p.nchunks = (n_faces + 63) / 64;
        raster_project<<<cdiv((long)B * n_verts, 256), 256, 0, st>>>(p);
        raster_chunk_boxes<<<dim3(p.nchunks, B), 64, 0, st>>>(p);
        if (split) raster_tiles<false, WAVES><<<grid, 64 * WAVES, 0, st>>>(p);
        else raster_tiles<false, 1><<<grid, 64 * WAVES, 0, st>>>(p);
    }
    return check_launch("g2s_raster_depth_fwd");
}

extern "C" size_t g2s_raster_bwd_workspace_bytes(int B, int n_verts) {
    if (B <= 0 || n_verts <= 0) return 0;
    return (size_t)B * n_verts * 3 * sizeof(long long) + 256;
}

extern "C" int g2s_raster_depth_bwd_ex(const float *verts, const int32_t *faces,
                                       const float *grad_depth, const int32_t *face_idx,
                                       const float *bary, int B, int n_verts, int n_faces, int S,
                                       const float *K, float orig_size, int ssaa, float *grad_verts,
                                       void *workspace, size_t workspace_bytes, g2s_stream_t stream) {
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
    const int bw_tiles = cdiv(p.is, TILE) * cdiv(p.is, TILE);
    if (deterministic()) {
        // float atomics would make the vertex sums depend on the order the tiles finish in
        if (!workspace || workspace_bytes < g2s_raster_bwd_workspace_bytes(B, n_verts))
            return fail(G2S_ERR_WORKSPACE, "deterministic mode: the backward needs its fixed-point workspace "
                        "(g2s_raster_bwd_workspace_bytes = %zu bytes, got %zu)",
                        g2s_raster_bwd_workspace_bytes(B, n_verts), workspace ? workspace_bytes : (size_t)0);
        p.gfix = reinterpret_cast<long long *>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
        if (!precleared() && hipMemsetAsync(p.gfix, 0, (size_t)B * n_verts * 3 * sizeof(long long), st) != hipSuccess)
            return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(workspace) failed");
        raster_bwd_samples<long long><<<dim3(cdiv(bw_tiles, 4), B), 256, 0, st>>>(p);
    } else {
        if (!precleared() && hipMemsetAsync(grad_verts, 0, (size_t)B * n_verts * 3 * sizeof(float), st) != hipSuccess)
            return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(grad_verts) failed");
        raster_bwd_samples<float><<<dim3(cdiv(bw_tiles, 4), B), 256, 0, st>>>(p);
    }
    raster_bwd_project<<<cdiv((long)B * n_verts, 256), 256, 0, st>>>(p);
    return check_launch("g2s_raster_depth_bwd");
}

extern "C" int g2s_raster_depth_bwd(const float *verts, const int32_t *faces,
                                    const float *grad_depth, const int32_t *face_idx,
                                    const float *bary, int B, int n_verts, int n_faces, int S,
                                    const float *K, float orig_size, int ssaa, float *grad_verts,
                                    g2s_stream_t stream) {
    return g2s_raster_depth_bwd_ex(verts, faces, grad_depth, face_idx, bary, B, n_verts, n_faces, S, K,
                                   orig_size, ssaa, grad_verts, nullptr, 0, stream);
}
