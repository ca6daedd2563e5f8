// raster_core.h — per-vertex / per-face / per-fragment arithmetic of the depth rasterizer.
//
// Plain inline functions usable from HIP device code and from a host C++ harness
// (tests/raster_tile_emulation.cpp) so that the tile culling logic can be exercised on the CPU.
// The arithmetic mirrors, operation for operation, the CUDA kernels of the external
// neural_renderer package as recorded in SURVEY.md Appendix A (projection.py,
// forward_face_index_map kernels 1 and 2, backward_depth_map); the translation unit that
// includes this header must be compiled with -ffp-contract=off so that a*b+c is never fused.
//
// Design (not in the reference): faces are culled per 8x8-sample tile (one wavefront per tile)
// through two levels of bounding boxes, survivors are compacted into an LDS list, every sample
// tests the list with the reference's three edge inequalities, and only covering (sample, face)
// pairs run the expensive barycentric/depth arithmetic.  The winner is the lexicographic minimum
// of (depth, face id) — identical to the reference's in-order strict-< scan.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define G2S_HD __host__ __device__ __forceinline__
#else
#define G2S_HD inline
#endif

namespace g2s {

struct Cam {  // pinhole intrinsics (third row 0 0 1) + neural_renderer orig_size
    float k00, k01, k02, k10, k11, k12, orig_size;
};

struct F4 { float x, y, z, w; };

// projection.py [recalled] with R = I, t = 0, zero distortion: camera xyz -> (u_n, v_n, z).
G2S_HD void project(float x, float y, float z, const Cam &c, float &u, float &v) {
    const float eps = 1e-9f;
    float x_ = x / (z + eps);
    float y_ = y / (z + eps);
    float uu = (x_ * c.k00 + y_ * c.k01) + c.k02;
    float vv = (x_ * c.k10 + y_ * c.k11) + c.k12;
    vv = c.orig_size - vv;
    u = 2.0f * (uu - c.orig_size / 2.0f) / c.orig_size;
    v = 2.0f * (vv - c.orig_size / 2.0f) / c.orig_size;
}

// "return if backside" test of kernel_1/kernel_2 for the ordered face (p0, p1, p2).
G2S_HD bool back_facing(float x0, float y0, float x1, float y1, float x2, float y2) {
    return (y2 - y0) * (x1 - x0) < (y1 - y0) * (x2 - x0);
}

// Centre of raster sample i in normalised device coordinates: (2 i + 1 - is) / is.
G2S_HD float sample_centre(int i, int is) { return (float)((2.0 * i + 1 - is) / is); }

// Conservative overlap of a bounding box with the sample-centre range of a tile.  The box is
// grown by a small tolerance so that fp32 rounding in the edge inequalities can never accept a
// sample the culling dropped.  NaN bounds compare false -> no overlap (such faces never win).
G2S_HD bool bbox_overlaps(float bxmin, float bymin, float bxmax, float bymax, float txlo,
                          float tylo, float txhi, float tyhi) {
    float mx = bxmax > -bxmin ? bxmax : -bxmin;
    float my = bymax > -bymin ? bymax : -bymin;
    float m = mx > my ? mx : my;
    float e = 1e-5f * (1.0f + m);
    return (bxmin - e <= txhi) && (bxmax + e >= txlo) && (bymin - e <= tyhi) && (bymax + e >= tylo);
}

// Conservative triangle-vs-tile test for the ordered (front-facing) face (a, b, c): the kernel_2
// coverage test accepts a sample iff all three edge values  E_k(p) = (yp - y_k) * dx_k - (xp - x_k) * dy_k
// are >= 0.  E_k is linear in p, so its maximum over the tile's sample-centre rectangle is attained
// at a corner; if that maximum is negative (beyond a rounding margin) for any edge, no sample of
// the tile can be covered.  This removes long thin faces (depth discontinuities seen from the
// side) from every tile of their bounding box that they do not actually cross.
G2S_HD bool edge_may_reach(float x0, float y0, float dx, float dy, float txlo, float tylo,
                           float txhi, float tyhi) {
    const float yp = dx > 0.0f ? tyhi : tylo;
    const float xp = dy > 0.0f ? txlo : txhi;
    const float a = (yp - y0) * dx, b = (xp - x0) * dy;
    const float aa = a < 0.0f ? -a : a, bb = b < 0.0f ? -b : b;
    return a - b >= -1e-5f * (aa + bb) - 1e-12f;
}

G2S_HD bool tile_may_cover(float ax, float ay, float bx, float by, float cx, float cy, float txlo,
                           float tylo, float txhi, float tyhi) {
    return edge_may_reach(ax, ay, bx - ax, by - ay, txlo, tylo, txhi, tyhi) &&
           edge_may_reach(bx, by, cx - bx, cy - by, txlo, tylo, txhi, tyhi) &&
           edge_may_reach(cx, cy, ax - cx, ay - cy, txlo, tylo, txhi, tyhi);
}

// One candidate face in the tile list: the three float4 the coverage test reads, plus depths/id.
struct FaceRec {
    F4 e0;  // x0, y0, x1-x0, y1-y0
    F4 e1;  // x1, y1, x2-x1, y2-y1
    F4 e2;  // x2, y2, x0-x2, y0-y2
    F4 zf;  // z0, z1, z2, face id (int bits)
};

G2S_HD FaceRec make_rec(float x0, float y0, float z0, float x1, float y1, float z1, float x2,
                        float y2, float z2, int fn) {
    FaceRec r;
    r.e0 = {x0, y0, x1 - x0, y1 - y0};
    r.e1 = {x1, y1, x2 - x1, y2 - y1};
    r.e2 = {x2, y2, x0 - x2, y0 - y2};
    union { int i; float f; } cv;
    cv.i = fn;
    r.zf = {z0, z1, z2, cv.f};
    return r;
}

// kernel_2 "check [py, px] is inside the face".
G2S_HD bool covers(const F4 &e0, const F4 &e1, const F4 &e2, float xp, float yp) {
    return !(((yp - e0.y) * e0.z < (xp - e0.x) * e0.w) || ((yp - e1.y) * e1.z < (xp - e1.x) * e1.w) ||
             ((yp - e2.y) * e2.z < (xp - e2.x) * e2.w));
}

// kernel_1: inverse of the pixel-space vertex matrix.
G2S_HD void face_inverse(float x0, float y0, float x1, float y1, float x2, float y2, int is,
                         float fi[9]) {
    const float fis = (float)is;
    float p00 = 0.5f * (x0 * fis + fis - 1.0f), p01 = 0.5f * (y0 * fis + fis - 1.0f);
    float p10 = 0.5f * (x1 * fis + fis - 1.0f), p11 = 0.5f * (y1 * fis + fis - 1.0f);
    float p20 = 0.5f * (x2 * fis + fis - 1.0f), p21 = 0.5f * (y2 * fis + fis - 1.0f);
    float den = (p20 * (p01 - p11) + p00 * (p11 - p21) + p10 * (p21 - p01));
    fi[0] = (p11 - p21) / den;
    fi[1] = (p20 - p10) / den;
    fi[2] = (p10 * p21 - p20 * p11) / den;
    fi[3] = (p21 - p01) / den;
    fi[4] = (p00 - p20) / den;
    fi[5] = (p20 * p01 - p00 * p21) / den;
    fi[6] = (p01 - p11) / den;
    fi[7] = (p10 - p00) / den;
    fi[8] = (p00 * p11 - p10 * p01) / den;
}

G2S_HD float clamp01(float w) {  // CUDA min(max(w, 0.), 1.): NaN -> 0
    float a = (w > 0.0f) ? w : 0.0f;
    return (a < 1.0f) ? a : 1.0f;
}

// kernel_2 fragment arithmetic.  Returns false when the fragment is rejected by near/far.
G2S_HD bool fragment(const float fi[9], float z0, float z1, float z2, int xi, int yi, float near_,
                     float far_, float w[3], float &zp) {
    w[0] = fi[0] * (float)xi + fi[1] * (float)yi + fi[2];
    w[1] = fi[3] * (float)xi + fi[4] * (float)yi + fi[5];
    w[2] = fi[6] * (float)xi + fi[7] * (float)yi + fi[8];
    float w_sum = 0.0f;
    for (int k = 0; k < 3; k++) {
        w[k] = clamp01(w[k]);
        w_sum += w[k];
    }
    for (int k = 0; k < 3; k++) w[k] /= w_sum;
    zp = 1.0f / (w[0] / z0 + w[1] / z1 + w[2] / z2);
    if (zp <= near_ || far_ <= zp) return false;
    return true;
}

// Lexicographic (depth, face id) minimum == the reference's ascending scan with strict <.
G2S_HD bool wins(float zp, int fn, float best_zp, int best_fn) {
    return zp < best_zp || (zp == best_zp && fn < best_fn);
}

// Vertex ids of implicit-grid face g in [0, 2Q), Q = (S-1)^2 (renderer/utils.py:76-80):
// g < Q: faces1 of quad g = (i,j),(i+1,j),(i,j+1); g >= Q: faces2 = (i,j+1),(i+1,j),(i+1,j+1).
G2S_HD void implicit_face(int g, int S, int v[3]) {
    const int Q = (S - 1) * (S - 1);
    const int q = g < Q ? g : g - Q;
    const int i = q / (S - 1), j = q % (S - 1);
    if (g < Q) {
        v[0] = i * S + j;
        v[1] = (i + 1) * S + j;
        v[2] = i * S + j + 1;
    } else {
        v[0] = i * S + j + 1;
        v[1] = (i + 1) * S + j;
        v[2] = (i + 1) * S + j + 1;
    }
}

// backward_depth_map [recalled] for one raster sample + projection backward.
// Inputs: projected face (x,y,z per vertex), saved weights, incoming gradient g (already divided
// by the pooling factor).  Outputs gradient w.r.t. the projected vertices (gx, gy, gz per vertex).
G2S_HD void fragment_backward(const float px[3], const float py[3], const float pz[3],
                              const float w[3], int is, float g, float gpx[3], float gpy[3],
                              float gpz[3]) {
    float fi[9];
    face_inverse(px[0], py[0], px[1], py[1], px[2], py[2], is, fi);
    const float depth = 1.0f / (w[0] / pz[0] + w[1] / pz[1] + w[2] / pz[2]);
    const float depth2 = depth * depth;
    float tmp[2] = {0.0f, 0.0f};
    for (int l = 0; l < 3; l++) {
        tmp[0] += -fi[3 * l + 0] / pz[l];
        tmp[1] += -fi[3 * l + 1] / pz[l];
    }
    for (int k = 0; k < 3; k++) {
        gpz[k] = g * w[k] * depth2 / (pz[k] * pz[k]);
        gpx[k] = -g * tmp[0] * w[k] * depth2 * (float)is / 2.0f;
        gpy[k] = -g * tmp[1] * w[k] * depth2 * (float)is / 2.0f;
    }
}

// projection backward: gradient w.r.t. (u_n, v_n, z) -> gradient w.r.t. camera xyz.
G2S_HD void project_backward(float x, float y, float z, const Cam &c, float gu_n, float gv_n,
                             float gz_direct, float &gx, float &gy, float &gz) {
    const float eps = 1e-9f;
    const float gu = gu_n * (2.0f / c.orig_size);
    const float gv = -gv_n * (2.0f / c.orig_size);
    const float gx_ = gu * c.k00 + gv * c.k10;
    const float gy_ = gu * c.k01 + gv * c.k11;
    const float zz = z + eps;
    gx = gx_ / zz;
    gy = gy_ / zz;
    gz = gz_direct - (gx_ * x + gy_ * y) / (zz * zz);
}

}  // namespace g2s
