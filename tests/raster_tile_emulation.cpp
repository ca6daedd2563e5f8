// Host emulation of the tile algorithm of gan-2d-to-3d_amd/csrc/raster.hip, built by
// tests/test_raster_core_cpu.py with g++ -ffp-contract=off.  It uses the SAME arithmetic header
// (raster_core.h) as the HIP kernels and the same two-level culling / candidate list / hit list /
// lexicographic-minimum structure, with the 64 lanes of a wavefront run as a serial loop.  It lets
// the CPU test-suite check the culling logic and the arithmetic against the oracle; it is test
// code, not a product path (the product has no CPU path).
#include <climits>
#include <cmath>
#include <cstring>
#include <vector>
#include "../gan-2d-to-3d_amd/csrc/raster_core.h"

using namespace g2s;

namespace {
struct P4 { float x, y, z; };
struct Box { float xmin, ymin, xmax, ymax; };

void grow(Box &b, const P4 &p) {
    b.xmin = fminf(b.xmin, p.x); b.xmax = fmaxf(b.xmax, p.x);
    b.ymin = fminf(b.ymin, p.y); b.ymax = fmaxf(b.ymax, p.y);
}
}  // namespace

extern "C" int g2s_emul_render_depth(const float *verts, const int *faces, int B, int N, int F, int S,
                                     const float *K, float orig_size, int ssaa, int fill_back,
                                     float near_, float far_, float *depth_out, int *face_idx,
                                     float *bary, long *stats /*[4]: candidates, fragments, max candidates per tile, max chunk hits per tile*/) {
    const int is = S * ssaa;
    Cam cam{K[0], K[1], K[2], K[3], K[4], K[5], orig_size};
    const bool implicit = faces == nullptr;
    const int nblk = (S - 1 + 7) / 8;
    const int nchunks = implicit ? nblk * nblk : (F + 63) / 64;
    std::vector<float> dss((size_t)is * is);
    stats[0] = stats[1] = stats[2] = stats[3] = 0;
    for (int b = 0; b < B; b++) {
        std::vector<P4> proj(N);
        for (int i = 0; i < N; i++) {
            const float *v = verts + ((size_t)b * N + i) * 3;
            project(v[0], v[1], v[2], cam, proj[i].x, proj[i].y);
            proj[i].z = v[2];
        }
        // chunk -> list of geometric faces + box
        std::vector<Box> cbox(nchunks, Box{INFINITY, INFINITY, -INFINITY, -INFINITY});
        std::vector<std::vector<int>> cfaces(nchunks);
        if (implicit) {
            const int Sm1 = S - 1, Q = Sm1 * Sm1;
            for (int c = 0; c < nchunks; c++) {
                const int bi = c / nblk, bj = c % nblk;
                for (int t = 0; t < 81; t++) {
                    int i = bi * 8 + t / 9, j = bj * 8 + t % 9;
                    if (i < S && j < S) grow(cbox[c], proj[i * S + j]);
                }
                for (int l = 0; l < 64; l++) {
                    int qi = bi * 8 + (l >> 3), qj = bj * 8 + (l & 7);
                    if (qi < Sm1 && qj < Sm1) {
                        cfaces[c].push_back(qi * Sm1 + qj);
                        cfaces[c].push_back(Q + qi * Sm1 + qj);
                    }
                }
            }
        } else {
            for (int f = 0; f < F; f++) {
                cfaces[f / 64].push_back(f);
                for (int k = 0; k < 3; k++) grow(cbox[f / 64], proj[faces[3 * f + k]]);
            }
        }
        const int tiles = (is + 7) / 8;
        for (int ty = 0; ty < tiles; ty++)
            for (int tx = 0; tx < tiles; tx++) {
                const int xl = std::min(tx * 8 + 7, is - 1), yl = std::min(ty * 8 + 7, is - 1);
                const float txlo = sample_centre(tx * 8, is), txhi = sample_centre(xl, is);
                const float tylo = sample_centre(ty * 8, is), tyhi = sample_centre(yl, is);
                std::vector<FaceRec> list;
                long chunk_hits = 0;
                for (int c = 0; c < nchunks; c++) {
                    if (!bbox_overlaps(cbox[c].xmin, cbox[c].ymin, cbox[c].xmax, cbox[c].ymax, txlo,
                                       tylo, txhi, tyhi))
                        continue;
                    chunk_hits++;
                    for (int g : cfaces[c]) {
                        int v[3];
                        if (implicit) implicit_face(g, S, v);
                        else { v[0] = faces[3 * g]; v[1] = faces[3 * g + 1]; v[2] = faces[3 * g + 2]; }
                        const P4 &p0 = proj[v[0]], &p1 = proj[v[1]], &p2 = proj[v[2]];
                        Box bb{INFINITY, INFINITY, -INFINITY, -INFINITY};
                        grow(bb, p0); grow(bb, p1); grow(bb, p2);
                        if (!bbox_overlaps(bb.xmin, bb.ymin, bb.xmax, bb.ymax, txlo, tylo, txhi, tyhi))
                            continue;
                        if (!back_facing(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y) &&
                            tile_may_cover(p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, txlo, tylo, txhi, tyhi))
                            list.push_back(make_rec(p0.x, p0.y, p0.z, p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, g));
                        if (fill_back && !back_facing(p2.x, p2.y, p1.x, p1.y, p0.x, p0.y) &&
                            tile_may_cover(p2.x, p2.y, p1.x, p1.y, p0.x, p0.y, txlo, tylo, txhi, tyhi))
                            list.push_back(make_rec(p2.x, p2.y, p2.z, p1.x, p1.y, p1.z, p0.x, p0.y, p0.z, g + F));
                    }
                }
                stats[0] += (long)list.size();
                if ((long)list.size() > stats[2]) stats[2] = (long)list.size();
                if (chunk_hits > stats[3]) stats[3] = chunk_hits;
                for (int l = 0; l < 64; l++) {
                    const int xi = tx * 8 + (l & 7), yi = ty * 8 + (l >> 3);
                    if (xi >= is || yi >= is) continue;
                    const float xp = sample_centre(xi, is), yp = sample_centre(yi, is);
                    float best = far_, bw[3] = {0, 0, 0};
                    int bfn = INT_MAX;
                    // walk the list back to front: the result must not depend on the order
                    for (int e = (int)list.size() - 1; e >= 0; e--) {
                        const FaceRec &r = list[e];
                        if (!covers(r.e0, r.e1, r.e2, xp, yp)) continue;
                        stats[1]++;
                        float fi[9], w[3], zp;
                        face_inverse(r.e0.x, r.e0.y, r.e1.x, r.e1.y, r.e2.x, r.e2.y, is, fi);
                        if (!fragment(fi, r.zf.x, r.zf.y, r.zf.z, xi, yi, near_, far_, w, zp)) continue;
                        int fn;
                        memcpy(&fn, &r.zf.w, 4);
                        if (wins(zp, fn, best, bfn)) { best = zp; bfn = fn; bw[0] = w[0]; bw[1] = w[1]; bw[2] = w[2]; }
                    }
                    const size_t si = ((size_t)b * is + yi) * is + xi;
                    dss[(size_t)yi * is + xi] = bfn == INT_MAX ? far_ : best;
                    face_idx[si] = bfn == INT_MAX ? -1 : bfn;
                    for (int k = 0; k < 3; k++) bary[3 * si + k] = bw[k];
                }
            }
        for (int r = 0; r < S; r++)
            for (int c = 0; c < S; c++) {
                float out;
                if (ssaa == 2) {
                    const int yi = is - 2 - 2 * r, xi = 2 * c;
                    const float d = dss[(size_t)yi * is + xi], d_r = dss[(size_t)yi * is + xi + 1];
                    const float d_u = dss[(size_t)(yi + 1) * is + xi], d_ur = dss[(size_t)(yi + 1) * is + xi + 1];
                    out = (((d_u + d_ur) + d) + d_r) / 4.0f;
                } else {
                    out = dss[(size_t)(is - 1 - r) * is + c];
                }
                depth_out[((size_t)b * S + r) * S + c] = out;
            }
    }
    return 0;
}

// backward: same per-sample arithmetic as raster_bwd_samples + raster_bwd_project (serial adds).
extern "C" int g2s_emul_render_depth_bwd(const float *verts, const int *faces, const float *grad_depth,
                                         const int *face_idx, const float *bary, int B, int N, int F,
                                         int S, const float *K, float orig_size, int ssaa,
                                         float *grad_verts) {
    const int is = S * ssaa;
    Cam cam{K[0], K[1], K[2], K[3], K[4], K[5], orig_size};
    std::vector<double> acc((size_t)B * N * 3, 0.0);
    for (long i = 0; i < (long)B * is * is; i++) {
        const int fn = face_idx[i];
        if (fn < 0) continue;
        const int b = (int)(i / ((long)is * is)), pn = (int)(i % ((long)is * is));
        const int yi = pn / is, xi = pn % is, fr = is - 1 - yi;
        const float g = grad_depth[((size_t)b * S + fr / ssaa) * S + xi / ssaa] / (float)(ssaa * ssaa);
        if (g == 0.0f) continue;
        int v[3];
        const int gi = fn % F;
        if (faces) { v[0] = faces[3 * gi]; v[1] = faces[3 * gi + 1]; v[2] = faces[3 * gi + 2]; }
        else implicit_face(gi, S, v);
        if (fn >= F) std::swap(v[0], v[2]);
        float px[3], py[3], pz[3];
        for (int k = 0; k < 3; k++) {
            const float *q = verts + ((size_t)b * N + v[k]) * 3;
            project(q[0], q[1], q[2], cam, px[k], py[k]);
            pz[k] = q[2];
        }
        const float w[3] = {bary[3 * i], bary[3 * i + 1], bary[3 * i + 2]};
        float gx[3], gy[3], gz[3];
        fragment_backward(px, py, pz, w, is, g, gx, gy, gz);
        for (int k = 0; k < 3; k++) {
            double *d = &acc[((size_t)b * N + v[k]) * 3];
            d[0] += gx[k]; d[1] += gy[k]; d[2] += gz[k];
        }
    }
    for (long i = 0; i < (long)B * N; i++) {
        const float *q = verts + i * 3;
        project_backward(q[0], q[1], q[2], cam, (float)acc[3 * i], (float)acc[3 * i + 1],
                         (float)acc[3 * i + 2], grad_verts[3 * i], grad_verts[3 * i + 1],
                         grad_verts[3 * i + 2]);
    }
    return 0;
}
