// geometry.hip — fused renderer geometry / loss glue of the GAN2Shape step (HBM-bound elementwise
// chains that the reference runs as dozens of tiny launches each, forward and again in autograd).
//
//   g2s_view_transform_*   model.py:330-335 get_view_transformation + renderer/utils.py:33-73
//                          (view[B,6] -> R = Rz Ry Rx [B,3,3], t [B,3])
//   g2s_warp_verts_*       renderer.py:74-80,64-72,90-95  depth -> d*ray -> R(X - c) + c + t
//   g2s_inv_warp_grid_*    renderer.py:97-102,82-88,110-114  depth -> R^T(d*ray - t - c) + c -> K -> [-1,1]
//   g2s_smooth_loss_*      losses.py:54-79  second-order smoothness of a [N,H,W] map
//
// Every backward is analytic; per-batch reductions (gradients of R and t) are reduced in the
// workgroup and added with one float atomic per workgroup and component.
#include <algorithm>
#include "g2s_common.h"

namespace g2s {

__device__ __forceinline__ float wave_sum_g(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum `v` over the 256 threads of the workgroup and atomically add it to *dst (thread 0).
__device__ __forceinline__ void block_atomic_add(float v, float *dst, float *red) {
    v = wave_sum_g(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(dst, red[0] + red[1] + red[2] + red[3]);
}

// ------------------------------------------------------------------ view -> (R, t)
struct ViewScale { float rot, txy, tz; };  // pi/180*xyz_rotation_range, xy_translation_range, z_translation_range

__global__ void view_transform_fwd(const float *__restrict__ view, ViewScale s, float *__restrict__ R,
                                   float *__restrict__ t, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *v = view + b * 6;
    const float ax = v[0] * s.rot, ay = v[1] * s.rot, az = v[2] * s.rot;
    const float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay), cz = cosf(az), sz = sinf(az);
    float *r = R + b * 9;
    // Rz * Ry * Rx
    r[0] = cz * cy;  r[1] = cz * sy * sx - sz * cx;  r[2] = cz * sy * cx + sz * sx;
    r[3] = sz * cy;  r[4] = sz * sy * sx + cz * cx;  r[5] = sz * sy * cx - cz * sx;
    r[6] = -sy;      r[7] = cy * sx;                 r[8] = cy * cx;
    t[b * 3 + 0] = v[3] * s.txy;
    t[b * 3 + 1] = v[4] * s.txy;
    t[b * 3 + 2] = v[5] * s.tz;
}

__global__ void view_transform_bwd(const float *__restrict__ view, ViewScale s,
                                   const float *__restrict__ gR, const float *__restrict__ gt,
                                   float *__restrict__ gview, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *v = view + b * 6;
    const float ax = v[0] * s.rot, ay = v[1] * s.rot, az = v[2] * s.rot;
    const float cx = cosf(ax), sx = sinf(ax), cy = cosf(ay), sy = sinf(ay), cz = cosf(az), sz = sinf(az);
    const float *g = gR + b * 9;
    // d/dax
    const float dax = g[1] * (cz * sy * cx + sz * sx) + g[2] * (-cz * sy * sx + sz * cx) +
                      g[4] * (sz * sy * cx - cz * sx) + g[5] * (-sz * sy * sx - cz * cx) +
                      g[7] * (cy * cx) + g[8] * (-cy * sx);
    const float day = g[0] * (-cz * sy) + g[1] * (cz * cy * sx) + g[2] * (cz * cy * cx) +
                      g[3] * (-sz * sy) + g[4] * (sz * cy * sx) + g[5] * (sz * cy * cx) +
                      g[6] * (-cy) + g[7] * (-sy * sx) + g[8] * (-sy * cx);
    const float daz = g[0] * (-sz * cy) + g[1] * (-sz * sy * sx - cz * cx) + g[2] * (-sz * sy * cx + cz * sx) +
                      g[3] * (cz * cy) + g[4] * (cz * sy * sx - sz * cx) + g[5] * (cz * sy * cx + sz * sx);
    float *o = gview + b * 6;
    o[0] = dax * s.rot;
    o[1] = day * s.rot;
    o[2] = daz * s.rot;
    o[3] = gt[b * 3 + 0] * s.txy;
    o[4] = gt[b * 3 + 1] * s.txy;
    o[5] = gt[b * 3 + 2] * s.tz;
}

// ------------------------------------------------------------------ depth -> warped vertices
// grid (ceil(P / 256), B)
__global__ __launch_bounds__(256) void warp_verts_fwd(const float *__restrict__ depth,
                                                      const float *__restrict__ rays,
                                                      const float *__restrict__ R,
                                                      const float *__restrict__ t, float rcd,
                                                      float *__restrict__ verts, int P) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float *r = R + b * 9, *tt = t + b * 3;
    const float d = depth[(size_t)b * P + p];
    const float x = rays[3 * p] * d, y = rays[3 * p + 1] * d, z = rays[3 * p + 2] * d - rcd;
    float *o = verts + ((size_t)b * P + p) * 3;
    o[0] = (x * r[0] + y * r[1] + z * r[2]) + tt[0];
    o[1] = (x * r[3] + y * r[4] + z * r[5]) + tt[1];
    o[2] = ((x * r[6] + y * r[7] + z * r[8]) + rcd) + tt[2];
}

// gRt [B, 12] (9 of R row-major, then 3 of t) must be zero-filled by the caller.
__global__ __launch_bounds__(256) void warp_verts_bwd(const float *__restrict__ depth,
                                                      const float *__restrict__ rays,
                                                      const float *__restrict__ R,
                                                      const float *__restrict__ gverts, float rcd,
                                                      float *__restrict__ gdepth,
                                                      float *__restrict__ gRt, int P, int need_rt) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float *r = R + b * 9;
    float s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // one pass when the grid covers P (the default launch); a single workgroup per item strides over
    // all pixels in deterministic mode, so every per-item sum has one fixed order
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const float *g = gverts + ((size_t)b * P + p) * 3;
        const float g0 = g[0], g1 = g[1], g2 = g[2];
        const float d = depth[(size_t)b * P + p];
        const float rx = rays[3 * p], ry = rays[3 * p + 1], rz = rays[3 * p + 2];
        const float x = rx * d, y = ry * d, z = rz * d - rcd;
        // gX = R^T g ; gd = gX . ray
        const float gx = r[0] * g0 + r[3] * g1 + r[6] * g2;
        const float gy = r[1] * g0 + r[4] * g1 + r[7] * g2;
        const float gz = r[2] * g0 + r[5] * g1 + r[8] * g2;
        gdepth[(size_t)b * P + p] = gx * rx + gy * ry + gz * rz;
        s[0] += g0 * x; s[1] += g0 * y; s[2] += g0 * z;
        s[3] += g1 * x; s[4] += g1 * y; s[5] += g1 * z;
        s[6] += g2 * x; s[7] += g2 * y; s[8] += g2 * z;
        s[9] += g0; s[10] += g1; s[11] += g2;
    }
    if (!need_rt) return;
    float *o = gRt + b * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) block_atomic_add(s[k], o + k, red);
}

// ------------------------------------------------------------------ depth -> inverse-warped 2-D grid
struct Intr { float k00, k01, k02, k10, k11, k12, sx, sy; };  // sx = 2/(W-1), sy = 2/(H-1)

__global__ __launch_bounds__(256) void inv_warp_grid_fwd(const float *__restrict__ depth,
                                                         const float *__restrict__ rays,
                                                         const float *__restrict__ R,
                                                         const float *__restrict__ t, float rcd,
                                                         Intr K, float *__restrict__ grid, int P) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float *r = R + b * 9, *tt = t + b * 3;
    const float d = depth[(size_t)b * P + p];
    const float zx = rays[3 * p] * d - tt[0], zy = rays[3 * p + 1] * d - tt[1],
                zz = (rays[3 * p + 2] * d - tt[2]) - rcd;
    // Y = R^T (Z - c) + c
    const float yx = zx * r[0] + zy * r[3] + zz * r[6];
    const float yy = zx * r[1] + zy * r[4] + zz * r[7];
    const float yz = (zx * r[2] + zy * r[5] + zz * r[8]) + rcd;
    const float a = yx / yz, bq = yy / yz;
    const float u = (a * K.k00 + bq * K.k01) + K.k02, v = (a * K.k10 + bq * K.k11) + K.k12;
    float *o = grid + ((size_t)b * P + p) * 2;
    o[0] = u * K.sx - 1.0f;
    o[1] = v * K.sy - 1.0f;
}

__global__ __launch_bounds__(256) void inv_warp_grid_bwd(const float *__restrict__ depth,
                                                         const float *__restrict__ rays,
                                                         const float *__restrict__ R,
                                                         const float *__restrict__ t, float rcd,
                                                         Intr K, const float *__restrict__ ggrid,
                                                         float *__restrict__ gdepth,
                                                         float *__restrict__ gRt, int P, int need_rt) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float *r = R + b * 9, *tt = t + b * 3;
    float s[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {
        const float d = depth[(size_t)b * P + p];
        const float rx = rays[3 * p], ry = rays[3 * p + 1], rz = rays[3 * p + 2];
        const float zx = rx * d - tt[0], zy = ry * d - tt[1], zz = (rz * d - tt[2]) - rcd;
        const float yx = zx * r[0] + zy * r[3] + zz * r[6];
        const float yy = zx * r[1] + zy * r[4] + zz * r[7];
        const float yz = (zx * r[2] + zy * r[5] + zz * r[8]) + rcd;
        const float *g = ggrid + ((size_t)b * P + p) * 2;
        const float gu = g[0] * K.sx, gv = g[1] * K.sy;
        const float ga = gu * K.k00 + gv * K.k10, gb = gu * K.k01 + gv * K.k11;
        const float gyx = ga / yz, gyy = gb / yz, gyz = -(ga * yx + gb * yy) / (yz * yz);
        // gZ = R gY
        const float gzx = r[0] * gyx + r[1] * gyy + r[2] * gyz;
        const float gzy = r[3] * gyx + r[4] * gyy + r[5] * gyz;
        const float gzz = r[6] * gyx + r[7] * gyy + r[8] * gyz;
        gdepth[(size_t)b * P + p] = gzx * rx + gzy * ry + gzz * rz;
        // Y_k = sum_j R_jk (Z - c)_j  =>  gR_jk = (Z - c)_j gY_k
        s[0] += zx * gyx; s[1] += zx * gyy; s[2] += zx * gyz;
        s[3] += zy * gyx; s[4] += zy * gyy; s[5] += zy * gyz;
        s[6] += zz * gyx; s[7] += zz * gyy; s[8] += zz * gyz;
        s[9] -= gzx; s[10] -= gzy; s[11] -= gzz;
    }
    if (!need_rt) return;
    float *o = gRt + b * 12;
#pragma unroll
    for (int k = 0; k < 12; k++) block_atomic_add(s[k], o + k, red);
}

// ------------------------------------------------------------------ smoothness loss
// loss = mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2| over a [N, H, W] map (losses.py:54-79);
// dxdy == dydx element-wise (mixed second difference).  grid (ceil(W/32), ceil(H/8), N), block (32, 8)
__device__ __forceinline__ float sgn(float v) { return (v > 0.0f) ? 1.0f : ((v < 0.0f) ? -1.0f : 0.0f); }

__global__ __launch_bounds__(256) void smooth_loss_fwd(const float *__restrict__ pm, float *__restrict__ loss,
                                                       int N, int H, int W, float w_xx, float w_xy, float w_yy) {
    __shared__ float red[4];
    float acc = 0.0f;
    // grid (ceil(W/32), ceil(H/8), N): one 32x8 patch per workgroup; a (1, 1, 1) grid (deterministic
    // mode) walks all patches of all maps in a fixed order instead
    for (int n = blockIdx.z; n < N; n += gridDim.z)
        for (int y = blockIdx.y * 8 + (threadIdx.x >> 5); y < H; y += gridDim.y * 8)
            for (int x = blockIdx.x * 32 + (threadIdx.x & 31); x < W; x += gridDim.x * 32) {
                const float *p = pm + (size_t)n * H * W;
                const float c = p[y * W + x];
                if (x + 2 < W) acc += w_xx * fabsf(p[y * W + x + 2] - 2.0f * p[y * W + x + 1] + c);
                if (y + 2 < H) acc += w_yy * fabsf(p[(y + 2) * W + x] - 2.0f * p[(y + 1) * W + x] + c);
                if (x + 1 < W && y + 1 < H)
                    acc += w_xy * fabsf((p[(y + 1) * W + x + 1] - p[(y + 1) * W + x]) - (p[y * W + x + 1] - c));
            }
    block_atomic_add(acc, loss, red);
}

__global__ __launch_bounds__(256) void smooth_loss_bwd(const float *__restrict__ pm,
                                                       const float *__restrict__ gloss,
                                                       float *__restrict__ gp, int H, int W,
                                                       float w_xx, float w_xy, float w_yy) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const float *p = pm + (size_t)blockIdx.z * H * W;
    auto P = [&](int yy, int xx) { return p[yy * W + xx]; };
    auto sxx = [&](int yy, int j) {  // sign of dx2 at (yy, j), j in [0, W-3]
        return (j >= 0 && j + 2 < W) ? sgn(P(yy, j + 2) - 2.0f * P(yy, j + 1) + P(yy, j)) : 0.0f;
    };
    auto syy = [&](int i, int xx) {
        return (i >= 0 && i + 2 < H) ? sgn(P(i + 2, xx) - 2.0f * P(i + 1, xx) + P(i, xx)) : 0.0f;
    };
    auto sxy = [&](int i, int j) {
        return (i >= 0 && j >= 0 && i + 1 < H && j + 1 < W)
                   ? sgn((P(i + 1, j + 1) - P(i + 1, j)) - (P(i, j + 1) - P(i, j))) : 0.0f;
    };
    float g = w_xx * (sxx(y, x - 2) - 2.0f * sxx(y, x - 1) + sxx(y, x));
    g += w_yy * (syy(y - 2, x) - 2.0f * syy(y - 1, x) + syy(y, x));
    g += w_xy * (sxy(y - 1, x - 1) - sxy(y - 1, x) - sxy(y, x - 1) + sxy(y, x));
    gp[(size_t)blockIdx.z * H * W + y * W + x] = g * gloss[0];
}

// ------------------------------------------------------------------ normals from depth
// renderer.py:127-139: g = d * ray; n = (g[y,x+1] - g[y,x-1]) x (g[y+1,x] - g[y-1,x]) in the interior,
// (0,0,1) on the border; normal = n / (|n| + 1e-7).  grid (ceil(W/32), ceil(H/8), B), block (32, 8)
constexpr float NORMAL_EPS = 1e-7f;

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 cross3(V3 a, V3 b) { return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 sub3(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }

__device__ __forceinline__ V3 pt3(const float *d, const float *rays, int W, int y, int x) {
    const float dd = d[y * W + x];
    const float *r = rays + 3 * (y * W + x);
    return V3{r[0] * dd, r[1] * dd, r[2] * dd};
}

__global__ __launch_bounds__(256) void normal_fwd(const float *__restrict__ depth,
                                                  const float *__restrict__ rays,
                                                  float *__restrict__ normal, int H, int W) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const float *d = depth + (size_t)blockIdx.z * H * W;
    V3 n{0.0f, 0.0f, 1.0f};
    if (x > 0 && y > 0 && x < W - 1 && y < H - 1)
        n = cross3(sub3(pt3(d, rays, W, y, x + 1), pt3(d, rays, W, y, x - 1)),
                   sub3(pt3(d, rays, W, y + 1, x), pt3(d, rays, W, y - 1, x)));
    const float inv = 1.0f / (sqrtf(n.x * n.x + n.y * n.y + n.z * n.z) + NORMAL_EPS);
    float *o = normal + ((size_t)blockIdx.z * H * W + y * W + x) * 3;
    o[0] = n.x * inv; o[1] = n.y * inv; o[2] = n.z * inv;
}

// gradient of the un-normalised normal n at interior pixel (y, x) -> (g_tu, g_tv)
__device__ __forceinline__ void normal_tangent_grads(const float *d, const float *rays,
                                                     const float *gnormal, int H, int W, int y, int x,
                                                     V3 &gtu, V3 &gtv) {
    gtu = V3{0, 0, 0};
    gtv = V3{0, 0, 0};
    if (!(x > 0 && y > 0 && x < W - 1 && y < H - 1)) return;
    const V3 tu = sub3(pt3(d, rays, W, y, x + 1), pt3(d, rays, W, y, x - 1));
    const V3 tv = sub3(pt3(d, rays, W, y + 1, x), pt3(d, rays, W, y - 1, x));
    const V3 n = cross3(tu, tv);
    const float r = sqrtf(n.x * n.x + n.y * n.y + n.z * n.z), re = r + NORMAL_EPS;
    const float *g = gnormal + 3 * (y * W + x);
    const float dot = n.x * g[0] + n.y * g[1] + n.z * g[2];
    const float k = (r > 0.0f) ? dot / (re * re * r) : 0.0f;
    const V3 gn{g[0] / re - n.x * k, g[1] / re - n.y * k, g[2] / re - n.z * k};
    gtu = cross3(tv, gn);   // d(tu x tv)/d tu
    gtv = cross3(gn, tu);   // d(tu x tv)/d tv
}

__global__ __launch_bounds__(256) void normal_bwd(const float *__restrict__ depth,
                                                  const float *__restrict__ rays,
                                                  const float *__restrict__ gnormal,
                                                  float *__restrict__ gdepth, int H, int W) {
    const int x = blockIdx.x * 32 + (threadIdx.x & 31), y = blockIdx.y * 8 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    const float *d = depth + (size_t)blockIdx.z * H * W;
    const float *gn = gnormal + (size_t)blockIdx.z * H * W * 3;
    // g_point[p] = gtu(y, x-1) - gtu(y, x+1) + gtv(y-1, x) - gtv(y+1, x)
    V3 a, b, acc{0, 0, 0};
    if (x - 1 >= 0) { normal_tangent_grads(d, rays, gn, H, W, y, x - 1, a, b); acc.x += a.x; acc.y += a.y; acc.z += a.z; }
    if (x + 1 < W) { normal_tangent_grads(d, rays, gn, H, W, y, x + 1, a, b); acc.x -= a.x; acc.y -= a.y; acc.z -= a.z; }
    if (y - 1 >= 0) { normal_tangent_grads(d, rays, gn, H, W, y - 1, x, a, b); acc.x += b.x; acc.y += b.y; acc.z += b.z; }
    if (y + 1 < H) { normal_tangent_grads(d, rays, gn, H, W, y + 1, x, a, b); acc.x -= b.x; acc.y -= b.y; acc.z -= b.z; }
    const float *r = rays + 3 * (y * W + x);
    gdepth[(size_t)blockIdx.z * H * W + y * W + x] = acc.x * r[0] + acc.y * r[1] + acc.z * r[2];
}

// ------------------------------------------------------------------ lighting + shading
// model.py:347-360: a = l0/2+.5, b = l1/2+.5, dir = normalize(l2, l3, 1);
// diffuse = relu(n . dir); shading = a + b * diffuse; texture = (albedo/2 + .5) * shading * 2 - 1.
// normal [Bn,H,W,3] and albedo [Ba,3,H,W] broadcast over the light batch B (Bn, Ba in {1, B}).
__global__ __launch_bounds__(256) void shading_fwd(const float *__restrict__ normal,
                                                   const float *__restrict__ light,
                                                   const float *__restrict__ albedo,
                                                   float *__restrict__ diffuse,
                                                   float *__restrict__ texture, int P, int Bn, int Ba) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float *l = light + b * 4;
    const float la = l[0] / 2.0f + 0.5f, lb = l[1] / 2.0f + 0.5f;
    const float nrm = sqrtf(l[2] * l[2] + l[3] * l[3] + 1.0f);
    const float dx = l[2] / nrm, dy = l[3] / nrm, dz = 1.0f / nrm;
    const float *n = normal + ((size_t)(Bn == 1 ? 0 : b) * P + p) * 3;
    const float dot = n[0] * dx + n[1] * dy + n[2] * dz;
    const float dif = dot > 0.0f ? dot : 0.0f;
    diffuse[(size_t)b * P + p] = dif;
    const float sh = la + lb * dif;
    const float *al = albedo + (size_t)(Ba == 1 ? 0 : b) * 3 * P + p;
    float *tx = texture + (size_t)b * 3 * P + p;
    for (int c = 0; c < 3; c++) tx[(size_t)c * P] = (al[(size_t)c * P] / 2.0f + 0.5f) * sh * 2.0f - 1.0f;
}

// per-b gradients: gnormal [B,P,3], galbedo [B,3,P], glight [B,4] (zero-filled by the caller)
__global__ __launch_bounds__(256) void shading_bwd(const float *__restrict__ normal,
                                                   const float *__restrict__ light,
                                                   const float *__restrict__ albedo,
                                                   const float *__restrict__ gdiffuse,
                                                   const float *__restrict__ gtexture,
                                                   float *__restrict__ gnormal,
                                                   float *__restrict__ galbedo,
                                                   float *__restrict__ glight, int P, int Bn, int Ba) {
    __shared__ float red[4];
    const int b = blockIdx.y;
    const float *l = light + b * 4;
    const float la = l[0] / 2.0f + 0.5f, lb = l[1] / 2.0f + 0.5f;
    const float nrm = sqrtf(l[2] * l[2] + l[3] * l[3] + 1.0f);
    const float dx = l[2] / nrm, dy = l[3] / nrm, dz = 1.0f / nrm;
    float s[4] = {0, 0, 0, 0};
    for (int p = blockIdx.x * 256 + threadIdx.x; p < P; p += gridDim.x * 256) {   // one pass, or all of P in deterministic mode
        const float *n = normal + ((size_t)(Bn == 1 ? 0 : b) * P + p) * 3;
        const float dot = n[0] * dx + n[1] * dy + n[2] * dz;
        const float dif = dot > 0.0f ? dot : 0.0f;
        const float sh = la + lb * dif;
        const float *al = albedo + (size_t)(Ba == 1 ? 0 : b) * 3 * P + p;
        const float *gt = gtexture + (size_t)b * 3 * P + p;
        float g_sh = 0.0f;
        for (int c = 0; c < 3; c++) {
            const float g = gt[(size_t)c * P];
            g_sh += g * (al[(size_t)c * P] / 2.0f + 0.5f) * 2.0f;
            galbedo[(size_t)b * 3 * P + (size_t)c * P + p] = g * sh;  // d/d albedo = 0.5 * sh * 2
        }
        const float g_dif = g_sh * lb + (gdiffuse ? gdiffuse[(size_t)b * P + p] : 0.0f);
        const float g_dot = dot > 0.0f ? g_dif : 0.0f;
        float *gn = gnormal + ((size_t)b * P + p) * 3;
        gn[0] = g_dot * dx; gn[1] = g_dot * dy; gn[2] = g_dot * dz;
        const float g_dx = g_dot * n[0], g_dy = g_dot * n[1], g_dz = g_dot * n[2];
        // direction = (l2, l3, 1) / nrm:  g_l2 = (g_dx - dx * (g . dir)) / nrm, same for l3
        const float gd_dot = g_dx * dx + g_dy * dy + g_dz * dz;
        s[0] += g_sh * 0.5f;
        s[1] += (g_sh * dif) * 0.5f;
        s[2] += (g_dx - dx * gd_dot) / nrm;
        s[3] += (g_dy - dy * gd_dot) / nrm;
    }
    float *o = glight + b * 4;
#pragma unroll
    for (int k = 0; k < 4; k++) block_atomic_add(s[k], o + k, red);
}

// ------------------------------------------------------------------ depth head
// model.py:337-345 get_clamped_depth + :323-324 rescale_depth on the depth net's raw map [n = B*H*W]:
//   t = tanh(raw - mean),  d = (1 + t) / 2 * hi + (1 - t) / 2 * lo,
//   out = d * (1 - b) + b * border_depth,  b = 1.02 in the two left / right columns, else 0
// (mean: device scalar, the whole-batch mean the caller reduced).  The reference spends 14 launches on
// this chain and as many again in autograd.
struct DepthHead { float lo, hi, border_depth; int W, border; };

__device__ __forceinline__ float border_weight(long i, const DepthHead &h) {
    if (!h.border) return 0.0f;
    const int col = (int)(i % h.W);
    return (col < 2 || col >= h.W - 2) ? 1.02f : 0.0f;
}

__global__ __launch_bounds__(256) void depth_head_fwd(const float *__restrict__ raw, const float *__restrict__ mean,
                                                      float *__restrict__ out, long n, DepthHead h) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float t = tanhf(raw[i] - mean[0]);
    const float d = (1.0f + t) / 2.0f * h.hi + (1.0f - t) / 2.0f * h.lo;
    const float b = border_weight(i, h);
    out[i] = d * (1.0f - b) + b * h.border_depth;
}

// gc = d out / d (raw - mean) * g; *gsum += sum gc (zeroed by the launcher).  The centring's own backward
// (g_raw = gc - mean(gc)) is the second kernel.
__global__ __launch_bounds__(256) void depth_head_bwd(const float *__restrict__ raw, const float *__restrict__ mean,
                                                      const float *__restrict__ g, float *__restrict__ gc,
                                                      float *__restrict__ gsum, long n, DepthHead h) {
    __shared__ float red[4];
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float t = tanhf(raw[i] - mean[0]);
        const float gd = g[i] * (1.0f - border_weight(i, h));
        const float gt = gd * h.hi / 2.0f - gd * h.lo / 2.0f;
        const float v = gt * (1.0f - t * t);
        gc[i] = v;
        acc += v;
    }
    block_atomic_add(acc, gsum, red);
}

__global__ __launch_bounds__(256) void sub_mean_kernel(float *__restrict__ x, const float *__restrict__ sum, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] -= sum[0] / (float)n;
}

static Intr make_intr(const float *K, int H, int W) {
    return Intr{K[0], K[1], K[2], K[3], K[4], K[5], 2.0f / (float)(W - 1), 2.0f / (float)(H - 1)};
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_view_transform_fwd(const float *view, float rot_scale, float txy_scale,
                                      float tz_scale, float *R, float *t, int B, g2s_stream_t stream) {
    G2S_REQUIRE(view && R && t && B > 0, "bad argument");
    view_transform_fwd<<<cdiv(B, 64), 64, 0, as_stream(stream)>>>(view, ViewScale{rot_scale, txy_scale, tz_scale}, R, t, B);
    return check_launch("g2s_view_transform_fwd");
}

extern "C" int g2s_view_transform_bwd(const float *view, float rot_scale, float txy_scale,
                                      float tz_scale, const float *gR, const float *gt, float *gview,
                                      int B, g2s_stream_t stream) {
    G2S_REQUIRE(view && gR && gt && gview && B > 0, "bad argument");
    view_transform_bwd<<<cdiv(B, 64), 64, 0, as_stream(stream)>>>(view, ViewScale{rot_scale, txy_scale, tz_scale}, gR, gt, gview, B);
    return check_launch("g2s_view_transform_bwd");
}

extern "C" int g2s_warp_verts_fwd(const float *depth, const float *rays, const float *R, const float *t,
                                  float rot_center_depth, float *verts, int B, int P,
                                  g2s_stream_t stream) {
    G2S_REQUIRE(depth && rays && R && t && verts && B > 0 && P > 0 && B <= 65535, "bad argument");
    warp_verts_fwd<<<dim3(cdiv(P, 256), B), 256, 0, as_stream(stream)>>>(depth, rays, R, t, rot_center_depth, verts, P);
    return check_launch("g2s_warp_verts_fwd");
}

extern "C" int g2s_warp_verts_bwd(const float *depth, const float *rays, const float *R,
                                  const float *gverts, float rot_center_depth, float *gdepth,
                                  float *gRt, int B, int P, g2s_stream_t stream) {
    G2S_REQUIRE(depth && rays && R && gverts && gdepth && B > 0 && P > 0 && B <= 65535, "bad argument");
    hipStream_t st = as_stream(stream);
    if (gRt && !precleared() && hipMemsetAsync(gRt, 0, (size_t)B * 12 * sizeof(float), st) != hipSuccess)
        return fail(G2S_ERR_LAUNCH, "hipMemsetAsync failed");
    warp_verts_bwd<<<dim3(deterministic() ? 1 : cdiv(P, 256), B), 256, 0, st>>>(depth, rays, R, gverts, rot_center_depth, gdepth, gRt, P, gRt != nullptr);
    return check_launch("g2s_warp_verts_bwd");
}

extern "C" int g2s_inv_warp_grid_fwd(const float *depth, const float *rays, const float *R,
                                     const float *t, const float *K, float rot_center_depth,
                                     float *grid, int B, int H, int W, g2s_stream_t stream) {
    G2S_REQUIRE(depth && rays && R && t && K && grid && B > 0 && H > 1 && W > 1 && B <= 65535, "bad argument");
    inv_warp_grid_fwd<<<dim3(cdiv(H * W, 256), B), 256, 0, as_stream(stream)>>>(
        depth, rays, R, t, rot_center_depth, make_intr(K, H, W), grid, H * W);
    return check_launch("g2s_inv_warp_grid_fwd");
}

extern "C" int g2s_inv_warp_grid_bwd(const float *depth, const float *rays, const float *R,
                                     const float *t, const float *K, float rot_center_depth,
                                     const float *ggrid, float *gdepth, float *gRt, int B, int H,
                                     int W, g2s_stream_t stream) {
    G2S_REQUIRE(depth && rays && R && t && K && ggrid && gdepth && B > 0 && H > 1 && W > 1 && B <= 65535, "bad argument");
    hipStream_t st = as_stream(stream);
    if (gRt && !precleared() && hipMemsetAsync(gRt, 0, (size_t)B * 12 * sizeof(float), st) != hipSuccess)
        return fail(G2S_ERR_LAUNCH, "hipMemsetAsync failed");
    inv_warp_grid_bwd<<<dim3(deterministic() ? 1 : cdiv(H * W, 256), B), 256, 0, st>>>(
        depth, rays, R, t, rot_center_depth, make_intr(K, H, W), ggrid, gdepth, gRt, H * W, gRt != nullptr);
    return check_launch("g2s_inv_warp_grid_bwd");
}

static void smooth_weights(int N, int H, int W, float &wxx, float &wxy, float &wyy) {
    wxx = (W > 2) ? 1.0f / ((float)N * H * (W - 2)) : 0.0f;
    wyy = (H > 2) ? 1.0f / ((float)N * (H - 2) * W) : 0.0f;
    wxy = (H > 1 && W > 1) ? 2.0f / ((float)N * (H - 1) * (W - 1)) : 0.0f;  // dxdy and dydx
}

extern "C" int g2s_smooth_loss_fwd(const float *p, float *loss, int N, int H, int W, g2s_stream_t stream) {
    G2S_REQUIRE(p && loss && N > 0 && H > 0 && W > 0 && N <= 65535, "bad argument");
    float wxx, wxy, wyy;
    smooth_weights(N, H, W, wxx, wxy, wyy);
    hipStream_t st = as_stream(stream);
    if (!precleared() && hipMemsetAsync(loss, 0, sizeof(float), st) != hipSuccess) return fail(G2S_ERR_LAUNCH, "hipMemsetAsync failed");
    smooth_loss_fwd<<<deterministic() ? dim3(1, 1, 1) : dim3(cdiv(W, 32), cdiv(H, 8), N), 256, 0, st>>>(p, loss, N, H, W, wxx, wxy, wyy);
    return check_launch("g2s_smooth_loss_fwd");
}

extern "C" int g2s_smooth_loss_bwd(const float *p, const float *gloss, float *gp, int N, int H, int W,
                                   g2s_stream_t stream) {
    G2S_REQUIRE(p && gloss && gp && N > 0 && H > 0 && W > 0 && N <= 65535, "bad argument");
    float wxx, wxy, wyy;
    smooth_weights(N, H, W, wxx, wxy, wyy);
    smooth_loss_bwd<<<dim3(cdiv(W, 32), cdiv(H, 8), N), 256, 0, as_stream(stream)>>>(p, gloss, gp, H, W, wxx, wxy, wyy);
    return check_launch("g2s_smooth_loss_bwd");
}

extern "C" int g2s_normal_fwd(const float *depth, const float *rays, float *normal, int B, int H, int W,
                              g2s_stream_t stream) {
    G2S_REQUIRE(depth && rays && normal && B > 0 && H > 0 && W > 0 && B <= 65535, "bad argument");
    normal_fwd<<<dim3(cdiv(W, 32), cdiv(H, 8), B), 256, 0, as_stream(stream)>>>(depth, rays, normal, H, W);
    return check_launch("g2s_normal_fwd");
}

extern "C" int g2s_normal_bwd(const float *depth, const float *rays, const float *gnormal, float *gdepth,
                              int B, int H, int W, g2s_stream_t stream) {
    G2S_REQUIRE(depth && rays && gnormal && gdepth && B > 0 && H > 0 && W > 0 && B <= 65535, "bad argument");
    normal_bwd<<<dim3(cdiv(W, 32), cdiv(H, 8), B), 256, 0, as_stream(stream)>>>(depth, rays, gnormal, gdepth, H, W);
    return check_launch("g2s_normal_bwd");
}

extern "C" int g2s_shading_fwd(const float *normal, const float *light, const float *albedo,
                               float *diffuse, float *texture, int B, int Bn, int Ba, int P,
                               g2s_stream_t stream) {
    G2S_REQUIRE(normal && light && albedo && diffuse && texture && B > 0 && P > 0 && B <= 65535, "bad argument");
    G2S_REQUIRE((Bn == 1 || Bn == B) && (Ba == 1 || Ba == B), "normal / albedo batch must be 1 or B");
    shading_fwd<<<dim3(cdiv(P, 256), B), 256, 0, as_stream(stream)>>>(normal, light, albedo, diffuse, texture, P, Bn, Ba);
    return check_launch("g2s_shading_fwd");
}

extern "C" int g2s_shading_bwd(const float *normal, const float *light, const float *albedo,
                               const float *gdiffuse, const float *gtexture, float *gnormal,
                               float *galbedo, float *glight, int B, int Bn, int Ba, int P,
                               g2s_stream_t stream) {
    G2S_REQUIRE(normal && light && albedo && gtexture && gnormal && galbedo && glight && B > 0 && P > 0 && B <= 65535,
                "bad argument");
    G2S_REQUIRE((Bn == 1 || Bn == B) && (Ba == 1 || Ba == B), "normal / albedo batch must be 1 or B");
    hipStream_t st = as_stream(stream);
    if (!precleared() && hipMemsetAsync(glight, 0, (size_t)B * 4 * sizeof(float), st) != hipSuccess)
        return fail(G2S_ERR_LAUNCH, "hipMemsetAsync failed");
    shading_bwd<<<dim3(deterministic() ? 1 : cdiv(P, 256), B), 256, 0, st>>>(normal, light, albedo, gdiffuse, gtexture, gnormal, galbedo, glight, P, Bn, Ba);
    return check_launch("g2s_shading_bwd");
}

extern "C" int g2s_depth_head_fwd(const float *raw, const float *mean, float *out, int64_t n, int W, float lo,
                                  float hi, int clamp_border, float border_depth, g2s_stream_t stream) {
    G2S_REQUIRE(raw && mean && out && n > 0 && W >= 4 && n % W == 0, "bad argument (W >= 4 must divide n)");
    depth_head_fwd<<<cdiv((long)n, 256), 256, 0, as_stream(stream)>>>(raw, mean, out, (long)n,
                                                                     DepthHead{lo, hi, border_depth, W, clamp_border});
    return check_launch("g2s_depth_head_fwd");
}

extern "C" int g2s_depth_head_bwd(const float *raw, const float *mean, const float *g, float *g_raw, float *gsum,
                                  int64_t n, int W, float lo, float hi, int clamp_border, float border_depth,
                                  g2s_stream_t stream) {
    G2S_REQUIRE(raw && mean && g && g_raw && gsum && n > 0 && W >= 4 && n % W == 0, "bad argument");
    hipStream_t st = as_stream(stream);
    if (!precleared() && hipMemsetAsync(gsum, 0, sizeof(float), st) != hipSuccess) return fail(G2S_ERR_LAUNCH, "hipMemsetAsync failed");
    const int blocks = deterministic() ? 1 : (int)std::min<long>(cdiv((long)n, 256), 1024);
    depth_head_bwd<<<blocks, 256, 0, st>>>(raw, mean, g, g_raw, gsum, (long)n, DepthHead{lo, hi, border_depth, W, clamp_border});
    sub_mean_kernel<<<cdiv((long)n, 256), 256, 0, st>>>(g_raw, gsum, (long)n);
    return check_launch("g2s_depth_head_bwd");
}
