// winograd4.hip — 3x3 stride-1 convolution (padding 1) as Winograd F(4x4, 3x3) on the fp32 matrix
// cores of gfx950: the large-map layers of the FROZEN networks (generator StyledConvs at >= 32^2,
// stylegan2-pytorch/model.py:285-289; discriminator ResBlock conv1, model.py:679-697; VGG16 trunk of
// LPIPS, lpips/pretrained_networks.py:97-135) and their data-gradients.  Same contract as
// g2s_conv3x3_wino (winograd.hip); 36 multiplications per 16 outputs instead of 64 (F(2x2)) or 144
// (direct): 1.78x fewer MFMA operations than winograd.hip, 4x fewer than the direct implicit GEMM.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A      per 4x4 output tile, 6x6 input patch d
//
// with the interpolation points 0, +-1, +-2, inf (Lavin & Gray, "Fast Algorithms for Convolutional
// Neural Networks", 2016, section 4.2): 36 independent GEMMs M_p[Cout][tiles] = U_p[Cout][Cin] V_p[Cin][tiles].
//
// Why a different kernel shape than winograd.hip: a wave cannot hold 36 accumulator blocks (576
// registers).  The 6x6 positions are split into four 3x3 quadrants, one per wave; the output transform
// is linear in the positions, so every wave applies A^T . A to ITS nine positions (84 VALU operations
// per element) and the four partial 4x4 output tiles meet in LDS.
//   * Workgroup = 64 output channels x 32 tiles (512 output pixels per channel), 8 waves: wave
//     (g, q) owns channel half g and quadrant q: 9 accumulators of v_mfma_f32_32x32x2_f32 = 144
//     registers -> TWO waves per SIMD (256 registers each): group 0 stages first and multiplies after,
//     group 1 the other way round, so the two waves of a SIMD take turns on the matrix pipe and the vector
//     ALU without hand placement.
//   * U = G g G^T once per weight tensor (computed in double, rounded once) in the tiled layout
//     [m-tile][k-tile][g][position][channel pair][32 m][2]: a K tile (4 channels) of an m-tile is one
//     contiguous 36 KB block in LDS order -> 36 LDS-DMA pieces of 1 KB, ring of two.
//   * The input patches go through LDS raw: per K tile the 6 rows x 4 own columns of every (tile, channel)
//     patch are fetched ONCE (12 LDS-DMA pieces; rows outside the image come from a zero row), two stages.
//     A thread then transforms one 3x3 quadrant of V = B^T d B for one (tile, channel): five 16-byte LDS
//     reads (a 5x5 corner of the patch; the outer column from the neighbour lane by a whole-wave DPP
//     shift), 48 VALU operations, 9 values to the V stage.  (Loading the corners from global memory
//     instead — 3.3x the bytes — ran at the same speed; so did a deeper U ring, requests placed behind
//     MFMAs, and no wait for them at all: what the requests cost is shared capacity beside the DS traffic,
//     not latency — DESIGN §4.3b.)
//   * Every transfer is an LDS-DMA request issued from inline assembly; the one s_waitcnt vmcnt(0) before the
//     hand-over barrier states their completion (the compiler would add vmcnt(0) before every DS instruction
//     after a __builtin_amdgcn_global_load_lds it cannot prove disjoint).
//   * One ds_read_b64 per operand and position feeds two MFMAs (channels 2 kk, 2 kk + 1); after the barrier
//     the fragments of the next tile are read under the last four MFMAs of this one.
//   * Partition: whole tiles, or K slices through a workspace + split_reduce (no atomics: fixed summation
//     order on every path).
#include <algorithm>
#include "g2s_common.h"
#include "split_reduce.h"
#include "xcd_tile.h"

namespace g2s {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int W4BM = 64;        // output channels per workgroup
constexpr int W4BT = 32;        // 4x4 output tiles per workgroup
constexpr int W4KC = 4;         // channels per K tile
constexpr int W4THREADS = 512;
constexpr int W4UBLOCK = 36 * W4KC * W4BM;   // floats of one (m-tile, k-tile) block of U
constexpr int W4VBLOCK = 36 * W4KC * W4BT;

struct Wino4Desc {
    const float *x, *U, *in_scale, *out_scale, *bias, *noise, *noise_w;
    float *y;
    int B, Cr, M, H, W;
    int TH, TW;          // tiles per image (H / 4, W / 4)
    int ktiles, splitk;
    int act;
    float act_alpha, act_gain;
    float *part;         // split-K: slice s stores to part[s * part_n + offset in y]
    long part_n;
};

// B^T along one axis, three of its six outputs from the five inputs they use:
// rows 0..2 of B^T on d0..d4, rows 3..5 on d1..d5 (passed as e0..e4)
__device__ __forceinline__ void bt_lo3(float d0, float d1, float d2, float d3, float d4, float (&o)[3]) {
    o[0] = fmaf(4.0f, d0, fmaf(-5.0f, d2, d4));
    const float a = fmaf(-4.0f, d2, d4), b = fmaf(-4.0f, d1, d3);
    o[1] = a + b;
    o[2] = a - b;
}
__device__ __forceinline__ void bt_hi3(float e0, float e1, float e2, float e3, float e4, float (&o)[3]) {
    const float c = e3 - e1, e = e2 - e0;
    o[0] = fmaf(2.0f, e, c);
    o[1] = fmaf(-2.0f, e, c);
    o[2] = fmaf(4.0f, e0, fmaf(-5.0f, e2, e4));
}

// A^T restricted to three positions of one axis: positions 0..2 (hi = false) or 3..5 (hi = true)
__device__ __forceinline__ void at3(bool hi, float m0, float m1, float m2, float (&o)[4]) {
    if (!hi) {
        const float s = m1 + m2, dd = m1 - m2;
        o[0] = m0 + s;
        o[1] = dd;
        o[2] = s;
        o[3] = dd;
    } else {
        const float s = m0 + m1, dd = m0 - m1;
        o[0] = s;
        o[1] = 2.0f * dd;
        o[2] = 4.0f * s;
        o[3] = fmaf(8.0f, dd, m2);
    }
}

// 16 bytes of zeros: the source of the patch rows above / below the image
__device__ __attribute__((aligned(16))) float w4_zero_row[4] = {0.f, 0.f, 0.f, 0.f};

// LDS (floats): U ring of two K tiles | V, two stages | raw patch rows, two stages | the style scales of the image
constexpr int W4RAW = 12 * 64 * 4;                 // 4 channels x 3 row pairs x (32 tiles x 2 rows) x 4 columns
constexpr int W4_LU = 0, W4_LV = 2 * W4UBLOCK, W4_LR = W4_LV + 2 * W4VBLOCK, W4_LS = W4_LR + 2 * W4RAW;
constexpr int W4MAXC = 512;                        // reduction channels whose scales fit the LDS table
constexpr int W4LDS = W4_LS + W4MAXC;

template <bool SCALE>
__global__ __launch_bounds__(W4THREADS) void wino4_kernel(Wino4Desc d) {
    __shared__ __attribute__((aligned(16))) float L[W4LDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;     // channel half of the MFMA block; staging order (0: stage first, 1: multiply first)
    const int quad = wave & 3;     // positions i in 3 (quad >> 1) .. + 2, j in 3 (quad & 1) .. + 2
    const int l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = d.M / W4BM;
    const int per_img = d.TH * d.TW;
    const int g = xcd_logical_tile();
    const int mt = g % tiles_m, nt = g / tiles_m;
    const int per = (d.ktiles + d.splitk - 1) / d.splitk;
    const int kt_begin = blockIdx.y * per;
    const int n_it = min(d.ktiles, kt_begin + per) - kt_begin;
    const int HW = d.H * d.W;
    // the 32 tiles of a block lie in ONE image (per_img % 32 == 0, checked by the host)
    const int img = (nt * W4BT) / per_img;
    const int tile0 = nt * W4BT - img * per_img;

    // Every global -> LDS transfer is an LDS-DMA request in inline assembly: the compiler does not see
    // them, so it adds no waits of its own (with __builtin_amdgcn_global_load_lds every later DS instruction
    // that may alias waits for ALL outstanding requests, vmcnt(0), which makes the copy synchronous); their
    // completion is stated once per K tile by the s_waitcnt vmcnt(0) before the hand-over barrier.  (M0 is
    // written inside the asm block without a clobber — clang rejects reserved registers in clobber lists with a
    // warning; nothing else in this kernel uses M0: gfx9 DS instructions do not read it.)
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(&L[0]);
    auto dma16 = [&](unsigned lds_float, const char *gp) {
        const unsigned la = lds0 + lds_float * 4;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(la), "v"(gp) : "memory");
    };
    // ---- requests.  48 pieces of 1 KB per K tile: 36 of U, 12 of raw patch rows; 6 per wave.
    //   U block (m-tile, K tile): contiguous 36 KB in LDS order; waves 0..3 take pieces 3 w .. 3 w + 2, waves
    //     4..7 pieces 12 + 6 (w - 4) .. + 5.
    //   raw rows: piece c = 3 channel + row pair; lane = tile + 32 (row & 1); waves 0..3 take c = w, w + 4, w + 8.
    const char *ubase = reinterpret_cast<const char *>(d.U) + (size_t)mt * d.ktiles * W4UBLOCK * 4 + lane * 16;
    auto dma_u = [&](int kt, int slot) {
        const int first = wave < 4 ? 3 * wave : 12 + 6 * (wave - 4), cnt = wave < 4 ? 3 : 6;
#pragma unroll
        for (int e = 0; e < 6; e++)
            if (e < cnt) dma16(W4_LU + slot * W4UBLOCK + (first + e) * 256, ubase + (size_t)kt * W4UBLOCK * 4 + (first + e) * 1024);
    };
    // per-lane sources of the three raw pieces of this wave (waves 0..3): piece c = wave + 4 e
    const char *rsrc[3];
    bool rok[3];
    {
        const int n = tile0 + l31, ty = n / d.TW, tx = n % d.TW;
#pragma unroll
        for (int e = 0; e < 3; e++) {
            const int c = (wave & 3) + 4 * e, ch = c / 3, rp = c % 3;
            const int iy = 4 * ty - 1 + 2 * rp + lk;
            rok[e] = iy >= 0 && iy < d.H;
            rsrc[e] = rok[e] ? reinterpret_cast<const char *>(d.x + ((size_t)(img * d.Cr + ch) * d.H + iy) * d.W + 4 * tx)
                             : reinterpret_cast<const char *>(w4_zero_row);
        }
    }
    auto dma_raw = [&](int kt, int st) {
        if (wave < 4) {
#pragma unroll
            for (int e = 0; e < 3; e++)
                dma16(W4_LR + st * W4RAW + ((wave & 3) + 4 * e) * 256, rsrc[e] + (rok[e] ? (size_t)kt * W4KC * HW * 4 : 0));
        }
    };

    // ---- staging role: tile l31 of the block, channel sc of the K tile, rows 3 sh .. + 2 and columns 3 sq .. + 2
    // of V = B^T d B (one 3x3 quadrant of the 6x6 positions: it needs a 5x5 corner of the 6x6 patch)
    const int sc = 2 * (wave & 1) + lk;
    const int sh = (wave >> 1) & 1;
    const int sq = grp;
    bool edge_col;   // the outer patch column this thread needs lies outside the image
    {
        const int tx = (tile0 + l31) % d.TW;
        edge_col = sq ? tx == d.TW - 1 : tx == 0;
    }
    // the quadrant of B^T d B of raw stage `rst` -> stage `st` of V, layout [position][channel pair][tile][2]
    auto stage_patch = [&](int kt, int rst, int st) {
        const f32x4 *raw = reinterpret_cast<const f32x4 *>(&L[W4_LR + rst * W4RAW]);
        float t[5][3];
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const int r = sh + i;                                    // patch row
            const f32x4 v = raw[(sc * 3 + (r >> 1)) * 64 + l31 + 32 * (r & 1)];
            // outer columns from the neighbour lanes = neighbour tiles of the tile row (whole-wave DPP shifts);
            // (__float_as_int, not __builtin_bit_cast: bit_cast of a vector ELEMENT reads element 0 with this compiler)
            const int nl = __builtin_amdgcn_mov_dpp(__float_as_int(v.w), 0x138, 0xf, 0xf, true);   // lane - 1
            const int nr = __builtin_amdgcn_mov_dpp(__float_as_int(v.x), 0x130, 0xf, 0xf, true);   // lane + 1
            if (sq == 0)   // columns 0..2 of V from patch columns -1..3
                bt_lo3(edge_col ? 0.0f : __builtin_bit_cast(float, nl), v.x, v.y, v.z, v.w, t[i]);
            else           // columns 3..5 of V from patch columns 0..4
                bt_hi3(v.x, v.y, v.z, v.w, edge_col ? 0.0f : __builtin_bit_cast(float, nr), t[i]);
        }
        const float rs = SCALE ? L[W4_LS + kt * W4KC + sc] : 1.0f;
        float *vw = &L[W4_LV + st * W4VBLOCK + ((sh * 2 + sq) * 9) * 128 + (sc >> 1) * 64 + l31 * 2 + (sc & 1)];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float c3[3];
            if (sh == 0) bt_lo3(t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], c3);
            else bt_hi3(t[0][j], t[1][j], t[2][j], t[3][j], t[4][j], c3);
#pragma unroll
            for (int i = 0; i < 3; i++) vw[(i * 3 + j) * 128] = c3[i] * rs;
        }
    };

    f32x16 acc[9];
#pragma unroll
    for (int p = 0; p < 9; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.0f;

    // ---- prologue: U(0), raw(0), raw(1), the image's style scales; tile 0 staged
    dma_u(kt_begin, 0);
    dma_raw(kt_begin, 0);
    if (n_it > 1) dma_raw(kt_begin + 1, 1);
    if constexpr (SCALE) {
        if (tid < d.Cr) L[W4_LS + tid] = d.in_scale[(size_t)img * d.Cr + tid];
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    stage_patch(kt_begin, 0, 0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const int foff = lk * 64 + l31 * 2;
    f32x2 fa[9], fb[9];
    auto read_frags = [&](int slot, int st, const int p0, const int p1) {
        const float *ua = &L[W4_LU + slot * W4UBLOCK + (grp * 36 + quad * 9) * 128 + foff];
        const float *vb = &L[W4_LV + st * W4VBLOCK + (quad * 9) * 128 + foff];
#pragma unroll
        for (int p = p0; p < p1; p++) {
            fa[p] = *reinterpret_cast<const f32x2 *>(ua + p * 128);
            fb[p] = *reinterpret_cast<const f32x2 *>(vb + p * 128);
        }
    };
    auto mfmas = [&](const int p0, const int p1) {
#pragma unroll
        for (int p = p0; p < p1; p++) {
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p].x, fb[p].x, acc[p], 0, 0, 0);
            acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[p].y, fb[p].y, acc[p], 0, 0, 0);
        }
    };
    read_frags(0, 0, 0, 9);
    // One K tile per iteration, one barrier.  Between barrier it - 1 and barrier it every wave
    //   * requests U(it + 1) into the ring slot tile it - 1 has left and the raw rows of tile it + 2 into the
    //     stage tile it has left (6 pieces per wave; they land before barrier it: vmcnt(0)),
    //   * stages its quadrant of tile it + 1 from the raw rows barrier it - 1 has published,
    //   * multiplies positions 0..6 of tile it.
    // The two waves of a SIMD (one of each group) take turns on the matrix pipe and the vector ALU: group 0
    // stages FIRST and multiplies after, group 1 the other way round.  After the barrier the fragments of tile
    // it + 1 go into the registers positions 0..6 have released, under the MFMAs of positions 7, 8.
    for (int it = 0; it < n_it; it++) {
        const bool more = it + 1 < n_it;
        const int kt = kt_begin + it;
        if (more) dma_u(kt + 1, (it + 1) & 1);
        if (it + 2 < n_it) dma_raw(kt + 2, it & 1);
        if (grp == 0 && more) stage_patch(kt + 1, (it + 1) & 1, (it + 1) & 1);
        mfmas(0, 7);
        if (grp != 0 && more) stage_patch(kt + 1, (it + 1) & 1, (it + 1) & 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) read_frags((it + 1) & 1, (it + 1) & 1, 0, 7);
        mfmas(7, 9);
        if (more) read_frags((it + 1) & 1, (it + 1) & 1, 7, 9);
    }
    __syncthreads();   // the last tile's fragment reads are done before the exchange buffer reuses the LDS

    // ---- epilogue: partial output transform of this wave's quadrant, the four partial 4x4 tiles meet in
    // LDS (wave `quad` owns output row `quad` of every tile), scale / bias / noise / activation, store.
    // C layout of the 32x32 MFMA: row m = (r & 3) + 8 (r >> 2) + 4 lk, column n = l31.
    const bool qi = quad >> 1, qj = quad & 1;
    f32x4 *X = reinterpret_cast<f32x4 *>(&L[0]);   // exchange buffer [grp][dest row a][source slot 3][rr 4][lane 64]
    const int Ntiles = d.B * per_img;
    const int n = nt * W4BT + l31;
    const bool valid = n < Ntiles;
    const int b = valid ? n / per_img : 0, rr_ = valid ? n % per_img : 0;
    const int oy = 4 * (rr_ / d.TW) + quad, ox = 4 * (rr_ % d.TW);
    const bool raw = d.part != nullptr;
    f32x4 nz{0.f, 0.f, 0.f, 0.f};
    if (d.noise && !raw) {
        const float nw = d.noise_w[0];
        nz = *reinterpret_cast<const f32x4 *>(d.noise + oy * d.W + ox) * nw;
    }
    float *yb = (raw ? d.part + (size_t)blockIdx.y * d.part_n : d.y) + ((size_t)b * d.M * d.H + oy) * d.W + ox;
    const float *ob = d.out_scale ? d.out_scale + (size_t)b * d.M : nullptr;
#pragma unroll
    for (int ch = 0; ch < 4; ch++) {
        f32x4 own[4];
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            const int r = ch * 4 + rr;
            float t[3][4], Y[4][4];
#pragma unroll
            for (int i = 0; i < 3; i++) at3(qj, acc[i * 3 + 0][r], acc[i * 3 + 1][r], acc[i * 3 + 2][r], t[i]);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float col[4];
                at3(qi, t[0][c], t[1][c], t[2][c], col);
#pragma unroll
                for (int a = 0; a < 4; a++) Y[a][c] = col[a];
            }
#pragma unroll
            for (int a = 0; a < 4; a++) {
                const f32x4 v{Y[a][0], Y[a][1], Y[a][2], Y[a][3]};
                if (a == quad) own[rr] = v;
                else X[(((grp * 4 + a) * 3 + (((quad - a) & 3) - 1)) * 4 + rr) * 64 + lane] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 4; rr++) {
            f32x4 v = own[rr];
#pragma unroll
            for (int s = 0; s < 3; s++) v += X[(((grp * 4 + quad) * 3 + s) * 4 + rr) * 64 + lane];
            const int m = mt * W4BM + grp * 32 + rr + 8 * ch + 4 * lk;
            const float sc_o = ob ? ob[m] : 1.0f;
            if (raw) {
                v *= sc_o;
            } else {
                const float bi = d.bias ? d.bias[m] : 0.0f;
                v = v * sc_o + bi + nz;
                if (d.act) {
#pragma unroll
                    for (int q = 0; q < 4; q++) v[q] = (v[q] > 0.0f ? v[q] : v[q] * d.act_alpha) * d.act_gain;
                }
            }
            if (valid) *reinterpret_cast<f32x4 *>(yb + (size_t)m * HW) = v;
        }
        if (ch < 3) __syncthreads();
    }
}

// U = G g G^T of every (output channel m, reduction channel c) into the tiled layout (double arithmetic,
// one rounding).  w(m, c, ky, kx) = w[m * w_ms + c * w_ks + (flip ? 8 - (3 ky + kx) : 3 ky + kx)].
__global__ void wino4_weights_kernel(const float *w, float *U, int M, int Cr, int w_ms, int w_ks, int flip,
                                     int ktiles) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * Cr) return;
    const int m = idx % M, c = idx / M;
    const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                            {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
    double gg[9], h[6][3];
    for (int t = 0; t < 9; t++) gg[t] = (double)w[(size_t)m * w_ms + (size_t)c * w_ks + (flip ? 8 - t : t)];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 3; j++) h[i][j] = G[i][0] * gg[j] + G[i][1] * gg[3 + j] + G[i][2] * gg[6 + j];
    float *dst = U + ((size_t)((m / W4BM) * ktiles + c / W4KC) * 2 + (m % W4BM) / 32) * 36 * 128 + ((c % W4KC) >> 1) * 64 +
                 (m % 32) * 2 + (c & 1);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) {
            const double u = h[i][0] * G[j][0] + h[i][1] * G[j][1] + h[i][2] * G[j][2];
            const int P = ((i / 3) * 2 + j / 3) * 9 + (i % 3) * 3 + j % 3;
            dst[P * 128] = (float)u;
        }
}

}  // namespace g2s

using namespace g2s;

extern "C" size_t g2s_wino4_weights_floats(int M, int Cr) {
    if (M <= 0 || Cr <= 0) return 0;
    return (size_t)cdiv(M, W4BM) * cdiv(Cr, W4KC) * W4UBLOCK;
}

extern "C" int g2s_wino4_weights(const float *w, float *U, int Cout, int Cin, int transpose, g2s_stream_t stream) {
    G2S_REQUIRE(w && U, "w, U must not be NULL");
    G2S_REQUIRE(Cout > 0 && Cin > 0, "sizes must be positive");
    const int M = transpose ? Cin : Cout, Cr = transpose ? Cout : Cin;
    const int w_ms = transpose ? 9 : Cin * 9, w_ks = transpose ? Cin * 9 : 9;
    hipStream_t st = as_stream(stream);
    const size_t n = g2s_wino4_weights_floats(M, Cr);
    if (hipMemsetAsync(U, 0, n * sizeof(float), st) != hipSuccess) return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(U) failed");
    wino4_weights_kernel<<<cdiv((long)M * Cr, 256), 256, 0, st>>>(w, U, M, Cr, w_ms, w_ks, transpose ? 1 : 0, cdiv(Cr, W4KC));
    return check_launch("g2s_wino4_weights");
}

// 1 when g2s_conv3x3_wino4 takes the shape: whole 4x4 tiles, whole tile rows per 32-tile block, whole
// 64-channel blocks and K tiles.
extern "C" int g2s_wino4_supported(int B, int Cr, int M, int H, int W) {
    if (B <= 0 || Cr <= 0 || M <= 0 || H < 4 || W < 4) return 0;
    if (H % 4 || W % 4 || M % W4BM || Cr % W4KC) return 0;
    const int TW = W / 4;
    if (TW > W4BT || W4BT % TW) return 0;
    if (((H / 4) * TW) % W4BT || Cr > W4MAXC) return 0;   // a block's 32 tiles lie in one image; its scale table
    if ((long)B * Cr * H * W >= (1l << 29) || (long)B * M * H * W >= (1l << 29)) return 0;
    return g2s_wino4_weights_floats(M, Cr) < ((size_t)1 << 29);
}

extern "C" int g2s_conv3x3_wino4(const float *x, const float *U, const float *in_scale, const float *out_scale,
                                 const float *bias, const float *noise, const float *noise_w, float *y, int B, int Cr,
                                 int M, int H, int W, int act, float alpha, float gain, int splitk, float *ws,
                                 int64_t ws_floats, g2s_stream_t stream) {
    G2S_REQUIRE(x && U && y, "x, U, y must not be NULL");
    G2S_REQUIRE(g2s_wino4_supported(B, Cr, M, H, W), "shape not supported by the F(4x4,3x3) kernel");
    G2S_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-ReLU)");
    G2S_REQUIRE(!noise || noise_w, "noise needs noise_w");
    Wino4Desc d{};
    d.x = x;
    d.U = U;
    d.in_scale = in_scale;
    d.out_scale = out_scale;
    d.bias = bias;
    d.noise = noise;
    d.noise_w = noise_w;
    d.y = y;
    d.B = B;
    d.Cr = Cr;
    d.M = M;
    d.H = H;
    d.W = W;
    d.TH = H / 4;
    d.TW = W / 4;
    d.ktiles = Cr / W4KC;
    d.act = act;
    d.act_alpha = alpha;
    d.act_gain = gain;
    const int tiles = (M / W4BM) * cdiv((long)B * d.TH * d.TW, W4BT);
    const size_t y_floats = (size_t)B * M * H * W;
    // Partition: whole tiles, or split-K slices that store side by side in the workspace and meet in the
    // reduce pass (split_reduce.h) when whole tiles would leave CUs idle.  splitk = 0: the built-in choice.
    constexpr int NCU = 256;
    if (splitk <= 0) {
        splitk = 1;
        if (tiles < NCU) {
            const int want = std::min(NCU / tiles, SPLIT_REDUCE_MAX);
            if (want >= 2 && d.ktiles / want >= 16) splitk = want;
        }
    }
    splitk = std::min(splitk, d.ktiles);
    splitk = cdiv(d.ktiles, cdiv(d.ktiles, splitk));   // no empty slices
    if (splitk > 1 && !(ws && splitk <= SPLIT_REDUCE_MAX && (size_t)splitk * y_floats <= (size_t)ws_floats)) splitk = 1;
    d.splitk = splitk;
    if (splitk > 1) {
        d.part = ws;
        d.part_n = (long)y_floats;
    }
    hipStream_t st = as_stream(stream);
    dim3 grid(tiles, splitk, 1);
    if (in_scale) wino4_kernel<true><<<grid, W4THREADS, 0, st>>>(d);
    else wino4_kernel<false><<<grid, W4THREADS, 0, st>>>(d);
    int rc = check_launch("g2s_conv3x3_wino4");
    if (rc == G2S_OK && splitk > 1)
        return split_reduce_launch(ws, splitk, (int64_t)y_floats, y, bias, (int64_t)H * W, M, act, alpha, gain, stream,
                                   noise, noise_w);
    return rc;
}
