// winograd.hip — 3x3 stride-1 convolution (padding 1) as Winograd F(2x2, 3x3) on the fp32 matrix
// cores of gfx950.
//
// Serves the stride-1 3x3 layers of the FROZEN networks of a GAN2Shape step — the generator's plain
// StyledConvs (stylegan2-pytorch/model.py:285-289), the discriminator's ResBlock conv1
// (model.py:679-697), the whole VGG16 trunk of LPIPS (lpips/pretrained_networks.py:97-135) — and
// their data-gradients: two thirds of the step's convolution FLOP.  Same contract as g2s_modconv
// (input scale = style, output scale = demodulation, optional bias + leaky-ReLU epilogue), results
// equal up to fp32 rounding (exact-arithmetic identity; transform constants are 0, +-1, +-1/2).
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        per 2x2 output tile, 4x4 input patch d
//
// i.e. 16 independent GEMMs  M_p[Cout][tiles] = U_p[Cout][Cin] V_p[Cin][tiles]  (p = 4i + j): 16
// multiplications per 4 outputs instead of 36 — 2.25x fewer MFMA operations than the direct
// implicit GEMM.
//
//   * U = G g G^T is computed ONCE per weight tensor (g2s_wino_weights; the weights of G / D / VGG
//     are constants of the training step) into a tiled layout [m-tile][k-tile][i][4 ch][64 m][j]
//     (p = 4i + j), so a K tile of one m-tile is one contiguous 16 KB block in exactly its LDS
//     order: it goes global -> LDS by direct DMA (global_load_lds_dwordx4: no registers, no
//     VGPR->LDS transfer), and one ds_read_b128 fetches the MFMA operands of 4 positions.
//   * V = B^T d B is never materialised in memory: each thread gathers the 4x4 patch of one
//     (tile, channel) from global memory (buffer loads with loop-invariant offsets; the hardware
//     range check supplies the zero padding), applies the style scale, transforms in registers and
//     writes the 16 values to the 16 LDS planes.
//   * Workgroup = 64 output channels x 64 tiles (256 output pixels), 4 waves (2 x 2), each wave owns
//     a 32 x 32 block of ALL 16 positions: 16 accumulators of v_mfma_f32_32x32x2_f32 = 256
//     registers per lane, one wave per SIMD.  The output transform A^T M A then happens entirely in
//     registers (the 16 positions of one (channel, tile) live in one lane) and feeds the
//     demodulation / bias / activation epilogue; outputs leave as float2 rows.
//   * K tile = 4 channels = 32 MFMAs per wave (2048 matrix-pipe cycles); the staging of tile t+1
//     and the loads of tile t+2 are issued between those MFMAs, one hand-placed slice per MFMA gap
//     (one wave per SIMD issues in order: whatever is not between two MFMAs stalls the matrix pipe);
//     V double-buffered in LDS, U in a ring of three (its DMA runs two tiles ahead), one raw
//     s_barrier per K tile.
//   * Two waves per SIMD (the positions of a block split over two waves, 128 accumulator registers each) was built
//     and measured in round 4: correct, 1.16 - 1.27x slower — the bare loop of two waves sharing the matrix pipe
//     (16 MFMAs + 8 fragment reads each, nothing else) takes as long per K tile as this kernel's complete loop
//     (profiles/r04_experiments_dropped.txt).
//   * Partition: whole tiles, split-K, or stream-K (equal runs of (tile, K tile) units over 256
//     workgroups, a run may cross tile borders; partial sums meet by float atomics in a cleared y).
#include <algorithm>
#include "g2s_common.h"
#include "split_reduce.h"
#include "xcd_tile.h"

namespace g2s {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int WBM = 64;       // output channels per workgroup
constexpr int WBT = 64;       // 2x2 output tiles per workgroup
constexpr int WKC = 4;        // channels per K tile
constexpr int WTHREADS = 256;
constexpr int WBLOCK = 16 * WKC * WBM;  // floats of one (m-tile, k-tile) block of U

struct WinoDesc {
    const float *x, *U, *in_scale, *out_scale, *bias;
    float *y;
    int B, Cr, M, H, W;
    int TH, TW;          // tiles per image (ceil(H/2), ceil(W/2))
    int ktiles, splitk;
    int upw;             // stream-K: K-tile units per workgroup (0 = one (tile, K slice) per workgroup)
    int units;           // stream-K: tiles * ktiles
    int rr;              // persistent whole tiles over exactly 256 workgroups (see the kernel)
    int act;
    float act_alpha, act_gain;
    float *part;         // split-K with a workspace: slice s stores to part[s * part_n + offset in y]
    long part_n;
    // NoiseInjection of StyledConv in the epilogue (with the bias): + noise_w[0] * noise[oy * W + ox]; NULL: none
    const float *noise, *noise_w;
};

// PARTIAL: Cr is not a multiple of the K tile (the last tile's surplus channels read as zero).
// FAST: the 64 tiles of a workgroup are whole tile rows of one image (64 % TW == 0, W even): a lane
// then loads only the two MIDDLE columns of its 4x4 patch (one aligned 8-byte load per patch row,
// the wave reads 512 contiguous bytes) and takes the outer columns from its neighbour lanes — 4
// fully coalesced loads per K tile instead of 16 strided ones (the vector L1 is the busiest unit of
// this kernel: the matrix work per loaded byte is 2.25x smaller than in the direct GEMM).
template <bool SCALE, bool PARTIAL, bool FAST>
__global__ __launch_bounds__(WTHREADS) void wino_kernel(WinoDesc d) {
    __shared__ __attribute__((aligned(16))) float Us[3][16 * WKC * WBM];   // ring of 3: DMA runs 2 tiles ahead
    __shared__ __attribute__((aligned(16))) float Vs[2][16 * WKC * WBT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (d.M + WBM - 1) / WBM;
    const int per_img = d.TH * d.TW, Ntiles = d.B * per_img;
    // Work of this workgroup: a run of (tile, K tile) units.  Plain mode: one tile, one K slice
    // (blockIdx.y of split-K).  Stream-K mode: an equal share of ALL units, crossing tile borders —
    // every CU gets the same amount of matrix work whatever the tile count (one workgroup per CU
    // fits: a launch of 1.1 rounds of equal tiles would otherwise take 2).
    int unit, unit_end;
    const int g = xcd_logical_tile();   // stream-K: index of this workgroup's run
    {
        if (d.rr) {
            // 256 workgroups = 8 XCDs x 32 (g is XCD-major): XCD x owns the x-th eighth of the tiles — the chunk it
            // gets in a plain launch — and its 32 workgroups walk that chunk side by side
            const int tiles = d.units / d.ktiles, per_xcd = (tiles + 7) >> 3;
            const int x = g >> 5, l = g & 31;
            unit = (x * per_xcd + l) * d.ktiles;
            unit_end = min(tiles, (x + 1) * per_xcd) * d.ktiles;
        } else if (d.upw) {
            unit = g * d.upw;
            unit_end = min(d.units, unit + d.upw);
        } else {
            const int per = (d.ktiles + d.splitk - 1) / d.splitk;
            const int k0 = blockIdx.y * per;
            unit = g * d.ktiles + k0;
            unit_end = g * d.ktiles + min(d.ktiles, k0 + per);
        }
    }
    const int HW = d.H * d.W;
    constexpr int OOB = 0x7fffffff;

    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)d.x, 0, d.B * d.Cr * HW * 4, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc((void *)(SCALE ? d.in_scale : d.x), 0,
                                                       SCALE ? d.B * d.Cr * 4 : 4, 0x00020000);

    const int kc = wave;
    while (unit < unit_end) {
        const int tile_id = unit / d.ktiles;
        const int kt_begin = unit - tile_id * d.ktiles;
        const int kt_end = min(d.ktiles, kt_begin + (unit_end - unit));
        unit += kt_end - kt_begin;
        if (d.rr) unit += 31 * d.ktiles;   // the next tile of this workgroup's share: 32 tiles further on
        const int mt = tile_id % tiles_m, nt = tile_id / tiles_m;
        __syncthreads();   // the previous segment's LDS reads are done before this one's staging
        // ---- input patch of this thread: tile t = lane of the block, channel kc = wave of the K tile
        constexpr int ND = FAST ? 4 : 16;   // loads per patch
        int offD[ND];
        int offS = OOB;
        bool first_col = false, last_col = false;
        {
            const int n = nt * WBT + lane;
            const bool valid = n < Ntiles;
            const int b = valid ? n / per_img : 0, r = valid ? n % per_img : 0;
            const int ty = r / d.TW, tx = r % d.TW;
            const int base = (b * d.Cr + kc) * HW;
            first_col = tx == 0;
            last_col = tx == d.TW - 1;
    #pragma unroll
            for (int i = 0; i < 4; i++) {
                const int iy = 2 * ty - 1 + i;
                const bool row_ok = valid & (iy >= 0) & (iy < d.H);
                if constexpr (FAST) {
                    offD[i] = row_ok ? (base + iy * d.W + 2 * tx) * 4 : OOB;   // columns 2tx, 2tx+1
                } else {
    #pragma unroll
                    for (int j = 0; j < 4; j++) {
                        const int ix = 2 * tx - 1 + j;
                        offD[i * 4 + j] = (row_ok & (ix >= 0) & (ix < d.W)) ? (base + iy * d.W + ix) * 4 : OOB;
                    }
                }
            }
            if (valid) offS = (b * d.Cr + kc) * 4;
        }
        // U block of K tile kt -> LDS buffer `buf` by direct DMA (global_load_lds_dwordx4: no registers,
        // no VGPR->LDS transfer): 16 wave-instructions of 1 KB, chunk e * 4 + wave for e = 0..3
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const char *ubase = reinterpret_cast<const char *>(d.U) + (size_t)mt * d.ktiles * WBLOCK * 4 + lane * 16;
        auto dma_u = [&](int kt, int buf, const int e) {
            const int chunk = e * 4 + wave_u;   // 1 KB chunk of the 16 KB block
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(ubase + (size_t)kt * WBLOCK * 4 + chunk * 1024),
                (__attribute__((address_space(3))) void *)(&Us[0][0] + buf * (16 * WKC * WBM) + chunk * 256), 16, 0, 0);
        };

        float rd[16], rs = 1.0f;    // FAST: rd[i*4+1], rd[i*4+2] hold the loaded middle columns
        auto load_tile = [&](int kt) {
            // surplus channels of the last tile: an out-of-range offset makes the hardware return 0
            const int kill = (PARTIAL && kt * WKC + kc >= d.Cr) ? OOB : 0;  // wave-uniform
            const int so = kt * WKC * HW * 4;
            if constexpr (FAST) {
    #pragma unroll
                for (int i = 0; i < 4; i++) {
                    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, offD[i] | kill, so, 0);
                    rd[i * 4 + 1] = __uint_as_float(v.x);
                    rd[i * 4 + 2] = __uint_as_float(v.y);
                }
            } else {
    #pragma unroll
                for (int e = 0; e < 16; e++)
                    rd[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, offD[e] | kill, so, 0));
            }
            if constexpr (SCALE)
                rs = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, offS | kill, kt * WKC * 4, 0));
        };
        // B^T d B of the registers -> the 16 planes of buffer `buf`, as [i][channel][tile][j]
        auto stage_tile = [&](int buf) {
            float t[16];
            if constexpr (FAST) {
                // outer columns from the neighbour lanes = neighbour tiles of the same tile row (whole-wave
                // DPP shifts: no LDS traffic); zero at the image border
    #pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int left = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rd[i * 4 + 2]), 0x138, 0xf, 0xf, false);
                    const int right = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, rd[i * 4 + 1]), 0x130, 0xf, 0xf, false);
                    rd[i * 4 + 0] = first_col ? 0.0f : __builtin_bit_cast(float, left);
                    rd[i * 4 + 3] = last_col ? 0.0f : __builtin_bit_cast(float, right);
                }
            }
            if constexpr (SCALE) {
    #pragma unroll
                for (int e = 0; e < 16; e++) rd[e] *= rs;
            }
    #pragma unroll
            for (int j = 0; j < 4; j++) {
                t[0 * 4 + j] = rd[0 * 4 + j] - rd[2 * 4 + j];
                t[1 * 4 + j] = rd[1 * 4 + j] + rd[2 * 4 + j];
                t[2 * 4 + j] = rd[2 * 4 + j] - rd[1 * 4 + j];
                t[3 * 4 + j] = rd[1 * 4 + j] - rd[3 * 4 + j];
            }
            f32x4 *vb = reinterpret_cast<f32x4 *>(&Vs[0][0] + buf * (16 * WKC * WBT)) + kc * WBT + lane;
    #pragma unroll
            for (int i = 0; i < 4; i++)
                vb[i * WKC * WBT] = f32x4{t[i * 4 + 0] - t[i * 4 + 2], t[i * 4 + 1] + t[i * 4 + 2],
                                          t[i * 4 + 2] - t[i * 4 + 1], t[i * 4 + 1] - t[i * 4 + 3]};
        };

        f32x16 acc[16];
    #pragma unroll
        for (int p = 0; p < 16; p++)
    #pragma unroll
            for (int r = 0; r < 16; r++) acc[p][r] = 0.0f;

        const int kt_last = kt_end - 1;
        // operand fragments of one k-step (2 channels): one 16-byte read per operand and patch row i
        const int aoff = lk * WBM + wm * 32 + l31;   // float4 index; + (i * WKC + 2 ks) * WBM
        const int boff = lk * WBT + wn * 32 + l31;
        f32x4 fa0[4], fb0[4], fa1[4], fb1[4];
        auto read_frags = [&](int buf, const int ks, f32x4 (&fa)[4], f32x4 (&fb)[4]) {
            const f32x4 *ua = reinterpret_cast<const f32x4 *>(&Us[0][0] + buf * (16 * WKC * WBM)) + aoff;
            const f32x4 *vb = reinterpret_cast<const f32x4 *>(&Vs[0][0] + buf * (16 * WKC * WBT)) + boff;
    #pragma unroll
            for (int i = 0; i < 4; i++) {
                fa[i] = ua[(i * WKC + 2 * ks) * WBM];
                fb[i] = vb[(i * WKC + 2 * ks) * WBT];
            }
        };

    #pragma unroll
        for (int e = 0; e < 4; e++) dma_u(kt_begin, 0, e);
    #pragma unroll
        for (int e = 0; e < 4; e++) dma_u(min(kt_begin + 1, kt_last), 1, e);
        load_tile(kt_begin);
        stage_tile(0);
        load_tile(min(kt_begin + 1, kt_last));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        read_frags(0, 0, fa0, fb0);

        // Software pipeline of one K tile (ONE loop body: the 256 accumulator registers must not be
        // shuffled between copies of the body).  A wave issues in order and a second MFMA waits ~64
        // cycles for the matrix pipe, so everything else sits BETWEEN the MFMAs:
        //   * the U block of tile kt+1 goes to the free LDS buffer by DMA, tile kt+1's patch (in
        //     registers since the previous iteration: a full MFMA phase of latency budget) is
        //     transformed and written there, and the registers are re-loaded with tile kt+2;
        //   * the fragments of k-step 1 are read under the MFMAs of k-step 0;
        //   * after the hand-over barrier the k-step-0 fragments of tile kt+1 are read under the last 4
        //     MFMAs of k-step 1, so that the matrix pipe does not wait for LDS after the barrier.
        // Hand-placed schedule: a wave issues in order and a second MFMA waits ~64 cycles for the
        // matrix pipe, so each of the 32 MFMA gaps of a K tile gets ONE small slice of the other
        // work (<= ~8 instructions), pinned with sched_barrier(0):
        //   gaps  0-3   read 2 fragments of k-step 1 each
        //   gaps  4-7   patch row i: outer columns from the neighbour lanes, style scale; one DMA chunk of
        //               the U block of tile kt+2 (ring of 3 LDS buffers: everything a wave loads in
        //               iteration kt is first needed in iteration kt+1, a whole MFMA phase later)
        //   gaps  8-11  row transform of patch column j
        //   gaps 12-15  reload patch row i (kt+2)
        //   gaps 16-23  column transform of row i (even gap), its 16-byte LDS write (odd gap)
        //   gaps 24-27  nothing
        //   barrier (LDS hand-over only: no vmcnt wait)
        //   gaps 28-31  read the k-step-0 fragments of tile kt+1
        int cur = 0;
        int ub0 = 0, ub1 = 1, ub2 = 2;   // U ring: tile kt, kt+1, kt+2
        for (int kt = kt_begin; kt < kt_end; kt++, cur ^= 1) {
            const f32x4 *ua = reinterpret_cast<const f32x4 *>(&Us[0][0] + ub0 * (16 * WKC * WBM)) + aoff;
            const f32x4 *va = reinterpret_cast<const f32x4 *>(&Vs[0][0] + cur * (16 * WKC * WBT)) + boff;
            const f32x4 *un = reinterpret_cast<const f32x4 *>(&Us[0][0] + ub1 * (16 * WKC * WBM)) + aoff;
            const f32x4 *vn = reinterpret_cast<const f32x4 *>(&Vs[0][0] + (cur ^ 1) * (16 * WKC * WBT)) + boff;
            f32x4 *vw = reinterpret_cast<f32x4 *>(&Vs[0][0] + (cur ^ 1) * (16 * WKC * WBT)) + kc * WBT + lane;
            const int kt2 = min(kt + 2, kt_last);
            const int kill2 = (PARTIAL && kt2 * WKC + kc >= d.Cr) ? OOB : 0;
            const int so2 = kt2 * WKC * HW * 4;
            float t[16];
            f32x4 v4;
    #pragma unroll
            for (int q = 0; q < 32; q++) {
                if (q == 28) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                if (q < 16) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[q >> 2][q & 3], fb0[q >> 2][q & 3], acc[q], 0, 0, 0);
                else acc[q - 16] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[(q - 16) >> 2][q & 3], fb1[(q - 16) >> 2][q & 3], acc[q - 16], 0, 0, 0);
                if (q < 4) {
                    fa1[q] = ua[(q * WKC + 2) * WBM];
                    fb1[q] = va[(q * WKC + 2) * WBT];
                } else if (q < 8) {
                    const int i = q - 4;
                    if constexpr (FAST) {
                        // bound_ctrl: lanes without a source read 0 (no `old` operand to initialise)
                        const int left = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, rd[i * 4 + 2]), 0x138, 0xf, 0xf, true);
                        const int right = __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, rd[i * 4 + 1]), 0x130, 0xf, 0xf, true);
                        rd[i * 4 + 0] = first_col ? 0.0f : __builtin_bit_cast(float, left);
                        rd[i * 4 + 3] = last_col ? 0.0f : __builtin_bit_cast(float, right);
                    }
                    if constexpr (SCALE) {
    #pragma unroll
                        for (int j = 0; j < 4; j++) rd[i * 4 + j] *= rs;
                    }
                    dma_u(kt2, ub2, i);   // U of tile kt+2 -> the ring slot that held tile kt-1
                } else if (q < 12) {
                    const int j = q - 8;
                    t[0 * 4 + j] = rd[0 * 4 + j] - rd[2 * 4 + j];
                    t[1 * 4 + j] = rd[1 * 4 + j] + rd[2 * 4 + j];
                    t[2 * 4 + j] = rd[2 * 4 + j] - rd[1 * 4 + j];
                    t[3 * 4 + j] = rd[1 * 4 + j] - rd[3 * 4 + j];
                } else if (q < 16) {
                    // registers of patch row i are free (row transform done): reload them with tile kt+2
                    const int i = q - 12;
                    if constexpr (FAST) {
                        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rx, offD[i] | kill2, so2, 0);
                        rd[i * 4 + 1] = __uint_as_float(v.x);
                        rd[i * 4 + 2] = __uint_as_float(v.y);
                    } else {
    #pragma unroll
                        for (int j = 0; j < 4; j++)
                            rd[i * 4 + j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, offD[i * 4 + j] | kill2, so2, 0));
                    }
                    if (SCALE && q == 15)
                        rs = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, offS | kill2, kt2 * WKC * 4, 0));
                } else if (q < 24) {
                    const int i = (q - 16) >> 1;
                    if ((q & 1) == 0) {   // column transform of row i ...
                        v4[0] = t[i * 4 + 0] - t[i * 4 + 2];
                        v4[1] = t[i * 4 + 1] + t[i * 4 + 2];
                        v4[2] = t[i * 4 + 2] - t[i * 4 + 1];
                        v4[3] = t[i * 4 + 1] - t[i * 4 + 3];
                    } else {              // ... and its 16-byte LDS write in the next gap
                        vw[i * WKC * WBT] = v4;
                    }
                } else if (q >= 28) {
                    const int i = q - 28;
                    fa0[i] = un[(i * WKC) * WBM];
                    fb0[i] = vn[(i * WKC) * WBT];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            const int ufree = ub0;
            ub0 = ub1;
            ub1 = ub2;
            ub2 = ufree;
        }

        // ---- epilogue: output transform A^T M A in registers, scale / bias / activation, store.
        // C layout of the 32x32 MFMA: row m = (r & 3) + 8 (r >> 2) + 4 lk, column n = l31.
        const int n = nt * WBT + wn * 32 + l31;
        if (n >= Ntiles) continue;
        const int b = n / per_img, rr = n % per_img;
        const int oy = 2 * (rr / d.TW), ox = 2 * (rr % d.TW);
        const bool row1 = oy + 1 < d.H, col1 = ox + 1 < d.W;
        const bool split = kt_begin != 0 || kt_end != d.ktiles;   // partial sum of the tile
        const bool vec = col1 && (d.W & 1) == 0 && !split;
        float *yb = d.y + ((size_t)b * d.M * d.H + oy) * d.W + ox;
        const float *ob = d.out_scale ? d.out_scale + (size_t)b * d.M : nullptr;
        float nz[4] = {0.f, 0.f, 0.f, 0.f};     // the noise term of this lane's 2x2 outputs (same for every channel)
        if (d.noise && !(split && d.part != nullptr)) {
            const float nw = d.noise_w[0];
            const float *np_ = d.noise + oy * d.W + ox;
            nz[0] = nw * np_[0];
            if (col1) nz[1] = nw * np_[1];
            if (row1) nz[2] = nw * np_[d.W];
            if (row1 && col1) nz[3] = nw * np_[d.W + 1];
        }
    #pragma unroll
        for (int r = 0; r < 16; r++) {
            const int m = mt * WBM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (m >= d.M) continue;
            float s0[4], s1[4];
    #pragma unroll
            for (int j = 0; j < 4; j++) {
                s0[j] = acc[0 + j][r] + acc[4 + j][r] + acc[8 + j][r];
                s1[j] = acc[4 + j][r] - acc[8 + j][r] - acc[12 + j][r];
            }
            float y[4] = {s0[0] + s0[1] + s0[2], s0[1] - s0[2] - s0[3], s1[0] + s1[1] + s1[2], s1[1] - s1[2] - s1[3]};
            const float sc = ob ? ob[m] : 1.0f;
            // a partial sum that goes to a workspace slot carries the (linear) output scale only: the
            // reduce pass adds bias and activation; whole tiles and the atomic path finish here (for
            // the atomic path the host has cleared bias / activation: they run as a deferred launch)
            const bool raw = split && d.part != nullptr;
            const float bi = (d.bias && !raw) ? d.bias[m] : 0.0f;
    #pragma unroll
            for (int q = 0; q < 4; q++) {
                float v = y[q] * sc + bi + nz[q];
                if (d.act && !raw) v = (v > 0.0f ? v : v * d.act_alpha) * d.act_gain;
                y[q] = v;
            }
            float *dst = yb + (size_t)m * HW;
            if (vec) {
                *reinterpret_cast<float2 *>(dst) = float2{y[0], y[1]};
                if (row1) *reinterpret_cast<float2 *>(dst + d.W) = float2{y[2], y[3]};
            } else if (split && d.part) {
                // slot: the split-K slice, or the position of this run among the runs that share the tile
                const int slot = d.upw ? g - (tile_id * d.ktiles) / d.upw : (int)blockIdx.y;
                float *pd = d.part + (size_t)slot * d.part_n + (dst - d.y);
                if (col1 && (d.W & 1) == 0) {
                    *reinterpret_cast<float2 *>(pd) = float2{y[0], y[1]};
                    if (row1) *reinterpret_cast<float2 *>(pd + d.W) = float2{y[2], y[3]};
                } else {
                    pd[0] = y[0];
                    if (col1) pd[1] = y[1];
                    if (row1) pd[d.W] = y[2];
                    if (row1 && col1) pd[d.W + 1] = y[3];
                }
            } else if (split) {
                unsafeAtomicAdd(dst, y[0]);
                if (col1) unsafeAtomicAdd(dst + 1, y[1]);
                if (row1) unsafeAtomicAdd(dst + d.W, y[2]);
                if (row1 && col1) unsafeAtomicAdd(dst + d.W + 1, y[3]);
            } else {
                dst[0] = y[0];
                if (col1) dst[1] = y[1];
                if (row1) dst[d.W] = y[2];
                if (row1 && col1) dst[d.W + 1] = y[3];
            }
        }
    }
}

// Second pass of a stream-K launch with a workspace: every output pair (2 pixels of one 2x2 tile)
// finds its tile, the number of runs that shared it, adds their slots, applies bias + activation.
// Tiles that one run computed whole were finished by the convolution kernel and are left alone.
__global__ __launch_bounds__(256) void wino_streamk_reduce_kernel(WinoDesc d) {
    const int W2 = d.W >> 1;                       // W is even on this path
    const long pairs = (long)d.B * d.M * d.H * W2;
    const int tiles_m = (d.M + WBM - 1) / WBM, per_img = d.TH * d.TW;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += (long)gridDim.x * blockDim.x) {
        const int px = (int)(i % W2);
        long r = i / W2;
        const int oy = (int)(r % d.H);
        r /= d.H;
        const int m = (int)(r % d.M), b = (int)(r / d.M);
        const int n = b * per_img + (oy >> 1) * d.TW + px;
        const int tile_id = (n / WBT) * tiles_m + m / WBM;
        const int u0 = tile_id * d.ktiles;
        const int parts = (u0 + d.ktiles - 1) / d.upw - u0 / d.upw + 1;
        if (parts == 1) continue;
        const size_t o = (((size_t)b * d.M + m) * d.H + oy) * d.W + 2 * px;
        float2 acc{0.f, 0.f};
        for (int s = 0; s < parts; s++) {
            const float2 v = *reinterpret_cast<const float2 *>(d.part + (size_t)s * d.part_n + o);
            acc.x += v.x;
            acc.y += v.y;
        }
        const float bi = d.bias ? d.bias[m] : 0.0f;
        acc.x += bi;
        acc.y += bi;
        if (d.noise) {
            const float nw = d.noise_w[0];
            acc.x += nw * d.noise[oy * d.W + 2 * px];
            acc.y += nw * d.noise[oy * d.W + 2 * px + 1];
        }
        if (d.act) {
            acc.x = (acc.x > 0.0f ? acc.x : acc.x * d.act_alpha) * d.act_gain;
            acc.y = (acc.y > 0.0f ? acc.y : acc.y * d.act_alpha) * d.act_gain;
        }
        *reinterpret_cast<float2 *>(d.y + o) = acc;
    }
}

// U = G g G^T of every (output channel m, reduction channel c) into the tiled layout.
// w(m, c, ky, kx) = w[m * w_ms + c * w_ks + (flip ? 8 - (3 ky + kx) : 3 ky + kx)].
__global__ void wino_weights_kernel(const float *w, float *U, int M, int Cr, int w_ms, int w_ks, int flip,
                                    int ktiles) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * Cr) return;
    const int m = idx % M, c = idx / M;
    float g[9];
#pragma unroll
    for (int t = 0; t < 9; t++) g[t] = w[(size_t)m * w_ms + (size_t)c * w_ks + (flip ? 8 - t : t)];
    // rows: G g  (4 x 3)
    float h[12];
#pragma unroll
    for (int j = 0; j < 3; j++) {
        h[0 * 3 + j] = g[0 * 3 + j];
        h[1 * 3 + j] = 0.5f * (g[0 * 3 + j] + g[1 * 3 + j] + g[2 * 3 + j]);
        h[2 * 3 + j] = 0.5f * (g[0 * 3 + j] - g[1 * 3 + j] + g[2 * 3 + j]);
        h[3 * 3 + j] = g[2 * 3 + j];
    }
    // block [i][channel][m][j] (p = 4 i + j): row i of G g G^T is one float4
    f32x4 *dst = reinterpret_cast<f32x4 *>(U + (size_t)((m / WBM) * ktiles + c / WKC) * WBLOCK) + (c % WKC) * WBM + (m % WBM);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float u0 = h[i * 3 + 0];
        const float u1 = 0.5f * (h[i * 3 + 0] + h[i * 3 + 1] + h[i * 3 + 2]);
        const float u2 = 0.5f * (h[i * 3 + 0] - h[i * 3 + 1] + h[i * 3 + 2]);
        const float u3 = h[i * 3 + 2];
        dst[i * WKC * WBM] = f32x4{u0, u1, u2, u3};
    }
}

}  // namespace g2s

using namespace g2s;

extern "C" size_t g2s_wino_weights_floats(int M, int Cr) {
    if (M <= 0 || Cr <= 0) return 0;
    return (size_t)cdiv(M, WBM) * cdiv(Cr, WKC) * WBLOCK;
}

extern "C" int g2s_wino_weights(const float *w, float *U, int Cout, int Cin, int transpose, g2s_stream_t stream) {
    G2S_REQUIRE(w && U, "w, U must not be NULL");
    G2S_REQUIRE(Cout > 0 && Cin > 0, "sizes must be positive");
    // forward: M = Cout, reduction over Cin, taps as stored.  transpose (data-gradient): M = Cin,
    // reduction over Cout, taps flipped: w'(i, o, ky, kx) = w(o, i, 2 - ky, 2 - kx).
    const int M = transpose ? Cin : Cout, Cr = transpose ? Cout : Cin;
    const int w_ms = transpose ? 9 : Cin * 9, w_ks = transpose ? Cin * 9 : 9;
    hipStream_t st = as_stream(stream);
    const size_t n = g2s_wino_weights_floats(M, Cr);
    if (hipMemsetAsync(U, 0, n * sizeof(float), st) != hipSuccess) return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(U) failed");
    wino_weights_kernel<<<cdiv((long)M * Cr, 256), 256, 0, st>>>(w, U, M, Cr, w_ms, w_ks, transpose ? 1 : 0, cdiv(Cr, WKC));
    return check_launch("g2s_wino_weights");
}

static int wino_launch(const float *x, const float *U, const float *in_scale, const float *out_scale,
                       const float *bias, const float *noise, const float *noise_w, float *y, int B, int Cr, int M,
                       int H, int W, int act, float alpha, float gain, int splitk, float *ws, int64_t ws_floats,
                       g2s_stream_t stream) {
    G2S_REQUIRE(x && U && y, "x, U, y must not be NULL");
    G2S_REQUIRE(B > 0 && Cr > 0 && M > 0 && H >= 2 && W >= 2, "sizes must be positive (H, W >= 2)");
    G2S_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-ReLU)");
    G2S_REQUIRE((long)B * Cr * H * W < (1l << 29) && (long)B * M * H * W < (1l << 29) &&
                    g2s_wino_weights_floats(M, Cr) < ((size_t)1 << 29),
                "problem too large for 32-bit byte offsets");
    WinoDesc d{};
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
    d.TH = (H + 1) / 2;
    d.TW = (W + 1) / 2;
    d.ktiles = cdiv(Cr, WKC);
    d.act = act;
    d.act_alpha = alpha;
    d.act_gain = gain;
    const int tiles = cdiv(M, WBM) * cdiv((long)B * d.TH * d.TW, WBT);
    // Partition.  One workgroup per CU is resident (512 registers per lane), all tiles cost the same:
    // a launch of `tiles` workgroups takes ceil(tiles / 256) rounds.  splitk > 0: that K split of
    // every tile (tests, tuning).  splitk = 0: whole tiles when the rounds are well filled, else
    // stream-K — 256 workgroups with equal runs of (tile, K tile) units.
    constexpr int NCU = 256;
    int grid_x = tiles;
    d.upw = 0;
    d.units = tiles * d.ktiles;
    bool partial_sums;
    const size_t y_floats = (size_t)B * M * H * W;
    bool use_part = false;
    // Whole tiles over two or more rounds: 256 persistent workgroups walk consecutive tiles instead of the
    // dispatcher starting `tiles` workgroups one round after the other (same arithmetic per tile; measured
    // 2-6 % on the short-K launches, e.g. 18 x 64 -> 64 @ 128^2: 144 -> 136 us).
    const bool whole = splitk == 1 || (splitk == 0 && (double)tiles / ((double)cdiv(tiles, NCU) * NCU) >= 0.9);
    if (whole && tiles >= 2 * NCU) splitk = -10000 - NCU;
    if (splitk > 0) {
        splitk = std::min(splitk, d.ktiles);
        splitk = cdiv(d.ktiles, cdiv(d.ktiles, splitk));   // no empty slices
        partial_sums = splitk > 1;
        // with a workspace the slices store their partial sums side by side and a reduce pass finishes
        // the launch (split_reduce.h): no clear of y, no atomic burst, no separate epilogue launch
        use_part = partial_sums && ws && splitk <= SPLIT_REDUCE_MAX && (size_t)splitk * y_floats <= (size_t)ws_floats;
        if (use_part) {
            d.part = ws;
            d.part_n = (long)y_floats;
        }
    } else if (splitk <= -10000) {
        // persistent whole tiles: 256 workgroups, each walking WHOLE tiles of its XCD's chunk (no partial sums, no
        // reduce pass)
        d.rr = 1;
        grid_x = NCU;
        splitk = 1;
        partial_sums = false;
    } else if (splitk < 0) {   // stream-K over -splitk workgroups (tests)
        d.upw = cdiv(d.units, std::min(-splitk, d.units));
        grid_x = cdiv(d.units, d.upw);
        splitk = 1;
        partial_sums = true;
    } else {
        splitk = 1;
        const int rounds = cdiv(tiles, NCU);
        const double fill = (double)tiles / ((double)rounds * NCU);
        const int upw = cdiv(d.units, NCU);
        if (fill < 0.9 && upw >= 4) {
            d.upw = upw;
            grid_x = cdiv(d.units, upw);
        }
        partial_sums = d.upw != 0;
    }
    // stream-K with a workspace: every run stores its share of a split tile to a slot of its own
    // (at most ceil(ktiles / upw) + 1 runs meet in a tile) and wino_streamk_reduce_kernel finishes them
    bool streamk_part = false;
    if (d.upw && partial_sums && ws && W % 2 == 0) {
        const int slots = cdiv(d.ktiles, d.upw) + 1;
        if (slots <= SPLIT_REDUCE_MAX && (size_t)slots * y_floats <= (size_t)ws_floats) {
            streamk_part = use_part = true;
            d.part = ws;
            d.part_n = (long)y_floats;
        }
    }
    if (deterministic() && partial_sums && !use_part) {
        // partial sums would meet by float atomics (no workspace): whole tiles instead
        splitk = 1;
        grid_x = tiles;
        d.upw = 0;
        partial_sums = false;
    }
    const bool deferred = partial_sums && !streamk_part && (bias != nullptr || act != 0 || noise != nullptr);
    if (deferred) {
        d.bias = nullptr;
        d.noise = d.noise_w = nullptr;
        d.act = 0;
    }
    d.splitk = splitk;
    hipStream_t st = as_stream(stream);
    if (partial_sums && !use_part && hipMemsetAsync(y, 0, y_floats * sizeof(float), st) != hipSuccess)
        return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(y) failed");
    dim3 grid(grid_x, splitk, 1);
    const bool partial = Cr % WKC != 0;
    // whole tile rows per workgroup and 8-byte aligned row pairs: see FAST above
    const bool fast = d.TW <= WBT && WBT % d.TW == 0 && W % 2 == 0;
#define G2S_WINO_LAUNCH(S_, P_, F_) wino_kernel<S_, P_, F_><<<grid, WTHREADS, 0, st>>>(d)
    if (in_scale) {
        if (partial) { if (fast) G2S_WINO_LAUNCH(true, true, true); else G2S_WINO_LAUNCH(true, true, false); }
        else { if (fast) G2S_WINO_LAUNCH(true, false, true); else G2S_WINO_LAUNCH(true, false, false); }
    } else {
        if (partial) { if (fast) G2S_WINO_LAUNCH(false, true, true); else G2S_WINO_LAUNCH(false, true, false); }
        else { if (fast) G2S_WINO_LAUNCH(false, false, true); else G2S_WINO_LAUNCH(false, false, false); }
    }
#undef G2S_WINO_LAUNCH
    int rc = check_launch("g2s_conv3x3_wino");
    if (rc == G2S_OK && streamk_part) {
        const long pairs = (long)y_floats / 2;
        wino_streamk_reduce_kernel<<<(unsigned)std::min<long>((pairs + 255) / 256, 256 * 8), 256, 0, st>>>(d);
        return check_launch("g2s_conv3x3_wino (stream-K reduce)");
    }
    if (rc == G2S_OK && use_part)
        return split_reduce_launch(ws, splitk, (int64_t)y_floats, y, bias, (int64_t)H * W, M, act, alpha, gain, stream,
                                   noise, noise_w);
    if (rc != G2S_OK || !deferred) return rc;
    if (noise)
        return g2s_noise_bias_act(y, noise, noise_w, bias, y, B, M, H * W, alpha, gain, stream);
    return g2s_fused_bias_act(y, bias, nullptr, y, (int64_t)B * M * H * W, (int64_t)H * W, M, act ? 3 : 1, 0,
                              alpha, act ? gain : 1.0f, G2S_F32, stream);
}

extern "C" int g2s_conv3x3_wino(const float *x, const float *U, const float *in_scale, const float *out_scale,
                                const float *bias, float *y, int B, int Cr, int M, int H, int W, int act,
                                float alpha, float gain, int splitk, float *ws, int64_t ws_floats,
                                g2s_stream_t stream) {
    return wino_launch(x, U, in_scale, out_scale, bias, nullptr, nullptr, y, B, Cr, M, H, W, act, alpha, gain, splitk,
                       ws, ws_floats, stream);
}

// The same with the whole StyledConv tail (stylegan2-pytorch/model.py:349-355) in the epilogue:
// y = gain * leaky_relu(out_scale * conv(in_scale * x) + noise_w[0] * noise[h, w] + bias[m], alpha).
extern "C" int g2s_conv3x3_wino_nba(const float *x, const float *U, const float *in_scale, const float *out_scale,
                                    const float *bias, const float *noise, const float *noise_w, float *y, int B, int Cr,
                                    int M, int H, int W, float alpha, float gain, int splitk, float *ws,
                                    int64_t ws_floats, g2s_stream_t stream) {
    G2S_REQUIRE(bias && noise && noise_w, "bias, noise and noise_w must not be NULL");
    return wino_launch(x, U, in_scale, out_scale, bias, noise, noise_w, y, B, Cr, M, H, W, 1, alpha, gain, splitk, ws,
                       ws_floats, stream);
}
