// modconv.hip — StyleGAN2 modulated convolution as an fp32-MFMA implicit GEMM for gfx950.
//
// Replaces the grouped F.conv2d / F.conv_transpose2d of ModulatedConv2d.forward
// (GAN2Shape/stylegan2/stylegan2-pytorch/model.py:250-291) and their data-gradients.
// The reference materialises one weight set per sample (B*Cout*Cin*9 floats) and runs a grouped
// convolution with B groups.  Here the weights stay shared and batch-independent:
//     y[b,o] = demod[b,o] * sum_{i,t} W[o,i,t] * (style[b,i] * x[b,i,.])
// so the whole batch is ONE GEMM  C[M = out channels][N = B*OH*OW] = A[M][K] * Bm[K][N]  with
//   A  = the weight tensor itself, gathered in place (no packing; the transposed use for the
//        data-gradient only changes two strides),
//   Bm = im2col of x, never materialised: gathered from global memory straight into LDS with the
//        per-(b,i) style scale applied on the way,
//   epilogue = per-(b,o) demodulation scale.
// v_mfma_f32_32x32x2_f32 (exact fp32, 256 FLOP/clk/CU) — results differ from an fp32 FMA chain
// only by summation order.  LDS holds k-major operand tiles so that the MFMA operand fetch is a
// conflict-free ds_read_b32 (lanes 0-31 consecutive, lanes 32-63 the next k row).  Global loads
// for tile t+1 are issued before the MFMAs of tile t and written to the other LDS buffer after
// them (one barrier per K tile).
//
// The transposed convolution (UP2) is run as its four output-parity classes (4/2/2/1 taps): no
// multiplications by inserted zeros; classes are blockIdx.z of one launch.
// Small layers (4x4 .. 16x16) use 64x64 tiles and split-K with float atomics to fill 256 CUs.
#include "g2s_common.h"
#include "conv_wgrad_core.h"
#include "thinconv.h"
#include "xcd_tile.h"

namespace g2s {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int NTHREADS = 256;

struct ConvClass {  // one output-parity class (a single class for gather geometries)
    int OH, OW;     // class output extent
    int T;          // taps in this class
    int oy0, ox0;   // offset of the class in the full output
    int tab[25];    // per tap: (dy + 8) | (dx + 8) << 8 | wtap << 16
};

struct ConvDesc {
    const float *x, *w, *in_scale, *out_scale;
    float *y;
    int B, Cr, M;        // batch, reduction channels, output channels
    int H, W;            // input spatial size
    int OHf, OWf;        // full output spatial size
    int w_ms, w_ks;      // strides (in floats) of the m index and the reduction-channel index in w
    int is;              // input stride of the gather (1 or 2)
    int os;              // output stride of a class (1, or 2 for the polyphase classes)
    int ncls;
    int splitk;
    int w_bytes;         // size of ONE group's w in bytes (buffer-resource range)
    // Grouped launch (two structurally identical trained nets run as one, their channels side by
    // side: op/conv.py PairConvFunction): `groups` independent convolutions of Cr -> M channels;
    // x has Cx = groups * Cr channels per sample, y has My = groups * M, w / bias hold the groups
    // back to back.  groups = 1: Cx = Cr, My = M.
    int groups, Cx, My;
    int cls_splitk[4];   // split-K slices of each class (<= splitk = grid.y): lighter classes get fewer
    const float *bias;   // [M] added after out_scale, or NULL
    // StyledConv's NoiseInjection (stylegan2-pytorch/model.py:294-305,349-355) in the epilogue: + noise_w[0] *
    // noise[oy * OWf + ox], one [OHf, OWf] map for all samples and channels; NULL: none
    const float *noise, *noise_w;
    int act;             // 0: none, 1: leaky-ReLU(alpha) * gain applied after the bias
    float act_alpha, act_gain;
    ConvClass cls[4];
};

// K tiles hold WHOLE reduction channels (T taps each): 2 channels x 9 taps = 18 for 3x3, else 16.
// A thread's elements of a tile then keep the same (channel-in-tile, tap) for the entire K loop,
// so tap offsets, bounds checks and LDS addresses are computed ONCE; per tile only the channel base
// moves.  This keeps the VALU work per tile (~60 instructions) far below the MFMA time.
// 5x5 kernels (25 taps) take one channel per tile, padded to 26 rows, in a kernel instance of their
// own (KMAX = 26) so that the common instances keep their smaller LDS footprint.
template <int T> struct KTile {
    static constexpr int BKT = (T == 9) ? 18 : (T == 25) ? 26 : 16;
    static constexpr int CPT = BKT / T;
    static constexpr int KMAX = (T == 25) ? 26 : 18;
};

typedef unsigned int u32x3 __attribute__((ext_vector_type(3)));

// PARTIAL: the last K tile may run past the last reduction channel (Cr % channels-per-tile != 0);
// SCALE: an input scale (style / demod) multiplies the im2col operand on its way into LDS.
template <int BM, int BN, int T, bool PARTIAL, bool SCALE>
__device__ __forceinline__ void modconv_body(const ConvDesc &d, const ConvClass &c,
                                             float (&As)[2][KTile<T>::KMAX][BM + 1],
                                             float (&Bs)[2][KTile<T>::KMAX + 1][BN],
                                             const int (&stab)[25], const int tile_id, const int grp,
                                             const int kt_begin, const int kt_end, const bool atomic) {
    constexpr int BKT = KTile<T>::BKT, CPT = KTile<T>::CPT, BK_MAX = KTile<T>::KMAX;
    // 4 waves: 2 x 2 over the tile, or 1 x 4 for the 32-row tile (layers of <= 32 output channels
    // would leave half of a 64-row tile's matrix work on padding); 32x32 MFMA tiles per wave
    constexpr int WAVES_M = BM >= 64 ? 2 : 1, WAVES_N = 4 / WAVES_M;
    constexpr int WMT = BM / (32 * WAVES_M), WNT = BN / (32 * WAVES_N);
    constexpr bool WIDE = (T == 9);              // 9 contiguous taps per (m, channel): 3 x dwordx3
    constexpr int EA = WIDE ? (BM * 6 + NTHREADS - 1) / NTHREADS : (BM * BKT + NTHREADS - 1) / NTHREADS;
    constexpr int EB = (BN * BKT + NTHREADS - 1) / NTHREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int Ncls = d.B * c.OH * c.OW;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int m0 = (tile_id % tiles_m) * BM;
    const int n0 = (tile_id / tiles_m) * BN;
    const int HW = d.H * d.W;
    constexpr int OOB = 0x7fffffff;  // voffset beyond num_records: the buffer load returns 0

    // Buffer resources (SGPR descriptors built from kernel arguments only): 32-bit per-lane byte
    // offsets that stay CONSTANT over the K loop + a scalar offset for the channel base; the
    // hardware range check zeroes masked elements, so the K loop carries no address arithmetic.
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)(d.x + (size_t)grp * d.Cr * HW), 0,
                                                      (d.B * d.Cx - grp * d.Cr) * HW * 4, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)(d.w + (size_t)grp * (d.w_bytes / 4)), 0, d.w_bytes, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(SCALE ? d.in_scale : d.w), 0, SCALE ? d.B * d.Cr * 4 : 4, 0x00020000);

    // ---- im2col elements of this thread: column n fixed, rows k_e = tid / BN + e * (256 / BN)
    const int nB = tid % BN;
    const int ng = n0 + nB;
    const bool n_ok = ng < Ncls;
    int bb = 0, iy0 = 0, ix0 = 0;
    if (n_ok) {
        bb = ng / (c.OH * c.OW);
        const int r = ng % (c.OH * c.OW);
        iy0 = (r / c.OW) * d.is;
        ix0 = (r % c.OW) * d.is;
    }
    int offB[EB], dchB[EB];
#pragma unroll
    for (int e = 0; e < EB; e++) {
        const int k = tid / BN + e * (NTHREADS / BN);
        const int dch = k / T, t = k - dch * T;
        const int tb = stab[t < T ? t : 0];
        const int iy = iy0 + (tb & 0xff) - 8, ix = ix0 + ((tb >> 8) & 0xff) - 8;
        const bool ok = n_ok && k < BKT && dch < CPT && iy >= 0 && iy < d.H && ix >= 0 && ix < d.W;
        offB[e] = ok ? (((bb * d.Cx + dch) * d.H + iy) * d.W + ix) * 4 : OOB;
        dchB[e] = dch;
    }
    const int offS = n_ok ? bb * d.Cr * 4 : OOB;  // + channel * 4
    // ---- weight elements.  WIDE: idx -> (m, channel-in-tile, tap triple): one dwordx3 each;
    //      else idx -> k = idx % BKT (lanes along contiguous taps), one dword each.
    int offA[EA], ldsA[EA], dchA[EA];
#pragma unroll
    for (int e = 0; e < EA; e++) {
        const int idx = tid + e * NTHREADS;
        if (WIDE) {
            const int m = idx / 6, r = idx % 6, dch = r / 3, tg = r % 3;
            const bool in = idx < BM * 6, ok = in && m0 + m < d.M;
            offA[e] = ok ? ((m0 + m) * d.w_ms + dch * d.w_ks + 3 * tg) * 4 : OOB;
            ldsA[e] = in ? (dch * 9 + 3 * tg) * (BM + 1) + m : BM;  // surplus: the pad column of rows 0..2
            dchA[e] = dch;
        } else {
            const int k = idx % BKT, m = idx / BKT;
            const int dch = k / T, t = k - dch * T;
            const bool ok = idx < BM * BKT && dch < CPT && m0 + m < d.M;
            offA[e] = ok ? ((m0 + m) * d.w_ms + dch * d.w_ks + (stab[t] >> 16)) * 4 : OOB;
            ldsA[e] = (idx < BM * BKT) ? k * (BM + 1) + m : BM;  // surplus lands in the row padding
            dchA[e] = dch;
        }
    }

    // Software pipeline, one barrier per K tile:
    //   registers hold tile t+1 (loads issued one iteration earlier), LDS buffer t&1 is being
    //   multiplied, buffer (t+1)&1 is free.  Inside the MFMA sequence of tile t, after each k-step,
    //   a slice of tile t+1 is written to the free buffer and its registers are immediately
    //   re-loaded with tile t+2 — the staging LDS / VMEM work issues in the shadow of the 64-cycle
    //   MFMAs instead of in a phase of its own.
    float ra[EA][WIDE ? 3 : 1], rb[EB];
    float rs[WIDE ? 2 : EB], rsn[2];  // WIDE: the tile's two channel scales (current / next tile)
    auto loadA = [&](int e, int kt) {
        const int ch0 = kt * CPT;
        const int vo = (PARTIAL && dchA[e] >= d.Cr - ch0) ? OOB : offA[e];
        if constexpr (WIDE) {
            const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rw, vo, ch0 * d.w_ks * 4, 0);
            ra[e][0] = __uint_as_float(v.x);
            ra[e][1] = __uint_as_float(v.y);
            ra[e][2] = __uint_as_float(v.z);
        } else {
            ra[e][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, vo, ch0 * d.w_ks * 4, 0));
        }
    };
    auto loadB = [&](int e, int kt) {
        const int ch0 = kt * CPT;
        const bool out = PARTIAL && dchB[e] >= d.Cr - ch0;
        rb[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, out ? OOB : offB[e], ch0 * HW * 4, 0));
        if constexpr (SCALE && !WIDE)
            rs[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rsc, out ? OOB : offS + dchB[e] * 4, ch0 * 4, 0));
    };
    auto loadS = [&](int kt, float (&dst)[2]) {  // WIDE only: scales of the tile's two channels
        if constexpr (SCALE && WIDE) {
            const int ch0 = kt * CPT;
            dst[0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, offS, ch0 * 4, 0));
            dst[1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rsc, (PARTIAL && 1 >= d.Cr - ch0) ? OOB : offS + 4, ch0 * 4, 0));
        }
    };
    auto storeA = [&](int e, float *abuf) {
        if constexpr (WIDE) {
            abuf[ldsA[e]] = ra[e][0];
            abuf[ldsA[e] + (BM + 1)] = ra[e][1];
            abuf[ldsA[e] + 2 * (BM + 1)] = ra[e][2];
        } else {
            abuf[ldsA[e]] = ra[e][0];
        }
    };
    auto storeB = [&](int e, float (*bbuf)[BN]) {
        const int k = tid / BN + e * (NTHREADS / BN);
        float v = rb[e];
        if constexpr (SCALE) v *= WIDE ? (dchB[e] ? rs[1] : rs[0]) : rs[e];
        bbuf[k < BKT ? k : BK_MAX][nB] = v;  // row BK_MAX = dump row
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; i++)
#pragma unroll
        for (int j = 0; j < WNT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    const int kt_last = kt_end - 1;
    // prologue: tile kt_begin -> LDS buffer 0, tile kt_begin+1 -> registers
#pragma unroll
    for (int e = 0; e < EA; e++) loadA(e, kt_begin);
#pragma unroll
    for (int e = 0; e < EB; e++) loadB(e, kt_begin);
    if constexpr (WIDE) loadS(kt_begin, rs);
#pragma unroll
    for (int e = 0; e < EA; e++) { storeA(e, &As[0][0][0]); loadA(e, min(kt_begin + 1, kt_last)); }
#pragma unroll
    for (int e = 0; e < EB; e++) { storeB(e, Bs[0]); loadB(e, min(kt_begin + 1, kt_last)); }
    if constexpr (WIDE) loadS(min(kt_begin + 1, kt_last), rs);
    __syncthreads();
    const int l31 = lane & 31, lk = lane >> 5;

    // one K tile: multiply buffer `cur`, stage tile kt+1 into `cur ^ 1`, fetch tile kt+2.
    // Operand fragments are register double-buffered (the ds_reads of k-step s+1 are issued before
    // the MFMAs of step s) and the staging slice is spread BETWEEN the MFMAs of the step, so one
    // wave alone keeps the matrix pipe issuing back to back; two co-resident waves that run in
    // lockstep (same program, same phase) then no longer leave the pipe idle together.
    auto ktile = [&](int kt, const int cur) {
        const int kt2 = min(kt + 2, kt_last);
        float *anext = &As[cur ^ 1][0][0];
        float(*bnext)[BN] = Bs[cur ^ 1];
        float a[2][WMT], b[2][WNT];
#pragma unroll
        for (int i = 0; i < WMT; i++) a[0][i] = As[cur][lk][wm * (BM / WAVES_M) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < WNT; j++) b[0][j] = Bs[cur][lk][wn * (BN / WAVES_N) + j * 32 + l31];
#pragma unroll
        for (int st = 0; st < BKT / 2; st++) {
            const int p = st & 1, k2n = 2 * st + 2;
            if (st + 1 < BKT / 2) {
#pragma unroll
                for (int i = 0; i < WMT; i++) a[p ^ 1][i] = As[cur][k2n + lk][wm * (BM / WAVES_M) + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < WNT; j++) b[p ^ 1][j] = Bs[cur][k2n + lk][wn * (BN / WAVES_N) + j * 32 + l31];
            }
            if (WIDE && st == 0) loadS(kt2, rsn);  // scales of tile kt+2, used from the next iteration on
            int slot = 0;
#pragma unroll
            for (int i = 0; i < WMT; i++)
#pragma unroll
                for (int j = 0; j < WNT; j++) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p][i], b[p][j], acc[i][j], 0, 0, 0);
                    // staging slice of this k-step, one quarter after each MFMA
                    if (slot == 0 && st < EA) storeA(st, anext);
                    if (slot == (WMT * WNT > 1 ? 1 : 0) && st < EA) loadA(st, kt2);
                    if (slot == (WMT * WNT > 2 ? 2 : 0) && st < EB) storeB(st, bnext);
                    if (slot == (WMT * WNT > 3 ? 3 : 0) && st < EB) loadB(st, kt2);
                    slot++;
                }
            // pin the interleave: next fragments first, then MFMA / staging alternately
            __builtin_amdgcn_sched_group_barrier(0x100, WMT + WNT, 0);  // DS read
            for (int q = 0; q < WMT * WNT; q++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);      // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
            }
        }
        if constexpr (WIDE && SCALE) { rs[0] = rsn[0]; rs[1] = rsn[1]; }
        // LDS hand-over only: __syncthreads() would also wait (vmcnt(0)) for the global loads of
        // tile kt+2 issued a few instructions ago and expose their latency at every K tile
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    int kt = kt_begin;
    for (; kt + 1 < kt_end; kt += 2) {  // unrolled by two: LDS buffer indices are compile-time
        ktile(kt, 0);
        ktile(kt + 1, 1);
    }
    if (kt < kt_end) ktile(kt, 0);

    // ---- epilogue: C[m][n], m = (r&3) + 8*(r>>2) + 4*(lane>>5), n = lane&31 within a 32x32 tile
#pragma unroll
    for (int j = 0; j < WNT; j++) {
        const int n = n0 + wn * (BN / WAVES_N) + j * 32 + l31;
        if (n >= Ncls) continue;
        const int b = n / (c.OH * c.OW);
        const int r_ = n % (c.OH * c.OW);
        const int oy = (r_ / c.OW) * d.os + c.oy0, ox = (r_ % c.OW) * d.os + c.ox0;
        float *yb = d.y + (((size_t)b * d.My + grp * d.M) * d.OHf + oy) * d.OWf + ox;
        const float *ob = d.out_scale ? d.out_scale + (size_t)b * d.M : nullptr;
        const float *gbias = d.bias ? d.bias + grp * d.M : nullptr;
        const float nz = d.noise ? d.noise_w[0] * d.noise[oy * d.OWf + ox] : 0.0f;
#pragma unroll
        for (int i = 0; i < WMT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * (BM / WAVES_M) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= d.M) continue;
                float v = acc[i][j][r];
                if (ob) v *= ob[m];
                if (gbias) v += gbias[m];
                v += nz;
                if (d.act) v = (v > 0.0f ? v : v * d.act_alpha) * d.act_gain;
                float *dst = yb + (size_t)m * d.OHf * d.OWf;
                if (atomic) unsafeAtomicAdd(dst, v);
                else *dst = v;
            }
    }
}

static __host__ __device__ inline int ktiles_of(int Cr, int T) {
    const int cpt = (T == 9) ? 2 : (T == 25) ? 1 : 16 / T;
    return (Cr + cpt - 1) / cpt;
}

// One (class, tile, K range) segment: compile-time tap count; no partial-tile checks when the
// channels fill the K tiles.
template <int BM, int BN, int KMAX>
__device__ __forceinline__ void modconv_segment(const ConvDesc &d, const ConvClass &c,
                                                float (&As)[2][KMAX][BM + 1], float (&Bs)[2][KMAX + 1][BN],
                                                const int (&stab)[25], const int tile_id, const int grp,
                                                const int kb, const int ke, const bool atomic) {
    const bool scale = d.in_scale != nullptr;
#define G2S_BODY(T, PART, SC) modconv_body<BM, BN, T, PART, SC>(d, c, As, Bs, stab, tile_id, grp, kb, ke, atomic)
#define G2S_TAPS(T, CPT)                                   \
    if (d.Cr % CPT) {                                      \
        if (scale) G2S_BODY(T, true, true);                \
        else G2S_BODY(T, true, false);                     \
    } else {                                               \
        if (scale) G2S_BODY(T, false, true);               \
        else G2S_BODY(T, false, false);                    \
    }
    if constexpr (KMAX == 26) {  // 5x5 (never modulated)
        G2S_BODY(25, true, false);
    } else {
        switch (c.T) {
        case 9: G2S_TAPS(9, 2) break;
        case 16: G2S_BODY(16, true, false); break;  // 4x4 (never modulated)
        case 4: G2S_TAPS(4, 4) break;
        case 2: G2S_TAPS(2, 8) break;
        default: G2S_TAPS(1, 16) break;
        }
    }
#undef G2S_TAPS
#undef G2S_BODY
}

// One workgroup of the convolution: bx = workgroup within the grid row of nx (XCD-aware tile order),
// by = split-K slice, bz = polyphase class; row = by + ny * bz.
template <int BM, int BN, int KMAX>
__device__ __forceinline__ void modconv_block(const ConvDesc &d, float (&As)[2][KMAX][BM + 1],
                                              float (&Bs)[2][KMAX + 1][BN], int (&stab)[25], const int nx,
                                              const int bx, const int by, const int bz, const int row) {
    const int tiles_m = (d.M + BM - 1) / BM;
    // grouped launch: the groups' m-tiles follow each other within a pixel tile (they read
    // neighbouring channels of the same pixels)
    // readfirstlane: the tile / group indices feed buffer descriptors, which must live in SGPRs (a
    // descriptor the compiler cannot prove uniform costs a waterfall loop around EVERY load)
    const int tile_all0 = __builtin_amdgcn_readfirstlane(xcd_logical_tile(nx, bx, row));
    const ConvClass &c = d.cls[bz];
    // uniform early exits: smaller parity classes need fewer tiles; empty split-K slices
    const int mt_all = tile_all0 % (tiles_m * d.groups);
    const int grp = __builtin_amdgcn_readfirstlane(mt_all / tiles_m);   // integer division runs on the VALU
    const int tile_id = __builtin_amdgcn_readfirstlane((tile_all0 / (tiles_m * d.groups)) * tiles_m + mt_all % tiles_m);
    if ((int)(tile_id / tiles_m) * BN >= d.B * c.OH * c.OW) return;
    const int ktiles = ktiles_of(d.Cr, c.T);
    const int slices = d.cls_splitk[bz];
    const int per = (ktiles + slices - 1) / slices;
    const int kb = by * per;
    if (by >= slices || kb >= ktiles) return;
    if (threadIdx.x < 25) stab[threadIdx.x] = c.tab[threadIdx.x];
    __syncthreads();
    modconv_segment<BM, BN, KMAX>(d, c, As, Bs, stab, tile_id, grp, kb, min(ktiles, kb + per), slices > 1);
}

template <int BM, int BN, int KMAX>
__global__ __launch_bounds__(NTHREADS) void modconv_kernel(ConvDesc d) {
    __shared__ float As[2][KMAX][BM + 1];
    __shared__ float Bs[2][KMAX + 1][BN];
    __shared__ int stab[25];
    modconv_block<BM, BN, KMAX>(d, As, Bs, stab, gridDim.x, blockIdx.x, blockIdx.y, blockIdx.z,
                                blockIdx.y + gridDim.y * blockIdx.z);
}

// The weight-gradient GEMM alone (fallback of g2s_conv2d_bwd when the data-gradient takes a 128-wide tile).
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_rider_kernel(WgradParams p) {
    __shared__ float Wa[WG_BK][WG_BM + 1];
    __shared__ float Wb[WG_BK][WG_BN + 1];
    wgrad_block(p, Wa, Wb, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Backward of one layer of the trained nets in ONE grid: workgroups [0, n_dgrad) run the
// data-gradient (the convolution kernel, 64x64 / 32x128 / 64x128 tiles), the rest the weight-gradient GEMM
// (conv_wgrad_core.h).  Both are latency-bound at these sizes (10-25 us each for a few hundred
// workgroups): dealt to the CUs together, the layer costs the longer of the two, and one graph node.
// The flattened workgroup order equals that of the two stand-alone launches, so the XCD-aware tile
// order of the convolution is unchanged.
template <int BM, int BN, int KMAX>
__global__ __launch_bounds__(NTHREADS) void conv_bwd_kernel(ConvDesc d, WgradParams p, int nx, int ny, int n_dgrad,
                                                            int w_tiles, int w_split) {
    __shared__ float As[2][KMAX][BM + 1];
    __shared__ float Bs[2][KMAX + 1][BN];
    __shared__ int stab[25];
    __shared__ float Wa[WG_BK][WG_BM + 1];
    __shared__ float Wb[WG_BK][WG_BN + 1];
    const int lin = blockIdx.x;
    if (lin < n_dgrad) {
        const int bx = lin % nx, row = lin / nx;
        modconv_block<BM, BN, KMAX>(d, As, Bs, stab, nx, bx, row % ny, row / ny, row);
    } else {
        const int l = lin - n_dgrad;
        const int bx = l % w_tiles, r = l / w_tiles;
        wgrad_block(p, Wa, Wb, bx, r % w_split, r / w_split);
    }
}


// ------------------------------------------------------------------------------------------------
// fp16-OPERAND variant (BASELINE config 5: "fp16 MFMA path"): same geometry machinery, tensors stay
// fp32 in HBM, both GEMM operands are rounded to fp16 (RTNE) on their way into LDS and multiplied
// by v_mfma_f32_32x32x8_f16 with fp32 accumulation — 16x the matrix rate of the fp32 MFMA, half the
// LDS bytes.  NOT used by the headline (fp32) workload: results differ from fp32 at the 1e-3 level.
// Deliberately simple schedule: one LDS buffer, {gather + convert + write, barrier, MFMAs, barrier}
// per K tile; latencies overlap across the 2-3 workgroups a CU holds (LDS 37 KB, < 170 registers).
// LDS images are [row][k] with k contiguous (4 halves = one MFMA operand = one ds_read_b64).
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int T> struct KTile16 {       // channels per K tile such that K = CPT * T is a multiple of 8
    static constexpr int CPT = (T == 9) ? 8 : (T == 4) ? 8 : (T == 2) ? 16 : 32;
    static constexpr int KT = CPT * T;  // 72, 32, 32, 32
};


// Epilogue of the fp16-operand bodies (as the fp32 kernel): C[m][n], m = (r&3) + 8*(r>>2) + 4*(lane>>5), n = lane&31.
template <int BM, int BN>
__device__ __forceinline__ void f16_epilogue(const ConvDesc &d, const ConvClass &c, f32x16 (&acc)[BM / 64][BN / 64],
                                             const int m0, const int n0, const int wm, const int wn, const int l31,
                                             const int lk, const int Ncls) {
    constexpr int WMT = BM / 64, WNT = BN / 64;
#pragma unroll
    for (int j = 0; j < WNT; j++) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= Ncls) continue;
        const int b = n / (c.OH * c.OW);
        const int r_ = n % (c.OH * c.OW);
        const int oy = (r_ / c.OW) * d.os + c.oy0, ox = (r_ % c.OW) * d.os + c.ox0;
        float *yb = d.y + ((size_t)b * d.M * d.OHf + oy) * d.OWf + ox;
        const float *ob = d.out_scale ? d.out_scale + (size_t)b * d.M : nullptr;
#pragma unroll
        for (int i = 0; i < WMT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= d.M) continue;
                float v = acc[i][j][r];
                if (ob) v *= ob[m];
                if (d.bias) v += d.bias[m];
                if (d.act) v = (v > 0.0f ? v : v * d.act_alpha) * d.act_gain;
                float *dst = yb + (size_t)m * d.OHf * d.OWf;
                if (d.splitk > 1) unsafeAtomicAdd(dst, v);
                else *dst = v;
            }
    }
}

template <int BM, int BN, int T, bool SCALE>
__device__ __forceinline__ void modconv_f16_body(const ConvDesc &d, const ConvClass &c, _Float16 *As, _Float16 *Bs,
                                                 const int (&stab)[25], const int tile_id) {
    constexpr int CPT = KTile16<T>::CPT, KT = KTile16<T>::KT, KTP = KT + 4;   // +4 halves: bank spread
    constexpr int WMT = BM / 64, WNT = BN / 64;
    constexpr int GA = (BM * (KT / 4) + NTHREADS - 1) / NTHREADS;   // groups of 4 consecutive k per thread
    constexpr int GB = (BN * (KT / 4) + NTHREADS - 1) / NTHREADS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lk = lane >> 5;
    const int Ncls = d.B * c.OH * c.OW;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int m0 = (tile_id % tiles_m) * BM;
    const int n0 = (tile_id / tiles_m) * BN;
    const int ktiles = (d.Cr + CPT - 1) / CPT;
    const int per = (ktiles + d.splitk - 1) / d.splitk;
    const int kt_begin = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_begin + per);
    const int HW = d.H * d.W;
    constexpr int OOB = 0x7fffffff;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)d.x, 0, d.B * d.Cr * HW * 4, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)d.w, 0, d.w_bytes, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(SCALE ? d.in_scale : d.w), 0, SCALE ? d.B * d.Cr * 4 : 4, 0x00020000);

    // ---- im2col groups: column n = gidx % BN (coalesced pixels), k = 4 * (gidx / BN) .. + 3
    int offB[GB][4], offS[GB][4];
    unsigned dchB[GB];          // 4 x 8 bits: channel-in-tile of each element (partial last tile)
#pragma unroll
    for (int e = 0; e < GB; e++) {
        const int gidx = tid + e * NTHREADS;
        const int nB = gidx % BN, kq = gidx / BN;
        const int ng = n0 + nB;
        const bool n_ok = ng < Ncls && kq < KT / 4;
        int bb = 0, iy0 = 0, ix0 = 0;
        if (n_ok) {
            bb = ng / (c.OH * c.OW);
            const int r = ng % (c.OH * c.OW);
            iy0 = (r / c.OW) * d.is;
            ix0 = (r % c.OW) * d.is;
        }
        dchB[e] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = kq * 4 + j;
            const int dch = k / T, t = k - dch * T;
            const int tb = stab[t];
            const int iy = iy0 + (tb & 0xff) - 8, ix = ix0 + ((tb >> 8) & 0xff) - 8;
            const bool ok = n_ok & (iy >= 0) & (iy < d.H) & (ix >= 0) & (ix < d.W);
            offB[e][j] = ok ? (((bb * d.Cr + dch) * d.H + iy) * d.W + ix) * 4 : OOB;
            offS[e][j] = (SCALE && n_ok) ? (bb * d.Cr + dch) * 4 : OOB;
            dchB[e] |= (unsigned)dch << (8 * j);
        }
    }
    // ---- weight groups: k fastest (contiguous taps of the forward layout): k4 = gidx % (KT/4), m = gidx / (KT/4)
    int offA[GA][4];
    unsigned dchA[GA];
#pragma unroll
    for (int e = 0; e < GA; e++) {
        const int gidx = tid + e * NTHREADS;
        const int kq = gidx % (KT / 4), m = gidx / (KT / 4);
        const bool ok = m < BM && m0 + m < d.M;
        dchA[e] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = kq * 4 + j;
            const int dch = k / T, t = k - dch * T;
            offA[e][j] = ok ? ((m0 + m) * d.w_ms + dch * d.w_ks + (stab[t] >> 16)) * 4 : OOB;
            dchA[e] |= (unsigned)dch << (8 * j);
        }
    }

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; i++)
#pragma unroll
        for (int j = 0; j < WNT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    for (int kt = kt_begin; kt < kt_end; kt++) {
        const int ch0 = kt * CPT, left = d.Cr - ch0;    // channels left (partial last tile)
        // gather + convert + write
#pragma unroll
        for (int e = 0; e < GB; e++) {
            const int gidx = tid + e * NTHREADS;
            if (gidx >= BN * (KT / 4)) break;
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool out = (int)((dchB[e] >> (8 * j)) & 0xff) >= left;
                float v = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, out ? OOB : offB[e][j], ch0 * HW * 4, 0));
                if constexpr (SCALE)
                    v *= __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, out ? OOB : offS[e][j], ch0 * 4, 0));
                h[j] = (_Float16)v;
            }
            *reinterpret_cast<f16x4 *>(&Bs[(gidx % BN) * KTP + (gidx / BN) * 4]) = h;
        }
#pragma unroll
        for (int e = 0; e < GA; e++) {
            const int gidx = tid + e * NTHREADS;
            if (gidx >= BM * (KT / 4)) break;
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool out = (int)((dchA[e] >> (8 * j)) & 0xff) >= left;
                h[j] = (_Float16)__uint_as_float(
                    __builtin_amdgcn_raw_buffer_load_b32(rw, out ? OOB : offA[e][j], ch0 * d.w_ks * 4, 0));
            }
            *reinterpret_cast<f16x4 *>(&As[(gidx / (KT / 4)) * KTP + (gidx % (KT / 4)) * 4]) = h;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KT / 8; ks++) {
            f16x4 a[WMT], b[WNT];
#pragma unroll
            for (int i = 0; i < WMT; i++)
                a[i] = *reinterpret_cast<const f16x4 *>(&As[(wm * (BM / 2) + i * 32 + l31) * KTP + ks * 8 + lk * 4]);
#pragma unroll
            for (int j = 0; j < WNT; j++)
                b[j] = *reinterpret_cast<const f16x4 *>(&Bs[(wn * (BN / 2) + j * 32 + l31) * KTP + ks * 8 + lk * 4]);
#pragma unroll
            for (int i = 0; i < WMT; i++)
#pragma unroll
                for (int j = 0; j < WNT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue (as the fp32 kernel): C[m][n], m = (r&3) + 8*(r>>2) + 4*(lane>>5), n = lane&31
#pragma unroll
    for (int j = 0; j < WNT; j++) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= Ncls) continue;
        const int b = n / (c.OH * c.OW);
        const int r_ = n % (c.OH * c.OW);
        const int oy = (r_ / c.OW) * d.os + c.oy0, ox = (r_ % c.OW) * d.os + c.ox0;
        float *yb = d.y + ((size_t)b * d.M * d.OHf + oy) * d.OWf + ox;
        const float *ob = d.out_scale ? d.out_scale + (size_t)b * d.M : nullptr;
#pragma unroll
        for (int i = 0; i < WMT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= d.M) continue;
                float v = acc[i][j][r];
                if (ob) v *= ob[m];
                if (d.bias) v += d.bias[m];
                if (d.act) v = (v > 0.0f ? v : v * d.act_alpha) * d.act_gain;
                float *dst = yb + (size_t)m * d.OHf * d.OWf;
                if (d.splitk > 1) unsafeAtomicAdd(dst, v);
                else *dst = v;
            }
    }
}


// ---- fp16 operands, the 4- / 2- / 1-tap polyphase classes and 1x1 kernels: the generic gather of
// modconv_f16_body (4 consecutive k per group, dword loads) in the pipelined schedule of the 3x3 form
// below — K tile of 32, LDS double-buffered, tile t+1 converted and written and tile t+2 requested
// behind tile t's two v_mfma_f32_32x32x16_f16 steps, one raw barrier per K tile.
template <int BM, int BN, int T, bool SCALE>
__device__ __forceinline__ void modconv_f16p_body(const ConvDesc &d, const ConvClass &c, _Float16 *smem,
                                                  const int (&stab)[25], const int tile_id) {
    static_assert(T == 4 || T == 2 || T == 1, "polyphase classes / 1x1 kernels");
    constexpr int CPT = KTile16<T>::CPT, KT = KTile16<T>::KT, KTP = KT + 8;   // 32 k per tile, pitch 40 halves
    static_assert(KT == 32, "two K = 16 steps per tile");
    constexpr int WMT = BM / 64, WNT = BN / 64;
    constexpr int GA = BM * (KT / 4) / NTHREADS, GB = BN * (KT / 4) / NTHREADS;
    static_assert(GA * NTHREADS == BM * (KT / 4) && GB * NTHREADS == BN * (KT / 4), "whole groups per thread");
    constexpr int BUF = (BM + BN) * KTP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lk = lane >> 5;
    const int Ncls = d.B * c.OH * c.OW;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int m0 = (tile_id % tiles_m) * BM;
    const int n0 = (tile_id / tiles_m) * BN;
    const int ktiles = (d.Cr + CPT - 1) / CPT;
    const int per = (ktiles + d.splitk - 1) / d.splitk;
    const int kt_begin = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_begin + per);
    const int HW = d.H * d.W;
    constexpr int OOB = 0x7fffffff;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)d.x, 0, d.B * d.Cr * HW * 4, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)d.w, 0, d.w_bytes, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(SCALE ? d.in_scale : d.w), 0, SCALE ? d.B * d.Cr * 4 : 4, 0x00020000);
    const int nB = tid % BN;
    const int ng = n0 + nB;
    const bool n_ok = ng < Ncls;
    int bb = 0, iy0 = 0, ix0 = 0;
    if (n_ok) {
        bb = ng / (c.OH * c.OW);
        const int r = ng % (c.OH * c.OW);
        iy0 = (r / c.OW) * d.is;
        ix0 = (r % c.OW) * d.is;
    }
    int offB[GB][4], offS[GB][4], offA[GA][4];
    unsigned dchB[GB], dchA[GA];    // 4 x 8 bits: channel-in-tile of each element (partial last tile)
#pragma unroll
    for (int e = 0; e < GB; e++) {
        const int kq = (tid + e * NTHREADS) / BN;
        dchB[e] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = kq * 4 + j;
            const int dch = k / T, t = k - dch * T;
            const int tb = stab[t];
            const int iy = iy0 + (tb & 0xff) - 8, ix = ix0 + ((tb >> 8) & 0xff) - 8;
            const bool ok = n_ok & (iy >= 0) & (iy < d.H) & (ix >= 0) & (ix < d.W);
            offB[e][j] = ok ? (((bb * d.Cr + dch) * d.H + iy) * d.W + ix) * 4 : OOB;
            offS[e][j] = (SCALE && n_ok) ? (bb * d.Cr + dch) * 4 : OOB;
            dchB[e] |= (unsigned)dch << (8 * j);
        }
    }
#pragma unroll
    for (int e = 0; e < GA; e++) {
        const int gidx = tid + e * NTHREADS;
        const int kq = gidx % (KT / 4), m = gidx / (KT / 4);
        const bool ok = m0 + m < d.M;
        dchA[e] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = kq * 4 + j;
            const int dch = k / T, t = k - dch * T;
            offA[e][j] = ok ? ((m0 + m) * d.w_ms + dch * d.w_ks + (stab[t] >> 16)) * 4 : OOB;
            dchA[e] |= (unsigned)dch << (8 * j);
        }
    }
    float ra[GA][4], rb[GB][4], rsv[SCALE ? GB : 1][4];
    auto load_tile = [&](int kt) {
        const int ch0 = kt * CPT, have = d.Cr - ch0;
#pragma unroll
        for (int e = 0; e < GA; e++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool out = (int)((dchA[e] >> (8 * j)) & 0xff) >= have;
                ra[e][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rw, out ? OOB : offA[e][j], ch0 * d.w_ks * 4, 0));
            }
#pragma unroll
        for (int e = 0; e < GB; e++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const bool out = (int)((dchB[e] >> (8 * j)) & 0xff) >= have;
                rb[e][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, out ? OOB : offB[e][j], ch0 * HW * 4, 0));
                if constexpr (SCALE)
                    rsv[e][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, out ? OOB : offS[e][j], ch0 * 4, 0));
            }
    };
    auto store_tile = [&](_Float16 *abuf, _Float16 *bbuf) {
#pragma unroll
        for (int e = 0; e < GA; e++) {
            const int gidx = tid + e * NTHREADS;
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; j++) h[j] = (_Float16)ra[e][j];
            *reinterpret_cast<f16x4 *>(&abuf[(gidx / (KT / 4)) * KTP + (gidx % (KT / 4)) * 4]) = h;
        }
#pragma unroll
        for (int e = 0; e < GB; e++) {
            f16x4 h;
#pragma unroll
            for (int j = 0; j < 4; j++) h[j] = (_Float16)(SCALE ? rb[e][j] * rsv[SCALE ? e : 0][j] : rb[e][j]);
            *reinterpret_cast<f16x4 *>(&bbuf[nB * KTP + ((tid + e * NTHREADS) / BN) * 4]) = h;
        }
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; i++)
#pragma unroll
        for (int j = 0; j < WNT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    load_tile(kt_begin);
    store_tile(smem, smem + BM * KTP);
    if (kt_begin + 1 < kt_end) load_tile(kt_begin + 1);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; kt++) {
        const int cur = (kt - kt_begin) & 1;
        const _Float16 *As = smem + cur * BUF, *Bs = As + BM * KTP;
#pragma unroll
        for (int ks = 0; ks < KT / 16; ks++) {
            f16x8 a[WMT], b[WNT];
#pragma unroll
            for (int i = 0; i < WMT; i++)
                a[i] = *reinterpret_cast<const f16x8 *>(&As[(wm * (BM / 2) + i * 32 + l31) * KTP + ks * 16 + lk * 8]);
#pragma unroll
            for (int j = 0; j < WNT; j++)
                b[j] = *reinterpret_cast<const f16x8 *>(&Bs[(wn * (BN / 2) + j * 32 + l31) * KTP + ks * 16 + lk * 8]);
#pragma unroll
            for (int i = 0; i < WMT; i++)
#pragma unroll
                for (int j = 0; j < WNT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt_end) {
            _Float16 *an = smem + (cur ^ 1) * BUF;
            store_tile(an, an + BM * KTP);
            if (kt + 2 < kt_end) load_tile(kt + 2);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    f16_epilogue<BM, BN>(d, c, acc, m0, n0, wm, wn, l31, lk, Ncls);
}

// ---- fp16 operands, 3x3 classes (T = 9) on maps at least 4 wide: the pipelined form.
// K is re-ordered into QUADS: (channel, kernel row) -> 4 k-slots = the row's 3 taps + one zero, so that
// every operand element group is ONE dwordx3 buffer load (weights: the 3 contiguous taps of a kernel
// row; activations: 3 horizontally adjacent pixels — at the left / right image border the load is
// shifted inwards by one pixel and the triple re-aligned in registers) and ONE ds_write_b64 of 4
// halves: 12 + 12 wide loads per thread and K tile of 4 channels instead of 72 + 72 dword loads for 8.
// The 25 % zero slots cost matrix time only, and v_mfma_f32_32x32x16_f16 (gfx950: K = 16 in 32 cycles)
// is 12x ahead of the fp32 MFMA per channel: this kernel is paced by its operand traffic (L1 / LDS),
// so the schedule is the fp32 kernel's — LDS double-buffered, tile t+1 converted and written and tile
// t+2 requested in the shadow of tile t's MFMAs, one raw barrier per K tile (no vmcnt drain).
template <int BM, int BN, bool SCALE>
__device__ __forceinline__ void modconv_f16w_body(const ConvDesc &d, const ConvClass &c, _Float16 *smem,
                                                  const int (&stab)[25], const int tile_id) {
    constexpr int CPT = 4, QT = CPT * 3, KT = QT * 4, KTP = KT + 8;   // 12 quads = 48 k-slots, pitch 56 halves
    constexpr int WMT = BM / 64, WNT = BN / 64;
    constexpr int GA = (BM * QT + NTHREADS - 1) / NTHREADS;
    constexpr int GB = (BN * QT + NTHREADS - 1) / NTHREADS;
    static_assert((BM * QT) % NTHREADS == 0 && (BN * QT) % NTHREADS == 0, "whole groups per thread");
    constexpr int BUF = (BM + BN) * KTP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lk = lane >> 5;
    const int Ncls = d.B * c.OH * c.OW;
    const int tiles_m = (d.M + BM - 1) / BM;
    const int m0 = (tile_id % tiles_m) * BM;
    const int n0 = (tile_id / tiles_m) * BN;
    const int ktiles = (d.Cr + CPT - 1) / CPT;
    const int per = (ktiles + d.splitk - 1) / d.splitk;
    const int kt_begin = blockIdx.y * per;
    const int kt_end = min(ktiles, kt_begin + per);
    const int HW = d.H * d.W;
    constexpr int OOB = 0x7fffffff;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void *)d.x, 0, d.B * d.Cr * HW * 4, 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void *)d.w, 0, d.w_bytes, 0x00020000);
    const auto rsc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(SCALE ? d.in_scale : d.w), 0, SCALE ? d.B * d.Cr * 4 : 4, 0x00020000);

    // ---- this thread's pixel column (the same for all its activation quads)
    const int nB = tid % BN;
    const int ng = n0 + nB;
    const bool n_ok = ng < Ncls;
    int bb = 0, iy0 = 0, ix0 = 0;
    if (n_ok) {
        bb = ng / (c.OH * c.OW);
        const int r = ng % (c.OH * c.OW);
        iy0 = (r / c.OW) * d.is;
        ix0 = (r % c.OW) * d.is;
    }
    // taps 3ky .. 3ky+2 of a kernel row share dy and step through x by +1 (gather) or -1 (the adjoint
    // of a stride-1 convolution): one triple starting at the smaller x, reversed in registers if needed
    const int dxa = ((stab[0] >> 8) & 0xff) - 8, dxb = ((stab[2] >> 8) & 0xff) - 8;
    const bool rev = dxa > dxb;
    const int ixmin = ix0 + (rev ? dxb : dxa);
    const int xa = min(max(ixmin, 0), d.W - 3);         // the triple is loaded from [xa, xa + 2]
    const bool left = ixmin < xa, right = ixmin > xa;  // shifted by one pixel at the image border
    int offB[GB];
    int dchB[GB];
#pragma unroll
    for (int e = 0; e < GB; e++) {
        const int q = (tid + e * NTHREADS) / BN;
        const int dch = q / 3, ky = q - dch * 3;
        const int iy = iy0 + (stab[3 * ky] & 0xff) - 8;
        const bool ok = n_ok && iy >= 0 && iy < d.H;
        offB[e] = ok ? (((bb * d.Cr + dch) * d.H + iy) * d.W + xa) * 4 : OOB;
        dchB[e] = dch;
    }
    const int offS = (SCALE && n_ok) ? bb * d.Cr * 4 : OOB;
    // ---- weight quads: lanes along the 12 quads of one output channel (144 contiguous bytes forward)
    int offA[GA], ldsA[GA], dchA[GA];
#pragma unroll
    for (int e = 0; e < GA; e++) {
        const int idx = tid + e * NTHREADS;
        const int m = idx / QT, q = idx - m * QT;
        const int dch = q / 3, ky = q - dch * 3;
        offA[e] = (m0 + m < d.M) ? ((m0 + m) * d.w_ms + dch * d.w_ks + 3 * ky) * 4 : OOB;
        ldsA[e] = m * KTP + q * 4;
        dchA[e] = dch;
    }

    u32x3 ra[GA], rb[GB];
    float rs[CPT];
    auto load_tile = [&](int kt) {
        const int ch0 = kt * CPT, have = d.Cr - ch0;     // channels present in this tile (partial last tile)
#pragma unroll
        for (int e = 0; e < GA; e++)
            ra[e] = __builtin_amdgcn_raw_buffer_load_b96(rw, dchA[e] >= have ? OOB : offA[e], ch0 * d.w_ks * 4, 0);
#pragma unroll
        for (int e = 0; e < GB; e++)
            rb[e] = __builtin_amdgcn_raw_buffer_load_b96(rx, dchB[e] >= have ? OOB : offB[e], ch0 * HW * 4, 0);
        if constexpr (SCALE) {
#pragma unroll
            for (int cc = 0; cc < CPT; cc++)
                rs[cc] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsc, cc >= have ? OOB : offS + 4 * cc, ch0 * 4, 0));
        }
    };
    auto store_tile = [&](_Float16 *abuf, _Float16 *bbuf) {
#pragma unroll
        for (int e = 0; e < GA; e++) {
            f16x4 h;
            h[0] = (_Float16)__uint_as_float(ra[e].x);
            h[1] = (_Float16)__uint_as_float(ra[e].y);
            h[2] = (_Float16)__uint_as_float(ra[e].z);
            h[3] = (_Float16)0.0f;
            *reinterpret_cast<f16x4 *>(&abuf[ldsA[e]]) = h;
        }
#pragma unroll
        for (int e = 0; e < GB; e++) {
            const float l0 = __uint_as_float(rb[e].x), l1 = __uint_as_float(rb[e].y), l2 = __uint_as_float(rb[e].z);
            float e0 = left ? 0.0f : (right ? l1 : l0);
            float e1 = left ? l0 : (right ? l2 : l1);
            float e2 = left ? l1 : (right ? 0.0f : l2);
            if constexpr (SCALE) {
                const float sc = dchB[e] == 0 ? rs[0] : dchB[e] == 1 ? rs[1] : dchB[e] == 2 ? rs[2] : rs[3];
                e0 *= sc;
                e1 *= sc;
                e2 *= sc;
            }
            f16x4 h;
            h[0] = (_Float16)(rev ? e2 : e0);
            h[1] = (_Float16)e1;
            h[2] = (_Float16)(rev ? e0 : e2);
            h[3] = (_Float16)0.0f;
            const int q = (tid + e * NTHREADS) / BN;
            *reinterpret_cast<f16x4 *>(&bbuf[nB * KTP + q * 4]) = h;
        }
    };

    f32x16 acc[WMT][WNT];
#pragma unroll
    for (int i = 0; i < WMT; i++)
#pragma unroll
        for (int j = 0; j < WNT; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.0f;

    load_tile(kt_begin);
    store_tile(smem, smem + BM * KTP);
    if (kt_begin + 1 < kt_end) load_tile(kt_begin + 1);
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; kt++) {
        const int cur = (kt - kt_begin) & 1;
        const _Float16 *As = smem + cur * BUF, *Bs = As + BM * KTP;
#pragma unroll
        for (int ks = 0; ks < KT / 16; ks++) {
            f16x8 a[WMT], b[WNT];
#pragma unroll
            for (int i = 0; i < WMT; i++)
                a[i] = *reinterpret_cast<const f16x8 *>(&As[(wm * (BM / 2) + i * 32 + l31) * KTP + ks * 16 + lk * 8]);
#pragma unroll
            for (int j = 0; j < WNT; j++)
                b[j] = *reinterpret_cast<const f16x8 *>(&Bs[(wn * (BN / 2) + j * 32 + l31) * KTP + ks * 16 + lk * 8]);
#pragma unroll
            for (int i = 0; i < WMT; i++)
#pragma unroll
                for (int j = 0; j < WNT; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt_end) {
            _Float16 *an = smem + (cur ^ 1) * BUF;
            store_tile(an, an + BM * KTP);                 // tile kt+1 (requested one iteration ago)
            if (kt + 2 < kt_end) load_tile(kt + 2);
        }
        // LDS hand-over only (no vmcnt drain: the loads of tile kt+2 stay in flight over the barrier)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }

    // ---- epilogue (as the fp32 kernel): C[m][n], m = (r&3) + 8*(r>>2) + 4*(lane>>5), n = lane&31
#pragma unroll
    for (int j = 0; j < WNT; j++) {
        const int n = n0 + wn * (BN / 2) + j * 32 + l31;
        if (n >= Ncls) continue;
        const int b = n / (c.OH * c.OW);
        const int r_ = n % (c.OH * c.OW);
        const int oy = (r_ / c.OW) * d.os + c.oy0, ox = (r_ % c.OW) * d.os + c.ox0;
        float *yb = d.y + ((size_t)b * d.M * d.OHf + oy) * d.OWf + ox;
        const float *ob = d.out_scale ? d.out_scale + (size_t)b * d.M : nullptr;
#pragma unroll
        for (int i = 0; i < WMT; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
                if (m >= d.M) continue;
                float v = acc[i][j][r];
                if (ob) v *= ob[m];
                if (d.bias) v += d.bias[m];
                if (d.act) v = (v > 0.0f ? v : v * d.act_alpha) * d.act_gain;
                float *dst = yb + (size_t)m * d.OHf * d.OWf;
                if (d.splitk > 1) unsafeAtomicAdd(dst, v);
                else *dst = v;
            }
    }
}

template <int BM, int BN>
__global__ __launch_bounds__(NTHREADS) void modconv_f16_kernel(ConvDesc d) {
    constexpr int KTP = 72 + 4;
    // one buffer for both forms: 2 x (BM + BN) rows of 56 halves (pipelined 3x3 form) >= (BM + BN) rows of 76
    __shared__ __attribute__((aligned(16))) _Float16 smem[2 * (BM + BN) * 56];
    _Float16 *As = smem, *Bs = smem + BM * KTP;
    __shared__ int stab[25];
    const ConvClass &c = d.cls[blockIdx.z];
    const int tiles_m = (d.M + BM - 1) / BM;
    const int tile_id = xcd_logical_tile();
    if ((int)(tile_id / tiles_m) * BN >= d.B * c.OH * c.OW) return;
    const bool wide = c.T == 9 && d.W >= 4;   // the pipelined quad form (4 channels per K tile)
    const int cpt = wide ? 4 : c.T == 9 ? 8 : c.T == 4 ? 8 : c.T == 2 ? 16 : 32;
    const int ktiles = (d.Cr + cpt - 1) / cpt;
    const int per = (ktiles + d.splitk - 1) / d.splitk;
    if ((int)blockIdx.y * per >= ktiles) return;
    if (threadIdx.x < 25) stab[threadIdx.x] = c.tab[threadIdx.x];
    __syncthreads();
    const bool scale = d.in_scale != nullptr;
    if (wide) {
        if (scale) modconv_f16w_body<BM, BN, true>(d, c, smem, stab, tile_id);
        else modconv_f16w_body<BM, BN, false>(d, c, smem, stab, tile_id);
        return;
    }
    switch (c.T) {
    case 9:
        if (scale) modconv_f16_body<BM, BN, 9, true>(d, c, As, Bs, stab, tile_id);
        else modconv_f16_body<BM, BN, 9, false>(d, c, As, Bs, stab, tile_id);
        break;
    case 4:
        if (scale) modconv_f16p_body<BM, BN, 4, true>(d, c, smem, stab, tile_id);
        else modconv_f16p_body<BM, BN, 4, false>(d, c, smem, stab, tile_id);
        break;
    case 2:
        if (scale) modconv_f16p_body<BM, BN, 2, true>(d, c, smem, stab, tile_id);
        else modconv_f16p_body<BM, BN, 2, false>(d, c, smem, stab, tile_id);
        break;
    default:
        if (scale) modconv_f16p_body<BM, BN, 1, true>(d, c, smem, stab, tile_id);
        else modconv_f16p_body<BM, BN, 1, false>(d, c, smem, stab, tile_id);
        break;
    }
}

static int pack(int dy, int dx, int wt) { return (dy + 8) | ((dx + 8) << 8) | (wt << 16); }

}  // namespace g2s

using namespace g2s;

// Measured (tile, split-K) choices for the call signatures of the north-star workload.
struct TunedConv { int B, Cin, Cout, H, k, mode, transpose, fused, tile, splitk; };
static const TunedConv kTuned[] = {
#include "modconv_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

// ... and of g2s_conv2d (the trained nets' layers): keyed by the call signature of g2s_conv2d.
struct TunedConv2d { int B, Cr, M, H, k, stride, pad, adjoint, m_major, fused, groups, tile, splitk; };
static const TunedConv2d kTuned2d[] = {
#include "conv2d_tuned.inc"
    {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}};

// g2s_modconv_tune: per-thread override of the tile / split-K heuristic (tools/tune_modconv.py).
static thread_local int g_force_tile = -1, g_force_splitk = -1;

extern "C" int g2s_modconv_tune(int tile, int splitk) {
    G2S_REQUIRE(tile >= -2 && tile <= 4 && splitk >= -1 && splitk <= 64 && splitk != 0,
                "tile must be -1 (built-in), -2 (heuristic without the tuned table) or 0..4, "
                "splitk -1 (built-in) or 1..64");
    g_force_tile = tile;
    g_force_splitk = splitk;
    return G2S_OK;
}

// General geometry of one launch.
//   adjoint = 0 (gather):  y[b,m,oy,ox]            = sum_{c,ky,kx} x[b,c,oy*s+ky-p,ox*s+kx-p] W(m,c,ky,kx)
//   adjoint = 1 (scatter): y[b,m,iy*s+ky-p,ix*s+kx-p] += x[b,c,iy,ix] W(m,c,ky,kx)   (out_h x out_w given)
//   W(m,c,ky,kx) = w[m*w_ms + c*w_ks + ky*k + kx]:  m_major: w is [M, Cr, k, k], else [Cr, M, k, k].
struct ConvGeom {
    int k, stride, pad, adjoint, m_major;
    int out_h, out_w;  // adjoint only; 0 = (H-1)*s - 2p + k
};

// The weight-gradient GEMM that rides in the same grid as a data-gradient launch (g2s_conv2d_bwd).
struct WgradRider {
    WgradParams p;
    int tiles, split;
    bool dw_is_zero;
};

// Noise term of the epilogue for the NEXT launch of this thread (set and cleared by g2s_modconv_nba around its call:
// an implementation detail of this file, no state survives an entry point).
static thread_local const float *t_noise = nullptr, *t_noise_w = nullptr;

static int conv_launch(const float *x, const float *w, const float *in_scale, const float *out_scale,
                       const float *bias, int act, float act_alpha, float act_gain, float *y, int B,
                       int Cr, int M, int H, int W, const ConvGeom &g, int tuned_tile, int tuned_splitk,
                       g2s_stream_t stream, bool y_is_zero = false, bool f16_operands = false, int groups = 1,
                       const WgradRider *rider = nullptr, int *plan_needs_zero = nullptr) {
    G2S_REQUIRE(plan_needs_zero || (x && w && y), "x, w, y must not be NULL");
    G2S_REQUIRE(B > 0 && Cr > 0 && M > 0 && H > 0 && W > 0, "sizes must be positive");
    const int k = g.k, s_ = g.stride, p_ = g.pad, KK = k * k;
    G2S_REQUIRE(k >= 1 && k <= 5, "kernel size must be 1..5 (got %d)", k);
    G2S_REQUIRE(s_ == 1 || s_ == 2, "stride must be 1 or 2 (got %d)", s_);
    G2S_REQUIRE(p_ >= 0 && p_ < k, "padding must be in [0, k)");
    ConvDesc d{};
    d.x = x;
    d.w = w;
    d.in_scale = in_scale;
    d.out_scale = out_scale;
    d.bias = bias;
    d.noise = t_noise;
    d.noise_w = t_noise_w;
    G2S_REQUIRE(!d.noise || (!f16_operands && groups == 1), "the noise epilogue exists for the fp32 kernel only");
    d.act = act;
    d.act_alpha = act_alpha;
    d.act_gain = act_gain;
    d.y = y;
    d.B = B;
    d.H = H;
    d.W = W;
    d.Cr = Cr;
    d.M = M;
    d.w_ms = g.m_major ? Cr * KK : KK;
    d.w_ks = g.m_major ? KK : M * KK;
    d.w_bytes = M * Cr * KK * 4;
    G2S_REQUIRE(groups >= 1 && groups <= 8 && (groups == 1 || (!in_scale && !out_scale && !f16_operands)),
                "grouped launches take no input / output scales and fp32 operands");
    d.groups = groups;
    d.Cx = groups * Cr;
    d.My = groups * M;
    bool holes = false;  // output positions no class writes (zero-filled)
    if (!g.adjoint) {
        G2S_REQUIRE(H + 2 * p_ >= k && W + 2 * p_ >= k, "input smaller than the kernel");
        d.ncls = 1;
        d.os = 1;
        d.is = s_;
        d.OHf = (H + 2 * p_ - k) / s_ + 1;
        d.OWf = (W + 2 * p_ - k) / s_ + 1;
        ConvClass &c = d.cls[0];
        c.T = KK;
        c.oy0 = c.ox0 = 0;
        c.OH = d.OHf;
        c.OW = d.OWf;
        for (int ky = 0; ky < k; ky++)
            for (int kx = 0; kx < k; kx++) c.tab[ky * k + kx] = pack(ky - p_, kx - p_, ky * k + kx);
    } else {
        d.OHf = g.out_h ? g.out_h : (H - 1) * s_ - 2 * p_ + k;
        d.OWf = g.out_w ? g.out_w : (W - 1) * s_ - 2 * p_ + k;
        G2S_REQUIRE(d.OHf >= (H - 1) * s_ - 2 * p_ + k && d.OWf >= (W - 1) * s_ - 2 * p_ + k &&
                        d.OHf <= (H - 1) * s_ - 2 * p_ + k + s_ - 1 && d.OWf <= (W - 1) * s_ - 2 * p_ + k + s_ - 1,
                    "adjoint output size %dx%d inconsistent with the input", d.OHf, d.OWf);
        d.is = 1;
        d.os = s_;
        d.ncls = 0;
        // output row oy = iy*s + ky - p: residue class r = oy mod s uses the taps ky == r + p (mod s),
        // read at iy = oy' + (r + p - ky) / s  (oy = oy'*s + r).  No multiplications by inserted zeros.
        for (int ry = 0; ry < s_; ry++)
            for (int rx = 0; rx < s_; rx++) {
                ConvClass c{};
                c.oy0 = ry;
                c.ox0 = rx;
                c.OH = (d.OHf - ry + s_ - 1) / s_;
                c.OW = (d.OWf - rx + s_ - 1) / s_;
                c.T = 0;
                for (int ky = 0; ky < k; ky++)
                    for (int kx = 0; kx < k; kx++)
                        if ((ry + p_ - ky) % s_ == 0 && (rx + p_ - kx) % s_ == 0)
                            c.tab[c.T++] = pack((ry + p_ - ky) / s_, (rx + p_ - kx) / s_, ky * k + kx);
                if (c.T == 0 || c.OH <= 0 || c.OW <= 0) {
                    holes = holes || (c.OH > 0 && c.OW > 0);
                    continue;
                }
                d.cls[d.ncls++] = c;
            }
        G2S_REQUIRE(d.ncls > 0, "empty convolution");
    }
    bool big = false;
    for (int i = 0; i < d.ncls; i++) {
        const int T = d.cls[i].T;
        G2S_REQUIRE(T == 1 || T == 2 || T == 4 || T == 9 || T == 16 || T == 25,
                    "unsupported tap count %d (k=%d stride=%d)", T, k, s_);
        G2S_REQUIRE(!(in_scale && T > 9), "input scales are limited to kernels of up to 3x3");
        big = big || T == 25;
    }
    // tile configuration + split-K: aim for >= 2 workgroups (8 waves) per CU
    long nmax = 0;
    int kt_min = 1 << 30;
    for (int i = 0; i < d.ncls; i++) {
        nmax = std::max(nmax, (long)B * d.cls[i].OH * d.cls[i].OW);
        kt_min = std::min(kt_min, ktiles_of(d.Cr, d.cls[i].T));
    }
    G2S_REQUIRE(nmax < (1l << 30) && (long)B * d.Cx * H * W < (1l << 29) && (long)groups * M * Cr * KK < (1l << 29),
                "problem too large for 32-bit byte offsets");
    hipStream_t st = as_stream(stream);
    const int cfgs[5][2] = {{128, 128}, {128, 64}, {64, 64}, {32, 128}, {64, 128}};
    int pick = 2;
    for (int i = 0; i < 3; i++) {
        const long blocks = (long)groups * cdiv(d.M, cfgs[i][0]) * cdiv(nmax, cfgs[i][1]) * d.ncls;
        if (d.M > cfgs[i][0] / 2 && blocks >= 512) { pick = i; break; }
    }
    if (d.M <= 32 && nmax >= 1024 && !f16_operands) pick = 3;   // a 64-row tile would be half padding
    if (f16_operands && pick == 1) pick = 2;   // fp16 form: 128x128 or 64x64 only; fewer than 512 big tiles -> small ones
    if (tuned_tile >= 0 && g_force_tile != -2) pick = tuned_tile;
    else tuned_splitk = -1;
    if (g_force_tile >= 0) pick = g_force_tile;
    const int BMv = cfgs[pick][0], BNv = cfgs[pick][1];
    const int tiles = groups * cdiv(d.M, BMv) * cdiv(nmax, BNv);
    int splitk = 1;
    while ((long)tiles * d.ncls * splitk < 512 && kt_min / (splitk * 2) >= 8 && splitk < 64) splitk *= 2;
    if (tuned_splitk > 0) splitk = tuned_splitk;
    if (g_force_splitk > 0) splitk = g_force_splitk;
    if (deterministic()) splitk = 1;   // no float atomics: one workgroup owns every output element
    splitk = std::max(1, std::min(splitk, kt_min));
    dim3 grid(tiles, splitk, d.ncls);
    // A bias / activation epilogue needs the complete sum: with split-K it runs as a second,
    // elementwise launch (g2s_fused_bias_act in place) after the partial sums have been added.
    const bool split = splitk > 1;
    const bool deferred_epilogue = split && (bias != nullptr || act != 0 || d.noise != nullptr);
    const float *noise = d.noise, *noise_w = d.noise_w;
    if (deferred_epilogue) {
        d.bias = nullptr;
        d.noise = d.noise_w = nullptr;
        d.act = 0;
    }
    d.splitk = splitk;
    // the polyphase classes of a strided scatter differ in depth (4 / 2 / 2 / 1 taps of a 3x3):
    // slices in proportion, so that every workgroup carries about the same run of K tiles
    int kt_max = 1;
    for (int i = 0; i < d.ncls; i++) kt_max = std::max(kt_max, ktiles_of(d.Cr, d.cls[i].T));
    for (int i = 0; i < d.ncls; i++) {
        const int kt = ktiles_of(d.Cr, d.cls[i].T);
        d.cls_splitk[i] = std::max(1, std::min(kt, (int)(((long)splitk * kt + kt_max / 2) / kt_max)));
    }
    if (plan_needs_zero) {   // g2s_modconv_needs_zero: report, launch nothing
        *plan_needs_zero = (split || holes) ? 1 : 0;
        return G2S_OK;
    }
    if ((split || holes) && !y_is_zero) {
        if (hipMemsetAsync(y, 0, (size_t)B * d.My * d.OHf * d.OWf * sizeof(float), st) != hipSuccess)
            return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(y) failed");
    }
    if (f16_operands) {
        for (int i = 0; i < d.ncls; i++)
            G2S_REQUIRE(d.cls[i].T == 1 || d.cls[i].T == 2 || d.cls[i].T == 4 || d.cls[i].T == 9,
                        "fp16 operands: 1x1 / 3x3 kernels (stride 1 or 2) only");
        G2S_REQUIRE(pick <= 2, "fp16 operands: tiles 0..2 only");
        if (pick == 2) modconv_f16_kernel<64, 64><<<grid, NTHREADS, 0, st>>>(d);
        else modconv_f16_kernel<128, 128><<<grid, NTHREADS, 0, st>>>(d);
    } else if (rider) {
        // data-gradient + weight-gradient of one layer in one grid (64x64 / 32x128 tiles: what the
        // trained nets' layers use); 128-row tiles: two launches
        const WgradParams &p = rider->p;
        if (p.atomic && !rider->dw_is_zero &&
            hipMemsetAsync(p.dw, 0, (size_t)p.groups * p.Ca * p.N * sizeof(float), st) != hipSuccess)
            return fail(G2S_ERR_LAUNCH, "hipMemsetAsync(dw) failed");
        if (pick >= 2) {
            const int n_dgrad = tiles * splitk * d.ncls, n_wgrad = rider->tiles * rider->split * p.groups;
#define G2S_BWD(BM_, BN_, K_) \
    conv_bwd_kernel<BM_, BN_, K_><<<n_dgrad + n_wgrad, NTHREADS, 0, st>>>(d, p, tiles, splitk, n_dgrad, rider->tiles, rider->split)
            if (pick == 2) { if (big) G2S_BWD(64, 64, 26); else G2S_BWD(64, 64, 18); }
            else if (pick == 3) { if (big) G2S_BWD(32, 128, 26); else G2S_BWD(32, 128, 18); }
            else { if (big) G2S_BWD(64, 128, 26); else G2S_BWD(64, 128, 18); }
#undef G2S_BWD
        } else {
            if (big) {
                if (pick == 0) modconv_kernel<128, 128, 26><<<grid, NTHREADS, 0, st>>>(d);
                else modconv_kernel<128, 64, 26><<<grid, NTHREADS, 0, st>>>(d);
            } else {
                if (pick == 0) modconv_kernel<128, 128, 18><<<grid, NTHREADS, 0, st>>>(d);
                else modconv_kernel<128, 64, 18><<<grid, NTHREADS, 0, st>>>(d);
            }
            conv_wgrad_rider_kernel<<<dim3(rider->tiles, rider->split, p.groups), NTHREADS, 0, st>>>(p);
        }
    } else if (big) {
        if (pick == 0) modconv_kernel<128, 128, 26><<<grid, NTHREADS, 0, st>>>(d);
        else if (pick == 1) modconv_kernel<128, 64, 26><<<grid, NTHREADS, 0, st>>>(d);
        else if (pick == 2) modconv_kernel<64, 64, 26><<<grid, NTHREADS, 0, st>>>(d);
        else if (pick == 3) modconv_kernel<32, 128, 26><<<grid, NTHREADS, 0, st>>>(d);
        else modconv_kernel<64, 128, 26><<<grid, NTHREADS, 0, st>>>(d);
    } else {
        if (pick == 0) modconv_kernel<128, 128, 18><<<grid, NTHREADS, 0, st>>>(d);
        else if (pick == 1) modconv_kernel<128, 64, 18><<<grid, NTHREADS, 0, st>>>(d);
        else if (pick == 2) modconv_kernel<64, 64, 18><<<grid, NTHREADS, 0, st>>>(d);
        else if (pick == 3) modconv_kernel<32, 128, 18><<<grid, NTHREADS, 0, st>>>(d);
        else modconv_kernel<64, 128, 18><<<grid, NTHREADS, 0, st>>>(d);
    }
    int rc = check_launch("g2s_modconv");
    if (rc != G2S_OK || !deferred_epilogue) return rc;
    if (noise) {    // StyledConv tail in place (bias is required there, act = leaky-ReLU)
        G2S_REQUIRE(bias && act == 1, "noise epilogue: needs a bias and act = 1");
        return g2s_noise_bias_act(y, noise, noise_w, bias, y, B, d.My, d.OHf * d.OWf, act_alpha, act_gain, stream);
    }
    return g2s_fused_bias_act(y, bias, nullptr, y, (int64_t)B * d.My * d.OHf * d.OWf,
                              (int64_t)d.OHf * d.OWf, d.My, act ? 3 : 1, 0, act_alpha,
                              act ? act_gain : 1.0f, G2S_F32, stream);
}

// The StyleGAN2 modes of g2s_modconv in terms of the general geometry (w is always [Cout, Cin, k, k]).
static int modconv_launch(const float *x, const float *w, const float *in_scale,
                          const float *out_scale, const float *bias, int act, float act_alpha,
                          float act_gain, float *y, int B, int Cin, int Cout, int H, int W, int k,
                          int mode, int transpose, g2s_stream_t stream, bool f16_operands = false,
                          bool y_is_zero = false, int *plan_needs_zero = nullptr) {
    G2S_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "sizes must be positive");
    G2S_REQUIRE(k == 1 || k == 3, "kernel size must be 1 or 3 (got %d)", k);
    G2S_REQUIRE(mode == G2S_CONV_PLAIN || mode == G2S_CONV_UP2 || mode == G2S_CONV_DOWN2,
                "unsupported mode %d", mode);
    // fromRGB (1x1 from 3 channels over >= 64 K pixels) runs as a streaming kernel (thinconv.hip) instead
    // of an MFMA tile whose K is padding.  A forced tile keeps the call on the MFMA kernel (tests).
    if (!f16_operands && !in_scale && !out_scale && mode == G2S_CONV_PLAIN && g_force_tile == -1 &&
        g_force_splitk == -1 && thin_conv_eligible(B, Cin, Cout, H, W, k, transpose)) {
        if (plan_needs_zero) {
            *plan_needs_zero = 0;
            return G2S_OK;
        }
        return thin_conv_launch(x, w, bias, y, B, Cin, Cout, H, W, act, act_alpha, act_gain, stream);
    }
    ConvGeom g{};
    g.k = k;
    g.stride = mode == G2S_CONV_PLAIN ? 1 : 2;
    g.pad = mode == G2S_CONV_PLAIN ? k / 2 : 0;
    // UP2 is the scatter; a transposed call runs the adjoint geometry with the roles of Cin / Cout
    // (and therefore the two weight strides) swapped
    g.adjoint = (mode == G2S_CONV_UP2) != (transpose != 0);
    g.m_major = !transpose;
    if (!g.adjoint && mode != G2S_CONV_PLAIN) G2S_REQUIRE(H >= k && W >= k, "input smaller than the kernel");
    int tile = -1, splitk = -1;
    if (H == W && !f16_operands)
        for (int pass = 0; pass < 2 && tile == -1 && splitk == -1; pass++)   // pass 1: the row measured without / with the epilogue
            for (const TunedConv *t = kTuned; t->B; ++t)
                if (t->B == B && t->Cin == Cin && t->Cout == Cout && t->H == H && t->k == k && t->mode == mode &&
                    t->transpose == transpose && (pass == 1 || t->fused == (bias != nullptr || act != 0))) {
                    tile = t->tile;
                    splitk = t->splitk;
                    break;
                }
    return conv_launch(x, w, in_scale, out_scale, bias, act, act_alpha, act_gain, y, B,
                       transpose ? Cout : Cin, transpose ? Cin : Cout, H, W, g, tile, splitk, stream, y_is_zero,
                       f16_operands, 1, nullptr, plan_needs_zero);
}

// g2s_modconv + g2s_conv_bias_act in one entry point, with the caller's promise that y is already
// zero (a slice of a pool cleared once per training step: the split-K / polyphase-hole paths then
// issue no clear of their own — one graph node less per such launch).
extern "C" int g2s_modconv_ex(const float *x, const float *w, const float *in_scale, const float *out_scale,
                              const float *bias, float *y, int B, int Cin, int Cout, int H, int W, int k, int mode,
                              int transpose, int act, float alpha, float gain, int y_is_zero, g2s_stream_t stream) {
    G2S_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-ReLU)");
    return modconv_launch(x, w, in_scale, out_scale, bias, act, alpha, gain, y, B, Cin, Cout, H, W, k, mode,
                          transpose, stream, false, y_is_zero != 0);
}

// g2s_modconv_ex with the whole StyledConv tail (stylegan2-pytorch/model.py:349-355) in the epilogue:
// y = gain * leaky_relu(out_scale * conv(in_scale * x) + noise_w[0] * noise[h, w] + bias[c], alpha).
extern "C" int g2s_modconv_nba(const float *x, const float *w, const float *in_scale, const float *out_scale,
                               const float *bias, const float *noise, const float *noise_w, float *y, int B, int Cin,
                               int Cout, int H, int W, int k, int mode, int transpose, float alpha, float gain,
                               int y_is_zero, g2s_stream_t stream) {
    G2S_REQUIRE(bias && noise && noise_w, "bias, noise and noise_w must not be NULL");
    t_noise = noise;
    t_noise_w = noise_w;
    const int rc = modconv_launch(x, w, in_scale, out_scale, bias, 1, alpha, gain, y, B, Cin, Cout, H, W, k, mode,
                                  transpose, stream, false, y_is_zero != 0);
    t_noise = t_noise_w = nullptr;
    return rc;
}

// 1 if that launch adds into a cleared output (split-K slices / polyphase holes), else 0; < 0: error.
extern "C" int g2s_modconv_needs_zero(int B, int Cin, int Cout, int H, int W, int k, int mode, int transpose,
                                      int has_scales, int fused) {
    int needs = 0;
    static const float dummy = 0.0f;   // never dereferenced: the plan mode launches nothing
    const float *sc = has_scales ? &dummy : nullptr;
    const int rc = modconv_launch(nullptr, nullptr, sc, nullptr, fused ? &dummy : nullptr, fused ? 1 : 0, 0.0f, 1.0f,
                                  nullptr, B, Cin, Cout, H, W, k, mode, transpose, nullptr, false, false, &needs);
    return rc == G2S_OK ? needs : rc;
}

extern "C" int g2s_modconv_f16(const float *x, const float *w, const float *in_scale, const float *out_scale,
                               const float *bias, float *y, int B, int Cin, int Cout, int H, int W, int k,
                               int mode, int transpose, int act, float alpha, float gain, g2s_stream_t stream) {
    G2S_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-ReLU)");
    return modconv_launch(x, w, in_scale, out_scale, bias, act, alpha, gain, y, B, Cin, Cout, H, W, k, mode,
                          transpose, stream, true);
}

extern "C" int g2s_modconv(const float *x, const float *w, const float *in_scale,
                           const float *out_scale, float *y, int B, int Cin, int Cout, int H, int W,
                           int k, int mode, int transpose, g2s_stream_t stream) {
    return modconv_launch(x, w, in_scale, out_scale, nullptr, 0, 0.0f, 1.0f, y, B, Cin, Cout, H, W, k,
                          mode, transpose, stream);
}

extern "C" int g2s_conv_bias_act(const float *x, const float *w, const float *bias, float *y, int B,
                                 int Cin, int Cout, int H, int W, int k, int mode, int act,
                                 float alpha, float gain, g2s_stream_t stream) {
    G2S_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-ReLU)");
    return modconv_launch(x, w, nullptr, nullptr, bias, act, alpha, gain, y, B, Cin, Cout, H, W, k,
                          mode, 0, stream);
}

static int conv2d_impl(const float *x, const float *w, const float *bias, float *y, int B, int Cr, int M, int H,
                       int W, int k, int stride, int pad, int adjoint, int w_m_major, int out_h, int out_w,
                       int act, float alpha, float gain, int y_is_zero, int groups, g2s_stream_t stream,
                       const WgradRider *rider = nullptr) {
    G2S_REQUIRE(act == 0 || act == 1, "act must be 0 (none) or 1 (leaky-ReLU)");
    ConvGeom g{k, stride, pad, adjoint ? 1 : 0, w_m_major ? 1 : 0, adjoint ? out_h : 0, adjoint ? out_w : 0};
    int tile = -1, splitk = -1;
    if (H == W)
        for (const TunedConv2d *t = kTuned2d; t->B; ++t)
            if (t->B == B && t->Cr == Cr && t->M == M && t->H == H && t->k == k && t->stride == stride &&
                t->pad == pad && t->adjoint == g.adjoint && t->m_major == g.m_major &&
                t->fused == (bias != nullptr || act != 0) && t->groups == groups) {
                tile = t->tile;
                splitk = t->splitk;
                break;
            }
    return conv_launch(x, w, nullptr, nullptr, bias, act, alpha, gain, y, B, Cr, M, H, W, g, tile, splitk, stream,
                       y_is_zero != 0, false, groups, rider);
}

extern "C" int g2s_conv2d(const float *x, const float *w, const float *bias, float *y, int B, int Cr,
                          int M, int H, int W, int k, int stride, int pad, int adjoint, int w_m_major,
                          int out_h, int out_w, int act, float alpha, float gain, int y_is_zero,
                          g2s_stream_t stream) {
    return conv2d_impl(x, w, bias, y, B, Cr, M, H, W, k, stride, pad, adjoint, w_m_major, out_h, out_w, act, alpha,
                       gain, y_is_zero, 1, stream);
}

extern "C" int g2s_conv2d_grouped(const float *x, const float *w, const float *bias, float *y, int B, int Cr,
                                  int M, int H, int W, int k, int stride, int pad, int adjoint, int w_m_major,
                                  int out_h, int out_w, int act, float alpha, float gain, int y_is_zero,
                                  int groups, g2s_stream_t stream) {
    return conv2d_impl(x, w, bias, y, B, Cr, M, H, W, k, stride, pad, adjoint, w_m_major, out_h, out_w, act, alpha,
                       gain, y_is_zero, groups, stream);
}

extern "C" int g2s_conv2d_bwd(const float *gy, const float *w, float *gx, int B, int Cr, int M, int H, int W, int k,
                              int stride, int pad, int adjoint, int w_m_major, int out_h, int out_w, int gx_is_zero,
                              const float *A, const float *G, float *dw, int Ca, int Cg, int PH, int PW, int GH,
                              int GW, int dw_is_zero, int groups, g2s_stream_t stream) {
    WgradRider rider;
    const int rc = wgrad_plan(A, G, dw, B, Ca, Cg, PH, PW, GH, GW, k, stride, pad, groups, rider.p, rider.tiles,
                              rider.split);
    if (rc != G2S_OK) return rc;
    rider.dw_is_zero = dw_is_zero != 0;
    return conv2d_impl(gy, w, nullptr, gx, B, Cr, M, H, W, k, stride, pad, adjoint, w_m_major, out_h, out_w, 0, 0.0f,
                       1.0f, gx_is_zero, groups, stream, &rider);
}
