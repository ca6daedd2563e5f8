// conv_wgrad_core.h — the weight-gradient GEMM of the small trained nets (see conv_wgrad.hip) as a
// device function + host-side planning, shared by the stand-alone kernel (conv_wgrad.hip) and the
// fused backward launch of modconv.hip (data-gradient and weight-gradient of one layer in ONE grid).
#pragma once
#include <algorithm>
#include "g2s_common.h"

namespace g2s {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_BM = 64, WG_BN = 64, WG_BK = 32, WG_THREADS = 256;
constexpr int WG_E = WG_BM * WG_BK / WG_THREADS;  // elements per thread and operand per K tile (8)

struct WgradParams {
    const float *A, *G;
    float *dw;
    int B, Ca, Cg, PH, PW, GH, GW, k, stride, pad;
    int N;       // Cg * k * k
    int K;       // B * PH * PW
    int ktiles;  // ceil(K / 32)
    int per;     // K tiles per pixel slice (by)
    int atomic;  // more than one slice
    int groups;  // independent problems (bz): A has groups * Ca channels per sample, G has
                 // groups * Cg, dw holds the groups back to back (op/conv.py PairConvFunction)
};

// One workgroup of the weight-gradient GEMM: (bx, by, bz) = (output tile, pixel slice, group).
__device__ __forceinline__ void wgrad_block(const WgradParams &p, float (&As)[WG_BK][WG_BM + 1],
                                            float (&Bs)[WG_BK][WG_BN + 1], const int bx, const int by,
                                            const int bz) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, lk = lane >> 5;
    const int tiles_m = (p.Ca + WG_BM - 1) / WG_BM;
    const int m0 = (bx % tiles_m) * WG_BM, n0 = (bx / tiles_m) * WG_BN;
    const int kt0 = by * p.per, kt1 = min(p.ktiles, kt0 + p.per);
    const int P = p.PH * p.PW, KK = p.k * p.k;
    const int grp = bz;
    const int Ca_tot = p.Ca * p.groups, Cg_tot = p.Cg * p.groups;

    // this thread's elements: pixel lane kl fixed, rows (a or n) r0 + 8 e
    const int kl = tid & 31, r0 = tid >> 5;
    int gbase[WG_E], dy[WG_E], dx[WG_E];
    bool nok[WG_E], aok[WG_E];
#pragma unroll
    for (int e = 0; e < WG_E; e++) {
        const int n = n0 + r0 + 8 * e;
        nok[e] = n < p.N;
        const int g = n / KK, t = n - g * KK;
        gbase[e] = g * p.GH * p.GW;
        dy[e] = t / p.k - p.pad;
        dx[e] = t % p.k - p.pad;
        aok[e] = m0 + r0 + 8 * e < p.Ca;
    }
    float ra[WG_E], rb[WG_E];
    auto load = [&](int kt) {
        const int kg = kt * WG_BK + kl;
        const bool kok = kg < p.K;
        const int b = kok ? kg / P : 0, pix = kok ? kg - b * P : 0;
        const int py = pix / p.PW, px = pix - py * p.PW;
        const float *ap = p.A + ((size_t)b * Ca_tot + grp * p.Ca + m0 + r0) * P + pix;
        const float *gp = p.G + ((size_t)b * Cg_tot + grp * p.Cg) * p.GH * p.GW;
        const int gy0 = py * p.stride, gx0 = px * p.stride;
#pragma unroll
        for (int e = 0; e < WG_E; e++) {
            ra[e] = (kok && aok[e]) ? ap[(size_t)8 * e * P] : 0.0f;
            const int gy = gy0 + dy[e], gx = gx0 + dx[e];
            const bool ok = kok && nok[e] && gy >= 0 && gy < p.GH && gx >= 0 && gx < p.GW;
            rb[e] = ok ? gp[gbase[e] + gy * p.GW + gx] : 0.0f;
        }
    };
    wg_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] = 0.0f;
    if (kt0 < kt1) load(kt0);
    for (int kt = kt0; kt < kt1; kt++) {
        __syncthreads();  // previous tile's fragments have been read
#pragma unroll
        for (int e = 0; e < WG_E; e++) {
            As[kl][r0 + 8 * e] = ra[e];
            Bs[kl][r0 + 8 * e] = rb[e];
        }
        __syncthreads();
        if (kt + 1 < kt1) load(kt + 1);  // in flight during the MFMAs
#pragma unroll
        for (int k2 = 0; k2 < WG_BK; k2 += 2) {
            const float a = As[k2 + lk][wm * 32 + l31];
            const float b = Bs[k2 + lk][wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // C[m][n]: m = (r&3) + 8*(r>>2) + 4*(lane>>5), n = lane&31 within the wave's 32x32 tile
    const int n = n0 + wn * 32 + l31;
    if (n >= p.N) return;
#pragma unroll
    for (int r = 0; r < 16; r++) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= p.Ca) continue;
        float *dst = p.dw + ((size_t)grp * p.Ca + m) * p.N + n;
        if (p.atomic) unsafeAtomicAdd(dst, acc[r]);
        else *dst = acc[r];
    }
}

}  // namespace g2s

// Validates the arguments and fills the launch parameters; tiles x split x groups workgroups.
static inline int wgrad_plan(const float *A, const float *G, float *dw, int B, int Ca, int Cg, int PH, int PW, int GH,
                             int GW, int k, int stride, int pad, int groups, g2s::WgradParams &p, int &tiles,
                             int &split) {
    using namespace g2s;
    G2S_REQUIRE(A && G && dw, "NULL pointer argument");
    G2S_REQUIRE(B > 0 && Ca > 0 && Cg > 0 && PH > 0 && PW > 0 && GH > 0 && GW > 0, "sizes must be positive");
    G2S_REQUIRE(k >= 1 && k <= 5 && (stride == 1 || stride == 2) && pad >= 0 && pad < k,
                "k must be 1..5, stride 1 or 2, 0 <= pad < k");
    G2S_REQUIRE(groups >= 1 && groups <= 8, "groups must be 1..8");
    G2S_REQUIRE((long)B * groups * Cg * GH * GW < (1l << 31) && (long)B * groups * Ca * PH * PW < (1l << 31) &&
                    (long)groups * Ca * Cg * k * k < (1l << 31), "tensor too large");
    p = WgradParams{};
    p.A = A;
    p.G = G;
    p.dw = dw;
    p.B = B;
    p.Ca = Ca;
    p.Cg = Cg;
    p.PH = PH;
    p.PW = PW;
    p.GH = GH;
    p.GW = GW;
    p.k = k;
    p.stride = stride;
    p.pad = pad;
    p.groups = groups;
    p.N = Cg * k * k;
    p.K = B * PH * PW;
    p.ktiles = cdiv(p.K, WG_BK);
    tiles = cdiv(Ca, WG_BM) * cdiv(p.N, WG_BN);
    split = std::max(1, std::min(p.ktiles, cdiv(768, tiles * groups)));
    if (deterministic()) split = 1;   // the whole pixel reduction in one workgroup: fixed order, no atomics
    p.per = cdiv(p.ktiles, split);
    split = cdiv(p.ktiles, p.per);
    p.atomic = split > 1;
    return G2S_OK;
}
