// probe.hip — measurement aid, not part of the training path: what the fp32 matrix pipe of this GPU sustains.
// g2s_mfma_probe launches `blocks` workgroups of `waves` wavefronts; every wave issues `iters` x 8 independent
// v_mfma_f32_32x32x2_f32 (8 accumulators, no memory traffic, operands in registers) — the speed of light the
// convolution kernels are priced against is the NOMINAL 157.3 TFLOP/s (256 CUs x 4 SIMDs x 256 FLOP/cycle x 2.4 GHz);
// this kernel shows how much of it a wave (or two per SIMD) can actually draw (tools/bench_mfma_peak.py).
#include "g2s_common.h"

namespace g2s {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void mfma_probe_kernel(float *out, int iters) {
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.0f;
    float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < 8; p++) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[p], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[p][r];
    if (s == 12345.678f) out[0] = s;   // keeps the accumulators alive
}

// The same with the operand traffic of the Winograd kernel's inner loop: per 16 MFMAs a wave reads 8 x 16 bytes per lane
// from LDS into the MFMA operand registers (mode 1: accumulators where the compiler puts them — ArchVGPRs at this register
// count; mode 2: accumulators forced into AccVGPRs through inline assembly).
typedef float f32x4_ __attribute__((ext_vector_type(4)));
template <bool ACC_AGPR>
__global__ __launch_bounds__(512) void mfma_lds_probe_kernel(float *out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = 1.0f + i * 1e-7f;
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[p][r] = 0.0f;
    const f32x4_ *src = reinterpret_cast<const f32x4_ *>(lds) + (threadIdx.x & 63) + (threadIdx.x >> 6) * 64;
    for (int it = 0; it < iters; it++) {
        f32x4_ f[8];
#pragma unroll
        for (int e = 0; e < 8; e++) f[e] = src[(e * 512 + it * 8) & 4095 & ~63];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const float a = f[(q >> 2) & 3][q & 3], b = f[4 + ((q >> 2) & 3)][q & 3];
            if constexpr (ACC_AGPR) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[q & 7]) : "v"(a), "v"(b));
            } else {
                acc[q & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[q & 7], 0, 0, 0);
            }
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[p][r];
    if (s == 12345.678f) out[0] = s;
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_mfma_lds_probe(float *out, int blocks, int waves, int iters, int acc_agpr, g2s_stream_t stream) {
    G2S_REQUIRE(out && blocks > 0 && waves >= 1 && waves <= 8 && iters > 0, "bad argument");
    if (acc_agpr) mfma_lds_probe_kernel<true><<<blocks, 64 * waves, 0, as_stream(stream)>>>(out, iters);
    else mfma_lds_probe_kernel<false><<<blocks, 64 * waves, 0, as_stream(stream)>>>(out, iters);
    return check_launch("g2s_mfma_lds_probe");
}

extern "C" int g2s_mfma_probe(float *out, int blocks, int waves, int iters, g2s_stream_t stream) {
    G2S_REQUIRE(out && blocks > 0 && waves >= 1 && waves <= 8 && iters > 0, "bad argument");
    mfma_probe_kernel<<<blocks, 64 * waves, 0, as_stream(stream)>>>(out, iters);
    return check_launch("g2s_mfma_probe");
}
