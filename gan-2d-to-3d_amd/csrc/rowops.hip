// rowops.hip — row-wise fused reductions that remove the elementwise/reduce launch chains around the
// modulated convolution (HBM-bound: every operand is read exactly once).
//
//   g2s_rows_dot_scale   gradients of  y = demod * conv(W, s * x)  that are reductions over H*W:
//                          dot[r]  = (sum_i a[r,i] * b[r,i]) * (inv ? 1 / inv[r] : 1)
//                          out[r,i] = b[r,i] * s[r]                       (optional, may alias b)
//                        used as  gs = sum_hw x * gxs ; gx = gxs * s     and   gd = sum_hw gy * y / demod
//   g2s_demod_fwd/_bwd   demod[b,o] = rsqrt(sum_i wsq[o,i] * s[b,i]^2 + eps)
//                        (ModulatedConv2d.forward, stylegan2-pytorch/model.py:254-258, with
//                        weight = scale * W * style  =>  sum_{i,t} weight^2 = sum_i wsq[o,i] s[b,i]^2)
#include "g2s_common.h"

namespace g2s {

__device__ __forceinline__ float wave_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// one wavefront per row, 4 rows per workgroup.  Rows of any length: a scalar head brings the row
// to 16-byte alignment (all operands share the row offset and have 16-byte aligned bases), then
// float4 body, scalar tail.
__global__ __launch_bounds__(256) void rows_dot_scale(const float *__restrict__ a,
                                                      const float *__restrict__ b,
                                                      const float *__restrict__ s,
                                                      const float *__restrict__ inv,
                                                      float *__restrict__ out,
                                                      float *__restrict__ dot, int rows, int n,
                                                      int aligned_bases) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t off = (size_t)row * n;
    const float *ar = a ? a + off : nullptr;
    const float *br = b + off;
    float *orow = out ? out + off : nullptr;
    const float sc = s ? s[row] : 1.0f;
    float acc = 0.0f;
    int head = aligned_bases ? (int)((4 - (off & 3)) & 3) : n;  // unaligned bases: all scalar
    if (head > n) head = n;
    const int n4 = (n - head) >> 2, tail0 = head + 4 * n4;
    for (int i = lane; i < head; i += 64) {
        const float bv = br[i];
        if (ar) acc += ar[i] * bv;
        if (orow) orow[i] = bv * sc;
    }
    for (int i = lane; i < n4; i += 64) {
        const float4 bv = reinterpret_cast<const float4 *>(br + head)[i];
        if (ar) {
            const float4 av = reinterpret_cast<const float4 *>(ar + head)[i];
            acc += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
        }
        if (orow)
            reinterpret_cast<float4 *>(orow + head)[i] = make_float4(bv.x * sc, bv.y * sc, bv.z * sc, bv.w * sc);
    }
    for (int i = tail0 + lane; i < n; i += 64) {
        const float bv = br[i];
        if (ar) acc += ar[i] * bv;
        if (orow) orow[i] = bv * sc;
    }
    if (dot) {
        acc = wave_sum(acc);
        if (lane == 0) dot[row] = inv ? acc / inv[row] : acc;
    }
}

// one wavefront per (b, o)
__global__ __launch_bounds__(256) void demod_fwd(const float *__restrict__ wsq,
                                                 const float *__restrict__ s,
                                                 float *__restrict__ demod, int B, int Cin, int Cout,
                                                 float eps) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= B * Cout) return;
    const int b = idx / Cout, o = idx % Cout;
    const float *w = wsq + (size_t)o * Cin, *sb = s + (size_t)b * Cin;
    float acc = 0.0f;
    for (int i = lane; i < Cin; i += 64) acc += w[i] * sb[i] * sb[i];
    acc = wave_sum(acc);
    if (lane == 0) demod[idx] = 1.0f / sqrtf(acc + eps);
}

// gs[b,i] = -s[b,i] * sum_o gd[b,o] * demod[b,o]^3 * wsq[o,i]
// grid (ceil(Cin / 64), B); 4 waves split the o range, lanes run along i (coalesced wsq rows).
__global__ __launch_bounds__(256) void demod_bwd(const float *__restrict__ wsq,
                                                 const float *__restrict__ s,
                                                 const float *__restrict__ demod,
                                                 const float *__restrict__ gd,
                                                 const float *gs_add,   // may alias gs (in-place add)
                                                 float *gs, int B, int Cin, int Cout) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lane, b = blockIdx.y;
    float acc = 0.0f;
    if (i < Cin) {
#pragma unroll 8
        for (int o = wave; o < Cout; o += 4) {
            const float d = demod[b * Cout + o];
            acc += gd[b * Cout + o] * d * d * d * wsq[(size_t)o * Cin + i];
        }
    }
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < Cin) {
        const float g = -s[b * Cin + i] * (red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
        gs[b * Cin + i] = gs_add ? gs_add[b * Cin + i] + g : g;
    }
}

}  // namespace g2s

using namespace g2s;

extern "C" int g2s_rows_dot_scale(const float *a, const float *b, const float *s, const float *inv,
                                  float *out, float *dot, int rows, int n, g2s_stream_t stream) {
    G2S_REQUIRE(b != nullptr && rows > 0 && n > 0, "b must not be NULL; rows, n positive");
    G2S_REQUIRE(out || dot, "nothing to compute");
    G2S_REQUIRE(!dot || a, "dot needs a");
    const int aligned = !((uintptr_t)b & 15) && (!a || !((uintptr_t)a & 15)) && (!out || !((uintptr_t)out & 15));
    hipStream_t st = as_stream(stream);
    rows_dot_scale<<<cdiv(rows, 4), 256, 0, st>>>(a, b, s, inv, out, dot, rows, n, aligned);
    return check_launch("g2s_rows_dot_scale");
}

extern "C" int g2s_demod_fwd(const float *wsq, const float *s, float *demod, int B, int Cin, int Cout,
                             float eps, g2s_stream_t stream) {
    G2S_REQUIRE(wsq && s && demod && B > 0 && Cin > 0 && Cout > 0, "bad argument");
    demod_fwd<<<cdiv((long)B * Cout, 4), 256, 0, as_stream(stream)>>>(wsq, s, demod, B, Cin, Cout, eps);
    return check_launch("g2s_demod_fwd");
}

extern "C" int g2s_demod_bwd(const float *wsq, const float *s, const float *demod, const float *gd,
                             float *gs, int B, int Cin, int Cout, g2s_stream_t stream) {
    G2S_REQUIRE(wsq && s && demod && gd && gs && B > 0 && Cin > 0 && Cout > 0, "bad argument");
    demod_bwd<<<dim3(cdiv(Cin, 64), B), 256, 0, as_stream(stream)>>>(wsq, s, demod, gd, nullptr, gs, B, Cin, Cout);
    return check_launch("g2s_demod_bwd");
}

extern "C" int g2s_demod_bwd_add(const float *wsq, const float *s, const float *demod, const float *gd,
                                 const float *gs_add, float *gs, int B, int Cin, int Cout, g2s_stream_t stream) {
    G2S_REQUIRE(wsq && s && demod && gd && gs && B > 0 && Cin > 0 && Cout > 0, "bad argument");
    demod_bwd<<<dim3(cdiv(Cin, 64), B), 256, 0, as_stream(stream)>>>(wsq, s, demod, gd, gs_add, gs, B, Cin, Cout);
    return check_launch("g2s_demod_bwd_add");
}
