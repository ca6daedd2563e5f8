// adam.hip — one-launch Adam step over a list of parameter tensors (g2s_adam_step, include/g2s.h).
//
// The reference trains its five small nets with torch.optim.Adam(lr 1e-4, betas (0.9, 0.999),
// weight_decay 5e-4: classic L2, GAN2Shape/trainer.py:163-171), one optimiser per step kind over
// up to 22.6 M parameters in ~90 tensors.  The update is a pure streaming pass (read p, g, m, v;
// write p, m, v: 28 bytes per parameter); here it is ONE grid for all tensors: the table of the
// persistent pointers (parameter, moments, counters) and the chunk prefix live in device memory,
// uploaded once; the gradient pointers — the only ones that change from step to step — travel as
// kernel arguments (so nothing is copied per step, and a captured graph carries them in its node);
// a block finds its tensor by bisection and updates one chunk of 16-byte lanes.  Step counters are device
// scalars, one per tensor as in torch (a tensor without a gradient skips the step and its count), so
// a captured HIP graph replays the right bias correction: every block of a tensor reads the count,
// the block of that tensor that finishes last increments it.
//
//   g' = g + wd p;  m = b1 m + (1 - b1) g';  v = b2 v + (1 - b2) g'^2;  t = step + 1
//   p -= lr / (1 - b1^t) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          (torch's Adam, amsgrad off)
#include "g2s_common.h"

namespace g2s {

typedef float ad_v4 __attribute__((ext_vector_type(4)));

constexpr int AD_THREADS = 256;
constexpr int AD_CHUNK = 256 * 4 * 16;   // elements per block: 16 float4 per thread

struct AdamGrads { const float *g[G2S_ADAM_MAX_TENSORS]; };

__global__ __launch_bounds__(AD_THREADS) void adam_kernel(const g2s_adam_tensor *tensors, const int *chunk0,
                                                          AdamGrads grads, int n_tensors, float lr, float b1,
                                                          float b2, float eps, float wd) {
    // tensor of this chunk: chunk0[i] <= blockIdx.x < chunk0[i + 1]
    int lo = 0, hi = n_tensors;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (chunk0[mid] <= (int)blockIdx.x) lo = mid;
        else hi = mid;
    }
    const g2s_adam_tensor t = tensors[lo];
    const float *tg = grads.g[lo];
    const int64_t base = (int64_t)(blockIdx.x - chunk0[lo]) * AD_CHUNK;
    const float tstep = *t.step + 1.0f;
    const float bc1 = 1.0f - powf(b1, tstep), bc2 = 1.0f - powf(b2, tstep);
    const float step_size = lr / bc1, inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
    auto update = [&](float &p, float g, float &m, float &v) {
        g = g + wd * p;
        m = b1 * m + (1.0f - b1) * g;
        v = b2 * v + (1.0f - b2) * g * g;
        p = p - step_size * (m / (sqrtf(v) * inv_sqrt_bc2 + eps));
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(t.p) | reinterpret_cast<uintptr_t>(tg) |
                       reinterpret_cast<uintptr_t>(t.m) | reinterpret_cast<uintptr_t>(t.v)) & 15) == 0;
    // 4 lanes-of-16-bytes per trip: all 16 loads of a trip are issued before its first store (p, m, v
    // may alias as far as the compiler knows, so a one-element loop would serialise on its stores);
    // gradients are read once and the results are not read again this step: non-temporal accesses
    // (177 vs 196 us on the step-3 set; a block-cyclic layout and larger trips changed nothing — the
    // in-place read-modify-write of three arrays stays at ~3.5 TB/s, as torch's fused kernel does)
    for (int it = 0; it < 16; it += 4) {
        int64_t idx[4];
        float4 p[4], g[4], m[4], v[4];
        bool full[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            idx[u] = base + ((int64_t)(it + u) * AD_THREADS + threadIdx.x) * 4;
            full[u] = vec && idx[u] + 4 <= t.n;
            if (full[u]) {
                p[u] = *reinterpret_cast<const float4 *>(t.p + idx[u]);
                const ad_v4 gg = __builtin_nontemporal_load(reinterpret_cast<const ad_v4 *>(tg + idx[u]));
                g[u] = float4{gg.x, gg.y, gg.z, gg.w};
                m[u] = *reinterpret_cast<const float4 *>(t.m + idx[u]);
                v[u] = *reinterpret_cast<const float4 *>(t.v + idx[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (full[u]) {
                update(p[u].x, g[u].x, m[u].x, v[u].x);
                update(p[u].y, g[u].y, m[u].y, v[u].y);
                update(p[u].z, g[u].z, m[u].z, v[u].z);
                update(p[u].w, g[u].w, m[u].w, v[u].w);
                __builtin_nontemporal_store(ad_v4{p[u].x, p[u].y, p[u].z, p[u].w}, reinterpret_cast<ad_v4 *>(t.p + idx[u]));
                __builtin_nontemporal_store(ad_v4{m[u].x, m[u].y, m[u].z, m[u].w}, reinterpret_cast<ad_v4 *>(t.m + idx[u]));
                __builtin_nontemporal_store(ad_v4{v[u].x, v[u].y, v[u].z, v[u].w}, reinterpret_cast<ad_v4 *>(t.v + idx[u]));
            } else {
                for (int64_t j = idx[u]; j < idx[u] + 4 && j < t.n; j++) update(t.p[j], tg[j], t.m[j], t.v[j]);
            }
        }
        if (idx[3] + 4 >= t.n) break;
    }
    // every block of this tensor has read its count before it draws a ticket: the last one advances it
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(t.ticket, 1) == chunk0[lo + 1] - chunk0[lo] - 1) {
            *t.step = tstep;
            *t.ticket = 0;
        }
    }
}

}  // namespace g2s

using namespace g2s;

extern "C" int64_t g2s_adam_chunk(void) { return AD_CHUNK; }

extern "C" int g2s_adam_step(const g2s_adam_tensor *tensors, const int *chunk0, const float *const *grads,
                             int n_tensors, int n_chunks, float lr, float beta1, float beta2, float eps,
                             float weight_decay, g2s_stream_t stream) {
    G2S_REQUIRE(tensors && chunk0 && grads, "NULL pointer argument");
    G2S_REQUIRE(n_tensors > 0 && n_tensors <= G2S_ADAM_MAX_TENSORS && n_chunks >= n_tensors,
                "1 <= n_tensors <= %d, a chunk per tensor at least", G2S_ADAM_MAX_TENSORS);
    AdamGrads ag{};
    for (int i = 0; i < n_tensors; i++) {
        G2S_REQUIRE(grads[i], "NULL gradient pointer");
        ag.g[i] = grads[i];
    }
    G2S_REQUIRE(lr >= 0 && beta1 >= 0 && beta1 < 1 && beta2 >= 0 && beta2 < 1 && eps >= 0 && weight_decay >= 0,
                "invalid hyper-parameters");
    adam_kernel<<<n_chunks, AD_THREADS, 0, as_stream(stream)>>>(tensors, chunk0, ag, n_tensors, lr, beta1, beta2,
                                                                 eps, weight_decay);
    return check_launch("g2s_adam_step");
}
