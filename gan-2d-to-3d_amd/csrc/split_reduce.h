// split_reduce.h — second half of a split-K Winograd launch with a workspace (split_reduce.hip): the K
// slices of a tile STORE their partial sums to consecutive copies of the output in the caller's
// workspace; this pass adds the copies, applies bias + activation and writes y.
//
// Why only there: the Winograd kernel holds one workgroup per CU and all of them reach their epilogue
// together, so the float atomics of a split launch arrive as one burst with nothing left to overlap
// it — measured 8.5 us per million atomics (8 x 512 x 16 x 16 outputs in 4 slices: 35 of the launch's
// 72 us).  Plain stores + this pass cost a quarter of that and make the sums deterministic.  The
// direct kernel keeps 2-4 workgroups per CU, whose atomics hide behind the others' matrix work: there
// the same scheme measured 1-8 us SLOWER per launch, so it keeps its atomics.
#pragma once
#include "g2s_common.h"

namespace g2s {

constexpr int SPLIT_REDUCE_MAX = 8;   // most slices one pass adds (all copies are requested before the first add)

// y[i] = act(sum_{s < slices} part[s * n + i] + bias[channel(i)]),  channel(i) = (i / hw) % channels;
// bias may be NULL; act 0: none, 1: leaky-ReLU(alpha) * gain.
// noise / noise_w (optional): + noise_w[0] * noise[i % hw] with the bias (StyledConv's NoiseInjection).
int split_reduce_launch(const float *part, int slices, int64_t n, float *y, const float *bias, int64_t hw,
                        int channels, int act, float alpha, float gain, g2s_stream_t stream,
                        const float *noise = nullptr, const float *noise_w = nullptr);

}  // namespace g2s
