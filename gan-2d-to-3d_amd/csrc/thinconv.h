// thinconv.h — 1x1 convolution from <= 4 input channels over many pixels (thinconv.hip).
#pragma once
#include <cstdint>
#include "g2s_common.h"

namespace g2s {

// Cr reduction channels -> M output channels, no input / output scales.
bool thin_conv_eligible(int B, int Cr, int M, int H, int W, int k, int transpose);

// y = act(conv1x1(x, w) + bias); w is [M, Cr, 1, 1].
int thin_conv_launch(const float *x, const float *w, const float *bias, float *y, int B, int Cr, int M, int H, int W,
                     int act, float alpha, float gain, g2s_stream_t stream);

}  // namespace g2s
