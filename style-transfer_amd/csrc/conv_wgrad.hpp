// Launch parameters of the pixel-reduction GEMM kernels (weight gradients, Gram matrices).
#pragma once
#include "common.hpp"

namespace stmi {

struct WgradParams {
    const void* x;
    const void* dy;
    float* dw;
    int N, Hi, Wi, Ci;  // x
    int Ho, Wo, Co;     // dy
    int stride;
    int ntaps, klen, cdiv;
    long tap_out_stride;
    long batch_out_stride;  // != 0: one output per sample (Gram)
    int pix_per_block;
    int splits;
    float alpha;
    Tap taps[kMaxTaps];
};

// LDS-DMA variant with producer / consumer waves (conv_wgrad_glds.hip).  Returns true and
// launches when the problem fits it (channel counts that are multiples of 128 bytes on both
// operands); *rc receives the status.  The output must already hold the values to add to.
bool try_launch_wgrad_glds(const WgradParams& p, int dtype, hipStream_t s, int* rc);

}  // namespace stmi
