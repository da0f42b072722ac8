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
    int splits;             // blocks that share one output tile, each with a slice of the pixels
    float alpha;
    // The pixel slices of a tile are combined WITHOUT atomics, so that the sum has one fixed order:
    // with splits > 1 slice s writes its partial tile (plain stores) to ws + s * out_elems at the offset the
    // tile has in dw, and fold_partials adds the planes up in slice order.  splits == 1: the tile goes
    // straight to dw (added to what dw holds when `accumulate`).
    float* ws;
    long out_elems;   // floats of the whole output (all samples, taps, tiles)
    int accumulate;
    Tap taps[kMaxTaps];
};

// How a launch cuts the pixels: the workspace it needs follows from it.
struct WgPlan {
    int splits, pix_per_block;
};

// LDS-DMA variant with producer / consumer waves (conv_wgrad_glds.hip).  wgrad_glds_plan returns false when
// the problem does not fit it (channel counts that are not multiples of 64 bytes on both operands, unaligned
// tensors); launch_wgrad_glds runs the plan that the caller stored in p.splits / p.pix_per_block.
bool wgrad_glds_plan(const WgradParams& p, int dtype, WgPlan* plan);
int launch_wgrad_glds(const WgradParams& p, int dtype, hipStream_t s);

// Resident-patch variant for 3x3 stride-1 layers (conv_wgrad_res.hip): a block owns a 32 x 32 tile under all
// nine taps and a slice of whole image rows.  wgrad_res_plan returns false when the problem does not fit it.
bool wgrad_res_plan(const WgradParams& p, int dtype, WgPlan* plan);
int launch_wgrad_res(const WgradParams& p, int dtype, hipStream_t s);

}  // namespace stmi
