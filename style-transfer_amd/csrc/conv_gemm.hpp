// Shared pieces of the gather-GEMM convolution kernels: launch parameters and the epilogue.
#pragma once
#include "common.hpp"

namespace stmi {

struct ConvGemmParams {
    const void* in;
    const void* wp;
    void* out;
    void* out_pre;
    const float* bias;
    const void* mask;
    const void* addend;
    int N, Hi, Wi, Ci;  // gathered tensor
    int Ho, Wo, Co;     // written tensor
    int Hs, Ws, oy0, ox0, oys, oxs;  // output sub-grid
    int in_stride;      // input base position = sub-grid index * in_stride
    int ntaps, klen, kpad, cdiv, co_pad;
    long w_batch_stride;  // != 0: one weight set per sample (Gram backward)
    float alpha;
    int relu;
    Tap taps[kMaxTaps];
};

// Epilogue shared by the kernels.  Accumulator tile (tc, tp) of a wave holds, in lane
// (g = lane>>4, r16 = lane&15), channels co..co+3 (co = co0 + (wc*TC+tc)*16 + 4g) of pixel
// column pix0 + (wpx*TP+tp)*16 + r16:   v = (acc + bias) * alpha ; masked ; + addend ;
// optional pre-ReLU copy ; optional ReLU ; store.
template <typename T, int TC, int TP>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, const f32x4 (&acc)[TC][TP], int co0, long pix0,
                                              int wc, int wpx, int g, int r16, long Mz, long zoff, int HWs) {
    T* out = (T*)p.out;
    T* out_pre = (T*)p.out_pre;
    const T* mask = (const T*)p.mask;
    const T* addend = (const T*)p.addend;
    const bool vec_ok = (p.Co & 3) == 0;
#pragma unroll
    for (int tp = 0; tp < TP; tp++) {
        const long m = pix0 + (wpx * TP + tp) * 16 + r16;
        if (m >= Mz) continue;
        const long mm = m + zoff;
        const int n = (int)(mm / HWs);
        const int rem = (int)(mm - (long)n * HWs);
        const int i = rem / p.Ws, j = rem - i * p.Ws;
        const long off = (((long)n * p.Ho + (p.oy0 + p.oys * i)) * p.Wo + (p.ox0 + p.oxs * j)) * p.Co;
#pragma unroll
        for (int tc = 0; tc < TC; tc++) {
            const int co = co0 + (wc * TC + tc) * 16 + 4 * g;
            if (co >= p.Co) continue;
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = acc[tc][tp][u];
            if (p.bias) {
#pragma unroll
                for (int u = 0; u < 4; u++)
                    if (co + u < p.Co) v[u] += p.bias[co + u];
            }
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] *= p.alpha;
            if (vec_ok) {
                if (mask) {
                    float mk[4];
                    load_vec<T, 4>(mask + off + co, mk);
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = mk[u] > 0.f ? v[u] : 0.f;
                }
                if (addend) {
                    float ad[4];
                    load_vec<T, 4>(addend + off + co, ad);
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] += ad[u];
                }
                if (out_pre) store_vec<T, 4>(out_pre + off + co, v);
                if (p.relu) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = fmaxf(v[u], 0.f);
                }
                store_vec<T, 4>(out + off + co, v);
            } else {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (co + u >= p.Co) continue;
                    float x = v[u];
                    if (mask) x = to_f32<T>(mask[off + co + u]) > 0.f ? x : 0.f;
                    if (addend) x += to_f32<T>(addend[off + co + u]);
                    if (out_pre) out_pre[off + co + u] = from_f32<T>(x);
                    if (p.relu) x = fmaxf(x, 0.f);
                    out[off + co + u] = from_f32<T>(x);
                }
            }
        }
    }
}

// LDS-DMA variant (conv_gemm_glds.hip).  Returns true and launches when the problem fits
// it (K per tap a multiple of 128 bytes, per-pixel validity); *rc receives the status.
bool try_launch_conv_gemm_glds(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc);
const char* conv_gemm_glds_cfg_name(const ConvGemmParams& p, int dtype);

}  // namespace stmi
