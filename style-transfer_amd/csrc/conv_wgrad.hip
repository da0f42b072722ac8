// Pixel-reduction GEMM for gfx950: convolution weight gradients and Gram matrices.
//
//   D[tap][e][co] = alpha * sum_pixels X[pixel @ tap][e] * DY[pixel][co]
//
// Both operands are stored channels-last, i.e. with the reduction index (the
// pixel) as the slow axis, so MFMA fragments need a transposed read.  Tiles are
// staged in LDS as [pixel][channel]; bf16 fragments come from
// ds_read_b64_tr_b16 (hardware 4x16 transpose), f32 fragments from column
// reads in the native 16x16x4 operand layout.  The K mapping inside a step is
// chosen for conflict-free banking (lane group g takes pixels 4g..4g+3 and
// 16+4g..16+4g+3); it only has to agree between the two operands.
// The pixel range is split over blocks.  Blocks that share an output tile write their partial tiles
// to planes of a workspace (plain stores) and fold_partials_kernel sums the planes in slice order:
// no float atomics anywhere, so a result does not depend on the order in which blocks finish.
#include "conv_wgrad.hpp"

#include <stdlib.h>

namespace stmi {

template <typename T> struct WgFrag;

template <> struct WgFrag<bf16_t> {
    static constexpr int BKP = 32;  // pixels per step
    // tile: [BKP][ld] bf16, ld*2 bytes per row, ld % 16 == 0
    static __device__ __forceinline__ bf16x8 load(const bf16_t* tile, int ld, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const bf16_t* a0 = tile + (4 * g + q) * ld + col0 + 4 * pp;
        const bf16_t* a1 = a0 + 16 * ld;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
        union {
            s16x4 h[2];
            bf16x8 f;
        } u;
        u.h[0] = lo;
        u.h[1] = hi;
        return u.f;
    }
};

template <> struct WgFrag<float> {
    static constexpr int BKP = 16;
    static __device__ __forceinline__ f32x4 load(const float* tile, int ld, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        f32x4 f;
#pragma unroll
        for (int s = 0; s < 4; s++) f[s] = tile[(4 * s + g) * ld + col0 + i];
        return f;
    }
};

template <typename T, int TR, int TCN, int WR, int WCN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    static_assert(WR * WCN == 4, "4 waves per block");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BKP = WgFrag<T>::BKP;
    constexpr int BLK_R = 16 * TR * WR, BLK_C = 16 * TCN * WCN;
    constexpr int LDR = BLK_R + 16, LDC = BLK_C + 16;  // padded leading dims (elements)
    constexpr int CPR_X = BLK_R / EPC, CPR_D = BLK_C / EPC;  // 16-byte chunks per pixel row
    constexpr int NXQ = (BKP * CPR_X + 255) / 256, NDQ = (BKP * CPR_D + 255) / 256;
    typedef typename Mma<T>::frag frag;

    __shared__ __attribute__((aligned(16))) T lds[2][BKP * (LDR + LDC)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WCN, wcn = wave % WCN;
    const int rtiles = (p.klen + BLK_R - 1) / BLK_R;
    const int tap_i = blockIdx.x / rtiles;
    const int r0 = (blockIdx.x - tap_i * rtiles) * BLK_R;
    const int c0 = blockIdx.y * BLK_C;
    const Tap tap = p.taps[tap_i];
    const bool batched = p.batch_out_stride != 0;
    const int HWo = p.Ho * p.Wo;
    int sample = 0, split = blockIdx.z;
    if (batched) {
        sample = blockIdx.z / p.splits;
        split = blockIdx.z - sample * p.splits;
    }
    const long pix_total = batched ? (long)HWo : (long)p.N * HWo;
    const long pbeg = (long)split * p.pix_per_block;
    long pend = pbeg + p.pix_per_block;
    if (pend > pix_total) pend = pix_total;
    const long poff = batched ? (long)sample * HWo : 0;
    const T* x = (const T*)p.x;
    const T* dy = (const T*)p.dy;

    f32x4 acc[TR][TCN];
#pragma unroll
    for (int a = 0; a < TR; a++)
#pragma unroll
        for (int b = 0; b < TCN; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 xreg[NXQ], dreg[NDQ];
    const bool dy_vec = (p.Co % EPC) == 0;

    auto gload = [&](long pstep) {
#pragma unroll
        for (int q = 0; q < NXQ; q++) {
            const int idx = tid + 256 * q;
            const int pk = idx / CPR_X, cc = idx - pk * CPR_X;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const long pix = pstep + pk;
            if (pk < BKP && pix < pend) {
                const long gp = pix + poff;
                const int n = (int)(gp / HWo);
                const int rem = (int)(gp - (long)n * HWo);
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int iy = oy * p.stride + tap.dy, ix = ox * p.stride + tap.dx;
                const int e0 = r0 + cc * EPC;
                if ((unsigned)iy < (unsigned)p.Hi) {
                    const long base = (((long)n * p.Hi + iy) * p.Wi + ix) * p.Ci;
                    if (p.cdiv == 0) {
                        if ((unsigned)ix < (unsigned)p.Wi && e0 < p.klen)
                            v = *reinterpret_cast<const uint4*>(x + base + e0);
                    } else {
                        T tmp[EPC];
#pragma unroll
                        for (int u = 0; u < EPC; u++) {
                            const int e = e0 + u;
                            const int xx = ix + e / p.cdiv;
                            tmp[u] = (e < p.klen && (unsigned)xx < (unsigned)p.Wi) ? x[base + e] : from_f32<T>(0.f);
                        }
                        v = *reinterpret_cast<const uint4*>(tmp);
                    }
                }
            }
            xreg[q] = v;
        }
#pragma unroll
        for (int q = 0; q < NDQ; q++) {
            const int idx = tid + 256 * q;
            const int pk = idx / CPR_D, cc = idx - pk * CPR_D;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            const long pix = pstep + pk;
            if (pk < BKP && pix < pend) {
                const long base = (pix + poff) * p.Co;
                const int ch = c0 + cc * EPC;
                if (dy_vec) {
                    if (ch < p.Co) v = *reinterpret_cast<const uint4*>(dy + base + ch);
                } else {
                    T tmp[EPC];
#pragma unroll
                    for (int u = 0; u < EPC; u++) tmp[u] = (ch + u < p.Co) ? dy[base + ch + u] : from_f32<T>(0.f);
                    v = *reinterpret_cast<const uint4*>(tmp);
                }
            }
            dreg[q] = v;
        }
    };

    auto sstore = [&](int buf) {
        T* xs = lds[buf];
        T* ds = lds[buf] + BKP * LDR;
#pragma unroll
        for (int q = 0; q < NXQ; q++) {
            const int idx = tid + 256 * q;
            const int pk = idx / CPR_X, cc = idx - pk * CPR_X;
            if (pk < BKP) *reinterpret_cast<uint4*>(xs + pk * LDR + cc * EPC) = xreg[q];
        }
#pragma unroll
        for (int q = 0; q < NDQ; q++) {
            const int idx = tid + 256 * q;
            const int pk = idx / CPR_D, cc = idx - pk * CPR_D;
            if (pk < BKP) *reinterpret_cast<uint4*>(ds + pk * LDC + cc * EPC) = dreg[q];
        }
    };

    auto compute = [&](int buf) {
        const T* xs = lds[buf];
        const T* ds = lds[buf] + BKP * LDR;
        frag a[TR], b[TCN];
#pragma unroll
        for (int t = 0; t < TR; t++) a[t] = WgFrag<T>::load(xs, LDR, (wr * TR + t) * 16, lane);
#pragma unroll
        for (int t = 0; t < TCN; t++) b[t] = WgFrag<T>::load(ds, LDC, (wcn * TCN + t) * 16, lane);
#pragma unroll
        for (int i = 0; i < TR; i++)
#pragma unroll
            for (int j = 0; j < TCN; j++) Mma<T>::mma(a[i], b[j], acc[i][j]);
    };

    if (pbeg < pend) {
        const int nsteps = (int)((pend - pbeg + BKP - 1) / BKP);
        gload(pbeg);
        sstore(0);
        __syncthreads();
        for (int s = 0; s < nsteps; s++) {
            const bool more = s + 1 < nsteps;
            if (more) gload(pbeg + (long)(s + 1) * BKP);
            compute(s & 1);
            if (more) sstore((s + 1) & 1);
            __syncthreads();
        }
    }
    // a slice of a shared tile writes its plane of the workspace; a tile with one slice goes to dw itself
    float* out = (p.splits > 1 ? p.ws + (long)split * p.out_elems : p.dw) + (long)sample * p.batch_out_stride +
                 (long)tap_i * p.tap_out_stride;
    const bool add = p.splits == 1 && p.accumulate;
    const int g = lane >> 4, i = lane & 15;
#pragma unroll
    for (int tr = 0; tr < TR; tr++) {
#pragma unroll
        for (int tc = 0; tc < TCN; tc++) {
            const int col = c0 + (wcn * TCN + tc) * 16 + i;
            if (col >= p.Co) continue;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int row = r0 + (wr * TR + tr) * 16 + 4 * g + u;
                if (row >= p.klen) continue;
                float* dst = out + (long)row * p.Co + col;
                float v = acc[tr][tc][u] * p.alpha;
                if (add) v += *dst;
                *dst = v;
            }
        }
    }
}

// out[i] = (accumulate ? out[i] : 0) + ws[0][i] + ws[1][i] + ... in slice order; planes are `elems` floats apart
__global__ __launch_bounds__(256) void fold_partials_kernel(const float* __restrict__ ws, float* __restrict__ out,
                                                            long elems, int splits, int accumulate) {
    const long first = blockIdx.x * 256L + threadIdx.x;
    if ((elems & 3) == 0 && ((((uintptr_t)ws) | ((uintptr_t)out)) & 15) == 0) {
        const long quads = elems >> 2;
        if (first >= quads) return;
        const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
        f32x4 a = accumulate ? reinterpret_cast<const f32x4*>(out)[first] : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int s = 0; s < splits; s++) a += w4[s * quads + first];
        reinterpret_cast<f32x4*>(out)[first] = a;
    } else {
        for (long i = first * 4; i < first * 4 + 4 && i < elems; i++) {
            float a = accumulate ? out[i] : 0.f;
            for (int s = 0; s < splits; s++) a += ws[s * elems + i];
            out[i] = a;
        }
    }
}

static int fold_partials(const WgradParams& p, hipStream_t s) {
    const long quads = (p.out_elems + 3) / 4;
    hipLaunchKernelGGL(fold_partials_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, p.ws, p.dw,
                       p.out_elems, p.splits, p.accumulate);
    STMI_LAUNCH_CHECK("fold_partials_kernel");
    return STMI_OK;
}

// Tile of the register-staged kernel for a problem.  With stores instead of atomics a finished tile costs
// its bytes once, so the choice follows the operands' widths alone.
struct RegCfg {
    int id, blk_r, blk_c;
};
static RegCfg reg_cfg(const WgradParams& p) {
    static const RegCfg table[5] = {{0, 128, 128}, {1, 64, 64}, {2, 128, 16}, {3, 16, 128}, {4, 32, 32}};
    if (STMI_ENV_SET("STMI_WG_CFG")) {
        const int c = STMI_ENV_INT("STMI_WG_CFG", -1);
        if (c >= 0 && c < 5) return table[c];
    }
    if (p.klen > 32 && p.Co > 32) return table[1];
    if (p.klen > 32) return table[2];
    if (p.Co > 32) return table[3];
    return table[4];
}

static WgPlan reg_plan(const WgradParams& p, int dtype) {
    const RegCfg c = reg_cfg(p);
    const int bkp = dtype == STMI_BF16 ? WgFrag<bf16_t>::BKP : WgFrag<float>::BKP;
    const bool batched = p.batch_out_stride != 0;
    const long pix_total = batched ? (long)p.Ho * p.Wo : (long)p.N * p.Ho * p.Wo;
    const int rtiles = ceil_div(p.klen, c.blk_r), ctiles = ceil_div(p.Co, c.blk_c);
    const long tiles = (long)p.ntaps * rtiles * ctiles * (batched ? p.N : 1);
    const int tb = STMI_ENV_INT("STMI_WG_BLOCKS", 0);
    const long target = tb > 0 ? tb : (batched ? 512 : 1024);
    long splits = (target + tiles - 1) / tiles;
    const long max_splits = (pix_total + 4 * bkp - 1) / (4 * bkp);  // at least 4 steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long ppb = (pix_total + splits - 1) / splits;
    ppb = (ppb + bkp - 1) / bkp * bkp;
    splits = (pix_total + ppb - 1) / ppb;
    WgPlan plan;
    plan.pix_per_block = (int)ppb;
    plan.splits = (int)splits;
    return plan;
}

template <typename T, int TR, int TCN, int WR, int WCN>
static int launch_wg(const WgradParams& p, hipStream_t s) {
    constexpr int BLK_R = 16 * TR * WR, BLK_C = 16 * TCN * WCN;
    const bool batched = p.batch_out_stride != 0;
    const int rtiles = ceil_div(p.klen, BLK_R), ctiles = ceil_div(p.Co, BLK_C);
    dim3 grid((unsigned)(p.ntaps * rtiles), (unsigned)ctiles, (unsigned)(p.splits * (batched ? p.N : 1)));
    hipLaunchKernelGGL((wgrad_kernel<T, TR, TCN, WR, WCN>), grid, dim3(256), 0, s, p);
    STMI_LAUNCH_CHECK("wgrad_kernel");
    return STMI_OK;
}

template <typename T>
static int launch_wg_typed(const WgradParams& p, hipStream_t s) {
    switch (reg_cfg(p).id) {
        case 0: return launch_wg<T, 4, 4, 2, 2>(p, s);
        case 1: return launch_wg<T, 2, 2, 2, 2>(p, s);   // 64 x 64
        case 2: return launch_wg<T, 2, 1, 4, 1>(p, s);   // 128 x 16
        case 3: return launch_wg<T, 1, 2, 1, 4>(p, s);   // 16 x 128
        default: return launch_wg<T, 1, 1, 2, 2>(p, s);  // 32 x 32
    }
}

static thread_local char g_last_wgrad[64] = "";
const char* last_wgrad_kernel() { return g_last_wgrad; }

// plans the launch (which kernel: 2 resident-patch, 1 LDS-DMA, 0 register-staged; how many pixel slices per tile);
// *ws_floats: the workspace it needs
static int plan_wgrad(WgradParams& p, int dtype, long* ws_floats) {
    WgPlan plan;
    int kind = 0;
    if (wgrad_res_plan(p, dtype, &plan)) kind = 2;
    else if (wgrad_glds_plan(p, dtype, &plan)) kind = 1;
    else plan = reg_plan(p, dtype);
    p.splits = plan.splits;
    p.pix_per_block = plan.pix_per_block;
    *ws_floats = plan.splits > 1 ? (long)plan.splits * p.out_elems : 0;
    return kind;
}

static int run_wgrad(WgradParams& p, int dtype, hipStream_t s, const char* who) {
    long need = 0;
    // the LDS-DMA kernel wants 16-byte aligned planes: plan with the workspace the call will really use
    const int kind = plan_wgrad(p, dtype, &need);
    STMI_REQUIRE(need == 0 || p.ws, "%s: this geometry cuts every output tile into %d pixel slices and needs a "
                 "workspace of %ld floats (see the *_ws_floats query)", who, p.splits, need);
    snprintf(g_last_wgrad, sizeof(g_last_wgrad), "%s", kind == 2 ? "wgrad_res_kernel" : kind == 1 ? "wgrad_glds_kernel" : "wgrad_kernel");
    int rc = kind == 2 ? launch_wgrad_res(p, dtype, s)
             : kind == 1 ? launch_wgrad_glds(p, dtype, s)
                         : (dtype == STMI_BF16 ? launch_wg_typed<bf16_t>(p, s) : launch_wg_typed<float>(p, s));
    if (rc) return rc;
    return p.splits > 1 ? fold_partials(p, s) : STMI_OK;
}

static int conv_wgrad_params(WgradParams& p, const void* x, const void* dy, float* dw, float* ws, int n, int h, int w,
                             int ci, int co, int kh, int kw, int stride, int accumulate) {
    int ho, wo, pt, pl;
    same_pads(h, kh, stride, &ho, &pt);
    same_pads(w, kw, stride, &wo, &pl);
    const bool row = (ci % 8) != 0;
    STMI_REQUIRE(!row || stride == 1, "conv wgrad: fewer than 8 input channels needs stride 1 (ci=%d)", ci);
    memset(&p, 0, sizeof(p));
    p.x = x; p.dy = dy; p.dw = dw; p.ws = ws;
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = ci;
    p.Ho = ho; p.Wo = wo; p.Co = co;
    p.stride = stride;
    p.alpha = 1.f;
    p.accumulate = accumulate;
    p.ntaps = row ? kh : kh * kw;
    STMI_REQUIRE(p.ntaps <= kMaxTaps, "conv wgrad: %d taps exceed the table", p.ntaps);
    p.klen = row ? kw * ci : ci;
    p.cdiv = row ? ci : 0;
    p.tap_out_stride = (long)p.klen * co;
    p.out_elems = (long)kh * kw * ci * co;
    for (int t = 0; t < p.ntaps; t++) {
        if (row) {
            p.taps[t].dy = (short)(t - pt); p.taps[t].dx = (short)(-pl);
        } else {
            p.taps[t].dy = (short)(t / kw - pt); p.taps[t].dx = (short)(t % kw - pl);
        }
        p.taps[t].widx = t;
    }
    return STMI_OK;
}

// dw fp32 [kh,kw,ci,co] of conv(n,h,w,ci -> co, k, stride): overwritten, or added to when `accumulate`
int conv_wgrad_impl(const void* x, const void* dy, float* dw, float* ws, int n, int h, int w, int ci, int co, int kh,
                    int kw, int stride, int accumulate, int dtype, hipStream_t s) {
    WgradParams p;
    int rc = conv_wgrad_params(p, x, dy, dw, ws, n, h, w, ci, co, kh, kw, stride, accumulate);
    if (rc) return rc;
    return run_wgrad(p, dtype, s, "conv wgrad");
}

// floats of workspace conv_wgrad_impl needs for this geometry; a 16-byte aligned workspace is assumed
long conv_wgrad_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    WgradParams p;
    if (conv_wgrad_params(p, nullptr, nullptr, nullptr, nullptr, n, h, w, ci, co, kh, kw, stride, 0)) return 0;
    long need = 0;
    plan_wgrad(p, dtype, &need);
    // the register-staged kernel (which an unaligned tensor falls back to) may cut differently: the larger need
    const WgPlan reg = reg_plan(p, dtype);
    const long need_reg = reg.splits > 1 ? (long)reg.splits * p.out_elems : 0;
    return need > need_reg ? need : need_reg;
}

static void gram_params(WgradParams& p, const void* f, float* g, float* ws, int n, int h, int w, int c, int accumulate) {
    memset(&p, 0, sizeof(p));
    p.x = f; p.dy = f; p.dw = g; p.ws = ws;
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = c;
    p.Ho = h; p.Wo = w; p.Co = c;
    p.stride = 1;
    p.alpha = 1.f / (float)((long)h * w);
    p.accumulate = accumulate;
    p.ntaps = 1; p.klen = c; p.cdiv = 0;
    p.tap_out_stride = 0;
    p.batch_out_stride = (long)c * c;
    p.out_elems = (long)n * c * c;
}

// g fp32 [n,c,c] = F^T F / (h*w): overwritten, or added to when `accumulate`
int gram_fwd_impl(const void* f, float* g, float* ws, int n, int h, int w, int c, int dtype, hipStream_t s,
                  int accumulate) {
    STMI_REQUIRE(c % 8 == 0, "gram: channel count must be a multiple of 8 (c=%d)", c);
    WgradParams p;
    gram_params(p, f, g, ws, n, h, w, c, accumulate);
    return run_wgrad(p, dtype, s, "gram");
}

long gram_ws_floats(int n, int h, int w, int c, int dtype) {
    if (c % 8 != 0) return 0;
    WgradParams p;
    gram_params(p, nullptr, nullptr, nullptr, n, h, w, c, 0);
    long need = 0;
    plan_wgrad(p, dtype, &need);
    const WgPlan reg = reg_plan(p, dtype);
    const long need_reg = reg.splits > 1 ? (long)reg.splits * p.out_elems : 0;
    return need > need_reg ? need : need_reg;
}

}  // namespace stmi
