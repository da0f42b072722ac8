// Stride-1 SAME convolution of a 3-channel fp32 image (gfx950): the stylizer's first 9x9 layer
// (full_graph_stylizer_network.py:47 -> model_utilities.py:235-242) and VGG conv1_1 (vgg.py:92-96,
// with the preprocessing of vgg.py:11-17 applied while the image is read).
//
// With 3 input channels a tap has 6 bytes of K: no gather-GEMM tile shape fits it.  Here a block
// owns an 8 x 32 rectangle of output pixels and keeps its input patch in LDS with the pixels
// padded to FOUR channels (one pixel = 4 elements = 8 bytes of bf16 / 16 bytes of fp32).  A
// kernel row then is a run of RUNPX = K + 1 pixels = 4 * RUNPX contiguous K elements at a pixel-
// aligned LDS address, so the MFMA's 16-byte K fragments are plain LDS reads at 8-byte alignment
// and the whole K axis is  ky-major runs: k = (ky * RUNPX + kx) * 4 + c, zero weights where
// kx >= K or c == 3.  The weights (CO rows of KL = round_up(4 * K * RUNPX, 32) elements, packed with
// STMI_PACK_FWD_RGB) sit in LDS beside the patch for the whole block.
//
// The image is read as fp32 and converted on the way in (x * scale - mean[c]), which folds the
// cast of the stylizer's input and VGG's preprocessing into the convolution; the input never
// exists as a 32-wide row-patch tensor.  Per output pixel the kernel reads 12 bytes and writes
// 2 * CO (bf16): it is bound by its stores.
#include "conv_gemm.hpp"

namespace stmi {

namespace {
constexpr int kRgbPH = 8, kRgbPW = 32;  // output pixels per block: 4 waves x (2 rows x 32)
}

// XT: element type of the image: float (forward layers), unsigned char (8-bit frames straight from a decoder:
// x / 255 is `scale`), or T (backward-data of a layer that WRITES 3 channels: its input gradient is this same
// convolution of dy with the flipped, transposed kernel).
// STATS: the block also leaves the per-tile sums of its output (ConvGemmParams::tile_stats, mode 1; one row per
// block) for the instance norm behind the stylizer's first layer.  That form has its own epilogue: a wave
// stages its 2 x 32 pixels as [64][CO] rows of T, stores them as whole rows and takes the sums from the staged
// rows (conv_gemm.hpp, wave_stats_mma / WaveStatsLanes); no bias, ReLU or pre-activation copy.
template <typename T, int K, int TC, typename XT, bool STATS>
__global__ __launch_bounds__(256) void conv_rgb_kernel(const XT* __restrict__ x, float scale, const float* mean,
                                                       const T* __restrict__ wq, const ConvGemmParams p) {
    constexpr int EPC = 16 / (int)sizeof(T);     // elements per 16-byte K fragment
    constexpr int RUNPX = K + 1;                 // pixels per kernel-row run (K is odd)
    constexpr int CPK = RUNPX * 4 / EPC;         // fragments per run
    constexpr int KL = (4 * K * RUNPX + 31) / 32 * 32;
    constexpr int NCH = KL / EPC, NSTEP = NCH / 4;
    constexpr int CO = 16 * TC, TP = 4;
    constexpr int PH = kRgbPH, PW = kRgbPW, PAD = (K - 1) / 2;
    constexpr int PHH = PH + K - 1, PWH = PW + RUNPX - 1;
    constexpr int PIX_B = 4 * (int)sizeof(T);    // bytes per padded pixel
    constexpr int W_STRIDE = KL * (int)sizeof(T) + 16;  // +16 bytes: sixteen rows hit sixteen bank groups
    constexpr int W_BYTES = CO * W_STRIDE;
    static_assert(RUNPX * 4 % EPC == 0 && NCH % 4 == 0, "runs and steps must be whole fragments");
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl = lds;
    char* pl = lds + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Hi, W = p.Wi;
    const int tiles_x = (W + PW - 1) / PW, tiles_y = (H + PH - 1) / PH;
    // Blocks b and b + 8 run on the same XCD: every XCD gets one contiguous run of tiles (row-major), so that the
    // halo a tile shares with its neighbours is fetched through ONE 4 MiB L2 (in plain block order the eight L2s
    // each fetched it: 2.5x the input from memory, by the counters).
    int t;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        t = (int)((xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot);
    }
    const int n = t / (tiles_x * tiles_y);
    t -= n * tiles_x * tiles_y;
    const int ty0 = (t / tiles_x) * PH, tx0 = (t % tiles_x) * PW;

    // weights -> LDS, 16 bytes per access
    for (int i = tid; i < CO * NCH; i += 256) {
        const int row = i / NCH, q = i - row * NCH;
        *reinterpret_cast<uint4*>(wl + row * W_STRIDE + q * 16) = *reinterpret_cast<const uint4*>(wq + (long)row * KL + q * EPC);
    }
    // patch -> LDS: fp32 image elements, preprocessed, converted; zeros outside the image and in channel 3
    {
        const XT* img = x + (long)n * H * W * 3;
        float m3[3] = {0.f, 0.f, 0.f};
        if (mean) { m3[0] = mean[0]; m3[1] = mean[1]; m3[2] = mean[2]; }
        for (int i = tid; i < PHH * PWH; i += 256) {
            const int r = i / PWH, c = i - r * PWH;
            const int iy = ty0 - PAD + r, ix = tx0 - PAD + c;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const XT* s = img + ((long)iy * W + ix) * 3;
                v[0] = to_f32<XT>(s[0]) * scale - m3[0];
                v[1] = to_f32<XT>(s[1]) * scale - m3[1];
                v[2] = to_f32<XT>(s[2]) * scale - m3[2];
            }
            T* d = reinterpret_cast<T*>(pl + (r * PWH + c) * PIX_B);
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = from_f32<T>(v[u]);
        }
    }
    __syncthreads();

    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
    // patch position under tap (0, 0) of this lane's pixel in each of the wave's four fragments
    int pbase[TP];
#pragma unroll
    for (int tp = 0; tp < TP; tp++) {
        const int f = wave * TP + tp;  // fragment f: row f / 2, columns (f % 2) * 16 ..
        pbase[tp] = ((f >> 1) * PWH + (f & 1) * 16 + r16) * PIX_B;
    }
#pragma unroll
    for (int s = 0; s < NSTEP; s++) {
        const int q = 4 * s + g;
        int ky = q / CPK;
        const int cx = q - ky * CPK;
        ky = ky < K ? ky : K - 1;  // past the last run the weights are zero: read something valid
        const int koff = (ky * PWH) * PIX_B + cx * 16;
        frag a[TC], b[TP];
#pragma unroll
        for (int tc = 0; tc < TC; tc++) a[tc] = *reinterpret_cast<const frag*>(wl + (tc * 16 + r16) * W_STRIDE + q * 16);
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
            const char* src = pl + pbase[tp] + koff;  // 8-byte aligned for bf16, 16 for fp32
            uint2 lo = *reinterpret_cast<const uint2*>(src), hi = *reinterpret_cast<const uint2*>(src + 8);
            uint4 v = {lo.x, lo.y, hi.x, hi.y};
            b[tp] = __builtin_bit_cast(frag, v);
        }
#pragma unroll
        for (int tc = 0; tc < TC; tc++)
#pragma unroll
            for (int tp = 0; tp < TP; tp++) Mma<T>::mma(a[tc], b[tp], acc[tc][tp]);
    }

    if constexpr (STATS) {
        constexpr int SROW = CO * (int)sizeof(T) + 16;    // bytes
        constexpr int LPR = CO / EPC;                     // lanes per pixel row
        __syncthreads();                                  // every wave is done with the weights and the patch
        char* stage = lds + wave * (64 * SROW);
        // staged row px of the wave = tile pixel (2 wave + px / 32, px % 32); fragment tp covers px 16 tp ..
#pragma unroll
        for (int tc = 0; tc < TC; tc++)
#pragma unroll
            for (int tp = 0; tp < TP; tp++) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = acc[tc][tp][u];
                store_vec<T, 4>(reinterpret_cast<T*>(stage + (tp * 16 + r16) * SROW) + tc * 16 + 4 * g, v);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own rows: no barrier
        const int oy0 = ty0 + 2 * wave;
        const int rows_valid = max(0, min(2, H - oy0)), px_valid = min(PW, W - tx0);
        const int lp = lane / LPR, lc = (lane % LPR) * EPC;
        WaveStatsLanes<T, CO> vs;
        vs.init();
        if constexpr (sizeof(T) == 4) load_vec<T, EPC>(reinterpret_cast<const T*>(stage + lc * (int)sizeof(T)), vs.sk);
        T* out = (T*)p.out;
#pragma unroll
        for (int i = 0; i < LPR; i++) {
            const int px = lp + i * (64 / LPR);
            const int oy = oy0 + (px >> 5), ox = tx0 + (px & 31);
            if (oy < H && ox < W) {
                const uint4 v = *reinterpret_cast<const uint4*>(stage + px * SROW + lc * (int)sizeof(T));
                *reinterpret_cast<uint4*>(out + (((long)n * H + oy) * W + ox) * p.Co + lc) = v;
                if constexpr (sizeof(T) == 4) vs.add(v);
            }
        }
        float* part = reinterpret_cast<float*>(lds);      // [wave][plane k, s1, s2, count][CO]
        f32x4 acc_s[TC], acc_q[TC];
        float k_first[TC];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int k = 0; k < TC; k++) {
                acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                k_first[k] = to_f32<T>(reinterpret_cast<const T*>(stage)[k * 16 + r16]);
            }
            // pixels outside the image were computed but are not part of the tensor: their rows become zeros
            if (rows_valid * px_valid < 64 && ((lane >> 5) >= rows_valid || (lane & 31) >= px_valid)) {
#pragma unroll
                for (int c = 0; c < CO * (int)sizeof(T); c += 16)
                    *reinterpret_cast<uint4*>(stage + lane * SROW + c) = make_uint4(0u, 0u, 0u, 0u);
            }
            asm volatile("" ::: "memory");
            if (rows_valid > 0) wave_stats_mma<CO>(stage, SROW, lane, acc_s, acc_q);
        } else {
            vs.reduce();
        }
        lds_barrier();                                    // every wave is done with its staging rows
        if constexpr (sizeof(T) == 2)
            wave_stats_part_mma<CO>(part + wave * 4 * CO, lane, acc_s, acc_q, k_first, (float)(rows_valid * px_valid));
        else
            vs.write_part(part + wave * 4 * CO, lane);
        lds_barrier();
        // wave 0's first pixel (ty0, tx0) is always inside the image
        block_stats_row<CO>(part, tid, p.tile_stats + (((long)n * (tiles_x * tiles_y) + t) * 4) * p.Co, p.Co);
        return;
    }
    constexpr int TPG = conv_stage_tpg(TC, TP, 1, 4, 160 * 1024 / 2);
    static_assert(TPG >= 1, "staging area");
    const long img_row = (long)n * H;
    conv_epilogue_lds<T, TC, TP, 1, 4, 256, TPG>(p, acc, 0, reinterpret_cast<float*>(lds), [&](int pix) -> long {
        const int f = pix >> 4;
        const int oy = ty0 + (f >> 1), ox = tx0 + (f & 1) * 16 + (pix & 15);
        return oy < H && ox < W ? ((img_row + oy) * W + ox) * p.Co : -1;
    });
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
int conv_rgb_klen(int k) { return (4 * k * (k + 1) + 31) / 32 * 32; }

template <typename T, int K, int TC, typename XT, bool STATS = false>
static int launch_rgb(const XT* x, float scale, const float* mean, const void* wq, const ConvGemmParams& p,
                      hipStream_t s) {
    constexpr int KL = (4 * K * (K + 1) + 31) / 32 * 32;
    constexpr int CO = 16 * TC;
    constexpr int W_BYTES = CO * (KL * (int)sizeof(T) + 16);
    constexpr int P_BYTES = (kRgbPH + K - 1) * (kRgbPW + K) * 4 * (int)sizeof(T);
    constexpr int TPG = conv_stage_tpg(TC, 4, 1, 4, 160 * 1024 / 2);
    constexpr int S_BYTES = 16 * TPG * 4 * (CO + 4) * 4;
    constexpr int S_WAVES = 4 * 64 * (CO * (int)sizeof(T) + 16);   // the STATS epilogue's per-wave rows
    constexpr int S_NEED = STATS ? S_WAVES : S_BYTES;
    constexpr size_t kLds = (size_t)((W_BYTES + P_BYTES > S_NEED ? W_BYTES + P_BYTES : S_NEED) + 15) / 16 * 16;
    STMI_DYN_LDS((conv_rgb_kernel<T, K, TC, XT, STATS>), (int)kLds, "hipFuncSetAttribute(conv_rgb_kernel)");
    {
        char name[160];
        const char* tn = sizeof(T) == 2 ? "__bf16" : "float";
        const char* xn = sizeof(XT) == 1 ? "unsigned char" : (sizeof(XT) == 2 ? "__bf16" : "float");
        snprintf(name, sizeof(name), "void stmi::conv_rgb_kernel<%s, %d, %d, %s, %s>(%s const*, float, float const*, %s const*, stmi::ConvGemmParams)",
                 tn, K, TC, xn, STATS ? "true" : "false", xn, tn);
        set_last_conv_kernel(name);
    }
    const long tiles = (long)p.N * ceil_div(p.Hi, kRgbPH) * ceil_div(p.Wi, kRgbPW);
    if (tiles <= 0) return STMI_OK;
    STMI_REQUIRE(tiles < (1L << 31), "conv_rgb: too many tiles for one launch");
    hipLaunchKernelGGL((conv_rgb_kernel<T, K, TC, XT, STATS>), dim3((unsigned)tiles), dim3(256), kLds, s, x, scale, mean, (const T*)wq, p);
    STMI_LAUNCH_CHECK("conv_rgb_kernel");
    return STMI_OK;
}

// Statistics rows per sample conv_rgb_fwd_impl(..., tile_stats) leaves for (h, w) planes: one per 8 x 32 block of
// output pixels (0: this configuration has none).
int conv_rgb_tile_stats_tiles(int h, int w, int co, int k) {
    if (!(k == 9 && co == 32)) return 0;
    return ceil_div(h, kRgbPH) * ceil_div(w, kRgbPW);
}

int conv_rgb_fwd_impl(const float* x, float scale, const float* mean, const void* wq, const float* bias, void* y,
                      void* y_pre, int n, int h, int w, int co, int k, int relu, int dtype, hipStream_t s,
                      float* tile_stats) {
    ConvGemmParams p;
    memset(&p, 0, sizeof(p));
    p.out = y; p.out_pre = y_pre; p.bias = bias; p.alpha = 1.f; p.relu = relu;
    p.tile_stats = tile_stats; p.tile_stats_mode = tile_stats ? 1 : 0;
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = 3;
    p.Ho = h; p.Wo = w; p.Co = co;
    const uintptr_t ptrs = (uintptr_t)y | (uintptr_t)y_pre | (uintptr_t)wq;
    STMI_REQUIRE((ptrs & 15) == 0, "conv_rgb: y, y_pre and the packed weights must be 16-byte aligned");
    STMI_REQUIRE((long)n * h * w * 3 < (1L << 31), "conv_rgb: image batch too large for 32-bit indexing");
    p.row_store_ok = 1;
    const bool bf = dtype == STMI_BF16;
    if (tile_stats) {
        STMI_REQUIRE(k == 9 && co == 32 && !bias && !relu && !y_pre, "conv_rgb: tile statistics are built for the plain (9, 32) layer");
        return bf ? launch_rgb<bf16_t, 9, 2, float, true>(x, scale, mean, wq, p, s) : launch_rgb<float, 9, 2, float, true>(x, scale, mean, wq, p, s);
    }
    if (k == 9 && co == 32) return bf ? launch_rgb<bf16_t, 9, 2, float>(x, scale, mean, wq, p, s) : launch_rgb<float, 9, 2, float>(x, scale, mean, wq, p, s);
    if (k == 3 && co == 64) return bf ? launch_rgb<bf16_t, 3, 4, float>(x, scale, mean, wq, p, s) : launch_rgb<float, 3, 4, float>(x, scale, mean, wq, p, s);
    if (k == 3 && co == 32) return bf ? launch_rgb<bf16_t, 3, 2, float>(x, scale, mean, wq, p, s) : launch_rgb<float, 3, 2, float>(x, scale, mean, wq, p, s);
    set_error("conv_rgb: unsupported (k, co) = (%d, %d); built for (9, 32), (3, 64) and (3, 32)", k, co);
    return STMI_ERR_INVALID;
}

// The plain (9, 32) layer on 8-bit frames [n,h,w,3]: conv(x * scale - mean) computed in dtype; tile_stats optional.
int conv_rgb_u8_fwd_impl(const unsigned char* x, float scale, const float* mean, const void* wq, void* y, int n, int h,
                         int w, int co, int k, int dtype, hipStream_t s, float* tile_stats) {
    ConvGemmParams p;
    memset(&p, 0, sizeof(p));
    p.out = y; p.alpha = 1.f;
    p.tile_stats = tile_stats; p.tile_stats_mode = tile_stats ? 1 : 0;
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = 3;
    p.Ho = h; p.Wo = w; p.Co = co;
    STMI_REQUIRE((((uintptr_t)y | (uintptr_t)wq) & 15) == 0, "conv_rgb: y and the packed weights must be 16-byte aligned");
    STMI_REQUIRE((long)n * h * w * 3 < (1L << 31), "conv_rgb: image batch too large for 32-bit indexing");
    STMI_REQUIRE(k == 9 && co == 32, "conv_rgb (8-bit frames): built for the (9, 32) layer, got (%d, %d)", k, co);
    p.row_store_ok = 1;
    const bool bf = dtype == STMI_BF16;
    if (tile_stats)
        return bf ? launch_rgb<bf16_t, 9, 2, unsigned char, true>(x, scale, mean, wq, p, s)
                  : launch_rgb<float, 9, 2, unsigned char, true>(x, scale, mean, wq, p, s);
    return bf ? launch_rgb<bf16_t, 9, 2, unsigned char>(x, scale, mean, wq, p, s)
              : launch_rgb<float, 9, 2, unsigned char>(x, scale, mean, wq, p, s);
}

// Backward-data of a k x k stride-1 SAME convolution ci -> 3 (the stylizer's last layer,
// full_graph_stylizer_network.py:78): dx[n,h,w,ci] = conv(dy[n,h,w,3], flipped and transposed kernel),
// the weights packed with STMI_PACK_DGRAD_RGB.
int conv_rgb_dgrad_impl(const void* dy, const void* wq, void* dx, int n, int h, int w, int ci, int k, int dtype,
                        hipStream_t s) {
    ConvGemmParams p;
    memset(&p, 0, sizeof(p));
    p.out = dx; p.alpha = 1.f;
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = 3;
    p.Ho = h; p.Wo = w; p.Co = ci;
    STMI_REQUIRE((((uintptr_t)dx | (uintptr_t)wq) & 15) == 0, "conv_rgb dgrad: dx and the packed weights must be 16-byte aligned");
    STMI_REQUIRE((long)n * h * w * 3 < (1L << 31), "conv_rgb dgrad: gradient batch too large for 32-bit indexing");
    p.row_store_ok = 1;
    if (k == 9 && ci == 32) {
        return dtype == STMI_BF16 ? launch_rgb<bf16_t, 9, 2, bf16_t>((const bf16_t*)dy, 1.f, nullptr, wq, p, s)
                                  : launch_rgb<float, 9, 2, float>((const float*)dy, 1.f, nullptr, wq, p, s);
    }
    set_error("conv_rgb dgrad: unsupported (k, ci) = (%d, %d); built for (9, 32)", k, ci);
    return STMI_ERR_INVALID;
}

}  // namespace stmi
