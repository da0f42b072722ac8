// Stride-1 SAME convolution that WRITES three channels (gfx950): the stylizer's last 9x9 layer
// (full_graph_stylizer_network.py:78) and the backward-data of VGG conv1_1 (a 3x3 convolution
// 64 -> 3 with the flipped, transposed kernel).
//
// With 3 output channels the MFMA's 16 output rows would be 81 % padding.  Instead the kernel
// row is moved to the OUTPUT side: t[pixel][(kx, co)] = sum_ky sum_ci x[pixel @ ky][ci] * W[ky][ci][(kx, co)]
// is an ordinary GEMM with kw * 3 (27 of 32, or 9 of 16) live output rows and only kh vertical
// taps, and y[oy][ox][co] = sum_kx t[oy][ox + kx - pad][(kx, co)] folds it back.  The weights are
// the STMI_PACK_FWD_ROWOUT layout.  A block owns 8 x PW output pixels, keeps the (8 + K - 1) x
// (PW + K - 1) input patch and all weights in LDS, computes t for its 8 x (PW + K - 1) columns
// with MFMA, leaves it in LDS as fp32 and folds it there: the (kx, co) tensor never reaches
// memory (as a separate pass it costs a 32-wide write and read per pixel, more than the layer's
// input), and the patch is fetched once instead of once per vertical tap.
#include "conv_gemm.hpp"

namespace stmi {

namespace {
template <int RB>
__device__ __forceinline__ int to3_swz(int r) {
    return RB == 128 ? (r >> 1) & 7 : ((r >> 3) & 1) << 1;
}
constexpr int kTo3PH = 8;
constexpr int kTo3Stride = 33;  // floats per pixel of t in LDS: odd, so the fold reads without bank conflicts
}  // namespace

// K: kernel size; RB: bytes of input channels per pixel (64 or 128); COLS = PW + K - 1 columns of t per
// block (a multiple of 8, so that the 8 x COLS pixels are whole 16-pixel fragments for each of the 4 waves).
// CLAMP (bf16): the launch absorbs the instance norm + ReLU in front of it (see ConvGemmParams::clamp_t): the patch
// goes to LDS as max(x, t[n][ci]) -- t itself outside the image --, the weights are the sample's own scaled copy
// (w_stride elements apart), and the fold starts from the constant terms of its (kx, co) rows (bias_rows [n][32]).
struct To3Clamp {
    const void* t;          // [N][CI] of T
    const float* bias_rows; // [N][32]
    long w_stride;
};

template <typename T, int K, int RB, int COLS, bool CLAMP = false>
__global__ __launch_bounds__(256) void conv_to3_kernel(const T* __restrict__ x, const T* __restrict__ wq,
                                                       const T* __restrict__ addend, T* __restrict__ y, int N, int H, int W,
                                                       To3Clamp cl) {
    static_assert(!CLAMP || sizeof(T) == 2, "the clamp compares bf16 bit patterns");
    constexpr int EPC = 16 / (int)sizeof(T);
    constexpr int CI = RB / (int)sizeof(T);
    constexpr int CPR = RB / 16;               // 16-byte chunks per pixel row
    constexpr int KS = RB / 64;                // MFMA K steps per vertical tap
    constexpr int PH = kTo3PH, PW = COLS - (K - 1), PAD = (K - 1) / 2;
    constexpr int PROWS = PH + K - 1;
    constexpr int TC = K * 3 > 16 ? 2 : 1;     // 16-row tiles of (kx, co) outputs
    constexpr int NF = PH * COLS / 64;         // 16-pixel fragments per wave
    constexpr int W_BYTES = K * 32 * RB;       // the packed weights keep 32 rows per tap
    constexpr int P_PIX = PROWS * COLS;
    static_assert(PH * COLS % 64 == 0 && COLS % 8 == 0, "t columns must split into whole fragments per wave");
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* wl = lds;
    char* pl = lds + W_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // weights and patch -> LDS, 16 bytes per access, chunk p of row r at position p ^ swz(r)
    const T* wn = CLAMP ? wq + (long)n * cl.w_stride : wq;
    for (int i = tid; i < K * 32 * CPR; i += 256) {
        const int row = i / CPR, pos = i - row * CPR;
        const int chunk = pos ^ to3_swz<RB>(row & 31);
        *reinterpret_cast<uint4*>(wl + row * RB + pos * 16) = *reinterpret_cast<const uint4*>(wn + (long)row * CI + chunk * EPC);
    }
    {
        constexpr int TOTAL = P_PIX * CPR, ROUNDS = (TOTAL + 255) / 256;
        const T* img = x + (long)n * H * W * CI;
        uint4 v[ROUNDS];
        // CLAMP: the clamp values of the two chunks this thread ever loads (position tid % CPR, swizzled or not)
        static_assert(!CLAMP || (RB == 64 && 256 % CPR == 0), "one chunk position per thread, two swizzle states");
        uint4 tq_plain = make_uint4(0u, 0u, 0u, 0u), tq_swz = tq_plain;
        if constexpr (CLAMP) {
            const T* t_n = (const T*)cl.t + (long)n * CI;
            tq_plain = *reinterpret_cast<const uint4*>(t_n + (tid % CPR) * EPC);
            tq_swz = *reinterpret_cast<const uint4*>(t_n + ((tid % CPR) ^ 2) * EPC);
        }
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            const int i = tid + j * 256;
            const int pr = i / CPR, pos = i - pr * CPR;
            const int r = pr / COLS, c = pr - r * COLS;
            const int iy = ty0 - PAD + r, ix = tx0 - PAD + c;
            v[j] = make_uint4(0u, 0u, 0u, 0u);
            const int swz = to3_swz<RB>(pr);
            const int chunk = pos ^ swz;
            uint4 tq = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (CLAMP) {
                tq = swz ? tq_swz : tq_plain;   // (a thread's chunk position is the same in every round)
                v[j] = tq;   // outside the image: the value the absorbed norm maps to zero
            }
            if (i < TOTAL && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
                v[j] = *reinterpret_cast<const uint4*>(img + ((long)iy * W + ix) * CI + chunk * EPC);
        }
        // (the maxima come in a second loop: behind each load they would wait for it, one round trip per round)
#pragma unroll
        for (int j = 0; j < ROUNDS; j++) {
            const int i = tid + j * 256;
            if constexpr (CLAMP) {   // bf16 bit patterns order like fp16 ones: one packed maximum per register
                const uint4 tq = to3_swz<RB>(i / CPR) ? tq_swz : tq_plain;
                asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[j].x) : "v"(tq.x));
                asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[j].y) : "v"(tq.y));
                asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[j].z) : "v"(tq.z));
                asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(v[j].w) : "v"(tq.w));
            }
            if (i < TOTAL) *reinterpret_cast<uint4*>(pl + i * 16) = v[j];
        }
    }
    __syncthreads();

    f32x4 acc[TC][NF];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < NF; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int ky = 0; ky < K; ky++) {
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            frag a[TC], b[NF];
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
                const int row = tc * 16 + r16;
                a[tc] = *reinterpret_cast<const frag*>(wl + (ky * 32 + row) * RB + (((4 * kk + g) ^ to3_swz<RB>(row)) << 4));
            }
#pragma unroll
            for (int nf = 0; nf < NF; nf++) {
                const int pr = (wave * NF + nf) * 16 + r16 + ky * COLS;  // pixel (py, px') under vertical tap ky
                b[nf] = *reinterpret_cast<const frag*>(pl + pr * RB + (((4 * kk + g) ^ to3_swz<RB>(pr)) << 4));
            }
#pragma unroll
            for (int tc = 0; tc < TC; tc++)
#pragma unroll
                for (int nf = 0; nf < NF; nf++) Mma<T>::mma(a[tc], b[nf], acc[tc][nf]);
        }
    }

    // t -> LDS (over the operands), then the fold over kx
    __syncthreads();
    float* tl = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int tc = 0; tc < TC; tc++)
#pragma unroll
        for (int nf = 0; nf < NF; nf++) {
            const int pp = (wave * NF + nf) * 16 + r16;
#pragma unroll
            for (int u = 0; u < 4; u++) tl[pp * kTo3Stride + tc * 16 + 4 * g + u] = acc[tc][nf][u];
        }
    __syncthreads();
    for (int o = tid; o < PH * PW; o += 256) {
        const int py = o / PW, px = o - py * PW;
        const int oy = ty0 + py, ox = tx0 + px;
        if (oy >= H || ox >= W) continue;
        float s[3] = {0.f, 0.f, 0.f};
        const float* row = tl + (py * COLS + px) * kTo3Stride;
#pragma unroll
        for (int kx = 0; kx < K; kx++) {
#pragma unroll
            for (int co = 0; co < 3; co++) {
                s[co] += row[kx * kTo3Stride + kx * 3 + co];
                if constexpr (CLAMP) s[co] += cl.bias_rows[n * 32 + kx * 3 + co];
            }
        }
        const long e = (((long)n * H + oy) * W + ox) * 3;
#pragma unroll
        for (int co = 0; co < 3; co++) {
            float v = s[co];
            if (addend) v += to_f32<T>(addend[e + co]);
            y[e + co] = from_f32<T>(v);
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, int K, int RB, int COLS, bool CLAMP = false>
static int launch_to3(const void* x, const void* wq, const void* addend, void* y, int n, int h, int w, hipStream_t s,
                      To3Clamp cl = To3Clamp{nullptr, nullptr, 0}) {
    constexpr int PW = COLS - (K - 1);
    constexpr int OPS = K * 32 * RB + (kTo3PH + K - 1) * COLS * RB;
    constexpr int TBYTES = kTo3PH * COLS * kTo3Stride * 4;
    constexpr size_t kLds = (size_t)((OPS > TBYTES ? OPS : TBYTES) + 15) / 16 * 16;
    STMI_DYN_LDS((conv_to3_kernel<T, K, RB, COLS, CLAMP>), (int)kLds, "hipFuncSetAttribute(conv_to3_kernel)");
    {
        char name[160];
        const char* tn = sizeof(T) == 2 ? "__bf16" : "float";
        snprintf(name, sizeof(name), "void stmi::conv_to3_kernel<%s, %d, %d, %d, %s>(%s const*, %s const*, %s const*, %s*, int, int, int, stmi::To3Clamp)",
                 tn, K, RB, COLS, CLAMP ? "true" : "false", tn, tn, tn, tn);
        set_last_conv_kernel(name);
    }
    const long tiles = (long)n * ceil_div(h, kTo3PH) * ceil_div(w, PW);
    if (tiles <= 0) return STMI_OK;
    STMI_REQUIRE(tiles < (1L << 31), "conv_to3: too many tiles for one launch");
    hipLaunchKernelGGL((conv_to3_kernel<T, K, RB, COLS, CLAMP>), dim3((unsigned)tiles), dim3(256), kLds, s, (const T*)x, (const T*)wq,
                       (const T*)addend, (T*)y, n, h, w, cl);
    STMI_LAUNCH_CHECK("conv_to3_kernel");
    return STMI_OK;
}

// 0 when no instantiation exists for (k, ci, dtype)
int conv_to3_supported(int k, int ci, int dtype) {
    const int rb = ci * (dtype == STMI_BF16 ? 2 : 4);
    return ((k == 9 && ci == 32) || (k == 3 && (ci == 64 || ci == 32))) && (rb == 64 || rb == 128);
}

int conv_to3_impl(const void* x, const void* wq, const void* addend, void* y, int n, int h, int w, int ci, int k,
                  int dtype, hipStream_t s) {
    STMI_REQUIRE(conv_to3_supported(k, ci, dtype), "conv_to3: unsupported (k, ci, dtype) = (%d, %d, %d)", k, ci, dtype);
    STMI_REQUIRE((((uintptr_t)x | (uintptr_t)wq) & 15) == 0, "conv_to3: x and the packed weights must be 16-byte aligned");
    const bool bf = dtype == STMI_BF16;
    if (k == 9) {  // ci == 32
        return bf ? launch_to3<bf16_t, 9, 64, 40>(x, wq, addend, y, n, h, w, s) : launch_to3<float, 9, 128, 40>(x, wq, addend, y, n, h, w, s);
    }
    if (ci == 64) return launch_to3<bf16_t, 3, 128, 32>(x, wq, addend, y, n, h, w, s);  // fp32 rows would be 256 bytes
    return bf ? launch_to3<bf16_t, 3, 64, 32>(x, wq, addend, y, n, h, w, s) : launch_to3<float, 3, 128, 32>(x, wq, addend, y, n, h, w, s);
}

// The 9 x 9 layer absorbing the norm + ReLU in front of it (bf16, 32 input channels): see To3Clamp.
int conv_to3_clamp_impl(const void* x, const void* wq_samples, long w_stride, const void* clamp_t, const float* bias_rows,
                        void* y, int n, int h, int w, int ci, int k, int dtype, hipStream_t s) {
    STMI_REQUIRE(dtype == STMI_BF16 && k == 9 && ci == 32, "conv_to3 (absorbing a norm): bf16, 9 x 9, 32 input channels only");
    STMI_REQUIRE((((uintptr_t)x | (uintptr_t)wq_samples | (uintptr_t)clamp_t) & 15) == 0 && w_stride % 8 == 0,
                 "conv_to3: x, the weights and the clamp values must be 16-byte aligned");
    return launch_to3<bf16_t, 9, 64, 40, true>(x, wq_samples, nullptr, y, n, h, w, s, To3Clamp{clamp_t, bias_rows, w_stride});
}

}  // namespace stmi
