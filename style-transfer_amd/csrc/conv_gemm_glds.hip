// Gather-GEMM convolution, LDS-DMA variant (gfx950).
//
// Same contraction as conv_gemm.hip, for the heavy layers: K per tap a multiple of 128
// bytes (>= 64 bf16 / 32 fp32 input channels) and per-pixel validity.  Both operand tiles
// go global -> LDS with global_load_lds_dwordx4 (no VGPR round trip, no ds_write pass):
//   * one wave-instruction moves 8 rows x 128 B; lane l fills chunk position l&7 of row
//     l>>3, linearly in LDS (the hardware writes base + lane*16);
//   * the bank swizzle therefore sits on the SOURCE side: position p of row r is loaded
//     from chunk p ^ ((r>>1)&7), and fragment reads apply the same XOR, which makes the
//     16-row x 4-chunk ds_read_b128 pattern conflict-free on 128-byte rows;
//   * rows that fall outside the image are fetched from a 128-byte page of zeros.
// Two LDS buffers; the DMA for step s+1 is in flight while step s is multiplied.
#include "conv_gemm.hpp"

#include <stdlib.h>

namespace stmi {

__device__ uint4 g_zero_page[8];

template <typename T>
__device__ __forceinline__ void glds16(const T* src, uint4* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <typename T, int TC, int TP, int WC, int WP>
__global__ __launch_bounds__(256) void conv_gemm_glds_kernel(const ConvGemmParams p) {
    static_assert(WC * WP == 4, "4 waves per block");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 8 * EPC;  // K elements per step: 128 bytes per row
    constexpr int BLK_CO = 16 * TC * WC, BLK_PIX = 16 * TP * WP;
    constexpr int NI_W = BLK_CO / 32, NI_P = BLK_PIX / 32;  // DMA instructions per wave and step
    static_assert(BLK_CO % 32 == 0 && BLK_PIX % 32 == 0, "tile rows must split into 8-row DMA pieces over 4 waves");
    typedef typename Mma<T>::frag frag;

    __shared__ uint4 lds[2][(BLK_CO + BLK_PIX) * 8];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wpx = wave % WP;
    const int g = lane >> 4, r16 = lane & 15;
    const int sub = lane >> 3, pos = lane & 7;
    const int co0 = blockIdx.y * BLK_CO;
    const long pix0 = (long)blockIdx.x * BLK_PIX;
    const int z = blockIdx.z;
    const int HWs = p.Hs * p.Ws;
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)HWs : (long)p.N * HWs;
    const long zoff = batched ? (long)z * HWs : 0;
    const T* in = (const T*)p.in;
    const T* wp = (const T*)p.wp + (long)z * p.w_batch_stride;

    // what this lane fetches in every step
    int wrow_off[NI_W];
#pragma unroll
    for (int j = 0; j < NI_W; j++) {
        const int row = (wave * NI_W + j) * 8 + sub;
        wrow_off[j] = row * p.kpad + (pos ^ ((row >> 1) & 7)) * EPC;
    }
    int py[NI_P], px[NI_P];
    long pbase[NI_P];
#pragma unroll
    for (int j = 0; j < NI_P; j++) {
        const int row = (wave * NI_P + j) * 8 + sub;
        const long m = pix0 + row;
        if (m < Mz) {
            const long mm = m + zoff;
            const int n = (int)(mm / HWs);
            const int rem = (int)(mm - (long)n * HWs);
            const int i = rem / p.Ws;
            py[j] = i * p.in_stride;
            px[j] = (rem - i * p.Ws) * p.in_stride;
            pbase[j] = (((long)n * p.Hi + py[j]) * p.Wi + px[j]) * p.Ci + (pos ^ ((row >> 1) & 7)) * EPC;
        } else {
            py[j] = -(1 << 20);
            px[j] = 0;
            pbase[j] = 0;
        }
    }
    const T* zero_src = (const T*)(g_zero_page + pos);

    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage = [&](int buf, int t, int k0) {
        const Tap tap = p.taps[t];
        const T* wsrc = wp + ((long)tap.widx * p.co_pad + co0) * p.kpad + k0;
#pragma unroll
        for (int j = 0; j < NI_W; j++) glds16(wsrc + wrow_off[j], &lds[buf][(wave * NI_W + j) * 64]);
        const long tdelta = ((long)tap.dy * p.Wi + tap.dx) * p.Ci + k0;
#pragma unroll
        for (int j = 0; j < NI_P; j++) {
            const int iy = py[j] + tap.dy, ix = px[j] + tap.dx;
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            glds16(ok ? in + pbase[j] + tdelta : zero_src, &lds[buf][(BLK_CO / 8 + wave * NI_P + j) * 64]);
        }
    };

    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            frag a[TC], b[TP];
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
                const int row = (wc * TC + tc) * 16 + r16;
                const uint4 raw = lds[buf][row * 8 + ((4 * kk + g) ^ ((row >> 1) & 7))];
                a[tc] = *reinterpret_cast<const frag*>(&raw);
            }
#pragma unroll
            for (int tp = 0; tp < TP; tp++) {
                const int row = (wpx * TP + tp) * 16 + r16;
                const uint4 raw = lds[buf][(BLK_CO + row) * 8 + ((4 * kk + g) ^ ((row >> 1) & 7))];
                b[tp] = *reinterpret_cast<const frag*>(&raw);
            }
#pragma unroll
            for (int tc = 0; tc < TC; tc++)
#pragma unroll
                for (int tp = 0; tp < TP; tp++) Mma<T>::mma(a[tc], b[tp], acc[tc][tp]);
        }
    };

    const int ksteps = p.klen / BK;
    const int nsteps = p.ntaps * ksteps;
    if (nsteps > 0) {
        int t = 0, ks = 0;
        stage(0, 0, 0);
        __syncthreads();  // drains the DMA (vmcnt) before the barrier
        for (int s = 0; s < nsteps; s++) {
            const bool more = s + 1 < nsteps;
            if (more) {
                if (++ks == ksteps) { ks = 0; ++t; }
                stage((s + 1) & 1, t, ks * BK);
            }
            compute(s & 1);
            __syncthreads();
        }
    }
    conv_epilogue<T, TC, TP>(p, acc, co0, pix0, wc, wpx, g, r16, Mz, zoff, HWs);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct GldsCfg {
    int blk_co, blk_pix;
    const char* name;
};
static const GldsCfg kCfgs[] = {
    {128, 128, "4, 4, 2, 2"},  // 0
    {128, 64, "4, 2, 2, 2"},   // 1
    {64, 128, "2, 4, 2, 2"},   // 2
    {64, 64, "2, 2, 2, 2"},    // 3
    {32, 128, "2, 2, 1, 4"},   // 4
};

static long tiles_of(const ConvGemmParams& p, int c) {
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)p.Hs * p.Ws : (long)p.N * p.Hs * p.Ws;
    return (long)ceil_div(Mz, kCfgs[c].blk_pix) * (p.co_pad / kCfgs[c].blk_co) * (batched ? p.N : 1);
}

static bool eligible(const ConvGemmParams& p, int dtype) {
    const int bk = dtype == STMI_BF16 ? 64 : 32;
    return p.cdiv == 0 && p.klen > 0 && p.klen % bk == 0 && p.co_pad % 32 == 0 && p.ntaps > 0;
}

// Picks the tile: big tiles when they still give every CU two blocks, otherwise smaller
// ones.  STMI_CONV_CFG=<index> forces a configuration (tuning aid).
static int pick(const ConvGemmParams& p) {
    const char* force = getenv("STMI_CONV_CFG");
    if (force) {
        const int c = atoi(force);
        if (c >= 0 && c < 5 && p.co_pad % kCfgs[c].blk_co == 0) return c;
    }
    if (p.co_pad % 64 != 0) return 4;
    const bool wide = p.co_pad % 128 == 0;
    if (wide && tiles_of(p, 0) >= 512) return 0;
    if (wide && tiles_of(p, 1) >= 256) return 1;
    if (tiles_of(p, 2) >= 512) return 2;
    return 3;
}

template <typename T, int TC, int TP, int WC, int WP>
static int launch_one(const ConvGemmParams& p, hipStream_t s) {
    constexpr int BLK_CO = 16 * TC * WC, BLK_PIX = 16 * TP * WP;
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)p.Hs * p.Ws : (long)p.N * p.Hs * p.Ws;
    if (Mz <= 0) return STMI_OK;
    dim3 grid((unsigned)ceil_div(Mz, BLK_PIX), (unsigned)(p.co_pad / BLK_CO), (unsigned)(batched ? p.N : 1));
    hipLaunchKernelGGL((conv_gemm_glds_kernel<T, TC, TP, WC, WP>), grid, dim3(256), 0, s, p);
    STMI_LAUNCH_CHECK("conv_gemm_glds_kernel");
    return STMI_OK;
}

template <typename T>
static int launch_typed(const ConvGemmParams& p, hipStream_t s) {
    switch (pick(p)) {
        case 0: return launch_one<T, 4, 4, 2, 2>(p, s);
        case 1: return launch_one<T, 4, 2, 2, 2>(p, s);
        case 2: return launch_one<T, 2, 4, 2, 2>(p, s);
        case 3: return launch_one<T, 2, 2, 2, 2>(p, s);
        default: return launch_one<T, 2, 2, 1, 4>(p, s);
    }
}

bool try_launch_conv_gemm_glds(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc) {
    if (!eligible(p, dtype) || getenv("STMI_NO_GLDS")) return false;
    *rc = dtype == STMI_BF16 ? launch_typed<bf16_t>(p, s) : launch_typed<float>(p, s);
    return true;
}

const char* conv_gemm_glds_cfg_name(const ConvGemmParams& p, int dtype) {
    return eligible(p, dtype) && !getenv("STMI_NO_GLDS") ? kCfgs[pick(p)].name : nullptr;
}

}  // namespace stmi
