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
// The K loop is an NS-deep ring of LDS buffers.  At these problem sizes a CU holds one or
// two blocks, so throughput is (bytes in flight) / (DMA latency): NS-1 steps stay in flight
// behind a COUNTED s_waitcnt vmcnt and one raw s_barrier per step (a __syncthreads() would
// drain the ring with vmcnt(0)).
#include "conv_gemm.hpp"

#include <stdlib.h>

namespace stmi {

__device__ uint4 g_zero_page[8];

template <typename T>
__device__ __forceinline__ void glds16(const T* src, uint4* lds_dst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ds_read_b128 the compiler does not see as an LDS access; the value is valid only after
// lds_wait() followed by lds_landed() on the same register.
__device__ __forceinline__ u32x4 lds_read16(unsigned addr, int imm) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_landed(u32x4& v) { asm volatile("" : "+v"(v)); }

// all waves' DMA for the step has landed (each waited on its own counter) and every wave
// is done reading the buffer that is refilled next
__device__ __forceinline__ void ring_barrier() {
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <typename T, int TC, int TP, int WC, int WP, int NS>
__global__ __launch_bounds__(256) void conv_gemm_glds_kernel(const ConvGemmParams p) {
    static_assert(WC * WP == 4, "4 waves per block");
    static_assert(NS >= 2 && NS <= 4, "ring depth");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 8 * EPC;  // K elements per step: 128 bytes per row
    constexpr int BLK_CO = 16 * TC * WC, BLK_PIX = 16 * TP * WP;
    constexpr int NI_W = BLK_CO / 32, NI_P = BLK_PIX / 32;  // DMA instructions per wave and step
    constexpr int NI = NI_W + NI_P;
    constexpr int STAGE = (BLK_CO + BLK_PIX) * 8;  // uint4 per ring slot
    static_assert(BLK_CO % 32 == 0 && BLK_PIX % 32 == 0, "tile rows must split into 8-row DMA pieces over 4 waves");
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(16))) uint4 lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wpx = wave % WP;
    const int g = lane >> 4, r16 = lane & 15;
    const int sub = lane >> 3, pos = lane & 7;
    const int HWs = p.Hs * p.Ws;
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)HWs : (long)p.N * HWs;
    // Blocks b and b+8 run on the same XCD: hand each XCD one contiguous run of tiles
    // (channel tile fastest, then pixels, then sample), so that the pixel rows it gathers
    // and their halos stay inside that XCD's own 4 MiB L2.
    unsigned tile;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    const unsigned n_co = p.co_pad / BLK_CO;
    const unsigned n_pix = (unsigned)((Mz + BLK_PIX - 1) / BLK_PIX);
    const int co0 = (tile % n_co) * BLK_CO;
    const unsigned tile_p = tile / n_co;
    const long pix0 = (long)(tile_p % n_pix) * BLK_PIX;
    const int z = tile_p / n_pix;
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

    const int ksteps = p.klen / BK;
    const int nsteps = p.ntaps * ksteps;
    int st_t = 0, st_k = 0;  // (tap, k-chunk) of the next step to stage

    auto stage_next = [&](int slot) {
        uint4* buf = lds + slot * STAGE;
        const Tap tap = p.taps[st_t];
        const int k0 = st_k * BK;
        const T* wsrc = wp + ((long)tap.widx * p.co_pad + co0) * p.kpad + k0;
#pragma unroll
        for (int j = 0; j < NI_W; j++) glds16(wsrc + wrow_off[j], buf + (wave * NI_W + j) * 64);
        const long tdelta = ((long)tap.dy * p.Wi + tap.dx) * p.Ci + k0;
#pragma unroll
        for (int j = 0; j < NI_P; j++) {
            const int iy = py[j] + tap.dy, ix = px[j] + tap.dx;
            const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
            glds16(ok ? in + pbase[j] + tdelta : zero_src, buf + (BLK_CO / 8 + wave * NI_P + j) * 64);
        }
        if (++st_k == ksteps) { st_k = 0; ++st_t; }
    };

    // Fragment reads go through inline asm: a compiler-visible LDS load after an LDS-DMA
    // write that it cannot prove disjoint is preceded by s_waitcnt vmcnt(0), which would
    // drain the ring every step.  Only lgkmcnt(0) waits are used (scalar loads share the
    // counter and return out of order).
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned frag_off = r16 * 128 + ((g ^ ((r16 >> 1) & 7)) << 4);  // kk = 1 flips chunk bit 2
    const unsigned a_addr0 = lds_base + wc * TC * 2048 + frag_off;
    const unsigned a_addr1 = lds_base + wc * TC * 2048 + (frag_off ^ 64);
    const unsigned b_addr0 = lds_base + BLK_CO * 128 + wpx * TP * 2048 + frag_off;
    const unsigned b_addr1 = lds_base + BLK_CO * 128 + wpx * TP * 2048 + (frag_off ^ 64);

    auto compute = [&](int slot) {
        const unsigned so = slot * (STAGE * 16);
        u32x4 a0[TC], b0[TP], a1[TC], b1[TP];
#pragma unroll
        for (int tc = 0; tc < TC; tc++) a0[tc] = lds_read16(a_addr0 + so, tc * 2048);
#pragma unroll
        for (int tp = 0; tp < TP; tp++) b0[tp] = lds_read16(b_addr0 + so, tp * 2048);
        lds_wait();
#pragma unroll
        for (int tc = 0; tc < TC; tc++) lds_landed(a0[tc]);
#pragma unroll
        for (int tp = 0; tp < TP; tp++) lds_landed(b0[tp]);
#pragma unroll
        for (int tc = 0; tc < TC; tc++) a1[tc] = lds_read16(a_addr1 + so, tc * 2048);
#pragma unroll
        for (int tp = 0; tp < TP; tp++) b1[tp] = lds_read16(b_addr1 + so, tp * 2048);
#pragma unroll
        for (int tc = 0; tc < TC; tc++)
#pragma unroll
            for (int tp = 0; tp < TP; tp++)
                Mma<T>::mma(__builtin_bit_cast(frag, a0[tc]), __builtin_bit_cast(frag, b0[tp]), acc[tc][tp]);
        lds_wait();
#pragma unroll
        for (int tc = 0; tc < TC; tc++) lds_landed(a1[tc]);
#pragma unroll
        for (int tp = 0; tp < TP; tp++) lds_landed(b1[tp]);
#pragma unroll
        for (int tc = 0; tc < TC; tc++)
#pragma unroll
            for (int tp = 0; tp < TP; tp++)
                Mma<T>::mma(__builtin_bit_cast(frag, a1[tc]), __builtin_bit_cast(frag, b1[tp]), acc[tc][tp]);
    };

    if (nsteps > 0) {
        // prologue: NS-1 steps in flight
#pragma unroll
        for (int s = 0; s < NS - 1; s++)
            if (s < nsteps) stage_next(s);
        int slot = 0;
        for (int s = 0; s < nsteps; s++) {
            // wait for step s: everything issued after it may stay in flight
            const int later = nsteps - 1 - s;  // steps issued after s so far: min(later, NS-2)
            if (NS >= 4 && later >= 2) wait_vmcnt<2 * NI>();
            else if (NS >= 3 && later >= 1) wait_vmcnt<NI>();
            else wait_vmcnt<0>();
            ring_barrier();
            if (s + NS - 1 < nsteps) stage_next(slot == 0 ? NS - 1 : slot - 1);  // refill the slot step s-1 used
            compute(slot);
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
    }
    conv_epilogue_linear<T, TC, TP, WC, WP, NS * STAGE * 16>(p, acc, co0, pix0, Mz, zoff, HWs, lds);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
struct GldsCfg {
    int blk_co, blk_pix;
    const char* name;
};
static const GldsCfg kCfgs[] = {
    {128, 128, "4, 4, 2, 2, 3"},  // 0
    {128, 64, "4, 2, 2, 2, 3"},   // 1
    {64, 128, "2, 4, 2, 2, 3"},   // 2
    {64, 64, "2, 2, 2, 2, 4"},    // 3
    {32, 128, "2, 2, 1, 4, 3"},   // 4
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
    const int c = STMI_ENV_INT("STMI_CONV_CFG", -1);
    if (c >= 0) {
        if (c >= 0 && c < 5 && p.co_pad % kCfgs[c].blk_co == 0) return c;
    }
    if (p.co_pad % 64 != 0) return 4;
    const bool wide = p.co_pad % 128 == 0;
    if (wide && tiles_of(p, 0) >= 512) return 0;
    if (wide && tiles_of(p, 1) >= 256) return 1;
    if (tiles_of(p, 2) >= 512) return 2;
    return 3;
}

template <typename T, int TC, int TP, int WC, int WP, int NS>
static int launch_one(const ConvGemmParams& p, hipStream_t s) {
    constexpr int BLK_CO = 16 * TC * WC, BLK_PIX = 16 * TP * WP;
    constexpr size_t kLds = (size_t)NS * (BLK_CO + BLK_PIX) * 128;
    STMI_DYN_LDS((conv_gemm_glds_kernel<T, TC, TP, WC, WP, NS>), (int)kLds, "hipFuncSetAttribute(conv_gemm_glds_kernel)");
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)p.Hs * p.Ws : (long)p.N * p.Hs * p.Ws;
    if (Mz <= 0) return STMI_OK;
    const long tiles = (long)ceil_div(Mz, BLK_PIX) * (p.co_pad / BLK_CO) * (batched ? p.N : 1);
    STMI_REQUIRE(tiles < (1L << 31), "conv: too many tiles for one launch");
    dim3 grid((unsigned)tiles);
    hipLaunchKernelGGL((conv_gemm_glds_kernel<T, TC, TP, WC, WP, NS>), grid, dim3(256), kLds, s, p);
    STMI_LAUNCH_CHECK("conv_gemm_glds_kernel");
    return STMI_OK;
}

template <typename T>
static int launch_typed(const ConvGemmParams& p, hipStream_t s) {
    switch (pick(p)) {
        case 0: return launch_one<T, 4, 4, 2, 2, 3>(p, s);
        case 1: return launch_one<T, 4, 2, 2, 2, 3>(p, s);
        case 2: return launch_one<T, 2, 4, 2, 2, 3>(p, s);
        case 3: return launch_one<T, 2, 2, 2, 2, 4>(p, s);
        default: return launch_one<T, 2, 2, 1, 4, 3>(p, s);
    }
}

bool try_launch_conv_gemm_glds(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc) {
    if (!eligible(p, dtype) || STMI_ENV_SET("STMI_NO_GLDS")) return false;
    *rc = dtype == STMI_BF16 ? launch_typed<bf16_t>(p, s) : launch_typed<float>(p, s);
    return true;
}

const char* conv_gemm_glds_cfg_name(const ConvGemmParams& p, int dtype) {
    return eligible(p, dtype) && !STMI_ENV_SET("STMI_NO_GLDS") ? kCfgs[pick(p)].name : nullptr;
}

}  // namespace stmi
