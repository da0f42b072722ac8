// Gather-GEMM convolution with producer / consumer waves (gfx950).
//
// Same contraction as conv_gemm.hip, D[co][pixel] = sum_tap sum_k Wp[tap][co][k] * In[pixel @ tap][k],
// for every layer whose K per tap is a multiple of 64 bytes: stride-2 and transposed
// convolutions (one launch per output-parity class), Gram backward (per-sample weights) and
// the 32-channel row-patch layers.  A K step is one 128- or 64-byte row piece per tile row;
// both operand tiles go global -> LDS with global_load_lds_dwordx4 through a 4-deep ring.
//
// This contraction re-fetches every pixel row once per tap, so it is bound by the fill path
// (about 24 KB per MFLOP), and a CU's fill rate scales with the number of waves that ISSUE
// fills (an issue stalls its wave for 60-180 cycles).  Hence eight producer waves that only
// issue DMA and keep counted s_waitcnt vmcnt(N) books, and four consumer waves (2 x 2) that
// only read fragments and issue MFMAs; one s_barrier per step hands a filled slot over and a
// drained one back.
//
// The DMA writes LDS linearly, so the bank swizzle sits on the SOURCE side: position p of
// tile row r holds chunk p ^ f(r), f(r) = (r >> 1) & 7 for 128-byte rows and ((r >> 3) & 1) << 1
// for 64-byte rows (both make the 16-row ds_read_b128 fragment pattern conflict-free); rows
// outside the image come from a page of zeros.
#include "conv_gemm.hpp"

#include <stdlib.h>

#include <type_traits>

namespace stmi {

__device__ uint4 g_pc_zero_page[8];  // zero-initialised

namespace {

#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"  // host pass only: LDS pointers are 32-bit on the device
template <typename T>
__device__ __forceinline__ void glds16(const T* src, unsigned lds_byte_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_byte_addr, 16, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int RB>
__device__ __forceinline__ int row_swz(int r) {
    return RB == 128 ? (r >> 1) & 7 : ((r >> 3) & 1) << 1;
}

constexpr int kRing = 4;
constexpr int kProducers = 8;
constexpr int kThreads = 64 * (4 + kProducers);

}  // namespace

// Block tile: 32*TC output channels x 32*TP pixels; RB bytes of K per row and step; RING slots of operand
// tiles (4: three steps prefetched; 2: one, for launches of a few steps only, where the LDS goes into a wider
// pixel tile instead: fewer blocks, each with the same fixed start and epilogue cost).
template <typename T, int TC, int TP, int RB, int RING>
__global__ __launch_bounds__(kThreads) void conv_gemm_pc_kernel(const ConvGemmParams p) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = RB / (int)sizeof(T);  // K elements per step
    constexpr int CPR = RB / 16;             // 16-byte chunks per row
    constexpr int RPP = 1024 / RB;           // rows per DMA piece
    constexpr int BLK_CO = 32 * TC, BLK_PIX = 32 * TP;
    constexpr int W_PIECES = BLK_CO / RPP, P_PIECES = BLK_PIX / RPP;
    constexpr int NWX = (W_PIECES + kProducers - 1) / kProducers;  // weight pieces of the waves that have any
    constexpr int NPX = P_PIECES / kProducers;                     // pixel pieces per producer wave
    constexpr int W_BYTES = BLK_CO * RB, SLOT = (BLK_CO + BLK_PIX) * RB;
    static_assert(P_PIECES % kProducers == 0 && NPX >= 1, "pixel pieces must split over the producer waves");
    static_assert(W_PIECES % NWX == 0, "weight pieces must split evenly over the waves that fetch them");
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(128))) char lds[];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Blocks b and b+8 run on the same XCD: hand each XCD one contiguous run of tiles
    // (channel tile fastest, then pixels, then sample), so that the pixel rows it gathers
    // stay inside that XCD's own 4 MiB L2.
    unsigned tile;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    // sub-grid of this block (one launch may cover several, see ConvGemmParams::SubGrid)
    int Hs = p.Hs, Ws = p.Ws, oy0 = p.oy0, ox0 = p.ox0, ntaps = p.ntaps, tap0 = 0;
    if (p.nclasses > 0) {
        int c = 0;
        while (c + 1 < p.nclasses && tile >= p.cls[c].tile_end) c++;
        Hs = p.cls[c].Hs; Ws = p.cls[c].Ws; oy0 = p.cls[c].oy0; ox0 = p.cls[c].ox0;
        ntaps = p.cls[c].ntaps; tap0 = p.cls[c].tap0;
        if (c > 0) tile -= p.cls[c - 1].tile_end;
    }
    const int HWs = Hs * Ws;
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)HWs : (long)p.N * HWs;
    const unsigned n_co = p.co_pad / BLK_CO;
    const unsigned n_pix = (unsigned)((Mz + BLK_PIX - 1) / BLK_PIX);
    const int co0 = (tile % n_co) * BLK_CO;
    const unsigned tile_p = tile / n_co;
    const long pix0 = (long)(tile_p % n_pix) * BLK_PIX;
    const int z = tile_p / n_pix;
    const long zoff = batched ? (long)z * HWs : 0;
    const int ksteps = p.klen / BK;
    const int nsteps = ntaps * ksteps;

    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4;
        const int sub = lane / CPR, pos = lane % CPR;
        const T* in = (const T*)p.in;
        const T* wp = (const T*)p.wp + (long)z * p.w_batch_stride;
        const T* zero_src = (const T*)g_pc_zero_page;
        // pixel rows this lane fetches in every step
        int py[NPX], px[NPX];
        long pbase[NPX];
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const int row = (pw * NPX + j) * RPP + sub;
            const long m = pix0 + row;
            if (m < Mz) {
                const long mm = m + zoff;
                const int n = (int)(mm / HWs);
                const int rem = (int)(mm - (long)n * HWs);
                const int i = rem / Ws;
                py[j] = i * p.in_stride;
                px[j] = (rem - i * Ws) * p.in_stride;
                pbase[j] = (((long)n * p.Hi + py[j]) * p.Wi + px[j]) * p.Ci + (pos ^ row_swz<RB>(row)) * EPC;
            } else {
                py[j] = -(1 << 20);
                px[j] = 0;
                pbase[j] = 0;
            }
        }
        // weight rows (only the first W_PIECES / NWX producer waves have any)
        int wrow_off[NWX];
#pragma unroll
        for (int j = 0; j < NWX; j++) {
            const int row = (pw * NWX + j) * RPP + sub;
            wrow_off[j] = row * p.kpad + (pos ^ row_swz<RB>(row)) * EPC;
        }
        int st_t = 0, st_k = 0;  // (tap, k-chunk) of the next step to stage
        auto run = [&](auto has_w_tag) {
            constexpr bool HAS_W = decltype(has_w_tag)::value;
            constexpr int NI = NPX + (HAS_W ? NWX : 0);
            auto stage_next = [&](int slot) {
                const unsigned base = lds_base + slot * SLOT;
                const Tap tap = p.taps[tap0 + st_t];
                const int k0 = st_k * BK;
                if (HAS_W) {
                    const T* wsrc = wp + ((long)tap.widx * p.co_pad + co0) * p.kpad + k0;
#pragma unroll
                    for (int j = 0; j < NWX; j++) glds16(wsrc + wrow_off[j], base + (pw * NWX + j) * 1024);
                }
                const long tdelta = ((long)tap.dy * p.Wi + tap.dx) * p.Ci + k0;
#pragma unroll
                for (int j = 0; j < NPX; j++) {
                    const int iy = py[j] + tap.dy, ix = px[j] + tap.dx;
                    const bool ok = (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
                    glds16(ok ? in + pbase[j] + tdelta : zero_src, base + W_BYTES + (pw * NPX + j) * 1024);
                }
                if (++st_k == ksteps) { st_k = 0; ++st_t; }
            };
#pragma unroll
            for (int s = 0; s < RING - 1; s++)
                if (s < nsteps) stage_next(s);
            for (int s = 0; s < nsteps; s++) {
                // steps issued after s so far: min(nsteps - 1 - s, RING - 2)
                if constexpr (RING == 4) {
                    const int later = nsteps - 1 - s;
                    if (later >= 2) wait_vmcnt<2 * NI>();
                    else if (later >= 1) wait_vmcnt<NI>();
                    else wait_vmcnt<0>();
                } else {
                    static_assert(RING == 2, "ring depth");
                    wait_vmcnt<0>();
                }
                __builtin_amdgcn_s_barrier();  // slot of step s is filled; slot of step s-1 is drained
                asm volatile("" ::: "memory");
                if (s + RING - 1 < nsteps) stage_next((s + RING - 1) % RING);
            }
        };
        if (pw * NWX < W_PIECES) run(std::true_type{});
        else run(std::false_type{});
    } else {
        // ------------------------------------------------------------------ consumers
        const int wc = wave >> 1, wpx = wave & 1;
        const int g = lane >> 4, r16 = lane & 15;
        for (int s = 0; s < nsteps; s++) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* ws = lds + (s % RING) * SLOT;
            const char* ps = ws + W_BYTES;
#pragma unroll
            for (int kk = 0; kk < RB / 64; kk++) {
                frag a[TC], b[TP];
#pragma unroll
                for (int tc = 0; tc < TC; tc++) {
                    const int row = (wc * TC + tc) * 16 + r16;
                    a[tc] = *reinterpret_cast<const frag*>(ws + row * RB + (((4 * kk + g) ^ row_swz<RB>(row)) << 4));
                }
#pragma unroll
                for (int tp = 0; tp < TP; tp++) {
                    const int row = (wpx * TP + tp) * 16 + r16;
                    b[tp] = *reinterpret_cast<const frag*>(ps + row * RB + (((4 * kk + g) ^ row_swz<RB>(row)) << 4));
                }
#pragma unroll
                for (int tc = 0; tc < TC; tc++)
#pragma unroll
                    for (int tp = 0; tp < TP; tp++) Mma<T>::mma(a[tc], b[tp], acc[tc][tp]);
            }
        }
    }

    // epilogue: row-wise through LDS when the tensors allow it, lane-wise otherwise
    constexpr int TPG = conv_stage_tpg(TC, TP, 2, 2, RING * SLOT);
    static_assert(TPG >= 1, "staging area larger than the ring");
    if (p.row_store_ok) {
        conv_epilogue_lds<T, TC, TP, 2, 2, kThreads, TPG>(p, acc, co0, reinterpret_cast<float*>(lds), [&](int pl) -> long {
            const long m = pix0 + pl;
            if (m >= Mz) return -1;
            const long mm = m + zoff;
            const int n = (int)(mm / HWs);
            const int rem = (int)(mm - (long)n * HWs);
            const int i = rem / Ws, j = rem - i * Ws;
            return (((long)n * p.Ho + (oy0 + p.oys * i)) * p.Wo + (ox0 + p.oxs * j)) * p.Co;
        });
    } else if (wave < 4) {  // single sub-grid only (the launcher merges sub-grids only for row-wise stores)
        conv_epilogue<T, TC, TP>(p, acc, co0, pix0, wave >> 1, wave & 1, lane >> 4, lane & 15, Mz, zoff, HWs);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static int pc_row_bytes(const ConvGemmParams& p, int dtype) {
    const int es = dtype == STMI_BF16 ? 2 : 4;
    if (p.cdiv != 0 || p.klen <= 0 || p.ntaps <= 0) return 0;
    if (p.klen * es % 128 == 0) return 128;
    if (p.klen * es % 64 == 0) return 64;
    return 0;
}

static long pc_tiles(const ConvGemmParams& p, int blk_co, int blk_pix) {
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)p.Hs * p.Ws : (long)p.N * p.Hs * p.Ws;
    return (long)ceil_div(Mz, blk_pix) * (p.co_pad / blk_co) * (batched ? p.N : 1);
}

// Fills in the tile ranges of the sub-grids of a merged launch; returns the total.
static long pc_assign_tiles(ConvGemmParams& p, int blk_co, int blk_pix) {
    long total = 0;
    for (int c = 0; c < p.nclasses; c++) {
        total += (long)ceil_div((long)p.N * p.cls[c].Hs * p.cls[c].Ws, blk_pix) * (p.co_pad / blk_co);
        p.cls[c].tile_end = (unsigned)total;
    }
    return total;
}

template <typename T, int TC, int TP, int RB, int RING = kRing>
static int launch_pc(const ConvGemmParams& p_in, hipStream_t s, char* name, int name_len, const char* tname) {
    ConvGemmParams p = p_in;
    constexpr size_t kLds = (size_t)RING * (32 * TC + 32 * TP) * RB;
    STMI_DYN_LDS((conv_gemm_pc_kernel<T, TC, TP, RB, RING>), (int)kLds, "hipFuncSetAttribute(conv_gemm_pc_kernel)");
    if (name)
        snprintf(name, name_len, "void stmi::conv_gemm_pc_kernel<%s, %d, %d, %d, %d>(stmi::ConvGemmParams)", tname, TC, TP, RB, RING);
    const long tiles = p.nclasses > 0 ? pc_assign_tiles(p, 32 * TC, 32 * TP) : pc_tiles(p, 32 * TC, 32 * TP);
    if (tiles <= 0) return STMI_OK;
    STMI_REQUIRE(tiles < (1L << 31), "conv: too many tiles for one launch");
    hipLaunchKernelGGL((conv_gemm_pc_kernel<T, TC, TP, RB, RING>), dim3((unsigned)tiles), dim3(kThreads), kLds, s, p);
    STMI_LAUNCH_CHECK("conv_gemm_pc_kernel");
    return STMI_OK;
}

// Tile choice: 128 x 64 when that still gives the chip most of a block per CU, else 64 x 64;
// 32-row weight sets take 32 x 128; 64-byte rows take 128 or 256 pixels by the number of K steps.  (128 x 128 does not fit the 168-register budget of a
// 768-thread block without spilling.)  STMI_PC_CFG=1|2 forces a configuration (tuning aid).
template <typename T>
static int launch_pc_typed(const ConvGemmParams& p, int rb, hipStream_t s, char* name, int name_len, const char* tname) {
    if (rb == 64) {
        // 256-pixel tiles once a block has enough K steps to amortise its start and its epilogue
        // (the 9-tap layers); the 3-step conv1_1 row form is bound by its stores and keeps 128
        const bool deep = p.ntaps * (p.klen / (64 / (int)sizeof(T))) >= 6;
        if (p.co_pad % 64 == 0)
            return deep ? launch_pc<T, 2, 8, 64>(p, s, name, name_len, tname) : launch_pc<T, 2, 4, 64>(p, s, name, name_len, tname);
        return deep ? launch_pc<T, 1, 8, 64>(p, s, name, name_len, tname) : launch_pc<T, 1, 4, 64>(p, s, name, name_len, tname);
    }
    // launches of at most eight K steps (the parity classes of the strided layers): a 2-slot ring and twice
    // the pixels per block.  STMI_PC_WIDE=0 switches that off (tuning aid).
    const bool wide = STMI_ENV_INT("STMI_PC_WIDE", 1) != 0;
    const bool shallow = wide && p.ntaps * (p.klen / (128 / (int)sizeof(T))) <= 8;
    if (p.co_pad % 64 != 0)
        return shallow ? launch_pc<T, 1, 8, 128, 2>(p, s, name, name_len, tname) : launch_pc<T, 1, 4, 128>(p, s, name, name_len, tname);
    if (shallow && p.co_pad % 128 != 0) return launch_pc<T, 2, 4, 128, 2>(p, s, name, name_len, tname);
    int cfg = STMI_ENV_INT("STMI_PC_CFG", 0);
    // Launches of many block rounds and few K steps (the 64 -> 128 stride-2 layer of a 1080p batch: 126 rounds of
    // nine steps) take the 2-slot ring as well: 48 KB of LDS lets two blocks share a CU, so one block's start and
    // epilogue run under the other's steps (885 -> about 750 us per launch; a training batch, one round, is unmoved).
    if (cfg == 3 || (cfg == 0 && wide && p.ntaps * (p.klen / (128 / (int)sizeof(T))) <= 9 && pc_tiles(p, 64, 128) >= 4 * 256))
        return launch_pc<T, 2, 4, 128, 2>(p, s, name, name_len, tname);
    if (cfg != 1 && cfg != 2) cfg = p.co_pad % 128 == 0 && pc_tiles(p, 128, 64) >= 192 ? 1 : 2;
    if (cfg == 1 && (p.co_pad % 128 != 0 || sizeof(T) == 4)) cfg = 2;  // the fp32 128 x 64 tile spills registers
    if (cfg == 1) return launch_pc<T, 4, 2, 128>(p, s, name, name_len, tname);
    return launch_pc<T, 2, 2, 128>(p, s, name, name_len, tname);
}

// Several sub-grids (p.nclasses > 0, p.cls[] filled in but for tile_end; p.Hs, p.Ws, p.ntaps describing the
// largest of them, for the tile choice) in one launch.  Needs row-wise stores and one weight set.
bool try_launch_conv_gemm_pc_multi(ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len) {
    const int rb = pc_row_bytes(p, dtype);
    if (rb == 0 || p.co_pad % 32 != 0 || STMI_ENV_SET("STMI_NO_PC") || STMI_ENV_SET("STMI_NO_PC_MULTI")) return false;
    if ((((uintptr_t)p.in | (uintptr_t)p.wp) & 15) != 0 || p.Ci % (dtype == STMI_BF16 ? 8 : 4) != 0) return false;
    if (!p.row_store_ok || p.w_batch_stride != 0 || p.nclasses < 1 || p.nclasses > 4) return false;
    *rc = dtype == STMI_BF16 ? launch_pc_typed<bf16_t>(p, rb, s, name, name_len, "__bf16")
                             : launch_pc_typed<float>(p, rb, s, name, name_len, "float");
    return true;
}

bool try_launch_conv_gemm_pc(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len) {
    const int rb = pc_row_bytes(p, dtype);
    if (rb == 0 || p.co_pad % 32 != 0 || STMI_ENV_SET("STMI_NO_PC")) return false;
    if ((((uintptr_t)p.in | (uintptr_t)p.wp) & 15) != 0 || p.Ci % (dtype == STMI_BF16 ? 8 : 4) != 0) return false;
    {
        // A launch of fewer than five K steps is all prologue and epilogue: eight producer waves and a 4-deep
        // ring bring nothing there, and the smaller blocks of the same-wave kernel hide more of the latency
        // (Gram backward of the 64..256-channel taps: 60 -> 50 us per step).  STMI_PC_MIN_STEPS: tuning aid.
        const int min_steps = STMI_ENV_INT("STMI_PC_MIN_STEPS", 5);
        if (p.ntaps * (p.klen / (rb / (dtype == STMI_BF16 ? 2 : 4))) < min_steps) return false;
    }
    *rc = dtype == STMI_BF16 ? launch_pc_typed<bf16_t>(p, rb, s, name, name_len, "__bf16")
                             : launch_pc_typed<float>(p, rb, s, name, name_len, "float");
    return true;
}

}  // namespace stmi
