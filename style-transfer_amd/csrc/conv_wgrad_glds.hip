// Pixel-reduction GEMM, LDS-DMA variant with producer / consumer waves (gfx950).
//
//   D[tap][e][co] += alpha * sum_pixels X[pixel @ tap][e] * DY[pixel][co]
//
// for operands whose channel counts are multiples of 128 bytes.  A block owns one
// (tap, 128-byte channel chunk of X, 128-byte channel chunk of DY) tile and a slice of the
// pixels.  Per step, 128 pixel rows of each operand (128 B each) go global -> LDS with
// global_load_lds_dwordx4 through a 4-deep ring; rows outside the image or the slice come
// from a page of zeros.  Eight producer waves only issue the DMA (each issue stalls its wave
// for 60-180 cycles) and keep per-row pixel coordinates incrementally; four consumer waves
// (one per SIMD beside two producers) read transposed fragments and run the MFMAs.  One
// s_barrier per step hands a filled slot to the consumers and a drained one back.
//
// LDS rows are [pixel][channel chunk]; the 16-byte pieces of row r sit at position
// p ^ (((r >> 1) & 3) << 1), which makes the ds_read_b64_tr_b16 fragment reads (4 rows x 32
// bytes per 16 lanes) conflict-free.  The K order inside a 32-pixel group follows the
// hardware transpose; it only has to agree between the two operands.
//
// The finished tile goes through LDS so that every wave stores whole 256-byte rows.  Blocks that
// share a tile (each with a slice of the pixels) do not add into it: a slice writes its partial tile
// to a plane of its own in the workspace and a fold kernel sums the planes in slice order, so the
// result has one fixed summation order (float atomics would make it depend on the arrival order).
#include "conv_wgrad.hpp"

#include <stdlib.h>

namespace stmi {

__device__ uint4 g_wg_zero_page[8];  // zero-initialised

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

// chunk swizzle of a tile with RB-byte rows (conflict-free ds_read_b64_tr_b16 fragment reads)
template <int RB>
__device__ __forceinline__ int row_swz(int r) {
    // 256-byte rows (fp32 tiles of 64 channels, four-byte reads): the four rows a read touches go to four
    // different 64-byte groups
    return RB == 256 ? (r & 3) << 2 : RB == 128 ? ((r >> 1) & 3) << 1 : ((r >> 2) & 1) << 1;
}

// fragment of 16 channels (col0..) x one K group of pixels (row0..) from a swizzled tile
template <typename T, int RB> struct TileFrag;

template <int RB> struct TileFrag<bf16_t, RB> {
    static constexpr int KPIX = 32;
    static __device__ __forceinline__ bf16x8 load(const char* tile, int row0, int col0, int lane) {
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        const int r = row0 + 4 * g + q;  // r + 16 has the same swizzle
        const int chunk = (col0 >> 3) + (pp >> 1);
        const char* a0 = tile + r * RB + ((chunk ^ row_swz<RB>(r)) << 4) + ((pp & 1) << 3);
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * RB));
        union {
            s16x4 h[2];
            bf16x8 f;
        } u;
        u.h[0] = lo;
        u.h[1] = hi;
        return u.f;
    }
};

template <int RB> struct TileFrag<float, RB> {
    static constexpr int KPIX = 16;
    static __device__ __forceinline__ f32x4 load(const char* tile, int row0, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const int ch = col0 + i;
        f32x4 f;
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int r = row0 + 4 * s + g;
            f[s] = *reinterpret_cast<const float*>(tile + r * RB + (((ch >> 2) ^ row_swz<RB>(r)) << 4) + ((ch & 3) << 2));
        }
        return f;
    }
};

constexpr int kStepPix = 128;  // pixel rows per ring slot and operand (64 for the 256-byte rows of fp32)
constexpr int kRing = 4;
constexpr int kProducers = 8;  // DMA-issuing waves: the fill rate of a CU scales with the waves that issue
constexpr int kThreads = 64 * (4 + kProducers);

}  // namespace

// XB / DB: bytes per pixel row of the X / DY tile (128 or 64: 64 or 32 bf16 channels)
template <typename T, int XB, int DB, int SP = kStepPix>
__global__ __launch_bounds__(kThreads) void wgrad_glds_kernel(const WgradParams p) {
    constexpr int kStepPix = SP;   // (shadows the default: everything below counts in this launch's steps)
    constexpr int EPC = 16 / sizeof(T);
    constexpr int TE = XB / (int)sizeof(T), TD = DB / (int)sizeof(T);  // tile: TE x-channels by TD dy-channels
    constexpr int TWX = TE / 32, TWD = TD / 32;  // 16-channel fragments per consumer wave (2 x 2 waves)
    constexpr int KPIX = TileFrag<T, XB>::KPIX;
    constexpr int XOPER = kStepPix * XB, SLOT = kStepPix * (XB + DB);  // bytes
    constexpr int XRP = 1024 / XB, DRP = 1024 / DB;                    // rows per 1 KiB DMA piece
    constexpr int XPW = kStepPix / XRP / kProducers, DPW = kStepPix / DRP / kProducers;  // pieces per producer and step
    constexpr int NI = XPW + DPW;
    static_assert(TWX >= 1 && TWD >= 1, "tile too small for 2 x 2 waves");
    static_assert(XPW >= 1 && DPW >= 1, "fewer pieces than producer waves");
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(128))) char lds[];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rtiles = p.klen / TE, ctiles = p.Co / TD;
    // Blocks b and b+8 share an XCD.  Every XCD gets one contiguous run of (pixel slice, tile)
    // pairs, tiles fastest: the blocks that read the same pixels then share one 4 MiB L2 and
    // the two operands are fetched through it once instead of once per XCD.
    const int ntiles = p.ntaps * rtiles * ctiles;
    unsigned logical;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        logical = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    const int slice = logical / ntiles;
    int t = logical - slice * ntiles;
    const int ct = t % ctiles;
    t /= ctiles;
    const int rt = t % rtiles;
    const int tap_i = t / rtiles;
    const int r0 = rt * TE, c0 = ct * TD;
    const bool batched = p.batch_out_stride != 0;
    const int HWo = p.Ho * p.Wo;
    int sample = 0, split = slice;
    if (batched) {
        sample = slice / p.splits;
        split = slice - sample * p.splits;
    }
    const int pix_total = batched ? HWo : p.N * HWo;
    const int pbeg = split * p.pix_per_block;
    const int pend = min(pbeg + p.pix_per_block, pix_total);
    const int nsteps = pbeg < pend ? (pend - pbeg + kStepPix - 1) / kStepPix : 0;

    f32x4 acc[TWX][TWD];
#pragma unroll
    for (int a = 0; a < TWX; a++)
#pragma unroll
        for (int b = 0; b < TWD; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = wave - 4;
        const Tap tap = p.taps[tap_i];
        const T* x = (const T*)p.x;
        const T* dy = (const T*)p.dy;
        // lane -> (row of the piece, 16-byte position in the row); pieces start at multiples of
        // their row count, so the swizzle only depends on the row inside the piece
        const int xsub = lane / (XB / 16), xpos = lane % (XB / 16);
        const int dsub = lane / (DB / 16), dpos = lane % (DB / 16);
        const T* zero_src = (const T*)g_wg_zero_page;
        // Per X row slot: the pixel's index, its input coordinates under this tap and the element
        // offset of its row, advanced by one step (128 pixels) with additions only (integer
        // multiplies run at quarter rate).  DY rows are consecutive pixels: an offset is enough.
        int xpix[XPW], iy[XPW], ix[XPW], oyv[XPW], xoff[XPW];
#pragma unroll
        for (int j = 0; j < XPW; j++) {
            const int pi = pbeg + (pw * XPW + j) * XRP + xsub;
            xpix[j] = pi;
            const int n = batched ? sample : pi / HWo;
            const int rem = pi - (batched ? 0 : n * HWo);
            const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
            oyv[j] = oy;
            iy[j] = oy * p.stride + tap.dy;
            ix[j] = ox * p.stride + tap.dx;
            xoff[j] = ((n * p.Hi + iy[j]) * p.Wi + ix[j]) * p.Ci + r0 + (xpos ^ row_swz<XB>(xsub)) * EPC;
        }
        int dpix[DPW], doff[DPW];
#pragma unroll
        for (int j = 0; j < DPW; j++) {
            const int pi = pbeg + (pw * DPW + j) * DRP + dsub;
            dpix[j] = pi;
            doff[j] = ((batched ? sample * HWo : 0) + pi) * p.Co + c0 + (dpos ^ row_swz<DB>(dsub)) * EPC;
        }
        const int dq = kStepPix / p.Wo, dr = kStepPix - dq * p.Wo;
        const int x_col = p.stride * p.Ci, x_row = p.stride * p.Wi * p.Ci;   // one output column / row, in x elements
        const int x_adv = dr * x_col + dq * x_row;                           // dr columns and dq rows
        const int x_wrap = x_row - p.Wo * x_col;                             // column wrap: back Wo columns, down a row
        const int x_img = (p.Hi * p.Wi * p.Ci) - p.Ho * x_row;               // row wrap: back Ho rows, on to the next sample
        const int d_adv = kStepPix * p.Co;
        auto stage = [&](int slot) {
            const unsigned xdst = lds_base + slot * SLOT + pw * XPW * 1024;
            const unsigned ddst = lds_base + slot * SLOT + XOPER + pw * DPW * 1024;
#pragma unroll
            for (int j = 0; j < XPW; j++) {
                const bool ok = xpix[j] < pend && (unsigned)iy[j] < (unsigned)p.Hi && (unsigned)ix[j] < (unsigned)p.Wi;
                glds16(ok ? x + xoff[j] : zero_src, xdst + j * 1024);
                // advance this row slot by one step
                xpix[j] += kStepPix;
                ix[j] += dr * p.stride;
                iy[j] += dq * p.stride;
                oyv[j] += dq;
                xoff[j] += x_adv;
                if (ix[j] - tap.dx >= p.Wo * p.stride) {
                    ix[j] -= p.Wo * p.stride;
                    iy[j] += p.stride;
                    oyv[j]++;
                    xoff[j] += x_wrap;
                }
                if (!batched) {
                    while (oyv[j] >= p.Ho) {
                        oyv[j] -= p.Ho;
                        iy[j] -= p.Ho * p.stride;
                        xoff[j] += x_img;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < DPW; j++) {
                glds16(dpix[j] < pend ? dy + doff[j] : zero_src, ddst + j * 1024);
                dpix[j] += kStepPix;
                doff[j] += d_adv;
            }
        };
#pragma unroll
        for (int s = 0; s < kRing - 1; s++)
            if (s < nsteps) stage(s);
        for (int s = 0; s < nsteps; s++) {
            // steps issued after s so far: min(nsteps - 1 - s, kRing - 2)
            const int later = nsteps - 1 - s;
            if (kRing >= 4 && later >= 2) wait_vmcnt<2 * NI>();
            else if (later >= 1) wait_vmcnt<NI>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // slot of step s is filled; slot of step s-1 is drained
            asm volatile("" ::: "memory");
            if (s + kRing - 1 < nsteps) stage((s + kRing - 1) % kRing);
        }
    } else {
        // ------------------------------------------------------------------ consumers
        const int wr = wave >> 1, wcn = wave & 1;
        for (int s = 0; s < nsteps; s++) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* xs = lds + (s % kRing) * SLOT;
            const char* ds = xs + XOPER;
#pragma unroll
            for (int kk = 0; kk < kStepPix / KPIX; kk++) {
                frag a[TWX], b[TWD];
#pragma unroll
                for (int i = 0; i < TWX; i++) a[i] = TileFrag<T, XB>::load(xs, kk * KPIX, (wr * TWX + i) * 16, lane);
#pragma unroll
                for (int i = 0; i < TWD; i++) b[i] = TileFrag<T, DB>::load(ds, kk * KPIX, (wcn * TWD + i) * 16, lane);
#pragma unroll
                for (int i = 0; i < TWX; i++)
#pragma unroll
                    for (int j = 0; j < TWD; j++) Mma<T>::mma(a[i], b[j], acc[i][j]);
            }
        }
    }

    // epilogue: tile -> LDS -> whole rows, 16 bytes per lane
    constexpr int LD = TD + 4;
    float* stage_f = reinterpret_cast<float*>(lds);
    __syncthreads();
    if (wave < 4) {
        const int wr = wave >> 1, wcn = wave & 1, g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int a = 0; a < TWX; a++)
#pragma unroll
            for (int b = 0; b < TWD; b++)
#pragma unroll
                for (int u = 0; u < 4; u++)
                    stage_f[((wr * TWX + a) * 16 + 4 * g + u) * LD + (wcn * TWD + b) * 16 + i] = acc[a][b][u];
    }
    __syncthreads();
    // a slice of a shared tile writes its plane of the workspace; a tile with one slice goes to dw itself
    float* out = (p.splits > 1 ? p.ws + (long)split * p.out_elems : p.dw) + (long)sample * p.batch_out_stride +
                 (long)tap_i * p.tap_out_stride + (long)r0 * p.Co + c0;
    const bool add = p.splits == 1 && p.accumulate;
    constexpr int QPR = TD / 4;   // 16-byte pieces per tile row
    const int q = tid % QPR;
    for (int row = tid / QPR; row < TE; row += kThreads / QPR) {
        f32x4 v = *reinterpret_cast<const f32x4*>(stage_f + row * LD + 4 * q) * p.alpha;
        f32x4* dst = reinterpret_cast<f32x4*>(out + (long)row * p.Co + 4 * q);
        if (add) v += *dst;
        *dst = v;
    }
}

// ---------------------------------------------------------------------------
// bytes per tile row for a channel count, 0 when the kernel cannot take it
static int row_bytes(int channels, int dtype) {
    const int es = dtype == STMI_BF16 ? 2 : 4;
    // fp32: 64 channels a row where both operands have them (tile_dims): twice the products per byte of fill of
    // the 32-channel tiles, and the fill is what bounds these launches
    if (dtype != STMI_BF16 && channels * es % 256 == 0 && !STMI_ENV_SET("STMI_NO_WG_F32_WIDE")) return 256;
    if (channels * es % 128 == 0) return 128;
    if (dtype == STMI_BF16 && channels * es % 64 == 0) return 64;  // 32 bf16 channels; f32 tiles would be too narrow
    return 0;
}

static bool wg_eligible(const WgradParams& p, int dtype) {
    if (p.cdiv != 0 || p.klen != p.Ci || row_bytes(p.Ci, dtype) == 0 || row_bytes(p.Co, dtype) == 0) return false;
    if ((long)p.N * p.Hi * p.Wi * p.Ci >= (1L << 31) || (long)p.N * p.Ho * p.Wo * p.Co >= (1L << 31)) return false;
    if ((((uintptr_t)p.x | (uintptr_t)p.dy | (uintptr_t)p.dw | (uintptr_t)p.ws) & 15) != 0) return false;
    if (p.out_elems % 4 != 0) return false;   // planes of the workspace stay 16-byte aligned
    return true;
}

// pixel slices per tile for TE x TD tiles
static WgPlan plan_for(const WgradParams& p, int te, int td) {
    const bool batched = p.batch_out_stride != 0;
    const long pix_total = batched ? (long)p.Ho * p.Wo : (long)p.N * p.Ho * p.Wo;
    const long tiles = (long)p.ntaps * (p.klen / te) * (p.Co / td);
    const long groups = tiles * (batched ? p.N : 1);
    // about one block per CU for 64 x 64 tiles, two for the smaller tiles whose steps are short
    const int tb = STMI_ENV_INT("STMI_WG_BLOCKS", 0);
    const long target = tb > 0 ? tb : (te * td >= 64 * 64 ? 256 : 512);
    long splits = target / groups;
    const long max_splits = (pix_total + 2 * kStepPix - 1) / (2 * kStepPix);  // at least two steps per block
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    long ppb = (pix_total + splits - 1) / splits;
    ppb = (ppb + kStepPix - 1) / kStepPix * kStepPix;
    splits = (pix_total + ppb - 1) / ppb;
    WgPlan plan;
    plan.pix_per_block = (int)ppb;
    plan.splits = (int)splits;
    return plan;
}

// bytes per tile row of the two operands: bf16 128 or 64 each; fp32 256 for both where both have 64 channels, else 128
static void tile_row_bytes(const WgradParams& p, int dtype, int* xb, int* db) {
    *xb = row_bytes(p.Ci, dtype);
    *db = row_bytes(p.Co, dtype);
    if (dtype != STMI_BF16 && (*xb != 256 || *db != 256)) *xb = *db = 128;
}

static void tile_dims(const WgradParams& p, int dtype, int* te, int* td) {
    const int es = dtype == STMI_BF16 ? 2 : 4;
    int xb, db;
    tile_row_bytes(p, dtype, &xb, &db);
    *te = xb / es;
    *td = db / es;
}

bool wgrad_glds_plan(const WgradParams& p, int dtype, WgPlan* plan) {
    if (!wg_eligible(p, dtype) || STMI_ENV_SET("STMI_NO_WG_GLDS")) return false;
    int te, td;
    tile_dims(p, dtype, &te, &td);
    *plan = plan_for(p, te, td);
    return true;
}

template <typename T, int XB, int DB, int SP = kStepPix>
static int launch_wg_glds(const WgradParams& p, hipStream_t s) {
    constexpr int TE = XB / (int)sizeof(T), TD = DB / (int)sizeof(T);
    constexpr size_t kLds = (size_t)kRing * SP * (XB + DB);
    static_assert(kLds <= 160 * 1024, "the ring exceeds the CU's LDS");
    STMI_DYN_LDS((wgrad_glds_kernel<T, XB, DB, SP>), (int)kLds, "hipFuncSetAttribute(wgrad_glds_kernel)");
    const bool batched = p.batch_out_stride != 0;
    const long tiles = (long)p.ntaps * (p.klen / TE) * (p.Co / TD);
    const long blocks = tiles * p.splits * (batched ? p.N : 1);
    STMI_REQUIRE(blocks < (1L << 31), "wgrad: grid too large");
    STMI_REQUIRE(p.splits == 1 || p.ws, "wgrad: %d pixel slices per tile need the workspace", p.splits);
    dim3 grid((unsigned)blocks);
    hipLaunchKernelGGL((wgrad_glds_kernel<T, XB, DB, SP>), grid, dim3(kThreads), kLds, s, p);
    STMI_LAUNCH_CHECK("wgrad_glds_kernel");
    return STMI_OK;
}

int launch_wgrad_glds(const WgradParams& p, int dtype, hipStream_t s) {
    int xb, db;
    tile_row_bytes(p, dtype, &xb, &db);
    if (dtype != STMI_BF16) return xb == 256 ? launch_wg_glds<float, 256, 256, 64>(p, s) : launch_wg_glds<float, 128, 128>(p, s);
    if (xb == 128) return db == 128 ? launch_wg_glds<bf16_t, 128, 128>(p, s) : launch_wg_glds<bf16_t, 128, 64>(p, s);
    return db == 128 ? launch_wg_glds<bf16_t, 64, 128>(p, s) : launch_wg_glds<bf16_t, 64, 64>(p, s);
}

}  // namespace stmi
