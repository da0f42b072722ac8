// Weight gradient of a 3x3 stride-1 SAME convolution with the input patch resident in LDS (gfx950, bf16 and fp32).
//
//   D[tap][ci][co] = alpha * sum_pixels X[pixel @ tap][ci] * DY[pixel][co]
//
// The tap-per-block kernel (conv_wgrad_glds.hip) streams every pixel row of X nine times, once under each tap:
// on a residual layer of a training batch that is 148 MB through the L2 -> LDS path for two tensors of 4 MB.
// Here a block owns a 32 x 32 (ci, co) tile under ALL nine taps and a slice of whole image rows.  A stage is R image
// rows: the rows' halo patch of X ((R + 2) x (W + 2) pixels, 32 channels = 64 or 128 bytes each) and the R x W pixels of
// DY go global -> LDS once (global_load_lds_dwordx4, two or three buffers, four DMA waves) and the four MFMA waves read the
// nine shifted views of the patch from LDS: 3.5 x fewer fill bytes, and the pixel slices of a tile are 16 instead of
// 64 of the block count, so the planes a launch leaves for the fold stay small.
//
// The MFMA waves split the K loop (32-pixel groups of a stage, round robin), each holding all nine taps' 2 x 2
// fragments (144 accumulator registers); their sums meet through LDS in a fixed order ((w0 + w2) + (w1 + w3)), and
// the slices of a tile are planes of the workspace summed in slice order by the fold kernel: no float atomics, one
// summation order.
//
// LDS rows are [pixel][32 channels]; the 16-byte pieces of row r sit at position p ^ (((r >> 2) & 1) << 1) (bf16,
// 64-byte rows) or p ^ (((r >> 1) & 3) << 1) (fp32, 128-byte rows) -- the swizzles of conv_wgrad_glds.hip, functions
// of the LDS row alone, so a view shifted by a tap reads the same bytes the DMA wrote.  bf16: the K order inside a
// 32-pixel group follows the hardware transpose (ds_read_b64_tr_b16); it only has to agree between the two
// operands.  fp32: a fragment is 16 channels x 16 pixels, four 4-byte reads per lane (v_mfma_f32_16x16x4_f32 x 4).
// In fp32 the tap-per-block kernel has 32 x 32 tiles too (128-byte rows) and is bound by its fill (8 FLOP per byte
// through the LDS); here the fill is 9 x smaller for X and the launch is bound by the fp32 matrix rate.
#include "conv_wgrad.hpp"

#include <stdlib.h>

namespace stmi {

__device__ uint4 g_wgr_zero_page[8];  // zero-initialised: one 128-byte row

namespace {

#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"  // host pass only: LDS pointers are 32-bit on the device
template <typename T>
__device__ __forceinline__ void glds16(const T* src, unsigned lds_byte_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_byte_addr, 16, 0, 0);
}

constexpr int kTile = 32;               // channels of X and of DY per block
constexpr int kThreads = 512;           // 4 MFMA waves + 4 DMA waves
constexpr int kMaxPatchPieces = 13;     // 1 KiB DMA pieces (16 or 8 pixel rows) per DMA wave and stage
constexpr int kMaxDyPieces = 6;
constexpr int kRedLd = 36;              // floats per row of a staged 32 x 32 tile (conflict-free fragment stores)
constexpr int kRedWave = 9 * kTile * kRedLd * 4;   // bytes of one wave's nine staged tiles

template <int RB>
__device__ __forceinline__ int row_swz(int r) { return RB == 128 ? ((r >> 1) & 3) << 1 : ((r >> 2) & 1) << 1; }

// s_waitcnt vmcnt(n) for a run-time n (the instruction takes an immediate): n is uniform, at most 19
__device__ __forceinline__ void wait_vmcnt_rt(int n) {
#define STMI_WAIT_CASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        STMI_WAIT_CASE(1) STMI_WAIT_CASE(2) STMI_WAIT_CASE(3) STMI_WAIT_CASE(4) STMI_WAIT_CASE(5) STMI_WAIT_CASE(6)
        STMI_WAIT_CASE(7) STMI_WAIT_CASE(8) STMI_WAIT_CASE(9) STMI_WAIT_CASE(10) STMI_WAIT_CASE(11) STMI_WAIT_CASE(12)
        STMI_WAIT_CASE(13) STMI_WAIT_CASE(14) STMI_WAIT_CASE(15) STMI_WAIT_CASE(16) STMI_WAIT_CASE(17) STMI_WAIT_CASE(18)
        STMI_WAIT_CASE(19)
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
#undef STMI_WAIT_CASE
}

struct ResGeom {
    int nbuf;         // operand buffers in LDS (2 or 3): a stage is issued nbuf - 1 stages ahead
    int R;            // image rows per stage
    int stages;       // N * H / R
    int spb;          // stages per block
    int patch_pp;     // DMA pieces per DMA wave and stage: patch, DY
    int dy_pp;
};

}  // namespace

template <typename T>
__global__ __launch_bounds__(kThreads) void wgrad_res_kernel(const WgradParams p, const ResGeom gm) {
    constexpr int kRB = kTile * (int)sizeof(T);   // bytes of a pixel row
    constexpr int RPP = 1024 / kRB;               // pixel rows per DMA piece
    constexpr int CPR = kRB / 16;                 // 16-byte chunks per row
    constexpr int EPC = 16 / (int)sizeof(T);      // elements per chunk
    constexpr int KPIX = sizeof(T) == 2 ? 32 : 16;   // pixels per K group of a fragment
    extern __shared__ __attribute__((aligned(128))) char lds[];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Hi, W = p.Wi, R = gm.R;
    const int PWH = W + 2;
    const int patch_bytes = 4 * gm.patch_pp * 1024, dy_bytes = 4 * gm.dy_pp * 1024;
    const int stage_bytes = patch_bytes + dy_bytes;

    // Blocks b and b+8 share an XCD.  Every XCD gets one contiguous run of (row slice, tile) pairs, tiles
    // fastest: the blocks that read the same rows share one L2.
    const int rtiles = p.Ci / kTile, ctiles = p.Co / kTile, ntiles = rtiles * ctiles;
    unsigned logical;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        logical = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    const int split = logical / ntiles;
    const int t = logical - split * ntiles;
    const int r0 = (t / ctiles) * kTile, c0 = (t % ctiles) * kTile;
    const int st_beg = split * gm.spb, st_end = min(st_beg + gm.spb, gm.stages);
    const int nst = st_end - st_beg;   // >= 1 by the plan

    f32x4 acc[9][2][2];
#pragma unroll
    for (int k = 0; k < 9; k++)
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[k][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wave >= 4) {
        // ------------------------------------------------------------------ DMA waves
        const int pw = wave - 4;
        const int sub = lane / CPR, pos = lane % CPR;   // row of the piece, 16-byte position in the row
        const T* x = (const T*)p.x;
        const T* dy = (const T*)p.dy;
        const T* zero_src = (const T*)g_wgr_zero_page + pos * EPC;
        const int patch_rows = (R + 2) * PWH, dy_rows = R * W;
        // Per piece of this lane, the part of the address that does not change from stage to stage: pieces
        // start at multiples of their row count, so the swizzle only depends on the row inside the piece.
        int p_off[kMaxPatchPieces], p_py[kMaxPatchPieces];   // element offset relative to the stage's first row; patch row or -1
        const int chunk_e = (pos ^ row_swz<kRB>(sub)) * EPC;
        int py = (pw * RPP + sub) / PWH, px = pw * RPP + sub - py * PWH;   // one division, then steps of 4 pieces
#pragma unroll
        for (int j = 0; j < kMaxPatchPieces; j++) {
            const int pr = ((pw + 4 * j) * RPP) + sub;
            const int ix = px - 1;
            const bool ok = j < gm.patch_pp && pr < patch_rows && (unsigned)ix < (unsigned)W;
            p_py[j] = ok ? py : -1;
            p_off[j] = ((py - 1) * W + ix) * p.Ci + r0 + chunk_e;
            px += 4 * RPP;
            while (px >= PWH) {
                px -= PWH;
                py++;
            }
        }
        int d_off[kMaxDyPieces];
        bool d_ok[kMaxDyPieces];
#pragma unroll
        for (int j = 0; j < kMaxDyPieces; j++) {
            const int dr = ((pw + 4 * j) * RPP) + sub;
            d_ok[j] = j < gm.dy_pp && dr < dy_rows;
            d_off[j] = dr * p.Co + c0 + chunk_e;
        }
        auto issue = [&](int st, int buf) {
            const int gr0 = st * R;                  // first image row of the stage, counted over all samples
            const int n = gr0 / H, y0 = gr0 - n * H;
            const long xbase = ((long)n * H + y0) * W * p.Ci;
            const long dbase = (long)gr0 * W * p.Co;
            const unsigned pdst = lds_base + buf * stage_bytes + pw * 1024;
#pragma unroll
            for (int j = 0; j < kMaxPatchPieces; j++) {
                if (j < gm.patch_pp) {
                    const bool ok = p_py[j] >= 0 && (unsigned)(y0 + p_py[j] - 1) < (unsigned)H;
                    glds16(ok ? x + (xbase + p_off[j]) : zero_src, pdst + j * 4096);
                }
            }
            const unsigned ddst = lds_base + buf * stage_bytes + patch_bytes + pw * 1024;
#pragma unroll
            for (int j = 0; j < kMaxDyPieces; j++) {
                if (j < gm.dy_pp) glds16(d_ok[j] ? dy + (dbase + d_off[j]) : zero_src, ddst + j * 4096);
            }
        };
        const int np = gm.patch_pp + gm.dy_pp;   // DMA instructions of this wave per stage
        for (int s = 0; s < gm.nbuf - 1 && s < nst; s++) issue(st_beg + s, s);
        for (int s = 0; s < nst; s++) {
            // stage s has landed: all but the stages issued after it (at most nbuf - 2) may still be in flight
            wait_vmcnt_rt(np * min(nst - 1 - s, gm.nbuf - 2));
            __builtin_amdgcn_s_barrier();                      // ... and the MFMA waves are done with stage s - 1
            asm volatile("" ::: "memory");
            if (s + gm.nbuf - 1 < nst) issue(st_beg + s + gm.nbuf - 1, (s + gm.nbuf - 1) % gm.nbuf);
        }
    } else {
        // ------------------------------------------------------------------ MFMA waves
        // Fragment addresses.  The swizzle flips address bits with row bits, and the row bits are bits of the byte
        // address too (a row is 64 or 128 bytes): with a = lane part + row0 * row bytes (row0 is the same for the
        // whole wave), bf16 flips bit 5 with bit 8 (bit 2 of the row) and fp32 flips bits 5, 6 with bits 8, 9
        // (bits 1, 2 of the row).  Channels 16 .. 31 sit 32 (bf16) or 64 (fp32) bytes further: one more flip.
        //   bf16: a lane reads 8 bytes of row (row0 + 4 g + q) and of the row 16 below (same swizzle) with the
        //         transposing read: 16 channels x 32 pixels;
        //   fp32: a lane reads 4 bytes of rows row0 + g, + 4, + 8, + 12 (channel col0 + i): 16 channels x 16 pixels.
        typedef typename Mma<T>::frag frag;
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int g4 = lane >> 4, i4 = lane & 15;
        const unsigned lane_a = sizeof(T) == 2 ? (unsigned)((4 * g4 + (i4 >> 2)) * kRB + ((i4 >> 1) & 1) * 16 + (i4 & 1) * 8)
                                               : (unsigned)(g4 * kRB + i4 * 4);
        constexpr unsigned kHalf = sizeof(T) == 2 ? 32u : 64u;   // channels 16 .. 31
        auto swz = [&](unsigned a) -> unsigned {
            return sizeof(T) == 2 ? a ^ ((a >> 3) & 32u) : a ^ ((a >> 3) & 0x60u);
        };
        // fragment of channels (hi ? 16.. : 0..) x KPIX pixels from tile row byte offset `rowb`
        auto frag_at = [&](const char* tile, unsigned rowb, bool hi) -> frag {
            if constexpr (sizeof(T) == 2) {
                const unsigned a = swz(lane_a + rowb) ^ (hi ? kHalf : 0u);
                union {
                    s16x4 h[2];
                    bf16x8 f;
                } u;
                u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a));
                u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a + 16 * kRB));
                return u.f;
            } else {
                f32x4 f;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    f[q] = *reinterpret_cast<const float*>(tile + (swz(lane_a + rowb + q * 4 * kRB) ^ (hi ? kHalf : 0u)));
                return f;
            }
        };
        const int wpr = W / KPIX;        // K groups per image row
        const int ksteps = R * wpr;
        const int tap_row = PWH * kRB;   // bytes from a patch row to the one below
        for (int s = 0; s < nst; s++) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* xs = lds + (s % gm.nbuf) * stage_bytes;
            const char* ds = xs + patch_bytes;
            int y = 0, col = wave * KPIX;        // this wave's first group: image row of the stage, first pixel
            while (col >= W) { col -= W; y++; }
            for (int ks = wave; ks < ksteps; ks += 4) {
                const unsigned arow = (unsigned)((y * PWH + col) * kRB);   // tap (0, 0): patch row y, column col
                const unsigned brow = (unsigned)((y * W + col) * kRB);
                frag b[2], a[3][2];
                b[0] = frag_at(ds, brow, false);
                b[1] = frag_at(ds, brow, true);
                // the reads of tap k + 2 are issued before the products of tap k
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const unsigned ar = arow + (k / 3) * tap_row + (k % 3) * kRB;
                    a[k][0] = frag_at(xs, ar, false);
                    a[k][1] = frag_at(xs, ar, true);
                }
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    if (k + 2 < 9) {
                        const unsigned ar = arow + ((k + 2) / 3) * tap_row + ((k + 2) % 3) * kRB;
                        a[(k + 2) % 3][0] = frag_at(xs, ar, false);
                        a[(k + 2) % 3][1] = frag_at(xs, ar, true);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    Mma<T>::mma(a[k % 3][0], b[0], acc[k][0][0]);
                    Mma<T>::mma(a[k % 3][0], b[1], acc[k][0][1]);
                    Mma<T>::mma(a[k % 3][1], b[0], acc[k][1][0]);
                    Mma<T>::mma(a[k % 3][1], b[1], acc[k][1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                col += 4 * KPIX;
                while (col >= W) { col -= W; y++; }
            }
        }
    }

    // ---------------------------------------------------------------- the four waves' sums, in a fixed order
    float* red = reinterpret_cast<float*>(lds);
    const int g = lane >> 4, i16 = lane & 15;
    auto stage_acc = [&](float* dst) {
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        dst[(k * kTile + a * 16 + 4 * g + u) * kRedLd + b * 16 + i16] = acc[k][a][b][u];
    };
    __syncthreads();                                   // every wave is done with the operand buffers
    if (wave == 2 || wave == 3) stage_acc(red + (wave - 2) * (kRedWave / 4));
    __syncthreads();
    if (wave < 2) {
        const float* src = red + wave * (kRedWave / 4);
#pragma unroll
        for (int k = 0; k < 9; k++)
#pragma unroll
            for (int a = 0; a < 2; a++)
#pragma unroll
                for (int b = 0; b < 2; b++)
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        acc[k][a][b][u] += src[(k * kTile + a * 16 + 4 * g + u) * kRedLd + b * 16 + i16];
    }
    __syncthreads();
    if (wave < 2) stage_acc(red + wave * (kRedWave / 4));
    __syncthreads();
    // a slice of a shared tile writes its plane of the workspace; a tile with one slice goes to dw itself
    float* out = (p.splits > 1 ? p.ws + (long)split * p.out_elems : p.dw) + (long)r0 * p.Co + c0;
    const bool add = p.splits == 1 && p.accumulate;
    const float* red1 = red + kRedWave / 4;
    for (int e = tid; e < 9 * kTile * (kTile / 4); e += kThreads) {
        const int q = e & 7, row = (e >> 3) & 31, k = e >> 8;   // 16-byte piece of a tile row, row, tap
        const int o = (k * kTile + row) * kRedLd + 4 * q;
        f32x4 v = (*reinterpret_cast<const f32x4*>(red + o) + *reinterpret_cast<const f32x4*>(red1 + o)) * p.alpha;
        // (tap k writes weight slice k: checked on the host; a per-thread index into the tap table would be a
        // vector load from the kernel-argument buffer, which lives in host memory)
        f32x4* dst = reinterpret_cast<f32x4*>(out + (long)k * p.tap_out_stride + (long)row * p.Co + 4 * q);
        if (add) v += *dst;
        *dst = v;
    }
}

// ---------------------------------------------------------------------------
static bool res_geom(const WgradParams& p, int dtype, ResGeom* gm) {
    if ((dtype != STMI_BF16 && dtype != STMI_F32) || p.ntaps != 9 || p.stride != 1 || p.cdiv != 0 || p.klen != p.Ci || p.batch_out_stride != 0) return false;
    if (p.Hi != p.Ho || p.Wi != p.Wo || p.Ci % kTile != 0 || p.Co % kTile != 0 || p.Wi % 32 != 0) return false;
    if (p.tap_out_stride != (long)p.Ci * p.Co || p.out_elems % 4 != 0) return false;
    for (int t = 0; t < 9; t++)
        if (p.taps[t].dy != t / 3 - 1 || p.taps[t].dx != t % 3 - 1 || p.taps[t].widx != t) return false;
    if ((long)p.N * p.Hi * p.Wi * p.Ci >= (1L << 31) || (long)p.N * p.Ho * p.Wo * p.Co >= (1L << 31)) return false;
    // Measured (128 -> 128 channels, wgrad + fold, us): 4 x 64 x 64 pixels 18.7 tap-per-block against 20.4 here,
    // 8 x 64 x 64 26.0 against 25.1, 4 x 128 x 128 46.9 against 33.7.  With few pixels the 16 planes a launch
    // leaves (16 tiles instead of 36 share the 256 blocks) cost more than the smaller fill saves.
    // fp32 (32 x 32 tiles in the tap-per-block kernel as well, which is bound by its fill): measured below.
    if (dtype == STMI_BF16 && (long)p.N * p.Hi * p.Wi < STMI_ENV_INT("STMI_WG_RES_MIN_PIX", 32768)) return false;
    if (dtype == STMI_F32 && STMI_ENV_SET("STMI_NO_WG_RES_F32")) return false;
    if ((((uintptr_t)p.x | (uintptr_t)p.dy | (uintptr_t)p.dw | (uintptr_t)p.ws) & 15) != 0) return false;
    // the most image rows per stage (4, 2 or 1) that divide the height and whose two operand buffers fit the LDS
    const int W = p.Wi;
    const int rpp = dtype == STMI_BF16 ? 16 : 8;   // pixel rows per 1 KiB DMA piece
    int R = 0;
    for (int r = 4; r >= 1 && R == 0; r >>= 1) {
        const int patch_pp = (((r + 2) * (W + 2) + rpp - 1) / rpp + 3) / 4, dy_pp = (r * W / rpp + 3) / 4;
        if (p.Hi % r == 0 && patch_pp <= kMaxPatchPieces && dy_pp <= kMaxDyPieces && 2 * 4 * (patch_pp + dy_pp) <= 160) R = r;
    }
    if (R == 0) return false;
    gm->R = R;
    gm->stages = p.N * p.Hi / R;
    gm->patch_pp = (((R + 2) * (W + 2) + rpp - 1) / rpp + 3) / 4;
    gm->dy_pp = (R * W / rpp + 3) / 4;
    gm->nbuf = 3 * 4 * (gm->patch_pp + gm->dy_pp) <= 160 ? 3 : 2;
    // one launch's worth of blocks: the tiles times as many row slices as give every CU a block
    const long tiles = (long)(p.Ci / kTile) * (p.Co / kTile);
    const int tb = STMI_ENV_INT("STMI_WG_BLOCKS", 0);
    long splits = (tb > 0 ? tb : 256) / tiles;
    if (splits > gm->stages) splits = gm->stages;
    if (splits < 1) splits = 1;
    gm->spb = (int)((gm->stages + splits - 1) / splits);
    return true;
}

bool wgrad_res_plan(const WgradParams& p, int dtype, WgPlan* plan) {
    ResGeom gm;
    if (STMI_ENV_SET("STMI_NO_WG_RES") || !res_geom(p, dtype, &gm)) return false;
    plan->splits = (gm.stages + gm.spb - 1) / gm.spb;
    plan->pix_per_block = gm.spb * gm.R * p.Wi;
    return true;
}

int launch_wgrad_res(const WgradParams& p, int dtype, hipStream_t s) {
    ResGeom gm;
    STMI_REQUIRE(res_geom(p, dtype, &gm), "wgrad: the geometry does not fit the resident-patch kernel");
    STMI_REQUIRE(p.splits == (gm.stages + gm.spb - 1) / gm.spb, "wgrad: plan and launch disagree (%d slices)", p.splits);
    STMI_REQUIRE(p.splits == 1 || p.ws, "wgrad: %d row slices per tile need the workspace", p.splits);
    const size_t stage_bytes = (size_t)4 * (gm.patch_pp + gm.dy_pp) * 1024;
    const size_t red_bytes = 2 * (size_t)kRedWave;
    const size_t lds_bytes = gm.nbuf * stage_bytes > red_bytes ? gm.nbuf * stage_bytes : red_bytes;
    STMI_REQUIRE(lds_bytes <= 160 * 1024, "wgrad: operand buffers exceed the CU's LDS");
    const long blocks = (long)(p.Ci / kTile) * (p.Co / kTile) * p.splits;
    STMI_REQUIRE(blocks < (1L << 31), "wgrad: grid too large");
    if (dtype == STMI_BF16) {
        STMI_DYN_LDS(wgrad_res_kernel<bf16_t>, 160 * 1024, "hipFuncSetAttribute(wgrad_res_kernel)");
        hipLaunchKernelGGL(wgrad_res_kernel<bf16_t>, dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, p, gm);
    } else {
        STMI_DYN_LDS(wgrad_res_kernel<float>, 160 * 1024, "hipFuncSetAttribute(wgrad_res_kernel)");
        hipLaunchKernelGGL(wgrad_res_kernel<float>, dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, p, gm);
    }
    STMI_LAUNCH_CHECK("wgrad_res_kernel");
    return STMI_OK;
}

}  // namespace stmi
