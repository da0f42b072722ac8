// Weight gradient of a 3x3 stride-1 SAME convolution with the input patch resident in LDS (gfx950, bf16).
//
//   D[tap][ci][co] = alpha * sum_pixels X[pixel @ tap][ci] * DY[pixel][co]
//
// The tap-per-block kernel (conv_wgrad_glds.hip) streams every pixel row of X nine times, once under each tap:
// on a residual layer of a training batch that is 148 MB through the L2 -> LDS path for two tensors of 4 MB.
// Here a block owns a 32 x 32 (ci, co) tile under ALL nine taps and a slice of whole image rows.  A stage is R image
// rows: the rows' halo patch of X ((R + 2) x (W + 2) pixels, 32 channels = 64 bytes each) and the R x W pixels of
// DY go global -> LDS once (global_load_lds_dwordx4, two or three buffers, four DMA waves) and the four MFMA waves read the
// nine shifted views of the patch from LDS: 3.5 x fewer fill bytes, and the pixel slices of a tile are 16 instead of
// 64 of the block count, so the planes a launch leaves for the fold stay small.
//
// The MFMA waves split the K loop (32-pixel groups of a stage, round robin), each holding all nine taps' 2 x 2
// fragments (144 accumulator registers); their sums meet through LDS in a fixed order ((w0 + w2) + (w1 + w3)), and
// the slices of a tile are planes of the workspace summed in slice order by the fold kernel: no float atomics, one
// summation order.
//
// LDS rows are [pixel][32 channels]; the 16-byte pieces of row r sit at position p ^ (((r >> 2) & 1) << 1) -- the
// swizzle of the 64-byte tiles of conv_wgrad_glds.hip, a function of the LDS row alone, so a view shifted by a tap
// reads the same bytes the DMA wrote.  The K order inside a 32-pixel group follows the hardware transpose
// (ds_read_b64_tr_b16); it only has to agree between the two operands.
#include "conv_wgrad.hpp"

#include <stdlib.h>

namespace stmi {

__device__ uint4 g_wgr_zero_page[4];  // zero-initialised: one 64-byte row

namespace {

#pragma clang diagnostic ignored "-Wint-to-void-pointer-cast"  // host pass only: LDS pointers are 32-bit on the device
__device__ __forceinline__ void glds16(const bf16_t* src, unsigned lds_byte_addr) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_byte_addr, 16, 0, 0);
}

constexpr int kRB = 64;                 // bytes of a pixel row: 32 bf16 channels
constexpr int kTile = 32;               // channels of X and of DY per block
constexpr int kThreads = 512;           // 4 MFMA waves + 4 DMA waves
constexpr int kMaxPatchPieces = 13;     // 1 KiB DMA pieces (16 pixel rows) per DMA wave and stage
constexpr int kMaxDyPieces = 6;
constexpr int kRedLd = 36;              // floats per row of a staged 32 x 32 tile (conflict-free fragment stores)
constexpr int kRedWave = 9 * kTile * kRedLd * 4;   // bytes of one wave's nine staged tiles

__device__ __forceinline__ int row_swz(int r) { return ((r >> 2) & 1) << 1; }

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

__global__ __launch_bounds__(kThreads) void wgrad_res_kernel(const WgradParams p, const ResGeom gm) {
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
        const int sub = lane >> 2, pos = lane & 3;   // row of the piece, 16-byte position in the row
        const bf16_t* x = (const bf16_t*)p.x;
        const bf16_t* dy = (const bf16_t*)p.dy;
        const bf16_t* zero_src = (const bf16_t*)g_wgr_zero_page + pos * 8;
        const int patch_rows = (R + 2) * PWH, dy_rows = R * W;
        // Per piece of this lane, the part of the address that does not change from stage to stage: pieces
        // start at multiples of 16 rows, so the swizzle only depends on the row inside the piece.
        int p_off[kMaxPatchPieces], p_py[kMaxPatchPieces];   // element offset relative to the stage's first row; patch row or -1
        const int chunk_e = (pos ^ row_swz(sub)) * 8;
        int py = (pw * 16 + sub) / PWH, px = pw * 16 + sub - py * PWH;   // one division, then steps of 64 rows
#pragma unroll
        for (int j = 0; j < kMaxPatchPieces; j++) {
            const int pr = ((pw + 4 * j) * 16) + sub;
            const int ix = px - 1;
            const bool ok = j < gm.patch_pp && pr < patch_rows && (unsigned)ix < (unsigned)W;
            p_py[j] = ok ? py : -1;
            p_off[j] = ((py - 1) * W + ix) * p.Ci + r0 + chunk_e;
            px += 64;
            while (px >= PWH) {
                px -= PWH;
                py++;
            }
        }
        int d_off[kMaxDyPieces];
        bool d_ok[kMaxDyPieces];
#pragma unroll
        for (int j = 0; j < kMaxDyPieces; j++) {
            const int dr = ((pw + 4 * j) * 16) + sub;
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
        // Fragment addresses: a lane reads 8 bytes of row (row0 + 4 g + q) of a tile -- byte (4 g + q) 64 + (pp >> 1)
        // 16 + (pp & 1) 8 plus row0 * 64 -- and 16 rows further down (+1024: the swizzle is the same); the swizzle
        // flips address bit 5 (the 16-byte chunk pair) with bit 2 of the row, which is bit 8 of the byte address.
        // row0 is the same for the whole wave: one add and the flip per fragment, channels 16 .. 31 at the address
        // with bit 5 flipped once more.
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const int g4 = lane >> 4, i4 = lane & 15;
        const unsigned lane_a = (unsigned)((4 * g4 + (i4 >> 2)) * kRB + ((i4 >> 1) & 1) * 16 + (i4 & 1) * 8);
        auto frag_at = [&](const char* tile, unsigned a) -> bf16x8 {
            union {
                s16x4 h[2];
                bf16x8 f;
            } u;
            u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a));
            u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(tile + a + 16 * kRB));
            return u.f;
        };
        auto swz = [&](unsigned row_bytes) -> unsigned {
            const unsigned a = lane_a + row_bytes;
            return a ^ ((a >> 3) & 32u);
        };
        const int wpr = W >> 5;          // 32-pixel groups per image row
        const int ksteps = R * wpr;
        const int tap_row = PWH * kRB;   // bytes from a patch row to the one below
        for (int s = 0; s < nst; s++) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const char* xs = lds + (s % gm.nbuf) * stage_bytes;
            const char* ds = xs + patch_bytes;
            int y = 0, col = wave << 5;          // this wave's first group: image row of the stage, first pixel
            while (col >= W) { col -= W; y++; }
            for (int ks = wave; ks < ksteps; ks += 4) {
                const unsigned arow = (unsigned)((y * PWH + col) * kRB);   // tap (0, 0): patch row y, column col
                const unsigned bsw = swz((unsigned)((y * W + col) * kRB));
                bf16x8 b[2], a[3][2];
                b[0] = frag_at(ds, bsw);
                b[1] = frag_at(ds, bsw ^ 32u);
                // the reads of tap k + 2 are issued before the products of tap k
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    const unsigned asw = swz(arow + (k / 3) * tap_row + (k % 3) * kRB);
                    a[k][0] = frag_at(xs, asw);
                    a[k][1] = frag_at(xs, asw ^ 32u);
                }
#pragma unroll
                for (int k = 0; k < 9; k++) {
                    if (k + 2 < 9) {
                        const unsigned asw = swz(arow + ((k + 2) / 3) * tap_row + ((k + 2) % 3) * kRB);
                        a[(k + 2) % 3][0] = frag_at(xs, asw);
                        a[(k + 2) % 3][1] = frag_at(xs, asw ^ 32u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    Mma<bf16_t>::mma(a[k % 3][0], b[0], acc[k][0][0]);
                    Mma<bf16_t>::mma(a[k % 3][0], b[1], acc[k][0][1]);
                    Mma<bf16_t>::mma(a[k % 3][1], b[0], acc[k][1][0]);
                    Mma<bf16_t>::mma(a[k % 3][1], b[1], acc[k][1][1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                col += 128;
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
    if (dtype != STMI_BF16 || p.ntaps != 9 || p.stride != 1 || p.cdiv != 0 || p.klen != p.Ci || p.batch_out_stride != 0) return false;
    if (p.Hi != p.Ho || p.Wi != p.Wo || p.Ci % kTile != 0 || p.Co % kTile != 0 || p.Wi % 32 != 0) return false;
    if (p.tap_out_stride != (long)p.Ci * p.Co || p.out_elems % 4 != 0) return false;
    for (int t = 0; t < 9; t++)
        if (p.taps[t].dy != t / 3 - 1 || p.taps[t].dx != t % 3 - 1 || p.taps[t].widx != t) return false;
    if ((long)p.N * p.Hi * p.Wi * p.Ci >= (1L << 31) || (long)p.N * p.Ho * p.Wo * p.Co >= (1L << 31)) return false;
    // Measured (128 -> 128 channels, wgrad + fold, us): 4 x 64 x 64 pixels 18.7 tap-per-block against 20.4 here,
    // 8 x 64 x 64 26.0 against 25.1, 4 x 128 x 128 46.9 against 33.7.  With few pixels the 16 planes a launch
    // leaves (16 tiles instead of 36 share the 256 blocks) cost more than the smaller fill saves.
    if ((long)p.N * p.Hi * p.Wi < STMI_ENV_INT("STMI_WG_RES_MIN_PIX", 32768)) return false;
    if ((((uintptr_t)p.x | (uintptr_t)p.dy | (uintptr_t)p.dw | (uintptr_t)p.ws) & 15) != 0) return false;
    // the most image rows per stage (4, 2 or 1) that divide the height and whose two operand buffers fit the LDS
    const int W = p.Wi;
    int R = 0;
    for (int r = 4; r >= 1 && R == 0; r >>= 1) {
        const int patch_pp = (((r + 2) * (W + 2) + 15) / 16 + 3) / 4, dy_pp = (r * W / 16 + 3) / 4;
        if (p.Hi % r == 0 && patch_pp <= kMaxPatchPieces && dy_pp <= kMaxDyPieces && 2 * 4 * (patch_pp + dy_pp) <= 160) R = r;
    }
    if (R == 0) return false;
    gm->R = R;
    gm->stages = p.N * p.Hi / R;
    gm->patch_pp = (((R + 2) * (W + 2) + 15) / 16 + 3) / 4;
    gm->dy_pp = (R * W / 16 + 3) / 4;
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

int launch_wgrad_res(const WgradParams& p, hipStream_t s) {
    ResGeom gm;
    STMI_REQUIRE(res_geom(p, STMI_BF16, &gm), "wgrad: the geometry does not fit the resident-patch kernel");
    STMI_REQUIRE(p.splits == (gm.stages + gm.spb - 1) / gm.spb, "wgrad: plan and launch disagree (%d slices)", p.splits);
    STMI_REQUIRE(p.splits == 1 || p.ws, "wgrad: %d row slices per tile need the workspace", p.splits);
    const size_t stage_bytes = (size_t)4 * (gm.patch_pp + gm.dy_pp) * 1024;
    const size_t red_bytes = 2 * (size_t)kRedWave;
    const size_t lds_bytes = gm.nbuf * stage_bytes > red_bytes ? gm.nbuf * stage_bytes : red_bytes;
    STMI_REQUIRE(lds_bytes <= 160 * 1024, "wgrad: operand buffers exceed the CU's LDS");
    STMI_DYN_LDS(wgrad_res_kernel, 160 * 1024, "hipFuncSetAttribute(wgrad_res_kernel)");
    const long blocks = (long)(p.Ci / kTile) * (p.Co / kTile) * p.splits;
    STMI_REQUIRE(blocks < (1L << 31), "wgrad: grid too large");
    hipLaunchKernelGGL(wgrad_res_kernel, dim3((unsigned)blocks), dim3(kThreads), lds_bytes, s, p, gm);
    STMI_LAUNCH_CHECK("wgrad_res_kernel");
    return STMI_OK;
}

}  // namespace stmi
