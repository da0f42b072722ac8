// Stride-2 backward-data / transposed 3x3 convolution with the input patch held in LDS (gfx950).
//
// A transposed convolution writes four output parity classes (oy & 1, ox & 1); class c is a small stride-1
// convolution of the input (1, 2, 2 and 4 taps of the 3 x 3 kernel).  The gather-GEMM form fetches every
// input pixel row once per tap, nine times in all, and these layers have few channels (64 -> 32, 128 -> 64):
// they are bound by that fetch, not by the MFMA.  Here a block owns a 4 x 32 rectangle of input positions,
// keeps the halo patch of that rectangle (ALL channels) in LDS for the whole block, runs the taps of the
// four classes over it with the weight tiles streamed through a small ring, and holds the four classes'
// accumulators until the end: the input is fetched about 1.2 times, and a wave leaves with two complete
// output rows of 64 pixels, staged through LDS and stored as whole contiguous rows.
//
// 256 threads, wave w = tile row w (32 positions = two 16-pixel fragments) x all output channels.  Every wave
// both issues the DMA pieces (global_load_lds_dwordx4, source-side XOR swizzle, a page of zeros behind
// out-of-image rows) and multiplies; several blocks fit a CU (33 .. 66 KB of LDS), so one block's loads
// overlap another's arithmetic and stores.
#include "conv_gemm.hpp"

#include <type_traits>

namespace stmi {

__device__ uint4 g_up2_zero_page[8];  // zero-initialised: source of out-of-image rows

namespace {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
// fragment reads are inline asm: a compiler-visible LDS read behind an LDS-DMA write is preceded by vmcnt(0)
__device__ __forceinline__ u32x4 lds_read16(unsigned addr) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_landed(u32x4& v) { asm volatile("" : "+v"(v)); }

constexpr int kPH = 4, kPW = 32;      // input positions per block
constexpr int kRing = 3;              // weight slots
constexpr int kMaxPatchRows = (kPH + 1) * (kPW + 1);      // a stride-2 3x3 kernel reaches one row and one column
constexpr int kPatchPieces = (kMaxPatchRows + 31) / 32;   // 8-row pieces per wave and chunk (4 waves)

}  // namespace

// NCH: 128-byte chunks of an input pixel row (Ci * sizeof(T) / 128);  COF: 16-channel fragments of the output
// STATS: the block also leaves the per-tile sums of its output (ConvGemmParams::tile_stats, mode 1: shift, sum,
// sum of squares, count per channel; ONE row per block) for the instance norm behind a transposed convolution.
// They are sums of the STORED, rounded tensor (what a separate statistics pass would read), taken from the staged
// rows: in bf16 with two small MFMAs per fragment (ones^T Y and the diagonal of Y^T Y; plain sums are exact
// enough for a tensor that itself has eight bits of mantissa, and each wave moves them to a shift before
// they leave), in fp32 as shifted sums in the lanes that store the rows.
// CLAMP (bf16, one chunk per pixel): ConvGemmParams::clamp_t -- the launch absorbs the instance norm + ReLU in front
// of it: fragments are clamped from below by the sample's t values, rows outside the image hold t, the weights are
// the sample's own scaled copy, and every output class starts from the constant terms of ITS taps
// (clamp_bias [n][4][Co], by class slot 2 (oy & 1) + (ox & 1)).
template <typename T, int NCH, int COF, bool STATS, bool CLAMP>
__device__ __forceinline__ void conv_up2_body(const ConvGemmParams& p) {
    static_assert(!CLAMP || (sizeof(T) == 2 && NCH == 1), "the clamp is for bf16 pixels of one 128-byte chunk");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 8 * EPC;                    // channels per chunk
    constexpr int CO = 16 * COF;
    constexpr int W_SLOT = CO * 128;               // bytes of one (tap, chunk) weight tile
    constexpr int NI_W = W_SLOT / 1024 / 4;        // weight DMA instructions per wave and step
    static_assert(NI_W * 4 * 1024 == W_SLOT, "weight pieces must tile the four waves");
    constexpr int P_CHUNK = 4 * kPatchPieces * 1024;  // bytes of one chunk's patch
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(128))) uint4 lds[];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned patch_base = lds_base + kRing * W_SLOT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Hi, W = p.Wi, C = p.Ci;

    // tile of this block: blocks b and b+8 share an XCD, each XCD gets one contiguous run
    unsigned tile;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    const unsigned tiles_x = p.pt_div[1], tiles_y = p.pt_div[2];
    const unsigned q = tile / tiles_x;
    const int tx0 = (int)(tile - q * tiles_x) * kPW;
    const int n = (int)(q / tiles_y);
    const int ty0 = (int)(q - (unsigned)n * tiles_y) * kPH;
    // halo of the tap table (host): patch row (py, px) holds input (ty0 + dy_min + py, tx0 + dx_min + px)
    const int dy_min = (int)p.pt_magic[0] - 1, dx_min = (int)p.pt_magic[1] - 1;
    const int PWH = kPW + (int)p.pt_magic[2];            // patch width
    const int nrows = (kPH + (int)p.pt_div[0]) * PWH;    // patch rows

    // ---------------------------------------------------------------- loads: the whole patch, two weight tiles
    const int sub = lane >> 3, pos = lane & 7;
    {
        const T* in = (const T*)p.in;
        const T* zero_src = (const T*)(g_up2_zero_page + pos);
        // CLAMP: outside the image the sample's clamp values (this lane's piece of the chunk: pr & 7 == sub)
        const T* t_src = CLAMP ? (const T*)p.clamp_t + (long)n * C + (pos ^ sub) * EPC : nullptr;
#pragma unroll
        for (int i = 0; i < kPatchPieces; i++) {
            const int pr = (wave * kPatchPieces + i) * 8 + sub;
            const int py = pr / PWH, px = pr - py * PWH;
            const int iy = ty0 + dy_min + py, ix = tx0 + dx_min + px;
            const bool ok = pr < nrows && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const T* src = in + ((long)(n * H + iy) * W + ix) * C + (pos ^ (pr & 7)) * EPC;
#pragma unroll
            for (int c = 0; c < NCH; c++)
                glds16(ok ? src + c * BK : (CLAMP ? t_src + c * BK : zero_src), patch_base + c * P_CHUNK + (wave * kPatchPieces + i) * 1024);
        }
    }
    // weight tile of step s = (tap index in the table, chunk): rows CO x 128 bytes, swizzled like the consumers read
    const T* wp = (const T*)p.wp + (CLAMP ? (long)n * p.w_batch_stride : 0L);
    int wrow_off[NI_W];
#pragma unroll
    for (int j = 0; j < NI_W; j++) {
        const int row = (wave * NI_W + j) * 8 + sub;
        wrow_off[j] = row * p.kpad + (pos ^ ((row >> 1) & 7)) * EPC;
    }
    const int nsteps = p.ntaps * NCH;    // ntaps: taps of all classes together
    auto stage_weights = [&](int s) {
        const int t = s / NCH, c = s - t * NCH;
        const Tap tap = p.taps[t];
        const T* wsrc = wp + (long)tap.widx * p.co_pad * p.kpad + c * BK;
        const unsigned dst = lds_base + (s % kRing) * W_SLOT + wave * (NI_W * 1024);
#pragma unroll
        for (int j = 0; j < NI_W; j++) glds16(wsrc + wrow_off[j], dst + j * 1024);   // the same count in every wave
    };
    stage_weights(0);
    if (nsteps > 1) stage_weights(1);

    // ---------------------------------------------------------------- K loop
    const int g = lane >> 4, r16 = lane & 15;
    f32x4 acc[4][COF][2];
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
        for (int a = 0; a < COF; a++) {
            f32x4 start = {0.f, 0.f, 0.f, 0.f};
            // CLAMP: a class starts from the constant terms of its own taps (C/D rows 4 g + u of channel fragment a)
            if constexpr (CLAMP) start = *reinterpret_cast<const f32x4*>(p.clamp_bias + ((long)n * 4 + c) * p.Co + a * 16 + 4 * g);
#pragma unroll
            for (int b = 0; b < 2; b++) acc[c][a][b] = start;
        }
    // CLAMP: the eight clamp values of this lane's fragment registers, per 64-byte half of the chunk
    u32x4 tv[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};
    if constexpr (CLAMP) {
        const T* t_n = (const T*)p.clamp_t + (long)n * C;
        const uint4 t0 = *reinterpret_cast<const uint4*>(t_n + 8 * g);
        const uint4 t1 = *reinterpret_cast<const uint4*>(t_n + 32 + 8 * g);
        tv[0] = (u32x4){t0.x, t0.y, t0.z, t0.w};
        tv[1] = (u32x4){t1.x, t1.y, t1.z, t1.w};
    }

    const unsigned a_off = lds_base + r16 * 128 + ((g ^ ((r16 >> 1) & 7)) << 4);
    // patch row of this lane's two pixel columns under a tap with offset (0, 0)
    const int pr0 = (wave - dy_min) * PWH + r16 - dx_min;

    int s = 0;
    auto class_steps = [&](auto cls_tag, int tap0, int ntaps) __attribute__((always_inline)) {
        constexpr int CLS = decltype(cls_tag)::value;
        for (int t = tap0; t < tap0 + ntaps; t++) {
            const Tap tap = p.taps[t];
            const int prt = pr0 + tap.dy * PWH + tap.dx;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                // weights of step s have landed (those of step s + 1 may still be in flight) ...
                if (s + 1 < nsteps) wait_vmcnt<NI_W>();
                else wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();   // ... in every wave's share; the slot of step s - 1 is drained
                asm volatile("" ::: "memory");
                if (s + 2 < nsteps) stage_weights(s + 2);
                const unsigned wbase = a_off + (s % kRing) * W_SLOT;
                const unsigned pbuf = patch_base + c * P_CHUNK;
                u32x4 fa[2][COF], fb[2][2];
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int k = 0; k < COF; k++) fa[h][k] = lds_read16((wbase + k * 2048) ^ (h ? 64u : 0u));
#pragma unroll
                    for (int f = 0; f < 2; f++) {
                        const int pr = prt + f * 16;
                        fb[h][f] = lds_read16((pbuf + pr * 128 + ((g ^ (pr & 7)) << 4)) ^ (h ? 64u : 0u));
                    }
                }
                lds_wait();
#pragma unroll
                for (int h = 0; h < 2; h++) {
#pragma unroll
                    for (int k = 0; k < COF; k++) lds_landed(fa[h][k]);
#pragma unroll
                    for (int f = 0; f < 2; f++) lds_landed(fb[h][f]);
                }
                if constexpr (CLAMP) {
#pragma unroll
                    for (int h = 0; h < 2; h++)
#pragma unroll
                        for (int f = 0; f < 2; f++)
#pragma unroll
                            for (int u = 0; u < 4; u++) {
                                unsigned r;
                                asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(fb[h][f][u]), "v"(tv[h][u]));
                                fb[h][f][u] = r;
                            }
                }
#pragma unroll
                for (int h = 0; h < 2; h++)
#pragma unroll
                    for (int k = 0; k < COF; k++)
#pragma unroll
                        for (int f = 0; f < 2; f++)
                            Mma<T>::mma(__builtin_bit_cast(frag, fa[h][k]), __builtin_bit_cast(frag, fb[h][f]), acc[CLS][k][f]);
                s++;
            }
        }
    };
    // class slot = 2 * (oy0 & 1) + (ox0 & 1) (host): a class that does not exist has no taps
    class_steps(std::integral_constant<int, 0>{}, p.cls[0].tap0, p.cls[0].ntaps);
    class_steps(std::integral_constant<int, 1>{}, p.cls[1].tap0, p.cls[1].ntaps);
    class_steps(std::integral_constant<int, 2>{}, p.cls[2].tap0, p.cls[2].ntaps);
    class_steps(std::integral_constant<int, 3>{}, p.cls[3].tap0, p.cls[3].ntaps);

    // ---------------------------------------------------------------- epilogue
    // Wave w: input row ty0 + w -> output rows 2 (ty0 + w) + cy, 64 output pixels 2 (tx0 + j) + cx each.
    // One output row at a time goes through the wave's own staging rows ([64 pixels][CO] of T, +16 bytes a
    // row), then out as 16 bytes per lane along whole pixel rows.
    constexpr int SROW = CO * (int)sizeof(T) + 16;        // bytes
    constexpr int LPR = CO / EPC;                         // lanes per pixel row
    constexpr bool MSTATS = STATS && sizeof(T) == 2;      // bf16: the sums run on the matrix cores
    constexpr bool VSTATS = STATS && sizeof(T) == 4;      // fp32: shifted sums in the lanes that store
    __syncthreads();                                      // every wave is done with the patch and the ring
    char* stage = reinterpret_cast<char*>(lds) + wave * (64 * SROW);
    T* out = (T*)p.out;
    const int iy = ty0 + wave;
    const int lp = lane / LPR, lc = (lane % LPR) * EPC;
    const int nvalid_px = min(64, p.Wo - 2 * tx0);        // >= 1: the tile exists
    WaveStatsLanes<T, CO> vs;                             // VSTATS; shift: the wave's first pixel
    vs.init();
    f32x4 acc_s[COF], acc_q[COF];                         // MSTATS (wave_stats_mma)
    float k_first[COF];                                   // ... shift: the wave's first stored pixel, channel 16 k + r16
#pragma unroll
    for (int k = 0; k < COF; k++) {
        acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        k_first[k] = 0.f;
    }
    int rows_valid = 0;
#pragma unroll
    for (int cy = 0; cy < 2; cy++) {
        const int oy = 2 * iy + cy;
#pragma unroll
        for (int cx = 0; cx < 2; cx++)
#pragma unroll
            for (int f = 0; f < 2; f++)
#pragma unroll
                for (int k = 0; k < COF; k++) {
                    float v[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = acc[2 * cy + cx][k][f][u] * p.alpha;
                    store_vec<T, 4>(reinterpret_cast<T*>(stage + (2 * (f * 16 + r16) + cx) * SROW) + k * 16 + 4 * g, v);
                }
        lds_wait();   // the wave's own rows: no barrier
        asm volatile("" ::: "memory");
        if constexpr (VSTATS) {
            if (cy == 0) load_vec<T, EPC>(reinterpret_cast<const T*>(stage + lc * (int)sizeof(T)), vs.sk);   // pixel 0 of the row
        }
        if (oy < p.Ho) {
            T* orow = out + ((long)(n * p.Ho + oy) * p.Wo) * p.Co;
#pragma unroll
            for (int i = 0; i < LPR; i++) {
                const int px = lp + i * (64 / LPR);
                const int ox = 2 * tx0 + px;
                if (ox < p.Wo) {
                    const uint4 v = *reinterpret_cast<const uint4*>(stage + px * SROW + lc * (int)sizeof(T));
                    *reinterpret_cast<uint4*>(orow + (long)ox * p.Co + lc) = v;
                    if constexpr (VSTATS) vs.add(v);
                }
            }
            if constexpr (MSTATS) {
                if (cy == 0) {
#pragma unroll
                    for (int k = 0; k < COF; k++) k_first[k] = to_f32<T>(reinterpret_cast<const T*>(stage)[k * 16 + r16]);
                }
                // pixels past the right edge were computed but are not part of the tensor: their rows become zeros
                if (nvalid_px < 64 && lane >= nvalid_px) {
#pragma unroll
                    for (int c = 0; c < CO * (int)sizeof(T); c += 16)
                        *reinterpret_cast<uint4*>(stage + lane * SROW + c) = make_uint4(0u, 0u, 0u, 0u);
                }
                asm volatile("" ::: "memory");
                wave_stats_mma<CO>(stage, SROW, lane, acc_s, acc_q);
                rows_valid++;
            }
        }
        lds_wait();
        asm volatile("" ::: "memory");
    }
    if constexpr (STATS) {
        float* part = reinterpret_cast<float*>(lds);      // [wave][plane k, s1, s2, count][CO]
        if constexpr (VSTATS) vs.reduce();
        lds_barrier();                                    // every wave is done with its staging rows
        if constexpr (VSTATS) vs.write_part(part + wave * 4 * CO, lane);
        else wave_stats_part_mma<CO>(part + wave * 4 * CO, lane, acc_s, acc_q, k_first, (float)(rows_valid * nvalid_px));
        lds_barrier();
        // wave 0's first pixel (2 ty0, 2 tx0) is always inside the image
        const unsigned tile_in_sample = tile - (unsigned)n * tiles_x * tiles_y;
        block_stats_row<CO>(part, tid, p.tile_stats + (((long)n * p.tile_stats_ts + tile_in_sample) * 4) * p.Co, p.Co);
    }
}

template <typename T, int NCH, int COF, bool STATS, bool CLAMP = false>
__global__ __launch_bounds__(256) void conv_up2_kernel(const ConvGemmParams p) {
    conv_up2_body<T, NCH, COF, STATS, CLAMP>(p);
}

// The absorbing form is bound by its stores like the plain one and wants the same four blocks per CU: the clamp
// values and pointers would otherwise push it over 128 registers.
template <typename T, int NCH, int COF>
__global__ __launch_bounds__(256, 4) void conv_up2_clamp_kernel(const ConvGemmParams p) {
    conv_up2_body<T, NCH, COF, true, true>(p);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, int NCH, int COF, bool STATS, bool CLAMP = false>
static int launch_up2(const ConvGemmParams& p, hipStream_t s) {
    constexpr size_t kOperands = (size_t)kRing * 16 * COF * 128 + (size_t)NCH * 4 * kPatchPieces * 1024;
    constexpr size_t kStage = (size_t)4 * 64 * (16 * COF * sizeof(T) + 16);
    constexpr size_t kLds = kOperands > kStage ? kOperands : kStage;
    if constexpr (CLAMP) {
        STMI_DYN_LDS((conv_up2_clamp_kernel<T, NCH, COF>), (int)kLds, "hipFuncSetAttribute(conv_up2_clamp_kernel)");
        hipLaunchKernelGGL((conv_up2_clamp_kernel<T, NCH, COF>), dim3(p.pt_div[1] * p.pt_div[2] * (unsigned)p.N), dim3(256), kLds, s, p);
    } else {
        STMI_DYN_LDS((conv_up2_kernel<T, NCH, COF, STATS, CLAMP>), (int)kLds, "hipFuncSetAttribute(conv_up2_kernel)");
        hipLaunchKernelGGL((conv_up2_kernel<T, NCH, COF, STATS, CLAMP>), dim3(p.pt_div[1] * p.pt_div[2] * (unsigned)p.N), dim3(256), kLds, s, p);
    }
    STMI_LAUNCH_CHECK("conv_up2_kernel");
    return STMI_OK;
}

template <typename T, int NCH, int COF>
static int launch_up2_s(const ConvGemmParams& p, hipStream_t s) {
    return p.tile_stats ? launch_up2<T, NCH, COF, true>(p, s) : launch_up2<T, NCH, COF, false>(p, s);
}

// p: the multi-class form of a stride-2 backward-data launch (nclasses sub-grids with oys = oxs = 2, in_stride 1).
// Fills q, the launch form of p (halo and grid in the pt_ fields); false: the layer is not one of this kernel's.
static bool up2_prepare(const ConvGemmParams& p, int dtype, ConvGemmParams& q) {
    if (STMI_ENV_SET("STMI_NO_UP2")) return false;
    const int es = dtype == STMI_BF16 ? 2 : 4;
    if (p.nclasses < 1 || p.nclasses > 4 || p.oys != 2 || p.oxs != 2 || p.in_stride != 1 || p.cdiv != 0) return false;
    if (p.klen != p.Ci || (p.Ci * es) % 128 != 0 || (p.Co != 32 && p.Co != 64) || p.co_pad < p.Co) return false;
    const int nch = p.Ci * es / 128;
    if (nch != 1 && nch != 2 && nch != 4) return false;
    if (p.bias || p.mask || p.addend || p.out_pre || p.relu || (p.w_batch_stride != 0 && !p.clamp_t)) return false;
    // the absorbed-norm form: bf16, pixels of one chunk, with the statistics the next norm wants
    if (p.clamp_t && (dtype != STMI_BF16 || p.Ci * es != 128 || !p.tile_stats || p.tile_stats_mode != 1 || p.w_batch_stride <= 0 ||
                      !p.clamp_bias || (((uintptr_t)p.clamp_t | (uintptr_t)p.clamp_bias) & 15) != 0))
        return false;
    if (p.tile_stats && p.tile_stats_mode != 1) return false;
    if ((((uintptr_t)p.in | (uintptr_t)p.wp | (uintptr_t)p.out) & 15) != 0) return false;
    if ((long)p.N * p.Hi * p.Wi * p.Ci >= (1L << 31) || (long)p.N * p.Ho * p.Wo * p.Co >= (1L << 31)) return false;
    q = p;
    int dy_min = 0, dy_max = 0, dx_min = 0, dx_max = 0, hs = 0, ws = 0, nt = 0;
    ConvGemmParams::SubGrid slots[4] = {};
    for (int c = 0; c < p.nclasses; c++) {
        const ConvGemmParams::SubGrid& g = p.cls[c];
        if (g.oy0 < 0 || g.oy0 > 1 || g.ox0 < 0 || g.ox0 > 1) return false;
        ConvGemmParams::SubGrid& d = slots[2 * g.oy0 + g.ox0];
        if (d.ntaps) return false;   // two classes of one parity
        d = g;
        hs = g.Hs > hs ? g.Hs : hs;
        ws = g.Ws > ws ? g.Ws : ws;
        for (int t = g.tap0; t < g.tap0 + g.ntaps; t++) {
            dy_min = p.taps[t].dy < dy_min ? p.taps[t].dy : dy_min;
            dy_max = p.taps[t].dy > dy_max ? p.taps[t].dy : dy_max;
            dx_min = p.taps[t].dx < dx_min ? p.taps[t].dx : dx_min;
            dx_max = p.taps[t].dx > dx_max ? p.taps[t].dx : dx_max;
        }
        nt += g.ntaps;
    }
    if (dy_min < -1 || dy_max > 1 || dx_min < -1 || dx_max > 1 || dy_max - dy_min > 1 || dx_max - dx_min > 1 || nt < 1) return false;
    // every output pixel belongs to a class position inside the input-position grid the tiles cover
    if (2 * hs < p.Ho || 2 * ws < p.Wo) return false;
    // the kernel walks the tap table front to back: classes in slot order, their taps consecutive
    for (int c = 0, at = 0; c < 4; c++) {
        if (slots[c].ntaps && slots[c].tap0 != at) return false;
        at += slots[c].ntaps;
        q.cls[c] = slots[c];
    }
    q.ntaps = nt;
    q.pt_div[0] = (unsigned)(dy_max - dy_min);
    q.pt_div[1] = (unsigned)ceil_div(ws, kPW);
    q.pt_div[2] = (unsigned)ceil_div(hs, kPH);
    q.pt_magic[0] = (unsigned)(dy_min + 1);
    q.pt_magic[1] = (unsigned)(dx_min + 1);
    q.pt_magic[2] = (unsigned)(dx_max - dx_min);
    if ((long)q.pt_div[1] * q.pt_div[2] * p.N >= (1L << 31)) return false;
    return true;
}

// Statistics rows per sample this kernel leaves for the launch p (one per block), 0 when p is not its layer.
int conv_up2_tile_stats_tiles(const ConvGemmParams& p_in, int dtype) {
    ConvGemmParams q;
    ConvGemmParams p = p_in;   // a question about the geometry only
    p.clamp_t = nullptr;
    p.w_batch_stride = 0;
    if (!up2_prepare(p, dtype, q)) return 0;
    return (int)(q.pt_div[1] * q.pt_div[2]);
}

bool try_launch_conv_up2(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len) {
    ConvGemmParams q;
    if (!up2_prepare(p, dtype, q)) return false;
    if (p.tile_stats && p.tile_stats_ts != (int)(q.pt_div[1] * q.pt_div[2])) return false;
    const int es = dtype == STMI_BF16 ? 2 : 4;
    const int nch = p.Ci * es / 128;
    const int cof = p.Co / 16;
    if (name)
        snprintf(name, name_len, p.clamp_t ? "void stmi::conv_up2_clamp_kernel<%s, %d, %d>(stmi::ConvGemmParams)"
                                           : "void stmi::conv_up2_kernel<%s, %d, %d, %s, false>(stmi::ConvGemmParams)",
                 dtype == STMI_BF16 ? "__bf16" : "float", nch, cof, p.tile_stats ? "true" : "false");
    if (p.clamp_t) {
        *rc = cof == 2 ? launch_up2<bf16_t, 1, 2, true, true>(q, s) : launch_up2<bf16_t, 1, 4, true, true>(q, s);
        return true;
    }
    if (dtype == STMI_BF16) {
        if (nch == 1) *rc = cof == 2 ? launch_up2_s<bf16_t, 1, 2>(q, s) : launch_up2_s<bf16_t, 1, 4>(q, s);
        else if (nch == 2) *rc = cof == 2 ? launch_up2_s<bf16_t, 2, 2>(q, s) : launch_up2_s<bf16_t, 2, 4>(q, s);
        else *rc = cof == 2 ? launch_up2_s<bf16_t, 4, 2>(q, s) : launch_up2_s<bf16_t, 4, 4>(q, s);
    } else {
        if (nch == 1) *rc = cof == 2 ? launch_up2_s<float, 1, 2>(q, s) : launch_up2_s<float, 1, 4>(q, s);
        else if (nch == 2) *rc = cof == 2 ? launch_up2_s<float, 2, 2>(q, s) : launch_up2_s<float, 2, 4>(q, s);
        else *rc = cof == 2 ? launch_up2_s<float, 4, 2>(q, s) : launch_up2_s<float, 4, 4>(q, s);
    }
    return true;
}

}  // namespace stmi
