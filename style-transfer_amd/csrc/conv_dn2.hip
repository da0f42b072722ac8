// Stride-2 3x3 forward convolution (the stylizer's two down-sampling layers, 32 -> 64 and 64 -> 128 channels,
// model_utilities.py:236-237 with strides [1,2,2,1]) with the input patch held in LDS (gfx950).
//
// On even image sides 'SAME' pads below and to the right only, so output (oy, ox) reads input rows 2 oy .. 2 oy + 2
// and columns 2 ox .. 2 ox + 2.  Seen as PAIRS of horizontally adjacent pixels (pair j = columns 2 j, 2 j + 1:
// 2 Ci contiguous channels in NHWC) the column stride disappears: output ox reads all of pair ox (kx = 0, 1) and the
// first half of pair ox + 1 (kx = 2).  A block owns 4 x PW output pixels, keeps the 9 x (PW + 1) pair patch (all
// channels) in LDS for its whole life, and runs per kernel row ky the K steps [pair ox | first half of pair ox + 1]
// over it with the weight tiles streamed through a small ring: the input is fetched about 1.16 times where the
// gather-GEMM form fetches every pixel row once per tap it serves (2.25 times), and every K step is a dense
// 16-byte-per-lane LDS read (64-byte pixel rows feed the gather-GEMM half-empty).
//
// 256 threads, wave w = output row w of the tile (PW pixels = PW / 16 fragments) x all output channels; every wave
// both issues the DMA pieces (global_load_lds_dwordx4, source-side XOR swizzle, a page of zeros behind
// out-of-image rows) and multiplies.  The packed weights are the plain forward pack ([tap][co][ci]): a weight tile
// row is put together from the taps (ky, 0) and (ky, 1) piece by piece when a pixel has only 64 bytes of channels.
#include "conv_gemm.hpp"

#include <type_traits>

namespace stmi {

__device__ uint4 g_dn2_zero_page[8];  // zero-initialised: source of out-of-image rows

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

constexpr int kPH = 4;                // output rows per block (one per wave)
constexpr int kRing = 3;              // weight slots

template <int F, int L, typename Fn>
__device__ __forceinline__ void static_for(Fn&& fn) {
    if constexpr (F < L) {
        fn(std::integral_constant<int, F>{});
        static_for<F + 1, L>(fn);
    }
}

}  // namespace

// NCH: 128-byte chunks of an input PAIR row (2 Ci sizeof(T) / 128);  COF: 16-channel fragments of the output;
// PW: output pixels per tile row.  STATS: the block also leaves the per-tile sums of its output
// (ConvGemmParams::tile_stats, mode 1; ONE row per block) for the instance norm behind the layer, taken from the
// staged rows as in conv_up2_kernel.
// CLAMP (bf16, at most two chunks per pair): ConvGemmParams::clamp_t -- the launch absorbs the instance norm + ReLU
// in front of it: pixel fragments are clamped from below by the sample's t values (one packed maximum per register),
// rows outside the image hold t, and the weights are the sample's own (scaled) copy.
template <typename T, int NCH, int COF, int PW, bool STATS, bool CLAMP = false>
__global__ __launch_bounds__(256) void conv_dn2_kernel(const ConvGemmParams p) {
    static_assert(!CLAMP || (sizeof(T) == 2 && NCH <= 2), "the clamp is for bf16 pixels of 64 or 128 bytes");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 8 * EPC;                    // elements per 128-byte chunk
    constexpr int CO = 16 * COF;
    constexpr int NF = PW / 16;                    // pixel fragments per wave
    constexpr int PWH = PW + 1;                    // pair columns of the patch
    constexpr int NROWS = (2 * kPH + 1) * PWH;     // pair rows of the patch
    constexpr int PIECES = (NROWS + 31) / 32;      // 8-row pieces per wave and chunk
    constexpr int P_CHUNK = 4 * PIECES * 1024;     // bytes of one chunk's patch
    constexpr int W_SLOT = CO * 128;               // bytes of one weight tile
    constexpr int NI_W = W_SLOT / 1024 / 4;        // weight DMA instructions per wave and step
    static_assert(NI_W * 4 * 1024 == W_SLOT, "weight pieces must tile the four waves");
    // K steps of one kernel row: the NCH chunks of pair ox (kx = 0, 1), then the first half of pair ox + 1 (kx = 2):
    // NCH / 2 chunks, or the first 64 bytes of the only chunk
    constexpr int HSTEPS = NCH >= 2 ? NCH / 2 : 1;
    constexpr int KY_STEPS = NCH + HSTEPS;
    constexpr int NS = 3 * KY_STEPS;
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
    const int tx0 = (int)(tile - q * tiles_x) * PW;
    const int n = (int)(q / tiles_y);
    const int ty0 = (int)(q - (unsigned)n * tiles_y) * kPH;

    // ---------------------------------------------------------------- loads: the whole patch, two weight tiles
    const int sub = lane >> 3, pos = lane & 7;
    {
        const T* in = (const T*)p.in;
        const T* zero_src = (const T*)(g_dn2_zero_page + pos);
        // CLAMP: the sample's clamp values stand in for rows outside the image; this lane's 16-byte piece of chunk c
        // of a pair row holds the channels (c BK + piece * EPC) mod C of one of the pair's pixels
        const T* t_n = CLAMP ? (const T*)p.clamp_t + (long)n * C : nullptr;
        const int sp_e = (pos ^ sub) * EPC;   // pr & 7 == sub for every piece
        const int pairs = W >> 1;
#pragma unroll
        for (int i = 0; i < PIECES; i++) {
            const int pr = (wave * PIECES + i) * 8 + sub;
            const int py = pr / PWH, px = pr - py * PWH;
            const int iy = 2 * ty0 + py, pc = tx0 + px;
            const bool ok = pr < NROWS && iy < H && pc < pairs;
            const T* src = in + ((long)(n * H + iy) * W + 2 * pc) * C + (pos ^ (pr & 7)) * EPC;
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const T* outside = CLAMP ? t_n + (c * BK + sp_e) % C : zero_src;
                glds16(ok ? src + c * BK : outside, patch_base + c * P_CHUNK + (wave * PIECES + i) * 1024);
            }
        }
    }
    // Weight tile of a step: rows CO x 128 bytes, swizzled like the consumers read.  The 16-byte piece a lane
    // fetches sits at source piece sp of its row: with one chunk per pair (NCH == 1) pieces 0 .. 3 are the channels
    // of tap (ky, 0) or (ky, 2) and pieces 4 .. 7 those of tap (ky, 1) (unused behind kx = 2).
    const T* wp = (const T*)p.wp + (CLAMP ? (long)n * p.w_batch_stride : 0L);
    int wrow_off[NI_W];
    bool wrow_hi[NI_W];
#pragma unroll
    for (int j = 0; j < NI_W; j++) {
        const int row = (wave * NI_W + j) * 8 + sub;
        const int sp = pos ^ ((row >> 1) & 7);
        wrow_hi[j] = NCH == 1 && sp >= 4;
        wrow_off[j] = row * p.kpad + (NCH == 1 ? (sp & 3) : sp) * EPC;
    }
    const long tap_stride = (long)p.co_pad * p.kpad;
    auto stage_weights = [&](auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        constexpr int KY = S / KY_STEPS, R = S % KY_STEPS;
        constexpr bool HALF = R >= NCH;               // the kx = 2 part
        constexpr int CH = HALF ? R - NCH : R;        // chunk of the pair row
        // channel chunk and kernel column the tile's low pieces come from
        constexpr int KX = HALF ? 2 : (NCH >= 2 ? CH / HSTEPS : 0);
        constexpr int CC = NCH >= 2 ? CH % HSTEPS : 0;
        const T* w0 = wp + (long)(KY * 3 + KX) * tap_stride + CC * BK;
        const unsigned dst = lds_base + (S % kRing) * W_SLOT + wave * (NI_W * 1024);
#pragma unroll
        for (int j = 0; j < NI_W; j++)
            glds16(w0 + wrow_off[j] + ((wrow_hi[j] && !HALF) ? tap_stride : 0), dst + j * 1024);   // the same count in every wave
    };
    stage_weights(std::integral_constant<int, 0>{});
    stage_weights(std::integral_constant<int, 1>{});

    // ---------------------------------------------------------------- K loop
    const int g = lane >> 4, r16 = lane & 15;
    f32x4 acc[COF][NF];
#pragma unroll
    for (int a = 0; a < COF; a++) {
        f32x4 start = {0.f, 0.f, 0.f, 0.f};
        if constexpr (CLAMP) start = *reinterpret_cast<const f32x4*>(p.clamp_bias + (long)n * p.Co + a * 16 + 4 * g);   // C/D rows 4 g + u
#pragma unroll
        for (int b = 0; b < NF; b++) acc[a][b] = start;
    }

    // CLAMP: the eight clamp values of this lane's fragment registers, per 64-byte half of a chunk: a half is 32
    // channels of ONE pixel of the pair (all of its 32 channels, or half h of its 64), whatever the chunk
    u32x4 tv[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};
    if constexpr (CLAMP) {
        const T* t_n = (const T*)p.clamp_t + (long)n * C;
        const uint4 t0 = *reinterpret_cast<const uint4*>(t_n + 8 * g);
        const uint4 t1 = *reinterpret_cast<const uint4*>(t_n + (NCH == 1 ? 0 : 32) + 8 * g);
        tv[0] = (u32x4){t0.x, t0.y, t0.z, t0.w};
        tv[1] = (u32x4){t1.x, t1.y, t1.z, t1.w};
    }
    const unsigned a_off = lds_base + r16 * 128 + ((g ^ ((r16 >> 1) & 7)) << 4);
    // pair row of this lane's first pixel column under kernel row 0, pair offset 0
    const int pr0 = (2 * wave) * PWH + r16;

    static_for<0, NS>([&](auto s_tag) {
        constexpr int S = decltype(s_tag)::value;
        constexpr int KY = S / KY_STEPS, R = S % KY_STEPS;
        constexpr bool HALF = R >= NCH;
        constexpr int CH = HALF ? R - NCH : R;
        constexpr int NH = (NCH == 1 && HALF) ? 1 : 2;   // 64-byte halves of the chunk that carry channels
        // weights of step S have landed (those of step S + 1 may still be in flight) ...
        if constexpr (S + 1 < NS) wait_vmcnt<NI_W>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();   // ... in every wave's share; the slot of step S - 1 is drained
        asm volatile("" ::: "memory");
        if constexpr (S + 2 < NS) stage_weights(std::integral_constant<int, S + 2>{});
        const unsigned wbase = a_off + (S % kRing) * W_SLOT;
        const unsigned pbuf = patch_base + CH * P_CHUNK;
        const int prt = pr0 + KY * PWH + (HALF ? 1 : 0);
        u32x4 fa[NH][COF], fb[NH][NF];
#pragma unroll
        for (int h = 0; h < NH; h++) {
#pragma unroll
            for (int k = 0; k < COF; k++) fa[h][k] = lds_read16((wbase + k * 2048) ^ (h ? 64u : 0u));
#pragma unroll
            for (int f = 0; f < NF; f++) {
                const int pr = prt + f * 16;
                fb[h][f] = lds_read16((pbuf + pr * 128 + ((g ^ (pr & 7)) << 4)) ^ (h ? 64u : 0u));
            }
        }
        lds_wait();
#pragma unroll
        for (int h = 0; h < NH; h++) {
#pragma unroll
            for (int k = 0; k < COF; k++) lds_landed(fa[h][k]);
#pragma unroll
            for (int f = 0; f < NF; f++) lds_landed(fb[h][f]);
        }
        if constexpr (CLAMP) {
            // a pixel fragment is clamped behind the first product of the fragment before it: all but the first
            // four packed maxima of a step run under matrix work
            auto clamp = [&](int h, int f) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    unsigned r;
                    asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(fb[h][f][u]), "v"(tv[h][u]));
                    fb[h][f][u] = r;
                }
            };
            clamp(0, 0);
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int f = 0; f < NF; f++)
#pragma unroll
                    for (int k = 0; k < COF; k++) {
                        Mma<T>::mma(__builtin_bit_cast(frag, fa[h][k]), __builtin_bit_cast(frag, fb[h][f]), acc[k][f]);
                        if (k == 0) {
                            if (f + 1 < NF) clamp(h, f + 1);
                            else if (h + 1 < NH) clamp(h + 1, 0);
                        }
                    }
        } else {
#pragma unroll
            for (int h = 0; h < NH; h++)
#pragma unroll
                for (int k = 0; k < COF; k++)
#pragma unroll
                    for (int f = 0; f < NF; f++)
                        Mma<T>::mma(__builtin_bit_cast(frag, fa[h][k]), __builtin_bit_cast(frag, fb[h][f]), acc[k][f]);
        }
    });

    // ---------------------------------------------------------------- epilogue
    // Wave w: output row ty0 + w, PW pixels: through the wave's own staging rows ([pixels][CO] of T, +16 bytes a
    // row), then out as 16 bytes per lane along whole pixel rows.
    constexpr int SROW = CO * (int)sizeof(T) + 16;        // bytes
    constexpr int LPR = CO / EPC;                         // lanes per pixel row
    constexpr int SPX = STATS && sizeof(T) == 2 ? 32 : PW;   // staged rows per wave (the matrix-core sums read 32)
    constexpr bool MSTATS = STATS && sizeof(T) == 2;      // bf16: the sums run on the matrix cores
    constexpr bool VSTATS = STATS && sizeof(T) == 4;      // fp32: shifted sums in the lanes that store
    static_assert(!MSTATS || PW == 32, "the matrix-core sums take 32 staged pixels");
    __syncthreads();                                      // every wave is done with the patch and the ring
    char* stage = reinterpret_cast<char*>(lds) + wave * (SPX * SROW);
    T* out = (T*)p.out;
    const int oy = ty0 + wave;
    const int lp = lane / LPR, lc = (lane % LPR) * EPC;
    const int nvalid_px = min(PW, p.Wo - tx0);            // >= 1: the tile exists
    WaveStatsLanes<T, CO> vs;                             // VSTATS; shift: the wave's first pixel
    vs.init();
    f32x4 acc_s[COF], acc_q[COF];                         // MSTATS (wave_stats_mma)
    float k_first[COF];                                   // ... shift: the wave's first stored pixel, channel 16 k + r16
#pragma unroll
    for (int k = 0; k < COF; k++) {
        acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        k_first[k] = 0.f;
    }
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int k = 0; k < COF; k++) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = acc[k][f][u] * p.alpha;
            store_vec<T, 4>(reinterpret_cast<T*>(stage + (f * 16 + r16) * SROW) + k * 16 + 4 * g, v);
        }
    lds_wait();   // the wave's own rows: no barrier
    asm volatile("" ::: "memory");
    if constexpr (VSTATS) load_vec<T, EPC>(reinterpret_cast<const T*>(stage + lc * (int)sizeof(T)), vs.sk);   // pixel 0 of the row
    const bool row_ok = oy < p.Ho;
    if (row_ok) {
        T* orow = out + ((long)(n * p.Ho + oy) * p.Wo) * p.Co;
        constexpr int PPI = 64 / LPR;                     // pixels per pass of the wave
#pragma unroll
        for (int i = 0; i < (PW + PPI - 1) / PPI; i++) {
            const int px = lp + i * PPI;
            const int ox = tx0 + px;
            if (px < PW && ox < p.Wo) {
                const uint4 v = *reinterpret_cast<const uint4*>(stage + px * SROW + lc * (int)sizeof(T));
                *reinterpret_cast<uint4*>(orow + (long)ox * p.Co + lc) = v;
                if constexpr (VSTATS) vs.add(v);
            }
        }
        if constexpr (MSTATS) {
#pragma unroll
            for (int k = 0; k < COF; k++) k_first[k] = to_f32<T>(reinterpret_cast<const T*>(stage)[k * 16 + r16]);
            // pixels past the right edge were computed but are not part of the tensor: their rows become zeros
            if (nvalid_px < 32 && lane >= nvalid_px && lane < 32) {
#pragma unroll
                for (int c = 0; c < CO * (int)sizeof(T); c += 16)
                    *reinterpret_cast<uint4*>(stage + lane * SROW + c) = make_uint4(0u, 0u, 0u, 0u);
            }
            lds_wait();
            asm volatile("" ::: "memory");
            wave_stats_mma<CO, 1>(stage, SROW, lane, acc_s, acc_q);
        }
    }
    if constexpr (STATS) {
        float* part = reinterpret_cast<float*>(lds);      // [wave][plane k, s1, s2, count][CO]
        if constexpr (VSTATS) vs.reduce();
        lds_wait();
        lds_barrier();                                    // every wave is done with its staging rows
        if constexpr (VSTATS) vs.write_part(part + wave * 4 * CO, lane);
        else wave_stats_part_mma<CO>(part + wave * 4 * CO, lane, acc_s, acc_q, k_first, row_ok ? (float)nvalid_px : 0.f);
        lds_barrier();
        // wave 0's first pixel (ty0, tx0) is always inside the image
        const unsigned tile_in_sample = tile - (unsigned)n * tiles_x * tiles_y;
        block_stats_row<CO>(part, tid, p.tile_stats + (((long)n * p.tile_stats_ts + tile_in_sample) * 4) * p.Co, p.Co);
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, int NCH, int COF, int PW, bool STATS, bool CLAMP = false>
static int launch_dn2(const ConvGemmParams& p, hipStream_t s) {
    constexpr int PIECES = ((2 * kPH + 1) * (PW + 1) + 31) / 32;
    constexpr size_t kOperands = (size_t)kRing * 16 * COF * 128 + (size_t)NCH * 4 * PIECES * 1024;
    constexpr size_t kStage = (size_t)4 * (STATS && sizeof(T) == 2 ? 32 : PW) * (16 * COF * sizeof(T) + 16);
    constexpr size_t kPart = (size_t)4 * 4 * 16 * COF * 4;
    constexpr size_t kLds = kOperands > kStage ? (kOperands > kPart ? kOperands : kPart) : (kStage > kPart ? kStage : kPart);
    static_assert(kLds <= 160 * 1024, "patch + weight ring exceed the CU's LDS");
    STMI_DYN_LDS((conv_dn2_kernel<T, NCH, COF, PW, STATS, CLAMP>), (int)kLds, "hipFuncSetAttribute(conv_dn2_kernel)");
    hipLaunchKernelGGL((conv_dn2_kernel<T, NCH, COF, PW, STATS, CLAMP>), dim3(p.pt_div[1] * p.pt_div[2] * (unsigned)p.N), dim3(256),
                       kLds, s, p);
    STMI_LAUNCH_CHECK("conv_dn2_kernel");
    return STMI_OK;
}

template <typename T, int NCH, int COF, int PW>
static int launch_dn2_s(const ConvGemmParams& p, hipStream_t s) {
    return p.tile_stats ? launch_dn2<T, NCH, COF, PW, true>(p, s) : launch_dn2<T, NCH, COF, PW, false>(p, s);
}

// Pixels per tile row for this layer and type: 32, or 16 where the 32-wide patch would not fit the LDS (fp32 with
// four chunks per pair); 0: not one of this kernel's layers.
static int dn2_width(int nch, int dtype) {
    if (nch == 1 || nch == 2) return 32;
    if (nch == 4 && dtype != STMI_BF16) return 16;   // (no bf16 layer of the network has 128 input channels at stride 2)
    return 0;
}

// p: a forward launch.  Fills q, the launch form of p (grid in the pt_ fields); false: not one of this kernel's.
static bool dn2_prepare(const ConvGemmParams& p, int dtype, ConvGemmParams& q) {
    if (STMI_ENV_SET("STMI_NO_DN2")) return false;
    const int es = dtype == STMI_BF16 ? 2 : 4;
    if (p.in_stride != 2 || p.ntaps != 9 || p.cdiv != 0 || p.nclasses != 0 || (p.w_batch_stride != 0 && !p.clamp_t)) return false;
    if (p.oys != 1 || p.oxs != 1 || p.oy0 != 0 || p.ox0 != 0 || p.Hs != p.Ho || p.Ws != p.Wo) return false;
    if ((p.Hi & 1) || (p.Wi & 1) || 2 * p.Ho != p.Hi || 2 * p.Wo != p.Wi) return false;   // pads below / right only
    if (p.klen != p.Ci || (2 * p.Ci * es) % 128 != 0 || (p.Co != 64 && p.Co != 128) || p.co_pad < p.Co) return false;
    const int nch = 2 * p.Ci * es / 128;
    if (dn2_width(nch, dtype) == 0) return false;
    // the absorbed-norm form: bf16, pixels of 64 or 128 bytes, with the statistics the next norm wants
    if (p.clamp_t && (dtype != STMI_BF16 || nch > 2 || !p.tile_stats || p.w_batch_stride <= 0 || !p.clamp_bias ||
                      (((uintptr_t)p.clamp_t | (uintptr_t)p.clamp_bias) & 15) != 0))
        return false;
    if (p.kpad < p.Ci || (p.kpad * es) % 16 != 0) return false;
    for (int t = 0; t < 9; t++)
        if (p.taps[t].dy != t / 3 || p.taps[t].dx != t % 3 || p.taps[t].widx != t) return false;
    if (p.bias || p.mask || p.addend || p.out_pre || p.relu || p.pool_out) return false;
    if (p.tile_stats && p.tile_stats_mode != 1) return false;
    if ((((uintptr_t)p.in | (uintptr_t)p.wp | (uintptr_t)p.out) & 15) != 0) return false;
    if ((long)p.N * p.Hi * p.Wi * p.Ci >= (1L << 31) || (long)p.N * p.Ho * p.Wo * p.Co >= (1L << 31)) return false;
    q = p;
    q.pt_div[1] = (unsigned)ceil_div(p.Wo, dn2_width(nch, dtype));
    q.pt_div[2] = (unsigned)ceil_div(p.Ho, kPH);
    if ((long)q.pt_div[1] * q.pt_div[2] * p.N >= (1L << 31)) return false;
    return true;
}

// Statistics rows per sample this kernel leaves for the launch p (one per block), 0 when p is not its layer.
int conv_dn2_tile_stats_tiles(const ConvGemmParams& p, int dtype) {
    ConvGemmParams q;
    ConvGemmParams plain = p;   // a question about the geometry only
    plain.tile_stats = nullptr;
    plain.clamp_t = nullptr;
    plain.w_batch_stride = 0;
    if (!dn2_prepare(plain, dtype, q)) return 0;
    return (int)(q.pt_div[1] * q.pt_div[2]);
}

bool try_launch_conv_dn2(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len) {
    ConvGemmParams q;
    if (!dn2_prepare(p, dtype, q)) return false;
    if (p.tile_stats && p.tile_stats_ts != (int)(q.pt_div[1] * q.pt_div[2])) return false;
    const int es = dtype == STMI_BF16 ? 2 : 4;
    const int nch = 2 * p.Ci * es / 128;
    const int cof = p.Co / 16;
    if (name)
        snprintf(name, name_len, "void stmi::conv_dn2_kernel<%s, %d, %d, %d, %s, %s>(stmi::ConvGemmParams)",
                 dtype == STMI_BF16 ? "__bf16" : "float", nch, cof, dn2_width(nch, dtype), p.tile_stats ? "true" : "false",
                 p.clamp_t ? "true" : "false");
    if (p.clamp_t) {
        if (nch == 1) *rc = cof == 4 ? launch_dn2<bf16_t, 1, 4, 32, true, true>(q, s) : launch_dn2<bf16_t, 1, 8, 32, true, true>(q, s);
        else *rc = cof == 4 ? launch_dn2<bf16_t, 2, 4, 32, true, true>(q, s) : launch_dn2<bf16_t, 2, 8, 32, true, true>(q, s);
        return true;
    }
    if (dtype == STMI_BF16) {
        if (nch == 1) *rc = cof == 4 ? launch_dn2_s<bf16_t, 1, 4, 32>(q, s) : launch_dn2_s<bf16_t, 1, 8, 32>(q, s);
        else *rc = cof == 4 ? launch_dn2_s<bf16_t, 2, 4, 32>(q, s) : launch_dn2_s<bf16_t, 2, 8, 32>(q, s);
    } else {
        if (nch == 1) *rc = cof == 4 ? launch_dn2_s<float, 1, 4, 32>(q, s) : launch_dn2_s<float, 1, 8, 32>(q, s);
        else if (nch == 2) *rc = cof == 4 ? launch_dn2_s<float, 2, 4, 32>(q, s) : launch_dn2_s<float, 2, 8, 32>(q, s);
        else *rc = cof == 4 ? launch_dn2_s<float, 4, 4, 16>(q, s) : launch_dn2_s<float, 4, 8, 16>(q, s);
    }
    return true;
}

}  // namespace stmi
