// 3x3 stride-1 convolution with the input patch held in LDS (gfx950).
//
// The LDS-DMA gather-GEMM of conv_gemm_glds.hip re-fetches every pixel row once per tap, and
// the global -> LDS path moves only about 70 GB/s per CU, so that kernel is bound by its
// fill traffic, not by the MFMA.  Here a block owns a PH x PW rectangle of 128 output pixels
// of one image and, per 128-byte channel chunk, keeps the (PH+2) x (PW+2) halo patch of
// that rectangle in LDS: the nine taps read the same patch at shifted rows, and only the
// weight tile (BLK_CO rows x 128 B) is streamed per tap.  Fill bytes per MFMA drop by
// 2.3x (BLK_CO = 128) to 2.9x (BLK_CO = 64).
//
//   * weight tiles go through an NSW-deep ring, the patch is double-buffered; all fills
//     are global_load_lds_dwordx4 (8 rows x 128 B per wave-instruction, XOR swizzle on
//     the source side, a page of zeros behind out-of-image rows);
//   * the nine taps of a chunk are unrolled, so the number of DMA instructions issued
//     after the step being waited for is a compile-time constant: every step waits with
//     a counted s_waitcnt vmcnt(N) and meets the other waves at one s_barrier;
//   * fragment reads are inline asm (a compiler-visible LDS read after an LDS-DMA write
//     is preceded by vmcnt(0)), software-pipelined across steps: the second K half of a
//     step is multiplied after the barrier of the next one, under that step's first reads.
#include "conv_gemm.hpp"

#include <stdlib.h>

#include <type_traits>

namespace stmi {

__device__ uint4 g_patch_zero_page[8];  // zero-initialised: source of out-of-image rows

#ifdef STMI_STAMPS
// tuning aid (`make stamps`, scripts/patch_stamps.py; not in the product build): phase time stamps of the
// first 256 blocks, consumer wave 0 and producer wave 4: 100 MHz wall counter and shader cycle counter
__device__ unsigned long long g_stamps[256 * 16];
#define STMI_STAMP(i)                                                                   \
    if ((threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == 4) && blockIdx.x < 256) { \
        g_stamps[blockIdx.x * 16 + ((threadIdx.x >> 6) ? 8 : 0) + 2 * (i)] = wall_clock64();   \
        g_stamps[blockIdx.x * 16 + ((threadIdx.x >> 6) ? 8 : 0) + 2 * (i) + 1] = clock64();    \
    }
#else
#define STMI_STAMP(i)
#endif


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
__device__ __forceinline__ u32x4 lds_read16(unsigned addr, int imm) {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(imm) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_landed(u32x4& v) { asm volatile("" : "+v"(v)); }

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int kProloguePieces = 5;  // pieces of the FIRST patch the producer wave issues itself
constexpr int kPatchPieces = 9;  // 8-row DMA pieces per wave and chunk: 4 * 9 * 8 = 288 patch rows

// pieces of the NEXT chunk's patch a wave issues in tap iteration t (they must land with the
// group that carries the weights of the next chunk's first tap, NSW-1 steps ahead)
constexpr int pieces_at(int nsw, int t) {
    const int span = 11 - nsw;  // iterations 0 .. 10-nsw
    const int extra = kPatchPieces - span;
    return t < 0 || t >= span ? 0 : (t < extra ? 2 : 1);
}
constexpr int pieces_before(int nsw, int t) {
    int n = 0;
    for (int i = 0; i < t; i++) n += pieces_at(nsw, i);
    return n;
}
// DMA instructions in the group issued in tap iteration t (t < 0: the previous chunk)
constexpr int group_size(int nsw, int ni_w, bool last, int t) {
    if (t < 0) return ni_w + pieces_at(nsw, t + 9);
    if (last) return t + nsw - 1 <= 8 ? ni_w : 0;
    return ni_w + pieces_at(nsw, t);
}
// instructions that may still be in flight when tap iteration t starts computing
constexpr int in_flight_ok(int nsw, int ni_w, bool last, int t) {
    int n = 0;
    for (int j = 1; j <= nsw - 2; j++) n += group_size(nsw, ni_w, last, t - j);
    return n;
}

}  // namespace

// One float from LDS at a byte address plus an immediate offset (no address arithmetic per row); asm so that
// the read is not merged or moved over the barriers around it.  The caller waits (lds_wait) before use.
template <int OFF>
__device__ __forceinline__ float lds_read_f32_at(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

// Epilogue of the one-tile kernel among the four MFMA waves alone, for plain bf16 convolutions that leave tile
// statistics (ConvGemmParams::pt_wepi).  A wave stages its 64 pixels x 16 TC channels as rows of T in a free patch
// buffer (no barrier: nobody else reads them; bias and alpha go in before) and stores them as whole row pieces.
// The statistics (mode 1) are sums of the stored rows on the matrix cores (wave_stats_mma); the two pixel halves
// of a channel half meet in `spart` behind ONE barrier of the four waves (the DMA waves have left the kernel),
// one row per tile.  The generic epilogue needs two to four barriers of all eight waves, fp32 staging and a column pass.
// LDS reads behind the global stores are inline asm: a compiler-visible one would be preceded by vmcnt(0), i.e.
// by the stores' round trip to memory.
template <typename T, int TC, int PW>
__device__ __forceinline__ void patch_wave_epilogue(const ConvGemmParams& p, const f32x4 (&acc)[TC][4], char* stage_w,
                                                    unsigned stage_w_addr, unsigned spart_addr, int wave, int lane, int n,
                                                    int ty0, int tx0, int co0) {
    static_assert(sizeof(T) == 2 && (TC == 2 || TC == 4), "bf16 tiles of 64 or 128 channels");
    constexpr int TP = 4, PH = 128 / PW, FPR = PW / 16;
    constexpr int WCH = TC * 16;                          // channels of a wave
    constexpr int SROW = WCH * (int)sizeof(T) + 16;       // bytes of a staged row
    constexpr int LPR = WCH / 8;                          // lanes per staged row (16 bytes each)
    constexpr int PXP = 64 / LPR, NPASS = LPR;            // pixels per pass, passes
    const int H = p.Hi, W = p.Wi;
    const int wc = wave >> 1, wpx = wave & 1, g = lane >> 4, r16 = lane & 15;
#pragma unroll
    for (int tc = 0; tc < TC; tc++) {
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        const int ch = co0 + wc * WCH + tc * 16 + 4 * g;
        if (p.bias) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (ch + u < p.Co) bias[u] = p.bias[ch + u];
        }
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = (acc[tc][tp][u] + bias[u]) * p.alpha;
            store_vec<T, 4>(reinterpret_cast<T*>(stage_w + (tp * 16 + r16) * SROW) + tc * 16 + 4 * g, v);
        }
    }
    lds_wait();
    asm volatile("" ::: "memory");
    if (p.tile_stats) {
        // this wave's half of the tile's pixels; rows of pixels outside the image become zeros first
        const int rows_in = max(0, min(TP / FPR, H - ty0 - wpx * (TP / FPR))), cols_in = min(PW, W - tx0);
        if (rows_in * cols_in < 64) {
            const int fl = lane >> 4;
            if (fl / FPR >= rows_in || (fl % FPR) * 16 + (lane & 15) >= cols_in) {
#pragma unroll
                for (int cb = 0; cb < WCH * (int)sizeof(T); cb += 16)
                    *reinterpret_cast<uint4*>(stage_w + lane * SROW + cb) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
        asm volatile("" ::: "memory");
        f32x4 acc_s[TC], acc_q[TC];
#pragma unroll
        for (int k = 0; k < TC; k++) acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
        wave_stats_mma<WCH>(stage_w, SROW, lane, acc_s, acc_q);
        // C/D layout: the lanes with g == r16 / 4 hold the column sum and the diagonal for channel 16 k + r16
        if (g == (r16 >> 2)) {
#pragma unroll
            for (int k = 0; k < TC; k++) {
                const unsigned a = spart_addr + ((wave * 2 * WCH) + k * 16 + r16) * 4;
                asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(acc_s[k][r16 & 3]) : "memory");
                asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(a), "v"(acc_q[k][r16 & 3]), "n"(WCH * 4) : "memory");
            }
        }
        lds_wait();
        __builtin_amdgcn_s_barrier();                      // the four MFMA waves: both halves' sums are in LDS
        asm volatile("" ::: "memory");
        if (wpx == 0 && lane < WCH) {
            const unsigned a0 = spart_addr + (wave * 2 * WCH + lane) * 4, a1 = a0 + 2 * WCH * 4;   // this wave, wave + 1
            const float s0 = lds_read_f32_at<0>(a0), q0 = lds_read_f32_at<WCH * 4>(a0);
            const float s1 = lds_read_f32_at<0>(a1), q1 = lds_read_f32_at<WCH * 4>(a1);
            unsigned kraw;
            asm volatile("ds_read_u16 %0, %1" : "=v"(kraw) : "v"(stage_w_addr + lane * 2) : "memory");
            lds_wait();
            asm volatile("" : "+v"(kraw));
            const float S = s0 + s1, Q = q0 + q1;
            const float kf = __uint_as_float(kraw << 16);   // the tile's first pixel
            const float nt = (float)(min(PH, H - ty0) * min(PW, W - tx0));
            const int ch = co0 + wc * WCH + lane;
            if (ch < p.Co) {
                const int tile_in_sample = (ty0 / PH) * (int)p.pt_div[1] + tx0 / PW;
                float* row = p.tile_stats + (((long)n * p.tile_stats_ts + tile_in_sample) * 4) * p.Co + ch;
                row[0] = kf;
                row[p.Co] = S - nt * kf;
                row[2 * p.Co] = fmaxf(Q - 2.f * kf * S + nt * kf * kf, 0.f);
                row[3 * p.Co] = nt;
            }
        }
    }
    // out: pixel PXP j + lane / LPR of the wave's 64, 16 bytes (8 channels) per lane, all reads in flight first
    const int lr = lane / LPR, lchunk = lane % LPR;
    const int ch = co0 + wc * WCH + lchunk * 8;
    u32x4 dv[NPASS];
#pragma unroll
    for (int j = 0; j < NPASS; j++) dv[j] = lds_read16(stage_w_addr + (PXP * j + lr) * SROW + (lchunk << 4), 0);
    lds_wait();
    T* out = (T*)p.out;
#pragma unroll
    for (int j = 0; j < NPASS; j++) {
        lds_landed(dv[j]);
        const int px = PXP * j + lr;
        const int f = wpx * TP + (px >> 4);
        const int oy = ty0 + f / FPR, ox = tx0 + (f % FPR) * 16 + (px & 15);
        if (oy < H && ox < W && ch < p.Co)
            *reinterpret_cast<u32x4*>(out + (((long)n * H + oy) * W + ox) * p.Co + ch) = dv[j];
    }
}

// ONE: the layer has a single channel chunk (Cin * sizeof(T) == 128): every instruction runs once,
// i.e. misses the instruction cache once, so that instantiation keeps its code small (rolled tap
// loop in the MFMA waves, no next-patch logic in the DMA waves).
// packed maximum of eight 16-bit floats by their bit patterns (see conv_patch_pers_kernel, CLAMP)
__device__ __forceinline__ void pk_max_bits(u32x4& v, const u32x4& t) {
#pragma unroll
    for (int u = 0; u < 4; u++) {
        unsigned r;
        asm volatile("v_pk_max_f16 %0, %1, %2" : "=v"(r) : "v"(v[u]), "v"(t[u]));
        v[u] = r;
    }
}

template <typename T, int TC, int PW, int NSW, bool ONE>
__global__ __launch_bounds__(512) void conv_patch_kernel(const ConvGemmParams p) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 8 * EPC;       // channels per chunk (128 bytes)
    constexpr int TP = 4;             // 16-pixel fragments per consumer wave
    constexpr int BLK_CO = 32 * TC;   // 2 x 2 consumer waves
    constexpr int PH = 128 / PW;      // 128 output pixels per block
    constexpr int FPR = PW / 16;      // fragments per patch row
    constexpr int PWH = PW + 2;
    constexpr int NROWS = (PH + 2) * PWH;
    constexpr int NI_W = BLK_CO / 32;
    constexpr int W_SLOT = BLK_CO * 128;                  // bytes
    constexpr int P_BUF = 4 * kPatchPieces * 8 * 128;     // bytes
    static_assert(NROWS <= 4 * kPatchPieces * 8, "patch does not fit its pieces");
    static_assert(PW == 16 || PW == 32 || PW == 64, "patch width");
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(128))) uint4 lds[];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned patch_base = lds_base + NSW * W_SLOT;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Hi, W = p.Wi, C = p.Ci;

    // tile of this block: blocks b and b+8 share an XCD, each XCD gets one contiguous run
    unsigned tile;
    {
        const unsigned G = gridDim.x, b = blockIdx.x;
        const unsigned per = G >> 3, rem = G & 7, xcd = b & 7, slot = b >> 3;
        tile = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    // divisions by host-provided reciprocals (exact for the operand range the launcher checked)
    auto div_by = [&](unsigned a, int i) -> unsigned {
        return p.pt_magic[i] ? __umulhi(a, p.pt_magic[i]) : (p.pt_div[i] == 1 ? a : a / p.pt_div[i]);
    };
    // split K: the grid is pt_ksplit copies of the tile list, copy s owning channel chunks kc0 .. kc0 + nchunks
    int kslice = 0;
    if (p.pt_ksplit > 1) {
        const unsigned per_slice = gridDim.x / (unsigned)p.pt_ksplit;
        kslice = (int)(tile / per_slice);
        tile -= (unsigned)kslice * per_slice;
    }
    unsigned tp_id = div_by(tile, 0);
    const int co0 = (tile - tp_id * p.pt_div[0]) * BLK_CO;
    unsigned q = div_by(tp_id, 1);
    const int tx0 = (tp_id - q * p.pt_div[1]) * PW;
    const unsigned n_img = div_by(q, 2);
    const int ty0 = (q - n_img * p.pt_div[2]) * PH;
    const int n = (int)n_img;
    const int kc0 = kslice * p.pt_kchunks;
    const int nchunks = p.pt_ksplit > 1 ? min(p.pt_kchunks, p.klen / BK - kc0) : p.klen / BK;

    STMI_STAMP(0);
    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (wave >= 4) {
        // ------------------------------------------------------------------ producer waves
        // Issuing a DMA piece stalls the issuing wave for 60-180 cycles; kept out of the
        // MFMA waves' instruction streams, that cost overlaps their work (each SIMD runs
        // one producer and one consumer wave).
        const int pw = wave - 4;
        const int sub = lane >> 3, pos = lane & 7;
        const T* in = (const T*)p.in + kc0 * BK;
        const T* wp = (const T*)p.wp + kc0 * BK;
        const T* zero_src = (const T*)(g_patch_zero_page + pos);
        int wrow_off[NI_W];
#pragma unroll
        for (int j = 0; j < NI_W; j++) {
            const int row = (pw * NI_W + j) * 8 + sub;
            wrow_off[j] = row * p.kpad + (pos ^ ((row >> 1) & 7)) * EPC;
        }
        auto stage_weights = [&](int tap_i, int chunk, int slot) {
            const Tap tap = p.taps[tap_i];
            const T* wsrc = wp + ((long)tap.widx * p.co_pad + co0) * p.kpad + chunk * BK;
            const unsigned dst = lds_base + slot * W_SLOT + pw * (NI_W * 1024);
#pragma unroll
            for (int j = 0; j < NI_W; j++) glds16(wsrc + wrow_off[j], dst + j * 1024);
        };
        stage_weights(0, 0, 0);
        // patch rows this lane fetches in every chunk: element offset of the row's chunk, or -1.
        // Consecutive pieces are 8 patch rows apart: one division, then steps.
        int poff[kPatchPieces];
        {
            int pr = pw * kPatchPieces * 8 + sub;
            const int py0 = pr / PWH;
            int px = pr - py0 * PWH;
            int iy = ty0 - 1 + py0, ix = tx0 - 1 + px;
            const int swz = (pos ^ (sub & 7)) * EPC;  // pr & 7 == sub for every piece
            const int step = 8 * C, wrap = (W - PWH) * C;
            int lin = ((n * H + iy) * W + ix) * C + swz;
#pragma unroll
            for (int i = 0; i < kPatchPieces; i++) {
                const bool ok = pr < NROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                poff[i] = ok ? lin : -1;
                pr += 8;
                px += 8;
                ix += 8;
                lin += step;
                const bool w = px >= PWH;
                px -= w ? PWH : 0;
                ix -= w ? PWH : 0;
                iy += w ? 1 : 0;
                lin += w ? wrap : 0;
            }
        }
        auto stage_patch_piece = [&](int i, int chunk) {
            const unsigned dst = patch_base + (chunk & 1) * P_BUF + (pw * kPatchPieces + i) * 1024;
            glds16(poff[i] >= 0 ? in + (poff[i] + chunk * BK) : zero_src, dst);
        };
        // The first patch is split with the consumer wave of the same index (idle until the
        // first barrier): issuing a piece costs the issuing wave 60-180 cycles.
#pragma unroll
        for (int i = 0; i < kProloguePieces; i++) stage_patch_piece(i, 0);
#pragma unroll
        for (int s = 1; s < NSW - 1; s++) stage_weights(s, 0, s);

        STMI_STAMP(1);
        int slot = 0;  // ring slot of the current step
        auto chunk_body = [&](auto last_tag, int c) {
            constexpr bool LAST = decltype(last_tag)::value;
            static_for<0, 9>([&](auto t_tag) {
                constexpr int t = decltype(t_tag)::value;
                wait_vmcnt<in_flight_ok(NSW, NI_W, LAST, t)>();
                __builtin_amdgcn_s_barrier();  // step (c, t) is in LDS; step (c, t) - 1 has been read
                asm volatile("" ::: "memory");
                constexpr int tn = (t + NSW - 1) % 9, cn = (t + NSW - 1) / 9;
                if (!LAST || cn == 0) stage_weights(tn, c + cn, slot == 0 ? NSW - 1 : slot - 1);
                if (!LAST) {
#pragma unroll
                    for (int q = 0; q < pieces_at(NSW, t); q++) stage_patch_piece(pieces_before(NSW, t) + q, c + 1);
                }
                slot = slot + 1 == NSW ? 0 : slot + 1;
            });
        };
        if constexpr (!ONE) {
            for (int c = 0; c + 1 < nchunks; c++) chunk_body(std::false_type{}, c);
        }
        chunk_body(std::true_type{}, nchunks - 1);
        STMI_STAMP(2);
        if constexpr (sizeof(T) == 2 && (TC == 2 || TC == 4)) {
            if (p.pt_wepi) return;   // the MFMA waves finish the tile among themselves (patch_wave_epilogue)
        }
    } else {
        // ------------------------------------------------------------------ consumer waves
        const int wc = wave >> 1, wpx = wave & 1;
        const int g = lane >> 4, r16 = lane & 15;
        // patch row under the centre tap of each of this lane's pixel columns
        int pr_c[TP];
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
            const int f = wpx * TP + tp;
            pr_c[tp] = (f / FPR + 1) * PWH + (f % FPR) * 16 + r16 + 1;
        }
        const unsigned a_off = lds_base + wc * TC * 2048 + r16 * 128 + ((g ^ ((r16 >> 1) & 7)) << 4);
        // this wave's share of the first patch (the producer wave of the same index takes the rest)
        {
            const int sub = lane >> 3, pos = lane & 7;
            const T* in = (const T*)p.in + kc0 * BK;
#pragma unroll
            for (int i = kProloguePieces; i < kPatchPieces; i++) {
                const int pr = (wave * kPatchPieces + i) * 8 + sub;
                const int py = pr / PWH, px = pr - py * PWH;
                const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
                const bool ok = pr < NROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                const T* src = ok ? in + (((n * H + iy) * W + ix) * C + (pos ^ (pr & 7)) * EPC)
                                  : (const T*)(g_patch_zero_page + pos);
                glds16(src, patch_base + (wave * kPatchPieces + i) * 1024);
            }
        }

        u32x4 fa[2][TC], fb[2][TP];  // fragments of the two K halves of a step
#pragma unroll
        for (int tc = 0; tc < TC; tc++) fa[1][tc] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int tp = 0; tp < TP; tp++) fb[1][tp] = (u32x4){0u, 0u, 0u, 0u};

        // multiply half `use` while the reads of half `load` are issued between the MFMAs
        auto mma_and_read = [&](auto use_tag, unsigned wbase, const unsigned (&baddr)[TP], bool do_read) {
            constexpr int USE = decltype(use_tag)::value, LOAD = 1 - USE;
            const unsigned flip = LOAD ? 64u : 0u;
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
#pragma unroll
                for (int tp = 0; tp < TP; tp++) {
                    Mma<T>::mma(__builtin_bit_cast(frag, fa[USE][tc]), __builtin_bit_cast(frag, fb[USE][tp]), acc[tc][tp]);
                    // reads go behind the first MFMAs so that the rest of the block covers their
                    // latency: one per MFMA when there are sixteen MFMAs, two when there are eight
                    if (do_read) {
                        constexpr int RPG = TC * TP >= 16 ? 1 : 2;
#pragma unroll
                        for (int k = RPG * (tc * TP + tp); k < RPG * (tc * TP + tp + 1); k++) {
                            if (k < TC) fa[LOAD][k] = lds_read16(wbase ^ flip, k * 2048);
                            else if (k < TC + TP) fb[LOAD][k - TC] = lds_read16(baddr[k - TC] ^ flip, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto landed = [&](auto half_tag) {
            constexpr int HALF = decltype(half_tag)::value;
            lds_wait();
#pragma unroll
            for (int tc = 0; tc < TC; tc++) lds_landed(fa[HALF][tc]);
#pragma unroll
            for (int tp = 0; tp < TP; tp++) lds_landed(fb[HALF][tp]);
        };

        int slot = 0;
        unsigned baddr[TP];
#pragma unroll
        for (int tp = 0; tp < TP; tp++) baddr[tp] = 0;
        auto tap_step = [&](int c, int t, unsigned pbuf) {
            if (c == 0 && t == 0) wait_vmcnt<0>();  // own share of the first patch
            __builtin_amdgcn_s_barrier();
            if (c == 0 && t == 0) { STMI_STAMP(1); }
            asm volatile("" ::: "memory");
            const Tap tap = p.taps[t];
            const int shift = tap.dy * PWH + tap.dx;
            const unsigned wbase = a_off + slot * W_SLOT;
#pragma unroll
            for (int tp = 0; tp < TP; tp++) {
                const int pr = pr_c[tp] + shift;
                baddr[tp] = pbuf + pr * 128 + ((g ^ (pr & 7)) << 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            // second half of the previous step under the reads of this step's first half
            mma_and_read(std::integral_constant<int, 1>{}, wbase, baddr, true);
            landed(std::integral_constant<int, 0>{});
            mma_and_read(std::integral_constant<int, 0>{}, wbase, baddr, true);
            landed(std::integral_constant<int, 1>{});  // every read of the step is back before the next barrier
            slot = slot + 1 == NSW ? 0 : slot + 1;
        };
        if constexpr (ONE) {
#pragma nounroll
            for (int t = 0; t < 9; t++) tap_step(0, t, patch_base);
        } else {
            // Unrolled, the per-tap fragment addresses are loop invariants of the chunk loop.
            for (int c = 0; c < nchunks; c++) {
                const unsigned pbuf = patch_base + (c & 1) * P_BUF;
                static_for<0, 9>([&](auto t_tag) { tap_step(c, decltype(t_tag)::value, pbuf); });
            }
        }
        mma_and_read(std::integral_constant<int, 1>{}, 0u, baddr, false);
        STMI_STAMP(2);
        if constexpr (sizeof(T) == 2 && (TC == 2 || TC == 4)) {
            if (p.pt_wepi) {
                // free LDS: the patch buffer the last chunk did not use (read last before that chunk's first
                // barrier), and the ring slot after the last step's (consumed before the last step's barrier)
                const unsigned stage_off = NSW * W_SLOT + (nchunks & 1) * P_BUF + wave * (64 * (TC * 32 + 16));
                patch_wave_epilogue<T, TC, PW>(p, acc, reinterpret_cast<char*>(lds) + stage_off, lds_base + stage_off,
                                               lds_base + slot * W_SLOT, wave, lane, n, ty0, tx0, co0);
                STMI_STAMP(3);
                return;
            }
        }
    }

    // epilogue
    const long img = (long)n * H;
    if (p.row_store_ok) {
        TileStatsOut ts_out{nullptr, 0, 0, 0, nullptr};
        if (p.tile_stats) {
            const int ts = (int)(p.pt_div[1] * p.pt_div[2]);
            const int tile_in_sample = (ty0 / PH) * (int)p.pt_div[1] + tx0 / PW;
            const int planes = p.tile_stats_mode == 1 ? 4 : 2;
            ts_out.base = p.tile_stats + (((long)n * ts + tile_in_sample) * planes) * p.Co + co0;
            ts_out.C = p.Co;
            ts_out.count = min(PH, H - ty0) * min(PW, W - tx0);
            ts_out.mode = p.tile_stats_mode;
            ts_out.stats_n = p.bn_stats ? p.bn_stats + (long)n * p.Co * 2 : nullptr;
        }
        conv_epilogue_lds<T, TC, TP, 2, 2, 512, TP>(p, acc, co0, reinterpret_cast<float*>(lds), [&](int pl) -> long {
            const int f = pl >> 4;
            const int oy = ty0 + f / FPR, ox = tx0 + (f % FPR) * 16 + (pl & 15);
            return oy < H && ox < W ? ((img + oy) * W + ox) * p.Co : -1;
        }, ts_out, p.pt_ksplit > 1 ? p.ksplit_ws + (long)kslice * p.ksplit_plane : nullptr);
        if (p.pool_out) {
            // 2 x 2 max pool of the tile from the staged rows (finished values; staged row = local pixel):
            // tiles start at even coordinates and have even sides, so every window lies in one tile
            constexpr int STRIDE = BLK_CO + 4, LPR = BLK_CO / EPC, QW = PW / 2, QN = (PH / 2) * QW;
            const float* stage = reinterpret_cast<const float*>(lds);
            T* __restrict__ pooled = (T*)p.pool_out;
            const int Hq = (H + 1) >> 1, Wq = (W + 1) >> 1;
            __syncthreads();
            for (int task = tid; task < QN * LPR; task += 512) {
                const int q = task / LPR, cl = (task - q * LPR) * EPC;
                const int qy = q / QW, qx = q - qy * QW;
                const int oy = ty0 + 2 * qy, ox = tx0 + 2 * qx;
                if (oy >= H || ox >= W || co0 + cl >= p.Co) continue;
                float m[EPC];
#pragma unroll
                for (int u = 0; u < EPC; u++) m[u] = -INFINITY;
#pragma unroll
                for (int dy = 0; dy < 2; dy++) {
#pragma unroll
                    for (int dx = 0; dx < 2; dx++) {
                        if (oy + dy >= H || ox + dx >= W) continue;
                        const int ly = 2 * qy + dy, lx = 2 * qx + dx;
                        const int pl = (ly * FPR + (lx >> 4)) * 16 + (lx & 15);
#pragma unroll
                        for (int u = 0; u < EPC; u += 4) {
                            const f32x4 t = *reinterpret_cast<const f32x4*>(stage + pl * STRIDE + cl + u);
                            m[u] = fmaxf(m[u], t[0]); m[u + 1] = fmaxf(m[u + 1], t[1]);
                            m[u + 2] = fmaxf(m[u + 2], t[2]); m[u + 3] = fmaxf(m[u + 3], t[3]);
                        }
                    }
                }
                store_vec<T, EPC>(pooled + (((long)n * Hq + (oy >> 1)) * Wq + (ox >> 1)) * p.Co + co0 + cl, m);
            }
        }
    } else if (wave < 4) {
        const int wc = wave >> 1, wpx = wave & 1;
        const int g = lane >> 4, r16 = lane & 15;
        conv_epilogue_at<T, TC, TP>(p, acc, co0, wc, g, [&](int tp) -> long {
            const int f = wpx * TP + tp;
            const int oy = ty0 + f / FPR, ox = tx0 + (f % FPR) * 16 + r16;
            return oy < H && ox < W ? ((img + oy) * W + ox) * p.Co : -1;
        });
    }
    STMI_STAMP(3);
}

// ---------------------------------------------------------------------------
// Persistent form for layers of several block rounds per CU (1080p frames, 512 x 512 Gatys images, the first VGG
// layers of a training batch).  With one resident block per CU (88-136 KB of LDS) nothing overlaps a block's
// prologue (first patch + three weight tiles: 1.3-2.0 us) or its epilogue; here a block walks a list of tiles and
// the K steps of all its tiles form ONE stream through the same ring: while the MFMA waves stage and store tile i
// (in staging rows of their own, behind the operand buffers), the DMA waves have already issued the first patch
// and the first weight tiles of tile i + 1.  The DMA waves issue every patch piece themselves (the MFMA waves
// have no idle prologue to lend), as many pieces as the patch has rows, and join the epilogue's barriers with
// bare s_barriers.
// ---------------------------------------------------------------------------
namespace {
constexpr int pieces_at_pp(int nsw, int pp, int t) {
    const int span = 11 - nsw;
    return t < 0 || t >= span ? 0 : pp / span + (t < pp % span ? 1 : 0);
}
constexpr int pieces_before_pp(int nsw, int pp, int t) {
    int n = 0;
    for (int i = 0; i < t; i++) n += pieces_at_pp(nsw, pp, i);
    return n;
}
constexpr int group_size_pp(int nsw, int ni_w, int pp, bool last, int t) {
    if (t < 0) return ni_w + pieces_at_pp(nsw, pp, t + 9);
    if (last) return t + nsw - 1 <= 8 ? ni_w : 0;
    return ni_w + pieces_at_pp(nsw, pp, t);
}
constexpr int in_flight_ok_pp(int nsw, int ni_w, int pp, bool last, int t) {
    int n = 0;
    for (int j = 1; j <= nsw - 2; j++) n += group_size_pp(nsw, ni_w, pp, last, t - j);
    return n;
}
}  // namespace

//
// STATS: the launch also leaves the per-tile sums of ConvGemmParams::tile_stats (mode 1) for the instance norm
// behind the layer.  The MFMA waves' epilogue is untouched: while they store a staging group's rows, the DMA
// waves -- idle there, their next loads already issued -- add up the same rows by column (lane = channel x
// one of 256 / BLK_CO row slices, the slices then added across lanes): one statistics row per tile, as the
// plain kernel leaves, at 2 % of the launch (1080p residual layers, measured).

//
// WEPI (bf16, 128-channel tiles, plain convolutions: no bias, mask, addend, ReLU or second output): the MFMA waves
// finish a tile without the rest of the block.  A wave leaves its 64 pixels x 64 channels as bf16 rows in a staging
// area of its own (no barrier: nobody else reads them), stores them as 128-byte row pieces, takes the tile
// statistics from the same rows on the matrix cores (wave_stats_mma; TWO statistics rows per tile, one per pixel
// half) and starts the next tile.  The DMA waves then have no part in a tile change at all: the four epilogue
// barriers, the fp32 staging groups and the DMA waves' statistics arithmetic are gone.
// CLAMP (with WEPI): ConvGemmParams::clamp_t -- the launch absorbs the instance norm + ReLU in front of it.
template <typename T, int TC, int PW, int NSW, bool STATS, bool WEPI = false, bool CLAMP = false>
__global__ __launch_bounds__(512) void conv_patch_pers_kernel(const ConvGemmParams p) {
    static_assert(!CLAMP || (WEPI && sizeof(T) == 2), "the clamp rides on the bf16 wave-private form");
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BK = 8 * EPC;
    constexpr int TP = 4;
    constexpr int BLK_CO = 32 * TC;
    constexpr int PH = 128 / PW;
    constexpr int FPR = PW / 16;
    constexpr int PWH = PW + 2;
    constexpr int NROWS = (PH + 2) * PWH;
    constexpr int NI_W = BLK_CO / 32;
    constexpr int PP = (NROWS + 31) / 32;                 // 8-row patch pieces per DMA wave and chunk
    constexpr int W_SLOT = BLK_CO * 128;
    constexpr int P_BUF = 4 * PP * 8 * 128;
    constexpr int TPG = TC >= 4 ? 2 : 4;                  // pixel fragments per staging group (about 34 KB of rows)
    constexpr int EPI_BARRIERS = 2 * (TP / TPG);          // barriers of conv_epilogue_lds for that grouping
    typedef typename Mma<T>::frag frag;

    extern __shared__ __attribute__((aligned(128))) uint4 lds[];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned patch_base = lds_base + NSW * W_SLOT;
    float* stage = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + NSW * W_SLOT + 2 * P_BUF);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = p.Hi, W = p.Wi, C = p.Ci;
    const int nchunks = p.klen / BK;

    // Tiles of this block.  Blocks b and b+8 share an XCD; XCD x walks the contiguous run
    // [x * total / 8, (x + 1) * total / 8) of the tile order (channel tile fastest, then pixels), its blocks
    // interleaved: one 4 MiB L2 serves neighbouring tiles at any moment of the launch.
    const unsigned total = p.pt_npers;
    const unsigned xcd = blockIdx.x & 7, lane_blk = blockIdx.x >> 3, blk_per_xcd = gridDim.x >> 3;
    const unsigned run0 = (unsigned)((unsigned long)total * xcd / 8), run1 = (unsigned)((unsigned long)total * (xcd + 1) / 8);
    const int my_tiles = run0 + lane_blk < run1 ? (int)((run1 - run0 - lane_blk + blk_per_xcd - 1) / blk_per_xcd) : 0;
    auto div_by = [&](unsigned a, int i) -> unsigned {
        return p.pt_magic[i] ? __umulhi(a, p.pt_magic[i]) : (p.pt_div[i] == 1 ? a : a / p.pt_div[i]);
    };
    struct TilePos { int co0, tx0, ty0, n; };
    auto tile_pos = [&](int it) -> TilePos {
        const unsigned tile = run0 + lane_blk + (unsigned)it * blk_per_xcd;
        const unsigned tp_id = div_by(tile, 0);
        const unsigned q = div_by(tp_id, 1);
        const unsigned n_img = div_by(q, 2);
        return TilePos{(int)(tile - tp_id * p.pt_div[0]) * BLK_CO, (int)(tp_id - q * p.pt_div[1]) * PW,
                       (int)(q - n_img * p.pt_div[2]) * PH, (int)n_img};
    };
    if (my_tiles == 0) return;   // whole block: no barrier has been passed

    if (wave >= 4) {
        // ------------------------------------------------------------------ DMA waves
        const int pw = wave - 4;
        const int sub = lane >> 3, pos = lane & 7;
        const T* in = (const T*)p.in;
        const T* wp = (const T*)p.wp;
        const T* zero_src = (const T*)(g_patch_zero_page + pos);
        int wrow_off[NI_W];
#pragma unroll
        for (int j = 0; j < NI_W; j++) {
            const int row = (pw * NI_W + j) * 8 + sub;
            wrow_off[j] = row * p.kpad + (pos ^ ((row >> 1) & 7)) * EPC;
        }
        int co0 = 0, wn = 0;   // channel tile and sample of the weights being staged
        auto stage_weights = [&](int tap_i, int chunk, int slot) {
            const Tap tap = p.taps[tap_i];
            const T* wsrc = wp + (CLAMP ? (long)wn * p.w_batch_stride : 0L) + ((long)tap.widx * p.co_pad + co0) * p.kpad + chunk * BK;
            const unsigned dst = lds_base + slot * W_SLOT + pw * (NI_W * 1024);
#pragma unroll
            for (int j = 0; j < NI_W; j++) glds16(wsrc + wrow_off[j], dst + j * 1024);
        };
        // element offsets of the patch rows this lane fetches in every chunk of a tile, or -1
        auto patch_offsets = [&](const TilePos& tp, int (&poff)[PP]) {
            int pr = pw * PP * 8 + sub;
            const int py0 = pr / PWH;
            int px = pr - py0 * PWH;
            int iy = tp.ty0 - 1 + py0, ix = tp.tx0 - 1 + px;
            const int swz = (pos ^ (sub & 7)) * EPC;
            const int step = 8 * C, wrap = (W - PWH) * C;
            int lin = ((tp.n * H + iy) * W + ix) * C + swz;
#pragma unroll
            for (int i = 0; i < PP; i++) {
                const bool ok = pr < NROWS && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                poff[i] = ok ? lin : -1;
                pr += 8;
                px += 8;
                ix += 8;
                lin += step;
                const bool w = px >= PWH;
                px -= w ? PWH : 0;
                ix -= w ? PWH : 0;
                iy += w ? 1 : 0;
                lin += w ? wrap : 0;
            }
        };
        int poff[PP], poff_next[PP];
        unsigned gc = 0;   // chunks staged so far: selects the patch buffer
        // rows outside the image: zeros, or (CLAMP) the sample's clamp values, which the absorbed norm maps to zero
        const T* t_rows = CLAMP ? (const T*)p.clamp_t + (pos ^ (sub & 7)) * EPC : nullptr;
        auto stage_patch_piece = [&](int i, const int (&off)[PP], int chunk, unsigned buf, int n_off) {
            const unsigned dst = patch_base + (buf & 1) * P_BUF + (pw * PP + i) * 1024;
            const T* outside = CLAMP ? t_rows + ((long)n_off * C + chunk * BK) : zero_src;
            glds16(off[i] >= 0 ? in + (off[i] + chunk * BK) : outside, dst);
        };
        TilePos cur = tile_pos(0);
        co0 = cur.co0;
        wn = cur.n;
        patch_offsets(cur, poff);
        stage_weights(0, 0, 0);
#pragma unroll
        for (int i = 0; i < PP; i++) stage_patch_piece(i, poff, 0, 0, cur.n);
#pragma unroll
        for (int s = 1; s < NSW - 1; s++) stage_weights(s, 0, s);

        int slot = 0;
        // chunk c of the current tile; `next_off` / `next_chunk`: where the chunk after it lives (the next
        // chunk of this tile, or chunk 0 of the next tile); co0_next: the weights' channel tile for that chunk
        auto chunk_body = [&](auto last_tag, int c, const int (&next_off)[PP], int next_chunk, int co0_next, int n_next) {
            constexpr bool LAST = decltype(last_tag)::value;
            static_for<0, 9>([&](auto t_tag) {
                constexpr int t = decltype(t_tag)::value;
                wait_vmcnt<in_flight_ok_pp(NSW, NI_W, PP, LAST, t)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                constexpr int tn = (t + NSW - 1) % 9, cn = (t + NSW - 1) / 9;
                if (cn == 0) {
                    stage_weights(tn, c, slot == 0 ? NSW - 1 : slot - 1);
                } else if (!LAST) {
                    const int keep = co0, keep_n = wn;
                    co0 = co0_next;
                    wn = n_next;
                    stage_weights(tn, next_chunk, slot == 0 ? NSW - 1 : slot - 1);
                    co0 = keep;
                    wn = keep_n;
                }
                if (!LAST) {
#pragma unroll
                    for (int q = 0; q < pieces_at_pp(NSW, PP, t); q++)
                        stage_patch_piece(pieces_before_pp(NSW, PP, t) + q, next_off, next_chunk, gc + 1, n_next);
                }
                slot = slot + 1 == NSW ? 0 : slot + 1;
            });
            gc++;
        };
        // Column sums of the tile the MFMA waves are storing.  A DMA wave owns BLK_CO / 4 channels; its lanes
        // are (channel, row slice rs): slice rs takes rows rs * RPS .. of every staging group, and the slices
        // of a channel -- lanes of one wave -- are added with lane exchanges, so a tile leaves ONE statistics
        // row.  Shift: the tile's first pixel (always inside the image).  The reads are inline asm (see
        // lds_read16).
        constexpr int RS = 256 / BLK_CO, ROWS = 16 * TPG * 2, RPS = ROWS / RS, STRIDE = BLK_CO + 4, CPW = BLK_CO / 4;
        static_assert(RPS == 32, "a row slice is two pixel fragments of a staging group");
        const int cc = pw * CPW + lane % CPW, rs = lane / CPW;
        const unsigned stage_addr = lds_base + NSW * W_SLOT + 2 * P_BUF;
        auto epilogue_barriers = [&](const TilePos& tp) {
            if constexpr (WEPI) {
                // nothing: the MFMA waves finish a tile among themselves
            } else if constexpr (!STATS) {
#pragma unroll
                for (int b = 0; b < EPI_BARRIERS; b++) __builtin_amdgcn_s_barrier();
            } else {
                float k = 0.f, s1 = 0.f, s2 = 0.f, cnt = 0.f;
                // A tile inside the image takes the short path: rows read with immediate offsets (no
                // address arithmetic) and summed two at a time; validity is worked out only for the tiles
                // on the bottom and right edges, once per pixel fragment (16 rows: 16 pixels of one line).
                const bool interior = tp.ty0 + PH <= H && tp.tx0 + PW <= W;
                const unsigned row0 = stage_addr + (rs * RPS * STRIDE + cc) * 4;
                // row rr = rs * 32 + j of a group: MFMA wave w_ = rr / (16 TPG), fragment (rr % (16 TPG)) / 16
                const int w_ = rs * RPS / (16 * TPG), f_in_group = (rs * RPS % (16 * TPG)) >> 4;
                f32x2 s1v = {0.f, 0.f}, s2v = {0.f, 0.f};
#pragma unroll
                for (int gi = 0; gi < TP / TPG; gi++) {
                    __builtin_amdgcn_s_barrier();   // the previous group's rows are consumed
                    __builtin_amdgcn_s_barrier();   // this group's rows are staged
                    asm volatile("" ::: "memory");
                    float v[RPS];
                    if (gi == 0) k = lds_read_f32_at<0>(stage_addr + cc * 4);
                    static_for<0, RPS>([&](auto j_tag) {
                        constexpr int j = decltype(j_tag)::value;
                        v[j] = lds_read_f32_at<j * STRIDE * 4>(row0);
                    });
                    lds_wait();
                    asm volatile("" : "+v"(k));
#pragma unroll
                    for (int j = 0; j < RPS; j++) asm volatile("" : "+v"(v[j]));
                    if (interior) {
                        const f32x2 kk = {k, k};
#pragma unroll
                        for (int j = 0; j < RPS; j += 2) {
                            const f32x2 d = f32x2{v[j], v[j + 1]} - kk;
                            s1v += d;
                            s2v += d * d;
                        }
                    } else {
#pragma unroll
                        for (int h = 0; h < 2; h++) {
                            const int f = w_ * TP + gi * TPG + f_in_group + h;   // pixel fragment of the tile
                            const int oy = tp.ty0 + f / FPR, ox0 = tp.tx0 + (f % FPR) * 16;
                            const int nvalid = oy < H ? min(16, W - ox0) : 0;    // may be negative: none
#pragma unroll
                            for (int j = 0; j < 16; j++) {
                                const float ok = j < nvalid ? 1.f : 0.f;
                                const float d = (v[h * 16 + j] - k) * ok;
                                s1 += d;
                                s2 += d * d;
                                cnt += ok;
                            }
                        }
                    }
                }
                if (interior) {
                    s1 = s1v[0] + s1v[1];
                    s2 = s2v[0] + s2v[1];
                    cnt = float(RPS * (TP / TPG));
                }
#pragma unroll
                for (int off = CPW; off < 64; off <<= 1) {
                    s1 += __shfl_xor(s1, off);
                    s2 += __shfl_xor(s2, off);
                    cnt += __shfl_xor(cnt, off);
                }
                if (rs == 0 && tp.co0 + cc < p.Co) {
                    const int tile_in_sample = (tp.ty0 / PH) * (int)p.pt_div[1] + tp.tx0 / PW;
                    float* row = p.tile_stats + (((long)tp.n * p.tile_stats_ts + tile_in_sample) * 4) * p.Co + tp.co0 + cc;
                    row[0] = k;
                    row[p.Co] = s1;
                    row[2 * p.Co] = s2;
                    row[3 * p.Co] = cnt;
                }
            }
        };
        for (int it = 0; it < my_tiles; it++) {
            const bool has_next = it + 1 < my_tiles;
            TilePos nxt = cur;
            if (has_next) {
                nxt = tile_pos(it + 1);
                patch_offsets(nxt, poff_next);
            }
            for (int c = 0; c + 1 < nchunks; c++) chunk_body(std::false_type{}, c, poff, c + 1, co0, cur.n);
            if (has_next) {
                chunk_body(std::false_type{}, nchunks - 1, poff_next, 0, nxt.co0, nxt.n);
                // the MFMA waves stage and store tile `it`: their barriers need every wave of the block
                epilogue_barriers(cur);
#pragma unroll
                for (int i = 0; i < PP; i++) poff[i] = poff_next[i];
                cur = nxt;
                co0 = cur.co0;
                wn = cur.n;
            } else {
                chunk_body(std::true_type{}, nchunks - 1, poff, 0, co0, cur.n);
                epilogue_barriers(cur);
            }
        }
    } else {
        // ------------------------------------------------------------------ MFMA waves
        const int wc = wave >> 1, wpx = wave & 1;
        const int g = lane >> 4, r16 = lane & 15;
        int pr_c[TP];
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
            const int f = wpx * TP + tp;
            pr_c[tp] = (f / FPR + 1) * PWH + (f % FPR) * 16 + r16 + 1;
        }
        const unsigned a_off = lds_base + wc * TC * 2048 + r16 * 128 + ((g ^ ((r16 >> 1) & 7)) << 4);
        f32x4 acc[TC][TP];
        u32x4 fa[2][TC], fb[2][TP];
        u32x4 tv[2] = {(u32x4){0u, 0u, 0u, 0u}, (u32x4){0u, 0u, 0u, 0u}};   // CLAMP: see below
        auto mma_and_read = [&](auto use_tag, unsigned wbase, const unsigned (&baddr)[TP], bool do_read) {
            constexpr int USE = decltype(use_tag)::value, LOAD = 1 - USE;
            const unsigned flip = LOAD ? 64u : 0u;
            // CLAMP: a pixel fragment is clamped just before its first product, behind the previous fragment's
            // MFMA, so that all but the first four packed maxima of a half run under matrix work
            if constexpr (CLAMP) pk_max_bits(fb[USE][0], tv[USE]);
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
#pragma unroll
                for (int tp = 0; tp < TP; tp++) {
                    Mma<T>::mma(__builtin_bit_cast(frag, fa[USE][tc]), __builtin_bit_cast(frag, fb[USE][tp]), acc[tc][tp]);
                    if constexpr (CLAMP) {
                        if (tc == 0 && tp + 1 < TP) pk_max_bits(fb[USE][tp + 1], tv[USE]);
                    }
                    if (do_read) {
                        constexpr int RPG = TC * TP >= 16 ? 1 : 2;
#pragma unroll
                        for (int k = RPG * (tc * TP + tp); k < RPG * (tc * TP + tp + 1); k++) {
                            if (k < TC) fa[LOAD][k] = lds_read16(wbase ^ flip, k * 2048);
                            else if (k < TC + TP) fb[LOAD][k - TC] = lds_read16(baddr[k - TC] ^ flip, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        auto landed = [&](auto half_tag) {
            constexpr int HALF = decltype(half_tag)::value;
            lds_wait();
#pragma unroll
            for (int tc = 0; tc < TC; tc++) lds_landed(fa[HALF][tc]);
#pragma unroll
            for (int tp = 0; tp < TP; tp++) lds_landed(fb[HALF][tp]);
        };
        int slot = 0;
        unsigned gc = 0;
        unsigned baddr[TP];
#pragma unroll
        for (int tp = 0; tp < TP; tp++) baddr[tp] = 0;
        // WEPI: this wave's staging rows
        constexpr int SROW = TC * 16 * (int)sizeof(T) + 16;           // bytes of a staged row (this wave's channels)
        constexpr int S_WAVE = 64 * SROW;
        char* const stage_w = reinterpret_cast<char*>(stage) + wave * S_WAVE;
        const unsigned stage_w_addr = lds_base + NSW * W_SLOT + 2 * P_BUF + wave * S_WAVE;
        // CLAMP: this wave's copy of the current sample's clamp values ([C] of T behind the staging rows), and the
        // eight values of each K half of the current chunk that go with this lane's fragment registers.  bf16 bit
        // patterns order like fp16 ones (sign, then magnitude) as long as none reads as an fp16 NaN (|x| >= 2^121),
        // so the maximum of two packed pairs is ONE v_pk_max_f16.
        const unsigned tab_addr = lds_base + NSW * W_SLOT + 2 * P_BUF + 4 * S_WAVE + wave * 256;
        int n_tab = -1;
        u32x4 tv1_chunk = (u32x4){0u, 0u, 0u, 0u};   // the current chunk's values for K half 1 (tv[1] lags one product)
        auto tap_step = [&](int t, unsigned pbuf) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const Tap tap = p.taps[t];
            const int shift = tap.dy * PWH + tap.dx;
            const unsigned wbase = a_off + slot * W_SLOT;
#pragma unroll
            for (int tp = 0; tp < TP; tp++) {
                const int pr = pr_c[tp] + shift;
                baddr[tp] = pbuf + pr * 128 + ((g ^ (pr & 7)) << 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            mma_and_read(std::integral_constant<int, 1>{}, wbase, baddr, true);   // (second half of the PREVIOUS step)
            if constexpr (CLAMP) {
                if (t == 0) tv[1] = tv1_chunk;   // ... which at a chunk's first tap still belonged to the chunk before
            }
            landed(std::integral_constant<int, 0>{});
            mma_and_read(std::integral_constant<int, 0>{}, wbase, baddr, true);
            landed(std::integral_constant<int, 1>{});
            slot = slot + 1 == NSW ? 0 : slot + 1;
        };
        for (int it = 0; it < my_tiles; it++) {
            const TilePos tp0 = tile_pos(it);
#pragma unroll
            for (int a = 0; a < TC; a++) {
                f32x4 start = {0.f, 0.f, 0.f, 0.f};
                if constexpr (CLAMP)   // the constant term of the absorbed norm; C/D rows 4 g + u of channel fragment a
                    start = *reinterpret_cast<const f32x4*>(p.clamp_bias + (long)tp0.n * p.Co + tp0.co0 + wc * (TC * 16) + a * 16 + 4 * g);
#pragma unroll
                for (int b = 0; b < TP; b++) acc[a][b] = start;
            }
#pragma unroll
            for (int tc = 0; tc < TC; tc++) fa[1][tc] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int tp = 0; tp < TP; tp++) fb[1][tp] = (u32x4){0u, 0u, 0u, 0u};
            if constexpr (CLAMP) {
                if (tp0.n != n_tab) {   // a block's run of tiles changes sample a few times per launch
                    n_tab = tp0.n;
                    if (lane * EPC < C) {
                        const uint4 tvals = *reinterpret_cast<const uint4*>((const T*)p.clamp_t + (long)n_tab * C + lane * EPC);
                        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(lds) + (tab_addr - lds_base) + lane * 16) = tvals;
                    }
                    lds_wait();
                    asm volatile("" ::: "memory");
                }
            }
            for (int c = 0; c < nchunks; c++, gc++) {
                const unsigned pbuf = patch_base + (gc & 1) * P_BUF;
                if constexpr (CLAMP) {
                    tv[0] = lds_read16(tab_addr + (c * BK + 8 * g) * 2, 0);
                    tv1_chunk = lds_read16(tab_addr + (c * BK + 32 + 8 * g) * 2, 0);
                    lds_wait();
                    lds_landed(tv[0]);
                    lds_landed(tv1_chunk);
                }
                static_for<0, 9>([&](auto t_tag) { tap_step(decltype(t_tag)::value, pbuf); });
            }
            mma_and_read(std::integral_constant<int, 1>{}, 0u, baddr, false);
            if constexpr (WEPI) {
                // this wave's 64 pixels x 16 TC channels -> rows of T of its own; no barrier: nobody else reads them
#pragma unroll
                for (int tc = 0; tc < TC; tc++)
#pragma unroll
                    for (int tp = 0; tp < TP; tp++) {
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) v[u] = acc[tc][tp][u] * p.alpha;
                        store_vec<T, 4>(reinterpret_cast<T*>(stage_w + (tp * 16 + r16) * SROW) + tc * 16 + 4 * g, v);
                    }
                lds_wait();
                asm volatile("" ::: "memory");
                // out: one eighth of the rows per instruction, pixel 8 j + lane / 8 of the wave's 64, 16 bytes
                // (8 channels) per lane; all eight reads in flight before the first store.  Everything about the
                // pixel but lane / 8 is the same for the whole wave: fragment f = wpx TP + j / 2, tile row f / FPR,
                // first column (f % FPR) 16 + (j & 1) 8.
                {
                    const int l3 = lane >> 3;
                    const int ch = tp0.co0 + wc * (TC * 16) + ((lane & 7) << 3);
                    u32x4 dv[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) dv[j] = lds_read16(stage_w_addr + (8 * j + l3) * SROW + ((lane & 7) << 4), 0);
                    lds_wait();
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        lds_landed(dv[j]);
                        const int f = wpx * TP + (j >> 1);
                        const int oy = tp0.ty0 + f / FPR, ox = tp0.tx0 + (f % FPR) * 16 + (j & 1) * 8 + l3;
                        if (oy < H && ox < W && ch < p.Co)
                            *reinterpret_cast<u32x4*>((T*)p.out + (((long)tp0.n * H + oy) * W + ox) * p.Co + ch) = dv[j];
                    }
                }
                if constexpr (STATS) {
                    // The wave's own statistics row (two rows per tile, one per pixel half): sums of the stored
                    // rows on the matrix cores.  Pixels outside the image were computed but are not part of the
                    // tensor: their rows become zeros first.
                    const int rows_in = max(0, min(PH - wpx * (TP / FPR), min(TP / FPR, H - tp0.ty0 - wpx * (TP / FPR))));
                    const int cols_in = min(PW, W - tp0.tx0);
                    asm volatile("" ::: "memory");
                    if (rows_in * cols_in < 64) {
                        const int fl = lane >> 4;   // fragment of the wave this staged row belongs to
                        if (fl / FPR >= rows_in || (fl % FPR) * 16 + (lane & 15) >= cols_in) {
#pragma unroll
                            for (int cb = 0; cb < TC * 16 * (int)sizeof(T); cb += 16)
                                *reinterpret_cast<uint4*>(stage_w + lane * SROW + cb) = make_uint4(0u, 0u, 0u, 0u);
                        }
                    }
                    asm volatile("" ::: "memory");
                    f32x4 acc_s[TC], acc_q[TC];
#pragma unroll
                    for (int k = 0; k < TC; k++) acc_s[k] = acc_q[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    wave_stats_mma<TC * 16>(stage_w, SROW, lane, acc_s, acc_q);
                    // C/D layout: the lanes with g == r16 / 4 hold the column sum and the diagonal for channel 16 k + r16
                    if (g == (r16 >> 2)) {
                        const float nw = (float)(rows_in * cols_in);
                        const int tile_in_sample = (tp0.ty0 / PH) * (int)p.pt_div[1] + tp0.tx0 / PW;
                        float* row = p.tile_stats + (((long)tp0.n * p.tile_stats_ts + tile_in_sample * 2 + wpx) * 4) * p.Co;
#pragma unroll
                        for (int k = 0; k < TC; k++) {
                            const int ch = tp0.co0 + wc * (TC * 16) + k * 16 + r16;
                            if (ch < p.Co) {
                                const float S = acc_s[k][r16 & 3], Q = acc_q[k][r16 & 3];
                                // shift: the wave's first pixel (zero when the whole half is outside the image)
                                const float kf = to_f32<T>(reinterpret_cast<const T*>(stage_w)[k * 16 + r16]);
                                row[ch] = kf;
                                row[p.Co + ch] = S - nw * kf;
                                row[2 * p.Co + ch] = fmaxf(Q - 2.f * kf * S + nw * kf * kf, 0.f);
                                row[3 * p.Co + ch] = nw;
                            }
                        }
                    }
                    lds_wait();
                    asm volatile("" ::: "memory");
                }
                continue;
            }
            const long img = (long)tp0.n * H;
            conv_epilogue_lds<T, TC, TP, 2, 2, 256, TPG, true>(p, acc, tp0.co0, stage, [&](int pl) -> long {
                const int f = pl >> 4;
                const int oy = tp0.ty0 + f / FPR, ox = tp0.tx0 + (f % FPR) * 16 + (pl & 15);
                return oy < H && ox < W ? ((img + oy) * W + ox) * p.Co : -1;
            });
        }
    }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
static bool patch_eligible(const ConvGemmParams& p, int dtype) {
    const int bk = dtype == STMI_BF16 ? 64 : 32;
    if (p.ntaps != 9 || p.cdiv != 0 || p.klen <= 0 || p.klen % bk != 0 || p.klen != p.Ci) return false;
    if ((p.w_batch_stride != 0 && !p.clamp_t) || p.in_stride != 1 || p.oys != 1 || p.oxs != 1 || p.oy0 != 0 || p.ox0 != 0) return false;
    if (p.Hs != p.Hi || p.Ws != p.Wi || p.Ho != p.Hi || p.Wo != p.Wi) return false;
    if (p.co_pad % 32 != 0) return false;
    if ((long)p.N * p.Hi * p.Wi * p.Ci >= (1L << 31)) return false;
    for (int t = 0; t < 9; t++)
        if (p.taps[t].dy < -1 || p.taps[t].dy > 1 || p.taps[t].dx < -1 || p.taps[t].dx > 1) return false;
    return true;
}

// Patch width: 4 x 32 output pixels on wide images (6 x 34 = 204 halo rows per chunk; the 2 x 64 shape needs
// 4 x 66 = 264 and measured 1-2 % slower on every layer), the image width class below that.
// STMI_PATCH_PW=16|32|64 caps it (tuning aid).
static int patch_width(const ConvGemmParams& p) {
    const int cap = STMI_ENV_INT("STMI_PATCH_PW", 32);
    const int w = p.Wi > 32 ? 64 : (p.Wi > 16 ? 32 : 16);
    return w > cap && (cap == 16 || cap == 32) ? cap : w;
}

static long patch_tiles(const ConvGemmParams& p, int blk_co) {
    const int pw = patch_width(p), ph = 128 / pw;
    return (long)p.N * ceil_div(p.Hi, ph) * ceil_div(p.Wi, pw) * (p.co_pad / blk_co);
}

// The widest channel tile (128, 64, 32) that still spreads the layer over the chip; the narrow
// ones re-fetch the patch more often.  STMI_PATCH_CO=32|64|128 forces the choice (tuning aid).
static int patch_blk_co(const ConvGemmParams& p, int dtype) {
    {
        const int c = STMI_ENV_INT("STMI_PATCH_CO", 0);
        if ((c == 32 || c == 64 || c == 128) && p.co_pad % c == 0) return c;
    }
    // single-chunk layers are all prologue and epilogue: more, smaller blocks hide more of it
    const bool one_chunk = p.klen == (dtype == STMI_BF16 ? 64 : 32);
    if (p.co_pad % 128 == 0 && patch_tiles(p, 128) >= 256 && !one_chunk) return 128;
    if (p.co_pad % 64 == 0 && patch_tiles(p, 64) >= 64) return 64;
    return 32;
}

template <typename T, int TC, int PW, int NSW, bool ONE>
static int launch_patch(const ConvGemmParams& p, hipStream_t s) {
    constexpr size_t kLds = (size_t)NSW * 32 * TC * 128 + 2 * 4 * kPatchPieces * 8 * 128;
    STMI_DYN_LDS((conv_patch_kernel<T, TC, PW, NSW, ONE>), (int)kLds, "hipFuncSetAttribute(conv_patch_kernel)");
    const long tiles = patch_tiles(p, 32 * TC);
    if (tiles <= 0) return STMI_OK;
    ConvGemmParams q = p;
    {
        constexpr int PH = 128 / PW;
        const unsigned div[3] = {(unsigned)(p.co_pad / (32 * TC)), (unsigned)ceil_div(p.Wi, PW), (unsigned)ceil_div(p.Hi, PH)};
        for (int i = 0; i < 3; i++) {
            q.pt_div[i] = div[i];
            // ceil(2^32 / d) gives exact quotients while dividend * d < 2^32
            const bool fits = div[i] > 1 && (unsigned long)tiles * div[i] < (1UL << 32);
            q.pt_magic[i] = fits ? (unsigned)(((1UL << 32) + div[i] - 1) / div[i]) : 0u;
        }
    }
    STMI_REQUIRE(tiles < (1L << 31), "conv: too many tiles for one launch");
    // The MFMA waves' own epilogue (patch_wave_epilogue): plain bf16 convolutions on tiles of 64 / 128 channels that
    // leave tile statistics (10.2 -> 9.7 us on a residual layer of a training batch).  Launches without statistics
    // gain nothing from it (+-0.5 us, measured on every VGG layer: their tail is the stores' drain, not the
    // barriers), and single-chunk layers lose their second resident block to its staging rows.
    // STMI_PATCH_WEPI1=0: off (tuning aid).
    if constexpr (sizeof(T) == 2 && (TC == 2 || TC == 4)) {
        q.pt_wepi = !ONE && p.row_store_ok && !p.mask && !p.addend && !p.relu && !p.out_pre && p.tile_stats &&
                    p.tile_stats_mode == 1 && STMI_ENV_INT("STMI_PATCH_WEPI1", 1) != 0;
    }
    // a layer with a single channel chunk never touches the second patch buffer
    const size_t lds_bytes = ONE ? kLds - 4 * kPatchPieces * 8 * 128 : kLds;
    const long blocks = tiles * (q.pt_ksplit > 1 ? q.pt_ksplit : 1);
    STMI_REQUIRE(blocks < (1L << 31) && !(ONE && q.pt_ksplit > 1), "conv: bad split-K launch");
    if (q.pt_ksplit > 1) q.pt_wepi = 0;
    hipLaunchKernelGGL((conv_patch_kernel<T, TC, PW, NSW, ONE>), dim3((unsigned)blocks), dim3(512), lds_bytes, s, q);
    STMI_LAUNCH_CHECK("conv_patch_kernel");
    return STMI_OK;
}

template <typename T, int TC>
static int launch_patch_w(const ConvGemmParams& p, hipStream_t s) {
    const bool one = p.klen == 128 / (int)sizeof(T);
    switch (patch_width(p)) {
        case 64: return one ? launch_patch<T, TC, 64, 4, true>(p, s) : launch_patch<T, TC, 64, 4, false>(p, s);
        case 32: return one ? launch_patch<T, TC, 32, 4, true>(p, s) : launch_patch<T, TC, 32, 4, false>(p, s);
        default: return one ? launch_patch<T, TC, 16, 4, true>(p, s) : launch_patch<T, TC, 16, 4, false>(p, s);
    }
}

// Compute units of the current device (the persistent kernel's grid), looked up once per device.
static int device_cus() {
    constexpr int kMaxDev = 64;
    static std::atomic<int> cus[kMaxDev];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return 256;
    int v = cus[dev].load(std::memory_order_acquire);
    if (v == 0) {
        hipDeviceProp_t prop;
        v = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
        if (v < 8) v = 8;
        cus[dev].store(v, std::memory_order_release);
    }
    return v;
}

// The persistent form pays from three block rounds per CU on (measured: 64 rounds, 1080p residual layers,
// -8 % per launch inside the pipeline, -14 % alone; two rounds, conv2_2 of a training batch: +0.4 % on the step).
// STMI_PATCH_PERS: fewest rounds that take it, 0 = never.
static bool patch_persistent_shape(const ConvGemmParams& p, int dtype, int blk_co, long tiles) {
    const int min_rounds = STMI_ENV_INT("STMI_PATCH_PERS", 3);
    if (min_rounds <= 0 || blk_co < 64 || !p.row_store_ok || patch_width(p) > 32) return false;
    if (dtype != STMI_BF16 && blk_co > 64) return false;   // fp32 fragments + the tile walk do not fit 256 VGPRs
    if (p.klen == (dtype == STMI_BF16 ? 64 : 32)) return false;   // single-chunk layers keep two resident blocks per CU
    return tiles >= (long)min_rounds * (device_cus() / 8 * 8);
}

// A geometry whose launches are persistent at ANY batch size (one sample alone has the rounds): there the
// persistent kernel's DMA waves leave the tile statistics almost for free.  Depends on the plane and the channel
// counts only, so that a sample's result never depends on the batch it is part of.
static bool patch_stats_persistent(const ConvGemmParams& p, int dtype) {
    const int co = patch_blk_co(p, dtype);
    ConvGemmParams one = p;
    one.N = 1;
    return patch_persistent_shape(p, dtype, co, patch_tiles(one, co));
}

// Persistent-statistics geometries whose tiles are 128 channels wide in bf16 run the wave-private epilogue, which
// leaves TWO statistics rows per tile (one per pixel half).  Geometry only; STMI_PATCH_WEPI=0: off.
static bool patch_stats_two_rows(const ConvGemmParams& p, int dtype) {
    return dtype == STMI_BF16 && patch_blk_co(p, dtype) == 128 && p.row_store_ok && patch_stats_persistent(p, dtype) &&
           STMI_ENV_INT("STMI_PATCH_WEPI", 1) != 0;
}

static bool patch_persistent(const ConvGemmParams& p, int dtype, int blk_co) {
    if (p.tile_stats) {
        // forward statistics of a plain convolution only; the decision must not depend on the batch size
        if (p.tile_stats_mode != 1 || p.mask || p.addend || p.relu) return false;
        return patch_stats_persistent(p, dtype);
    }
    return patch_persistent_shape(p, dtype, blk_co, patch_tiles(p, blk_co));
}

template <typename T, int TC, int PW, int NSW, bool STATS, bool WEPI = false, bool CLAMP = false>
static int launch_patch_pers(const ConvGemmParams& p, hipStream_t s) {
    constexpr int PH = 128 / PW;
    constexpr size_t kPatch = (size_t)4 * (((PH + 2) * (PW + 2) + 31) / 32) * 8 * 128;
    // WEPI: four waves' rows of 16 TC channels of T (+ 16 bytes)
    constexpr size_t kStage = WEPI ? (size_t)4 * 64 * (16 * TC * sizeof(T) + 16)
                                    : (size_t)16 * (TC >= 4 ? 2 : 4) * 2 * (32 * TC + 4) * 4;
    // CLAMP: a 256-byte table of clamp values per MFMA wave behind the staging rows
    constexpr size_t kLds = (size_t)NSW * 32 * TC * 128 + 2 * kPatch + kStage + (CLAMP ? 4 * 256 : 0);
    static_assert(kLds <= 160 * 1024, "operand buffers + staging rows exceed the CU's LDS");
    STMI_DYN_LDS((conv_patch_pers_kernel<T, TC, PW, NSW, STATS, WEPI, CLAMP>), (int)kLds, "hipFuncSetAttribute(conv_patch_pers_kernel)");
    const long tiles = patch_tiles(p, 32 * TC);
    STMI_REQUIRE(tiles > 0 && tiles < (1L << 31), "conv: tile count out of range");
    ConvGemmParams q = p;
    const unsigned div[3] = {(unsigned)(p.co_pad / (32 * TC)), (unsigned)ceil_div(p.Wi, PW), (unsigned)ceil_div(p.Hi, PH)};
    for (int i = 0; i < 3; i++) {
        q.pt_div[i] = div[i];
        const bool fits = div[i] > 1 && (unsigned long)tiles * div[i] < (1UL << 32);
        q.pt_magic[i] = fits ? (unsigned)(((1UL << 32) + div[i] - 1) / div[i]) : 0u;
    }
    q.pt_npers = (unsigned)tiles;
    unsigned grid = (unsigned)(device_cus() / 8 * 8);   // a multiple of the eight XCDs
    if ((long)grid > tiles) grid = (unsigned)((tiles + 7) / 8 * 8);
    hipLaunchKernelGGL((conv_patch_pers_kernel<T, TC, PW, NSW, STATS, WEPI, CLAMP>), dim3(grid), dim3(512), kLds, s, q);
    STMI_LAUNCH_CHECK("conv_patch_pers_kernel");
    return STMI_OK;
}

// The wave-private epilogue (WEPI, see the kernel): plain bf16 convolutions on 128-channel tiles.  STMI_PATCH_WEPI=0: off.
static bool patch_wepi(const ConvGemmParams& p, int dtype, int blk_co) {
    if (dtype != STMI_BF16 || blk_co != 128 || !p.row_store_ok) return false;
    if (p.bias || p.mask || p.addend || p.relu || p.out_pre) return false;
    if (p.tile_stats && p.tile_stats_mode != 1) return false;
    if ((long)p.N * p.Ho * p.Wo * p.Co >= (1L << 31)) return false;
    return STMI_ENV_INT("STMI_PATCH_WEPI", 1) != 0;
}

template <typename T, int TC>
static int launch_patch_pers_w(const ConvGemmParams& p, hipStream_t s) {
    if constexpr (sizeof(T) == 2 && TC == 4) {
        if (patch_wepi(p, STMI_BF16, 32 * TC)) {
            if (p.tile_stats)
                return patch_width(p) == 32 ? launch_patch_pers<T, TC, 32, 4, true, true>(p, s) : launch_patch_pers<T, TC, 16, 4, true, true>(p, s);
            return patch_width(p) == 32 ? launch_patch_pers<T, TC, 32, 4, false, true>(p, s) : launch_patch_pers<T, TC, 16, 4, false, true>(p, s);
        }
    }
    if (p.tile_stats)
        return patch_width(p) == 32 ? launch_patch_pers<T, TC, 32, 4, true>(p, s) : launch_patch_pers<T, TC, 16, 4, true>(p, s);
    return patch_width(p) == 32 ? launch_patch_pers<T, TC, 32, 4, false>(p, s) : launch_patch_pers<T, TC, 16, 4, false>(p, s);
}

// Tiles per sample the patch kernel would cut this layer into (the number of statistics rows per sample of
// ConvGemmParams::tile_stats), or 0 when the layer does not run on the patch kernel with row-wise stores.
int conv_patch_tiles_per_sample(const ConvGemmParams& p, int dtype, int mode) {
    if (!patch_eligible(p, dtype) || STMI_ENV_SET("STMI_NO_PATCH") || !p.row_store_ok) return 0;
    const int pw = patch_width(p), ph = 128 / pw;
    const int tiles = ceil_div(p.Hi, ph) * ceil_div(p.Wi, pw);
    return mode == 1 && patch_stats_two_rows(p, dtype) ? 2 * tiles : tiles;
}

// 1 when the statistics rows of this geometry come at no cost to the launch (the persistent kernel)
int conv_patch_stats_free(const ConvGemmParams& p, int dtype) {
    return patch_eligible(p, dtype) && !STMI_ENV_SET("STMI_NO_PATCH") && p.row_store_ok && patch_stats_persistent(p, dtype);
}

// Second launch of a split-K convolution: out = epilogue(sum of the slices' planes, in slice order).
template <typename T>
__global__ __launch_bounds__(256) void conv_ksplit_fold_kernel(const float* __restrict__ ws, long plane, int ksplit,
                                                               T* __restrict__ out, T* __restrict__ out_pre,
                                                               const float* __restrict__ bias, const T* __restrict__ mask,
                                                               const T* __restrict__ addend, float alpha, int relu,
                                                               long pieces, int Co) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long idx = blockIdx.x * 256L + threadIdx.x;   // 16-byte piece of the written tensor ([pixel][Co], Co % EPC == 0)
    if (idx >= pieces) return;
    const long e = idx * EPC;
    float v[EPC];
#pragma unroll
    for (int u = 0; u < EPC; u++) v[u] = 0.f;
    for (int sl = 0; sl < ksplit; sl++) {
        float t[EPC];
        load_vec<float, EPC>(ws + sl * plane + e, t);
#pragma unroll
        for (int u = 0; u < EPC; u++) v[u] += t[u];
    }
    const int c = (int)(e % Co);
#pragma unroll
    for (int u = 0; u < EPC; u++) v[u] = (v[u] + (bias ? bias[c + u] : 0.f)) * alpha;
    if (mask) {
        float mk[EPC];
        load_vec<T, EPC>(mask + e, mk);
#pragma unroll
        for (int u = 0; u < EPC; u++) v[u] = mk[u] > 0.f ? v[u] : 0.f;
    }
    if (addend) {
        float ad[EPC];
        load_vec<T, EPC>(addend + e, ad);
#pragma unroll
        for (int u = 0; u < EPC; u++) v[u] += ad[u];
    }
    if (out_pre) store_vec<T, EPC>(out_pre + e, v);
    if (relu) {
#pragma unroll
        for (int u = 0; u < EPC; u++) v[u] = fmaxf(v[u], 0.f);
    }
    store_vec<T, EPC>(out + e, v);
}

// K slices for a layer whose tiles do not fill the chip (the 16 x 16 maps of VGG conv5_x at 256 x 256 pictures:
// 64 tiles of 144 K steps on 256 CUs); 1 = no split.  A split shortens the K loop every block walks, and pays
// for it with a second launch and with the planes of fp32 partial sums written and read back; measured on the
// VGG shapes (scripts/bench_ksplit.py), per call:
//   saved   = chunks * (1 - 1/ks) * 1.9 us   (bf16, LDS-bound whatever the tile's channel count; fp32, bound by
//             its matrix rate: 15 us per chunk on 128-channel tiles)
//   paid    = 4.5 us + 8 bytes * ks * outputs / 3 TB/s
// Among the powers of two that divide the chunk count, keep two chunks per slice, stay within one block per CU
// and fit the workspace, the one with the largest net saving is taken, if that is above 1 us.
static int patch_ksplit(const ConvGemmParams& p, int dtype, int blk_co) {
    if (!p.ksplit_ws || !p.row_store_ok || p.tile_stats || STMI_ENV_SET("STMI_NO_KSPLIT")) return 1;
    const int chunks = p.klen / (dtype == STMI_BF16 ? 64 : 32);
    const long tiles = patch_tiles(p, blk_co);
    const long plane = (long)p.N * p.Ho * p.Wo * p.Co;
    const int forced = STMI_ENV_INT("STMI_KSPLIT", 0);   // measurement knob: any divisor of the chunk count
    if (forced >= 1) return chunks % forced == 0 && plane * forced <= p.ksplit_ws_floats ? forced : 1;
    const double chunk_us = dtype == STMI_BF16 ? 1.9 : 15.0 * blk_co / 128.0;
    int best = 1;
    double best_net = 1.0;
    for (int ks = 2; tiles * ks <= 256 && chunks / ks >= 2 && chunks % ks == 0 && plane * ks <= p.ksplit_ws_floats; ks *= 2) {
        const double net = chunks * (1.0 - 1.0 / ks) * chunk_us - (4.5 + 8.0 * ks * (double)plane / 3e6);
        if (net > best_net) best = ks, best_net = net;
    }
    return best;
}

long conv_patch_ksplit_ws_floats(const ConvGemmParams& p_in, int dtype) {
    if (!patch_eligible(p_in, dtype) || STMI_ENV_SET("STMI_NO_PATCH")) return 0;
    ConvGemmParams p = p_in;
    p.ksplit_ws = reinterpret_cast<float*>(16);   // any non-null value: the plan below only compares sizes
    p.ksplit_ws_floats = 1L << 60;
    const int co = patch_blk_co(p, dtype);
    if (patch_persistent(p, dtype, co)) return 0;
    const int ks = patch_ksplit(p, dtype, co);
    return ks > 1 ? (long)ks * p.N * p.Ho * p.Wo * p.Co : 0;
}

template <typename T>
static int launch_ksplit_fold(const ConvGemmParams& p, int ksplit, hipStream_t s) {
    constexpr int EPC = 16 / (int)sizeof(T);
    const long plane = (long)p.N * p.Ho * p.Wo * p.Co, pieces = plane / EPC;
    hipLaunchKernelGGL(conv_ksplit_fold_kernel<T>, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, s, p.ksplit_ws, plane,
                       ksplit, (T*)p.out, (T*)p.out_pre, p.bias, (const T*)p.mask, (const T*)p.addend, p.alpha, p.relu,
                       pieces, p.Co);
    STMI_LAUNCH_CHECK("conv_ksplit_fold_kernel");
    return STMI_OK;
}

// The one-tile kernel at channel tile `co`; true (the launch was made or refused with *rc set).
static bool launch_patch_any(const ConvGemmParams& p, int dtype, int co, int ksplit, hipStream_t s, int* rc, char* name,
                             int name_len, const char* suffix) {
    (void)ksplit;
    if (name)
        snprintf(name, name_len, "void stmi::conv_patch_kernel<%s, %d, %d, 4, %s>(stmi::ConvGemmParams)%s",
                 dtype == STMI_BF16 ? "__bf16" : "float", co / 32, patch_width(p),
                 p.klen == (dtype == STMI_BF16 ? 64 : 32) ? "true" : "false", suffix);
    if (dtype == STMI_BF16)
        *rc = co == 128 ? launch_patch_w<bf16_t, 4>(p, s)
                        : (co == 64 ? launch_patch_w<bf16_t, 2>(p, s) : launch_patch_w<bf16_t, 1>(p, s));
    else
        *rc = co == 128 ? launch_patch_w<float, 4>(p, s)
                        : (co == 64 ? launch_patch_w<float, 2>(p, s) : launch_patch_w<float, 1>(p, s));
    return true;
}

bool try_launch_conv_patch(const ConvGemmParams& p_in, int dtype, hipStream_t s, int* rc, char* name, int name_len) {
    if (!patch_eligible(p_in, dtype) || STMI_ENV_SET("STMI_NO_PATCH")) return false;
    ConvGemmParams p = p_in;
    const int co = patch_blk_co(p, dtype);
    if (p.clamp_t) {
        // the absorbed-norm form exists as the persistent bf16 kernel with the wave-private epilogue and statistics
        if (dtype != STMI_BF16 || co != 128 || !patch_wepi(p, dtype, co) || !p.tile_stats || p.tile_stats_mode != 1 ||
            !patch_stats_two_rows(p, dtype) || p.Ci > 128 || p.w_batch_stride <= 0 || !p.clamp_bias || p.Co % 128 != 0 ||
            (((uintptr_t)p.clamp_t | (uintptr_t)p.clamp_bias) & 15) != 0 || p.pool_out || p.ksplit_ws)
            return false;
        if (name)
            snprintf(name, name_len, "void stmi::conv_patch_pers_kernel<__bf16, 4, %d, 4, true, true, true>(stmi::ConvGemmParams)",
                     patch_width(p));
        *rc = patch_width(p) == 32 ? launch_patch_pers<bf16_t, 4, 32, 4, true, true, true>(p, s)
                                   : launch_patch_pers<bf16_t, 4, 16, 4, true, true, true>(p, s);
        return true;
    }
    if (p.pool_out) {
        // the pool comes from the staged rows of the one-tile kernel's row-wise epilogue
        // (launches long enough for the persistent kernel gain more from it than from the saved pool launch:
        // conv2_2 of eight 256 x 256 pictures 45.9 + 6.9 us in two launches against 56.2 us fused in one-tile blocks)
        if (!p.row_store_ok || p.tile_stats || patch_persistent(p, dtype, co) || STMI_ENV_SET("STMI_NO_POOL_FUSE")) return false;
        return launch_patch_any(p, dtype, co, 1, s, rc, name, name_len, " [+pool]");
    }
    const int ksplit = patch_persistent(p, dtype, co) || p.klen == (dtype == STMI_BF16 ? 64 : 32) ? 1 : patch_ksplit(p, dtype, co);
    if (ksplit > 1) {
        // the slices leave raw sums; everything the epilogue would have applied moves to the fold launch
        p.pt_ksplit = ksplit;
        p.pt_kchunks = p.klen / (dtype == STMI_BF16 ? 64 : 32) / ksplit;
        p.ksplit_plane = (long)p.N * p.Ho * p.Wo * p.Co;
        p.bias = nullptr; p.mask = nullptr; p.addend = nullptr; p.out_pre = nullptr; p.relu = 0; p.alpha = 1.f;
    } else {
        p.pt_ksplit = 1;
    }
    if (p.tile_stats && p.tile_stats_mode == 1 && patch_stats_two_rows(p, dtype) && !patch_wepi(p, dtype, co)) {
        set_error("conv: the tile statistics of this geometry (two rows per tile) come from the plain convolution only: "
                  "no bias, ReLU, mask, addend or pre-activation copy, fewer than 2^31 output elements");
        *rc = STMI_ERR_INVALID;
        return true;
    }
    if (patch_persistent(p, dtype, co)) {
        if (name)
            snprintf(name, name_len, "void stmi::conv_patch_pers_kernel<%s, %d, %d, 4, %s, %s>(stmi::ConvGemmParams)",
                     dtype == STMI_BF16 ? "__bf16" : "float", co / 32, patch_width(p), p.tile_stats ? "true" : "false",
                     patch_wepi(p, dtype, co) ? "true" : "false");
        if (dtype == STMI_BF16) *rc = co == 128 ? launch_patch_pers_w<bf16_t, 4>(p, s) : launch_patch_pers_w<bf16_t, 2>(p, s);
        else *rc = launch_patch_pers_w<float, 2>(p, s);
        return true;
    }
    launch_patch_any(p, dtype, co, ksplit, s, rc, name, name_len, ksplit > 1 ? " [split K]" : "");
    if (*rc == STMI_OK && ksplit > 1)
        *rc = dtype == STMI_BF16 ? launch_ksplit_fold<bf16_t>(p_in, ksplit, s) : launch_ksplit_fold<float>(p_in, ksplit, s);
    return true;
}


}  // namespace stmi

#ifdef STMI_STAMPS
extern "C" int stmi_debug_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(stmi::g_stamps), sizeof(unsigned long long) * n);
}
#endif
