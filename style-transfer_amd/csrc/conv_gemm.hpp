// Shared pieces of the gather-GEMM convolution kernels: launch parameters and the epilogue.
#pragma once
#include "common.hpp"

namespace stmi {

struct ConvGemmParams {
    const void* in;
    const void* wp;
    void* out;
    void* out_pre;
    const float* bias;
    const void* mask;
    const void* addend;
    int N, Hi, Wi, Ci;  // gathered tensor
    int Ho, Wo, Co;     // written tensor
    int Hs, Ws, oy0, ox0, oys, oxs;  // output sub-grid
    int in_stride;      // input base position = sub-grid index * in_stride
    int ntaps, klen, kpad, cdiv, co_pad;
    long w_batch_stride;  // != 0: one weight set per sample (Gram backward)
    float alpha;
    int relu;
    int row_store_ok;  // Co and the tensors' alignment allow whole-row 16-byte stores (set by the launcher)
    unsigned pt_div[3], pt_magic[3];  // patch kernel: channel tiles, tiles per row, tile rows per image (+ reciprocals)
    unsigned pt_npers;                // persistent patch kernel: tiles of the launch
    int pt_wepi;                      // one-tile patch kernel: the MFMA waves finish the tile among themselves
    // Split K (one-tile patch kernel, layers with too few tiles to fill the chip): pt_ksplit > 1 slices of
    // pt_kchunks channel chunks each; slice s of a tile leaves its raw fp32 sums in plane s of ksplit_ws
    // (ksplit_plane floats apart, laid out like the written tensor) and conv_ksplit_fold adds the planes in slice
    // order and applies bias, alpha, mask, addend, pre-activation copy and ReLU.  ksplit_ws / ksplit_ws_floats:
    // the caller's workspace (nullptr: never split).
    int pt_ksplit, pt_kchunks;
    long ksplit_plane;
    float* ksplit_ws;
    long ksplit_ws_floats;
    // != nullptr (one-tile patch kernel, row-wise epilogue): the block also leaves the 2 x 2 / stride 2 'SAME' max
    // pool of the values it stores, [N][(Ho + 1) / 2][(Wo + 1) / 2][Co], taken from its staged rows
    void* pool_out;
    // != nullptr (persistent patch kernel, bf16, wave-private epilogue, at most 128 input channels): the input is the
    // raw input of an instance norm + ReLU that this convolution absorbs.  With a = scale * rstd > 0, b = shift -
    // mean * a, t = -b / a:  relu(a x + b) = a * max(x, t) + b, so the launch reads max(x, clamp_t[n][ci]) (one packed
    // max per fragment register), takes t for taps outside the image (a t + b = 0: the zero padding of the
    // normalised tensor) and multiplies with per-sample weights W * a (w_batch_stride apart).  The constant term
    // sum(W b) = -sum(W a t) per (sample, output channel) is what the accumulators start from: clamp_bias[n][Co].
    const void* clamp_t;
    const float* clamp_bias;
    Tap taps[kMaxTaps];
    // Several output sub-grids in ONE launch (the stride^2 parity classes of a strided backward-data /
    // transposed convolution; producer/consumer kernel only).  nclasses == 0: the fields above describe the
    // only sub-grid.  Otherwise class c covers block tiles [cls[c-1].tile_end, cls[c].tile_end), has its own
    // sub-grid and uses taps[tap0 .. tap0 + ntaps) of the shared table.
    struct SubGrid {
        int Hs, Ws, oy0, ox0, ntaps, tap0;
        unsigned tile_end;
    };
    int nclasses;
    SubGrid cls[4];
    // Per-tile sums of the written tensor for the instance norm next to this convolution (LDS-resident patch
    // kernel only; tile_stats == nullptr: off), so that the norm needs no reduction pass of its own over the
    // tensor.  tile_stats_ts = tiles per sample.
    //   mode 1 (forward, the norm FOLLOWS the convolution): tile_stats[sample][tile][plane][channel], planes =
    //          shift k, sum(y - k), sum((y - k)^2), pixel count
    //   mode 2 (backward-data, the written tensor is the gradient g w.r.t. the OUTPUT of the norm below, whose
    //          input bn_x has the written tensor's shape): tile_stats[sample][tile][2][channel] =
    //          sum(g * act'(z)), sum(g * act'(z) * xhat), xhat = (bn_x - mean) * rstd, z = scale * xhat + shift --
    //          the partial rows the norm's backward apply pass folds.
    float* tile_stats;
    int tile_stats_ts, tile_stats_mode;
    const void* bn_x;
    const float *bn_stats, *bn_scale, *bn_shift;  // stats [n][c][2] = (mean, rstd)
    int bn_act;
};

// Where one block's sums go (element (plane, local channel c) is base[plane * C + c]) and what they are of.
struct TileStatsOut {
    float* base;      // nullptr = no statistics
    int C, count;     // channels of the tensor; pixels of this tile inside the image
    int mode;         // ConvGemmParams::tile_stats_mode
    const float* stats_n;  // mode 2: statistics of this block's sample
};

// Epilogue shared by the kernels.  Accumulator tile (tc, tp) of a wave holds, in lane
// (g = lane>>4, r16 = lane&15), channels co..co+3 (co = co0 + (wc*TC+tc)*16 + 4g) of the
// lane's pixel column tp;  pixel_offset(tp) gives the element offset of that pixel in the
// written tensor, or a negative value when the column is outside it:
//   v = (acc + bias) * alpha ; masked ; + addend ; optional pre-ReLU copy ; optional ReLU ; store.
template <typename T, int TC, int TP, typename PixelOffset>
__device__ __forceinline__ void conv_epilogue_at(const ConvGemmParams& p, const f32x4 (&acc)[TC][TP], int co0, int wc,
                                                 int g, PixelOffset pixel_offset) {
    T* __restrict__ out = (T*)p.out;
    T* __restrict__ out_pre = (T*)p.out_pre;
    const T* __restrict__ mask = (const T*)p.mask;
    const T* __restrict__ addend = (const T*)p.addend;
    const float* __restrict__ bias = p.bias;
    const bool vec_ok = (p.Co & 3) == 0;
    // Loads first, stores last: a load issued between two stores waits for its data with
    // the stores still queued behind it, and sixteen such round trips cost microseconds.
    float bias_r[TC][4];
#pragma unroll
    for (int tc = 0; tc < TC; tc++) {
        const int co = co0 + (wc * TC + tc) * 16 + 4 * g;
#pragma unroll
        for (int u = 0; u < 4; u++) bias_r[tc][u] = bias && co + u < p.Co ? bias[co + u] : 0.f;
    }
#pragma unroll
    for (int tp = 0; tp < TP; tp++) {
        const long off = pixel_offset(tp);
        if (off < 0) continue;
        if (vec_ok) {
            float mk[TC][4], ad[TC][4];
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
                const int co = co0 + (wc * TC + tc) * 16 + 4 * g;
                if (co >= p.Co) continue;
                if (mask) load_vec<T, 4>(mask + off + co, mk[tc]);
                if (addend) load_vec<T, 4>(addend + off + co, ad[tc]);
            }
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
                const int co = co0 + (wc * TC + tc) * 16 + 4 * g;
                if (co >= p.Co) continue;
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (acc[tc][tp][u] + bias_r[tc][u]) * p.alpha;
                if (mask) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = mk[tc][u] > 0.f ? v[u] : 0.f;
                }
                if (addend) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] += ad[tc][u];
                }
                if (out_pre) store_vec<T, 4>(out_pre + off + co, v);
                if (p.relu) {
#pragma unroll
                    for (int u = 0; u < 4; u++) v[u] = fmaxf(v[u], 0.f);
                }
                store_vec<T, 4>(out + off + co, v);
            }
        } else {
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
                const int co = co0 + (wc * TC + tc) * 16 + 4 * g;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (co + u >= p.Co) continue;
                    float x = (acc[tc][tp][u] + bias_r[tc][u]) * p.alpha;
                    if (mask) x = to_f32<T>(mask[off + co + u]) > 0.f ? x : 0.f;
                    if (addend) x += to_f32<T>(addend[off + co + u]);
                    if (out_pre) out_pre[off + co + u] = from_f32<T>(x);
                    if (p.relu) x = fmaxf(x, 0.f);
                    out[off + co + u] = from_f32<T>(x);
                }
            }
        }
    }
}

// The gather-GEMM kernels number their pixel columns linearly over the output sub-grid:
// column tp of the wave is pixel pix0 + (wpx*TP+tp)*16 + r16.
template <typename T, int TC, int TP>
__device__ __forceinline__ void conv_epilogue(const ConvGemmParams& p, const f32x4 (&acc)[TC][TP], int co0, long pix0,
                                              int wc, int wpx, int g, int r16, long Mz, long zoff, int HWs) {
    conv_epilogue_at<T, TC, TP>(p, acc, co0, wc, g, [&](int tp) -> long {
        const long m = pix0 + (wpx * TP + tp) * 16 + r16;
        if (m >= Mz) return -1;
        const long mm = m + zoff;
        const int n = (int)(mm / HWs);
        const int rem = (int)(mm - (long)n * HWs);
        const int i = rem / p.Ws, j = rem - i * p.Ws;
        return (((long)n * p.Ho + (p.oy0 + p.oys * i)) * p.Wo + (p.ox0 + p.oxs * j)) * p.Co;
    });
}

// Epilogue through LDS for the kernels whose block tile is done with its LDS: the
// accumulators' native layout gives each lane 4 channels of one pixel, i.e. 8-byte stores
// into sixteen different pixel rows per wave-instruction, and sixteen of those per lane
// cost about 5 us of store issue.  Staged as fp32 [pixel][channel] in LDS and read back
// row-wise, every wave-instruction instead stores whole pixel rows, 16 bytes per lane.
// The tile goes through the staging area TPG pixel fragments (per wave) at a time, so the
// area is 16*TPG*WP * (BLK_CO + 4) floats: conv_stage_tpg() picks the largest TPG that fits.
// Needs Co a multiple of the 16-byte vector and 16-byte aligned tensors (p.row_store_ok);
// pixel_offset(local pixel) is the element offset in the written tensor or negative.  All NT
// threads of the block must call it (waves past WC*WP only store).
constexpr int conv_stage_tpg(int tc, int tp, int wc, int wp, int lds_bytes) {
    int best = 0;
    for (int g = 1; g <= tp; g++)
        if (tp % g == 0 && 16 * g * wp * (16 * tc * wc + 4) * 4 <= lds_bytes) best = g;
    return best;
}

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() also waits for the wave's global
// memory traffic (vmcnt(0)), and on this chip that counter includes STORES until they are acknowledged: placed
// behind a tile's output stores it would cost their round trip to memory (about 2 us per block, measured).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float conv_act_grad(float z, int act) {
    if (act == STMI_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == STMI_ACT_SIGMOID) {
        const float sg = 1.f / (1.f + __expf(-z));
        return sg * (1.f - sg);
    }
    return 1.f;
}

// With ts_out.base set the block also leaves per-channel sums over its tile (see ConvGemmParams::tile_stats):
// the finished values go back into the staging rows, and column-owner threads (consecutive threads =
// consecutive channels: conflict-free LDS reads, coalesced global loads) add them up over the rows.  That needs
// a single staging group (TPG == TP) and 8 * NT + ROWS words of LDS behind the staging rows.
// LDSB: the block's barriers wait for LDS traffic only (lds_barrier), for a caller whose other waves are
// busy loading and must not wait for this tile's output stores; the caller then guarantees that no LDS-DMA
// targets the staging rows.
template <typename T, int TC, int TP, int WC, int WP, int NT, int TPG, bool LDSB = false, typename PixelOffset>
__device__ __forceinline__ void conv_epilogue_lds(const ConvGemmParams& p, const f32x4 (&acc)[TC][TP], int co0,
                                                  float* stage, PixelOffset pixel_offset,
                                                  TileStatsOut ts_out = TileStatsOut{nullptr, 0, 0, 0, nullptr},
                                                  float* raw_out = nullptr) {
    constexpr int EPC = 16 / sizeof(T);
    constexpr int BLK_CO = 16 * TC * WC;
    constexpr int STRIDE = BLK_CO + 4;   // floats; +16 bytes keeps the 16-byte writes conflict-free
    constexpr int LPR = BLK_CO / EPC;    // lanes per pixel row
    constexpr int RPP = NT / LPR;        // pixel rows per pass of the block
    constexpr int ROWS = 16 * TPG * WP;  // pixel rows per group
    static_assert(TPG >= 1 && TP % TPG == 0, "fragment groups must tile the wave's pixels");
    static_assert(BLK_CO % EPC == 0 && NT % LPR == 0, "row pieces must tile the block");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wpx = wave % WP;
    const int g = lane >> 4, r16 = lane & 15;
    T* __restrict__ out = (T*)p.out;
    T* __restrict__ out_pre = (T*)p.out_pre;
    const T* __restrict__ mask = (const T*)p.mask;
    const T* __restrict__ addend = (const T*)p.addend;
    const int cl_rd = (tid % LPR) * EPC;
    const bool col_ok = co0 + cl_rd < p.Co;
    const bool want_stats = TPG == TP && ts_out.base != nullptr;   // block-uniform
    // the sums (and a fused pool) are of the FINISHED values
    const bool restage = (want_stats || (TPG == TP && p.pool_out != nullptr)) && (mask || addend || p.relu);
    constexpr int CT = BLK_CO / 4;                                  // column threads per staged row (4 channels each)
    constexpr int NRG = CT <= NT ? NT / CT : 1;                     // row groups: column thread = (row group, 4 channels)
    float* part = stage + ROWS * STRIDE;                            // [NRG][2][BLK_CO]
    int* rowoff = reinterpret_cast<int*>(part + 8 * NT);            // [ROWS] element offset of a staged row, or < 0

    f32x4 bias_r[TC];
#pragma unroll
    for (int tc = 0; tc < TC; tc++) {
        bias_r[tc] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int co = co0 + (wc * TC + tc) * 16 + 4 * g;
        if (p.bias && wave < WC * WP) {
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (co + u < p.Co) bias_r[tc][u] = p.bias[co + u];
        }
    }
#pragma unroll
    for (int tp0 = 0; tp0 < TP; tp0 += TPG) {
        if constexpr (LDSB) lds_barrier();
        else __syncthreads();  // operand tiles (first group) or the previous group's rows are consumed
        if (wave < WC * WP) {
#pragma unroll
            for (int tc = 0; tc < TC; tc++) {
                const int cl = (wc * TC + tc) * 16 + 4 * g;
#pragma unroll
                for (int t = 0; t < TPG; t++) {
                    const int rr = (wpx * TPG + t) * 16 + r16;
                    *reinterpret_cast<f32x4*>(stage + rr * STRIDE + cl) = (acc[tc][tp0 + t] + bias_r[tc]) * p.alpha;
                }
            }
        }
        if constexpr (LDSB) lds_barrier();
        else __syncthreads();
        if (!col_ok) continue;
        for (int rr = tid / LPR; rr < ROWS; rr += RPP) {
            const int w_ = rr / (16 * TPG), rem = rr - w_ * (16 * TPG);
            const int pl = (w_ * TP + tp0 + (rem >> 4)) * 16 + (rem & 15);
            const long off = pixel_offset(pl);
            if (want_stats && cl_rd == 0) rowoff[rr] = (int)off;   // element offsets fit 31 bits (launcher)
            if (off < 0) continue;
            const long e = off + co0 + cl_rd;
            float v[EPC];
#pragma unroll
            for (int u = 0; u < EPC; u += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(stage + rr * STRIDE + cl_rd + u);
                v[u] = t[0]; v[u + 1] = t[1]; v[u + 2] = t[2]; v[u + 3] = t[3];
            }
            if (raw_out) {   // a K slice's partial sums: fp32 rows of a workspace plane, finished by the fold kernel
                store_vec<float, EPC>(raw_out + e, v);
                continue;
            }
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
            if (p.relu) {
#pragma unroll
                for (int u = 0; u < EPC; u++) v[u] = fmaxf(v[u], 0.f);
            }
            if (restage) {
#pragma unroll
                for (int u = 0; u < EPC; u += 4)
                    *reinterpret_cast<f32x4*>(stage + rr * STRIDE + cl_rd + u) = (f32x4){v[u], v[u + 1], v[u + 2], v[u + 3]};
            }
            store_vec<T, EPC>(out + e, v);
        }
    }
    if constexpr (TPG == TP && CT <= NT && NT % CT == 0 && ROWS % NRG == 0 && EPC % 4 == 0) {
        if (want_stats) {
            constexpr int RPT = ROWS / NRG;        // staged rows per column thread
            const int col = (tid % CT) * 4, rg = tid / CT;
            const bool live = co0 + col < p.Co;    // channel counts are multiples of 4 here (row_store_ok)
            lds_barrier();                         // row offsets (and finished values) are in LDS
            // Every LDS read is issued before the first use (a loop that waits per row pays an LDS round
            // trip for each row: 2 us per block, measured), and the reads are 16 bytes wide: four channels
            // per thread keep the column phase a fraction of the LDS time of the row phase.
            f32x4 sv[RPT];
            int ro[RPT];
#pragma unroll
            for (int j = 0; j < RPT; j++) {
                sv[j] = *reinterpret_cast<const f32x4*>(stage + (rg + j * NRG) * STRIDE + col);
                ro[j] = rowoff[rg + j * NRG];
            }
            f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
            if (ts_out.mode == 1) {
                const f32x4 k = *reinterpret_cast<const f32x4*>(stage + col);  // the tile's first pixel: always in the image
#pragma unroll
                for (int j = 0; j < RPT; j++) {
                    const float m = ro[j] >= 0 ? 1.f : 0.f;   // selects, not branches: the rows stay one straight run
                    const f32x4 d = (sv[j] - k) * m;
                    s1 += d;
                    s2 += d * d;
                }
                if (rg == 0 && live) *reinterpret_cast<f32x4*>(ts_out.base + col) = k;
            } else if (live) {
                const T* __restrict__ bx = (const T*)p.bn_x;
                const int c = co0 + col;
                float mean[4], rstd[4], sc[4], sh[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    mean[u] = ts_out.stats_n[2 * (c + u)];
                    rstd[u] = ts_out.stats_n[2 * (c + u) + 1];
                    sc[u] = p.bn_scale[c + u];
                    sh[u] = p.bn_shift[c + u];
                }
                float xv[RPT][4];
#pragma unroll
                for (int j = 0; j < RPT; j++) load_vec<T, 4>(bx + (ro[j] >= 0 ? ro[j] : 0) + c, xv[j]);
#pragma unroll
                for (int j = 0; j < RPT; j++) {
                    const float m = ro[j] >= 0 ? 1.f : 0.f;
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const float xh = (xv[j][u] - mean[u]) * rstd[u];
                        const float gg = m * sv[j][u] * conv_act_grad(sc[u] * xh + sh[u], p.bn_act);
                        s1[u] += gg;
                        s2[u] += gg * xh;
                    }
                }
            }
            *reinterpret_cast<f32x4*>(part + (rg * 2 + 0) * BLK_CO + col) = s1;
            *reinterpret_cast<f32x4*>(part + (rg * 2 + 1) * BLK_CO + col) = s2;
            lds_barrier();
            if (tid < 2 * BLK_CO) {
                const int which = tid / BLK_CO, cc = tid - which * BLK_CO;
                if (co0 + cc < p.Co) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < NRG; j++) a += part[(j * 2 + which) * BLK_CO + cc];
                    if (ts_out.mode == 1) {
                        ts_out.base[(1 + which) * ts_out.C + cc] = a;
                        if (which == 0) ts_out.base[3 * ts_out.C + cc] = (float)ts_out.count;
                    } else {
                        ts_out.base[which * ts_out.C + cc] = a;
                    }
                }
            }
        }
    }
}

// Epilogue of the gather-GEMM kernels (256 threads, WC x WP waves): row-wise through the
// block's LDS (LDS_BYTES of it at `stage`) when the tensors allow it, lane-wise otherwise.
template <typename T, int TC, int TP, int WC, int WP, int LDS_BYTES>
__device__ __forceinline__ void conv_epilogue_linear(const ConvGemmParams& p, const f32x4 (&acc)[TC][TP], int co0,
                                                     long pix0, long Mz, long zoff, int HWs, void* stage) {
    constexpr int TPG = conv_stage_tpg(TC, TP, WC, WP, LDS_BYTES);
    if constexpr (TPG > 0) {
        if (p.row_store_ok) {
            conv_epilogue_lds<T, TC, TP, WC, WP, 256, TPG>(p, acc, co0, reinterpret_cast<float*>(stage), [&](int pl) -> long {
                const long m = pix0 + pl;
                if (m >= Mz) return -1;
                const long mm = m + zoff;
                const int n = (int)(mm / HWs);
                const int rem = (int)(mm - (long)n * HWs);
                const int i = rem / p.Ws, j = rem - i * p.Ws;
                return (((long)n * p.Ho + (p.oy0 + p.oys * i)) * p.Wo + (p.ox0 + p.oxs * j)) * p.Co;
            });
            return;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    conv_epilogue<T, TC, TP>(p, acc, co0, pix0, wave / WP, wave % WP, lane >> 4, lane & 15, Mz, zoff, HWs);
}

// ---------------------------------------------------------------------------
// Tile statistics (ConvGemmParams::tile_stats, mode 1) for kernels whose waves stage their output as
// [64 pixels][CO] rows of T in LDS (row stride SROW bytes, a multiple of 16) before storing whole rows: sums
// of the STORED, rounded tensor -- what a separate statistics pass over it would read.
//   bf16: for every 16-channel group, ones^T Y (column sums of the staged rows Y) and the diagonal of Y^T Y (sums
//         of squares) on the matrix cores, K = pixels; the operand is the hardware-transposed read of the rows,
//         whose K order is the same on both sides of a product.  Plain sums are exact enough for a tensor that
//         itself has eight bits of mantissa; each wave moves them to a shift before they leave.  Rows of pixels
//         outside the tensor must hold zeros.
//   fp32: shifted sums in the lanes that store the rows (WaveStatsLanes).
// The four waves' partial rows meet in LDS (`part`: 4 x 4 x CO floats) and leave as ONE row per block.
// ---------------------------------------------------------------------------
// NH: 32-pixel halves of the staged rows (2: 64 pixels, 1: 32)
template <int CO, int NH = 2>
__device__ __forceinline__ void wave_stats_mma(const char* stage, int srow, int lane, f32x4 (&acc_s)[CO / 16],
                                               f32x4 (&acc_q)[CO / 16]) {
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    const int g = lane >> 4, r16 = lane & 15, q = r16 >> 2, pp = r16 & 3;
    bf16x8 ones;
#pragma unroll
    for (int u = 0; u < 8; u++) ones[u] = (__bf16)1.0f;
#pragma unroll
    for (int h = 0; h < NH; h++)
#pragma unroll
        for (int k = 0; k < CO / 16; k++) {
            // 16 channels (16 k ..) x 32 pixels (32 h ..): lane (g, q, pp) reads 8 bytes of rows 4 g + q and + 16
            const char* a0 = stage + (32 * h + 4 * g + q) * srow + (2 * k + (pp >> 1)) * 16 + ((pp & 1) << 3);
            union {
                s16x4 hv[2];
                bf16x8 f;
            } y;
            y.hv[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
            y.hv[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 16 * srow));
            acc_s[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, y.f, acc_s[k], 0, 0, 0);
            acc_q[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y.f, y.f, acc_q[k], 0, 0, 0);
        }
}

// The wave's row of `part` from the MFMA sums.  C/D layout: lane (g, r16), register u = row 4 g + u, column r16;
// column sums sit in any row, squares on the diagonal: the lanes with g == r16 / 4 hold both for channel
// 16 k + r16.  k_first: the shift (a stored pixel of the wave), nw: pixels summed.
//   sum(y - k) = S - n k,  sum((y - k)^2) = Q - 2 k S + n k^2
template <int CO>
__device__ __forceinline__ void wave_stats_part_mma(float* part_wave, int lane, const f32x4 (&acc_s)[CO / 16],
                                                    const f32x4 (&acc_q)[CO / 16], const float (&k_first)[CO / 16], float nw) {
    const int g = lane >> 4, r16 = lane & 15;
    if (g != (r16 >> 2)) return;
#pragma unroll
    for (int k = 0; k < CO / 16; k++) {
        const float S = acc_s[k][r16 & 3], Q = acc_q[k][r16 & 3], kf = k_first[k];
        float* e = part_wave + k * 16 + r16;
        e[0] = kf;
        e[CO] = S - nw * kf;
        e[2 * CO] = fmaxf(Q - 2.f * kf * S + nw * kf * kf, 0.f);
        e[3 * CO] = nw;
    }
}

// fp32 form: a lane stores EPC channels (lc ..) of some of the wave's pixels and sums them as it goes.
template <typename T, int CO>
struct WaveStatsLanes {
    static constexpr int EPC = 16 / (int)sizeof(T), LPR = CO / EPC;
    float sk[EPC], s1[EPC], s2[EPC], cnt;
    __device__ __forceinline__ void init() {
        cnt = 0.f;
#pragma unroll
        for (int u = 0; u < EPC; u++) sk[u] = s1[u] = s2[u] = 0.f;
    }
    __device__ __forceinline__ void add(const uint4& v) {
        float f[EPC];
        unpack_vec<T, EPC>(v, f);
#pragma unroll
        for (int u = 0; u < EPC; u++) {
            const float d = f[u] - sk[u];
            s1[u] += d;
            s2[u] += d * d;
        }
        cnt += 1.f;
    }
    // lanes of one channel group (lane % LPR) -> the wave's sums -> its row of `part`
    __device__ __forceinline__ void reduce() {
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
            for (int u = 0; u < EPC; u++) {
                s1[u] += __shfl_xor(s1[u], off);
                s2[u] += __shfl_xor(s2[u], off);
            }
            cnt += __shfl_xor(cnt, off);
        }
    }
    __device__ __forceinline__ void write_part(float* part_wave, int lane) const {
        if (lane >= LPR) return;
#pragma unroll
        for (int u = 0; u < EPC; u++) {
            float* e = part_wave + lane * EPC + u;
            e[0] = sk[u];
            e[CO] = s1[u];
            e[2 * CO] = s2[u];
            e[3 * CO] = cnt;
        }
    }
};

// Waves -> the block's statistics row, moved to wave 0's shift (its first pixel must be inside the tensor):
// sum(x - k0) = s1 + n d, sum((x - k0)^2) = s2 + 2 d s1 + n d^2,  d = k_w - k0.  Threads 0 .. CO - 1; row: the
// block's row of the tile_stats tensor (planes C apart).
template <int CO>
__device__ __forceinline__ void block_stats_row(const float* part, int tid, float* row, int C) {
    if (tid >= CO || tid >= C) return;
    const float k0 = part[tid];
    float t1 = 0.f, t2 = 0.f, tc = 0.f;
#pragma unroll
    for (int w_ = 0; w_ < 4; w_++) {
        const float* e = part + w_ * 4 * CO + tid;
        const float c_ = e[3 * CO];
        if (c_ > 0.f) {
            const float d = e[0] - k0;
            t1 += e[CO] + c_ * d;
            t2 += e[2 * CO] + 2.f * d * e[CO] + c_ * d * d;
            tc += c_;
        }
    }
    row[tid] = k0;
    row[C + tid] = t1;
    row[2 * C + tid] = t2;
    row[3 * C + tid] = tc;
}

// Optional extra of a backward-data convolution: per-tile sums for the backward pass of the instance norm
// whose OUTPUT gradient the convolution writes (ConvGemmParams::tile_stats, mode 2).
struct InBwdSums {
    float* rows;      // [n][tiles][2][channels]
    int tiles;
    const void* x;    // the norm's input, shaped like the written gradient
    const float *stats, *scale, *shift;
    int act;
};

// Optional extra of a launch that absorbs the instance norm + ReLU in front of it (ConvGemmParams::clamp_t,
// clamp_bias, w_batch_stride).
struct AbsorbClamp {
    const void* t;
    const float* bias;
    long w_stride;
};

// Name (as a profiler prints it) of the conv kernel launched last by this thread.
void set_last_conv_kernel(const char* name);

// LDS-DMA variant (conv_gemm_glds.hip).  Returns true and launches when the problem fits
// it (K per tap a multiple of 128 bytes, per-pixel validity); *rc receives the status.
bool try_launch_conv_gemm_glds(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc);
const char* conv_gemm_glds_cfg_name(const ConvGemmParams& p, int dtype);

// Producer / consumer gather-GEMM (conv_gemm_pc.hip) for K per tap a multiple of 64 bytes.
// Returns true and launches when the problem fits it; name (optional) receives the kernel's
// name as a profiler prints it.
bool try_launch_conv_gemm_pc(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len);
// The same for p.nclasses > 0 sub-grids in one launch (fills in cls[].tile_end); false when it does not fit.
bool try_launch_conv_gemm_pc_multi(ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len);

// Stride-2 backward-data / transposed 3x3 convolution with the input patch resident in LDS (conv_up2.hip), for
// the multi-class form of the launch (p.nclasses parity classes); false when it does not fit.
int conv_up2_tile_stats_tiles(const ConvGemmParams& p, int dtype);
bool try_launch_conv_up2(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len);

// Stride-2 3x3 forward convolution on even image sides with the input patch resident in LDS (conv_dn2.hip);
// false when the launch is not one of its layers.  conv_dn2_tile_stats_tiles: statistics rows per sample it leaves.
int conv_dn2_tile_stats_tiles(const ConvGemmParams& p, int dtype);
bool try_launch_conv_dn2(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len);

// 3x3 stride-1 kernel with the input patch resident in LDS (conv_patch.hip).  Returns true
// and launches when the problem fits it; name (optional) receives the kernel's name as a
// profiler prints it.
bool try_launch_conv_patch(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len);
int conv_patch_tiles_per_sample(const ConvGemmParams& p, int dtype, int mode = 1);   // mode: ConvGemmParams::tile_stats_mode
int conv_patch_stats_free(const ConvGemmParams& p, int dtype);
long conv_patch_ksplit_ws_floats(const ConvGemmParams& p, int dtype);   // workspace of a split-K launch of this layer, 0: none

}  // namespace stmi
