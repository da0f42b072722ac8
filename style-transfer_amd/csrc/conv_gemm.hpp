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
    // Per-tile statistics of the written tensor for a following instance norm (LDS-resident patch kernel
    // only; nullptr = off): tile_stats[sample][tile of the sample][plane][channel], planes = shift k,
    // sum(x - k), sum((x - k)^2), pixel count; tile_stats_ts = tiles per sample.  One block writes
    // four contiguous runs of its channels.
    float* tile_stats;
    int tile_stats_ts;
};

// Where one block's statistics go: element (plane, local channel c) is base[plane * C + c].
struct TileStatsOut {
    float* base;   // nullptr = no statistics
    int C, count;
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

template <typename T, int TC, int TP, int WC, int WP, int NT, int TPG, typename PixelOffset>
__device__ __forceinline__ void conv_epilogue_lds(const ConvGemmParams& p, const f32x4 (&acc)[TC][TP], int co0,
                                                  float* stage, PixelOffset pixel_offset,
                                                  TileStatsOut ts_out = TileStatsOut{nullptr, 0, 0}) {
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
    const bool want_stats = ts_out.base != nullptr;   // block-uniform
    float st_k[EPC], st_1[EPC], st_2[EPC];
#pragma unroll
    for (int u = 0; u < EPC; u++) st_k[u] = st_1[u] = st_2[u] = 0.f;

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
        __syncthreads();  // operand tiles (first group) or the previous group's rows are consumed
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
        __syncthreads();
        if (want_stats && tp0 == 0) {
            // shift of the block's sums: the tile's first staged row (in the image or not, it is a
            // finite value of the right size, and any shift gives the same statistics)
#pragma unroll
            for (int u = 0; u < EPC; u += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(stage + cl_rd + u);
                st_k[u] = t[0]; st_k[u + 1] = t[1]; st_k[u + 2] = t[2]; st_k[u + 3] = t[3];
            }
        }
        if (!col_ok) continue;
        for (int rr = tid / LPR; rr < ROWS; rr += RPP) {
            const int w_ = rr / (16 * TPG), rem = rr - w_ * (16 * TPG);
            const int pl = (w_ * TP + tp0 + (rem >> 4)) * 16 + (rem & 15);
            const long off = pixel_offset(pl);
            if (off < 0) continue;
            const long e = off + co0 + cl_rd;
            float v[EPC];
#pragma unroll
            for (int u = 0; u < EPC; u += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(stage + rr * STRIDE + cl_rd + u);
                v[u] = t[0]; v[u + 1] = t[1]; v[u + 2] = t[2]; v[u + 3] = t[3];
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
            if (want_stats) {
#pragma unroll
                for (int u = 0; u < EPC; u++) {
                    const float d = v[u] - st_k[u];
                    st_1[u] += d;
                    st_2[u] += d * d;
                }
            }
            store_vec<T, EPC>(out + e, v);
        }
    }
    if (want_stats) {
        // fold the RPP row owners of every channel: lanes of a wave first, then the waves through LDS
        static_assert(LPR <= 64 && 64 % LPR == 0, "a wave holds whole pixel rows");
#pragma unroll
        for (int o = LPR; o < 64; o <<= 1) {
#pragma unroll
            for (int u = 0; u < EPC; u++) {
                st_1[u] += __shfl_xor(st_1[u], o, 64);
                st_2[u] += __shfl_xor(st_2[u], o, 64);
            }
        }
        constexpr int NW = NT / 64;
        __syncthreads();  // the last group's rows are consumed
        if (lane < LPR) {
#pragma unroll
            for (int u = 0; u < EPC; u++) {
                stage[(wave * 2 + 0) * BLK_CO + cl_rd + u] = st_1[u];
                stage[(wave * 2 + 1) * BLK_CO + cl_rd + u] = st_2[u];
            }
        }
        __syncthreads();
        if (tid < BLK_CO && co0 + tid < p.Co) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < NW; w_++) {
                a1 += stage[(w_ * 2 + 0) * BLK_CO + tid];
                a2 += stage[(w_ * 2 + 1) * BLK_CO + tid];
            }
            float* o = ts_out.base + tid;
            o[ts_out.C] = a1;
            o[2 * ts_out.C] = a2;
            o[3 * ts_out.C] = (float)ts_out.count;
        }
        if (wave == 1 && lane < LPR && col_ok) store_vec<float, EPC>(ts_out.base + cl_rd, st_k);
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

// 3x3 stride-1 kernel with the input patch resident in LDS (conv_patch.hip).  Returns true
// and launches when the problem fits it; name (optional) receives the kernel's name as a
// profiler prints it.
bool try_launch_conv_patch(const ConvGemmParams& p, int dtype, hipStream_t s, int* rc, char* name, int name_len);
int conv_patch_tiles_per_sample(const ConvGemmParams& p, int dtype);

}  // namespace stmi
