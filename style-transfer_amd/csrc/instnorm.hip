// Instance normalisation for NHWC activations on gfx950 (HBM-bound).
//
// A thread owns V consecutive channels (16 bytes when the channel count allows) and walks
// pixels, so every wave access is a run of whole NHWC pixel rows.  Each direction is three
// launches:
//   partial  per-block sums over a slice of the plane (enough slices to fill the chip),
//            shifted by the sample's first pixel so that the sum of squares stays well
//            conditioned; written to a workspace, no atomics, bit-reproducible
//   finalize one block per sample folds the partials into per-channel constants
//   apply    streams the tensor once more and writes the result
// No atomics: the per-channel parameter gradients (sums over samples and partial rows) are folded in a fixed
// order by a small kernel of their own (in_param_grads_kernel).
#include "common.hpp"

namespace stmi {

constexpr int kInMaxSplits = 256;
constexpr int kInPpt = 4;  // pixels a thread keeps in flight

struct InGeom {
    int N, HW, C, S, chunk;  // S slices of `chunk` pixels per sample
};

__device__ __forceinline__ float act_fwd(float z, int act) {
    if (act == STMI_ACT_RELU) return fmaxf(z, 0.f);
    if (act == STMI_ACT_SIGMOID) return 1.f / (1.f + __expf(-z));
    return z;
}

__device__ __forceinline__ float act_grad(float z, int act) {
    if (act == STMI_ACT_RELU) return z > 0.f ? 1.f : 0.f;
    if (act == STMI_ACT_SIGMOID) {
        const float s = 1.f / (1.f + __expf(-z));
        return s * (1.f - s);
    }
    return 1.f;
}

// Folds a block's per-thread sums (two per channel) over its pixel slots and writes 2C
// floats to dst: [c] = first sum, [C + c] = second sum.
template <int V>
__device__ __forceinline__ void block_combine(const float (&s1)[V], const float (&s2)[V], float* red, float* dst,
                                              int C, int CG, int slots, bool active) {
    const int tid = threadIdx.x;
    const int slot = tid / CG, cg = tid - slot * CG;
    if (active) {
#pragma unroll
        for (int j = 0; j < V; j++) {
            red[slot * 2 * C + cg * V + j] = s1[j];
            red[slot * 2 * C + C + cg * V + j] = s2[j];
        }
    }
    __syncthreads();
    for (int o = tid; o < 2 * C; o += 256) {
        float acc = 0.f;
        for (int sl = 0; sl < slots; sl++) acc += red[sl * 2 * C + o];
        dst[o] = acc;
    }
}

template <typename T, int V>
__global__ __launch_bounds__(256) void in_stats_kernel(const T* __restrict__ x, float* __restrict__ ws, InGeom g) {
    extern __shared__ float red[];
    const int CG = g.C / V, slots = 256 / CG;
    const int tid = threadIdx.x;
    const bool active = tid < slots * CG;
    const int slot = tid / CG, cg = tid - slot * CG;
    const int n = blockIdx.y, s = blockIdx.x;
    const T* xs = x + (long)n * g.HW * g.C;
    float s1[V], s2[V], k[V];
#pragma unroll
    for (int j = 0; j < V; j++) s1[j] = s2[j] = 0.f;
    if (active) {
        load_vec<T, V>(xs + cg * V, k);
        const int pend = min(g.HW, (s + 1) * g.chunk);
        // kInPpt loads in flight per thread: at these sizes the pass is bound by latency, not bytes
        for (int pix0 = s * g.chunk + slot; pix0 < pend; pix0 += slots * kInPpt) {
            float v[kInPpt][V];
#pragma unroll
            for (int u = 0; u < kInPpt; u++) {
                const int pix = pix0 + u * slots;
                if (pix < pend) load_vec<T, V>(xs + (long)pix * g.C + cg * V, v[u]);
            }
#pragma unroll
            for (int u = 0; u < kInPpt; u++) {
                if (pix0 + u * slots >= pend) continue;
#pragma unroll
                for (int j = 0; j < V; j++) {
                    const float d = v[u][j] - k[j];
                    s1[j] += d;
                    s2[j] += d * d;
                }
            }
        }
    }
    block_combine<V>(s1, s2, red, ws + ((long)n * g.S + s) * 2 * g.C, g.C, CG, slots, active);
}

// Finalize kernels: grid (N, ceil(C/8)); 32 lanes cooperate on one channel's S partials.
__device__ __forceinline__ void fold_partials(const float* __restrict__ ws, const InGeom& g, int n, int c, float& s1,
                                              float& s2) {
    const int l32 = threadIdx.x & 31;
    s1 = 0.f;
    s2 = 0.f;
    if (c < g.C) {
        for (int s = l32; s < g.S; s += 32) {
            const float* w = ws + ((long)n * g.S + s) * 2 * g.C;
            s1 += w[c];
            s2 += w[g.C + c];
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
}

// stats[n][c] = (mean, 1/sqrt(var + eps)) from the shifted partial sums
template <typename T>
__global__ __launch_bounds__(256) void in_stats_finalize_kernel(const T* __restrict__ x, const float* __restrict__ ws,
                                                                float* __restrict__ stats, float eps, InGeom g) {
    const int n = blockIdx.x, c = blockIdx.y * 8 + (threadIdx.x >> 5);
    float s1, s2;
    fold_partials(ws, g, n, c, s1, s2);
    if (c < g.C && (threadIdx.x & 31) == 0) {
        const float inv = 1.f / (float)g.HW;
        const float k = to_f32<T>(x[(long)n * g.HW * g.C + c]);
        const float m = s1 * inv;
        const float var = fmaxf(s2 * inv - m * m, 0.f);
        stats[((long)n * g.C + c) * 2] = k + m;
        stats[((long)n * g.C + c) * 2 + 1] = 1.f / sqrtf(var + eps);
    }
}

// ---------------------------------------------------------------------------
// Statistics from the per-tile sums a convolution's epilogue leaves behind (conv_gemm.hpp,
// ConvGemmParams::tile_stats): per sample, plane and channel a row of `ts` tiles holding the tile's
// shift k, sum(x - k), sum((x - k)^2) and pixel count.  Tiles are combined as groups of samples with
// their own means (Chan et al.): no cancellation however large the mean is next to the deviation.
// ---------------------------------------------------------------------------
struct TileFold {
    float mean, m2, count;
};

// tiles[sample][tile][plane][channel]: element (t, plane, c) of sample n
__device__ __forceinline__ const float* tile_row(const float* tiles, int n, int t, int C, int ts) {
    return tiles + (((long)n * ts + t) * 4) * C;
}

// one thread folds all tiles of channel c of sample n (loads are coalesced over the channel threads)
__device__ __forceinline__ TileFold fold_tiles_serial(const float* __restrict__ tiles, int n, int c, int C, int ts) {
    const float k0 = tile_row(tiles, n, 0, C, ts)[c];
    float cnt = 0.f, s = 0.f;
    for (int t = 0; t < ts; t++) {
        const float* r = tile_row(tiles, n, t, C, ts);
        const float ct = r[3 * C + c];
        cnt += ct;
        s += ct * (r[c] - k0) + r[C + c];
    }
    const float mean = k0 + s / cnt;
    float m2 = 0.f;
    for (int t = 0; t < ts; t++) {
        const float* r = tile_row(tiles, n, t, C, ts);
        const float ct = r[3 * C + c], s1 = r[C + c], s2 = r[2 * C + c];
        const float rc = ct > 0.f ? __frcp_rn(ct) : 0.f;
        const float d = r[c] + s1 * rc - mean;
        m2 += s2 - s1 * s1 * rc + ct * d * d;
    }
    return TileFold{mean, m2, cnt};
}

// Statistics from long tile rows (more than the apply blocks fold themselves), one pass.  Every tile's sums are
// moved to ONE shift per channel, the first tile's (a pixel of the plane, so (mean - shift)^2 is of the order of
// the variance and the plain-sum form is as well conditioned as the statistics pass, which shifts by the
// sample's first pixel too); rows with a zero count add nothing.
// grid (N, C / CPB): a block owns CPB channels of one sample, a thread one channel quad of every TL-th tile
// (TL = 1024 / CPB tile lanes).  Sixteen-byte loads, eight tiles (32 loads) of a thread in flight at once: the
// rows are read at memory speed, not at one round trip per tile.  C must be a multiple of CPB.  (CPB 8 and 32
// take the same time on 16 x 1020 rows of 128 channels; 8 keeps small batches spread over more blocks.)
template <int CPB>
__global__ __launch_bounds__(256) void in_stats_from_tiles_kernel(const float* __restrict__ tiles,
                                                                  float* __restrict__ stats, float eps, int C, int ts) {
    constexpr int QN = CPB / 4, TL = 256 / QN;
    __shared__ f32x4 red[3][TL][QN + 1];
    const int n = blockIdx.x, q = threadIdx.x % QN, tl = threadIdx.x / QN;
    const int c = blockIdx.y * CPB + q * 4;
    const f32x4 k0 = *reinterpret_cast<const f32x4*>(tile_row(tiles, n, 0, C, ts) + c);
    f32x4 cnt = {0.f, 0.f, 0.f, 0.f}, s1 = cnt, s2 = cnt;
    constexpr int U = 8;
    for (int t0 = tl; t0 < ts; t0 += TL * U) {
        f32x4 k[U], a[U], b[U], ct[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int t = t0 + u * TL;
            const bool ok = t < ts;
            const float* r = tile_row(tiles, n, ok ? t : 0, C, ts) + c;
            k[u] = *reinterpret_cast<const f32x4*>(r);
            a[u] = *reinterpret_cast<const f32x4*>(r + C);
            b[u] = *reinterpret_cast<const f32x4*>(r + 2 * C);
            ct[u] = *reinterpret_cast<const f32x4*>(r + 3 * C);
            if (!ok) ct[u] = a[u] = b[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const f32x4 d = k[u] - k0;
            cnt += ct[u];
            s1 += a[u] + ct[u] * d;
            s2 += b[u] + 2.f * d * a[u] + ct[u] * d * d;
        }
    }
    red[0][tl][q] = cnt;
    red[1][tl][q] = s1;
    red[2][tl][q] = s2;
    __syncthreads();
    if (threadIdx.x < CPB) {
        const int qq = threadIdx.x >> 2, e = threadIdx.x & 3;
        float tc = 0.f, t1 = 0.f, t2 = 0.f;
#pragma unroll 8
        for (int j = 0; j < TL; j++) {
            tc += red[0][j][qq][e];
            t1 += red[1][j][qq][e];
            t2 += red[2][j][qq][e];
        }
        const int ch = blockIdx.y * CPB + threadIdx.x;
        const float m = t1 / tc;
        stats[((long)n * C + ch) * 2] = tile_row(tiles, n, 0, C, ts)[ch] + m;
        stats[((long)n * C + ch) * 2 + 1] = 1.f / sqrtf(fmaxf(t2 / tc - m * m, 0.f) + eps);
    }
}

// The same for any channel count: 8 channels x 32 tile lanes per block, four-byte loads.
__global__ __launch_bounds__(256) void in_stats_from_tiles_any_kernel(const float* __restrict__ tiles,
                                                                      float* __restrict__ stats, float eps, int C, int ts) {
    __shared__ float red[3][32][9];
    const int n = blockIdx.x, cl = threadIdx.x & 7, tl = threadIdx.x >> 3;
    const int c = blockIdx.y * 8 + cl;
    const bool live = c < C;
    const int cc = live ? c : 0;
    const float k0 = tile_row(tiles, n, 0, C, ts)[cc];
    float cnt = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll 8
    for (int t = tl; t < ts; t += 32) {
        const float* r = tile_row(tiles, n, t, C, ts);
        const float ct = r[3 * C + cc], d = r[cc] - k0, a = r[C + cc], b = r[2 * C + cc];
        cnt += ct;
        s1 += a + ct * d;
        s2 += b + 2.f * d * a + ct * d * d;
    }
    red[0][tl][cl] = cnt;
    red[1][tl][cl] = s1;
    red[2][tl][cl] = s2;
    __syncthreads();
    if (tl == 0 && live) {
        cnt = s1 = s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 32; j++) {
            cnt += red[0][j][cl];
            s1 += red[1][j][cl];
            s2 += red[2][j][cl];
        }
        const float m = s1 / cnt;
        stats[((long)n * C + c) * 2] = k0 + m;
        stats[((long)n * C + c) * 2 + 1] = 1.f / sqrtf(fmaxf(s2 / cnt - m * m, 0.f) + eps);
    }
}

// Sums the S partial rows of sample n (2C floats each: [c] first sum, [C + c] second sum) into out[0 .. 2C)
// in shared memory.  tmp: 256 floats of shared memory.  Every thread of the 256-thread block calls it; the
// result is visible after the call.  All loads of a thread are independent, so they are in flight together.
__device__ __forceinline__ void fold_rows_block(const float* __restrict__ rows, int S, int C, float* out, float* tmp) {
    const int W = 2 * C, tid = threadIdx.x;
    if (W >= 256) {
        for (int o = tid; o < W; o += 256) {
            float acc = 0.f;
#pragma unroll 8
            for (int s_ = 0; s_ < S; s_++) acc += rows[(long)s_ * W + o];
            out[o] = acc;
        }
        __syncthreads();
        return;
    }
    const int G = 256 / W;  // row groups
    const int rg = tid / W, o = tid - rg * W;
    float acc = 0.f;
    if (rg < G) {
#pragma unroll 8
        for (int s_ = rg; s_ < S; s_ += G) acc += rows[(long)s_ * W + o];
        tmp[rg * W + o] = acc;
    }
    __syncthreads();
    if (tid < W) {
        float t = 0.f;
        for (int j = 0; j < G; j++) t += tmp[j * W + tid];
        out[tid] = t;
    }
    __syncthreads();
}

// SRC: where the per-channel constants come from
//   0  stats[n][c] = (mean, 1/sqrt(var + eps)), made by an earlier launch
//   1  the partial rows of in_stats_kernel (ws), folded by every block for itself; block 0 of a sample
//      leaves (mean, rstd) in `stats` for the backward pass
//   2  a convolution's tile sums (short rows only), folded the same way
template <typename T, int V, int SRC>
__global__ __launch_bounds__(256) void in_apply_kernel(const T* __restrict__ x, float* __restrict__ stats,
                                                       const float* __restrict__ src, int ts, float eps,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* residual, T* y, float* y_f32, int act, InGeom g) {
    extern __shared__ float sh[];  // SRC != 0: [2C] a, b  |  [2C] sums  |  [256] scratch
    const int CG = g.C / V, slots = 256 / CG;
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    if constexpr (SRC == 1) {
        float* sums = sh + 2 * g.C;
        fold_rows_block(src + (long)n * g.S * 2 * g.C, g.S, g.C, sums, sums + 2 * g.C);
        const float inv = 1.f / (float)g.HW;
        for (int c = tid; c < g.C; c += 256) {
            const float k = to_f32<T>(x[(long)n * g.HW * g.C + c]);
            const float m = sums[c] * inv;
            const float var = fmaxf(sums[g.C + c] * inv - m * m, 0.f);
            const float mean = k + m, rstd = 1.f / sqrtf(var + eps);
            const float a = scale[c] * rstd;
            sh[c] = a;
            sh[g.C + c] = shift[c] - mean * a;
            if (blockIdx.x == 0) {
                stats[((long)n * g.C + c) * 2] = mean;
                stats[((long)n * g.C + c) * 2 + 1] = rstd;
            }
        }
        __syncthreads();
    } else if constexpr (SRC == 2) {
        // Tile sums are moved to ONE shift per channel (the first tile's), which makes them plain sums: with
        // d = k_t - k_0,  sum(x - k_0) = s1_t + n_t d,  sum((x - k_0)^2) = s2_t + 2 d s1_t + n_t d^2.
        // Thread = (tile group, channel): loads are coalesced over channels and all in flight.
        float* sums = sh + 2 * g.C;         // [2C]
        float* tmp = sums + 2 * g.C;        // [256]
        const int W = g.C < 256 ? g.C : 256, G = 256 / W;
        for (int c0 = 0; c0 < g.C; c0 += W) {
            const int rg = tid / W, c = c0 + tid % W;
            float a1 = 0.f, a2 = 0.f;
            if (rg < G && c < g.C) {
                const float k0 = tile_row(src, n, 0, g.C, ts)[c];
#pragma unroll 4
                for (int t = rg; t < ts; t += G) {
                    const float* r = tile_row(src, n, t, g.C, ts);
                    const float d = r[c] - k0, s1 = r[g.C + c], s2 = r[2 * g.C + c], ct = r[3 * g.C + c];
                    a1 += s1 + ct * d;
                    a2 += s2 + 2.f * d * s1 + ct * d * d;
                }
            }
            for (int which = 0; which < 2; which++) {
                if (rg < G) tmp[rg * W + tid % W] = which ? a2 : a1;
                __syncthreads();
                if (tid < W && c0 + tid < g.C) {
                    float t_ = 0.f;
                    for (int j = 0; j < G; j++) t_ += tmp[j * W + tid];
                    sums[which * g.C + c0 + tid] = t_;
                }
                __syncthreads();
            }
        }
        const float inv = 1.f / (float)g.HW;
        for (int c = tid; c < g.C; c += 256) {
            const float k0 = tile_row(src, n, 0, g.C, ts)[c];
            const float m = sums[c] * inv;
            const float var = fmaxf(sums[g.C + c] * inv - m * m, 0.f);
            const float mean = k0 + m, rstd = 1.f / sqrtf(var + eps);
            const float a = scale[c] * rstd;
            sh[c] = a;
            sh[g.C + c] = shift[c] - mean * a;
            if (blockIdx.x == 0) {
                stats[((long)n * g.C + c) * 2] = mean;
                stats[((long)n * g.C + c) * 2 + 1] = rstd;
            }
        }
        __syncthreads();
    }
    if (tid >= slots * CG) return;
    const int slot = tid / CG, cg = tid - slot * CG;
    float a[V], b[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int c = cg * V + j;
        if constexpr (SRC != 0) {
            a[j] = sh[c];
            b[j] = sh[g.C + c];
        } else {
            const float mean = stats[((long)n * g.C + c) * 2], rstd = stats[((long)n * g.C + c) * 2 + 1];
            a[j] = scale[c] * rstd;
            b[j] = shift[c] - mean * a[j];
        }
    }
    const long sample = (long)n * g.HW * g.C;
    const int stride = gridDim.x * slots;
    for (int pix0 = blockIdx.x * slots + slot; pix0 < g.HW; pix0 += stride * kInPpt) {
        float v[kInPpt][V], r[kInPpt][V];
#pragma unroll
        for (int u = 0; u < kInPpt; u++) {
            const int pix = pix0 + u * stride;
            if (pix >= g.HW) continue;
            const long off = sample + (long)pix * g.C + cg * V;
            load_vec<T, V>(x + off, v[u]);
            if (residual) load_vec<T, V>(residual + off, r[u]);
        }
#pragma unroll
        for (int u = 0; u < kInPpt; u++) {
            const int pix = pix0 + u * stride;
            if (pix >= g.HW) continue;
            const long off = sample + (long)pix * g.C + cg * V;
#pragma unroll
            for (int j = 0; j < V; j++) v[u][j] = act_fwd(a[j] * v[u][j] + b[j], act);
            if (residual) {
#pragma unroll
                for (int j = 0; j < V; j++) v[u][j] += r[u][j];
            }
            store_vec<T, V>(y + off, v[u]);
            if (y_f32) store_vec<float, V>(y_f32 + off, v[u]);
        }
    }
}

// The last layer's apply pass for callers that want 8-bit frames: y_u8 = (unsigned char)(act(a x + b) * 255),
// truncated like the reference's (img * 255).astype(uint8) (image_utilities.py:195-196), clamped to 0 .. 255.
// A thread owns PX consecutive pixels of C channels (3 channels: 4 pixels = 24 bytes of bf16 in, 12 bytes out).
template <typename T, int C, int PX>
__global__ __launch_bounds__(256) void in_apply_u8_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          unsigned char* __restrict__ y, int act, int HW, long groups) {
    const long gidx = blockIdx.x * 256L + threadIdx.x;
    if (gidx >= groups) return;
    const int gps = (HW + PX - 1) / PX;                 // pixel groups per sample (the last one may be short)
    const int n = (int)(gidx / gps);
    const int pix0 = (int)(gidx - (long)n * gps) * PX;
    float a[C], b[C];
#pragma unroll
    for (int c = 0; c < C; c++) {
        const float mean = stats[((long)n * C + c) * 2], rstd = stats[((long)n * C + c) * 2 + 1];
        a[c] = scale[c] * rstd;
        b[c] = shift[c] - mean * a[c];
    }
    const long base = ((long)n * HW + pix0) * C;
    const bool whole = pix0 + PX <= HW;
    float v[PX * C];
    if (whole && (((uintptr_t)(x + base)) & 7) == 0 && (PX * C * sizeof(T)) % 8 == 0) {
        constexpr int NW = PX * C * (int)sizeof(T) / 8;
        uint2 raw[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) raw[q] = reinterpret_cast<const uint2*>(x + base)[q];
        const T* e = reinterpret_cast<const T*>(raw);
#pragma unroll
        for (int i = 0; i < PX * C; i++) v[i] = to_f32<T>(e[i]);
    } else {
#pragma unroll
        for (int i = 0; i < PX * C; i++) v[i] = pix0 + i / C < HW ? to_f32<T>(x[base + i]) : 0.f;
    }
    unsigned char out[PX * C];
#pragma unroll
    for (int i = 0; i < PX * C; i++) {
        const float z = act_fwd(a[i % C] * v[i] + b[i % C], act) * 255.f;
        out[i] = (unsigned char)fminf(fmaxf(z, 0.f), 255.f);
    }
    if (whole && (((uintptr_t)(y + base)) & 3) == 0 && (PX * C) % 4 == 0) {
#pragma unroll
        for (int q = 0; q < PX * C / 4; q++) reinterpret_cast<unsigned int*>(y + base)[q] = reinterpret_cast<const unsigned int*>(out)[q];
    } else {
#pragma unroll
        for (int i = 0; i < PX * C; i++)
            if (pix0 + i / C < HW) y[base + i] = out[i];
    }
}

int in_apply_u8_impl(const void* x, const float* stats, const float* scale, const float* shift, unsigned char* y,
                     int act, int n, int h, int w, int c, int dtype, hipStream_t s) {
    STMI_REQUIRE(c == 3, "instance norm apply to 8-bit frames: built for 3-channel images (c=%d)", c);
    constexpr int PX = 4;
    const int HW = h * w;
    const long groups = (long)n * ((HW + PX - 1) / PX);
    const dim3 grid((unsigned)((groups + 255) / 256));
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL((in_apply_u8_kernel<bf16_t, 3, PX>), grid, dim3(256), 0, s, (const bf16_t*)x, stats, scale, shift, y, act, HW, groups);
    else
        hipLaunchKernelGGL((in_apply_u8_kernel<float, 3, PX>), grid, dim3(256), 0, s, (const float*)x, stats, scale, shift, y, act, HW, groups);
    STMI_LAUNCH_CHECK("in_apply_u8_kernel");
    return STMI_OK;
}

// backward pass 1: per-(n,c) sums of g and g*xhat, g = dy * act'(z)
template <typename T, int V>
__global__ __launch_bounds__(256) void in_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ ws,
                                                            int act, InGeom g) {
    extern __shared__ float red[];
    const int CG = g.C / V, slots = 256 / CG;
    const int tid = threadIdx.x;
    const bool active = tid < slots * CG;
    const int slot = tid / CG, cg = tid - slot * CG;
    const int n = blockIdx.y, s = blockIdx.x;
    float s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; j++) s1[j] = s2[j] = 0.f;
    if (active) {
        float mean[V], rstd[V], sc[V], sh[V];
#pragma unroll
        for (int j = 0; j < V; j++) {
            const int c = cg * V + j;
            mean[j] = stats[((long)n * g.C + c) * 2];
            rstd[j] = stats[((long)n * g.C + c) * 2 + 1];
            sc[j] = scale[c];
            sh[j] = shift[c];
        }
        const long sample = (long)n * g.HW * g.C;
        const int pend = min(g.HW, (s + 1) * g.chunk);
        for (int pix0 = s * g.chunk + slot; pix0 < pend; pix0 += slots * kInPpt) {
            float v[kInPpt][V], d[kInPpt][V];
#pragma unroll
            for (int u = 0; u < kInPpt; u++) {
                const int pix = pix0 + u * slots;
                if (pix >= pend) continue;
                const long off = sample + (long)pix * g.C + cg * V;
                load_vec<T, V>(x + off, v[u]);
                load_vec<T, V>(dy + off, d[u]);
            }
#pragma unroll
            for (int u = 0; u < kInPpt; u++) {
                if (pix0 + u * slots >= pend) continue;
#pragma unroll
                for (int j = 0; j < V; j++) {
                    const float xh = (v[u][j] - mean[j]) * rstd[j];
                    const float gg = d[u][j] * act_grad(sc[j] * xh + sh[j], act);
                    s1[j] += gg;
                    s2[j] += gg * xh;
                }
            }
        }
    }
    block_combine<V>(s1, s2, red, ws + ((long)n * g.S + s) * 2 * g.C, g.C, CG, slots, active);
}

// Parameter gradients of an instance norm from the partial rows of its backward reduction ([n][nrows][2C]:
// sum(g) in [0, C), sum(g * xhat) in [C, 2C)): dshift[c] (+)= sum over samples and rows of the first,
// dscale[c] (+)= of the second.  A block owns 64 of the 2C columns (`cb`: which 64); its four waves take every
// fourth row (loads coalesced over the columns, independent, so in flight together) and their four sums are
// joined in wave order: one fixed summation order whatever the launch looks like around it.
__device__ __forceinline__ void param_grads_block(const float* __restrict__ rows, int total_rows, int C, int cb,
                                                  float* dscale, float* dshift, int accumulate, float (*part)[64]) {
    const int W = 2 * C, col = cb * 64 + (threadIdx.x & 63), grp = threadIdx.x >> 6;
    float acc = 0.f;
    if (col < W) {
#pragma unroll 8
        for (int r = grp; r < total_rows; r += 4) acc += rows[(long)r * W + col];
    }
    part[grp][threadIdx.x & 63] = acc;
    __syncthreads();
    if (grp == 0 && col < W) {
        const int l = threadIdx.x;
        const float t = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
        float* dst = col < C ? dshift + col : dscale + (col - C);
        *dst = accumulate ? *dst + t : t;
    }
}

__global__ __launch_bounds__(256) void in_param_grads_kernel(const float* __restrict__ rows, int total_rows, int C,
                                                             float* dscale, float* dshift, int accumulate) {
    __shared__ float part[4][64];
    param_grads_block(rows, total_rows, C, blockIdx.x, dscale, dshift, accumulate, part);
}

// backward pass 2 (of two): folds the partial rows of pass 1 itself, then
// dx = scale*rstd*(g - mean(g) - xhat*mean(g*xhat)).  The grid's x extent is `apply_x` apply blocks plus, when
// the parameter gradients are wanted, ceil(2C / 64) more blocks (of sample 0 only) that fold ALL samples' rows
// into dscale / dshift in a fixed order while the apply blocks stream the tensor: no launch of their own, no
// atomics.
template <typename T, int V>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ ws, T* dx, int act, InGeom g,
                                                           int apply_x, float* dscale, float* dshift, int accumulate) {
    extern __shared__ float sh[];  // [2C] sums | [256] scratch
    const int CG = g.C / V, slots = 256 / CG;
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    if ((int)blockIdx.x >= apply_x) {
        if (n == 0)
            param_grads_block(ws, g.N * g.S, g.C, blockIdx.x - apply_x, dscale, dshift, accumulate,
                              reinterpret_cast<float(*)[64]>(sh));
        return;
    }
    fold_rows_block(ws + (long)n * g.S * 2 * g.C, g.S, g.C, sh, sh + 2 * g.C);
    if (tid >= slots * CG) return;
    const int slot = tid / CG, cg = tid - slot * CG;
    const float inv = 1.f / (float)g.HW;
    float mean[V], rstd[V], sc[V], shf[V], m1[V], m2[V];
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int c = cg * V + j;
        mean[j] = stats[((long)n * g.C + c) * 2];
        rstd[j] = stats[((long)n * g.C + c) * 2 + 1];
        m1[j] = sh[c] * inv;
        m2[j] = sh[g.C + c] * inv;
        sc[j] = scale[c];
        shf[j] = shift[c];
    }
    const long sample = (long)n * g.HW * g.C;
    const int stride = apply_x * slots;
    for (int pix0 = blockIdx.x * slots + slot; pix0 < g.HW; pix0 += stride * kInPpt) {
        float v[kInPpt][V], d[kInPpt][V];
#pragma unroll
        for (int u = 0; u < kInPpt; u++) {
            const int pix = pix0 + u * stride;
            if (pix >= g.HW) continue;
            const long off = sample + (long)pix * g.C + cg * V;
            load_vec<T, V>(x + off, v[u]);
            load_vec<T, V>(dy + off, d[u]);
        }
#pragma unroll
        for (int u = 0; u < kInPpt; u++) {
            const int pix = pix0 + u * stride;
            if (pix >= g.HW) continue;
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float xh = (v[u][j] - mean[j]) * rstd[j];
                const float gg = d[u][j] * act_grad(sc[j] * xh + shf[j], act);
                v[u][j] = sc[j] * rstd[j] * (gg - m1[j] - xh * m2[j]);
            }
            store_vec<T, V>(dx + sample + (long)pix * g.C + cg * V, v[u]);
        }
    }
}

// V = widest vector that divides C (C / V <= 256 threads)
template <typename T>
static int pick_v(int c) {
    constexpr int VMAX = 16 / (int)sizeof(T);
    if (c % VMAX == 0 && c / VMAX <= 256) return VMAX;
    if (c <= 256) return 1;
    return 0;
}

// Partial rows per sample.  The apply kernels fold the rows themselves (no finalize launch), so a sample has
// at most kInFoldFloats / 2C of them (32 KiB to fold per apply block), and never more than the plane has
// rounds of loads for.  The choice depends on the plane and the channel count only, never on the batch size:
// a sample's result is bit-identical whatever batch it is part of.
constexpr int kInFoldFloats = 8192;
constexpr int kInFoldMaxRows = 64;

// Planes of 8 MB and more (the 1080p activations): the pass is bound by bytes there, and the few long blocks
// that suit the fold (32 a sample at 128 channels: 8 waves a CU) leave the loads of a CU too thin to
// stream; the forward statistics pass then takes as many slices as the workspace holds and a finalize launch
// (measured at 16 x 270 x 480 x 128: 184 -> 120 us).
constexpr long kInBigPlaneBytes = 8L << 20;

template <typename T>
static InGeom make_geom(int n, int h, int w, int c, bool fwd_stats = false) {
    InGeom g;
    g.N = n; g.HW = h * w; g.C = c;
    const int v = pick_v<T>(c) > 0 ? pick_v<T>(c) : 1;
    const int slots = 256 / (c / v);
    long s = ((long)g.HW + slots * kInPpt - 1) / (slots * kInPpt);  // one round of loads per thread
    long cap = kInFoldFloats / (2L * c);
    if (cap > kInFoldMaxRows) cap = kInFoldMaxRows;
    if (fwd_stats && (long)g.HW * c * (long)sizeof(T) >= kInBigPlaneBytes) cap = kInMaxSplits;
    if (cap < 1) cap = 1;
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    g.S = (int)s;
    g.chunk = (g.HW + g.S - 1) / g.S;
    return g;
}

// ~2 pixels per thread on small planes (two blocks per CU at the training sizes overlap their folds and loads:
// 2.060 -> 2.050 ms per step against 4 pixels, 2.125 with 8; STMI_IN_PPT / STMI_IN_BMAX: tuning aids); on large
// planes few, long blocks (each folds the partial rows first)
static int apply_blocks(const InGeom& g, int slots) {
    const int ppt = STMI_ENV_INT("STMI_IN_PPT", 2);
    long b = ((long)g.HW + slots * ppt - 1) / (slots * ppt);
    if (b > STMI_ENV_INT("STMI_IN_BMAX", 128)) b = STMI_ENV_INT("STMI_IN_BMAX", 128);
    if (b < 1) b = 1;
    return (int)b;
}

#define IN_DISPATCH(T, c, ...)                                                    \
    do {                                                                          \
        const int v_ = pick_v<T>(c);                                              \
        if (v_ == 16 / (int)sizeof(T)) { constexpr int V = 16 / (int)sizeof(T); __VA_ARGS__; } \
        else if (v_ == 1) { constexpr int V = 1; __VA_ARGS__; }                   \
        else { set_error("instance norm: unsupported channel count %d", c); return STMI_ERR_INVALID; } \
    } while (0)

// workspace (floats): N*S*2C partial sums; sized for the most rows a sample can have
static inline long partial_floats(int n, int c) { return (long)n * kInMaxSplits * 2 * c; }
long in_ws_floats(int n, int h, int w, int c) { return partial_floats(n, c) + (long)n * c * 2; }

template <typename T>
static int stats_typed(const void* x, float* ws, float* stats, float eps, int n, int h, int w, int c, hipStream_t s) {
    const InGeom g = make_geom<T>(n, h, w, c, true);
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        hipLaunchKernelGGL((in_stats_kernel<T, V>), dim3(g.S, n), dim3(256), slots * 2 * c * sizeof(float), s,
                           (const T*)x, ws, g);
    });
    hipLaunchKernelGGL((in_stats_finalize_kernel<T>), dim3(n, (c + 7) / 8), dim3(256), 0, s, (const T*)x, ws, stats, eps, g);
    STMI_LAUNCH_CHECK("in_stats kernels");
    return STMI_OK;
}

constexpr int kFoldMaxTiles = 32;  // longer tile rows get a finalize launch instead of a fold in every apply block

static inline size_t fold_shared_bytes(int c) { return (size_t)(4 * c + 256) * sizeof(float); }

// src_kind: 0 statistics already in `stats`; 1 partial rows of the statistics pass in `src`; 2 tile sums in `src`
// rows_s: src_kind 1 only -- the number of partial rows per sample the statistics launch wrote
template <typename T>
static int apply_typed(const void* x, float* stats, int src_kind, const float* src, int ts, float eps,
                       const float* scale, const float* shift, const void* residual, void* y, float* y_f32, int act,
                       int n, int h, int w, int c, hipStream_t s, int rows_s = 0) {
    InGeom g = make_geom<T>(n, h, w, c);
    if (src_kind == 1) {
        STMI_REQUIRE(rows_s > 0 && rows_s <= kInFoldMaxRows, "instance norm: %d partial rows cannot be folded by the apply blocks", rows_s);
        g.S = rows_s;   // the apply blocks fold exactly the rows the statistics launch wrote
    }
    if (src_kind == 2 && ts > kFoldMaxTiles) {
        // the launch shape depends on the channel count only: a sample's statistics must not depend on its batch
        if (c % 8 == 0)
            hipLaunchKernelGGL(in_stats_from_tiles_kernel<8>, dim3(n, c / 8), dim3(256), 0, s, src, stats, eps, c, ts);
        else
            hipLaunchKernelGGL(in_stats_from_tiles_any_kernel, dim3(n, (c + 7) / 8), dim3(256), 0, s, src, stats, eps, c, ts);
        src_kind = 0;
    }
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        const dim3 grid(apply_blocks(g, slots), n);
        if (src_kind == 1)
            hipLaunchKernelGGL((in_apply_kernel<T, V, 1>), grid, dim3(256), fold_shared_bytes(c), s, (const T*)x, stats,
                               src, ts, eps, scale, shift, (const T*)residual, (T*)y, y_f32, act, g);
        else if (src_kind == 2)
            hipLaunchKernelGGL((in_apply_kernel<T, V, 2>), grid, dim3(256), fold_shared_bytes(c), s, (const T*)x, stats,
                               src, ts, eps, scale, shift, (const T*)residual, (T*)y, y_f32, act, g);
        else
            hipLaunchKernelGGL((in_apply_kernel<T, V, 0>), grid, dim3(256), 0, s, (const T*)x, stats, src, ts, eps,
                               scale, shift, (const T*)residual, (T*)y, y_f32, act, g);
    });
    STMI_LAUNCH_CHECK("in_apply_kernel");
    return STMI_OK;
}

// statistics pass + apply in two launches (the apply blocks fold the partial rows themselves)
template <typename T>
static int fwd_typed(const void* x, float* ws, float* stats, float eps, const float* scale, const float* shift,
                     const void* residual, void* y, float* y_f32, int act, int n, int h, int w, int c, hipStream_t s) {
    const InGeom g = make_geom<T>(n, h, w, c, true);
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        hipLaunchKernelGGL((in_stats_kernel<T, V>), dim3(g.S, n), dim3(256), slots * 2 * c * sizeof(float), s,
                           (const T*)x, ws, g);
    });
    STMI_LAUNCH_CHECK("in_stats_kernel");
    if (g.S > kInFoldMaxRows) {   // too many rows for the apply blocks to fold: one finalize launch
        hipLaunchKernelGGL((in_stats_finalize_kernel<T>), dim3(n, (c + 7) / 8), dim3(256), 0, s, (const T*)x, ws, stats, eps, g);
        STMI_LAUNCH_CHECK("in_stats_finalize_kernel");
        return apply_typed<T>(x, stats, 0, nullptr, 0, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, s);
    }
    return apply_typed<T>(x, stats, 1, ws, 0, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, s, g.S);
}

int in_param_grads_impl(const float* rows, int nrows, int n, int c, float* dscale, float* dshift, int accumulate,
                        hipStream_t s) {
    hipLaunchKernelGGL(in_param_grads_kernel, dim3((2 * c + 63) / 64), dim3(256), 0, s, rows, n * nrows, c, dscale,
                       dshift, accumulate);
    STMI_LAUNCH_CHECK("in_param_grads_kernel");
    return STMI_OK;
}

// partial rows per sample the backward reduction pass leaves in its workspace for this geometry
int in_bwd_rows_count(int n, int h, int w, int c, int dtype) {
    return dtype == STMI_BF16 ? make_geom<bf16_t>(n, h, w, c).S : make_geom<float>(n, h, w, c).S;
}

template <typename T>
static int bwd_typed(const void* x, const void* dy, const float* stats, const float* scale, const float* shift,
                     void* dx, float* dscale, float* dshift, int accumulate, float* ws, int act, int n, int h, int w,
                     int c, hipStream_t s) {
    const InGeom g = make_geom<T>(n, h, w, c);
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        hipLaunchKernelGGL((in_bwd_reduce_kernel<T, V>), dim3(g.S, n), dim3(256), slots * 2 * c * sizeof(float), s,
                           (const T*)x, (const T*)dy, stats, scale, shift, ws, act, g);
        const int ax = apply_blocks(g, slots), px = dscale ? (2 * c + 63) / 64 : 0;
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, V>), dim3(ax + px, n), dim3(256), fold_shared_bytes(c),
                           s, (const T*)x, (const T*)dy, stats, scale, shift, (const float*)ws, (T*)dx, act, g, ax,
                           dscale, dshift, accumulate);
    });
    STMI_LAUNCH_CHECK("in_bwd kernels");
    return STMI_OK;
}

// backward with the reduction pass already done elsewhere (a backward-data convolution's epilogue left `nrows`
// partial rows per sample, [n][nrows][2C]): only the apply pass runs
template <typename T>
static int bwd_rows_typed(const void* x, const void* dy, const float* stats, const float* scale, const float* shift,
                          void* dx, float* dscale, float* dshift, int accumulate, const float* rows, int nrows, int act,
                          int n, int h, int w, int c, hipStream_t s) {
    InGeom g = make_geom<T>(n, h, w, c);
    g.S = nrows;
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        const int ax = apply_blocks(g, slots), px = dscale ? (2 * c + 63) / 64 : 0;
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, V>), dim3(ax + px, n), dim3(256), fold_shared_bytes(c),
                           s, (const T*)x, (const T*)dy, stats, scale, shift, rows, (T*)dx, act, g, ax, dscale, dshift,
                           accumulate);
    });
    STMI_LAUNCH_CHECK("in_bwd_apply_kernel");
    return STMI_OK;
}

int in_bwd_rows_impl(const void* x, const void* dy, const float* stats, const float* scale, const float* shift, void* dx,
                     float* dscale, float* dshift, int accumulate, const float* rows, int nrows, int act, int n, int h,
                     int w, int c, int dtype, hipStream_t s) {
    return dtype == STMI_BF16
               ? bwd_rows_typed<bf16_t>(x, dy, stats, scale, shift, dx, dscale, dshift, accumulate, rows, nrows, act, n, h, w, c, s)
               : bwd_rows_typed<float>(x, dy, stats, scale, shift, dx, dscale, dshift, accumulate, rows, nrows, act, n, h, w, c, s);
}

int in_stats_impl(const void* x, float* ws, float* stats, float eps, int n, int h, int w, int c, int dtype,
                  hipStream_t s) {
    return dtype == STMI_BF16 ? stats_typed<bf16_t>(x, ws, stats, eps, n, h, w, c, s)
                              : stats_typed<float>(x, ws, stats, eps, n, h, w, c, s);
}

// ---------------------------------------------------------------------------
// Statistics pass for callers that cannot allocate: the per-block partial sums go to a
// fixed array in device memory instead of a caller workspace.  Every GPU the process uses
// holds its own copy of the array for the life of the process; its address on a device is
// looked up once, with that device current, and kept.
// ---------------------------------------------------------------------------
constexpr long kStaticWsFloats = (long)STMI_IN_STATIC_MAX_NC * 2 * kInMaxSplits;   // 16 MiB
static __device__ __align__(256) float scratch[kStaticWsFloats];

constexpr int kMaxDevices = 64;
static std::atomic<float*> g_scratch_ptr[kMaxDevices];

int in_stats_static_impl(const void* x, float* stats, float eps, int n, int h, int w, int c, int dtype,
                         hipStream_t s) {
    const long need = partial_floats(n, c);
    if (need > kStaticWsFloats) {
        set_error("instance_norm_nhwc_stats_static: n * c = %ld exceeds the fixed workspace (n * c <= %ld); "
                  "use stmi_instance_norm_nhwc_stats with a caller workspace", (long)n * c,
                  (long)STMI_IN_STATIC_MAX_NC);
        return STMI_ERR_INVALID;
    }
    int dev = -1;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    if (dev < 0 || dev >= kMaxDevices) {
        set_error("instance_norm_nhwc_stats_static: device ordinal %d out of range", dev);
        return STMI_ERR_INVALID;
    }
    float* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        // host code must never hand the array's NAME to a kernel: the name is a host-side stand-in,
        // the device address comes from the runtime (and differs per GPU)
        rc = check_hip(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)), "hipGetSymbolAddress(scratch)");
        if (rc) return rc;
        if (!ptr) {
            set_error("instance_norm_nhwc_stats_static: no device address for the fixed workspace");
            return STMI_ERR_HIP;
        }
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    // every partial row the finalize pass reads is written by the partial pass of the same call:
    // nothing in the array has to start at zero
    return in_stats_impl(x, ptr, stats, eps, n, h, w, c, dtype, s);
}

int in_apply_impl(const void* x, const float* stats, const float* scale, const float* shift, const void* residual,
                  void* y, float* y_f32, int act, int n, int h, int w, int c, int dtype, hipStream_t s) {
    float* st = const_cast<float*>(stats);  // read only on this path
    return dtype == STMI_BF16
               ? apply_typed<bf16_t>(x, st, 0, nullptr, 0, 0.f, scale, shift, residual, y, y_f32, act, n, h, w, c, s)
               : apply_typed<float>(x, st, 0, nullptr, 0, 0.f, scale, shift, residual, y, y_f32, act, n, h, w, c, s);
}

// statistics (written to `stats` for the backward pass) from a convolution's tile sums, then apply
int in_apply_tiles_impl(const void* x, const float* tiles, int ts, float* stats, float eps, const float* scale,
                        const float* shift, const void* residual, void* y, float* y_f32, int act, int n, int h, int w,
                        int c, int dtype, hipStream_t s) {
    return dtype == STMI_BF16
               ? apply_typed<bf16_t>(x, stats, 2, tiles, ts, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, s)
               : apply_typed<float>(x, stats, 2, tiles, ts, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, s);
}

// statistics [n][c][2] = (mean, rstd) from a convolution's tile sums alone (any number of rows per sample)
int in_stats_from_tiles_impl(const float* tiles, int ts, float* stats, float eps, int n, int c, hipStream_t s) {
    if (c % 8 == 0)
        hipLaunchKernelGGL(in_stats_from_tiles_kernel<8>, dim3(n, c / 8), dim3(256), 0, s, tiles, stats, eps, c, ts);
    else
        hipLaunchKernelGGL(in_stats_from_tiles_any_kernel, dim3(n, (c + 7) / 8), dim3(256), 0, s, tiles, stats, eps, c, ts);
    STMI_LAUNCH_CHECK("in_stats_from_tiles_kernel");
    return STMI_OK;
}

// What a convolution that absorbs the instance norm + ReLU in front of it needs (ConvGemmParams::clamp_t): per
// sample the clamp values t = mean - shift / a (a = scale * rstd; relu(a x + b) = a (max(x, t) - t)), a copy of the
// packed weights [taps][rows][kpad] (K = input channel, contiguous) scaled by a along K, and the constant term
// bias[n][co] = -sum over taps and channels of W' t -- from the ROUNDED weights and clamp values, the ones the kernel
// multiplies, so that a region the clamp flattens comes out as the exact zero of the normalised tensor.  scale > 0.
// Two launches: weights and clamp values, then one wave per (output channel, sample) for the constant terms.
template <typename T>
__global__ __launch_bounds__(256) void in_absorb_prep_kernel(const float* __restrict__ stats, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const T* __restrict__ wp,
                                                             T* __restrict__ clamp_t, T* __restrict__ wps, int C, int kpad,
                                                             long welems) {
    const int n = blockIdx.y;
    const float* st = stats + (long)n * C * 2;
    for (long e = blockIdx.x * 256L + threadIdx.x; e < welems; e += (long)gridDim.x * 256) {
        const int k = (int)(e % kpad);
        wps[(long)n * welems + e] = from_f32<T>(k < C ? to_f32<T>(wp[e]) * (scale[k] * st[2 * k + 1]) : 0.f);
    }
    if (blockIdx.x == 0) {
        for (int c = threadIdx.x; c < C; c += 256) {
            const float a = scale[c] * st[2 * c + 1];
            clamp_t[(long)n * C + c] = from_f32<T>(fminf(fmaxf(st[2 * c] - shift[c] / a, -1e30f), 1e30f));   // far below the fp16-NaN patterns
        }
    }
}

// groups: the taps' constant terms are summed by group (a transposed convolution's output classes use different
// taps): bias is [n][groups][co] with tap `tap` adding to group group_of[tap] (< 0: to none); groups == 1: [n][co].
struct TapGroups {
    int groups;
    signed char group_of[32];
};

template <typename T>
__global__ __launch_bounds__(64) void in_absorb_bias_kernel(const T* __restrict__ clamp_t, const T* __restrict__ wps,
                                                            float* __restrict__ bias, int C, int kpad, int rows_pad, int taps,
                                                            int co, TapGroups tg) {
    const int o = blockIdx.x, n = blockIdx.y, lane = threadIdx.x;
    const T* w_n = wps + (long)n * taps * rows_pad * kpad;
    const T* t_n = clamp_t + (long)n * C;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int tap = 0; tap < taps; tap++) {
        const T* row = w_n + ((long)tap * rows_pad + o) * kpad;
        float a = 0.f;
        for (int k = lane; k < C; k += 64) a += to_f32<T>(row[k]) * to_f32<T>(t_n[k]);
        const int gi = tg.groups == 1 ? 0 : tg.group_of[tap];
#pragma unroll
        for (int j = 0; j < 4; j++) acc[j] += gi == j ? a : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc[j] += __shfl_xor(acc[j], off);
        if (lane == 0 && j < tg.groups) bias[((long)n * tg.groups + j) * co + o] = -acc[j];
    }
}

int in_absorb_prep_impl(const float* stats, const float* scale, const float* shift, const void* wp, void* clamp_t,
                        void* wps, float* bias, int n, int c, int kpad, int rows_pad, int taps, int co, int dtype,
                        hipStream_t s, const signed char* tap_groups, int groups) {
    TapGroups tg;
    tg.groups = groups > 1 ? groups : 1;
    for (int i = 0; i < 32; i++) tg.group_of[i] = tap_groups && groups > 1 && i < taps ? tap_groups[i] : (signed char)0;
    STMI_REQUIRE(taps <= 32 && tg.groups <= 4, "absorb prep: at most 32 taps in 4 groups");
    const long welems = (long)taps * rows_pad * kpad;
    const int bx = (int)((welems + 1023) / 1024 > 256 ? 256 : (welems + 1023) / 1024 > 0 ? (welems + 1023) / 1024 : 1);
    if (dtype == STMI_BF16) {
        hipLaunchKernelGGL(in_absorb_prep_kernel<bf16_t>, dim3(bx, n), dim3(256), 0, s, stats, scale, shift, (const bf16_t*)wp,
                           (bf16_t*)clamp_t, (bf16_t*)wps, c, kpad, welems);
        hipLaunchKernelGGL(in_absorb_bias_kernel<bf16_t>, dim3(co, n), dim3(64), 0, s, (const bf16_t*)clamp_t, (const bf16_t*)wps,
                           bias, c, kpad, rows_pad, taps, co, tg);
    } else {
        hipLaunchKernelGGL(in_absorb_prep_kernel<float>, dim3(bx, n), dim3(256), 0, s, stats, scale, shift, (const float*)wp,
                           (float*)clamp_t, (float*)wps, c, kpad, welems);
        hipLaunchKernelGGL(in_absorb_bias_kernel<float>, dim3(co, n), dim3(64), 0, s, (const float*)clamp_t, (const float*)wps,
                           bias, c, kpad, rows_pad, taps, co, tg);
    }
    STMI_LAUNCH_CHECK("in_absorb_prep kernels");
    return STMI_OK;
}

// whole forward: statistics pass, then apply (which also leaves the statistics in `stats`)
int in_fwd_impl(const void* x, float* ws, float* stats, float eps, const float* scale, const float* shift,
                const void* residual, void* y, float* y_f32, int act, int n, int h, int w, int c, int dtype,
                hipStream_t s) {
    return dtype == STMI_BF16
               ? fwd_typed<bf16_t>(x, ws, stats, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, s)
               : fwd_typed<float>(x, ws, stats, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, s);
}

int in_bwd_impl(const void* x, const void* dy, const float* stats, const float* scale, const float* shift, void* dx,
                float* dscale, float* dshift, int accumulate, float* ws, int act, int n, int h, int w, int c, int dtype,
                hipStream_t s) {
    return dtype == STMI_BF16
               ? bwd_typed<bf16_t>(x, dy, stats, scale, shift, dx, dscale, dshift, accumulate, ws, act, n, h, w, c, s)
               : bwd_typed<float>(x, dy, stats, scale, shift, dx, dscale, dshift, accumulate, ws, act, n, h, w, c, s);
}

}  // namespace stmi
