// Bandwidth-bound kernels of the style-transfer path on gfx950: VGG input
// scaling, 2x2 SAME max-pool (forward and routed backward), fused loss
// reductions (content / Gram MSE, total variation) that emit the scaled
// gradient in the same pass, TF1-style Adam, dtype casts.
#include "common.hpp"

namespace stmi {

static inline int flat_blocks(long work_items) {
    long b = (work_items + 255) / 256;
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------- scale/shift
template <typename T>
__global__ void scale_shift_kernel(const float* __restrict__ x, const float* __restrict__ mean, T* __restrict__ y,
                                   float scale, long count, int c) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        y[i] = from_f32<T>(x[i] * scale - mean[i % c]);
}

int scale_shift_impl(const float* x, const float* mean, void* y, float scale, long count, int c, int dtype,
                     hipStream_t s) {
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL(scale_shift_kernel<bf16_t>, dim3(flat_blocks(count)), dim3(256), 0, s, x, mean, (bf16_t*)y, scale, count, c);
    else
        hipLaunchKernelGGL(scale_shift_kernel<float>, dim3(flat_blocks(count)), dim3(256), 0, s, x, mean, (float*)y, scale, count, c);
    STMI_LAUNCH_CHECK("scale_shift_kernel");
    return STMI_OK;
}

// ---------------------------------------------------------------------- casts
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ src, D* __restrict__ dst, long count) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        dst[i] = from_f32<D>(to_f32<S>(src[i]));
}

int cast_impl(const void* src, void* dst, long count, int dtype, int to_f32_flag, hipStream_t s) {
    const dim3 grid(flat_blocks(count)), block(256);
    if (dtype == STMI_BF16) {
        if (to_f32_flag) hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, s, (const bf16_t*)src, (float*)dst, count);
        else hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, s, (const float*)src, (bf16_t*)dst, count);
    } else {
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, s, (const float*)src, (float*)dst, count);
    }
    STMI_LAUNCH_CHECK("cast_kernel");
    return STMI_OK;
}

template <typename T>
__global__ void activation_kernel(const T* __restrict__ x, T* __restrict__ y, long count, int act) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float v = to_f32<T>(x[i]);
        y[i] = from_f32<T>(act == STMI_ACT_RELU ? fmaxf(v, 0.f) : act == STMI_ACT_SIGMOID ? 1.f / (1.f + __expf(-v)) : v);
    }
}

int activation_impl(const void* x, void* y, long count, int act, int dtype, hipStream_t s) {
    const dim3 grid(flat_blocks(count)), block(256);
    if (dtype == STMI_BF16) hipLaunchKernelGGL(activation_kernel<bf16_t>, grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, count, act);
    else hipLaunchKernelGGL(activation_kernel<float>, grid, block, 0, s, (const float*)x, (float*)y, count, act);
    STMI_LAUNCH_CHECK("activation_kernel");
    return STMI_OK;
}

__global__ void clip_kernel(float* x, long count, float lo, float hi) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x)
        x[i] = fminf(fmaxf(x[i], lo), hi);
}

int clip_impl(float* x, long count, float lo, float hi, hipStream_t s) {
    hipLaunchKernelGGL(clip_kernel, dim3(flat_blocks(count)), dim3(256), 0, s, x, count, lo, hi);
    STMI_LAUNCH_CHECK("clip_kernel");
    return STMI_OK;
}

// ------------------------------------------------------------------- max pool
// one thread = one output window x V channels
template <typename T, int V>
__global__ void pool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w, int c) {
    const int ho = (h + 1) / 2, wo = (w + 1) / 2, cg = c / V;
    const unsigned total = (unsigned)n * ho * wo * cg;  // < 2^31, checked by the launcher: 32-bit divisions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        unsigned r = i / cg;
        const int ox = (int)(r % wo); r /= wo;
        const int oy = (int)(r % ho);
        const int b = (int)(r / ho);
        float m[V];
#pragma unroll
        for (int j = 0; j < V; j++) m[j] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 2; dy++)
#pragma unroll
            for (int dx = 0; dx < 2; dx++) {
                const int iy = 2 * oy + dy, ix = 2 * ox + dx;
                if (iy < h && ix < w) {
                    float v[V];
                    load_vec<T, V>(x + (((long)b * h + iy) * w + ix) * c + g * V, v);
#pragma unroll
                    for (int j = 0; j < V; j++) m[j] = fmaxf(m[j], v[j]);
                }
            }
        store_vec<T, V>(y + (((long)b * ho + oy) * wo + ox) * c + g * V, m);
    }
}

template <typename T, int V>
__global__ void pool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, int n, int h,
                                int w, int c) {
    const int ho = (h + 1) / 2, wo = (w + 1) / 2, cg = c / V;
    const unsigned total = (unsigned)n * ho * wo * cg;  // < 2^31, checked by the launcher: 32-bit divisions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        unsigned r = i / cg;
        const int ox = (int)(r % wo); r /= wo;
        const int oy = (int)(r % ho);
        const int b = (int)(r / ho);
        float v[4][V], m[V], d[V];
        bool taken[V];
#pragma unroll
        for (int j = 0; j < V; j++) { m[j] = -INFINITY; taken[j] = false; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            if (iy < h && ix < w) {
                load_vec<T, V>(x + (((long)b * h + iy) * w + ix) * c + g * V, v[k]);
#pragma unroll
                for (int j = 0; j < V; j++) m[j] = fmaxf(m[j], v[k][j]);
            }
        }
        load_vec<T, V>(dy + (((long)b * ho + oy) * wo + ox) * c + g * V, d);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int iy = 2 * oy + (k >> 1), ix = 2 * ox + (k & 1);
            if (iy < h && ix < w) {
                float o[V];
#pragma unroll
                for (int j = 0; j < V; j++) {
                    // first maximum of the window takes the gradient; the ReLU
                    // that produced x passes it only where x > 0
                    const bool hit = !taken[j] && v[k][j] == m[j];
                    taken[j] = taken[j] || hit;
                    o[j] = (hit && v[k][j] > 0.f) ? d[j] : 0.f;
                }
                store_vec<T, V>(dx + (((long)b * h + iy) * w + ix) * c + g * V, o);
            }
        }
    }
}

template <typename T>
static int pool_typed(bool bwd, const void* x, const void* dy, void* out, int n, int h, int w, int c, hipStream_t s) {
    constexpr int VMAX = 16 / (int)sizeof(T);
    const int ho = (h + 1) / 2, wo = (w + 1) / 2;
    STMI_REQUIRE((long)n * ho * wo * c < (1L << 31), "max pool: more than 2^31 work items");
    if (c % VMAX == 0) {
        const dim3 grid(flat_blocks((long)n * ho * wo * (c / VMAX)));
        if (bwd) hipLaunchKernelGGL((pool_bwd_kernel<T, VMAX>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, n, h, w, c);
        else hipLaunchKernelGGL((pool_fwd_kernel<T, VMAX>), grid, dim3(256), 0, s, (const T*)x, (T*)out, n, h, w, c);
    } else {
        const dim3 grid(flat_blocks((long)n * ho * wo * c));
        if (bwd) hipLaunchKernelGGL((pool_bwd_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x, (const T*)dy, (T*)out, n, h, w, c);
        else hipLaunchKernelGGL((pool_fwd_kernel<T, 1>), grid, dim3(256), 0, s, (const T*)x, (T*)out, n, h, w, c);
    }
    STMI_LAUNCH_CHECK("pool kernel");
    return STMI_OK;
}

int pool_fwd_impl(const void* x, void* y, int n, int h, int w, int c, int dtype, hipStream_t s) {
    return dtype == STMI_BF16 ? pool_typed<bf16_t>(false, x, nullptr, y, n, h, w, c, s)
                              : pool_typed<float>(false, x, nullptr, y, n, h, w, c, s);
}

int pool_bwd_impl(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int dtype, hipStream_t s) {
    return dtype == STMI_BF16 ? pool_typed<bf16_t>(true, x, dy, dx, n, h, w, c, s)
                              : pool_typed<float>(true, x, dy, dx, n, h, w, c, s);
}

// --------------------------------------------------------------------- losses
// A loss kernel does not add its term into one scalar (float atomics: the sum would depend on the order in
// which blocks finish).  Block b writes ITS partial sum to slot[b] of a "loss slot" of STMI_LOSS_SLOT_FLOATS
// floats, block 0 clears the entries no block owns, and loss_slots_sum_kernel adds a term's slots up in a
// fixed order.  Grids are capped at the slot length.
constexpr int kSlot = STMI_LOSS_SLOT_FLOATS;

__device__ __forceinline__ void slot_write(float* slot, float block_total) {
    if (threadIdx.x == 0) slot[blockIdx.x] = block_total;
    if (blockIdx.x == 0)
        for (int i = gridDim.x + threadIdx.x; i < kSlot; i += blockDim.x) slot[i] = 0.f;
}

static inline int reduce_blocks(long work_items) {
    long b = (work_items + 2047) / 2048;
    if (b > 128) b = 128;
    if (b < 1) b = 1;
    return (int)b;
}

// slot[block] = loss_scale * sum((a-b)^2) over the block's elements; grad = grad_scale * (a-b)
template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ grad,
                                                  float* slot, float loss_scale, float grad_scale, long count) {
    __shared__ float red[4];
    float acc = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float d = to_f32<T>(a[i]) - to_f32<T>(b[i]);
        acc += d * d;
        if (grad) grad[i] = from_f32<T>(grad_scale * d);
    }
    const float tot = block_sum_256(acc, red);
    slot_write(slot, tot * loss_scale);
}

int mse_impl(const void* a, const void* b, void* grad, float* slot, float loss_scale, float grad_scale, long count,
             int dtype, hipStream_t s) {
    const dim3 grid(reduce_blocks(count));
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL(mse_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)grad, slot, loss_scale, grad_scale, count);
    else
        hipLaunchKernelGGL(mse_kernel<float>, grid, dim3(256), 0, s, (const float*)a, (const float*)b, (float*)grad, slot, loss_scale, grad_scale, count);
    STMI_LAUNCH_CHECK("mse_kernel");
    return STMI_OK;
}

// out[t] = sum of the slots first[t] .. first[t + 1] - 1, one block per term.  A slot is summed as 256 threads x
// 4 consecutive entries, wave butterflies, then the four waves in order; the slots of a term are added in index
// order: the order never depends on timing.
struct SlotTerms {
    int first[STMI_LOSS_TERMS_MAX + 1];
};

__global__ __launch_bounds__(256) void loss_slots_sum_kernel(const float* __restrict__ slots, const SlotTerms terms,
                                                             float* __restrict__ out) {
    __shared__ float red[4];
    const int t = blockIdx.x;
    float total = 0.f;
    for (int sl = terms.first[t]; sl < terms.first[t + 1]; sl++) {
        const f32x4 v = reinterpret_cast<const f32x4*>(slots + (long)sl * kSlot)[threadIdx.x];
        const float r = block_sum_256((v[0] + v[1]) + (v[2] + v[3]), red);
        total += r;   // meaningful in thread 0 only
    }
    if (threadIdx.x == 0) out[t] = total;
}

int loss_slots_sum_impl(const float* slots, const int* first, int nterms, float* out, hipStream_t s) {
    SlotTerms terms;
    for (int i = 0; i <= STMI_LOSS_TERMS_MAX; i++) terms.first[i] = first[i <= nterms ? i : nterms];
    hipLaunchKernelGGL(loss_slots_sum_kernel, dim3(nterms), dim3(256), 0, s, slots, terms, out);
    STMI_LAUNCH_CHECK("loss_slots_sum_kernel");
    return STMI_OK;
}

// Gram form: g fp32 [n,c,c] against target fp32 [c,c]; writes the scaled
// difference into the K-contiguous padded layout the Gram-backward GEMM reads
// ([n][rows_pad][kpad], zero padding).  (dG + dG^T)/2 needs the transposed element as well: a
// block takes TS x TS tiles (tile t, t + grid, ...), brings the mirrored tile's differences through LDS
// (both tiles are read along their rows), and ends with its partial sum in the loss slot.
template <typename T, int TS>
__global__ __launch_bounds__(256) void gram_mse_kernel(const float* __restrict__ g, const float* __restrict__ target,
                                                       T* __restrict__ sout, float* slot, float loss_scale,
                                                       float grad_scale, int n, int c, int rows_pad, int kpad) {
    __shared__ float red[4];
    __shared__ float mir[TS][TS + 1];
    constexpr int QR = TS / 4;               // 16-byte pieces per tile row
    constexpr int PER = TS * QR / 256;       // pieces per thread and tile (4 for 64 x 64 tiles, 1 for 32 x 32)
    const int tiles_c = kpad / TS, tiles_r = (rows_pad + TS - 1) / TS;
    const int total = n * tiles_r * tiles_c;
    float acc = 0.f;
    for (int tile = blockIdx.x; tile < total; tile += gridDim.x) {
        int t = tile;
        const int b = t / (tiles_r * tiles_c);
        t -= b * tiles_r * tiles_c;
        const int row0 = (t / tiles_c) * TS, col0 = (t % tiles_c) * TS;
        const float* gb = g + (long)b * c * c;
        // Both the tile and its mirror image are read along their rows, 16 bytes per access (c is a multiple of
        // 8, so a piece is inside the matrix or outside as a whole); all loads of a thread are issued before
        // the first use: the kernel is one memory round trip long.
        f32x4 dd[PER], dm[PER];
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = threadIdx.x + u * 256;
            const int i = e / QR, q4 = (e - i * QR) * 4;
            const int row = row0 + i, col = col0 + q4;          // the tile's own piece
            const int mrow = col0 + i, mcol = row0 + q4;        // the mirrored tile's piece
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            dd[u] = row < c && col < c ? *reinterpret_cast<const f32x4*>(gb + (long)row * c + col) -
                                             *reinterpret_cast<const f32x4*>(target + (long)row * c + col)
                                       : zero;
            dm[u] = mrow < c && mcol < c ? *reinterpret_cast<const f32x4*>(gb + (long)mrow * c + mcol) -
                                               *reinterpret_cast<const f32x4*>(target + (long)mrow * c + mcol)
                                         : zero;
        }
        if (tile != (int)blockIdx.x) __syncthreads();   // the previous tile's mirror has been read
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = threadIdx.x + u * 256;
            const int i = e / QR, q4 = (e - i * QR) * 4;
#pragma unroll
            for (int k = 0; k < 4; k++) mir[i][q4 + k] = dm[u][k];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < PER; u++) {
            const int e = threadIdx.x + u * 256;
            const int i = e / QR, q4 = (e - i * QR) * 4;
            const int row = row0 + i, col = col0 + q4;
            if (row >= rows_pad) continue;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                acc += dd[u][k] * dd[u][k];   // zero outside the matrix
                // (dG + dG^T)/2: exact also for a non-symmetric target; element (row, col + k) mirrors (col + k, row)
                v[k] = row < c && col < c ? grad_scale * 0.5f * (dd[u][k] + mir[q4 + k][i]) : 0.f;
            }
            store_vec<T, 4>(sout + ((long)b * rows_pad + row) * kpad + col, v);
        }
    }
    const float tot = block_sum_256(acc, red);
    slot_write(slot, tot * loss_scale);
}

template <typename T>
static void launch_gram_mse(const float* g, const float* target, void* sout, float* slot, float loss_scale,
                            float grad_scale, int n, int c, int rows_pad, int kpad, hipStream_t s) {
    // 64 x 64 tiles when the padded matrix is made of them, 32 x 32 for the small ones; at most one block per
    // entry of the loss slot (a block then takes several tiles)
    if (rows_pad % 64 == 0 && kpad % 64 == 0) {
        const long tiles = (long)n * (rows_pad / 64) * (kpad / 64);
        const dim3 grid((unsigned)(tiles < kSlot ? tiles : kSlot));
        hipLaunchKernelGGL((gram_mse_kernel<T, 64>), grid, dim3(256), 0, s, g, target, (T*)sout, slot, loss_scale, grad_scale, n, c, rows_pad, kpad);
    } else {
        const long tiles = (long)n * ((rows_pad + 31) / 32) * (kpad / 32);
        const dim3 grid((unsigned)(tiles < kSlot ? tiles : kSlot));
        hipLaunchKernelGGL((gram_mse_kernel<T, 32>), grid, dim3(256), 0, s, g, target, (T*)sout, slot, loss_scale, grad_scale, n, c, rows_pad, kpad);
    }
}

int gram_mse_impl(const float* g, const float* target, void* sout, float* slot, float loss_scale, float grad_scale,
                  int n, int c, int dtype, hipStream_t s) {
    const int rows_pad = packed_rows(c), kpad = round_up(c, 32);
    if (dtype == STMI_BF16) launch_gram_mse<bf16_t>(g, target, sout, slot, loss_scale, grad_scale, n, c, rows_pad, kpad, s);
    else launch_gram_mse<float>(g, target, sout, slot, loss_scale, grad_scale, n, c, rows_pad, kpad, s);
    STMI_LAUNCH_CHECK("gram_mse_kernel");
    return STMI_OK;
}

// total variation: one thread per element; forward differences to the right
// and below feed the loss, the gradient gathers the up-to-four differences the
// element takes part in.
// A block owns a contiguous span of the flat [n*h][w*c] tensor and walks it U x 256 elements at a time with the
// loads of all U elements (the value and its four neighbours, mostly cache hits) issued before the first use:
// the launch is a few memory round trips long, not one per row.  32-bit indices (the launcher checks the count).
template <typename T>
__global__ __launch_bounds__(256) void tv_kernel(const float* __restrict__ img, T* __restrict__ grad, float* slot,
                                                 float wx, float wy, int total, int span, int h, int w, int c) {
    __shared__ float red[4];
    constexpr int U = 4;
    const int rowst = w * c;
    const int begin = blockIdx.x * span, end = min(total, begin + span);
    float acc = 0.f;
    for (int e0 = begin + (int)threadIdx.x; e0 < end; e0 += 256 * U) {
        float v[U], right[U], left[U], down[U], up[U];
        bool has_r[U], has_l[U], has_d[U], has_u[U], live[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int e = e0 + u * 256;
            live[u] = e < end;
            const int ee = live[u] ? e : begin;
            const int row = ee / rowst, j = ee - row * rowst, y = row % h, x = j / c;
            has_r[u] = x + 1 < w;
            has_l[u] = x > 0;
            has_d[u] = y + 1 < h;
            has_u[u] = y > 0;
            v[u] = img[ee];
            right[u] = has_r[u] ? img[ee + c] : 0.f;
            left[u] = has_l[u] ? img[ee - c] : 0.f;
            down[u] = has_d[u] ? img[ee + rowst] : 0.f;
            up[u] = has_u[u] ? img[ee - rowst] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (!live[u]) continue;
            float gsum = 0.f;
            if (has_r[u]) { const float d = right[u] - v[u]; acc += wx * d * d; gsum -= 2.f * wx * d; }
            if (has_l[u]) { const float d = v[u] - left[u]; gsum += 2.f * wx * d; }
            if (has_d[u]) { const float d = down[u] - v[u]; acc += wy * d * d; gsum -= 2.f * wy * d; }
            if (has_u[u]) { const float d = v[u] - up[u]; gsum += 2.f * wy * d; }
            grad[e0 + u * 256] = from_f32<T>(gsum);
        }
    }
    const float tot = block_sum_256(acc, red);
    slot_write(slot, tot);
}

int tv_impl(const float* img, void* grad, float* slot, float weight, int n, int h, int w, int c, int dtype,
            hipStream_t s) {
    // reduce_mean over [n,h,w-1,c] and [n,h-1,w,c]; a 1-wide axis gives an empty mean (NaN in TF): treated as 0 here
    const float wx = w > 1 ? weight / ((float)n * h * (w - 1) * c) : 0.f;
    const float wy = h > 1 ? weight / ((float)n * (h - 1) * w * c) : 0.f;
    const long count = (long)n * h * w * c;
    if (count == 0) return STMI_OK;
    STMI_REQUIRE(count < (1L << 31) - 256 * 8, "tv: more than 2^31 elements");
    // at most 256 blocks (each ends with one entry of the loss slot), spans a multiple of the block's stride
    const int blocks = (int)(count < 256 * 1024 ? (count + 1023) / 1024 : 256);
    const int span = (int)(((count + blocks - 1) / blocks + 1023) / 1024 * 1024);
    const dim3 grid((unsigned)((count + span - 1) / span));
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL(tv_kernel<bf16_t>, grid, dim3(256), 0, s, img, (bf16_t*)grad, slot, wx, wy, (int)count, span, h, w, c);
    else
        hipLaunchKernelGGL(tv_kernel<float>, grid, dim3(256), 0, s, img, (float*)grad, slot, wx, wy, (int)count, span, h, w, c);
    STMI_LAUNCH_CHECK("tv_kernel");
    return STMI_OK;
}

__global__ void bump_step_kernel(int* step);

// ------------------------------------------------------------- step begin
// Clears several fp32 buffers and advances a step counter in ONE launch: the accumulators of a
// training step (losses, flat gradient, Gram matrices, row-patch weight gradients) would otherwise
// each cost a launch of their own.  The table travels by value in the kernel arguments, so a
// captured graph replays it without a device-side table.
struct ZeroBatch {
    float* p[STMI_ZERO_MULTI_MAX];
    long len[STMI_ZERO_MULTI_MAX];   // floats of the range
    int n;
    int* step;
};

// grid (blocks, ranges): a block clears a part of ONE range (no search per element), 16 bytes per thread and
// store where the buffer is 16-byte aligned
__global__ __launch_bounds__(256) void zero_multi_kernel(const ZeroBatch b) {
    if (b.step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) b.step[0] += 1;
    float* base = b.p[blockIdx.y];
    const long len = b.len[blockIdx.y];
    const long first = blockIdx.x * (long)blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    if ((((uintptr_t)base) & 15) == 0) {
        const long quads = len >> 2;
        for (long q = first; q < quads; q += stride) reinterpret_cast<f32x4*>(base)[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (blockIdx.x == 0 && threadIdx.x < (len & 3)) base[(quads << 2) + threadIdx.x] = 0.f;
    } else {
        for (long i = first; i < len; i += stride) base[i] = 0.f;
    }
}

int zero_multi_impl(float* const* ptrs, const long* counts, int n, int* step, hipStream_t s) {
    ZeroBatch b;
    memset(&b, 0, sizeof(b));
    long run = 0;
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (counts[i] <= 0) continue;
        run = counts[i] > run ? counts[i] : run;   // the longest range sizes the grid
        b.p[m] = ptrs[i];
        b.len[m] = counts[i];
        m++;
    }
    b.n = m;
    b.step = step;
    if (m == 0) {
        if (step) hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(64), 0, s, step);
        STMI_LAUNCH_CHECK("bump_step_kernel");
        return STMI_OK;
    }
    hipLaunchKernelGGL(zero_multi_kernel, dim3(flat_blocks((run + 3) / 4), m), dim3(256), 0, s, b);
    STMI_LAUNCH_CHECK("zero_multi_kernel");
    return STMI_OK;
}

// ----------------------------------------------------------------------- Adam
__global__ void bump_step_kernel(int* step) {
    if (threadIdx.x == 0 && blockIdx.x == 0) step[0] += 1;
}

// lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), in double (1 - b2^t must stay exact for small t, b2 = .999); one thread
// of the block works it out (two double pow are hundreds of instructions: not something for every thread).
__device__ __forceinline__ float adam_lr_t(float lr, float b1, float b2, const int* step_dev, int t_host, float* sh) {
    if (threadIdx.x == 0) {
        const int t = step_dev ? step_dev[0] : t_host;
        *sh = (float)((double)lr * sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t)));
    }
    __syncthreads();
    return *sh;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long count, float lr, float b1, float b2, float eps,
                                                   float grad_scale, const int* step_dev, int t_host) {
    __shared__ float sh;
    const float lr_t = adam_lr_t(lr, b1, b2, step_dev, t_host, &sh);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < count; i += (long)gridDim.x * blockDim.x) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// the same on 16-byte aligned buffers: four parameters per thread and access; the tail (count % 4) is the last block's
__global__ __launch_bounds__(256) void adam_vec_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long count, float lr, float b1, float b2,
                                                       float eps, float grad_scale, const int* step_dev, int t_host) {
    __shared__ float sh;
    const float lr_t = adam_lr_t(lr, b1, b2, step_dev, t_host, &sh);
    const long quads = count >> 2;
    for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < quads; q += (long)gridDim.x * blockDim.x) {
        const f32x4 gi = reinterpret_cast<const f32x4*>(g)[q] * grad_scale;
        const f32x4 mi = b1 * reinterpret_cast<const f32x4*>(m)[q] + (1.f - b1) * gi;
        const f32x4 vi = b2 * reinterpret_cast<const f32x4*>(v)[q] + (1.f - b2) * gi * gi;
        f32x4 pi = reinterpret_cast<const f32x4*>(p)[q];
#pragma unroll
        for (int u = 0; u < 4; u++) pi[u] -= lr_t * mi[u] / (sqrtf(vi[u]) + eps);
        reinterpret_cast<f32x4*>(m)[q] = mi;
        reinterpret_cast<f32x4*>(v)[q] = vi;
        reinterpret_cast<f32x4*>(p)[q] = pi;
    }
    if (blockIdx.x == gridDim.x - 1) {
        for (long i = (quads << 2) + threadIdx.x; i < count; i += blockDim.x) {
            const float gi = g[i] * grad_scale;
            const float mi = b1 * m[i] + (1.f - b1) * gi;
            const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
            m[i] = mi;
            v[i] = vi;
            p[i] -= lr_t * mi / (sqrtf(vi) + eps);
        }
    }
}

int adam_impl(float* p, const float* g, float* m, float* v, long count, float lr, float b1, float b2, float eps,
              float grad_scale, int* step_dev, int t, hipStream_t s) {
    if (step_dev && t >= 0) {
        hipLaunchKernelGGL(bump_step_kernel, dim3(1), dim3(64), 0, s, step_dev);
        STMI_LAUNCH_CHECK("bump_step_kernel");
    }
    if (((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) && count >= 4)
        hipLaunchKernelGGL(adam_vec_kernel, dim3(flat_blocks(count >> 2)), dim3(256), 0, s, p, g, m, v, count, lr, b1, b2, eps, grad_scale, (const int*)step_dev, t);
    else
        hipLaunchKernelGGL(adam_kernel, dim3(flat_blocks(count)), dim3(256), 0, s, p, g, m, v, count, lr, b1, b2, eps, grad_scale, (const int*)step_dev, t);
    STMI_LAUNCH_CHECK("adam_kernel");
    return STMI_OK;
}

}  // namespace stmi

// ---------------------------------------------------------------------------
// Row patches for layers with very few channels on one side (the 9x9 RGB layers).
// A kw x C window of a C-channel NHWC tensor is kw*C CONTIGUOUS elements; copying it out
// once per pixel into a 32-wide row turns the kw x kh convolution into kh vertical taps
// with K = 32 (or 32 outputs), which the GEMM kernels run at full MFMA efficiency.
// ---------------------------------------------------------------------------
namespace stmi {

// rows[n,y,x,e] = x[n, y, x + dx0 + e/c, e%c] for e < kw*c and an in-range column, else 0
template <typename T>
__global__ void pixels_to_rows_kernel(const T* __restrict__ x, T* __restrict__ rows, int n, int h, int w, int c,
                                      int kw, int dx0) {
    const unsigned total = (unsigned)n * h * w * 4;  // 4 chunks of 8 elements per pixel; < 2^31 (launcher)
    const int klen = kw * c;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int chunk = (int)(i & 3);
        const long pix = i >> 2;
        const int xx = (int)((i >> 2) % (unsigned)w);
        const long rowbase = (pix - xx) * c;  // element offset of column 0 of this image row
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = chunk * 8 + u;
            const int col = xx + dx0 + e / c;
            v[u] = (e < klen && (unsigned)col < (unsigned)w) ? to_f32<T>(x[rowbase + (long)(xx + dx0) * c + e]) : 0.f;
        }
        if constexpr (sizeof(T) == 2) {
            store_vec<T, 8>(rows + pix * 32 + chunk * 8, v);
        } else {
            float lo[4] = {v[0], v[1], v[2], v[3]}, hi[4] = {v[4], v[5], v[6], v[7]};
            store_vec<T, 4>(rows + pix * 32 + chunk * 8, lo);
            store_vec<T, 4>(rows + pix * 32 + chunk * 8 + 4, hi);
        }
    }
}

// y[n,yy,x,co] = sum_kx rows[n, yy, x + dx0 + kx, kx*c + co] over in-range columns (+ addend)
template <typename T>
__global__ void rows_to_pixels_sum_kernel(const T* __restrict__ rows, const T* __restrict__ addend, T* __restrict__ y,
                                          int n, int h, int w, int c, int kw, int dx0) {
    const unsigned total = (unsigned)n * h * w * c;  // < 2^31 (launcher): 32-bit divisions
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int co = (int)(i % (unsigned)c);
        const long pix = i / (unsigned)c;
        const int xx = (int)((i / (unsigned)c) % (unsigned)w);
        float acc = 0.f;
        for (int kx = 0; kx < kw; kx++) {
            const int col = xx + dx0 + kx;
            if ((unsigned)col < (unsigned)w) acc += to_f32<T>(rows[(pix + dx0 + kx) * 32 + kx * c + co]);
        }
        if (addend) acc += to_f32<T>(addend[i]);
        y[i] = from_f32<T>(acc);
    }
}

// mode 0: dw[ky][e][co]             = d[ky][e][co]            (e < kw*c; rows on the reduction side)
// mode 1: dw[ky][kw-1-j][ci][co]    = d[ky][ci][j*co_n + co]  (rows on the output side)
__global__ void rows_wgrad_to_hwio_kernel(const float* __restrict__ d, float* __restrict__ dw, int kh, int kw, int ci,
                                          int co, int mode) {
    const int total = kh * kw * ci * co;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i % co;
        const int c = (i / co) % ci;
        const int kx = (i / (co * ci)) % kw;
        const int ky = i / (co * ci * kw);
        dw[i] = mode == 0 ? d[((long)ky * 32 + kx * ci + c) * co + o]
                          : d[((long)ky * ci + c) * 32 + (kw - 1 - kx) * co + o];
    }
}

int pixels_to_rows_impl(const void* x, void* rows, int n, int h, int w, int c, int kw, int dx0, int dtype,
                        hipStream_t s) {
    STMI_REQUIRE((long)n * h * w * 4 < (1L << 31), "pixels_to_rows: more than 2^31 work items");
    const dim3 grid(flat_blocks((long)n * h * w * 4));
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL(pixels_to_rows_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)x, (bf16_t*)rows, n, h, w, c, kw, dx0);
    else
        hipLaunchKernelGGL(pixels_to_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)rows, n, h, w, c, kw, dx0);
    STMI_LAUNCH_CHECK("pixels_to_rows_kernel");
    return STMI_OK;
}

int rows_to_pixels_sum_impl(const void* rows, const void* addend, void* y, int n, int h, int w, int c, int kw, int dx0, int dtype,
                            hipStream_t s) {
    STMI_REQUIRE((long)n * h * w * c < (1L << 31), "rows_to_pixels_sum: more than 2^31 work items");
    const dim3 grid(flat_blocks((long)n * h * w * c));
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL(rows_to_pixels_sum_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)rows, (const bf16_t*)addend, (bf16_t*)y, n, h, w, c, kw, dx0);
    else
        hipLaunchKernelGGL(rows_to_pixels_sum_kernel<float>, grid, dim3(256), 0, s, (const float*)rows, (const float*)addend, (float*)y, n, h, w, c, kw, dx0);
    STMI_LAUNCH_CHECK("rows_to_pixels_sum_kernel");
    return STMI_OK;
}

int rows_wgrad_to_hwio_impl(const float* d, float* dw, int kh, int kw, int ci, int co, int mode, hipStream_t s) {
    hipLaunchKernelGGL(rows_wgrad_to_hwio_kernel, dim3(flat_blocks((long)kh * kw * ci * co)), dim3(256), 0, s, d, dw, kh, kw, ci, co, mode);
    STMI_LAUNCH_CHECK("rows_wgrad_to_hwio_kernel");
    return STMI_OK;
}

}  // namespace stmi
