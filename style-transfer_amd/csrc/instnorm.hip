// Instance normalisation for NHWC activations on gfx950 (HBM-bound).
//
// A thread owns V consecutive channels (16 bytes when the channel count allows)
// and walks pixels, so every wave access is a run of whole NHWC pixel rows.
// Statistics are two-level: per-block partial sums (shifted by the sample's
// first pixel so that sum-of-squares stays well conditioned) written to a
// workspace, combined in the prologue of the consuming kernel.  Nothing is
// atomically accumulated except the per-channel parameter gradients.
#include "common.hpp"

namespace stmi {

constexpr int kInMaxSplits = 64;

static inline int in_splits(long hw) {
    long s = (hw + 511) / 512;
    if (s > kInMaxSplits) s = kInMaxSplits;
    if (s < 1) s = 1;
    return (int)s;
}

struct InGeom {
    int N, HW, C, S, chunk;  // S splits of `chunk` pixels
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

// Combines a block's per-thread partials (2V floats each) over pixel slots and
// writes 2C floats: for channel group cg, V first-moment then V second-moment sums.
template <int V>
__device__ __forceinline__ void block_combine(const float (&s1)[V], const float (&s2)[V], float* red, float* dst,
                                              int CG, int slots, bool active) {
    const int tid = threadIdx.x;
    if (active) {
#pragma unroll
        for (int j = 0; j < V; j++) {
            red[tid * 2 * V + j] = s1[j];
            red[tid * 2 * V + V + j] = s2[j];
        }
    }
    __syncthreads();
    const int total = CG * 2 * V;
    for (int o = tid; o < total; o += 256) {
        const int cg = o / (2 * V), j = o - cg * 2 * V;
        float acc = 0.f;
        for (int sl = 0; sl < slots; sl++) acc += red[(sl * CG + cg) * 2 * V + j];
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
        for (int pix = s * g.chunk + slot; pix < pend; pix += slots) {
            float v[V];
            load_vec<T, V>(xs + (long)pix * g.C + cg * V, v);
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float d = v[j] - k[j];
                s1[j] += d;
                s2[j] += d * d;
            }
        }
    }
    block_combine<V>(s1, s2, red, ws + ((long)n * g.S + s) * 2 * g.C, CG, slots, active);
}

template <typename T, int V>
__global__ __launch_bounds__(256) void in_apply_kernel(const T* __restrict__ x, const float* __restrict__ ws,
                                                       const float* __restrict__ scale, const float* __restrict__ shift,
                                                       const T* residual, T* y, float* y_f32, float* stats_out, float eps,
                                                       int act, InGeom g) {
    const int CG = g.C / V, slots = 256 / CG;
    const int tid = threadIdx.x;
    if (tid >= slots * CG) return;
    const int slot = tid / CG, cg = tid - slot * CG;
    const int n = blockIdx.y;
    const T* xs = x + (long)n * g.HW * g.C;
    float a[V], b[V];
    {
        float k[V], s1[V], s2[V];
        load_vec<T, V>(xs + cg * V, k);
#pragma unroll
        for (int j = 0; j < V; j++) s1[j] = s2[j] = 0.f;
        for (int s = 0; s < g.S; s++) {
            const float* w = ws + ((long)n * g.S + s) * 2 * g.C + cg * 2 * V;
#pragma unroll
            for (int j = 0; j < V; j++) {
                s1[j] += w[j];
                s2[j] += w[V + j];
            }
        }
        const float inv = 1.f / (float)g.HW;
#pragma unroll
        for (int j = 0; j < V; j++) {
            const float m = s1[j] * inv;
            const float var = fmaxf(s2[j] * inv - m * m, 0.f);
            const float mean = k[j] + m;
            const float rstd = 1.f / sqrtf(var + eps);
            const int c = cg * V + j;
            a[j] = scale[c] * rstd;
            b[j] = shift[c] - mean * a[j];
            if (stats_out && blockIdx.x == 0 && slot == 0) {
                stats_out[((long)n * g.C + c) * 2] = mean;
                stats_out[((long)n * g.C + c) * 2 + 1] = rstd;
            }
        }
    }
    const long sample = (long)n * g.HW * g.C;
    for (int pix = blockIdx.x * slots + slot; pix < g.HW; pix += gridDim.x * slots) {
        const long off = sample + (long)pix * g.C + cg * V;
        float v[V];
        load_vec<T, V>(x + off, v);
#pragma unroll
        for (int j = 0; j < V; j++) v[j] = act_fwd(a[j] * v[j] + b[j], act);
        if (residual) {
            float r[V];
            load_vec<T, V>(residual + off, r);
#pragma unroll
            for (int j = 0; j < V; j++) v[j] += r[j];
        }
        store_vec<T, V>(y + off, v);
        if (y_f32) store_vec<float, V>(y_f32 + off, v);
    }
}

// pass 1 of the backward: per-(n,c) sums of g and g*xhat, g = dy * act'(z)
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
        for (int pix = s * g.chunk + slot; pix < pend; pix += slots) {
            const long off = sample + (long)pix * g.C + cg * V;
            float v[V], d[V];
            load_vec<T, V>(x + off, v);
            load_vec<T, V>(dy + off, d);
#pragma unroll
            for (int j = 0; j < V; j++) {
                const float xh = (v[j] - mean[j]) * rstd[j];
                const float gg = d[j] * act_grad(sc[j] * xh + sh[j], act);
                s1[j] += gg;
                s2[j] += gg * xh;
            }
        }
    }
    block_combine<V>(s1, s2, red, ws + ((long)n * g.S + s) * 2 * g.C, CG, slots, active);
}

// pass 2: dx = scale*rstd*(g - mean(g) - xhat*mean(g*xhat)); parameter gradients added once per sample
template <typename T, int V>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift,
                                                           const float* __restrict__ ws, T* dx, float* dscale,
                                                           float* dshift, int act, InGeom g) {
    const int CG = g.C / V, slots = 256 / CG;
    const int tid = threadIdx.x;
    if (tid >= slots * CG) return;
    const int slot = tid / CG, cg = tid - slot * CG;
    const int n = blockIdx.y;
    float mean[V], rstd[V], sc[V], sh[V], m1[V], m2[V];
#pragma unroll
    for (int j = 0; j < V; j++) m1[j] = m2[j] = 0.f;
    for (int s = 0; s < g.S; s++) {
        const float* w = ws + ((long)n * g.S + s) * 2 * g.C + cg * 2 * V;
#pragma unroll
        for (int j = 0; j < V; j++) {
            m1[j] += w[j];
            m2[j] += w[V + j];
        }
    }
    const float inv = 1.f / (float)g.HW;
#pragma unroll
    for (int j = 0; j < V; j++) {
        const int c = cg * V + j;
        mean[j] = stats[((long)n * g.C + c) * 2];
        rstd[j] = stats[((long)n * g.C + c) * 2 + 1];
        sc[j] = scale[c];
        sh[j] = shift[c];
        if (blockIdx.x == 0 && slot == 0) {
            atomicAdd(dshift + c, m1[j]);
            atomicAdd(dscale + c, m2[j]);
        }
        m1[j] *= inv;
        m2[j] *= inv;
    }
    const long sample = (long)n * g.HW * g.C;
    for (int pix = blockIdx.x * slots + slot; pix < g.HW; pix += gridDim.x * slots) {
        const long off = sample + (long)pix * g.C + cg * V;
        float v[V], d[V];
        load_vec<T, V>(x + off, v);
        load_vec<T, V>(dy + off, d);
#pragma unroll
        for (int j = 0; j < V; j++) {
            const float xh = (v[j] - mean[j]) * rstd[j];
            const float gg = d[j] * act_grad(sc[j] * xh + sh[j], act);
            v[j] = sc[j] * rstd[j] * (gg - m1[j] - xh * m2[j]);
        }
        store_vec<T, V>(dx + off, v);
    }
}

static InGeom make_geom(int n, int h, int w, int c) {
    InGeom g;
    g.N = n; g.HW = h * w; g.C = c;
    g.S = in_splits((long)h * w);
    g.chunk = (g.HW + g.S - 1) / g.S;
    return g;
}

static int apply_blocks(const InGeom& g, int slots) {
    long b = ((long)g.HW + slots * 4 - 1) / (slots * 4);  // ~4 pixels per thread
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

// V = widest vector that divides C; C up to 1024 with V >= C/256
template <typename T>
static int pick_v(int c) {
    constexpr int VMAX = 16 / (int)sizeof(T);
    if (c % VMAX == 0 && c / VMAX <= 256) return VMAX;
    if (c <= 256) return 1;
    return 0;
}

#define IN_DISPATCH(T, c, ...)                                                    \
    do {                                                                          \
        const int v_ = pick_v<T>(c);                                              \
        if (v_ == 16 / (int)sizeof(T)) { constexpr int V = 16 / (int)sizeof(T); __VA_ARGS__; } \
        else if (v_ == 1) { constexpr int V = 1; __VA_ARGS__; }                   \
        else { set_error("instance norm: unsupported channel count %d", c); return STMI_ERR_INVALID; } \
    } while (0)

template <typename T>
static int stats_typed(const void* x, float* ws, int n, int h, int w, int c, hipStream_t s) {
    const InGeom g = make_geom(n, h, w, c);
    IN_DISPATCH(T, c, {
        hipLaunchKernelGGL((in_stats_kernel<T, V>), dim3(g.S, n), dim3(256), 256 * 2 * V * sizeof(float), s,
                           (const T*)x, ws, g);
    });
    STMI_LAUNCH_CHECK("in_stats_kernel");
    return STMI_OK;
}

template <typename T>
static int apply_typed(const void* x, const float* ws, const float* scale, const float* shift, const void* residual,
                       void* y, float* y_f32, float* stats_out, float eps, int act, int n, int h, int w, int c,
                       hipStream_t s) {
    const InGeom g = make_geom(n, h, w, c);
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        hipLaunchKernelGGL((in_apply_kernel<T, V>), dim3(apply_blocks(g, slots), n), dim3(256), 0, s, (const T*)x, ws,
                           scale, shift, (const T*)residual, (T*)y, y_f32, stats_out, eps, act, g);
    });
    STMI_LAUNCH_CHECK("in_apply_kernel");
    return STMI_OK;
}

template <typename T>
static int bwd_typed(const void* x, const void* dy, const float* stats, const float* scale, const float* shift,
                     void* dx, float* dscale, float* dshift, float* ws, int act, int n, int h, int w, int c,
                     hipStream_t s) {
    const InGeom g = make_geom(n, h, w, c);
    IN_DISPATCH(T, c, {
        const int slots = 256 / (c / V);
        hipLaunchKernelGGL((in_bwd_reduce_kernel<T, V>), dim3(g.S, n), dim3(256), 256 * 2 * V * sizeof(float), s,
                           (const T*)x, (const T*)dy, stats, scale, shift, ws, act, g);
        hipLaunchKernelGGL((in_bwd_apply_kernel<T, V>), dim3(apply_blocks(g, slots), n), dim3(256), 0, s, (const T*)x,
                           (const T*)dy, stats, scale, shift, ws, (T*)dx, dscale, dshift, act, g);
    });
    STMI_LAUNCH_CHECK("in_bwd kernels");
    return STMI_OK;
}

long in_ws_floats(int n, int h, int w, int c) { return (long)n * kInMaxSplits * 2 * c; }

int in_stats_impl(const void* x, float* ws, int n, int h, int w, int c, int dtype, hipStream_t s) {
    return dtype == STMI_BF16 ? stats_typed<bf16_t>(x, ws, n, h, w, c, s) : stats_typed<float>(x, ws, n, h, w, c, s);
}

int in_apply_impl(const void* x, const float* ws, const float* scale, const float* shift, const void* residual,
                  void* y, float* y_f32, float* stats_out, float eps, int act, int n, int h, int w, int c, int dtype,
                  hipStream_t s) {
    return dtype == STMI_BF16
               ? apply_typed<bf16_t>(x, ws, scale, shift, residual, y, y_f32, stats_out, eps, act, n, h, w, c, s)
               : apply_typed<float>(x, ws, scale, shift, residual, y, y_f32, stats_out, eps, act, n, h, w, c, s);
}

int in_bwd_impl(const void* x, const void* dy, const float* stats, const float* scale, const float* shift, void* dx,
                float* dscale, float* dshift, float* ws, int act, int n, int h, int w, int c, int dtype,
                hipStream_t s) {
    return dtype == STMI_BF16
               ? bwd_typed<bf16_t>(x, dy, stats, scale, shift, dx, dscale, dshift, ws, act, n, h, w, c, s)
               : bwd_typed<float>(x, dy, stats, scale, shift, dx, dscale, dshift, ws, act, n, h, w, c, s);
}

}  // namespace stmi
