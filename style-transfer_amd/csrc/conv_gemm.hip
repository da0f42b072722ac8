// Gather-GEMM convolution for gfx950.
//
// One kernel serves every data-path contraction of the stylizer and of VGG19:
//   conv2d forward, conv2d backward-data (= transposed-conv forward),
//   transposed-conv backward-data (= conv2d forward) and the Gram backward
//   (a 1x1 conv with per-sample weights).
//
// GEMM view (per block):  D[co][pixel] = sum_taps sum_k  Wp[tap][co][k] * In[pixel @ tap][k]
//   - rows  = output channels, read from a packed, K-contiguous weight copy
//   - cols  = output pixels of a (possibly strided) output sub-grid
//   - a tap = one (dy,dx) input offset with `klen` contiguous K elements.  With
//     >= 8 input channels klen is the channel count; with fewer (RGB inputs,
//     3-channel gradients) a tap is a whole kernel ROW: kw*C contiguous NHWC
//     elements, validity checked per element.
//   - stride-s backward-data is run once per output parity class so that only
//     the taps that actually hit an input pixel are multiplied.
// Accumulators are laid out co-major (row = 4*(lane>>4)+reg), so a lane owns 4
// consecutive channels of one pixel and the epilogue stores 8 or 16 bytes per
// lane into NHWC.
#include "conv_gemm.hpp"

namespace stmi {

// chunk swizzle of a 64-byte LDS row: conflict-free ds_read_b128 for the
// 16-row x 4-chunk fragment pattern (rows r and r+4 land on distinct slots
// inside every 16-lane service group).
__device__ __forceinline__ int row_swz(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

template <typename T, int TC, int TP, int WC, int WP>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const ConvGemmParams p) {
    static_assert(WC * WP == 4, "4 waves per block");
    constexpr int EPC = 16 / sizeof(T);  // elements per 16-byte chunk
    constexpr int BK = 4 * EPC;          // K elements per step (64 bytes per row)
    constexpr int BLK_CO = 16 * TC * WC;
    constexpr int BLK_PIX = 16 * TP * WP;
    constexpr int NW = (BLK_CO * 4 + 255) / 256;
    constexpr int NP = (BLK_PIX * 4 + 255) / 256;
    typedef typename Mma<T>::frag frag;

    __shared__ uint4 lds[2][(BLK_CO + BLK_PIX) * 4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wpx = wave % WP;
    const int g = lane >> 4, r16 = lane & 15;
    const int co0 = blockIdx.y * BLK_CO;
    const long pix0 = (long)blockIdx.x * BLK_PIX;
    const int z = blockIdx.z;
    const int HWs = p.Hs * p.Ws;
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)HWs : (long)p.N * HWs;
    const long zoff = batched ? (long)z * HWs : 0;
    const T* in = (const T*)p.in;
    const T* wp = (const T*)p.wp + (long)z * p.w_batch_stride;

    // per-thread gather rows
    int pn[NP], py[NP], px[NP];
#pragma unroll
    for (int q = 0; q < NP; q++) {
        const int row = (tid + 256 * q) >> 2;
        const long m = pix0 + row;
        if (row < BLK_PIX && m < Mz) {
            const long mm = m + zoff;
            const int n = (int)(mm / HWs);
            const int rem = (int)(mm - (long)n * HWs);
            const int i = rem / p.Ws;
            pn[q] = n;
            py[q] = i * p.in_stride;
            px[q] = (rem - i * p.Ws) * p.in_stride;
        } else {
            pn[q] = 0;
            py[q] = -(1 << 20);
            px[q] = 0;
        }
    }

    f32x4 acc[TC][TP];
#pragma unroll
    for (int a = 0; a < TC; a++)
#pragma unroll
        for (int b = 0; b < TP; b++) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    uint4 wreg[NW], preg[NP];

    auto gload = [&](int t, int k0) {
        const Tap tap = p.taps[t];
#pragma unroll
        for (int q = 0; q < NW; q++) {
            const int idx = tid + 256 * q;
            const int row = idx >> 2, c = idx & 3;
            if (row < BLK_CO)
                wreg[q] = *reinterpret_cast<const uint4*>(
                    wp + ((long)tap.widx * p.co_pad + co0 + row) * p.kpad + k0 + c * EPC);
        }
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int idx = tid + 256 * q;
            const int c = idx & 3;
            const int iy = py[q] + tap.dy, ix = px[q] + tap.dx;
            const int e0 = k0 + c * EPC;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (p.cdiv == 0) {
                if ((unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi && e0 < p.klen)
                    v = *reinterpret_cast<const uint4*>(
                        in + (((long)pn[q] * p.Hi + iy) * p.Wi + ix) * p.Ci + e0);
            } else if ((unsigned)iy < (unsigned)p.Hi) {
                const long base = (((long)pn[q] * p.Hi + iy) * p.Wi + ix) * p.Ci;
                T tmp[EPC];
#pragma unroll
                for (int u = 0; u < EPC; u++) {
                    const int e = e0 + u;
                    const int xx = ix + e / p.cdiv;
                    tmp[u] = (e < p.klen && (unsigned)xx < (unsigned)p.Wi) ? in[base + e] : from_f32<T>(0.f);
                }
                v = *reinterpret_cast<const uint4*>(tmp);
            }
            preg[q] = v;
        }
    };

    auto sstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NW; q++) {
            const int idx = tid + 256 * q;
            const int row = idx >> 2, c = idx & 3;
            if (row < BLK_CO) lds[buf][row * 4 + (c ^ row_swz(row))] = wreg[q];
        }
#pragma unroll
        for (int q = 0; q < NP; q++) {
            const int idx = tid + 256 * q;
            const int row = idx >> 2, c = idx & 3;
            if (row < BLK_PIX) lds[buf][(BLK_CO + row) * 4 + (c ^ row_swz(row))] = preg[q];
        }
    };

    auto compute = [&](int buf) {
        frag a[TC], b[TP];
#pragma unroll
        for (int tc = 0; tc < TC; tc++) {
            const int row = (wc * TC + tc) * 16 + r16;
            const uint4 raw = lds[buf][row * 4 + (g ^ row_swz(row))];
            a[tc] = *reinterpret_cast<const frag*>(&raw);
        }
#pragma unroll
        for (int tp = 0; tp < TP; tp++) {
            const int row = (wpx * TP + tp) * 16 + r16;
            const uint4 raw = lds[buf][(BLK_CO + row) * 4 + (g ^ row_swz(row))];
            b[tp] = *reinterpret_cast<const frag*>(&raw);
        }
#pragma unroll
        for (int tc = 0; tc < TC; tc++)
#pragma unroll
            for (int tp = 0; tp < TP; tp++) Mma<T>::mma(a[tc], b[tp], acc[tc][tp]);
    };

    const int ksteps = p.kpad / BK;
    const int nsteps = p.ntaps * ksteps;
    if (nsteps > 0) {
        int t = 0, ks = 0;
        gload(0, 0);
        sstore(0);
        __syncthreads();
        for (int s = 0; s < nsteps; s++) {
            const bool more = s + 1 < nsteps;
            if (more) {
                if (++ks == ksteps) { ks = 0; ++t; }
                gload(t, ks * BK);
            }
            compute(s & 1);
            if (more) sstore((s + 1) & 1);
            __syncthreads();
        }
    }

    conv_epilogue_linear<T, TC, TP, WC, WP, (int)sizeof(lds)>(p, acc, co0, pix0, Mz, zoff, HWs, lds);
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------
template <typename T, int TC, int TP, int WC, int WP>
static int launch_cfg(const ConvGemmParams& p, long Mz, int nz, hipStream_t s) {
    constexpr int BLK_CO = 16 * TC * WC, BLK_PIX = 16 * TP * WP;
    dim3 grid((unsigned)ceil_div(Mz, BLK_PIX), (unsigned)(p.co_pad / BLK_CO), (unsigned)nz);
    hipLaunchKernelGGL((conv_gemm_kernel<T, TC, TP, WC, WP>), grid, dim3(256), 0, s, p);
    STMI_LAUNCH_CHECK("conv_gemm_kernel");
    return STMI_OK;
}

// tile configuration for a problem: 0 = 128co x 128pix, 1 = 64x64, 2 = 64x256, 3 = 32x256, 4 = 16x256
static int pick_cfg(int rows, long Mz, int nz) {
    if (rows % 128 == 0) {
        const long blocks = (long)ceil_div(Mz, 128) * (rows / 128) * nz;
        return blocks >= 384 ? 0 : 1;
    }
    if (rows % 64 == 0) {
        const long blocks = (long)ceil_div(Mz, 256) * (rows / 64) * nz;
        return blocks >= 384 ? 2 : 1;
    }
    return rows % 32 == 0 ? 3 : 4;
}

template <typename T>
static int launch_typed(const ConvGemmParams& p, hipStream_t s) {
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)p.Hs * p.Ws : (long)p.N * p.Hs * p.Ws;
    const int nz = batched ? p.N : 1;
    if (Mz <= 0) return STMI_OK;
    switch (pick_cfg(p.co_pad, Mz, nz)) {
        case 0: return launch_cfg<T, 4, 4, 2, 2>(p, Mz, nz, s);
        case 1: return launch_cfg<T, 2, 2, 2, 2>(p, Mz, nz, s);
        case 2: return launch_cfg<T, 4, 4, 1, 4>(p, Mz, nz, s);
        case 3: return launch_cfg<T, 2, 4, 1, 4>(p, Mz, nz, s);
        default: return launch_cfg<T, 1, 4, 1, 4>(p, Mz, nz, s);
    }
}

int pool_fwd_impl(const void* x, void* y, int n, int h, int w, int c, int dtype, hipStream_t s);   // pointwise.hip

static thread_local char g_last_conv_kernel[192] = "";
static const char* kV1Cfg[5] = {"4, 4, 2, 2", "2, 2, 2, 2", "4, 4, 1, 4", "2, 4, 1, 4", "1, 4, 1, 4"};

// Kernel choice: LDS-resident patch kernel for 3x3 stride-1 layers, producer / consumer LDS-DMA
// gather-GEMM for the other layers with a multiple of 64 bytes of K per tap (the same-wave LDS-DMA
// kernel stays behind it as a tuning reference), register-staged gather-GEMM for the rest.
// The name of the instantiation launched is kept per thread for profiling front ends.
static void set_row_store(ConvGemmParams& p, int dtype) {
    const int epc = dtype == STMI_BF16 ? 8 : 4;
    const uintptr_t ptrs = (uintptr_t)p.out | (uintptr_t)p.out_pre | (uintptr_t)p.mask | (uintptr_t)p.addend;
    p.row_store_ok = p.Co % epc == 0 && (ptrs & 15) == 0;
}

int launch_conv_gemm(const ConvGemmParams& p_in, int dtype, hipStream_t s) {
    int rc = STMI_OK;
    const char* tname = dtype == STMI_BF16 ? "__bf16" : "float";
    ConvGemmParams p = p_in;
    set_row_store(p, dtype);
    if (p.tile_stats) {
        const int ts = p.in_stride == 2 ? conv_dn2_tile_stats_tiles(p, dtype) : conv_patch_tiles_per_sample(p, dtype, p.tile_stats_mode);
        STMI_REQUIRE(ts > 0 && ts == p.tile_stats_ts,
                     "conv: tile statistics asked for a layer whose kernel does not produce them (tiles %d, caller's %d)",
                     ts, p.tile_stats_ts);
        STMI_REQUIRE(!conv_patch_stats_free(p, dtype) || !(p.relu || p.mask || p.addend),
                     "conv: the persistent kernel's tile statistics are those of the plain convolution (no ReLU, mask or addend)");
    }
    if (try_launch_conv_dn2(p, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel))) return rc;
    STMI_REQUIRE(!(p.tile_stats && p.in_stride == 2),
                 "conv: the tile statistics of a stride-2 layer come from the plain convolution on 16-byte aligned tensors "
                 "only (no bias, ReLU, mask, addend or pre-activation copy)");
    if (p.pool_out) {
        // the one-tile patch kernel pools the tile it stores; any other geometry pools y in a launch of its own
        void* pool = p.pool_out;
        if (((uintptr_t)pool & 15) == 0 &&
            try_launch_conv_patch(p, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel)))
            return rc;
        p.pool_out = nullptr;
        rc = launch_conv_gemm(p, dtype, s);
        return rc ? rc : pool_fwd_impl(p.out, pool, p.N, p.Ho, p.Wo, p.Co, dtype, s);
    }
    if (try_launch_conv_patch(p, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel))) return rc;
    if (try_launch_conv_gemm_pc(p, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel))) return rc;
    if (try_launch_conv_gemm_glds(p, dtype, s, &rc)) {
        snprintf(g_last_conv_kernel, sizeof(g_last_conv_kernel),
                 "void stmi::conv_gemm_glds_kernel<%s, %s>(stmi::ConvGemmParams)", tname, conv_gemm_glds_cfg_name(p, dtype));
        return rc;
    }
    const bool batched = p.w_batch_stride != 0;
    const long Mz = batched ? (long)p.Hs * p.Ws : (long)p.N * p.Hs * p.Ws;
    snprintf(g_last_conv_kernel, sizeof(g_last_conv_kernel), "void stmi::conv_gemm_kernel<%s, %s>(stmi::ConvGemmParams)",
             tname, kV1Cfg[pick_cfg(p.co_pad, Mz, batched ? p.N : 1)]);
    if (dtype == STMI_BF16) return launch_typed<bf16_t>(p, s);
    return launch_typed<float>(p, s);
}

const char* last_conv_kernel() { return g_last_conv_kernel; }
void set_last_conv_kernel(const char* name) { snprintf(g_last_conv_kernel, sizeof(g_last_conv_kernel), "%s", name); }

// rows of a packed weight set are padded so that every block tile height divides them
int packed_rows(int rows) {
    if (rows > 64) return round_up(rows, 128);
    if (rows > 32) return 64;
    if (rows > 16) return 32;
    return 16;
}

int conv_rgb_klen(int k);  // conv_rgb.hip

static bool row_mode(int kchannels) { return (kchannels % 8) != 0; }

// mode FWD : rows = co, K = ci     mode DGRAD: rows = ci, K = co
PackGeom pack_geom(int kh, int kw, int ci, int co, int mode) {
    PackGeom g;
    if (mode == STMI_PACK_FWD_ROWOUT) {  // rows = (kx, co) pairs, one vertical tap per ky, K = ci
        g.rows = kw * co;
        g.rows_pad = packed_rows(32);  // the row buffer is 32 wide whatever kw * co is
        g.row = false;
        g.ntaps = kh;
        g.klen = ci;
        g.kpad = round_up(ci, 32);
        return g;
    }
    if (mode == STMI_PACK_FWD_RGB || mode == STMI_PACK_DGRAD_RGB) {  // conv_rgb.hip: rows of ky-major runs of (kw + 1) four-channel pixels
        g.rows = mode == STMI_PACK_FWD_RGB ? co : ci;
        g.rows_pad = round_up(g.rows, 16);
        g.row = false;
        g.ntaps = 1;
        g.klen = g.kpad = conv_rgb_klen(kh);
        return g;
    }
    const int kch = mode == STMI_PACK_FWD ? ci : co;
    g.rows = mode == STMI_PACK_FWD ? co : ci;
    g.rows_pad = packed_rows(g.rows);
    g.row = row_mode(kch);
    g.ntaps = g.row ? kh : kh * kw;
    g.klen = g.row ? kw * kch : kch;
    g.kpad = round_up(g.klen, 32);
    return g;
}

template <typename T>
__device__ __forceinline__ void pack_range(const float* __restrict__ w, T* __restrict__ out, int kh, int kw, int ci,
                                           int co, int mode, const PackGeom& g, long first, long stride) {
    // 32-bit indices (the launchers check the packed size): three 64-bit divisions per element were most of the kernel
    const unsigned total = (unsigned)g.ntaps * g.rows_pad * g.kpad, kpad = g.kpad, per_tap = kpad * g.rows_pad;
    for (unsigned idx = (unsigned)first; idx < total; idx += (unsigned)stride) {
        const int t = (int)(idx / per_tap);
        const unsigned in_tap = idx - t * per_tap;
        const int row = (int)(in_tap / kpad);
        const int e = (int)(in_tap - row * kpad);
        float v = 0.f;
        if (mode == STMI_PACK_FWD_ROWOUT) {
            if (row < g.rows && e < g.klen) v = w[(((long)t * kw + row / co) * ci + e) * co + row % co];
        } else if (mode == STMI_PACK_FWD_RGB || mode == STMI_PACK_DGRAD_RGB) {
            const int run = 4 * (kw + 1);
            const int ky = e / run, r = e - ky * run, kx = r >> 2, c = r & 3;
            if (row < g.rows && ky < kh && kx < kw && c < 3) {
                // backward-data: the flipped kernel with its channel axes exchanged (row = ci, c = co)
                v = mode == STMI_PACK_FWD_RGB ? w[(((long)ky * kw + kx) * ci + c) * co + row]
                                              : w[(((long)(kh - 1 - ky) * kw + (kw - 1 - kx)) * ci + row) * co + c];
            }
        } else if (row < g.rows && e < g.klen) {
            int ky, kx, kc;
            const int kch = mode == STMI_PACK_FWD ? ci : co;
            if (g.row) {
                ky = t;
                const int pos = e / kch;
                kc = e - pos * kch;
                // forward rows run left to right over kx; backward-data rows
                // visit the taps in ascending pixel order, i.e. descending kx
                kx = mode == STMI_PACK_FWD ? pos : kw - 1 - pos;
            } else {
                ky = t / kw;
                kx = t - ky * kw;
                kc = e;
            }
            const int c_in = mode == STMI_PACK_FWD ? kc : row;
            const int c_out = mode == STMI_PACK_FWD ? row : kc;
            v = w[(((long)ky * kw + kx) * ci + c_in) * co + c_out];
        }
        out[idx] = from_f32<T>(v);
    }
}

// Forward packing of a layer with whole 16-byte channel groups (mode FWD, not the row form): per tap the [ci][co]
// matrix goes out transposed, [co rows][ci], through a 32 x 32 LDS tile, so that both the reads (along co) and the
// writes (along K) are contiguous; element by element the reads stride by co floats and fetch every line many times
// over (53 MB from memory to pack 6.7 MB of weights, by the counters).  256 threads; unit = (tap, row tile, K tile).
template <typename T>
__device__ __forceinline__ void pack_fwd_tiles(const float* __restrict__ w, T* __restrict__ out, int ci, int co,
                                               const PackGeom& g, unsigned first_unit, unsigned unit_stride) {
    __shared__ float tile[32][33];
    const unsigned tr = (g.rows_pad + 31) / 32, te = (g.kpad + 31) / 32, per_tap = tr * te;
    const unsigned units = (unsigned)g.ntaps * per_tap;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (unsigned u = first_unit; u < units; u += unit_stride) {
        const unsigned t = u / per_tap, in_tap = u - t * per_tap;
        const int r0 = (int)(in_tap / te) * 32, e0 = (int)(in_tap % te) * 32;
        const float* wt = w + (long)t * ci * co;
        __syncthreads();   // the previous unit's tile has been written out
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int e = e0 + ty + 8 * i, r = r0 + tx;
            tile[ty + 8 * i][tx] = e < g.klen && r < g.rows ? wt[(long)e * co + r] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int r = r0 + ty + 8 * i, e = e0 + tx;
            if (r < g.rows_pad && e < g.kpad) out[((long)t * g.rows_pad + r) * g.kpad + e] = from_f32<T>(tile[tx][ty + 8 * i]);
        }
    }
}

template <typename T>
__global__ void pack_weights_kernel(const float* __restrict__ w, T* __restrict__ out, int kh, int kw, int ci, int co,
                                    int mode, PackGeom g) {
    pack_range<T>(w, out, kh, kw, ci, co, mode, g, blockIdx.x * (long)blockDim.x + threadIdx.x,
                  (long)gridDim.x * blockDim.x);
}

// Several weight tensors of one flat fp32 buffer in one launch; the table travels by value
// in the kernel arguments, so a captured graph replays it without any device-side table.
constexpr int kMaxPackBatch = 48;
struct PackEntry {
    long src_off, dst_off;
    int kh, kw, ci, co, mode;
    PackGeom g;
};
struct PackBatch {
    PackEntry e[kMaxPackBatch];
};

template <typename T>
__global__ void pack_weights_multi_kernel(const float* __restrict__ src, T* __restrict__ dst, const PackBatch b) {
    const PackEntry& d = b.e[blockIdx.y];
    if (d.mode == STMI_PACK_FWD && !d.g.row) {   // block-uniform
        pack_fwd_tiles<T>(src + d.src_off, dst + d.dst_off, d.ci, d.co, d.g, blockIdx.x, gridDim.x);
        return;
    }
    pack_range<T>(src + d.src_off, dst + d.dst_off, d.kh, d.kw, d.ci, d.co, d.mode, d.g,
                  blockIdx.x * (long)blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x);
}

int pack_weights_multi_impl(const float* src, void* dst, const stmi_pack_desc* descs, int n, int dtype, hipStream_t s) {
    for (int first = 0; first < n; first += kMaxPackBatch) {
        const int cnt = n - first < kMaxPackBatch ? n - first : kMaxPackBatch;
        PackBatch b;
        memset(&b, 0, sizeof(b));
        for (int i = 0; i < cnt; i++) {
            const stmi_pack_desc& d = descs[first + i];
            b.e[i].src_off = d.src_off; b.e[i].dst_off = d.dst_off;
            b.e[i].kh = d.kh; b.e[i].kw = d.kw; b.e[i].ci = d.ci; b.e[i].co = d.co; b.e[i].mode = d.mode;
            b.e[i].g = pack_geom(d.kh, d.kw, d.ci, d.co, d.mode);
            STMI_REQUIRE((long)b.e[i].g.ntaps * b.e[i].g.rows_pad * b.e[i].g.kpad < (1L << 31), "pack_conv_weights: packed tensor too large");
        }
        if (dtype == STMI_BF16)
            hipLaunchKernelGGL(pack_weights_multi_kernel<bf16_t>, dim3(48, cnt), dim3(256), 0, s, src, (bf16_t*)dst, b);
        else
            hipLaunchKernelGGL(pack_weights_multi_kernel<float>, dim3(48, cnt), dim3(256), 0, s, src, (float*)dst, b);
        STMI_LAUNCH_CHECK("pack_weights_multi_kernel");
    }
    return STMI_OK;
}

// Builds the tap table of a forward convolution (gathers x, writes y).
static int setup_fwd(ConvGemmParams& p, int n, int h, int w, int ci, int co, int kh, int kw, int stride) {
    int ho, wo, pt, pl;
    same_pads(h, kh, stride, &ho, &pt);
    same_pads(w, kw, stride, &wo, &pl);
    const PackGeom g = pack_geom(kh, kw, ci, co, STMI_PACK_FWD);
    STMI_REQUIRE(g.ntaps <= kMaxTaps, "conv fwd: %d taps exceed the table (%d)", g.ntaps, kMaxTaps);
    STMI_REQUIRE(!g.row || stride == 1, "conv fwd: fewer than 8 input channels needs stride 1 (ci=%d stride=%d)", ci, stride);
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = ci;
    p.Ho = ho; p.Wo = wo; p.Co = co;
    p.Hs = ho; p.Ws = wo; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
    p.in_stride = stride;
    p.ntaps = g.ntaps; p.klen = g.klen; p.kpad = g.kpad; p.co_pad = g.rows_pad;
    p.cdiv = g.row ? ci : 0;
    for (int t = 0; t < g.ntaps; t++) {
        if (g.row) {
            p.taps[t].dy = (short)(t - pt); p.taps[t].dx = (short)(-pl); p.taps[t].widx = t;
        } else {
            const int ky = t / kw, kx = t % kw;
            p.taps[t].dy = (short)(ky - pt); p.taps[t].dx = (short)(kx - pl); p.taps[t].widx = t;
        }
    }
    return STMI_OK;
}

static void clear_params(ConvGemmParams& p) {
    memset(&p, 0, sizeof(p));
    p.alpha = 1.f;
}

int conv_fwd_impl(const void* x, const void* wp, const float* bias, void* y, void* y_pre, const void* addend,
                  int n, int h, int w, int ci, int co, int kh, int kw, int stride, int relu, int dtype,
                  hipStream_t s, float* tile_stats, int tile_stats_ts, float* ws, long ws_floats, void* y_pool) {
    ConvGemmParams p;
    clear_params(p);
    int rc = setup_fwd(p, n, h, w, ci, co, kh, kw, stride);
    if (rc) return rc;
    p.in = x; p.wp = wp; p.out = y; p.out_pre = y_pre; p.bias = bias; p.addend = addend; p.relu = relu;
    p.tile_stats = tile_stats; p.tile_stats_ts = tile_stats_ts; p.tile_stats_mode = tile_stats ? 1 : 0;
    p.ksplit_ws = ws_floats > 0 ? ws : nullptr; p.ksplit_ws_floats = ws_floats;
    p.pool_out = y_pool;
    return launch_conv_gemm(p, dtype, s);
}

int in_stats_from_tiles_impl(const float* tiles, int ts, float* stats, float eps, int n, int c, hipStream_t s);   // instnorm.hip
int in_absorb_prep_impl(const float* stats, const float* scale, const float* shift, const void* wp, void* clamp_t,
                        void* wps, float* bias, int n, int c, int kpad, int rows_pad, int taps, int co, int dtype,
                        hipStream_t s, const signed char* tap_groups = nullptr, int groups = 1);

// 3x3 convolution (stride 1 or 2) of relu(IN(x)) given x, the tile sums of x (from the convolution that wrote it)
// and the norm's scale (> 0) / shift -- without the normalised tensor ever being written: see
// ConvGemmParams::clamp_t.  ws: n * ci clamp values + n packed-weight copies of the compute type + n * co floats
// of constant terms.  stats: receives (mean, rstd) of the absorbed norm.
static bool absorb_setup(ConvGemmParams& p, int n, int h, int w, int ci, int co, int stride, int dtype) {
    clear_params(p);
    if (setup_fwd(p, n, h, w, ci, co, 3, 3, stride)) return false;
    p.row_store_ok = co % (dtype == STMI_BF16 ? 8 : 4) == 0;
    // the questions the kernels' launchers ask, on a launch with stand-in pointers
    p.clamp_t = reinterpret_cast<const void*>(16);
    p.clamp_bias = reinterpret_cast<const float*>(16);
    p.tile_stats = reinterpret_cast<float*>(16);
    p.tile_stats_mode = 1;
    p.w_batch_stride = (long)p.ntaps * p.co_pad * p.kpad;
    return true;
}

// statistics rows per sample of the output, 0: the geometry has no absorbing kernel
static int absorb_tiles(ConvGemmParams& p, int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (dtype != STMI_BF16 || kh != 3 || kw != 3 || !absorb_setup(p, n, h, w, ci, co, stride, dtype)) return 0;
    if (stride == 2) {
        const int ts = conv_dn2_tile_stats_tiles(p, dtype);
        p.tile_stats_ts = ts;
        return ts > 0 && 2 * ci * 2 / 128 <= 2 ? ts : 0;
    }
    if (stride != 1 || ci != 128 || p.co_pad != 128 || !conv_patch_stats_free(p, dtype)) return 0;
    p.tile_stats_ts = conv_patch_tiles_per_sample(p, dtype, 1);
    return p.tile_stats_ts;
}

long conv_absorb_norm_ws_elems(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    ConvGemmParams p;
    if (absorb_tiles(p, n, h, w, ci, co, kh, kw, stride, dtype) <= 0) return 0;
    // clamp values, the samples' weight copies, and the fp32 bias rows (two elements of the compute type each)
    return ((long)n * ci + 7) / 8 * 8 + (long)n * p.w_batch_stride + 2 * ((long)n * co + 3) / 4 * 4;
}

int conv_absorb_norm_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    ConvGemmParams p;
    return absorb_tiles(p, n, h, w, ci, co, kh, kw, stride, dtype);
}

int conv_fwd_absorb_norm_impl(const void* x, const float* in_tiles, int in_ts, const float* in_scale, const float* in_shift,
                              float eps, float* in_stats, const void* wp, void* y, float* tile_stats, void* ws, int n, int h,
                              int w, int ci, int co, int kh, int kw, int stride, int dtype, hipStream_t s) {
    ConvGemmParams p;
    STMI_REQUIRE(absorb_tiles(p, n, h, w, ci, co, kh, kw, stride, dtype) > 0,
                 "conv absorb norm: this geometry is not supported (stmi_conv2d_absorb_norm_ws_bytes returned 0)");
    const long welems = p.w_batch_stride;
    const size_t es = dtype == STMI_BF16 ? 2 : 4;
    void* clamp_t = ws;
    char* wps = static_cast<char*>(ws) + (((long)n * ci + 7) / 8 * 8) * es;
    float* bias = reinterpret_cast<float*>(wps + (((long)n * welems * es + 15) / 16) * 16);
    int rc = in_stats_from_tiles_impl(in_tiles, in_ts, in_stats, eps, n, ci, s);
    if (rc) return rc;
    rc = in_absorb_prep_impl(in_stats, in_scale, in_shift, wp, clamp_t, wps, bias, n, ci, p.kpad, p.co_pad, p.ntaps, co, dtype, s);
    if (rc) return rc;
    p.in = x; p.wp = wps; p.out = y;
    p.clamp_t = clamp_t;
    p.clamp_bias = bias;
    p.tile_stats = tile_stats;
    set_row_store(p, dtype);
    rc = STMI_OK;
    const bool taken = stride == 2 ? try_launch_conv_dn2(p, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel))
                                   : try_launch_conv_patch(p, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel));
    STMI_REQUIRE(taken, "conv absorb norm: the tensors must be 16-byte aligned");
    return rc;
}

// The k x k layer INTO three channels (conv_to3.hip) absorbing the norm + ReLU in front of it: bf16, 9 x 9, 32 input
// channels.  ws as above, with 32 constant terms per sample (one per (kx, co) row of the packed weights).
int conv_to3_supported(int k, int ci, int dtype);
int conv_to3_clamp_impl(const void* x, const void* wq_samples, long w_stride, const void* clamp_t, const float* bias_rows,
                        void* y, int n, int h, int w, int ci, int k, int dtype, hipStream_t s);

long conv_to3_absorb_norm_ws_elems(int n, int ci, int k, int dtype) {
    if (dtype != STMI_BF16 || k != 9 || ci != 32 || !conv_to3_supported(k, ci, dtype)) return 0;
    return ((long)n * ci + 7) / 8 * 8 + (long)n * k * 32 * ci + 2 * (long)n * 32;
}

int conv_to3_absorb_norm_impl(const void* x, const float* in_tiles, int in_ts, const float* in_scale, const float* in_shift,
                              float eps, float* in_stats, const void* wq, void* y, void* ws, int n, int h, int w, int ci, int k,
                              int dtype, hipStream_t s) {
    STMI_REQUIRE(conv_to3_absorb_norm_ws_elems(n, ci, k, dtype) > 0,
                 "conv_to3 absorb norm: this geometry is not supported (stmi_conv2d_to3_absorb_norm_ws_bytes returned 0)");
    const long welems = (long)k * 32 * ci;
    const size_t es = 2;
    void* clamp_t = ws;
    char* wps = static_cast<char*>(ws) + (((long)n * ci + 7) / 8 * 8) * es;
    float* bias = reinterpret_cast<float*>(wps + (((long)n * welems * es + 15) / 16) * 16);
    int rc = in_stats_from_tiles_impl(in_tiles, in_ts, in_stats, eps, n, ci, s);
    if (rc) return rc;
    rc = in_absorb_prep_impl(in_stats, in_scale, in_shift, wq, clamp_t, wps, bias, n, ci, ci, 32, k, 32, dtype, s);
    if (rc) return rc;
    return conv_to3_clamp_impl(x, wps, welems, clamp_t, bias, y, n, h, w, ci, k, dtype, s);
}

// Floats of workspace with which a forward (dgrad == 0) or stride-1 backward-data (dgrad != 0) convolution of this
// geometry splits its K loop over several blocks per tile; 0: it never does.  Assumes 16-byte aligned tensors.
long conv_ksplit_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, int dgrad) {
    if (stride != 1 || kh != 3 || kw != 3) return 0;
    ConvGemmParams p;
    clear_params(p);
    // backward-data of a 3x3 stride-1 layer is the same gather with the roles of ci and co exchanged
    if (setup_fwd(p, n, h, w, dgrad ? co : ci, dgrad ? ci : co, kh, kw, stride)) return 0;
    p.row_store_ok = (dgrad ? ci : co) % (dtype == STMI_BF16 ? 8 : 4) == 0;
    return conv_patch_ksplit_ws_floats(p, dtype);
}

static int tile_stats_rows(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, int mode);

// Tiles per sample of the sums a stride-1 backward-data convolution of this geometry can emit (0: it cannot).
int conv_dgrad_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (stride != 1 || kh != 3 || kw != 3) return 0;
    // backward-data of a 3x3 stride-1 layer is the same gather with the roles of ci and co exchanged
    return tile_stats_rows(n, h, w, co, ci, kh, kw, 1, dtype, 2);
}

// Tiles per sample of the statistics a forward convolution of this geometry can emit (0: it cannot).
// The answer assumes 16-byte aligned tensors, which is what makes the row-wise epilogue eligible.
static int tile_stats_rows(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, int mode) {
    ConvGemmParams p;
    clear_params(p);
    if (setup_fwd(p, n, h, w, ci, co, kh, kw, stride)) return 0;
    p.row_store_ok = co % (dtype == STMI_BF16 ? 8 : 4) == 0;
    if (stride == 2 && mode == 1) return conv_dn2_tile_stats_tiles(p, dtype);
    return conv_patch_tiles_per_sample(p, dtype, mode);
}

int conv_fwd_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    return tile_stats_rows(n, h, w, ci, co, kh, kw, stride, dtype, 1);
}

// 1 when the forward statistics of this geometry cost the convolution nothing (the persistent kernel leaves them)
int conv_fwd_tile_stats_free(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    ConvGemmParams p;
    clear_params(p);
    if (setup_fwd(p, n, h, w, ci, co, kh, kw, stride)) return 0;
    p.row_store_ok = co % (dtype == STMI_BF16 ? 8 : 4) == 0;
    if (stride == 2) return conv_dn2_tile_stats_tiles(p, dtype) > 0;   // taken from the staged rows, on the matrix cores in bf16
    return conv_patch_stats_free(p, dtype);
}

// Backward-data of conv(n,h,w,ci -> co, k, stride): gathers dy, writes dx.
// The multi-class form of a strided backward-data launch: every output parity class (cy, cx) is a sub-grid with
// its own taps of the shared table (input row = (iy + pt - ky) / stride must be integral, so class (cy, cx) uses
// the taps with ky = cy + pt, kx = cx + pl (mod stride)).  p: the launch with everything but the sub-grids set.
// false: a class has no taps or the table overflows (the caller launches class by class).
static bool build_multi_class(const ConvGemmParams& p, int h, int w, int kh, int kw, int stride, int pt, int pl,
                              int dtype, ConvGemmParams& m) {
    m = p;
    m.oys = stride; m.oxs = stride;
    int nt = 0, nc = 0, most = 0;
    for (int cy = 0; cy < stride; cy++) {
        for (int cx = 0; cx < stride; cx++) {
            const int hs = (h - cy + stride - 1) / stride, ws = (w - cx + stride - 1) / stride;
            if (hs <= 0 || ws <= 0) continue;
            if (nc >= 4) return false;
            ConvGemmParams::SubGrid& g2 = m.cls[nc];
            g2.Hs = hs; g2.Ws = ws; g2.oy0 = cy; g2.ox0 = cx; g2.tap0 = nt; g2.ntaps = 0; g2.tile_end = 0;
            for (int ky = 0; ky < kh; ky++) {
                const int ny = cy + pt - ky;
                if (((ny % stride) + stride) % stride) continue;
                for (int kx = 0; kx < kw; kx++) {
                    const int nx = cx + pl - kx;
                    if (((nx % stride) + stride) % stride) continue;
                    if (nt >= kMaxTaps) return false;
                    m.taps[nt].dy = (short)((ny - (((ny % stride) + stride) % stride)) / stride);
                    m.taps[nt].dx = (short)((nx - (((nx % stride) + stride) % stride)) / stride);
                    m.taps[nt].widx = ky * kw + kx;
                    nt++;
                    g2.ntaps++;
                }
            }
            if (g2.ntaps == 0) return false;   // a class without taps is all zeros: left to the per-class launches
            if (g2.ntaps > most) { most = g2.ntaps; m.Hs = hs; m.Ws = ws; }
            nc++;
        }
    }
    if (nc == 0) return false;
    m.nclasses = nc;
    m.ntaps = most;
    set_row_store(m, dtype);
    return true;
}

// tile_stats (with tile_stats_ts rows per sample, conv_transpose_tile_stats_tiles): the forward of a transposed
// convolution also leaves the per-tile sums of dx for the instance norm behind it.
int conv_dgrad_impl(const void* dy, const void* wp, void* dx, const void* mask, const void* addend, float alpha,
                    int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, hipStream_t s,
                    const InBwdSums* bn, float* tile_stats, int tile_stats_ts, float* ws, long ws_floats,
                    const AbsorbClamp* cl) {
    int ho, wo, pt, pl;
    STMI_REQUIRE(!cl || (stride == 2 && tile_stats && !bn), "conv dgrad: a norm is absorbed by the up-sampling kernel only");
    same_pads(h, kh, stride, &ho, &pt);
    same_pads(w, kw, stride, &wo, &pl);
    const PackGeom g = pack_geom(kh, kw, ci, co, STMI_PACK_DGRAD);
    STMI_REQUIRE(!g.row || stride == 1, "conv dgrad: fewer than 8 output channels needs stride 1 (co=%d stride=%d)", co, stride);
    ConvGemmParams p;
    clear_params(p);
    p.in = dy; p.wp = wp; p.out = dx; p.mask = mask; p.addend = addend; p.alpha = alpha;
    p.N = n; p.Hi = ho; p.Wi = wo; p.Ci = co;
    p.Ho = h; p.Wo = w; p.Co = ci;
    p.klen = g.klen; p.kpad = g.kpad; p.co_pad = g.rows_pad;
    p.cdiv = g.row ? co : 0;
    p.in_stride = 1;
    if (stride == 1) {
        STMI_REQUIRE(g.ntaps <= kMaxTaps, "conv dgrad: %d taps exceed the table", g.ntaps);
        p.Hs = h; p.Ws = w; p.oy0 = 0; p.ox0 = 0; p.oys = 1; p.oxs = 1;
        p.ntaps = g.ntaps;
        for (int t = 0; t < g.ntaps; t++) {
            if (g.row) {
                p.taps[t].dy = (short)(pt - t); p.taps[t].dx = (short)(pl - (kw - 1)); p.taps[t].widx = t;
            } else {
                const int ky = t / kw, kx = t % kw;
                p.taps[t].dy = (short)(pt - ky); p.taps[t].dx = (short)(pl - kx); p.taps[t].widx = t;
            }
        }
        if (bn) {
            p.tile_stats = bn->rows; p.tile_stats_ts = bn->tiles; p.tile_stats_mode = 2;
            p.bn_x = bn->x; p.bn_stats = bn->stats; p.bn_scale = bn->scale; p.bn_shift = bn->shift; p.bn_act = bn->act;
        }
        STMI_REQUIRE(!tile_stats, "conv dgrad: forward tile statistics are for the strided (transposed-conv) form");
        p.ksplit_ws = ws_floats > 0 && !bn ? ws : nullptr; p.ksplit_ws_floats = ws_floats;
        return launch_conv_gemm(p, dtype, s);
    }
    STMI_REQUIRE(!bn, "conv dgrad: instance-norm sums are produced by stride-1 layers only");
    // First try all classes in ONE launch (each is only 1..4 taps deep: four launches are mostly launch overhead).
    if (stride * stride <= 4) {
        ConvGemmParams m;
        if (build_multi_class(p, h, w, kh, kw, stride, pt, pl, dtype, m)) {
            int rc = STMI_OK;
            if (tile_stats) {
                m.tile_stats = tile_stats; m.tile_stats_ts = tile_stats_ts; m.tile_stats_mode = 1;
            }
            if (cl) {
                m.clamp_t = cl->t; m.clamp_bias = cl->bias; m.w_batch_stride = cl->w_stride;
            }
            if (try_launch_conv_up2(m, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel))) return rc;
            STMI_REQUIRE(!cl, "transposed conv: this geometry cannot absorb a norm");
            STMI_REQUIRE(!tile_stats, "transposed conv: this geometry leaves no tile statistics");
            if (try_launch_conv_gemm_pc_multi(m, dtype, s, &rc, g_last_conv_kernel, (int)sizeof(g_last_conv_kernel))) return rc;
        }
    }
    STMI_REQUIRE(!tile_stats, "transposed conv: this geometry leaves no tile statistics");
    // one launch per class
    for (int cy = 0; cy < stride; cy++) {
        for (int cx = 0; cx < stride; cx++) {
            p.oy0 = cy; p.ox0 = cx; p.oys = stride; p.oxs = stride;
            p.Hs = (h - cy + stride - 1) / stride;
            p.Ws = (w - cx + stride - 1) / stride;
            if (p.Hs <= 0 || p.Ws <= 0) continue;
            int nt = 0;
            for (int ky = 0; ky < kh; ky++) {
                const int ny = cy + pt - ky;
                if (((ny % stride) + stride) % stride) continue;
                for (int kx = 0; kx < kw; kx++) {
                    const int nx = cx + pl - kx;
                    if (((nx % stride) + stride) % stride) continue;
                    STMI_REQUIRE(nt < kMaxTaps, "conv dgrad: tap table overflow");
                    // exact division, also for negative numerators
                    p.taps[nt].dy = (short)((ny - (((ny % stride) + stride) % stride)) / stride);
                    p.taps[nt].dx = (short)((nx - (((nx % stride) + stride) % stride)) / stride);
                    p.taps[nt].widx = ky * kw + kx;
                    nt++;
                }
            }
            p.ntaps = nt;
            int rc = launch_conv_gemm(p, dtype, s);
            if (rc) return rc;
        }
    }
    // the profiling name says that this call was several launches
    const size_t used = strlen(g_last_conv_kernel);
    snprintf(g_last_conv_kernel + used, sizeof(g_last_conv_kernel) - used, " [one launch per parity class]");
    return STMI_OK;
}

// Statistics rows per sample the forward of a transposed convolution x[n,h,w,ci] -> y[n,s*h,s*w,co] can leave
// (conv_dgrad_impl(..., tile_stats)), 0 when the kernel that runs this geometry leaves none.
int conv_transpose_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (stride < 2 || stride * stride > 4) return 0;
    // as conv_dgrad_impl sees it: the backward-data of a conv (n, s*h, s*w, co) -> ci
    const int H = h * stride, W = w * stride;
    int ho, wo, pt, pl;
    same_pads(H, kh, stride, &ho, &pt);
    same_pads(W, kw, stride, &wo, &pl);
    const PackGeom g = pack_geom(kh, kw, co, ci, STMI_PACK_DGRAD);
    if (g.row) return 0;
    ConvGemmParams p;
    clear_params(p);
    p.alpha = 1.f;
    p.N = n; p.Hi = ho; p.Wi = wo; p.Ci = ci;
    p.Ho = H; p.Wo = W; p.Co = co;
    p.klen = g.klen; p.kpad = g.kpad; p.co_pad = g.rows_pad;
    p.in_stride = 1;
    ConvGemmParams m;
    if (!build_multi_class(p, H, W, kh, kw, stride, pt, pl, dtype, m)) return 0;
    return conv_up2_tile_stats_tiles(m, dtype);
}

// The forward of a transposed 3x3 stride-2 convolution x[n,h,w,ci] -> y[n,2h,2w,co] absorbing the norm + ReLU in front
// of it (up-sampling kernel, bf16, 64 input channels).  ws: n * ci clamp values, n packed-weight copies, n * 9 * co
// floats of constant terms by output class (the four classes use different taps).
long conv_transpose_absorb_norm_ws_elems(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (dtype != STMI_BF16 || kh != 3 || kw != 3 || stride != 2 || ci * 2 != 128) return 0;
    if (conv_transpose_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype) <= 0) return 0;
    const PackGeom g = pack_geom(kh, kw, co, ci, STMI_PACK_DGRAD);
    return ((long)n * ci + 7) / 8 * 8 + (long)n * g.ntaps * g.rows_pad * g.kpad + 2 * (long)n * 4 * co;
}

int conv_transpose_fwd_absorb_norm_impl(const void* x, const float* in_tiles, int in_ts, const float* in_scale,
                                        const float* in_shift, float eps, float* in_stats, const void* wp, void* y,
                                        float* tile_stats, void* ws, int n, int h, int w, int ci, int co, int kh, int kw,
                                        int stride, int dtype, hipStream_t s) {
    STMI_REQUIRE(conv_transpose_absorb_norm_ws_elems(n, h, w, ci, co, kh, kw, stride, dtype) > 0,
                 "transposed conv absorb norm: this geometry is not supported (stmi_conv2d_transpose_absorb_norm_ws_bytes returned 0)");
    const PackGeom g = pack_geom(kh, kw, co, ci, STMI_PACK_DGRAD);
    const long welems = (long)g.ntaps * g.rows_pad * g.kpad;
    const size_t es = 2;
    void* clamp_t = ws;
    char* wps = static_cast<char*>(ws) + (((long)n * ci + 7) / 8 * 8) * es;
    float* bias = reinterpret_cast<float*>(wps + (((long)n * welems * es + 15) / 16) * 16);
    int rc = in_stats_from_tiles_impl(in_tiles, in_ts, in_stats, eps, n, ci, s);
    if (rc) return rc;
    // class slot (2 (oy & 1) + (ox & 1), as the kernel numbers its accumulators) of every tap of the 3 x 3 kernel
    signed char slot_of[32];
    for (int i = 0; i < 32; i++) slot_of[i] = -1;
    {
        const int H = h * stride, W = w * stride;
        int ho, wo, pt, pl;
        same_pads(H, kh, stride, &ho, &pt);
        same_pads(W, kw, stride, &wo, &pl);
        ConvGemmParams p, m;
        clear_params(p);
        p.N = n; p.Hi = ho; p.Wi = wo; p.Ci = ci; p.Ho = H; p.Wo = W; p.Co = co;
        p.klen = g.klen; p.kpad = g.kpad; p.co_pad = g.rows_pad; p.in_stride = 1;
        STMI_REQUIRE(build_multi_class(p, H, W, kh, kw, stride, pt, pl, dtype, m), "transposed conv absorb norm: no class table");
        for (int c = 0; c < m.nclasses; c++)
            for (int t = m.cls[c].tap0; t < m.cls[c].tap0 + m.cls[c].ntaps; t++)
                slot_of[m.taps[t].widx] = (signed char)(2 * m.cls[c].oy0 + m.cls[c].ox0);
    }
    rc = in_absorb_prep_impl(in_stats, in_scale, in_shift, wp, clamp_t, wps, bias, n, ci, g.kpad, g.rows_pad, g.ntaps, co, dtype, s,
                             slot_of, 4);
    if (rc) return rc;
    const AbsorbClamp cl{clamp_t, bias, welems};
    const int ts = conv_transpose_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
    return conv_dgrad_impl(x, wps, y, nullptr, nullptr, 1.f, n, h * stride, w * stride, co, ci, kh, kw, stride, dtype, s, nullptr,
                           tile_stats, ts, nullptr, 0, &cl);
}

int gram_bwd_impl(const void* f, const void* sp, void* df, int accumulate, int n, int h, int w, int c, int dtype,
                  hipStream_t s) {
    ConvGemmParams p;
    clear_params(p);
    p.in = f; p.wp = sp; p.out = df; p.addend = accumulate ? df : nullptr;
    p.N = n; p.Hi = h; p.Wi = w; p.Ci = c;
    p.Ho = h; p.Wo = w; p.Co = c;
    p.Hs = h; p.Ws = w; p.oys = 1; p.oxs = 1;
    p.in_stride = 1;
    p.ntaps = 1; p.klen = c; p.kpad = round_up(c, 32); p.co_pad = packed_rows(c);
    p.cdiv = 0;
    p.taps[0].dy = 0; p.taps[0].dx = 0; p.taps[0].widx = 0;
    p.w_batch_stride = (long)p.co_pad * p.kpad;
    STMI_REQUIRE(c % 8 == 0, "gram bwd: channel count must be a multiple of 8 (c=%d)", c);
    return launch_conv_gemm(p, dtype, s);
}

int pack_weights_impl(const float* w, void* out, int kh, int kw, int ci, int co, int mode, int dtype, hipStream_t s) {
    const PackGeom g = pack_geom(kh, kw, ci, co, mode);
    const long total = (long)g.ntaps * g.rows_pad * g.kpad;
    STMI_REQUIRE(total < (1L << 31), "pack_conv_weights: packed tensor too large");
    const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
    if (dtype == STMI_BF16)
        hipLaunchKernelGGL(pack_weights_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, w, (bf16_t*)out, kh, kw, ci, co, mode, g);
    else
        hipLaunchKernelGGL(pack_weights_kernel<float>, dim3(blocks), dim3(256), 0, s, w, (float*)out, kh, kw, ci, co, mode, g);
    STMI_LAUNCH_CHECK("pack_weights_kernel");
    return STMI_OK;
}

}  // namespace stmi
