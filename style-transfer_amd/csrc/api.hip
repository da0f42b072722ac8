// extern "C" surface of libstmi355.so (declared in include/stmi.h).
#include <stdarg.h>

#include "conv_gemm.hpp"

namespace stmi {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return STMI_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return STMI_ERR_HIP;
}

int ensure_dynamic_lds(const void* fn, int bytes, const char* what, std::atomic<unsigned long long>& done) {
    int dev = 0;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    const unsigned long long bit = dev >= 0 && dev < 64 ? 1ull << dev : 0;   // beyond 64 devices: set on every launch
    if (bit && (done.load(std::memory_order_acquire) & bit)) return STMI_OK;
    rc = check_hip(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
    if (rc) return rc;
    if (bit) done.fetch_or(bit, std::memory_order_release);
    return STMI_OK;
}

// implemented in the kernel translation units
const char* last_conv_kernel();
int pack_weights_impl(const float*, void*, int, int, int, int, int, int, hipStream_t);
int pack_weights_multi_impl(const float*, void*, const stmi_pack_desc*, int, int, hipStream_t);
int conv_fwd_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int conv_fwd_tile_stats_free(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int in_apply_tiles_impl(const void* x, const float* tiles, int ts, float* stats, float eps, const float* scale,
                        const float* shift, const void* residual, void* y, float* y_f32, int act, int n, int h, int w,
                        int c, int dtype, hipStream_t s);
int conv_fwd_impl(const void* x, const void* wp, const float* bias, void* y, void* y_pre, const void* addend,
                  int n, int h, int w, int ci, int co, int kh, int kw, int stride, int relu, int dtype, hipStream_t s,
                  float* tile_stats = nullptr, int tile_stats_ts = 0, float* ws = nullptr, long ws_floats = 0,
                  void* y_pool = nullptr);
int conv_dgrad_impl(const void* dy, const void* wp, void* dx, const void* mask, const void* addend, float alpha,
                    int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, hipStream_t s,
                    const InBwdSums* bn = nullptr, float* tile_stats = nullptr, int tile_stats_ts = 0,
                    float* ws = nullptr, long ws_floats = 0, const AbsorbClamp* cl = nullptr);
long conv_transpose_absorb_norm_ws_elems(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int conv_transpose_fwd_absorb_norm_impl(const void* x, const float* in_tiles, int in_ts, const float* in_scale,
                                        const float* in_shift, float eps, float* in_stats, const void* wp, void* y,
                                        float* tile_stats, void* ws, int n, int h, int w, int ci, int co, int kh, int kw,
                                        int stride, int dtype, hipStream_t s);
long conv_absorb_norm_ws_elems(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int conv_absorb_norm_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int conv_fwd_absorb_norm_impl(const void* x, const float* in_tiles, int in_ts, const float* in_scale, const float* in_shift,
                              float eps, float* in_stats, const void* wp, void* y, float* tile_stats, void* ws, int n, int h,
                              int w, int ci, int co, int kh, int kw, int stride, int dtype, hipStream_t s);
long conv_to3_absorb_norm_ws_elems(int n, int ci, int k, int dtype);
int conv_to3_absorb_norm_impl(const void* x, const float* in_tiles, int in_ts, const float* in_scale, const float* in_shift,
                              float eps, float* in_stats, const void* wq, void* y, void* ws, int n, int h, int w, int ci, int k,
                              int dtype, hipStream_t s);
long conv_ksplit_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, int dgrad);
int conv_transpose_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int conv_dgrad_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int in_bwd_rows_impl(const void* x, const void* dy, const float* stats, const float* scale, const float* shift, void* dx,
                     float* dscale, float* dshift, int accumulate, const float* rows, int nrows, int act, int n, int h,
                     int w, int c, int dtype, hipStream_t s);
int conv_rgb_fwd_impl(const float* x, float scale, const float* mean, const void* wq, const float* bias, void* y,
                      void* y_pre, int n, int h, int w, int co, int k, int relu, int dtype, hipStream_t s,
                      float* tile_stats = nullptr);
int conv_rgb_tile_stats_tiles(int h, int w, int co, int k);
int conv_rgb_u8_fwd_impl(const unsigned char* x, float scale, const float* mean, const void* wq, void* y, int n, int h,
                         int w, int co, int k, int dtype, hipStream_t s, float* tile_stats);
int in_apply_u8_impl(const void* x, const float* stats, const float* scale, const float* shift, unsigned char* y,
                     int act, int n, int h, int w, int c, int dtype, hipStream_t s);
int conv_rgb_dgrad_impl(const void* dy, const void* wq, void* dx, int n, int h, int w, int ci, int k, int dtype,
                        hipStream_t s);
int conv_to3_supported(int k, int ci, int dtype);
int conv_to3_impl(const void* x, const void* wq, const void* addend, void* y, int n, int h, int w, int ci, int k,
                  int dtype, hipStream_t s);
const char* last_wgrad_kernel();
int conv_wgrad_impl(const void* x, const void* dy, float* dw, float* ws, int n, int h, int w, int ci, int co, int kh,
                    int kw, int stride, int accumulate, int dtype, hipStream_t s);
long conv_wgrad_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype);
int gram_fwd_impl(const void* f, float* g, float* ws, int n, int h, int w, int c, int dtype, hipStream_t s,
                  int accumulate);
long gram_ws_floats(int n, int h, int w, int c, int dtype);
int zero_multi_impl(float* const* ptrs, const long* counts, int n, int* step, hipStream_t s);
int gram_bwd_impl(const void* f, const void* sp, void* df, int accumulate, int n, int h, int w, int c, int dtype,
                  hipStream_t s);
long in_ws_floats(int n, int h, int w, int c);
int in_param_grads_impl(const float* rows, int nrows, int n, int c, float* dscale, float* dshift, int accumulate,
                        hipStream_t s);
int in_bwd_rows_count(int n, int h, int w, int c, int dtype);
int in_stats_impl(const void* x, float* ws, float* stats, float eps, int n, int h, int w, int c, int dtype,
                  hipStream_t s);
int in_fwd_impl(const void* x, float* ws, float* stats, float eps, const float* scale, const float* shift,
                const void* residual, void* y, float* y_f32, int act, int n, int h, int w, int c, int dtype,
                hipStream_t s);
int in_stats_static_impl(const void* x, float* stats, float eps, int n, int h, int w, int c, int dtype, hipStream_t s);
int in_apply_impl(const void* x, const float* stats, const float* scale, const float* shift, const void* residual,
                  void* y, float* y_f32, int act, int n, int h, int w, int c, int dtype, hipStream_t s);
int in_bwd_impl(const void* x, const void* dy, const float* stats, const float* scale, const float* shift, void* dx,
                float* dscale, float* dshift, int accumulate, float* ws, int act, int n, int h, int w, int c, int dtype,
                hipStream_t s);
int scale_shift_impl(const float*, const float*, void*, float, long, int, int, hipStream_t);
int cast_impl(const void*, void*, long, int, int, hipStream_t);
int clip_impl(float*, long, float, float, hipStream_t);
int activation_impl(const void*, void*, long, int, int, hipStream_t);
int pixels_to_rows_impl(const void*, void*, int, int, int, int, int, int, int, hipStream_t);
int rows_to_pixels_sum_impl(const void*, const void*, void*, int, int, int, int, int, int, int, hipStream_t);
int rows_wgrad_to_hwio_impl(const float*, float*, int, int, int, int, int, hipStream_t);
int pool_fwd_impl(const void*, void*, int, int, int, int, int, hipStream_t);
int pool_bwd_impl(const void*, const void*, void*, int, int, int, int, int, hipStream_t);
int mse_impl(const void*, const void*, void*, float*, float, float, long, int, hipStream_t);
int gram_mse_impl(const float*, const float*, void*, float*, float, float, int, int, int, hipStream_t);
int tv_impl(const float*, void*, float*, float, int, int, int, int, int, hipStream_t);
int loss_slots_sum_impl(const float* slots, const int* first, int nterms, float* out, hipStream_t s);
int adam_impl(float*, const float*, float*, float*, long, float, float, float, float, float, int*, int, hipStream_t);

static int check_dtype(int dtype) {
    if (dtype == STMI_F32 || dtype == STMI_BF16) return STMI_OK;
    set_error("unknown dtype %d", dtype);
    return STMI_ERR_INVALID;
}

static int check_conv(const char* who, int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    if (n < 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || kh > 9 || kw > 9) {
        set_error("%s: bad geometry n=%d h=%d w=%d ci=%d co=%d k=%dx%d stride=%d", who, n, h, w, ci, co, kh, kw, stride);
        return STMI_ERR_INVALID;
    }
    return STMI_OK;
}

}  // namespace stmi

using namespace stmi;

#define S(stream) ((hipStream_t)(stream))

extern "C" {

int stmi_version(void) { return 100; }

const char* stmi_last_error(void) { return g_err; }

int stmi_device_arch(char* buf, int buflen) {
    int dev = 0;
    int rc = check_hip(hipGetDevice(&dev), "hipGetDevice");
    if (rc) return rc;
    hipDeviceProp_t prop;
    rc = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties");
    if (rc) return rc;
    snprintf(buf, buflen, "%s", prop.gcnArchName);
    return STMI_OK;
}

int stmi_last_conv_kernel(char* buf, int buflen) {
    STMI_REQUIRE(buf && buflen > 0, "last_conv_kernel: bad arguments");
    snprintf(buf, buflen, "%s", last_conv_kernel());
    return STMI_OK;
}

long stmi_conv_packed_elems(int kh, int kw, int ci, int co, int mode) {
    const PackGeom g = pack_geom(kh, kw, ci, co, mode);
    return (long)g.ntaps * g.rows_pad * g.kpad;
}

int stmi_pack_conv_weights(const float* w_hwio, void* packed, int kh, int kw, int ci, int co, int mode, int dtype,
                           void* stream) {
    if (check_conv("pack_conv_weights", 0, 1, 1, ci, co, kh, kw, 1, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(mode >= STMI_PACK_FWD && mode <= STMI_PACK_DGRAD_RGB, "pack_conv_weights: bad mode %d", mode);
    STMI_REQUIRE(mode != STMI_PACK_FWD_RGB || (ci == 3 && kh == kw && (kh & 1) && co % 16 == 0),
                 "pack_conv_weights: the 3-channel layout needs ci == 3, an odd square kernel and co %% 16 == 0");
    STMI_REQUIRE(mode != STMI_PACK_DGRAD_RGB || (co == 3 && kh == kw && (kh & 1) && ci % 16 == 0),
                 "pack_conv_weights: the 3-channel backward layout needs co == 3, an odd square kernel and ci %% 16 == 0");
    STMI_REQUIRE(mode != STMI_PACK_FWD_ROWOUT || (kw * co <= 32 && ci % 8 == 0),
                 "pack_conv_weights: row-patch output needs kw*co <= 32 and ci %% 8 == 0 (kw=%d co=%d ci=%d)", kw, co, ci);
    STMI_REQUIRE(w_hwio && packed, "pack_conv_weights: null pointer");
    return pack_weights_impl(w_hwio, packed, kh, kw, ci, co, mode, dtype, S(stream));
}

int stmi_pack_conv_weights_multi(const float* src_base, void* dst_base, const stmi_pack_desc* descs, int n, int dtype,
                                 void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(src_base && dst_base && descs && n >= 0, "pack_conv_weights_multi: bad arguments");
    for (int i = 0; i < n; i++) {
        const stmi_pack_desc& d = descs[i];
        if (check_conv("pack_conv_weights_multi", 0, 1, 1, d.ci, d.co, d.kh, d.kw, 1, dtype)) return STMI_ERR_INVALID;
        STMI_REQUIRE(d.mode >= STMI_PACK_FWD && d.mode <= STMI_PACK_DGRAD_RGB && d.src_off >= 0 && d.dst_off >= 0,
                     "pack_conv_weights_multi: bad descriptor %d", i);
        STMI_REQUIRE(d.mode != STMI_PACK_FWD_RGB || (d.ci == 3 && d.kh == d.kw && (d.kh & 1) && d.co % 16 == 0),
                     "pack_conv_weights_multi: descriptor %d: the 3-channel layout needs ci == 3, an odd square kernel and co %% 16 == 0", i);
        STMI_REQUIRE(d.mode != STMI_PACK_DGRAD_RGB || (d.co == 3 && d.kh == d.kw && (d.kh & 1) && d.ci % 16 == 0),
                     "pack_conv_weights_multi: descriptor %d: the 3-channel backward layout needs co == 3, an odd square kernel and ci %% 16 == 0", i);
        STMI_REQUIRE(d.mode != STMI_PACK_FWD_ROWOUT || (d.kw * d.co <= 32 && d.ci % 8 == 0),
                     "pack_conv_weights_multi: descriptor %d: row-patch output needs kw*co <= 32 and ci %% 8 == 0", i);
    }
    if (n == 0) return STMI_OK;
    return pack_weights_multi_impl(src_base, dst_base, descs, n, dtype, S(stream));
}

int stmi_conv2d_nhwc_fwd(const void* x, const void* w_packed_fwd, const float* bias, void* y, void* y_pre,
                         float* ws, long ws_floats, int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                         int relu, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_fwd", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && w_packed_fwd && y, "conv2d_nhwc_fwd: null pointer");
    STMI_REQUIRE(ws_floats >= 0 && (ws_floats == 0 || (ws && (((uintptr_t)ws) & 15) == 0)),
                 "conv2d_nhwc_fwd: the workspace must be 16-byte aligned");
    return conv_fwd_impl(x, w_packed_fwd, bias, y, y_pre, nullptr, n, h, w, ci, co, kh, kw, stride, relu, dtype, S(stream),
                         nullptr, 0, ws, ws_floats);
}

int stmi_conv2d_nhwc_fwd_pool(const void* x, const void* w_packed_fwd, const float* bias, void* y, void* y_pool, int n,
                              int h, int w, int ci, int co, int kh, int kw, int stride, int relu, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_fwd_pool", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && w_packed_fwd && y && y_pool, "conv2d_nhwc_fwd_pool: null pointer");
    return conv_fwd_impl(x, w_packed_fwd, bias, y, nullptr, nullptr, n, h, w, ci, co, kh, kw, stride, relu, dtype, S(stream),
                         nullptr, 0, nullptr, 0, y_pool);
}

long stmi_conv2d_absorb_norm_ws_bytes(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || (dtype != STMI_F32 && dtype != STMI_BF16)) return 0;
    const long elems = conv_absorb_norm_ws_elems(n, h, w, ci, co, kh, kw, stride, dtype);
    return elems > 0 ? elems * (dtype == STMI_BF16 ? 2 : 4) + 32 : 0;
}

int stmi_conv2d_absorb_norm_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || (dtype != STMI_F32 && dtype != STMI_BF16)) return 0;
    return conv_absorb_norm_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
}

int stmi_conv2d_nhwc_fwd_absorb_norm(const void* x, const float* in_tile_stats, int in_tiles, const float* in_scale,
                                     const float* in_shift, float eps, float* in_stats, const void* w_packed_fwd, void* y,
                                     float* tile_stats, void* ws, int n, int h, int w, int ci, int co, int kh, int kw,
                                     int stride, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_fwd_absorb_norm", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x && in_tile_stats && in_scale && in_shift && in_stats && w_packed_fwd && y && tile_stats && ws,
                 "conv2d_nhwc_fwd_absorb_norm: null pointer");
    STMI_REQUIRE(in_tiles > 0 && (((uintptr_t)ws) & 15) == 0, "conv2d_nhwc_fwd_absorb_norm: bad tile count or unaligned workspace");
    return conv_fwd_absorb_norm_impl(x, in_tile_stats, in_tiles, in_scale, in_shift, eps, in_stats, w_packed_fwd, y,
                                     tile_stats, ws, n, h, w, ci, co, kh, kw, stride, dtype, S(stream));
}

long stmi_conv2d_ksplit_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype, int dgrad) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_ksplit_ws_floats(n, h, w, ci, co, kh, kw, stride, dtype, dgrad);
}

int stmi_conv2d_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_fwd_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
}

int stmi_conv2d_tile_stats_free(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_fwd_tile_stats_free(n, h, w, ci, co, kh, kw, stride, dtype);
}

int stmi_conv2d_nhwc_fwd_tile_stats(const void* x, const void* w_packed_fwd, const float* bias, void* y, void* y_pre,
                                    float* tile_stats, int n, int h, int w, int ci, int co, int kh, int kw, int stride,
                                    int relu, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_fwd_tile_stats", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x && w_packed_fwd && y && tile_stats, "conv2d_nhwc_fwd_tile_stats: null pointer");
    const int ts = conv_fwd_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
    STMI_REQUIRE(ts > 0, "conv2d_nhwc_fwd_tile_stats: this geometry produces no tile statistics "
                 "(stmi_conv2d_tile_stats_tiles returned 0)");
    return conv_fwd_impl(x, w_packed_fwd, bias, y, y_pre, nullptr, n, h, w, ci, co, kh, kw, stride, relu, dtype, S(stream),
                         tile_stats, ts);
}

int stmi_conv2d_rgb_nhwc_fwd(const float* x_f32, float scale, const float* mean, const void* w_packed_rgb,
                             const float* bias, void* y, void* y_pre, int n, int h, int w, int co, int k, int relu,
                             int dtype, void* stream) {
    if (check_conv("conv2d_rgb_nhwc_fwd", n, h, w, 3, co, k, k, 1, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x_f32 && w_packed_rgb && y, "conv2d_rgb_nhwc_fwd: null pointer");
    return conv_rgb_fwd_impl(x_f32, scale, mean, w_packed_rgb, bias, y, y_pre, n, h, w, co, k, relu, dtype, S(stream));
}

int stmi_conv2d_rgb_tile_stats_tiles(int h, int w, int co, int k) {
    if (h <= 0 || w <= 0 || co <= 0 || k <= 0) return 0;
    return conv_rgb_tile_stats_tiles(h, w, co, k);
}

int stmi_conv2d_rgb_nhwc_fwd_tile_stats(const float* x_f32, float scale, const float* mean, const void* w_packed_rgb,
                                        void* y, float* tile_stats, int n, int h, int w, int co, int k, int dtype,
                                        void* stream) {
    if (check_conv("conv2d_rgb_nhwc_fwd_tile_stats", n, h, w, 3, co, k, k, 1, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x_f32 && w_packed_rgb && y && tile_stats, "conv2d_rgb_nhwc_fwd_tile_stats: null pointer");
    STMI_REQUIRE(conv_rgb_tile_stats_tiles(h, w, co, k) > 0, "conv2d_rgb_nhwc_fwd_tile_stats: no tile statistics "
                 "for this (k, co) (stmi_conv2d_rgb_tile_stats_tiles returned 0)");
    return conv_rgb_fwd_impl(x_f32, scale, mean, w_packed_rgb, nullptr, y, nullptr, n, h, w, co, k, 0, dtype, S(stream),
                             tile_stats);
}

int stmi_conv2d_rgb_u8_nhwc_fwd(const unsigned char* x_u8, float scale, const float* mean, const void* w_packed_rgb,
                                void* y, float* tile_stats, int n, int h, int w, int co, int k, int dtype, void* stream) {
    if (check_conv("conv2d_rgb_u8_nhwc_fwd", n, h, w, 3, co, k, k, 1, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x_u8 && w_packed_rgb && y, "conv2d_rgb_u8_nhwc_fwd: null pointer");
    STMI_REQUIRE(!tile_stats || conv_rgb_tile_stats_tiles(h, w, co, k) > 0, "conv2d_rgb_u8_nhwc_fwd: no tile statistics "
                 "for this (k, co) (stmi_conv2d_rgb_tile_stats_tiles returned 0)");
    return conv_rgb_u8_fwd_impl(x_u8, scale, mean, w_packed_rgb, y, n, h, w, co, k, dtype, S(stream), tile_stats);
}

int stmi_conv2d_rgb_nhwc_dgrad(const void* dy, const void* w_packed_dgrad_rgb, void* dx, int n, int h, int w, int ci,
                               int k, int dtype, void* stream) {
    if (check_conv("conv2d_rgb_nhwc_dgrad", n, h, w, ci, 3, k, k, 1, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(dy && w_packed_dgrad_rgb && dx, "conv2d_rgb_nhwc_dgrad: null pointer");
    return conv_rgb_dgrad_impl(dy, w_packed_dgrad_rgb, dx, n, h, w, ci, k, dtype, S(stream));
}

int stmi_conv2d_to3_supported(int k, int ci, int dtype) { return conv_to3_supported(k, ci, dtype); }

int stmi_conv2d_to3_nhwc_fwd(const void* x, const void* w_packed_rowout, const void* addend, void* y, int n, int h,
                             int w, int ci, int k, int dtype, void* stream) {
    if (check_conv("conv2d_to3_nhwc_fwd", n, h, w, ci, 3, k, k, 1, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x && w_packed_rowout && y, "conv2d_to3_nhwc_fwd: null pointer");
    return conv_to3_impl(x, w_packed_rowout, addend, y, n, h, w, ci, k, dtype, S(stream));
}

long stmi_conv2d_to3_absorb_norm_ws_bytes(int n, int h, int w, int ci, int k, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || k <= 0 || (dtype != STMI_F32 && dtype != STMI_BF16)) return 0;
    const long elems = conv_to3_absorb_norm_ws_elems(n, ci, k, dtype);
    return elems > 0 ? elems * 2 + 32 : 0;
}

int stmi_conv2d_to3_nhwc_fwd_absorb_norm(const void* x, const float* in_tile_stats, int in_tiles, const float* in_scale,
                                         const float* in_shift, float eps, float* in_stats, const void* w_packed_rowout,
                                         void* y, void* ws, int n, int h, int w, int ci, int k, int dtype, void* stream) {
    if (check_conv("conv2d_to3_nhwc_fwd_absorb_norm", n, h, w, ci, 3, k, k, 1, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x && in_tile_stats && in_scale && in_shift && in_stats && w_packed_rowout && y && ws,
                 "conv2d_to3_nhwc_fwd_absorb_norm: null pointer");
    STMI_REQUIRE(in_tiles > 0 && (((uintptr_t)ws) & 15) == 0, "conv2d_to3_nhwc_fwd_absorb_norm: bad tile count or unaligned workspace");
    return conv_to3_absorb_norm_impl(x, in_tile_stats, in_tiles, in_scale, in_shift, eps, in_stats, w_packed_rowout, y, ws,
                                     n, h, w, ci, k, dtype, S(stream));
}

int stmi_conv2d_nhwc_dgrad(const void* dy, const void* w_packed_dgrad, void* dx, const void* relu_mask,
                           const void* addend, float alpha, float* ws, long ws_floats, int n, int h, int w, int ci,
                           int co, int kh, int kw, int stride, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_dgrad", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(dy && w_packed_dgrad && dx, "conv2d_nhwc_dgrad: null pointer");
    STMI_REQUIRE(ws_floats >= 0 && (ws_floats == 0 || (ws && (((uintptr_t)ws) & 15) == 0)),
                 "conv2d_nhwc_dgrad: the workspace must be 16-byte aligned");
    return conv_dgrad_impl(dy, w_packed_dgrad, dx, relu_mask, addend, alpha, n, h, w, ci, co, kh, kw, stride, dtype, S(stream),
                           nullptr, nullptr, 0, ws, ws_floats);
}

int stmi_conv2d_dgrad_in_sums_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_dgrad_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
}

int stmi_conv2d_nhwc_dgrad_in_sums(const void* dy, const void* w_packed_dgrad, void* dx, const void* relu_mask,
                                   const void* addend, float alpha, const void* bn_x, const float* bn_stats,
                                   const float* bn_scale, const float* bn_shift, int bn_act, float* rows, int n, int h,
                                   int w, int ci, int co, int kh, int kw, int stride, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_dgrad_in_sums", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(dy && w_packed_dgrad && dx && bn_x && bn_stats && bn_scale && bn_shift && rows,
                 "conv2d_nhwc_dgrad_in_sums: null pointer");
    STMI_REQUIRE(bn_act >= STMI_ACT_NONE && bn_act <= STMI_ACT_SIGMOID, "conv2d_nhwc_dgrad_in_sums: bad activation %d", bn_act);
    InBwdSums bn;
    bn.rows = rows;
    bn.tiles = conv_dgrad_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
    STMI_REQUIRE(bn.tiles > 0, "conv2d_nhwc_dgrad_in_sums: this geometry produces no sums "
                 "(stmi_conv2d_dgrad_in_sums_tiles returned 0)");
    bn.x = bn_x; bn.stats = bn_stats; bn.scale = bn_scale; bn.shift = bn_shift; bn.act = bn_act;
    return conv_dgrad_impl(dy, w_packed_dgrad, dx, relu_mask, addend, alpha, n, h, w, ci, co, kh, kw, stride, dtype,
                           S(stream), &bn);
}

int stmi_last_wgrad_kernel(char* buf, int buflen) {
    STMI_REQUIRE(buf && buflen > 0, "last_wgrad_kernel: bad arguments");
    snprintf(buf, buflen, "%s", last_wgrad_kernel());
    return STMI_OK;
}

long stmi_conv2d_wgrad_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || kh > 9 || kw > 9) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_wgrad_ws_floats(n, h, w, ci, co, kh, kw, stride, dtype);
}

int stmi_conv2d_nhwc_wgrad(const void* x, const void* dy, float* dw, float* ws, int n, int h, int w, int ci, int co,
                           int kh, int kw, int stride, int accumulate, int dtype, void* stream) {
    if (check_conv("conv2d_nhwc_wgrad", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && dy && dw, "conv2d_nhwc_wgrad: null pointer");
    return conv_wgrad_impl(x, dy, dw, ws, n, h, w, ci, co, kh, kw, stride, accumulate, dtype, S(stream));
}

// A transposed conv x[n,h,w,ci] -> y[n,sh,sw,co] with W[kh,kw,co,ci] is the backward-data of the
// conv (n, sh, sw, co) -> ci whose HWIO weight is W.
int stmi_conv2d_transpose_nhwc_fwd(const void* x, const void* w_packed_dgrad, void* y, int n, int h, int w, int ci,
                                   int co, int kh, int kw, int stride, int dtype, void* stream) {
    if (check_conv("conv2d_transpose_nhwc_fwd", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && w_packed_dgrad && y, "conv2d_transpose_nhwc_fwd: null pointer");
    return conv_dgrad_impl(x, w_packed_dgrad, y, nullptr, nullptr, 1.f, n, h * stride, w * stride, co, ci, kh, kw, stride, dtype, S(stream));
}

int stmi_conv2d_transpose_tile_stats_tiles(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_transpose_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
}

int stmi_conv2d_transpose_nhwc_fwd_tile_stats(const void* x, const void* w_packed_dgrad, void* y, float* tile_stats,
                                              int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype,
                                              void* stream) {
    if (check_conv("conv2d_transpose_nhwc_fwd_tile_stats", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x && w_packed_dgrad && y && tile_stats, "conv2d_transpose_nhwc_fwd_tile_stats: null pointer");
    const int ts = conv_transpose_tile_stats_tiles(n, h, w, ci, co, kh, kw, stride, dtype);
    STMI_REQUIRE(ts > 0, "conv2d_transpose_nhwc_fwd_tile_stats: this geometry produces no tile statistics "
                 "(stmi_conv2d_transpose_tile_stats_tiles returned 0)");
    return conv_dgrad_impl(x, w_packed_dgrad, y, nullptr, nullptr, 1.f, n, h * stride, w * stride, co, ci, kh, kw, stride,
                           dtype, S(stream), nullptr, tile_stats, ts);
}

long stmi_conv2d_transpose_absorb_norm_ws_bytes(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || (dtype != STMI_F32 && dtype != STMI_BF16)) return 0;
    const long elems = conv_transpose_absorb_norm_ws_elems(n, h, w, ci, co, kh, kw, stride, dtype);
    return elems > 0 ? elems * 2 + 32 : 0;
}

int stmi_conv2d_transpose_nhwc_fwd_absorb_norm(const void* x, const float* in_tile_stats, int in_tiles, const float* in_scale,
                                               const float* in_shift, float eps, float* in_stats, const void* w_packed_dgrad,
                                               void* y, float* tile_stats, void* ws, int n, int h, int w, int ci, int co,
                                               int kh, int kw, int stride, int dtype, void* stream) {
    if (check_conv("conv2d_transpose_nhwc_fwd_absorb_norm", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;  // an empty batch has no buffers
    STMI_REQUIRE(x && in_tile_stats && in_scale && in_shift && in_stats && w_packed_dgrad && y && tile_stats && ws,
                 "conv2d_transpose_nhwc_fwd_absorb_norm: null pointer");
    STMI_REQUIRE(in_tiles > 0 && (((uintptr_t)ws) & 15) == 0, "conv2d_transpose_nhwc_fwd_absorb_norm: bad tile count or unaligned workspace");
    return conv_transpose_fwd_absorb_norm_impl(x, in_tile_stats, in_tiles, in_scale, in_shift, eps, in_stats, w_packed_dgrad, y,
                                               tile_stats, ws, n, h, w, ci, co, kh, kw, stride, dtype, S(stream));
}

int stmi_conv2d_transpose_nhwc_dgrad(const void* dy, const void* w_packed_fwd, void* dx, const void* addend, int n,
                                     int h, int w, int ci, int co, int kh, int kw, int stride, int dtype,
                                     void* stream) {
    if (check_conv("conv2d_transpose_nhwc_dgrad", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(dy && w_packed_fwd && dx, "conv2d_transpose_nhwc_dgrad: null pointer");
    return conv_fwd_impl(dy, w_packed_fwd, nullptr, dx, nullptr, addend, n, h * stride, w * stride, co, ci, kh, kw, stride, 0, dtype, S(stream));
}

long stmi_conv2d_transpose_wgrad_ws_floats(int n, int h, int w, int ci, int co, int kh, int kw, int stride, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || ci <= 0 || co <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || kh > 9 || kw > 9) return 0;
    if (dtype != STMI_F32 && dtype != STMI_BF16) return 0;
    return conv_wgrad_ws_floats(n, h * stride, w * stride, co, ci, kh, kw, stride, dtype);
}

int stmi_conv2d_transpose_nhwc_wgrad(const void* x, const void* dy, float* dw, float* ws, int n, int h, int w, int ci,
                                     int co, int kh, int kw, int stride, int accumulate, int dtype, void* stream) {
    if (check_conv("conv2d_transpose_nhwc_wgrad", n, h, w, ci, co, kh, kw, stride, dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && dy && dw, "conv2d_transpose_nhwc_wgrad: null pointer");
    return conv_wgrad_impl(dy, x, dw, ws, n, h * stride, w * stride, co, ci, kh, kw, stride, accumulate, dtype, S(stream));
}

long stmi_instance_norm_ws_floats(int n, int h, int w, int c) { return in_ws_floats(n, h, w, c); }

int stmi_instance_norm_nhwc_stats(const void* x, float* ws, float* stats, float eps, int n, int h, int w, int c,
                                  int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && ws && stats && n >= 0 && h > 0 && w > 0 && c > 0, "instance_norm_nhwc_stats: bad arguments");
    if (n == 0) return STMI_OK;
    return in_stats_impl(x, ws, stats, eps, n, h, w, c, dtype, S(stream));
}

int stmi_instance_norm_nhwc_stats_static(const void* x, float* stats, float eps, int n, int h, int w, int c, int dtype,
                                         void* stream) {
    STMI_REQUIRE(x && stats && n >= 0 && h > 0 && w > 0 && c > 0, "instance_norm_nhwc_stats_static: bad arguments");
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    if (n == 0) return STMI_OK;
    return in_stats_static_impl(x, stats, eps, n, h, w, c, dtype, S(stream));
}

int stmi_instance_norm_nhwc_apply(const void* x, const float* stats, const float* scale, const float* shift,
                                  const void* residual, void* y, float* y_f32, int act,
                                  int n, int h, int w, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && stats && scale && shift && y && n >= 0 && h > 0 && w > 0 && c > 0, "instance_norm_nhwc_apply: bad arguments");
    STMI_REQUIRE(act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "instance_norm_nhwc_apply: bad activation %d", act);
    if (n == 0) return STMI_OK;
    return in_apply_impl(x, stats, scale, shift, residual, y, y_f32, act, n, h, w, c, dtype, S(stream));
}

int stmi_instance_norm_nhwc_apply_u8(const void* x, const float* stats, const float* scale, const float* shift,
                                     unsigned char* y_u8, int act, int n, int h, int w, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && stats && scale && shift && y_u8 && n >= 0 && h > 0 && w > 0 && c > 0, "instance_norm_nhwc_apply_u8: bad arguments");
    STMI_REQUIRE(act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "instance_norm_nhwc_apply_u8: bad activation %d", act);
    if (n == 0) return STMI_OK;
    return in_apply_u8_impl(x, stats, scale, shift, y_u8, act, n, h, w, c, dtype, S(stream));
}

int stmi_instance_norm_nhwc_fwd(const void* x, float* ws, float* stats, float eps, const float* scale, const float* shift,
                                const void* residual, void* y, float* y_f32, int act, int n, int h, int w, int c,
                                int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && ws && stats && scale && shift && y && n >= 0 && h > 0 && w > 0 && c > 0,
                 "instance_norm_nhwc_fwd: bad arguments");
    STMI_REQUIRE(act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "instance_norm_nhwc_fwd: bad activation %d", act);
    STMI_REQUIRE(y != x, "instance_norm_nhwc_fwd: y must not alias x (every apply block reads the statistics' shift "
                 "from x's first pixel while others already write y); stmi_instance_norm_nhwc_stats + _apply may run in place");
    if (n == 0) return STMI_OK;
    return in_fwd_impl(x, ws, stats, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, dtype, S(stream));
}

int stmi_instance_norm_nhwc_apply_tiles(const void* x, const float* tile_stats, int tiles, float* stats, float eps,
                                        const float* scale, const float* shift, const void* residual, void* y,
                                        float* y_f32, int act, int n, int h, int w, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && tile_stats && stats && scale && shift && y && n >= 0 && h > 0 && w > 0 && c > 0 && tiles > 0,
                 "instance_norm_nhwc_apply_tiles: bad arguments");
    STMI_REQUIRE(act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "instance_norm_nhwc_apply_tiles: bad activation %d", act);
    if (n == 0) return STMI_OK;
    return in_apply_tiles_impl(x, tile_stats, tiles, stats, eps, scale, shift, residual, y, y_f32, act, n, h, w, c, dtype,
                               S(stream));
}

int stmi_instance_norm_nhwc_bwd(const void* x, const void* dy, const float* stats, const float* scale,
                                const float* shift, void* dx, float* dscale, float* dshift, int accumulate, float* ws,
                                float eps, int act, int n, int h, int w, int c, int dtype, void* stream) {
    (void)eps;  // 1/sqrt(var+eps) is taken from `stats`
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && dy && stats && scale && shift && dx && ws && n >= 0 && h > 0 && w > 0 && c > 0 &&
                 (dscale == nullptr) == (dshift == nullptr), "instance_norm_nhwc_bwd: bad arguments");
    STMI_REQUIRE(act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "instance_norm_nhwc_bwd: bad activation %d", act);
    if (n == 0) return STMI_OK;
    return in_bwd_impl(x, dy, stats, scale, shift, dx, dscale, dshift, accumulate, ws, act, n, h, w, c, dtype, S(stream));
}

int stmi_instance_norm_nhwc_bwd_rows(const void* x, const void* dy, const float* stats, const float* scale,
                                     const float* shift, void* dx, float* dscale, float* dshift, int accumulate,
                                     const float* rows, int nrows, int act, int n, int h, int w, int c, int dtype,
                                     void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && dy && stats && scale && shift && dx && rows && nrows > 0 && n >= 0 && h > 0 && w > 0 && c > 0 &&
                 (dscale == nullptr) == (dshift == nullptr), "instance_norm_nhwc_bwd_rows: bad arguments");
    STMI_REQUIRE(act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "instance_norm_nhwc_bwd_rows: bad activation %d", act);
    if (n == 0) return STMI_OK;
    return in_bwd_rows_impl(x, dy, stats, scale, shift, dx, dscale, dshift, accumulate, rows, nrows, act, n, h, w, c, dtype,
                            S(stream));
}

int stmi_instance_norm_bwd_rows_count(int n, int h, int w, int c, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || (dtype != STMI_F32 && dtype != STMI_BF16)) return 0;
    return in_bwd_rows_count(n, h, w, c, dtype);
}

int stmi_instance_norm_param_grads(const float* rows, int nrows, int n, int c, float* dscale, float* dshift,
                                   int accumulate, void* stream) {
    STMI_REQUIRE(rows && dscale && dshift && nrows > 0 && n >= 0 && c > 0, "instance_norm_param_grads: bad arguments");
    if (n == 0) return STMI_OK;
    return in_param_grads_impl(rows, nrows, n, c, dscale, dshift, accumulate, S(stream));
}

int stmi_scale_shift(const float* x, const float* mean, void* y, float scale, long count, int c, int dtype,
                     void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && mean && y && count >= 0 && c > 0, "scale_shift: bad arguments");
    if (count == 0) return STMI_OK;
    return scale_shift_impl(x, mean, y, scale, count, c, dtype, S(stream));
}

int stmi_maxpool2x2_same_fwd(const void* x, void* y, int n, int h, int w, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && y && n >= 0 && h > 0 && w > 0 && c > 0, "maxpool2x2_same_fwd: bad arguments");
    if (n == 0) return STMI_OK;
    return pool_fwd_impl(x, y, n, h, w, c, dtype, S(stream));
}

int stmi_maxpool2x2_same_bwd(const void* x, const void* dy, void* dx, int n, int h, int w, int c, int dtype,
                             void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && dy && dx && n >= 0 && h > 0 && w > 0 && c > 0, "maxpool2x2_same_bwd: bad arguments");
    if (n == 0) return STMI_OK;
    return pool_bwd_impl(x, dy, dx, n, h, w, c, dtype, S(stream));
}

long stmi_gram_ws_floats(int n, int h, int w, int c, int dtype) {
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || (dtype != STMI_F32 && dtype != STMI_BF16)) return 0;
    return gram_ws_floats(n, h, w, c, dtype);
}

int stmi_gram_nhwc_fwd(const void* f, float* g, float* ws, int n, int h, int w, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(f && g && n >= 0 && h > 0 && w > 0 && c > 0, "gram_nhwc_fwd: bad arguments");
    if (n == 0) return STMI_OK;
    return gram_fwd_impl(f, g, ws, n, h, w, c, dtype, S(stream), 0);
}

int stmi_gram_nhwc_fwd_add(const void* f, float* g, float* ws, int n, int h, int w, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(f && g && n >= 0 && h > 0 && w > 0 && c > 0, "gram_nhwc_fwd_add: bad arguments");
    if (n == 0) return STMI_OK;
    return gram_fwd_impl(f, g, ws, n, h, w, c, dtype, S(stream), 1);
}

int stmi_zero_multi(float* const* ptrs, const long* counts, int n, int* step_dev, void* stream) {
    STMI_REQUIRE(n >= 0 && n <= STMI_ZERO_MULTI_MAX && (n == 0 || (ptrs && counts)), "zero_multi: bad arguments (n=%d, at most %d buffers)", n, STMI_ZERO_MULTI_MAX);
    for (int i = 0; i < n; i++) STMI_REQUIRE(counts[i] == 0 || ptrs[i], "zero_multi: null buffer %d", i);
    if (n == 0 && !step_dev) return STMI_OK;
    return zero_multi_impl(ptrs, counts, n, step_dev, S(stream));
}

int stmi_gram_nhwc_bwd(const void* f, const void* s, void* df, int accumulate, int n, int h, int w, int c, int dtype,
                       void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(f && s && df && n >= 0 && h > 0 && w > 0 && c > 0, "gram_nhwc_bwd: bad arguments");
    if (n == 0) return STMI_OK;
    return gram_bwd_impl(f, s, df, accumulate, n, h, w, c, dtype, S(stream));
}

long stmi_gram_packed_elems(int n, int c) {
    const PackGeom g = pack_geom(1, 1, c, c, STMI_PACK_FWD);
    return (long)n * g.rows_pad * g.kpad;
}

int stmi_gram_mse_fwd_bwd(const float* g, const float* target, void* s_out, float* slot, float loss_scale,
                          float grad_scale, int n, int c, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(g && target && s_out && slot && n >= 0 && c > 0 && c % 8 == 0, "gram_mse_fwd_bwd: bad arguments (c=%d)", c);
    STMI_REQUIRE(((((uintptr_t)g) | ((uintptr_t)target) | ((uintptr_t)s_out)) & 15) == 0, "gram_mse_fwd_bwd: g, target and s_out must be 16-byte aligned");
    if (n == 0) return STMI_OK;
    return gram_mse_impl(g, target, s_out, slot, loss_scale, grad_scale, n, c, dtype, S(stream));
}

int stmi_mse_fwd_bwd(const void* a, const void* b, void* grad, float* slot, float loss_scale, float grad_scale,
                     long count, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(a && b && slot && count >= 0, "mse_fwd_bwd: bad arguments");
    if (count == 0) return STMI_OK;
    return mse_impl(a, b, grad, slot, loss_scale, grad_scale, count, dtype, S(stream));
}

int stmi_tv_loss_fwd_bwd(const float* img, void* grad, float* slot, float weight, int n, int h, int w, int c,
                         int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(img && grad && slot && n >= 0 && h > 0 && w > 0 && c > 0, "tv_loss_fwd_bwd: bad arguments");
    if (n == 0) return STMI_OK;
    return tv_impl(img, grad, slot, weight, n, h, w, c, dtype, S(stream));
}

int stmi_loss_slots_sum(const float* slots, const int* first, int nterms, float* out, void* stream) {
    STMI_REQUIRE(slots && first && out && nterms >= 1 && nterms <= STMI_LOSS_TERMS_MAX,
                 "loss_slots_sum: bad arguments (1 .. %d terms)", STMI_LOSS_TERMS_MAX);
    for (int t = 0; t < nterms; t++)
        STMI_REQUIRE(first[t] >= 0 && first[t] <= first[t + 1], "loss_slots_sum: first[] must not decrease");
    STMI_REQUIRE((((uintptr_t)slots) & 15) == 0, "loss_slots_sum: slots must be 16-byte aligned");
    return loss_slots_sum_impl(slots, first, nterms, out, S(stream));
}

int stmi_adam_tf1(float* p, const float* g, float* m, float* v, long count, float lr, float beta1, float beta2,
                  float eps, float grad_scale, int* step_dev, int t, void* stream) {
    STMI_REQUIRE(p && g && m && v && count >= 0, "adam_tf1: bad arguments");
    STMI_REQUIRE(step_dev || t >= 1, "adam_tf1: step t must be >= 1 (got %d)", t);
    if (count == 0) return STMI_OK;
    return adam_impl(p, g, m, v, count, lr, beta1, beta2, eps, grad_scale, step_dev, t, S(stream));
}

int stmi_cast(const void* src, void* dst, long count, int dtype, int to_f32, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(src && dst && count >= 0, "cast: bad arguments");
    if (count == 0) return STMI_OK;
    return cast_impl(src, dst, count, dtype, to_f32, S(stream));
}

int stmi_pixels_to_rows(const void* x, void* rows, int n, int h, int w, int c, int kw, int dx0, int dtype,
                        void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && rows && n >= 0 && h > 0 && w > 0 && c > 0 && kw > 0 && kw * c <= 32, "pixels_to_rows: bad arguments");
    if (n == 0) return STMI_OK;
    return pixels_to_rows_impl(x, rows, n, h, w, c, kw, dx0, dtype, S(stream));
}

int stmi_rows_to_pixels_sum(const void* rows, const void* addend, void* y, int n, int h, int w, int c, int kw, int dx0, int dtype,
                            void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(rows && y && n >= 0 && h > 0 && w > 0 && c > 0 && kw > 0 && kw * c <= 32, "rows_to_pixels_sum: bad arguments");
    if (n == 0) return STMI_OK;
    return rows_to_pixels_sum_impl(rows, addend, y, n, h, w, c, kw, dx0, dtype, S(stream));
}

int stmi_rows_wgrad_to_hwio(const float* d, float* dw, int kh, int kw, int ci, int co, int mode, void* stream) {
    STMI_REQUIRE(d && dw && kh > 0 && kw > 0 && ci > 0 && co > 0 && (mode == 0 || mode == 1), "rows_wgrad_to_hwio: bad arguments");
    STMI_REQUIRE(mode == 0 ? kw * ci <= 32 : kw * co <= 32, "rows_wgrad_to_hwio: the patched side must fit 32 columns");
    return rows_wgrad_to_hwio_impl(d, dw, kh, kw, ci, co, mode, S(stream));
}

int stmi_activation(const void* x, void* y, long count, int act, int dtype, void* stream) {
    if (check_dtype(dtype)) return STMI_ERR_INVALID;
    STMI_REQUIRE(x && y && count >= 0 && act >= STMI_ACT_NONE && act <= STMI_ACT_SIGMOID, "activation: bad arguments");
    if (count == 0) return STMI_OK;
    return activation_impl(x, y, count, act, dtype, S(stream));
}

int stmi_clip(float* x, long count, float lo, float hi, void* stream) {
    STMI_REQUIRE(x && count >= 0, "clip: bad arguments");
    if (count == 0) return STMI_OK;
    return clip_impl(x, count, lo, hi, S(stream));
}

}  // extern "C"
