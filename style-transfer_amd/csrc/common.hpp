// Shared device/host helpers for the gfx950 style-transfer kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "../../include/stmi.h"

namespace stmi {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int kWave = 64;
constexpr int kMaxTaps = 81;

// One (tap) of a gather-GEMM: input offset relative to the output pixel's base
// position, and which [rows x K] slice of the packed weights it multiplies.
struct Tap {
    short dy, dx;
    int widx;
};

// ---------------------------------------------------------------------------
// error reporting: every extern "C" entry point returns 0 or a negative code
// and leaves a thread-local message for stmi_last_error().
// ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int check_hip(hipError_t e, const char* what);

#define STMI_REQUIRE(cond, ...)                      \
    do {                                             \
        if (!(cond)) {                               \
            ::stmi::set_error(__VA_ARGS__);          \
            return STMI_ERR_INVALID;                 \
        }                                            \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel AND device: the attribute belongs to a
// device's copy of the code object, so a process that uses several GPUs sets it on each.  `done` is a static
// of the call site (one per kernel instantiation) with one bit per device ordinal; the call is idempotent, so
// host threads racing here both succeed.
int ensure_dynamic_lds(const void* fn, int bytes, const char* what, std::atomic<unsigned long long>& done);

#define STMI_DYN_LDS(kernel, bytes, what)                                               \
    do {                                                                                \
        static std::atomic<unsigned long long> done_{0};                                \
        int rc_ = ::stmi::ensure_dynamic_lds((const void*)kernel, bytes, what, done_);  \
        if (rc_) return rc_;                                                            \
    } while (0)

#define STMI_LAUNCH_CHECK(name)                                         \
    do {                                                                \
        int _rc = ::stmi::check_hip(hipGetLastError(), name);           \
        if (_rc) return _rc;                                            \
    } while (0)

// ---------------------------------------------------------------------------
// element conversion
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<unsigned char>(unsigned char v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }

// V consecutive elements of T moved with the widest access that fits.
template <typename T, int V>
__device__ __forceinline__ void load_vec(const T* p, float (&out)[V]) {
    if constexpr ((sizeof(T) * V) % 16 == 0) {
        constexpr int NQ = sizeof(T) * V / 16;
        uint4 raw[NQ];
#pragma unroll
        for (int q = 0; q < NQ; q++) raw[q] = reinterpret_cast<const uint4*>(p)[q];
        const T* e = reinterpret_cast<const T*>(raw);
#pragma unroll
        for (int i = 0; i < V; i++) out[i] = to_f32<T>(e[i]);
    } else if constexpr ((sizeof(T) * V) == 8) {
        uint2 raw = *reinterpret_cast<const uint2*>(p);
        const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
        for (int i = 0; i < V; i++) out[i] = to_f32<T>(e[i]);
    } else {
#pragma unroll
        for (int i = 0; i < V; i++) out[i] = to_f32<T>(p[i]);
    }
}

// the V = 16 / sizeof(T) elements of a 16-byte register value
template <typename T, int V>
__device__ __forceinline__ void unpack_vec(const uint4& raw, float (&out)[V]) {
    static_assert(sizeof(T) * V == 16, "one 16-byte vector");
    const T* e = reinterpret_cast<const T*>(&raw);
#pragma unroll
    for (int i = 0; i < V; i++) out[i] = to_f32<T>(e[i]);
}

template <typename T, int V>
__device__ __forceinline__ void store_vec(T* p, const float (&in)[V]) {
    if constexpr ((sizeof(T) * V) % 16 == 0) {
        constexpr int NQ = sizeof(T) * V / 16;
        uint4 raw[NQ];
        T* e = reinterpret_cast<T*>(raw);
#pragma unroll
        for (int i = 0; i < V; i++) e[i] = from_f32<T>(in[i]);
#pragma unroll
        for (int q = 0; q < NQ; q++) reinterpret_cast<uint4*>(p)[q] = raw[q];
    } else if constexpr ((sizeof(T) * V) == 8) {
        uint2 raw;
        T* e = reinterpret_cast<T*>(&raw);
#pragma unroll
        for (int i = 0; i < V; i++) e[i] = from_f32<T>(in[i]);
        *reinterpret_cast<uint2*>(p) = raw;
    } else {
#pragma unroll
        for (int i = 0; i < V; i++) p[i] = from_f32<T>(in[i]);
    }
}

// ---------------------------------------------------------------------------
// MFMA wrapper: one "fragment" is 16 bytes of K per lane for a 16-row operand.
//   bf16: 8 elements, one v_mfma_f32_16x16x32_bf16 (K = 32 per step)
//   f32 : 4 elements, four v_mfma_f32_16x16x4_f32  (K = 16 per step); lane
//         group g = lane>>4 owns k = 4g..4g+3 of the step, instruction i takes
//         element i of every lane, so the four instructions cover all 16 k.
// Both produce the 16x16 C/D map row = 4*(lane>>4)+reg, col = lane&15.
// ---------------------------------------------------------------------------
template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static constexpr int kElems = 8;
    static __device__ __forceinline__ void mma(const frag& a, const frag& b, f32x4& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <> struct Mma<float> {
    typedef f32x4 frag;
    static constexpr int kElems = 4;
    static __device__ __forceinline__ void mma(const frag& a, const frag& b, f32x4& c) {
#pragma unroll
        for (int i = 0; i < 4; i++) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------
// reductions
// ---------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `red` holds 4 floats.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return r;
}

// Tuning knobs are environment variables read ONCE per process (function-local statics are
// initialised thread-safely): set them before the first launch.  STMI_ENV_LIVE=1 (tuning scripts that
// switch knobs inside one process) re-reads them on every use instead.
static inline bool env_live() {
    static const bool v = getenv("STMI_ENV_LIVE") != nullptr;
    return v;
}
#define STMI_ENV_SET(name) \
    ([]() -> bool { static const bool v = getenv(name) != nullptr; return ::stmi::env_live() ? getenv(name) != nullptr : v; }())
#define STMI_ENV_INT(name, dflt)                                                   \
    ([]() -> int {                                                                 \
        static const int v = getenv(name) ? atoi(getenv(name)) : (dflt);           \
        return ::stmi::env_live() ? (getenv(name) ? atoi(getenv(name)) : (dflt)) : v; \
    }())

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

// TensorFlow 'SAME' rule: out = ceil(in/s); pad_before = max((out-1)s+k-in,0)/2.
static inline void same_pads(int in, int k, int s, int* out, int* before) {
    int o = (in + s - 1) / s;
    int total = (o - 1) * s + k - in;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

// Geometry of a packed (K-contiguous, zero-padded) weight set.
//   mode STMI_PACK_FWD  : rows = co, K = ci      mode STMI_PACK_DGRAD: rows = ci, K = co
// With fewer than 8 K-channels a tap is a whole kernel row (kw * channels elements).
struct PackGeom {
    int ntaps, rows, rows_pad, klen, kpad;
    bool row;
};
PackGeom pack_geom(int kh, int kw, int ci, int co, int mode);
int packed_rows(int rows);

}  // namespace stmi
