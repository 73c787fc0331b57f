// Shared device/host helpers for libbbocr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

// round-to-nearest-even f32 -> bf16 bits (host); device code uses the hardware cvt via __bf16 casts
static inline uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32_host(uint16_t h) {
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short h) {
    return __uint_as_float(((unsigned int)h) << 16);
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
    __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 (RNE, NaN preserved)
    return __builtin_bit_cast(unsigned short, b);
}
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {   // one v_cvt_pk_bf16_f32 (RNE, same rounding as the scalar cast)
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_t));
}

// ---- element type of MFMA operands / stored activations: EL = 0 bf16 (default), EL = 1 IEEE fp16 (bbocr_config::precision >= 1).
// Both are 16-bit sign-magnitude formats with the same MFMA shape (v_mfma_f32_16x16x32_{bf16,f16}), so the kernels differ only in
// these conversions.  fp16 keeps 11 significand bits against bf16's 8 (8x smaller rounding steps) with a range of 6e-8 .. 65504.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;
static inline uint16_t f32_to_f16_host(float f) {          // round to nearest even, subnormals and overflow to inf handled
    const _Float16 h = (_Float16)f;
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
static inline float f16_to_f32_host(uint16_t u) {
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
}
static inline uint16_t f32_to_el_host(int el, float f) { return el ? f32_to_f16_host(f) : f32_to_bf16_host(f); }
static inline float el_to_f32_host(int el, uint16_t u) { return el ? f16_to_f32_host(u) : bf16_to_f32_host(u); }

template <int EL> struct El;
template <> struct El<0> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
    static __device__ __forceinline__ f32x2_t unpack2(unsigned int q) { return (f32x2_t){__uint_as_float(q << 16), __uint_as_float(q & 0xffff0000u)}; }
    static __device__ __forceinline__ unsigned short from_f32(float f) { return f32_to_bf16_bits(f); }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return bf16_bits_to_f32(h); }
};
template <> struct El<1> {
    typedef f16x8 v8;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned int pack2(float lo, float hi) {     // RNE (v_cvt_f16_f32 x2 / v_cvt_pk_f16_f32), never the RTZ pack
        const f32x2_t v = {lo, hi};
        return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2_t));
    }
    static __device__ __forceinline__ f32x2_t unpack2(unsigned int q) { return __builtin_convertvector(__builtin_bit_cast(f16x2_t, q), f32x2_t); }
    static __device__ __forceinline__ unsigned short from_f32(float f) { return __builtin_bit_cast(unsigned short, (_Float16)f); }
    static __device__ __forceinline__ float to_f32(unsigned short h) { return (float)__builtin_bit_cast(_Float16, h); }
};
// max of 16-bit sign-magnitude floats without knowing which: x ^ ((x >> 15) & 0x7fff) is monotone as int16 (and its own inverse)
__device__ __forceinline__ s16x8 sm16_key(s16x8 x) { return x ^ ((x >> 15) & (short)0x7fff); }

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous run of logical ids so
// neighbouring tiles (same activation patch / same weight panel) share one L2.  Bijective for any n.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7, i = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i;
}

// A/B switches of the measurement campaign (BBOCR_* environment variables) exist in diagnostic builds only (make DIAG=1): the shipped
// library always takes the measured-best default and does not read the environment.
#ifdef BBOCR_DIAG
#include <stdlib.h>
static inline int diag_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline constexpr int diag_knob(const char*, int dflt) { return dflt; }
#endif

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess) {                                                            \
            throw std::string(#expr) + ": " + hipGetErrorString(_e);                       \
        }                                                                                  \
    } while (0)

// ---- per-DEVICE launch state.  hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, device) and the CU count is the
// device's: a process may hold contexts on several cards (bbocr_config::device), so neither may live in a process-wide flag.
static constexpr int kMaxDev = 32;
struct LdsOptIn {
    std::atomic<size_t> hw[kMaxDev];   // per device: the largest dynamic LDS size this function has been opted in for (static storage: zero)
};
static inline hipError_t lds_opt_in(LdsOptIn& st, const void* fn, size_t smem) {
    int d = 0;
    hipError_t e = hipGetDevice(&d);
    if (e != hipSuccess) return e;
    if (d < 0 || d >= kMaxDev) return hipErrorInvalidDevice;
    size_t cur = st.hw[d].load(std::memory_order_acquire);
    if (smem <= cur) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);     // idempotent: a racing thread sets the same value
    if (e != hipSuccess) return e;
    while (cur < smem && !st.hw[d].compare_exchange_weak(cur, smem, std::memory_order_release)) {}
    return hipSuccess;
}
static inline int device_cus() {
    static std::atomic<int> n[kMaxDev];
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) return 256;
    int v = n[d].load(std::memory_order_acquire);
    if (v) return v;
    hipDeviceProp_t prop;
    v = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    n[d].store(v, std::memory_order_release);
    return v;
}
