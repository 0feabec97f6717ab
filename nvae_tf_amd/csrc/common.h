// Shared device helpers for the NVAE gfx950 kernels.  CDNA4 only: wave64, MFMA, 160 KB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/nvae_hip.h"

typedef __bf16 bf16;
typedef _Float16 f16;            // the third activation dtype (NVAE_F16): same kernels, v_mfma_f32_16x16x32_f16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

extern thread_local char g_nvae_err[512];

#define NVAE_FAIL(code, ...)                                   \
    do {                                                       \
        snprintf(g_nvae_err, sizeof(g_nvae_err), __VA_ARGS__); \
        return (code);                                         \
    } while (0)

#define NVAE_REQUIRE(cond, ...)                                \
    do {                                                       \
        if (!(cond)) NVAE_FAIL(NVAE_EINVAL, __VA_ARGS__);      \
    } while (0)

#define NVAE_LAUNCH_CHECK(name)                                                            \
    do {                                                                                   \
        hipError_t e_ = hipGetLastError();                                                 \
        if (e_ != hipSuccess) NVAE_FAIL(NVAE_ELAUNCH, "%s: %s", name, hipGetErrorString(e_)); \
    } while (0)

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// NVAE_DETERMINISTIC (nvae_set_deterministic): every sum across workgroups then has ONE adder per address or a fixed order, so
// a step's results do not depend on how the hardware schedules workgroups.  The launchers pick their slow-but-ordered
// configurations: one statistics-slab row per producing workgroup, pixel splits of the weight-gradient kernels combined
// through slabs or not split at all, no halo weight-gradient kernel (its pixel splits meet in f32 atomics).
extern int g_nvae_det;
static inline bool is16(int dtype) { return dtype == NVAE_BF16 || dtype == NVAE_F16; }     // 16-bit activation types
template <typename T> static constexpr int dtype_of() { return sizeof(T) == 4 ? NVAE_F32 : (__is_same(T, __bf16) ? NVAE_BF16 : NVAE_F16); }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
// rows of a statistics slab that `producers` workgroups add into: <= 64 adders per address, or one each (deterministic)
static inline int slab_rows_for(long producers) { return g_nvae_det ? (int)(producers > 0 ? producers : 1) : cdiv(producers, 64); }

// ---------------------------------------------------------------------------------------
// 8-element vector access: one 16-B access for bf16, two for f32.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned short f2bf(float f) {
    bf16 b = (bf16)f;   // v_cvt_pk_bf16_f32: RNE, NaN-preserving
    return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct V8;
template <> struct V8<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[8]) {
        float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
        ((float4*)p)[0] = make_float4(v[0], v[1], v[2], v[3]);
        ((float4*)p)[1] = make_float4(v[4], v[5], v[6], v[7]);
    }
};
template <> struct V8<bf16> {
    static __device__ __forceinline__ void ld(const bf16* p, float (&v)[8]) {
        uint4 r = *(const uint4*)p;
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
        v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
        v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16* p, const float (&v)[8]) {
        uint4 r;
        r.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        r.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        r.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
        r.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
        *(uint4*)p = r;
    }
};

template <> struct V8<f16> {
    static __device__ __forceinline__ void ld(const f16* p, float (&v)[8]) {
        const f16x8 r = *(const f16x8*)p;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)r[j];          // v_cvt_f32_f16
    }
    static __device__ __forceinline__ void st(f16* p, const float (&v)[8]) {
        f16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (f16)v[j];            // v_cvt_pk_f16_f32 pairs: RNE, saturating to inf
        *(f16x8*)p = r;
    }
};

// 16-B chunk of 8 sixteen-bit elements <-> 8 floats (the conv operand prologue works on raw uint4 chunks)
template <typename T> __device__ __forceinline__ void unpack8(uint4 raw, float (&v)[8]);
template <> __device__ __forceinline__ void unpack8<bf16>(uint4 raw, float (&v)[8]) {
    v[0] = __uint_as_float(raw.x << 16); v[1] = __uint_as_float(raw.x & 0xffff0000u);
    v[2] = __uint_as_float(raw.y << 16); v[3] = __uint_as_float(raw.y & 0xffff0000u);
    v[4] = __uint_as_float(raw.z << 16); v[5] = __uint_as_float(raw.z & 0xffff0000u);
    v[6] = __uint_as_float(raw.w << 16); v[7] = __uint_as_float(raw.w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack8<f16>(uint4 raw, float (&v)[8]) {
    const f16x8 r = __builtin_bit_cast(f16x8, raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)r[j];
}
template <typename T> __device__ __forceinline__ uint4 pack8(const float (&v)[8]);
template <> __device__ __forceinline__ uint4 pack8<bf16>(const float (&v)[8]) {
    uint4 r;
    r.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
    r.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
    r.z = (unsigned)f2bf(v[4]) | ((unsigned)f2bf(v[5]) << 16);
    r.w = (unsigned)f2bf(v[6]) | ((unsigned)f2bf(v[7]) << 16);
    return r;
}
template <> __device__ __forceinline__ uint4 pack8<f16>(const float (&v)[8]) {
    f16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (f16)v[j];
    return __builtin_bit_cast(uint4, r);
}

template <typename T> __device__ __forceinline__ float ldf(const T* p);
template <> __device__ __forceinline__ float ldf<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ldf<bf16>(const bf16* p) { return (float)*p; }
template <> __device__ __forceinline__ float ldf<f16>(const f16* p) { return (float)*p; }
template <typename T> __device__ __forceinline__ void stf(T* p, float v);
template <> __device__ __forceinline__ void stf<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void stf<bf16>(bf16* p, float v) { *p = (bf16)v; }
template <> __device__ __forceinline__ void stf<f16>(f16* p, float v) { *p = (f16)v; }

// ---------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------
// v_rcp_f32 (1 ulp) instead of an IEEE division: hipcc expands `1.0f / d` into 11 VALU instructions (two v_div_scale, v_rcp,
// four v_fma, v_mul, v_div_fmas, v_div_fixup), which made every Swish 14 instructions instead of 4 - measurable where
// the activation is applied redundantly (conv operand prologue, depthwise LDS prologue, BatchNorm-backward epilogue)
__device__ __forceinline__ float sigmoidf_(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float swishf_(float x) { return x * sigmoidf_(x); }
__device__ __forceinline__ float dswishf_(float x) {
    float s = sigmoidf_(x);
    return s * (1.0f + x * (1.0f - s));
}
__device__ __forceinline__ float eluf_(float x) { return x > 0.f ? x : expm1f(x); }
__device__ __forceinline__ float deluf_(float x) { return x > 0.f ? 1.f : __expf(x); }
__device__ __forceinline__ float softplusf_(float x) {
    return fmaxf(x, 0.f) + log1pf(__expf(-fabsf(x)));
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block (256 threads) sum via LDS; result valid in every thread
__device__ __forceinline__ float block_sum256(float v, float* sm /* >= 4 floats */) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

// exact unsigned division by a runtime constant: q = (n * M) >> 40, valid for n < 2^40 / d
struct FastDiv {
    unsigned long long M;
    unsigned d;
};
static inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f;
    f.d = d;
    f.M = ((1ull << 40) / d) + 1;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) {
    return (unsigned)(((unsigned long long)n * f.M) >> 40);
}

// loss scale of the backward seeds (include/nvae_hip.h, hyper layout): 0 or no buffer = 1
__device__ __forceinline__ float loss_scale_of(const float* hyper) {
    if (!hyper) return 1.f;
    const float s = hyper[NVAE_HY_LSCALE];
    return s != 0.f ? s : 1.f;
}

// Activation codes shared with the host
#define ACT_NONE 0
#define ACT_SWISH 1
#define ACT_ELU 2

#define DISPATCH_T(dtype, ...)                                        \
    if ((dtype) == NVAE_F32) { typedef float T; __VA_ARGS__ }         \
    else if ((dtype) == NVAE_BF16) { typedef bf16 T; __VA_ARGS__ }    \
    else if ((dtype) == NVAE_F16) { typedef f16 T; __VA_ARGS__ }      \
    else NVAE_FAIL(NVAE_EINVAL, "bad dtype %d", (int)(dtype));
