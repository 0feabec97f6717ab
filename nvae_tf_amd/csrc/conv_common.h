// Shared device helpers of the MFMA convolution kernels (conv_gemm.hip, conv_wgrad.hip).
#pragma once
#include "common.h"

template <typename T> struct Tr;
template <> struct Tr<bf16> { static constexpr int VE = 8; };
template <> struct Tr<f16> { static constexpr int VE = 8; };
template <> struct Tr<float> { static constexpr int VE = 4; };

// bijective XCD remap (cdna guide T1): consecutive logical tiles land on one XCD
__device__ __forceinline__ int xcd_remap(int id, int n) {
    int q = n >> 3, r = n & 7, x = id & 7, l = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

// one 16x16x32 MFMA on two raw 16-B operand chunks of 16-bit type T (bf16 or f16)
template <typename T>
__device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 acc) {
    if constexpr (sizeof(T) == 2 && !__is_same(T, bf16))
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
}

// the same MFMA with the accumulator pinned in AGPRs, in place (hand-scheduled loops; no hazard bookkeeping by the
// compiler: the caller keeps VALU reads of the accumulator >= 19 wait states behind the last of these)
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <typename T>
__device__ __forceinline__ void mfma16_agpr(const i32x4& a, const i32x4& b, f32x4& acc) {
    if constexpr (__is_same(T, bf16)) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

template <typename T>
__device__ __forceinline__ void mfma_step(const uint4& a, const uint4& b, f32x4& acc) {
    if constexpr (sizeof(T) == 2) {
        acc = mfma16<T>(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc);
    } else {
        // the lane's 16-B chunk holds 4 consecutive k; A and B use the same k permutation
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
}

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BKC> __device__ __forceinline__ int swz_row(int row) {
    return BKC == 8 ? ((row >> 1) & 7) : (row & 15);
}


// zero page for padded / out-of-range DMA lanes (one copy per translation unit)
static __device__ uint4 g_zero16[1];
static inline const uint4* zero_page() {
    uint4* z = nullptr;
    (void)hipGetSymbolAddress((void**)&z, HIP_SYMBOL(g_zero16));
    return z;
}

static int check_geom_mfma(const char* who, const NvaeConvGeom* g) {
    NVAE_REQUIRE(g, "%s: NULL geometry", who);
    NVAE_REQUIRE(g->B > 0 && g->Hin > 0 && g->Win > 0 && g->Cin > 0 && g->Hout > 0 && g->Wout > 0 && g->Cout > 0,
                 "%s: non-positive dimension", who);
    NVAE_REQUIRE(g->KH > 0 && g->KW > 0 && g->KH <= 7 && g->KW <= 7 && g->stride >= 1 && g->div >= 1,
                 "%s: bad kernel/stride/div", who);
    NVAE_REQUIRE(g->in_ld >= g->Cin && g->out_ld >= g->Cout, "%s: leading dimensions too small", who);
    NVAE_REQUIRE((long)g->B * g->Hout * g->Wout < (1L << 23) && (long)g->B * g->Hin * g->Win < (1L << 23),
                 "%s: more than 2^23 pixels per call unsupported", who);
    return NVAE_OK;
}

