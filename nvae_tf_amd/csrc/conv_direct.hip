// Scalar fallback convolutions for the handful of shapes the MFMA implicit GEMM does not take:
// the stem (Cin = 1), the latent half of DecoderSampleCombiner (Cin = 20) and the 1-channel logit
// head (Cout = 1).  Together < 0.1 % of the path's MACs; written for generality, not speed.
#include "common.h"

__device__ __forceinline__ bool src_coord(int o, int k, int stride, int pad, int div, int exact,
                                          int in_size, int& s) {
    int c = o * stride - pad + k;
    if (c < 0 || c >= in_size * div) return false;
    if (div == 1) { s = c; return true; }
    s = c / div;
    return !exact || (c - s * div) == 0;
}

template <typename T>
__global__ void k_conv_direct(NvaeConvGeom g, const T* __restrict__ src, const float* __restrict__ w,
                              long ws_tap, long ws_c, long ws_n, int flip,
                              const float* __restrict__ bias, const T* residual, void* out, int out_f32,
                              long total) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
        int n = (int)(i % g.Cout);
        long m = i / g.Cout;
        int wo = (int)(m % g.Wout);
        long q = m / g.Wout;
        int ho = (int)(q % g.Hout);
        long b = q / g.Hout;
        float acc = bias ? bias[n] : 0.f;
        for (int kh = 0; kh < g.KH; ++kh) {
            int hs;
            if (!src_coord(ho, kh, g.stride, g.pad_t, g.div, g.exact, g.Hin, hs)) continue;
            for (int kw = 0; kw < g.KW; ++kw) {
                int ws;
                if (!src_coord(wo, kw, g.stride, g.pad_l, g.div, g.exact, g.Win, ws)) continue;
                int tap = flip ? (g.KH - 1 - kh) * g.KW + (g.KW - 1 - kw) : kh * g.KW + kw;
                const T* sp = src + ((b * g.Hin + hs) * (long)g.Win + ws) * g.in_ld;
                const float* wp = w + tap * ws_tap + n * ws_n;
                for (int c = 0; c < g.Cin; ++c) acc += ldf<T>(sp + c) * wp[c * ws_c];
            }
        }
        if (residual) acc += ldf<T>(residual + m * g.res_ld + n);
        if (out_f32) ((float*)out)[m * g.out_ld + n] = acc;
        else stf<T>((T*)out + m * g.out_ld + n, acc);
    }
}

static int check_geom(const char* who, const NvaeConvGeom* g) {
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

extern "C" int nvae_conv_direct(int dtype, const NvaeConvGeom* g, const void* src, const float* w,
                                long ws_tap, long ws_c, long ws_n, int flip, const float* bias,
                                const void* residual, void* out, int out_f32, void* stream) {
    if (int e = check_geom("conv_direct", g)) return e;
    NVAE_REQUIRE(src && w && out, "conv_direct: NULL pointer");
    NVAE_REQUIRE(!residual || g->res_ld >= g->Cout, "conv_direct: res_ld too small");
    long total = (long)g->B * g->Hout * g->Wout * g->Cout;
    long gr = (total + 255) / 256;
    if (gr > 8192) gr = 8192;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_conv_direct<T>), (int)gr, 256, 0, (hipStream_t)stream, *g, (const T*)src, w, ws_tap, ws_c, ws_n, flip, bias, (const T*)residual, out, out_f32, total);)
    NVAE_LAUNCH_CHECK("conv_direct");
    return NVAE_OK;
}

// One block per (k, n) weight element (k == K means the bias row).
template <typename T>
__global__ void k_conv_direct_wgrad(NvaeConvGeom g, const T* __restrict__ x, const T* __restrict__ dy,
                                    float* dw, int dw_ld, float* db, int K) {
    __shared__ float sm[4];
    const int n = blockIdx.x % g.Cout;
    const int k = blockIdx.x / g.Cout;
    const bool is_bias = (k == K);
    int tap = 0, c = 0, kh = 0, kw = 0;
    if (!is_bias) {
        tap = k / g.Cin; c = k - tap * g.Cin;
        kh = tap / g.KW; kw = tap - kh * g.KW;
    }
    const long M = (long)g.B * g.Hout * g.Wout;
    float acc = 0.f;
    for (long m = threadIdx.x; m < M; m += 256) {
        float a = 1.f;
        if (!is_bias) {
            int wo = (int)(m % g.Wout);
            long q = m / g.Wout;
            int ho = (int)(q % g.Hout);
            long b = q / g.Hout;
            int hs, ws;
            if (!src_coord(ho, kh, g.stride, g.pad_t, g.div, g.exact, g.Hin, hs)) continue;
            if (!src_coord(wo, kw, g.stride, g.pad_l, g.div, g.exact, g.Win, ws)) continue;
            a = ldf<T>(x + ((b * g.Hin + hs) * (long)g.Win + ws) * g.in_ld + c);
        }
        acc += a * ldf<T>(dy + m * g.out_ld + n);
    }
    acc = block_sum256(acc, sm);
    if (threadIdx.x == 0) {
        if (is_bias) atomicAdd(db + n, acc);
        else atomicAdd(dw + (long)k * dw_ld + n, acc);
    }
}

extern "C" int nvae_conv_direct_wgrad(int dtype, const NvaeConvGeom* g, const void* x, const void* dy,
                                      float* dw, int dw_ld, float* db, void* stream) {
    if (int e = check_geom("conv_direct_wgrad", g)) return e;
    NVAE_REQUIRE(x && dy && dw && dw_ld >= g->Cout, "conv_direct_wgrad: bad args");
    int K = g->KH * g->KW * g->Cin;
    long blocks = (long)(K + (db ? 1 : 0)) * g->Cout;
    NVAE_REQUIRE(blocks < (1L << 31), "conv_direct_wgrad: too many weights");
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_conv_direct_wgrad<T>), (int)blocks, 256, 0, (hipStream_t)stream, *g, (const T*)x, (const T*)dy, dw, dw_ld, db, K);)
    NVAE_LAUNCH_CHECK("conv_direct_wgrad");
    return NVAE_OK;
}
