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

// ---------------------------------------------------------------------------------------
// "Thin" weight gradients: the three shapes above have K*N of a few hundred to a few thousand and
// M = B*H*W of 2k-131k, so the work is a skinny reduction over pixels.  One thread keeps KT x 8
// accumulators: out[j][c..c+7] += a_j(m) * v(m)[c..c+7] with a 16-B vector operand v and scalar a_j.
//   VARIANT 0 (stem, latent half of the combiner): v = dy[m, n..n+7], a_j = x[pix(m, tap_j), ci_j]
//   VARIANT 1 (1-channel logit head):              v = x[pix(m, tap_j), c..c+7], a = dy[m, 0]
// Lanes over 8-channel groups (coalesced), row-lanes over pixels, wave shuffles + one LDS hop, then
// f32 atomics with at most `S` (<= 32) adders per address.
// ---------------------------------------------------------------------------------------
#define THIN_KT 10
template <typename T, int VARIANT>
__global__ __launch_bounds__(256) void k_conv_thin_wgrad(NvaeConvGeom g, const T* __restrict__ x,
                                                         const T* __restrict__ dy, float* dw, int dw_ld,
                                                         float* db, int J, int rows_per_block) {
    const int CV = VARIANT == 0 ? g.Cout : g.Cin;           // vectorised channel count
    const int TGS = CV >= 64 ? 8 : (CV > 16 ? (CV > 32 ? 8 : 4) : (CV > 8 ? 2 : 1));
    const int RL = 256 / TGS;
    const int tg = threadIdx.x % TGS, rl = threadIdx.x / TGS;
    const int c0 = blockIdx.x * 64 + tg * 8;
    const bool cval = c0 < CV;
    const int j0 = blockIdx.z * THIN_KT;
    const long M = (long)g.B * g.Hout * g.Wout;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    int jkh[THIN_KT], jkw[THIN_KT], jci[THIN_KT];
#pragma unroll
    for (int j = 0; j < THIN_KT; ++j) {
        int jj = j0 + j;
        int tap = VARIANT == 0 ? jj / g.Cin : jj;
        jci[j] = VARIANT == 0 ? jj - tap * g.Cin : 0;
        jkh[j] = tap / g.KW;
        jkw[j] = tap - jkh[j] * g.KW;
    }
    float acc[THIN_KT][8];
    float accb[8];
#pragma unroll
    for (int j = 0; j < THIN_KT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) accb[e] = 0.f;
    const bool do_bias = db != nullptr && blockIdx.z == 0;
    if (cval) {
        for (long m = r0 + rl; m < r1; m += RL) {
            int wo = (int)(m % g.Wout);
            long q = m / g.Wout;
            int ho = (int)(q % g.Hout);
            long b = q / g.Hout;
            if (VARIANT == 0) {
                float v[8];
                V8<T>::ld(dy + m * g.out_ld + c0, v);
                if (do_bias) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) accb[e] += v[e];
                }
#pragma unroll
                for (int j = 0; j < THIN_KT; ++j) {
                    if (j0 + j >= J) continue;
                    int hs, ws;
                    if (!src_coord(ho, jkh[j], g.stride, g.pad_t, g.div, g.exact, g.Hin, hs)) continue;
                    if (!src_coord(wo, jkw[j], g.stride, g.pad_l, g.div, g.exact, g.Win, ws)) continue;
                    float a = ldf<T>(x + ((b * g.Hin + hs) * (long)g.Win + ws) * g.in_ld + jci[j]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[j][e] += a * v[e];
                }
            } else {
                float a = ldf<T>(dy + m * g.out_ld);
                if (do_bias && tg == 0 && blockIdx.x == 0) accb[0] += a;
#pragma unroll
                for (int j = 0; j < THIN_KT; ++j) {
                    if (j0 + j >= J) continue;
                    int hs, ws;
                    if (!src_coord(ho, jkh[j], g.stride, g.pad_t, g.div, g.exact, g.Hin, hs)) continue;
                    if (!src_coord(wo, jkw[j], g.stride, g.pad_l, g.div, g.exact, g.Win, ws)) continue;
                    float v[8];
                    V8<T>::ld(x + ((b * g.Hin + hs) * (long)g.Win + ws) * g.in_ld + c0, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[j][e] += a * v[e];
                }
            }
        }
    }
    // reduce over row lanes: shuffles inside the wave, then LDS across the 4 waves
    __shared__ float sm[4][8][(THIN_KT + 1) * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j <= THIN_KT; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = j < THIN_KT ? acc[j][e] : accb[e];
            for (int o = 32; o >= TGS; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane < TGS) sm[wave][lane][j * 8 + e] = v;
        }
    __syncthreads();
    if (threadIdx.x < TGS && cval) {
#pragma unroll
        for (int j = 0; j <= THIN_KT; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = sm[0][tg][j * 8 + e] + sm[1][tg][j * 8 + e] + sm[2][tg][j * 8 + e] + sm[3][tg][j * 8 + e];
                if (j == THIN_KT) {
                    if (!do_bias) continue;
                    if (VARIANT == 0) atomicAdd(db + c0 + e, v);
                    else if (e == 0 && tg == 0 && blockIdx.x == 0) atomicAdd(db, v);
                } else if (j0 + j < J) {
                    if (VARIANT == 0) atomicAdd(dw + (long)(j0 + j) * dw_ld + c0 + e, v);
                    else atomicAdd(dw + ((long)(j0 + j) * g.Cin + c0 + e) * dw_ld, v);
                }
            }
    }
}

template <typename T>
static bool launch_thin_wgrad(const NvaeConvGeom* g, const void* x, const void* dy, float* dw, int dw_ld,
                              float* db, hipStream_t s) {
    constexpr int ve = sizeof(T) == 2 ? 8 : 4;
    const long M = (long)g->B * g->Hout * g->Wout;
    int variant, J, CV;
    if (g->Cout % 8 == 0 && g->out_ld % ve == 0 && aligned16(dy) && g->KH * g->KW * g->Cin <= 64) {
        variant = 0; J = g->KH * g->KW * g->Cin; CV = g->Cout;
    } else if (g->Cout == 1 && g->Cin % 8 == 0 && g->in_ld % ve == 0 && aligned16(x)) {
        variant = 1; J = g->KH * g->KW; CV = g->Cin;
    } else {
        return false;
    }
    const int tgs = CV >= 64 ? 8 : (CV > 16 ? (CV > 32 ? 8 : 4) : (CV > 8 ? 2 : 1));
    const int rl = 256 / tgs;
    const int strips = (CV + 63) / 64, kch = (J + THIN_KT - 1) / THIN_KT;
    long S = 512 / (strips * kch);
    if (S > 32) S = 32;      // fewer splits are slower (121 vs 55 us at S = 8): the kernel is latency-, not atomics-bound
    long max_s = M / (rl * 2L);
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    long rpb = (M + S - 1) / S;
    S = (M + rpb - 1) / rpb;
    dim3 grid(strips, (unsigned)S, kch);
    if (variant == 0)
        hipLaunchKernelGGL((k_conv_thin_wgrad<T, 0>), grid, 256, 0, s, *g, (const T*)x, (const T*)dy, dw, dw_ld, db, J, (int)rpb);
    else
        hipLaunchKernelGGL((k_conv_thin_wgrad<T, 1>), grid, 256, 0, s, *g, (const T*)x, (const T*)dy, dw, dw_ld, db, J, (int)rpb);
    return true;
}

extern "C" int nvae_conv_direct_wgrad(int dtype, const NvaeConvGeom* g, const void* x, const void* dy,
                                      float* dw, int dw_ld, float* db, void* stream) {
    if (int e = check_geom("conv_direct_wgrad", g)) return e;
    NVAE_REQUIRE(x && dy && dw && dw_ld >= g->Cout, "conv_direct_wgrad: bad args");
    int K = g->KH * g->KW * g->Cin;
    long blocks = (long)(K + (db ? 1 : 0)) * g->Cout;
    NVAE_REQUIRE(blocks < (1L << 31), "conv_direct_wgrad: too many weights");
    {
        bool done = false;
        DISPATCH_T(dtype, done = launch_thin_wgrad<T>(g, x, dy, dw, dw_ld, db, (hipStream_t)stream);)
        if (done) {
            NVAE_LAUNCH_CHECK("conv_thin_wgrad");
            return NVAE_OK;
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_conv_direct_wgrad<T>), (int)blocks, 256, 0, (hipStream_t)stream, *g, (const T*)x, (const T*)dy, dw, dw_ld, db, K);)
    NVAE_LAUNCH_CHECK("conv_direct_wgrad");
    return NVAE_OK;
}
