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

// 1x1 convolutions with at most 32 input channels (forward of the latent half of DecoderSampleCombiner,
// decoder.py:110-117, accumulated onto the other half's output): a workgroup stages 64 rows of x and the
// [Cin][64] weight strip in LDS, thread (4 rows, 4 channels) does 16 FMAs per 2 LDS reads.
template <typename T>
__global__ __launch_bounds__(256) void k_conv_smallk_fwd(const T* __restrict__ src, int in_ld, int Cin,
                                                         const float* __restrict__ w, long ws_c, long ws_n,
                                                         const float* __restrict__ bias, const T* residual,
                                                         int res_ld, void* out, int out_ld, int out_f32, int Cout,
                                                         long M) {
    __shared__ float sx[64][33];
    __shared__ __attribute__((aligned(16))) float sw[32][64];
    const int n_base = blockIdx.x * 64;
    const long m_base = (long)blockIdx.y * 64;
    for (int q = threadIdx.x; q < 64 * 32; q += 256) {
        const int r = q >> 5, k = q & 31;
        sx[r][k] = (m_base + r < M && k < Cin) ? ldf<T>(src + (m_base + r) * in_ld + k) : 0.f;
    }
    for (int q = threadIdx.x; q < 32 * 64; q += 256) {
        const int k = q >> 6, n = q & 63;
        sw[k][n] = (k < Cin && n_base + n < Cout) ? w[k * ws_c + (long)(n_base + n) * ws_n] : 0.f;
    }
    __syncthreads();
    const int n4 = threadIdx.x & 15, rg = threadIdx.x >> 4;
    float acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[r][e] = 0.f;
#pragma unroll 4
    for (int k = 0; k < 32; ++k) {
        if (k >= Cin) break;
        const float4 wv = *(const float4*)(&sw[k][n4 * 4]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xv = sx[rg * 4 + r][k];
            acc[r][0] += xv * wv.x; acc[r][1] += xv * wv.y; acc[r][2] += xv * wv.z; acc[r][3] += xv * wv.w;
        }
    }
    const int n = n_base + n4 * 4;
    if (n >= Cout) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long m = m_base + rg * 4 + r;
        if (m >= M) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[r][e] + (bias ? bias[n + e] : 0.f);
            if (residual) v += ldf<T>(residual + m * res_ld + n + e);
            if (out_f32) ((float*)out)[m * out_ld + n + e] = v;
            else stf<T>((T*)out + m * out_ld + n + e, v);
        }
    }
}

extern "C" int nvae_conv_direct(int dtype, const NvaeConvGeom* g, const void* src, const float* w,
                                long ws_tap, long ws_c, long ws_n, int flip, const float* bias,
                                const void* residual, void* out, int out_f32, void* stream) {
    if (int e = check_geom("conv_direct", g)) return e;
    NVAE_REQUIRE(src && w && out, "conv_direct: NULL pointer");
    NVAE_REQUIRE(!residual || g->res_ld >= g->Cout, "conv_direct: res_ld too small");
    long total = (long)g->B * g->Hout * g->Wout * g->Cout;
    if (g->KH == 1 && g->KW == 1 && g->stride == 1 && g->div == 1 && g->Cin <= 32 && g->Cout % 4 == 0 &&
        g->Hin == g->Hout && g->Win == g->Wout) {
        const long M = (long)g->B * g->Hout * g->Wout;
        dim3 grid(cdiv(g->Cout, 64), (unsigned)cdiv(M, 64));
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_conv_smallk_fwd<T>), grid, 256, 0, (hipStream_t)stream, (const T*)src, g->in_ld, g->Cin, w, ws_c, ws_n, bias, (const T*)residual, g->res_ld, out, g->out_ld, out_f32, g->Cout, M);)
        NVAE_LAUNCH_CHECK("conv_smallk_fwd");
        return NVAE_OK;
    }
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

// ---------------------------------------------------------------------------------------
// 1x1 convolutions with at most 32 input channels (the 20 latent channels of DecoderSampleCombiner,
// decoder.py:110-117): dw[k, n] += sum_m x[m, k] * dy[m, n].  The generic thin kernel above spends its
// time in 2-B gathers and an 88-value shuffle tree per thread; here a workgroup stages 64 rows of x
// (all K channels) and of dy (a 64-channel strip) in LDS once and thread (kq, n4) accumulates
// dw[kq (+16), 4 channels] over the rows: 2-3 LDS reads per 4-8 FMAs, no cross-lane reduction at all.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_conv_smallk_wgrad(const T* __restrict__ x, int in_ld, int Cin,
                                                           const T* __restrict__ dy, int out_ld, int Cout,
                                                           float* dw, int dw_ld, float* db, long M,
                                                           int rows_per_block) {
    constexpr int RT = 64;
    __shared__ float sx[RT][33];                       // f32, padded: (row, k) reads are conflict-free
    __shared__ __attribute__((aligned(16))) T sd[RT][64];
    const int n_base = blockIdx.x * 64;
    const int n4 = threadIdx.x & 15, kq = threadIdx.x >> 4;      // 4 channels, k = kq and kq + 16
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > M) r1 = M;
    float a0[4] = {0.f, 0.f, 0.f, 0.f}, a1[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    const bool k1 = kq + 16 < Cin;
    constexpr int VE = 16 / (int)sizeof(T);
    for (long t0 = r0; t0 < r1; t0 += RT) {
        __syncthreads();
        for (int q = threadIdx.x; q < RT * 32; q += 256) {       // x tile, zero-padded to 32 channels
            const int r = q >> 5, k = q & 31;
            sx[r][k] = (t0 + r < r1 && k < Cin) ? ldf<T>(x + (t0 + r) * in_ld + k) : 0.f;
        }
        for (int q = threadIdx.x; q < RT * (64 / VE); q += 256) {
            const int r = q / (64 / VE), cc = (q - r * (64 / VE)) * VE;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (t0 + r < r1 && n_base + cc < Cout) v = *(const uint4*)(dy + (t0 + r) * out_ld + n_base + cc);
            *(uint4*)(&sd[r][cc]) = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < RT; ++r) {
            float d[4];
            if constexpr (sizeof(T) == 2) {
                const uint2 u = *(const uint2*)(&sd[r][n4 * 4]);
                float d8[8];
                unpack8<T>(make_uint4(u.x, u.y, 0u, 0u), d8);
                d[0] = d8[0]; d[1] = d8[1]; d[2] = d8[2]; d[3] = d8[3];
            } else {
                const float4 u = *(const float4*)(&sd[r][n4 * 4]);
                d[0] = u.x; d[1] = u.y; d[2] = u.z; d[3] = u.w;
            }
            const float x0 = sx[r][kq], x1 = sx[r][kq + 16];
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[e] += x0 * d[e]; a1[e] += x1 * d[e]; ab[e] += d[e]; }
        }
    }
    const int n = n_base + n4 * 4;
    if (n >= Cout) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (kq < Cin) atomicAdd(dw + (long)kq * dw_ld + n + e, a0[e]);
        if (k1) atomicAdd(dw + (long)(kq + 16) * dw_ld + n + e, a1[e]);
        if (db && kq == 0) atomicAdd(db + n + e, ab[e]);
    }
}

// ---------------------------------------------------------------------------------------
// 3x3 'same' stride-1 convolutions with ONE channel on one side and <= 32 on the other: the stem
// (Cin = 1, preprocess.py:19-23) and the logit head (Cout = 1, postprocess.py:27-30) of the MNIST model.
// Both weight gradients are the same correlation
//     out[t, c] += sum_pix a[pix + off(t)] * v[pix, c]        (a: 1-channel image, v: C-channel image)
// stem: a = x, v = dy, t = tap;  head: a = dy, v = x, and the result for offset -off(t) is tap t (flip).
// A persistent workgroup walks 16x16 tiles: the 18x18 halo of a (f32) and the 256 x 32 tile of v sit in
// LDS, thread (pixel lane, 4 channels) keeps 9 x 4 accumulators over 8 pixels per tile and all tiles.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_conv_1ch_wgrad(const T* __restrict__ a, int a_ld, const T* __restrict__ v,
                                                        int v_ld, int C, float* out, long out_t, long out_c,
                                                        int flip, float* db, int db_from_a, int B, int H, int W) {
    __shared__ float sa[18][19];
    __shared__ __attribute__((aligned(16))) T sv[256][32];
    __shared__ float red[4][8][41];
    const int cg = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    const long units = (long)B * tiles_x * tiles_y;
    float acc[9][4], sumv[4] = {0.f, 0.f, 0.f, 0.f}, suma = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    constexpr int VE = 16 / (int)sizeof(T);
    for (long u = blockIdx.x; u < units; u += gridDim.x) {
        const long b = u / (tiles_x * tiles_y);
        const int tile = (int)(u - b * (tiles_x * tiles_y));
        const int y0 = (tile / tiles_x) * 16, x0 = (tile % tiles_x) * 16;
        __syncthreads();
        for (int q = threadIdx.x; q < 18 * 18; q += 256) {
            const int py = q / 18, px = q - py * 18;
            const int gy = y0 - 1 + py, gx = x0 - 1 + px;
            sa[py][px] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? ldf<T>(a + ((b * H + gy) * (long)W + gx) * a_ld) : 0.f;
        }
        for (int q = threadIdx.x; q < 256 * (32 / VE); q += 256) {
            const int pix = q / (32 / VE), cc = (q - pix * (32 / VE)) * VE;
            const int gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
            uint4 w = make_uint4(0, 0, 0, 0);
            if (gy < H && gx < W && cc < C) w = *(const uint4*)(v + ((b * H + gy) * (long)W + gx) * v_ld + cc);
            *(uint4*)(&sv[pix][cc]) = w;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int pix = pl + 32 * i, py = pix >> 4, px = pix & 15;
            float d[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) d[e] = (float)sv[pix][cg * 4 + e];
#pragma unroll
            for (int e = 0; e < 4; ++e) sumv[e] += d[e];
            suma += sa[py + 1][px + 1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float av = sa[py + kh][px + kw];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[kh * 3 + kw][e] += av * d[e];
                }
        }
    }
    // reduce over the 32 pixel lanes: lanes with equal cg are 8 apart in a wave, then the 4 waves via LDS
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float vals[41];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) vals[t * 4 + e] = acc[t][e];
#pragma unroll
    for (int e = 0; e < 4; ++e) vals[36 + e] = sumv[e];
    vals[40] = suma;
#pragma unroll
    for (int j = 0; j < 41; ++j) {
        float x = vals[j];
        x += __shfl_xor(x, 8, 64); x += __shfl_xor(x, 16, 64); x += __shfl_xor(x, 32, 64);
        if (lane < 8) red[wave][lane][j] = x;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < 8 * 41; q += 256) {
        const int g8 = q / 41, j = q - g8 * 41;
        const float x = red[0][g8][j] + red[1][g8][j] + red[2][g8][j] + red[3][g8][j];
        if (j < 36) {
            const int t = j >> 2, c = g8 * 4 + (j & 3);
            if (c < C) atomicAdd(out + (long)(flip ? 8 - t : t) * out_t + (long)c * out_c, x);
        } else if (db) {
            if (!db_from_a && j < 40) { const int c = g8 * 4 + (j - 36); if (c < C) atomicAdd(db + c, x); }
            if (db_from_a && j == 40 && g8 == 0) atomicAdd(db, x);      // every cg lane summed the same a
        }
    }
}

template <typename T>
static bool launch_thin_wgrad(const NvaeConvGeom* g, const void* x, const void* dy, float* dw, int dw_ld,
                              float* db, hipStream_t s) {
    constexpr int ve = sizeof(T) == 2 ? 8 : 4;
    const long M = (long)g->B * g->Hout * g->Wout;
    if (g->KH == 1 && g->KW == 1 && g->stride == 1 && g->div == 1 && g->Cin <= 32 && g->Cout % 8 == 0 &&
        g->out_ld % ve == 0 && aligned16(dy) && g->Hin == g->Hout && g->Win == g->Wout) {
        const int strips = cdiv(g->Cout, 64);
        long S = 512 / strips;                 // ~512 workgroups, whole 64-row tiles each
        if (S > (M + 63) / 64) S = (M + 63) / 64;
        if (S < 1) S = 1;
        long rpb = ((M + S - 1) / S + 63) / 64 * 64;
        S = (M + rpb - 1) / rpb;
        hipLaunchKernelGGL((k_conv_smallk_wgrad<T>), dim3(strips, (unsigned)S), 256, 0, s, (const T*)x, g->in_ld, g->Cin,
                           (const T*)dy, g->out_ld, g->Cout, dw, dw_ld, db, M, (int)rpb);
        return true;
    }
    const bool same3 = g->KH == 3 && g->KW == 3 && g->stride == 1 && g->div == 1 && g->pad_t == 1 && g->pad_l == 1 &&
                       g->Hin == g->Hout && g->Win == g->Wout;
    if (same3 && g->Cin == 1 && g->Cout <= 32 && g->Cout % ve == 0 && g->out_ld % ve == 0 && aligned16(dy)) {
        // stem: a = x (1 channel), v = dy;  dw[tap][n], db[n] = sum dy
        hipLaunchKernelGGL((k_conv_1ch_wgrad<T>), 256, 256, 0, s, (const T*)x, g->in_ld, (const T*)dy, g->out_ld, g->Cout,
                           dw, (long)dw_ld, 1L, 0, db, 0, g->B, g->Hout, g->Wout);
        return true;
    }
    if (same3 && g->Cout == 1 && g->Cin <= 32 && g->Cin % ve == 0 && g->in_ld % ve == 0 && aligned16(x)) {
        // logit head: a = dy (1 channel), v = x;  dw[(tap, ci)][0] with the tap mirrored, db[0] = sum dy
        hipLaunchKernelGGL((k_conv_1ch_wgrad<T>), 256, 256, 0, s, (const T*)dy, g->out_ld, (const T*)x, g->in_ld, g->Cin,
                           dw, (long)g->Cin * dw_ld, (long)dw_ld, 1, db, 1, g->B, g->Hout, g->Wout);
        return true;
    }
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
        // (deterministic mode: the generic kernel below has one workgroup, i.e. one adder, per weight)
        if (!g_nvae_det) { DISPATCH_T(dtype, done = launch_thin_wgrad<T>(g, x, dy, dw, dw_ld, db, (hipStream_t)stream);) }
        if (done) {
            NVAE_LAUNCH_CHECK("conv_thin_wgrad");
            return NVAE_OK;
        }
    }
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_conv_direct_wgrad<T>), (int)blocks, 256, 0, (hipStream_t)stream, *g, (const T*)x, (const T*)dy, dw, dw_ld, db, K);)
    NVAE_LAUNCH_CHECK("conv_direct_wgrad");
    return NVAE_OK;
}
