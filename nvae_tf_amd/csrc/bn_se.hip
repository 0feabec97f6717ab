// BatchNorm(+Swish) and Squeeze-Excitation kernels.  HBM-bound.
//
// Reductions over the row axis of an NHWC tensor viewed as [rows, C] share one structure:
// a block owns a contiguous range of rows; thread (ty, tg) walks rows ty, ty+RPI, ... of that
// range and holds 8 consecutive channels (one 16-B bf16 load) in registers; partials are combined
// across ty through LDS and the block issues one f32 atomic per (channel, quantity).  Atomics are
// contiguous per wave (8 channels per lane), the shape the microarch guide prices at full rate.
#include "common.h"

#define RED_THREADS 256

// MODE 0: bn stats        q0 = x,          q1 = x*x           -> out[c], out[C + c]
// MODE 1: bn bwd          q0 = dpre,       q1 = dpre * xhat   -> out0[c] (dbeta), out1[c] (dgamma)
// MODE 2: per-image sum   q0 = x                              -> out[b*C + c]
// MODE 3: per-image sum   q0 = x*dy                           -> out[b*C + c]
// MODE 4: column sum with leading dimension ld, q0 = x        -> out[c]
template <typename T, int MODE>
__global__ void k_colreduce(const T* __restrict__ x, const T* __restrict__ dy, long rows_per_group,
                            int C, int ld, int rows_per_block, const float* __restrict__ scale,
                            const float* __restrict__ shift, const float* __restrict__ mean,
                            const float* __restrict__ invstd, int act, float* out0, float* out1) {
    constexpr int NQ = (MODE <= 1) ? 2 : 1;
    const int TG = C >> 3;                 // channel groups of 8
    const int RPI = RED_THREADS / TG;      // rows per iteration (>= 1)
    const int tg = threadIdx.x % TG, ty = threadIdx.x / TG;
    const long grp = blockIdx.y;           // image index for MODE 2/3, else 0
    const long r0 = (long)blockIdx.x * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows_per_group) r1 = rows_per_group;
    float acc[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    float sc[8], sh[8], mu[8], is[8];
    if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = tg * 8 + j;
            sc[j] = scale[c]; sh[j] = shift[c]; mu[j] = mean[c]; is[j] = invstd[c];
        }
    }
    if (ty < RPI) {
        for (long r = r0 + ty; r < r1; r += RPI) {
            const long off = (grp * rows_per_group + r) * (long)ld + tg * 8;
            float v[8];
            V8<T>::ld(x + off, v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[0][j] += v[j]; acc[1][j] += v[j] * v[j]; }
            } else if (MODE == 1) {
                float g[8];
                V8<T>::ld(dy + off, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float pre = v[j] * sc[j] + sh[j];
                    float dpre = (act == ACT_SWISH) ? g[j] * dswishf_(pre) : g[j];
                    acc[0][j] += dpre;
                    acc[1][j] += dpre * (v[j] - mu[j]) * is[j];
                }
            } else if (MODE == 3) {
                float g[8];
                V8<T>::ld(dy + off, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[0][j] += v[j] * g[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[0][j] += v[j];
            }
        }
    }
    __shared__ float sm[RED_THREADS * NQ * 8];
    float* mine = sm + threadIdx.x * (NQ * 8);
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) mine[q * 8 + j] = acc[q][j];
    __syncthreads();
    if (ty == 0) {
        for (int t = 1; t < RPI; ++t) {
            const float* o = sm + (t * TG + tg) * (NQ * 8);
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q][j] += o[q * 8 + j];
        }
        float* d0 = out0 + grp * C + tg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(d0 + j, acc[0][j]);
        if (NQ == 2) {
            float* d1 = out1 + tg * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j) atomicAdd(d1 + j, acc[1][j]);
        }
    }
}

template <typename T, int MODE>
static int launch_colreduce(const T* x, const T* dy, long groups, long rows_per_group, int C, int ld,
                            const float* scale, const float* shift, const float* mean,
                            const float* invstd, int act, float* out0, float* out1, hipStream_t s) {
    const int TG = C / 8;
    const int RPI = RED_THREADS / TG;
    // aim for ~1024 blocks in total, at least 4 iterations per block
    long want_blocks = 1024 / groups;
    if (want_blocks < 1) want_blocks = 1;
    long rpb = (rows_per_group + want_blocks - 1) / want_blocks;
    long min_rpb = (long)RPI * 4;
    if (rpb < min_rpb) rpb = min_rpb;
    int nblk = (int)((rows_per_group + rpb - 1) / rpb);
    dim3 grid(nblk, (unsigned)groups);
    hipLaunchKernelGGL((k_colreduce<T, MODE>), grid, RED_THREADS, 0, s, x, dy, rows_per_group, C, ld,
                       (int)rpb, scale, shift, mean, invstd, act, out0, out1);
    return 0;
}

static int check_c(const char* who, int C) {
    NVAE_REQUIRE(C >= 8 && C % 8 == 0 && C <= 2048, "%s: C=%d must be a multiple of 8 in [8, 2048]", who, C);
    return NVAE_OK;
}

extern "C" int nvae_bn_stats(int dtype, const void* x, long rows, int C, float* sums, void* stream) {
    if (int e = check_c("bn_stats", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x), "bn_stats: bad rows/alignment");
    DISPATCH_T(dtype, launch_colreduce<T, 0>((const T*)x, nullptr, 1, rows, C, C, nullptr, nullptr, nullptr, nullptr, 0, sums, sums + C, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("bn_stats");
    return NVAE_OK;
}

__global__ void k_bn_finalize(const float* __restrict__ sums, float inv_n, int C,
                              const float* __restrict__ gamma, const float* __restrict__ beta,
                              float* rm, float* rv, float momentum, float eps, float* scale,
                              float* shift, float* mean, float* invstd) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float m = sums[c] * inv_n;
    float var = fmaxf(sums[C + c] * inv_n - m * m, 0.f);
    float is = rsqrtf(var + eps);
    float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - m * sc;
    mean[c] = m;
    invstd[c] = is;
    rm[c] = rm[c] * momentum + m * (1.f - momentum);
    rv[c] = rv[c] * momentum + var * (1.f - momentum);
}

extern "C" int nvae_bn_finalize(const float* sums, long rows, int C, const float* gamma,
                                const float* beta, float* rm, float* rv, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* invstd, void* stream) {
    NVAE_REQUIRE(rows > 0 && C > 0, "bn_finalize: bad shape");
    hipLaunchKernelGGL(k_bn_finalize, cdiv(C, 256), 256, 0, (hipStream_t)stream, sums, 1.0f / (float)rows,
                       C, gamma, beta, rm, rv, momentum, eps, scale, shift, mean, invstd);
    NVAE_LAUNCH_CHECK("bn_finalize");
    return NVAE_OK;
}

__global__ void k_bn_eval_prepare(const float* gamma, const float* beta, const float* rm,
                                  const float* rv, int C, float eps, float* scale, float* shift) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float sc = gamma[c] * rsqrtf(rv[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
}

extern "C" int nvae_bn_eval_prepare(const float* gamma, const float* beta, const float* rm,
                                    const float* rv, int C, float eps, float* scale, float* shift,
                                    void* stream) {
    NVAE_REQUIRE(C > 0, "bn_eval_prepare: bad C");
    hipLaunchKernelGGL(k_bn_eval_prepare, cdiv(C, 256), 256, 0, (hipStream_t)stream, gamma, beta, rm, rv, C, eps, scale, shift);
    NVAE_LAUNCH_CHECK("bn_eval_prepare");
    return NVAE_OK;
}

static inline int ew_grid(long n8) {
    long g = (n8 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

template <typename T>
__global__ void k_bn_apply(const T* __restrict__ x, T* __restrict__ y, long n8, int C8,
                           const float* __restrict__ scale, const float* __restrict__ shift, int act) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        float v[8];
        V8<T>::ld(x + i * 8, v);
        float4 s0 = *(const float4*)(scale + c0), s1 = *(const float4*)(scale + c0 + 4);
        float4 h0 = *(const float4*)(shift + c0), h1 = *(const float4*)(shift + c0 + 4);
        float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float p = v[j] * sc[j] + sh[j];
            v[j] = (act == ACT_SWISH) ? swishf_(p) : p;
        }
        V8<T>::st(y + i * 8, v);
    }
}

extern "C" int nvae_bn_apply(int dtype, const void* x, void* y, long rows, int C, const float* scale,
                             const float* shift, int act, void* stream) {
    if (int e = check_c("bn_apply", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && aligned16(y) && aligned16(scale) && aligned16(shift), "bn_apply: bad rows/alignment");
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "bn_apply: act %d unsupported", act);
    long n8 = rows * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bn_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)x, (T*)y, n8, C / 8, scale, shift, act);)
    NVAE_LAUNCH_CHECK("bn_apply");
    return NVAE_OK;
}

extern "C" int nvae_bn_bwd_reduce(int dtype, const void* x, const void* dy, long rows, int C,
                                  const float* scale, const float* shift, const float* mean,
                                  const float* invstd, int act, float* dgamma, float* dbeta, void* stream) {
    if (int e = check_c("bn_bwd_reduce", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && aligned16(dy), "bn_bwd_reduce: bad rows/alignment");
    DISPATCH_T(dtype, launch_colreduce<T, 1>((const T*)x, (const T*)dy, 1, rows, C, C, scale, shift, mean, invstd, act, dbeta, dgamma, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("bn_bwd_reduce");
    return NVAE_OK;
}

// dx = scale * (dpre - mean(dpre) - xhat * mean(dpre * xhat)),  scale = gamma * invstd
template <typename T>
__global__ void k_bn_bwd_apply(const T* __restrict__ x, const T* __restrict__ dy, T* dx, long n8, int C8,
                               const float* __restrict__ scale, const float* __restrict__ shift,
                               const float* __restrict__ mean, const float* __restrict__ invstd, int act,
                               const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                               float inv_n, int acc) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        float v[8], g[8], o[8];
        V8<T>::ld(x + i * 8, v);
        V8<T>::ld(dy + i * 8, g);
        if (acc) V8<T>::ld(dx + i * 8, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = c0 + j;
            float sc = scale[c];
            float pre = v[j] * sc + shift[c];
            float dpre = (act == ACT_SWISH) ? g[j] * dswishf_(pre) : g[j];
            float xh = (v[j] - mean[c]) * invstd[c];
            float d = sc * (dpre - dbeta[c] * inv_n - xh * dgamma[c] * inv_n);
            o[j] = (acc ? o[j] : 0.f) + d;
        }
        V8<T>::st(dx + i * 8, o);
    }
}

extern "C" int nvae_bn_bwd_apply(int dtype, const void* x, const void* dy, void* dx, long rows, int C,
                                 const float* scale, const float* shift, const float* mean,
                                 const float* invstd, int act, const float* dgamma, const float* dbeta,
                                 int accumulate, void* stream) {
    if (int e = check_c("bn_bwd_apply", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && aligned16(dy) && aligned16(dx), "bn_bwd_apply: bad rows/alignment");
    long n8 = rows * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bn_bwd_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, n8, C / 8, scale, shift, mean, invstd, act, dgamma, dbeta, 1.0f / (float)rows, accumulate);)
    NVAE_LAUNCH_CHECK("bn_bwd_apply");
    return NVAE_OK;
}

extern "C" int nvae_colsum(int dtype, const void* x, long rows, int C, int ld, float* out, void* stream) {
    if (int e = check_c("colsum", C)) return e;
    NVAE_REQUIRE(rows > 0 && ld >= C && ld % 8 == 0 && aligned16(x), "colsum: bad rows/ld/alignment");
    DISPATCH_T(dtype, launch_colreduce<T, 4>((const T*)x, nullptr, 1, rows, C, ld, nullptr, nullptr, nullptr, nullptr, 0, out, nullptr, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("colsum");
    return NVAE_OK;
}

// ---------------------------------------------------------------------------------------
// Squeeze-Excitation
// ---------------------------------------------------------------------------------------
extern "C" int nvae_se_pool(int dtype, const void* x, int B, int HW, int C, float* pooled_sum, void* stream) {
    if (int e = check_c("se_pool", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x), "se_pool: bad shape/alignment");
    DISPATCH_T(dtype, launch_colreduce<T, 2>((const T*)x, nullptr, B, HW, C, C, nullptr, nullptr, nullptr, nullptr, 0, pooled_sum, nullptr, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("se_pool");
    return NVAE_OK;
}

extern "C" int nvae_se_bwd_reduce(int dtype, const void* x, const void* dy, int B, int HW, int C, float* r, void* stream) {
    if (int e = check_c("se_bwd_reduce", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x) && aligned16(dy), "se_bwd_reduce: bad shape/alignment");
    DISPATCH_T(dtype, launch_colreduce<T, 3>((const T*)x, (const T*)dy, B, HW, C, C, nullptr, nullptr, nullptr, nullptr, 0, r, nullptr, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("se_bwd_reduce");
    return NVAE_OK;
}

#define SE_MAX_C 2048
#define SE_MAX_H 128

// One block per image: hidden = relu(p W1 + b1); gate = sigmoid(hidden W2 + b2)
__global__ void k_se_gate(const float* __restrict__ pooled_sum, float inv_hw, int C, int Hd,
                          const float* __restrict__ w1, const float* __restrict__ b1,
                          const float* __restrict__ w2, const float* __restrict__ b2,
                          float* __restrict__ gate, float* __restrict__ hidden) {
    __shared__ float p[SE_MAX_C];
    __shared__ float hd[SE_MAX_H];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) p[c] = pooled_sum[(long)b * C + c] * inv_hw;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int h = wave; h < Hd; h += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += p[c] * w1[(long)c * Hd + h];
        a = wave_sum(a);
        if (lane == 0) {
            float v = fmaxf(a + b1[h], 0.f);
            hd[h] = v;
            hidden[(long)b * Hd + h] = v;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = b2[c];
        for (int h = 0; h < Hd; ++h) a += hd[h] * w2[(long)h * C + c];
        gate[(long)b * C + c] = sigmoidf_(a);
    }
}

extern "C" int nvae_se_gate(const float* pooled_sum, int B, int HW, int C, int Hd, const float* w1,
                            const float* b1, const float* w2, const float* b2, float* gate,
                            float* hidden, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= SE_MAX_C && Hd > 0 && Hd <= SE_MAX_H, "se_gate: bad shape C=%d Hd=%d", C, Hd);
    hipLaunchKernelGGL(k_se_gate, B, 256, 0, (hipStream_t)stream, pooled_sum, 1.0f / (float)HW, C, Hd, w1, b1, w2, b2, gate, hidden);
    NVAE_LAUNCH_CHECK("se_gate");
    return NVAE_OK;
}

template <typename T>
__global__ void k_se_apply(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y, long n8,
                           int C8, long hwc8, const float* __restrict__ gate, float ss, float bs) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        long b = i / hwc8;
        float v[8], k[8];
        V8<T>::ld(x + i * 8, v);
        V8<T>::ld(skip + i * 8, k);
        const float* g = gate + b * (C8 * 8) + c0;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ss * k[j] + bs * v[j] * g[j];
        V8<T>::st(y + i * 8, v);
    }
}

extern "C" int nvae_se_apply(int dtype, const void* x, const void* skip, void* y, int B, int HW, int C,
                             const float* gate, float skip_scale, float branch_scale, void* stream) {
    if (int e = check_c("se_apply", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x) && aligned16(skip) && aligned16(y), "se_apply: bad shape/alignment");
    long n8 = (long)B * HW * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)x, (const T*)skip, (T*)y, n8, C / 8, (long)HW * (C / 8), gate, skip_scale, branch_scale);)
    NVAE_LAUNCH_CHECK("se_apply");
    return NVAE_OK;
}

// One block per image.  r[b,c] = sum_hw dy*x.  dgate = bs * r.
__global__ void k_se_gate_bwd(const float* __restrict__ r, const float* __restrict__ pooled_sum,
                              const float* __restrict__ gate, const float* __restrict__ hidden,
                              float inv_hw, int C, int Hd, const float* __restrict__ w1,
                              const float* __restrict__ w2, float bs, float* dw1, float* db1,
                              float* dw2, float* db2, float* __restrict__ dpool) {
    __shared__ float dpre2[SE_MAX_C];
    __shared__ float hd[SE_MAX_H];
    __shared__ float dpre1[SE_MAX_H];
    const int b = blockIdx.x;
    for (int h = threadIdx.x; h < Hd; h += 256) hd[h] = hidden[(long)b * Hd + h];
    for (int c = threadIdx.x; c < C; c += 256) {
        float g = gate[(long)b * C + c];
        float d = bs * r[(long)b * C + c] * g * (1.f - g);
        dpre2[c] = d;
        atomicAdd(db2 + c, d);
    }
    __syncthreads();
    // dW2[h, c] += hd[h] * dpre2[c]
    for (int i = threadIdx.x; i < Hd * C; i += 256) {
        int h = i / C, c = i - h * C;
        float v = hd[h] * dpre2[c];
        if (v != 0.f) atomicAdd(dw2 + i, v);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int h = wave; h < Hd; h += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += w2[(long)h * C + c] * dpre2[c];
        a = wave_sum(a);
        if (lane == 0) {
            float d = hd[h] > 0.f ? a : 0.f;
            dpre1[h] = d;
            atomicAdd(db1 + h, d);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float p = pooled_sum[(long)b * C + c] * inv_hw;
        float a = 0.f;
        for (int h = 0; h < Hd; ++h) {
            float d = dpre1[h];
            a += w1[(long)c * Hd + h] * d;
            float v = p * d;
            if (v != 0.f) atomicAdd(dw1 + (long)c * Hd + h, v);
        }
        dpool[(long)b * C + c] = a * inv_hw;
    }
}

extern "C" int nvae_se_gate_bwd(const float* r, const float* pooled_sum, const float* gate,
                                const float* hidden, int B, int HW, int C, int Hd, const float* w1,
                                const float* w2, float branch_scale, float* dw1, float* db1, float* dw2,
                                float* db2, float* dpool, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= SE_MAX_C && Hd > 0 && Hd <= SE_MAX_H, "se_gate_bwd: bad shape C=%d Hd=%d", C, Hd);
    hipLaunchKernelGGL(k_se_gate_bwd, B, 256, 0, (hipStream_t)stream, r, pooled_sum, gate, hidden, 1.0f / (float)HW, C, Hd, w1, w2, branch_scale, dw1, db1, dw2, db2, dpool);
    NVAE_LAUNCH_CHECK("se_gate_bwd");
    return NVAE_OK;
}

// dx (+)= bs * dy * gate + dpool[b,c];   dskip (+)= ss * dy
template <typename T>
__global__ void k_se_bwd_apply(const T* __restrict__ dy, const float* __restrict__ gate,
                               const float* __restrict__ dpool, T* dx, T* dskip, long n8, int C8,
                               long hwc8, float ss, float bs, int acc_dx, int acc_dskip) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        long b = i / hwc8;
        float g[8], o[8], k[8];
        V8<T>::ld(dy + i * 8, g);
        const float* gt = gate + b * (C8 * 8) + c0;
        const float* dp = dpool + b * (C8 * 8) + c0;
        if (acc_dx) V8<T>::ld(dx + i * 8, o);
        if (acc_dskip) V8<T>::ld(dskip + i * 8, k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = (acc_dx ? o[j] : 0.f) + bs * g[j] * gt[j] + dp[j];
            k[j] = (acc_dskip ? k[j] : 0.f) + ss * g[j];
        }
        V8<T>::st(dx + i * 8, o);
        if (dskip) V8<T>::st(dskip + i * 8, k);
    }
}

extern "C" int nvae_se_bwd_apply(int dtype, const void* dy, const float* gate, const float* dpool,
                                 void* dx, void* dskip, int B, int HW, int C, float skip_scale,
                                 float branch_scale, int acc_dx, int acc_dskip, void* stream) {
    if (int e = check_c("se_bwd_apply", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(dy) && aligned16(dx) && aligned16(dskip), "se_bwd_apply: bad shape/alignment");
    NVAE_REQUIRE(dskip || !acc_dskip, "se_bwd_apply: acc_dskip without dskip");
    long n8 = (long)B * HW * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_bwd_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)dy, gate, dpool, (T*)dx, (T*)dskip, n8, C / 8, (long)HW * (C / 8), skip_scale, branch_scale, acc_dx, dskip ? acc_dskip : 0);)
    NVAE_LAUNCH_CHECK("se_bwd_apply");
    return NVAE_OK;
}
