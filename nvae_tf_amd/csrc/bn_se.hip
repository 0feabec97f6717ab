// BatchNorm(+Swish) and Squeeze-Excitation kernels.  HBM-bound.
//
// Reductions over the row axis of an NHWC tensor viewed as [rows, C] share one structure ("strip
// reduce"): a workgroup owns a strip of <= 64 channels and a contiguous range of rows; thread
// (rl, tg) walks rows rl, rl+RL, ... and holds 8 consecutive channels (one 16-B bf16 load), so a
// wave reads whole 128-B lines.  Partials are combined with wave64 shuffles, one LDS hop, and are
// written as a [S][NQ][C] slab that a tiny finalize kernel sums.  No contended atomics: measured on
// MI355X, 1024 workgroups adding into the same 2C addresses cost 250-300 us per call (same-address
// f32 atomics serialise at the memory side), 30x the streaming time of the tensor.
#include "common.h"
#include "bn_fin.h"

#define RED_THREADS 256
#define MAX_SPLITS 128

// MODE 0: bn stats        q0 = x,          q1 = x*x
// MODE 1: bn bwd          q0 = dpre,       q1 = dpre * x          (xhat folded in by the finalize)
// MODE 2: per-image sum   q0 = x           (blockIdx.z = image, S = 1, direct store)
// MODE 3: per-image sum   q0 = x*dy
// MODE 4: column sum with leading dimension ld, q0 = x   (atomics, S small)
// MODE 0 / 1 on f32 tensors accumulate and write their slab in f64 (bn_fin.h "statistics precision")
template <typename T, int MODE> struct StripAcc { typedef float type; };
template <> struct StripAcc<float, 0> { typedef double type; };
template <> struct StripAcc<float, 1> { typedef double type; };

template <typename T, int MODE>
__global__ __launch_bounds__(RED_THREADS) void k_stripreduce(
    const T* __restrict__ x, const T* __restrict__ dy, long rows_per_group, int C, int ld,
    int rows_per_block, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    float* out, BnFinArgs fin) {
    typedef typename StripAcc<T, MODE>::type A;
    constexpr int NQ = (MODE <= 1) ? 2 : 1;
    // thread groups of 8 channels per strip, padded to a power of two (1, 2, 4 or 8) so that lanes
    // holding the same channels are a fixed power-of-two apart
    const int TGS = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
    const int RL = RED_THREADS / TGS;        // row lanes
    const int tg = threadIdx.x % TGS, rl = threadIdx.x / TGS;
    const int c0 = blockIdx.x * 64 + tg * 8;
    const bool cval = c0 < C;
    const long grp = blockIdx.z;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows_per_group) r1 = rows_per_group;
    A acc[NQ][8];
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = (A)0;
    float sc[8], sh[8];
    if (MODE == 1 && cval) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { sc[j] = scale[c0 + j]; sh[j] = shift[c0 + j]; }
    }
    if (cval) {
        for (long r = r0 + rl; r < r1; r += RL) {
            const long off = (grp * rows_per_group + r) * (long)ld + c0;
            float v[8];
            V8<T>::ld(x + off, v);
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[0][j] += (A)v[j]; acc[1][j] += (A)v[j] * (A)v[j]; }
            } else if (MODE == 1) {
                float g[8];
                V8<T>::ld(dy + off, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float dpre = g[j];
                    if (act == ACT_SWISH) dpre *= dswishf_(v[j] * sc[j] + sh[j]);
                    acc[0][j] += (A)dpre;
                    acc[1][j] += (A)dpre * (A)v[j];
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
    // lanes with equal tg are TGS apart: xor-shuffle over the row-lane bits of the lane id
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            A v = acc[q][j];
            for (int o = 32; o >= TGS; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[q][j] = v;
        }
    __shared__ A sm[4][8][NQ * 8];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < TGS) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) sm[wave][lane][q * 8 + j] = acc[q][j];
    }
    __syncthreads();
    if (threadIdx.x < TGS && c0 < C) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                A v = sm[0][tg][q * 8 + j] + sm[1][tg][q * 8 + j] + sm[2][tg][q * 8 + j] + sm[3][tg][q * 8 + j];
                if (MODE <= 1) __hip_atomic_store((A*)out + ((long)blockIdx.y * NQ + q) * C + c0 + j, v, __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);               // slab [S][NQ][C]
                else if (MODE == 4) atomicAdd(out + c0 + j, (float)v);
                else out[grp * C + c0 + j] = (float)v;                                // [B][C], S == 1
            }
    }
    if (MODE <= 1) {
        // fused finalize: the last of this strip's gridDim.y workgroups turns the slabs into coefficients
        if (fin.counter == nullptr) return;
        if (!bn_last_arriver(fin.counter + blockIdx.x, (int)gridDim.y)) return;
        if constexpr (MODE == 0) bn_fin_fwd<sizeof(A) == 8>(fin, out, (int)gridDim.y, C, blockIdx.x * 64, 1);
        else bn_fin_bwd<sizeof(A) == 8>(fin, out, (int)gridDim.y, C, blockIdx.x * 64, 1);
    }
}

// Number of row splits the reductions use for a [rows, C] tensor (slab size = S * 2 * C floats).
extern "C" int nvae_reduce_splits(long rows, int C) {
    if (rows <= 0 || C <= 0) return 1;
    const int tgs = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
    const int rl = RED_THREADS / tgs;
    const int strips = (C + 63) / 64;
    long s = 1024 / strips;
    long max_s = rows / ((long)rl * 2);      // >= 2 rows per thread
    if (s > max_s) s = max_s;
    if (s > MAX_SPLITS) s = MAX_SPLITS;
    if (s < 1) s = 1;
    long rpb = (rows + s - 1) / s;
    return (int)((rows + rpb - 1) / rpb);
}

template <typename T, int MODE>
static void launch_strip(const T* x, const T* dy, long groups, long rows, int C, int ld, int S,
                         const float* scale, const float* shift, int act, float* out, hipStream_t s,
                         const BnFinArgs* fin = nullptr) {
    long rpb = (rows + S - 1) / S;
    dim3 grid((C + 63) / 64, S, (unsigned)groups);
    BnFinArgs f{};
    if (fin) f = *fin;
    hipLaunchKernelGGL((k_stripreduce<T, MODE>), grid, RED_THREADS, 0, s, x, dy, rows, C, ld, (int)rpb,
                       scale, shift, act, out, f);
}

static int check_c(const char* who, int C) {
    NVAE_REQUIRE(C >= 8 && C % 8 == 0 && C <= 8192, "%s: C=%d must be a multiple of 8 in [8, 8192]", who, C);
    return NVAE_OK;
}

extern "C" int nvae_bn_stats(int dtype, const void* x, long rows, int C, float* partials, void* stream) {
    if (int e = check_c("bn_stats", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && partials, "bn_stats: bad rows/alignment");
    const int S = nvae_reduce_splits(rows, C);
    DISPATCH_T(dtype, launch_strip<T, 0>((const T*)x, nullptr, 1, rows, C, C, S, nullptr, nullptr, 0, partials, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("bn_stats");
    return NVAE_OK;
}

template <bool F64>
__global__ void k_bn_finalize(const void* __restrict__ partials, int S, int C, BnFinArgs a) {
    int c; double s1, s2;
    if (bn_slab_sum64_t<F64>(partials, S, C, blockIdx.x * 64, c, s1, s2)) bn_fin_fwd_channel(a, c, s1, s2);
}

static void launch_bn_finalize(int dtype, const float* partials, int S, long rows, int C, const float* gamma,
                               const float* beta, float* rm, float* rv, float momentum, float eps, float* scale,
                               float* shift, float* mean, float* invstd, hipStream_t s) {
    BnFinArgs f{};
    f.inv_n = 1.0f / (float)rows; f.gamma = gamma; f.beta = beta; f.rm = rm; f.rv = rv; f.momentum = momentum;
    f.eps = eps; f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = invstd;
    if (dtype == NVAE_F32) hipLaunchKernelGGL(k_bn_finalize<true>, cdiv(C, 64), 256, 0, s, (const void*)partials, S, C, f);
    else hipLaunchKernelGGL(k_bn_finalize<false>, cdiv(C, 64), 256, 0, s, (const void*)partials, S, C, f);
}

extern "C" int nvae_bn_finalize(int dtype, const float* partials, long rows, int C, const float* gamma,
                                const float* beta, float* rm, float* rv, float momentum, float eps,
                                float* scale, float* shift, float* mean, float* invstd, void* stream) {
    NVAE_REQUIRE(rows > 0 && C > 0, "bn_finalize: bad shape");
    launch_bn_finalize(dtype, partials, nvae_reduce_splits(rows, C), rows, C, gamma, beta, rm, rv, momentum, eps, scale,
                       shift, mean, invstd, (hipStream_t)stream);
    NVAE_LAUNCH_CHECK("bn_finalize");
    return NVAE_OK;
}

extern "C" int nvae_bn_finalize_s(int dtype, const float* partials, int S, long rows, int C, const float* gamma,
                                  const float* beta, float* rm, float* rv, float momentum, float eps,
                                  float* scale, float* shift, float* mean, float* invstd, void* stream) {
    NVAE_REQUIRE(rows > 0 && C > 0 && S > 0, "bn_finalize_s: bad shape");
    launch_bn_finalize(dtype, partials, S, rows, C, gamma, beta, rm, rv, momentum, eps, scale, shift, mean, invstd,
                       (hipStream_t)stream);
    NVAE_LAUNCH_CHECK("bn_finalize_s");
    return NVAE_OK;
}

// Statistics + finalize in one launch (bn_fin.h).  `counters`: >= ceil(C/64) zero-initialised ints.
extern "C" int nvae_bn_stats_fin(int dtype, const void* x, long rows, int C, float* partials, int* counters,
                                 const float* gamma, const float* beta, float* rm, float* rv,
                                 float momentum, float eps, float* scale, float* shift, float* mean,
                                 float* invstd, void* stream) {
    if (int e = check_c("bn_stats_fin", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && partials && counters && gamma && beta && rm && rv && scale && shift && mean && invstd,
                 "bn_stats_fin: bad rows/alignment/null argument");
    const int S = nvae_reduce_splits(rows, C);
    BnFinArgs f{};
    f.counter = counters; f.inv_n = 1.0f / (float)rows; f.gamma = gamma; f.beta = beta; f.rm = rm; f.rv = rv;
    f.momentum = momentum; f.eps = eps; f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = invstd;
    DISPATCH_T(dtype, launch_strip<T, 0>((const T*)x, nullptr, 1, rows, C, C, S, nullptr, nullptr, 0, partials, (hipStream_t)stream, &f);)
    NVAE_LAUNCH_CHECK("bn_stats_fin");
    return NVAE_OK;
}

__global__ void k_bn_eval_prepare(const float* gamma, const float* beta, const float* rm,
                                  const float* rv, int C, float eps, float* scale, float* shift,
                                  float* mean, float* invstd) {
    int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    float is = rsqrtf(rv[c] + eps);
    float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - rm[c] * sc;
    if (mean) { mean[c] = rm[c]; invstd[c] = is; }
}

extern "C" int nvae_bn_eval_prepare(const float* gamma, const float* beta, const float* rm,
                                    const float* rv, int C, float eps, float* scale, float* shift,
                                    float* mean, float* invstd, void* stream) {
    NVAE_REQUIRE(C > 0 && ((mean == nullptr) == (invstd == nullptr)), "bn_eval_prepare: bad args");
    hipLaunchKernelGGL(k_bn_eval_prepare, cdiv(C, 256), 256, 0, (hipStream_t)stream, gamma, beta, rm, rv, C, eps, scale, shift, mean, invstd);
    NVAE_LAUNCH_CHECK("bn_eval_prepare");
    return NVAE_OK;
}

static inline int ew_grid(long n8) {
    long g = (n8 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

__device__ __forceinline__ void ld8f(const float* p, float (&v)[8]) {
    float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

template <typename T>
__global__ void k_bn_apply(const T* __restrict__ x, T* __restrict__ y, long n8, int C8,
                           const float* __restrict__ scale, const float* __restrict__ shift, int act) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        float v[8], sc[8], sh[8];
        V8<T>::ld(x + i * 8, v);
        ld8f(scale + c0, sc);
        ld8f(shift + c0, sh);
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
                                  const float* scale, const float* shift, int act, float* partials,
                                  void* stream) {
    if (int e = check_c("bn_bwd_reduce", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && aligned16(dy) && partials, "bn_bwd_reduce: bad rows/alignment");
    const int S = nvae_reduce_splits(rows, C);
    DISPATCH_T(dtype, launch_strip<T, 1>((const T*)x, (const T*)dy, 1, rows, C, C, S, scale, shift, act, partials, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("bn_bwd_reduce");
    return NVAE_OK;
}

// dbeta = sum dpre; dgamma = sum dpre*xhat = invstd * (sum dpre*x - mean * sum dpre).
// dx = scale*(dpre - dbeta/N - xhat*dgamma/N) = scale*dpre + k1*x + k0.
template <bool F64>
__global__ void k_bn_bwd_finalize(const void* __restrict__ partials, int S, int C, BnFinArgs a) {
    int c; double s1, s2;
    if (bn_slab_sum64_t<F64>(partials, S, C, blockIdx.x * 64, c, s1, s2)) bn_fin_bwd_channel(a, C, c, s1, s2);
}

static void launch_bn_bwd_finalize(int dtype, const float* partials, int S, long rows, int C, const float* scale,
                                   const float* mean, const float* invstd, float* dgamma, float* dbeta, float* k0k1,
                                   int frozen, hipStream_t s) {
    BnFinArgs f{};
    f.inv_n = 1.0f / (float)rows; f.scale = (float*)scale; f.mean = (float*)mean; f.invstd = (float*)invstd;
    f.dgamma = dgamma; f.dbeta = dbeta; f.k0k1 = k0k1; f.frozen = frozen;
    if (dtype == NVAE_F32) hipLaunchKernelGGL(k_bn_bwd_finalize<true>, cdiv(C, 64), 256, 0, s, (const void*)partials, S, C, f);
    else hipLaunchKernelGGL(k_bn_bwd_finalize<false>, cdiv(C, 64), 256, 0, s, (const void*)partials, S, C, f);
}

extern "C" int nvae_bn_bwd_finalize(int dtype, const float* partials, long rows, int C, const float* scale,
                                    const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                    float* k0k1, int frozen, void* stream) {
    NVAE_REQUIRE(rows > 0 && C > 0 && partials && k0k1, "bn_bwd_finalize: bad args");
    launch_bn_bwd_finalize(dtype, partials, nvae_reduce_splits(rows, C), rows, C, scale, mean, invstd, dgamma, dbeta, k0k1,
                           frozen, (hipStream_t)stream);
    NVAE_LAUNCH_CHECK("bn_bwd_finalize");
    return NVAE_OK;
}

// Backward reduction + finalize in one launch (bn_fin.h).
extern "C" int nvae_bn_bwd_reduce_fin(int dtype, const void* x, const void* dy, long rows, int C,
                                      const float* scale, const float* shift, const float* mean,
                                      const float* invstd, int act, float* partials, int* counters,
                                      float* dgamma, float* dbeta, float* k0k1, int frozen, void* stream) {
    if (int e = check_c("bn_bwd_reduce_fin", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && aligned16(dy) && partials && counters && scale && shift && mean && invstd && dgamma && dbeta && k0k1,
                 "bn_bwd_reduce_fin: bad rows/alignment/null argument");
    const int S = nvae_reduce_splits(rows, C);
    BnFinArgs f{};
    f.counter = counters; f.inv_n = 1.0f / (float)rows;
    f.scale = (float*)scale; f.mean = (float*)mean; f.invstd = (float*)invstd;
    f.dgamma = dgamma; f.dbeta = dbeta; f.k0k1 = k0k1; f.frozen = frozen;
    DISPATCH_T(dtype, launch_strip<T, 1>((const T*)x, (const T*)dy, 1, rows, C, C, S, scale, shift, act, partials, (hipStream_t)stream, &f);)
    NVAE_LAUNCH_CHECK("bn_bwd_reduce_fin");
    return NVAE_OK;
}

// as nvae_bn_bwd_finalize for a slab with an explicit number of rows (the accumulated slab of
// nvae_conv_gemm_bnbwd: S = nvae_conv_gemm_stats_rows)
extern "C" int nvae_bn_bwd_finalize_s(int dtype, const float* partials, int S, long rows, int C, const float* scale,
                                      const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                      float* k0k1, int frozen, void* stream) {
    NVAE_REQUIRE(rows > 0 && C > 0 && S > 0 && partials && k0k1, "bn_bwd_finalize_s: bad args");
    launch_bn_bwd_finalize(dtype, partials, S, rows, C, scale, mean, invstd, dgamma, dbeta, k0k1, frozen, (hipStream_t)stream);
    NVAE_LAUNCH_CHECK("bn_bwd_finalize_s");
    return NVAE_OK;
}

template <typename T>
__global__ void k_bn_bwd_apply(const T* __restrict__ x, const T* __restrict__ dy, T* dx, long n8, int C8,
                               const float* __restrict__ scale, const float* __restrict__ shift,
                               const float* __restrict__ k0k1, int act, int acc) {
    const int C = C8 * 8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        float v[8], g[8], o[8], sc[8], sh[8], k0[8], k1[8];
        V8<T>::ld(x + i * 8, v);
        V8<T>::ld(dy + i * 8, g);
        if (acc) V8<T>::ld(dx + i * 8, o);
        ld8f(scale + c0, sc);
        ld8f(k0k1 + c0, k0);
        ld8f(k0k1 + C + c0, k1);
        if (act == ACT_SWISH) ld8f(shift + c0, sh);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float dpre = g[j];
            if (act == ACT_SWISH) dpre *= dswishf_(v[j] * sc[j] + sh[j]);
            float d = sc[j] * dpre + k1[j] * v[j] + k0[j];
            o[j] = (acc ? o[j] : 0.f) + d;
        }
        V8<T>::st(dx + i * 8, o);
    }
}

extern "C" int nvae_bn_bwd_apply(int dtype, const void* x, const void* dy, void* dx, long rows, int C,
                                 const float* scale, const float* shift, const float* k0k1, int act,
                                 int accumulate, void* stream) {
    if (int e = check_c("bn_bwd_apply", C)) return e;
    NVAE_REQUIRE(rows > 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(scale) && aligned16(shift) && aligned16(k0k1),
                 "bn_bwd_apply: bad rows/alignment");
    long n8 = rows * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bn_bwd_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, n8, C / 8, scale, shift, k0k1, act, accumulate);)
    NVAE_LAUNCH_CHECK("bn_bwd_apply");
    return NVAE_OK;
}

extern "C" int nvae_colsum(int dtype, const void* x, long rows, int C, int ld, float* out, void* stream) {
    // no channel-count ceiling here: the decoder's learned constant h is summed over the batch as one
    // [B, H*W*C] matrix (32768 columns for the CIFAR-10 configuration)
    NVAE_REQUIRE(C >= 8 && C % 8 == 0 && C <= (1 << 22), "colsum: C=%d must be a multiple of 8 in [8, 2^22]", C);
    NVAE_REQUIRE(rows > 0 && ld >= C && ld % 8 == 0 && aligned16(x), "colsum: bad rows/ld/alignment");
    int S = nvae_reduce_splits(rows, C);
    if (S > 8) S = 8;   // atomics: keep the adders per address few
    if (g_nvae_det) S = 1;
    DISPATCH_T(dtype, launch_strip<T, 4>((const T*)x, nullptr, 1, rows, C, ld, S, nullptr, nullptr, 0, out, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("colsum");
    return NVAE_OK;
}

// ---------------------------------------------------------------------------------------
// Squeeze-Excitation
// ---------------------------------------------------------------------------------------
extern "C" int nvae_se_pool(int dtype, const void* x, int B, int HW, int C, float* pooled_sum, void* stream) {
    if (int e = check_c("se_pool", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x), "se_pool: bad shape/alignment");
    DISPATCH_T(dtype, launch_strip<T, 2>((const T*)x, nullptr, B, HW, C, C, 1, nullptr, nullptr, 0, pooled_sum, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("se_pool");
    return NVAE_OK;
}

extern "C" int nvae_se_bwd_reduce(int dtype, const void* x, const void* dy, int B, int HW, int C, float* r, void* stream) {
    if (int e = check_c("se_bwd_reduce", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x) && aligned16(dy), "se_bwd_reduce: bad shape/alignment");
    DISPATCH_T(dtype, launch_strip<T, 3>((const T*)x, (const T*)dy, B, HW, C, C, 1, nullptr, nullptr, 0, r, (hipStream_t)stream);)
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

// Pool + gate in one launch: one workgroup per image sums its own [HW, C] tile (48-96 KB at the tower shapes:
// 1-2 us at what one CU can pull in) and goes straight on to the two FC layers, instead of a separate
// pooling launch whose only consumer is this kernel.  pooled_sum is still written (the FC weight gradients
// need it).  MODE_R: the same for the backward pass, r[b,c] = sum_hw x*dy feeding k_se_gate_bwd's math.
template <typename T>
__device__ __forceinline__ void se_image_sums(const T* __restrict__ x, const T* __restrict__ dy, int HW, int C,
                                              float* __restrict__ p /* LDS [C] */, float* part /* LDS [2048] */) {
    const int CG = C / 8;                                    // 8-channel groups; C <= 2048 -> CG <= 256
    const int RLn = 256 / CG;                                // row lanes (>= 1)
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (rl < RLn)
        for (int r = rl; r < HW; r += RLn) {
            float v[8];
            V8<T>::ld(x + (long)r * C + tg * 8, v);
            if (dy) {
                float g[8];
                V8<T>::ld(dy + (long)r * C + tg * 8, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += v[j] * g[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) a[j] += v[j];
            }
        }
    if (rl < RLn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) part[rl * C + tg * 8 + j] = a[j];     // RLn * C <= 2048
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float v = 0.f;
        for (int k = 0; k < RLn; ++k) v += part[k * C + c];
        p[c] = v;
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void k_se_pool_gate(const T* __restrict__ x, int HW, float inv_hw, int C, int Hd,
                                                      const float* __restrict__ w1, const float* __restrict__ b1,
                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                      float* __restrict__ pooled_sum, float* __restrict__ gate,
                                                      float* __restrict__ hidden) {
    __shared__ float p[SE_MAX_C];
    __shared__ float part[2048];
    __shared__ float hd[SE_MAX_H];
    const int b = blockIdx.x;
    se_image_sums<T>(x + (long)b * HW * C, nullptr, HW, C, p, part);
    for (int c = threadIdx.x; c < C; c += 256) {
        pooled_sum[(long)b * C + c] = p[c];
        p[c] *= inv_hw;
    }
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

extern "C" int nvae_se_pool_gate(int dtype, const void* x, int B, int HW, int C, int Hd, const float* w1,
                                 const float* b1, const float* w2, const float* b2, float* pooled_sum,
                                 float* gate, float* hidden, void* stream) {
    if (int e = check_c("se_pool_gate", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && C <= SE_MAX_C && Hd > 0 && Hd <= SE_MAX_H && aligned16(x) && pooled_sum && gate && hidden,
                 "se_pool_gate: bad shape C=%d Hd=%d / alignment", C, Hd);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_pool_gate<T>), B, 256, 0, (hipStream_t)stream, (const T*)x, HW, 1.0f / (float)HW, C, Hd, w1, b1, w2, b2, pooled_sum, gate, hidden);)
    NVAE_LAUNCH_CHECK("se_pool_gate");
    return NVAE_OK;
}

template <typename T>
__global__ void k_se_apply(const T* __restrict__ x, const T* __restrict__ skip, T* __restrict__ y, long n8,
                           int C8, long hwc8, const float* __restrict__ gate, float ss, float bs) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        long b = i / hwc8;
        float v[8], k[8], g[8];
        V8<T>::ld(x + i * 8, v);
        V8<T>::ld(skip + i * 8, k);
        ld8f(gate + b * (C8 * 8) + c0, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = ss * k[j] + bs * v[j] * g[j];
        V8<T>::st(y + i * 8, v);
    }
}

extern "C" int nvae_se_apply(int dtype, const void* x, const void* skip, void* y, int B, int HW, int C,
                             const float* gate, float skip_scale, float branch_scale, void* stream) {
    if (int e = check_c("se_apply", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x) && aligned16(skip) && aligned16(y) && aligned16(gate), "se_apply: bad shape/alignment");
    long n8 = (long)B * HW * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)x, (const T*)skip, (T*)y, n8, C / 8, (long)HW * (C / 8), gate, skip_scale, branch_scale);)
    NVAE_LAUNCH_CHECK("se_apply");
    return NVAE_OK;
}

// SE gate + residual that also emits the BatchNorm statistics of its output (the next residual cell
// starts with BN): strip-reduce structure (a workgroup owns <= 64 channels x a row range), y is written
// and its column sums / sums of squares (from the f32 values) go to stats[blockIdx.y][2][C].
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_se_apply_stats(const T* __restrict__ x, const T* __restrict__ skip,
                                                               T* __restrict__ y, long rows, int C, int HW,
                                                               int rows_per_block, const float* __restrict__ gate,
                                                               float ss, float bs, float* __restrict__ stats_) {
    typedef typename StatT<T>::type ST;          // slab element type (bn_fin.h "statistics precision")
    ST* __restrict__ stats = (ST*)stats_;
    const int TGS = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
    const int RL = RED_THREADS / TGS;
    const int tg = threadIdx.x % TGS, rl = threadIdx.x / TGS;
    const int c0 = blockIdx.x * 64 + tg * 8;
    const bool cval = c0 < C;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float acc[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = 0.f;
    if (cval) {
        for (long r = r0 + rl; r < r1; r += RL) {
            const long off = r * (long)C + c0;
            float v[8], k[8], g[8];
            V8<T>::ld(x + off, v);
            V8<T>::ld(skip + off, k);
            ld8f(gate + (r / HW) * C + c0, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = ss * k[j] + bs * v[j] * g[j];
                acc[0][j] += v[j]; acc[1][j] += v[j] * v[j];
            }
            V8<T>::st(y + off, v);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[q][j];
            for (int o = 32; o >= TGS; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[q][j] = v;
        }
    __shared__ float sm[4][8][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < TGS) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) sm[wave][lane][q * 8 + j] = acc[q][j];
    }
    __syncthreads();
    if (threadIdx.x < TGS && cval) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                stats[((long)blockIdx.y * 2 + q) * C + c0 + j] =
                    sm[0][tg][q * 8 + j] + sm[1][tg][q * 8 + j] + sm[2][tg][q * 8 + j] + sm[3][tg][q * 8 + j];
    }
}

// stats: [nvae_reduce_splits(B*HW, C)][2][C] floats, consumed by nvae_bn_finalize_s
extern "C" int nvae_se_apply_stats(int dtype, const void* x, const void* skip, void* y, int B, int HW, int C,
                                   const float* gate, float skip_scale, float branch_scale, float* stats,
                                   void* stream) {
    if (int e = check_c("se_apply_stats", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(x) && aligned16(skip) && aligned16(y) && aligned16(gate) && stats,
                 "se_apply_stats: bad shape/alignment");
    const long rows = (long)B * HW;
    const int S = nvae_reduce_splits(rows, C);
    const long rpb = (rows + S - 1) / S;
    dim3 grid((C + 63) / 64, S);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_apply_stats<T>), grid, RED_THREADS, 0, (hipStream_t)stream, (const T*)x, (const T*)skip, (T*)y, rows, C, HW, (int)rpb, gate, skip_scale, branch_scale, stats);)
    NVAE_LAUNCH_CHECK("se_apply_stats");
    return NVAE_OK;
}

// Stage 1, one block per image, no atomics: dpre2[b,c] = bs*r*g*(1-g); dpre1[b,h]; dpool[b,c].
__global__ void k_se_gate_bwd(const float* __restrict__ r, const float* __restrict__ gate,
                              const float* __restrict__ hidden, float inv_hw, int C, int Hd,
                              const float* __restrict__ w1, const float* __restrict__ w2, float bs,
                              float* __restrict__ dpre2_out, float* __restrict__ dpre1_out,
                              float* __restrict__ dpool) {
    __shared__ float dpre2[SE_MAX_C];
    __shared__ float dpre1[SE_MAX_H];
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float g = gate[(long)b * C + c];
        float d = bs * r[(long)b * C + c] * g * (1.f - g);
        dpre2[c] = d;
        dpre2_out[(long)b * C + c] = d;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int h = wave; h < Hd; h += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += w2[(long)h * C + c] * dpre2[c];
        a = wave_sum(a);
        if (lane == 0) {
            float d = hidden[(long)b * Hd + h] > 0.f ? a : 0.f;
            dpre1[h] = d;
            dpre1_out[(long)b * Hd + h] = d;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int h = 0; h < Hd; ++h) a += w1[(long)c * Hd + h] * dpre1[h];
        dpool[(long)b * C + c] = a * inv_hw;
    }
}

// k_se_gate_bwd with the per-image reduction r[c] = sum_hw x*dy done in place (one workgroup per image), see
// k_se_pool_gate.
template <typename T>
__global__ __launch_bounds__(256) void k_se_reduce_gate_bwd(const T* __restrict__ x, const T* __restrict__ dy, int HW,
                                                            const float* __restrict__ gate,
                                                            const float* __restrict__ hidden, float inv_hw, int C,
                                                            int Hd, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, float bs,
                                                            float* __restrict__ dpre2_out,
                                                            float* __restrict__ dpre1_out, float* __restrict__ dpool) {
    __shared__ float dpre2[SE_MAX_C];
    __shared__ float part[2048];
    __shared__ float dpre1[SE_MAX_H];
    const int b = blockIdx.x;
    se_image_sums<T>(x + (long)b * HW * C, dy + (long)b * HW * C, HW, C, dpre2, part);     // dpre2 <- r
    for (int c = threadIdx.x; c < C; c += 256) {
        const float g = gate[(long)b * C + c];
        const float d = bs * dpre2[c] * g * (1.f - g);
        dpre2[c] = d;
        dpre2_out[(long)b * C + c] = d;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int h = wave; h < Hd; h += 4) {
        float a = 0.f;
        for (int c = lane; c < C; c += 64) a += w2[(long)h * C + c] * dpre2[c];
        a = wave_sum(a);
        if (lane == 0) {
            const float d = hidden[(long)b * Hd + h] > 0.f ? a : 0.f;
            dpre1[h] = d;
            dpre1_out[(long)b * Hd + h] = d;
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int h = 0; h < Hd; ++h) a += w1[(long)c * Hd + h] * dpre1[h];
        dpool[(long)b * C + c] = a * inv_hw;
    }
}

// Stage 2: weight gradients as batch contractions, one thread per output, no atomics.
//   dW2[h,c] += sum_b hid[b,h]*dpre2[b,c];  db2[c] += sum_b dpre2[b,c]
//   dW1[c,h] += sum_b p[b,c]*dpre1[b,h];    db1[h] += sum_b dpre1[b,h]
#define SE_BATCH_MAX 32
struct SeWgradBatch {       // layers of one shape share a launch: blockIdx.y selects the layer
    const float* pooled[SE_BATCH_MAX]; const float* hidden[SE_BATCH_MAX]; const float* scratch[SE_BATCH_MAX];
    float* dw1[SE_BATCH_MAX]; float* db1[SE_BATCH_MAX]; float* dw2[SE_BATCH_MAX]; float* db2[SE_BATCH_MAX];
};

__global__ void k_se_wgrad(SeWgradBatch bt, int B, float inv_hw, int C, int Hd) {
    const float* __restrict__ pooled_sum = bt.pooled[blockIdx.y];
    const float* __restrict__ hidden = bt.hidden[blockIdx.y];
    const float* __restrict__ dpre2 = bt.scratch[blockIdx.y];
    const float* __restrict__ dpre1 = dpre2 + (long)B * C;
    float* dw1 = bt.dw1[blockIdx.y]; float* db1 = bt.db1[blockIdx.y];
    float* dw2 = bt.dw2[blockIdx.y]; float* db2 = bt.db2[blockIdx.y];
    // 32 outputs per workgroup (lanes 0-31: consecutive outputs, coalesced over c), 8 batch lanes
    __shared__ float sm[8][32][2];
    const int ol = threadIdx.x & 31, bl = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + ol;
    const int n2 = Hd * C;
    float a2 = 0.f, a1 = 0.f;
    int h = 0, c = 0, kind = 3;
    if (i < n2) { kind = 0; h = i / C; c = i - h * C; }
    else if (i < n2 + C) { kind = 1; c = i - n2; }
    else if (i < n2 + C + Hd) { kind = 2; h = i - n2 - C; }
    if (kind == 0) {
        for (int b = bl; b < B; b += 8) {
            a2 += hidden[(long)b * Hd + h] * dpre2[(long)b * C + c];
            a1 += pooled_sum[(long)b * C + c] * dpre1[(long)b * Hd + h];
        }
    } else if (kind == 1) {
        for (int b = bl; b < B; b += 8) a2 += dpre2[(long)b * C + c];
    } else if (kind == 2) {
        for (int b = bl; b < B; b += 8) a2 += dpre1[(long)b * Hd + h];
    }
    sm[bl][ol][0] = a2; sm[bl][ol][1] = a1;
    __syncthreads();
    if (bl != 0 || kind == 3) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) { a2 += sm[k][ol][0]; a1 += sm[k][ol][1]; }
    if (kind == 0) { dw2[i] += a2; dw1[(long)c * Hd + h] += a1 * inv_hw; }
    else if (kind == 1) db2[c] += a2;
    else db1[h] += a2;
}

extern "C" int nvae_se_gate_bwd(const float* r, const float* pooled_sum, const float* gate,
                                const float* hidden, int B, int HW, int C, int Hd, const float* w1,
                                const float* w2, float branch_scale, float* dw1, float* db1, float* dw2,
                                float* db2, float* dpool, float* scratch, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= SE_MAX_C && Hd > 0 && Hd <= SE_MAX_H, "se_gate_bwd: bad shape C=%d Hd=%d", C, Hd);
    NVAE_REQUIRE(scratch, "se_gate_bwd: scratch [B*(C+Hd)] floats required");
    float* dpre2 = scratch;
    float* dpre1 = scratch + (long)B * C;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_se_gate_bwd, B, 256, 0, s, r, gate, hidden, 1.0f / (float)HW, C, Hd, w1, w2, branch_scale, dpre2, dpre1, dpool);
    if (dw1) {
        SeWgradBatch bt{};
        bt.pooled[0] = pooled_sum; bt.hidden[0] = hidden; bt.scratch[0] = scratch;
        bt.dw1[0] = dw1; bt.db1[0] = db1; bt.dw2[0] = dw2; bt.db2[0] = db2;
        hipLaunchKernelGGL(k_se_wgrad, dim3(cdiv((long)Hd * C + C + Hd, 32), 1), 256, 0, s, bt, B, 1.0f / (float)HW, C, Hd);
    }
    NVAE_LAUNCH_CHECK("se_gate_bwd");
    return NVAE_OK;
}

// The parameter gradients of the two FC layers from the scratch nvae_se_gate_bwd left (dw1 == NULL
// there): independent of the data-gradient chain, so the caller may enqueue it on another stream.
// nvae_se_bwd_reduce + nvae_se_gate_bwd (without the FC parameter gradients) in ONE launch; C <= 2048.
extern "C" int nvae_se_reduce_gate_bwd(int dtype, const void* x, const void* dy, const float* gate,
                                       const float* hidden, int B, int HW, int C, int Hd, const float* w1,
                                       const float* w2, float branch_scale, float* dpool, float* scratch,
                                       void* stream) {
    if (int e = check_c("se_reduce_gate_bwd", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && C <= SE_MAX_C && Hd > 0 && Hd <= SE_MAX_H && aligned16(x) && aligned16(dy) && gate &&
                 hidden && w1 && w2 && dpool && scratch, "se_reduce_gate_bwd: bad shape C=%d Hd=%d / alignment / NULL", C, Hd);
    float* dpre2 = scratch;
    float* dpre1 = scratch + (long)B * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_reduce_gate_bwd<T>), B, 256, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, HW, gate, hidden, 1.0f / (float)HW, C, Hd, w1, w2, branch_scale, dpre2, dpre1, dpool);)
    NVAE_LAUNCH_CHECK("se_reduce_gate_bwd");
    return NVAE_OK;
}

extern "C" int nvae_se_wgrad_batched(int n, const float* const* pooled_sum, const float* const* hidden,
                                     const float* const* scratch, int B, int HW, int C, int Hd, float* const* dw1,
                                     float* const* db1, float* const* dw2, float* const* db2, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && C > 0 && C <= SE_MAX_C && Hd > 0 && Hd <= SE_MAX_H, "se_wgrad: bad shape C=%d Hd=%d", C, Hd);
    NVAE_REQUIRE(n >= 1 && n <= SE_BATCH_MAX && pooled_sum && hidden && scratch && dw1 && db1 && dw2 && db2,
                 "se_wgrad: n=%d must be in [1, %d], no NULL table", n, SE_BATCH_MAX);
    SeWgradBatch bt{};
    for (int i = 0; i < n; ++i) {
        NVAE_REQUIRE(pooled_sum[i] && hidden[i] && scratch[i] && dw1[i] && db1[i] && dw2[i] && db2[i], "se_wgrad: NULL pointer in layer %d", i);
        bt.pooled[i] = pooled_sum[i]; bt.hidden[i] = hidden[i]; bt.scratch[i] = scratch[i];
        bt.dw1[i] = dw1[i]; bt.db1[i] = db1[i]; bt.dw2[i] = dw2[i]; bt.db2[i] = db2[i];
    }
    hipLaunchKernelGGL(k_se_wgrad, dim3(cdiv((long)Hd * C + C + Hd, 32), n), 256, 0, (hipStream_t)stream, bt, B,
                       1.0f / (float)HW, C, Hd);
    NVAE_LAUNCH_CHECK("se_wgrad");
    return NVAE_OK;
}

// The parameter gradients of the two FC layers from the scratch nvae_se_gate_bwd left (dw1 == NULL
// there): independent of the data-gradient chain, so the caller may enqueue it on another stream.
extern "C" int nvae_se_wgrad(const float* pooled_sum, const float* hidden, const float* scratch, int B, int HW,
                             int C, int Hd, float* dw1, float* db1, float* dw2, float* db2, void* stream) {
    return nvae_se_wgrad_batched(1, &pooled_sum, &hidden, &scratch, B, HW, C, Hd, &dw1, &db1, &dw2, &db2, stream);
}

// dx (+)= bs * dy * gate + dpool[b,c];   dskip (+)= ss * dy
template <typename T>
__global__ void k_se_bwd_apply(const T* __restrict__ dy, const float* __restrict__ gate,
                               const float* __restrict__ dpool, T* dx, T* dskip, long n8, int C8,
                               long hwc8, float ss, float bs, int acc_dx, int acc_dskip) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        long b = i / hwc8;
        float g[8], o[8], k[8], gt[8], dp[8];
        V8<T>::ld(dy + i * 8, g);
        ld8f(gate + b * (C8 * 8) + c0, gt);
        ld8f(dpool + b * (C8 * 8) + c0, dp);
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
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(dy) && aligned16(dx) && aligned16(dskip) && aligned16(gate) && aligned16(dpool), "se_bwd_apply: bad shape/alignment");
    NVAE_REQUIRE(dskip || !acc_dskip, "se_bwd_apply: acc_dskip without dskip");
    long n8 = (long)B * HW * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_bwd_apply<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)dy, gate, dpool, (T*)dx, (T*)dskip, n8, C / 8, (long)HW * (C / 8), skip_scale, branch_scale, acc_dx, dskip ? acc_dskip : 0);)
    NVAE_LAUNCH_CHECK("se_bwd_apply");
    return NVAE_OK;
}

// nvae_se_bwd_apply for the case where x (the SE input) is the output of a BatchNorm and has no other
// consumer: dx is final here, so the BatchNorm-backward sums (sum dpre, sum dpre*xb with
// dpre = dx * act'(scale*xb + shift), xb = the BatchNorm's input) are reduced in the same pass and written
// as partials[S][2][C] (S = nvae_reduce_splits(B*HW, C)) for nvae_bn_bwd_finalize_s.
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_se_bwd_apply_bn(
    const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ dpool, T* __restrict__ dx,
    T* dskip, long rows, int C, int HW, int rows_per_block, float ss, float bs, int acc_dskip,
    const T* __restrict__ xb, const float* __restrict__ scale, const float* __restrict__ shift, int act,
    float* __restrict__ partials_) {
    typedef typename StatT<T>::type ST;          // slab element type (bn_fin.h "statistics precision")
    ST* __restrict__ partials = (ST*)partials_;
    const int TGS = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
    const int RL = RED_THREADS / TGS;
    const int tg = threadIdx.x % TGS, rl = threadIdx.x / TGS;
    const int c0 = blockIdx.x * 64 + tg * 8;
    const bool cval = c0 < C;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    float acc[2][8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        acc[0][j] = 0.f; acc[1][j] = 0.f;
        sc[j] = cval ? scale[c0 + j] : 0.f;
        sh[j] = cval ? shift[c0 + j] : 0.f;
    }
    if (cval) {
        for (long r = r0 + rl; r < r1; r += RL) {
            const long off = r * (long)C + c0;
            const long b = r / HW;
            float g[8], o[8], k[8], gt[8], dp[8], xv[8];
            V8<T>::ld(dy + off, g);
            ld8f(gate + b * C + c0, gt);
            ld8f(dpool + b * C + c0, dp);
            V8<T>::ld(xb + off, xv);
            if (dskip && acc_dskip) V8<T>::ld(dskip + off, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                o[j] = bs * g[j] * gt[j] + dp[j];
                k[j] = ((dskip && acc_dskip) ? k[j] : 0.f) + ss * g[j];
                float dpre = o[j];
                if (act == ACT_SWISH) dpre *= dswishf_(xv[j] * sc[j] + sh[j]);
                acc[0][j] += dpre;
                acc[1][j] += dpre * xv[j];
            }
            V8<T>::st(dx + off, o);
            if (dskip) V8<T>::st(dskip + off, k);
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[q][j];
            for (int o = 32; o >= TGS; o >>= 1) v += __shfl_xor(v, o, 64);
            acc[q][j] = v;
        }
    __shared__ float sm[4][8][16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < TGS) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) sm[wave][lane][q * 8 + j] = acc[q][j];
    }
    __syncthreads();
    if (threadIdx.x < TGS && cval) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j)
                partials[((long)blockIdx.y * 2 + q) * C + c0 + j] =
                    sm[0][tg][q * 8 + j] + sm[1][tg][q * 8 + j] + sm[2][tg][q * 8 + j] + sm[3][tg][q * 8 + j];
    }
}

extern "C" int nvae_se_bwd_apply_bn(int dtype, const void* dy, const float* gate, const float* dpool, void* dx,
                                    void* dskip, int B, int HW, int C, float skip_scale, float branch_scale,
                                    int acc_dskip, const void* xb, const float* scale, const float* shift, int act,
                                    float* partials, void* stream) {
    if (int e = check_c("se_bwd_apply_bn", C)) return e;
    NVAE_REQUIRE(B > 0 && HW > 0 && aligned16(dy) && aligned16(dx) && aligned16(dskip) && aligned16(gate) &&
                 aligned16(dpool) && aligned16(xb) && scale && shift && partials, "se_bwd_apply_bn: bad shape/alignment");
    NVAE_REQUIRE(dskip || !acc_dskip, "se_bwd_apply_bn: acc_dskip without dskip");
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "se_bwd_apply_bn: act %d unsupported", act);
    const long rows = (long)B * HW;
    const int S = nvae_reduce_splits(rows, C);
    const long rpb = (rows + S - 1) / S;
    dim3 grid((C + 63) / 64, S);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_bwd_apply_bn<T>), grid, RED_THREADS, 0, (hipStream_t)stream, (const T*)dy, gate, dpool, (T*)dx, (T*)dskip, rows, C, HW, (int)rpb, skip_scale, branch_scale, acc_dskip, (const T*)xb, scale, shift, act, partials);)
    NVAE_LAUNCH_CHECK("se_bwd_apply_bn");
    return NVAE_OK;
}

// ---- apply passes that finish the reduction themselves ---------------------------------------------
// When the slab partials come from another kernel (conv / depthwise / SE epilogues), a separate finalize
// launch costs ~5 us plus its dependency edge, 380 times per step.  These apply kernels are
// strip-structured instead (a workgroup owns 64 channels x a row range): every workgroup first sums its
// strip's slab rows itself (one memory round trip, bn_slab_sum64) and derives the 64 coefficients into
// LDS; the workgroups with blockIdx.y == 0 also publish them (and update the moving statistics /
// parameter gradients) for the later passes.
template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_bn_apply_fin(
    const T* __restrict__ x, T* __restrict__ y, long rows, int C, int rows_per_block,
    const float* __restrict__ partials, int S, BnFinArgs a, int act) {
    __shared__ float s_sc[64], s_sh[64];
    {
        int c; double s1, s2;
        if (bn_slab_sum64_t<sizeof(T) == 4>(partials, S, C, blockIdx.x * 64, c, s1, s2)) {
            const double md = s1 * (double)a.inv_n;
            const float m = (float)md, var = (float)fmax(s2 * (double)a.inv_n - md * md, 0.0);
            const float is = rsqrtf(var + a.eps);
            const float sc = a.gamma[c] * is, sh = a.beta[c] - m * sc;
            s_sc[c & 63] = sc; s_sh[c & 63] = sh;
            if (blockIdx.y == 0) {
                a.scale[c] = sc; a.shift[c] = sh; a.mean[c] = m; a.invstd[c] = is;
                a.rm[c] = a.rm[c] * a.momentum + m * (1.f - a.momentum);
                a.rv[c] = a.rv[c] * a.momentum + var * (1.f - a.momentum);
            }
        }
    }
    __syncthreads();
    const int TGS = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
    const int RL = RED_THREADS / TGS;
    const int tg = threadIdx.x % TGS, rl = threadIdx.x / TGS;
    const int c0 = blockIdx.x * 64 + tg * 8;
    if (c0 >= C) return;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = s_sc[tg * 8 + j]; sh[j] = s_sh[tg * 8 + j]; }
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long r = r0 + rl; r < r1; r += RL) {
        float v[8];
        V8<T>::ld(x + r * (long)C + c0, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = v[j] * sc[j] + sh[j];
            v[j] = (act == ACT_SWISH) ? swishf_(p) : p;
        }
        V8<T>::st(y + r * (long)C + c0, v);
    }
}

extern "C" int nvae_bn_apply_fin(int dtype, const void* x, void* y, long rows, int C, const float* partials, int S,
                                 const float* gamma, const float* beta, float* rm, float* rv, float momentum,
                                 float eps, float* scale, float* shift, float* mean, float* invstd, int act,
                                 void* stream) {
    if (int e = check_c("bn_apply_fin", C)) return e;
    NVAE_REQUIRE(rows > 0 && S > 0 && aligned16(x) && aligned16(y) && partials && gamma && beta && rm && rv && scale &&
                 shift && mean && invstd, "bn_apply_fin: bad rows/alignment/null argument");
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "bn_apply_fin: act %d unsupported", act);
    BnFinArgs f{};
    f.inv_n = 1.0f / (float)rows; f.gamma = gamma; f.beta = beta; f.rm = rm; f.rv = rv; f.momentum = momentum;
    f.eps = eps; f.scale = scale; f.shift = shift; f.mean = mean; f.invstd = invstd;
    int S2 = nvae_reduce_splits(rows, C) * 2;            // one row per thread: these passes are latency-bound
    {
        const int tgs = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
        const long max_s = rows / (RED_THREADS / tgs);
        if (S2 > max_s) S2 = (int)max_s;
        if (S2 < 1) S2 = 1;
    }
    const long rpb = (rows + S2 - 1) / S2;
    dim3 grid((C + 63) / 64, S2);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bn_apply_fin<T>), grid, RED_THREADS, 0, (hipStream_t)stream, (const T*)x, (T*)y, rows, C, (int)rpb, partials, S, f, act);)
    NVAE_LAUNCH_CHECK("bn_apply_fin");
    return NVAE_OK;
}

template <typename T>
__global__ __launch_bounds__(RED_THREADS) void k_bn_bwd_apply_fin(
    const T* __restrict__ x, const T* __restrict__ dy, T* dx, long rows, int C, int rows_per_block,
    const float* __restrict__ partials, int S, BnFinArgs a, const float* __restrict__ shift, int act, int acc) {
    __shared__ float s_sc[64], s_sh[64], s_k0[64], s_k1[64];
    {
        int c; double s1, s2;
        if (bn_slab_sum64_t<sizeof(T) == 4>(partials, S, C, blockIdx.x * 64, c, s1, s2)) {
            float dg, k0, k1;
            bn_bwd_coefs(a, c, s1, s2, dg, k0, k1);
            s_sc[c & 63] = a.scale[c]; s_sh[c & 63] = shift[c]; s_k0[c & 63] = k0; s_k1[c & 63] = k1;
            if (blockIdx.y == 0) { a.dgamma[c] += dg; a.dbeta[c] += (float)s1; }
        }
    }
    __syncthreads();
    const int TGS = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
    const int RL = RED_THREADS / TGS;
    const int tg = threadIdx.x % TGS, rl = threadIdx.x / TGS;
    const int c0 = blockIdx.x * 64 + tg * 8;
    if (c0 >= C) return;
    float sc[8], sh[8], k0[8], k1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = s_sc[tg * 8 + j]; sh[j] = s_sh[tg * 8 + j]; k0[j] = s_k0[tg * 8 + j]; k1[j] = s_k1[tg * 8 + j];
    }
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block;
    if (r1 > rows) r1 = rows;
    for (long r = r0 + rl; r < r1; r += RL) {
        const long off = r * (long)C + c0;
        float v[8], g[8], o[8];
        V8<T>::ld(x + off, v);
        V8<T>::ld(dy + off, g);
        if (acc) V8<T>::ld(dx + off, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float dpre = g[j];
            if (act == ACT_SWISH) dpre *= dswishf_(v[j] * sc[j] + sh[j]);
            o[j] = (acc ? o[j] : 0.f) + sc[j] * dpre + k1[j] * v[j] + k0[j];
        }
        V8<T>::st(dx + off, o);
    }
}

extern "C" int nvae_bn_bwd_apply_fin(int dtype, const void* x, const void* dy, void* dx, long rows, int C,
                                     const float* partials, int S, const float* scale, const float* shift,
                                     const float* mean, const float* invstd, float* dgamma, float* dbeta, int act,
                                     int frozen, int accumulate, void* stream) {
    if (int e = check_c("bn_bwd_apply_fin", C)) return e;
    NVAE_REQUIRE(rows > 0 && S > 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && partials && scale && shift &&
                 mean && invstd && dgamma && dbeta, "bn_bwd_apply_fin: bad rows/alignment/null argument");
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "bn_bwd_apply_fin: act %d unsupported", act);
    BnFinArgs f{};
    f.inv_n = 1.0f / (float)rows; f.scale = (float*)scale; f.mean = (float*)mean; f.invstd = (float*)invstd;
    f.dgamma = dgamma; f.dbeta = dbeta; f.frozen = frozen;
    int S2 = nvae_reduce_splits(rows, C) * 2;            // one row per thread: these passes are latency-bound
    {
        const int tgs = C >= 64 ? 8 : (C > 16 ? (C > 32 ? 8 : 4) : (C > 8 ? 2 : 1));
        const long max_s = rows / (RED_THREADS / tgs);
        if (S2 > max_s) S2 = (int)max_s;
        if (S2 < 1) S2 = 1;
    }
    const long rpb = (rows + S2 - 1) / S2;
    dim3 grid((C + 63) / 64, S2);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bn_bwd_apply_fin<T>), grid, RED_THREADS, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, (T*)dx, rows, C, (int)rpb, partials, S, f, shift, act, accumulate);)
    NVAE_LAUNCH_CHECK("bn_bwd_apply_fin");
    return NVAE_OK;
}
