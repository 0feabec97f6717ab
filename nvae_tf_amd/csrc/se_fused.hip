// Squeeze-Excitation + residual as ONE launch per direction, with the BatchNorm in front of it folded in.
//
// The residual cells end in  y = ss*skip + bs * SE(BN(x))  (decoder.py:135-147, postprocess.py:84-88; the
// encoder / preprocess cells have no BatchNorm there: encoder.py:99-107, preprocess.py:100-107).  SE is
// per image: global average pool -> Dense(max(C/16,4)) -> ReLU -> Dense(C) -> sigmoid -> multiply
// (common.py:110-142).  Round 1 ran this as bn_apply + se_pool_gate + se_apply_stats (forward) and
// se_reduce_gate_bwd + se_bwd_apply_bn (backward): five launches of 6-11 us on 1-8 MB tensors, bound by
// launch latency.  Here a workgroup owns whole images (all channels), so the pooling, the two FC layers,
// the gate, the residual add AND the BatchNorm statistics of y (the next cell starts with a BatchNorm) are
// one pass; the BatchNorm in front is applied on the fly from its coefficient table (its output is never
// materialised; the average pool of an affine map is the affine map of the pool).
//
// Thread layout: CG = C/8 channel groups (power of two <= 256), RL = 256/CG row lanes; thread (rl, tg)
// owns channels [8 tg, 8 tg + 8) of pixels rl, rl + RL, ...: 16-B loads, whole 128-B lines per wave.
#include "common.h"
#include "bn_fin.h"

#define SEF_MAX_C 2048
#define SEF_MAX_H 128

__device__ __forceinline__ void sef_ld8(const float* p, float (&v)[8]) {
    const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// cross-row-lane reduction of NQ per-thread 8-vectors: part is [NQ][RL][C] = NQ * 2048 floats
template <int NQ>
__device__ __forceinline__ void sef_scatter(float* part, const float (&a)[NQ][8], int rl, int tg, int C) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) part[q * 2048 + rl * C + tg * 8 + j] = a[q][j];
}
__device__ __forceinline__ float sef_gather(const float* part, int q, int c, int C, int RL) {
    float v = 0.f;
    for (int k = 0; k < RL; ++k) v += part[q * 2048 + k * C + c];
    return v;
}

template <typename T>
__global__ __launch_bounds__(256) void k_se_fused_fwd(
    const T* __restrict__ x, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift,
    const T* __restrict__ skip, T* __restrict__ y, int B, int HW, int C, int Hd, int imgs,
    const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float ss, float bs, float* __restrict__ pooled_sum,
    float* __restrict__ gate_out, float* __restrict__ hidden_out, float* stats, BnFinArgs fin) {
    __shared__ float p[SEF_MAX_C];
    __shared__ float part[2 * 2048];
    __shared__ float hd[SEF_MAX_H];
    const int CG = C >> 3, RL = 256 / CG;
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_hw = 1.0f / (float)HW;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = bn_scale ? bn_scale[tg * 8 + j] : 1.f; sh[j] = bn_scale ? bn_shift[tg * 8 + j] : 0.f; }
    float st[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[0][j] = 0.f; st[1][j] = 0.f; }
    for (int im = 0; im < imgs; ++im) {
        const long b = (long)blockIdx.x * imgs + im;
        if (b >= B) break;
        const T* xb = x + b * HW * C;
        const T* kb = skip + b * HW * C;
        T* yb = y + b * HW * C;
        // ---- pool (raw sums; the BatchNorm is affine per channel)
        float a[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        for (int r = rl; r < HW; r += RL) {
            float v[8];
            V8<T>::ld(xb + (long)r * C + tg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[0][j] += v[j];
        }
        __syncthreads();                       // previous image's readers of p / part / hd are done
        sef_scatter<1>(part, a, rl, tg, C);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float v = sef_gather(part, 0, c, C, RL);
            if (bn_scale) v = bn_scale[c] * v + (float)HW * bn_shift[c];
            pooled_sum[b * C + c] = v;
            p[c] = v * inv_hw;
        }
        __syncthreads();
        // ---- hidden = relu(p W1 + b1); gate = sigmoid(hidden W2 + b2)
        for (int h = wave; h < Hd; h += 4) {
            float acc = 0.f;
            for (int c = lane; c < C; c += 64) acc += p[c] * w1[(long)c * Hd + h];
            acc = wave_sum(acc);
            if (lane == 0) {
                const float v = fmaxf(acc + b1[h], 0.f);
                hd[h] = v;
                hidden_out[b * Hd + h] = v;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = b2[c];
            for (int h = 0; h < Hd; ++h) acc += hd[h] * w2[(long)h * C + c];
            const float g = sigmoidf_(acc);
            gate_out[b * C + c] = g;
            p[c] = g;
        }
        __syncthreads();
        // ---- y = ss*skip + bs * BN(x) * gate (+ statistics of y)
        float g8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) g8[j] = p[tg * 8 + j] * bs;
        for (int r = rl; r < HW; r += RL) {
            float v[8], k[8];
            V8<T>::ld(xb + (long)r * C + tg * 8, v);
            V8<T>::ld(kb + (long)r * C + tg * 8, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = ss * k[j] + (v[j] * sc[j] + sh[j]) * g8[j];
                st[0][j] += v[j]; st[1][j] += v[j] * v[j];
            }
            V8<T>::st(yb + (long)r * C + tg * 8, v);
        }
    }
    if (!stats) return;
    __syncthreads();
    sef_scatter<2>(part, st, rl, tg, C);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        bn_store_partial(stats + ((long)blockIdx.x * 2) * C + c, sef_gather(part, 0, c, C, RL));
        bn_store_partial(stats + ((long)blockIdx.x * 2 + 1) * C + c, sef_gather(part, 1, c, C, RL));
    }
    if (fin.counter == nullptr) return;
    if (!bn_last_arriver(fin.counter, (int)gridDim.x)) return;
    bn_fin_fwd(fin, stats, (int)gridDim.x, C, 0, (C + 63) / 64);
}

// images per workgroup such that the statistics slab has at most MAX rows
static inline int sef_imgs(int B, int max_rows) { return (B + max_rows - 1) / max_rows; }

extern "C" int nvae_se_fused_rows(int B) { return B <= 0 ? 0 : cdiv(B, sef_imgs(B, 128)); }

static int sef_check(const char* who, int B, int HW, int C, int Hd) {
    NVAE_REQUIRE(B > 0 && HW > 0 && C >= 8 && C <= SEF_MAX_C && (C & (C - 1)) == 0 && Hd > 0 && Hd <= SEF_MAX_H,
                 "%s: C=%d must be a power of two in [8, %d], Hd=%d in [1, %d]", who, C, SEF_MAX_C, Hd, SEF_MAX_H);
    return NVAE_OK;
}

extern "C" int nvae_se_fused_fwd(int dtype, const void* x, const float* bn_scale, const float* bn_shift,
                                 const void* skip, void* y, int B, int HW, int C, int Hd, const float* w1,
                                 const float* b1, const float* w2, const float* b2, float skip_scale,
                                 float branch_scale, float* pooled_sum, float* gate, float* hidden, float* stats,
                                 const NvaeBnFin* fin, void* stream) {
    if (int e = sef_check("se_fused_fwd", B, HW, C, Hd)) return e;
    NVAE_REQUIRE(aligned16(x) && aligned16(skip) && aligned16(y) && w1 && b1 && w2 && b2 && pooled_sum && gate && hidden,
                 "se_fused_fwd: alignment / NULL argument");
    NVAE_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "se_fused_fwd: scale and shift go together");
    NVAE_REQUIRE(!fin || stats, "se_fused_fwd: an in-kernel finalize needs the statistics slab");
    BnFinArgs f{};
    if (fin) {
        NVAE_REQUIRE(fin->counter && fin->gamma && fin->beta && fin->rm && fin->rv && fin->scale && fin->shift &&
                     fin->mean && fin->invstd, "se_fused_fwd: NULL field in NvaeBnFin");
        f.counter = fin->counter; f.inv_n = 1.0f / (float)((long)B * HW); f.gamma = fin->gamma; f.beta = fin->beta;
        f.rm = fin->rm; f.rv = fin->rv; f.momentum = fin->momentum; f.eps = fin->eps; f.scale = fin->scale;
        f.shift = fin->shift; f.mean = fin->mean; f.invstd = fin->invstd;
    }
    const int imgs = sef_imgs(B, 128);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_fused_fwd<T>), cdiv(B, imgs), 256, 0, (hipStream_t)stream, (const T*)x,
                                         bn_scale, bn_shift, (const T*)skip, (T*)y, B, HW, C, Hd, imgs, w1, b1, w2, b2,
                                         skip_scale, branch_scale, pooled_sum, gate, hidden, stats, f);)
    NVAE_LAUNCH_CHECK("se_fused_fwd");
    return NVAE_OK;
}

// Backward of the same block.  With xs = BN(x) (or x itself), r[c] = sum_hw xs*dy:
//   dpre2 = bs*r*g*(1-g);  dpre1[h] = relu'(hidden[h]) * sum_c W2[h,c] dpre2[c];  dpool[c] = sum_h W1[c,h] dpre1[h] / HW
//   dxs = bs*dy*gate + dpool;   dskip (+)= ss*dy
// dpre2 | dpre1 go to `scratch` ([B*C] | [B*Hd]) for the FC parameter gradients (nvae_se_wgrad_batched).
// partials != NULL: xs = act(BN(x)) had no other consumer, so dxs is final and the BatchNorm-backward sums
// (sum dpre, sum dpre*x with dpre = dxs * act'(scale*x + shift)) are reduced here: partials[rows][2][C].
template <typename T>
__global__ __launch_bounds__(256) void k_se_fused_bwd(
    const T* __restrict__ x, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ hidden, T* dx, T* dskip,
    int B, int HW, int C, int Hd, int imgs, const float* __restrict__ w1, const float* __restrict__ w2, float ss,
    float bs, int acc_dx, int acc_dskip, float* __restrict__ scratch, float* partials, BnFinArgs fin) {
    __shared__ float d2[SEF_MAX_C];
    __shared__ float part[2 * 2048];
    __shared__ float d1[SEF_MAX_H];
    const int CG = C >> 3, RL = 256 / CG;
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_hw = 1.0f / (float)HW;
    float* dpre2_out = scratch;
    float* dpre1_out = scratch + (long)B * C;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = bn_scale ? bn_scale[tg * 8 + j] : 1.f; sh[j] = bn_scale ? bn_shift[tg * 8 + j] : 0.f; }
    float st[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[0][j] = 0.f; st[1][j] = 0.f; }
    for (int im = 0; im < imgs; ++im) {
        const long b = (long)blockIdx.x * imgs + im;
        if (b >= B) break;
        const T* xb = x + b * HW * C;
        const T* gb = dy + b * HW * C;
        // ---- r = sum xs*dy  (xs = act(scale*x + shift); for act = none: scale * sum x*dy + shift * sum dy)
        float a[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        for (int r = rl; r < HW; r += RL) {
            float v[8], g[8];
            V8<T>::ld(xb + (long)r * C + tg * 8, v);
            V8<T>::ld(gb + (long)r * C + tg * 8, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float xs = v[j] * sc[j] + sh[j];
                if (act == ACT_SWISH) xs = swishf_(xs);
                a[0][j] += xs * g[j];
            }
        }
        __syncthreads();
        sef_scatter<1>(part, a, rl, tg, C);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            const float r = sef_gather(part, 0, c, C, RL);
            const float g = gate[b * C + c];
            const float d = bs * r * g * (1.f - g);
            d2[c] = d;
            dpre2_out[b * C + c] = d;
        }
        __syncthreads();
        for (int h = wave; h < Hd; h += 4) {
            float acc = 0.f;
            for (int c = lane; c < C; c += 64) acc += w2[(long)h * C + c] * d2[c];
            acc = wave_sum(acc);
            if (lane == 0) {
                const float d = hidden[b * Hd + h] > 0.f ? acc : 0.f;
                d1[h] = d;
                dpre1_out[b * Hd + h] = d;
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float acc = 0.f;
            for (int h = 0; h < Hd; ++h) acc += w1[(long)c * Hd + h] * d1[h];
            d2[c] = acc * inv_hw;              // dpool
        }
        __syncthreads();
        float g8[8], dp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { g8[j] = gate[b * C + tg * 8 + j] * bs; dp[j] = d2[tg * 8 + j]; }
        T* dxb = dx + b * HW * C;
        T* dkb = dskip ? dskip + b * HW * C : nullptr;
        for (int r = rl; r < HW; r += RL) {
            const long off = (long)r * C + tg * 8;
            float g[8], o[8], k[8], v[8];
            V8<T>::ld(gb + off, g);
            if (acc_dx) V8<T>::ld(dxb + off, o);
            if (dkb && acc_dskip) V8<T>::ld(dkb + off, k);
            if (partials) V8<T>::ld(xb + off, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = g[j] * g8[j] + dp[j];
                o[j] = (acc_dx ? o[j] : 0.f) + d;
                k[j] = ((dkb && acc_dskip) ? k[j] : 0.f) + ss * g[j];
                if (partials) {
                    float dpre = d;
                    if (act == ACT_SWISH) dpre *= dswishf_(v[j] * sc[j] + sh[j]);
                    st[0][j] += dpre; st[1][j] += dpre * v[j];
                }
            }
            V8<T>::st(dxb + off, o);
            if (dkb) V8<T>::st(dkb + off, k);
        }
    }
    if (!partials) return;
    __syncthreads();
    sef_scatter<2>(part, st, rl, tg, C);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        bn_store_partial(partials + ((long)blockIdx.x * 2) * C + c, sef_gather(part, 0, c, C, RL));
        bn_store_partial(partials + ((long)blockIdx.x * 2 + 1) * C + c, sef_gather(part, 1, c, C, RL));
    }
    if (fin.counter == nullptr) return;
    if (!bn_last_arriver(fin.counter, (int)gridDim.x)) return;
    bn_fin_bwd(fin, partials, (int)gridDim.x, C, 0, (C + 63) / 64);
}

extern "C" int nvae_se_fused_bwd(int dtype, const void* x, const float* bn_scale, const float* bn_shift, int act,
                                 const void* dy, const float* gate, const float* hidden, void* dx, void* dskip, int B,
                                 int HW, int C, int Hd, const float* w1, const float* w2, float skip_scale,
                                 float branch_scale, int acc_dx, int acc_dskip, float* scratch, float* partials,
                                 void* stream) {
    if (int e = sef_check("se_fused_bwd", B, HW, C, Hd)) return e;
    NVAE_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(dskip) && gate && hidden && w1 && w2 && scratch,
                 "se_fused_bwd: alignment / NULL argument");
    NVAE_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "se_fused_bwd: scale and shift go together");
    NVAE_REQUIRE(act == ACT_NONE || (act == ACT_SWISH && bn_scale), "se_fused_bwd: act %d unsupported", act);
    NVAE_REQUIRE(dskip || !acc_dskip, "se_fused_bwd: acc_dskip without dskip");
    NVAE_REQUIRE(!partials || (bn_scale && !acc_dx), "se_fused_bwd: BatchNorm sums need the coefficients and a final dx");
    BnFinArgs f{};
    const int imgs = sef_imgs(B, 128);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_fused_bwd<T>), cdiv(B, imgs), 256, 0, (hipStream_t)stream, (const T*)x,
                                         bn_scale, bn_shift, act, (const T*)dy, gate, hidden, (T*)dx, (T*)dskip, B, HW, C,
                                         Hd, imgs, w1, w2, skip_scale, branch_scale, acc_dx, dskip ? acc_dskip : 0,
                                         scratch, partials, f);)
    NVAE_LAUNCH_CHECK("se_fused_bwd");
    return NVAE_OK;
}
