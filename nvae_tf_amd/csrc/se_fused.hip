// Squeeze-Excitation + residual as ONE launch per direction, with the BatchNorm in front of it folded in.
//
// The residual cells end in  y = ss*skip + bs * SE(BN(x))  (decoder.py:135-147, postprocess.py:84-88; the
// encoder / preprocess cells have no BatchNorm there: encoder.py:99-107, preprocess.py:100-107).  SE is
// per image: global average pool -> Dense(max(C/16,4)) -> ReLU -> Dense(C) -> sigmoid -> multiply
// (common.py:110-142).  Round 1 ran this as bn_apply + se_pool_gate + se_apply_stats (forward) and
// se_reduce_gate_bwd + se_bwd_apply_bn (backward): five launches of 6-11 us on 1-8 MB tensors, bound by
// launch latency.  Here a workgroup owns whole images (all channels), so the pooling, the two FC layers,
// the gate, the residual add AND the BatchNorm statistics of y (the next cell starts with a BatchNorm) are
// one pass; the BatchNorm in front is applied on the fly - from its coefficient table, or straight from the
// statistics slab its producer accumulated (bn_fin.h BnFromSlab: no finalize launch either); its output is
// never materialised (the average pool of an affine map is the affine map of the pool).
//
// Latency: these tensors are 1-8 MB, so the kernel is as long as its chain of dependent memory round trips.
// Everything it needs (the image's x and skip chunks, both FC matrices, the slab) is requested in the first
// instructions, so there is ONE global round trip before the stores; images of up to NCH * RL pixels stay in
// registers between the pooling pass and the apply pass.
//
// Thread layout: CG = C/8 channel groups (power of two <= 256), RL = 256/CG row lanes; thread (rl, tg)
// owns channels [8 tg, 8 tg + 8) of pixels rl, rl + RL, ...: 16-B loads, whole 128-B lines per wave.
#include "common.h"
#include "bn_fin.h"

#define SEF_MAX_C 2048
#define SEF_MAX_H 128
#define SEF_W_LDS 4096          // FC matrices of up to this many floats each are staged in LDS

// cross-row-lane reduction of NQ per-thread 8-vectors: part is [NQ][RL][C] = NQ * 2048 elements
template <int NQ, typename E>
__device__ __forceinline__ void sef_scatter(E* part, const E (&a)[NQ][8], int rl, int tg, int C) {
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) part[q * 2048 + rl * C + tg * 8 + j] = a[q][j];
}
template <typename E>
__device__ __forceinline__ E sef_gather(const E* part, int q, int c, int C, int RL) {
    E v = 0;
    for (int k = 0; k < RL; ++k) v += part[q * 2048 + k * C + c];
    return v;
}

struct SefOut {                 // statistics of y for the BatchNorm that follows: rows of a ZEROED slab
    void* stats; int rows;       // element type: StatT<T> (bn_fin.h "statistics precision")
};

// NCH > 0: HW == NCH * RL and the image's chunks stay in registers; NCH == 0: any HW, second pass re-reads
template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_se_fused_fwd(
    const T* __restrict__ x, BnFromSlab bn, const T* __restrict__ skip, T* __restrict__ y, int B, int HW, int C,
    int Hd, int imgs, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float ss, float bs, float* __restrict__ pooled_sum,
    float* __restrict__ gate_out, float* __restrict__ hidden_out, SefOut so) {
    typedef typename StatT<T>::type ST;
    __shared__ float p[SEF_MAX_C];
    __shared__ ST part_s[2 * 2048];
    float* part = (float*)part_s;                 // (the pooling pass uses the first 2048 floats of it)
    __shared__ float hd[SEF_MAX_H];
    __shared__ float s_w1[SEF_W_LDS], s_w2[SEF_W_LDS];
    __shared__ float s_sc[SEF_MAX_C], s_sh[SEF_MAX_C];
    const int CG = C >> 3, RL = 256 / CG;
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_hw = 1.0f / (float)HW;
    const bool has_bn = bn.slab != nullptr || bn.scale != nullptr;
    const bool w_lds = C * Hd <= SEF_W_LDS;
    constexpr int NR = NCH > 0 ? NCH : 1;
    float xv[NR][8], kv[NR][8];
    // ---- everything this workgroup will need, requested up front
    long b = (long)blockIdx.x * imgs;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            V8<T>::ld(x + (b * HW + rl + i * RL) * C + tg * 8, xv[i]);
            V8<T>::ld(skip + (b * HW + rl + i * RL) * C + tg * 8, kv[i]);
        }
    }
    if (w_lds)
        for (int i = threadIdx.x; i < C * Hd; i += 256) { s_w1[i] = w1[i]; s_w2[i] = w2[i]; }
    if (has_bn)
        for (int c = threadIdx.x; c < C; c += 256) {
            float sc, sh;
            bn_coef<sizeof(T) == 4>(bn, C, c, blockIdx.x == 0, sc, sh);
            s_sc[c] = sc; s_sh[c] = sh;
        }
    ST st[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[0][j] = 0; st[1][j] = 0; }
    for (int im = 0; im < imgs; ++im, ++b) {
        if (b >= B) break;
        const T* xb = x + b * HW * C;
        const T* kb = skip + b * HW * C;
        T* yb = y + b * HW * C;
        if (NCH > 0 && im > 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                V8<T>::ld(xb + (long)(rl + i * RL) * C + tg * 8, xv[i]);
                V8<T>::ld(kb + (long)(rl + i * RL) * C + tg * 8, kv[i]);
            }
        }
        // ---- pool (raw sums; the BatchNorm is affine per channel)
        float a[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        if constexpr (NCH > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) a[0][j] += xv[i][j];
        } else {
            for (int r = rl; r < HW; r += RL) {
                float v[8];
                V8<T>::ld(xb + (long)r * C + tg * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) a[0][j] += v[j];
            }
        }
        __syncthreads();                       // previous image's readers of p / part / hd are done; tables are written
        sef_scatter<1, float>(part, a, rl, tg, C);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float v = sef_gather<float>(part, 0, c, C, RL);
            if (has_bn) v = s_sc[c] * v + (float)HW * s_sh[c];
            pooled_sum[b * C + c] = v;
            p[c] = v * inv_hw;
        }
        __syncthreads();
        // ---- hidden = relu(p W1 + b1); gate = sigmoid(hidden W2 + b2)
        const float* W1 = w_lds ? s_w1 : w1;
        const float* W2 = w_lds ? s_w2 : w2;
        for (int h = wave; h < Hd; h += 4) {
            float acc = 0.f;
            for (int c = lane; c < C; c += 64) acc += p[c] * W1[(long)c * Hd + h];
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
            for (int h = 0; h < Hd; ++h) acc += hd[h] * W2[(long)h * C + c];
            const float g = sigmoidf_(acc);
            gate_out[b * C + c] = g;
            p[c] = g;
        }
        __syncthreads();
        // ---- y = ss*skip + bs * BN(x) * gate (+ statistics of y)
        float g8[8], sc[8], sh[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g8[j] = p[tg * 8 + j] * bs;
            sc[j] = has_bn ? s_sc[tg * 8 + j] : 1.f;
            sh[j] = has_bn ? s_sh[tg * 8 + j] : 0.f;
        }
        if constexpr (NCH > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = ss * kv[i][j] + (xv[i][j] * sc[j] + sh[j]) * g8[j];
                    st[0][j] += (ST)v[j]; st[1][j] += (ST)v[j] * (ST)v[j];
                }
                V8<T>::st(yb + (long)(rl + i * RL) * C + tg * 8, v);
            }
        } else {
            for (int r = rl; r < HW; r += RL) {
                float v[8], k[8];
                V8<T>::ld(xb + (long)r * C + tg * 8, v);
                V8<T>::ld(kb + (long)r * C + tg * 8, k);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[j] = ss * k[j] + (v[j] * sc[j] + sh[j]) * g8[j];
                    st[0][j] += (ST)v[j]; st[1][j] += (ST)v[j] * (ST)v[j];
                }
                V8<T>::st(yb + (long)r * C + tg * 8, v);
            }
        }
    }
    if (!so.stats) return;
    __syncthreads();
    sef_scatter<2, ST>(part_s, st, rl, tg, C);
    __syncthreads();
    const int row = blockIdx.x % so.rows;          // <= 64 adders per address (see conv_gemm.hip)
    ST* slab = (ST*)so.stats;
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(slab + ((long)row * 2) * C + c, sef_gather<ST>(part_s, 0, c, C, RL));
        atomicAdd(slab + ((long)row * 2 + 1) * C + c, sef_gather<ST>(part_s, 1, c, C, RL));
    }
}

// =========================================================================================
// Image-split variants (round 3): with <= 64 images per rank (BASELINE.json configs[3], [4]: batch 64 / 32) a launch of
// one workgroup per image leaves most of the chip idle (C5: 32 workgroups for 256 CUs, 52-64 us per launch, 19 of
// the step's 93 ms).  Here S workgroups share an image (S * B <= 256, all co-resident): each pools its slice of the
// pixels, publishes the partial vector (write-through stores) and takes a ticket on the image's counter; the last
// arriver sums the partials in slice order (deterministic), runs FC1 -> ReLU -> FC2 -> sigmoid (backward: dpre2,
// dpre1, dpool) and publishes the gate (dpool); the other slices wait for the counter to say so, read it with sc1
// loads and apply it to their slice - the hand-off of bn_fin.h / cdna guide 16 in its counter form.  Counter
// protocol per image (zero at rest): S arrivals -> the last arriver adds S when the vector is out ("ready" = 2S) ->
// each reader adds 1 when it has the vector -> the add that returns 3S - 2 resets the counter.  A waiting workgroup
// only ever waits for workgroups of its own launch that need nothing from it, and the grid fits the chip, so every
// wave reaches its exit; the poll is bounded all the same (a timed-out wait poisons the output with NaNs).
// =========================================================================================
struct SefSplit {
    int S;                      // slices per image (>= 2)
    float* part;                // [B][S][C] partial pooled sums (backward: partial r)
    float* vec;                 // [B][C] backward only: dpool
    int* counter;               // [B]
};

__device__ __forceinline__ void sef_store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float sef_load_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// all threads call; returns the ticket (uniform) after this workgroup's partial stores are out
__device__ __forceinline__ int sef_arrive(int* counter, int* s_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) *s_flag = atomicAdd(counter, 1);
    __syncthreads();
    return *s_flag;
}
// last arriver, after its vector stores: publish
__device__ __forceinline__ void sef_publish(int* counter, int S) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(counter, S);
}
// reader: wait until the vector is out; false after ~2 s (never observed: see above)
__device__ __forceinline__ bool sef_wait(int* counter, int S, int* s_flag) {
    if (threadIdx.x == 0) {
        int ok = 0;
        for (long it = 0; it < (1L << 24); ++it) {
            if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 2 * S) { ok = 1; break; }
            __builtin_amdgcn_s_sleep(8);
        }
        *s_flag = ok;
    }
    __syncthreads();
    return *s_flag != 0;
}
// reader, after it has the vector in LDS
__device__ __forceinline__ void sef_release(int* counter, int S) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(counter, 1);
        if (old == 3 * S - 2) atomicExch(counter, 0);
    }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_se_split_fwd(
    const T* __restrict__ x, BnFromSlab bn, const T* __restrict__ skip, T* __restrict__ y, int B, int HW, int C,
    int Hd, const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
    const float* __restrict__ b2, float ss, float bs, float* __restrict__ pooled_sum,
    float* __restrict__ gate_out, float* __restrict__ hidden_out, SefOut so, SefSplit sp) {
    typedef typename StatT<T>::type ST;
    __shared__ float p[SEF_MAX_C];
    __shared__ ST part_s[2 * 2048];
    float* part = (float*)part_s;
    __shared__ float hd[SEF_MAX_H];
    __shared__ float s_sc[SEF_MAX_C], s_sh[SEF_MAX_C];
    __shared__ int s_flag;
    const int CG = C >> 3, RL = 256 / CG;
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_hw = 1.0f / (float)HW;
    const bool has_bn = bn.slab != nullptr || bn.scale != nullptr;
    const int S = sp.S;
    const long b = blockIdx.x / S;
    const int sl = blockIdx.x - (int)b * S;
    const int P = HW / S, p0 = sl * P;              // this workgroup's pixels [p0, p0 + P)
    constexpr int NR = NCH > 0 ? NCH : 1;
    float xv[NR][8], kv[NR][8];
    const T* xb = x + (b * HW + p0) * C;
    const T* kb = skip + (b * HW + p0) * C;
    T* yb = y + (b * HW + p0) * C;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            V8<T>::ld(xb + (long)(rl + i * RL) * C + tg * 8, xv[i]);
            V8<T>::ld(kb + (long)(rl + i * RL) * C + tg * 8, kv[i]);
        }
    }
    if (has_bn)
        for (int c = threadIdx.x; c < C; c += 256) {
            float sc, sh;
            bn_coef<sizeof(T) == 4>(bn, C, c, blockIdx.x == 0, sc, sh);
            s_sc[c] = sc; s_sh[c] = sh;
        }
    // ---- partial pool of the slice (raw sums)
    float a[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[0][j] += xv[i][j];
    } else {
        for (int r = rl; r < P; r += RL) {
            float v[8];
            V8<T>::ld(xb + (long)r * C + tg * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) a[0][j] += v[j];
        }
    }
    sef_scatter<1, float>(part, a, rl, tg, C);
    __syncthreads();
    float* mine = sp.part + (b * S + sl) * C;
    for (int c = threadIdx.x; c < C; c += 256) sef_store_sc1(mine + c, sef_gather<float>(part, 0, c, C, RL));
    int* counter = sp.counter + b;
    const int ticket = sef_arrive(counter, &s_flag);
    if (ticket == S - 1) {
        // ---- last arriver: pooled vector in slice order, both FC layers, publish the gate
        for (int c = threadIdx.x; c < C; c += 256) {
            float v = 0.f;
            for (int q = 0; q < S; ++q) v += sef_load_sc1(sp.part + (b * S + q) * C + c);
            if (has_bn) v = s_sc[c] * v + (float)HW * s_sh[c];
            pooled_sum[b * C + c] = v;
            p[c] = v * inv_hw;
        }
        __syncthreads();
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
            sef_store_sc1(gate_out + b * C + c, g);
            p[c] = g;
        }
        sef_publish(counter, S);
        __syncthreads();
    } else {
        const bool ok = sef_wait(counter, S, &s_flag);
        for (int c = threadIdx.x; c < C; c += 256) p[c] = ok ? sef_load_sc1(gate_out + b * C + c) : __builtin_nanf("");
        sef_release(counter, S);
        __syncthreads();
    }
    // ---- y = ss*skip + bs * BN(x) * gate on the slice (+ statistics of y)
    ST st[2][8];
    float g8[8], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        st[0][j] = 0; st[1][j] = 0;
        g8[j] = p[tg * 8 + j] * bs;
        sc[j] = has_bn ? s_sc[tg * 8 + j] : 1.f;
        sh[j] = has_bn ? s_sh[tg * 8 + j] : 0.f;
    }
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = ss * kv[i][j] + (xv[i][j] * sc[j] + sh[j]) * g8[j];
                st[0][j] += (ST)v[j]; st[1][j] += (ST)v[j] * (ST)v[j];
            }
            V8<T>::st(yb + (long)(rl + i * RL) * C + tg * 8, v);
        }
    } else {
        for (int r = rl; r < P; r += RL) {
            float v[8], k[8];
            V8<T>::ld(xb + (long)r * C + tg * 8, v);
            V8<T>::ld(kb + (long)r * C + tg * 8, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = ss * k[j] + (v[j] * sc[j] + sh[j]) * g8[j];
                st[0][j] += (ST)v[j]; st[1][j] += (ST)v[j] * (ST)v[j];
            }
            V8<T>::st(yb + (long)r * C + tg * 8, v);
        }
    }
    if (!so.stats) return;
    __syncthreads();
    sef_scatter<2, ST>(part_s, st, rl, tg, C);
    __syncthreads();
    const int row = blockIdx.x % so.rows;
    ST* slab = (ST*)so.stats;
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(slab + ((long)row * 2) * C + c, sef_gather<ST>(part_s, 0, c, C, RL));
        atomicAdd(slab + ((long)row * 2 + 1) * C + c, sef_gather<ST>(part_s, 1, c, C, RL));
    }
}

template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_se_split_bwd(
    const T* __restrict__ x, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ hidden, T* dx, T* dskip,
    int B, int HW, int C, int Hd, const float* __restrict__ w1, const float* __restrict__ w2, float ss,
    float bs, int acc_dx, int acc_dskip, float* __restrict__ scratch, SefOut so, SefSplit sp) {
    typedef typename StatT<T>::type ST;
    __shared__ float d2[SEF_MAX_C];
    __shared__ ST part_s[2 * 2048];
    float* part = (float*)part_s;
    __shared__ float d1[SEF_MAX_H];
    __shared__ int s_flag;
    const int CG = C >> 3, RL = 256 / CG;
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_hw = 1.0f / (float)HW;
    float* dpre2_out = scratch;
    float* dpre1_out = scratch + (long)B * C;
    ST* partials = (ST*)so.stats;
    const int S = sp.S;
    const long b = blockIdx.x / S;
    const int sl = blockIdx.x - (int)b * S;
    const int P = HW / S, p0 = sl * P;
    constexpr int NR = NCH > 0 ? NCH : 1;
    float xv[NR][8], gv[NR][8];
    const T* xb = x + (b * HW + p0) * C;
    const T* gb = dy + (b * HW + p0) * C;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            V8<T>::ld(xb + (long)(rl + i * RL) * C + tg * 8, xv[i]);
            V8<T>::ld(gb + (long)(rl + i * RL) * C + tg * 8, gv[i]);
        }
    }
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = bn_scale ? bn_scale[tg * 8 + j] : 1.f; sh[j] = bn_scale ? bn_shift[tg * 8 + j] : 0.f; }
    // ---- partial r = sum xs*dy over the slice
    float a[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float xs = xv[i][j] * sc[j] + sh[j];
                if (act == ACT_SWISH) xs = swishf_(xs);
                a[0][j] += xs * gv[i][j];
            }
    } else {
        for (int r = rl; r < P; r += RL) {
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
    }
    sef_scatter<1, float>(part, a, rl, tg, C);
    __syncthreads();
    float* mine = sp.part + (b * S + sl) * C;
    for (int c = threadIdx.x; c < C; c += 256) sef_store_sc1(mine + c, sef_gather<float>(part, 0, c, C, RL));
    int* counter = sp.counter + b;
    const int ticket = sef_arrive(counter, &s_flag);
    float* dpool = sp.vec + b * C;
    if (ticket == S - 1) {
        for (int c = threadIdx.x; c < C; c += 256) {
            float r = 0.f;
            for (int q = 0; q < S; ++q) r += sef_load_sc1(sp.part + (b * S + q) * C + c);
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
            acc *= inv_hw;
            sef_store_sc1(dpool + c, acc);
            d2[c] = acc;
        }
        sef_publish(counter, S);
        __syncthreads();
    } else {
        const bool ok = sef_wait(counter, S, &s_flag);
        for (int c = threadIdx.x; c < C; c += 256) d2[c] = ok ? sef_load_sc1(dpool + c) : __builtin_nanf("");
        sef_release(counter, S);
        __syncthreads();
    }
    ST st[2][8];
    float g8[8], dp[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[0][j] = 0; st[1][j] = 0; g8[j] = gate[b * C + tg * 8 + j] * bs; dp[j] = d2[tg * 8 + j]; }
    T* dxb = dx + (b * HW + p0) * C;
    T* dkb = dskip ? dskip + (b * HW + p0) * C : nullptr;
    auto apply = [&](long off, const float (&g)[8], const float (&v)[8]) {
        float o[8], k[8];
        if (acc_dx) V8<T>::ld(dxb + off, o);
        if (dkb && acc_dskip) V8<T>::ld(dkb + off, k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = g[j] * g8[j] + dp[j];
            o[j] = (acc_dx ? o[j] : 0.f) + d;
            k[j] = ((dkb && acc_dskip) ? k[j] : 0.f) + ss * g[j];
            if (partials) {
                float dpre = d;
                if (act == ACT_SWISH) dpre *= dswishf_(v[j] * sc[j] + sh[j]);
                st[0][j] += (ST)dpre; st[1][j] += (ST)dpre * (ST)v[j];
            }
        }
        V8<T>::st(dxb + off, o);
        if (dkb) V8<T>::st(dkb + off, k);
    };
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) apply((long)(rl + i * RL) * C + tg * 8, gv[i], xv[i]);
    } else {
        for (int r = rl; r < P; r += RL) {
            const long off = (long)r * C + tg * 8;
            float g[8], v[8];
            V8<T>::ld(gb + off, g);
            V8<T>::ld(xb + off, v);
            apply(off, g, v);
        }
    }
    if (!partials) return;
    __syncthreads();
    sef_scatter<2, ST>(part_s, st, rl, tg, C);
    __syncthreads();
    const int row = blockIdx.x % so.rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(partials + ((long)row * 2) * C + c, sef_gather<ST>(part_s, 0, c, C, RL));
        atomicAdd(partials + ((long)row * 2 + 1) * C + c, sef_gather<ST>(part_s, 1, c, C, RL));
    }
}

// images per workgroup: at most 128 workgroups (more would not fill the chip any better at these sizes)
static inline int sef_imgs(int B) { return (B + 127) / 128; }
static inline int sef_wgs(int B) { return cdiv(B, sef_imgs(B)); }

// Workspace of the image-split variants: partial vectors [B][S][C] + [B][C], and per-image counters (ZEROED once by
// the caller; the kernels leave them zero).  Launches that share it must be stream-ordered.  Process-wide.
static float* g_se_ws = nullptr;
static size_t g_se_ws_bytes = 0;
static int* g_se_counters = nullptr;
static int g_se_ncounters = 0;
extern "C" int nvae_se_set_workspace(void* buf, size_t bytes, int* counters, int n_counters) {
    NVAE_REQUIRE((buf && counters && bytes > 0 && n_counters > 0) || (!buf && !counters),
                 "se_set_workspace: buffer and counters must both be given (or both NULL)");
    g_se_ws = (float*)buf; g_se_ws_bytes = buf ? bytes : 0; g_se_counters = counters; g_se_ncounters = buf ? n_counters : 0;
    return NVAE_OK;
}
static int g_se_force_split = -1;      // tuning / test hook: -1 = the launcher's choice, 1 = never split, S = force (if legal)
extern "C" int nvae_se_force_split(int S) { g_se_force_split = S; return NVAE_OK; }

// slices per image: 1 = whole images per workgroup (k_se_fused_*), S >= 2 = k_se_split_*
static int sef_split(int B, int HW, int C) {
    if (!g_se_ws || B > g_se_ncounters || g_se_force_split == 1) return 1;
    const int RL = 256 / (C / 8);
    auto legal = [&](int S) {
        return S >= 2 && S <= 16 && HW % S == 0 && HW / S >= RL && (long)S * B <= 256 &&
               ((size_t)B * S * C + (size_t)B * C) * 4 <= g_se_ws_bytes;
    };
    if (g_se_force_split > 1) return legal(g_se_force_split) ? g_se_force_split : 1;
    // worth it when one workgroup per image leaves most of the chip idle and an image is big enough to pay for the
    // hand-off (measured: profiles/r03_se_split.txt)
    if (B > 64 || (long)HW * C < 32768) return 1;
    // slices of >= 16 K elements (32 KB of 16-bit data): below that the hand-off costs more than the extra workgroups buy
    // (tools/mb_se.py: 16x16x256 at batch 32 runs 24 / 36 us whole, 18 / 24 us in 4 slices, 24 / 26 us in 8)
    int S = 1;
    while (legal(2 * S) && (long)(HW / (2 * S)) * C >= 16384) S *= 2;
    return S;
}

extern "C" int nvae_se_fused_rows(int B, int HW, int C) {
    if (B <= 0) return 0;
    const int S = sef_split(B, HW, C);
    return slab_rows_for(S > 1 ? B * S : sef_wgs(B));
}

static int sef_check(const char* who, int B, int HW, int C, int Hd) {
    NVAE_REQUIRE(B > 0 && HW > 0 && C >= 8 && C <= SEF_MAX_C && (C & (C - 1)) == 0 && Hd > 0 && Hd <= SEF_MAX_H,
                 "%s: C=%d must be a power of two in [8, %d], Hd=%d in [1, %d]", who, C, SEF_MAX_C, Hd, SEF_MAX_H);
    return NVAE_OK;
}

static int sef_bn(const char* who, const NvaeBnIn* in, long rows, BnFromSlab& bn) {
    bn = BnFromSlab{};
    if (!in) return NVAE_OK;
    NVAE_REQUIRE(in->scale && in->shift, "%s: NvaeBnIn needs scale / shift", who);
    bn.scale = in->scale; bn.shift = in->shift; bn.mean = in->mean; bn.invstd = in->invstd;
    if (in->slab) {
        NVAE_REQUIRE(in->rows > 0 && in->gamma && in->beta && in->rm && in->rv && in->mean && in->invstd,
                     "%s: NvaeBnIn with a slab needs rows, gamma, beta, rm, rv, mean, invstd", who);
        bn.slab = in->slab; bn.rows = in->rows; bn.inv_n = 1.0f / (float)rows; bn.eps = in->eps; bn.momentum = in->momentum;
        bn.gamma = in->gamma; bn.beta = in->beta; bn.rm = in->rm; bn.rv = in->rv;
    }
    return NVAE_OK;
}

extern "C" int nvae_se_fused_fwd(int dtype, const void* x, const NvaeBnIn* bn_in, const void* skip, void* y, int B,
                                 int HW, int C, int Hd, const float* w1, const float* b1, const float* w2,
                                 const float* b2, float skip_scale, float branch_scale, float* pooled_sum,
                                 float* gate, float* hidden, float* stats, void* stream) {
    if (int e = sef_check("se_fused_fwd", B, HW, C, Hd)) return e;
    NVAE_REQUIRE(aligned16(x) && aligned16(skip) && aligned16(y) && w1 && b1 && w2 && b2 && pooled_sum && gate && hidden,
                 "se_fused_fwd: alignment / NULL argument");
    BnFromSlab bn;
    if (int e = sef_bn("se_fused_fwd", bn_in, (long)B * HW, bn)) return e;
    const int RL = 256 / (C / 8);
    const int S = sef_split(B, HW, C);
    if (S > 1) {
        SefOut so{(void*)stats, slab_rows_for(B * S)};
        SefSplit sp{S, g_se_ws, g_se_ws + (size_t)B * S * C, g_se_counters};
        const int P = HW / S;
        const int nch = (P % RL == 0) ? P / RL : 0;
#define SEF_LAUNCH(N_)                                                                                              \
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_split_fwd<T, N_>), B * S, 256, 0, (hipStream_t)stream, (const T*)x, bn, \
                                             (const T*)skip, (T*)y, B, HW, C, Hd, w1, b1, w2, b2, skip_scale,        \
                                             branch_scale, pooled_sum, gate, hidden, so, sp);)
        if (nch == 1) SEF_LAUNCH(1) else if (nch == 2) SEF_LAUNCH(2) else if (nch == 4) SEF_LAUNCH(4) else if (nch == 8) SEF_LAUNCH(8) else SEF_LAUNCH(0)
#undef SEF_LAUNCH
        NVAE_LAUNCH_CHECK("se_fused_fwd (split)");
        return NVAE_OK;
    }
    const int imgs = sef_imgs(B), wgs = sef_wgs(B);
    SefOut so{(void*)stats, slab_rows_for(wgs)};
    const int nch = (HW % RL == 0) ? HW / RL : 0;
#define SEF_LAUNCH(N_)                                                                                              \
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_fused_fwd<T, N_>), wgs, 256, 0, (hipStream_t)stream, (const T*)x, bn, \
                                         (const T*)skip, (T*)y, B, HW, C, Hd, imgs, w1, b1, w2, b2, skip_scale,      \
                                         branch_scale, pooled_sum, gate, hidden, so);)
    if (nch == 1) SEF_LAUNCH(1) else if (nch == 2) SEF_LAUNCH(2) else if (nch == 4) SEF_LAUNCH(4) else if (nch == 8) SEF_LAUNCH(8) else SEF_LAUNCH(0)
#undef SEF_LAUNCH
    NVAE_LAUNCH_CHECK("se_fused_fwd");
    return NVAE_OK;
}

// Backward of the same block.  With xs = BN(x) (or x itself), r[c] = sum_hw xs*dy:
//   dpre2 = bs*r*g*(1-g);  dpre1[h] = relu'(hidden[h]) * sum_c W2[h,c] dpre2[c];  dpool[c] = sum_h W1[c,h] dpre1[h] / HW
//   dxs = bs*dy*gate + dpool;   dskip (+)= ss*dy
// dpre2 | dpre1 go to `scratch` ([B*C] | [B*Hd]) for the FC parameter gradients (nvae_se_wgrad_batched).
// partials != NULL: xs = act(BN(x)) had no other consumer, so dxs is final and the BatchNorm-backward sums
// (sum dpre, sum dpre*x with dpre = dxs * act'(scale*x + shift)) are reduced here into the ZEROED partials[rows][2][C].
template <typename T, int NCH>
__global__ __launch_bounds__(256) void k_se_fused_bwd(
    const T* __restrict__ x, const float* __restrict__ bn_scale, const float* __restrict__ bn_shift, int act,
    const T* __restrict__ dy, const float* __restrict__ gate, const float* __restrict__ hidden, T* dx, T* dskip,
    int B, int HW, int C, int Hd, int imgs, const float* __restrict__ w1, const float* __restrict__ w2, float ss,
    float bs, int acc_dx, int acc_dskip, float* __restrict__ scratch, SefOut so) {
    typedef typename StatT<T>::type ST;
    __shared__ float d2[SEF_MAX_C];
    __shared__ ST part_s[2 * 2048];
    float* part = (float*)part_s;
    __shared__ float d1[SEF_MAX_H];
    __shared__ float s_w1[SEF_W_LDS], s_w2[SEF_W_LDS];
    const int CG = C >> 3, RL = 256 / CG;
    const int tg = threadIdx.x % CG, rl = threadIdx.x / CG;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const float inv_hw = 1.0f / (float)HW;
    const bool w_lds = C * Hd <= SEF_W_LDS;
    float* dpre2_out = scratch;
    float* dpre1_out = scratch + (long)B * C;
    ST* partials = (ST*)so.stats;
    constexpr int NR = NCH > 0 ? NCH : 1;
    float xv[NR][8], gv[NR][8];
    long b = (long)blockIdx.x * imgs;
    if constexpr (NCH > 0) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            V8<T>::ld(x + (b * HW + rl + i * RL) * C + tg * 8, xv[i]);
            V8<T>::ld(dy + (b * HW + rl + i * RL) * C + tg * 8, gv[i]);
        }
    }
    if (w_lds)
        for (int i = threadIdx.x; i < C * Hd; i += 256) { s_w1[i] = w1[i]; s_w2[i] = w2[i]; }
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = bn_scale ? bn_scale[tg * 8 + j] : 1.f; sh[j] = bn_scale ? bn_shift[tg * 8 + j] : 0.f; }
    ST st[2][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { st[0][j] = 0; st[1][j] = 0; }
    for (int im = 0; im < imgs; ++im, ++b) {
        if (b >= B) break;
        const T* xb = x + b * HW * C;
        const T* gb = dy + b * HW * C;
        if (NCH > 0 && im > 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                V8<T>::ld(xb + (long)(rl + i * RL) * C + tg * 8, xv[i]);
                V8<T>::ld(gb + (long)(rl + i * RL) * C + tg * 8, gv[i]);
            }
        }
        // ---- r = sum xs*dy  (xs = act(scale*x + shift))
        float a[1][8] = {{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}};
        if constexpr (NCH > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float xs = xv[i][j] * sc[j] + sh[j];
                    if (act == ACT_SWISH) xs = swishf_(xs);
                    a[0][j] += xs * gv[i][j];
                }
        } else {
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
        }
        __syncthreads();
        sef_scatter<1, float>(part, a, rl, tg, C);
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            const float r = sef_gather<float>(part, 0, c, C, RL);
            const float g = gate[b * C + c];
            const float d = bs * r * g * (1.f - g);
            d2[c] = d;
            dpre2_out[b * C + c] = d;
        }
        __syncthreads();
        const float* W1 = w_lds ? s_w1 : w1;
        const float* W2 = w_lds ? s_w2 : w2;
        for (int h = wave; h < Hd; h += 4) {
            float acc = 0.f;
            for (int c = lane; c < C; c += 64) acc += W2[(long)h * C + c] * d2[c];
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
            for (int h = 0; h < Hd; ++h) acc += W1[(long)c * Hd + h] * d1[h];
            d2[c] = acc * inv_hw;              // dpool
        }
        __syncthreads();
        float g8[8], dp[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { g8[j] = gate[b * C + tg * 8 + j] * bs; dp[j] = d2[tg * 8 + j]; }
        T* dxb = dx + b * HW * C;
        T* dkb = dskip ? dskip + b * HW * C : nullptr;
        auto apply = [&](long off, const float (&g)[8], const float (&v)[8]) {
            float o[8], k[8];
            if (acc_dx) V8<T>::ld(dxb + off, o);
            if (dkb && acc_dskip) V8<T>::ld(dkb + off, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = g[j] * g8[j] + dp[j];
                o[j] = (acc_dx ? o[j] : 0.f) + d;
                k[j] = ((dkb && acc_dskip) ? k[j] : 0.f) + ss * g[j];
                if (partials) {
                    float dpre = d;
                    if (act == ACT_SWISH) dpre *= dswishf_(v[j] * sc[j] + sh[j]);
                    st[0][j] += (ST)dpre; st[1][j] += (ST)dpre * (ST)v[j];
                }
            }
            V8<T>::st(dxb + off, o);
            if (dkb) V8<T>::st(dkb + off, k);
        };
        if constexpr (NCH > 0) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) apply((long)(rl + i * RL) * C + tg * 8, gv[i], xv[i]);
        } else {
            for (int r = rl; r < HW; r += RL) {
                const long off = (long)r * C + tg * 8;
                float g[8], v[8];
                V8<T>::ld(gb + off, g);
                V8<T>::ld(xb + off, v);
                apply(off, g, v);
            }
        }
    }
    if (!partials) return;
    __syncthreads();
    sef_scatter<2, ST>(part_s, st, rl, tg, C);
    __syncthreads();
    const int row = blockIdx.x % so.rows;
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(partials + ((long)row * 2) * C + c, sef_gather<ST>(part_s, 0, c, C, RL));
        atomicAdd(partials + ((long)row * 2 + 1) * C + c, sef_gather<ST>(part_s, 1, c, C, RL));
    }
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
    const int RL = 256 / (C / 8);
    const int S = sef_split(B, HW, C);
    if (S > 1) {
        SefOut so{(void*)partials, slab_rows_for(B * S)};
        SefSplit sp{S, g_se_ws, g_se_ws + (size_t)B * S * C, g_se_counters};
        const int P = HW / S;
        const int nch = (P % RL == 0) ? P / RL : 0;
#define SEF_LAUNCH(N_)                                                                                                \
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_split_bwd<T, N_>), B * S, 256, 0, (hipStream_t)stream, (const T*)x, bn_scale, \
                                             bn_shift, act, (const T*)dy, gate, hidden, (T*)dx, (T*)dskip, B, HW, C, Hd,    \
                                             w1, w2, skip_scale, branch_scale, acc_dx, dskip ? acc_dskip : 0, scratch, so, sp);)
        if (nch == 1) SEF_LAUNCH(1) else if (nch == 2) SEF_LAUNCH(2) else if (nch == 4) SEF_LAUNCH(4) else if (nch == 8) SEF_LAUNCH(8) else SEF_LAUNCH(0)
#undef SEF_LAUNCH
        NVAE_LAUNCH_CHECK("se_fused_bwd (split)");
        return NVAE_OK;
    }
    const int imgs = sef_imgs(B), wgs = sef_wgs(B);
    SefOut so{(void*)partials, slab_rows_for(wgs)};
    const int nch = (HW % RL == 0) ? HW / RL : 0;
#define SEF_LAUNCH(N_)                                                                                                \
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_se_fused_bwd<T, N_>), wgs, 256, 0, (hipStream_t)stream, (const T*)x, bn_scale, \
                                         bn_shift, act, (const T*)dy, gate, hidden, (T*)dx, (T*)dskip, B, HW, C, Hd, imgs,  \
                                         w1, w2, skip_scale, branch_scale, acc_dx, dskip ? acc_dskip : 0, scratch, so);)
    if (nch == 1) SEF_LAUNCH(1) else if (nch == 2) SEF_LAUNCH(2) else if (nch == 4) SEF_LAUNCH(4) else if (nch == 8) SEF_LAUNCH(8) else SEF_LAUNCH(0)
#undef SEF_LAUNCH
    NVAE_LAUNCH_CHECK("se_fused_bwd");
    return NVAE_OK;
}
