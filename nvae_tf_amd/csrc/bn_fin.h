// "Last arriver finalizes": the kernels that produce BatchNorm slab partials ([S][2][C] floats, one
// slab per row split) also turn them into per-channel coefficients, so no separate finalize launch
// is needed.  Every workgroup that contributes to a channel group bumps that group's counter after
// publishing its partials; the one that observes count == contributors - 1 sums the slabs for the
// group's channels and writes scale/shift/mean/invstd (+ moving statistics) or dgamma/dbeta/k0k1.
// The counter is reset by the same workgroup, so a zero-initialised buffer stays valid for every
// later launch on the stream (and for every replay of a captured graph).
//
// Cross-XCD visibility without fences: measured on MI355X (tools/mb_sync.hip) an agent-scope
// __threadfence costs ~65 ns PER WORKGROUP, serialised device-wide (128 workgroups: +7.4 us, 1024:
// +64 us), because it writes back / invalidates the XCD's whole L2.  Instead the partials are stored
// with agent-scope (sc1, write-through) stores, each wave waits for its own stores (vmcnt(0)) before
// the workgroup barrier, the counter is an agent-scope atomic executed at the memory side, and the
// finalizing workgroup reads the slabs with agent-scope (sc1) loads, which do not hit stale lines of
// its own L2.  Same-address counter atomics cost ~10 ns each (<= 128 contributors per counter).
#pragma once
#include "common.h"

struct BnFinArgs {
    int* counter;                  // [channel groups]; nullptr = no fused finalize
    float inv_n;                   // 1 / rows
    // forward (statistics -> coefficients)
    const float* gamma; const float* beta;
    float* rm; float* rv;
    float momentum, eps;
    float* scale; float* shift; float* mean; float* invstd;
    // backward (sum dpre, sum dpre*x -> dgamma, dbeta, k0, k1); reads scale/mean/invstd above
    float* dgamma; float* dbeta; float* k0k1;
    int frozen;
};

// All threads of the workgroup call this after their partial stores.  Returns true (uniformly) in
// the workgroup that arrived last of `contributors`.
__device__ __forceinline__ void bn_store_partial(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float bn_load_partial(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool bn_last_arriver(int* counter, int contributors) {
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's partial stores have been written through
    __syncthreads();
    if (threadIdx.x == 0) {
        const int old = atomicAdd(counter, 1);
        const int last = old == contributors - 1;
        if (last) atomicExch(counter, 0);
        s_last = last;
    }
    __syncthreads();
    return s_last != 0;
}

// The same in two halves, for producers that have more stores to issue after their slab partials (the conv
// epilogue: output tile): bn_arrive() right after the partial stores - only those are waited for, the ticket's
// round trip then overlaps the remaining stores - and bn_was_last() at the very end of the kernel.
__device__ __forceinline__ int bn_arrive(int* counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's partial stores have been written through
    __syncthreads();
    return threadIdx.x == 0 ? atomicAdd(counter, 1) : 0;   // meaningful in thread 0 only
}
__device__ __forceinline__ bool bn_was_last(int* counter, int ticket, int contributors) {
    __shared__ int s_last2;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int last = ticket == contributors - 1;
        if (last) atomicExch(counter, 0);
        s_last2 = last;
    }
    __syncthreads();
    return s_last2 != 0;
}

// Coefficients of ONE channel straight from an accumulated statistics slab ([rows][2][C], rows = 1-8: the producers
// add into it with atomics).  Cheap enough for every consumer workgroup to do for the channels it touches, so
// neither a finalize launch nor a last-arriver hand-off is needed; the workgroup the caller elects (`publish`)
// also writes the table for the backward pass and updates the moving statistics.
struct BnFromSlab {
    const void* slab; int rows;          // nullptr: scale / shift below are already final (inputs)
    float inv_n, eps, momentum;
    const float* gamma; const float* beta;
    float* rm; float* rv;
    float* scale; float* shift; float* mean; float* invstd;
};
// F64: the slab holds doubles (the f32 activation path, see "statistics precision" below)
template <bool F64>
__device__ __forceinline__ void bn_coef(const BnFromSlab& a, int C, int c, bool publish, float& sc, float& sh) {
    if (a.slab == nullptr) { sc = a.scale[c]; sh = a.shift[c]; return; }
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < a.rows; ++r) {
        if constexpr (F64) {
            s1 += ((const double*)a.slab)[((long)r * 2) * C + c]; s2 += ((const double*)a.slab)[((long)r * 2 + 1) * C + c];
        } else {
            s1 += ((const float*)a.slab)[((long)r * 2) * C + c]; s2 += ((const float*)a.slab)[((long)r * 2 + 1) * C + c];
        }
    }
    const double md = s1 * (double)a.inv_n;
    const float m = (float)md, var = (float)fmax(s2 * (double)a.inv_n - md * md, 0.0);
    const float is = rsqrtf(var + a.eps);
    sc = a.gamma[c] * is;
    sh = a.beta[c] - m * sc;
    if (publish) {
        a.scale[c] = sc; a.shift[c] = sh; a.mean[c] = m; a.invstd[c] = is;
        a.rm[c] = a.rm[c] * a.momentum + m * (1.f - a.momentum);
        a.rv[c] = a.rv[c] * a.momentum + var * (1.f - a.momentum);
    }
}

// Sum slabs for channels [cbase, cbase + 64) with a workgroup of >= 256 threads.  The strip's slab rows
// (row = 2*s + q, 64 floats each) are read as 16-B agent-scope loads, 16 rows in flight per thread, so
// up to 128 splits cost ONE memory round trip (the loop form of this sum took 8).
// Returns true with (c, s1, s2) valid in the 64 threads that own a channel < C.  C % 4 == 0.
typedef float bn_f4 __attribute__((ext_vector_type(4)));
typedef unsigned int bn_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bool bn_slab_sum64(const float* partials, int S, int C, int cbase, int& c,
                                              float& s1, float& s2) {
    __shared__ float sm[16][64];
    const int fl = threadIdx.x & 15, rl = threadIdx.x >> 4;      // 16-B lane, row lane (q = rl & 1)
    const bool worker = threadIdx.x < 256;                       // larger workgroups: the rest only meet the barriers
    const int cq = cbase + fl * 4;
    const int rows = 2 * S;
    bn_f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (worker && cq < C) {
        // buffer loads with the sc1 (agent-coherent) cache policy: the compiler tracks their vmcnt
        const __amdgpu_buffer_rsrc_t rsrc =
            __builtin_amdgcn_make_buffer_rsrc((void*)partials, 0, (int)((long)rows * C * 4), 0x00020000);
        for (int base = 0; base < rows; base += 256) {
            bn_u4 v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int r = base + rl + 16 * i;
                r = r < rows ? r : rows - 2 + (rl & 1);          // clamped duplicate, weighted 0 below
                v[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r * C + cq) * 4, 0, 16 /* sc1 */);
            }
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (base + rl + 16 * i < rows) acc += __builtin_bit_cast(bn_f4, v[i]);
        }
    }
    __syncthreads();        // protects sm against the previous call's readers
    if (worker) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[rl][fl * 4 + j] = acc[j];
    }
    __syncthreads();
    c = cbase + (int)threadIdx.x;
    if (threadIdx.x >= 64 || c >= C) return false;
    s1 = 0.f; s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k += 2) { s1 += sm[k][threadIdx.x]; s2 += sm[k + 1][threadIdx.x]; }
    return true;
}

// The same for a slab of DOUBLES (the f32 activation path: "statistics precision" below).
__device__ __forceinline__ bool bn_slab_sum64_f64(const double* partials, int S, int C, int cbase, int& c,
                                                  double& s1, double& s2) {
    __shared__ double smd[4][2][64];
    const int cl = threadIdx.x & 63, part = (threadIdx.x >> 6) & 3;
    double a1 = 0.0, a2 = 0.0;
    if (threadIdx.x < 256 && cbase + cl < C)
        for (int s = part; s < S; s += 4) {
            a1 += __hip_atomic_load(partials + ((long)s * 2) * C + cbase + cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            a2 += __hip_atomic_load(partials + ((long)s * 2 + 1) * C + cbase + cl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    __syncthreads();
    if (threadIdx.x < 256) { smd[part][0][cl] = a1; smd[part][1][cl] = a2; }
    __syncthreads();
    c = cbase + (int)threadIdx.x;
    if (threadIdx.x >= 64 || c >= C) return false;
    s1 = smd[0][0][cl] + smd[1][0][cl] + smd[2][0][cl] + smd[3][0][cl];
    s2 = smd[0][1][cl] + smd[1][1][cl] + smd[2][1][cl] + smd[3][1][cl];
    return true;
}

// ---- statistics precision -------------------------------------------------------------------------------
// Every slab is [rows][2][C]: (sum x, sum x^2) forward, (sum dpre, sum dpre*x) backward.  On the bf16 activation path
// its elements are floats.  On the f32 path (the parity path) they are DOUBLES and the producers accumulate in f64:
// E[x^2] - E[x]^2 and sum dpre*x - mean * sum dpre are differences of nearly equal numbers when |mean| >> std
// (15-19 on the depthwise-conv outputs in front of bn3 at initialisation), and through the ~330 layers of the C2
// configuration an f32 rounding of those sums - or merely a different ORDER of the f32 atomic adds from run to
// run - moved the loss by up to 2.6e-3 (5 nats) between runs and three times further from the fp64 oracle than
// the PyTorch-CPU f32 run of the same model.  All finalize arithmetic below is done in double either way.
template <bool F64>
__device__ __forceinline__ bool bn_slab_sum64_t(const void* partials, int S, int C, int cbase, int& c, double& s1,
                                                double& s2) {
    if constexpr (F64) {
        return bn_slab_sum64_f64((const double*)partials, S, C, cbase, c, s1, s2);
    } else {
        float a = 0.f, b = 0.f;
        const bool ok = bn_slab_sum64((const float*)partials, S, C, cbase, c, a, b);
        s1 = a; s2 = b;
        return ok;
    }
}

__device__ __forceinline__ void bn_fin_fwd_channel(const BnFinArgs& a, int c, double s1, double s2) {
    const double md = s1 * (double)a.inv_n;
    const double vd = fmax(s2 * (double)a.inv_n - md * md, 0.0);
    const float m = (float)md, var = (float)vd;
    const float is = rsqrtf(var + a.eps);
    const float sc = a.gamma[c] * is;
    a.scale[c] = sc;
    a.shift[c] = a.beta[c] - m * sc;
    a.mean[c] = m;
    a.invstd[c] = is;
    a.rm[c] = a.rm[c] * a.momentum + m * (1.f - a.momentum);
    a.rv[c] = a.rv[c] * a.momentum + var * (1.f - a.momentum);
}

// dbeta = sum dpre; dgamma = invstd * (sum dpre*x - mean * sum dpre);
// dx = scale*dpre + k1*x + k0 (k0 = k1 = 0 for frozen statistics).
__device__ __forceinline__ void bn_bwd_coefs(const BnFinArgs& a, int c, double s1, double s2, float& dg, float& k0,
                                             float& k1) {
    const float m = a.mean[c], is = a.invstd[c], sc = a.scale[c];
    dg = (float)((double)is * (s2 - (double)m * s1));
    k1 = a.frozen ? 0.f : -sc * dg * is * a.inv_n;
    k0 = a.frozen ? 0.f : (float)(-(double)sc * s1 * (double)a.inv_n) - k1 * m;
}
__device__ __forceinline__ void bn_fin_bwd_channel(const BnFinArgs& a, int C, int c, double s1, double s2) {
    float dg, k0, k1;
    bn_bwd_coefs(a, c, s1, s2, dg, k0, k1);
    a.dgamma[c] += dg;
    a.dbeta[c] += (float)s1;
    a.k0k1[c] = k0;
    a.k0k1[C + c] = k1;
}

// Finalize `ngroups64` consecutive 64-channel groups starting at cbase.
template <bool F64>
__device__ __forceinline__ void bn_fin_fwd(const BnFinArgs& a, const void* partials, int S, int C,
                                           int cbase, int ngroups64) {
    for (int g = 0; g < ngroups64; ++g) {
        int c; double s1, s2;
        if (bn_slab_sum64_t<F64>(partials, S, C, cbase + g * 64, c, s1, s2)) bn_fin_fwd_channel(a, c, s1, s2);
    }
}
template <bool F64>
__device__ __forceinline__ void bn_fin_bwd(const BnFinArgs& a, const void* partials, int S, int C,
                                           int cbase, int ngroups64) {
    for (int g = 0; g < ngroups64; ++g) {
        int c; double s1, s2;
        if (bn_slab_sum64_t<F64>(partials, S, C, cbase + g * 64, c, s1, s2)) bn_fin_bwd_channel(a, C, c, s1, s2);
    }
}

// slab element type of an activation type
template <typename T> struct StatT { typedef float type; };
template <> struct StatT<float> { typedef double type; };
