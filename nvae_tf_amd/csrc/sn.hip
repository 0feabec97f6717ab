// Spectral normalisation (TFA SpectralNormalization.normalize_weights [3P]; 163-221 call sites, e.g.
// encoder.py:92-98) as four multi-tensor passes over the flat f32 parameter buffer, plus the
// preparation of the MFMA compute copies.  A workgroup owns 16 consecutive k rows of one weight
// matrix W2d[K, Cout]; descs[i].blk_off maps workgroups to matrices (binary search).
//   pass 1  t = W u            (row dots)
//   pass 2  partial column sums of t^T W per 16-row workgroup -> colpart (plain stores)
//   pass 3  |t|^2 and w2 = t^T W summed in a FIXED order;  v W = w2 / |t|;  u' = l2n(v W);  sigma = (v W) . u'
//           (no atomics anywhere: sigma, and with it W <- W / sigma, is bit-reproducible, so data-parallel replicas
//           that start identical stay identical without re-broadcasts)
//   pass 4  W *= 1/sigma in place; write wF[Cout][K] and wD[Cin][taps flipped][Cout] in `dtype`
// HBM traffic per step: 3 reads + 1 write of the masters and 2 writes of the copies.
#include "common.h"

#define SN_ROWS 16
#define SN_L2_EPS 1e-12f

__device__ __forceinline__ int find_desc(const NvaeConvDesc* __restrict__ d, int n, int blk) {
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        int mid = (lo + hi + 1) >> 1;
        if (d[mid].blk_off <= blk) lo = mid;
        else hi = mid - 1;
    }
    return lo;
}

// SN_BPW consecutive 16-row blocks per workgroup: one block is 1-24 KB of weights, too little for a workgroup of
// its own (15 855 workgroups of ~1 us each ran at 0.9 TB/s)
#define SN_BPW 8
__global__ void k_sn_rowdot(const float* __restrict__ params, const NvaeConvDesc* __restrict__ descs,
                            int n, int total_blocks, const float* __restrict__ sn_state, float* __restrict__ t_out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int blk = blockIdx.x * SN_BPW; blk < total_blocks && blk < (blockIdx.x + 1) * SN_BPW; ++blk) {
        const int di = find_desc(descs, n, blk);
        const NvaeConvDesc d = descs[di];
        const int k0 = (blk - d.blk_off) * SN_ROWS;
        const float* W = params + d.w_off;
        const float* u = sn_state + d.u_off;
        // a wave owns rows wave, wave+4, wave+8, wave+12 and walks them TOGETHER: four independent
        // accumulators per lane, u read once per column group, 16-B loads when Cout allows (it always does
        // on this path: Cout % 8 == 0 except the 1-channel logit head)
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        const float* Wr[4];
        bool rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + wave + 4 * q;
            rv[q] = k < d.K;
            Wr[q] = W + (long)(rv[q] ? k : k0) * d.Cout;
        }
        if ((d.Cout & 3) == 0 && ((size_t)W & 15) == 0 && ((size_t)u & 15) == 0) {
            for (int c = lane * 4; c < d.Cout; c += 256) {
                const float4 uv = *(const float4*)(u + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 w = *(const float4*)(Wr[q] + c);
                    a[q] += w.x * uv.x + w.y * uv.y + w.z * uv.z + w.w * uv.w;
                }
            }
        } else {
            for (int c = lane; c < d.Cout; c += 64) {
                const float uv = u[c];
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] += Wr[q][c] * uv;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v = wave_sum(a[q]);
            if (lane == 0 && rv[q]) t_out[d.t_off + k0 + wave + 4 * q] = v;
        }
    }
}

__global__ void k_sn_colsum(const float* __restrict__ params, const NvaeConvDesc* __restrict__ descs,
                            int n, const float* __restrict__ t_in, float* __restrict__ colpart) {
    const int di = find_desc(descs, n, blockIdx.x);
    const NvaeConvDesc d = descs[di];
    const int k0 = (blockIdx.x - d.blk_off) * SN_ROWS;
    int rows = d.K - k0;
    if (rows > SN_ROWS) rows = SN_ROWS;
    const float* W = params + d.w_off + (long)k0 * d.Cout;
    __shared__ float tt[SN_ROWS];
    if (threadIdx.x < SN_ROWS) tt[threadIdx.x] = threadIdx.x < rows ? t_in[d.t_off + k0 + threadIdx.x] : 0.f;
    __syncthreads();
    for (int c = threadIdx.x; c < d.Cout; c += 256) {
        float a = 0.f;
        for (int r = 0; r < rows; ++r) a += tt[r] * W[(long)r * d.Cout + c];
        colpart[d.p_off + (long)(blockIdx.x - d.blk_off) * d.Cout + c] = a;
    }
}

__global__ void k_sn_finish(const NvaeConvDesc* __restrict__ descs, float* __restrict__ sn_state,
                            const float* __restrict__ t_in, const float* __restrict__ colpart,
                            float* __restrict__ w2, float* __restrict__ inv_sigma) {
    __shared__ float sm[4];
    const NvaeConvDesc d = descs[blockIdx.x];
    const int nblk = (d.K + SN_ROWS - 1) / SN_ROWS;
    // |t|^2 and the column sums, each in a fixed order (thread-strided partial sums, then block_sum256's tree)
    float a = 0.f;
    for (int k = threadIdx.x; k < d.K; k += 256) { const float v = t_in[d.t_off + k]; a += v * v; }
    const float nt2 = block_sum256(a, sm);
    const float inv_nt = rsqrtf(fmaxf(nt2, SN_L2_EPS));
    a = 0.f;
    for (int c = threadIdx.x; c < d.Cout; c += 256) {
        const float* p = colpart + d.p_off + c;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int b = 0;
        for (; b + 4 <= nblk; b += 4) {
            s0 += p[(long)b * d.Cout]; s1 += p[(long)(b + 1) * d.Cout];
            s2 += p[(long)(b + 2) * d.Cout]; s3 += p[(long)(b + 3) * d.Cout];
        }
        for (; b < nblk; ++b) s0 += p[(long)b * d.Cout];
        const float v = ((s0 + s1) + (s2 + s3)) * inv_nt;
        w2[d.u_off + c] = v;                          // read back below by the thread that wrote it
        a += v * v;
    }
    a = block_sum256(a, sm);
    const float inv_nu = rsqrtf(fmaxf(a, SN_L2_EPS));
    for (int c = threadIdx.x; c < d.Cout; c += 256) sn_state[d.u_off + c] = w2[d.u_off + c] * inv_nu;
    if (threadIdx.x == 0) inv_sigma[d.idx] = 1.0f / (a * inv_nu);   // sigma = sum (vW)^2 * inv_nu
}

extern "C" int nvae_sn_power_iter(float* params, const NvaeConvDesc* descs, int n, int total_blocks,
                                  float* sn_state, float* sn_scratch_t, float* colpart, float* w2,
                                  float* inv_sigma, void* stream) {
    NVAE_REQUIRE(n > 0 && total_blocks > 0 && params && descs && sn_state && sn_scratch_t && colpart && w2 && inv_sigma,
                 "sn_power_iter: bad args");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(k_sn_rowdot, cdiv(total_blocks, SN_BPW), 256, 0, s, params, descs, n, total_blocks, sn_state, sn_scratch_t);
    hipLaunchKernelGGL(k_sn_colsum, total_blocks, 256, 0, s, params, descs, n, sn_scratch_t, colpart);
    hipLaunchKernelGGL(k_sn_finish, n, 256, 0, s, descs, sn_state, sn_scratch_t, colpart, w2, inv_sigma);
    NVAE_LAUNCH_CHECK("sn_power_iter");
    return NVAE_OK;
}

template <typename T> struct Pack16;
template <> struct Pack16<bf16> {
    static __device__ __forceinline__ void store(bf16* p, const float (&v)[16]) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = v[j]; b[j] = v[8 + j]; }
        V8<bf16>::st(p, a);
        V8<bf16>::st(p + 8, b);
    }
};
template <> struct Pack16<f16> {
    static __device__ __forceinline__ void store(f16* p, const float (&v)[16]) {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = v[j]; b[j] = v[8 + j]; }
        V8<f16>::st(p, a);
        V8<f16>::st(p + 8, b);
    }
};
template <> struct Pack16<float> {
    static __device__ __forceinline__ void store(float* p, const float (&v)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) ((float4*)p)[j] = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
    }
};

template <typename T>
__global__ void k_weight_prep(float* __restrict__ params, const NvaeConvDesc* __restrict__ descs, int n,
                              const float* __restrict__ inv_sigma, T* __restrict__ wcopies) {
    const int di = find_desc(descs, n, blockIdx.x);
    const NvaeConvDesc d = descs[di];
    const int k0 = (blockIdx.x - d.blk_off) * SN_ROWS;
    int rows = d.K - k0;
    if (rows > SN_ROWS) rows = SN_ROWS;
    const float sc = inv_sigma ? inv_sigma[d.idx] : 1.0f;
    float* W = params + d.w_off + (long)k0 * d.Cout;
    for (int c = threadIdx.x; c < d.Cout; c += 256) {
        float v[16];
#pragma unroll
        for (int r = 0; r < SN_ROWS; ++r) {
            v[r] = 0.f;
            if (r < rows) {
                v[r] = W[(long)r * d.Cout + c] * sc;
                if (inv_sigma) W[(long)r * d.Cout + c] = v[r];
                if (d.wd_off >= 0) {
                    int k = k0 + r;
                    int tap = k / d.Cin, ci = k - tap * d.Cin;
                    stf<T>(wcopies + d.wd_off + (long)ci * d.wd_ld + (long)(d.taps - 1 - tap) * d.Cout + c, v[r]);
                }
            }
        }
        if (d.wf_off >= 0) {
            T* dst = wcopies + d.wf_off + (long)c * d.wf_ld + k0;
            if (k0 + SN_ROWS <= d.wf_ld) Pack16<T>::store(dst, v);
            else
                for (int r = 0; r < SN_ROWS && k0 + r < d.wf_ld; ++r) stf<T>(dst + r, v[r]);
        }
    }
}

extern "C" int nvae_weight_prep(int dtype, float* params, const NvaeConvDesc* descs, int n,
                                int total_blocks, const float* inv_sigma, void* wcopies, void* stream) {
    NVAE_REQUIRE(n > 0 && total_blocks > 0 && params && descs && wcopies, "weight_prep: bad args");
    NVAE_REQUIRE(aligned16(wcopies), "weight_prep: wcopies must be 16-B aligned");
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_weight_prep<T>), total_blocks, 256, 0, (hipStream_t)stream, params, descs, n, inv_sigma, (T*)wcopies);)
    NVAE_LAUNCH_CHECK("weight_prep");
    return NVAE_OK;
}
