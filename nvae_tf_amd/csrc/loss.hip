// Latent-hierarchy kernels: Gaussian sampler + per-group KL (+ log q / log p), Bernoulli
// reconstruction, KL balancing / loss assembly, BN-gamma abs-max regulariser.
// All reductions are f32: wave64 shuffle reduction, then one LDS hop per block.
#include "common.h"
#include "bn_fin.h"

__device__ __forceinline__ float softclamp5_(float x) { return 5.0f * tanhf(x * 0.2f); }
__device__ __forceinline__ float dsoftclamp5_(float x) {
    float t = tanhf(x * 0.2f);
    return 1.0f - t * t;
}
#define HALF_LOG_2PI 0.9189385332046727f

// One block per image, or S blocks per image with an ordered last-arriver sum when the batch alone leaves most of the chip idle
// (C5: 32 images x 20 480 latent elements ran 35 us on 32 CUs).  common.py:76-102 + models.py:197-201 + util.py:39-46.
#define SAMP_MAX_B 4096
#define SAMP_MAX_S 16
static __device__ float g_samp_part[SAMP_MAX_B * SAMP_MAX_S * 3];    // (one launch at a time per device: stream-ordered use)
static __device__ int g_samp_count[SAMP_MAX_B];                       // zero at rest
template <typename T>
__global__ void k_sampler_fwd(const float* __restrict__ enc_p, const float* __restrict__ dec_p,
                              const float* __restrict__ eps, T* __restrict__ z, float* __restrict__ kl,
                              float* logq, float* logp, float* __restrict__ mu_sigma, int HW, int L,
                              long total) {
    __shared__ float sm[4];
    const int b = blockIdx.y;
    const int S = gridDim.x, sl = blockIdx.x;       // S workgroups per image (large latent maps at small batch: C4 / C5)
    const int n = HW * L;
    float akl = 0.f, aq = 0.f, ap = 0.f;
    for (int e = sl * 256 + threadIdx.x; e < n; e += S * 256) {
        int pix = e / L, l = e - pix * L;
        long pbase = ((long)b * HW + pix) * (2 * L) + l;
        long idx = (long)b * n + e;
        float a = enc_p[pbase], bb = enc_p[pbase + L];
        float mq, sq, mp, sp;
        if (dec_p) {
            float m = dec_p[pbase], s = dec_p[pbase + L];
            mp = softclamp5_(m);
            sp = expf(softclamp5_(s)) + 1e-2f;
            mq = softclamp5_(a + m);
            sq = expf(softclamp5_(bb + s)) + 1e-2f;
        } else {
            mp = 0.f; sp = 1.f;
            mq = softclamp5_(a);
            sq = expf(softclamp5_(bb)) + 1e-2f;
        }
        float ep = eps[idx];
        float zz = mq + ep * sq;
        stf<T>(z + idx, zz);
        float t1 = (mq - mp) / sp, t2 = sq / sp;
        akl += 0.5f * (t1 * t1 + t2 * t2) - 0.5f - logf(t2);
        if (logq) {
            // (z - mq)/sq == eps exactly
            aq += -0.5f * ep * ep - HALF_LOG_2PI - logf(sq);
            float nz = (zz - mp) / sp;
            ap += -0.5f * nz * nz - HALF_LOG_2PI - logf(sp);
        }
        if (mu_sigma) {
            mu_sigma[idx] = mq; mu_sigma[total + idx] = sq;
            mu_sigma[2 * total + idx] = mp; mu_sigma[3 * total + idx] = sp;
        }
    }
    akl = block_sum256(akl, sm);
    if (logq) {
        aq = block_sum256(aq, sm);
        ap = block_sum256(ap, sm);
    }
    if (S == 1) {
        if (threadIdx.x == 0) {
            kl[b] = akl;
            if (logq) { logq[b] += aq; logp[b] += ap; }
        }
        return;
    }
    // partial sums out, the image's last arriver adds them in slice order (independent of the arrival order)
    float* part = g_samp_part + ((long)b * SAMP_MAX_S + sl) * 3;
    if (threadIdx.x == 0) { bn_store_partial(part, akl); bn_store_partial(part + 1, aq); bn_store_partial(part + 2, ap); }
    if (!bn_last_arriver(g_samp_count + b, S)) return;
    if (threadIdx.x == 0) {
        float t0 = 0.f, t1 = 0.f, t2 = 0.f;
        for (int q = 0; q < S; ++q) {
            const float* pq = g_samp_part + ((long)b * SAMP_MAX_S + q) * 3;
            t0 += bn_load_partial(pq); t1 += bn_load_partial(pq + 1); t2 += bn_load_partial(pq + 2);
        }
        kl[b] = t0;
        if (logq) { logq[b] += t1; logp[b] += t2; }
    }
}

extern "C" int nvae_sampler_fwd(int dtype, const float* enc_p, const float* dec_p, const float* eps,
                                void* z, float* kl, float* logq, float* logp, float* mu_sigma, int B,
                                int HW, int L, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && L > 0 && enc_p && eps && z && kl, "sampler_fwd: bad args");
    NVAE_REQUIRE((logq == nullptr) == (logp == nullptr), "sampler_fwd: logq/logp must both be set or NULL");
    int S = 1;
    if (B <= SAMP_MAX_B) while (S < SAMP_MAX_S && (long)B * S < 512 && (long)HW * L / (2 * S) >= 1024) S *= 2;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_sampler_fwd<T>), dim3(S, B), 256, 0, (hipStream_t)stream, enc_p, dec_p, eps, (T*)z, kl, logq, logp, mu_sigma, HW, L, (long)B * HW * L);)
    NVAE_LAUNCH_CHECK("sampler_fwd");
    return NVAE_OK;
}

template <typename T>
__global__ void k_sampler_bwd(const float* __restrict__ enc_p, const float* __restrict__ dec_p,
                              const float* __restrict__ eps, const T* __restrict__ dz,
                              const float* __restrict__ coeff, const float* __restrict__ hyper,
                              float inv_batch, T* __restrict__ d_enc, T* __restrict__ d_dec, int L,
                              long n, const float* __restrict__ gscales, int gid) {
    inv_batch *= loss_scale_of(hyper);
    // dz arrives range-normalised (elementwise.hip "activation-gradient range normalisation"): the KL seed joins it on
    // the same exponent
    if (gscales) inv_batch *= exp2f(gscales[gid]);
    const float ckl = hyper[NVAE_HY_BETA] * coeff[0] * inv_batch;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        long pix = i / L;
        int l = (int)(i - pix * L);
        long pbase = pix * (2 * L) + l;
        float a = enc_p[pbase], bb = enc_p[pbase + L];
        float ep = eps[i];
        float g = dz ? ldf<T>(dz + i) : 0.f;
        if (dec_p) {
            float m = dec_p[pbase], s = dec_p[pbase + L];
            float mp = softclamp5_(m), ep_s = expf(softclamp5_(s)), sp = ep_s + 1e-2f;
            float mq = softclamp5_(a + m), eq_s = expf(softclamp5_(bb + s)), sq = eq_s + 1e-2f;
            float t1 = (mq - mp) / sp, t2 = sq / sp;
            float dmq = g + ckl * t1 / sp;
            float dsq = g * ep + ckl * (t2 / sp - 1.f / sq);
            float dmp = -ckl * t1 / sp;
            float dsp = ckl * (1.f - t1 * t1 - t2 * t2) / sp;
            float da = dmq * dsoftclamp5_(a + m);
            float db = dsq * eq_s * dsoftclamp5_(bb + s);
            stf<T>(d_enc + pbase, da);
            stf<T>(d_enc + pbase + L, db);
            stf<T>(d_dec + pbase, da + dmp * dsoftclamp5_(m));
            stf<T>(d_dec + pbase + L, db + dsp * ep_s * dsoftclamp5_(s));
        } else {
            float mq = softclamp5_(a), eq_s = expf(softclamp5_(bb)), sq = eq_s + 1e-2f;
            float dmq = g + ckl * mq;
            float dsq = g * ep + ckl * (sq - 1.f / sq);
            stf<T>(d_enc + pbase, dmq * dsoftclamp5_(a));
            stf<T>(d_enc + pbase + L, dsq * eq_s * dsoftclamp5_(bb));
        }
    }
}

extern "C" int nvae_sampler_bwd_scaled(int dtype, const float* enc_p, const float* dec_p, const float* eps,
                                       const void* dz, const float* coeff, const float* hyper, float inv_batch,
                                       void* d_enc, void* d_dec, int B, int HW, int L, const float* gscales, int gid,
                                       void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && L > 0 && enc_p && eps && coeff && hyper && d_enc, "sampler_bwd: bad args");
    NVAE_REQUIRE((dec_p == nullptr) == (d_dec == nullptr), "sampler_bwd: dec_p/d_dec mismatch");
    long n = (long)B * HW * L;
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_sampler_bwd<T>), (int)g, 256, 0, (hipStream_t)stream, enc_p, dec_p, eps, (const T*)dz, coeff, hyper, inv_batch, (T*)d_enc, (T*)d_dec, L, n, gscales, gid);)
    NVAE_LAUNCH_CHECK("sampler_bwd");
    return NVAE_OK;
}
extern "C" int nvae_sampler_bwd(int dtype, const float* enc_p, const float* dec_p, const float* eps,
                                const void* dz, const float* coeff, const float* hyper, float inv_batch,
                                void* d_enc, void* d_dec, int B, int HW, int L, void* stream) {
    return nvae_sampler_bwd_scaled(dtype, enc_p, dec_p, eps, dz, coeff, hyper, inv_batch, d_enc, d_dec, B, HW, L, nullptr, 0,
                                   stream);
}

// models.py:242-250; one block per image
template <typename T>
__global__ void k_bernoulli_fwd(const float* __restrict__ logits, const T* __restrict__ x,
                                float* __restrict__ recon, int H, int W, int C, int crop) {
    __shared__ float sm[4];
    const int b = blockIdx.x;
    const int n = H * W * C;
    float a = 0.f;
    for (int e = threadIdx.x; e < n; e += 256) {
        int pix = e / C;
        int h = pix / W, w = pix - h * W;
        if (crop && (h < 2 || h >= H - 2 || w < 2 || w >= W - 2)) continue;
        float l = logits[(long)b * n + e];
        float xv = ldf<T>(x + (long)b * n + e);
        a += softplusf_(l) - xv * l;
    }
    a = block_sum256(a, sm);
    if (threadIdx.x == 0) recon[b] = a;
}

extern "C" int nvae_bernoulli_fwd(int dtype, const float* logits, const void* x, float* recon, int B,
                                  int H, int W, int C, int crop, void* stream) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && logits && x && recon, "bernoulli_fwd: bad args");
    NVAE_REQUIRE(!crop || (H > 4 && W > 4), "bernoulli_fwd: crop needs H, W > 4");
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bernoulli_fwd<T>), B, 256, 0, (hipStream_t)stream, logits, (const T*)x, recon, H, W, C, crop);)
    NVAE_LAUNCH_CHECK("bernoulli_fwd");
    return NVAE_OK;
}

template <typename T>
__global__ void k_bernoulli_bwd(const float* __restrict__ logits, const T* __restrict__ x,
                                T* __restrict__ dl, long n, float inv_batch, const float* __restrict__ hyper) {
    inv_batch *= loss_scale_of(hyper);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L)
        stf<T>(dl + i, (sigmoidf_(logits[i]) - ldf<T>(x + i)) * inv_batch);
}

extern "C" int nvae_bernoulli_bwd(int dtype, const float* logits, const void* x, void* dlogits, long n,
                                  float inv_batch, const float* hyper, void* stream) {
    NVAE_REQUIRE(n > 0 && logits && x && dlogits, "bernoulli_bwd: bad args");
    long g = (n + 255) / 256;
    if (g > 2048) g = 2048;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_bernoulli_bwd<T>), (int)g, 256, 0, (hipStream_t)stream, logits, (const T*)x, (T*)dlogits, n, inv_batch, hyper);)
    NVAE_LAUNCH_CHECK("bernoulli_bwd");
    return NVAE_OK;
}

// am[g] = mean_b |kl_all[g,b]|   (models.py:210 before the +0.01)
__global__ void k_kl_absmean(const float* __restrict__ kl_all, int B, float* __restrict__ am) {
    __shared__ float sm[4];
    const int g = blockIdx.x;
    float a = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) a += fabsf(kl_all[(long)g * B + b]);
    a = block_sum256(a, sm);
    if (threadIdx.x == 0) am[g] = a / (float)B;
}

extern "C" int nvae_kl_absmean(const float* kl_all, int G, int B, float* am, void* stream) {
    NVAE_REQUIRE(G > 0 && B > 0 && kl_all && am, "kl_absmean: bad args");
    hipLaunchKernelGGL(k_kl_absmean, G, 256, 0, (hipStream_t)stream, kl_all, B, am);
    NVAE_LAUNCH_CHECK("kl_absmean");
    return NVAE_OK;
}

#define MAX_GROUPS 256
// models.py:204-222 + 124-126.  Single block.
__global__ void k_loss_finalize(const float* __restrict__ kl_all, const float* __restrict__ am,
                                const float* __restrict__ alphas, int G, int B,
                                const float* __restrict__ recon, const float* __restrict__ bn_loss,
                                const float* __restrict__ hyper, float* __restrict__ coeff,
                                float* __restrict__ kl_loss, float* __restrict__ results) {
    __shared__ float cf[MAX_GROUPS];
    __shared__ float sm[4];
    const float beta = hyper[NVAE_HY_BETA];
    const bool balance = hyper[NVAE_HY_BALANCE] != 0.f;
    if (threadIdx.x == 0) {
        if (balance) {
            float total = 0.f;
            for (int g = 0; g < G; ++g) total += am[g] + 0.01f;
            float mean = 0.f;
            for (int g = 0; g < G; ++g) {
                cf[g] = (am[g] + 0.01f) / alphas[g] * total;
                mean += cf[g];
            }
            mean /= (float)G;
            for (int g = 0; g < G; ++g) cf[g] /= mean;
        } else {
            for (int g = 0; g < G; ++g) cf[g] = 1.f;
        }
        for (int g = 0; g < G; ++g) coeff[g] = cf[g];
    }
    __syncthreads();
    float ar = 0.f, ak = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
        float k = 0.f;
        for (int g = 0; g < G; ++g) k += kl_all[(long)g * B + b] * cf[g];
        k *= beta;
        kl_loss[b] = k;
        ak += k;
        ar += recon[b];
    }
    ar = block_sum256(ar, sm);
    ak = block_sum256(ak, sm);
    if (threadIdx.x == 0) {
        float bn = bn_loss ? bn_loss[0] : 0.f;
        results[NVAE_RES_RECON] = ar / (float)B;
        results[NVAE_RES_KL] = ak / (float)B;
        results[NVAE_RES_BN] = bn;
        results[NVAE_RES_LOSS] = (ar + ak) / (float)B + bn;
    }
}

extern "C" int nvae_loss_finalize(const float* kl_all, const float* am, const float* alphas, int G,
                                  int B, const float* recon, const float* bn_loss, const float* hyper,
                                  float* coeff, float* kl_loss, float* results, void* stream) {
    NVAE_REQUIRE(G > 0 && G <= MAX_GROUPS && B > 0, "loss_finalize: G=%d must be in [1,%d]", G, MAX_GROUPS);
    NVAE_REQUIRE(kl_all && am && alphas && recon && hyper && coeff && kl_loss && results, "loss_finalize: NULL arg");
    hipLaunchKernelGGL(k_loss_finalize, 1, 256, 0, (hipStream_t)stream, kl_all, am, alphas, G, B, recon, bn_loss, hyper, coeff, kl_loss, results);
    NVAE_LAUNCH_CHECK("loss_finalize");
    return NVAE_OK;
}

// models.py:252-267: one wave per BN layer
__global__ void k_bn_absmax_fwd(const float* __restrict__ params, const int* __restrict__ table,
                                int n_layers, float lambda, float* bn_loss, int* __restrict__ argmax) {
    int layer = blockIdx.x * 4 + (threadIdx.x >> 6);
    int lane = threadIdx.x & 63;
    if (layer >= n_layers) return;
    int off = table[2 * layer], C = table[2 * layer + 1];
    float best = -1.f;
    int bi = 0;
    for (int c = lane; c < C; c += 64) {
        float v = fabsf(params[off + c]);
        if (v > best) { best = v; bi = c; }
    }
    // arg-max across the wave; ties resolved towards the lowest index (tf.reduce_max subgradient
    // goes to the first maximal element in our restatement)
    for (int o = 32; o > 0; o >>= 1) {
        float ob = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) {
        argmax[layer] = bi;
        atomicAdd(bn_loss, lambda * best);
    }
}

// deterministic twin: one workgroup, the layers' maxima are summed by one thread in layer order
__global__ void k_bn_absmax_fwd_ordered(const float* __restrict__ params, const int* __restrict__ table,
                                        int n_layers, float lambda, float* bn_loss, int* __restrict__ argmax) {
    __shared__ float s_best[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float total = 0.f;
    for (int l0 = 0; l0 < n_layers; l0 += 4) {
        const int layer = l0 + wave;
        float best = -1.f;
        int bi = 0;
        if (layer < n_layers) {
            const int off = table[2 * layer], C = table[2 * layer + 1];
            for (int c = lane; c < C; c += 64) {
                const float v = fabsf(params[off + c]);
                if (v > best) { best = v; bi = c; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (lane == 0) argmax[layer] = bi;
        }
        if (lane == 0) s_best[wave] = layer < n_layers ? lambda * best : 0.f;
        __syncthreads();
        if (threadIdx.x == 0) total += ((s_best[0] + s_best[1]) + s_best[2]) + s_best[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) *bn_loss += total;
}

extern "C" int nvae_bn_absmax_fwd(const float* params, const int* table, int n_layers, float lambda,
                                  float* bn_loss, int* argmax, void* stream) {
    NVAE_REQUIRE(n_layers > 0 && params && table && bn_loss && argmax, "bn_absmax_fwd: bad args");
    if (g_nvae_det) {
        hipLaunchKernelGGL(k_bn_absmax_fwd_ordered, 1, 256, 0, (hipStream_t)stream, params, table, n_layers, lambda, bn_loss, argmax);
        NVAE_LAUNCH_CHECK("bn_absmax_fwd (ordered)");
        return NVAE_OK;
    }
    hipLaunchKernelGGL(k_bn_absmax_fwd, cdiv(n_layers, 4), 256, 0, (hipStream_t)stream, params, table, n_layers, lambda, bn_loss, argmax);
    NVAE_LAUNCH_CHECK("bn_absmax_fwd");
    return NVAE_OK;
}

__global__ void k_bn_absmax_bwd(const float* __restrict__ params, float* grads,
                                const int* __restrict__ table, const int* __restrict__ argmax,
                                int n_layers, float lambda, const float* __restrict__ hyper) {
    int layer = blockIdx.x * 256 + threadIdx.x;
    if (layer >= n_layers) return;
    lambda *= loss_scale_of(hyper);
    int idx = table[2 * layer] + argmax[layer];
    float v = params[idx];
    grads[idx] += lambda * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
}

extern "C" int nvae_bn_absmax_bwd(const float* params, float* grads, const int* table,
                                  const int* argmax, int n_layers, float lambda, const float* hyper, void* stream) {
    NVAE_REQUIRE(n_layers > 0 && params && grads && table && argmax, "bn_absmax_bwd: bad args");
    hipLaunchKernelGGL(k_bn_absmax_bwd, cdiv(n_layers, 256), 256, 0, (hipStream_t)stream, params, grads, table, argmax, n_layers, lambda, hyper);
    NVAE_LAUNCH_CHECK("bn_absmax_bwd");
    return NVAE_OK;
}
