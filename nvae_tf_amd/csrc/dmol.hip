// Discretised mixture of logistics output head (forward NLL, gradient, sampling).
//
// Not in the reference (train.py:219 / README.md:25-27 leave the CIFAR / CelebA heads unimplemented);
// specified by oracle/nvae_oracle.py::dmol_* after the NVAE paper / PixelCNN++.  Per pixel the head
// has 10*M logits laid out as [M mixture logits | for colour c: M means, M log-scales, M coefficient
// logits].  The logits tensor is f32 NHWC with row stride `ld` >= 10*M (the conv that produces it
// pads its output channels to a multiple of 8), the image x is f32 in [0, 1] (k/255 is not
// representable in bf16, and the likelihood has a case split at the ends of the range).
//
// HBM-bound, tiny next to the conv stack: B*H*W*(10M) floats read once (fwd) / read once + written
// once (bwd).  One thread owns one pixel; per-image sums use a wave shuffle + one LDS hop.
#include "common.h"
#include "bn_fin.h"

#define DMOL_MAXM 16
#define DMOL_LOG1275 4.8481163645f     // log(127.5)

struct DmolTerm {        // d s_k / d (raw logits of mixture k), filled when GRAD
    float dmean[3], dls[3], dco[3];
};

// s_k = sum_c log p_c(x_c | mixture k) for the pixel whose logits start at l.
template <bool GRAD>
__device__ __forceinline__ float dmol_term(const float* __restrict__ l, int M, int k, const float (&x)[3],
                                           DmolTerm& g) {
    float co[3], mu[3], dmu[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) co[j] = tanhf(l[M + 3 * M * j + 2 * M + k]);
    mu[0] = l[M + k];
    mu[1] = l[M + 3 * M + k] + co[0] * x[0];
    mu[2] = l[M + 6 * M + k] + co[1] * x[0] + co[2] * x[1];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float raw_ls = l[M + 3 * M * c + M + k];
        const float ls = fmaxf(raw_ls, -7.0f);
        const float inv = __expf(-ls);
        const float cen = x[c] - mu[c];
        const float plus = inv * (cen + 1.0f / 255.0f), mn = inv * (cen - 1.0f / 255.0f), mid = inv * cen;
        float lp, dplus = 0.f, dmin = 0.f, dmid = 0.f, ddirect = 0.f;
        if (x[c] < -0.999f) {
            lp = plus - softplusf_(plus);
            dplus = sigmoidf_(-plus);
        } else if (x[c] > 0.99f) {
            lp = -softplusf_(mn);
            dmin = -sigmoidf_(mn);
        } else {
            const float sp = sigmoidf_(plus), sn = sigmoidf_(mn);
            const float delta = sp - sn;
            if (delta > 1e-5f) {
                lp = __logf(fmaxf(delta, 1e-10f));
                dplus = sp * (1.f - sp) / delta;
                dmin = -sn * (1.f - sn) / delta;
            } else {
                lp = mid - ls - 2.0f * softplusf_(mid) - DMOL_LOG1275;
                dmid = 1.f - 2.f * sigmoidf_(mid);
                ddirect = -1.f;
            }
        }
        s += lp;
        if (GRAD) {
            dmu[c] = -inv * (dplus + dmin + dmid);
            g.dmean[c] = dmu[c];
            g.dls[c] = raw_ls >= -7.0f ? ddirect - (dplus * plus + dmin * mn + dmid * mid) : 0.f;
        }
    }
    if (GRAD) {
        g.dco[0] = dmu[1] * x[0] * (1.f - co[0] * co[0]);
        g.dco[1] = dmu[2] * x[0] * (1.f - co[1] * co[1]);
        g.dco[2] = dmu[2] * x[1] * (1.f - co[2] * co[2]);
    }
    return s;
}

// log p(pixel) = logsumexp_k(log_softmax(lp)_k + s_k); also returns the two log-normalisers.
__device__ __forceinline__ float dmol_pixel(const float* __restrict__ l, int M, const float (&x)[3],
                                            float& lse_t, float& lse_lp) {
    float t[DMOL_MAXM];
    float mt = -INFINITY, ml = -INFINITY;
    DmolTerm unused;
#pragma unroll
    for (int k = 0; k < DMOL_MAXM; ++k)
        if (k < M) {
            const float lpk = l[k];
            t[k] = lpk + dmol_term<false>(l, M, k, x, unused);
            mt = fmaxf(mt, t[k]);
            ml = fmaxf(ml, lpk);
        }
    float st = 0.f, sl = 0.f;
#pragma unroll
    for (int k = 0; k < DMOL_MAXM; ++k)
        if (k < M) {
            st += __expf(t[k] - mt);
            sl += __expf(l[k] - ml);
        }
    lse_t = mt + __logf(st);
    lse_lp = ml + __logf(sl);
    return lse_t - lse_lp;
}

// nll[b] = -sum_pixels log p.  gridDim.x = S workgroups per image (round 3: one workgroup per image left 32-64 CUs with
// 16 pixels x 100 strided loads per thread: 563 us at C5); each publishes its partial sum, the image's last arriver adds
// the S partials in slice order (so the result does not depend on the arrival order) and writes nll[b].
#define DMOL_MAX_B 4096
#define DMOL_MAX_S 16
static __device__ float g_dmol_part[DMOL_MAX_B * DMOL_MAX_S];      // (one launch at a time per device: stream-ordered use)
static __device__ int g_dmol_count[DMOL_MAX_B];                     // zero at rest
__global__ __launch_bounds__(256) void k_dmol_fwd(const float* __restrict__ logits, int ld,
                                                  const float* __restrict__ x, float* __restrict__ nll,
                                                  int HW, int M) {
    __shared__ float sm[4];
    const long b = blockIdx.y;
    const int S = gridDim.x, s = blockIdx.x;
    float a = 0.f;
    for (int p = s * 256 + threadIdx.x; p < HW; p += S * 256) {
        const long pix = b * HW + p;
        float xv[3], lt, ll;
#pragma unroll
        for (int c = 0; c < 3; ++c) xv[c] = 2.0f * x[pix * 3 + c] - 1.0f;
        a -= dmol_pixel(logits + pix * ld, M, xv, lt, ll);
    }
    a = block_sum256(a, sm);
    if (S == 1) {
        if (threadIdx.x == 0) nll[b] = a;
        return;
    }
    if (threadIdx.x == 0) bn_store_partial(g_dmol_part + b * DMOL_MAX_S + s, a);
    if (!bn_last_arriver(g_dmol_count + b, S)) return;
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int q = 0; q < S; ++q) t += bn_load_partial(g_dmol_part + b * DMOL_MAX_S + q);
        nll[b] = t;
    }
}

extern "C" int nvae_dmol_fwd(const float* logits, int ld, const float* x, float* nll, int B, int HW, int M,
                             void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && logits && x && nll, "dmol_fwd: bad args");
    NVAE_REQUIRE(M >= 1 && M <= DMOL_MAXM && ld >= 10 * M, "dmol_fwd: M=%d must be in [1, %d] and ld=%d >= 10*M", M, DMOL_MAXM, ld);
    // enough workgroups for the chip; one per image when the batch alone fills it or exceeds the hand-off tables
    int S = 1;
    if (B <= DMOL_MAX_B) while (S < DMOL_MAX_S && (long)B * S < 1024 && HW / (2 * S) >= 256) S *= 2;
    hipLaunchKernelGGL(k_dmol_fwd, dim3(S, B), 256, 0, (hipStream_t)stream, logits, ld, x, nll, HW, M);
    NVAE_LAUNCH_CHECK("dmol_fwd");
    return NVAE_OK;
}

// dlogits = scale * d(-log p)/dlogits; pad channels [10M, ld) are written as zero
template <typename T>
__global__ __launch_bounds__(256) void k_dmol_bwd(const float* __restrict__ logits, int ld,
                                                  const float* __restrict__ x, T* __restrict__ dl,
                                                  long npix, int M, float scale, const float* __restrict__ hyper) {
    scale *= loss_scale_of(hyper);
    for (long pix = blockIdx.x * 256L + threadIdx.x; pix < npix; pix += gridDim.x * 256L) {
        const float* l = logits + pix * ld;
        T* d = dl + pix * ld;
        float xv[3], lse_t, lse_lp;
#pragma unroll
        for (int c = 0; c < 3; ++c) xv[c] = 2.0f * x[pix * 3 + c] - 1.0f;
        dmol_pixel(l, M, xv, lse_t, lse_lp);
        for (int k = 0; k < M; ++k) {
            DmolTerm g;
            const float lpk = l[k];
            const float s = dmol_term<true>(l, M, k, xv, g);
            const float post = __expf(lpk + s - lse_t);
            const float w = -scale * post;
            stf<T>(d + k, -scale * (post - __expf(lpk - lse_lp)));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                stf<T>(d + M + 3 * M * c + k, w * g.dmean[c]);
                stf<T>(d + M + 3 * M * c + M + k, w * g.dls[c]);
                stf<T>(d + M + 3 * M * c + 2 * M + k, w * g.dco[c]);
            }
        }
        for (int j = 10 * M; j < ld; ++j) stf<T>(d + j, 0.f);
    }
}

extern "C" int nvae_dmol_bwd(int dtype, const float* logits, int ld, const float* x, void* dlogits, int B,
                             int HW, int M, float scale, const float* hyper, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && logits && x && dlogits, "dmol_bwd: bad args");
    NVAE_REQUIRE(M >= 1 && M <= DMOL_MAXM && ld >= 10 * M, "dmol_bwd: M=%d must be in [1, %d] and ld=%d >= 10*M", M, DMOL_MAXM, ld);
    const long npix = (long)B * HW;
    long g = (npix + 255) / 256;
    if (g > 4096) g = 4096;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_dmol_bwd<T>), (int)g, 256, 0, (hipStream_t)stream, logits, ld, x, (T*)dlogits, npix, M, scale, hyper);)
    NVAE_LAUNCH_CHECK("dmol_bwd");
    return NVAE_OK;
}

// One draw per pixel given uniform noise (u_mix [npix, M], u_pix [npix, 3] in (0, 1)); out [npix, 3] in [0, 1].
__global__ __launch_bounds__(256) void k_dmol_sample(const float* __restrict__ logits, int ld,
                                                     const float* __restrict__ u_mix,
                                                     const float* __restrict__ u_pix, float* __restrict__ out,
                                                     long npix, int M, float t) {
    for (long pix = blockIdx.x * 256L + threadIdx.x; pix < npix; pix += gridDim.x * 256L) {
        const float* l = logits + pix * ld;
        int best = 0;
        float bv = -INFINITY;
        for (int k = 0; k < M; ++k) {
            const float v = l[k] / t - __logf(-__logf(u_mix[pix * M + k]));
            if (v > bv) { bv = v; best = k; }         // first maximum, as torch.argmax
        }
        float xs[3], co[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float mu = l[M + 3 * M * c + best];
            const float ls = fmaxf(l[M + 3 * M * c + M + best], -7.0f);
            co[c] = tanhf(l[M + 3 * M * c + 2 * M + best]);
            const float u = u_pix[pix * 3 + c];
            xs[c] = mu + __expf(ls) * t * (__logf(u) - __logf(1.0f - u));
        }
        const float x0 = fminf(fmaxf(xs[0], -1.f), 1.f);
        const float x1 = fminf(fmaxf(xs[1] + co[0] * x0, -1.f), 1.f);
        const float x2 = fminf(fmaxf(xs[2] + co[1] * x0 + co[2] * x1, -1.f), 1.f);
        out[pix * 3 + 0] = x0 * 0.5f + 0.5f;
        out[pix * 3 + 1] = x1 * 0.5f + 0.5f;
        out[pix * 3 + 2] = x2 * 0.5f + 0.5f;
    }
}

extern "C" int nvae_dmol_sample(const float* logits, int ld, const float* u_mix, const float* u_pix,
                                float* out, int B, int HW, int M, float temperature, void* stream) {
    NVAE_REQUIRE(B > 0 && HW > 0 && logits && u_mix && u_pix && out, "dmol_sample: bad args");
    NVAE_REQUIRE(M >= 1 && M <= DMOL_MAXM && ld >= 10 * M && temperature > 0.f, "dmol_sample: bad M / ld / temperature");
    const long npix = (long)B * HW;
    long g = (npix + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(k_dmol_sample, (int)g, 256, 0, (hipStream_t)stream, logits, ld, u_mix, u_pix, out, npix, M, temperature);
    NVAE_LAUNCH_CHECK("dmol_sample");
    return NVAE_OK;
}
