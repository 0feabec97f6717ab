// Elementwise kernels: activations, add, cast, nearest-upsample backward, Philox randn, Adamax.
// All are HBM-bound streaming kernels: 16-B (bf16) / 32-B (f32) per lane per iteration,
// grid capped at 2048 blocks with a grid-stride loop.
#include "common.h"

thread_local char g_nvae_err[512] = {0};
extern "C" const char* nvae_last_error(void) { return g_nvae_err; }
int g_nvae_det = 0;
extern "C" int nvae_set_deterministic(int on) { g_nvae_det = on != 0; return NVAE_OK; }
extern "C" int nvae_get_deterministic(void) { return g_nvae_det; }
extern "C" int nvae_abi_version(void) { return NVAE_ABI_VERSION; }

static inline int ew_grid(long n8) {
    long g = (n8 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

template <typename T, int OP>
__global__ void k_unary_fwd(const T* __restrict__ x, T* __restrict__ y, long n8, float a, float b) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float v[8];
        V8<T>::ld(x + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (OP == NVAE_OP_AFFINE) v[j] = a * v[j] + b;
            else if (OP == NVAE_OP_SWISH) v[j] = swishf_(v[j]);
            else v[j] = eluf_(v[j]);
        }
        V8<T>::st(y + i * 8, v);
    }
}

template <typename T, int OP>
__global__ void k_unary_bwd(const T* __restrict__ x, const T* __restrict__ dy, T* dx, long n8, int acc) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float v[8], g[8], o[8];
        V8<T>::ld(x + i * 8, v);
        V8<T>::ld(dy + i * 8, g);
        if (acc) V8<T>::ld(dx + i * 8, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float d = (OP == NVAE_OP_SWISH) ? dswishf_(v[j]) : deluf_(v[j]);
            o[j] = (acc ? o[j] : 0.f) + g[j] * d;
        }
        V8<T>::st(dx + i * 8, o);
    }
}

template <typename T>
__global__ void k_add(T* dst, const T* __restrict__ src, long n8, int acc) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float s[8], d[8];
        V8<T>::ld(src + i * 8, s);
        if (acc) {
            V8<T>::ld(dst + i * 8, d);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += d[j];
        }
        V8<T>::st(dst + i * 8, s);
    }
}

template <typename TS, typename TD>
__global__ void k_cast(const TS* __restrict__ src, TD* __restrict__ dst, long n8) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float v[8];
        V8<TS>::ld(src + i * 8, v);
        V8<TD>::st(dst + i * 8, v);
    }
}

// dx[b,h,w,c] (+)= sum_{i,j<f} dxu[b, h*f+i, w*f+j, c]
template <typename T>
__global__ void k_upsample_pool(const T* __restrict__ dxu, T* dx, int H, int W, int C8, int f, long n8,
                                int acc) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c8 = (int)(i % C8);
        long p = i / C8;
        int w = (int)(p % W);
        long q = p / W;
        int h = (int)(q % H);
        long b = q / H;
        float o[8];
        if (acc) V8<T>::ld(dx + i * 8, o);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = 0.f;
        }
        for (int ii = 0; ii < f; ++ii)
            for (int jj = 0; jj < f; ++jj) {
                long src = ((b * (H * f) + (h * f + ii)) * (long)(W * f) + (w * f + jj)) * C8 + c8;
                float v[8];
                V8<T>::ld(dxu + src * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] += v[j];
            }
        V8<T>::st(dx + i * 8, o);
    }
}

// ---------------------------------------------------------------------------------------
// Philox4x32-10 + Box-Muller.  Each thread produces 4 normals from counter (ctr_base + tid).
// The 64-bit base counter lives in device memory and is advanced by a trailing 1-thread kernel
// so that graph replays draw fresh noise.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    unsigned hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    unsigned hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    unsigned n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__global__ void k_randn(float* __restrict__ out, long n, unsigned long long seed,
                        const unsigned long long* __restrict__ counter) {
    unsigned long long base = *counter;
    long n4 = (n + 3) / 4;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        unsigned long long ctr = base + (unsigned long long)i;
        unsigned c[4] = {(unsigned)ctr, (unsigned)(ctr >> 32), 0x5eedu, 0u};
        unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            philox_round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        float u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = ((float)(c[j] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float r0 = sqrtf(-2.f * __logf(u[0])), r1 = sqrtf(-2.f * __logf(u[2]));
        float s0, c0, s1, c1;
        __sincosf(6.283185307179586f * u[1], &s0, &c0);
        __sincosf(6.283185307179586f * u[3], &s1, &c1);
        float z[4] = {r0 * c0, r0 * s0, r1 * c1, r1 * s1};
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (i * 4 + j < n) out[i * 4 + j] = z[j];
    }
}
__global__ void k_advance_counter(unsigned long long* counter, unsigned long long by) { *counter += by; }

// Adamax, Keras formulation: m = b1 m + (1-b1) g; u = max(b2 u, |g|); p -= lr_t * m / (u + eps)
__global__ void k_adamax(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                         float* __restrict__ u, long n4, const float* __restrict__ hyper, float b1,
                         float b2, float eps) {
    if (hyper[NVAE_HY_OVERFLOW] != 0.f) return;       // non-finite gradient (nvae_grad_guard): skip the step
    const float lr_t = hyper[NVAE_HY_LR];
    const float gs = hyper[NVAE_HY_GSCALE] != 0.f ? hyper[NVAE_HY_GSCALE] : 1.f;     // undo the loss scale (f16 path)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        float4 pv = ((float4*)p)[i], gv = ((const float4*)g)[i], mv = ((float4*)m)[i], uv = ((float4*)u)[i];
        float* pp = (float*)&pv; float* gp = (float*)&gv; float* mp = (float*)&mv; float* up = (float*)&uv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g1 = gp[j] * gs;
            mp[j] = b1 * mp[j] + (1.f - b1) * g1;
            up[j] = fmaxf(b2 * up[j], fabsf(g1));
            pp[j] -= lr_t * mp[j] / (up[j] + eps);
        }
        ((float4*)p)[i] = pv; ((float4*)m)[i] = mv; ((float4*)u)[i] = uv;
    }
}


// ---- dynamic loss scaling (f16 activations) ------------------------------------------------------------------
// Any non-finite element of the (all-reduced) gradient sets the overflow flag; the store is idempotent, so the race
// between workgroups is benign.
__global__ void k_grad_guard(const float* __restrict__ g, long n4, float* hyper) {
    bool bad = false;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
        const uint4 v = ((const uint4*)g)[i];
        bad |= ((v.x & 0x7f800000u) == 0x7f800000u) | ((v.y & 0x7f800000u) == 0x7f800000u) |
               ((v.z & 0x7f800000u) == 0x7f800000u) | ((v.w & 0x7f800000u) == 0x7f800000u);
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) hyper[NVAE_HY_OVERFLOW] = 1.0f;
}
__global__ void k_loss_scale_update(float* hyper, float min_scale, float max_scale) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = hyper[NVAE_HY_LSCALE] != 0.f ? hyper[NVAE_HY_LSCALE] : 1.f, good = hyper[NVAE_HY_GOOD];
    if (hyper[NVAE_HY_OVERFLOW] != 0.f) { s = fmaxf(s * 0.5f, min_scale); good = 0.f; }
    else if (++good >= (float)NVAE_LS_GROWTH_STEPS) { s = fminf(s * 2.0f, max_scale); good = 0.f; }
    hyper[NVAE_HY_LSCALE] = s; hyper[NVAE_HY_GSCALE] = 1.0f / s; hyper[NVAE_HY_GOOD] = good; hyper[NVAE_HY_OVERFLOW] = 0.f;
}
extern "C" int nvae_grad_guard(const float* g, long n, float* hyper, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 4 == 0 && g && hyper && aligned16(g), "grad_guard: bad args");
    hipLaunchKernelGGL(k_grad_guard, ew_grid(n / 4), 256, 0, (hipStream_t)stream, g, n / 4, hyper);
    NVAE_LAUNCH_CHECK("grad_guard");
    return NVAE_OK;
}
extern "C" int nvae_loss_scale_update(float* hyper, float min_scale, float max_scale, void* stream) {
    NVAE_REQUIRE(hyper && min_scale > 0.f && max_scale >= min_scale, "loss_scale_update: bad args");
    hipLaunchKernelGGL(k_loss_scale_update, 1, 64, 0, (hipStream_t)stream, hyper, min_scale, max_scale);
    NVAE_LAUNCH_CHECK("loss_scale_update");
    return NVAE_OK;
}

// ---- activation-gradient range normalisation (float16 path, deep hierarchies) ---------------------------------
// At a random initialisation the gradients of a 30-40-group NVAE grow by ~1.5-2x per latent group on the way back: 19
// decades from the last decoder group to the stem at C5 (profiles/r02_f16_gradient_range_c5.txt), against float16's 12
// including subnormals, so NO single loss scale fits.  The backward pass therefore renormalises the activation
// gradient at group boundaries, entirely on the device (nothing here needs the host, so it lives inside the captured
// graphs): k_grad_amax finds max |g| of the tensor, k_grad_rescale multiplies it by the power of two that brings that
// maximum to 2^target and records the CUMULATIVE exponent under a new scale id (`scales[id]` = log2 of the factor that
// every gradient tagged with that id carries on top of the loss scale).  Linear backward ops hand the tag on; two
// gradients with different tags are added by k_grad_merge on the smaller of the two exponents (no overflow); parameter
// gradients (f32, written with their dy's tag) are divided by it at the end (k_grad_unscale).
template <typename T>
__global__ void k_grad_amax(const T* __restrict__ g, long n8, float* __restrict__ slot) {
    float m = 0.f;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float v[8];
        V8<T>::ld(g + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));      // (fmaxf drops NaNs; infinities survive)
    }
    // one atomic per WORKGROUP on at most 512 workgroups: same-address atomics serialise at the memory side (round 3: one per
    // wave on a 2 048-workgroup grid made this kernel 51 us on a 4 MB tensor, 4.2 ms of the C5 float16 step)
    m = wave_max(m);
    __shared__ float s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicMax((unsigned*)slot, __float_as_uint(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3]))));   // non-negative floats order like uints
}
__device__ __forceinline__ float pow2i(float e) { return __uint_as_float((unsigned)((int)e + 127) << 23); }   // e in [-126, 127]
template <typename T>
__global__ void k_grad_rescale(T* __restrict__ g, long n8, const float* __restrict__ amax, float* __restrict__ scales,
                               int id_in, int id_out, float target_log2) {
    const float a = amax[0];
    float k = 0.f;
    if (a > 0.f && a < 3.0e38f) k = fminf(fmaxf(floorf(target_log2 - log2f(a)), -40.f), 40.f);
    if (blockIdx.x == 0 && threadIdx.x == 0) scales[id_out] = scales[id_in] + k;
    if (k == 0.f) return;
    const float f = pow2i(k);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float v[8];
        V8<T>::ld(g + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] *= f;
        V8<T>::st(g + i * 8, v);
    }
}
template <typename T>
__global__ void k_grad_merge(T* __restrict__ dst, const T* __restrict__ src, long n8, float* __restrict__ scales,
                             int id_dst, int id_src, int id_out) {
    const float sd = scales[id_dst], ss = scales[id_src];
    const float m = fminf(sd, ss);
    const float fd = pow2i(fmaxf(m - sd, -120.f)), fs = pow2i(fmaxf(m - ss, -120.f));
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        float a[8], b[8];
        V8<T>::ld(dst + i * 8, a);
        V8<T>::ld(src + i * 8, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = a[j] * fd + b[j] * fs;
        V8<T>::st(dst + i * 8, a);
    }
    // (every block read both exponents before anybody may overwrite one of them: id_out is a fresh id)
    if (blockIdx.x == 0 && threadIdx.x == 0) scales[id_out] = m;
}
// table: [n_ranges][4] ints = (offset / 4, float4 count, scale id, unused); one row of workgroups per range
__global__ void k_grad_unscale(float* __restrict__ grads, const int* __restrict__ table, const float* __restrict__ scales) {
    const int* r = table + 4 * blockIdx.y;
    const float e = -scales[r[2]];
    if (e == 0.f) return;
    // exponents beyond a float's range are applied in two steps
    const float f1 = pow2i(fminf(fmaxf(e, -100.f), 100.f)), f2 = pow2i(fminf(fmaxf(e - fminf(fmaxf(e, -100.f), 100.f), -100.f), 100.f));
    float4* p = (float4*)grads + r[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < r[1]; i += gridDim.x * 256) {
        float4 v = p[i];
        v.x = v.x * f1 * f2; v.y = v.y * f1 * f2; v.z = v.z * f1 * f2; v.w = v.w * f1 * f2;
        p[i] = v;
    }
}
extern "C" int nvae_grad_amax(int dtype, const void* g, long n, float* slot, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0 && g && slot && aligned16(g), "grad_amax: bad args (n=%ld)", n);
    const int grid = ew_grid(n / 8) < 512 ? ew_grid(n / 8) : 512;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_grad_amax<T>), grid, 256, 0, (hipStream_t)stream, (const T*)g, n / 8, slot);)
    NVAE_LAUNCH_CHECK("grad_amax");
    return NVAE_OK;
}
extern "C" int nvae_grad_rescale(int dtype, void* g, long n, const float* amax, float* scales, int id_in, int id_out,
                                 float target_log2, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0 && g && amax && scales && aligned16(g) && id_in >= 0 && id_out > 0 && id_in != id_out,
                 "grad_rescale: bad args (n=%ld, ids %d -> %d)", n, id_in, id_out);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_grad_rescale<T>), ew_grid(n / 8), 256, 0, (hipStream_t)stream, (T*)g, n / 8, amax,
                                         scales, id_in, id_out, target_log2);)
    NVAE_LAUNCH_CHECK("grad_rescale");
    return NVAE_OK;
}
extern "C" int nvae_grad_merge(int dtype, void* dst, const void* src, long n, float* scales, int id_dst, int id_src,
                               int id_out, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0 && dst && src && scales && aligned16(dst) && aligned16(src) && id_out > 0 &&
                 id_out != id_dst && id_out != id_src, "grad_merge: bad args (n=%ld, ids %d + %d -> %d)", n, id_dst, id_src, id_out);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_grad_merge<T>), ew_grid(n / 8), 256, 0, (hipStream_t)stream, (T*)dst, (const T*)src,
                                         n / 8, scales, id_dst, id_src, id_out);)
    NVAE_LAUNCH_CHECK("grad_merge");
    return NVAE_OK;
}
extern "C" int nvae_grad_unscale(float* grads, const int* table, int n_ranges, const float* scales, void* stream) {
    NVAE_REQUIRE(grads && table && scales && n_ranges > 0 && aligned16(grads), "grad_unscale: bad args");
    hipLaunchKernelGGL(k_grad_unscale, dim3(64, n_ranges), 256, 0, (hipStream_t)stream, grads, table, scales);
    NVAE_LAUNCH_CHECK("grad_unscale");
    return NVAE_OK;
}

extern "C" int nvae_unary_fwd(int dtype, int op, const void* x, void* y, long n, float a, float b,
                              void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0, "unary_fwd: n=%ld must be a positive multiple of 8", n);
    NVAE_REQUIRE(aligned16(x) && aligned16(y), "unary_fwd: pointers must be 16-B aligned");
    NVAE_REQUIRE(op >= 0 && op <= 2, "unary_fwd: bad op %d", op);
    long n8 = n / 8;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
        if (op == NVAE_OP_AFFINE) hipLaunchKernelGGL((k_unary_fwd<T, 0>), ew_grid(n8), 256, 0, s, (const T*)x, (T*)y, n8, a, b);
        else if (op == NVAE_OP_SWISH) hipLaunchKernelGGL((k_unary_fwd<T, 1>), ew_grid(n8), 256, 0, s, (const T*)x, (T*)y, n8, a, b);
        else hipLaunchKernelGGL((k_unary_fwd<T, 2>), ew_grid(n8), 256, 0, s, (const T*)x, (T*)y, n8, a, b);)
    NVAE_LAUNCH_CHECK("unary_fwd");
    return NVAE_OK;
}

extern "C" int nvae_unary_bwd(int dtype, int op, const void* x, const void* dy, void* dx, long n,
                              int accumulate, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0, "unary_bwd: n=%ld must be a positive multiple of 8", n);
    NVAE_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx), "unary_bwd: alignment");
    NVAE_REQUIRE(op == NVAE_OP_SWISH || op == NVAE_OP_ELU, "unary_bwd: bad op %d", op);
    long n8 = n / 8;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(dtype,
        if (op == NVAE_OP_SWISH) hipLaunchKernelGGL((k_unary_bwd<T, 1>), ew_grid(n8), 256, 0, s, (const T*)x, (const T*)dy, (T*)dx, n8, accumulate);
        else hipLaunchKernelGGL((k_unary_bwd<T, 2>), ew_grid(n8), 256, 0, s, (const T*)x, (const T*)dy, (T*)dx, n8, accumulate);)
    NVAE_LAUNCH_CHECK("unary_bwd");
    return NVAE_OK;
}

extern "C" int nvae_add(int dtype, void* dst, const void* src, long n, int accumulate, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0, "add: n=%ld must be a positive multiple of 8", n);
    NVAE_REQUIRE(aligned16(dst) && aligned16(src), "add: alignment");
    long n8 = n / 8;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_add<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (T*)dst, (const T*)src, n8, accumulate);)
    NVAE_LAUNCH_CHECK("add");
    return NVAE_OK;
}

extern "C" int nvae_cast(int sd, int dd, const void* src, void* dst, long n, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 8 == 0, "cast: n=%ld must be a positive multiple of 8", n);
    NVAE_REQUIRE(aligned16(dst) && aligned16(src), "cast: alignment");
    long n8 = n / 8;
    hipStream_t s = (hipStream_t)stream;
    if (sd == NVAE_F32 && dd == NVAE_BF16) hipLaunchKernelGGL((k_cast<float, bf16>), ew_grid(n8), 256, 0, s, (const float*)src, (bf16*)dst, n8);
    else if (sd == NVAE_BF16 && dd == NVAE_F32) hipLaunchKernelGGL((k_cast<bf16, float>), ew_grid(n8), 256, 0, s, (const bf16*)src, (float*)dst, n8);
    else if (sd == NVAE_F32 && dd == NVAE_F32) hipLaunchKernelGGL((k_cast<float, float>), ew_grid(n8), 256, 0, s, (const float*)src, (float*)dst, n8);
    else if (sd == NVAE_BF16 && dd == NVAE_BF16) hipLaunchKernelGGL((k_cast<bf16, bf16>), ew_grid(n8), 256, 0, s, (const bf16*)src, (bf16*)dst, n8);
    else if (sd == NVAE_F32 && dd == NVAE_F16) hipLaunchKernelGGL((k_cast<float, f16>), ew_grid(n8), 256, 0, s, (const float*)src, (f16*)dst, n8);
    else if (sd == NVAE_F16 && dd == NVAE_F32) hipLaunchKernelGGL((k_cast<f16, float>), ew_grid(n8), 256, 0, s, (const f16*)src, (float*)dst, n8);
    else if (sd == NVAE_F16 && dd == NVAE_F16) hipLaunchKernelGGL((k_cast<f16, f16>), ew_grid(n8), 256, 0, s, (const f16*)src, (f16*)dst, n8);
    else NVAE_FAIL(NVAE_EINVAL, "cast: bad dtypes %d->%d", sd, dd);
    NVAE_LAUNCH_CHECK("cast");
    return NVAE_OK;
}

extern "C" int nvae_upsample_pool_bwd(int dtype, const void* dxu, void* dx, int B, int H, int W, int C,
                                      int f, int accumulate, void* stream) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && f >= 1, "upsample_pool_bwd: bad shape");
    NVAE_REQUIRE(aligned16(dxu) && aligned16(dx), "upsample_pool_bwd: alignment");
    long n8 = (long)B * H * W * (C / 8);
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_upsample_pool<T>), ew_grid(n8), 256, 0, (hipStream_t)stream, (const T*)dxu, (T*)dx, H, W, C / 8, f, n8, accumulate);)
    NVAE_LAUNCH_CHECK("upsample_pool_bwd");
    return NVAE_OK;
}

extern "C" int nvae_randn(float* out, long n, unsigned long long seed, unsigned long long* counter_dev,
                          void* stream) {
    NVAE_REQUIRE(n > 0 && out && counter_dev, "randn: bad args");
    long n4 = (n + 3) / 4;
    hipLaunchKernelGGL(k_randn, ew_grid(n4), 256, 0, (hipStream_t)stream, out, n, seed, counter_dev);
    hipLaunchKernelGGL(k_advance_counter, 1, 1, 0, (hipStream_t)stream, counter_dev, (unsigned long long)n4);
    NVAE_LAUNCH_CHECK("randn");
    return NVAE_OK;
}

extern "C" int nvae_adamax(float* p, const float* g, float* m, float* u, long n, const float* hyper,
                           float beta1, float beta2, float eps, void* stream) {
    NVAE_REQUIRE(n > 0 && n % 4 == 0, "adamax: n=%ld must be a positive multiple of 4", n);
    NVAE_REQUIRE(aligned16(p) && aligned16(g) && aligned16(m) && aligned16(u), "adamax: alignment");
    long n4 = n / 4;
    hipLaunchKernelGGL(k_adamax, ew_grid(n4), 256, 0, (hipStream_t)stream, p, g, m, u, n4, hyper, beta1, beta2, eps);
    NVAE_LAUNCH_CHECK("adamax");
    return NVAE_OK;
}
