// Depthwise 5x5 convolution (decoder.py:130), NHWC, vectorised over channels.
// Spatial extents on this path are 4x4 / 8x8 with 768-1536 channels, so a whole image slab is a few
// hundred KB and the 25-tap re-reads are served by L1/L2: the kernel is bound by one HBM read + one
// write of the tensor.  Weights are the f32 masters [5,5,C] (25*C*4 B, L2-resident).
#include "common.h"

// two adjacent channels in one LDS / global access
template <typename T> __device__ __forceinline__ void ld2(const T* p, float& a, float& b);
template <> __device__ __forceinline__ void ld2<bf16>(const bf16* p, float& a, float& b) {
    const unsigned v = *(const unsigned*)p;
    a = __uint_as_float(v << 16); b = __uint_as_float(v & 0xffff0000u);
}
template <> __device__ __forceinline__ void ld2<float>(const float* p, float& a, float& b) {
    const float2 v = *(const float2*)p;
    a = v.x; b = v.y;
}
template <typename T> __device__ __forceinline__ void st2(T* p, float a, float b);
template <> __device__ __forceinline__ void st2<bf16>(bf16* p, float a, float b) {
    *(unsigned*)p = (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
}
template <> __device__ __forceinline__ void st2<float>(float* p, float a, float b) { *(float2*)p = make_float2(a, b); }

template <typename T>
__global__ void k_dwconv5(const T* __restrict__ x, const float* __restrict__ w,
                          const float* __restrict__ bias, T* y, int H, int W, int C8, long n8, int flip,
                          int acc) {
    const int C = C8 * 8;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n8; i += gridDim.x * 256L) {
        int c0 = (int)(i % C8) * 8;
        long p = i / C8;
        int wo = (int)(p % W);
        long q = p / W;
        int ho = (int)(q % H);
        long b = q / H;
        float o[8];
        if (acc) V8<T>::ld(y + i * 8, o);
        else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = bias ? bias[c0 + j] : 0.f;
        }
        for (int kh = 0; kh < 5; ++kh) {
            int hi = ho + kh - 2;
            if (hi < 0 || hi >= H) continue;
            for (int kw = 0; kw < 5; ++kw) {
                int wi = wo + kw - 2;
                if (wi < 0 || wi >= W) continue;
                float v[8];
                V8<T>::ld(x + (((b * H + hi) * (long)W + wi) * C + c0), v);
                int tap = flip ? (4 - kh) * 5 + (4 - kw) : kh * 5 + kw;
                const float* wp = w + (long)tap * C + c0;
                float4 w0 = *(const float4*)wp, w1 = *(const float4*)(wp + 4);
                o[0] += v[0] * w0.x; o[1] += v[1] * w0.y; o[2] += v[2] * w0.z; o[3] += v[3] * w0.w;
                o[4] += v[4] * w1.x; o[5] += v[5] * w1.y; o[6] += v[6] * w1.z; o[7] += v[7] * w1.w;
            }
        }
        V8<T>::st(y + i * 8, o);
    }
}

// LDS-resident variant for H*W <= 64 (the 4x4 / 8x8 towers): a workgroup owns 128 channels of a chunk
// of images; the [HW][128] input slab is staged in LDS once per image, each thread keeps the 25 taps of
// its two channels in registers and produces the outputs of every 4th pixel.
template <typename T>
__global__ __launch_bounds__(256) void k_dwconv5_lds(const T* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, T* y, int B, int H, int W,
                                                     int C, int flip, int acc, int imgs_per_block) {
    constexpr int CS = 128;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    T* sx = (T*)smem;
    const int HW = H * W;
    const int c_base = blockIdx.x * CS;
    const int cp = threadIdx.x & 63, pg = threadIdx.x >> 6;
    const int c = c_base + 2 * cp;
    const bool cval = c < C;
    float wr[25][2];
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const int tap = flip ? 24 - t : t;
        wr[t][0] = cval ? w[(long)tap * C + c] : 0.f;
        wr[t][1] = cval ? w[(long)tap * C + c + 1] : 0.f;
    }
    const float b0 = (bias && cval) ? bias[c] : 0.f, b1 = (bias && cval) ? bias[c + 1] : 0.f;
    constexpr int VE = 16 / (int)sizeof(T);
    const int chunks_per_row = CS / VE, nchunks = HW * chunks_per_row;
    const int img0 = blockIdx.y * imgs_per_block;
    int img1 = img0 + imgs_per_block;
    if (img1 > B) img1 = B;
    for (int b = img0; b < img1; ++b) {
        __syncthreads();
        for (int q = threadIdx.x; q < nchunks; q += 256) {
            int p = q / chunks_per_row, cc = (q - p * chunks_per_row) * VE;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (c_base + cc < C) v = *(const uint4*)(x + ((long)b * HW + p) * C + c_base + cc);
            *(uint4*)(sx + p * CS + cc) = v;
        }
        __syncthreads();
        if (!cval) continue;
        for (int p = pg; p < HW; p += 4) {
            const int h = p / W, wv = p - h * W;
            T* yp = y + ((long)b * HW + p) * C + c;
            float a0 = b0, a1 = b1;
            if (acc) ld2<T>(yp, a0, a1);
#pragma unroll
            for (int kh = 0; kh < 5; ++kh) {
                const int hi = h + kh - 2;
                if (hi < 0 || hi >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) {
                    const int wi = wv + kw - 2;
                    if (wi < 0 || wi >= W) continue;
                    float x0, x1;
                    ld2<T>(sx + (hi * W + wi) * CS + 2 * cp, x0, x1);
                    a0 += x0 * wr[kh * 5 + kw][0];
                    a1 += x1 * wr[kh * 5 + kw][1];
                }
            }
            st2<T>(yp, a0, a1);
        }
    }
}

extern "C" int nvae_dwconv5(int dtype, const void* x, const float* w, const float* bias, void* y, int B,
                            int H, int W, int C, int flip, int accumulate, void* stream) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "dwconv5: bad shape");
    NVAE_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w), "dwconv5: alignment");
    if (H * W <= 64) {
        const int strips = cdiv(C, 128);
        int want = 1024 / strips;
        if (want < 1) want = 1;
        int ipb = cdiv(B, want);
        if (ipb < 1) ipb = 1;
        dim3 grid(strips, cdiv(B, ipb));
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_dwconv5_lds<T>), grid, 256, (size_t)H * W * 128 * sizeof(T), (hipStream_t)stream, (const T*)x, w, bias, (T*)y, B, H, W, C, flip, accumulate, ipb);)
        NVAE_LAUNCH_CHECK("dwconv5_lds");
        return NVAE_OK;
    }
    long n8 = (long)B * H * W * (C / 8);
    long g = (n8 + 255) / 256;
    if (g > 4096) g = 4096;
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_dwconv5<T>), (int)g, 256, 0, (hipStream_t)stream, (const T*)x, w, bias, (T*)y, H, W, C / 8, n8, flip, accumulate);)
    NVAE_LAUNCH_CHECK("dwconv5");
    return NVAE_OK;
}

// dw[kh,kw,c] += sum_{b,h,w} x[b,h+kh-2,w+kw-2,c] * dy[b,h,w,c];  db[c] += sum dy.
// Thread = one channel (lanes over consecutive channels: coalesced), block.y = chunk of images;
// 25 taps + bias accumulate in registers, one atomic per (tap, channel) per block.
template <typename T>
__global__ void k_dwconv5_wgrad(const T* __restrict__ x, const T* __restrict__ dy, float* dw, float* db,
                                int B, int H, int W, int C, int imgs_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const int b0 = blockIdx.y * imgs_per_block;
    int b1 = b0 + imgs_per_block;
    if (b1 > B) b1 = B;
    float acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = 0.f;
    float ab = 0.f;
    for (int b = b0; b < b1; ++b)
        for (int h = 0; h < H; ++h)
            for (int wv = 0; wv < W; ++wv) {
                float g = ldf<T>(dy + (((long)b * H + h) * W + wv) * C + c);
                ab += g;
#pragma unroll
                for (int kh = 0; kh < 5; ++kh) {
                    int hi = h + kh - 2;
                    if (hi < 0 || hi >= H) continue;
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw) {
                        int wi = wv + kw - 2;
                        if (wi < 0 || wi >= W) continue;
                        acc[kh * 5 + kw] += g * ldf<T>(x + (((long)b * H + hi) * W + wi) * C + c);
                    }
                }
            }
#pragma unroll
    for (int t = 0; t < 25; ++t) atomicAdd(dw + (long)t * C + c, acc[t]);
    if (db) atomicAdd(db + c, ab);
}

// LDS-resident variant for the shapes on the path (H*W <= 64): a workgroup owns 128 channels and a
// chunk of images; per image the [HW][128] slabs of x and dy are staged in LDS once and every
// (tap, channel) product reads them from there.  Thread = channel pair x tap group (taps tg, tg+4, ..),
// lanes over consecutive channel pairs (conflict-free 4-B LDS reads).
template <typename T>
__global__ __launch_bounds__(256) void k_dwconv5_wgrad_lds(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* dw, float* db, int B, int H, int W, int C,
                                                           int imgs_per_block) {
    constexpr int CS = 128;                        // channels per workgroup
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int HW = H * W;
    T* sx = (T*)smem;
    T* sdy = sx + HW * CS;
    const int c_base = blockIdx.x * CS;
    const int cp = threadIdx.x & 63, tg = threadIdx.x >> 6;      // channel pair, tap group
    const int b0 = blockIdx.y * imgs_per_block;
    int b1 = b0 + imgs_per_block;
    if (b1 > B) b1 = B;
    float acc[7][2];
#pragma unroll
    for (int t = 0; t < 7; ++t) acc[t][0] = acc[t][1] = 0.f;
    float ab0 = 0.f, ab1 = 0.f;
    constexpr int VE = 16 / sizeof(T);
    const int chunks_per_row = CS / VE;
    const int nchunks = HW * chunks_per_row;
    for (int b = b0; b < b1; ++b) {
        __syncthreads();
        for (int q = threadIdx.x; q < nchunks; q += 256) {
            int p = q / chunks_per_row, cc = (q - p * chunks_per_row) * VE;
            uint4 vx = make_uint4(0, 0, 0, 0), vd = make_uint4(0, 0, 0, 0);
            if (c_base + cc < C) {
                long off = ((long)b * HW + p) * C + c_base + cc;
                vx = *(const uint4*)(x + off);
                vd = *(const uint4*)(dy + off);
            }
            *(uint4*)(sx + p * CS + cc) = vx;
            *(uint4*)(sdy + p * CS + cc) = vd;
        }
        __syncthreads();
        for (int p = 0; p < HW; ++p) {
            const int h = p / W, wv = p - h * W;
            float g0, g1;
            ld2<T>(sdy + p * CS + 2 * cp, g0, g1);
            if (tg == 0) { ab0 += g0; ab1 += g1; }
#pragma unroll
            for (int t = 0; t < 7; ++t) {
                const int tap = tg + 4 * t;
                if (tap >= 25) continue;
                const int kh = tap / 5, kw = tap - kh * 5;
                const int hi = h + kh - 2, wi = wv + kw - 2;
                if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                float x0, x1;
                ld2<T>(sx + (hi * W + wi) * CS + 2 * cp, x0, x1);
                acc[t][0] += g0 * x0;
                acc[t][1] += g1 * x1;
            }
        }
    }
    const int c = c_base + 2 * cp;
    if (c < C) {
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            const int tap = tg + 4 * t;
            if (tap >= 25) continue;
            atomicAdd(dw + (long)tap * C + c, acc[t][0]);
            atomicAdd(dw + (long)tap * C + c + 1, acc[t][1]);
        }
        if (db && tg == 0) { atomicAdd(db + c, ab0); atomicAdd(db + c + 1, ab1); }
    }
}

extern "C" int nvae_dwconv5_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, int B,
                                  int H, int W, int C, void* stream) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && x && dy && dw, "dwconv5_wgrad: bad args");
    if (H * W <= 64 && C % 8 == 0 && aligned16(x) && aligned16(dy)) {
        const int strips = cdiv(C, 128);
        int want = 768 / strips;
        if (want < 1) want = 1;
        int ipb = cdiv(B, want);
        if (ipb < 1) ipb = 1;
        dim3 grid(strips, cdiv(B, ipb));
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_dwconv5_wgrad_lds<T>), grid, 256, (size_t)2 * H * W * 128 * sizeof(T), (hipStream_t)stream, (const T*)x, (const T*)dy, dw, db, B, H, W, C, ipb);)
        NVAE_LAUNCH_CHECK("dwconv5_wgrad_lds");
        return NVAE_OK;
    }
    int cblocks = cdiv(C, 256);
    int want = 512 / cblocks;
    if (want < 1) want = 1;
    int ipb = cdiv(B, want);
    if (ipb < 1) ipb = 1;
    dim3 grid(cblocks, cdiv(B, ipb));
    DISPATCH_T(dtype, hipLaunchKernelGGL((k_dwconv5_wgrad<T>), grid, 256, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, db, B, H, W, C, ipb);)
    NVAE_LAUNCH_CHECK("dwconv5_wgrad");
    return NVAE_OK;
}
