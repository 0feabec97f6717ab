// Depthwise 5x5 convolution (decoder.py:130), NHWC, vectorised over channels.
// Spatial extents on this path are 4x4 / 8x8 with 768-1536 channels, so a whole image slab is a few
// hundred KB and the 25-tap re-reads are served by L1/L2: the kernel is bound by one HBM read + one
// write of the tensor.  Weights are the f32 masters [5,5,C] (25*C*4 B, L2-resident).
#include "common.h"

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

extern "C" int nvae_dwconv5(int dtype, const void* x, const float* w, const float* bias, void* y, int B,
                            int H, int W, int C, int flip, int accumulate, void* stream) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "dwconv5: bad shape");
    NVAE_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w), "dwconv5: alignment");
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

extern "C" int nvae_dwconv5_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, int B,
                                  int H, int W, int C, void* stream) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && x && dy && dw, "dwconv5_wgrad: bad args");
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
