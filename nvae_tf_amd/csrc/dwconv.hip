// Depthwise 5x5 convolution (decoder.py:130), NHWC: forward / data gradient (same kernel, flipped taps)
// and weight gradient.  HBM-bound: one read + one write of the tensor (forward), two reads (wgrad).
//
// A workgroup owns a TH x TW tile of output pixels of one image and a strip of 64 channels (128 B of
// bf16 per pixel = one cache line).  The (TH+4) x (TW+4) input halo is staged once in LDS with 16-B
// loads (zero outside the image: 'same' padding costs nothing afterwards), then thread (cg, pl) works
// on 4 channels (one 8-B / 16-B LDS read) for pixel lane pl:
//   forward: R vertically adjacent outputs per thread, so each halo row read feeds up to R outputs
//            (10 LDS reads per output instead of 25 for R = 4); the 25 x 4 taps live in registers;
//   wgrad:   4 horizontally adjacent dy pixels per thread against an 8-wide x row per tap row
//            (11 LDS reads per pixel), 25 x 4 accumulators in registers across every tile the
//            workgroup visits, one shuffle + LDS reduction and one atomic per (tap, channel) at the end.
// Tiles: 8x8 (R = 4) for H, W > 4, 4x4 (R = 1) for the 4x4 tower.  Weights are the f32 masters [5,5,C].
//
// Measured on MI355X before this layout (thread = channel pair / pixel, taps re-read from L1/L2): the
// CIFAR-10 shape B64 x 16x16 x 1536 took 173 us forward and 1 180 us for the weight gradient, i.e.
// 0.6 / 0.08 TB/s of algorithmic traffic.
#include "common.h"
#include "bn_fin.h"
#include "conv_common.h"      // glds16 (LDS-DMA), wait_vmcnt, zero_page

#define DW_CC 64          // channels per workgroup
__device__ __forceinline__ int cdiv_dev(int a, int b) { return (a + b - 1) / b; }

template <typename T> struct DwVec;     // 4 channels <-> float[4]
template <> struct DwVec<bf16> {
    static __device__ __forceinline__ void ld(const bf16* p, float (&v)[4]) {
        const uint2 u = *(const uint2*)p;
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16* p, const float (&v)[4]) {
        uint2 u;
        u.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
        u.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
        *(uint2*)p = u;
    }
};
template <> struct DwVec<f16> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ void ld(const f16* p, float (&v)[4]) {
        const h4 u = *(const h4*)p;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)u[j];
    }
    static __device__ __forceinline__ void st(f16* p, const float (&v)[4]) {
        h4 u;
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = (f16)v[j];
        *(h4*)p = u;
    }
};
template <> struct DwVec<float> {
    static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
        const float4 u = *(const float4*)p;
        v[0] = u.x; v[1] = u.y; v[2] = u.z; v[3] = u.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
        *(float4*)p = make_float4(v[0], v[1], v[2], v[3]);
    }
};

// Stage rows [y0, y0+NH) x cols [x0, x0+NW) x channels [c_base, c_base+64) of image b into LDS
// ([pixel][64] T, zero outside the image / beyond C).
template <typename T, int NH, int NW>
__device__ __forceinline__ void dw_stage(T* __restrict__ s, const T* __restrict__ src, long b, int H, int W,
                                         int C, int y0, int x0, int c_base) {
    constexpr int VE = 16 / (int)sizeof(T);          // elements per 16-B chunk
    constexpr int CPP = DW_CC / VE;                  // chunks per pixel
    for (int q = threadIdx.x; q < NH * NW * CPP; q += 256) {
        const int pix = q / CPP, cc = (q - pix * CPP) * VE;
        const int py = pix / NW, px = pix - py * NW;
        const int gy = y0 + py, gx = x0 + px;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + cc < C)
            v = *(const uint4*)(src + (((b * H + gy) * (long)W + gx) * C + c_base + cc));
        *(uint4*)(s + pix * DW_CC + cc) = v;
    }
}

// o[r] += sum over the 5x5 window of output (row0 + r, col) for R vertically adjacent outputs: each halo
// row is read once (5 LDS reads of 4 channels) and feeds every output whose window contains it.
template <typename T, int HTW, int R>
__device__ __forceinline__ void dw5_rows(const T* __restrict__ sx, const float (&wr)[25][4], float (&o)[R][4],
                                         int row0, int col, int cg) {
#pragma unroll
    for (int hr = 0; hr < R + 4; ++hr) {                 // halo row row0 + hr feeds outputs hr-4 .. hr
        float xin[5][4];
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) DwVec<T>::ld(sx + ((row0 + hr) * HTW + col + kw) * DW_CC + cg * 4, xin[kw]);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int kh = hr - r;
            if (kh < 0 || kh > 4) continue;
#pragma unroll
            for (int kw = 0; kw < 5; ++kw)
#pragma unroll
                for (int j = 0; j < 4; ++j) o[r][j] += xin[kw][j] * wr[kh * 5 + kw][j];
        }
    }
}

template <typename T, int TH, int TW, int R>
__global__ __launch_bounds__(256) void k_dw5_fwd(const T* __restrict__ x, const float* __restrict__ w,
                                                 const float* __restrict__ bias, T* y, int H, int W, int C,
                                                 int tiles_x, int flip, int acc) {
    static_assert(TW * (TH / R) == 16 && TH % R == 0, "16 pixel lanes per workgroup");
    constexpr int HTH = TH + 4, HTW = TW + 4;
    __shared__ __attribute__((aligned(16))) T sx[HTH * HTW * DW_CC];
    const int c_base = blockIdx.x * DW_CC;
    const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
    const long b = blockIdx.z;
    const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = c_base + cg * 4;
    const bool cval = c < C;
    // taps of this thread's 4 channels (issued before the staging so both latencies overlap)
    float wr[25][4];
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const int tap = flip ? 24 - t : t;
        const float4 u = cval ? *(const float4*)(w + (long)tap * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        wr[t][0] = u.x; wr[t][1] = u.y; wr[t][2] = u.z; wr[t][3] = u.w;
    }
    dw_stage<T, HTH, HTW>(sx, x, b, H, W, C, ty * TH - 2, tx * TW - 2, c_base);
    __syncthreads();
    if (!cval) return;
    const int col = pl % TW, row0 = (pl / TW) * R;       // tile coordinates of the first output
    const int gx = tx * TW + col;
    float o[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = ty * TH + row0 + r;
        if (acc && gy < H && gx < W) DwVec<T>::ld(y + (((b * H + gy) * (long)W + gx) * C + c), o[r]);
        else {
#pragma unroll
            for (int j = 0; j < 4; ++j) o[r][j] = (bias && !acc) ? bias[c + j] : 0.f;
        }
    }
    dw5_rows<T, TW + 4, R>(sx, wr, o, row0, col, cg);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = ty * TH + row0 + r;
        if (gy < H && gx < W) DwVec<T>::st(y + (((b * H + gy) * (long)W + gx) * C + c), o[r]);
    }
}

// bf16 production variant.  Persistent workgroups walk "units" (one 8x8 tile of one image, or four whole
// 4x4 images); the halo of the next unit is in flight (LDS-DMA, global_load_lds_dwordx4: no VGPRs,
// out-of-image chunks read a zero page) while the current one is computed, and the taps are loaded once
// per workgroup.  Thread = channel PAIR (one 4-B LDS read, conflict-free across the 32 pairs of a strip)
// x a 2-row x 4-column block of outputs: 48 halo reads feed 8 outputs (6 per output instead of 25), the
// 25 x 2 taps + 16 accumulators fit in ~100 VGPRs, so four workgroups share a CU and hide each other's
// LDS latency.  The FMAs compile to v_pk_fma_f32 on the channel pair.
typedef float dw_f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ dw_f2 dw_unpack(unsigned v);      // 16-bit pair -> f32 pair
template <> __device__ __forceinline__ dw_f2 dw_unpack<bf16>(unsigned v) {
    dw_f2 r;
    r.x = __uint_as_float(v << 16); r.y = __uint_as_float(v & 0xffff0000u);
    return r;
}
template <> __device__ __forceinline__ dw_f2 dw_unpack<f16>(unsigned v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = __builtin_bit_cast(h2, v);
    dw_f2 r;
    r.x = (float)h[0]; r.y = (float)h[1];
    return r;
}
template <typename T> __device__ __forceinline__ unsigned dw_pack(dw_f2 v);
template <> __device__ __forceinline__ unsigned dw_pack<bf16>(dw_f2 v) { return (unsigned)f2bf(v.x) | ((unsigned)f2bf(v.y) << 16); }
template <> __device__ __forceinline__ unsigned dw_pack<f16>(dw_f2 v) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    h2 h; h[0] = (f16)v.x; h[1] = (f16)v.y;
    return __builtin_bit_cast(unsigned, h);
}


// Operand prologue of the 16-bit ring kernels (round 2): the input is act(BN(x)) of the RAW tensor x, never
// materialised.  A thread DMA's 16-B chunks of a fixed 8-channel group (q & 7 == lane & 7), so it keeps that group's
// scale / shift in registers and transforms ITS OWN chunks in LDS right after its vmcnt wait and before the barrier
// that publishes the stage (no extra barrier); out-of-image chunks (the 'same' padding, DMA'd from the zero page) stay
// zero, because the padding applies to the activated tensor.  The whole image is one tile at 4x4 / 8x8, so every
// element is transformed exactly once per pass.
struct DwPre { BnFromSlab bn; int on; int act; };
// Data-gradient launches whose output is d act(BN(x0)) for a BatchNorm this conv was the only consumer of: the kernel also
// accumulates that BatchNorm's backward sums (sum dpre, sum dpre * x0 with dpre = dx * act'(scale * x0 + shift)) into the slab
// `stats`, read by nvae_bn_bwd_apply_fin - the counterpart of the BN-backward epilogue of the implicit GEMM (conv_gemm.hip).
struct DwBnBwd { const void* x0; const float* scale; const float* shift; int act; };
template <typename T>
__device__ __forceinline__ void dw_pre_coefs(const DwPre& pre, int C, int c_base, float (&sc)[8], float (&sh)[8]) {
    __shared__ float t_sc[DW_CC], t_sh[DW_CC];
    if (threadIdx.x < DW_CC) {
        const int c = c_base + threadIdx.x;
        float a = 0.f, b = 0.f;
        if (c < C) bn_coef<sizeof(T) == 4>(pre.bn, C, c, blockIdx.y == 0, a, b);
        t_sc[threadIdx.x] = a; t_sh[threadIdx.x] = b;
    }
    __syncthreads();
    const int g8 = (threadIdx.x & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) { sc[j] = t_sc[g8 + j]; sh[j] = t_sh[g8 + j]; }
}
template <typename T>
__device__ __forceinline__ uint4 dw_pre_chunk(uint4 raw, const float (&sc)[8], const float (&sh)[8], int act) {
    float v[8];
    unpack8<T>(raw, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float p = v[j] * sc[j] + sh[j];
        v[j] = act == ACT_SWISH ? swishf_(p) : p;
    }
    return pack8<T>(v);
}

template <typename T, int TH, int TW, int IMGS>
__global__ __launch_bounds__(256, 2) void k_dw5_fwd_ring(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, T* y, int B, int H,
                                                      int W, int C, int tiles_x, int tiles_per_img, int flip,
                                                      int acc, const uint4* __restrict__ zeros,
                                                      float* __restrict__ stats, DwPre pre, DwBnBwd bb) {
    static_assert(IMGS * (TH / 2) * (TW / 4) == 8, "8 pixel lanes of 2x4 outputs per workgroup");
    constexpr int NS = 2;
    constexpr int HTH = TH + 4, HTW = TW + 4, NCH = IMGS * HTH * HTW * 8;     // 16-B chunks per unit
    constexpr int KI = (NCH + 255) / 256;            // DMA instructions per thread and stage
    constexpr int STAGE = KI * 256;                  // chunks per stage (tail chunks read zeros)
    __shared__ uint4 lds[NS * STAGE];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_base = blockIdx.x * DW_CC;
    const int cp = tid & 31, pl = tid >> 5;
    const int c = c_base + cp * 2;
    const bool cval = c < C;
    dw_f2 wr[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        const int tap = (flip & 1) ? 24 - t : t;
        const float2 u = cval ? *(const float2*)(w + (long)tap * C + c) : make_float2(0.f, 0.f);
        wr[t].x = u.x; wr[t].y = u.y;
    }
    dw_f2 bv;
    bv.x = (bias && !acc && cval) ? bias[c] : 0.f;
    bv.y = (bias && !acc && cval) ? bias[c + 1] : 0.f;
    const T* bx0 = (const T*)bb.x0;
    dw_f2 bsc = {0.f, 0.f}, bsh = {0.f, 0.f};
    if (bx0 && cval) { bsc.x = bb.scale[c]; bsc.y = bb.scale[c + 1]; bsh.x = bb.shift[c]; bsh.y = bb.shift[c + 1]; }
    const long units = IMGS == 1 ? (long)B * tiles_per_img : (long)(B + IMGS - 1) / IMGS;
    const int nmine = blockIdx.y < units ? (int)((units - blockIdx.y + gridDim.y - 1) / gridDim.y) : 0;
    auto unit_of = [&](int i, long& b, int& ty, int& tx) {      // first image and tile of unit i
        const long u = blockIdx.y + (long)i * gridDim.y;
        if (IMGS == 1) {
            b = u / tiles_per_img;
            const int tile = (int)(u - b * tiles_per_img);
            ty = tile / tiles_x; tx = tile - ty * tiles_x;
        } else {
            b = u * IMGS; ty = 0; tx = 0;
        }
    };
    auto issue = [&](int i) -> unsigned {       // returns the mask of this thread's chunks that hold image data
        long b; int ty, tx;
        unit_of(i, b, ty, tx);
        const unsigned dst = lds_base + (unsigned)((i % NS) * STAGE) * 16u;
        unsigned mask = 0;
#pragma unroll
        for (int k = 0; k < KI; ++k) {
            const int q = (k * 4 + wave) * 64 + lane;
            const int pix = q >> 3, cc = (q & 7) * 8;
            const int img = pix / (HTH * HTW), pi = pix - img * (HTH * HTW);
            const int py = pi / HTW, px = pi - py * HTW;
            const int gy = ty * TH - 2 + py, gx = tx * TW - 2 + px;
            const bool ok = q < NCH && b + img < B && gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + cc < C;
            const void* p = ok ? (const void*)(x + ((((b + img) * H + gy) * (long)W + gx) * C + c_base + cc)) : (const void*)zeros;
            glds16(p, dst + (unsigned)((k * 4 + wave) * 64) * 16u);
            mask |= (ok ? 1u : 0u) << k;
        }
        return mask;
    };
    float psc[8], psh[8];
    if (pre.on) dw_pre_coefs<T>(pre, C, c_base, psc, psh);
    unsigned mask_cur = 0, mask_next = 0;
    if (nmine > 0) mask_cur = issue(0);
    // pixel lane -> (image within the unit, 2-row block, 4-column block)
    constexpr int CB = TW / 4, RB = TH / 2;
    const int cb = pl % CB, rb = (pl / CB) % RB, img = pl / (CB * RB);
    dw_f2 st1 = {0.f, 0.f}, st2 = {0.f, 0.f};       // BatchNorm statistics of this thread's outputs (stats != NULL)
    for (int i = 0; i < nmine; ++i) {
        wait_vmcnt<0>();             // this wave's part of unit i has landed (and its older y stores)
        if (pre.on) {
            uint4* st = lds + (i % NS) * STAGE;
#pragma unroll
            for (int k = 0; k < KI; ++k)
                if (mask_cur & (1u << k)) {
                    uint4* slot = st + (k * 4 + wave) * 64 + lane;
                    *slot = dw_pre_chunk<T>(*slot, psc, psh, pre.act);
                }
        }
        __syncthreads();             // everyone's part is in LDS; everyone finished reading stage (i+1) % 2
        if (i + 1 < nmine) mask_next = issue(i + 1);
        const unsigned mask_keep = mask_next;
        long b; int ty, tx;
        unit_of(i, b, ty, tx);
        b += img;
        const T* sx = (const T*)(lds + (i % NS) * STAGE) + (long)img * HTH * HTW * DW_CC;
        const int gy0 = ty * TH + rb * 2, gx0 = tx * TW + cb * 4;
        const bool live = cval && b < B;
        dw_f2 o[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                o[r][p] = bv;
                if (acc && live && gy0 + r < H && gx0 + p < W)
                    o[r][p] = dw_unpack<T>(*(const unsigned*)(y + (((b * H + gy0 + r) * (long)W + gx0 + p) * C + c)));
            }
        unsigned x0v[2][4];
        if (bx0) {        // the BatchNorm input at this thread's outputs, requested before the tap loop
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    x0v[r][p] = 0u;
                    if (live && gy0 + r < H && gx0 + p < W)
                        x0v[r][p] = *(const unsigned*)(bx0 + (((b * H + gy0 + r) * (long)W + gx0 + p) * C + c));
                }
        }
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {             // halo row rb*2 + hr feeds output rows hr-4 .. hr
            dw_f2 xr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                xr[q] = dw_unpack<T>(*(const unsigned*)(sx + ((rb * 2 + hr) * HTW + cb * 4 + q) * DW_CC + cp * 2));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int kh = hr - r;
                if (kh < 0 || kh > 4) continue;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw)
                        o[r][p] = __builtin_elementwise_fma(xr[p + kw], wr[kh * 5 + kw], o[r][p]);
            }
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (gy0 + r < H && gx0 + p < W) {
                        *(unsigned*)(y + (((b * H + gy0 + r) * (long)W + gx0 + p) * C + c)) =
                            dw_pack<T>(o[r][p]);
                        if (bx0) {
                            const dw_f2 xv = dw_unpack<T>(x0v[r][p]);
                            dw_f2 dpre = o[r][p];
                            if (bb.act == ACT_SWISH) {
                                const dw_f2 pa = xv * bsc + bsh;
                                dpre.x *= dswishf_(pa.x); dpre.y *= dswishf_(pa.y);
                            }
                            st1 += dpre;
                            st2 += dpre * xv;
                        } else {
                            st1 += o[r][p];
                            st2 += o[r][p] * o[r][p];
                        }
                    }
        }
        mask_cur = mask_keep;
    }
    if (stats) {
        // per-workgroup column sums, accumulated into the (zeroed) slab nvae_bn_finalize_s / nvae_bn_apply_fin consume
        st1.x += __shfl_xor(st1.x, 32, 64); st1.y += __shfl_xor(st1.y, 32, 64);
        st2.x += __shfl_xor(st2.x, 32, 64); st2.y += __shfl_xor(st2.y, 32, 64);
        __syncthreads();                 // the ring is dead: reuse it as [4 waves][32 pairs][4]
        float* red = (float*)lds;
        if (lane < 32) {
            float* r = red + (wave * 32 + lane) * 4;
            r[0] = st1.x; r[1] = st1.y; r[2] = st2.x; r[3] = st2.y;
        }
        __syncthreads();
        if (tid < 64) {
            const int pr = tid >> 1, e = tid & 1;
            const int cc = c_base + pr * 2 + e;
            if (cc < C) {
                float a1 = 0.f, a2 = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) { a1 += red[(wv * 32 + pr) * 4 + e]; a2 += red[(wv * 32 + pr) * 4 + 2 + e]; }
                // workgroup y adds into row y % rows of the zeroed slab (<= 64 adders per address, conv_gemm.hip)
                const int row = (flip & 2) ? (int)blockIdx.y : blockIdx.y % cdiv_dev((int)gridDim.y, 64);   // bit 1: one row per workgroup
                atomicAdd(stats + ((long)row * 2) * C + cc, a1);
                atomicAdd(stats + ((long)row * 2 + 1) * C + cc, a2);
            }
        }
    }
}

// persistent workgroups per channel strip of the bf16 ring kernel (= rows of its statistics slab)
static long dw_ring_rows(int B, int H, int W, int C) {
    const bool small = H <= 4 && W <= 4;
    const long units = small ? (B + 3) / 4 : (long)B * cdiv(W, 8) * cdiv(H, 8);
    long nb = 1024 / cdiv(C, DW_CC);     // 4 persistent workgroups per CU in flight
    if (nb < 1) nb = 1;
    if (nb > units) nb = units;
    return nb;
}

extern "C" int nvae_dwconv5_stats_rows(int dtype, int B, int H, int W, int C) {
    if (!is16(dtype) || B <= 0 || H <= 0 || W <= 0 || C < 8 || C % 8) return 0;
    return slab_rows_for(dw_ring_rows(B, H, W, C));
}

static int dw_pre_from(const char* who, const NvaeBnIn* in, int act, long rows, DwPre& pre) {
    pre = DwPre{};
    if (!in) return NVAE_OK;
    NVAE_REQUIRE(in->scale && in->shift, "%s: NvaeBnIn needs scale / shift", who);
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "%s: act %d unsupported", who, act);
    pre.on = 1; pre.act = act;
    pre.bn.scale = in->scale; pre.bn.shift = in->shift; pre.bn.mean = in->mean; pre.bn.invstd = in->invstd;
    if (in->slab) {
        NVAE_REQUIRE(in->rows > 0 && in->gamma && in->beta && in->rm && in->rv && in->mean && in->invstd,
                     "%s: NvaeBnIn with a slab needs rows, gamma, beta, rm, rv, mean, invstd", who);
        pre.bn.slab = in->slab; pre.bn.rows = in->rows; pre.bn.inv_n = 1.0f / (float)rows; pre.bn.eps = in->eps;
        pre.bn.momentum = in->momentum; pre.bn.gamma = in->gamma; pre.bn.beta = in->beta; pre.bn.rm = in->rm; pre.bn.rv = in->rv;
    }
    return NVAE_OK;
}

static int dwconv5_impl(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H, int W,
                        int C, int flip, int accumulate, float* stats, void* stream, const DwPre& pre = DwPre{},
                        const DwBnBwd& bb = DwBnBwd{}) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0, "dwconv5: bad shape");
    NVAE_REQUIRE(aligned16(x) && aligned16(y) && aligned16(w) && (!bias || aligned16(bias)), "dwconv5: alignment");
    NVAE_REQUIRE(!stats || (is16(dtype) && !accumulate && (!flip || bb.x0)), "dwconv5: statistics only from the 16-bit forward (or BatchNorm-backward sums from the data gradient)");
    NVAE_REQUIRE(!bb.x0 || (stats && flip && is16(dtype)), "dwconv5_bnbwd: needs a slab, a data-gradient launch and a 16-bit activation type");
    const int strips = cdiv(C, DW_CC);
    NVAE_REQUIRE(B <= 65535, "dwconv5: batch too large for the grid");
    if (is16(dtype)) {
        const bool small = H <= 4 && W <= 4;
        const int tx = small ? 1 : cdiv(W, 8), ty = small ? 1 : cdiv(H, 8);
        dim3 grid(strips, (unsigned)dw_ring_rows(B, H, W, C));
#define DW_RING(T_)                                                                                              \
        if (small)                                                                                               \
            hipLaunchKernelGGL((k_dw5_fwd_ring<T_, 4, 4, 4>), grid, 256, 0, (hipStream_t)stream, (const T_*)x, w, bias, (T_*)y, B, H, W, C, tx, tx * ty, flip | (g_nvae_det ? 2 : 0), accumulate, zero_page(), stats, pre, bb); \
        else                                                                                                     \
            hipLaunchKernelGGL((k_dw5_fwd_ring<T_, 8, 8, 1>), grid, 256, 0, (hipStream_t)stream, (const T_*)x, w, bias, (T_*)y, B, H, W, C, tx, tx * ty, flip | (g_nvae_det ? 2 : 0), accumulate, zero_page(), stats, pre, bb);
        if (dtype == NVAE_BF16) { DW_RING(bf16) } else { DW_RING(f16) }
#undef DW_RING
        NVAE_LAUNCH_CHECK("dwconv5");
        return NVAE_OK;
    }
    NVAE_REQUIRE(!pre.on, "dwconv5_pre: the operand prologue exists for the 16-bit activation types only");
    if (H <= 4 && W <= 4) {
        dim3 grid(strips, 1, B);
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_dw5_fwd<T, 4, 4, 1>), grid, 256, 0, (hipStream_t)stream, (const T*)x, w, bias, (T*)y, H, W, C, 1, flip, accumulate);)
    } else {
        const int tx = cdiv(W, 8), ty = cdiv(H, 8);
        NVAE_REQUIRE((long)tx * ty <= 65535 && B <= 65535, "dwconv5: image too large for the tile grid");
        dim3 grid(strips, tx * ty, B);
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_dw5_fwd<T, 8, 8, 4>), grid, 256, 0, (hipStream_t)stream, (const T*)x, w, bias, (T*)y, H, W, C, tx, flip, accumulate);)
    }
    NVAE_LAUNCH_CHECK("dwconv5");
    return NVAE_OK;
}

extern "C" int nvae_dwconv5(int dtype, const void* x, const float* w, const float* bias, void* y, int B,
                            int H, int W, int C, int flip, int accumulate, void* stream) {
    return dwconv5_impl(dtype, x, w, bias, y, B, H, W, C, flip, accumulate, nullptr, stream);
}

// Forward pass that also emits the BatchNorm statistics of its output: stats[rows][2][C] with
// rows = nvae_dwconv5_stats_rows(...) > 0 (per-workgroup column sums and sums of squares, taken from the
// f32 accumulators), to be consumed by nvae_bn_finalize_s.
extern "C" int nvae_dwconv5_stats(int dtype, const void* x, const float* w, const float* bias, void* y, int B,
                                  int H, int W, int C, float* stats, void* stream) {
    NVAE_REQUIRE(stats, "dwconv5_stats: NULL statistics slab");
    return dwconv5_impl(dtype, x, w, bias, y, B, H, W, C, 0, 0, stats, stream);
}

// Forward pass on act(BN(x)) of the RAW tensor x (16-bit activation types): the BatchNorm in front - given as its
// final table or as the statistics slab its producer left (NvaeBnIn, as for nvae_se_fused_fwd) - is applied to the
// halo tile in LDS, the normalised activation is never written.  stats: NULL or the output's statistics slab.
extern "C" int nvae_dwconv5_pre(int dtype, const void* x, const NvaeBnIn* bn, int act, const float* w,
                                const float* bias, void* y, int B, int H, int W, int C, float* stats, void* stream) {
    NVAE_REQUIRE(bn && is16(dtype), "dwconv5_pre: needs a BatchNorm description and a 16-bit activation type");
    DwPre pre;
    if (int e = dw_pre_from("dwconv5_pre", bn, act, (long)B * H * W, pre)) return e;
    return dwconv5_impl(dtype, x, w, bias, y, B, H, W, C, 0, 0, stats, stream, pre);
}

// Data gradient dx = dwconv5^T(dy) that also reduces the backward sums of the BatchNorm(+act) whose OUTPUT dx is the gradient of
// (x0 = that BatchNorm's input, scale / shift = its final coefficients): partials[rows][2][C], rows =
// nvae_dwconv5_stats_rows(...), zeroed by the caller, consumed by nvae_bn_bwd_apply_fin.  16-bit activation types.
extern "C" int nvae_dwconv5_bnbwd(int dtype, const void* dy, const float* w, void* dx, int B, int H, int W, int C,
                                  const void* x0, const float* scale, const float* shift, int act, float* partials,
                                  void* stream) {
    NVAE_REQUIRE(x0 && scale && shift && partials && aligned16(x0), "dwconv5_bnbwd: NULL / unaligned argument");
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "dwconv5_bnbwd: act %d unsupported", act);
    DwBnBwd bb{x0, scale, shift, act};
    return dwconv5_impl(dtype, dy, w, nullptr, dx, B, H, W, C, 1, 0, partials, stream, DwPre{}, bb);
}

// dw[kh,kw,c] += sum_{b,h,w} x[b,h+kh-2,w+kw-2,c] * dy[b,h,w,c];  db[c] += sum dy.
// PX = dy pixels per thread (adjacent in a row): 4 for 8x8 tiles, 1 for 4x4 tiles.
template <typename T, int TH, int TW, int PX>
__global__ __launch_bounds__(256) void k_dw5_wgrad(const T* __restrict__ x, const T* __restrict__ dy,
                                                   float* dw, float* db, int B, int H, int W, int C,
                                                   int tiles_x, int tiles_per_img, int units_per_block) {
    static_assert(TH * TW == 16 * PX && TW % PX == 0, "16 pixel lanes per workgroup");
    constexpr int HTH = TH + 4, HTW = TW + 4;
    constexpr int SX = HTH * HTW * DW_CC, SD = TH * TW * DW_CC;
    constexpr int RED_BYTES = 4 * 16 * 104 * 4;        // cross-wave reduction buffer
    constexpr int STG_BYTES = (SX + SD) * (int)sizeof(T);
    __shared__ __attribute__((aligned(16))) unsigned char smem[STG_BYTES > RED_BYTES ? STG_BYTES : RED_BYTES];
    T* sx = (T*)smem;
    T* sd = sx + SX;
    const int c_base = blockIdx.x * DW_CC;
    const int cg = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int row = pl / (TW / PX), col0 = (pl % (TW / PX)) * PX;
    float acc[25][4];
#pragma unroll
    for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
    float ab[4] = {0.f, 0.f, 0.f, 0.f};
    const long units = (long)B * tiles_per_img;          // (image, tile) pairs
    long u0 = (long)blockIdx.y * units_per_block, u1 = u0 + units_per_block;
    if (u1 > units) u1 = units;
    for (long u = u0; u < u1; ++u) {
        const long b = u / tiles_per_img;
        const int tile = (int)(u - b * tiles_per_img);
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        __syncthreads();
        dw_stage<T, HTH, HTW>(sx, x, b, H, W, C, ty * TH - 2, tx * TW - 2, c_base);
        dw_stage<T, TH, TW>(sd, dy, b, H, W, C, ty * TH, tx * TW, c_base);
        __syncthreads();
        float g[PX][4];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            DwVec<T>::ld(sd + (row * TW + col0 + p) * DW_CC + cg * 4, g[p]);
#pragma unroll
            for (int j = 0; j < 4; ++j) ab[j] += g[p][j];
        }
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
            float xr[PX + 4][4];
#pragma unroll
            for (int q = 0; q < PX + 4; ++q) DwVec<T>::ld(sx + ((row + kh) * HTW + col0 + q) * DW_CC + cg * 4, xr[q]);
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int kw = 0; kw < 5; ++kw)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[kh * 5 + kw][j] += g[p][j] * xr[p + kw][j];
        }
    }
    // reduce over the 16 pixel lanes: lanes with equal cg are 16 apart (4 per wave), then 4 waves via LDS
#pragma unroll
    for (int t = 0; t < 25; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = acc[t][j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][j] = v;
        }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float v = ab[j];
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        ab[j] = v;
    }
    __syncthreads();                                     // staging buffers are dead: reuse as [4][16][104] f32
    float* red = (float*)smem;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 16) {
        float* r = red + (wave * 16 + lane) * 104;
#pragma unroll
        for (int t = 0; t < 25; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) r[t * 4 + j] = acc[t][j];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[100 + j] = ab[j];
    }
    __syncthreads();
    // 16 cg x 104 values: value e = t*4 + j of channel group g -> dw[t][c_base + 4g + j]
    for (int q = threadIdx.x; q < 16 * 104; q += 256) {
        const int gq = q / 104, e = q - gq * 104;
        const float v = red[(0 * 16 + gq) * 104 + e] + red[(1 * 16 + gq) * 104 + e] +
                        red[(2 * 16 + gq) * 104 + e] + red[(3 * 16 + gq) * 104 + e];
        const int cc = c_base + gq * 4 + (e & 3);
        if (cc >= C) continue;
        if (e < 100) atomicAdd(dw + (long)(e >> 2) * C + cc, v);
        else if (db) atomicAdd(db + cc, v);
    }
}

// bf16 production variant of the weight gradient for images larger than 4x4: same persistent LDS-DMA
// ring and thread layout as k_dw5_fwd_ring (channel pair x 2x4 block of dy pixels): per unit 48 reads
// of the x halo + 8 of dy feed 200 packed FMAs into the thread's 25 tap-pair accumulators, which live in
// registers across all the units the workgroup visits.
template <typename T, int TH, int TW, int IMGS>
__global__ __launch_bounds__(256, 2) void k_dw5_wgrad_ring(const T* __restrict__ x, const T* __restrict__ dy,
                                                           float* dw, float* db, int B, int H, int W, int C,
                                                           int tiles_x, int tiles_per_img,
                                                           const uint4* __restrict__ zeros, DwPre pre) {
    static_assert(IMGS * (TH / 2) * (TW / 4) == 8, "8 pixel lanes of 2x4 dy pixels per workgroup");
    constexpr int NS = 2;
    constexpr int HTH = TH + 4, HTW = TW + 4;
    constexpr int XCH = IMGS * HTH * HTW * 8, DCH = IMGS * TH * TW * 8;      // 16-B chunks: x halos, dy tiles
    constexpr int KX = (XCH + 255) / 256, KD = (DCH + 255) / 256;
    constexpr int STAGE = (KX + KD) * 256;
    constexpr int RED = 4 * 32 * 52 / 4;                         // reduction buffer in uint4
    __shared__ uint4 lds[NS * STAGE > RED ? NS * STAGE : RED];
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c_base = blockIdx.x * DW_CC;
    const int cp = tid & 31, pl = tid >> 5;
    constexpr int CB = TW / 4, RB = TH / 2;
    const int cb = pl % CB, rb = (pl / CB) % RB, img = pl / (CB * RB);
    dw_f2 acc[25];
#pragma unroll
    for (int t = 0; t < 25; ++t) acc[t] = dw_f2{0.f, 0.f};
    dw_f2 ab = {0.f, 0.f};
    const long units = IMGS == 1 ? (long)B * tiles_per_img : (long)(B + IMGS - 1) / IMGS;
    const int nmine = blockIdx.y < units ? (int)((units - blockIdx.y + gridDim.y - 1) / gridDim.y) : 0;
    auto issue = [&](int i) -> unsigned {       // returns the mask of this thread's x chunks that hold image data
        const long u = blockIdx.y + (long)i * gridDim.y;
        long b; int ty, tx;
        unsigned mask = 0;
        if (IMGS == 1) {
            b = u / tiles_per_img;
            const int tile = (int)(u - b * tiles_per_img);
            ty = tile / tiles_x; tx = tile - ty * tiles_x;
        } else {
            b = u * IMGS; ty = 0; tx = 0;
        }
        const unsigned dst = lds_base + (unsigned)((i % NS) * STAGE) * 16u;
#pragma unroll
        for (int k = 0; k < KX; ++k) {
            const int q = (k * 4 + wave) * 64 + lane;
            const int pix = q >> 3, cc = (q & 7) * 8;
            const int im = pix / (HTH * HTW), pi = pix - im * (HTH * HTW);
            const int py = pi / HTW, px = pi - py * HTW;
            const int gy = ty * TH - 2 + py, gx = tx * TW - 2 + px;
            const bool ok = q < XCH && b + im < B && gy >= 0 && gy < H && gx >= 0 && gx < W && c_base + cc < C;
            const void* p = ok ? (const void*)(x + ((((b + im) * H + gy) * (long)W + gx) * C + c_base + cc)) : (const void*)zeros;
            glds16(p, dst + (unsigned)((k * 4 + wave) * 64) * 16u);
            mask |= (ok ? 1u : 0u) << k;
        }
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            const int q = (k * 4 + wave) * 64 + lane;
            const int pix = q >> 3, cc = (q & 7) * 8;
            const int im = pix / (TH * TW), pi = pix - im * (TH * TW);
            const int py = pi / TW, px = pi - py * TW;
            const int gy = ty * TH + py, gx = tx * TW + px;
            const bool ok = q < DCH && b + im < B && gy < H && gx < W && c_base + cc < C;
            const void* p = ok ? (const void*)(dy + ((((b + im) * H + gy) * (long)W + gx) * C + c_base + cc)) : (const void*)zeros;
            glds16(p, dst + (unsigned)(KX * 256 + (k * 4 + wave) * 64) * 16u);
        }
        return mask;
    };
    float psc[8], psh[8];
    if (pre.on) dw_pre_coefs<T>(pre, C, c_base, psc, psh);
    unsigned mask_cur = 0, mask_next = 0;
    if (nmine > 0) mask_cur = issue(0);
    for (int i = 0; i < nmine; ++i, mask_cur = mask_next) {
        wait_vmcnt<0>();
        if (pre.on) {
            uint4* st = lds + (i % NS) * STAGE;
#pragma unroll
            for (int k = 0; k < KX; ++k)
                if (mask_cur & (1u << k)) {
                    uint4* slot = st + (k * 4 + wave) * 64 + lane;
                    *slot = dw_pre_chunk<T>(*slot, psc, psh, pre.act);
                }
        }
        __syncthreads();
        if (i + 1 < nmine) mask_next = issue(i + 1);
        const T* sx = (const T*)(lds + (i % NS) * STAGE) + (long)img * HTH * HTW * DW_CC;
        const T* sd = (const T*)(lds + (i % NS) * STAGE) + KX * 256 * 8 + (long)img * TH * TW * DW_CC;
        dw_f2 g[2][4];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                g[r][p] = dw_unpack<T>(*(const unsigned*)(sd + ((rb * 2 + r) * TW + cb * 4 + p) * DW_CC + cp * 2));
                ab += g[r][p];
            }
#pragma unroll
        for (int hr = 0; hr < 6; ++hr) {
            dw_f2 xr[8];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                xr[q] = dw_unpack<T>(*(const unsigned*)(sx + ((rb * 2 + hr) * HTW + cb * 4 + q) * DW_CC + cp * 2));
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int kh = hr - r;
                if (kh < 0 || kh > 4) continue;
#pragma unroll
                for (int p = 0; p < 4; ++p)
#pragma unroll
                    for (int kw = 0; kw < 5; ++kw)
                        acc[kh * 5 + kw] = __builtin_elementwise_fma(g[r][p], xr[p + kw], acc[kh * 5 + kw]);
            }
        }
    }
    // reduce over the 8 pixel lanes: the two of a wave by shuffle, the four waves through LDS
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        acc[t].x += __shfl_xor(acc[t].x, 32, 64);
        acc[t].y += __shfl_xor(acc[t].y, 32, 64);
    }
    ab.x += __shfl_xor(ab.x, 32, 64);
    ab.y += __shfl_xor(ab.y, 32, 64);
    __syncthreads();                                 // stages are dead: reuse as [4 waves][32 pairs][52] f32
    float* red = (float*)lds;
    if (lane < 32) {
        float* r = red + (wave * 32 + lane) * 52;
#pragma unroll
        for (int t = 0; t < 25; ++t) { r[t * 2] = acc[t].x; r[t * 2 + 1] = acc[t].y; }
        r[50] = ab.x; r[51] = ab.y;
    }
    __syncthreads();
    for (int q = tid; q < 32 * 52; q += 256) {
        const int pr = q / 52, e = q - pr * 52;
        const float v = red[(0 * 32 + pr) * 52 + e] + red[(1 * 32 + pr) * 52 + e] + red[(2 * 32 + pr) * 52 + e] +
                        red[(3 * 32 + pr) * 52 + e];
        const int cc = c_base + pr * 2 + (e & 1);
        if (cc >= C) continue;
        if (e < 50) atomicAdd(dw + (long)(e >> 1) * C + cc, v);
        else if (db) atomicAdd(db + cc, v);
    }
}

static int dwconv5_wgrad_impl(int dtype, const void* x, const void* dy, float* dw, float* db, int B,
                              int H, int W, int C, void* stream, const DwPre& pre) {
    NVAE_REQUIRE(B > 0 && H > 0 && W > 0 && C >= 8 && C % 8 == 0 && x && dy && dw, "dwconv5_wgrad: bad args");
    NVAE_REQUIRE(aligned16(x) && aligned16(dy), "dwconv5_wgrad: alignment");
    const int strips = cdiv(C, DW_CC);
    const bool small = H <= 4 && W <= 4;
    const int tx = small ? 1 : cdiv(W, 8), ty = small ? 1 : cdiv(H, 8);
    const long units = (long)B * tx * ty;
    // Each workgroup ends with one atomic per (tap, channel) of its strip, so its share of units must be
    // worth that tail: >= 4 units per workgroup when that still fills the 256 CUs, at most ~1024 workgroups.
    long want = units / (small ? 16 : 4);        // a 4x4 unit is only 16 pixels
    if (want > 1024 / strips) want = 1024 / strips;
    if (want < cdiv(256, strips)) want = cdiv(256, strips);
    if (want > units) want = units;
    if (want < 1 || g_nvae_det) want = 1;          // deterministic: one workgroup (one adder) per channel strip
    long upb = (units + want - 1) / want;
    const long chunks = (units + upb - 1) / upb;
    NVAE_REQUIRE(chunks <= 65535, "dwconv5_wgrad: too many tiles");
    dim3 grid(strips, (unsigned)chunks);
    if (is16(dtype) && !small) {
        if (dtype == NVAE_BF16)
            hipLaunchKernelGGL((k_dw5_wgrad_ring<bf16, 8, 8, 1>), grid, 256, 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, dw, db, B, H, W, C, tx, tx * ty, zero_page(), pre);
        else
            hipLaunchKernelGGL((k_dw5_wgrad_ring<f16, 8, 8, 1>), grid, 256, 0, (hipStream_t)stream, (const f16*)x, (const f16*)dy, dw, db, B, H, W, C, tx, tx * ty, zero_page(), pre);
        NVAE_LAUNCH_CHECK("dwconv5_wgrad");
        return NVAE_OK;
    }
    if (is16(dtype)) {            // 4x4 tower: units of four images
        const long u4 = (B + 3) / 4;
        long w4 = u4 / 4;
        if (w4 > 1024 / strips) w4 = 1024 / strips;
        if (w4 < cdiv(256, strips)) w4 = cdiv(256, strips);
        if (w4 > u4) w4 = u4;
        if (w4 < 1 || g_nvae_det) w4 = 1;
        if (dtype == NVAE_BF16)
            hipLaunchKernelGGL((k_dw5_wgrad_ring<bf16, 4, 4, 4>), dim3(strips, (unsigned)w4), 256, 0, (hipStream_t)stream, (const bf16*)x, (const bf16*)dy, dw, db, B, H, W, C, 1, 1, zero_page(), pre);
        else
            hipLaunchKernelGGL((k_dw5_wgrad_ring<f16, 4, 4, 4>), dim3(strips, (unsigned)w4), 256, 0, (hipStream_t)stream, (const f16*)x, (const f16*)dy, dw, db, B, H, W, C, 1, 1, zero_page(), pre);
        NVAE_LAUNCH_CHECK("dwconv5_wgrad");
        return NVAE_OK;
    }
    NVAE_REQUIRE(!pre.on, "dwconv5_wgrad_pre: the operand prologue exists for the 16-bit activation types only");
    if (small) {
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_dw5_wgrad<T, 4, 4, 1>), grid, 256, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, db, B, H, W, C, tx, tx * ty, (int)upb);)
    } else {
        DISPATCH_T(dtype, hipLaunchKernelGGL((k_dw5_wgrad<T, 8, 8, 4>), grid, 256, 0, (hipStream_t)stream, (const T*)x, (const T*)dy, dw, db, B, H, W, C, tx, tx * ty, (int)upb);)
    }
    NVAE_LAUNCH_CHECK("dwconv5_wgrad");
    return NVAE_OK;
}

extern "C" int nvae_dwconv5_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, int B,
                                  int H, int W, int C, void* stream) {
    return dwconv5_wgrad_impl(dtype, x, dy, dw, db, B, H, W, C, stream, DwPre{});
}

// Weight gradient against act(BN(x)) of the RAW tensor x, with the BatchNorm's FINAL coefficient table (the backward
// pass runs after the table was published): the counterpart of nvae_dwconv5_pre, 16-bit activation types only.
extern "C" int nvae_dwconv5_wgrad_pre(int dtype, const void* x, const float* scale, const float* shift, int act,
                                      const void* dy, float* dw, float* db, int B, int H, int W, int C, void* stream) {
    NVAE_REQUIRE(is16(dtype) && scale && shift, "dwconv5_wgrad_pre: needs a 16-bit activation type and a coefficient table");
    NVAE_REQUIRE(act == ACT_NONE || act == ACT_SWISH, "dwconv5_wgrad_pre: act %d unsupported", act);
    DwPre pre{};
    pre.on = 1; pre.act = act; pre.bn.scale = (float*)scale; pre.bn.shift = (float*)shift;
    return dwconv5_wgrad_impl(dtype, x, dy, dw, db, B, H, W, C, stream, pre);
}
