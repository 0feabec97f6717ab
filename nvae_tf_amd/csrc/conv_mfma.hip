// MFMA implicit-GEMM convolutions for gfx950: forward / data-gradient (k_conv_gemm) and
// weight-gradient (k_conv_wgrad).  bf16 inputs use v_mfma_f32_16x16x32_bf16, f32 inputs use the
// exact v_mfma_f32_16x16x4_f32; both accumulate in f32 and share one C/D fragment map
// (col = lane & 15, row = 4*(lane >> 4) + reg).
//
// k_conv_gemm:  C[M, N] = A[M, K] * B[K, N],  M = B*Hout*Wout, K = KH*KW*Cin, N = Cout.
//   A is gathered on the fly from the NHWC source (TF-'same' padding, stride, nearest-upsample or
//   gradient dilation are all folded into the gather, see NvaeConvGeom); B comes pre-transposed
//   ([N][K], k contiguous) so both LDS tiles are [rows][4 x 16-B chunks] and every MFMA operand is
//   one ds_read_b128.  The tile is 128 x BN x (4 chunks), 4 waves as 2 x 2, LDS double-buffered with
//   one barrier per K-step; the global loads of step t+1 are in flight behind the MFMAs of step t.
//   LDS chunk slots are XOR-swizzled (slot = chunk ^ f(row), f = {0,2,3,1}[(row>>2)&3]), which makes
//   the 16x16x32 operand read conflict-free over the hardware's 16-lane ds_read_b128 groups.
//   Workgroups are renumbered so that each XCD owns a contiguous range of tiles (the N-tiles of one
//   M-tile, and neighbouring M-tiles, share gathered activations and weight panels through that L2).
//
// k_conv_wgrad: dW[K, N] += A^T[K, M] * dY[M, N].  Both operands arrive pixel-major ([m][channels],
//   the natural NHWC order), are staged as such, and are transposed for free by ds_read_b64_tr_b16
//   (bf16) when the fragments are read.  The reduction over M is split across blockIdx.y and
//   combined with f32 atomics into the (zeroed) flat gradient buffer.
#include "common.h"

template <typename T> struct Tr;
template <> struct Tr<bf16> { static constexpr int VE = 8; };
template <> struct Tr<float> { static constexpr int VE = 4; };

__device__ __forceinline__ int swz4(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

// bijective XCD remap (cdna guide T1): consecutive logical tiles land on one XCD
__device__ __forceinline__ int xcd_remap(int id, int n) {
    int q = n >> 3, r = n & 7, x = id & 7, l = id >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

template <typename T>
__device__ __forceinline__ void mfma_step(const uint4& a, const uint4& b, f32x4& acc) {
    if constexpr (sizeof(T) == 2) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    } else {
        // the lane's 16-B chunk holds 4 consecutive k; A and B use the same k permutation
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
    }
}

// =========================================================================================
// k_conv_gemm2: the same implicit GEMM with the staging done by LDS-DMA (global_load_lds_dwordx4,
// no staging registers, no ds_write) into a 3-deep LDS ring, BK = 8 chunks (64 bf16 / 32 f32) per
// barrier.  The DMA is issued through inline asm so that hipcc does not know about the pending LDS
// writes (it would otherwise put s_waitcnt vmcnt(0) in front of every ds_read); completion is
// tracked by hand: one counted s_waitcnt vmcnt(NLOAD) + one raw s_barrier per K-step, with the loads
// of step t+2 issued right after the barrier of step t (ring slot (t+2)%3 was last read in step t-1,
// which every wave has finished once it passed that barrier).
//   LDS image per stage: [BM + BN rows][8 slots of 16 B]; slot = chunk ^ ((row >> 1) & 7): lane-linear
//   for the DMA (the swizzle is applied to the per-lane SOURCE address) and conflict-free for the
//   16x16x32 operand reads (checked against the ds_read_b128 16-lane groups).
//   Out-of-image (padding) and out-of-range lanes read a 16-B zero buffer instead of being masked:
//   LDS-DMA needs every lane to write its slot.
// =========================================================================================
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int BKC> __device__ __forceinline__ int swz_row(int row) {
    return BKC == 8 ? ((row >> 1) & 7) : (row & 15);
}

template <typename T, int BM, int BN, int WM, int WN, int STAGES, int BKC>
__global__ __launch_bounds__(WM* WN * 64) void k_conv_gemm2(
    NvaeConvGeom g, const T* __restrict__ src, const T* __restrict__ wT, int w_ld,
    const float* __restrict__ bias, const T* residual, void* out, int out_f32, int M, int K, int n_tiles,
    int total_tiles, FastDiv fd_hw, FastDiv fd_w, const uint4* __restrict__ zeros, float* stats,
    int vec_epi) {
    constexpr int NT = WM * WN * 64;
    constexpr int VE = Tr<T>::VE;
    constexpr int BKE = BKC * VE;                  // K elements per ring step (BKC 16-B chunks per row)
    constexpr int ACH = BM * BKC / NT, BCH = BN * BKC / NT;
    constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
    constexpr int STAGE = (BM + BN) * BKC;         // uint4 per ring slot
    constexpr int NLOAD = ACH + BCH;
    static_assert(ACH >= 1 && BCH >= 1 && (NT / BKC) % 16 == 0, "tile/thread mismatch");
    static_assert(BKC == 8 || BKC == 16, "row width");
    static_assert(STAGES * STAGE * 16 >= WM * WN * 16 * (BN / WN + 4) * 4, "epilogue staging must fit in the ring");
    static_assert(STAGES * STAGE * 16 >= WM * BN * 2 * 4, "stats scratch must fit in the ring");
    static_assert(STAGES == 2 || STAGES == 3, "ring depth");
    __shared__ uint4 lds[STAGES * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int tile = xcd_remap(blockIdx.x, total_tiles);
    const int bm = tile / n_tiles, bn = tile - bm * n_tiles;
    const int N = g.Cout;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    // ---- per-thread gather state: chunk q = tid + NT*i  ->  row q / BKC, physical slot q % BKC
    const int row0 = tid / BKC;
    const int kc = ((tid % BKC) ^ swz_row<BKC>(row0)) * VE;
    int tap = kc / g.Cin;
    int ci = kc - tap * g.Cin;
    int kh = tap / g.KW, kw = tap - kh * g.KW;
    int kabs = kc;
    const int hlim = g.Hin * g.div, wlim = g.Win * g.div;

    int hb[ACH], wb[ACH];
    long pb[ACH];
    bool mv[ACH];
#pragma unroll
    for (int i = 0; i < ACH; ++i) {
        int m = bm * BM + row0 + (NT / BKC) * i;
        mv[i] = m < M;
        unsigned mm = mv[i] ? (unsigned)m : 0u;
        unsigned b = fdiv(mm, fd_hw);
        unsigned rem = mm - b * fd_hw.d;
        unsigned ho = fdiv(rem, fd_w);
        unsigned wo = rem - ho * fd_w.d;
        hb[i] = (int)ho * g.stride - g.pad_t;
        wb[i] = (int)wo * g.stride - g.pad_l;
        pb[i] = (long)b * g.Hin * g.Win;
    }
    const T* bp[BCH];
    bool nv[BCH];
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        int n = bn * BN + row0 + (NT / BKC) * j;
        nv[j] = n < N;
        bp[j] = wT + (long)(nv[j] ? n : 0) * w_ld;
    }

    auto issue = [&](int slot) {
        const bool kval = kabs < K;
        const unsigned dst = lds_base + (unsigned)(slot * STAGE + wave * 64) * 16u;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            int hc = hb[i] + kh, wc = wb[i] + kw;
            bool ok = mv[i] && kval && hc >= 0 && hc < hlim && wc >= 0 && wc < wlim;
            int hs = hc, ws = wc;
            if (g.div != 1) {
                hs = hc / g.div; ws = wc / g.div;
                if (g.exact) ok = ok && (hs * g.div == hc) && (ws * g.div == wc);
            }
            const void* p = ok ? (const void*)(src + (pb[i] + (long)hs * g.Win + ws) * g.in_ld + ci) : (const void*)zeros;
            glds16(p, dst + (unsigned)(NT * i) * 16u);
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const void* p = (nv[j] && kval) ? (const void*)(bp[j] + kabs) : (const void*)zeros;
            glds16(p, dst + (unsigned)(BM * BKC + NT * j) * 16u);
        }
        kabs += BKE;
        ci += BKE;
        while (ci >= g.Cin) {
            ci -= g.Cin;
            if (++kw == g.KW) { kw = 0; ++kh; }
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (K + BKE - 1) / BKE;
    const int fr = lane & 15, fq = lane >> 4;
    issue(0);
    if (STAGES == 3 && nk > 1) issue(1);
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        if (STAGES == 3 && t + 1 < nk) wait_vmcnt<NLOAD>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // refill the slot that was read in step t-1 with the tile of step t + STAGES - 1
        if (t + STAGES - 1 < nk) issue(cur >= 1 ? cur - 1 : STAGES - 1);
        const uint4* buf = lds + cur * STAGE;
#pragma unroll
        for (int h = 0; h < BKC / 4; ++h) {
            uint4 af[MI], bf[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                int r = wm * (BM / WM) + i * 16 + fr;
                af[i] = buf[r * BKC + ((h * 4 + fq) ^ swz_row<BKC>(r))];
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                int r = wn * (BN / WN) + j * 16 + fr;
                bf[j] = buf[BM * BKC + r * BKC + ((h * 4 + fq) ^ swz_row<BKC>(r))];
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) mfma_step<T>(af[i], bf[j], acc[i][j]);
        }
        cur = cur == STAGES - 1 ? 0 : cur + 1;
    }

    // ---- epilogue -------------------------------------------------------------------------
    // (a) optional BatchNorm statistics of the output tile, straight from the accumulators:
    //     per column sum and sum of squares over the tile's valid rows -> stats[bm][2][N]
    if (stats) {
        __syncthreads();                                  // ring no longer read
        float* red = (float*)lds;                         // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = wn * (BN / WN) + j * 16 + fr;
            const int n = bn * BN + nl;
            const float bv = (bias && n < N) ? bias[n] : 0.f;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = bm * BM + wm * (BM / WM) + i * 16 + fq * 4 + r;
                    const float v = m < M ? acc[i][j][r] + bv : 0.f;
                    s1 += v; s2 += v * v;
                }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (fq == 0) { red[(wm * BN + nl) * 2] = s1; red[(wm * BN + nl) * 2 + 1] = s2; }
        }
        __syncthreads();
        for (int nl = tid; nl < BN; nl += NT) {
            const int n = bn * BN + nl;
            if (n >= N) continue;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { s1 += red[(w * BN + nl) * 2]; s2 += red[(w * BN + nl) * 2 + 1]; }
            stats[((long)bm * 2) * N + n] = s1;
            stats[((long)bm * 2 + 1) * N + n] = s2;
        }
    }
    // (b) output.  Vector path: each wave stages one 16-row slab of its tile in LDS (f32), then every
    //     lane stores 8 consecutive columns of one row (16 B bf16 / 32 B f32) - whole-line writes
    //     instead of the 2-byte column-strided stores the MFMA C layout gives directly.
    constexpr int WCOLS = BN / WN;
    constexpr int SROW = WCOLS + 4;                        // padded f32 row
    constexpr int VPR = WCOLS / 8;                         // 8-column vectors per row
    if (vec_epi) {
        __syncthreads();
        float* st = (float*)lds + wave * (16 * SROW);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(fq * 4 + r) * SROW + j * 16 + fr] = acc[i][j][r];
            __syncthreads();
            for (int v = lane; v < 16 * VPR; v += 64) {
                const int row = v / VPR, c8 = v - row * VPR;
                const int m = bm * BM + wm * (BM / WM) + i * 16 + row;
                const int n0 = bn * BN + wn * WCOLS + c8 * 8;
                if (m >= M || n0 >= N) continue;
                float o[8];
                const float4 lo = *(const float4*)(st + row * SROW + c8 * 8), hi = *(const float4*)(st + row * SROW + c8 * 8 + 4);
                o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
                if (n0 + 8 <= N) {
                    if (bias) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += bias[n0 + e];
                    }
                    if (residual) {
                        float rr[8];
                        V8<T>::ld(residual + (long)m * g.res_ld + n0, rr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rr[e];
                    }
                    if (out_f32) V8<float>::st((float*)out + (long)m * g.out_ld + n0, o);
                    else V8<T>::st((T*)out + (long)m * g.out_ld + n0, o);
                } else {
                    for (int e = 0; e < 8 && n0 + e < N; ++e) {
                        float vv = o[e] + (bias ? bias[n0 + e] : 0.f);
                        if (residual) vv += ldf<T>(residual + (long)m * g.res_ld + n0 + e);
                        if (out_f32) ((float*)out)[(long)m * g.out_ld + n0 + e] = vv;
                        else stf<T>((T*)out + (long)m * g.out_ld + n0 + e, vv);
                    }
                }
            }
            __syncthreads();
        }
        return;
    }
    // scalar path (unaligned channel slices, e.g. SkipScaler's 10-channel outputs)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = bn * BN + wn * (BN / WN) + j * 16 + fr;
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = bm * BM + wm * (BM / WM) + i * 16 + fq * 4 + r;
                if (m >= M) continue;
                float v = acc[i][j][r] + bv;
                if (residual) v += ldf<T>(residual + (long)m * g.res_ld + n);
                if (out_f32) ((float*)out)[(long)m * g.out_ld + n] = v;
                else stf<T>((T*)out + (long)m * g.out_ld + n, v);
            }
        }
    }
}

__device__ uint4 g_zero16[1];   // zero page for padded / out-of-range DMA lanes (zero-initialised)

static int check_geom_mfma(const char* who, const NvaeConvGeom* g) {
    NVAE_REQUIRE(g, "%s: NULL geometry", who);
    NVAE_REQUIRE(g->B > 0 && g->Hin > 0 && g->Win > 0 && g->Cin > 0 && g->Hout > 0 && g->Wout > 0 && g->Cout > 0,
                 "%s: non-positive dimension", who);
    NVAE_REQUIRE(g->KH > 0 && g->KW > 0 && g->KH <= 7 && g->KW <= 7 && g->stride >= 1 && g->div >= 1,
                 "%s: bad kernel/stride/div", who);
    NVAE_REQUIRE(g->in_ld >= g->Cin && g->out_ld >= g->Cout, "%s: leading dimensions too small", who);
    NVAE_REQUIRE((long)g->B * g->Hout * g->Wout < (1L << 23) && (long)g->B * g->Hin * g->Win < (1L << 23),
                 "%s: more than 2^23 pixels per call unsupported", who);
    return NVAE_OK;
}

static int conv_gemm_bm(const NvaeConvGeom* g) {
    // M-tile height the launcher will pick (must match launch_conv_gemm)
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    const long big_tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    const long w192 = (long)cdiv(N, 192) * 192, w128 = (long)cdiv(N, 128) * 128, w64 = (long)cdiv(N, 64) * 64;
    if ((long)cdiv(M, 256) * cdiv(N, 192) >= 224 && w192 <= w128 && w192 <= w64 && K >= 1024) return 256;
    if (big_tiles >= 192) return 128;
    return 64;
}

template <typename T>
static int launch_conv_gemm(const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                            const float* bias, const void* residual, void* out, int out_f32,
                            float* stats, hipStream_t s) {
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    FastDiv fd_hw = make_fastdiv((unsigned)(g->Hout * g->Wout)), fd_w = make_fastdiv((unsigned)g->Wout);
    uint4* zeros = nullptr;
    (void)hipGetSymbolAddress((void**)&zeros, HIP_SYMBOL(g_zero16));
    // 16-B row-contiguous stores need aligned rows; otherwise the scalar epilogue
    const int vo = out_f32 ? 4 : (int)(16 / sizeof(T));
    const int vec_epi = (g->out_ld % vo == 0) && aligned16(out) &&
                        (!residual || (g->res_ld % (int)(16 / sizeof(T)) == 0 && aligned16(residual)));
#define LAUNCH2(BM_, BN_, WM_, WN_, ST_, BKC_)                                                          \
    {                                                                                                   \
        int mt = cdiv(M, BM_), nt = cdiv(N, BN_);                                                       \
        hipLaunchKernelGGL((k_conv_gemm2<T, BM_, BN_, WM_, WN_, ST_, BKC_>), mt * nt, WM_ * WN_ * 64, 0, s, *g, \
                           (const T*)src, (const T*)wT, w_ld, bias, (const T*)residual, out, out_f32, M, \
                           K, nt, mt * nt, fd_hw, fd_w, zeros, stats, vec_epi);                         \
    }
    // Large problems: 128-row tiles, 8 waves; N tile with the least padding (ties -> larger).
    // Small problems (few tiles): 64 x 64 tiles, 4 waves, so that the grid covers the chip.
    const long big_tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    const long w192 = (long)cdiv(N, 192) * 192, w128 = (long)cdiv(N, 128) * 128, w64 = (long)cdiv(N, 64) * 64;
    if ((long)cdiv(M, 256) * cdiv(N, 192) >= 224 && w192 <= w128 && w192 <= w64 && K >= 1024) {
        // the FLOP-dominant layers: 256 x 192 tile (112 FLOP per staged byte), 2-deep ring (112 KB)
        LAUNCH2(256, 192, 4, 2, 2, 8)
    } else if (big_tiles >= 192) {
        if (w192 <= w128 && w192 <= w64) LAUNCH2(128, 192, 2, 4, 3, 8)
        else if (w128 <= w64) LAUNCH2(128, 128, 2, 4, 3, 8)
        else LAUNCH2(128, 64, 4, 2, 3, 8)
    } else if (K >= 512) {
        // small M: latency-bound K loop -> 8 waves, 128-deep ring steps (half the barriers)
        LAUNCH2(64, 64, 2, 4, 3, 16)
    } else {
        LAUNCH2(64, 64, 2, 2, 3, 8)
    }
#undef LAUNCH2
    return 0;
}

extern "C" int nvae_conv_gemm_mtiles(const NvaeConvGeom* g) {
    if (!g) return 0;
    return cdiv((long)g->B * g->Hout * g->Wout, conv_gemm_bm(g));
}

extern "C" int nvae_conv_gemm(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                              const float* bias, const void* residual, void* out, int out_f32,
                              float* stats, void* stream) {
    if (int e = check_geom_mfma("conv_gemm", g)) return e;
    NVAE_REQUIRE(src && wT && out, "conv_gemm: NULL pointer");
    const int ve = (dtype == NVAE_BF16) ? 8 : 4;
    NVAE_REQUIRE(g->Cin % ve == 0 && g->in_ld % ve == 0 && w_ld % ve == 0,
                 "conv_gemm: Cin=%d in_ld=%d w_ld=%d must be multiples of %d (use nvae_conv_direct)", g->Cin, g->in_ld, w_ld, ve);
    NVAE_REQUIRE(w_ld >= g->KH * g->KW * g->Cin, "conv_gemm: w_ld too small");
    NVAE_REQUIRE(aligned16(src) && aligned16(wT), "conv_gemm: src/wT must be 16-B aligned");
    NVAE_REQUIRE(!residual || g->res_ld >= g->Cout, "conv_gemm: res_ld too small");
    DISPATCH_T(dtype, launch_conv_gemm<T>(g, src, wT, w_ld, bias, residual, out, out_f32, stats, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("conv_gemm");
    return NVAE_OK;
}

// =========================================================================================
// k_conv_wgrad2: weight gradient with the LDS-DMA ring.  Per ring step RS = 8 chunks-worth of pixels
// (64 bf16 / 32 f32) of the pixel-major x-gather [RS][KT] and dy [RS][NTL] images are DMA'd into LDS;
// the images keep the XOR-swizzled 32-B segment layout of k_conv_wgrad (transposing ds_read_b64_tr_b16
// fragment reads), with the swizzle moved to the DMA's per-lane source address.
// =========================================================================================
template <typename T, int COLS>
__device__ __forceinline__ int img_src_chunk(int m, int pc) {
    // inverse of img_off: which logical 16-B chunk lands at physical chunk `pc` of pixel row `m`
    if constexpr (sizeof(T) == 2) {
        int s = (COLS >= 128) ? ((m & 3) | (((m >> 3) & 1) << 2)) : (((m >> 1) & 1) | (((m >> 3) & 1) << 1));
        return (((pc >> 1) ^ s) << 1) | (pc & 1);
    } else {
        return (((pc >> 2) ^ (m & 1)) << 2) | (pc & 3);
    }
}

template <typename T, int KT, int NTL, int WK, int WN, int STAGES>
__global__ __launch_bounds__(WK* WN * 64) void k_conv_wgrad2(
    NvaeConvGeom g, const T* __restrict__ x, const T* __restrict__ dy, float* dw, int dw_ld, float* db,
    int M, int K, int n_tiles, int m_per_split, FastDiv fd_hw, FastDiv fd_w,
    const uint4* __restrict__ zeros, float* slab) {
    constexpr int NT = WK * WN * 64;
    constexpr int VE = Tr<T>::VE;
    constexpr int RS = 8 * VE;                      // pixels per ring step
    constexpr int CPR_A = KT / VE, CPR_B = NTL / VE;
    constexpr int ACH = RS * CPR_A / NT, BCH = RS * CPR_B / NT;
    constexpr int MI = KT / WK / 16, NI = NTL / WN / 16;
    constexpr int A_BYTES = RS * KT * (int)sizeof(T), B_BYTES = RS * NTL * (int)sizeof(T);
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int NLOAD = ACH + BCH;
    static_assert(ACH >= 1 && BCH >= 1, "tile/thread mismatch");
    static_assert((NT / CPR_A) % 16 == 0 || sizeof(T) == 4, "A rows per pass must keep the swizzle invariant");
    static_assert((NT / CPR_B) % 16 == 0 || sizeof(T) == 4, "B rows per pass must keep the swizzle invariant");
    static_assert((NT / CPR_A) % 2 == 0 && (NT / CPR_B) % 2 == 0, "row step parity");
    __shared__ __attribute__((aligned(16))) unsigned char lds[STAGES * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / WN, wn = wave - wk * WN;
    // consecutive tiles (the n-tiles of one k-tile, then the next k-tile) share an XCD's L2
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int kt = tile / n_tiles, nt = tile - kt * n_tiles;
    const int k0 = kt * KT, n0 = nt * NTL;
    const int N = g.Cout;
    const int m_begin = blockIdx.y * m_per_split;
    int m_end = m_begin + m_per_split;
    if (m_end > M) m_end = M;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    // A side: physical chunk (a_row0 + pass*rows, a_pc); logical k column fixed per thread
    const int a_pc = tid % CPR_A, a_row0 = tid / CPR_A;
    const int kcol = k0 + img_src_chunk<T, KT>(a_row0, a_pc) * VE;
    const bool kval = kcol < K;
    const int tap = kval ? kcol / g.Cin : 0;
    const int ci = kval ? kcol - tap * g.Cin : 0;
    const int kh = tap / g.KW, kw = tap - kh * g.KW;
    const int hlim = g.Hin * g.div, wlim = g.Win * g.div;
    const int b_pc = tid % CPR_B, b_row0 = tid / CPR_B;
    const int ncol = n0 + img_src_chunk<T, NTL>(b_row0, b_pc) * VE;
    const bool nval = ncol < N;

    auto issue = [&](int slot, int mbase) {
        const unsigned dst = lds_base + (unsigned)(slot * STAGE_BYTES) + (unsigned)(wave * 64) * 16u;
#pragma unroll
        for (int i = 0; i < ACH; ++i) {
            const int m = mbase + a_row0 + i * (NT / CPR_A);
            const void* p = zeros;
            if (kval && m < m_end) {
                unsigned b = fdiv((unsigned)m, fd_hw);
                unsigned rem = (unsigned)m - b * fd_hw.d;
                unsigned ho = fdiv(rem, fd_w);
                unsigned wo = rem - ho * fd_w.d;
                int hc = (int)ho * g.stride - g.pad_t + kh, wc = (int)wo * g.stride - g.pad_l + kw;
                bool ok = hc >= 0 && hc < hlim && wc >= 0 && wc < wlim;
                int hs = hc, ws = wc;
                if (g.div != 1) {
                    hs = hc / g.div; ws = wc / g.div;
                    if (g.exact) ok = ok && (hs * g.div == hc) && (ws * g.div == wc);
                }
                if (ok) p = x + ((long)b * g.Hin * g.Win + (long)hs * g.Win + ws) * g.in_ld + ci;
            }
            glds16(p, dst + (unsigned)(NT * i) * 16u);
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int m = mbase + b_row0 + j * (NT / CPR_B);
            const void* p = (nval && m < m_end) ? (const void*)(dy + (long)m * g.out_ld + ncol) : (const void*)zeros;
            glds16(p, dst + (unsigned)A_BYTES + (unsigned)(NT * j) * 16u);
        }
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (db != nullptr) && (kt == 0);
    float bsum[VE];
#pragma unroll
    for (int j = 0; j < VE; ++j) bsum[j] = 0.f;

    const int fr = lane & 15, fq = lane >> 4;
    const int nsteps = (m_end - m_begin + RS - 1) / RS;
    if (nsteps > 0) issue(0, m_begin);
    if (STAGES == 3 && nsteps > 1) issue(1, m_begin + RS);
    int cur = 0;
    for (int t = 0; t < nsteps; ++t) {
        if (STAGES == 3 && t + 1 < nsteps) wait_vmcnt<NLOAD>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (t + STAGES - 1 < nsteps) issue(cur >= 1 ? cur - 1 : STAGES - 1, m_begin + (t + STAGES - 1) * RS);
        const unsigned char* bufA = lds + cur * STAGE_BYTES;
        const unsigned char* bufB = bufA + A_BYTES;
        if (do_bias) {
            // re-read this thread's own DMA'd dy chunks (fixed column, RS/(NT/CPR_B) rows)
#pragma unroll
            for (int j = 0; j < BCH; ++j) {
                uint4 v = *(const uint4*)(bufB + (tid + NT * j) * 16);
                if constexpr (sizeof(T) == 2) {
                    bsum[0] += __uint_as_float(v.x << 16); bsum[1] += __uint_as_float(v.x & 0xffff0000u);
                    bsum[2] += __uint_as_float(v.y << 16); bsum[3] += __uint_as_float(v.y & 0xffff0000u);
                    bsum[4] += __uint_as_float(v.z << 16); bsum[5] += __uint_as_float(v.z & 0xffff0000u);
                    bsum[6] += __uint_as_float(v.w << 16); bsum[7] += __uint_as_float(v.w & 0xffff0000u);
                } else {
                    bsum[0] += __uint_as_float(v.x); bsum[1] += __uint_as_float(v.y);
                    bsum[2] += __uint_as_float(v.z); bsum[3] += __uint_as_float(v.w);
                }
            }
        }
        if constexpr (sizeof(T) == 2) {
            const int q = fr >> 2, p = fr & 3;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int mr = ks * 32 + 8 * fq + q, mr1 = mr + 4;
                const int s8_0 = (mr & 3) | (((mr >> 3) & 1) << 2), s8_1 = (mr1 & 3) | (((mr1 >> 3) & 1) << 2);
                const int s4_0 = ((mr >> 1) & 1) | (((mr >> 3) & 1) << 1), s4_1 = ((mr1 >> 1) & 1) | (((mr1 >> 3) & 1) << 1);
                const int sa0 = KT >= 128 ? s8_0 : s4_0, sa1 = KT >= 128 ? s8_1 : s4_1;
                const int sb0 = NTL >= 128 ? s8_0 : s4_0, sb1 = NTL >= 128 ? s8_1 : s4_1;
                bf16x8 af[MI], bfr[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int seg = wk * (KT / WK / 16) + i;
                    auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(bufA + mr * (KT * 2) + ((seg ^ sa0) << 5) + p * 8));
                    auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(bufA + mr1 * (KT * 2) + ((seg ^ sa1) << 5) + p * 8));
                    af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int seg = wn * (NTL / WN / 16) + j;
                    auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(bufB + mr * (NTL * 2) + ((seg ^ sb0) << 5) + p * 8));
                    auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(bufB + mr1 * (NTL * 2) + ((seg ^ sb1) << 5) + p * 8));
                    bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int sI = 0; sI < RS / 4; ++sI) {
                const int m = 4 * sI + fq;
                float af[MI], bfr[NI];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int col = wk * (KT / WK) + i * 16 + fr;
                    af[i] = *(const float*)(bufA + m * (KT * 4) + (((col >> 4) ^ (m & 1)) << 6) + ((col & 15) << 2));
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int col = wn * (NTL / WN) + j * 16 + fr;
                    bfr[j] = *(const float*)(bufB + m * (NTL * 4) + (((col >> 4) ^ (m & 1)) << 6) + ((col & 15) << 2));
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        cur = cur == STAGES - 1 ? 0 : cur + 1;
    }

    // With a slab, every split writes its partial [K + 1][N] block (row K = bias) with plain stores and
    // k_slab_reduce sums the splits; without, f32 atomics into the gradient buffer.
    float* part = slab ? slab + (long)blockIdx.y * (K + 1) * N : nullptr;
    if (do_bias) {
        __syncthreads();
        float* red = (float*)lds;                    // [NT / CPR_B rows][CPR_B][VE]
#pragma unroll
        for (int j = 0; j < VE; ++j) red[tid * VE + j] = bsum[j];
        __syncthreads();
        if (tid < CPR_B * VE) {
            // logical column (chunk lc, element j): physical chunk differs per row through the swizzle
            const int lc = tid / VE, j = tid - lc * VE;
            float a = 0.f;
            for (int r = 0; r < NT / CPR_B; ++r) {
                int pc = img_src_chunk<T, NTL>(r, lc);      // the swizzle is an involution
                a += red[(r * CPR_B + pc) * VE + j];
            }
            const int n = n0 + lc * VE + j;
            if (n < N) {
                if (part) part[(long)K * N + n] = a;
                else atomicAdd(db + n, a);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int n = n0 + wn * (NTL / WN) + j * 16 + fr;
        if (n >= N) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + wk * (KT / WK) + i * 16 + fq * 4 + r;
                if (k < K) {
                    if (part) part[(long)k * N + n] = acc[i][j][r];
                    else atomicAdd(dw + (long)k * dw_ld + n, acc[i][j][r]);
                }
            }
    }
}

// dw[k*dw_ld + n] += sum_s slab[s][k][n]  (k < K);  db[n] += sum_s slab[s][K][n].  32 outputs per
// workgroup, 8 lane groups walk the splits in parallel.
__global__ void k_slab_reduce(const float* __restrict__ slab, int S, int K, int N, float* dw, int dw_ld,
                              float* db) {
    __shared__ float sm[8][32];
    const int ol = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const long total = (long)(K + 1) * N;
    const long i = (long)blockIdx.x * 32 + ol;
    float a = 0.f;
    if (i < total)
        for (int s = sl; s < S; s += 8) a += slab[(long)s * total + i];
    sm[sl][ol] = a;
    __syncthreads();
    if (sl != 0 || i >= total) return;
#pragma unroll
    for (int k = 1; k < 8; ++k) a += sm[k][ol];
    const int k = (int)(i / N), n = (int)(i - (long)k * N);
    if (k < K) dw[(long)k * dw_ld + n] += a;
    else if (db) db[n] += a;
}

// Split policy shared by the launcher and the scratch-size query.
struct WgradPlan { int cfg, tiles, n_tiles, nsplit, mps; bool slab; };
template <typename T>
static WgradPlan plan_conv_wgrad(const NvaeConvGeom* g, long scratch_floats) {
    constexpr int RS = 8 * Tr<T>::VE;
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    const long flops = 2L * M * K * N;
    WgradPlan p;
    int KT, NTL;
    if (flops >= (1L << 36) && N >= 128 && K >= 256) { p.cfg = 0; KT = 256; NTL = 128; }
    else if (flops >= (1L << 32) && N >= 128 && K >= 128) { p.cfg = 1; KT = 128; NTL = 128; }
    else { p.cfg = 2; KT = 64; NTL = 64; }
    p.n_tiles = cdiv(N, NTL);
    p.tiles = cdiv(K, KT) * p.n_tiles;
    // enough workgroups to fill the chip (one wave of 256 for the 8-wave config, ~2 per CU otherwise),
    // at least 4 ring steps each
    int nsplit = (p.cfg == 0 ? 256 : 512) / p.tiles;
    int max_split = M / (RS * 4);
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit > 256) nsplit = 256;
    if (nsplit < 1) nsplit = 1;
    auto settle = [&](int want) {
        int mps = cdiv(M, want);
        mps = ((mps + RS - 1) / RS) * RS;
        p.mps = mps;
        p.nsplit = cdiv(M, mps);
    };
    settle(nsplit);
    // > 4 splits are combined through a slab (same-address f32 atomics serialise, ~0.3 us each);
    // without enough scratch fall back to 4 atomically combined splits
    p.slab = p.nsplit > 4;
    if (p.slab && scratch_floats < (long)p.nsplit * (K + 1) * N) {
        p.slab = false;
        settle(4);
    }
    return p;
}

template <typename T>
static int launch_conv_wgrad(const NvaeConvGeom* g, const void* x, const void* dy, float* dw, int dw_ld,
                             float* db, float* scratch, long scratch_floats, hipStream_t s) {
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    FastDiv fd_hw = make_fastdiv((unsigned)(g->Hout * g->Wout)), fd_w = make_fastdiv((unsigned)g->Wout);
    uint4* zeros = nullptr;
    (void)hipGetSymbolAddress((void**)&zeros, HIP_SYMBOL(g_zero16));
    const WgradPlan p = plan_conv_wgrad<T>(g, scratch ? scratch_floats : 0);
    float* slab = p.slab ? scratch : nullptr;
    dim3 grid(p.tiles, p.nsplit);
#define LAUNCHW(KT_, NTL_, WK_, WN_, ST_)                                                               \
    hipLaunchKernelGGL((k_conv_wgrad2<T, KT_, NTL_, WK_, WN_, ST_>), grid, WK_ * WN_ * 64, 0, s, *g,    \
                       (const T*)x, (const T*)dy, dw, dw_ld, db, M, K, p.n_tiles, p.mps, fd_hw, fd_w,   \
                       zeros, slab);
    if (p.cfg == 0) LAUNCHW(256, 128, 4, 2, 3)
    else if (p.cfg == 1) LAUNCHW(128, 128, 2, 2, 3)
    else LAUNCHW(64, 64, 2, 2, 3)
#undef LAUNCHW
    if (slab)
        hipLaunchKernelGGL(k_slab_reduce, cdiv((long)(K + 1) * N, 32), 256, 0, s, slab, p.nsplit, K, N, dw, dw_ld, db);
    return 0;
}

extern "C" long nvae_conv_wgrad_scratch(int dtype, const NvaeConvGeom* g) {
    if (!g) return 0;
    const long K = (long)g->KH * g->KW * g->Cin, N = g->Cout;
    WgradPlan p = dtype == NVAE_BF16 ? plan_conv_wgrad<bf16>(g, 1L << 60) : plan_conv_wgrad<float>(g, 1L << 60);
    return p.slab ? (long)p.nsplit * (K + 1) * N : 0;
}

extern "C" int nvae_conv_wgrad(int dtype, const NvaeConvGeom* g, const void* x, const void* dy, float* dw,
                               int dw_ld, float* db, float* scratch, long scratch_floats, void* stream) {
    if (int e = check_geom_mfma("conv_wgrad", g)) return e;
    NVAE_REQUIRE(x && dy && dw && dw_ld >= g->Cout, "conv_wgrad: bad args");
    const int ve = (dtype == NVAE_BF16) ? 8 : 4;
    NVAE_REQUIRE(g->Cin % ve == 0 && g->in_ld % ve == 0 && g->Cout % ve == 0 && g->out_ld % ve == 0,
                 "conv_wgrad: Cin=%d Cout=%d and their lds must be multiples of %d (use nvae_conv_direct_wgrad)", g->Cin, g->Cout, ve);
    NVAE_REQUIRE(aligned16(x) && aligned16(dy), "conv_wgrad: x/dy must be 16-B aligned");
    DISPATCH_T(dtype, launch_conv_wgrad<T>(g, x, dy, dw, dw_ld, db, scratch, scratch_floats, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("conv_wgrad");
    return NVAE_OK;
}
