// Weight-gradient implicit GEMM (see conv_gemm.hip for the forward / data-gradient kernels).
#include "conv_common.h"

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

// Layers of one geometry are launched together (nvae_conv_wgrad_batched): blockIdx.z selects the layer's
// pointers from this by-value table (kernel arguments, no device-side table to upload) and its slab.
#define WGRAD_BATCH_MAX 32
struct WgradBatch {
    const void* x[WGRAD_BATCH_MAX];
    const void* dy[WGRAD_BATCH_MAX];
    float* dw[WGRAD_BATCH_MAX];
    float* db[WGRAD_BATCH_MAX];
    long slab_stride;               // floats between the layers' slabs
};

template <typename T, int KT, int NTL, int WK, int WN, int STAGES>
__global__ __launch_bounds__(WK* WN * 64) void k_conv_wgrad2(
    NvaeConvGeom g, WgradBatch bt, int dw_ld,
    int M, int K, int n_tiles, int m_per_split, FastDiv fd_hw, FastDiv fd_w,
    const uint4* __restrict__ zeros, float* slab) {
    const T* __restrict__ x = (const T*)bt.x[blockIdx.z];
    const T* __restrict__ dy = (const T*)bt.dy[blockIdx.z];
    float* dw = bt.dw[blockIdx.z];
    float* db = bt.db[blockIdx.z];
    if (slab) slab += (long)blockIdx.z * bt.slab_stride;
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
                    float u8[8];
                    unpack8<T>(v, u8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) bsum[e] += u8[e];
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
                        acc[i][j] = mfma16<T>(af[i], bfr[j], acc[i][j]);
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
__global__ void k_slab_reduce(const float* __restrict__ slab, int S, int K, int N, WgradBatch bt, int dw_ld) {
    float* dw = bt.dw[blockIdx.y];
    float* db = bt.db[blockIdx.y];
    slab += (long)blockIdx.y * bt.slab_stride;
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

// =========================================================================================
// k_wgrad_halo (bf16): weight gradient of the dense K x K stride-1 'same' layers (the FLOP-dominant 5x5
// convs).  k_conv_wgrad2 re-streams x and dy once per k-tile, i.e. once per tap: 85 FLOP per staged
// byte, L2/Infinity-Cache bound at ~450 TFLOP/s.  Here a workgroup owns one kernel row kh, 64 input
// channels and 192 output channels and keeps the K taps' [64 x 192] accumulators in registers
// (K * 6 MFMA blocks per wave = 120 VGPRs for K = 5).  Per step it DMAs, for one 8 x 16 pixel
// half-patch, the x halo rows of its kh ([8][16+K-1] pixels x 64 ch = 20 KB) and the dy tile
// ([128 px][192] = 48 KB) ONCE and feeds all K taps from them: 231 FLOP per staged byte.
//   Both images are pixel-major and read with the transposing ds_read_b64_tr_b16.  x rows (128 B) use
//   the 4-segment swizzle keyed on the HALO row, which is conflict-free at any tap shift; dy rows
//   (384 B = 12 segments) swizzle segments 0-7 with the 8-segment key and 8-11 with the 4-segment key.
// =========================================================================================
__device__ __forceinline__ int seg_s8(int m) { return (m & 3) | (((m >> 3) & 1) << 2); }
__device__ __forceinline__ int seg_s4(int m) { return ((m >> 1) & 1) | (((m >> 3) & 1) << 1); }

template <typename T, int KS>
__global__ __launch_bounds__(512) void k_wgrad_halo(NvaeConvGeom g, const T* __restrict__ x,
                                                    const T* __restrict__ dy, float* dw, int dw_ld,
                                                    int n_tiles, int cchunks, int hp_per_split,
                                                    int hp_w /*W/16*/, int hp_per_img /*(H/8)*(W/16)*/,
                                                    int hp_total, const uint4* __restrict__ zeros) {
    constexpr int NT = 512, NTL = 192, CCH = 64;
    constexpr int HW_ = 16 + KS - 1;                     // halo width in pixels
    constexpr int HWL = 32;                              // halo row pitch in LDS (pixels): a power of two keeps
    //                                                      the swizzle key (bits of the pixel index) independent
    //                                                      of the halo ROW, so every operand address below is a
    //                                                      per-lane constant + a compile-time offset
    constexpr int A_ROWS = 8 * HWL;                      // LDS rows per step (one kernel row; columns >= HW_ unused)
    constexpr int A_CHUNKS = (A_ROWS * 8 + 63) / 64 * 64;
    constexpr int A_PASSES = (A_CHUNKS + NT - 1) / NT;
    constexpr int B_CHUNKS = 128 * 24, BCH = B_CHUNKS / NT;   // 6
    constexpr int A_BYTES = A_CHUNKS * 16, B_BYTES = B_CHUNKS * 16;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int PAD = (KS - 1) / 2;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 2, wn = wave & 3;             // 2 (ci halves of 32) x 4 (n quarters of 48)
    // blockIdx.x = ((cc * n_tiles) + nt) * KS + kh ; blockIdx.y = pixel split
    const int kh = blockIdx.x % KS;
    const int t2 = blockIdx.x / KS;
    const int nt = t2 % n_tiles, cc = t2 / n_tiles;
    const int n0 = nt * NTL, ci0 = cc * CCH;
    const int H = g.Hin, W = g.Win;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    // ---- per-thread DMA constants
    // x halo: chunk q = tid + NT*i -> halo pixel q >> 3, physical chunk q & 7 (16 B = 8 channels)
    int a_hy[A_PASSES], a_hx[A_PASSES], a_col[A_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int q = tid + NT * i, hrow = q >> 3, pc = q & 7;
        a_hy[i] = hrow / HWL; a_hx[i] = hrow - a_hy[i] * HWL;
        if (hrow >= A_ROWS || a_hx[i] >= HW_) a_hy[i] = -1000;      // unused columns / tail lanes DMA zeros
        a_col[i] = ((((pc >> 1) ^ seg_s4(hrow)) << 1) | (pc & 1)) * 8;
    }
    // dy tile: chunk q -> pixel q / 24, physical chunk q % 24
    int b_p[BCH], b_col[BCH];
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        const int q = tid + NT * j, p = q / 24, pc = q - p * 24;
        const int sp = pc >> 1;
        const int seg = sp < 8 ? (sp ^ seg_s8(p)) : 8 + ((sp - 8) ^ seg_s4(p));
        b_p[j] = p;
        b_col[j] = n0 + (seg * 2 + (pc & 1)) * 8;
    }
    auto issue = [&](int slot, int hp) {                  // hp = global half-patch index
        const int b = hp / hp_per_img, r = hp - b * hp_per_img;
        const int py0 = (r / hp_w) * 8, px0 = (r % hp_w) * 16;
        const unsigned dst = lds_base + (unsigned)(slot * STAGE_BYTES) + (unsigned)(wave * 64) * 16u;
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            if (NT * i + wave * 64 >= A_CHUNKS) continue;
            const int iy = py0 + a_hy[i] + kh - PAD, ix = px0 + a_hx[i] - PAD;
            const void* p = zeros;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                p = x + (((long)b * H + iy) * W + ix) * g.in_ld + ci0 + a_col[i];
            glds16(p, dst + (unsigned)(NT * i) * 16u);
        }
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const int py = py0 + (b_p[j] >> 4), px = px0 + (b_p[j] & 15);
            const void* p = dy + (((long)b * H + py) * W + px) * g.out_ld + b_col[j];
            glds16(p, dst + (unsigned)A_BYTES + (unsigned)(NT * j) * 16u);
        }
    };

    f32x4 acc[KS][2][3];
#pragma unroll
    for (int t = 0; t < KS; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int hp_begin = blockIdx.y * hp_per_split;
    int hp_end = hp_begin + hp_per_split;
    if (hp_end > hp_total) hp_end = hp_total;
    const int nsteps = hp_end - hp_begin;
    const int fr = lane & 15, fq = lane >> 4;
    const int q4 = fr >> 2, p4 = fr & 3;
    // Per-lane operand offsets.  This lane's two 4-pixel blocks of a 32-pixel k-step are pixels 8*fq + q4 and
    // + 4 (patch row 0 or 1 of the step, column 0..15); the k-step index only adds a compile-time constant
    // (2 halo rows / 32 dy pixels), and the swizzle keys depend on the column bits alone.
    int offA[KS][2][2], offB[3][2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int pl = 8 * fq + q4 + 4 * blk, prow = pl >> 4, pcol = pl & 15;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int seg = wn * 3 + j;
            const int gsw = seg < 8 ? (seg ^ seg_s8(pl)) : 8 + ((seg - 8) ^ seg_s4(pl));
            offB[j][blk] = pl * 384 + (gsw << 5) + p4 * 8;
        }
#pragma unroll
        for (int kw = 0; kw < KS; ++kw) {
            const int h = prow * HWL + pcol + kw;
#pragma unroll
            for (int i = 0; i < 2; ++i) offA[kw][i][blk] = h * 128 + (((wk * 2 + i) ^ seg_s4(h)) << 5) + p4 * 8;
        }
    }
    if (nsteps > 0) issue(0, hp_begin);
    for (int s = 0; s < nsteps; ++s) {
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 1 < nsteps) issue((s + 1) & 1, hp_begin + s + 1);
        const unsigned char* bufA = lds + (s & 1) * STAGE_BYTES;
        const unsigned char* bufB = bufA + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                  // 32 pixels per MFMA k-step: patch rows 2ks, 2ks+1
            bf16x8 bfr[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(bufB + offB[j][0] + ks * (32 * 384)));
                auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                    (__attribute__((address_space(3))) bf16x4*)(bufB + offB[j][1] + ks * (32 * 384)));
                bfr[j] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                bf16x8 af[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    auto lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(bufA + offA[kw][i][0] + ks * (2 * HWL * 128)));
                    auto hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                        (__attribute__((address_space(3))) bf16x4*)(bufA + offA[kw][i][1] + ks * (2 * HWL * 128)));
                    af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        acc[kw][i][j] = mfma16<T>(af[i], bfr[j], acc[kw][i][j]);
            }
        }
    }
    // ---- epilogue: f32 atomics (few adders per address: the pixel splits)
#pragma unroll
    for (int kw = 0; kw < KS; ++kw)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + wn * 48 + j * 16 + fr;
            if (n >= g.Cout) continue;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long k = (long)(kh * KS + kw) * g.Cin + ci0 + wk * 32 + i * 16 + fq * 4 + r;
                    atomicAdd(dw + k * dw_ld + n, acc[kw][i][j][r]);
                }
        }
}

static bool wgrad_halo_ok(int dtype, const NvaeConvGeom* g, const float* db) {
    return !g_nvae_det && is16(dtype) && db == nullptr && g->KH == g->KW && (g->KH == 5 || g->KH == 3) && g->stride == 1 &&
           g->div == 1 && g->pad_t == (g->KH - 1) / 2 && g->pad_l == (g->KW - 1) / 2 && g->Hin == g->Hout &&
           g->Win == g->Wout && g->Hin % 8 == 0 && g->Win % 16 == 0 && g->Cin % 64 == 0 && g->Cout % 192 == 0 &&
           (long)g->B * g->Hin * g->Win >= 16384;
}

template <typename T>
static void launch_wgrad_halo(const NvaeConvGeom* g, const void* x, const void* dy, float* dw, int dw_ld,
                              hipStream_t s) {
    const int n_tiles = g->Cout / 192, cchunks = g->Cin / 64, KS = g->KH;
    const int hp_w = g->Win / 16, hp_per_img = (g->Hin / 8) * hp_w, hp_total = g->B * hp_per_img;
    const int tiles = cchunks * n_tiles * KS;
    int nsplit = 256 / tiles;                            // one wave of 512-thread workgroups
    if (nsplit < 1) nsplit = 1;
    if (nsplit > hp_total / 8) nsplit = hp_total / 8 > 0 ? hp_total / 8 : 1;
    int hps = cdiv(hp_total, nsplit);
    nsplit = cdiv(hp_total, hps);
    dim3 grid(tiles, nsplit);
    const uint4* zeros = zero_page();
    if (KS == 5)
        hipLaunchKernelGGL((k_wgrad_halo<T, 5>), grid, 512, 0, s, *g, (const T*)x, (const T*)dy, dw, dw_ld, n_tiles,
                           cchunks, hps, hp_w, hp_per_img, hp_total, zeros);
    else
        hipLaunchKernelGGL((k_wgrad_halo<T, 3>), grid, 512, 0, s, *g, (const T*)x, (const T*)dy, dw, dw_ld, n_tiles,
                           cchunks, hps, hp_w, hp_per_img, hp_total, zeros);
}

// Split policy shared by the launcher and the scratch-size query.
struct WgradPlan { int cfg, tiles, n_tiles, nsplit, mps; bool slab; };
template <typename T>
static WgradPlan plan_conv_wgrad(const NvaeConvGeom* g, long scratch_floats, int n_layers = 1) {
    constexpr int RS = 8 * Tr<T>::VE;
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    const long flops = 2L * M * K * N;
    WgradPlan p;
    int KT, NTL;
    // the largest tile that still fills the chip: alone a layer needs the FLOPs for it, in a batch of n_layers
    // same-shape layers (the towers) the tile count is multiplied by n_layers
    const long nl = n_layers < 1 ? 1 : n_layers;
    const long tiles0 = (long)cdiv(K, 256) * cdiv(N, 128) * nl, tiles1 = (long)cdiv(K, 128) * cdiv(N, 128) * nl;
    if (N >= 128 && K >= 256 && (flops >= (1L << 33) || (nl > 1 && tiles0 >= 192))) { p.cfg = 0; KT = 256; NTL = 128; }
    else if (N >= 128 && K >= 128 && (flops >= (1L << 32) || (nl > 1 && tiles1 >= 256))) { p.cfg = 1; KT = 128; NTL = 128; }
    else { p.cfg = 2; KT = 64; NTL = 64; }
    p.n_tiles = cdiv(N, NTL);
    p.tiles = cdiv(K, KT) * p.n_tiles;
    // enough workgroups to fill the chip (one wave of 256 for the 8-wave config, ~2 per CU otherwise),
    // at least 4 ring steps each.  A batched launch already has n_layers times the tiles: the towers' 20-30
    // same-shape layers fill the chip without any pixel split, i.e. without partial slabs (round 1 split every
    // layer as if it were alone: 5 splits x 30 layers of the 256 -> 1536 convs wrote and re-read 237 MB of slabs)
    int nsplit = (p.cfg == 0 ? 256 : 512) / (p.tiles * (n_layers < 1 ? 1 : n_layers));
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
    p.slab = p.nsplit > (g_nvae_det ? 1 : 4);
    if (p.slab && scratch_floats < (long)p.nsplit * (K + 1) * N) {
        p.slab = false;
        settle(g_nvae_det ? 1 : 4);          // (deterministic: atomically combined splits are not an option)
    }
    return p;
}

template <typename T>
static int launch_conv_wgrad(const NvaeConvGeom* g, int n, const void* const* x, const void* const* dy,
                             float* const* dw, int dw_ld, float* const* db, float* scratch, long scratch_floats,
                             hipStream_t s) {
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    FastDiv fd_hw = make_fastdiv((unsigned)(g->Hout * g->Wout)), fd_w = make_fastdiv((unsigned)g->Wout);
    const uint4* zeros = zero_page();
    // scratch_floats is the budget PER LAYER: all layers of a batch share one plan
    const WgradPlan p = plan_conv_wgrad<T>(g, scratch ? scratch_floats : 0, n);
    float* slab = p.slab ? scratch : nullptr;
    WgradBatch bt{};
    for (int i = 0; i < n; ++i) { bt.x[i] = x[i]; bt.dy[i] = dy[i]; bt.dw[i] = dw[i]; bt.db[i] = db ? db[i] : nullptr; }
    bt.slab_stride = p.slab ? (long)p.nsplit * (K + 1) * N : 0;
    dim3 grid(p.tiles, p.nsplit, n);
#define LAUNCHW(KT_, NTL_, WK_, WN_, ST_)                                                               \
    hipLaunchKernelGGL((k_conv_wgrad2<T, KT_, NTL_, WK_, WN_, ST_>), grid, WK_ * WN_ * 64, 0, s, *g, bt,   \
                       dw_ld, M, K, p.n_tiles, p.mps, fd_hw, fd_w, zeros, slab);
    if (p.cfg == 0) LAUNCHW(256, 128, 4, 2, 3)
    else if (p.cfg == 1) LAUNCHW(128, 128, 2, 2, 3)
    else LAUNCHW(64, 64, 2, 2, 3)
#undef LAUNCHW
    if (slab)
        hipLaunchKernelGGL(k_slab_reduce, dim3(cdiv((long)(K + 1) * N, 32), n), 256, 0, s, slab, p.nsplit, K, N, bt, dw_ld);
    return 0;
}

extern "C" long nvae_conv_wgrad_scratch_n(int dtype, const NvaeConvGeom* g, int n_layers) {
    if (!g) return 0;
    const long K = (long)g->KH * g->KW * g->Cin, N = g->Cout;
    WgradPlan p = is16(dtype) ? plan_conv_wgrad<bf16>(g, 1L << 60, n_layers) : plan_conv_wgrad<float>(g, 1L << 60, n_layers);
    return p.slab ? (long)p.nsplit * (K + 1) * N : 0;
}
extern "C" long nvae_conv_wgrad_scratch(int dtype, const NvaeConvGeom* g) { return nvae_conv_wgrad_scratch_n(dtype, g, 1); }

static int check_wgrad_args(const char* who, int dtype, const NvaeConvGeom* g, int dw_ld) {
    if (int e = check_geom_mfma(who, g)) return e;
    const int ve = is16(dtype) ? 8 : 4;
    NVAE_REQUIRE(dw_ld >= g->Cout, "%s: dw_ld too small", who);
    NVAE_REQUIRE(g->Cin % ve == 0 && g->in_ld % ve == 0 && g->Cout % ve == 0 && g->out_ld % ve == 0,
                 "%s: Cin=%d Cout=%d and their lds must be multiples of %d (use nvae_conv_direct_wgrad)", who, g->Cin, g->Cout, ve);
    return NVAE_OK;
}

extern "C" int nvae_conv_wgrad(int dtype, const NvaeConvGeom* g, const void* x, const void* dy, float* dw,
                               int dw_ld, float* db, float* scratch, long scratch_floats, void* stream) {
    if (int e = check_wgrad_args("conv_wgrad", dtype, g, dw_ld)) return e;
    NVAE_REQUIRE(x && dy && dw, "conv_wgrad: bad args");
    NVAE_REQUIRE(aligned16(x) && aligned16(dy), "conv_wgrad: x/dy must be 16-B aligned");
    if (wgrad_halo_ok(dtype, g, db)) {
        if (dtype == NVAE_BF16) launch_wgrad_halo<bf16>(g, x, dy, dw, dw_ld, (hipStream_t)stream);
        else launch_wgrad_halo<f16>(g, x, dy, dw, dw_ld, (hipStream_t)stream);
        NVAE_LAUNCH_CHECK("wgrad_halo");
        return NVAE_OK;
    }
    DISPATCH_T(dtype, launch_conv_wgrad<T>(g, 1, &x, &dy, &dw, dw_ld, db ? &db : nullptr, scratch, scratch_floats, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("conv_wgrad");
    return NVAE_OK;
}

// n <= 32 layers of the SAME geometry in one launch (the residual towers repeat one conv shape 10-40 times per
// step and each of those weight gradients is a 10-15 us kernel): x / dy / dw / db are host arrays of n
// device pointers (db: all NULL or none), scratch holds n * nvae_conv_wgrad_scratch(dtype, g) floats.
extern "C" int nvae_conv_wgrad_batched(int dtype, const NvaeConvGeom* g, int n, const void* const* x,
                                       const void* const* dy, float* const* dw, int dw_ld, float* const* db,
                                       float* scratch, long scratch_floats_per_layer, void* stream) {
    if (int e = check_wgrad_args("conv_wgrad_batched", dtype, g, dw_ld)) return e;
    NVAE_REQUIRE(n >= 1 && n <= WGRAD_BATCH_MAX && x && dy && dw, "conv_wgrad_batched: n=%d must be in [1, %d]", n, WGRAD_BATCH_MAX);
    for (int i = 0; i < n; ++i) {
        NVAE_REQUIRE(x[i] && dy[i] && dw[i] && aligned16(x[i]) && aligned16(dy[i]), "conv_wgrad_batched: bad pointer in layer %d", i);
        NVAE_REQUIRE(!db || db[i], "conv_wgrad_batched: db must be given for all layers or none");
    }
    if (wgrad_halo_ok(dtype, g, db ? db[0] : nullptr)) {
        for (int i = 0; i < n; ++i) {
            if (dtype == NVAE_BF16) launch_wgrad_halo<bf16>(g, x[i], dy[i], dw[i], dw_ld, (hipStream_t)stream);
            else launch_wgrad_halo<f16>(g, x[i], dy[i], dw[i], dw_ld, (hipStream_t)stream);
        }
        NVAE_LAUNCH_CHECK("wgrad_halo");
        return NVAE_OK;
    }
    DISPATCH_T(dtype, launch_conv_wgrad<T>(g, n, x, dy, dw, dw_ld, db, scratch, scratch_floats_per_layer, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("conv_wgrad_batched");
    return NVAE_OK;
}
