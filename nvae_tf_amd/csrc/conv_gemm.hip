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
#include "conv_common.h"
#include "bn_fin.h"

// Shared epilogue of the forward / data-gradient kernels.  `row_m(r)` maps a tile-local output row to
// its global pixel index (or -1 if it does not exist); `lds_f` is the (no longer read) staging ring.
// Fused BatchNorm-backward reduction (nvae_conv_gemm_bnbwd): this launch computes the data gradient dy of
// a conv whose INPUT was y = act(BN(x)).  The epilogue, which holds the finished dy tile anyway, also
// forms dpre = dy * act'(scale*x + shift) against the matching tile of x and emits the per-tile column
// sums (sum dpre, sum dpre*x) as a [m_tiles][2][C] slab; the last M-tile of each N-tile column to arrive
// turns the slab into dgamma / dbeta / k0,k1 (bn_fin.h).  The separate read of x and dy by
// nvae_bn_bwd_reduce disappears; nvae_bn_bwd_apply follows as before.
struct ConvBnBwd {
    const void* x;          // BN input, [M, x_ld] in the activation dtype; nullptr = no fusion
    int x_ld, act, m_tiles, rows;   // M-tiles of the launch; rows of the (zeroed) slab they accumulate into
    const float* scale; const float* shift;
    float* partials;
    BnFinArgs fin;
};

// A-operand prologue (nvae_conv_gemm_ex): the source tensor is the INPUT of a BatchNorm(+Swish) whose
// coefficient table is final; the gathered chunks are turned into act(scale*x + shift) inside LDS, after
// their DMA has landed and before the workgroup's barrier, by the thread that issued the DMA (it knows
// which chunks are padding and must stay zero).  The normalised activation never makes a round trip of its
// own through HBM (round 1: one nvae_bn_apply_fin launch of 6-40 us in front of every conv).  act_out
// (optional): the activated tensor is written once, by the one N-tile workgroup whose turn it is, for the
// weight-gradient kernel of the same conv (which runs much later, on the side stream).
#define PRE_MAXC 2048            // channels of the coefficient table in LDS (k_conv_gemm2)
#define PRE_MAXC_2CU 512         // (k_conv_gemm2's two-workgroups-per-CU family)
#define PRE_MAXC_HALO 512        // (k_conv_halo: 152 KB of its 160 KB are the ring)
struct ConvPre {
    BnFromSlab bn;              // final coefficient table, or the statistics slab to finish in-kernel (bn_fin.h)
    bool on;
    int act;
    void* act_out; int act_ld;
};

// Split-K of k_conv_gemm2 (small-M layers whose K loop is long: the 3x3 convs and the 6x-wide 1x1 convs of the
// 4x4 / 8x8 towers): S workgroups per output tile, each `steps` ring steps of the K loop.  slab / counter: the
// workspace registered with nvae_conv_set_workspace (launches that share it must be stream-ordered).
struct ConvSplitK {
    int S, steps;
    float* slab;                // [tiles][S][BM*BN] f32 partial tiles, lane-linear
    int* counter;               // [tiles], zero at rest
};

template <typename T> __device__ __forceinline__ uint4 pre_chunk(uint4 raw, const float* sc, const float* sh, int act) {
    if constexpr (sizeof(T) == 2) {
        float v[8];
        unpack8<T>(raw, v);
        const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4), h0 = *(const float4*)sh, h1 = *(const float4*)(sh + 4);
        const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float p = v[j] * scv[j] + shv[j];
            v[j] = act == ACT_SWISH ? swishf_(p) : p;
        }
        return pack8<T>(v);
    } else {
        float v[4] = {__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w)};
        const float4 s0 = *(const float4*)sc, h0 = *(const float4*)sh;
        const float scv[4] = {s0.x, s0.y, s0.z, s0.w}, shv[4] = {h0.x, h0.y, h0.z, h0.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float p = v[j] * scv[j] + shv[j];
            v[j] = act == ACT_SWISH ? swishf_(p) : p;
        }
        return make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3]));
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool BNBWD, typename RowMap>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[BM / WM / 16][BN / WN / 16], float* lds_f,
                                              const NvaeConvGeom& g, const float* __restrict__ bias,
                                              const T* residual, void* out, int out_f32, int bm, int bn,
                                              float* stats, int vec_epi, RowMap row_m, const ConvBnBwd& be,
                                              int* stats_counter, int& ticket) {
    constexpr int NT = WM * WN * 64;
    constexpr int MI = BM / WM / 16, NI = BN / WN / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int fr = lane & 15, fq = lane >> 4;
    const int N = g.Cout;
    // ---- epilogue -------------------------------------------------------------------------
    // (a) optional BatchNorm statistics of the output tile, straight from the accumulators:
    //     per column sum and sum of squares over the tile's valid rows -> stats[bm][2][N]
    typedef typename StatT<T>::type ST;                  // slab element type (bn_fin.h "statistics precision")
    if (stats) {
        __syncthreads();                                  // ring no longer read
        ST* red = (ST*)lds_f;                       // [WM][BN][2]
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int nl = wn * (BN / WN) + j * 16 + fr;
            const int n = bn * BN + nl;
            const float bv = (bias && n < N) ? bias[n] : 0.f;
            ST s1 = 0, s2 = 0;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long m = row_m(wm * (BM / WM) + i * 16 + fq * 4 + r);
                    // (statistics of the OUTPUT: a residual - e.g. the accumulating second GEMM of a concat-free conv - counts)
                    float o = acc[i][j][r] + bv;
                    if (residual && m >= 0 && n < N) o += ldf<T>(residual + m * g.res_ld + n);
                    const ST v = m >= 0 ? (ST)o : (ST)0;
                    s1 += v; s2 += v * v;
                }
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (fq == 0) { red[(wm * BN + nl) * 2] = s1; red[(wm * BN + nl) * 2 + 1] = s2; }
        }
        __syncthreads();
        ST* slab = (ST*)stats;
        for (int nl = tid; nl < BN; nl += NT) {
            const int n = bn * BN + nl;
            if (n >= N) continue;
            ST s1 = 0, s2 = 0;
#pragma unroll
            for (int w = 0; w < WM; ++w) { s1 += red[(w * BN + nl) * 2]; s2 += red[(w * BN + nl) * 2 + 1]; }
            // M-tile bm adds into row bm % rows of the zeroed slab: at most 64 adders per address (measured on
            // MI355X, tools/mb_atomic.hip: <= 64 same-address float adders cost < 0.6 us and nobody waits for them;
            // 256 cost 7 us, 1024 26 us), and a consumer sums `rows` (1-8) rows instead of up to 512
            const int row = bm % be.rows;
            atomicAdd(slab + ((long)row * 2) * N + n, s1);
            atomicAdd(slab + ((long)row * 2 + 1) * N + n, s2);
        }
        // forward statistics with an in-kernel finalize: take the arrival ticket now (only the slab adds are
        // waited for; the ticket's round trip overlaps the output stores below), look at it when the kernel ends
        if (stats_counter) ticket = bn_arrive(stats_counter + bn);
    }
    // (b) output.  Vector path: each wave stages one 16-row slab of its tile in LDS (f32), then every
    //     lane stores 8 consecutive columns of one row (16 B bf16 / 32 B f32) - whole-line writes
    //     instead of the 2-byte column-strided stores the MFMA C layout gives directly.
    constexpr int WCOLS = BN / WN;
    constexpr int SROW = WCOLS + 4;                        // padded f32 row
    constexpr int VPR = WCOLS / 8;                         // 8-column vectors per row
    if constexpr (BNBWD) {
        // (b') vector stores with a FIXED column group per lane (so it can keep running column sums):
        //      lane -> (c8 = lane % VPR, row lane rl = lane / VPR), RPP rows of the 16-row slab per pass.
        constexpr int RPP = (64 / VPR) < 16 ? (64 / VPR) : 16;
        const int c8 = lane % VPR, rl = lane / VPR;
        const bool act_lane = rl < RPP;
        const int n0 = bn * BN + wn * WCOLS + c8 * 8;
        const bool nval = n0 < N;                          // N % 8 == 0 (checked by the host)
        float sc[8], sh[8];
        ST s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sc[e] = nval ? be.scale[n0 + e] : 0.f;
            sh[e] = nval ? be.shift[n0 + e] : 0.f;
            s1[e] = 0; s2[e] = 0;
        }
        const T* bx = (const T*)be.x;
        // 16-bit tensors: every x vector this lane will need (MI slabs x RPL rows) is requested NOW, before the slabs
        // are staged through LDS - issued inside the loop, each slab exposed one global-load latency
        constexpr int RPL = (16 + RPP - 1) / RPP;
        uint4 xq[sizeof(T) == 2 ? MI : 1][sizeof(T) == 2 ? RPL : 1];
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    const int row = rl + k * RPP;
                    xq[i][k] = make_uint4(0u, 0u, 0u, 0u);
                    if (act_lane && nval && row < 16) {
                        const long m = row_m(wm * (BM / WM) + i * 16 + row);
                        if (m >= 0) xq[i][k] = *(const uint4*)(bx + m * be.x_ld + n0);
                    }
                }
        }
        __syncthreads();
        float* st = lds_f + wave * (16 * SROW);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(fq * 4 + r) * SROW + j * 16 + fr] = acc[i][j][r];
            __syncthreads();
            if (act_lane && nval)
#pragma unroll
                for (int k = 0; k < RPL; ++k) {
                    const int row = rl + k * RPP;
                    if (row >= 16) continue;
                    const long m = row_m(wm * (BM / WM) + i * 16 + row);
                    if (m < 0) continue;
                    float o[8], xv[8];
                    const float4 lo = *(const float4*)(st + row * SROW + c8 * 8), hi = *(const float4*)(st + row * SROW + c8 * 8 + 4);
                    o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
                    if (bias) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += bias[n0 + e];
                    }
                    if (residual) {
                        float rr[8];
                        V8<T>::ld(residual + m * g.res_ld + n0, rr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rr[e];
                    }
                    V8<T>::st((T*)out + m * g.out_ld + n0, o);
                    if constexpr (sizeof(T) == 2) unpack8<T>(xq[i][k], xv);
                    else V8<T>::ld(bx + m * be.x_ld + n0, xv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        float dpre = o[e];
                        if (be.act == ACT_SWISH) dpre *= dswishf_(xv[e] * sc[e] + sh[e]);
                        s1[e] += (ST)dpre;
                        s2[e] += (ST)dpre * (ST)xv[e];
                    }
                }
            __syncthreads();
        }
        // column sums of the tile: [WM * RPP row lanes][BN][2] through LDS, then one slab row per M-tile
        ST* red = (ST*)lds_f;
        if (act_lane) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int idx = ((wm * RPP + rl) * BN + wn * WCOLS + c8 * 8 + e) * 2;
                red[idx] = s1[e]; red[idx + 1] = s2[e];
            }
        }
        __syncthreads();
        for (int nl = tid; nl < BN; nl += NT) {
            const int n = bn * BN + nl;
            if (n >= N) continue;
            ST a1 = 0, a2 = 0;
            for (int w = 0; w < WM * RPP; ++w) { a1 += red[(w * BN + nl) * 2]; a2 += red[(w * BN + nl) * 2 + 1]; }
            const int row = bm % be.rows;
            atomicAdd((ST*)be.partials + ((long)row * 2) * N + n, a1);
            atomicAdd((ST*)be.partials + ((long)row * 2 + 1) * N + n, a2);
        }
        // With a counter the last M-tile of this column of tiles finalizes in place.  That makes every
        // workgroup wait for its own output stores and an atomic round trip before it frees its LDS
        // (measured: +4..8 us on kernels of 15-25 us), so the host normally passes no counter and runs
        // nvae_bn_bwd_finalize_s on the slab instead.
        if (be.fin.counter && bn_last_arriver(be.fin.counter + bn, be.m_tiles))
            bn_fin_bwd<sizeof(T) == 4>(be.fin, be.partials, be.rows, N, bn * BN, (BN + 63) / 64);
        return;
    }
    if (vec_epi) {
        // 16-bit residual (skip connections, accumulating data gradients): all of this lane's residual vectors are
        // requested before the slabs are staged, like the x vectors of the BN-backward epilogue above
        constexpr int VPL = (16 * VPR + 63) / 64;
        uint4 rq[sizeof(T) == 2 ? MI : 1][sizeof(T) == 2 ? VPL : 1];
        const bool rpre = sizeof(T) == 2 && residual != nullptr;
        if constexpr (sizeof(T) == 2) {
            if (rpre) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int k = 0; k < VPL; ++k) {
                        const int v = lane + 64 * k;
                        rq[i][k] = make_uint4(0u, 0u, 0u, 0u);
                        if (v < 16 * VPR) {
                            const int row = v / VPR, c8 = v - row * VPR;
                            const long m = row_m(wm * (BM / WM) + i * 16 + row);
                            const int n0 = bn * BN + wn * WCOLS + c8 * 8;
                            if (m >= 0 && n0 + 8 <= N) rq[i][k] = *(const uint4*)(residual + m * g.res_ld + n0);
                        }
                    }
            }
        }
        __syncthreads();
        float* st = lds_f + wave * (16 * SROW);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[(fq * 4 + r) * SROW + j * 16 + fr] = acc[i][j][r];
            __syncthreads();
#pragma unroll
            for (int k = 0; k < VPL; ++k) {
                const int v = lane + 64 * k;
                if (v >= 16 * VPR) continue;
                const int row = v / VPR, c8 = v - row * VPR;
                const long m = row_m(wm * (BM / WM) + i * 16 + row);
                const int n0 = bn * BN + wn * WCOLS + c8 * 8;
                if (m < 0 || n0 >= N) continue;
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
                        if constexpr (sizeof(T) == 2) unpack8<T>(rq[i][k], rr);
                        else V8<T>::ld(residual + m * g.res_ld + n0, rr);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rr[e];
                    }
                    if (out_f32) V8<float>::st((float*)out + m * g.out_ld + n0, o);
                    else V8<T>::st((T*)out + m * g.out_ld + n0, o);
                } else {
                    for (int e = 0; e < 8 && n0 + e < N; ++e) {
                        float vv = o[e] + (bias ? bias[n0 + e] : 0.f);
                        if (residual) vv += ldf<T>(residual + m * g.res_ld + n0 + e);
                        if (out_f32) ((float*)out)[m * g.out_ld + n0 + e] = vv;
                        else stf<T>((T*)out + m * g.out_ld + n0 + e, vv);
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
                const long m = row_m(wm * (BM / WM) + i * 16 + fq * 4 + r);
                if (m < 0) continue;
                float v = acc[i][j][r] + bv;
                if (residual) v += ldf<T>(residual + m * g.res_ld + n);
                if (out_f32) ((float*)out)[m * g.out_ld + n] = v;
                else stf<T>((T*)out + m * g.out_ld + n, v);
            }
        }
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
template <typename T, int BM, int BN, int WM, int WN, int STAGES, int BKC, bool BNBWD, bool PRE>
__global__ __launch_bounds__(WM* WN * 64, (STAGES == 2 && BM == 128) ? 2 : 1) void k_conv_gemm2(
    NvaeConvGeom g, const T* __restrict__ src, const T* __restrict__ wT, int w_ld,
    const float* __restrict__ bias, const T* residual, void* out, int out_f32, int M, int K, int n_tiles,
    int total_tiles, FastDiv fd_hw, FastDiv fd_w, const uint4* __restrict__ zeros, float* stats,
    int vec_epi, ConvBnBwd be, ConvPre pre, BnFinArgs sfin, ConvSplitK sk) {
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
    static_assert(STAGES * STAGE * 16 >= WM * BN * 2 * 8, "stats scratch must fit in the ring");
    static_assert(!BNBWD || STAGES * STAGE * 16 >= WM * ((64 / (BN / WN / 8)) < 16 ? (64 / (BN / WN / 8)) : 16) * BN * 2 * (int)sizeof(typename StatT<T>::type),
                  "BN-backward scratch must fit in the ring");
    static_assert(STAGES >= 2 && STAGES <= 6, "ring depth");
    __shared__ uint4 lds[STAGES * STAGE];
    // (the 2-deep 128-row family is sized for TWO workgroups per CU - short-K layers whose time is prologue + epilogue: one
    //  workgroup's stores overlap the other's loads - so its coefficient table is small: K <= 512 means Cin <= 512)
    constexpr int PTC = (STAGES == 2 && BM == 128) ? PRE_MAXC_2CU : PRE_MAXC;
    __shared__ __attribute__((aligned(16))) float pre_tab[PRE ? 2 * PTC : 4];   // [scale | shift] of the prologue

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    // split-K (sk.S > 1): the S workgroups of a tile are neighbours in the remapped numbering (one XCD), each takes
    // `sk.steps` ring steps of the K loop; see the hand-off after the loop
    const int wg = xcd_remap(blockIdx.x, total_tiles * sk.S);
    const int tile = wg / sk.S, ks = wg - tile * sk.S;
    const int bm = tile / n_tiles, bn = tile - bm * n_tiles;
    const int N = g.Cout;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const int t_begin = ks * sk.steps;             // first ring step of this workgroup

    // ---- per-thread gather state: chunk q = tid + NT*i  ->  row q / BKC, physical slot q % BKC
    const int row0 = tid / BKC;
    const int kc = t_begin * BKE + ((tid % BKC) ^ swz_row<BKC>(row0)) * VE;
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

    // prologue state: the (tap, channel) position of ring step t, which lags `issue` by STAGES-1 steps
    int p_kh = kh, p_kw = kw, p_ci = ci, p_kabs = kabs;
    const int nk_all = (K + BKE - 1) / BKE;
    const int nk = nk_all - t_begin < sk.steps ? nk_all - t_begin : sk.steps;
    const int fr = lane & 15, fq = lane >> 4;
    issue(0);
    if constexpr (PRE) {
        // (behind the first stage's DMA: the slab -> coefficient chain is a dependent global round trip + f64 arithmetic)
        for (int c = tid; c < g.Cin; c += NT) bn_coef<sizeof(T) == 4>(pre.bn, g.Cin, c, blockIdx.x == 0, pre_tab[c], pre_tab[PTC + c]);
        __syncthreads();
    }
#pragma unroll
    for (int q = 1; q < STAGES - 1; ++q)
        if (nk > q) issue(q);
    int cur = 0;
    for (int t = 0; t < nk; ++t) {
        // stage t must have landed; up to STAGES-2 younger stages may stay in flight
        const int younger = nk - 1 - t;
        if (STAGES >= 6 && younger >= 4) wait_vmcnt<4 * NLOAD>();
        else if (STAGES >= 5 && younger >= 3) wait_vmcnt<3 * NLOAD>();
        else if (STAGES >= 4 && younger >= 2) wait_vmcnt<2 * NLOAD>();
        else if (STAGES >= 3 && younger >= 1) wait_vmcnt<NLOAD>();
        else wait_vmcnt<0>();
        if constexpr (PRE) {
            // this thread's own A chunks of stage t are in LDS (its DMA, its vmcnt): normalise + activate them in
            // place; padding / out-of-range chunks stay zero.  The barrier below publishes them with the rest.
            const bool kval = p_kabs < K;
            const bool my_turn = pre.act_out && ((t_begin + t) % n_tiles) == bn && p_kh == g.pad_t && p_kw == g.pad_l;
#pragma unroll
            for (int i = 0; i < ACH; ++i) {
                const int hc = hb[i] + p_kh, wc = wb[i] + p_kw;
                if (mv[i] && kval && hc >= 0 && hc < hlim && wc >= 0 && wc < wlim) {
                    uint4* slot = lds + cur * STAGE + tid + NT * i;
                    const uint4 v = pre_chunk<T>(*slot, pre_tab + p_ci, pre_tab + PTC + p_ci, pre.act);
                    *slot = v;
                    if (my_turn) *(uint4*)((T*)pre.act_out + (pb[i] + (long)hc * g.Win + wc) * pre.act_ld + p_ci) = v;
                }
            }
            p_kabs += BKE;
            p_ci += BKE;
            while (p_ci >= g.Cin) {
                p_ci -= g.Cin;
                if (++p_kw == g.KW) { p_kw = 0; ++p_kh; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // refill the slot that was read in step t-1 with the tile of step t + STAGES - 1
        if (t + STAGES - 1 < nk) issue(cur >= 1 ? cur - 1 : STAGES - 1);
        const uint4* buf = lds + cur * STAGE;
        if constexpr (sizeof(T) == 4) {
            // f32 (the parity path): TWO-LEVEL accumulation.  One accumulator fed K products in K order carries a
            // rounding error ~ eps * K / sqrt(2) (in units of the products' spread); K is 2 304-9 600 here, and measured
            // against fp64 that was 3-5x the error of a blocked CPU GEMM (tools/diag_f32_ops.py).  Each ring step's
            // 32 / 64 products are summed in a fresh accumulator and added to the running one: error ~ eps *
            // sqrt(c*K/2 + K^2/(2c)), 5-7x smaller at these K.
            uint4 af[BKC / 4][MI], bf[BKC / 4][NI];
#pragma unroll
            for (int h = 0; h < BKC / 4; ++h) {
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    int r = wm * (BM / WM) + i * 16 + fr;
                    af[h][i] = buf[r * BKC + ((h * 4 + fq) ^ swz_row<BKC>(r))];
                }
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    int r = wn * (BN / WN) + j * 16 + fr;
                    bf[h][j] = buf[BM * BKC + r * BKC + ((h * 4 + fq) ^ swz_row<BKC>(r))];
                }
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int h = 0; h < BKC / 4; ++h) mfma_step<T>(af[h][i], bf[h][j], part);
                    acc[i][j] += part;
                }
        } else {
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
        }
        cur = cur == STAGES - 1 ? 0 : cur + 1;
    }

    // (big tiles are never split: their slab registers would spill in the common path)
    constexpr int SPLIT_QC = MI * NI <= 2 ? 4 : 2;            // slabs requested per round trip by the reducer
    if constexpr (MI * NI <= 4) if (sk.S > 1) {
        // Split-K hand-off (cdna guide, "Projection GEMM at M = 256" item 2, the sc1 form): every workgroup writes its
        // f32 partial tile lane-linearly (accumulator (i, j) of thread tid at [(i*NI + j)*NT + tid], 16 B per lane:
        // whole lines, no layout work) with write-through stores, waits for them, and takes a ticket on the tile's
        // counter; the one that draws S - 1 reads the other slabs with sc1 loads and runs the epilogue.  The sum is
        // formed in slice order whoever arrives last (own slice from registers), so the result does not depend on
        // the arrival order.  Counters are zero at rest: the last arriver resets its tile's.
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(sk.slab + (long)tile * sk.S * (BM * BN)), 0, sk.S * BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bn_u4, acc[i][j]), rs,
                                                       ((ks * MI * NI + i * NI + j) * NT + tid) * 16, 0, 16 /* sc1 */);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = (int*)lds;                       // (the ring is no longer read)
        if (tid == 0) {
            const int old = atomicAdd(sk.counter + tile, 1);
            if (old == sk.S - 1) atomicExch(sk.counter + tile, 0);
            *flag = old;
        }
        __syncthreads();
        const bool last = *flag == sk.S - 1;
        __syncthreads();                             // everybody has read the flag before the epilogue reuses the ring
        if (!last) return;
        // ordered sum over the slices, four slabs requested per round trip; no branch around a load (a slice index
        // past the end, or the own slice, is loaded anyway and not used): hipcc would wait for each load separately
        f32x4 own[MI][NI];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) own[i][j] = acc[i][j];
        for (int q0 = 0; q0 < sk.S; q0 += SPLIT_QC) {
            bn_u4 v[SPLIT_QC][MI][NI];
#pragma unroll
            for (int u = 0; u < SPLIT_QC; ++u) {
                const int q = q0 + u < sk.S ? q0 + u : sk.S - 1;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        v[u][i][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, ((q * MI * NI + i * NI + j) * NT + tid) * 16, 0, 16 /* sc1 */);
            }
#pragma unroll
            for (int u = 0; u < SPLIT_QC; ++u) {
                const int q = q0 + u;
                if (q < sk.S) {
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const f32x4 val = q == ks ? own[i][j] : __builtin_bit_cast(f32x4, v[u][i][j]);
                            if (q == 0) acc[i][j] = val;
                            else acc[i][j] += val;
                        }
                }
            }
        }
    }
    int ticket = 0;
    conv_epilogue<T, BM, BN, WM, WN, BNBWD>(acc, (float*)lds, g, bias, residual, out, out_f32, bm, bn, stats, vec_epi,
                                            [&](int r) -> long { int m = bm * BM + r; return m < M ? (long)m : -1L; }, be,
                                            sfin.counter, ticket);
    if constexpr (!BNBWD) {
        // the last M-tile of this column of tiles turns the statistics slab into the next BatchNorm's coefficients
        if (stats && sfin.counter && bn_was_last(sfin.counter + bn, ticket, be.m_tiles))
            bn_fin_fwd<sizeof(T) == 4>(sfin, stats, be.rows, N, bn * BN, (BN + 63) / 64);
    }
}

// =========================================================================================
// k_conv_halo: K x K (3 or 5), stride 1, 'same' convolutions on images whose sides are multiples of
// 16 -- the FLOP-dominant dense 5x5 layers of Postprocess and their data gradients.
//   k_conv_gemm2 is bound by the bytes it stages: per 64-deep K-step a 256 x 192 tile DMAs 32 KB of
//   gathered activations + 24 KB of weights, and every activation is fetched again for each of the 25
//   taps.  Here the M-tile is one 16 x 16 pixel patch: its (16+K-1)^2 halo x 64 channels (51 KB for
//   K = 5) is DMA'd into LDS ONCE per channel chunk and all K*K taps read their shifted 16 x 16 window
//   out of it; only the weight tile streams per step (24 KB): 2.1x fewer staged bytes per FLOP and no
//   per-step gather addressing.  The next chunk's halo is prefetched one DMA pass per step during the
//   first 7 taps.  LDS: 2 halo buffers + 2 weight slots = 148 KB.
//   Halo rows are [pixel][8 chunks], slot = chunk ^ (row & 7): conflict-free for the 16x16x32 operand
//   read at ANY row offset (the tap shift moves the 16-row window by kh*20 + kw rows).
// =========================================================================================
static __device__ unsigned long long g_halo_stamps[2048];     // diagnostic stamps, see nvae_conv_halo_stamps

template <typename T, int BN, int KS, int WM, bool BNBWD, bool PRE>
__global__ __launch_bounds__(WM * 128) void k_conv_halo(
    NvaeConvGeom g, const T* __restrict__ src, const T* __restrict__ wT, int w_ld,
    const float* __restrict__ bias, const T* residual, void* out, int out_f32, int n_tiles, int total_tiles,
    int patches_w, int patches_per_img, const uint4* __restrict__ zeros, float* stats, int vec_epi,
    ConvBnBwd be, ConvPre pre, BnFinArgs sfin) {
    constexpr int BM = 256, WN = 2, NT = WM * WN * 64;
    constexpr int VE = Tr<T>::VE;
    constexpr int CCH = 8 * VE;                       // channels per halo chunk (64 bf16 / 32 f32)
    constexpr int HP = 16 + KS - 1, HROWS = HP * HP;
    constexpr int A_CHUNKS = (HROWS * 8 + 63) / 64 * 64;   // whole wave-instructions; the tail DMAs zeros
    constexpr int A_PASSES = (A_CHUNKS + NT - 1) / NT;
    constexpr int B_CHUNKS = BN * 8, BCH = B_CHUNKS / NT;
    constexpr int MI = 16 / WM, NI = BN / WN / 16;    // a wave owns MI rows of the 16 x 16 patch
    constexpr int PAD = (KS - 1) / 2, TAPS = KS * KS;
    static_assert(B_CHUNKS % NT == 0 && A_PASSES + 1 <= TAPS, "tile/thread mismatch");
    __shared__ uint4 lds[2 * A_CHUNKS + 2 * B_CHUNKS];
    __shared__ __attribute__((aligned(16))) float pre_tab[PRE ? 2 * PRE_MAXC_HALO : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    // (PMC rounds 1-3: 110 MB per launch at 16x16x384 = 1.90x the algorithmic bytes.  The excess is L2 FILL traffic, served by
    // the Infinity Cache: eight XCD-private L2s each pull both 3.7 MB weight panels.  Giving every XCD one panel - bn = XCD % 2 -
    // was measured in round 3: weights 59 -> 30 MB, but the two N-tiles of a patch then sit on different XCDs and its
    // activations are pulled twice, 25 -> 50 MB: 1.83x, same time.  With two N-tiles one operand is duplicated either way.)
    const int tile_ = xcd_remap(blockIdx.x, total_tiles);
    const int bp = tile_ / n_tiles, bn = tile_ - bp * n_tiles;
    const int tile = bp * n_tiles + bn;
    const int b = bp / patches_per_img, pidx = bp - b * patches_per_img;
    const int py0 = (pidx / patches_w) * 16, px0 = (pidx % patches_w) * 16;
    const int N = g.Cout, H = g.Hin, W = g.Win;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;

    const int row0 = tid >> 3, phys = tid & 7;
    const int sc = (phys ^ (row0 & 7)) * VE;          // this thread's element offset inside a chunk row
    // weight rows of this thread
    const T* bp_[BCH];
    bool nv[BCH];
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
        int n = bn * BN + row0 + (NT / 8) * j;
        nv[j] = n < N;
        bp_[j] = wT + (long)(nv[j] ? n : 0) * w_ld + sc;
    }
    // halo pixels of this thread (one per DMA pass)
    long apix[A_PASSES];                               // pixel offset (elements) or -1
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int hrow = row0 + (NT / 8) * i;
        apix[i] = -1;
        if (hrow < HROWS) {
            const int hy = hrow / HP, hx = hrow - hy * HP;
            const int iy = py0 - PAD + hy, ix = px0 - PAD + hx;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) apix[i] = (((long)b * H + iy) * W + ix) * g.in_ld + sc;
        }
    }
    auto issue_a = [&](int buf, int cc, int i) {
        // pass i covers chunks [NT*i, NT*i + NT): wave-uniform validity (A_CHUNKS is a multiple of 64)
        if (NT * i + wave * 64 >= A_CHUNKS) return;
        const void* p = apix[i] >= 0 ? (const void*)(src + apix[i] + (long)cc * CCH) : (const void*)zeros;
        glds16(p, lds_base + (unsigned)(buf * A_CHUNKS + NT * i + wave * 64) * 16u);
    };
    auto issue_b = [&](int slot, int cc, int tap) {
        const long k = (long)tap * g.Cin + (long)cc * CCH;
#pragma unroll
        for (int j = 0; j < BCH; ++j) {
            const void* p = nv[j] ? (const void*)(bp_[j] + k) : (const void*)zeros;
            glds16(p, lds_base + (unsigned)(2 * A_CHUNKS + slot * B_CHUNKS + NT * j + wave * 64) * 16u);
        }
    };
    auto issue_b_part = [&](int slot, int cc, int tap, int j) {      // j: compile-time after unrolling
        const long k = (long)tap * g.Cin + (long)cc * CCH;
        const void* p = nv[j] ? (const void*)(bp_[j] + k) : (const void*)zeros;
        glds16(p, lds_base + (unsigned)(2 * A_CHUNKS + slot * B_CHUNKS + NT * j + wave * 64) * 16u);
    };

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ncc = g.Cin / CCH;
    const int S = ncc * TAPS;
    const int fr = lane & 15, fq = lane >> 4;
    // prologue (see ConvPre): pass i of halo chunk `pcc` (buffer pcc & 1) is normalised + activated in place by the
    // thread that DMA'd it, after its wait and before a barrier; halo pixels outside the image stay zero
    auto pre_pass = [&](int pcc, int i) {
        if (NT * i + wave * 64 >= A_CHUNKS || apix[i] < 0) return;
        uint4* slot = lds + (pcc & 1) * A_CHUNKS + NT * i + tid;
        const int c0 = pcc * CCH + sc;
        const uint4 v = pre_chunk<T>(*slot, pre_tab + c0, pre_tab + PRE_MAXC_HALO + c0, pre.act);
        *slot = v;
        if (pre.act_out && (pcc % n_tiles) == bn) {
            // interior of the halo = the patch's own 16 x 16 pixels: each is written by exactly one N-tile
            const int hrow = row0 + (NT / 8) * i;
            const int hy = hrow / HP, hx = hrow - hy * HP;
            if (hy >= PAD && hy < PAD + 16 && hx >= PAD && hx < PAD + 16)
                *(uint4*)((T*)pre.act_out + (apix[i] / g.in_ld) * pre.act_ld + (long)pcc * CCH + sc) = v;
        }
    };
    if constexpr (PRE) {
        for (int c = tid; c < g.Cin; c += NT) bn_coef<sizeof(T) == 4>(pre.bn, g.Cin, c, blockIdx.x == 0, pre_tab[c], pre_tab[PRE_MAXC_HALO + c]);
        __syncthreads();
    }
    constexpr bool PIPE4 = sizeof(T) == 2 && WM * WN == 4;
    const int dbg = vec_epi >> 8;          // diagnostic stamps (nvae_conv_halo4_enable(form | 16 or 32)); 0 in production
    vec_epi &= 0xff;
    unsigned long long st_e = 0;
    if (dbg & 16) st_e = __builtin_amdgcn_s_memrealtime();
    constexpr bool PINGPONG = sizeof(T) == 2 && WM * WN == 8;
    const bool lag = PINGPONG && wave >= 4;
    // diagnostic stamps (DVFS give-back check of the microarchitecture guide: clock = d memtime / d memrealtime x 100 MHz)
    unsigned long long st_c = 0, st_r = 0;
    if (dbg & 24) { st_c = __builtin_amdgcn_s_memtime(); st_r = __builtin_amdgcn_s_memrealtime(); }
    if constexpr (PIPE4) {
        // 16-bit, FOUR waves of 128 x 96 (one per SIMD, accumulators 192 registers): 14 operand reads per 48 MFMAs instead
        // of 20, and no second wave on the SIMD to hide them - so the loop is a hand-made software pipeline over HALF
        // k-steps ("phases", 32 channels of one tap): while the 48 MFMAs of phase p run out of one register set, the 14
        // ds_read_b128 of phase p + 1 are issued between them into the other (sched_group_barrier: one read per two
        // MFMAs, all in flight after 28 MFMAs, 20 MFMAs of slack for the last one's latency).
        // One barrier per STEP, at the end of its phase 0:
        //   weight tile B(s) lives in ring slot s & 1 and is read in phase (s-1, 1) [first half] and (s, 0) [second
        //   half]; B(s + 2) is DMA'd into the same slot in phase (s, 1), i.e. after the barrier of step s, which a wave
        //   passes only with its reads of phase (s, 0) complete; every thread waits for its own pieces of B(s + 2)
        //   before the barrier of step s + 1 (counted: at most the halo pass issued in that phase stays in flight), after
        //   which phase (s + 1, 1) reads it.  Flight time of a weight piece: almost two phases, as in the 8-wave form.
        //   Halo pass i of the next chunk is issued in phase (i, 0), is complete at the barrier of step i + 1 (where the
        //   thread that issued it normalises it in place: ConvPre) and is read from phase (24, 1) on; the buffer it
        //   lands in was last read in phase (24, 0) of the chunk before, one barrier earlier.
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) issue_a(0, 0, i);
        issue_b(0, 0, 0);
        if (S > 1) issue_b(1, 0, 1);
        wait_vmcnt<0>();
        if constexpr (PRE) {
#pragma unroll
            for (int i = 0; i < A_PASSES; ++i) pre_pass(0, i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int lb = wm * MI * HP + fr;                           // halo row of this lane for (i = 0, tap 0)
        const int bo0 = (wn * (BN / WN) + fr) * 8 + (fq ^ (fr & 7));          // weight rows are multiples of 8 apart
        const int bo1 = (wn * (BN / WN) + fr) * 8 + ((4 + fq) ^ (fr & 7));
        i32x4 af[2][MI], bf[2][NI];
        // One phase: the 14 reads of the NEXT phase's fragments (6 weight, then 8 halo), each followed by two MFMAs of
        // THIS phase, then the other 20 MFMAs.  The MFMAs are inline asm with the accumulators pinned in AGPRs ("+a"):
        // as builtins the register allocator kept half of the 192 accumulator registers in VGPRs and moved every tile
        // in and out of AGPRs around its MFMA (116 v_accvgpr_write per phase); sched_barrier keeps the source order.
#define HALO4_PHASE(CUR, NXT, HN, ABUF, BBUF, TAPOFF)                                                              \
        {                                                                                                          \
            const i32x4* bb_ = (const i32x4*)(BBUF) + ((HN) ? bo1 : bo0);                                          \
            const i32x4* ab_ = (const i32x4*)(ABUF);                                                               \
            const int tb_ = lb + (TAPOFF);                                                                         \
            _Pragma("unroll") for (int q = 0; q < MI + NI; ++q) {                                                  \
                if (q < NI) bf[NXT][q < NI ? q : 0] = bb_[q * 128];                                                \
                else {                                                                                             \
                    const int hrow = tb_ + (q - NI) * HP;                                                          \
                    af[NXT][q < NI ? 0 : q - NI] = ab_[hrow * 8 + (((HN) * 4 + fq) ^ (hrow & 7))];                 \
                }                                                                                                  \
                mfma16_agpr<T>(af[CUR][(2 * q) / NI], bf[CUR][(2 * q) % NI], acc[(2 * q) / NI][(2 * q) % NI]);      \
                mfma16_agpr<T>(af[CUR][(2 * q + 1) / NI], bf[CUR][(2 * q + 1) % NI], acc[(2 * q + 1) / NI][(2 * q + 1) % NI]); \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }                                                                                                      \
            _Pragma("unroll") for (int k = 2 * (MI + NI); k < MI * NI; ++k)                                        \
                mfma16_agpr<T>(af[CUR][k / NI], bf[CUR][k % NI], acc[k / NI][k % NI]);                             \
            __builtin_amdgcn_sched_barrier(0);                                                                     \
        }
        {   // fragments of (step 0, first half)
            const i32x4* bb_ = (const i32x4*)(lds + 2 * A_CHUNKS) + bo0;
#pragma unroll
            for (int j = 0; j < NI; ++j) bf[0][j] = bb_[j * 128];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int hrow = lb + i * HP;
                af[0][i] = ((const i32x4*)lds)[hrow * 8 + (fq ^ (hrow & 7))];
            }
        }
        int cc = 0, tap = 0, tap2 = 2 % TAPS, cc2 = 2 / TAPS;      // (tap2, cc2): two steps ahead
        for (int s = 0; s < S; ++s) {
            const int kh = tap / KS, kw = tap - kh * KS;
            int tap1 = tap + 1, cc1 = cc;
            if (tap1 == TAPS) { tap1 = 0; ++cc1; }
            const int kh1 = tap1 / KS, kw1 = tap1 - kh1 * KS;
            // ---- phase 0: MFMAs of (s, first half), reads of (s, second half), one halo pass of the next chunk
            const bool a_pass = tap < A_PASSES && cc + 1 < ncc;
            const bool a_mine = a_pass && NT * tap + wave * 64 < A_CHUNKS;
            if (a_pass) {
#pragma unroll
                for (int i = 0; i < A_PASSES; ++i)
                    if (i == tap) issue_a((cc + 1) & 1, cc + 1, i);
            }
            __builtin_amdgcn_sched_barrier(0);
            HALO4_PHASE(0, 1, 1, lds + (cc & 1) * A_CHUNKS, lds + 2 * A_CHUNKS + (s & 1) * B_CHUNKS, kh * HP + kw)
            if (a_mine) wait_vmcnt<1>(); else wait_vmcnt<0>();
            if constexpr (PRE) {
                if (tap >= 1 && tap <= A_PASSES && cc + 1 < ncc) {
#pragma unroll
                    for (int i = 0; i < A_PASSES; ++i)
                        if (i == tap - 1) pre_pass(cc + 1, i);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // ---- phase 1: MFMAs of (s, second half), reads of (s + 1, first half), DMA of the weight tile of step s + 2
            if (s + 2 < S) issue_b(s & 1, cc2, tap2);
            __builtin_amdgcn_sched_barrier(0);
            // (after the last step this reads a stale but valid tile; nothing uses it)
            HALO4_PHASE(1, 0, 0, lds + (cc1 & 1) * A_CHUNKS, lds + 2 * A_CHUNKS + ((s + 1) & 1) * B_CHUNKS, kh1 * HP + kw1)
            tap = tap1; cc = cc1;
            if (++tap2 == TAPS) { tap2 = 0; ++cc2; }
        }
#undef HALO4_PHASE
        // (the compiler does not know these were MFMAs: cover the XDL-write -> VALU-read distance by hand)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    } else {
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) issue_a(0, 0, i);
        issue_b(0, 0, 0);
        int cc = 0, tap = 0;
        // bf16, 8 waves: PING-PONG.  A SIMD hosts waves w and w + 4.  If all eight leave the per-step barrier together,
        // both waves of a SIMD read LDS at the same time (matrix pipe idle) and then fight for the matrix pipe (each
        // stalled half the time): measured 53-55 % MFMA-busy, 28 % of wave-cycles parked.  Here a step is
        // [barrier R | 20 operand reads + this wave's share of the next DMA | barrier M | 48 MFMAs from registers] and
        // waves 4-7 run ONE BARRIER behind waves 0-3 (they pass one extra barrier before the loop, waves 0-3 one after
        // it): in every interval one wave of a SIMD issues MFMAs while the other reads.  Same code for both halves.
        // Ring-slot lifetimes: B(s + 1) goes into the slot of B(s - 1), whose last reader (a lagging wave) finished
        // before the leading waves' barrier R of step s, the first barrier after which anyone issues that DMA; every
        // wave waits for its own DMA before EVERY barrier, so B(s + 1) is complete before the leading waves' barrier R
        // of step s + 1 (the lagging waves' barrier M of step s).
        // prefetch(q): what step q's leading waves issue - the weight tile of step q + 1 and, during the first taps of a
        // chunk, one pass of the next chunk's halo.  Leading waves call prefetch(s) in their read interval of step s
        // (after barrier R of step s: the lagging waves finished reading that slot in the interval before).  Lagging
        // waves call prefetch(s + 1) after THEIR barrier M of step s - the same point in time - so that their DMA
        // also has a whole MFMA phase to land before their next loop-top wait (they are the last to pass a barrier
        // before the leading waves read the tile).
        auto prefetch = [&](int q, int tapq, int ccq) {
            if (q >= S) return;
            int ntap = tapq + 1, ncc_ = ccq;
            if (ntap == TAPS) { ntap = 0; ++ncc_; }
            if (q + 1 < S) issue_b((q + 1) & 1, ncc_, ntap);
            if (tapq < A_PASSES && ccq + 1 < ncc) {
#pragma unroll
                for (int i = 0; i < A_PASSES; ++i)
                    if (i == tapq) issue_a((ccq + 1) & 1, ccq + 1, i);
            }
        };
        if (lag) {
            prefetch(0, 0, 0);
            wait_vmcnt<0>();
            if constexpr (PRE) {
                // the lagging waves' share of chunk 0's halo must be normalised BEFORE this barrier: it pairs with the
                // leading waves' barrier R(0), after which they read the halo for tap 0
#pragma unroll
                for (int i = 0; i < A_PASSES; ++i) pre_pass(0, i);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
        }
        for (int s = 0; s < S; ++s) {
            wait_vmcnt<0>();
            if constexpr (PRE) {
                if (s == 0) {
                    if (!lag) {
#pragma unroll
                        for (int i = 0; i < A_PASSES; ++i) pre_pass(0, i);
                    }
                } else if (tap >= 1 && tap <= A_PASSES && cc + 1 < ncc) {
                    // the pass issued one step ago (tap - 1) for the NEXT chunk has landed; its buffer is not read
                    // before chunk cc + 1 starts
#pragma unroll
                    for (int i = 0; i < A_PASSES; ++i)
                        if (i == tap - 1) pre_pass(cc + 1, i);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const uint4* abuf = lds + (cc & 1) * A_CHUNKS;
            const uint4* bbuf = lds + 2 * A_CHUNKS + (s & 1) * B_CHUNKS;
            const int kh = tap / KS, kw = tap - kh * KS;
            int tap1 = tap + 1, cc1 = cc;
            if (tap1 == TAPS) { tap1 = 0; ++cc1; }
            if constexpr (PINGPONG) {
                uint4 af[2][MI], bf[2][NI];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int hrow = (wm * MI + i + kh) * HP + kw + fr;
                        af[h][i] = abuf[hrow * 8 + ((h * 4 + fq) ^ (hrow & 7))];
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int r = wn * (BN / WN) + j * 16 + fr;
                        bf[h][j] = bbuf[r * 8 + ((h * 4 + fq) ^ (r & 7))];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!lag) prefetch(s, tap, cc);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // (the lagging waves issue their DMA share at the head of their MFMA interval: it must be waited for one
                // barrier earlier than the leading waves' share.  Spreading the DMA instructions between the MFMAs of
                // both halves measured 223 us against 171: the fences and M0 writes break the MFMA stream.)
                if (lag) prefetch(s + 1, tap1, cc1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) mfma_step<T>(af[h][i], bf[h][j], acc[i][j]);
            } else if constexpr (sizeof(T) == 4) {
                // f32: two-level accumulation, one partial per (tap, 32-channel chunk) -- see k_conv_gemm2
                prefetch(s, tap, cc);
                uint4 af[2][MI], bf[2][NI];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int hrow = (wm * MI + i + kh) * HP + kw + fr;
                        af[h][i] = abuf[hrow * 8 + ((h * 4 + fq) ^ (hrow & 7))];
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int r = wn * (BN / WN) + j * 16 + fr;
                        bf[h][j] = bbuf[r * 8 + ((h * 4 + fq) ^ (r & 7))];
                    }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};
                        mfma_step<T>(af[0][i], bf[0][j], part);
                        mfma_step<T>(af[1][i], bf[1][j], part);
                        acc[i][j] += part;
                    }
            } else {
                prefetch(s, tap, cc);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    uint4 af[MI], bf[NI];
#pragma unroll
                    for (int i = 0; i < MI; ++i) {
                        const int hrow = (wm * MI + i + kh) * HP + kw + fr;
                        af[i] = abuf[hrow * 8 + ((h * 4 + fq) ^ (hrow & 7))];
                    }
#pragma unroll
                    for (int j = 0; j < NI; ++j) {
                        const int r = wn * (BN / WN) + j * 16 + fr;
                        bf[j] = bbuf[r * 8 + ((h * 4 + fq) ^ (r & 7))];
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int j = 0; j < NI; ++j) mfma_step<T>(af[i], bf[j], acc[i][j]);
                }
            }
            tap = tap1; cc = cc1;
        }
    }
    if (PINGPONG && !lag) { wait_vmcnt<0>(); __builtin_amdgcn_s_barrier(); }
    if (dbg & 8) {
        st_c = __builtin_amdgcn_s_memtime() - st_c; st_r = __builtin_amdgcn_s_memrealtime() - st_r;
        if (tid == 0 && tile < 1024) { g_halo_stamps[2 * tile] = st_c; g_halo_stamps[2 * tile + 1] = st_r; }
        return;
    }
    unsigned long long st_l = 0;
    if (dbg & 16) st_l = __builtin_amdgcn_s_memrealtime();
    int ticket = 0;
    conv_epilogue<T, BM, BN, WM, WN, BNBWD>(acc, (float*)lds, g, bias, residual, out, out_f32, bp, bn, stats, vec_epi,
                                            [&](int r) -> long { return ((long)b * H + py0 + (r >> 4)) * W + px0 + (r & 15); }, be,
                                            sfin.counter, ticket);
    if constexpr (!BNBWD) {
        if (stats && sfin.counter && bn_was_last(sfin.counter + bn, ticket, be.m_tiles))
            bn_fin_fwd<sizeof(T) == 4>(sfin, stats, be.rows, N, bn * BN, (BN + 63) / 64);
    }
    if (dbg & 16) {
        // absolute 100 MHz stamps: entry, loop start, loop end, exit (stores issued, not necessarily landed)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0 && tile < 512) {
            g_halo_stamps[4 * tile] = st_e; g_halo_stamps[4 * tile + 1] = st_r; g_halo_stamps[4 * tile + 2] = st_l;
            g_halo_stamps[4 * tile + 3] = __builtin_amdgcn_s_memrealtime();
        }
    }
}

// =========================================================================================
// k_conv_img: 3 x 3, stride 1, 'same' convolutions on WHOLE small images (4 x 4 or 8 x 8) -- the two convs of every
// EncodingResidualCell of the 4x4 / 8x8 towers (encoder.py:92-98) and their data gradients, 120 launches per C2 step.
//   Measured (tools/mb_smallconv.py, round 3): k_conv_gemm2 on these shapes is bound by the bytes a CU can pull through
//   its L1 (~50 GB/s per CU whatever the ring depth, tile or K split: 32 x 64 tiles stage 442 KB per workgroup for the
//   256 -> 256 conv at 4x4, 12.8 us; split-K moves the cost into its slab hand-off).  An im2col gather fetches every
//   activation nine times.  Here the M-tile is 128 output pixels = 8 (4x4) or 2 (8x8) WHOLE images: their activations
//   (128 x Cin) are DMA'd into LDS once and every tap reads its shifted window from there (a lane whose source pixel
//   lies outside the image contributes zeros); with a 16-column N-tile a workgroup stages 64 + 72 KB for the same conv
//   (3.2x fewer bytes) and the launch still has 256 workgroups.  K order is (64-channel chunk, tap), so the first
//   chunk's MFMAs start when a quarter of the operands has landed.
//   LDS: A [Cin/64][128 rows][8 x 16 B], B [Cin/64][9 taps][16 cols][8 x 16 B], slot = chunk ^ (row & 7) as in
//   k_conv_halo (conflict-free at any row shift).  One wave = 16 pixels x 16 columns, one accumulator tile.
//   The BatchNorm(+Swish) in front of the conv is applied to the staged tile ONCE per workgroup (ConvPre): unlike the
//   im2col kernels' prologue (redone per tap) that costs 64 elements per thread, so the apply launch disappears.
// =========================================================================================
template <typename T, int HW, int CIN, bool BNBWD, bool PRE>
__global__ __launch_bounds__(512) void k_conv_img(
    NvaeConvGeom g, const T* __restrict__ src, const T* __restrict__ wT, int w_ld,
    const float* __restrict__ bias, const T* residual, void* out, int out_f32, int M, int n_tiles, int total_tiles,
    const uint4* __restrict__ zeros, float* stats, int vec_epi, ConvBnBwd be, ConvPre pre, BnFinArgs sfin) {
    static_assert(sizeof(T) == 2, "16-bit activations only");
    constexpr int BM = 128, BN = 16, NT = 512, P = HW * HW;
    constexpr int NCC = CIN / 64;                       // 64-channel chunks
    constexpr int A_CC = BM * 8, B_CC = 9 * BN * 8;     // 16-B chunks per channel chunk: 1024, 1152
    constexpr int B_MAIN = 1024, B_LEFT = B_CC - B_MAIN;   // per chunk: 2 per thread + 128 left over
    constexpr int LEFT = NCC * B_LEFT;                  // <= 512: one DMA of (some of) the waves, issued first
    static_assert(CIN % 64 == 0 && LEFT <= NT && B_LEFT == 128, "channel count");
    __shared__ uint4 lds[NCC * (A_CC + B_CC)];
    __shared__ __attribute__((aligned(16))) float pre_tab[PRE ? 2 * CIN : 4];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = xcd_remap(blockIdx.x, total_tiles);
    const int bm = tile / n_tiles, bn = tile - bm * n_tiles;
    const int N = g.Cout;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
    const unsigned b_base = lds_base + (unsigned)(NCC * A_CC) * 16u;

    // ---- this thread's DMA sources.  A: chunk q = tid + 512*i of a channel chunk -> row q >> 3, slot q & 7
    const int phys = tid & 7;
    const T* a_src[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = (tid >> 3) + 64 * i;
        const long m = (long)bm * BM + row;
        a_ok[i] = m < M;
        a_src[i] = src + (a_ok[i] ? m : 0) * g.in_ld + ((phys ^ (row & 7)) << 3);
    }
    // B main part: chunk q = tid + 512*i (< 1024) of a channel chunk -> (tap, col) = q >> 3, slot q & 7
    const T* b_src[2];
    bool b_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int rc = (tid >> 3) + 64 * i;              // tap * 16 + col
        const int tap = rc >> 4, col = rc & 15;
        const int n = bn * BN + col;
        b_ok[i] = n < N;
        b_src[i] = wT + (long)(b_ok[i] ? n : 0) * w_ld + tap * CIN + ((phys ^ (col & 7)) << 3);
    }
    // B left-over: the last 128 chunks (tap 8) of every channel chunk, one DMA per thread of the first LEFT threads
    if (wave * 64 < LEFT) {
        const int cc = wave >> 1, q = B_MAIN + (wave & 1) * 64 + lane;        // (B_LEFT = 128 chunks = two waves per channel chunk)
        const int rc = q >> 3, col = rc & 15, tap = rc >> 4;
        const int n = bn * BN + col;
        const void* p = n < N ? (const void*)(wT + (long)n * w_ld + tap * CIN + cc * 64 + (((q & 7) ^ (col & 7)) << 3))
                              : (const void*)zeros;
        glds16(p, b_base + (unsigned)(cc * B_CC + B_MAIN + (wave & 1) * 64) * 16u);
    }
    auto issue_cc = [&](int cc) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(a_ok[i] ? (const void*)(a_src[i] + cc * 64) : (const void*)zeros,
                   lds_base + (unsigned)(cc * A_CC + 512 * i + wave * 64) * 16u);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            glds16(b_ok[i] ? (const void*)(b_src[i] + cc * 64) : (const void*)zeros,
                   b_base + (unsigned)(cc * B_CC + 512 * i + wave * 64) * 16u);
    };
    issue_cc(0);
    if constexpr (PRE) {
        // the slab -> coefficient chain (a dependent global round trip + f64 arithmetic) runs behind chunk 0's DMA; hipcc
        // waits vmcnt(0) for these loads, i.e. for chunk 0 as well, which is needed first anyway
        for (int c = tid; c < CIN; c += NT) bn_coef<false>(pre.bn, CIN, c, blockIdx.x == 0, pre_tab[c], pre_tab[CIN + c]);
    }
#pragma unroll
    for (int cc = 1; cc < NCC; ++cc) issue_cc(cc);

    // ---- per-lane tap geometry: this lane's output pixel and which of the 9 source pixels exist
    const int fr = lane & 15, fq = lane >> 4;
    const int r = wave * 16 + fr;                        // tile row of the A fragment this lane reads
    const int pix = r % P, py = pix / HW, px = pix % HW;
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cc = 0; cc < NCC; ++cc) {
        // chunk cc (and the left-over DMA, which is older) has landed; 4 DMAs per later chunk may stay in flight
        if (cc == NCC - 1) wait_vmcnt<0>();
        else if (cc == NCC - 2) wait_vmcnt<4>();
        else if (cc == NCC - 3) wait_vmcnt<8>();
        else wait_vmcnt<12>();
        if constexpr (PRE) {
            if (cc == 0) __syncthreads();                // coefficient table complete
            const bool mine = pre.act_out != nullptr;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!a_ok[i]) continue;
                const int row = (tid >> 3) + 64 * i;
                const int c0 = cc * 64 + ((phys ^ (row & 7)) << 3);
                uint4* slot = lds + cc * A_CC + 512 * i + tid;
                const uint4 v = pre_chunk<T>(*slot, pre_tab + c0, pre_tab + CIN + c0, pre.act);
                *slot = v;
                // the activated tensor is written once (for this conv's weight gradient): chunk c0/8 by N-tile (c0/8) % n_tiles
                if (mine && ((c0 >> 3) % n_tiles) == bn)
                    *(uint4*)((T*)pre.act_out + ((long)bm * BM + row) * pre.act_ld + c0) = v;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint4* abuf = lds + cc * A_CC;
        const uint4* bbuf = lds + NCC * A_CC + cc * B_CC;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const bool ok = (unsigned)(py + dy) < (unsigned)HW && (unsigned)(px + dx) < (unsigned)HW;
            const int rs = ok ? r + dy * HW + dx : r;    // source row (own row when the tap falls outside: zeroed below)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                uint4 a = abuf[rs * 8 + ((h * 4 + fq) ^ (rs & 7))];
                const uint4 b = bbuf[(tap * 16 + fr) * 8 + ((h * 4 + fq) ^ (fr & 7))];
                if (!ok) a = make_uint4(0u, 0u, 0u, 0u);
                if ((tap + h) & 1) mfma_step<T>(a, b, acc1);
                else mfma_step<T>(a, b, acc0);
            }
        }
    }
    f32x4 acc[1][1];
    acc[0][0] = acc0 + acc1;
    int ticket = 0;
    conv_epilogue<T, BM, BN, 8, 1, BNBWD>(acc, (float*)lds, g, bias, residual, out, out_f32, bm, bn, stats, vec_epi,
                                          [&](int rr) -> long { long m = (long)bm * BM + rr; return m < M ? m : -1L; }, be,
                                          sfin.counter, ticket);
    if constexpr (!BNBWD) {
        if (stats && sfin.counter && bn_was_last(sfin.counter + bn, ticket, be.m_tiles))
            bn_fin_fwd<false>(sfin, stats, be.rows, N, bn * BN, 1);
    }
}

// 5x5 halo kernel, 16-bit: four waves of 128 x 96 with a software-pipelined loop (1) or eight ping-pong waves of 64 x 96 (0)
static bool g_halo4 = false;
static int g_halo4_dbg = 0;
extern "C" int nvae_conv_halo4_enable(int on) { g_halo4 = (on & 1) != 0; g_halo4_dbg = on >> 1; return NVAE_OK; }
extern "C" int nvae_conv_halo_stamps(unsigned long long* host_out, int n) {
    NVAE_REQUIRE(host_out && n > 0 && n <= 2048, "nvae_conv_halo_stamps: bad arguments");
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_halo_stamps), (size_t)n * sizeof(unsigned long long));
    if (e != hipSuccess) NVAE_FAIL(NVAE_ELAUNCH, "nvae_conv_halo_stamps: %s", hipGetErrorString(e));
    return NVAE_OK;
}
// whole-image kernel eligibility (must agree between the launcher and nvae_conv_gemm_stats_rows)
static bool g_conv_img = true;
extern "C" int nvae_conv_img_enable(int on) { g_conv_img = on != 0; return NVAE_OK; }
static bool conv_img_ok(int dtype, const NvaeConvGeom* g) {
    return g_conv_img && is16(dtype) && g->KH == 3 && g->KW == 3 && g->stride == 1 && g->div == 1 && g->pad_t == 1 &&
           g->pad_l == 1 && g->Hin == g->Hout && g->Win == g->Wout && g->Hin == g->Win && (g->Hin == 4 || g->Hin == 8) &&
           (g->Cin == 128 || g->Cin == 256) && g->Cout >= 32 && (long)g->B * g->Hin * g->Win >= 256;
}

#ifndef HALO_WM
#define HALO_WM 4      // waves along M of the halo kernel: 4 -> 8 waves of 64 x 96, 2 -> 4 waves of 128 x 96
#endif
// halo kernel eligibility (must agree between the launcher and nvae_conv_gemm_stats_rows)
static bool conv_halo_ok(int dtype, const NvaeConvGeom* g) {
    const int cch = is16(dtype) ? 64 : 32;
    const int N = g->Cout;
    const long w192 = (long)cdiv(N, 192) * 192, w128 = (long)cdiv(N, 128) * 128;
    return g->KH == g->KW && (g->KH == 5 || g->KH == 3) && g->stride == 1 && g->div == 1 &&
           g->pad_t == (g->KH - 1) / 2 && g->pad_l == (g->KW - 1) / 2 && g->Hin == g->Hout && g->Win == g->Wout &&
           g->Hin % 16 == 0 && g->Win % 16 == 0 && g->Cin % cch == 0 && g->Cin >= 2 * cch && w192 <= w128 &&
           (long)g->B * (g->Hin / 16) * (g->Win / 16) * cdiv(N, 192) >= 64;
}

// Tile family of the generic kernel for a geometry (tools/tune_conv.py sweeps them on the tower shapes):
//  1: 256x192 (2-deep ring)  2: 128x192  3: 128x128  4: 128x64  5: 32x64, 128-deep steps  6: 64x64, 128-deep  7: 64x64
//  14: 128x128, 2-deep ring, two workgroups per CU (8-13: experimental, nvae_conv_gemm_force_tile only)
static int conv_tile_family(const NvaeConvGeom* g) {
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    const long big_tiles = (long)cdiv(M, 128) * cdiv(N, 128);
    const long w192 = (long)cdiv(N, 192) * 192, w128 = (long)cdiv(N, 128) * 128, w64 = (long)cdiv(N, 64) * 64;
    // the FLOP-dominant layers: 256 x 192 tile (112 FLOP per staged byte), 2-deep ring (112 KB); also the
    // single-K-step 1x1 convs on >= 16x16 images, which are bound by their output stores (fewer, fatter epilogues)
    if ((long)cdiv(M, 256) * cdiv(N, 192) >= 224 && w192 <= w128 && w192 <= w64 && (K >= 1024 || K <= 64)) return 1;
    // large problems: 128-row tiles, 8 waves; N tile with the least padding (ties -> larger).  With a short K loop
    // (K < 512) the 128-row tiles need twice the tiles to beat 64 x 64 (256 -> 1536 at 4x4: 10.2 vs 7.7 us)
    if (big_tiles >= (K < 512 ? 256 : 192)) {
        // short K loop and at least two rounds of workgroups: these launches are prologue + epilogue (C4's 256 -> 1536 1x1 convs
        // at 16x16: 4 ring steps, then 48 KB of stores per tile), and a 3-deep 128 x 192 ring (123 KB) leaves one workgroup per
        // CU, so nothing overlaps them.  128 x 128 with a 2-deep ring is 70 KB: two workgroups per CU
        if (K <= 512 && big_tiles >= 512 && w128 <= w64) return 14;
        if (w192 <= w128 && w192 <= w64) return 2;
        if (w128 <= w64) return 3;
        return 4;
    }
    if (K >= 512) {
        // small M: latency-bound K loop -> 8 waves, 128-deep ring steps (half the barriers).  These are bound by the
        // bytes each CU can pull in ((BM + BN) * K * 2 per workgroup at ~22 B/cycle/CU): when 64-row tiles leave CUs
        // idle, 32-row tiles use the whole chip and cut the per-CU bytes by 25 %
        return (long)cdiv(M, 64) * cdiv(N, 64) <= 160 ? 5 : 6;
    }
    return 7;
}

static int conv_gemm_bm_noimg(int dtype, const NvaeConvGeom* g) {
    if (conv_halo_ok(dtype, g)) return 256;
    static const int bm[15] = {0, 256, 128, 128, 128, 32, 64, 64, 0, 0, 0, 0, 0, 0, 128};
    return bm[conv_tile_family(g)];
}
static int conv_gemm_bm(int dtype, const NvaeConvGeom* g) {
    // M-tile height the launcher will pick (the statistics slab is sized from it: nvae_conv_gemm_stats_rows)
    if (conv_img_ok(dtype, g)) return 128;
    return conv_gemm_bm_noimg(dtype, g);
}

// which kernel a geometry gets (tests): -1 whole-image kernel, -2 halo kernel, else the tile family of k_conv_gemm2
extern "C" int nvae_conv_gemm_family(int dtype, const NvaeConvGeom* g) {
    if (!g) return 0;
    if (is16(dtype) && conv_img_ok(dtype, g)) return -1;
    if (conv_halo_ok(dtype, g)) return -2;
    return conv_tile_family(g);
}
// tuning hook (tools/tune_conv.py): 0 = the launcher's own choice, 1..7 = force a tile family of k_conv_gemm2
static int g_force_tile = 0;
extern "C" int nvae_conv_gemm_force_tile(int t) { g_force_tile = t; return NVAE_OK; }
// tuning hook (tools/mb_smallconv.py): 0 = the launcher's own choice, S >= 1 = force S K-slices per tile
static int g_force_split = 0;
extern "C" int nvae_conv_gemm_force_split(int s) { g_force_split = s; return NVAE_OK; }

// Split-K workspace: partial-tile slabs + per-tile arrival counters (ZEROED by the caller once; the kernels leave
// them zero).  One workspace serves every launch of a stream (stream order separates their use of it); launches
// on different streams that may overlap must not share one.  Without a workspace no launch is split.
static float* g_ws_slab = nullptr;
static size_t g_ws_bytes = 0;
static int* g_ws_counters = nullptr;
static int g_ws_ncounters = 0;
extern "C" int nvae_conv_set_workspace(void* slab, size_t bytes, int* counters, int n_counters) {
    NVAE_REQUIRE((slab && counters && bytes > 0 && n_counters > 0) || (!slab && !counters),
                 "conv_set_workspace: slab and counters must both be given (or both NULL)");
    NVAE_REQUIRE(aligned16(slab), "conv_set_workspace: slab must be 16-B aligned");
    g_ws_slab = (float*)slab; g_ws_bytes = slab ? bytes : 0; g_ws_counters = counters; g_ws_ncounters = slab ? n_counters : 0;
    return NVAE_OK;
}
// K-slices per output tile for a geometry run with BM x BN tiles and `nk` ring steps (0/1 = not split)
// Measured (tools/mb_smallconv.py, profiles/r03_mb_smallconv.txt): on the tower shapes with >= 128 output channels
// no split beats the unsplit launch by more than 0.5 us - the kernels are bound by the bytes a CU pulls through its
// L1 and the slab hand-off costs what the shorter K loop saves - so those are not split (the 3x3 ones run
// k_conv_img instead).  Narrow layers whose grid leaves most of the chip idle are: the 3x3 sampler convs
// (common.py:36-47, 256 -> 40 at 4x4: 64 workgroups) run 12.2 -> 8.2 us with three K-slices.
static int conv_split_choice(const NvaeConvGeom* g, int BM, int BN, int nk, bool eight_waves) {
    const long tiles = (long)cdiv((long)g->B * g->Hout * g->Wout, BM) * cdiv(g->Cout, BN);
    if (eight_waves && g->Cout <= 64 && nk >= 9 && tiles <= 96) return 3;
    return 1;
}

template <typename T>
static int launch_conv_gemm(const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                            const float* bias, const void* residual, void* out, int out_f32,
                            float* stats, hipStream_t s, const ConvBnBwd* fuse = nullptr,
                            const ConvPre* prologue = nullptr, const BnFinArgs* stats_fin = nullptr) {
    ConvBnBwd be{};
    if (fuse) be = *fuse;
    ConvPre pre{};
    if (prologue) pre = *prologue;
    BnFinArgs sfin{};
    if (stats_fin) sfin = *stats_fin;
    const bool use_pre = pre.on;
    // whole-image kernel unless an in-kernel finalize is requested (that works on 64-channel groups, wider than its
    // 16-column tiles): then the generic kernel runs on a slab the host sized for 128-row tiles (fewer rows, more adders)
    const bool img_run = sizeof(T) == 2 && conv_img_ok(dtype_of<T>(), g) && !sfin.counter && !be.fin.counter;
    be.m_tiles = cdiv((long)g->B * g->Hout * g->Wout, img_run ? 128 : conv_gemm_bm_noimg(dtype_of<T>(), g));
    be.rows = slab_rows_for(cdiv((long)g->B * g->Hout * g->Wout, conv_gemm_bm(dtype_of<T>(), g)));
    const int M = g->B * g->Hout * g->Wout, K = g->KH * g->KW * g->Cin, N = g->Cout;
    FastDiv fd_hw = make_fastdiv((unsigned)(g->Hout * g->Wout)), fd_w = make_fastdiv((unsigned)g->Wout);
    const uint4* zeros = zero_page();
    // 16-B row-contiguous stores need aligned rows; otherwise the scalar epilogue
    const int vo = out_f32 ? 4 : (int)(16 / sizeof(T));
    const int vec_epi = (g->out_ld % vo == 0) && aligned16(out) &&
                        (!residual || (g->res_ld % (int)(16 / sizeof(T)) == 0 && aligned16(residual)));
    if (be.x && (!vec_epi || out_f32 || N % 8 != 0)) return 1;     // fusion needs the vector epilogue
    if constexpr (sizeof(T) == 2) {
        if (img_run) {
            const int mt = cdiv(M, 128), nt = cdiv(N, 16);
#define LAUNCH_IMG(HW_, CIN_, F_, P_)                                                                         \
            hipLaunchKernelGGL((k_conv_img<T, HW_, CIN_, F_, P_>), mt * nt, 512, 0, s, *g, (const T*)src, (const T*)wT, w_ld, \
                               bias, (const T*)residual, out, out_f32, M, nt, mt * nt, zeros, stats, vec_epi, be, pre, sfin);
#define LAUNCH_IMG_V(HW_, CIN_)                                                                               \
            { if (be.x) LAUNCH_IMG(HW_, CIN_, true, false) else if (use_pre) LAUNCH_IMG(HW_, CIN_, false, true)  \
              else LAUNCH_IMG(HW_, CIN_, false, false) }
            if (g->Hin == 4 && g->Cin == 256) LAUNCH_IMG_V(4, 256)
            else if (g->Hin == 4) LAUNCH_IMG_V(4, 128)
            else if (g->Cin == 256) LAUNCH_IMG_V(8, 256)
            else LAUNCH_IMG_V(8, 128)
#undef LAUNCH_IMG_V
#undef LAUNCH_IMG
            return 0;
        }
    }
    if (conv_halo_ok(dtype_of<T>(), g)) {
        const int pw = g->Win / 16, ppi = (g->Hin / 16) * pw;
        const int mt = g->B * ppi, nt = cdiv(N, 192);
        const int vec_epi_plain = vec_epi;
        const int vec_epi = vec_epi_plain | (g_halo4_dbg << 8);     // experiment bits ride in the high bits (0 in production)
        if (use_pre && g->Cin > PRE_MAXC_HALO) return 2;
#define LAUNCH_HALO(KS_, WM_, F_, P_)                                                                     \
        hipLaunchKernelGGL((k_conv_halo<T, 192, KS_, WM_, F_, P_>), mt * nt, WM_ * 128, 0, s, *g, (const T*)src, \
                           (const T*)wT, w_ld, bias, (const T*)residual, out, out_f32, nt, mt * nt, pw, ppi,  \
                           zeros, stats, vec_epi, be, pre, sfin);
        if constexpr (sizeof(T) == 2) {
            if (g->KH == 5 && g_halo4) {
                if (be.x) LAUNCH_HALO(5, 2, true, false) else if (use_pre) LAUNCH_HALO(5, 2, false, true) else LAUNCH_HALO(5, 2, false, false)
                return 0;
            }
        }
        if (g->KH == 5) { if (be.x) LAUNCH_HALO(5, HALO_WM, true, false) else if (use_pre) LAUNCH_HALO(5, HALO_WM, false, true) else LAUNCH_HALO(5, HALO_WM, false, false) }
        else { if (be.x) LAUNCH_HALO(3, 4, true, false) else if (use_pre) LAUNCH_HALO(3, 4, false, true) else LAUNCH_HALO(3, 4, false, false) }
#undef LAUNCH_HALO
        return 0;
    }
    if (use_pre && g->Cin > PRE_MAXC) return 2;
#define LAUNCH2F(BM_, BN_, WM_, WN_, ST_, BKC_, F_, P_)                                                 \
    {                                                                                                   \
        int mt = cdiv(M, BM_), nt = cdiv(N, BN_);                                                       \
        const int nk = cdiv(K, BKC_ * (is16(dtype_of<T>()) ? 8 : 4));                                   \
        ConvSplitK sk{1, nk, g_ws_slab, g_ws_counters};                                                 \
        int S = g_force_split ? g_force_split : conv_split_choice(g, BM_, BN_, nk, WM_ * WN_ == 8);     \
        if (S > nk) S = nk;                                                                             \
        if (S > 1 && (BM_ / WM_ / 16) * (BN_ / WN_ / 16) <= 4 && g_ws_slab && !sfin.counter && mt * nt <= g_ws_ncounters && \
            (size_t)mt * nt * S * BM_ * BN_ * 4 <= g_ws_bytes) {                                        \
            sk.steps = cdiv(nk, S); sk.S = cdiv(nk, sk.steps);                                          \
        }                                                                                               \
        hipLaunchKernelGGL((k_conv_gemm2<T, BM_, BN_, WM_, WN_, ST_, BKC_, F_, P_>), mt * nt * sk.S, WM_ * WN_ * 64, 0, s, *g, \
                           (const T*)src, (const T*)wT, w_ld, bias, (const T*)residual, out, out_f32, M, \
                           K, nt, mt * nt, fd_hw, fd_w, zeros, stats, vec_epi, be, pre, sfin, sk);      \
    }
#define LAUNCH2(BM_, BN_, WM_, WN_, ST_, BKC_)                                                          \
    { if (be.x) LAUNCH2F(BM_, BN_, WM_, WN_, ST_, BKC_, true, false)                                    \
      else if (use_pre) LAUNCH2F(BM_, BN_, WM_, WN_, ST_, BKC_, false, true)                            \
      else LAUNCH2F(BM_, BN_, WM_, WN_, ST_, BKC_, false, false) }
    // Large problems: 128-row tiles, 8 waves; N tile with the least padding (ties -> larger).
    // Small problems (few tiles): 64 x 64 tiles, 4 waves, so that the grid covers the chip.
    int family = conv_tile_family(g);
    // An operand prologue is redone by every N-tile that stages the element, so its cost follows the number of N-tiles:
    // the short-K expand convs (256 -> 1536 at 4x4) run 64-wide tiles without it (7.9 us) and 128-wide ones with it
    // (11.8 us against 16.5; tools/tune_wide.py).  The statistics slab is indexed modulo its row count, so a tile
    // height other than conv_gemm_bm()'s stays correct (the in-kernel finalize counts M-tiles, so not with it).
    if (use_pre && !sfin.counter && family == 7 && N >= 1024 && N % 128 == 0 && M % 128 == 0) family = 3;
    // deeper rings (experimental families 8-11): no operand-prologue instantiation (its table would not fit beside the ring)
#define LAUNCH2X(BM_, BN_, WM_, WN_, ST_, BKC_)                                                         \
    { if (be.x) LAUNCH2F(BM_, BN_, WM_, WN_, ST_, BKC_, true, false)                                    \
      else LAUNCH2F(BM_, BN_, WM_, WN_, ST_, BKC_, false, false) }
    int pick = (g_force_tile && !be.x) ? g_force_tile : family;
    if (pick >= 8 && pick <= 13 && use_pre) pick = family;
    if (pick == 14 && use_pre && g->Cin > PRE_MAXC_2CU) pick = 3;
    switch (pick) {
        case 8: LAUNCH2X(32, 64, 2, 4, 4, 16) break;
        case 9: LAUNCH2X(32, 64, 2, 4, 6, 16) break;
        case 10: LAUNCH2X(64, 64, 2, 4, 4, 16) break;
        case 11: LAUNCH2X(32, 64, 2, 2, 6, 8) break;
        case 12: LAUNCH2X(64, 128, 2, 4, 3, 16) break;
        case 13: LAUNCH2X(64, 128, 2, 4, 4, 8) break;
        case 14: LAUNCH2(128, 128, 2, 4, 2, 8) break;
        case 1: LAUNCH2(256, 192, 4, 2, 2, 8) break;
        case 2: LAUNCH2(128, 192, 2, 4, 3, 8) break;
        case 3: LAUNCH2(128, 128, 2, 4, 3, 8) break;
        case 4: LAUNCH2(128, 64, 4, 2, 3, 8) break;
        case 5: LAUNCH2(32, 64, 2, 4, 3, 16) break;
        case 6: LAUNCH2(64, 64, 2, 4, 3, 16) break;
        default: LAUNCH2(64, 64, 2, 2, 3, 8) break;
    }
#undef LAUNCH2
#undef LAUNCH2X
#undef LAUNCH2F
    return 0;
}

// 1 if the geometry runs on the whole-image 3x3 kernel (whose operand prologue is worth using: ops.conv2d asks)
extern "C" int nvae_conv_img_ok(int dtype, const NvaeConvGeom* g) { return g && conv_img_ok(dtype, g) ? 1 : 0; }

extern "C" int nvae_conv_gemm_stats_rows(int dtype, const NvaeConvGeom* g) {
    if (!g) return 0;
    return slab_rows_for(cdiv((long)g->B * g->Hout * g->Wout, conv_gemm_bm(dtype, g)));
}

extern "C" int nvae_conv_gemm(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                              const float* bias, const void* residual, void* out, int out_f32,
                              float* stats, void* stream) {
    if (int e = check_geom_mfma("conv_gemm", g)) return e;
    NVAE_REQUIRE(src && wT && out, "conv_gemm: NULL pointer");
    const int ve = is16(dtype) ? 8 : 4;
    NVAE_REQUIRE(g->Cin % ve == 0 && g->in_ld % ve == 0 && w_ld % ve == 0,
                 "conv_gemm: Cin=%d in_ld=%d w_ld=%d must be multiples of %d (use nvae_conv_direct)", g->Cin, g->in_ld, w_ld, ve);
    NVAE_REQUIRE(w_ld >= g->KH * g->KW * g->Cin, "conv_gemm: w_ld too small");
    NVAE_REQUIRE(aligned16(src) && aligned16(wT), "conv_gemm: src/wT must be 16-B aligned");
    NVAE_REQUIRE(!residual || g->res_ld >= g->Cout, "conv_gemm: res_ld too small");
    DISPATCH_T(dtype, launch_conv_gemm<T>(g, src, wT, w_ld, bias, residual, out, out_f32, stats, (hipStream_t)stream);)
    NVAE_LAUNCH_CHECK("conv_gemm");
    return NVAE_OK;
}


// nvae_conv_gemm with (a) the BatchNorm(+Swish) in FRONT of the conv applied to the gathered operand inside the
// kernel (ConvPre) and (b) the BatchNorm BEHIND it finalized by the last workgroups to arrive (fin): with both, a
// conv -> BN -> act -> conv chain is one launch per conv and the normalised activations never touch HBM on their own.
extern "C" int nvae_conv_gemm_ex(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                                 const float* bias, const void* residual, void* out, int out_f32, float* stats,
                                 const NvaeConvPre* pre, const NvaeBnFin* fin, void* stream) {
    if (int e = check_geom_mfma("conv_gemm_ex", g)) return e;
    NVAE_REQUIRE(src && wT && out, "conv_gemm_ex: NULL pointer");
    const int ve = is16(dtype) ? 8 : 4;
    NVAE_REQUIRE(g->Cin % ve == 0 && g->in_ld % ve == 0 && w_ld % ve == 0,
                 "conv_gemm_ex: Cin=%d in_ld=%d w_ld=%d must be multiples of %d (use nvae_conv_direct)", g->Cin, g->in_ld, w_ld, ve);
    NVAE_REQUIRE(w_ld >= g->KH * g->KW * g->Cin, "conv_gemm_ex: w_ld too small");
    NVAE_REQUIRE(aligned16(src) && aligned16(wT), "conv_gemm_ex: src/wT must be 16-B aligned");
    NVAE_REQUIRE(!residual || g->res_ld >= g->Cout, "conv_gemm_ex: res_ld too small");
    ConvPre p{};
    if (pre) {
        const NvaeBnIn* in = &pre->bn;
        NVAE_REQUIRE(in->scale && in->shift, "conv_gemm_ex: prologue needs scale / shift");
        p.on = true;
        p.bn.scale = in->scale; p.bn.shift = in->shift; p.bn.mean = in->mean; p.bn.invstd = in->invstd;
        if (in->slab) {
            NVAE_REQUIRE(in->rows > 0 && in->gamma && in->beta && in->rm && in->rv && in->mean && in->invstd,
                         "conv_gemm_ex: NvaeBnIn with a slab needs rows, gamma, beta, rm, rv, mean, invstd");
            p.bn.slab = in->slab; p.bn.rows = in->rows; p.bn.inv_n = 1.0f / (float)((long)g->B * g->Hin * g->Win);
            p.bn.eps = in->eps; p.bn.momentum = in->momentum; p.bn.gamma = in->gamma; p.bn.beta = in->beta;
            p.bn.rm = in->rm; p.bn.rv = in->rv;
        }
        NVAE_REQUIRE(pre->act == ACT_NONE || pre->act == ACT_SWISH, "conv_gemm_ex: prologue act %d unsupported", pre->act);
        NVAE_REQUIRE(g->div == 1, "conv_gemm_ex: the prologue does not combine with an upsampled / dilated gather");
        NVAE_REQUIRE(!pre->act_out || (g->stride == 1 && g->Hin == g->Hout && g->Win == g->Wout && g->pad_t >= 0 &&
                                       g->pad_t < g->KH && g->pad_l >= 0 && g->pad_l < g->KW && aligned16(pre->act_out) &&
                                       pre->act_ld % ve == 0 && pre->act_ld >= g->Cin),
                     "conv_gemm_ex: act_out needs a stride-1 'same' geometry (every source pixel is some output's centre tap)");
        p.act = pre->act; p.act_out = pre->act_out; p.act_ld = pre->act_ld;
    }
    BnFinArgs f{};
    if (fin) {
        NVAE_REQUIRE(stats && fin->counter && fin->gamma && fin->beta && fin->rm && fin->rv && fin->scale && fin->shift &&
                     fin->mean && fin->invstd, "conv_gemm_ex: in-kernel finalize needs the slab and every NvaeBnFin field");
        f.counter = fin->counter; f.inv_n = 1.0f / (float)((long)g->B * g->Hout * g->Wout); f.gamma = fin->gamma;
        f.beta = fin->beta; f.rm = fin->rm; f.rv = fin->rv; f.momentum = fin->momentum; f.eps = fin->eps;
        f.scale = fin->scale; f.shift = fin->shift; f.mean = fin->mean; f.invstd = fin->invstd;
    }
    int rc = 0;
    DISPATCH_T(dtype, rc = launch_conv_gemm<T>(g, src, wT, w_ld, bias, residual, out, out_f32, stats, (hipStream_t)stream,
                                               nullptr, pre ? &p : nullptr, fin ? &f : nullptr);)
    NVAE_REQUIRE(rc == 0, "conv_gemm_ex: Cin=%d exceeds the prologue's coefficient table for this tile family", g->Cin);
    NVAE_LAUNCH_CHECK("conv_gemm_ex");
    return NVAE_OK;
}

// Largest Cin the prologue of nvae_conv_gemm_ex takes for this geometry (0: unsupported geometry).
extern "C" int nvae_conv_gemm_pre_max_cin(int dtype, const NvaeConvGeom* g) {
    if (!g || g->div != 1) return 0;
    return conv_halo_ok(dtype, g) ? PRE_MAXC_HALO : PRE_MAXC;
}

// nvae_conv_gemm for a data gradient whose destination feeds a BatchNorm backward: see ConvBnBwd.
extern "C" int nvae_conv_gemm_bnbwd(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                                    const float* bias, const void* residual, void* out,
                                    const NvaeBnBwdFuse* f, void* stream) {
    if (int e = check_geom_mfma("conv_gemm_bnbwd", g)) return e;
    NVAE_REQUIRE(src && wT && out && f, "conv_gemm_bnbwd: NULL pointer");
    const int ve = is16(dtype) ? 8 : 4;
    NVAE_REQUIRE(g->Cin % ve == 0 && g->in_ld % ve == 0 && w_ld % ve == 0,
                 "conv_gemm_bnbwd: Cin=%d in_ld=%d w_ld=%d must be multiples of %d", g->Cin, g->in_ld, w_ld, ve);
    NVAE_REQUIRE(w_ld >= g->KH * g->KW * g->Cin, "conv_gemm_bnbwd: w_ld too small");
    NVAE_REQUIRE(aligned16(src) && aligned16(wT), "conv_gemm_bnbwd: src/wT must be 16-B aligned");
    NVAE_REQUIRE(!residual || g->res_ld >= g->Cout, "conv_gemm_bnbwd: res_ld too small");
    NVAE_REQUIRE(f->x && f->scale && f->shift && f->partials, "conv_gemm_bnbwd: NULL field in NvaeBnBwdFuse");
    NVAE_REQUIRE(!f->counters || (f->mean && f->invstd && f->dgamma && f->dbeta && f->k0k1),
                 "conv_gemm_bnbwd: in-kernel finalize (counters) needs mean/invstd/dgamma/dbeta/k0k1");
    NVAE_REQUIRE(g->Cout % 8 == 0 && f->x_ld % 8 == 0 && f->x_ld >= g->Cout && aligned16(f->x),
                 "conv_gemm_bnbwd: Cout=%d / x_ld=%d must be multiples of 8 and x 16-B aligned", g->Cout, f->x_ld);
    NVAE_REQUIRE(f->act == ACT_NONE || f->act == ACT_SWISH, "conv_gemm_bnbwd: act %d unsupported", f->act);
    ConvBnBwd be{};
    be.x = f->x; be.x_ld = f->x_ld; be.act = f->act; 
    be.scale = f->scale; be.shift = f->shift; be.partials = f->partials;
    be.fin.counter = f->counters;
    be.fin.inv_n = 1.0f / (float)((long)g->B * g->Hout * g->Wout);
    be.fin.scale = (float*)f->scale; be.fin.mean = (float*)f->mean; be.fin.invstd = (float*)f->invstd;
    be.fin.dgamma = f->dgamma; be.fin.dbeta = f->dbeta; be.fin.k0k1 = f->k0k1; be.fin.frozen = f->frozen;
    int rc = 0;
    DISPATCH_T(dtype, rc = launch_conv_gemm<T>(g, src, wT, w_ld, bias, residual, out, 0, nullptr, (hipStream_t)stream, &be);)
    NVAE_REQUIRE(rc == 0, "conv_gemm_bnbwd: output rows are not 16-B aligned (out_ld=%d res_ld=%d)", g->out_ld, g->res_ld);
    NVAE_LAUNCH_CHECK("conv_gemm_bnbwd");
    return NVAE_OK;
}
