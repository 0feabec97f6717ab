/* nvae_hip.h -- C ABI of libnvae_hip.so: the gfx950 (MI355X) kernels behind the NVAE hot path.
 *
 * The reference (stevensdavid/nvae-tf) has no FFI: its hot path is Keras layers executed by the
 * TensorFlow runtime.  Each entry point below therefore cites the reference Python call site whose
 * arithmetic it replaces (file:line under the reference root).  INTEGRATION.md shows the ctypes
 * binding a maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless stated otherwise;
 *  - activations are NHWC, contiguous in C, element type `dtype` (NVAE_F32 / NVAE_BF16 / NVAE_F16);
 *    statistics, losses, master weights, gradients of weights and optimizer slots are always f32;
 *  - every call enqueues on `stream` (a hipStream_t passed as void*) and never synchronises,
 *    allocates or frees, so a sequence of calls can be captured into a hipGraph;
 *  - buffers documented "zeroed" must be zero when the call is enqueued (they are accumulated
 *    into with atomics);
 *  - return value: NVAE_OK, or an error code with a message available from nvae_last_error().
 *    Inputs are borrowed, outputs are caller-allocated.  Entry points are thread-safe (no global
 *    mutable state except the thread-local error string).
 */
#ifndef NVAE_HIP_H
#define NVAE_HIP_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NVAE_OK 0
#define NVAE_EINVAL 1   /* bad argument / unsupported shape */
#define NVAE_ELAUNCH 2  /* HIP launch failure */

#define NVAE_F32 0
#define NVAE_BF16 1
#define NVAE_F16 2      /* IEEE half activations, f32 accumulation / statistics / losses (BASELINE.json configs[4]) */

#define NVAE_ACT_NONE 0
#define NVAE_ACT_SWISH 1
#define NVAE_ACT_ELU 2

#define NVAE_OP_AFFINE 0 /* y = a*x + b   (preprocess.py:39, `2 * inputs - 1`) */
#define NVAE_OP_SWISH 1  /* activations.swish (preprocess.py:66) */
#define NVAE_OP_ELU 2    /* layers.ELU (common.py:54, encoder.py:60-65, postprocess.py:27) */

/* Bumped whenever an entry point is added or a signature changes; the Python binding (nvae_tf_amd/_lib.py
 * ABI_VERSION) refuses to load a library that reports another value. */
#define NVAE_ABI_VERSION 6

/* Deterministic mode (NVAE_DETERMINISTIC=1 in the Python host, read by _lib.load()).  With on != 0 every sum across
 * workgroups has ONE adder per address or a fixed order, so the bits a step produces do not depend on workgroup scheduling:
 * one statistics-slab row per producing workgroup (the *_rows queries return the larger counts: call this BEFORE sizing
 * slabs), weight-gradient pixel splits combined through slabs or not split, no halo weight-gradient kernel, an ordered
 * BatchNorm-regulariser sum.  Slower (ordered sums of up to 1 024 slab rows; unsplit depthwise / small weight gradients). */
int nvae_set_deterministic(int on);
int nvae_get_deterministic(void);

const char* nvae_last_error(void);
int nvae_abi_version(void);

/* Geometry of one convolution-shaped gather GEMM.
 *   out[b, ho, wo, n] = sum_{kh,kw,c} src[b, hs, ws, c] * w[(kh,kw,c), n]
 *   hc = ho*stride - pad_t + kh;  valid iff 0 <= hc < Hin*div and (!exact || hc % div == 0);
 *   hs = hc / div                                   (same for the w axis)
 * Forward conv:       div = nearest-upsample factor (1 or 2), exact = 0, stride = conv stride.
 * Data gradient:      src = dy, stride = 1, div = forward stride, exact = 1, pad = K-1-pad_fwd,
 *                     weights tap-flipped and transposed (see nvae_weight_prep).
 * TF 'same' padding is expressed by the caller through pad_t/pad_l (may be negative).         */
typedef struct NvaeConvGeom {
    int B, Hin, Win, Cin;    /* source tensor; Cin = channels contracted                       */
    int Hout, Wout, Cout;    /* output tensor                                                   */
    int KH, KW;
    int stride, pad_t, pad_l;
    int div, exact;
    int in_ld, out_ld, res_ld; /* pixel strides (elements) of src / out / residual             */
} NvaeConvGeom;

/* ---- dense convolutions: Conv2D call sites preprocess.py:19-23,92-99 encoder.py:92-98,12,47-57,61
 *      decoder.py:110-112,126-134 postprocess.py:29,78-82,96-105 common.py:41-62,150-163 ------ */
/* MFMA implicit GEMM.  wT: [Cout][w_ld] (k contiguous, k = (kh*KW+kw)*Cin + c), element type
 * `dtype`.  bias (f32, may be NULL), residual (dtype, may be NULL; may alias out = accumulate).
 * Requires Cin, in_ld, w_ld multiples of 8 (bf16) / 4 (f32) and 16-B aligned src/wT.            */
/* stats (f32, may be NULL, ZEROED): [nvae_conv_gemm_stats_rows(g)][2][Cout] column sums and sums of squares
 * of the output (bias included) - the BatchNorm statistics slab of the layer that follows, consumed by
 * nvae_bn_finalize_s / nvae_bn_apply_fin.  M-tile i ADDS (f32 atomics) into row i % rows: at most 64 adders
 * per address, so nobody waits and a consumer sums 1-8 rows (tools/mb_atomic.hip).                  */
int nvae_conv_gemm_stats_rows(int dtype, const NvaeConvGeom* g);
int nvae_conv_gemm(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                   const float* bias, const void* residual, void* out, int out_f32, float* stats,
                   void* stream);
/* nvae_conv_gemm for the conv -> BN -> act -> conv chains of every residual cell (encoder.py:91-103,
 * decoder.py:125-136, preprocess.py:84-99, postprocess.py:71-108), so that each BatchNorm costs no launch:
 *  pre (may be NULL): `src` is the INPUT of a BatchNorm(+Swish) (NvaeBnIn: final coefficient table, or
 *    the accumulated statistics slab, which the kernel then finishes itself); the kernel applies
 *    act(scale*x + shift) to the gathered operand inside LDS (zero padding stays zero).  act_out (may be NULL, needs a stride-1 'same' geometry): the activated tensor is also written
 *    out, [pixels][act_ld], for the weight gradient of this conv (nvae_conv_wgrad*).  Needs div == 1 and
 *    Cin <= nvae_conv_gemm_pre_max_cin(dtype, g).
 *  fin (may be NULL, needs stats): the BatchNorm that FOLLOWS this conv is finalized in-kernel from the
 *    statistics slab by the last M-tile of every N-tile column to arrive (NvaeBnFin, declared below with the
 *    SE kernels; counter: >= nvae N-tiles ints, zero at rest).                                          */
/* A BatchNorm applied by its consumer.  Either the coefficient table is final (slab == NULL; scale, shift
 * are inputs), or the consumer finishes the statistics itself: slab [rows][2][C] as accumulated by the
 * producer (rows = 1-8), and scale / shift / mean / invstd (for the backward pass) and the moving statistics
 * rm / rv are OUTPUTS written by one workgroup of the consuming kernel.                                 */
typedef struct NvaeBnIn {
    const float* slab; int rows;
    float momentum, eps;
    const float* gamma; const float* beta;
    float* rm; float* rv;
    float* scale; float* shift; float* mean; float* invstd;
} NvaeBnIn;
typedef struct NvaeConvPre {
    NvaeBnIn bn;                /* the BatchNorm in front of the conv (over the B*Hin*Win source pixels) */
    int act;
    void* act_out; int act_ld;
} NvaeConvPre;
/* In-kernel BatchNorm finalize ("the last workgroup to arrive turns the statistics slab into coefficients",
 * csrc/bn_fin.h): what BatchNormalization(momentum, epsilon) needs besides the slab.                  */
typedef struct NvaeBnFin {
    int* counter;               /* zero at rest (reset by the kernel); one int per 64-column tile group */
    const float* gamma; const float* beta;
    float* rm; float* rv;       /* moving statistics, updated with Keras momentum semantics */
    float momentum, eps;
    float* scale; float* shift; float* mean; float* invstd;     /* [C] each: outputs */
} NvaeBnFin;
int nvae_conv_gemm_ex(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                      const float* bias, const void* residual, void* out, int out_f32, float* stats,
                      const NvaeConvPre* pre, const NvaeBnFin* fin, void* stream);
int nvae_conv_gemm_pre_max_cin(int dtype, const NvaeConvGeom* g);
/* Tuning hook (tools/tune_conv.py): 0 = the launcher's own tile choice (default), 1..7 = force one tile family of
 * the generic implicit-GEMM kernel for plain launches.  Process-wide; not for production use.       */
int nvae_conv_gemm_force_tile(int tile);
/* Split-K workspace of the implicit-GEMM kernel.  The small-M layers with a long K loop (the 3x3 convs of
 * EncodingResidualCell at 4x4 / 8x8, encoder.py:92-98, and the K = 6*C 1x1 convs of GenerativeResidualCell,
 * decoder.py:126-134, forward and data gradient) are launched as S workgroups per output tile, each over 1/S of K:
 * the f32 partial tiles go through `slab` (write-through stores), the last workgroup to arrive at the tile's
 * counter sums them in slice order (bit-identical whatever the arrival order) and runs the usual epilogue.
 *   slab: >= bytes of device memory, 16-B aligned; counters: n_counters ints, ZEROED once by the caller (the
 *   kernels leave them zero).  Launches that share a workspace must be stream-ordered (one workspace per stream
 *   that issues convolutions).  NULL / NULL: no launch is split.  Process-wide setting.                     */
int nvae_conv_set_workspace(void* slab, size_t bytes, int* counters, int n_counters);
/* Tuning hook (tools/mb_smallconv.py): 0 = the launcher's own choice, S >= 1 = force S K-slices per tile. */
int nvae_conv_gemm_force_split(int S);
/* 1 if nvae_conv_gemm* runs this geometry on the whole-image 3x3 kernel (k_conv_img: 16-bit activations, 3x3 'same'
 * stride-1 convs on 4x4 / 8x8 images with 128 or 256 input channels - EncodingResidualCell at the two latent scales,
 * encoder.py:92-98, forward and data gradient).  Its NvaeConvPre prologue transforms the staged tile once per
 * workgroup, so the host uses the prologue for these convs even behind a Swish.                              */
int nvae_conv_img_ok(int dtype, const NvaeConvGeom* g);
/* Tuning / test hook: 0 = never select the whole-image kernel (the generic implicit GEMM runs instead). */
int nvae_conv_img_enable(int on);
/* which kernel nvae_conv_gemm* picks for a geometry: -1 whole-image kernel, -2 halo kernel, else k_conv_gemm2's tile family
 * (1: 256x192, 2: 128x192, 3: 128x128, 4: 128x64, 5: 32x64, 6 / 7: 64x64, 14: 128x128 with two workgroups per CU) */
int nvae_conv_gemm_family(int dtype, const NvaeConvGeom* g);
/* 16-bit dense 5x5 halo kernel: 1 = four waves of 128 x 96 with a software-pipelined loop, 0 = eight ping-pong waves of
 * 64 x 96.  Results are bit-identical (same accumulation order per output element). */
int nvae_conv_halo4_enable(int on);
/* Diagnostics of the halo kernel (tools/mb_halo.py).  nvae_conv_halo4_enable(form | 16): the kernel stamps s_memtime /
 * s_memrealtime around its main loop and returns WITHOUT its epilogue; stamps = [tile][2] {d memtime, d memrealtime} (clock =
 * ratio x 100 MHz).  nvae_conv_halo4_enable(form | 32): complete launches; stamps = [tile][4] absolute 100 MHz times at
 * entry, loop start, loop end, exit.  This copies the first n values to the host (n <= 2048). */
int nvae_conv_halo_stamps(unsigned long long* host_out, int n);
/* nvae_conv_gemm used as the DATA GRADIENT of a conv whose input was y = act(BN(x)) (the BNSwishConv /
 * ConvBNSwish pairs, encoder.py:91-98, decoder.py:125-135, postprocess.py:84-107): `out` receives dy
 * as usual and the epilogue additionally reduces dpre = dy * act'(scale*x + shift) against the same
 * tile of x: partials[mtiles][2][Cout] <- (sum dpre, sum dpre*x), and the last M-tile of each column
 * of tiles finishes dgamma += / dbeta += / k0k1 exactly as nvae_bn_bwd_reduce + nvae_bn_bwd_finalize
 * would, so only nvae_bn_bwd_apply remains for that layer.  Requires 16-B aligned output rows and
 * Cout % 8 == 0.  counters: >= ceil(Cout/64) ints, zero at rest (shared with nvae_bn_stats_fin); with
 * counters == NULL the kernel only writes the slab and the caller runs nvae_bn_bwd_finalize_s with
 * S = nvae_conv_gemm_stats_rows (the faster arrangement: workgroups retire without waiting); the
 * partials slab must be ZEROED (the M-tiles add into it).          */
typedef struct NvaeBnBwdFuse {
    const void* x;          /* BN input [B*Hout*Wout, x_ld], activation dtype */
    int x_ld, act, frozen;
    const float* scale; const float* shift; const float* mean; const float* invstd;
    float* partials; int* counters;
    float* dgamma; float* dbeta; float* k0k1;
} NvaeBnBwdFuse;
int nvae_conv_gemm_bnbwd(int dtype, const NvaeConvGeom* g, const void* src, const void* wT, int w_ld,
                         const float* bias, const void* residual, void* out, const NvaeBnBwdFuse* f,
                         void* stream);
/* Weight gradient: dw[k, n] += sum_m gather(x)[m, k] * dy[m, n]  (f32 atomics, dw zeroed or
 * holding a partial sum).  g describes the FORWARD conv; dy has pixel stride g->out_ld.
 * dw_ld = row stride of dw in floats.  db (may be NULL): db[n] += sum_m dy[m, n].
 * scratch (f32, uninitialised, may be NULL): when it holds nvae_conv_wgrad_scratch(...) floats the
 * pixel reduction is split over up to 256 workgroups per tile and combined through it instead of
 * through atomics.                                                                             */
long nvae_conv_wgrad_scratch(int dtype, const NvaeConvGeom* g);
/* per-layer scratch of an n-layer nvae_conv_wgrad_batched launch (the pixel split shrinks with n: a batch of
 * same-shape layers fills the chip by itself, so from a few layers on no scratch is needed at all)  */
long nvae_conv_wgrad_scratch_n(int dtype, const NvaeConvGeom* g, int n_layers);
int nvae_conv_wgrad(int dtype, const NvaeConvGeom* g, const void* x, const void* dy, float* dw,
                    int dw_ld, float* db, float* scratch, long scratch_floats, void* stream);
/* nvae_conv_wgrad for n <= 32 layers of the SAME geometry in one launch (the residual towers repeat one
 * conv shape 10-40 times per step; each of those weight gradients alone is a 10-15 us kernel).  x, dy, dw,
 * db: host arrays of n device pointers (db: NULL, or a pointer for every layer); scratch: n times
 * nvae_conv_wgrad_scratch_n(dtype, g, n) floats (scratch_floats_per_layer each).                     */
int nvae_conv_wgrad_batched(int dtype, const NvaeConvGeom* g, int n, const void* const* x,
                            const void* const* dy, float* const* dw, int dw_ld, float* const* db,
                            float* scratch, long scratch_floats_per_layer, void* stream);
/* Scalar fallback for shapes the MFMA path does not take (Cin = 1, 20; Cout = 1).  w is the f32
 * master [KH, KW, *, *] addressed as w[tap*ws_tap + c*ws_c + n*ws_n] (tap order flipped if
 * flip != 0), so the same kernel serves forward and data-gradient.                              */
int nvae_conv_direct(int dtype, const NvaeConvGeom* g, const void* src, const float* w, long ws_tap,
                     long ws_c, long ws_n, int flip, const float* bias, const void* residual,
                     void* out, int out_f32, void* stream);
/* dw[(tap*Cin + c)*dw_ld + n] += ..., db[n] += sum_m dy[m, n] (db may be NULL).                 */
int nvae_conv_direct_wgrad(int dtype, const NvaeConvGeom* g, const void* x, const void* dy,
                           float* dw, int dw_ld, float* db, void* stream);
/* out[c] += sum_r x[r*ld + c]  (bias gradients).  C multiple of 8.                              */
int nvae_colsum(int dtype, const void* x, long rows, int C, int ld, float* out, void* stream);

/* ---- DepthwiseConv2D((5,5), 'same'), decoder.py:130.  w f32 [5,5,C], bias f32 or NULL.
 *      flip=1 gives the data gradient.                                                         */
int nvae_dwconv5(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H,
                 int W, int C, int flip, int accumulate, void* stream);
/* forward pass that also emits the BatchNorm statistics of its output (decoder.py:130-131: the
 * depthwise conv feeds BN3): stats[rows][2][C] (ZEROED; the workgroups add into it with atomics, <= 64
 * adders per address), rows = nvae_dwconv5_stats_rows(...) (0 = unsupported for this dtype: use
 * nvae_bn_stats), consumed by nvae_bn_finalize_s / nvae_bn_apply_fin.                              */
int nvae_dwconv5_stats_rows(int dtype, int B, int H, int W, int C);
int nvae_dwconv5_stats(int dtype, const void* x, const float* w, const float* bias, void* y, int B, int H,
                       int W, int C, float* stats, void* stream);
int nvae_dwconv5_wgrad(int dtype, const void* x, const void* dy, float* dw, float* db, int B, int H,
                       int W, int C, void* stream);
/* The same two passes on act(BN(x)) of the RAW tensor x (decoder.py:125-131: BatchNormalization -> swish ->
 * DepthwiseConv2D), 16-bit activation types: the BatchNorm is applied to the halo tile in LDS and the normalised
 * activation is never materialised.  Forward: bn = the final table or the producer's statistics slab (as for
 * nvae_se_fused_fwd); weight gradient: the final scale / shift table.                                        */
int nvae_dwconv5_pre(int dtype, const void* x, const NvaeBnIn* bn, int act, const float* w, const float* bias,
                     void* y, int B, int H, int W, int C, float* stats /* NULL or output slab */, void* stream);
/* Data gradient of the depthwise conv that also reduces the backward sums of the BatchNorm(+act) in front of the conv (x0 = that
 * BatchNorm's input, scale / shift = its final coefficient table): partials[nvae_dwconv5_stats_rows(...)][2][C], zeroed by the
 * caller, consumed by nvae_bn_bwd_apply_fin.  16-bit activation types only (decoder.py:128-131 backward). */
int nvae_dwconv5_bnbwd(int dtype, const void* dy, const float* w, void* dx, int B, int H, int W, int C, const void* x0,
                       const float* scale, const float* shift, int act, float* partials, void* stream);
int nvae_dwconv5_wgrad_pre(int dtype, const void* x, const float* scale, const float* shift, int act,
                           const void* dy, float* dw, float* db, int B, int H, int W, int C, void* stream);

/* ---- BatchNormalization(momentum=.05, eps=1e-5) (+Swish), encoder.py:91-103, decoder.py:125-146,
 *      postprocess.py:71,84,107-108, preprocess.py:88-89, common.py:148,165-166 ------------------- */
/* STATISTICS SLABS.  Every slab is [rows][2][C]: (sum x, sum x^2) for the forward statistics, (sum dpre,
 * sum dpre*x) for the backward sums.  Its ELEMENT TYPE follows the activation dtype of the call that writes or
 * reads it: float for NVAE_BF16, DOUBLE for NVAE_F32 (the parity path: E[x^2]-E[x]^2 and sum dpre*x - mean*sum dpre
 * cancel 8-9 bits where |mean| >> std, e.g. on the depthwise-conv outputs in front of decoder.py:131, and f32
 * sums - or just their run-to-run order under atomics - moved the C2 loss by up to 2.6e-3).  Slabs declared
 * `float*` below are therefore S*2*C floats or S*2*C doubles; entry points that only read a slab take the
 * dtype of its producer as their first argument.  Slabs that producers ADD into (conv / depthwise / fused-SE
 * epilogues) must be zeroed.
 * Row splits S used by the strip reductions (nvae_bn_stats, nvae_bn_bwd_reduce):                  */
int nvae_reduce_splits(long rows, int C);
/* partials[S][2][C] <- per-split (sum x, sum x^2).  No atomics, nothing needs zeroing.
 * dtype == NVAE_F32: the slab holds S*2*C DOUBLES (sums accumulated and combined in f64: E[x^2]-E[x]^2
 * of an f32 tensor with |mean| >> std - the depthwise-conv outputs in front of decoder.py:131 - otherwise
 * cancels 8-9 bits); allocate 2 * S*2*C floats.  The same holds for nvae_bn_stats_fin.             */
int nvae_bn_stats(int dtype, const void* x, long rows, int C, float* partials, void* stream);
/* mean/var from the slab nvae_bn_stats(dtype, ...) wrote; scale = gamma*invstd, shift = beta -
 * mean*scale; moving stats updated with Keras semantics: moving = moving*momentum + batch*(1-momentum). */
int nvae_bn_finalize(int dtype, const float* partials, long rows, int C, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     float* scale, float* shift, float* mean, float* invstd, void* stream);
/* as nvae_bn_finalize for a slab with an explicit number of row splits S (conv-epilogue statistics) */
int nvae_bn_finalize_s(int dtype, const float* partials, int S, long rows, int C, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, float momentum,
                       float eps, float* scale, float* shift, float* mean, float* invstd, void* stream);
/* nvae_bn_stats + nvae_bn_finalize in ONE launch: the last workgroup of each 64-channel strip sums the
 * slabs and writes the coefficients ("last arriver finalizes", csrc/bn_fin.h).  counters: at least
 * ceil(C/64) ints, zero before the first use; the kernel leaves them zero again.                  */
int nvae_bn_stats_fin(int dtype, const void* x, long rows, int C, float* partials, int* counters,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* scale, float* shift, float* mean,
                      float* invstd, void* stream);
/* inference mode: scale/shift from the moving statistics; mean/invstd (may both be NULL) receive the
 * moving mean and 1/sqrt(moving_var + eps) for a backward pass through the frozen layer.          */
int nvae_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, int C, float eps, float* scale, float* shift,
                         float* mean, float* invstd, void* stream);
int nvae_bn_apply(int dtype, const void* x, void* y, long rows, int C, const float* scale,
                  const float* shift, int act, void* stream);
/* nvae_bn_finalize_s + nvae_bn_apply in ONE launch for a statistics slab that another kernel produced
 * (conv / depthwise / SE epilogues): every workgroup sums its own 64-channel strip of the slab, the
 * first row of workgroups also publishes scale/shift/mean/invstd and updates the moving statistics.  */
int nvae_bn_apply_fin(int dtype, const void* x, void* y, long rows, int C, const float* partials, int S,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* scale, float* shift, float* mean, float* invstd,
                      int act, void* stream);
/* nvae_bn_bwd_finalize_s + nvae_bn_bwd_apply in ONE launch (slab from nvae_conv_gemm_bnbwd /
 * nvae_se_bwd_apply_bn): dgamma +=, dbeta +=, dx (+)= scale*dpre + k1*x + k0.                       */
int nvae_bn_bwd_apply_fin(int dtype, const void* x, const void* dy, void* dx, long rows, int C,
                          const float* partials, int S, const float* scale, const float* shift,
                          const float* mean, const float* invstd, float* dgamma, float* dbeta, int act,
                          int frozen, int accumulate, void* stream);
/* partials[S][2][C] <- per-split (sum dpre, sum dpre*x); dpre = dy * act'(scale*x + shift).      */
int nvae_bn_bwd_reduce(int dtype, const void* x, const void* dy, long rows, int C, const float* scale,
                       const float* shift, int act, float* partials, void* stream);
/* dgamma += sum dpre*xhat, dbeta += sum dpre; k0k1[2][C] = coefficients of
 * dx = scale*dpre + k1*x + k0 (the batch-statistics terms of the BN gradient; zero when `frozen`,
 * i.e. the layer normalised with moving statistics).                                            */
int nvae_bn_bwd_finalize(int dtype, const float* partials, long rows, int C, const float* scale,
                         const float* mean, const float* invstd, float* dgamma, float* dbeta,
                         float* k0k1, int frozen, void* stream);
/* nvae_bn_bwd_reduce + nvae_bn_bwd_finalize in ONE launch (counters as for nvae_bn_stats_fin).    */
int nvae_bn_bwd_reduce_fin(int dtype, const void* x, const void* dy, long rows, int C,
                           const float* scale, const float* shift, const float* mean,
                           const float* invstd, int act, float* partials, int* counters,
                           float* dgamma, float* dbeta, float* k0k1, int frozen, void* stream);
/* as nvae_bn_bwd_finalize for a slab with S row splits (S = nvae_conv_gemm_stats_rows for the slab that
 * nvae_conv_gemm_bnbwd leaves when it is given no counters)                                        */
int nvae_bn_bwd_finalize_s(int dtype, const float* partials, int S, long rows, int C, const float* scale,
                           const float* mean, const float* invstd, float* dgamma, float* dbeta,
                           float* k0k1, int frozen, void* stream);
int nvae_bn_bwd_apply(int dtype, const void* x, const void* dy, void* dx, long rows, int C,
                      const float* scale, const float* shift, const float* k0k1, int act,
                      int accumulate, void* stream);

/* ---- SqueezeExcitation + residual, common.py:127-142 with encoder.py:107 / decoder.py:147 /
 *      preprocess.py:107 / postprocess.py:58:   y = skip_scale*skip + branch_scale*(x * gate) ----- */
int nvae_se_pool(int dtype, const void* x, int B, int HW, int C, float* pooled_sum /*[B,C]*/,
                 void* stream);
int nvae_se_gate(const float* pooled_sum, int B, int HW, int C, int Hd, const float* w1,
                 const float* b1, const float* w2, const float* b2, float* gate, float* hidden,
                 void* stream);
/* nvae_se_pool + nvae_se_gate in ONE launch (one workgroup per image; C <= 2048) */
int nvae_se_pool_gate(int dtype, const void* x, int B, int HW, int C, int Hd, const float* w1,
                      const float* b1, const float* w2, const float* b2, float* pooled_sum, float* gate,
                      float* hidden, void* stream);
int nvae_se_apply(int dtype, const void* x, const void* skip, void* y, int B, int HW, int C,
                  const float* gate, float skip_scale, float branch_scale, void* stream);
/* nvae_se_apply that also emits the BatchNorm statistics of y (the next residual cell starts with a
 * BatchNorm): stats[nvae_reduce_splits(B*HW, C)][2][C], consumed by nvae_bn_finalize_s.            */
int nvae_se_apply_stats(int dtype, const void* x, const void* skip, void* y, int B, int HW, int C,
                        const float* gate, float skip_scale, float branch_scale, float* stats, void* stream);
int nvae_se_bwd_reduce(int dtype, const void* x, const void* dy, int B, int HW, int C,
                       float* r /*[B,C]*/, void* stream);
int nvae_se_gate_bwd(const float* r, const float* pooled_sum, const float* gate, const float* hidden,
                     int B, int HW, int C, int Hd, const float* w1, const float* w2, float branch_scale,
                     float* dw1, float* db1, float* dw2, float* db2, float* dpool,
                     float* scratch /*[B*(C+Hd)]*/, void* stream);
/* nvae_se_bwd_apply (without accumulation into dx) when x was the output of y = act(BN(xb)) with no other
 * consumer (decoder.py:135-146, postprocess.py:107-108): also reduces the BatchNorm-backward sums of that
 * layer, partials[nvae_reduce_splits(B*HW, C)][2][C], for nvae_bn_bwd_finalize_s.                  */
int nvae_se_bwd_apply_bn(int dtype, const void* dy, const float* gate, const float* dpool, void* dx,
                         void* dskip, int B, int HW, int C, float skip_scale, float branch_scale,
                         int acc_dskip, const void* xb, const float* scale, const float* shift, int act,
                         float* partials, void* stream);
/* nvae_se_bwd_reduce + nvae_se_gate_bwd (FC parameter gradients left to nvae_se_wgrad) in ONE launch    */
int nvae_se_reduce_gate_bwd(int dtype, const void* x, const void* dy, const float* gate, const float* hidden,
                            int B, int HW, int C, int Hd, const float* w1, const float* w2,
                            float branch_scale, float* dpool, float* scratch, void* stream);
/* dw1 == NULL above skips the FC parameter gradients; nvae_se_wgrad computes them later from the same
 * scratch (they do not feed the data-gradient chain: the host enqueues them on its side stream).   */
int nvae_se_wgrad(const float* pooled_sum, const float* hidden, const float* scratch, int B, int HW,
                  int C, int Hd, float* dw1, float* db1, float* dw2, float* db2, void* stream);
/* the same for n <= 32 SE layers of one shape in one launch (host arrays of n device pointers) */
int nvae_se_wgrad_batched(int n, const float* const* pooled_sum, const float* const* hidden,
                          const float* const* scratch, int B, int HW, int C, int Hd, float* const* dw1,
                          float* const* db1, float* const* dw2, float* const* db2, void* stream);
int nvae_se_bwd_apply(int dtype, const void* dy, const float* gate, const float* dpool, void* dx,
                      void* dskip, int B, int HW, int C, float skip_scale, float branch_scale,
                      int acc_dx, int acc_dskip, void* stream);

/* ---- SE + residual (+ the BatchNorm in front of it) as one launch per direction: se_fused.hip ------
 * The tail of every residual cell, y = skip_scale*skip + branch_scale * SE(xs), xs = BN(x)
 * (decoder.py:135-147 bn4 -> se -> 0.1*inputs + ., postprocess.py:84-88 + 58) or xs = x
 * (encoder.py:99-107, preprocess.py:100-107).  A workgroup owns whole images: pool, both FC layers, gate,
 * residual add and the BatchNorm statistics of y are one pass with one global round trip.
 * bn (may be NULL: no BatchNorm): see NvaeBnIn - the BatchNorm's output is never materialised.
 * C: a power of two in [8, 2048].  pooled_sum [B,C] (sum over HW of xs), gate [B,C], hidden [B,Hd] are
 * outputs (the backward pass and nvae_se_wgrad_batched read them).
 * stats (may be NULL; ZEROED): [nvae_se_fused_rows(B, HW, C)][2][C] statistics slab of y for the BatchNorm that
 * follows (accumulated with atomics, <= 64 adders per address).                                        */
int nvae_se_fused_rows(int B, int HW, int C);
/* Workspace of the image-split form of the two kernels below: with <= 64 images per launch (BASELINE.json configs[3],
 * [4]) and images of >= 32 K elements, S workgroups share an image (S * B <= 256) and hand the pooled vector / the gate
 * (backward: r / dpool) to each other through `buf` (>= (B*S*C + B*C) * 4 bytes) and per-image arrival counters
 * (n_counters >= B ints, ZEROED once by the caller; the kernels leave them zero).  Results are independent of the
 * arrival order (slice-ordered sums).  Launches sharing a workspace must be stream-ordered.  NULL / NULL: never split. */
int nvae_se_set_workspace(void* buf, size_t bytes, int* counters, int n_counters);
/* Tuning / test hook: -1 = the launcher's choice (default), 1 = never split, S >= 2 = S slices per image where legal. */
int nvae_se_force_split(int S);
int nvae_se_fused_fwd(int dtype, const void* x, const NvaeBnIn* bn, const void* skip, void* y, int B, int HW,
                      int C, int Hd, const float* w1, const float* b1, const float* w2, const float* b2,
                      float skip_scale, float branch_scale, float* pooled_sum, float* gate, float* hidden,
                      float* stats, void* stream);
/* Backward of the same block: dx (+)= d/dxs, dskip (+)= skip_scale*dy (dskip may be NULL), FC gradient
 * scratch [B*(C+Hd)] for nvae_se_wgrad_batched.  act: activation of the folded BatchNorm (none / swish).
 * partials (may be NULL; needs bn_scale and acc_dx == 0; ZEROED): dxs is final, so the BatchNorm-backward
 * sums of the folded layer are accumulated in the same pass: partials[nvae_se_fused_rows(B, HW, C)][2][C] for
 * nvae_bn_bwd_apply_fin / nvae_bn_bwd_finalize_s.                                                    */
int nvae_se_fused_bwd(int dtype, const void* x, const float* bn_scale, const float* bn_shift, int act,
                      const void* dy, const float* gate, const float* hidden, void* dx, void* dskip, int B,
                      int HW, int C, int Hd, const float* w1, const float* w2, float skip_scale,
                      float branch_scale, int acc_dx, int acc_dskip, float* scratch, float* partials,
                      void* stream);

/* ---- elementwise ------------------------------------------------------------------------- */
int nvae_unary_fwd(int dtype, int op, const void* x, void* y, long n, float a, float b, void* stream);
int nvae_unary_bwd(int dtype, int op, const void* x, const void* dy, void* dx, long n, int accumulate,
                   void* stream);
int nvae_add(int dtype, void* dst, const void* src, long n, int accumulate, void* stream);
int nvae_cast(int src_dtype, int dst_dtype, const void* src, void* dst, long n, void* stream);
/* backward of tf.image.resize(nearest, xf) (common.py:168-172): dx = sum over fxf blocks        */
int nvae_upsample_pool_bwd(int dtype, const void* dxu, void* dx, int B, int H, int W, int C, int f,
                           int accumulate, void* stream);
/* eps ~ N(0,1): Philox4x32-10 + Box-Muller, counter read from / advanced in device memory so
 * the call is graph-replayable (common.py:67 tf.random.normal).                                 */
int nvae_randn(float* out, long n, unsigned long long seed, unsigned long long* counter_dev,
               void* stream);

/* ---- Sampler.call + KL term + log q / log p, common.py:76-102, models.py:197-201,
 *      util.py:39-50, decoder.py:69-71,84-90.  enc_p / dec_p: f32 [B,HW,2L] raw conv outputs
 *      (dec_p NULL for group 0), eps f32 [B,HW,L].  z: dtype.  kl: f32 [B].  logq/logp: f32 [B]
 *      accumulated (+=) when non-NULL.  mu_sigma (f32 [4,B,HW,L], may be NULL) receives
 *      enc_mu, enc_sigma, dec_mu, dec_sigma for DistributionParams.                            */
int nvae_sampler_fwd(int dtype, const float* enc_p, const float* dec_p, const float* eps, void* z,
                     float* kl, float* logq, float* logp, float* mu_sigma, int B, int HW, int L,
                     void* stream);
/* d_enc/d_dec (dtype) = gradient wrt the raw conv outputs, given dz (dtype, may be NULL) and
 * dL/dKL[b] = hyper[NVAE_HY_BETA] * coeff[0] * inv_batch.                                       */
int nvae_sampler_bwd(int dtype, const float* enc_p, const float* dec_p, const float* eps,
                     const void* dz, const float* coeff, const float* hyper, float inv_batch,
                     void* d_enc, void* d_dec, int B, int HW, int L, void* stream);
/* nvae_sampler_bwd when dz carries a range-normalisation tag (below): the KL seed is multiplied by 2^gscales[gid] so that
 * both terms of the output share dz's exponent.  gscales NULL: plain.                                       */
int nvae_sampler_bwd_scaled(int dtype, const float* enc_p, const float* dec_p, const float* eps, const void* dz,
                            const float* coeff, const float* hyper, float inv_batch, void* d_enc, void* d_dec, int B,
                            int HW, int L, const float* gscales, int gid, void* stream);
/* ---- activation-gradient range normalisation (float16 activations on deep hierarchies: BASELINE.json configs[4]) ----
 * Gradients of a 40-group NVAE span 19 decades at initialisation, float16 holds 12: the backward pass renormalises the
 * activation gradient at latent-group boundaries ON THE DEVICE.  `scales` is a zeroed per-step float array; entry id holds
 * the cumulative log2 factor of every gradient tagged id (entry 0 stays 0 = untouched).
 *   nvae_grad_amax:    slot[0] = max(slot[0], max |g|)                         (slot ZEROED per step)
 *   nvae_grad_rescale: g *= 2^k in place, k = floor(target_log2 - log2 amax) clamped to +-40; scales[id_out] = scales[id_in] + k
 *   nvae_grad_merge:   dst = dst * 2^(m - scales[id_dst]) + src * 2^(m - scales[id_src]), m = min of the two; scales[id_out] = m
 *   nvae_grad_unscale: grads[4*off .. 4*(off + count)) *= 2^-scales[id] for each table row (off, count, id, 0), in float4 units:
 *                      parameter gradients are f32 and were written with their dy's tag                       */
int nvae_grad_amax(int dtype, const void* g, long n, float* slot, void* stream);
int nvae_grad_rescale(int dtype, void* g, long n, const float* amax, float* scales, int id_in, int id_out,
                      float target_log2, void* stream);
int nvae_grad_merge(int dtype, void* dst, const void* src, long n, float* scales, int id_dst, int id_src, int id_out,
                    void* stream);
int nvae_grad_unscale(float* grads, const int* table /*device [n_ranges][4]*/, int n_ranges, const float* scales,
                      void* stream);

/* ---- Bernoulli reconstruction, models.py:242-250: recon[b] = sum softplus(l) - x*l;
 *      crop != 0 restricts to rows/cols [2, H-2) (evaluate.py:117).  logits f32, x dtype.        */
int nvae_bernoulli_fwd(int dtype, const float* logits, const void* x, float* recon, int B, int H,
                       int W, int C, int crop, void* stream);
int nvae_bernoulli_bwd(int dtype, const float* logits, const void* x, void* dlogits, long n,
                       float inv_batch, const float* hyper /* NULL or loss scale source */, void* stream);

/* ---- discretised mixture of logistics head (csrc/dmol.hip) ------------------------------------
 * NOT in the reference (train.py:219, README.md:25-27 list the CIFAR / CelebA heads as to-do); the
 * specification is oracle/nvae_oracle.py::dmol_log_prob / dmol_sample (NVAE paper, PixelCNN++).
 * logits: f32 [B*HW, ld], ld >= 10*M, channel layout [M mixture logits | per colour: M means,
 * M log-scales (clamped at -7), M coefficient logits (tanh)].  x: f32 [B*HW, 3] in [0, 1].
 *   nll[b]  = -sum_pixels log p(x)                                    (one workgroup per image)
 *   dlogits = scale * d nll / d logits, in `dtype`, channels [10M, ld) written as zero
 *   sample: one draw per pixel from uniform noise u_mix [B*HW, M], u_pix [B*HW, 3] in (0, 1);
 *           temperature divides the mixture logits and multiplies the logistic scale.           */
int nvae_dmol_fwd(const float* logits, int ld, const float* x, float* nll, int B, int HW, int M,
                  void* stream);
int nvae_dmol_bwd(int dtype, const float* logits, int ld, const float* x, void* dlogits, int B,
                  int HW, int M, float scale, const float* hyper /* NULL or loss scale source */, void* stream);
int nvae_dmol_sample(const float* logits, int ld, const float* u_mix, const float* u_pix, float* out,
                     int B, int HW, int M, float temperature, void* stream);

/* ---- KL balancing + loss assembly, models.py:121-126, 204-222 ------------------------------ */
#define NVAE_HY_LR 0       /* lr / (1 - beta1^t)                       */
#define NVAE_HY_BETA 1     /* KL warm-up coefficient (models.py:122)   */
#define NVAE_HY_BALANCE 2  /* 1.0 if beta < 1 (models.py:123)          */
/* loss scaling (f16 activations): the kernels that seed the backward pass (nvae_bernoulli_bwd, nvae_dmol_bwd,
 * nvae_sampler_bwd, nvae_bn_absmax_bwd) multiply by hyper[NVAE_HY_LSCALE], nvae_adamax divides it out again through
 * hyper[NVAE_HY_GSCALE] and skips the step when hyper[NVAE_HY_OVERFLOW] is set; nvae_grad_guard sets that flag when
 * the gradient buffer holds a non-finite value and nvae_loss_scale_update adapts the scale (x 0.5 on overflow, x 2 after
 * NVAE_LS_GROWTH_STEPS clean steps).  A zero in LSCALE / GSCALE means 1 (no scaling): the bf16 / f32 paths never set them. */
#define NVAE_HY_GSCALE 3   /* 1 / loss scale                            */
#define NVAE_HY_LSCALE 4   /* loss scale                                */
#define NVAE_HY_GOOD 5     /* clean steps since the last scale change   */
#define NVAE_HY_OVERFLOW 6 /* 1.0 = this step's gradient is non-finite  */
#define NVAE_LS_GROWTH_STEPS 200
#define NVAE_HY_SIZE 8
#define NVAE_RES_LOSS 0
#define NVAE_RES_BN 1
#define NVAE_RES_RECON 2   /* mean_b recon     */
#define NVAE_RES_KL 3      /* mean_b beta*KL   */
#define NVAE_RES_SIZE 8
/* am[g] = mean_b |kl_all[g,b]|  (all-reduce-averaged across ranks by the caller when DP)         */
int nvae_kl_absmean(const float* kl_all, int G, int B, float* am, void* stream);
int nvae_loss_finalize(const float* kl_all, const float* am, const float* alphas, int G, int B,
                       const float* recon, const float* bn_loss, const float* hyper, float* coeff,
                       float* kl_loss, float* results, void* stream);
/* calculate_bn_loss, models.py:252-267.  table: int32 [n_layers,2] = (offset of gamma in the flat
 * parameter buffer, C).                                                                         */
int nvae_bn_absmax_fwd(const float* params, const int* table, int n_layers, float lambda,
                       float* bn_loss /*zeroed*/, int* argmax, void* stream);
/* hyper: NULL, or the hyper buffer whose NVAE_HY_LSCALE multiplies the subgradient (loss scaling) */
int nvae_bn_absmax_bwd(const float* params, float* grads, const int* table, const int* argmax,
                       int n_layers, float lambda, const float* hyper, void* stream);

/* ---- Adamax (train.py:131; Keras defaults) over a flat buffer ------------------------------ */
int nvae_adamax(float* p, const float* g, float* m, float* u, long n, const float* hyper, float beta1,
                float beta2, float eps, void* stream);
/* dynamic loss scaling (f16 path; see the hyper layout): set hyper[NVAE_HY_OVERFLOW] if g[0..n) holds a non-finite
 * value (before nvae_adamax), and adapt the scale afterwards.  Both are no-ops for the optimizer when never called. */
int nvae_grad_guard(const float* g, long n, float* hyper, void* stream);
int nvae_loss_scale_update(float* hyper, float min_scale, float max_scale, void* stream);

/* ---- SpectralNormalization (TFA) power iteration + compute-copy preparation.
 *      One descriptor per wrapped conv; all offsets are element offsets into flat buffers.      */
typedef struct NvaeConvDesc {
    long long w_off;   /* master f32 [KH,KW,Cin,Cout] in `params`                                */
    long long wf_off;  /* forward copy  [Cout][wf_ld]   in `wcopies` (dtype)                      */
    long long wd_off;  /* dgrad copy    [Cin][wd_ld], taps flipped, in `wcopies`; -1 = none       */
    int u_off;         /* u [Cout] in `sn_state`                                                 */
    int t_off;         /* scratch t [K] in `sn_scratch`                                          */
    int K, Cout, Cin, taps;
    int wf_ld, wd_ld;
    int idx;           /* position in per-matrix scratch arrays                                  */
    int blk_off;       /* first workgroup of this matrix; a workgroup owns 16 consecutive k rows  */
    long long p_off;   /* this matrix's [ceil(K/16)][Cout] block in `colpart` (nvae_sn_power_iter)   */
} NvaeConvDesc;
/* One power iteration per matrix: v = l2n(u W^T); u' = l2n(v W); sigma = v W u'^T.  Writes u' into
 * sn_state and 1/sigma into inv_sigma; W itself is rescaled by nvae_weight_prep.  Every sum is taken in a
 * fixed order (no atomics): sigma is bit-reproducible, so data-parallel replicas do not drift.
 * colpart: scratch of sum_i ceil(K_i/16)*Cout_i floats (NvaeConvDesc.p_off); w2: scratch laid out like
 * sn_state.  Neither needs zeroing.                                                                */
int nvae_sn_power_iter(float* params, const NvaeConvDesc* descs /*device*/, int n, int total_blocks,
                       float* sn_state, float* sn_scratch_t, float* colpart, float* w2,
                       float* inv_sigma /*[n]*/, void* stream);
/* W *= inv_sigma[i] in place (skipped if inv_sigma NULL) and (re)write both compute copies.      */
int nvae_weight_prep(int dtype, float* params, const NvaeConvDesc* descs, int n, int total_blocks,
                     const float* inv_sigma, void* wcopies, void* stream);

#ifdef __cplusplus
}
#endif
#endif
