"""Host-side operator layer: a minimal tape (explicit backward closures, no autograd engine) over
the C-ABI kernels.  Every arithmetic op of the hot path is a libnvae_hip.so launch; torch supplies
device buffers only.  Gradients accumulate through the kernels' own `accumulate` epilogues."""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib as L
from ._lib import call, ptr


def same_pad(in_size: int, k: int, s: int) -> Tuple[int, int]:
    """TF padding='same' (SURVEY Q6): low pad = total // 2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


class Var:
    """An NHWC activation and (lazily) its gradient.

    A Var returned by bn_act is LAZY: `raw` is the BatchNorm's input and `pre` (a LazyBN) knows the
    coefficient table; consumers that can apply the BatchNorm themselves (conv2d's operand prologue, the fused
    SE kernels) read `raw` + `pre.coef()`, everybody else reads `t`, which materialises the normalised
    activation on first use (one apply launch, as in round 1)."""
    __slots__ = ("_t", "g", "needs_grad", "stats", "uses", "bn_src", "pre", "fin", "gid")

    def __init__(self, t: torch.Tensor, needs_grad: bool = True):
        self._t = t
        self.g: Optional[torch.Tensor] = None
        self.gid = 0        # range-normalisation tag of .g (GradScale): .g = 2^scales[gid] x the (loss-scaled) gradient
        self.needs_grad = needs_grad
        self.stats = None   # (slab [S,2,C] f32, S): BN statistics partials emitted by the producing kernel
        self.fin = None     # (bn, coef [4,C]): the producing kernel also finalized THAT BatchNorm's coefficients
        self.uses = 0       # ops that consumed this activation (each will add to .g in backward)
        self.bn_src = None  # set by bn_act on its output: what a consumer's data-gradient kernel needs
        #                     to reduce the BatchNorm backward sums in its epilogue (nvae_conv_gemm_bnbwd)
        self.pre = None     # LazyBN: the value is act(scale * raw + shift)

    @property
    def t(self) -> torch.Tensor:
        return self._t if self.pre is None else self.pre.materialize()

    @t.setter
    def t(self, value: torch.Tensor):
        self._t = value

    @property
    def raw(self) -> torch.Tensor:
        return self._t

    @property
    def shape(self):
        return self._t.shape


FUSED_FIN = os.environ.get("NVAE_BN_FUSED_FIN", "1") != "0"
APPLY_FIN = os.environ.get("NVAE_BN_APPLY_FIN", "1") != "0"   # slab -> coefficients inside the apply kernels
SE_STATS = os.environ.get("NVAE_SE_STATS", "1") != "0"         # BN statistics out of the SE + residual kernel
SE_FUSED = os.environ.get("NVAE_SE_FUSED", "1") != "0"         # SE + residual (+ the BatchNorm in front) as one launch
# BatchNorm(+Swish) applied in the consuming conv's operand prologue (nvae_conv_gemm_ex): "0" never; "1" 1x1 convs behind a
# BatchNorm WITHOUT activation (+ the whole-image 3x3 kernel, CONV_PRE_IMG); "2" (default since round 3) 1x1 convs behind
# any BatchNorm - an im2col operand is transformed once per N-tile (and per tap for k > 1), which for Swish used to cost more
# VALU time than the apply pass it removes (tools/bench_pre.py); with the 4-instruction sigmoid of round 3 (common.h) the
# 1x1 case pays: 19.42 / 19.40 -> 19.39 / 19.36 ms per step in two interleaved pairs, 34 launches fewer; "all" wherever the
# geometry allows (tests)
CONV_PRE = os.environ.get("NVAE_CONV_PRE", "2")
# producers finalize the next BatchNorm in-kernel ("last arriver"): measured +5..12 us per conv launch against the
# 4.8 us of a finalize launch or the ~0 of a consumer that reads the accumulated slab itself, so off by default
CONV_PRE_IMG = os.environ.get("NVAE_CONV_PRE_IMG", "1") != "0"   # ... and 3x3 convs on the whole-image kernel (nvae_conv_img_ok)
STATS_FIN = os.environ.get("NVAE_STATS_FIN", "0") != "0"
FUSE_BN_BWD = os.environ.get("NVAE_BN_BWD_FUSE", "1") != "0"   # BN backward sums in the dgrad epilogue
WGRAD_ORDER = os.environ.get("NVAE_WGRAD_ORDER", "queue")     # queue | small_first | big_first (within one flush)
DW_PRE = os.environ.get("NVAE_DW_PRE", "1") != "0"           # BN(+Swish) in front of a depthwise conv applied in its LDS tile
DW_BNBWD = os.environ.get("NVAE_DW_BNBWD", "1") != "0"       # ... and that BatchNorm's backward sums in the data-gradient kernel
BN_BWD_SPLIT = os.environ.get("NVAE_BN_BWD_SPLIT", "0") != "0"  # unfused BN backward: reduce + self-finishing apply


GRAD_TARGET_LOG2 = float(os.environ.get("NVAE_GRAD_TARGET_LOG2", "-4"))   # renormalised activation gradients peak at 2^-4 (see GradScale)


class GradScale:
    """Device-side range normalisation of the activation gradients (float16 activations on deep hierarchies; kernels and
    rationale: csrc/elementwise.hip).  `grad_boundary` ops renormalise the gradient that crosses them and give it a new
    tag; every backward closure hands its output gradient's tag (Var.gid) to the gradients it produces; gradients with
    different tags are merged on the smaller exponent; parameter gradients are recorded with the tag of the dy they were
    computed from and divided by 2^scales[tag] after the backward pass (`unscale`).  Nothing here reads device memory
    on the host, so the whole scheme is captured into the step's hipGraphs."""

    MAX_IDS = 2048

    def __init__(self, ctx: "Ctx"):
        self.ctx = ctx
        self.scales = ctx.zeros_f32(self.MAX_IDS)        # zeroed with the per-step pool; entry 0 stays 0
        self.amax = ctx.zeros_f32(self.MAX_IDS)
        self.next_id = 1
        self.ranges: List[Tuple[int, int, int]] = []     # (flat parameter offset, numel, tag) of gradients written so far
        self.pending: List[tuple] = []                   # (var, temporary gradient, its tag): merged after the closure

    def new_id(self) -> int:
        i = self.next_id
        if i >= self.MAX_IDS:
            raise RuntimeError("GradScale: out of scale ids")
        self.next_id += 1
        return i

    def note(self, slot, gid: int):
        """A parameter gradient (params.Slot) was / will be accumulated from a dy tagged gid."""
        if slot is not None:
            self.ranges.append((slot.off, slot.numel, gid))

    def merge(self, v: "Var", src: torch.Tensor, src_id: int):
        nid = self.new_id()
        call("nvae_grad_merge", self.ctx.dt, ptr(v.g), ptr(src), src.numel(), ptr(self.scales), v.gid, src_id, nid)
        v.gid = nid

    def flush(self):
        for v, tmp, gid in self.pending:
            self.merge(v, tmp, gid)
        self.pending.clear()

    def rescale(self, g: torch.Tensor, gid: int) -> int:
        """Renormalise g in place; returns its new tag."""
        nid = self.new_id()
        slot = ptr(self.amax) + 4 * nid
        call("nvae_grad_amax", self.ctx.dt, ptr(g), g.numel(), slot)
        call("nvae_grad_rescale", self.ctx.dt, ptr(g), g.numel(), slot, ptr(self.scales), gid, nid, GRAD_TARGET_LOG2)
        return nid

    def unscale(self, lo: int = 0, hi: Optional[int] = None):
        """Divide the parameter gradients recorded so far (those inside [lo, hi) of the flat buffer) by their tags' factors
        and forget them.  Call after the weight-gradient kernels have been joined."""
        ps = self.ctx.ps
        hi = ps.grads.numel() if hi is None else hi
        keep, by_off = [], {}
        for off, n, gid in self.ranges:
            if lo <= off < hi:
                n8 = (n + 7) // 8 * 8                               # (slots are padded to 8 floats: params.ALIGN)
                assert by_off.setdefault(off, (n8, gid)) == (n8, gid), "one parameter, two gradient tags"
            else:
                keep.append((off, n, gid))
        self.ranges = keep
        merged = []
        for off in sorted(by_off):
            n, gid = by_off[off]
            if not gid:
                continue
            if merged and merged[-1][2] == gid and merged[-1][0] + merged[-1][1] == off:
                merged[-1] = (merged[-1][0], merged[-1][1] + n, gid)
            else:
                merged.append((off, n, gid))
        if not merged:
            return
        key = tuple(merged)
        cache = ps.__dict__.setdefault("_unscale_tables", {})
        if key not in cache:
            flat = [v for off, n, gid in merged for v in (off // 4, n // 4, gid, 0)]
            cache[key] = torch.tensor(flat, dtype=torch.int32, device=ps.grads.device)
        call("nvae_grad_unscale", ptr(ps.grads), ptr(cache[key]), len(merged), ptr(self.scales))


class Ctx:
    """Per-step execution context: dtype, mode, tape, and the per-step zeroed f32 scratch pool."""

    def __init__(self, ps, dtype: torch.dtype, training: bool, record: bool, side_stream=None):
        self.ps = ps
        # Weight-gradient kernels do not feed the backward critical path (dgrad -> BN -> dgrad ...)
        # and most of them fill only a fraction of the chip; with a side stream they are forked off
        # the main stream (also inside hipGraph capture) and joined at the end of backward().
        self.side = side_stream
        self.keep: List[torch.Tensor] = []
        self.deferred: List[Callable[[], None]] = []
        self.wgrad_q: List[tuple] = []      # MFMA conv weight gradients waiting to be launched in same-shape batches
        self.se_q: List[tuple] = []         # SE FC parameter gradients, likewise
        self.flush_every = int(os.environ.get("NVAE_WGRAD_FLUSH", "256"))
        # BN(+Swish) -> depthwise 5x5 without materialising the activation: the forward kernel gets 4-7 us slower and
        # saves a 10 us launch, the weight-gradient kernel (side stream) gets 7-15 us slower.  Worth it when the side
        # stream runs under the main chain (single GPU: +0.3 %), not when every backward segment joins it (data
        # parallel) - the model switches it off there.
        self.dw_pre = True
        self.dtype = dtype
        self.dt = L.dtype_code(dtype)
        self.ve = 4 if dtype == torch.float32 else 8
        self.training = training
        self.record = record
        self.tape: List[Callable[[], None]] = []
        self.dev = ps.params.device
        self.zero_pool = ps.zero_pool
        self._counters = None
        self.zero_cursor = ps.zero_reserved
        self.gs: Optional[GradScale] = None     # set by the model for float16 runs of deep hierarchies

    # ---- memory -------------------------------------------------------------------------
    def empty(self, shape, dtype=None) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.dev)

    def zeros_f32(self, n: int) -> torch.Tensor:
        """A slice of the pool that is zeroed once per step (ParamStore.begin_step)."""
        n_al = (n + 7) // 8 * 8
        if self.zero_cursor + n_al > self.zero_pool.numel():
            raise RuntimeError("zero pool exhausted: enlarge ParamStore.zero_pool")
        out = self.zero_pool[self.zero_cursor:self.zero_cursor + n]
        self.zero_cursor += n_al
        return out

    def slab_dtype(self) -> torch.dtype:
        """Element type of every BatchNorm statistics slab: f64 on the f32 activation path (include/nvae_hip.h
        "STATISTICS SLABS")."""
        return torch.float64 if self.dtype == torch.float32 else torch.float32

    def zero_slab(self, rows: int, C_: int) -> torch.Tensor:
        """A zeroed [rows, 2, C] statistics slab (producers add into it with atomics) from the per-step pool."""
        n = rows * 2 * C_
        if self.dtype == torch.float32:
            return self.zeros_f32(2 * n).view(torch.float64).view(rows, 2, C_)
        return self.zeros_f32(n).view(rows, 2, C_)

    def empty_slab(self, rows: int, C_: int) -> torch.Tensor:
        return self.empty((rows, 2, C_), self.slab_dtype())

    def counters(self) -> int:
        """Arrival counters of the fused BatchNorm finalizes (csrc/bn_fin.h): zero at rest, shared by
        all launches of the main stream (they are serialised there)."""
        if self._counters is None:
            self._counters = self.zeros_f32(256)
        return ptr(self._counters)

    def grad_of(self, v: Var, gid: int = 0) -> Tuple[torch.Tensor, int]:
        """Gradient buffer of v and whether the next writer must accumulate into it.  gid: range-normalisation tag of
        the gradient about to be written (GradScale); a tag other than the buffer's sends the writer to a temporary that
        is merged in after the current backward closure."""
        if v.g is None:
            v.g = torch.empty(v.shape, dtype=self.dtype, device=self.dev)
            v.gid = gid
            return v.g, 0
        if v.gid == gid:
            return v.g, 1
        assert self.gs is not None, "gradient tags without a GradScale"
        tmp = torch.empty(v.shape, dtype=self.dtype, device=self.dev)
        self.gs.pending.append((v, tmp, gid))
        return tmp, 0

    def side_launch(self, fn: Callable[[], None], *keep):
        """Enqueue independent gradient work after everything issued so far, off the main stream.
        `keep` are temporaries that must outlive the side-stream kernels (held until the join)."""
        if self.side is None:
            fn()
            return
        # Deferred: cross-stream edges are not free in a hipGraph, so the work is forked in batches
        # (its inputs are kept alive; nothing on the main stream overwrites them).
        self.deferred.append(fn)
        self.keep.extend(t for t in keep if t is not None)
        if len(self.deferred) + len(self.wgrad_q) + len(self.se_q) >= self.flush_every:
            self.flush_side()

    def defer_wgrad(self, gw, x_t: torch.Tensor, dy_t: torch.Tensor, dy_ptr: int, dw: int, dw_ld: int, db):
        """Queue an MFMA conv weight gradient.  Layers of one geometry (the residual towers repeat a conv shape
        10-40 times) are launched together by nvae_conv_wgrad_batched when the queue is flushed."""
        key = (tuple(getattr(gw, f) for f, _ in gw._fields_), dw_ld, db is None)
        self.wgrad_q.append((key, gw, ptr(x_t), dy_ptr, dw, db))
        self.keep.extend((x_t, dy_t))
        if self.side is None or len(self.wgrad_q) + len(self.deferred) >= self.flush_every:
            self.flush_side()

    def defer_se_wgrad(self, shape: tuple, pooled, hidden, scratch, grads: tuple):
        """Queue the FC parameter gradients of one SE layer (shape = (B, HW, C, Hd)); same-shape layers share a launch."""
        self.se_q.append((shape, ptr(pooled), ptr(hidden), ptr(scratch)) + tuple(grads))
        self.keep.extend((pooled, hidden, scratch))
        if self.side is None or len(self.wgrad_q) + len(self.se_q) + len(self.deferred) >= self.flush_every:
            self.flush_side()

    def _launch_se_wgrads(self):
        groups: Dict[tuple, list] = {}
        for item in self.se_q:
            groups.setdefault(item[0], []).append(item)
        self.se_q.clear()
        for (B, HW, Cc, Hd), items in groups.items():
            for i0 in range(0, len(items), 32):
                run = items[i0:i0 + 32]
                n = len(run)
                arr = lambda idx: (C.c_void_p * n)(*[it[idx] for it in run])
                call("nvae_se_wgrad_batched", n, arr(1), arr(2), arr(3), B, HW, Cc, Hd, arr(4), arr(5), arr(6), arr(7))

    def _launch_wgrads(self):
        self._launch_se_wgrads()
        groups: Dict[tuple, list] = {}
        for item in self.wgrad_q:
            groups.setdefault(item[0], []).append(item)
        self.wgrad_q.clear()
        order = list(groups.items())
        if WGRAD_ORDER != "queue":
            # per-launch work of a group: pixels x K x N of one layer (x layers in the batch)
            work = lambda kv: (kv[1][0][1].B * kv[1][0][1].Hout * kv[1][0][1].Wout * kv[1][0][1].KH * kv[1][0][1].KW
                               * kv[1][0][1].Cin * kv[1][0][1].Cout)
            order.sort(key=work, reverse=(WGRAD_ORDER == "big_first"))
        for key, items in order:
            gw, dw_ld = items[0][1], key[1]
            for i0 in range(0, len(items), 32):
                chunk = items[i0:i0 + 32]
                seen, run = set(), []
                for it in chunk + [None]:            # a weight that appears twice must not share a launch
                    if it is None or it[4] in seen:
                        self._launch_wgrad_batch(gw, dw_ld, run)
                        seen, run = set(), []
                    if it is not None:
                        seen.add(it[4]); run.append(it)

    def _launch_wgrad_batch(self, gw, dw_ld, run):
        n = len(run)
        if n == 0:
            return
        need = L.load().nvae_conv_wgrad_scratch_n(self.dt, C.byref(gw), n)     # the pixel split shrinks with n
        scratch = self.empty((n * need,), torch.float32) if need else None
        if scratch is not None:
            self.keep.append(scratch)
        arr = lambda idx: (C.c_void_p * n)(*[it[idx] for it in run])
        has_db = run[0][5] is not None
        call("nvae_conv_wgrad_batched", self.dt, C.byref(gw), n, arr(2), arr(3), arr(4), dw_ld,
             arr(5) if has_db else None, ptr(scratch), need)

    def flush_side(self):
        if not self.deferred and not self.wgrad_q and not self.se_q:
            return
        if self.side is None:
            self._launch_wgrads()
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self._launch_wgrads()
            for fn in self.deferred:
                fn()
        self.deferred.clear()

    def backward(self, lo: int = 0, hi: Optional[int] = None, join: bool = True, flush: bool = True,
                 pre_flush: bool = False):
        """Run the tape entries [lo, hi) in reverse and join the side stream, so that every gradient
        those entries produce is complete on the current stream.  backward() runs the whole tape;
        data-parallel steps run it in segments and all-reduce each segment's parameters meanwhile.
        join=False leaves the queued side-stream work forked (the caller continues on the side stream with
        `fork` and joins once at the end with `join_side`).  flush=False does not even launch it: the weight-gradient
        work of the range stays queued for `flush_inline` (data parallel: it becomes a graph / a side-stream batch of
        its own, so that the main chain never waits for it between segments).  pre_flush=True first forks what earlier
        calls left queued (data parallel: the PREVIOUS segment's weight gradients run under this segment's main chain)."""
        hi = len(self.tape) if hi is None else hi
        if pre_flush:
            self.flush_side()
        if self.gs is None:
            for fn in reversed(self.tape[lo:hi]):
                fn()
        else:
            for fn in reversed(self.tape[lo:hi]):
                fn()
                self.gs.flush()
        if flush:
            self.flush_side()
        if join:
            self.join_side()
        del self.tape[lo:hi]
        if not self.tape and join and flush:
            self.keep.clear()

    def flush_inline(self):
        """Launch everything queued for the side stream on the CURRENT stream (the caller has switched to the side stream,
        or is capturing this batch as a graph of its own)."""
        self._launch_wgrads()
        for fn in self.deferred:
            fn()
        self.deferred.clear()

    def fork(self, fn: Callable[[], None]):
        """Run fn on the side stream after everything issued so far on either stream."""
        if self.side is None:
            fn()
            return
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            fn()

    def join_side(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)
        if not self.tape:
            self.keep.clear()


# ------------------------------------------------------------------------------------------
# elementwise
# ------------------------------------------------------------------------------------------
def affine(ctx: Ctx, x: torch.Tensor, a: float, b: float) -> Var:
    """y = a*x + b on an input tensor (no gradient): preprocess.py:39."""
    y = ctx.empty(x.shape)
    call("nvae_unary_fwd", ctx.dt, L.OP_AFFINE, ptr(x), ptr(y), x.numel(), a, b)
    return Var(y, needs_grad=False)


def unary(ctx: Ctx, x: Var, op: int) -> Var:
    x.uses += 1
    y = Var(ctx.empty(x.t.shape), x.needs_grad)
    call("nvae_unary_fwd", ctx.dt, op, ptr(x.t), ptr(y.t), x.t.numel(), 0.0, 0.0)
    if ctx.record and x.needs_grad:
        def bwd():
            g, acc = ctx.grad_of(x, y.gid)
            call("nvae_unary_bwd", ctx.dt, op, ptr(x.t), ptr(y.g), ptr(g), x.t.numel(), acc)
        ctx.tape.append(bwd)
    return y


def grad_boundary(ctx: Ctx, x: Var) -> Var:
    """Identity in the forward pass; in the backward pass the gradient that crosses it is renormalised to a fixed
    range and retagged (GradScale).  Placed between latent groups / cells by the towers when ctx.gs is set."""
    if ctx.gs is None or not ctx.record or not x.needs_grad:
        return x
    x.uses += 1
    y = Var(x.t, True)
    y.stats, y.fin = x.stats, x.fin

    def bwd():
        if y.g is None:
            return
        nid = ctx.gs.rescale(y.g, y.gid)
        if x.g is None:
            x.g, x.gid = y.g, nid
        else:
            ctx.gs.merge(x, y.g, nid)
    ctx.tape.append(bwd)
    return y


def add_grad(ctx: Ctx, dst: Var, src_grad: torch.Tensor, gid: int = 0):
    g, acc = ctx.grad_of(dst, gid)
    call("nvae_add", ctx.dt, ptr(g), ptr(src_grad), src_grad.numel(), acc)


# ------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------
def _geom(B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad_t, pad_l, div, exact, in_ld,
          out_ld, res_ld) -> L.ConvGeom:
    return L.ConvGeom(B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad_t, pad_l, div, exact,
                      in_ld, out_ld, res_ld)


def conv2d(ctx: Ctx, x: Var, conv, *, stride: int = 1, up: int = 1, pad: Optional[Tuple[int, int]] = None,
           out_hw: Optional[Tuple[int, int]] = None, c_off: int = 0, cin: Optional[int] = None,
           bias: bool = True, out: Optional[Var] = None, out_coff: int = 0, accumulate: bool = False,
           residual: Optional[Var] = None, out_f32: bool = False, want_stats: bool = False,
           stats_bn=None) -> Var:
    """Conv2D(padding='same') (+ folded nearest upsample, + residual add).  `conv` is a
    params.ConvParam.  c_off/cin select a row slice of a 1x1 kernel (concat-free
    DecoderSampleCombiner, decoder.py:115-117); out/out_coff write a channel slice of an existing
    tensor (concat-free SkipScaler, preprocess.py:65-74).
    want_stats / stats_bn: the output feeds a BatchNorm (stats_bn: that layer, finalized in-kernel).
    A lazy x (bn_act) is normalised + activated in the kernel's operand prologue when the geometry allows."""
    ps = ctx.ps
    x.uses += 1
    if residual is not None:
        residual.uses += 1
    want_stats = want_stats or stats_bn is not None
    B, H, W, Cx = x.shape
    k = conv.k
    cin = conv.cin if cin is None else cin
    assert cin <= Cx and (c_off == 0 and cin == conv.cin or k == 1)
    Hu, Wu = H * up, W * up
    if pad is None:
        pad = (same_pad(Hu, k, stride)[0], same_pad(Wu, k, stride)[0])
    if out_hw is None:
        out_hw = (-(-Hu // stride), -(-Wu // stride))
    Ho, Wo = out_hw
    cout = conv.cout
    if out is None:
        out = Var(ctx.empty((B, Ho, Wo, cout), torch.float32 if out_f32 else None))
    Cy = out.t.shape[3]
    assert out.t.shape[:3] == (B, Ho, Wo) and out_coff + cout <= Cy
    esz_out = out.t.element_size()
    out_ptr = ptr(out.t) + out_coff * esz_out
    res_ptr, res_ld = None, Cy
    if accumulate:
        res_ptr, res_ld = out_ptr, Cy
        assert not out_f32
    elif residual is not None:
        assert residual.t.shape == (B, Ho, Wo, cout)
        res_ptr, res_ld = ptr(residual.t), cout
    bias_ptr = ptr(ps.view(conv.b)) if (bias and conv.b is not None) else None
    ve = ctx.ve
    g = _geom(B, H, W, cin, Ho, Wo, cout, k, k, stride, pad[0], pad[1], up, 0, Cx, Cy, res_ld)
    fwd_mfma = cin % ve == 0 and Cx % ve == 0 and c_off % ve == 0 and conv.wf_off >= 0
    # operand prologue: x = act(BN(raw)) is applied inside the conv kernel (and, for the weight gradient, written
    # out once by it) when the gather is not upsampled and covers every source pixel as some output's centre tap
    lib = L.load()
    pre_pays = CONV_PRE == "all" or (CONV_PRE in ("1", "2") and x.pre is not None and (
        (k == 1 and (x.pre.act == L.ACT_NONE or CONV_PRE == "2")) or
        # whole-image 3x3 kernel (4x4 / 8x8 towers): the tile is staged once, so the prologue costs 64 elements per thread
        (k == 3 and CONV_PRE_IMG and lib.nvae_conv_img_ok(ctx.dt, C.byref(g)) == 1)))
    use_pre = (pre_pays and fwd_mfma and x.pre is not None and x.pre.mat is None and up == 1 and cin == Cx and c_off == 0
               and 0 < Cx <= lib.nvae_conv_gemm_pre_max_cin(ctx.dt, C.byref(g))
               and (not ctx.record or (stride == 1 and (Ho, Wo) == (H, W) and 0 <= pad[0] < k and 0 <= pad[1] < k)))
    x_act: Optional[torch.Tensor] = None      # the activated input as the backward pass will read it
    if fwd_mfma:
        wT = ptr(ps.wcopies) + (conv.wf_off + c_off) * ps.wcopies.element_size()
        slab, fin = None, None
        if want_stats and ctx.training and out_coff == 0 and Cy == cout:       # (accumulate / residual: the epilogue counts it in)
            # the conv's epilogue emits the BatchNorm statistics of its output (consumed by bn_act)
            S = lib.nvae_conv_gemm_stats_rows(ctx.dt, C.byref(g))
            slab = ctx.zero_slab(S, cout)      # accumulated into with atomics
            out.stats = (slab, S)
            if want_fin(ctx, stats_bn):
                coef = ctx.empty((4, cout), torch.float32)
                fin = C.byref(bn_fin_struct(ctx, stats_bn, coef))
                out.fin = (stats_bn, coef)
        if use_pre:
            if ctx.record:
                x_act = ctx.empty(x.shape)
            pre = L.ConvPre(x.pre.bn_in(), x.pre.act, ptr(x_act), Cx)
            call("nvae_conv_gemm_ex", ctx.dt, C.byref(g), ptr(x.raw), wT, conv.wf_ld, bias_ptr, res_ptr, out_ptr,
                 int(out_f32), ptr(slab), C.byref(pre), fin)
        elif fin is not None:
            call("nvae_conv_gemm_ex", ctx.dt, C.byref(g), ptr(x.t), wT, conv.wf_ld, bias_ptr, res_ptr, out_ptr,
                 int(out_f32), ptr(slab), None, fin)
        else:
            call("nvae_conv_gemm", ctx.dt, C.byref(g), ptr(x.t), wT, conv.wf_ld, bias_ptr, res_ptr, out_ptr,
                 int(out_f32), ptr(slab))
    else:
        w = ptr(ps.view(conv.w)) + c_off * cout * 4
        call("nvae_conv_direct", ctx.dt, C.byref(g), ptr(x.t), w, conv.cin * cout, cout, 1, 0, bias_ptr,
             res_ptr, out_ptr, int(out_f32))
    if x_act is None and ctx.record:
        x_act = x.t

    if ctx.record:
        def bwd():
            dy = out.g
            assert dy is not None, f"no gradient reached conv {conv.name}"
            esz = dy.element_size()
            dy_ptr = ptr(dy) + out_coff * esz
            # ---- weight / bias gradient
            dw = ptr(ps.grads) + (conv.w.off + c_off * cout) * 4
            db = (ptr(ps.grads) + conv.b.off * 4) if (bias and conv.b is not None) else None
            gid = out.gid
            if ctx.gs is not None:
                ctx.gs.note(conv.w, gid)
                ctx.gs.note(conv.b if (bias and conv.b is not None) else None, gid)
            gw = _geom(B, H, W, cin, Ho, Wo, cout, k, k, stride, pad[0], pad[1], up, 0, Cx, Cy, Cy)
            w_mfma = (cin % ve == 0 and Cx % ve == 0 and cout % ve == 0 and Cy % ve == 0
                      and out_coff % ve == 0)
            if w_mfma:
                ctx.defer_wgrad(gw, x_act, dy, dy_ptr, dw, cout, db)
            else:
                ctx.side_launch(lambda: call("nvae_conv_direct_wgrad", ctx.dt, C.byref(gw), ptr(x_act), dy_ptr, dw,
                                             cout, db), dy)
            # ---- residual
            if residual is not None and residual.needs_grad:
                add_grad(ctx, residual, dy, gid)
            # ---- data gradient
            if x.needs_grad:
                if up == 1:
                    dst, acc = ctx.grad_of(x, gid)
                else:
                    dst, acc = ctx.empty((B, Hu, Wu, Cx)), 0
                    assert cin == Cx
                gd = _geom(B, Ho, Wo, cout, Hu, Wu, cin, k, k, 1, k - 1 - pad[0], k - 1 - pad[1],
                           stride, 1, Cy, Cx, Cx)
                d_mfma = (cout % ve == 0 and Cy % ve == 0 and out_coff % ve == 0 and conv.wd_off >= 0)
                resid = ptr(dst) if acc else None
                if cin != Cx and not acc:
                    # partial-channel write into a fresh buffer is not expected on this path
                    raise RuntimeError("conv2d backward: channel-sliced input must be accumulated")
                src = x.bn_src
                fuse = (FUSE_BN_BWD and d_mfma and src is not None and x.uses == 1 and up == 1 and not acc
                        and cin == Cx and c_off == 0 and Cx % 8 == 0)
                if fuse:
                    # x = act(BN(x0)) and this conv is its only consumer: reduce the BN backward sums in
                    # the epilogue of the data-gradient kernel (the BN closure then only applies)
                    wD = ptr(ps.wcopies) + conv.wd_off * ps.wcopies.element_size()
                    mt = L.load().nvae_conv_gemm_stats_rows(ctx.dt, C.byref(gd))
                    src["partials"] = ctx.zero_slab(mt, Cx)
                    src["k0k1"] = ctx.empty((2, Cx), torch.float32)
                    src["mtiles"] = mt
                    f = L.BnBwdFuse(ptr(src["x"]), Cx, src["act"], src["frozen"], src["scale"], src["shift"],
                                    src["mean"], src["invstd"], ptr(src["partials"]), None,
                                    src["dgamma"], src["dbeta"], ptr(src["k0k1"]))
                    call("nvae_conv_gemm_bnbwd", ctx.dt, C.byref(gd), dy_ptr, wD, conv.wd_ld, None, None,
                         ptr(dst), C.byref(f))
                    src["fused"] = True
                elif d_mfma:
                    wD = ptr(ps.wcopies) + (conv.wd_off + c_off * conv.wd_ld) * ps.wcopies.element_size()
                    call("nvae_conv_gemm", ctx.dt, C.byref(gd), dy_ptr, wD, conv.wd_ld, None, resid,
                         ptr(dst), 0, None)
                else:
                    w = ptr(ps.view(conv.w)) + c_off * cout * 4
                    call("nvae_conv_direct", ctx.dt, C.byref(gd), dy_ptr, w, conv.cin * cout, 1, cout, 1,
                         None, resid, ptr(dst), 0)
                if up != 1:
                    gx, accx = ctx.grad_of(x, gid)
                    call("nvae_upsample_pool_bwd", ctx.dt, ptr(dst), ptr(gx), B, H, W, Cx, up, accx)
        ctx.tape.append(bwd)
    return out


def dwconv5(ctx: Ctx, x: Var, dw, want_stats: bool = False) -> Var:
    """DepthwiseConv2D((5,5), padding='same') with bias, decoder.py:130.  want_stats: the output feeds a
    BatchNorm, let the kernel emit its statistics slab (bf16 path)."""
    ps = ctx.ps
    x.uses += 1
    B, H, W, Cc = x.shape
    # a lazy BatchNorm(+Swish) input is applied inside the 16-bit ring kernels (to the halo tile in LDS, and again by
    # the weight-gradient kernel): the normalised activation is never written
    lazy_in = DW_PRE and ctx.dw_pre and ctx.dtype != torch.float32 and x.pre is not None and x.pre.mat is None
    y = Var(ctx.empty(x.shape))
    rows = L.load().nvae_dwconv5_stats_rows(ctx.dt, B, H, W, Cc) if (want_stats and ctx.training) else 0
    slab = None
    if rows > 0:
        slab = ctx.zero_slab(rows, Cc)       # accumulated into with atomics
        y.stats = (slab, rows)
    if lazy_in:
        pre = x.pre
        xt = x.raw
        call("nvae_dwconv5_pre", ctx.dt, ptr(xt), C.byref(pre.bn_in()), pre.act, ptr(ps.view(dw.w)), ptr(ps.view(dw.b)),
             ptr(y.t), B, H, W, Cc, ptr(slab))
    else:
        xt = x.t
        if rows > 0:
            call("nvae_dwconv5_stats", ctx.dt, ptr(xt), ptr(ps.view(dw.w)), ptr(ps.view(dw.b)), ptr(y.t), B, H, W, Cc,
                 ptr(slab))
        else:
            call("nvae_dwconv5", ctx.dt, ptr(xt), ptr(ps.view(dw.w)), ptr(ps.view(dw.b)), ptr(y.t), B, H, W, Cc, 0, 0)
    if ctx.record:
        def bwd():
            if ctx.gs is not None:
                ctx.gs.note(dw.w, y.gid); ctx.gs.note(dw.b, y.gid)
            if lazy_in:
                ctx.side_launch(lambda: call("nvae_dwconv5_wgrad_pre", ctx.dt, ptr(xt), pre.scale, pre.shift, pre.act,
                                             ptr(y.g), ptr(ps.grads) + dw.w.off * 4, ptr(ps.grads) + dw.b.off * 4,
                                             B, H, W, Cc), y.g)
            else:
                ctx.side_launch(lambda: call("nvae_dwconv5_wgrad", ctx.dt, ptr(xt), ptr(y.g), ptr(ps.grads) + dw.w.off * 4,
                                             ptr(ps.grads) + dw.b.off * 4, B, H, W, Cc), y.g)
            g, acc = ctx.grad_of(x, y.gid)
            src = x.bn_src
            if (FUSE_BN_BWD and DW_BNBWD and APPLY_FIN and src is not None and x.uses == 1 and not acc
                    and ctx.dtype != torch.float32):
                # x = act(BN(x0)) and this conv is its only consumer: the data-gradient kernel also reduces the BatchNorm's
                # backward sums (as the implicit GEMM does in its epilogue); the BN closure then only applies
                rows_b = L.load().nvae_dwconv5_stats_rows(ctx.dt, B, H, W, Cc)
                src["partials"] = ctx.zero_slab(rows_b, Cc)
                src["mtiles"] = rows_b
                call("nvae_dwconv5_bnbwd", ctx.dt, ptr(y.g), ptr(ps.view(dw.w)), ptr(g), B, H, W, Cc, ptr(src["x"]),
                     src["scale"], src["shift"], src["act"], ptr(src["partials"]))
                src["fused"] = True
            else:
                call("nvae_dwconv5", ctx.dt, ptr(y.g), ptr(ps.view(dw.w)), None, ptr(g), B, H, W, Cc, 1, acc)
        ctx.tape.append(bwd)
    return y


# ------------------------------------------------------------------------------------------
# BatchNorm (+ Swish)
# ------------------------------------------------------------------------------------------
BN_MOMENTUM = 0.05   # Keras semantics: fraction of the OLD moving statistic kept
BN_EPS = 1e-5


def _stats_slab_dtype(ctx: Ctx) -> torch.dtype:
    """Element type of the slab nvae_bn_stats / nvae_bn_stats_fin write: f64 on the f32 activation path."""
    return torch.float64 if ctx.dtype == torch.float32 else torch.float32


def bn_fin_struct(ctx: Ctx, bn, coef: torch.Tensor) -> "L.BnFin":
    """NvaeBnFin for a producer kernel that finalizes BatchNorm `bn` in-kernel into `coef` [4, C]."""
    ps = ctx.ps
    Cc = bn.c
    base = ptr(coef)
    return L.BnFin(ctx.counters(), ptr(ps.view(bn.gamma)), ptr(ps.view(bn.beta)), ptr(ps.sview(bn.rm)),
                   ptr(ps.sview(bn.rv)), BN_MOMENTUM, BN_EPS, base, base + Cc * 4, base + 2 * Cc * 4, base + 3 * Cc * 4)


def want_fin(ctx: Ctx, stats_bn) -> bool:
    return stats_bn is not None and ctx.training and STATS_FIN


class LazyBN:
    """A BatchNorm(+act) whose application is left to its consumer.  State: where the coefficient table stands
    (`ready`), and the materialised activation if somebody asked for it."""

    def __init__(self, ctx: Ctx, x: Var, bn, act: int, coef: torch.Tensor, ready: bool):
        self.ctx, self.x, self.bn, self.act, self.coef_t, self.ready = ctx, x, bn, act, coef, ready
        Cc = bn.c
        base = ptr(coef)
        self.scale, self.shift, self.mean, self.invstd = (base + i * Cc * 4 for i in range(4))
        self.mat: Optional[torch.Tensor] = None

    def _bn_args(self):
        ps, bn = self.ctx.ps, self.bn
        return ptr(ps.view(bn.gamma)), ptr(ps.view(bn.beta)), ptr(ps.sview(bn.rm)), ptr(ps.sview(bn.rv))

    def coef(self) -> Tuple[int, int]:
        """Device pointers (scale, shift) of the FINAL coefficient table (one finalize launch if the producer left
        only a statistics slab)."""
        if not self.ready:
            slab, Sx = self.x.stats
            B, H, W, Cc = self.x.shape
            call("nvae_bn_finalize_s", self.ctx.dt, ptr(slab), Sx, B * H * W, Cc, *self._bn_args(), BN_MOMENTUM, BN_EPS, self.scale,
                 self.shift, self.mean, self.invstd)
            self.ready = True
        return self.scale, self.shift

    def bn_in(self) -> "L.BnIn":
        """NvaeBnIn for a consumer kernel that applies this BatchNorm itself: the final table, or (first consumer
        after a slab-only producer) the slab, which that kernel turns into the table."""
        g, b, rm, rv = self._bn_args()
        if self.ready:
            return L.BnIn(None, 0, BN_MOMENTUM, BN_EPS, g, b, rm, rv, self.scale, self.shift, self.mean, self.invstd)
        slab, Sx = self.x.stats
        self.ready = True          # the consumer launched next publishes the table
        return L.BnIn(ptr(slab), Sx, BN_MOMENTUM, BN_EPS, g, b, rm, rv, self.scale, self.shift, self.mean, self.invstd)

    def materialize(self) -> torch.Tensor:
        if self.mat is None:
            ctx, x = self.ctx, self.x
            B, H, W, Cc = x.shape
            rows = B * H * W
            self.mat = ctx.empty(x.shape)
            if not self.ready and APPLY_FIN:
                # statistics slab from the producing kernel: finalize + apply in one launch
                slab, Sx = x.stats
                call("nvae_bn_apply_fin", ctx.dt, ptr(x.t), ptr(self.mat), rows, Cc, ptr(slab), Sx, *self._bn_args(),
                     BN_MOMENTUM, BN_EPS, self.scale, self.shift, self.mean, self.invstd, self.act)
                self.ready = True
            else:
                sc, sh = self.coef()
                call("nvae_bn_apply", ctx.dt, ptr(x.t), ptr(self.mat), rows, Cc, sc, sh, self.act)
        return self.mat


def bn_act(ctx: Ctx, x: Var, bn, act: int = L.ACT_NONE, lazy: bool = True) -> Var:
    """BatchNormalization(momentum=0.05, epsilon=1e-5) (+ Swish).  Returns a LAZY Var (see Var): this call only
    makes sure the statistics exist; whether the normalised activation is ever written to memory is the
    consumer's decision.  lazy=False materialises at once."""
    ps = ctx.ps
    x.uses += 1
    B, H, W, Cc = x.shape
    rows = B * H * W
    gamma, beta = ptr(ps.view(bn.gamma)), ptr(ps.view(bn.beta))
    rm, rv = ptr(ps.sview(bn.rm)), ptr(ps.sview(bn.rv))
    S = L.load().nvae_reduce_splits(rows, Cc)
    if ctx.training and x.fin is not None and x.fin[0] is bn:
        coef, ready = x.fin[1], True                    # finalized by the kernel that produced x
    else:
        coef = ctx.empty((4, Cc), torch.float32)        # scale, shift, mean, invstd
        scale, shift, mean, invstd = (ptr(coef) + i * Cc * 4 for i in range(4))
        if ctx.training and x.stats is not None:
            ready = False                               # slab only: finalized by whoever needs the table first
        elif ctx.training:
            partials = ctx.empty_slab(S, Cc)
            if FUSED_FIN:
                call("nvae_bn_stats_fin", ctx.dt, ptr(x.t), rows, Cc, ptr(partials), ctx.counters(), gamma, beta, rm,
                     rv, BN_MOMENTUM, BN_EPS, scale, shift, mean, invstd)
            else:
                call("nvae_bn_stats", ctx.dt, ptr(x.t), rows, Cc, ptr(partials))
                call("nvae_bn_finalize", ctx.dt, ptr(partials), rows, Cc, gamma, beta, rm, rv, BN_MOMENTUM, BN_EPS,
                     scale, shift, mean, invstd)
            ready = True
        else:
            call("nvae_bn_eval_prepare", gamma, beta, rm, rv, Cc, BN_EPS, scale, shift, mean, invstd)
            ready = True
    scale, shift, mean, invstd = (ptr(coef) + i * Cc * 4 for i in range(4))
    xt = x.t                                            # the BatchNorm's input (materialised if x itself is lazy)
    y = Var(xt, x.needs_grad)
    y.pre = LazyBN(ctx, x, bn, act, coef, ready)
    if not lazy:
        y.pre.materialize()
    if ctx.record:
        frozen = 0 if ctx.training else 1      # tf_literal: backward through moving-statistics BN
        dgamma = ptr(ps.grads) + bn.gamma.off * 4
        dbeta = ptr(ps.grads) + bn.beta.off * 4
        info = dict(x=xt, act=act, frozen=frozen, scale=scale, shift=shift, mean=mean, invstd=invstd,
                    dgamma=dgamma, dbeta=dbeta, fused=False, coef=coef)
        y.bn_src = info

        def bwd():
            if ctx.gs is not None:
                ctx.gs.note(bn.gamma, y.gid); ctx.gs.note(bn.beta, y.gid)
            y.pre.coef()        # (a lazy output nobody consumed in the forward pass: finalize now)
            if info["fused"] and APPLY_FIN and x.needs_grad:
                # the sums were produced by the consumer's backward kernel: finalize + apply in one launch
                g, acc = ctx.grad_of(x, y.gid)
                call("nvae_bn_bwd_apply_fin", ctx.dt, ptr(xt), ptr(y.g), ptr(g), rows, Cc, ptr(info["partials"]),
                     info["mtiles"], scale, shift, mean, invstd, dgamma, dbeta, act, frozen, acc)
                return
            if info["fused"]:
                call("nvae_bn_bwd_finalize_s", ctx.dt, ptr(info["partials"]), info["mtiles"], rows, Cc, scale, mean, invstd,
                     dgamma, dbeta, ptr(info["k0k1"]), frozen)
                if x.needs_grad:
                    g, acc = ctx.grad_of(x, y.gid)
                    call("nvae_bn_bwd_apply", ctx.dt, ptr(xt), ptr(y.g), ptr(g), rows, Cc, scale, shift,
                         ptr(info["k0k1"]), act, acc)
                return
            part = ctx.empty_slab(S, Cc)
            if BN_BWD_SPLIT and APPLY_FIN and x.needs_grad:
                # plain strip reduce (no last-arriver hand-off) + the apply pass that finishes the slab itself
                call("nvae_bn_bwd_reduce", ctx.dt, ptr(xt), ptr(y.g), rows, Cc, scale, shift, act, ptr(part))
                g, acc = ctx.grad_of(x, y.gid)
                call("nvae_bn_bwd_apply_fin", ctx.dt, ptr(xt), ptr(y.g), ptr(g), rows, Cc, ptr(part), S, scale, shift,
                     mean, invstd, dgamma, dbeta, act, frozen, acc)
                return
            k0k1 = ctx.empty((2, Cc), torch.float32)
            if FUSED_FIN:
                call("nvae_bn_bwd_reduce_fin", ctx.dt, ptr(xt), ptr(y.g), rows, Cc, scale, shift, mean, invstd,
                     act, ptr(part), ctx.counters(), dgamma, dbeta, ptr(k0k1), frozen)
            else:
                call("nvae_bn_bwd_reduce", ctx.dt, ptr(xt), ptr(y.g), rows, Cc, scale, shift, act, ptr(part))
                call("nvae_bn_bwd_finalize", ctx.dt, ptr(part), rows, Cc, scale, mean, invstd, dgamma, dbeta, ptr(k0k1),
                     frozen)
            if x.needs_grad:
                g, acc = ctx.grad_of(x, y.gid)
                call("nvae_bn_bwd_apply", ctx.dt, ptr(xt), ptr(y.g), ptr(g), rows, Cc, scale, shift,
                     ptr(k0k1), act, acc)
            _ = coef   # keep alive
        ctx.tape.append(bwd)
    return y


# ------------------------------------------------------------------------------------------
# Squeeze-Excitation + residual
# ------------------------------------------------------------------------------------------
SE_FUSED_MIN_B = int(os.environ.get("NVAE_SE_FUSED_MIN_B", "0"))


def _se_fused_ok(Cc: int, Hd: int, B: int) -> bool:
    # a fused SE workgroup owns whole images, so the launch has at most B workgroups.  Falling back to the strip-structured
    # three-kernel path for small batches (NVAE_SE_FUSED_MIN_B) was measured on the side workloads: C1 (batch 32) 3 %
    # slower, C4 (batch 64) equal, C5 (batch 32) 2 % faster - no clear winner, so the fused kernels stay the default
    return SE_FUSED and Cc & (Cc - 1) == 0 and 8 <= Cc <= 2048 and Hd <= 128 and B >= SE_FUSED_MIN_B


def se_residual(ctx: Ctx, x: Var, se, skip: Var, skip_scale: float, branch_scale: float, stats_bn=None) -> Var:
    """y = skip_scale*skip + branch_scale*SE(x)   (SURVEY Q3 for which side carries the 0.1).  x may be a lazy
    BatchNorm output (Var.pre): the fused kernels apply its coefficients on the fly.  stats_bn: the BatchNorm
    that consumes y, if the caller knows it (finalized in-kernel)."""
    ps = ctx.ps
    x.uses += 1
    skip.uses += 1
    B, H, W, Cc = x.shape
    HW, Hd = H * W, se.hidden
    pooled = ctx.empty((B, Cc), torch.float32)
    gate = ctx.empty((B, Cc), torch.float32)
    hidden = ctx.empty((B, Hd), torch.float32)
    w1, b1, w2, b2 = (ptr(ps.view(p)) for p in (se.w1, se.b1, se.w2, se.b2))
    fused = _se_fused_ok(Cc, Hd, B)
    lazy_in = fused and x.pre is not None and x.pre.act == L.ACT_NONE and x.pre.mat is None
    if lazy_in:
        bn_in = C.byref(x.pre.bn_in())          # the kernel finishes the statistics itself if nobody has yet
        pre_scale, pre_shift, pre_act, xin_t = x.pre.scale, x.pre.shift, x.pre.act, x.raw
    else:
        bn_in, pre_scale, pre_shift, pre_act, xin_t = None, None, None, L.ACT_NONE, x.t
    y = Var(ctx.empty(x.shape))
    if fused:
        # pool + FC + gate + residual add (+ the BatchNorm in front, + the statistics of y) in ONE launch
        slab = None
        if ctx.training and SE_STATS:
            S = L.load().nvae_se_fused_rows(B, HW, Cc)
            slab = ctx.zero_slab(S, Cc)
            y.stats = (slab, S)
        call("nvae_se_fused_fwd", ctx.dt, ptr(xin_t), bn_in, ptr(skip.t), ptr(y.t), B, HW, Cc, Hd,
             w1, b1, w2, b2, skip_scale, branch_scale, ptr(pooled), ptr(gate), ptr(hidden), ptr(slab))
    else:
        if Cc <= 2048:
            call("nvae_se_pool_gate", ctx.dt, ptr(x.t), B, HW, Cc, Hd, w1, b1, w2, b2, ptr(pooled), ptr(gate), ptr(hidden))
        else:
            call("nvae_se_pool", ctx.dt, ptr(x.t), B, HW, Cc, ptr(pooled))
            call("nvae_se_gate", ptr(pooled), B, HW, Cc, Hd, w1, b1, w2, b2, ptr(gate), ptr(hidden))
        if ctx.training and SE_STATS:
            # the consumer is almost always the next cell's BatchNorm: emit its statistics slab here
            S = L.load().nvae_reduce_splits(B * HW, Cc)
            slab = ctx.empty_slab(S, Cc)
            call("nvae_se_apply_stats", ctx.dt, ptr(x.t), ptr(skip.t), ptr(y.t), B, HW, Cc, ptr(gate), skip_scale,
                 branch_scale, ptr(slab))
            y.stats = (slab, S)
        else:
            call("nvae_se_apply", ctx.dt, ptr(x.t), ptr(skip.t), ptr(y.t), B, HW, Cc, ptr(gate), skip_scale,
                 branch_scale)
    if ctx.record:
        def bwd():
            if ctx.gs is not None:
                for sl in (se.w1, se.b1, se.w2, se.b2):
                    ctx.gs.note(sl, y.gid)
            scratch = ctx.empty((B, Cc + Hd), torch.float32)
            gp = ptr(ps.grads)
            src = x.bn_src
            fuse = FUSE_BN_BWD and src is not None and x.uses == 1 and x.g is None
            if fused:
                gx, accx = ctx.grad_of(x, y.gid)
                if skip.needs_grad:
                    gs, accs = ctx.grad_of(skip, y.gid)
                    gs_ptr = ptr(gs)
                else:
                    gs_ptr, accs = None, 0
                part = None
                if fuse:
                    # x = act(BN(xb)) with this SE as its only consumer: dx is final, reduce the BN backward sums here
                    S = L.load().nvae_se_fused_rows(B, HW, Cc)
                    src["partials"] = ctx.zero_slab(S, Cc)
                    src["k0k1"] = ctx.empty((2, Cc), torch.float32)
                    src["mtiles"] = S
                    src["fused"] = True
                    part = ptr(src["partials"])
                    assert accx == 0
                if lazy_in:
                    xin, f_scale, f_shift, f_act = ptr(xin_t), pre_scale, pre_shift, pre_act
                elif fuse:
                    # materialised BatchNorm output: the sums need the BatchNorm's input and coefficients, and
                    # r = sum xs*dy is recomputed from them as well (one tensor read instead of two)
                    xin, f_scale, f_shift, f_act = ptr(src["x"]), src["scale"], src["shift"], src["act"]
                else:
                    xin, f_scale, f_shift, f_act = ptr(x.t), None, None, L.ACT_NONE
                call("nvae_se_fused_bwd", ctx.dt, xin, f_scale, f_shift, f_act, ptr(y.g), ptr(gate), ptr(hidden),
                     ptr(gx), gs_ptr, B, HW, Cc, Hd, w1, w2, skip_scale, branch_scale, accx, accs, ptr(scratch), part)
                ctx.defer_se_wgrad((B, HW, Cc, Hd), pooled, hidden, scratch,
                                   (gp + se.w1.off * 4, gp + se.b1.off * 4, gp + se.w2.off * 4, gp + se.b2.off * 4))
                return
            r = ctx.empty((B, Cc), torch.float32)
            dpool = ctx.empty((B, Cc), torch.float32)
            if Cc <= 2048:
                call("nvae_se_reduce_gate_bwd", ctx.dt, ptr(x.t), ptr(y.g), ptr(gate), ptr(hidden), B, HW, Cc, Hd,
                     w1, w2, branch_scale, ptr(dpool), ptr(scratch))
            else:
                call("nvae_se_bwd_reduce", ctx.dt, ptr(x.t), ptr(y.g), B, HW, Cc, ptr(r))
                call("nvae_se_gate_bwd", ptr(r), ptr(pooled), ptr(gate), ptr(hidden), B, HW, Cc, Hd, w1, w2,
                     branch_scale, None, None, None, None, ptr(dpool), ptr(scratch))
            # the FC parameter gradients are off the data-gradient chain: side stream, like the conv wgrads
            ctx.defer_se_wgrad((B, HW, Cc, Hd), pooled, hidden, scratch,
                               (gp + se.w1.off * 4, gp + se.b1.off * 4, gp + se.w2.off * 4, gp + se.b2.off * 4))
            gx, accx = ctx.grad_of(x, y.gid)
            if skip.needs_grad:
                gs, accs = ctx.grad_of(skip, y.gid)
                gs_ptr = ptr(gs)
            else:
                gs_ptr, accs = None, 0
            if fuse:
                # x = act(BN(xb)) with this SE as its only consumer: dx is final, reduce the BN backward sums here
                S = L.load().nvae_reduce_splits(B * HW, Cc)
                src["partials"] = ctx.empty_slab(S, Cc)
                src["k0k1"] = ctx.empty((2, Cc), torch.float32)
                src["mtiles"] = S
                call("nvae_se_bwd_apply_bn", ctx.dt, ptr(y.g), ptr(gate), ptr(dpool), ptr(gx), gs_ptr, B, HW, Cc,
                     skip_scale, branch_scale, accs, ptr(src["x"]), src["scale"], src["shift"], src["act"],
                     ptr(src["partials"]))
                src["fused"] = True
            else:
                call("nvae_se_bwd_apply", ctx.dt, ptr(y.g), ptr(gate), ptr(dpool), ptr(gx), gs_ptr, B, HW, Cc,
                     skip_scale, branch_scale, accx, accs)
        ctx.tape.append(bwd)
    return y


# ------------------------------------------------------------------------------------------
# latent hierarchy
# ------------------------------------------------------------------------------------------
def sampler(ctx: Ctx, enc_p: Var, dec_p: Optional[Var], eps: torch.Tensor, kl_out: torch.Tensor,
            coeff: torch.Tensor, hyper: torch.Tensor, inv_batch: float,
            logq: Optional[torch.Tensor] = None, logp: Optional[torch.Tensor] = None,
            mu_sigma: Optional[torch.Tensor] = None) -> Var:
    """Sampler.call + this group's KL (+ log q / log p): common.py:76-102, models.py:197-201."""
    enc_p.uses += 1
    if dec_p is not None:
        dec_p.uses += 1
    B, H, W, L2 = enc_p.t.shape
    Lc = L2 // 2
    assert enc_p.t.dtype == torch.float32 and eps.dtype == torch.float32 and eps.shape == (B, H, W, Lc)
    z = Var(ctx.empty((B, H, W, Lc)))
    dp = ptr(dec_p.t) if dec_p is not None else None
    call("nvae_sampler_fwd", ctx.dt, ptr(enc_p.t), dp, ptr(eps), ptr(z.t), ptr(kl_out), ptr(logq), ptr(logp),
         ptr(mu_sigma), B, H * W, Lc)
    if ctx.record:
        def bwd():
            enc_p.g = ctx.empty(enc_p.t.shape)
            enc_p.gid = z.gid
            d_dec = None
            if dec_p is not None:
                dec_p.g = ctx.empty(dec_p.t.shape)
                dec_p.gid = z.gid
                d_dec = ptr(dec_p.g)
            if ctx.gs is not None:      # the KL seed joins dz on dz's exponent
                call("nvae_sampler_bwd_scaled", ctx.dt, ptr(enc_p.t), dp, ptr(eps), ptr(z.g), ptr(coeff), ptr(hyper),
                     inv_batch, ptr(enc_p.g), d_dec, B, H * W, Lc, ptr(ctx.gs.scales), z.gid)
                return
            call("nvae_sampler_bwd", ctx.dt, ptr(enc_p.t), dp, ptr(eps), ptr(z.g), ptr(coeff), ptr(hyper),
                 inv_batch, ptr(enc_p.g), d_dec, B, H * W, Lc)
        ctx.tape.append(bwd)
    return z


def bernoulli_nll(ctx: Ctx, logits: Var, x: torch.Tensor, recon_out: torch.Tensor, inv_batch: float,
                  crop: bool = False, hyper: Optional[torch.Tensor] = None):
    """calculate_recon_loss, models.py:242-250.  hyper: the step's hyper buffer (its loss scale multiplies the
    backward seed), or None."""
    B, H, W, Cc = logits.t.shape
    assert logits.t.dtype == torch.float32
    call("nvae_bernoulli_fwd", ctx.dt, ptr(logits.t), ptr(x), ptr(recon_out), B, H, W, Cc, int(crop))
    if ctx.record:
        def bwd():
            logits.g = ctx.empty(logits.t.shape)
            call("nvae_bernoulli_bwd", ctx.dt, ptr(logits.t), ptr(x), ptr(logits.g), logits.t.numel(),
                 inv_batch, ptr(hyper))
        ctx.tape.append(bwd)


def dmol_nll(ctx: Ctx, logits: Var, x32: torch.Tensor, recon_out: torch.Tensor, inv_batch: float, n_mix: int,
             hyper: Optional[torch.Tensor] = None):
    """-log p(x) under the discretised mixture of logistics (oracle dmol_log_prob); x32 f32 in [0, 1]."""
    B, H, W, ld = logits.t.shape
    assert logits.t.dtype == torch.float32 and x32.dtype == torch.float32 and x32.shape == (B, H, W, 3)
    call("nvae_dmol_fwd", ptr(logits.t), ld, ptr(x32), ptr(recon_out), B, H * W, n_mix)
    if ctx.record:
        def bwd():
            logits.g = ctx.empty(logits.t.shape)
            call("nvae_dmol_bwd", ctx.dt, ptr(logits.t), ld, ptr(x32), ptr(logits.g), B, H * W, n_mix, inv_batch,
                 ptr(hyper))
        ctx.tape.append(bwd)


def dmol_sample(logits: torch.Tensor, n_mix: int, temperature: float = 1.0, u_mix=None, u_pix=None,
                generator=None) -> torch.Tensor:
    """One RGB draw in [0, 1] per pixel (oracle dmol_sample).  Noise defaults to torch's device RNG."""
    B, H, W, ld = logits.shape
    dev = logits.device
    if u_mix is None:
        u_mix = torch.empty(B, H, W, n_mix, device=dev).uniform_(1e-5, 1 - 1e-5, generator=generator)
    if u_pix is None:
        u_pix = torch.empty(B, H, W, 3, device=dev).uniform_(1e-5, 1 - 1e-5, generator=generator)
    u_mix = u_mix.to(dev, torch.float32).contiguous()
    u_pix = u_pix.to(dev, torch.float32).contiguous()
    out = torch.empty(B, H, W, 3, dtype=torch.float32, device=dev)
    call("nvae_dmol_sample", ptr(logits), ld, ptr(u_mix), ptr(u_pix), ptr(out), B, H * W, n_mix, float(temperature))
    return out


def randn(ctx: Ctx, shape, seed: int, counter: torch.Tensor) -> torch.Tensor:
    out = ctx.empty(shape, torch.float32)
    call("nvae_randn", ptr(out), out.numel(), seed, ptr(counter))
    return out
