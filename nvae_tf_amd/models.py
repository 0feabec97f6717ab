"""NVAE model, mirroring the reference's models.py (constructor arguments, call / train_step /
sample / sample_with_z / calculate_* methods), executed by libnvae_hip.so on one MI355X per process.

Semantics (SURVEY "Quirks"): `training` is explicit (Q1): train_step runs batch-statistics BN and
one spectral-norm power iteration per conv; call/sample run moving-statistics BN and no SN.  The
`steps` resume bug (Q14) is fixed by the caller (train.py)."""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops
from .common import DistributionParams
from .decoder import Decoder, DecoderSampleCombiner
from .encoder import Encoder
from .ops import Ctx, Var
from .params import ParamStore
from .postprocess import Postprocess
from .preprocess import Preprocess

ADAMAX_B1, ADAMAX_B2, ADAMAX_EPS = 0.9, 0.999, 1e-7   # Keras Adamax defaults [3P], train.py:131
# single-GPU overlap of the step's HBM-bound bookkeeping with its launch-bound compute (both default on):
# per-step scalars uploaded without a host sync: measured SLOWER (6 097-6 144 vs 6 211-6 225 images/s on one box) - with
# the host running ahead, the next step's side-stream prologue competes with the tail of the current step
CAPTURE_PRIORITY = os.environ.get("NVAE_CAPTURE_PRIORITY", "0") != "0"
ASYNC_HYPER = os.environ.get("NVAE_ASYNC_HYPER", "0") != "0"
OVERLAP_PREP = os.environ.get("NVAE_OVERLAP_PREP", "1") != "0"      # SN + weight copies of later modules on the side stream
# Adamax of a finished backward segment on the side stream: measured SLOWER (21.1-21.3 vs 20.4 ms/step) - cutting the
# backward pass into segments flushes the weight-gradient queue five times, and smaller same-shape batches cost more
# than the 0.3 ms of Adamax that gets hidden (the NVAE_WGRAD_FLUSH sweep shows the same: 32 -> 22.0, 256 -> 20.7 ms)
OVERLAP_ADAMAX = os.environ.get("NVAE_OVERLAP_ADAMAX", "0") != "0"
# data parallel: backward segments (each its own hipGraph, its gradient range all-reduced while the next one runs):
# 5 = postprocess | decoder | three encoder + preprocess pieces, 4 / 3 = fewer encoder cuts, 2 = postprocess + decoder |
# encoder + preprocess, 1 = one backward graph and one all-reduce behind it (no overlap)
DP_SEGMENTS = int(os.environ.get("NVAE_DP_SEGMENTS", "4"))
DP_DW_PRE = os.environ.get("NVAE_DP_DW_PRE", "0") != "0"      # keep the depthwise BN prologue when data parallel
# single-GPU diagnostics of the data-parallel step's overhead (tools: bench.py --force-dp): leave out one of its collectives
DP_DIAG_SKIP_AM = os.environ.get("NVAE_DP_DIAG_SKIP_AM", "0") != "0"
DP_DIAG_SKIP_GRADS = os.environ.get("NVAE_DP_DIAG_SKIP_GRADS", "0") != "0"


class NVAE:
    def __init__(self, n_encoder_channels, n_decoder_channels, res_cells_per_group, n_preprocess_blocks,
                 n_preprocess_cells, n_latent_per_group, n_latent_scales, n_groups_per_scale,
                 n_postprocess_blocks, n_post_process_cells, sr_lambda, scale_factor, total_epochs,
                 n_total_iterations, step_based_warmup, input_shape, *, device="cuda:0",
                 dtype=torch.bfloat16, seed=1, lr_decay_steps: Optional[int] = None, base_lr=1e-3,
                 head: Optional[str] = None, num_mixture_dec: int = 10, loss_scale=None):
        """Positional arguments as models.py:17-36.  input_shape = [B, H, W, C] (B ignored).
        dtype: activation type of the kernels - torch.bfloat16 (default), torch.float16 or torch.float32; master
        weights, statistics, KL / losses and optimizer state are f32 in every case.  loss_scale: factor on the
        backward seeds (dL/dlogits, dL/dKL, the BN regulariser), undone inside Adamax, for float16 activations, whose
        gradients have 5 exponent bits.  A float = static scale; "dynamic" (the default for float16) = device-side
        dynamic scaling: a step whose gradient holds a non-finite value is skipped and halves the scale, 200 clean
        steps double it (nvae_grad_guard / nvae_loss_scale_update; no host round trip, so it lives inside the
        captured graphs).  None = 1.0 for bf16 / f32.
        head: "bernoulli" (the reference's binary-MNIST likelihood, default for C == 1) or "dmol"
        (discretised mixture of `num_mixture_dec` logistics over RGB, default for C == 3; the
        reference leaves it unimplemented)."""
        assert len(n_groups_per_scale) == n_latent_scales
        self.sr_lambda = sr_lambda
        if loss_scale is None:
            loss_scale = "dynamic" if dtype == torch.float16 else 1.0
        self.dynamic_loss_scale = loss_scale == "dynamic"
        self.loss_scale = 2.0 ** -8 if self.dynamic_loss_scale else float(loss_scale)     # (initial value if dynamic)
        self.n_latent_per_group = n_latent_per_group
        self.n_latent_scales = n_latent_scales
        self.n_groups_per_scale = [int(g) for g in n_groups_per_scale]
        self.n_total_iterations = n_total_iterations
        self.n_preprocess_blocks = n_preprocess_blocks
        self.total_epochs = total_epochs
        self.step_based_warmup = step_based_warmup
        self.scale_factor = scale_factor
        self.device = torch.device(device)
        self.dtype = dtype
        self.seed = seed
        self.lr_decay_steps = lr_decay_steps or n_total_iterations
        self.base_lr = base_lr
        input_shape = [int(v) for v in input_shape]
        self.input_shape = input_shape
        L.load()   # fail loudly now if the HIP library is missing

        ps = ParamStore(seed)
        self.ps = ps
        marks = [0]             # flat-buffer offsets where preprocess / encoder / decoder / postprocess end
        cmarks = [0]            # the same boundaries as indices into ps.convs (ParamStore.prepare_weights(part=k))
        self.preprocess = Preprocess(ps, n_encoder_channels, n_preprocess_blocks, n_preprocess_cells,
                                     scale_factor, input_shape)
        mult = self.preprocess.mult
        marks.append(ps._p_cursor); cmarks.append(len(ps.convs))
        self.encoder = Encoder(ps, n_encoder_channels, n_decoder_channels, n_latent_per_group,
                               res_cells_per_group, n_latent_scales, self.n_groups_per_scale, mult,
                               scale_factor, self.preprocess.output_shape_)
        mult = self.encoder.mult
        marks.append(ps._p_cursor); cmarks.append(len(ps.convs))
        self.decoder = Decoder(ps, n_encoder_channels, n_decoder_channels, n_latent_per_group,
                               res_cells_per_group, n_latent_scales,
                               list(reversed(self.n_groups_per_scale)), mult, scale_factor,
                               self.encoder.output_shape_)
        mult = self.decoder.mult
        marks.append(ps._p_cursor); cmarks.append(len(ps.convs))
        self.head = head or ("bernoulli" if input_shape[3] == 1 else "dmol")
        assert self.head in ("bernoulli", "dmol")
        assert self.head == "bernoulli" or input_shape[3] == 3, "the mixture-of-logistics head models RGB"
        self.num_mixture_dec = int(num_mixture_dec)
        out_channels = input_shape[3] if self.head == "bernoulli" else 10 * self.num_mixture_dec
        self.postprocess = Postprocess(ps, n_postprocess_blocks, n_post_process_cells, mult,
                                       n_decoder_channels, scale_factor, out_channels=out_channels)
        marks.append(ps._p_cursor); cmarks.append(len(ps.convs))
        self.param_marks = marks
        ps.prep_marks = cmarks
        self.n_groups = self.decoder.n_groups
        # per-image activation footprint decides the scratch pool; generous fixed size
        # (deterministic mode: one statistics-slab row per producing workgroup instead of one per 64)
        ps.finalize(self.device, dtype, zero_pool_floats=(1 << 27) if L.load().nvae_get_deterministic() else (1 << 24))

        self.epoch = 0          # updated at the start of each epoch (models.py:82-83)
        self.steps = 0          # updated for each training step (models.py:86-87)
        self.opt_iterations = 0
        self.reducer = None     # parallel.GradReducer when data-parallel
        # DP: backward runs in tape segments (postprocess | decoder | encoder top | encoder middle | encoder bottom +
        # preprocess); the parameters of a finished segment are all-reduced while the next one computes
        self.overlap_allreduce = DP_SEGMENTS > 1
        self._tape_marks = (0, 0)
        self._segments = []          # [(tape_lo, tape_hi, grad_lo, grad_hi)] in backward order, set by _forward
        dev = self.device
        self.hyper = torch.zeros(L.HY_SIZE, dtype=torch.float32, device=dev)
        self.hyper[L.HY_LSCALE], self.hyper[L.HY_GSCALE] = self.loss_scale, 1.0 / self.loss_scale
        self._hyper_ring = None
        self.results = torch.zeros(L.RES_SIZE, dtype=torch.float32, device=dev)
        self.alphas = self.calculate_kl_alphas(n_latent_scales, self.n_groups_per_scale).to(dev)
        self.coeff = torch.ones(self.n_groups, dtype=torch.float32, device=dev)
        self.am = torch.zeros(self.n_groups, dtype=torch.float32, device=dev)
        self.rng_counter = torch.zeros(1, dtype=torch.int64, device=dev)
        self._buf: Dict[int, Dict[str, torch.Tensor]] = {}
        self._plan = None
        self.overlap_wgrad = True   # weight-gradient kernels run on a side stream, joined before Adamax
        # the reference never passes training=True, which under TF-2.3 Keras most likely
        # leaves BatchNorm in inference mode and SpectralNormalization a no-op while training.
        # tf_literal=True reproduces that; the default is the author-intended training semantics.
        self.tf_literal = False
        self._side = None
        # float16 activations on a deep hierarchy: the backward pass renormalises the activation gradient at every
        # latent-group / cell boundary on the device (ops.GradScale; BASELINE.json configs[4] names fp16 for the 40-group
        # CelebA-64 model, whose gradients span 19 decades at initialisation against float16's 12).  NVAE_GRAD_RESCALE=0/1
        # overrides; default: float16 with more than 16 latent groups.
        env = os.environ.get("NVAE_GRAD_RESCALE")
        self.grad_rescale = (dtype == torch.float16 and self.n_groups > 16) if env is None else (env != "0" and dtype != torch.float32)

    # ------------------------------------------------------------------ helpers
    def n_trainable(self) -> int:
        return self.ps.n_trainable

    def eps_shapes(self, B: int):
        shapes, hw = [], self.decoder.z0_shape[0]
        for g in reversed(self.n_groups_per_scale):
            shapes += [(B, hw, hw, self.n_latent_per_group)] * g
            hw *= self.scale_factor
        return shapes

    def _buffers(self, B: int) -> Dict[str, torch.Tensor]:
        if B not in self._buf:
            dev = self.device
            self._buf[B] = {
                "kl_all": torch.zeros(self.n_groups, B, dtype=torch.float32, device=dev),
                "kl_loss": torch.zeros(B, dtype=torch.float32, device=dev),
                "recon": torch.zeros(B, dtype=torch.float32, device=dev),
                "log_p": torch.zeros(B, dtype=torch.float32, device=dev),
                "log_q": torch.zeros(B, dtype=torch.float32, device=dev),
            }
        return self._buf[B]

    def _as_input(self, data) -> torch.Tensor:
        if isinstance(data, (tuple, list)):
            data = data[0]   # labelled data: drop the label (models.py:113-115)
        x = data.to(self.device)
        # the mixture-of-logistics likelihood needs the exact 8-bit levels: keep the image in f32 and
        # cast a copy for the network inside _forward
        want = torch.float32 if self.head == "dmol" else self.dtype
        if x.dtype != want:
            x = x.to(want)
        return x.contiguous()

    def _draw_eps(self, ctx: Ctx, B: int, eps_list):
        if eps_list is not None:
            return [e.to(self.device, torch.float32).contiguous() for e in eps_list]
        # one Philox launch for all groups (15 launches + 15 counter updates per step otherwise); each group's
        # slice starts on a 16-B boundary
        shapes = self.eps_shapes(B)
        sizes = [(int(np.prod(s)) + 3) // 4 * 4 for s in shapes]
        flat = ops.randn(ctx, (sum(sizes),), self.seed, self.rng_counter)
        out, off = [], 0
        for s, n in zip(shapes, sizes):
            out.append(flat[off:off + int(np.prod(s))].view(s))
            off += n
        return out

    # ------------------------------------------------------------------ forward
    def _forward(self, ctx: Ctx, x: torch.Tensor, eps_list, nll=False, mu_sigma_list=None) -> Var:
        B = x.shape[0]
        buf = self._buffers(B)
        eps = self._draw_eps(ctx, B, eps_list)
        if x.dtype != self.dtype:
            xin = ctx.empty(x.shape)
            L.call("nvae_cast", L.F32, ctx.dt, L.ptr(x), L.ptr(xin), x.numel())
            x = xin
        h = self.preprocess(ctx, x)
        self._await_weights(1)
        enc_dec_combiners, final_x = self.encoder(ctx, h)
        enc_mark = len(ctx.tape)
        enc_dec_combiners.reverse()    # bottom-up -> top-down, models.py:93
        if nll:
            buf["log_p"].zero_(); buf["log_q"].zero_()
        self._await_weights(2)
        s = self.decoder(ctx, final_x, enc_dec_combiners, eps, buf["kl_all"], self.coeff, self.hyper,
                         1.0 / B, nll=nll, log_p=buf["log_p"], log_q=buf["log_q"],
                         mu_sigma_list=mu_sigma_list)
        self._tape_marks = (enc_mark, len(ctx.tape))
        dec_mark = len(ctx.tape)
        self._await_weights(3)
        logits = self.postprocess(ctx, s)
        if ctx.record:
            self._segments = self._make_segments(enc_mark, dec_mark, len(ctx.tape))
        return logits

    def _make_segments(self, enc_mark: int, dec_mark: int, end: int):
        """Backward segments (tape range, flat gradient range that is FINAL once the range has run), in backward
        order.  Parameters live in construction order (pre | enc | dec | post) and the tape runs in reverse, so
        after a tape suffix has run the gradients of everything constructed at or after its first layer are final
        (the encoder's combiner convs run inside the decoder's tape: final even earlier).  The encoder + preprocess
        range (45 % of the parameters) is cut twice at encoder group boundaries, so that the all-reduce left
        exposed after the last kernel covers about a quarter of it (24 MB at C2) instead of 108 MB."""
        m = self.param_marks
        nseg = DP_SEGMENTS
        if nseg <= 2:      # postprocess + decoder | encoder + preprocess
            return [(enc_mark, end, m[2], m[4]), (0, enc_mark, m[0], m[2])]
        segs = [(dec_mark, end, m[3], m[4]), (enc_mark, dec_mark, m[2], m[3])]
        enc = self.encoder
        offs, idxs = enc.group_param_off, enc.group_tape_idx
        total = m[2] - m[0]
        cuts = []
        for frac in {3: (), 4: (0.45,)}.get(nseg, (0.6, 0.25)):
            cand = [i for i in range(1, len(offs)) if offs[i] - m[0] <= frac * total and idxs[i] > 0]
            if cand and (not cuts or cand[-1] < cuts[-1]):
                cuts.append(cand[-1])
        hi_t, hi_p = enc_mark, m[2]
        for ci in cuts:                       # descending group index = backward order
            if idxs[ci] < hi_t and offs[ci] < hi_p:
                segs.append((idxs[ci], hi_t, offs[ci], hi_p))
                hi_t, hi_p = idxs[ci], offs[ci]
        segs.append((0, hi_t, m[0], hi_p))
        return segs

    def __call__(self, inputs, nll=False, eps_list=None, training=False):
        """NVAE.call, models.py:89-98 -> (reconstruction logits, z_params, log_p, log_q)."""
        x = self._as_input(inputs)
        B = x.shape[0]
        self.ps.begin_step()
        self.ps.prepare_weights(spectral_norm=False)
        ctx = Ctx(self.ps, self.dtype, training=training, record=False)
        ms = [torch.empty((4,) + s, dtype=torch.float32, device=self.device) for s in self.eps_shapes(B)]
        logits = self._forward(ctx, x, eps_list, nll=nll, mu_sigma_list=ms)
        buf = self._buffers(B)
        z_params = [DistributionParams(m[0], m[1], m[2], m[3]) for m in ms]
        if self.head == "dmol":     # hide the head conv's padding channels
            logits = Var(logits.t[..., :10 * self.num_mixture_dec], False)
        if nll:
            return logits.t, z_params, buf["log_p"].clone(), buf["log_q"].clone()
        zeros = torch.zeros(B, dtype=torch.float32, device=self.device)
        return logits.t, z_params, zeros, zeros.clone()

    # ------------------------------------------------------------------ losses (models.py:191-267)
    @staticmethod
    def calculate_kl_alphas(num_scales, groups_per_scale) -> torch.Tensor:
        """models.py:227-237."""
        coeffs = []
        for i in range(num_scales):
            g = groups_per_scale[num_scales - i - 1]
            coeffs.append(np.square(2 ** i) / g * np.ones(g, dtype=np.float32))
        c = np.concatenate(coeffs)
        return torch.from_numpy((c / c.min()).astype(np.float32))

    def beta(self) -> float:
        """models.py:121-122."""
        m = self.steps if self.step_based_warmup else self.epoch
        return min(m / (0.3 * self.n_total_iterations), 1)

    def learning_rate(self, it: int) -> float:
        """CosineDecay(1e-3, decay_steps), train.py:128-130 [3P]."""
        it = min(it, self.lr_decay_steps)
        return self.base_lr * 0.5 * (1 + math.cos(math.pi * it / self.lr_decay_steps))

    def on_epoch_begin(self, epoch, logs=None):   # models.py:239-240
        self.epoch = epoch

    def calculate_recon_loss(self, inputs, reconstruction, crop_output=False) -> torch.Tensor:
        """models.py:242-250 -> [B]."""
        x = self._as_input(inputs)
        B = x.shape[0]
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        ctx = Ctx(self.ps, self.dtype, training=False, record=False)
        logits = Var(reconstruction.to(torch.float32).contiguous(), False)
        if self.head == "dmol":
            assert not crop_output, "crop_output is the MNIST 28x28 crop (models.py:243-245)"
            ops.dmol_nll(ctx, logits, x, out, 1.0 / B, self.num_mixture_dec)
        else:
            ops.bernoulli_nll(ctx, logits, x, out, 1.0 / B, crop=crop_output)
        return out

    def calculate_bn_loss(self) -> torch.Tensor:
        """models.py:252-267."""
        out = torch.zeros(1, dtype=torch.float32, device=self.device)
        L.call("nvae_bn_absmax_fwd", L.ptr(self.ps.params), L.ptr(self.ps.bn_table),
               len(self.ps.bn_loss_layers), float(self.sr_lambda), L.ptr(out), L.ptr(self.ps.bn_argmax))
        return out[0]

    def calculate_kl_loss(self, kl_all: torch.Tensor, balancing: bool, beta: float = 1.0) -> torch.Tensor:
        """models.py:191-223 on a [G, B] matrix of per-group KL terms -> [B] (times beta)."""
        G, B = kl_all.shape
        hyper = torch.zeros(L.HY_SIZE, dtype=torch.float32, device=self.device)
        hyper[L.HY_BETA] = beta
        hyper[L.HY_BALANCE] = 1.0 if balancing else 0.0
        am = torch.empty(G, dtype=torch.float32, device=self.device)
        coeff = torch.empty(G, dtype=torch.float32, device=self.device)
        out = torch.empty(B, dtype=torch.float32, device=self.device)
        res = torch.empty(L.RES_SIZE, dtype=torch.float32, device=self.device)
        zero = torch.zeros(B, dtype=torch.float32, device=self.device)
        L.call("nvae_kl_absmean", L.ptr(kl_all), G, B, L.ptr(am))
        L.call("nvae_loss_finalize", L.ptr(kl_all), L.ptr(am), L.ptr(self.alphas), G, B, L.ptr(zero), None,
               L.ptr(hyper), L.ptr(coeff), L.ptr(out), L.ptr(res))
        return out

    # ------------------------------------------------------------------ training step
    def _set_hyper(self):
        beta = self.beta()
        t = self.opt_iterations + 1
        lr_t = self.learning_rate(self.opt_iterations) / (1 - ADAMAX_B1 ** t)
        # slots 0..2 come from the host every step (slots 3.. hold the loss-scale state, maintained on the device),
        # out of a small ring of pinned buffers; blocking by default (ASYNC_HYPER above)
        if self._hyper_ring is None:
            self._hyper_ring = [[torch.zeros(3, dtype=torch.float32).pin_memory(), None] for _ in range(8)]
            self._hyper_slot = 0
        slot = self._hyper_ring[self._hyper_slot]
        self._hyper_slot = (self._hyper_slot + 1) % len(self._hyper_ring)
        if slot[1] is not None:
            slot[1].synchronize()                      # the copy that last used this buffer (8 steps ago) has run
        h = slot[0]
        h[L.HY_LR], h[L.HY_BETA], h[L.HY_BALANCE] = lr_t, beta, 1.0 if beta < 1 else 0.0
        self.hyper[:3].copy_(h, non_blocking=ASYNC_HYPER)
        slot[1] = torch.cuda.Event()
        slot[1].record()

    def _seg_forward(self, x: torch.Tensor, eps_list, spectral_norm=True):
        """SN + forward + per-rank loss statistics.  Returns the context holding the tape."""
        ps = self.ps
        B = x.shape[0]
        buf = self._buffers(B)
        if self.overlap_wgrad and self._side is None and self.device.type == "cuda":
            self._side = torch.cuda.Stream(device=self.device)
        self._prepare_weights_staged(spectral_norm and not self.tf_literal)
        ctx = Ctx(ps, self.dtype, training=not self.tf_literal, record=True,
                  side_stream=self._side if self.overlap_wgrad else None)
        ctx.dw_pre = self.reducer is None or DP_DW_PRE
        if self.grad_rescale:
            ctx.gs = ops.GradScale(ctx)
        self._bn_loss = ctx.zeros_f32(1)
        nb = len(ps.bn_loss_layers)
        if nb:
            L.call("nvae_bn_absmax_fwd", L.ptr(ps.params), L.ptr(ps.bn_table), nb, float(self.sr_lambda),
                   L.ptr(self._bn_loss), L.ptr(ps.bn_argmax))
        logits = self._forward(ctx, x, eps_list)
        if self.head == "dmol":
            ops.dmol_nll(ctx, logits, x, buf["recon"], 1.0 / B, self.num_mixture_dec, hyper=self.hyper)
        else:
            ops.bernoulli_nll(ctx, logits, x, buf["recon"], 1.0 / B, hyper=self.hyper)
        L.call("nvae_kl_absmean", L.ptr(buf["kl_all"]), self.n_groups, B, L.ptr(self.am))
        self._logits = logits
        return ctx

    def _prepare_weights_staged(self, sn: bool):
        """Per-step zeroing, spectral norm + compute copies (1.7 GB of HBM traffic per step at C2, 0.8 ms).  Only the preprocess convs
        are prepared on the main stream; the encoder / decoder / postprocess parts run on the side stream, in the
        order the forward pass needs them, while the earlier modules already compute (`_await_weights` makes the
        main stream wait for a part just before its module starts).  The parts touch disjoint slices of every
        buffer involved (masters, SN vectors and scratch, copies)."""
        ps = self.ps
        self._prep_events = {}
        if not (OVERLAP_PREP and self.overlap_wgrad and self._side is not None and ps.n_prep_parts() == 4):
            ps.begin_step()
            ps.prepare_weights(spectral_norm=sn)
            return
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)
        ps.begin_step(zero_grads=False)
        ps.prepare_weights(spectral_norm=sn, part=0)
        with torch.cuda.stream(self._side):
            ps.grads.zero_()                 # 250 MB memset: needed by the backward pass only
            for k in (1, 2, 3):
                ps.prepare_weights(spectral_norm=sn, part=k)
                ev = torch.cuda.Event()
                ev.record(self._side)
                self._prep_events[k] = ev

    def _await_weights(self, part: int):
        ev = getattr(self, "_prep_events", {}).pop(part, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def n_segments(self) -> int:
        return len(self._segments)

    def _seg_backward(self, ctx: Ctx, B: int, part: Optional[int] = None, join: bool = True, flush: bool = True):
        """Loss + backward.  part=None: everything.  part k: segment k of `self._segments` (postprocess, decoder,
        then the encoder + preprocess pieces; run in that order); after part k the gradients of flat-buffer
        range `self.grad_range(k)` are final."""
        ps = self.ps
        if part in (None, 0):
            buf = self._buffers(B)
            L.call("nvae_loss_finalize", L.ptr(buf["kl_all"]), L.ptr(self.am), L.ptr(self.alphas), self.n_groups,
                   B, L.ptr(buf["recon"]), L.ptr(self._bn_loss), L.ptr(self.hyper), L.ptr(self.coeff),
                   L.ptr(buf["kl_loss"]), L.ptr(self.results))
            nb = len(ps.bn_loss_layers)
            if nb and ctx.gs is None:      # subgradient of the BN regulariser: depends on the parameters only, so it goes first
                L.call("nvae_bn_absmax_bwd", L.ptr(ps.params), L.ptr(ps.grads), L.ptr(ps.bn_table),
                       L.ptr(ps.bn_argmax), nb, float(self.sr_lambda), L.ptr(self.hyper))
        if part is None:
            ctx.backward()
            glo, ghi = 0, ps.grads.numel()
        else:
            lo, hi = self._segments[part][:2]
            ctx.backward(lo, hi if part else None, join=join, flush=flush)      # (the loss ops appended after the forward pass belong to segment 0)
            glo, ghi = self.grad_range(part)
        if flush:
            self._seg_finish_grads(ctx, glo, ghi, join)

    def _seg_finish_grads(self, ctx: Ctx, glo: int, ghi: int, join: bool):
        ps = self.ps
        if ctx.gs is not None:
            # range-normalised backward pass: the parameter gradients of this range carry their dy's power-of-two tag until
            # here; the regulariser's subgradient (true scale) is added afterwards, for the layers of this range
            assert join, "GradScale.unscale needs the weight-gradient kernels joined"
            ctx.gs.unscale(glo, ghi)
            rows = [i for i, b in enumerate(ps.bn_loss_layers) if glo <= b.gamma.off < ghi]
            if rows:
                assert rows == list(range(rows[0], rows[-1] + 1))
                L.call("nvae_bn_absmax_bwd", L.ptr(ps.params), L.ptr(ps.grads), L.ptr(ps.bn_table) + 8 * rows[0],
                       L.ptr(ps.bn_argmax) + 4 * rows[0], len(rows), float(self.sr_lambda), L.ptr(self.hyper))

    def grad_range(self, part: int):
        """Flat gradient range completed by backward segment `part` (see _make_segments)."""
        return self._segments[part][2:]

    def sync_replicas(self):
        """Align the replicas on rank 0's parameters, state, optimizer slots and loss-scale state (start-up / resume).
        During training the PARAMETERS and both Adamax slots do not drift: the all-reduced gradient is the same bits
        on every rank, Adamax is elementwise and spectral normalisation is deterministic (sn.hip)
        (tests/test_dist_cli_gpu.py asserts it over five graphed steps without any re-broadcast).  The non-trainable
        STATE does: every rank normalises its own shard of the batch, so the BatchNorm moving statistics are
        rank-local from step 1 on - `sync_state` averages them before anything reads them (evaluation, sampling,
        checkpoints, the early-stopping snapshot)."""
        if self.reducer is not None:
            for t in (self.ps.params, self.ps.state, self.ps.adam_m, self.ps.adam_u, self.hyper):
                self.reducer.broadcast_(t)

    def sync_state(self):
        """Data parallel: replace the rank-local BatchNorm moving statistics by their mean over ranks (the moving
        average of the per-shard batch statistics of ALL shards; what a single process that saw the global batch in
        `world` pieces would hold).  The spectral-norm vectors in the same buffer are identical on every rank and
        come back unchanged (a mean of equal numbers, up to one rounding for world sizes that are not a power of two -
        the same rounding on every rank).  Collective: every rank must call it at the same point."""
        if self.reducer is not None:
            self.reducer.allreduce_mean_(self.ps.state)

    def loss_scale_report(self):
        """(scale, clean steps since the last change) of the dynamic loss scale, read from the device (a host sync:
        call it once per epoch, not per step).  A scale pinned at the floor means every step is being skipped."""
        h = self.hyper.detach().cpu()
        return float(h[L.HY_LSCALE]), int(h[L.HY_GOOD])

    def _dp_segments(self) -> bool:
        return self.reducer is not None and self.overlap_allreduce

    def _seg_update(self, lo: int = 0, hi: Optional[int] = None):
        ps = self.ps
        hi = ps.params.numel() if hi is None else hi
        whole = lo == 0 and hi == ps.params.numel()
        if self.dynamic_loss_scale and whole:      # a non-finite gradient anywhere -> Adamax skips the step
            L.call("nvae_grad_guard", L.ptr(ps.grads), ps.grads.numel(), L.ptr(self.hyper))
        if hi > lo:
            L.call("nvae_adamax", L.ptr(ps.params) + 4 * lo, L.ptr(ps.grads) + 4 * lo, L.ptr(ps.adam_m) + 4 * lo,
                   L.ptr(ps.adam_u) + 4 * lo, hi - lo, L.ptr(self.hyper), ADAMAX_B1, ADAMAX_B2, ADAMAX_EPS)
        if self.dynamic_loss_scale and whole:
            L.call("nvae_loss_scale_update", L.ptr(self.hyper), 2.0 ** -24, 2.0 ** 16)

    def _fused_update(self) -> bool:
        """Single GPU: no collective stands between a backward segment and its optimizer step, so the Adamax launch
        of a finished segment's parameter range goes onto the side stream behind that segment's weight gradients
        and runs under the next segment's data-gradient chain (0.3 ms of HBM-bound work per step at C2)."""
        return (OVERLAP_ADAMAX and self.reducer is None and self.overlap_wgrad and self._side is not None
                and not self.dynamic_loss_scale)

    def _backward_with_update(self, ctx: Ctx, B: int):
        for part in range(self.n_segments()):
            self._seg_backward(ctx, B, part, join=False)
            lo, hi = self.grad_range(part)
            ctx.fork(lambda lo=lo, hi=hi: self._seg_update(lo, hi))
        ctx.join_side()

    def train_step(self, data, eps_list=None, spectral_norm=True, update=True):
        """NVAE.train_step, models.py:100-135 (eager launch path).  Returns device tensors:
        loss (scalar), reconstruction_loss [B], kl_loss [B] (already beta-scaled, Q11), bn_loss,
        plus kl_per_group [G, B] (unscaled)."""
        x = self._as_input(data)
        B = x.shape[0]
        self._set_hyper()
        ctx = self._seg_forward(x, eps_list, spectral_norm)
        if self.reducer is not None:
            self.reducer.allreduce_mean_(self.am)
        fused = update and self._fused_update()
        if self._dp_segments():
            works = []
            for part in range(self.n_segments()):
                self._seg_backward(ctx, B, part)
                self.reducer.start_allreduce_(self.ps.grads, *self.grad_range(part), works)
            self.reducer.finish_allreduce_(works)
        elif fused:
            self._backward_with_update(ctx, B)
        else:
            self._seg_backward(ctx, B)
            if self.reducer is not None:
                self.reducer.allreduce_grads_(self.ps.grads)
        if update:
            if not fused:
                self._seg_update()
            self.opt_iterations += 1
        self.steps += 1
        return self._step_outputs(B)

    def _step_outputs(self, B):
        buf = self._buffers(B)
        return {"loss": self.results[L.RES_LOSS], "reconstruction_loss": buf["recon"],
                "kl_loss": buf["kl_loss"], "bn_loss": self.results[L.RES_BN],
                "kl_per_group": buf["kl_all"]}

    # ------------------------------------------------------------------ hipGraph-replayed step
    def capture_train_step(self, batch_shape, warmup: int = 2):
        """Capture the training step into hipGraphs for a fixed batch shape.  The step is
        [graph: SN + forward + loss stats] -> (all-reduce of the [G] KL statistic when DP)
        -> [graph: loss + backward] -> (gradient all-reduce when DP) -> [graph: Adamax].
        With a reducer and overlap_allreduce the backward graph is one graph per segment (postprocess | decoder
        | three encoder + preprocess pieces) and each segment's gradient range is all-reduced while the next replays.
        Host-side scalars (lr, beta) reach the kernels through the `hyper` device buffer, noise is
        drawn in-graph from a device counter, so replays are exact continuations of training.

        Capture has no side effects on the training state: the warm-up steps (they load the code objects
        and size torch's allocator pools before capture) run real kernels on a zero batch, so parameters,
        Adamax slots, BN moving statistics / SN vectors and the noise counter are snapshotted before and
        restored after them.  A fresh or resumed run therefore continues exactly where its state says
        (train.py:46-55,133-135 of the reference: load_weights + initial_epoch)."""
        B = int(batch_shape[0])
        self._static_x = torch.zeros(tuple(batch_shape), device=self.device,
                                     dtype=torch.float32 if self.head == "dmol" else self.dtype)
        ps = self.ps
        snap = [(t, t.clone()) for t in (ps.params, ps.state, ps.adam_m, ps.adam_u, self.rng_counter, self.hyper,
                                         self.coeff, self.am, self.results)]
        self._set_hyper()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                ctx = self._seg_forward(self._static_x, None)
                if self._fused_update():
                    self._backward_with_update(ctx, B)
                else:
                    self._seg_backward(ctx, B)
                    self._seg_update()
            for t, saved in snap:
                t.copy_(saved)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize(self.device)
        del snap
        pool = torch.cuda.graph_pool_handle()
        g1, g3 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: a NCCL/RCCL watchdog thread may touch the HIP API while we capture
        kw = dict(pool=pool, capture_error_mode="thread_local")
        if CAPTURE_PRIORITY:
            # capture on a high-priority stream so that the main chain's kernel nodes outrank the side stream's
            # weight-gradient nodes when both are runnable.  Measured SLOWER: 6 190-6 215 against 6 300-6 320 images/s
            # (and 3 905 with the weight gradients flushed every 64), so off by default (NVAE_CAPTURE_PRIORITY=1)
            lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
            kw["stream"] = torch.cuda.Stream(device=self.device, priority=hi)
        with torch.cuda.graph(g1, **kw):
            ctx = self._seg_forward(self._static_x, None)
        if self._dp_segments():
            g2 = []
            for part in range(self.n_segments()):
                g2.append(torch.cuda.CUDAGraph())
                with torch.cuda.graph(g2[-1], **kw):
                    self._seg_backward(ctx, B, part)
        elif self._fused_update():
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, **kw):
                self._backward_with_update(ctx, B)
            g3 = None                       # the optimizer step is part of the backward graph
        else:
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, **kw):
                self._seg_backward(ctx, B)
        if g3 is not None:
            with torch.cuda.graph(g3, **kw):
                self._seg_update()
        self._plan = (g1, g2, g3, B)
        return self

    def train_step_graphed(self, data=None):
        """Replay the captured step (data=None reuses the static input buffer contents)."""
        g1, g2, g3, B = self._plan
        if data is not None:
            self._static_x.copy_(self._as_input(data))
        self._set_hyper()
        g1.replay()
        if self.reducer is not None and not DP_DIAG_SKIP_AM:
            self.reducer.allreduce_mean_(self.am)
        if isinstance(g2, list):
            works = []
            for part, g in enumerate(g2):
                g.replay()
                if not DP_DIAG_SKIP_GRADS:
                    self.reducer.start_allreduce_(self.ps.grads, *self.grad_range(part), works)
            self.reducer.finish_allreduce_(works)
        else:
            g2.replay()
            if self.reducer is not None and not DP_DIAG_SKIP_GRADS:
                self.reducer.allreduce_grads_(self.ps.grads)
        if g3 is not None:
            g3.replay()
        self.opt_iterations += 1
        self.steps += 1
        return self._step_outputs(B)

    # ------------------------------------------------------------------ sampling (models.py:137-189)
    def _images(self, logits: torch.Tensor, greyscale: bool, dmol_noise=None) -> torch.Tensor:
        if self.head == "dmol":      # one draw from the mixture per pixel (t = 1, as NVAE does for its output)
            u_mix, u_pix = dmol_noise if dmol_noise is not None else (None, None)
            return ops.dmol_sample(logits, self.num_mixture_dec, 1.0, u_mix, u_pix)
        return torch.sigmoid(logits) if greyscale else torch.bernoulli(torch.sigmoid(logits))

    def sample(self, n_samples=16, temperature=1.0, greyscale=True, eps_list=None, dmol_noise=None):
        """Ancestral sampling.  Temperature scales z0's sigma only (Q8).  Returns
        (images [B,H,W,C] f32 in [0,1], last_s, z1, z2) like the reference.  dmol_noise: optional
        (u_mix [B,H,W,M], u_pix [B,H,W,3]) uniforms for the mixture-of-logistics head."""
        ps = self.ps
        B = n_samples
        ps.begin_step()
        ps.prepare_weights(spectral_norm=False)
        ctx = Ctx(ps, self.dtype, training=False, record=False)
        eps = self._draw_eps(ctx, B, eps_list)
        buf = self._buffers(B)
        dec = self.decoder
        # group 0: mu = softclamp5(0) = 0, sigma = exp(softclamp5(0)) + 1e-2 = 1.01 (models.py:141-144)
        zero_p = Var(torch.zeros((B,) + dec.z0_shape[:2] + (2 * self.n_latent_per_group,),
                                 dtype=torch.float32, device=self.device), False)
        eps0 = eps[0] * temperature if temperature != 1.0 else eps[0]
        z = ops.sampler(ctx, zero_p, None, eps0, buf["kl_all"][0], self.coeff[0:1], self.hyper, 1.0 / B)
        s = dec.tiled_h(ctx, B)
        decoder_index = 0
        last_s = None
        last_params = None
        for layer in dec.groups:
            if isinstance(layer, DecoderSampleCombiner):
                if decoder_index > 0:
                    p = dec.sampler.get_params(ctx, dec.sampler.dec_sampler, decoder_index, s)
                    # prior sample: mu = sc(mu_p), sigma = exp(sc(log_sigma_p)) + 1e-2 == the group-0
                    # formula applied to the decoder's raw parameters (models.py:153-159)
                    z = ops.sampler(ctx, p, None, eps[decoder_index], buf["kl_all"][decoder_index],
                                    self.coeff[0:1], self.hyper, 1.0 / B)
                    last_params = p
                last_s = s
                s = layer(ctx, s, z)
                decoder_index += 1
            elif isinstance(layer, list):
                for cell in layer:
                    s = cell(ctx, s)
            else:
                s = layer(ctx, s)
        logits = self.postprocess(ctx, s).t
        images = self._images(logits, greyscale, dmol_noise)
        # z1, z2: two more draws from the last group's prior (models.py:175-176)
        zs = []
        for _ in range(2):
            e = ops.randn(ctx, eps[-1].shape, self.seed, self.rng_counter)
            src = last_params if last_params is not None else zero_p
            zs.append(ops.sampler(ctx, src, None, e, buf["kl_all"][0], self.coeff[0:1], self.hyper, 1.0 / B).t)
        return images, (last_s.t if last_s is not None else None), zs[0], zs[1]

    def sample_with_z(self, z: torch.Tensor, s: torch.Tensor):
        """models.py:181-189: decode from a fixed last-group z and its decoder state s."""
        ps = self.ps
        ps.begin_step()
        ps.prepare_weights(spectral_norm=False)
        ctx = Ctx(ps, self.dtype, training=False, record=False)
        last = self.decoder.groups[-1]
        out = last(ctx, Var(s.to(self.device, self.dtype).contiguous(), False),
                   Var(z.to(self.device, self.dtype).contiguous(), False))
        logits = self.postprocess(ctx, out).t
        return self._images(logits, True)
