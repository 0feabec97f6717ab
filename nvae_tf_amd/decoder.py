"""Top-down tower, mirroring the reference's decoder.py."""
from __future__ import annotations

import os

from typing import List

import torch

from . import _lib as L
from . import ops
from .common import Rescaler, RescaleType, Sampler, SqueezeExcitation
from .ops import Ctx, Var


COMBINER_STATS = os.environ.get("NVAE_COMBINER_STATS", "1") != "0"    # the combiner's GEMM emits the next BatchNorm's statistics


class DecoderSampleCombiner:
    """decoder.py:107-117: SN conv1x1(concat(x, z)).  Concat-free: two accumulating GEMMs over the
    row slices [0, Cx) and [Cx, Cx+L) of the one kernel."""

    def __init__(self, ps, name: str, x_channels: int, z_channels: int, output_channels: int):
        self.conv = ps.conv(name + ".conv", 1, x_channels + z_channels, output_channels)
        self.cx, self.cz = x_channels, z_channels

    def __call__(self, ctx: Ctx, x: Var, z: Var) -> Var:
        # the thin z slice first (direct kernel, carries the bias), then the MFMA GEMM over x accumulates onto it and emits
        # the BatchNorm statistics of the SUM in its epilogue: the next cell's first BatchNorm needs no statistics pass
        if not COMBINER_STATS:
            out = ops.conv2d(ctx, x, self.conv, c_off=0, cin=self.cx)
            return ops.conv2d(ctx, z, self.conv, c_off=self.cx, cin=self.cz, bias=False, out=out, accumulate=True)
        out = ops.conv2d(ctx, z, self.conv, c_off=self.cx, cin=self.cz)
        return ops.conv2d(ctx, x, self.conv, c_off=0, cin=self.cx, bias=False, out=out, accumulate=True, want_stats=True)


class GenerativeResidualCell:
    """decoder.py:120-147: 0.1*x + SE(BN(conv1x1(Swish(BN(dw5x5(Swish(BN(conv1x1(BN(x))))))))))."""

    def __init__(self, ps, name: str, output_channels: int, expansion_ratio: int = 6):
        c, e = output_channels, expansion_ratio * output_channels
        self.batch_norm1 = ps.bn(name + ".bn1", c, True)
        self.conv1 = ps.conv(name + ".conv1", 1, c, e)
        self.batch_norm2 = ps.bn(name + ".bn2", e, True)
        self.depth_conv = ps.dw(name + ".dw", e)          # not spectrally normalised, has bias
        self.batch_norm3 = ps.bn(name + ".bn3", e, True)
        self.conv2 = ps.conv(name + ".conv2", 1, e, c)
        self.batch_norm4 = ps.bn(name + ".bn4", c, True)
        self.se = SqueezeExcitation(ps, name + ".se", c)

    def __call__(self, ctx: Ctx, inputs: Var) -> Var:
        x = ops.bn_act(ctx, inputs, self.batch_norm1)
        x = ops.conv2d(ctx, x, self.conv1, stats_bn=self.batch_norm2)
        x = ops.bn_act(ctx, x, self.batch_norm2, L.ACT_SWISH)
        x = ops.dwconv5(ctx, x, self.depth_conv, want_stats=True)   # feeds bn3
        x = ops.bn_act(ctx, x, self.batch_norm3, L.ACT_SWISH)
        x = ops.conv2d(ctx, x, self.conv2, stats_bn=self.batch_norm4)
        x = ops.bn_act(ctx, x, self.batch_norm4)       # applied inside the SE kernel
        return self.se(ctx, x, inputs, 0.1, 1.0)


class Decoder:
    """decoder.py:9-104.  `n_groups_per_scale` arrives already reversed (models.py:69)."""

    def __init__(self, ps, n_encoder_channels, n_decoder_channels, n_latent_per_group, res_cells_per_group,
                 n_latent_scales, n_groups_per_scale: List[int], mult, scale_factor, input_shape):
        self.n_decoder_channels = n_decoder_channels
        self.n_latent_per_group = n_latent_per_group
        enc_ch, dec_ch = [], []
        m = mult
        for scale in range(n_latent_scales):
            for _ in range(n_groups_per_scale[scale]):
                enc_ch.append(n_encoder_channels * m)
                dec_ch.append(n_decoder_channels * m)
            m //= scale_factor
        self.sampler = Sampler(ps, n_latent_scales, n_groups_per_scale, n_latent_per_group, enc_ch, dec_ch)
        self.groups = []
        zi = 0
        for scale in range(n_latent_scales):
            for group in range(n_groups_per_scale[scale]):
                c = n_decoder_channels * mult
                if not (scale == 0 and group == 0):
                    self.groups.append([GenerativeResidualCell(ps, f"dec.g{zi}.c{i}", c)
                                        for i in range(res_cells_per_group)])
                    self.groups.append(DecoderSampleCombiner(ps, f"dec.comb{zi}", c, n_latent_per_group, c))
                else:
                    # first combiner consumes h (n_decoder_channels wide) and z0
                    self.groups.append(DecoderSampleCombiner(ps, f"dec.comb{zi}", n_decoder_channels,
                                                             n_latent_per_group, c))
                zi += 1
            if scale < n_latent_scales - 1:
                c = n_decoder_channels * mult
                self.groups.append(Rescaler(ps, f"dec.up{scale}", c, c // scale_factor, scale_factor,
                                            RescaleType.UP, in_bn_loss=True))
                self.groups[-1].feeds_bn = True        # the next scale's first cell starts with a BatchNorm over its output
                mult //= scale_factor
        self.mult = mult
        self.n_groups = zi
        hw = (int(input_shape[1]), int(input_shape[2]))
        self.z0_shape = (hw[0], hw[1], n_latent_per_group)
        self.h = ps.tensor("dec.h", torch.rand((hw[0], hw[1], n_decoder_channels), generator=ps.gen,
                                               dtype=torch.float64))   # decoder.py:60-62

    def tiled_h(self, ctx: Ctx, batch: int) -> Var:
        """tf.tile(h) as a broadcast cast kernel into a [B,H,W,D] activation; its gradient is summed
        back over the batch by the caller (h_backward)."""
        ps = ctx.ps
        hh, hw, d = self.h.shape
        src = ps.view(self.h).reshape(1, hh, hw, d).expand(batch, hh, hw, d).contiguous()
        out = Var(ctx.empty((batch, hh, hw, d)))
        L.call("nvae_cast", L.F32, ctx.dt, L.ptr(src), L.ptr(out.t), src.numel())
        if ctx.record:
            def bwd():
                if ctx.gs is not None:
                    ctx.gs.note(self.h, out.gid)
                if out.g is not None:
                    # dh[c'] = sum_b dH[b, c']: column sum over the batch with c' = (h,w,c) flattened
                    L.call("nvae_colsum", ctx.dt, L.ptr(out.g), batch, hh * hw * d, hh * hw * d,
                           L.ptr(ps.grads) + self.h.off * 4)
            ctx.tape.append(bwd)
        return out

    def __call__(self, ctx: Ctx, prior: Var, enc_dec_combiners: List, eps_list, kl_all: torch.Tensor,
                 coeff: torch.Tensor, hyper: torch.Tensor, inv_batch: float, nll: bool = False,
                 log_p=None, log_q=None, mu_sigma_list=None):
        B = prior.t.shape[0]
        lq, lp = (log_q, log_p) if nll else (None, None)
        ms = (lambda i: mu_sigma_list[i]) if mu_sigma_list is not None else (lambda i: None)
        z0 = self.sampler(ctx, prior, 0, eps_list[0], kl_all[0], coeff[0:1], hyper, inv_batch,
                          logq=lq, logp=lp, mu_sigma=ms(0))
        h = self.tiled_h(ctx, B)
        x = self.groups[0](ctx, h, z0)
        combine_idx = 0
        for group in self.groups[1:]:
            if isinstance(group, DecoderSampleCombiner):
                x = ops.grad_boundary(ctx, x)      # (float16 on deep hierarchies: renormalise the gradient per latent group)
                enc_prior = enc_dec_combiners[combine_idx](x)
                zi = combine_idx + 1
                z = self.sampler(ctx, x, zi, eps_list[zi], kl_all[zi], coeff[zi:zi + 1], hyper, inv_batch,
                                 enc_prior=enc_prior, logq=lq, logp=lp, mu_sigma=ms(zi))
                x = group(ctx, x, z)
                combine_idx += 1
            elif isinstance(group, list):
                for cell in group:
                    x = cell(ctx, x)
            else:
                x = group(ctx, x)
        return x
