"""Shared blocks, mirroring the reference's common.py: DistributionParams, Sampler,
SqueezeExcitation (fused with the residual add it always feeds), Rescaler."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import List, Optional

import torch

from . import _lib as L
from . import ops
from .ops import Ctx, Var


@dataclass
class DistributionParams:   # common.py:12-17
    enc_mu: torch.Tensor
    enc_sigma: torch.Tensor
    dec_mu: torch.Tensor
    dec_sigma: torch.Tensor


class RescaleType(Enum):    # common.py:105-107
    UP = auto()
    DOWN = auto()


class SqueezeExcitation:
    """common.py:110-142.  Always followed by a scaled residual add in the reference
    (encoder.py:107, decoder.py:147, preprocess.py:107, postprocess.py:58), so the call takes the
    skip tensor and both scales and runs gate + scale + add as one pass."""

    def __init__(self, ps, name: str, channels: int):
        self.p = ps.se(name, channels)

    def __call__(self, ctx: Ctx, x: Var, skip: Var, skip_scale: float, branch_scale: float) -> Var:
        return ops.se_residual(ctx, x, self.p, skip, skip_scale, branch_scale)


class Rescaler:
    """common.py:145-174: BN -> Swish -> [nearest x factor] -> SN conv3x3 (stride = factor if DOWN).
    The nearest upsample is folded into the conv's gather and never materialised."""

    def __init__(self, ps, name: str, in_channels: int, n_channels: int, scale_factor: int,
                 rescale_type: RescaleType, in_bn_loss: bool = False):
        self.bn = ps.bn(name + ".bn", in_channels, in_bn_loss)
        self.conv = ps.conv(name + ".conv", 3, in_channels, n_channels)
        self.mode = rescale_type
        self.feeds_bn = False      # set by the owner when the output goes straight into a BatchNorm
        self.stats_bn = None       # ... and which one (finalized by the conv kernel)
        self.factor = scale_factor

    def __call__(self, ctx: Ctx, x: Var) -> Var:
        y = ops.bn_act(ctx, x, self.bn, L.ACT_SWISH)
        if self.mode == RescaleType.UP:
            return ops.conv2d(ctx, y, self.conv, up=self.factor, want_stats=self.feeds_bn, stats_bn=self.stats_bn)
        # DOWN (encoder.py:47-57): the next tower's first cell starts with a BatchNorm over this output
        return ops.conv2d(ctx, y, self.conv, stride=self.factor, want_stats=True)


class Sampler:
    """common.py:20-102.  enc_sampler[i]: SN conv3x3 -> 2L; dec_sampler[i]: ELU + SN conv1x1 -> 2L
    (None for i = 0).  `call` fuses split + softclamp + exp + reparameterised sample + this group's
    KL (+ log q / log p) into one kernel; eps is an explicit input so results are reproducible."""

    def __init__(self, ps, n_latent_scales: int, n_groups_per_scale: List[int], n_latent_per_group: int,
                 enc_channels: List[int], dec_channels: List[int]):
        self.n_latent_per_group = n_latent_per_group
        self.enc_sampler, self.dec_sampler = [], []
        zi = 0
        for scale in range(n_latent_scales):
            for group in range(n_groups_per_scale[scale]):
                self.enc_sampler.append(ps.conv(f"dec.samp.enc{zi}.conv", 3, enc_channels[zi],
                                                2 * n_latent_per_group))
                if scale == 0 and group == 0:
                    self.dec_sampler.append(None)   # common.py:49-51
                else:
                    self.dec_sampler.append(ps.conv(f"dec.samp.dec{zi}.conv", 1, dec_channels[zi],
                                                    2 * n_latent_per_group))
                zi += 1

    def get_params(self, ctx: Ctx, sampler_list, z_idx: int, prior: Var) -> Var:
        """Raw (mu, log_sigma) conv output, kept in f32 (common.py:70-74; no squeeze)."""
        if sampler_list is self.dec_sampler:
            prior = ops.unary(ctx, prior, L.OP_ELU)
        return ops.conv2d(ctx, prior, sampler_list[z_idx], out_f32=True)

    def __call__(self, ctx: Ctx, prior: Var, z_idx: int, eps: torch.Tensor, kl_out: torch.Tensor,
                 coeff: torch.Tensor, hyper: torch.Tensor, inv_batch: float,
                 enc_prior: Optional[Var] = None, logq=None, logp=None, mu_sigma=None) -> Var:
        if enc_prior is None:
            enc_prior = prior
        enc_p = self.get_params(ctx, self.enc_sampler, z_idx, enc_prior)
        dec_p = None if z_idx == 0 else self.get_params(ctx, self.dec_sampler, z_idx, prior)
        return ops.sampler(ctx, enc_p, dec_p, eps, kl_out, coeff, hyper, inv_batch, logq, logp, mu_sigma)
