"""Bottom-up tower, mirroring the reference's encoder.py."""
from __future__ import annotations

from functools import partial
from typing import List

from . import _lib as L
from . import ops
from .common import Rescaler, RescaleType, SqueezeExcitation
from .ops import Ctx, Var


class EncoderDecoderCombiner:
    """encoder.py:9-16: encoder_x + SN conv1x1(decoder_x); the add is the conv's residual epilogue."""

    def __init__(self, ps, name: str, dec_channels: int, n_channels: int):
        self.decoder_conv = ps.conv(name + ".conv", 1, dec_channels, n_channels)

    def __call__(self, ctx: Ctx, encoder_x: Var, decoder_x: Var) -> Var:
        return ops.conv2d(ctx, decoder_x, self.decoder_conv, residual=encoder_x)


class EncodingResidualCell:
    """encoder.py:86-107: 0.1*x + SE(conv3x3(Swish(BN(conv3x3(Swish(BN(x)))))))."""

    def __init__(self, ps, name: str, output_channels: int):
        c = output_channels
        self.batch_norm1 = ps.bn(name + ".bn1", c, True)
        self.conv1 = ps.conv(name + ".conv1", 3, c, c)
        self.batch_norm2 = ps.bn(name + ".bn2", c, True)
        self.conv2 = ps.conv(name + ".conv2", 3, c, c)
        self.se = SqueezeExcitation(ps, name + ".se", c)

    def __call__(self, ctx: Ctx, inputs: Var) -> Var:
        x = ops.bn_act(ctx, inputs, self.batch_norm1, L.ACT_SWISH)
        x = ops.conv2d(ctx, x, self.conv1, stats_bn=self.batch_norm2)
        x = ops.bn_act(ctx, x, self.batch_norm2, L.ACT_SWISH)
        x = ops.conv2d(ctx, x, self.conv2)
        return self.se(ctx, x, inputs, 0.1, 1.0)


class Encoder:
    """encoder.py:19-83."""

    def __init__(self, ps, n_encoder_channels, n_decoder_channels, n_latent_per_group, res_cells_per_group,
                 n_latent_scales, n_groups_per_scale: List[int], mult, scale_factor, input_shape):
        self.groups = []
        # flat-parameter offset at which each entry of `groups` starts, and (filled in by __call__) the tape index
        # at which its forward work starts: cut points for the segmented, all-reduce-overlapped backward pass
        self.group_param_off: List[int] = []
        self.group_tape_idx: List[int] = []
        gi = ti = 0
        shape = list(input_shape)
        for scale in range(n_latent_scales):
            n_groups = n_groups_per_scale[scale]
            for group_idx in range(n_groups):
                c = n_encoder_channels * mult
                self.group_param_off.append(ps._p_cursor)
                self.groups.append([EncodingResidualCell(ps, f"enc.g{gi}.c{i}", c)
                                    for i in range(res_cells_per_group)])
                gi += 1
                if not (scale == n_latent_scales - 1 and group_idx == n_groups - 1):
                    self.group_param_off.append(ps._p_cursor)
                    self.groups.append(EncoderDecoderCombiner(ps, f"enc.comb{ti}",
                                                              n_decoder_channels * mult, c))
                    ti += 1
            if scale < n_latent_scales - 1:
                c = n_encoder_channels * mult
                self.group_param_off.append(ps._p_cursor)
                self.groups.append(Rescaler(ps, f"enc.down{scale}", c, c * scale_factor, scale_factor,
                                            RescaleType.DOWN, in_bn_loss=True))
                mult *= scale_factor
                shape = [shape[0], shape[1] // scale_factor, shape[2] // scale_factor, shape[3] * scale_factor]
        self.final_conv = ps.conv("enc.final.conv", 1, n_encoder_channels * mult, n_encoder_channels * mult)
        self.mult = mult
        self.output_shape_ = shape

    def __call__(self, ctx: Ctx, x: Var):
        enc_dec_combiners = []
        self.group_tape_idx = []
        for group in self.groups:
            self.group_tape_idx.append(len(ctx.tape))
            if isinstance(group, EncoderDecoderCombiner):
                enc_dec_combiners.append(partial(group, ctx, x))   # evaluated later by the decoder
            elif isinstance(group, list):
                xb = ops.grad_boundary(ctx, x)     # (float16 on deep hierarchies; the combiner tap keeps the unwrapped x)
                for cell in group:
                    xb = cell(ctx, xb)
                x = xb
            else:
                x = group(ctx, x)
        # final_enc: ELU -> SN conv1x1 -> ELU, encoder.py:58-66
        f = ops.unary(ctx, x, L.OP_ELU)
        f = ops.conv2d(ctx, f, self.final_conv)
        f = ops.unary(ctx, f, L.OP_ELU)
        return enc_dec_combiners, f
