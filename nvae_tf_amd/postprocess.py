"""Postprocess tower, mirroring the reference's postprocess.py.  89.7 % of the forward MACs of the
default config are the six dense 5x5 convolutions in here (groups=1: postprocess.py:74-76)."""
from __future__ import annotations

from . import _lib as L
from . import ops
from .common import Rescaler, RescaleType, SqueezeExcitation
from .ops import Ctx, Var


class ConvBNSwish:
    """postprocess.py:91-111: SN conv (no bias, dense) -> BN -> Swish."""

    def __init__(self, ps, name_conv: str, name_bn: str, in_channels: int, n_channels: int, kernel_size: int):
        self.conv = ps.conv(name_conv, kernel_size, in_channels, n_channels, bias=False)
        self.bn = ps.bn(name_bn, n_channels)

    def __call__(self, ctx: Ctx, x: Var) -> Var:
        x = ops.conv2d(ctx, x, self.conv, bias=False, stats_bn=self.bn)
        return ops.bn_act(ctx, x, self.bn, L.ACT_SWISH)


class PostprocessNode:
    """postprocess.py:61-88 (SE is applied by the owning cell together with the residual add)."""

    def __init__(self, ps, name: str, in_channels: int, n_channels: int, scale_factor: int, upscale: bool,
                 expansion_ratio: int = 6):
        self.up = Rescaler(ps, name + ".up", in_channels, n_channels, scale_factor, RescaleType.UP) if upscale else None
        if self.up is not None:
            self.up.feeds_bn = True        # node path: Rescaler -> bn0
        self.bn0 = ps.bn(name + ".bn0", n_channels)
        if self.up is not None:
            self.up.stats_bn = self.bn0
        hidden = n_channels * expansion_ratio
        self.cbs1 = ConvBNSwish(ps, name + ".conv1", name + ".bn1", n_channels, hidden, 1)
        self.cbs5 = ConvBNSwish(ps, name + ".conv5", name + ".bn2", hidden, hidden, 5)
        self.conv3 = ps.conv(name + ".conv3", 1, hidden, n_channels, bias=False)
        self.bn3 = ps.bn(name + ".bn3", n_channels)
        self.se = SqueezeExcitation(ps, name + ".se", n_channels)

    def __call__(self, ctx: Ctx, x: Var, skip: Var) -> Var:
        if self.up is not None:
            x = self.up(ctx, x)
        x = ops.bn_act(ctx, x, self.bn0)
        x = self.cbs1(ctx, x)
        x = self.cbs5(ctx, x)
        x = ops.conv2d(ctx, x, self.conv3, bias=False, stats_bn=self.bn3)
        x = ops.bn_act(ctx, x, self.bn3)       # applied inside the SE kernel
        return self.se(ctx, x, skip, 1.0, 0.1)      # skip + 0.1 * sequence, postprocess.py:58


class PostprocessCell:
    """postprocess.py:37-58 (n_nodes = 1 at every call site, postprocess.py:21)."""

    def __init__(self, ps, name: str, in_channels: int, n_channels: int, n_nodes: int, scale_factor: int,
                 upscale: bool = False):
        assert n_nodes == 1
        self.skip = Rescaler(ps, name + ".skip", in_channels, n_channels, scale_factor, RescaleType.UP) if upscale else None
        self.node = PostprocessNode(ps, name, in_channels, n_channels, scale_factor, upscale)

    def __call__(self, ctx: Ctx, inputs: Var) -> Var:
        skip = inputs if self.skip is None else self.skip(ctx, inputs)
        return self.node(ctx, inputs, skip)


class Postprocess:
    """postprocess.py:8-34."""

    def __init__(self, ps, n_blocks, n_cells, mult, n_channels_decoder, scale_factor, out_channels=1):
        self.cells = []
        idx = 0
        for _ in range(n_blocks):
            c_in = n_channels_decoder * mult
            mult //= scale_factor
            c = n_channels_decoder * mult
            for cell_idx in range(n_cells):
                up = cell_idx == 0
                self.cells.append(PostprocessCell(ps, f"post.cell{idx}", c_in if up else c, c, 1, scale_factor, up))
                idx += 1
        # a mixture-of-logistics head has 10*M (= 100) output channels: padded to a multiple of 8 so
        # that its data/weight gradients run on the MFMA kernels (params.ParamStore.conv)
        self.final_conv = ps.conv("post.final.conv", 3, n_channels_decoder * mult, out_channels,
                                  pad_cout=8 if out_channels > 8 else 1)
        self.mult = mult

    def __call__(self, ctx: Ctx, x: Var) -> Var:
        for cell in self.cells:
            x = cell(ctx, ops.grad_boundary(ctx, x))
        x = ops.unary(ctx, x, L.OP_ELU)                              # postprocess.py:27
        return ops.conv2d(ctx, x, self.final_conv, out_f32=True)     # logits stay f32
