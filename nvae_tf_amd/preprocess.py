"""Preprocess tower, mirroring the reference's preprocess.py."""
from __future__ import annotations

from . import _lib as L
from . import ops
from .common import SqueezeExcitation
from .ops import Ctx, Var


class SkipScaler:
    """preprocess.py:42-74: Swish, then four SN conv1x1 stride-2 on spatially shifted views, channel
    concat.  Here: four gather-GEMMs with (negative) padding = shift, each writing its channel slice
    of one output tensor, so neither the shifted views nor the concat are materialised."""

    SHIFTS = [(0, 0), (1, 1), (0, 1), (1, 0)]   # (h, w) offsets of conv1..conv4, preprocess.py:68-71

    def __init__(self, ps, name: str, in_channels: int, n_channels: int):
        q = n_channels // 4
        outs = [q, q, q, n_channels - 3 * q]
        self.convs = [ps.conv(f"{name}.conv{i + 1}", 1, in_channels, co) for i, co in enumerate(outs)]
        self.n_channels = n_channels

    def __call__(self, ctx: Ctx, x: Var) -> Var:
        B, H, W, _ = x.t.shape
        o = ops.unary(ctx, x, L.OP_SWISH)
        out = Var(ctx.empty((B, (H + 1) // 2, (W + 1) // 2, self.n_channels)))
        coff = 0
        for conv, (sh, sw) in zip(self.convs, self.SHIFTS):
            # TF 'same' on the shifted (H - sh) view with stride 2 has zero padding and ceil((H-sh)/2)
            # rows; the unified output has ceil(H/2) rows, identical for even H.
            assert (H - sh + 1) // 2 == (H + 1) // 2 and (W - sw + 1) // 2 == (W + 1) // 2, \
                "SkipScaler needs even spatial extents"
            ops.conv2d(ctx, o, conv, stride=2, pad=(-sh, -sw), out=out, out_coff=coff,
                       out_hw=((H + 1) // 2, (W + 1) // 2))
            coff += conv.cout
        return out


class BNSwishConv:
    """preprocess.py:77-107: skip(x) + 0.1 * SE(nodes(x)), nodes = n x [BN, Swish, SN conv3x3]."""

    def __init__(self, ps, name: str, n_nodes: int, in_channels: int, n_channels: int, stride: int):
        self.stride = stride
        self.skip = SkipScaler(ps, name + ".skip", in_channels, n_channels) if stride == 2 else None
        self.bns, self.convs = [], []
        c = in_channels
        for i in range(n_nodes):
            self.bns.append(ps.bn(f"{name}.bn{i}", c))
            self.convs.append(ps.conv(f"{name}.conv{i}", 3, c, n_channels))
            c = n_channels
        self.se = SqueezeExcitation(ps, name + ".se", n_channels)

    def __call__(self, ctx: Ctx, x: Var) -> Var:
        y = x
        for i, (bn, conv) in enumerate(zip(self.bns, self.convs)):
            y = ops.bn_act(ctx, y, bn, L.ACT_SWISH)
            y = ops.conv2d(ctx, y, conv, stride=self.stride if i == 0 else 1,
                           stats_bn=self.bns[i + 1] if i + 1 < len(self.convs) else None)
        skipped = x if self.skip is None else self.skip(ctx, x)
        return self.se(ctx, y, skipped, 1.0, 0.1)


class Preprocess:
    """preprocess.py:7-39."""

    def __init__(self, ps, n_encoder_channels, n_blocks, n_cells, scale_factor, input_shape, mult=1):
        in_ch = int(input_shape[3])
        self.stem = ps.conv("pre.stem", 3, in_ch, n_encoder_channels)
        self.cells = []
        idx = 0
        shape = list(input_shape)
        for _ in range(n_blocks):
            for _ in range(n_cells - 1):
                c = mult * n_encoder_channels
                self.cells.append(BNSwishConv(ps, f"pre.cell{idx}", 2, c, c, 1)); idx += 1
            c_out = mult * n_encoder_channels * scale_factor
            self.cells.append(BNSwishConv(ps, f"pre.cell{idx}", 2, mult * n_encoder_channels, c_out, 2)); idx += 1
            mult *= scale_factor
            shape = [shape[0], shape[1] // scale_factor, shape[2] // scale_factor, shape[3] * scale_factor]
        self.mult = mult
        self.output_shape_ = shape

    def __call__(self, ctx: Ctx, inputs) -> Var:
        x = ops.affine(ctx, inputs, 2.0, -1.0)   # [0,1] -> [-1,1], preprocess.py:38-39
        x = ops.conv2d(ctx, x, self.stem)
        for cell in self.cells:
            x = cell(ctx, ops.grad_boundary(ctx, x))
        return x
