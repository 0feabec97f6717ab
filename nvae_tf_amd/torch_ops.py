"""`torch.ops.nvae.*`: the hot-path kernel families registered with the PyTorch dispatcher (`torch.library`), each
with a fake (meta) implementation and an autograd formula whose backward is again HIP kernels of `libnvae_hip.so`.

north_star / SURVEY 8b name "PyTorch-ROCm custom ops" as the upper half of the boundary.  The training step itself
does not go through the dispatcher (nvae_tf_amd/ops.py drives the same C ABI from its own tape, which is what lets a
whole step be three hipGraph replays); these registrations make the same kernels usable from a stock torch autograd
loop, `torch.compile` graphs (fake impls) and `torch.library.opcheck`:

  nvae::conv2d_same(x, w, bias?, stride, upsample)  dense K x K (1, 3, 5) TF-'same' conv, NHWC x HWIO, stride 1 or 2
                                                 (Rescaler DOWN, common.py:156-162) or behind a nearest upsample by 2
                                                 that is never materialised (Rescaler UP, common.py:150-155)  (Conv2D,
                                                 postprocess.py:96-105, encoder.py:92-98)
  nvae::dwconv5(x, w, bias)                      depthwise 5 x 5 'same'                     (decoder.py:130)
  nvae::bn_act(x, gamma, beta, act, eps)         batch-statistics BatchNorm (+ Swish)       (decoder.py:125-146)
  nvae::se_residual(x, skip, w1, b1, w2, b2,     y = skip_scale*skip + branch_scale * x * sigmoid(FC2(relu(FC1(mean x))))
                    skip_scale, branch_scale)                                               (common.py:127-142)
  nvae::bernoulli_nll(logits, x)                 per-image -log p(x | logits)               (models.py:242-250)
  nvae::gauss_sample_kl(enc_p, dec_p?, eps)      z = mu_q + sigma_q * eps and KL(q || p) per image of one latent group,
                                                 soft-clamped residual parameterisation  (common.py:76-102, models.py:197-201)
  nvae::kl_balance(kl_all, alphas)               per-group balancing coefficients         (models.py:203-213; no gradient)
  nvae::bn_gamma_absmax(params, table, lambda)   lambda * sum over layers of max |gamma|  (models.py:252-267) + subgradient
  nvae::spectral_norm_step(w, u)                 one TFA power iteration: (sigma, u')     (SpectralNormalization, common.py:8)
  nvae::adamax_step(p, g, m, u, lr_t, b1, b2, eps)  Keras Adamax update in place          (train.py:131)

All tensors are NHWC on the GPU, activations bf16, f16 or f32, parameters f32.  No CPU implementation is registered: on a
CPU tensor the dispatcher raises, like every other entry into this package."""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from ._lib import call, ptr

_lib = torch.library.Library("nvae", "DEF")
_lib.define("conv2d_same(Tensor x, Tensor w, Tensor? bias, int stride=1, int upsample=1) -> Tensor")
_lib.define("conv2d_same_backward(Tensor dy, Tensor x, Tensor w, bool need_dx, int stride, int upsample) -> (Tensor, Tensor, Tensor)")
_lib.define("gauss_sample_kl(Tensor enc_p, Tensor? dec_p, Tensor eps) -> (Tensor, Tensor)")
_lib.define("gauss_sample_kl_backward(Tensor enc_p, Tensor? dec_p, Tensor eps, Tensor dz, Tensor dkl) -> (Tensor, Tensor)")
_lib.define("kl_balance(Tensor kl_all, Tensor alphas) -> Tensor")
_lib.define("bn_gamma_absmax(Tensor params, Tensor table, float lam) -> (Tensor, Tensor)")
_lib.define("bn_gamma_absmax_backward(Tensor params, Tensor table, Tensor argmax, float lam) -> Tensor")
_lib.define("spectral_norm_step(Tensor w, Tensor u) -> (Tensor, Tensor)")
_lib.define("adamax_step(Tensor(a!) p, Tensor g, Tensor(b!) m, Tensor(c!) u, float lr_t, float beta1, float beta2, float eps) -> ()")
_lib.define("dwconv5(Tensor x, Tensor w, Tensor bias) -> Tensor")
_lib.define("dwconv5_backward(Tensor dy, Tensor x, Tensor w) -> (Tensor, Tensor, Tensor)")
_lib.define("bn_act(Tensor x, Tensor gamma, Tensor beta, int act, float eps) -> (Tensor, Tensor, Tensor)")
_lib.define("bn_act_backward(Tensor dy, Tensor x, Tensor gamma, Tensor beta, Tensor mean, Tensor invstd, int act) -> (Tensor, Tensor, Tensor)")
_lib.define("se_residual(Tensor x, Tensor skip, Tensor w1, Tensor b1, Tensor w2, Tensor b2, float skip_scale, float branch_scale) -> (Tensor, Tensor, Tensor, Tensor)")
_lib.define("se_residual_backward(Tensor dy, Tensor x, Tensor w1, Tensor w2, Tensor pooled, Tensor gate, Tensor hidden, float skip_scale, float branch_scale) -> (Tensor, Tensor, Tensor, Tensor, Tensor, Tensor)")
_lib.define("bernoulli_nll(Tensor logits, Tensor x) -> Tensor")
_lib.define("bernoulli_nll_backward(Tensor logits, Tensor x, float scale) -> Tensor")


def _dt(t: Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return L.BF16
    if t.dtype == torch.float32:
        return L.F32
    if t.dtype == torch.float16:
        return L.F16
    raise TypeError(f"nvae ops take bf16, f16 or f32 activations, got {t.dtype}")


def _check_nhwc(x: Tensor, name: str):
    if x.dim() != 4 or not x.is_cuda:
        raise ValueError(f"{name}: expected a 4-d NHWC tensor on the GPU")


def _stat_dtype(x: Tensor) -> torch.dtype:
    return torch.float64 if x.dtype == torch.float32 else torch.float32     # include/nvae_hip.h "STATISTICS SLABS"


# ----------------------------------------------------------------------------------------------- conv2d_same
def _same_pad(n: int, k: int, s: int) -> int:
    """TF padding='same': low pad = total // 2 (a stride-2 3x3 on an even size pads bottom / right only)."""
    out = -(-n // s)
    return max((out - 1) * s + k - n, 0) // 2


def _conv_geoms(B, H, W, cin, cout, k, stride, up):
    Hu, Wu = H * up, W * up
    Ho, Wo = -(-Hu // stride), -(-Wu // stride)
    pt, pl = _same_pad(Hu, k, stride), _same_pad(Wu, k, stride)
    fwd = L.ConvGeom(B, H, W, cin, Ho, Wo, cout, k, k, stride, pt, pl, up, 0, cin, cout, cout)
    # data gradient: the forward kernel on dy, gradient dilation = forward stride, into the UPSAMPLED input grid
    dgrad = L.ConvGeom(B, Ho, Wo, cout, Hu, Wu, cin, k, k, 1, k - 1 - pt, k - 1 - pl, stride, 1, cout, cin, cin)
    return fwd, dgrad, (Ho, Wo), (Hu, Wu)


def _conv_fwd(x: Tensor, w: Tensor, bias: Optional[Tensor], stride: int = 1, upsample: int = 1) -> Tensor:
    _check_nhwc(x, "conv2d_same")
    k, k2, cin, cout = w.shape
    B, H, W, Cx = x.shape
    ve = 16 // x.element_size()
    if k != k2 or k not in (1, 3, 5) or Cx != cin or cin % ve or cout % ve or w.dtype != torch.float32:
        raise ValueError("conv2d_same: w must be f32 [k,k,Cin,Cout], k in {1,3,5}, channels multiples of 16 bytes")
    if stride not in (1, 2) or upsample not in (1, 2) or (stride == 2 and upsample == 2):
        raise ValueError("conv2d_same: stride and upsample are 1 or 2, not both 2")
    x = x.contiguous()
    # B operand of the forward GEMM: wF[co][(kh*K + kw)*Cin + ci] in the activation dtype (nvae_weight_prep's layout)
    wF = w.permute(3, 0, 1, 2).reshape(cout, k * k * cin).to(x.dtype).contiguous()
    g, _, (Ho, Wo), _ = _conv_geoms(B, H, W, cin, cout, k, stride, upsample)
    y = torch.empty((B, Ho, Wo, cout), dtype=x.dtype, device=x.device)
    b = bias.float().contiguous() if bias is not None else None
    call("nvae_conv_gemm", _dt(x), C.byref(g), ptr(x), ptr(wF), k * k * cin, ptr(b), None, ptr(y), 0, None)
    return y


def _conv_bwd(dy: Tensor, x: Tensor, w: Tensor, need_dx: bool, stride: int, upsample: int) -> Tuple[Tensor, Tensor, Tensor]:
    k, _, cin, cout = w.shape
    B, H, W, _ = x.shape
    dy, x = dy.contiguous(), x.contiguous()
    dt = _dt(x)
    gw, gd, _, (Hu, Wu) = _conv_geoms(B, H, W, cin, cout, k, stride, upsample)
    dx = torch.empty(0, device=x.device, dtype=x.dtype)
    if need_dx:
        # data gradient = the forward kernel on dy with tap-flipped, transposed weights: wD[ci][(kh',kw'),co]
        wD = w.flip(0, 1).permute(2, 0, 1, 3).reshape(cin, k * k * cout).to(x.dtype).contiguous()
        du = torch.empty((B, Hu, Wu, cin), dtype=x.dtype, device=x.device)
        call("nvae_conv_gemm", dt, C.byref(gd), ptr(dy), ptr(wD), k * k * cout, None, None, ptr(du), 0, None)
        if upsample == 1:
            dx = du
        else:                       # backward of the nearest upsample: sum over each 2 x 2 block
            dx = torch.empty_like(x)
            call("nvae_upsample_pool_bwd", dt, ptr(du), ptr(dx), B, H, W, cin, upsample, 0)
    dw = torch.zeros_like(w)
    db = torch.zeros(cout, dtype=torch.float32, device=x.device)
    n = int(L.load().nvae_conv_wgrad_scratch(dt, C.byref(gw)))
    scratch = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
    call("nvae_conv_wgrad", dt, C.byref(gw), ptr(x), ptr(dy), ptr(dw), cout, ptr(db), ptr(scratch), n)
    return dx, dw, db


_lib.impl("conv2d_same", _conv_fwd, "CUDA")
_lib.impl("conv2d_same_backward", _conv_bwd, "CUDA")


@torch.library.register_fake("nvae::conv2d_same")
def _(x, w, bias, stride=1, upsample=1):
    B, H, W, _ = x.shape
    return x.new_empty((B, -(-H * upsample // stride), -(-W * upsample // stride), w.shape[3]))


@torch.library.register_fake("nvae::conv2d_same_backward")
def _(dy, x, w, need_dx, stride, upsample):
    return (torch.empty_like(x) if need_dx else x.new_empty(0)), torch.empty_like(w), w.new_empty(w.shape[3])


def _conv_setup(ctx, inputs, output):
    x, w, bias, stride, upsample = inputs
    ctx.save_for_backward(x, w)
    ctx.has_bias = bias is not None
    ctx.geom = (stride, upsample)


def _conv_autograd(ctx, dy):
    x, w = ctx.saved_tensors
    dx, dw, db = torch.ops.nvae.conv2d_same_backward(dy, x, w, ctx.needs_input_grad[0], *ctx.geom)
    return (dx if ctx.needs_input_grad[0] else None), dw, (db if ctx.has_bias else None), None, None


torch.library.register_autograd("nvae::conv2d_same", _conv_autograd, setup_context=_conv_setup)


# ----------------------------------------------------------------------------------------------- dwconv5
def _dw_fwd(x: Tensor, w: Tensor, bias: Tensor) -> Tensor:
    _check_nhwc(x, "dwconv5")
    B, H, W, Cc = x.shape
    if tuple(w.shape) != (5, 5, Cc) or w.dtype != torch.float32 or Cc % 8:
        raise ValueError("dwconv5: w must be f32 [5,5,C] with C a multiple of 8")
    x = x.contiguous()
    y = torch.empty_like(x)
    call("nvae_dwconv5", _dt(x), ptr(x), ptr(w.contiguous()), ptr(bias.float().contiguous()), ptr(y), B, H, W, Cc, 0, 0)
    return y


def _dw_bwd(dy: Tensor, x: Tensor, w: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    B, H, W, Cc = x.shape
    dy, x, w = dy.contiguous(), x.contiguous(), w.contiguous()
    dx = torch.empty_like(x)
    call("nvae_dwconv5", _dt(x), ptr(dy), ptr(w), None, ptr(dx), B, H, W, Cc, 1, 0)      # flipped taps
    dw = torch.zeros_like(w)
    db = torch.zeros(Cc, dtype=torch.float32, device=x.device)
    call("nvae_dwconv5_wgrad", _dt(x), ptr(x), ptr(dy), ptr(dw), ptr(db), B, H, W, Cc)
    return dx, dw, db


_lib.impl("dwconv5", _dw_fwd, "CUDA")
_lib.impl("dwconv5_backward", _dw_bwd, "CUDA")


@torch.library.register_fake("nvae::dwconv5")
def _(x, w, bias):
    return torch.empty_like(x)


@torch.library.register_fake("nvae::dwconv5_backward")
def _(dy, x, w):
    return torch.empty_like(x), torch.empty_like(w), w.new_empty(w.shape[2])


def _dw_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _dw_autograd(ctx, dy):
    x, w = ctx.saved_tensors
    return torch.ops.nvae.dwconv5_backward(dy, x, w)


torch.library.register_autograd("nvae::dwconv5", _dw_autograd, setup_context=_dw_setup)


# ----------------------------------------------------------------------------------------------- bn_act
def _bn_fwd(x: Tensor, gamma: Tensor, beta: Tensor, act: int, eps: float) -> Tuple[Tensor, Tensor, Tensor]:
    _check_nhwc(x, "bn_act")
    Cc = x.shape[3]
    rows = x.numel() // Cc
    x = x.contiguous()
    dev = x.device
    S = int(L.load().nvae_reduce_splits(rows, Cc))
    slab = torch.empty((S, 2, Cc), dtype=_stat_dtype(x), device=dev)
    coef = torch.empty((4, Cc), dtype=torch.float32, device=dev)          # scale, shift, mean, invstd
    rm = torch.zeros(Cc, dtype=torch.float32, device=dev)                 # moving statistics are the caller's business
    rv = torch.ones(Cc, dtype=torch.float32, device=dev)
    dt = _dt(x)
    call("nvae_bn_stats", dt, ptr(x), rows, Cc, ptr(slab))
    call("nvae_bn_finalize", dt, ptr(slab), rows, Cc, ptr(gamma.contiguous()), ptr(beta.contiguous()), ptr(rm), ptr(rv),
         0.0, float(eps), ptr(coef[0]), ptr(coef[1]), ptr(coef[2]), ptr(coef[3]))
    y = torch.empty_like(x)
    call("nvae_bn_apply", dt, ptr(x), ptr(y), rows, Cc, ptr(coef[0]), ptr(coef[1]), int(act))
    return y, coef[2].clone(), coef[3].clone()


def _bn_bwd(dy: Tensor, x: Tensor, gamma: Tensor, beta: Tensor, mean: Tensor, invstd: Tensor, act: int):
    Cc = x.shape[3]
    rows = x.numel() // Cc
    dy, x = dy.contiguous(), x.contiguous()
    dev, dt = x.device, _dt(x)
    scale = (gamma * invstd).contiguous()
    shift = (beta - mean * scale).contiguous()
    S = int(L.load().nvae_reduce_splits(rows, Cc))
    part = torch.empty((S, 2, Cc), dtype=_stat_dtype(x), device=dev)
    call("nvae_bn_bwd_reduce", dt, ptr(x), ptr(dy), rows, Cc, ptr(scale), ptr(shift), int(act), ptr(part))
    dgamma = torch.zeros(Cc, dtype=torch.float32, device=dev)
    dbeta = torch.zeros(Cc, dtype=torch.float32, device=dev)
    k0k1 = torch.empty((2, Cc), dtype=torch.float32, device=dev)
    call("nvae_bn_bwd_finalize", dt, ptr(part), rows, Cc, ptr(scale), ptr(mean.contiguous()), ptr(invstd.contiguous()),
         ptr(dgamma), ptr(dbeta), ptr(k0k1), 0)
    dx = torch.empty_like(x)
    call("nvae_bn_bwd_apply", dt, ptr(x), ptr(dy), ptr(dx), rows, Cc, ptr(scale), ptr(shift), ptr(k0k1), int(act), 0)
    return dx, dgamma, dbeta


_lib.impl("bn_act", _bn_fwd, "CUDA")
_lib.impl("bn_act_backward", _bn_bwd, "CUDA")


@torch.library.register_fake("nvae::bn_act")
def _(x, gamma, beta, act, eps):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


@torch.library.register_fake("nvae::bn_act_backward")
def _(dy, x, gamma, beta, mean, invstd, act):
    return torch.empty_like(x), torch.empty_like(gamma), torch.empty_like(gamma)


def _bn_setup(ctx, inputs, output):
    x, gamma, beta, act, eps = inputs
    ctx.save_for_backward(x, gamma, beta, output[1], output[2])
    ctx.act = act
    ctx.mark_non_differentiable(output[1], output[2])


def _bn_autograd(ctx, dy, _dmean, _dinvstd):
    x, gamma, beta, mean, invstd = ctx.saved_tensors
    dx, dgamma, dbeta = torch.ops.nvae.bn_act_backward(dy, x, gamma, beta, mean, invstd, ctx.act)
    return dx, dgamma, dbeta, None, None


torch.library.register_autograd("nvae::bn_act", _bn_autograd, setup_context=_bn_setup)


# ----------------------------------------------------------------------------------------------- se_residual
def _se_fwd(x, skip, w1, b1, w2, b2, skip_scale: float, branch_scale: float):
    _check_nhwc(x, "se_residual")
    B, H, W, Cc = x.shape
    Hd = w1.shape[1]
    if tuple(w1.shape) != (Cc, Hd) or tuple(w2.shape) != (Hd, Cc) or Cc > 2048 or Cc % 8:
        raise ValueError("se_residual: w1 [C,Hd], w2 [Hd,C] f32, C a multiple of 8, C <= 2048")
    x, skip = x.contiguous(), skip.contiguous()
    dev = x.device
    pooled = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    gate = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    hidden = torch.empty((B, Hd), dtype=torch.float32, device=dev)
    dt = _dt(x)
    call("nvae_se_pool_gate", dt, ptr(x), B, H * W, Cc, Hd, ptr(w1.contiguous()), ptr(b1.contiguous()),
         ptr(w2.contiguous()), ptr(b2.contiguous()), ptr(pooled), ptr(gate), ptr(hidden))
    y = torch.empty_like(x)
    call("nvae_se_apply", dt, ptr(x), ptr(skip), ptr(y), B, H * W, Cc, ptr(gate), float(skip_scale), float(branch_scale))
    return y, pooled, gate, hidden


def _se_bwd(dy, x, w1, w2, pooled, gate, hidden, skip_scale: float, branch_scale: float):
    B, H, W, Cc = x.shape
    Hd = w1.shape[1]
    dy, x = dy.contiguous(), x.contiguous()
    dev, dt = x.device, _dt(x)
    dpool = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    scratch = torch.empty((B, Cc + Hd), dtype=torch.float32, device=dev)
    call("nvae_se_reduce_gate_bwd", dt, ptr(x), ptr(dy), ptr(gate), ptr(hidden), B, H * W, Cc, Hd, ptr(w1.contiguous()),
         ptr(w2.contiguous()), float(branch_scale), ptr(dpool), ptr(scratch))
    dw1, dw2 = torch.zeros_like(w1), torch.zeros_like(w2)
    db1 = torch.zeros(Hd, dtype=torch.float32, device=dev)
    db2 = torch.zeros(Cc, dtype=torch.float32, device=dev)
    call("nvae_se_wgrad", ptr(pooled), ptr(hidden), ptr(scratch), B, H * W, Cc, Hd, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2))
    dx, dskip = torch.empty_like(x), torch.empty_like(x)
    call("nvae_se_bwd_apply", dt, ptr(dy), ptr(gate), ptr(dpool), ptr(dx), ptr(dskip), B, H * W, Cc, float(skip_scale),
         float(branch_scale), 0, 0)
    return dx, dskip, dw1, db1, dw2, db2


_lib.impl("se_residual", _se_fwd, "CUDA")
_lib.impl("se_residual_backward", _se_bwd, "CUDA")


@torch.library.register_fake("nvae::se_residual")
def _(x, skip, w1, b1, w2, b2, skip_scale, branch_scale):
    B, Cc, Hd = x.shape[0], x.shape[3], w1.shape[1]
    return torch.empty_like(x), w1.new_empty((B, Cc)), w1.new_empty((B, Cc)), w1.new_empty((B, Hd))


@torch.library.register_fake("nvae::se_residual_backward")
def _(dy, x, w1, w2, pooled, gate, hidden, skip_scale, branch_scale):
    return (torch.empty_like(x), torch.empty_like(x), torch.empty_like(w1), w1.new_empty(w1.shape[1]),
            torch.empty_like(w2), w1.new_empty(w1.shape[0]))


def _se_setup(ctx, inputs, output):
    x, skip, w1, b1, w2, b2, ss, bs = inputs
    ctx.save_for_backward(x, w1, w2, output[1], output[2], output[3])
    ctx.scales = (ss, bs)
    ctx.mark_non_differentiable(output[1], output[2], output[3])


def _se_autograd(ctx, dy, *_):
    x, w1, w2, pooled, gate, hidden = ctx.saved_tensors
    dx, dskip, dw1, db1, dw2, db2 = torch.ops.nvae.se_residual_backward(dy, x, w1, w2, pooled, gate, hidden, *ctx.scales)
    return dx, dskip, dw1, db1, dw2, db2, None, None


torch.library.register_autograd("nvae::se_residual", _se_autograd, setup_context=_se_setup)


# ----------------------------------------------------------------------------------------------- bernoulli_nll
def _bern_fwd(logits: Tensor, x: Tensor) -> Tensor:
    _check_nhwc(logits, "bernoulli_nll")
    if logits.dtype != torch.float32 or x.shape != logits.shape:
        raise ValueError("bernoulli_nll: f32 logits [B,H,W,C] and data of the same shape (bf16 or f32)")
    B, H, W, Cc = logits.shape
    out = torch.zeros(B, dtype=torch.float32, device=logits.device)
    call("nvae_bernoulli_fwd", _dt(x), ptr(logits.contiguous()), ptr(x.contiguous()), ptr(out), B, H, W, Cc, 0)
    return out


def _bern_bwd(logits: Tensor, x: Tensor, scale: float) -> Tensor:
    g = torch.empty(logits.shape, dtype=x.dtype, device=logits.device)      # the kernel writes the activation dtype
    call("nvae_bernoulli_bwd", _dt(x), ptr(logits.contiguous()), ptr(x.contiguous()), ptr(g), logits.numel(), float(scale), None)
    return g.float()


_lib.impl("bernoulli_nll", _bern_fwd, "CUDA")
_lib.impl("bernoulli_nll_backward", _bern_bwd, "CUDA")


@torch.library.register_fake("nvae::bernoulli_nll")
def _(logits, x):
    return logits.new_empty(logits.shape[0])


@torch.library.register_fake("nvae::bernoulli_nll_backward")
def _(logits, x, scale):
    return torch.empty_like(logits)


def _bern_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _bern_autograd(ctx, dnll):
    logits, x = ctx.saved_tensors
    # the kernel scales by one scalar (the training step's 1/B); a per-image upstream gradient is applied on top
    g = torch.ops.nvae.bernoulli_nll_backward(logits, x, 1.0)
    return g * dnll.view(-1, 1, 1, 1), None


torch.library.register_autograd("nvae::bernoulli_nll", _bern_autograd, setup_context=_bern_setup)


# ----------------------------------------------------------------------------------------------- gauss_sample_kl
def _gs_fwd(enc_p: Tensor, dec_p: Optional[Tensor], eps: Tensor) -> Tuple[Tensor, Tensor]:
    _check_nhwc(enc_p, "gauss_sample_kl")
    B, H, W, L2 = enc_p.shape
    Lc = L2 // 2
    if enc_p.dtype != torch.float32 or eps.dtype != torch.float32 or tuple(eps.shape) != (B, H, W, Lc) or \
            (dec_p is not None and (dec_p.shape != enc_p.shape or dec_p.dtype != torch.float32)):
        raise ValueError("gauss_sample_kl: f32 enc_p / dec_p [B,H,W,2L] and eps [B,H,W,L]")
    z = torch.empty((B, H, W, Lc), dtype=torch.float32, device=enc_p.device)
    kl = torch.empty(B, dtype=torch.float32, device=enc_p.device)
    call("nvae_sampler_fwd", L.F32, ptr(enc_p.contiguous()), ptr(dec_p.contiguous()) if dec_p is not None else None,
         ptr(eps.contiguous()), ptr(z), ptr(kl), None, None, None, B, H * W, Lc)
    return z, kl


def _gs_bwd(enc_p: Tensor, dec_p: Optional[Tensor], eps: Tensor, dz: Tensor, dkl: Tensor) -> Tuple[Tensor, Tensor]:
    """The kernel weighs the KL term by ONE scalar (the training step's beta * coefficient / B); an arbitrary per-image
    upstream gradient dkl[B] is served by two launches - the dz path alone (weight 0) and the KL path alone (weight 1,
    dz = NULL) - combined per image."""
    B, H, W, L2 = enc_p.shape
    Lc = L2 // 2
    dev = enc_p.device
    enc_p, eps, dz = enc_p.contiguous(), eps.contiguous(), dz.contiguous()
    dp = dec_p.contiguous() if dec_p is not None else None
    hyper = torch.zeros(L.HY_SIZE, dtype=torch.float32, device=dev)
    one = torch.ones(1, dtype=torch.float32, device=dev)
    outs = []
    for beta, dzp in ((0.0, ptr(dz)), (1.0, None)):
        hyper[L.HY_BETA] = beta
        de = torch.empty_like(enc_p)
        dd = torch.empty_like(enc_p) if dp is not None else None
        call("nvae_sampler_bwd", L.F32, ptr(enc_p), ptr(dp), ptr(eps), dzp, ptr(one), ptr(hyper), 1.0, ptr(de), ptr(dd),
             B, H * W, Lc)
        outs.append((de, dd))
    wgt = dkl.view(B, 1, 1, 1)
    d_enc = outs[0][0] + wgt * outs[1][0]
    d_dec = (outs[0][1] + wgt * outs[1][1]) if dp is not None else enc_p.new_empty(0)
    return d_enc, d_dec


_lib.impl("gauss_sample_kl", _gs_fwd, "CUDA")
_lib.impl("gauss_sample_kl_backward", _gs_bwd, "CUDA")


@torch.library.register_fake("nvae::gauss_sample_kl")
def _(enc_p, dec_p, eps):
    return torch.empty_like(eps), enc_p.new_empty(enc_p.shape[0])


@torch.library.register_fake("nvae::gauss_sample_kl_backward")
def _(enc_p, dec_p, eps, dz, dkl):
    return torch.empty_like(enc_p), (torch.empty_like(enc_p) if dec_p is not None else enc_p.new_empty(0))


def _gs_setup(ctx, inputs, output):
    enc_p, dec_p, eps = inputs
    ctx.has_dec = dec_p is not None
    ctx.save_for_backward(enc_p, eps, *([dec_p] if dec_p is not None else []))


def _gs_autograd(ctx, dz, dkl):
    saved = ctx.saved_tensors
    enc_p, eps = saved[0], saved[1]
    dec_p = saved[2] if ctx.has_dec else None
    d_enc, d_dec = torch.ops.nvae.gauss_sample_kl_backward(enc_p, dec_p, eps, dz, dkl)
    return d_enc, (d_dec if ctx.has_dec else None), None      # (eps is noise: no gradient)


torch.library.register_autograd("nvae::gauss_sample_kl", _gs_autograd, setup_context=_gs_setup)


# ----------------------------------------------------------------------------------------------- kl_balance
def _klb(kl_all: Tensor, alphas: Tensor) -> Tensor:
    """models.py:203-213: coefficient_g = mean_b|KL_g| / alpha_g * total / sum(...), normalised to mean 1 (the kernels
    nvae_kl_absmean + nvae_loss_finalize with balancing on); a constant in the reference (stop_gradient)."""
    G, B = kl_all.shape
    dev = kl_all.device
    kl_all = kl_all.float().contiguous()
    hyper = torch.zeros(L.HY_SIZE, dtype=torch.float32, device=dev)
    hyper[L.HY_BETA], hyper[L.HY_BALANCE] = 1.0, 1.0
    am = torch.empty(G, dtype=torch.float32, device=dev)
    coeff = torch.empty(G, dtype=torch.float32, device=dev)
    out = torch.empty(B, dtype=torch.float32, device=dev)
    res = torch.empty(L.RES_SIZE, dtype=torch.float32, device=dev)
    zero = torch.zeros(B, dtype=torch.float32, device=dev)
    call("nvae_kl_absmean", ptr(kl_all), G, B, ptr(am))
    call("nvae_loss_finalize", ptr(kl_all), ptr(am), ptr(alphas.float().contiguous()), G, B, ptr(zero), None, ptr(hyper),
         ptr(coeff), ptr(out), ptr(res))
    return coeff


_lib.impl("kl_balance", _klb, "CUDA")


@torch.library.register_fake("nvae::kl_balance")
def _(kl_all, alphas):
    return alphas.new_empty(alphas.shape)


# ----------------------------------------------------------------------------------------------- bn_gamma_absmax
def _bnl_fwd(params: Tensor, table: Tensor, lam: float) -> Tuple[Tensor, Tensor]:
    """params: the flat f32 parameter buffer; table [n,2] int32 = (offset of gamma, channels) per counted layer."""
    n = table.shape[0]
    out = torch.zeros(1, dtype=torch.float32, device=params.device)
    argmax = torch.zeros(n, dtype=torch.int32, device=params.device)
    call("nvae_bn_absmax_fwd", ptr(params), ptr(table.contiguous()), n, float(lam), ptr(out), ptr(argmax))
    return out[0], argmax


def _bnl_bwd(params: Tensor, table: Tensor, argmax: Tensor, lam: float) -> Tensor:
    grads = torch.zeros_like(params)
    hyper = torch.zeros(L.HY_SIZE, dtype=torch.float32, device=params.device)      # loss scale 0 -> 1
    call("nvae_bn_absmax_bwd", ptr(params), ptr(grads), ptr(table.contiguous()), ptr(argmax), table.shape[0], float(lam),
         ptr(hyper))
    return grads


_lib.impl("bn_gamma_absmax", _bnl_fwd, "CUDA")
_lib.impl("bn_gamma_absmax_backward", _bnl_bwd, "CUDA")


@torch.library.register_fake("nvae::bn_gamma_absmax")
def _(params, table, lam):
    return params.new_empty(()), table.new_empty(table.shape[0])


@torch.library.register_fake("nvae::bn_gamma_absmax_backward")
def _(params, table, argmax, lam):
    return torch.empty_like(params)


def _bnl_setup(ctx, inputs, output):
    params, table, lam = inputs
    ctx.save_for_backward(params, table, output[1])
    ctx.lam = lam
    ctx.mark_non_differentiable(output[1])


def _bnl_autograd(ctx, dloss, _):
    params, table, argmax = ctx.saved_tensors
    return torch.ops.nvae.bn_gamma_absmax_backward(params, table, argmax, ctx.lam) * dloss, None, None


torch.library.register_autograd("nvae::bn_gamma_absmax", _bnl_autograd, setup_context=_bnl_setup)


# ----------------------------------------------------------------------------------------------- spectral_norm_step
def _sn_step(w: Tensor, u: Tensor) -> Tuple[Tensor, Tensor]:
    """One TFA SpectralNormalization power iteration on a conv kernel w [kh,kw,Cin,Cout] (f32) with u [Cout]:
    v = l2n(u W^T), u' = l2n(v W), sigma = v W u'^T -> (sigma, u').  W itself is left alone (the training step divides it
    by sigma inside nvae_weight_prep)."""
    kh, kw, cin, cout = w.shape
    K = kh * kw * cin
    dev = w.device
    d = (L.ConvDesc * 1)()
    d[0].w_off, d[0].wf_off, d[0].wd_off = 0, 0, -1
    d[0].u_off, d[0].t_off, d[0].K, d[0].Cout, d[0].Cin, d[0].taps = 0, 0, K, cout, cin, kh * kw
    d[0].wf_ld, d[0].wd_ld, d[0].idx, d[0].blk_off, d[0].p_off = (K + 7) // 8 * 8, 0, 0, 0, 0
    blocks = (K + 15) // 16
    descs = torch.frombuffer(bytearray(bytes(d)), dtype=torch.uint8).to(dev)
    state = u.detach().float().clone().contiguous()
    t = torch.zeros((K + 7) // 8 * 8, dtype=torch.float32, device=dev)
    colpart = torch.empty(blocks * cout, dtype=torch.float32, device=dev)
    w2 = torch.empty(cout, dtype=torch.float32, device=dev)
    inv_sigma = torch.empty(1, dtype=torch.float32, device=dev)
    call("nvae_sn_power_iter", ptr(w.detach().float().contiguous()), ptr(descs), 1, blocks, ptr(state), ptr(t), ptr(colpart),
         ptr(w2), ptr(inv_sigma))
    return 1.0 / inv_sigma[0], state


_lib.impl("spectral_norm_step", _sn_step, "CUDA")


@torch.library.register_fake("nvae::spectral_norm_step")
def _(w, u):
    return w.new_empty(()), torch.empty_like(u)


# ----------------------------------------------------------------------------------------------- adamax_step
def _adamax(p: Tensor, g: Tensor, m: Tensor, u: Tensor, lr_t: float, beta1: float, beta2: float, eps: float) -> None:
    """Keras Adamax on flat f32 buffers, in place: m = b1 m + (1-b1) g; u = max(b2 u, |g|); p -= lr_t m / (u + eps), with
    lr_t = lr / (1 - b1^t) supplied by the caller (models.NVAE._set_hyper)."""
    if not (p.is_contiguous() and m.is_contiguous() and u.is_contiguous()) or p.dtype != torch.float32:
        raise ValueError("adamax_step: contiguous f32 buffers")
    hyper = torch.zeros(L.HY_SIZE, dtype=torch.float32, device=p.device)
    hyper[L.HY_LR], hyper[L.HY_GSCALE], hyper[L.HY_LSCALE] = lr_t, 1.0, 1.0
    call("nvae_adamax", ptr(p), ptr(g.contiguous()), ptr(m), ptr(u), p.numel(), ptr(hyper), float(beta1), float(beta2), float(eps))


_lib.impl("adamax_step", _adamax, "CUDA")


@torch.library.register_fake("nvae::adamax_step")
def _(p, g, m, u, lr_t, beta1, beta2, eps):
    return None
