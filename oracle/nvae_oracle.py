"""CPU oracle for the NVAE hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain PyTorch-CPU restatement (fp64 master, fp32 for timing) of the reference's arithmetic
for the train/sample path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file; the product (nvae_tf_amd/) never does.

PARITY UNPINNED: the reference's arithmetic lives in TensorFlow 2.3 / TF-Addons / TF-Probability,
none of which are importable here, and the reference holds no tests, fixtures or golden vectors
for this path.  This restatement is therefore pinned only by the
known-answer tests derivable from the reference source (tests/test_oracle_kat.py).

Each function cites the reference file:line it restates (paths relative to the reference root).

Layout: activations NHWC, conv kernels HWIO ([kh, kw, c_in, c_out]) as in Keras, depthwise
kernels [5, 5, C], dense kernels [in, out].

Explicit deviations from the literal reference (SURVEY.md "Quirks"):
  Q1  `training` is an explicit flag (reference never passes it); training=True means batch-stat
      BN + one spectral-norm power iteration per conv per step, training=False means moving-stat
      BN and no spectral normalisation.
  Q9  no tf.squeeze of the sampler parameters (it would drop a batch of 1).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration (argument names follow NVAE.__init__, models.py:17-36)
# --------------------------------------------------------------------------------------
@dataclass
class OracleConfig:
    n_encoder_channels: int = 32
    n_decoder_channels: int = 32
    res_cells_per_group: int = 1
    n_preprocess_blocks: int = 2
    n_preprocess_cells: int = 3
    n_latent_per_group: int = 20
    n_groups_per_scale: List[int] = field(default_factory=lambda: [5, 10])
    n_postprocess_blocks: int = 2
    n_post_process_cells: int = 3
    sr_lambda: float = 0.01
    scale_factor: int = 2
    total_epochs: int = 400
    n_total_iterations: int = 400 * 417
    step_based_warmup: bool = True
    input_hw: int = 32
    input_channels: int = 1
    # output head: "bernoulli" (the reference, models.py:242-250) or "dmol" - the discretised mixture of
    # logistics of the NVAE paper, which the reference leaves unimplemented (train.py:219, README.md:25-27)
    head: str = "bernoulli"
    num_mixture_dec: int = 10

    @property
    def output_channels(self) -> int:
        if self.head == "bernoulli":
            return self.input_channels
        assert self.input_channels == 3, "the mixture-of-logistics head models RGB"
        return 10 * self.num_mixture_dec

    @property
    def n_latent_scales(self) -> int:
        return len(self.n_groups_per_scale)


BN_MOMENTUM = 0.05   # encoder.py:91 etc.; Keras meaning: keep 5 % of the old moving stat (Q2)
BN_EPS = 1e-5
SN_EPS = 1e-12       # tf.math.l2_normalize epsilon [3P]


# --------------------------------------------------------------------------------------
# L1 math helpers
# --------------------------------------------------------------------------------------
def softclamp5(x):
    """util.py:49-50."""
    return 5.0 * torch.tanh(x / 5.0)


def calculate_log_p(z, mu, sigma):
    """util.py:39-46."""
    nz = (z - mu) / sigma
    return -0.5 * nz * nz - 0.5 * math.log(2 * math.pi) - torch.log(sigma)


def swish(x):
    return x * torch.sigmoid(x)


def same_pad(in_size: int, k: int, s: int) -> Tuple[int, int]:
    """TF padding='same' (Q6): total = max((ceil(in/s)-1)*s + k - in, 0), low = total//2."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return total // 2, total - total // 2


def conv2d(x, w, b=None, stride=1):
    """Keras Conv2D(padding='same') on NHWC input with HWIO kernel."""
    kh, kw = w.shape[0], w.shape[1]
    pt, pb = same_pad(x.shape[1], kh, stride)
    pl, pr = same_pad(x.shape[2], kw, stride)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xn.contiguous(), w.permute(3, 2, 0, 1).contiguous(), b, stride=stride)
    return y.permute(0, 2, 3, 1)


def dwconv5(x, w, b):
    """Keras DepthwiseConv2D((5,5), padding='same'), decoder.py:130."""
    c = x.shape[3]
    xn = F.pad(x.permute(0, 3, 1, 2), (2, 2, 2, 2))
    y = F.conv2d(xn.contiguous(), w.permute(2, 0, 1).unsqueeze(1).contiguous(), b, groups=c)
    return y.permute(0, 2, 3, 1)


def upsample_nearest(x, f):
    """tf.image.resize(method='nearest') by an integer factor, common.py:168-172."""
    return x.repeat_interleave(f, dim=1).repeat_interleave(f, dim=2)


def glorot_uniform(shape, fan_in, fan_out, gen, dtype):
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return ((torch.rand(shape, generator=gen, dtype=torch.float64) * 2 - 1) * lim).to(dtype)


# --------------------------------------------------------------------------------------
# parameter store
# --------------------------------------------------------------------------------------
class Store:
    """name -> tensor.  `params` are trainable (Keras trainable_weights), `state` holds the
    spectral-norm `u` vectors and BN moving statistics."""

    def __init__(self, dtype, seed):
        self.dtype = dtype
        self.gen = torch.Generator().manual_seed(seed)
        self.params: Dict[str, torch.Tensor] = {}
        self.state: Dict[str, torch.Tensor] = {}
        self.sn_convs: List[str] = []          # names of SpectralNormalization-wrapped convs
        self.bn_loss_layers: List[str] = []    # BN layers counted by calculate_bn_loss

    def conv(self, name, k, ci, co, bias=True, sn=True):
        self.params[name + ".w"] = glorot_uniform((k, k, ci, co), k * k * ci, k * k * co,
                                                  self.gen, self.dtype)
        if bias:
            self.params[name + ".b"] = torch.zeros(co, dtype=self.dtype)
        if sn:
            # TFA SpectralNormalization: u ~ TruncatedNormal(stddev=0.02), shape (1, co) [3P]
            u = torch.randn(co, generator=self.gen, dtype=torch.float64).clamp(-2, 2) * 0.02
            self.state[name + ".u"] = u.to(self.dtype)
            self.sn_convs.append(name)

    def dw(self, name, c):
        # Keras fans for a (5,5,C,1) depthwise kernel: fan_in = 25*C, fan_out = 25 [3P]
        self.params[name + ".w"] = glorot_uniform((5, 5, c), 25 * c, 25, self.gen, self.dtype)
        self.params[name + ".b"] = torch.zeros(c, dtype=self.dtype)

    def bn(self, name, c, in_bn_loss=False):
        self.params[name + ".gamma"] = torch.ones(c, dtype=self.dtype)
        self.params[name + ".beta"] = torch.zeros(c, dtype=self.dtype)
        self.state[name + ".rm"] = torch.zeros(c, dtype=self.dtype)
        self.state[name + ".rv"] = torch.ones(c, dtype=self.dtype)
        if in_bn_loss:
            self.bn_loss_layers.append(name)

    def se(self, name, c):
        h = int(max(c / 16, 4))   # common.py:125
        self.params[name + ".w1"] = glorot_uniform((c, h), c, h, self.gen, self.dtype)
        self.params[name + ".b1"] = torch.zeros(h, dtype=self.dtype)
        self.params[name + ".w2"] = glorot_uniform((h, c), h, c, self.gen, self.dtype)
        self.params[name + ".b2"] = torch.zeros(c, dtype=self.dtype)


@dataclass
class DistributionParams:   # common.py:12-17
    enc_mu: torch.Tensor
    enc_sigma: torch.Tensor
    dec_mu: torch.Tensor
    dec_sigma: torch.Tensor


# --------------------------------------------------------------------------------------
# the model
# --------------------------------------------------------------------------------------
class OracleNVAE:
    def __init__(self, cfg: OracleConfig, dtype=torch.float64, seed: int = 1):
        self.cfg = cfg
        self.dtype = dtype
        self.s = Store(dtype, seed)
        self.steps = 0
        self.epoch = 0
        self._build()
        for p in self.s.params.values():
            p.requires_grad_(True)
        # Adamax slots (train.py:131; Keras defaults beta1 .9, beta2 .999, eps 1e-7 [3P])
        self.opt_m = {k: torch.zeros_like(v) for k, v in self.s.params.items()}
        self.opt_u = {k: torch.zeros_like(v) for k, v in self.s.params.items()}
        self.opt_iter = 0

    # ---------------------------------------------------------------- construction
    def _build(self):
        c, s = self.cfg, self.s
        sf = c.scale_factor
        E = c.n_encoder_channels
        # ---- Preprocess (preprocess.py:7-35)
        s.conv("pre.stem", 3, c.input_channels, E)
        mult = 1
        self.pre_cells = []   # (name, c_in, c_out, stride)
        idx = 0
        for _ in range(c.n_preprocess_blocks):
            for _ in range(c.n_preprocess_cells - 1):
                ch = mult * E
                self._build_bnswishconv(f"pre.cell{idx}", ch, ch, 1)
                self.pre_cells.append((f"pre.cell{idx}", ch, ch, 1))
                idx += 1
            ch_out = mult * E * sf
            self._build_bnswishconv(f"pre.cell{idx}", mult * E, ch_out, 2)
            self.pre_cells.append((f"pre.cell{idx}", mult * E, ch_out, 2))
            idx += 1
            mult *= sf
        # ---- Encoder (encoder.py:20-68)
        self.enc_layers = []   # ("group", gi, C) | ("comb", ti, C) | ("down", si, C_in, C_out)
        gi = ti = 0
        for scale in range(c.n_latent_scales):
            n_groups = c.n_groups_per_scale[scale]
            for g in range(n_groups):
                C = E * mult
                for cell in range(c.res_cells_per_group):
                    n = f"enc.g{gi}.c{cell}"
                    s.bn(n + ".bn1", C, True); s.conv(n + ".conv1", 3, C, C)
                    s.bn(n + ".bn2", C, True); s.conv(n + ".conv2", 3, C, C)
                    s.se(n + ".se", C)
                self.enc_layers.append(("group", gi, C)); gi += 1
                if not (scale == c.n_latent_scales - 1 and g == n_groups - 1):
                    # EncoderDecoderCombiner conv acts on the decoder state (encoder.py:12-15)
                    s.conv(f"enc.comb{ti}.conv", 1, c.n_decoder_channels * mult, C)
                    self.enc_layers.append(("comb", ti, C)); ti += 1
            if scale < c.n_latent_scales - 1:
                C = E * mult
                s.bn(f"enc.down{scale}.bn", C, True)
                s.conv(f"enc.down{scale}.conv", 3, C, C * sf)
                self.enc_layers.append(("down", scale, C, C * sf))
                mult *= sf
        s.conv("enc.final.conv", 1, E * mult, E * mult)
        self.n_taps = ti
        # ---- Decoder (decoder.py:10-62); groups reversed (models.py:69)
        D = c.n_decoder_channels
        dec_groups = list(reversed(c.n_groups_per_scale))
        top_hw = c.input_hw // (sf ** c.n_preprocess_blocks) // (sf ** (c.n_latent_scales - 1))
        self.top_hw = top_hw
        self.dec_layers = []   # ("cells", zi, C) | ("samplecomb", zi, C) | ("up", si, C_in, C_out)
        zi = 0
        L = c.n_latent_per_group
        for scale in range(c.n_latent_scales):
            for g in range(dec_groups[scale]):
                C = D * mult
                # Sampler convs (common.py:36-63); input width is whatever reaches them
                s.conv(f"dec.samp.enc{zi}.conv", 3, c.n_encoder_channels * mult, 2 * L)
                if not (scale == 0 and g == 0):
                    s.conv(f"dec.samp.dec{zi}.conv", 1, C, 2 * L)
                    for cell in range(c.res_cells_per_group):
                        n = f"dec.g{zi}.c{cell}"
                        s.bn(n + ".bn1", C, True); s.conv(n + ".conv1", 1, C, 6 * C)
                        s.bn(n + ".bn2", 6 * C, True); s.dw(n + ".dw", 6 * C)
                        s.bn(n + ".bn3", 6 * C, True); s.conv(n + ".conv2", 1, 6 * C, C)
                        s.bn(n + ".bn4", C, True); s.se(n + ".se", C)
                    self.dec_layers.append(("cells", zi, C))
                    s.conv(f"dec.comb{zi}.conv", 1, C + L, C)
                else:
                    s.conv(f"dec.comb{zi}.conv", 1, D + L, C)   # Q12: h has D channels
                self.dec_layers.append(("samplecomb", zi, C))
                zi += 1
            if scale < c.n_latent_scales - 1:
                C = D * mult
                s.bn(f"dec.up{scale}.bn", C, True)
                s.conv(f"dec.up{scale}.conv", 3, C, C // sf)
                self.dec_layers.append(("up", scale, C, C // sf))
                mult //= sf
        self.n_groups = zi
        s.params["dec.h"] = torch.rand((top_hw, top_hw, D), generator=s.gen,
                                       dtype=torch.float64).to(self.dtype)   # decoder.py:60-62
        # ---- Postprocess (postprocess.py:8-31)
        self.post_cells = []   # (name, c_in, C, upscale)
        idx = 0
        for _ in range(c.n_postprocess_blocks):
            c_in = D * mult
            mult //= sf
            C = D * mult
            for cell in range(c.n_post_process_cells):
                up = cell == 0
                n = f"post.cell{idx}"
                cin = c_in if up else C
                if up:
                    s.bn(n + ".skip.bn", cin); s.conv(n + ".skip.conv", 3, cin, C)
                    s.bn(n + ".up.bn", cin); s.conv(n + ".up.conv", 3, cin, C)
                s.bn(n + ".bn0", C)
                s.conv(n + ".conv1", 1, C, 6 * C, bias=False); s.bn(n + ".bn1", 6 * C)
                s.conv(n + ".conv5", 5, 6 * C, 6 * C, bias=False); s.bn(n + ".bn2", 6 * C)
                s.conv(n + ".conv3", 1, 6 * C, C, bias=False); s.bn(n + ".bn3", C)
                s.se(n + ".se", C)
                self.post_cells.append((n, cin, C, up))
                idx += 1
        s.conv("post.final.conv", 3, D * mult, c.output_channels)

    def _build_bnswishconv(self, n, c_in, c_out, stride):
        """preprocess.py:77-101 (+ SkipScaler 42-63)."""
        s = self.s
        s.bn(n + ".bn0", c_in); s.conv(n + ".conv0", 3, c_in, c_out)
        s.bn(n + ".bn1", c_out); s.conv(n + ".conv1", 3, c_out, c_out)
        s.se(n + ".se", c_out)
        if stride == 2:
            q = c_out // 4
            for i, co in enumerate([q, q, q, c_out - 3 * q]):
                s.conv(f"{n}.skip.conv{i + 1}", 1, c_in, co)

    def n_trainable(self) -> int:
        return sum(p.numel() for p in self.s.params.values())

    # ---------------------------------------------------------------- primitive layers
    def P(self, name):
        return self.s.params[name]

    def spectral_norm_step(self):
        """TFA SpectralNormalization.normalize_weights [3P] for every wrapped conv (a22):
        one power iteration, then W <- W / sigma in place, u <- u_new (no gradient)."""
        with torch.no_grad():
            for name in self.s.sn_convs:
                w = self.s.params[name + ".w"]
                u = self.s.state[name + ".u"].reshape(1, -1)
                w2 = w.reshape(-1, w.shape[-1])
                v = u @ w2.t()
                v = v * torch.rsqrt(torch.clamp((v * v).sum(), min=SN_EPS))
                un = v @ w2
                un = un * torch.rsqrt(torch.clamp((un * un).sum(), min=SN_EPS))
                sigma = (v @ w2 @ un.t()).reshape(())
                w.div_(sigma)
                self.s.state[name + ".u"] = un.reshape(-1)

    # `act_round` (None = exact arithmetic): emulation of a 16-bit activation path for the parity tests' error model.
    # The HIP kernels keep f32 accumulators, statistics and weights master copies but STORE activations (conv /
    # depthwise outputs, materialised BatchNorm(+Swish) outputs, SE + residual outputs) and read weights in the 16-bit
    # type.  With act_round set, the oracle rounds the same tensors (straight-through for the gradient), which gives
    # the spread a correct 16-bit implementation has around the exact result, per gradient tensor
    # (tests/test_model_gpu.py::bf16_spread).  It is not a bit model of the kernels.
    act_round = None

    def _r(self, x):
        return x if self.act_round is None else x + (self.act_round(x.detach()) - x.detach())

    def conv(self, name, x, stride=1):
        w = self.P(name + ".w")
        return self._r(conv2d(x, w if self.act_round is None else self._r(w), self.s.params.get(name + ".b"), stride))

    def bn(self, name, x, training):
        """Keras BatchNormalization(momentum=.05, epsilon=1e-5) (a23)."""
        g, b = self.P(name + ".gamma"), self.P(name + ".beta")
        if training:
            mean = x.mean(dim=(0, 1, 2))
            var = x.var(dim=(0, 1, 2), unbiased=False)
            with torch.no_grad():
                self.s.state[name + ".rm"] = (BN_MOMENTUM * self.s.state[name + ".rm"]
                                              + (1 - BN_MOMENTUM) * mean.detach())
                self.s.state[name + ".rv"] = (BN_MOMENTUM * self.s.state[name + ".rv"]
                                              + (1 - BN_MOMENTUM) * var.detach())
        else:
            mean, var = self.s.state[name + ".rm"], self.s.state[name + ".rv"]
        return self._r((x - mean) * torch.rsqrt(var + BN_EPS) * g + b)

    def se(self, name, x):
        """SqueezeExcitation.call, common.py:127-142."""
        p = x.mean(dim=(1, 2))
        h = torch.relu(p @ self.P(name + ".w1") + self.P(name + ".b1"))
        gate = torch.sigmoid(h @ self.P(name + ".w2") + self.P(name + ".b2"))
        return self._r(x * gate[:, None, None, :])

    def rescaler(self, name, x, up: bool, training):
        """Rescaler.call, common.py:164-174."""
        x = swish(self.bn(name + ".bn", x, training))
        if up:
            x = upsample_nearest(x, self.cfg.scale_factor)
            return self.conv(name + ".conv", x)
        return self.conv(name + ".conv", x, stride=self.cfg.scale_factor)

    # ---------------------------------------------------------------- towers
    def preprocess(self, x, training):
        """Preprocess.call preprocess.py:37-39; BNSwishConv.call 103-107; SkipScaler.call 65-74."""
        x = self.conv("pre.stem", 2 * x - 1)
        for (n, c_in, c_out, stride) in self.pre_cells:
            y = swish(self.bn(n + ".bn0", x, training))
            y = self.conv(n + ".conv0", y, stride)
            y = swish(self.bn(n + ".bn1", y, training))
            y = self.conv(n + ".conv1", y)
            y = self.se(n + ".se", y)
            if stride == 1:
                skip = x
            else:
                o = swish(x)
                parts = [self.conv(n + ".skip.conv1", o, 2),
                         self.conv(n + ".skip.conv2", o[:, 1:, 1:, :], 2),
                         self.conv(n + ".skip.conv3", o[:, :, 1:, :], 2),
                         self.conv(n + ".skip.conv4", o[:, 1:, :, :], 2)]
                skip = torch.cat(parts, dim=3)
            x = skip + 0.1 * y
        return x

    def enc_cell(self, n, x, training):
        """EncodingResidualCell.call, encoder.py:101-107."""
        y = swish(self.bn(n + ".bn1", x, training))
        y = self.conv(n + ".conv1", y)
        y = swish(self.bn(n + ".bn2", y, training))
        y = self.conv(n + ".conv2", y)
        y = self.se(n + ".se", y)
        return 0.1 * x + y

    def encoder(self, x, training):
        """Encoder.call, encoder.py:70-83: returns (tap activations bottom-up, final)."""
        taps = []
        for layer in self.enc_layers:
            if layer[0] == "group":
                for cell in range(self.cfg.res_cells_per_group):
                    x = self.enc_cell(f"enc.g{layer[1]}.c{cell}", x, training)
            elif layer[0] == "comb":
                taps.append((layer[1], x))
            else:
                x = self.rescaler(f"enc.down{layer[1]}", x, False, training)
        final = F.elu(self.conv("enc.final.conv", F.elu(x)))   # encoder.py:58-66
        return taps, final

    def gen_cell(self, n, x, training):
        """GenerativeResidualCell.call, decoder.py:137-147."""
        y = self.bn(n + ".bn1", x, training)
        y = self.conv(n + ".conv1", y)
        y = swish(self.bn(n + ".bn2", y, training))
        y = self._r(dwconv5(y, self.P(n + ".dw.w"), self.P(n + ".dw.b")))
        y = swish(self.bn(n + ".bn3", y, training))
        y = self.conv(n + ".conv2", y)
        y = self.bn(n + ".bn4", y, training)
        y = self.se(n + ".se", y)
        return 0.1 * x + y

    def sampler(self, zi, prior, enc_prior, eps):
        """Sampler.call, common.py:76-102 (no squeeze: Q9)."""
        L = self.cfg.n_latent_per_group
        e = self.conv(f"dec.samp.enc{zi}.conv", enc_prior)
        e_mu, e_ls = e[..., :L], e[..., L:]
        if zi == 0:
            enc_mu = softclamp5(e_mu)
            enc_sigma = torch.exp(softclamp5(e_ls)) + 1e-2
            z = enc_mu + eps * enc_sigma
            return z, DistributionParams(enc_mu, enc_sigma, torch.zeros_like(enc_mu),
                                         torch.ones_like(enc_sigma))
        d = self.conv(f"dec.samp.dec{zi}.conv", F.elu(prior))
        r_mu, r_ls = d[..., :L], d[..., L:]
        dec_mu = softclamp5(r_mu)
        dec_sigma = torch.exp(softclamp5(r_ls)) + 1e-2
        enc_mu = softclamp5(e_mu + r_mu)
        enc_sigma = torch.exp(softclamp5(r_ls + e_ls)) + 1e-2
        z = enc_mu + eps * enc_sigma
        return z, DistributionParams(enc_mu, enc_sigma, dec_mu, dec_sigma)

    def decoder(self, final, taps, eps_list, training, nll=False):
        """Decoder.call, decoder.py:64-104.  `taps` bottom-up; consumed reversed (models.py:93)."""
        B = final.shape[0]
        taps = list(reversed(taps))
        z_params, zs = [], []
        log_p = torch.zeros(B, dtype=final.dtype)
        log_q = torch.zeros(B, dtype=final.dtype)

        def account(z, p):
            nonlocal log_p, log_q
            if nll:
                log_q = log_q + calculate_log_p(z, p.enc_mu, p.enc_sigma).sum(dim=(1, 2, 3))
                log_p = log_p + calculate_log_p(z, p.dec_mu, p.dec_sigma).sum(dim=(1, 2, 3))

        z, p = self.sampler(0, final, final, eps_list[0])
        account(z, p); z_params.append(p); zs.append(z)
        h = self.P("dec.h").unsqueeze(0).expand(B, -1, -1, -1)
        x = self.conv("dec.comb0.conv", torch.cat((h, z), dim=3))   # decoder.py:116-117
        ci = 0
        for layer in self.dec_layers[1:]:
            if layer[0] == "cells":
                for cell in range(self.cfg.res_cells_per_group):
                    x = self.gen_cell(f"dec.g{layer[1]}.c{cell}", x, training)
            elif layer[0] == "samplecomb":
                zi = layer[1]
                ti, enc_x = taps[ci]
                enc_prior = enc_x + self.conv(f"enc.comb{ti}.conv", x)   # encoder.py:14-16
                z, p = self.sampler(zi, x, enc_prior, eps_list[zi])
                account(z, p); z_params.append(p); zs.append(z)
                x = self.conv(f"dec.comb{zi}.conv", torch.cat((x, z), dim=3))
                ci += 1
            else:
                x = self.rescaler(f"dec.up{layer[1]}", x, True, training)
        return x, z_params, log_p, log_q, zs

    def postprocess(self, x, training):
        """Postprocess.call postprocess.py:33-34; PostprocessCell 57-58; PostprocessNode 61-88."""
        for (n, c_in, C, up) in self.post_cells:
            if up:
                skip = self.rescaler(n + ".skip", x, True, training)
                y = self.rescaler(n + ".up", x, True, training)
            else:
                skip, y = x, x
            y = self.bn(n + ".bn0", y, training)
            y = swish(self.bn(n + ".bn1", self.conv(n + ".conv1", y), training))
            y = swish(self.bn(n + ".bn2", self.conv(n + ".conv5", y), training))   # dense 5x5 (Q7)
            y = self.bn(n + ".bn3", self.conv(n + ".conv3", y), training)
            y = self.se(n + ".se", y)
            x = skip + 0.1 * y
        return self.conv("post.final.conv", F.elu(x))

    # ---------------------------------------------------------------- NVAE API (models.py)
    def eps_shapes(self, B) -> List[Tuple[int, ...]]:
        """Noise shapes per latent group, decoder order."""
        shapes = []
        hw = self.top_hw
        dec_groups = list(reversed(self.cfg.n_groups_per_scale))
        for scale in range(self.cfg.n_latent_scales):
            for _ in range(dec_groups[scale]):
                shapes.append((B, hw, hw, self.cfg.n_latent_per_group))
            hw *= self.cfg.scale_factor
        return shapes

    def call(self, x, eps_list, training=False, nll=False):
        """NVAE.call, models.py:89-98."""
        h = self.preprocess(x, training)
        taps, final = self.encoder(h, training)
        s, z_params, log_p, log_q, zs = self.decoder(final, taps, eps_list, training, nll)
        logits = self.postprocess(s, training)
        return logits, z_params, log_p, log_q, zs

    def calculate_kl_alphas(self):
        """models.py:227-237."""
        S, gps = self.cfg.n_latent_scales, self.cfg.n_groups_per_scale
        coeffs = []
        for i in range(S):
            g = gps[S - i - 1]
            coeffs.append(torch.full((g,), (2 ** i) ** 2 / g, dtype=self.dtype))
        coeffs = torch.cat(coeffs)
        return coeffs / coeffs.min()

    def kl_per_group(self, z_params):
        """models.py:197-201 -> [G, B]."""
        out = []
        for g in z_params:
            t1 = (g.enc_mu - g.dec_mu) / g.dec_sigma
            t2 = g.enc_sigma / g.dec_sigma
            kl = 0.5 * (t1 * t1 + t2 * t2) - 0.5 - torch.log(t2)
            out.append(kl.sum(dim=(1, 2, 3)))
        return torch.stack(out, 0)

    def calculate_kl_loss(self, z_params, balancing, kl_all=None):
        """models.py:191-223 -> ([B], coeff[G] or None)."""
        if kl_all is None:
            kl_all = self.kl_per_group(z_params)
        if balancing:
            alphas = self.calculate_kl_alphas()
            coeff = kl_all.abs().mean(dim=1) + 0.01
            total = coeff.sum()
            coeff = coeff / alphas * total
            coeff = coeff / coeff.mean()
            coeff = coeff.detach()
            return (kl_all * coeff[:, None]).sum(dim=0), coeff
        return kl_all.sum(dim=0), None

    def calculate_recon_loss(self, inputs, logits, crop_output=False):
        """models.py:242-250: -sum log Bernoulli(x; logits) = sum softplus(l) - x*l."""
        if crop_output:
            inputs = inputs[:, 2:30, 2:30, :]
            logits = logits[:, 2:30, 2:30, :]
        if self.cfg.head == "dmol":
            return -dmol_log_prob(inputs, logits, self.cfg.num_mixture_dec).sum(dim=(1, 2))
        return (F.softplus(logits) - inputs * logits).sum(dim=(1, 2, 3))

    def calculate_bn_loss(self):
        """models.py:252-267: lambda * sum over BN layers of enc/dec towers of max|gamma|."""
        tot = torch.zeros((), dtype=self.dtype)
        for n in self.s.bn_loss_layers:
            tot = tot + self.P(n + ".gamma").abs().max()
        return self.cfg.sr_lambda * tot

    def beta(self):
        """models.py:121-122."""
        m = self.steps if self.cfg.step_based_warmup else self.epoch
        return min(m / (0.3 * self.cfg.n_total_iterations), 1)

    def loss(self, x, eps_list, training=True):
        """Forward + ELBO as in train_step, models.py:116-126 (no SN step, no update)."""
        logits, z_params, _, _, zs = self.call(x, eps_list, training=training)
        recon = self.calculate_recon_loss(x, logits)
        bn_loss = self.calculate_bn_loss()
        beta = self.beta()
        kl_all = self.kl_per_group(z_params)
        kl, coeff = self.calculate_kl_loss(z_params, beta < 1, kl_all)
        kl = beta * kl
        total = (recon + kl).mean() + bn_loss
        return {"loss": total, "reconstruction_loss": recon, "kl_loss": kl, "bn_loss": bn_loss,
                "kl_per_group": kl_all, "kl_coeff": coeff, "logits": logits, "zs": zs,
                "z_params": z_params}

    def lr(self, it, decay_steps, base=1e-3):
        """CosineDecay(1e-3, decay_steps), train.py:128-130 [3P]."""
        it = min(it, decay_steps)
        return base * 0.5 * (1 + math.cos(math.pi * it / decay_steps))

    def adamax_apply(self, grads: Dict[str, torch.Tensor], lr, b1=0.9, b2=0.999, eps=1e-7):
        """Keras Adamax [3P] (a21)."""
        self.opt_iter += 1
        t = self.opt_iter
        with torch.no_grad():
            for k, p in self.s.params.items():
                g = grads[k]
                self.opt_m[k].mul_(b1).add_(g, alpha=1 - b1)
                self.opt_u[k] = torch.maximum(b2 * self.opt_u[k], g.abs())
                p.sub_(lr / (1 - b1 ** t) * self.opt_m[k] / (self.opt_u[k] + eps))

    def train_step(self, x, eps_list, decay_steps=None, spectral_norm=True, tf_literal=False):
        """NVAE.train_step, models.py:100-135 with training-mode semantics (Q1); tf_literal=True is the
        reference's literal behaviour (BN moving statistics, no spectral normalisation)."""
        if spectral_norm and not tf_literal:
            self.spectral_norm_step()
        out = self.loss(x, eps_list, training=not tf_literal)
        names = list(self.s.params.keys())
        grads = torch.autograd.grad(out["loss"], [self.s.params[k] for k in names],
                                    allow_unused=True)
        gd = {k: (g if g is not None else torch.zeros_like(self.s.params[k]))
              for k, g in zip(names, grads)}
        decay_steps = decay_steps or self.cfg.n_total_iterations
        self.adamax_apply(gd, self.lr(self.opt_iter, decay_steps))
        self.steps += 1
        out["grads"] = gd
        return out

    def sample(self, n_samples, temperature, eps_list, greyscale=True, return_last=False):
        """NVAE.sample, models.py:137-178 (BN inference mode; temperature on z0 only: Q8).
        return_last: also return (last_s, last_z): the decoder state entering the last
        DecoderSampleCombiner and the z it was combined with (what sample_with_z starts from)."""
        B = n_samples
        L = self.cfg.n_latent_per_group
        s = self.P("dec.h").unsqueeze(0).expand(B, -1, -1, -1)
        z0s = (B, self.top_hw, self.top_hw, L)
        mu = softclamp5(torch.zeros(z0s, dtype=self.dtype))
        sigma = torch.exp(softclamp5(torch.zeros(z0s, dtype=self.dtype))) + 1e-2
        if temperature != 1.0:
            sigma = sigma * temperature
        z = mu + eps_list[0] * sigma
        last_s, last_z = s, z
        s = self.conv("dec.comb0.conv", torch.cat((s, z), dim=3))
        for layer in self.dec_layers[1:]:
            if layer[0] == "cells":
                for cell in range(self.cfg.res_cells_per_group):
                    s = self.gen_cell(f"dec.g{layer[1]}.c{cell}", s, False)
            elif layer[0] == "samplecomb":
                zi = layer[1]
                d = self.conv(f"dec.samp.dec{zi}.conv", F.elu(s))
                mu = softclamp5(d[..., :L])
                sigma = torch.exp(softclamp5(d[..., L:])) + 1e-2
                z = mu + eps_list[zi] * sigma
                last_s, last_z = s, z
                s = self.conv(f"dec.comb{zi}.conv", torch.cat((s, z), dim=3))
            else:
                s = self.rescaler(f"dec.up{layer[1]}", s, True, False)
        logits = self.postprocess(s, False)
        if self.cfg.head == "dmol":
            out = logits          # draw pixels with dmol_sample(logits, M, u_mix, u_pix, t)
        else:
            out = torch.sigmoid(logits) if greyscale else logits
        return (out, last_s, last_z) if return_last else out

    def sample_with_z(self, z, s):
        """NVAE.sample_with_z, models.py:181-189: decoder.groups[-1] (the last DecoderSampleCombiner)
        on (s, z), then postprocess; Bernoulli mean of the logits."""
        zi = max(l[1] for l in self.dec_layers if l[0] == "samplecomb") if len(self.dec_layers) > 1 else 0
        s = self.conv(f"dec.comb{zi}.conv", torch.cat((s, z), dim=3))
        logits = self.postprocess(s, False)
        return logits if self.cfg.head == "dmol" else torch.sigmoid(logits)

    def neg_log_likelihood(self, x, eps_lists):
        """evaluate.py:111-123 for ONE batch: IWAE bound with k = len(eps_lists)."""
        logs = []
        for eps in eps_lists:
            logits, _, log_p, log_q, _ = self.call(x, eps, training=False, nll=True)
            crop = self.cfg.head == "bernoulli"      # the 28x28 crop is MNIST's zero padding (Q5)
            logs.append(-self.calculate_recon_loss(x, logits, crop_output=crop) - log_q + log_p)
        k = len(eps_lists)
        return -(torch.logsumexp(torch.stack(logs), dim=0) - math.log(float(k))).mean()


# --------------------------------------------------------------------------------------
# Discretised mixture of logistics (NOT in the reference: SURVEY 8f "ext"; specified here after the
# NVAE paper sec. 3 / PixelCNN++: M mixtures, RGB sub-pixel conditioning through tanh coefficients,
# 8-bit bins of half-width 1/255 on [-1, 1], log-scales clamped at -7).  Channel layout of the
# 10*M logits per pixel: [0, M) mixture logits; then for colour c in (R, G, B) a block of 3M at
# M + 3M*c: means [0, M), log-scales [M, 2M), coefficient logits [2M, 3M).
# --------------------------------------------------------------------------------------
def dmol_split(logits, M):
    lp = logits[..., :M]
    blk = logits[..., M:].reshape(*logits.shape[:-1], 3, 3 * M)
    means = blk[..., :M]                                  # [..., 3, M]
    log_scales = torch.clamp(blk[..., M:2 * M], min=-7.0)
    coeffs = torch.tanh(blk[..., 2 * M:])
    return lp, means, log_scales, coeffs


def dmol_log_prob(x01, logits, M):
    """log p(x) per pixel, [B, H, W]; x01 in [0, 1] with 3 channels."""
    x = 2.0 * x01 - 1.0
    lp, means, log_scales, coeffs = dmol_split(logits, M)
    xs = x.unsqueeze(-1)                                  # [B, H, W, 3, 1]
    m1 = means[..., 0, :]
    m2 = means[..., 1, :] + coeffs[..., 0, :] * xs[..., 0, :]
    m3 = means[..., 2, :] + coeffs[..., 1, :] * xs[..., 0, :] + coeffs[..., 2, :] * xs[..., 1, :]
    mu = torch.stack((m1, m2, m3), dim=-2)                # [B, H, W, 3, M]
    centered = xs - mu
    inv_stdv = torch.exp(-log_scales)
    plus_in = inv_stdv * (centered + 1.0 / 255.0)
    min_in = inv_stdv * (centered - 1.0 / 255.0)
    cdf_delta = torch.sigmoid(plus_in) - torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)
    log_one_minus_cdf_min = -F.softplus(min_in)
    mid_in = inv_stdv * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * F.softplus(mid_in)
    mid_safe = torch.where(cdf_delta > 1e-5, torch.log(torch.clamp(cdf_delta, min=1e-10)),
                           log_pdf_mid - math.log(127.5))
    lpc = torch.where(xs < -0.999, log_cdf_plus, torch.where(xs > 0.99, log_one_minus_cdf_min, mid_safe))
    return torch.logsumexp(lpc.sum(dim=-2) + F.log_softmax(lp, dim=-1), dim=-1)


def dmol_sample(logits, M, u_mix, u_pix, t=1.0):
    """One draw in [0, 1]^3 per pixel.  u_mix [B,H,W,M], u_pix [B,H,W,3] ~ U(1e-5, 1 - 1e-5).  The
    temperature sharpens the mixture choice (logits / t) and scales the logistic noise by t."""
    lp, means, log_scales, coeffs = dmol_split(logits, M)
    gumbel = -torch.log(-torch.log(u_mix))
    k = torch.argmax(lp / t + gumbel, dim=-1)             # [B, H, W]
    idx = k[..., None, None].expand(*k.shape, 3, 1)
    mu = torch.gather(means, -1, idx).squeeze(-1)         # [B, H, W, 3]
    ls = torch.gather(log_scales, -1, idx).squeeze(-1)
    co = torch.gather(coeffs, -1, idx).squeeze(-1)
    x = mu + torch.exp(ls) * t * (torch.log(u_pix) - torch.log(1.0 - u_pix))
    x0 = torch.clamp(x[..., 0], -1, 1)
    x1 = torch.clamp(x[..., 1] + co[..., 0] * x0, -1, 1)
    x2 = torch.clamp(x[..., 2] + co[..., 1] * x0 + co[..., 2] * x1, -1, 1)
    return torch.stack((x0, x1, x2), dim=-1) / 2.0 + 0.5


def synthetic_rgb_batch(B, hw=32, seed=1, dtype=torch.float64):
    """8-bit RGB images in [0, 1] (k/255): smooth random blobs plus noise, so that all three cases of
    the discretised likelihood (x = 0, x = 255, interior bins) occur."""
    g = torch.Generator().manual_seed(seed)
    low = torch.rand(B, 3, max(hw // 8, 1), max(hw // 8, 1), generator=g, dtype=torch.float64)
    img = F.interpolate(low, size=(hw, hw), mode="bilinear", align_corners=False)
    img = (img - 0.5) * 2.2 + 0.5 + 0.05 * torch.randn(B, 3, hw, hw, generator=g, dtype=torch.float64)
    img = torch.round(torch.clamp(img, 0, 1) * 255.0) / 255.0
    return img.permute(0, 2, 3, 1).contiguous().to(dtype)


def synthetic_batch(B, seed=1, hw=32, p=0.19, dtype=torch.float64):
    """SURVEY 8d synthetic MNIST-shaped input: inner 28x28 ~ Bernoulli(.19), 2-pixel zero border."""
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(B, hw, hw, 1, dtype=dtype)
    x[:, 2:hw - 2, 2:hw - 2, :] = (torch.rand(B, hw - 4, hw - 4, 1, generator=g) < p).to(dtype)
    return x
