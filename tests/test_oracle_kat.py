"""Known-answer tests pinning the CPU oracle to facts derivable from the reference source
(SURVEY.md section 8c).  The reference ships no tests or fixtures, so these - not TensorFlow
outputs - are what the oracle is pinned by ("parity unpinned" otherwise)."""
import math

import pytest
import torch

from oracle.nvae_oracle import (OracleConfig, OracleNVAE, calculate_log_p, conv2d, same_pad, softclamp5,
                                synthetic_batch)

D = torch.float64


def small(groups=(1, 1), **kw):
    cfg = dict(n_encoder_channels=8, n_decoder_channels=8, n_latent_per_group=4, n_groups_per_scale=list(groups),
               n_preprocess_cells=2, n_post_process_cells=2)
    cfg.update(kw)
    return OracleNVAE(OracleConfig(**cfg), dtype=D, seed=3)


def test_softclamp5_and_log_p():   # util.py:39-50
    assert float(softclamp5(torch.tensor(0.0, dtype=D))) == 0.0
    assert abs(float(softclamp5(torch.tensor(1e9, dtype=D))) - 5.0) < 1e-12
    x = torch.linspace(-7, 7, 15, dtype=D)
    assert torch.allclose(softclamp5(-x), -softclamp5(x))
    z = torch.tensor([0.3], dtype=D)
    assert abs(float(calculate_log_p(z, z, torch.ones(1, dtype=D))) + 0.5 * math.log(2 * math.pi)) < 1e-12


def test_parameter_totals():   # constructors: preprocess.py:19-33, encoder.py:34-66, decoder.py:30-62, postprocess.py:13-30
    for groups, cells, want in (([5, 10], 1, 40128893), ([1, 1], 1, 17243405), ([5, 10], 2, 62225021)):
        m = OracleNVAE(OracleConfig(n_groups_per_scale=groups, res_cells_per_group=cells), dtype=torch.float32)
        assert m.n_trainable() == want
    m = OracleNVAE(OracleConfig(), dtype=torch.float32)
    assert len(m.s.bn_loss_layers) == 88 and len(m.s.sn_convs) == 163
    assert abs(float(m.calculate_bn_loss()) - 0.88) < 1e-6                # 0.01 * 88 at gamma = 1


def test_kl_alphas():   # models.py:227-237
    assert small((5, 10)).calculate_kl_alphas().tolist() == [1.0] * 10 + [8.0] * 5
    assert small((1, 1)).calculate_kl_alphas().tolist() == [1.0, 4.0]
    assert small((2,)).calculate_kl_alphas().tolist() == [1.0, 1.0]


def test_kl_closed_forms():   # models.py:197-218
    m = small((5, 10))
    from oracle.nvae_oracle import DistributionParams
    mu = torch.randn(3, 4, 4, 4, dtype=D)
    sg = torch.rand(3, 4, 4, 4, dtype=D) + 0.5
    p = DistributionParams(mu, sg, mu.clone(), sg.clone())
    assert float(m.kl_per_group([p]).abs().max()) < 1e-12          # KL(q || q) = 0
    p = DistributionParams(mu, sg, torch.zeros_like(mu), torch.ones_like(sg))
    want = (0.5 * (mu * mu + sg * sg) - 0.5 - torch.log(sg)).sum((1, 2, 3))
    assert torch.allclose(m.kl_per_group([p])[0], want)
    # equal per-group KL -> balanced coefficients proportional to 1/alpha, mean 1
    kl_all = torch.full((15, 3), 2.0, dtype=D)
    _, coeff = m.calculate_kl_loss(None, True, kl_all)
    inv = 1.0 / m.calculate_kl_alphas()
    assert torch.allclose(coeff, inv / inv.mean())


def test_recon_at_zero_logits():   # models.py:242-250
    m = small()
    x = synthetic_batch(2, dtype=D)
    z = torch.zeros(2, 32, 32, 1, dtype=D)
    assert torch.allclose(m.calculate_recon_loss(x, z), torch.full((2,), 1024 * math.log(2), dtype=D))
    assert torch.allclose(m.calculate_recon_loss(x, z, crop_output=True), torch.full((2,), 784 * math.log(2), dtype=D))


def test_beta_and_lr_schedules():   # models.py:121-122, train.py:128-130
    m = small(n_total_iterations=1000)
    m.steps = 0
    assert m.beta() == 0
    m.steps = 150
    assert abs(m.beta() - 0.5) < 1e-12
    m.steps = 300
    assert m.beta() == 1
    assert abs(m.lr(0, 1000) - 1e-3) < 1e-15 and abs(m.lr(500, 1000) - 5e-4) < 1e-12 and m.lr(1000, 1000) < 1e-12


def test_same_padding_and_stage_shapes():   # SURVEY Q6 + shape table
    assert same_pad(32, 3, 2) == (0, 1) and same_pad(32, 3, 1) == (1, 1) and same_pad(16, 5, 1) == (2, 2)
    x = torch.zeros(1, 8, 8, 1, dtype=D)
    x[0, 7, 7, 0] = 1.0
    w = torch.zeros(3, 3, 1, 1, dtype=D)
    w[2, 2, 0, 0] = 1.0     # bottom-right tap: reaches the bottom/right zero padding only
    y = conv2d(x, w, None, stride=2)
    assert y.shape == (1, 4, 4, 1) and float(y.abs().sum()) == 0.0
    w[:] = 0
    w[1, 1, 0, 0] = 1.0     # with pad_top = 0 the tap (1,1) of output (3,3) reads input (7,7)
    assert float(conv2d(x, w, None, stride=2)[0, 3, 3, 0]) == 1.0
    m = OracleNVAE(OracleConfig(), dtype=torch.float32)
    xs = synthetic_batch(1, dtype=torch.float32)
    h = m.preprocess(xs, False)
    taps, final = m.encoder(h, False)
    assert h.shape == (1, 8, 8, 128) and final.shape == (1, 4, 4, 256) and len(taps) == 14
    assert [tuple(t[1].shape[1:]) for t in taps] == [(8, 8, 128)] * 5 + [(4, 4, 256)] * 9
    assert m.eps_shapes(1) == [(1, 4, 4, 20)] * 10 + [(1, 8, 8, 20)] * 5


def test_skipscaler_shifts():   # preprocess.py:68-71 on an index-coded tensor
    m = small()
    n = "pre.cell1"       # the stride-2 cell of block 0
    with torch.no_grad():
        for i in range(4):
            w = m.s.params[f"{n}.skip.conv{i + 1}.w"]
            w.zero_()
            w[0, 0, 0, :] = 1.0      # every output channel copies input channel 0
    x = torch.zeros(1, 32, 32, 8, dtype=D)
    hh, ww = torch.meshgrid(torch.arange(32.), torch.arange(32.), indexing="ij")
    x[0, :, :, 0] = (hh * 100 + ww).to(D) + 1000.0
    o = x * torch.sigmoid(x)
    parts = [conv2d(o, m.s.params[f"{n}.skip.conv1.w"], None, 2),
             conv2d(o[:, 1:, 1:, :], m.s.params[f"{n}.skip.conv2.w"], None, 2),
             conv2d(o[:, :, 1:, :], m.s.params[f"{n}.skip.conv3.w"], None, 2),
             conv2d(o[:, 1:, :, :], m.s.params[f"{n}.skip.conv4.w"], None, 2)]
    for part, (dh, dw) in zip(parts, [(0, 0), (1, 1), (0, 1), (1, 0)]):
        assert part.shape[1:3] == (16, 16)
        assert torch.allclose(part[0, 3, 5, 0], o[0, 6 + dh, 10 + dw, 0])


def test_group0_prior_sigma_and_se_of_constant():   # models.py:141-142, common.py:127-142
    m = small()
    eps = [torch.ones(s, dtype=D) for s in m.eps_shapes(2)]
    # sigma of group 0 in sample() is exp(softclamp5(0)) + 1e-2 = 1.01
    L = m.cfg.n_latent_per_group
    sigma0 = math.exp(0.0) + 1e-2
    assert abs(sigma0 - 1.01) < 1e-15
    x = torch.full((2, 4, 4, 8), 0.7, dtype=D)
    name = "pre.cell0.se"
    p = torch.full((2, 8), 0.7, dtype=D)
    h = torch.relu(p @ m.P(name + ".w1") + m.P(name + ".b1"))
    gate = torch.sigmoid(h @ m.P(name + ".w2") + m.P(name + ".b2"))
    assert torch.allclose(m.se(name, x), 0.7 * gate[:, None, None, :].expand(2, 4, 4, 8))


def test_spectral_norm_step_normalises_top_singular_direction():   # TFA normalize_weights [3P]
    m = small()
    name = m.s.sn_convs[3]
    for _ in range(300):
        m.spectral_norm_step()
    w2 = m.s.params[name + ".w"].detach().reshape(-1, m.s.params[name + ".w"].shape[-1])
    # power iteration under-estimates sigma, so the norm approaches 1 from above
    nrm = float(torch.linalg.matrix_norm(w2, 2))
    assert 1.0 - 1e-9 <= nrm < 1.02, nrm


def test_train_step_decreases_loss():
    m = small()
    m.steps = 10 ** 9
    x = synthetic_batch(4, dtype=D)
    g = torch.Generator().manual_seed(0)
    eps = [torch.randn(s, generator=g, dtype=D) for s in m.eps_shapes(4)]
    l0 = float(m.train_step(x, eps)["loss"].detach())
    for _ in range(5):
        out = m.train_step(x, eps)
    assert float(out["loss"].detach()) < l0
