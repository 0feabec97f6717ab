"""CPU: the oracle must reproduce the committed golden vectors (tests/golden/nvae_small.npz, made by
tests/golden/make_golden.py); guards the oracle itself against regressions."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


def close(a, b, tol=2e-6):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12)) < tol


def test_oracle_reproduces_golden_vectors():
    gold = np.load(os.path.join(HERE, "golden", "nvae_small.npz"))
    orc, x, eps = make_golden.build()
    assert np.allclose(make_golden.checksum(orc), gold["checksum"], rtol=1e-12, atol=1e-12)
    assert close(x.numpy(), gold["x"]) and all(close(e.numpy(), gold[f"eps{i}"]) for i, e in enumerate(eps))
    logits, zp, lp, lq, _ = orc.call(x, eps, training=False, nll=True)
    assert close(logits.detach(), gold["inf/logits"]) and close(lp.detach(), gold["inf/log_p"])
    assert close(lq.detach(), gold["inf/log_q"])
    for i, p in enumerate(zp):
        assert close(p.enc_mu.detach(), gold[f"inf/enc_mu{i}"]) and close(p.dec_sigma.detach(), gold[f"inf/dec_sigma{i}"])
    assert close(orc.sample(make_golden.B, 0.7, eps).detach(), gold["sample/t0.7"])
    orc.steps = 100
    o = orc.train_step(x, eps, decay_steps=1000)
    for k in ("loss", "reconstruction_loss", "kl_loss", "bn_loss", "kl_per_group", "kl_coeff"):
        assert close(o[k].detach(), gold["train/" + k]), k
    for k in make_golden.GRAD_KEYS:
        assert close(o["grads"][k].detach(), gold["grad/" + k], 2e-5), k
        assert close(orc.s.params[k].detach(), gold["updated/" + k]), k


def test_oracle_reproduces_rgb_golden_vectors():
    """The RGB / mixture-of-logistics fixture (tests/golden/nvae_rgb_small.npz, make_golden_rgb.py)."""
    import make_golden_rgb as mg
    from oracle.nvae_oracle import dmol_sample
    gold = np.load(os.path.join(HERE, "golden", "nvae_rgb_small.npz"))
    orc, x, eps, u_mix, u_pix = mg.build()
    assert np.allclose(make_golden.checksum(orc), gold["checksum"], rtol=1e-12, atol=1e-12)
    assert close(x.numpy(), gold["x"]) and close(u_mix.numpy(), gold["u_mix"])
    logits, zp, lp, lq, _ = orc.call(x, eps, training=False, nll=True)
    assert logits.shape[-1] == 10 * mg.M
    assert close(logits.detach(), gold["inf/logits"]) and close(lp.detach(), gold["inf/log_p"])
    assert close(orc.calculate_recon_loss(x, logits).detach(), gold["inf/recon"])
    s_logits = orc.sample(mg.B, 0.8, eps).detach()
    assert close(s_logits, gold["sample/logits_t0.8"])
    assert close(dmol_sample(s_logits, mg.M, u_mix, u_pix, 1.0), gold["sample/image_t0.8"], 1e-5)
    orc.steps = 100
    o = orc.train_step(x, eps, decay_steps=1000)
    for k in ("loss", "reconstruction_loss", "kl_loss", "bn_loss", "kl_per_group", "kl_coeff"):
        assert close(o[k].detach(), gold["train/" + k]), k
    for k in mg.GRAD_KEYS:
        assert close(o["grads"][k].detach(), gold["grad/" + k], 2e-5), k
