"""Generate tests/golden/nvae_small.npz from the CPU oracle (fp64): inputs, all weights/state, the
noise, and the expected outputs of one training step, the inference forward (IWAE terms) and ancestral
sampling for a shrunken NVAE.  The reference itself cannot be run here (TensorFlow is not installable
offline), so the vectors pin the build's two implementations of the reference arithmetic against
each other and against regressions; see DESIGN.md "Oracle".

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.nvae_oracle import OracleConfig, OracleNVAE, synthetic_batch  # noqa: E402

CFG = dict(n_encoder_channels=8, n_decoder_channels=8, res_cells_per_group=1, n_preprocess_blocks=2,
           n_preprocess_cells=2, n_latent_per_group=4, n_groups_per_scale=[1, 1], n_postprocess_blocks=2,
           n_post_process_cells=2, sr_lambda=0.01, scale_factor=2, total_epochs=10, n_total_iterations=1000,
           step_based_warmup=True)
B = 2
GRAD_KEYS = ["pre.stem.w", "enc.g1.c0.conv1.w", "dec.comb1.conv.w", "post.cell3.conv5.w", "dec.g1.c0.dw.w",
             "enc.g0.c0.bn1.gamma", "post.final.conv.b", "dec.h", "pre.cell0.se.w1"]


def build():
    """The fixture's model, inputs and noise, regenerated deterministically (torch CPU generator)."""
    orc = OracleNVAE(OracleConfig(**CFG), dtype=torch.float64, seed=11)
    g = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for k, v in orc.s.params.items():
            if k.endswith(".gamma"):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
            elif k.endswith((".beta", ".b", ".b1", ".b2")):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
            v.copy_(v.float().double())          # weights are exactly representable in f32
        for k in orc.s.state:
            if k.endswith(".rm"):
                orc.s.state[k] = torch.randn(orc.s.state[k].shape, generator=g, dtype=torch.float64) * 0.1
            elif k.endswith(".rv"):
                orc.s.state[k] = torch.rand(orc.s.state[k].shape, generator=g, dtype=torch.float64) + 0.5
            orc.s.state[k] = orc.s.state[k].float().double()
    x = synthetic_batch(B, seed=13)
    eps = [torch.randn(s, generator=g, dtype=torch.float64).float().double() for s in orc.eps_shapes(B)]
    return orc, x, eps


def checksum(orc):
    """Order-independent fingerprint of every weight/state tensor: guards the regeneration."""
    tot = []
    for d in (orc.s.params, orc.s.state):
        for k in sorted(d):
            v = d[k].detach()
            tot.append([float(v.sum()), float((v * v).sum()), float(v.abs().max())])
    return np.asarray(tot)


def main():
    orc, x, eps = build()
    out = {"x": x.numpy().astype(np.float32), "checksum": checksum(orc)}
    for i, e in enumerate(eps):
        out[f"eps{i}"] = e.numpy().astype(np.float32)
    # ---- inference forward with IWAE terms, and sampling (before any update)
    logits, zp, lp, lq, _ = orc.call(x, eps, training=False, nll=True)
    out["inf/logits"], out["inf/log_p"], out["inf/log_q"] = logits.detach().numpy(), lp.detach().numpy(), lq.detach().numpy()
    for i, p in enumerate(zp):
        out[f"inf/enc_mu{i}"], out[f"inf/enc_sigma{i}"] = p.enc_mu.detach().numpy(), p.enc_sigma.detach().numpy()
        out[f"inf/dec_mu{i}"], out[f"inf/dec_sigma{i}"] = p.dec_mu.detach().numpy(), p.dec_sigma.detach().numpy()
    out["inf/recon_crop"] = orc.calculate_recon_loss(x, logits, crop_output=True).detach().numpy()
    out["sample/t0.7"] = orc.sample(B, 0.7, eps).detach().numpy()
    # ---- one training step (SN + batch-stat BN + balanced KL at beta = 1/3 + Adamax)
    orc.steps = 100
    o = orc.train_step(x, eps, decay_steps=1000)
    for k in ("loss", "reconstruction_loss", "kl_loss", "bn_loss", "kl_per_group", "kl_coeff"):
        out["train/" + k] = np.asarray(o[k].detach().numpy())
    for k in GRAD_KEYS:
        out["grad/" + k] = o["grads"][k].detach().numpy()
        out["updated/" + k] = orc.s.params[k].detach().numpy()
    out["state_after/enc.g0.c0.bn1.rm"] = orc.s.state["enc.g0.c0.bn1.rm"].numpy()
    out["state_after/enc.g0.c0.bn1.rv"] = orc.s.state["enc.g0.c0.bn1.rv"].numpy()
    out["state_after/post.cell3.conv5.u"] = orc.s.state["post.cell3.conv5.u"].numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nvae_small.npz")
    out = {k: (np.asarray(v, dtype=np.float32) if k != "checksum" else v) for k, v in out.items()}
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1e6, "MB", orc.n_trainable(), "params")


if __name__ == "__main__":
    main()
