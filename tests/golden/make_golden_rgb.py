"""Generate tests/golden/nvae_rgb_small.npz from the CPU oracle (fp64): a shrunken RGB NVAE with the
discretised mixture-of-logistics head (BASELINE.json configs[3]/[4] family; SURVEY 8f "ext").  The
reference implements no RGB head, so these vectors pin the oracle's specification (dmol_log_prob /
dmol_sample) against regressions and the HIP path against it.

    python tests/golden/make_golden_rgb.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from make_golden import checksum  # noqa: E402
from oracle.nvae_oracle import OracleConfig, OracleNVAE, dmol_sample, synthetic_rgb_batch  # noqa: E402

CFG = dict(n_encoder_channels=8, n_decoder_channels=8, res_cells_per_group=1, n_preprocess_blocks=1,
           n_preprocess_cells=2, n_latent_per_group=4, n_groups_per_scale=[1, 2], n_postprocess_blocks=1,
           n_post_process_cells=2, sr_lambda=0.01, scale_factor=2, total_epochs=10, n_total_iterations=1000,
           step_based_warmup=True)
B, HW, M = 2, 16, 10
GRAD_KEYS = ["pre.stem.w", "post.final.conv.w", "post.final.conv.b", "dec.g1.c0.dw.w", "enc.g0.c0.bn1.gamma", "dec.h"]


def build():
    orc = OracleNVAE(OracleConfig(**CFG, input_hw=HW, input_channels=3, head="dmol", num_mixture_dec=M),
                     dtype=torch.float64, seed=21)
    g = torch.Generator().manual_seed(22)
    with torch.no_grad():
        for k, v in orc.s.params.items():
            if k.endswith(".gamma"):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
            elif k.endswith((".beta", ".b", ".b1", ".b2")):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
            v.copy_(v.float().double())
        for k in orc.s.state:
            if k.endswith(".rm"):
                orc.s.state[k] = torch.randn(orc.s.state[k].shape, generator=g, dtype=torch.float64) * 0.1
            elif k.endswith(".rv"):
                orc.s.state[k] = torch.rand(orc.s.state[k].shape, generator=g, dtype=torch.float64) + 0.5
            orc.s.state[k] = orc.s.state[k].float().double()
    x = synthetic_rgb_batch(B, hw=HW, seed=23).float().double()
    eps = [torch.randn(s, generator=g, dtype=torch.float64).float().double() for s in orc.eps_shapes(B)]
    u_mix = torch.rand(B, HW, HW, M, generator=g, dtype=torch.float64).clamp(1e-5, 1 - 1e-5).float().double()
    u_pix = torch.rand(B, HW, HW, 3, generator=g, dtype=torch.float64).clamp(1e-5, 1 - 1e-5).float().double()
    return orc, x, eps, u_mix, u_pix


def main():
    orc, x, eps, u_mix, u_pix = build()
    out = {"x": x.numpy(), "checksum": checksum(orc), "u_mix": u_mix.numpy(), "u_pix": u_pix.numpy()}
    for i, e in enumerate(eps):
        out[f"eps{i}"] = e.numpy()
    logits, zp, lp, lq, _ = orc.call(x, eps, training=False, nll=True)
    out["inf/logits"], out["inf/log_p"], out["inf/log_q"] = logits.detach().numpy(), lp.detach().numpy(), lq.detach().numpy()
    out["inf/recon"] = orc.calculate_recon_loss(x, logits).detach().numpy()
    out["sample/logits_t0.8"] = orc.sample(B, 0.8, eps).detach().numpy()
    out["sample/image_t0.8"] = dmol_sample(orc.sample(B, 0.8, eps).detach(), M, u_mix, u_pix, 1.0).numpy()
    orc.steps = 100
    o = orc.train_step(x, eps, decay_steps=1000)
    for k in ("loss", "reconstruction_loss", "kl_loss", "bn_loss", "kl_per_group", "kl_coeff"):
        out["train/" + k] = np.asarray(o[k].detach().numpy())
    for k in GRAD_KEYS:
        out["grad/" + k] = o["grads"][k].detach().numpy()
    path = os.path.join(HERE, "nvae_rgb_small.npz")
    out = {k: (np.asarray(v, dtype=np.float32) if k != "checksum" else v) for k, v in out.items()}
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path) / 1e6, "MB", orc.n_trainable(), "params")


if __name__ == "__main__":
    main()
