"""GPU: the HIP path (f32 compute) against the committed golden vectors, through the model API.
Tolerances: 1e-3 relative on outputs/losses, 5e-3 of the gradient scale (f32 kernels vs an fp64
oracle through ~40 layers with batch-2 BatchNorm)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))


def rel(a, b):
    a = np.asarray(a.detach().double().cpu() if torch.is_tensor(a) else a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12))


def build_model(dev):
    import make_golden
    from nvae_tf_amd.models import NVAE
    c = make_golden.CFG
    orc, x, eps = make_golden.build()
    m = NVAE(c["n_encoder_channels"], c["n_decoder_channels"], c["res_cells_per_group"], c["n_preprocess_blocks"],
             c["n_preprocess_cells"], c["n_latent_per_group"], len(c["n_groups_per_scale"]), c["n_groups_per_scale"],
             c["n_postprocess_blocks"], c["n_post_process_cells"], c["sr_lambda"], c["scale_factor"], c["total_epochs"],
             c["n_total_iterations"], c["step_based_warmup"], [make_golden.B, 32, 32, 1], device=dev,
             dtype=torch.float32)
    m.ps.load_named(orc.s.params, orc.s.state)
    return m, x.float(), [e.float() for e in eps], make_golden


def test_golden_inference_and_sampling(lib, dev):
    gold = np.load(os.path.join(HERE, "golden", "nvae_small.npz"))
    m, x, eps, mg = build_model(dev)
    logits, zp, lp, lq = m(x, nll=True, eps_list=eps)
    assert rel(logits, gold["inf/logits"]) < 1e-3
    assert rel(lp, gold["inf/log_p"]) < 1e-3 and rel(lq, gold["inf/log_q"]) < 1e-3
    for i, p in enumerate(zp):
        assert rel(p.enc_mu, gold[f"inf/enc_mu{i}"]) < 1e-3 and rel(p.enc_sigma, gold[f"inf/enc_sigma{i}"]) < 1e-3
        assert rel(p.dec_mu, gold[f"inf/dec_mu{i}"]) < 1e-3 and rel(p.dec_sigma, gold[f"inf/dec_sigma{i}"]) < 1e-3
    assert rel(m.calculate_recon_loss(x, logits, crop_output=True), gold["inf/recon_crop"]) < 1e-3
    img, _, _, _ = m.sample(mg.B, 0.7, eps_list=eps)
    assert rel(img, gold["sample/t0.7"]) < 1e-3


def test_golden_train_step(lib, dev):
    gold = np.load(os.path.join(HERE, "golden", "nvae_small.npz"))
    m, x, eps, mg = build_model(dev)
    m.steps = 100
    m.lr_decay_steps = 1000
    out = m.train_step(x, eps)
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(gold["train/loss"])) / float(gold["train/loss"]) < 1e-3
    assert rel(out["reconstruction_loss"], gold["train/reconstruction_loss"]) < 1e-3
    assert rel(out["kl_per_group"], gold["train/kl_per_group"]) < 3e-3
    assert rel(out["kl_loss"], gold["train/kl_loss"]) < 3e-3
    assert abs(float(out["bn_loss"]) - float(gold["train/bn_loss"])) < 1e-5
    assert rel(m.coeff, gold["train/kl_coeff"]) < 3e-3
    for k in mg.GRAD_KEYS:
        assert rel(m.ps.get_grad(k), gold["grad/" + k]) < 5e-3, k
        g = np.abs(gold["grad/" + k]) > 1e-4        # f32 noise on g stays < 0.4 % of lr
        d = np.abs(m.ps.get(k).double().cpu().numpy() - gold["updated/" + k])
        assert float((d * g).max()) < 2e-5, k
    assert rel(m.ps.get_state("enc.g0.c0.bn1.rm"), gold["state_after/enc.g0.c0.bn1.rm"]) < 1e-3
    assert rel(m.ps.get_state("enc.g0.c0.bn1.rv"), gold["state_after/enc.g0.c0.bn1.rv"]) < 1e-3
    assert rel(m.ps.get_state("post.cell3.conv5.u"), gold["state_after/post.cell3.conv5.u"]) < 1e-3


def test_golden_rgb_mixture_of_logistics(lib, dev):
    """RGB / mixture-of-logistics fixture: inference + IWAE terms, NLL, sampling with fixed uniforms and
    one training step of the f32 HIP path against tests/golden/nvae_rgb_small.npz."""
    import make_golden_rgb as mg
    from nvae_tf_amd.models import NVAE
    gold = np.load(os.path.join(HERE, "golden", "nvae_rgb_small.npz"))
    c = mg.CFG
    orc, x, eps, u_mix, u_pix = mg.build()
    m = NVAE(c["n_encoder_channels"], c["n_decoder_channels"], c["res_cells_per_group"], c["n_preprocess_blocks"],
             c["n_preprocess_cells"], c["n_latent_per_group"], len(c["n_groups_per_scale"]), c["n_groups_per_scale"],
             c["n_postprocess_blocks"], c["n_post_process_cells"], c["sr_lambda"], c["scale_factor"], c["total_epochs"],
             c["n_total_iterations"], c["step_based_warmup"], [mg.B, mg.HW, mg.HW, 3], device=dev, dtype=torch.float32,
             num_mixture_dec=mg.M)
    m.ps.load_named(orc.s.params, orc.s.state)
    xf, ef = x.float(), [e.float() for e in eps]
    logits, zp, lp, lq = m(xf, nll=True, eps_list=ef)
    assert rel(logits, gold["inf/logits"]) < 1e-3 and rel(lp, gold["inf/log_p"]) < 1e-3 and rel(lq, gold["inf/log_q"]) < 1e-3
    assert rel(m.calculate_recon_loss(xf, logits), gold["inf/recon"]) < 1e-3
    img, *_ = m.sample(mg.B, 0.8, eps_list=ef, dmol_noise=(u_mix.float(), u_pix.float()))
    bad = (np.abs(img.cpu().numpy() - gold["sample/image_t0.8"]).max(axis=-1) > 1e-3).mean()
    assert bad < 0.01            # a mixture pick may flip where two Gumbel scores tie within f32 noise
    m.steps = 100
    m.lr_decay_steps = 1000
    out = m.train_step(xf, ef)
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(gold["train/loss"])) / float(gold["train/loss"]) < 1e-3
    assert rel(out["reconstruction_loss"], gold["train/reconstruction_loss"]) < 1e-3
    assert rel(out["kl_per_group"], gold["train/kl_per_group"]) < 3e-3
    for k in mg.GRAD_KEYS:
        assert rel(m.ps.get_grad(k), gold["grad/" + k]) < 5e-3, k
