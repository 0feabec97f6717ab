"""Diagnostic: C2 architecture forward parity (f32 HIP vs fp64 oracle) at several batch sizes: loss terms per group."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_model_gpu as T

dev = "cuda:0"
cfg = dict(T.CFG, n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=2, n_preprocess_cells=3,
           n_post_process_cells=3, n_groups_per_scale=[5, 10])
for B in [int(a) for a in sys.argv[1:]] or [2, 8]:
    T.B = B
    orc, model, x, eps = T.build_pair(dev, torch.float32, cfg)
    orc.steps = model.steps = 100
    with torch.no_grad():
        orc.spectral_norm_step()
        out_o = orc.loss(x, eps, training=True)
    out = model.train_step(x.float(), [e.float() for e in eps], update=False)
    torch.cuda.synchronize()
    print("B", B, "loss", float(out["loss"]), float(out_o["loss"]))
    print("  recon rel", T.rel(out["reconstruction_loss"], out_o["reconstruction_loss"]))
    for gi in range(15):
        print("  kl group", gi, T.rel(out["kl_per_group"][gi], out_o["kl_per_group"][gi]),
              float(out["kl_per_group"][gi].mean()), float(out_o["kl_per_group"][gi].mean()))
