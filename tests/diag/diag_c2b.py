"""Diagnostic: C2 train-step parity at batch 8 (f32 HIP vs fp64 oracle): loss, KL groups, gradient cosine, worst tensors."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_model_gpu as T
dev = "cuda:0"
cfg = dict(T.CFG, n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=2, n_preprocess_cells=3,
           n_post_process_cells=3, n_groups_per_scale=[5, 10])
T.B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
orc, model, x, eps = T.build_pair(dev, torch.float32, cfg)
orc.steps = model.steps = 100
out_o = orc.train_step(x, eps, decay_steps=1000)
out = model.train_step(x.float(), [e.float() for e in eps])
torch.cuda.synchronize()
print("loss", float(out["loss"]), float(out_o["loss"]))
worst = sorted(((T.rel(model.ps.get_grad(k), g_o), float(g_o.abs().max()), k) for k, g_o in out_o["grads"].items()), reverse=True)
for w in worst[:15]: print("  ", w)
go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
gp = torch.cat([model.ps.get_grad(k).double().cpu().reshape(-1) for k in out_o["grads"]])
print("cos", float((go * gp).sum() / (go.norm() * gp.norm())), "norms", float(go.norm()), float(gp.norm()))
import collections
bad = [w for w in worst if w[0] > 2e-2 and w[1] > 1e-6]
print("tensors with rel err > 2e-2:", len(bad), "of", len(worst))
