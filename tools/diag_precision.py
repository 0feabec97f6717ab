"""Cross-precision ELBO of the SAME weights (trained in f32 / in bf16) on a held-out batch with fixed noise, in eval mode
(moving BN statistics) and in training mode (batch statistics), at several points of training.  Picks the regime
tests/test_precision_gpu.py asserts in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import configs
from nvae_tf_amd.datasets import synthetic_mnist

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
marks = [int(a) for a in sys.argv[2:]] or [240, 600, 1200, 2000]
nb = 8
raw = torch.from_numpy(synthetic_mnist(B * (nb + 1), 1)[0]).float()
data = torch.zeros(B * (nb + 1), 32, 32, 1); data[:, 2:30, 2:30, 0] = (raw > 0).float()
data = data.to(dev)
held_out = data[nb * B:]
models = {}
for tag, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
    m = configs.build("mnist_c2", batch=B, device=dev, dtype=dt, total_epochs=1, n_total_iterations=3000, seed=1)
    m.capture_train_step((B, 32, 32, 1))
    models[tag] = m
g = torch.Generator().manual_seed(77)
eps = [torch.randn(s, generator=g) for s in models["f32"].eps_shapes(B)]


def neg_elbo(m, training):
    logits, _, lp, lq = m(held_out, nll=True, eps_list=eps, training=training)
    return float((m.calculate_recon_loss(held_out, logits) + lq - lp).mean())


for i in range(max(marks)):
    x = data[(i % nb) * B:(i % nb + 1) * B]
    for m in models.values():
        out = m.train_step_graphed(x)
    if i + 1 in marks:
        torch.cuda.synchronize()
        for weights_of in ("f32", "bf16"):
            src = models[weights_of]
            vals = {}
            for run_in, m in models.items():
                keep = (m.ps.params.clone(), m.ps.state.clone())
                m.ps.params.copy_(src.ps.params); m.ps.state.copy_(src.ps.state)
                vals[run_in] = (neg_elbo(m, False), neg_elbo(m, True))
                m.ps.params.copy_(keep[0]); m.ps.state.copy_(keep[1])
            print(f"step {i + 1:5d} weights of {weights_of:4s}: eval-mode -ELBO f32 {vals['f32'][0]:10.3f} bf16 {vals['bf16'][0]:10.3f} (diff {vals['bf16'][0] - vals['f32'][0]:+8.3f}) | "
                  f"batch-stat -ELBO f32 {vals['f32'][1]:10.3f} bf16 {vals['bf16'][1]:10.3f} (diff {vals['bf16'][1] - vals['f32'][1]:+8.3f})", flush=True)
