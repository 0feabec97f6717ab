"""float16 training from step 0 (beta = 0, KL warm-up as in real training): loss curve next to bf16 with the same seed /
data / noise, and whether every gradient element stays finite.  usage: diag_f16_train.py [workload] [steps] [loss_scale]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import configs

name = sys.argv[1] if len(sys.argv) > 1 else "mnist_c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
scale = (None if sys.argv[3] == "dynamic" else float(sys.argv[3])) if len(sys.argv) > 3 else None
dev = torch.device("cuda:0")
c = configs.CONFIGS[name]
B, (H, W, C) = c["batch"], c["input_hwc"]
g = torch.Generator().manual_seed(3)
nb = 8
if C == 3:
    data = (torch.randint(0, 256, (nb * B, H, W, 3), generator=g).float() / 255.0).to(dev)
else:
    data = torch.zeros(nb * B, H, W, 1); data[:, 2:30, 2:30, 0] = (torch.rand(nb * B, 28, 28, generator=g) < 0.19).float(); data = data.to(dev)
models = {}
for tag, dt, s in (("f16", torch.float16, scale), ("bf16", torch.bfloat16, 1.0)):
    m = configs.build(name, device=dev, dtype=dt, loss_scale=s, total_epochs=1, n_total_iterations=3000)
    m.capture_train_step((B, H, W, C))
    models[tag] = m
every = max(steps // 25, 1)
bad = 0
for i in range(steps):
    x = data[(i % nb) * B:(i % nb + 1) * B]
    outs = {t: m.train_step_graphed(x) for t, m in models.items()}
    nf = int((~torch.isfinite(models["f16"].ps.grads)).sum())
    bad += nf > 0
    if i % every == 0 or (nf and bad < 30):
        a, b = outs["f16"], outs["bf16"]
        gmax = float(models["f16"].ps.grads.abs().max())
        print(f"step {i:4d} f16 loss {float(a['loss']):12.3f} recon {float(a['reconstruction_loss'].mean()):10.3f} kl {float(a['kl_per_group'].sum(0).mean()):12.3f} | "
              f"bf16 loss {float(b['loss']):12.3f} kl {float(b['kl_per_group'].sum(0).mean()):12.3f} | non-finite f16 grads {nf} max|g*scale| {gmax:.3e} "
              f"loss scale 2^{math.log2(float(models['f16'].hyper[4])):.0f}", flush=True)
print("steps with non-finite gradients:", bad, "params finite:", bool(torch.isfinite(models["f16"].ps.params).all()))
