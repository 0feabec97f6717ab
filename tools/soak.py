"""Soak run: N hipGraph-replayed training steps on fresh synthetic batches; prints the loss curve and
checks that every loss / parameter stays finite.  usage: python tools/soak.py [workload] [steps] [batch]"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import configs
from nvae_tf_amd.datasets import synthetic_mnist, synthetic_rgb

name = sys.argv[1] if len(sys.argv) > 1 else "mnist_c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
c = configs.CONFIGS[name]
B = int(sys.argv[3]) if len(sys.argv) > 3 else c["batch"]
dev = torch.device("cuda:0")
model = configs.build(name, batch=B, device=dev, dtype=torch.bfloat16, total_epochs=1, n_total_iterations=steps)
H, W, C = c["input_hwc"]
n = B * 16
if C == 1:
    raw = torch.from_numpy(synthetic_mnist(n, 1)[0]).float()
    data = torch.zeros(n, 32, 32, 1); data[:, 2:30, 2:30, 0] = (raw > 0).float()
else:
    data = torch.from_numpy(synthetic_rgb(n, H, 1)[0]).float() / 255.0
data = data.to(dev)
model.capture_train_step((B, H, W, C), warmup=1)
model.steps = model.opt_iterations = 0
curve = []
for i in range(steps):
    x = data[(i % 16) * B:(i % 16 + 1) * B]
    out = model.train_step_graphed(x)
    if i % max(steps // 15, 1) == 0 or i == steps - 1:
        l = float(out["loss"])
        curve.append((i, l, float(out["reconstruction_loss"].mean()), float(out["kl_loss"].mean())))
        print(f"step {i:5d} loss {l:10.3f} recon {curve[-1][2]:10.3f} beta*kl {curve[-1][3]:9.3f}", flush=True)
        assert math.isfinite(l), "non-finite loss"
torch.cuda.synchronize()
assert bool(torch.isfinite(model.ps.params).all()) and bool(torch.isfinite(model.ps.state).all())
assert curve[-1][2] < curve[0][2], "reconstruction loss did not decrease"     # the total adds beta * KL with beta: 0 -> 1
print("OK", name, "steps", steps, "first", curve[0][1], "last", curve[-1][1])
