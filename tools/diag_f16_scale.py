"""Which static loss scale does the float16 path need?  One training step (update=False) of a configuration at its full
batch with fixed noise, for a sweep of scales: fraction of finite gradient elements, and the cosine of the (unscaled)
f16 gradient with the bf16 gradient of the same weights.  usage: python tools/diag_f16_scale.py [workload] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import configs

name = sys.argv[1] if len(sys.argv) > 1 else "celeba64"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
dev = torch.device("cuda:0")
c = configs.CONFIGS[name]
B, (H, W, C) = c["batch"], c["input_hwc"]
g = torch.Generator().manual_seed(3)
if C == 3:
    x = (torch.randint(0, 256, (B, H, W, 3), generator=g).float() / 255.0).to(dev)
else:
    x = torch.zeros(B, H, W, 1); x[:, 2:30, 2:30, 0] = (torch.rand(B, 28, 28, generator=g) < 0.19).float(); x = x.to(dev)


def grads(dtype, scale):
    m = configs.build(name, device=dev, dtype=dtype, loss_scale=scale)
    m.steps = steps
    ge = torch.Generator().manual_seed(11)
    eps = [torch.randn(s, generator=ge) for s in m.eps_shapes(B)]
    out = m.train_step(x, eps_list=eps, update=False)
    torch.cuda.synchronize()
    gr = m.ps.grads.clone() / scale
    loss = float(out["loss"])
    del m
    torch.cuda.empty_cache()
    return gr, loss


ref, loss_ref = grads(torch.bfloat16, 1.0)
ref32, loss32 = grads(torch.float32, 1.0)
cos = lambda a, b: float((a.double() * b.double()).sum() / (a.double().norm() * b.double().norm()))
print(f"{name} beta-step {steps}: f32 loss {loss32:.4f} |g| {float(ref32.norm()):.3e} max|g| {float(ref32.abs().max()):.3e}; bf16 loss {loss_ref:.4f} cos(bf16, f32) {cos(ref, ref32):.6f}")
for e in (0, -4, -8, -12, -16, 4, 8):
    s = 2.0 ** e
    gr, loss = grads(torch.float16, s)
    fin = torch.isfinite(gr)
    frac = float(fin.float().mean())
    gz = torch.where(fin, gr, torch.zeros_like(gr))
    print(f"loss_scale 2^{e:+d}: loss {loss:.4f} finite {frac * 100:7.3f} %  |g| {float(gz.norm()):.3e}  cos(f16, f32) {cos(gz, ref32):.6f}  "
          f"zero-fraction {float((gz == 0).float().mean()):.4f} (f32: {float((ref32 == 0).float().mean()):.4f})", flush=True)
