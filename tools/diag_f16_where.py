"""Where do the float16 gradients of a configuration stop being finite?  One training step at step 0 (update=False)
with a static loss scale; per parameter-name prefix: tensors, tensors with a non-finite gradient, largest finite |g|
(unscaled), next to the bf16 run's largest |g|.  usage: diag_f16_where.py [workload] [log2 scale]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import configs

name = sys.argv[1] if len(sys.argv) > 1 else "celeba64"
e = int(sys.argv[2]) if len(sys.argv) > 2 else -12
dev = torch.device("cuda:0")
c = configs.CONFIGS[name]
B, (H, W, C) = c["batch"], c["input_hwc"]
g = torch.Generator().manual_seed(3)
x = (torch.randint(0, 256, (B, H, W, 3), generator=g).float() / 255.0).to(dev) if C == 3 else None
if x is None:
    x = torch.zeros(B, H, W, 1); x[:, 2:30, 2:30, 0] = (torch.rand(B, 28, 28, generator=g) < 0.19).float(); x = x.to(dev)


def run(dtype, scale):
    m = configs.build(name, device=dev, dtype=dtype, loss_scale=scale)
    m.steps = 2000          # beta = 0.04, as the full-batch test (training really begins in the KL warm-up)
    print(f"{dtype}: gradient range normalisation {'on' if m.grad_rescale else 'off'}", flush=True)
    ge = torch.Generator().manual_seed(11)
    eps = [torch.randn(s, generator=ge) for s in m.eps_shapes(B)]
    m.train_step(x, eps_list=eps, update=False)
    torch.cuda.synchronize()
    out = {k: (m.ps.get_grad(k) / scale).float().cpu() for k in m.ps.slots}
    del m; torch.cuda.empty_cache()
    return out


ref = run(torch.bfloat16, 1.0)
f16 = run(torch.float16, 2.0 ** e)
def prefix(k):
    p = k.split(".")
    return ".".join(p[:2]) if p[0] in ("enc", "dec") else p[0] + "." + p[1] if len(p) > 2 else p[0]
agg = collections.OrderedDict()
for k in f16:
    a = agg.setdefault(prefix(k), [0, 0, 0.0, 0.0])
    t = f16[k]
    a[0] += 1; a[1] += int(not bool(torch.isfinite(t).all()))
    fin = t[torch.isfinite(t)]
    a[2] = max(a[2], float(fin.abs().max()) if fin.numel() else 0.0)
    a[3] = max(a[3], float(ref[k].abs().max()))
bad = [k for k, a in agg.items() if a[1]]
print(f"SUMMARY target 2^{os.environ.get('NVAE_GRAD_TARGET_LOG2', '6')} loss scale 2^{e}: {len(bad)} of {len(agg)} prefixes hold a "
      f"non-finite gradient; first (in forward order) {bad[0] if bad else None}, last {bad[-1] if bad else None}", flush=True)
print(f"{name}, f16 loss scale 2^{e}: prefix  tensors  non-finite  max finite |g| (f16)  max |g| (bf16)")
for k, a in agg.items():
    print(f"  {k:22s} {a[0]:4d} {a[1]:4d}   {a[2]:10.3e}   {a[3]:10.3e}")
