"""Diagnostic: eager vs hipGraph-replayed training steps from identical state; prints per-tensor differences."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_model_gpu as T

dev = "cuda:0"
_, me, x, eps = T.build_pair(dev, torch.float32)
_, mg, _, _ = T.build_pair(dev, torch.float32)
_, me2, _, _ = T.build_pair(dev, torch.float32)
xs = x.float()
for m in (me, mg, me2):
    m.steps = 50
warm = int(os.environ.get("WARM", "2"))
mg.capture_train_step(xs.shape, warmup=warm)
for step in range(3):
    o1 = me.train_step(xs); o2 = mg.train_step_graphed(xs); o3 = me2.train_step(xs)
    torch.cuda.synchronize()
    print("step", step, "loss eager", float(o1["loss"]), "graph", float(o2["loss"]), "eager2", float(o3["loss"]))
    for name, a, b in (("graph-vs-eager", mg, me), ("eager2-vs-eager", me2, me)):
        dg = (a.ps.grads - b.ps.grads).abs()
        dp = (a.ps.params - b.ps.params).abs()
        print("  ", name, "grad maxdiff", float(dg.max()), "rel", float(dg.max() / b.ps.grads.abs().max()),
              "param q95", float(torch.quantile(dp[:1 << 20], 0.95)), "q50", float(torch.quantile(dp[:1 << 20], 0.5)), "max", float(dp.max()))
    if step == 0:
        worst = []
        for k in me.ps.slots:
            ga, gb = mg.ps.get_grad(k), me.ps.get_grad(k)
            worst.append((float((ga - gb).abs().max() / (gb.abs().max() + 1e-20)), float(gb.abs().max()), k))
        worst.sort(reverse=True)
        for w in worst[:12]:
            print("   ", w)
