"""Which gradient tensors of the shrunken parity model are ill-conditioned under bf16 storage, and by how much?
The fp64 oracle is run with its stored activations / conv weights rounded to bf16 (OracleNVAE.act_round, jittered by a
fraction of an ulp: tests/test_model_gpu.py::bf16_spread) and each gradient tensor is compared with the exact one; with
a GPU the HIP bf16 path's errors are printed beside it.  VERDICT r02 'weak' 2: post.cell0.se.b1 / bn3.beta / se.w1 at
1.00 / 0.53 / 0.37 in one run of four.   usage: python tests/diag/diag_bf16_spread.py > profiles/r03_bf16_spread.txt"""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("tm", os.path.join(ROOT, "tests", "test_model_gpu.py"))
tm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tm)

gpu = torch.cuda.is_available()
dev = torch.device("cuda:0") if gpu else None
if gpu:
    orc, model, x, eps = tm.build_pair(dev, torch.bfloat16)
else:
    from oracle.nvae_oracle import OracleConfig, OracleNVAE, synthetic_batch
    orc = OracleNVAE(OracleConfig(**tm.CFG), dtype=torch.float64, seed=5)
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for k, v in orc.s.params.items():
            if k.endswith(".gamma"):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
            elif k.endswith((".beta", ".b", ".b1", ".b2")):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
    x = synthetic_batch(tm.B, seed=3)
    eg = torch.Generator().manual_seed(8)
    eps = [torch.randn(s, generator=eg, dtype=torch.float64) for s in orc.eps_shapes(tm.B)]
orc.steps = 100
snap = ({k: v.detach().clone() for k, v in orc.s.params.items()}, {k: v.clone() for k, v in orc.s.state.items()})
out_o = orc.train_step(x, eps, decay_steps=1000)
orc.steps = 100
spread = tm.bf16_spread(orc, snap, x, eps, out_o["grads"], runs=12)
hip = {}
if gpu:
    for run in range(4):
        _, m2, _, _ = tm.build_pair(dev, torch.bfloat16)
        m2.steps = 100
        m2.train_step(x.float(), [e.float() for e in eps])
        torch.cuda.synchronize()
        for k in out_o["grads"]:
            hip.setdefault(k, []).append(tm.rel(m2.ps.get_grad(k), out_o["grads"][k]))
real = [k for k in out_o["grads"] if float(out_o["grads"][k].abs().max()) > 1e-6]
vals = sorted(spread[k] for k in real)
print(f"{len(real)} gradient tensors with a real gradient; oracle-with-bf16-storage vs exact fp64 oracle, max over 12 runs:")
print(f"  median {vals[len(vals) // 2]:.3e}   p90 {vals[len(vals) * 9 // 10]:.3e}   p98 {vals[len(vals) * 98 // 100]:.3e}   max {vals[-1]:.3e}")
if hip:
    hv = sorted(max(hip[k]) for k in real)
    print("HIP bf16 path vs exact fp64 oracle, max over 4 runs:")
    print(f"  median {hv[len(hv) // 2]:.3e}   p90 {hv[len(hv) * 9 // 10]:.3e}   p98 {hv[len(hv) * 98 // 100]:.3e}   max {hv[-1]:.3e}")
print(f"{'tensor':34s} {'emulated bf16 spread':>22s} {'HIP bf16 (4 runs)':>40s}")
for k in sorted(real, key=lambda k: -spread[k])[:20]:
    h = "  ".join(f"{e:.3f}" for e in hip.get(k, []))
    print(f"{k:34s} {spread[k]:22.3f} {h:>40s}")
