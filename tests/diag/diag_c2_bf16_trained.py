"""How well conditioned is the C2 architecture for a bf16-vs-fp64 gradient comparison once it has TRAINED a little?
At its random initialisation (batch 2) the bf16 gradient of ANY implementation is noise (HIP cosine 0.25, the oracle with
bf16 storage -0.09: tests/test_model_gpu.py::test_c2_architecture_parity).  Here the HIP f32 path trains the model for N
graph-replayed steps at batch 128, then one training-mode forward + backward at batch 4 is compared: HIP bf16 and the
oracle-with-bf16-storage, both against the fp64 oracle.   usage: python tests/diag/diag_c2_bf16_trained.py [steps]"""
import importlib.util, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("tm", os.path.join(ROOT, "tests", "test_model_gpu.py"))
tm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tm)
from nvae_tf_amd.models import NVAE
from oracle.nvae_oracle import synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
cfg = dict(tm.CFG, n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=2, n_preprocess_cells=3,
           n_post_process_cells=3, n_groups_per_scale=[5, 10])
tm.B = 4
orc, m32, x, eps = tm.build_pair(dev, torch.float32, cfg)


def make(dtype, B):
    return NVAE(cfg["n_encoder_channels"], cfg["n_decoder_channels"], cfg["res_cells_per_group"], cfg["n_preprocess_blocks"],
                cfg["n_preprocess_cells"], cfg["n_latent_per_group"], 2, cfg["n_groups_per_scale"], cfg["n_postprocess_blocks"],
                cfg["n_post_process_cells"], cfg["sr_lambda"], cfg["scale_factor"], cfg["total_epochs"], 20000, True,
                [B, 32, 32, 1], device=dev, dtype=dtype)
trainer = make(torch.float32, 128)
trainer.ps.load_named(orc.s.params, orc.s.state)
xb = synthetic_batch(128, seed=11).float().to(dev)
trainer.capture_train_step(xb.shape, warmup=1)
trainer._static_x.copy_(xb)
t0 = time.time()
for i in range(steps):
    out = trainer.train_step_graphed(None)
torch.cuda.synchronize()
print(f"trained {steps} f32 steps at batch 128 in {time.time() - t0:.1f} s, loss {float(out['loss']):.1f}", flush=True)
params = {k: v.detach().double().cpu() for k, v in trainer.ps.named().items()}
state = {k: v.detach().double().cpu() for k, v in trainer.ps.named_state().items()}
for k in orc.s.params:
    orc.s.params[k] = params[k].clone().requires_grad_(True)
for k in orc.s.state:
    orc.s.state[k] = state[k].clone()
snap = ({k: v.detach().clone() for k, v in orc.s.params.items()}, {k: v.clone() for k, v in orc.s.state.items()})
orc.steps = 6000
t0 = time.time()
out_o = orc.train_step(x, eps, decay_steps=20000)
print(f"fp64 oracle step at batch {tm.B}: {time.time() - t0:.1f} s, loss {float(out_o['loss']):.2f}, |g| "
      f"{float(torch.cat([g.reshape(-1) for g in out_o['grads'].values()]).norm()):.3e}", flush=True)
orc.steps = 6000
out_e, g_e = tm.oracle_16bit(orc, snap, x, eps)
go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
names = [k for k in out_o["grads"] if float(out_o["grads"][k].abs().max()) > 1e-6]


def stats(get):
    gq = torch.cat([get(k).double().cpu().reshape(-1) for k in out_o["grads"]])
    es = sorted(tm.rel(get(k), out_o["grads"][k]) for k in names)
    return float((go * gq).sum() / (go.norm() * gq.norm())), es[len(es) // 2], es[len(es) * 9 // 10]
for dtype in (torch.float32, torch.bfloat16):
    m = make(dtype, tm.B)
    m.ps.load_named(snap[0], snap[1])
    m.steps = 6000
    out = m.train_step(x.float(), [e.float() for e in eps])
    torch.cuda.synchronize()
    c, med, p90 = stats(lambda k: m.ps.get_grad(k))
    lo = float(out_o["loss"])
    kl = max(tm.rel(out["kl_per_group"][gi], out_o["kl_per_group"][gi]) for gi in range(15))
    print(f"HIP {dtype}: loss rel {abs(float(out['loss']) - lo) / abs(lo):.2e}, worst KL group {kl:.2e}, gradient cosine {c:.5f}, "
          f"per-tensor median {med:.2e} p90 {p90:.2e}", flush=True)
c, med, p90 = stats(lambda k: g_e[k])
kl = max(tm.rel(out_e["kl_per_group"][gi], out_o["kl_per_group"][gi]) for gi in range(15))
print(f"oracle with bf16 storage: loss rel {abs(float(out_e['loss']) - float(out_o['loss'])) / abs(float(out_o['loss'])):.2e}, "
      f"worst KL group {kl:.2e}, gradient cosine {c:.5f}, per-tensor median {med:.2e} p90 {p90:.2e}")
