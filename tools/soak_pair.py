"""bf16 against f32 training, same seed, same synthetic data, same in-graph noise stream: loss curves side by side
and the gap of the trailing-window means.  usage: python tools/soak_pair.py [workload] [steps] [batch] [json_out]

The two runs start bit-identical (f32 master weights, identical Philox noise) and drift apart chaotically, as any
two roundings of the same training run do; what must hold is that bf16 optimises the same objective equally
well: the ELBO levels (means over a trailing window of steps, and a fixed held-out batch under fixed noise at the
end) agree within the run-to-run spread of f32 training itself.  That spread is measured by the CONTROL pair:
the same two models again, same initial weights and data, only the Philox key of the in-graph noise differs
(tags "bf16/B", "f32/B"); |f32 - f32/B| is what two equally valid f32 runs differ by.
  NVAE_SOAK_CONTROL=0 skips the control pair."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import configs
from nvae_tf_amd.datasets import synthetic_mnist

name = sys.argv[1] if len(sys.argv) > 1 else "mnist_c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
c = configs.CONFIGS[name]
B = int(sys.argv[3]) if len(sys.argv) > 3 else c["batch"]
out_json = sys.argv[4] if len(sys.argv) > 4 else None
dev = torch.device("cuda:0")
H, W, C = c["input_hwc"]
assert C == 1
nb = 32
raw = torch.from_numpy(synthetic_mnist(B * (nb + 1), 1)[0]).float()
data = torch.zeros(B * (nb + 1), 32, 32, 1); data[:, 2:30, 2:30, 0] = (raw > 0).float()
data = data.to(dev)
held_out = data[nb * B:(nb + 1) * B]
models = {}
runs = [("bf16", torch.bfloat16, 1), ("f32", torch.float32, 1)]
if os.environ.get("NVAE_SOAK_CONTROL", "1") != "0":
    runs += [("bf16/B", torch.bfloat16, 1001), ("f32/B", torch.float32, 1001)]
for tag, dt, noise_key in runs:
    m = configs.build(name, batch=B, device=dev, dtype=dt, total_epochs=1, n_total_iterations=steps, seed=1)
    m.seed = noise_key                      # after construction: the weights are those of seed 1 in every run
    m.capture_train_step((B, H, W, C))
    models[tag] = m
assert torch.equal(models["bf16"].ps.params, models["f32"].ps.params)
win = max(steps // 10, 10)
curves = {t: [] for t in models}
every = max(steps // 30, 1)
for i in range(steps):
    x = data[(i % nb) * B:(i % nb + 1) * B]
    for tag, m in models.items():
        out = m.train_step_graphed(x)
        if i % every == 0 or i >= steps - win:
            # unscaled ELBO terms: reconstruction + sum of the per-group KLs (kl_loss itself is beta-scaled)
            rec = float(out["reconstruction_loss"].mean()); kl = float(out["kl_per_group"].sum(0).mean())
            curves[tag].append((i, float(out["loss"]), rec, kl))
    if i % every == 0:
        a, b = curves["bf16"][-1], curves["f32"][-1]
        print(f"step {i:5d}  bf16 loss {a[1]:9.3f} recon {a[2]:8.3f} kl {a[3]:8.3f} | f32 loss {b[1]:9.3f} recon {b[2]:8.3f} kl {b[3]:8.3f} | "
              f"neg-ELBO gap {a[2] + a[3] - b[2] - b[3]:+8.3f}", flush=True)
        assert math.isfinite(a[1]) and math.isfinite(b[1])
torch.cuda.synchronize()
res = {"workload": name, "steps": steps, "batch": B, "window": win}
for tag in models:
    tail = [r for r in curves[tag] if r[0] >= steps - win]
    res[tag] = {"neg_elbo_trailing_mean": sum(r[2] + r[3] for r in tail) / len(tail),
                "recon_trailing_mean": sum(r[2] for r in tail) / len(tail), "kl_trailing_mean": sum(r[3] for r in tail) / len(tail)}
# held-out batch, fixed noise, eval mode (moving statistics): single-sample negative ELBO of both trained models
g = torch.Generator().manual_seed(77)
eps = [torch.randn(s, generator=g) for s in models["f32"].eps_shapes(B)]
for tag, m in models.items():
    logits, zp, lp, lq = m(held_out, nll=True, eps_list=eps)
    rec = m.calculate_recon_loss(held_out, logits)
    res[tag]["held_out_neg_elbo"] = float((rec + lq - lp).mean())
# the same TRAINED weights evaluated in the other precision: isolates what bf16 arithmetic does to the ELBO of a given
# model from where chaotic training happened to take each run
def held_out_with(weights_of, run_in):
    src, dst = models[weights_of], models[run_in]
    keep = (dst.ps.params.clone(), dst.ps.state.clone())
    dst.ps.params.copy_(src.ps.params); dst.ps.state.copy_(src.ps.state)
    logits, zp, lp, lq = dst(held_out, nll=True, eps_list=eps)
    v = float((dst.calculate_recon_loss(held_out, logits) + lq - lp).mean())
    dst.ps.params.copy_(keep[0]); dst.ps.state.copy_(keep[1])
    return v
res["cross_precision_held_out"] = {
    "f32_weights_in_f32": res["f32"]["held_out_neg_elbo"], "f32_weights_in_bf16": held_out_with("f32", "bf16"),
    "bf16_weights_in_bf16": res["bf16"]["held_out_neg_elbo"], "bf16_weights_in_f32": held_out_with("bf16", "f32")}
c = res["cross_precision_held_out"]
res["bf16_inference_minus_f32_inference_nats"] = [c["f32_weights_in_bf16"] - c["f32_weights_in_f32"],
                                                  c["bf16_weights_in_bf16"] - c["bf16_weights_in_f32"]]
res["gap_trailing_nats"] = res["bf16"]["neg_elbo_trailing_mean"] - res["f32"]["neg_elbo_trailing_mean"]
res["gap_held_out_nats"] = res["bf16"]["held_out_neg_elbo"] - res["f32"]["held_out_neg_elbo"]
if "f32/B" in res:
    res["control_f32_vs_f32_trailing_nats"] = res["f32/B"]["neg_elbo_trailing_mean"] - res["f32"]["neg_elbo_trailing_mean"]
    res["control_f32_vs_f32_held_out_nats"] = res["f32/B"]["held_out_neg_elbo"] - res["f32"]["held_out_neg_elbo"]
    res["gap_B_trailing_nats"] = res["bf16/B"]["neg_elbo_trailing_mean"] - res["f32/B"]["neg_elbo_trailing_mean"]
    res["gap_B_held_out_nats"] = res["bf16/B"]["held_out_neg_elbo"] - res["f32/B"]["held_out_neg_elbo"]
    res["bf16_mean_minus_f32_mean_trailing_nats"] = 0.5 * (res["gap_trailing_nats"] + res["gap_B_trailing_nats"])
    res["bf16_mean_minus_f32_mean_held_out_nats"] = 0.5 * (res["gap_held_out_nats"] + res["gap_B_held_out_nats"])
print(json.dumps(res, indent=1))
if out_json:
    with open(out_json, "w") as fh:
        json.dump({"summary": res, "curves": curves}, fh)
