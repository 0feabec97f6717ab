"""bf16 training against f32 training of the benchmark model (BASELINE.json configs[1] architecture), same seed, same
data, same in-graph noise stream.

What can be asserted, and what cannot (tools/soak_pair.py, profiles/r02_soak_c2.json): two f32 runs that differ only in
the Philox key of the latent noise end 3 000 steps 4-6 nats apart in negative ELBO (training is chaotic and the
atomics-ordered reductions make even two runs of the SAME configuration diverge), so "bf16 ends within 0.5 nats of
f32" is not a resolvable statement about a pair of training runs.  What bf16 arithmetic does to the ELBO of a GIVEN
model is: the same trained weights evaluated in both precisions on a held-out batch with fixed noise.  That is the
0.5-nat bound (north_star: "NLL within +-0.5 nats") asserted here; the training curves are asserted to stay together at
the scale of their own run-to-run spread.

The evaluation here uses BATCH statistics: this early in training (500 steps, KL still ~100 nats) the moving
BatchNorm statistics (Keras momentum 0.05: essentially the last training batch's) make the eval-mode forward of a
held-out batch wildly off in ANY precision (tools/diag_precision.py: -ELBO 9e4 at step 240, 600 at step 600 against
500 with batch statistics), so its bf16-f32 difference says nothing about arithmetic.  Measured there (batch 64):
batch-statistics difference -0.17 / +0.29 nats at step 240, -0.05 / -0.02 at 600, -0.02 / +0.002 at 1 200; eval-mode
difference -0.007 / -0.001 at 1 200 and -0.007 / -0.067 after the 3 000-step soak."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

STEPS, BATCH, WINDOW = 500, 64, 50


def test_bf16_training_tracks_f32(lib, dev):
    from nvae_tf_amd import configs
    from nvae_tf_amd.datasets import synthetic_mnist
    nb = 32
    raw = torch.from_numpy(synthetic_mnist(BATCH * (nb + 1), 1)[0]).float()
    data = torch.zeros(BATCH * (nb + 1), 32, 32, 1)
    data[:, 2:30, 2:30, 0] = (raw > 0).float()
    data = data.to(dev)
    held_out = data[nb * BATCH:]
    models = {}
    for tag, dt in (("bf16", torch.bfloat16), ("f32", torch.float32)):
        m = configs.build("mnist_c2", batch=BATCH, device=dev, dtype=dt, total_epochs=1, n_total_iterations=3000, seed=1)
        m.capture_train_step((BATCH, 32, 32, 1))
        models[tag] = m
    assert torch.equal(models["bf16"].ps.params, models["f32"].ps.params)
    tail = {t: [] for t in models}
    for i in range(STEPS):
        x = data[(i % nb) * BATCH:(i % nb + 1) * BATCH]
        for tag, m in models.items():
            out = m.train_step_graphed(x)
            if i >= STEPS - WINDOW:
                rec = float(out["reconstruction_loss"].mean())
                kl = float(out["kl_per_group"].sum(0).mean())
                assert math.isfinite(rec) and math.isfinite(kl)
                tail[tag].append((rec, kl))
    torch.cuda.synchronize()
    mean = {t: (sum(r for r, _ in v) / len(v), sum(k for _, k in v) / len(v)) for t, v in tail.items()}
    print("trailing means (recon, KL):", mean)
    # the curves stay together: reconstruction within 1.5 %, negative ELBO within 8 % (the KL term is still falling by
    # ~0.3 nats per step here, so the two runs are some tens of steps apart, not at different levels)
    assert abs(mean["bf16"][0] - mean["f32"][0]) / mean["f32"][0] < 0.015
    nelbo = {t: mean[t][0] + mean[t][1] for t in mean}
    assert abs(nelbo["bf16"] - nelbo["f32"]) / nelbo["f32"] < 0.08

    # the same weights in both precisions: held-out single-sample negative ELBO under fixed noise, batch statistics
    g = torch.Generator().manual_seed(77)
    eps = [torch.randn(s, generator=g) for s in models["f32"].eps_shapes(BATCH)]

    def neg_elbo(m):
        logits, _, lp, lq = m(held_out, nll=True, eps_list=eps, training=True)
        return float((m.calculate_recon_loss(held_out, logits) + lq - lp).mean())

    for weights_of in ("f32", "bf16"):
        src = models[weights_of]
        vals = {}
        for run_in, m in models.items():
            keep = (m.ps.params.clone(), m.ps.state.clone())
            m.ps.params.copy_(src.ps.params); m.ps.state.copy_(src.ps.state)
            vals[run_in] = neg_elbo(m)
            m.ps.params.copy_(keep[0]); m.ps.state.copy_(keep[1])
        print(f"weights trained in {weights_of}: held-out -ELBO in f32 {vals['f32']:.3f}, in bf16 {vals['bf16']:.3f}")
        assert abs(vals["bf16"] - vals["f32"]) < 0.5 and abs(vals["bf16"] - vals["f32"]) < 1e-3 * vals["f32"], (weights_of, vals)
