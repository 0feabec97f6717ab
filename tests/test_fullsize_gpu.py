"""GPU tests at BASELINE.json's full sizes (batch 128, paper-default channel counts) through
size-independent properties, since the CPU oracle needs minutes at these sizes:
  * batch invariance: a conv over 128 images == the same conv over its two 64-image halves (the
    halves run a different tile configuration of the kernel, so this cross-checks 256x192 vs 128x192);
  * linearity of the weight gradient in the batch: wgrad(full) == wgrad(half0) + wgrad(half1);
  * one full-size bf16 training step of the benchmark model: finite losses, KL >= 0 per group, BN
    output statistics, loss decreases over a few steps, hipGraph replay keeps training."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _conv_pair(dev, dtype, B, H, cin, cout, k):
    from nvae_tf_amd.params import ParamStore
    ps = ParamStore(seed=2)
    conv = ps.conv("c", k, cin, cout)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    ps.begin_step(); ps.prepare_weights(False)
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, H, H, cin, generator=g).to(dev, dtype)
    dy = torch.randn(B, H, H, cout, generator=g).to(dev, dtype)
    return ps, conv, x, dy


def _run(ps, conv, x, dy, dtype):
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Ctx, Var
    ps.grads.zero_()
    ctx = Ctx(ps, dtype, True, True)
    xv = Var(x)
    y = ops.conv2d(ctx, xv, conv)
    y.g = dy
    ctx.backward()
    return y.t, xv.g, ps.get_grad("c.w").clone(), ps.get_grad("c.b").clone()


@pytest.mark.parametrize("shape", [(128, 16, 384, 384, 5), (128, 32, 192, 192, 5), (128, 4, 256, 256, 3)],
                         ids=["16x16_384", "32x32_192", "4x4_256"])
def test_conv_batch_invariance_and_wgrad_linearity(lib, dev, shape):
    B, H, cin, cout, k = shape
    dtype = torch.bfloat16
    ps, conv, x, dy = _conv_pair(dev, dtype, B, H, cin, cout, k)
    y, dx, dw, db = _run(ps, conv, x, dy, dtype)
    h = B // 2
    y0, dx0, dw0, db0 = _run(ps, conv, x[:h].contiguous(), dy[:h].contiguous(), dtype)
    y1, dx1, dw1, db1 = _run(ps, conv, x[h:].contiguous(), dy[h:].contiguous(), dtype)
    # forward / data gradient: per-output accumulation order over K is tile-independent -> bit-exact
    assert torch.equal(y[:h], y0) and torch.equal(y[h:], y1)
    assert torch.equal(dx[:h], dx0) and torch.equal(dx[h:], dx1)
    # weight gradient: f32 partial sums in a different order -> equal to f32 rounding of the sum
    scale = float(dw.abs().max())
    assert float((dw - (dw0 + dw1)).abs().max()) / scale < 2e-5
    assert float((db - (db0 + db1)).abs().max()) / float(db.abs().max()) < 2e-5
    assert math.isfinite(scale) and scale > 0


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_fullsize_train_steps(lib, dev, dtype):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    if dtype == torch.float16:
        return _fullsize_f16(bench, dev)
    model = bench.make_model(dev, torch.bfloat16, 128)
    assert model.n_trainable() == 62225021
    x = bench.synthetic_batch(128, 1, dev)
    model.steps = 10 ** 9            # beta = 1
    out = model.train_step(x)
    torch.cuda.synchronize()
    kl = out["kl_per_group"]
    assert kl.shape == (15, 128) and bool(torch.isfinite(kl).all()) and float(kl.min()) > -1e-3
    assert bool(torch.isfinite(out["reconstruction_loss"]).all())
    first = float(out["loss"])
    assert math.isfinite(first) and abs(float(out["bn_loss"]) - 0.01 * 174) < 1e-3     # gamma = 1 at init
    # gradient buffer is finite and non-trivial
    g = model.ps.grads
    assert bool(torch.isfinite(g).all()) and float(g.abs().max()) > 0
    model.capture_train_step(x.shape, warmup=1)
    model._static_x.copy_(x.to(torch.bfloat16))
    for _ in range(8):
        out = model.train_step_graphed(None)
    torch.cuda.synchronize()
    last = float(out["loss"])
    assert math.isfinite(last) and last < first, (first, last)


def _fullsize_f16(bench, dev):
    """The benchmark model (C2, batch 128) with float16 activations and the device-side dynamic loss scale, from step 0
    of the KL warm-up like a real run: every step is taken (no overflow from the initial scale 2^-8), the loss falls
    like the bf16 run's (tools/diag_f16_train.py: 720 -> 447 against 720 -> 427 nats after 384 steps)."""
    from nvae_tf_amd import _lib as L
    model = bench.make_model(dev, torch.float16, 128)
    assert model.dynamic_loss_scale and float(model.hyper[L.HY_LSCALE]) == 2.0 ** -8
    x = bench.synthetic_batch(128, 1, dev)
    model.capture_train_step(x.shape, warmup=1)
    model._static_x.copy_(x.to(torch.float16))
    losses = []
    for _ in range(60):
        losses.append(model.train_step_graphed(None)["loss"].clone())
    torch.cuda.synchronize()
    losses = [float(v) for v in losses]
    assert all(math.isfinite(v) for v in losses) and bool(torch.isfinite(model.ps.params).all())
    assert min(losses[-10:]) < 0.75 * losses[0], losses[::6]
    # (measured: no step skipped in 400; allow for one overflow, which halves the scale and restarts the count)
    assert float(model.hyper[L.HY_GOOD]) >= 30.0 and float(model.hyper[L.HY_LSCALE]) >= 2.0 ** -9


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json configs[3] (CIFAR-10, 30 groups, DMoL head, batch 64) and configs[4] (CelebA-64, 40 groups,
# batch 256 global = 32 per GPU) at their FULL per-GPU batch.  The CPU oracle needs minutes per step at these
# sizes (its parity runs at reduced batch live in test_model_gpu.py / test_golden_gpu.py), so the full batch is
# checked through properties that do not depend on size:
#   * inference is per-image: with moving BN statistics and fixed noise, a batch of B images gives the same
#     logits / log p / log q / reconstruction NLL as its two halves run on their own (the halves pick other
#     tile families and other split counts of every kernel);
#   * a training step yields KL >= 0 in every group, finite per-image losses, a finite non-trivial gradient;
#   * replaying the captured step keeps every loss / parameter finite and really steps the optimizer;
#   * ancestral samples are finite images in [0, 1] and sample_with_z(z, s) reproduces the decoder's last stage.
# eager vs graphed steps from one state: identical launches and noise, f32 atomics in a different order.
# Measured on C4 over four steps: reconstruction term 2e-4 .. 9.8e-3 (growing step over step); the total loss
# 7e-2 .. 6.4e-1 - at a random initialisation the 30 / 40-group KL inside it (1e6-1e7 nats at beta = 0.04, prior
# sigmas near their 0.01 floor) turns a one-ulp difference of a bf16 activation into percents, in the FIRST forward pass
# already, so only the reconstruction term carries a tight bound.
# (first step, any of the four steps) of the reconstruction term; the loss is printed, not bounded: measured 0.13-0.64 apart
EAGER_VS_GRAPH_TOL = (2e-3, 3e-2)


def _rgb_batch(B, hw, dev, seed=3):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randint(0, 256, (B, hw, hw, 3), generator=g).float() / 255.0).to(dev)


@pytest.mark.parametrize("name,n_groups,dtype", [("cifar10", 30, torch.bfloat16), ("celeba64", 40, torch.bfloat16),
                                                 ("celeba64", 40, torch.float16)],
                         ids=["C4_batch64", "C5_batch32", "C5_batch32_f16"])
def test_rgb_configs_full_batch_properties(lib, dev, name, n_groups, dtype):
    from nvae_tf_amd import configs
    B = configs.CONFIGS[name]["batch"]
    hw = configs.CONFIGS[name]["input_hwc"][0]
    # (float16 is the activation type BASELINE.json configs[4] names: "fp16 with fp32 KL accumulate")
    model = configs.build(name, device=dev, dtype=dtype)          # float16: dynamic loss scaling (the default)
    assert model.n_groups == n_groups
    assert model.n_trainable() == {"cifar10": 174561204, "celeba64": 394575888}[name]     # = the oracle's constructor
    x = _rgb_batch(B, hw, dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    eps = [torch.randn(s, generator=g) for s in model.eps_shapes(B)]

    # --- inference is per-image
    full = model(x, nll=True, eps_list=eps)
    rec_full = model.calculate_recon_loss(x, full[0])
    h = B // 2
    for lo, hi in ((0, h), (h, B)):
        part = model(x[lo:hi], nll=True, eps_list=[e[lo:hi] for e in eps])
        rec = model.calculate_recon_loss(x[lo:hi], part[0])
        scale = float(full[0].float().abs().max())
        assert float((part[0].float() - full[0][lo:hi].float()).abs().max()) <= 2e-2 * scale
        for a, b in ((part[2], full[2][lo:hi]), (part[3], full[3][lo:hi]), (rec, rec_full[lo:hi])):
            assert bool(torch.isfinite(a).all())
            assert float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) < 2e-3

    # --- one training step early in the KL warm-up (beta = 0.04, KL balancing on), as training really begins: at
    # beta = 1 the 30 / 40-group KL of a random initialisation is ~1e7 nats and its gradient ~1e23, a regime no run
    # ever visits and in which f32 itself overflows now and then
    model.steps = 2000
    assert 0.03 < model.beta() < 0.05
    out = model.train_step(x, eps_list=eps)
    torch.cuda.synchronize()
    kl = out["kl_per_group"]
    assert kl.shape == (n_groups, B) and bool(torch.isfinite(kl).all()) and float(kl.min()) > -1e-3
    assert bool(torch.isfinite(out["reconstruction_loss"]).all()) and float(out["reconstruction_loss"].min()) > 0
    first = float(out["loss"])
    assert math.isfinite(first)
    gr = model.ps.grads
    f16 = dtype == torch.float16
    if not f16:     # (float16: the first steps may overflow; the dynamic loss scale skips them and halves itself)
        assert bool(torch.isfinite(gr).all()) and float(gr.abs().max()) > 0

    # --- graph replay keeps training.  beta is held FIXED for this part (epoch-based warm-up with a constant epoch:
    # models.py:121-122 of the reference, Q10), so that "the model learns" is a statement about the reconstruction term
    # at a constant objective and not about the warm-up schedule.
    model.step_based_warmup, model.epoch = False, 2000
    assert 0.03 < model.beta() < 0.05
    model.capture_train_step(x.shape, warmup=1)
    model._static_x.copy_(x)
    kl_first = float(kl.sum(0).mean())
    p_before = model.ps.params.clone()
    ps = model.ps
    state0 = [(t, t.clone()) for t in (ps.params, ps.state, ps.adam_m, ps.adam_u, model.rng_counter, model.hyper,
                                       model.coeff, model.am, model.results)]
    it0 = (model.steps, model.opt_iterations)
    losses, kls, recs = [], [], []
    for _ in range(24):
        o = model.train_step_graphed(None)
        losses.append(o["loss"].clone()); kls.append(o["kl_per_group"].sum(0).mean())
        recs.append(o["reconstruction_loss"].mean())
    torch.cuda.synchronize()
    losses, kls, recs = [float(v) for v in losses], [float(v) for v in kls], [float(v) for v in recs]
    if f16:
        # float16 is the dtype BASELINE.json configs[4] names.  At a RANDOM INITIALISATION of this 40-group network the
        # parameter gradients grow by ~1.5-2x per group on the way back - 19 decades from the last decoder group to the stem
        # (profiles/r02_f16_gradient_range_c5.txt), float16 holds 12 - so no single loss scale fits (round 2: every step
        # skipped).  Round 3: the backward pass renormalises the activation gradient at every group / cell boundary on the
        # device (ops.GradScale); with it the model TRAINS in float16: steps are taken, the optimizer state is finite and
        # non-zero, and the reconstruction term falls as it does in bf16.
        from nvae_tf_amd import _lib as L
        assert model.grad_rescale
        scale = float(model.hyper[L.HY_LSCALE])
        print(f"float16 {name}: loss scale after {len(losses)} steps 2^{math.log2(scale):.0f}, Adamax slots max "
              f"{float(model.ps.adam_u.max()):.3e}; reconstruction term " + " ".join(f"{v:.0f}" for v in recs))
        assert bool(torch.isfinite(model.ps.params).all()) and bool(torch.isfinite(model.ps.adam_m).all())
        assert bool(torch.isfinite(model.ps.adam_u).all())
        assert float(model.ps.adam_u.max()) > 0 and float(model.ps.adam_m.abs().max()) > 0        # steps were TAKEN
        assert all(math.isfinite(v) for v in losses)
        assert sum(recs[-6:]) / 6 < recs[0], recs
        return
    # At a random initialisation the 30 / 40-group KL is 1e6-1e7 nats and neither the weighted loss nor the KL is
    # monotone over the first tens of steps (measured: C4's KL goes 2.0e6 -> 5.3e6 -> 2.3e6 within 24 steps, C5's loss
    # rises while beta climbs; the 60-step C4 soak of profiles/r01_soak_cifar10_60steps.txt shows the fall that
    # follows), so at this length the properties are: every step's loss and KL finite, every parameter finite, and the
    # optimizer really stepping (replay == eager and loss descent are asserted on C2, where 8 steps suffice).
    assert all(math.isfinite(v) for v in losses + kls) and kl_first > 0
    assert bool(torch.isfinite(model.ps.params).all()) and float(model.ps.adam_u.max()) > 0
    assert float((model.ps.params - p_before).abs().max()) > 1e-4
    # the model LEARNS at this batch: on the fixed batch, at fixed beta, the reconstruction term of the last six of the
    # 24 steps lies below the first step's (noise is redrawn every step, hence a window and not a single step)
    print(f"{name}: reconstruction term over 24 graphed steps at beta {model.beta():.3f}: "
          + " ".join(f"{v:.0f}" for v in recs))
    assert sum(recs[-6:]) / 6 < recs[0], recs
    # and the graphed steps ARE the eager steps: back to the state before the replays, four eager steps (same Philox
    # counter, so the same noise) reproduce the first four graphed losses up to the run-to-run noise of the f32 atomics
    for t, saved in state0:
        t.copy_(saved)
    model.steps, model.opt_iterations = it0
    dev_l, dev_r = [], []
    for i in range(4):
        o = model.train_step(x)
        dev_l.append(abs(float(o["loss"]) - losses[i]) / abs(losses[i]))
        dev_r.append(abs(float(o["reconstruction_loss"].mean()) - recs[i]) / abs(recs[i]))
    print(f"{name}: eager vs graphed over four steps from one state: loss {dev_l}, reconstruction term {dev_r}")
    assert dev_r[0] < EAGER_VS_GRAPH_TOL[0] and max(dev_r) < EAGER_VS_GRAPH_TOL[1], (dev_l, dev_r)

    # --- sampling
    images, last_s, z1, z2 = model.sample(n_samples=8, temperature=0.8)
    assert images.shape == (8, hw, hw, 3) and bool(torch.isfinite(images).all())
    assert float(images.min()) >= 0.0 and float(images.max()) <= 1.0
    again = model.sample_with_z(z1, last_s)
    assert again.shape == images.shape and bool(torch.isfinite(again).all())
