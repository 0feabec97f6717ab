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


def test_fullsize_train_steps(lib, dev):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
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
