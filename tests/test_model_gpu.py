"""End-to-end parity of the HIP path against the CPU oracle on a shrunken NVAE: identical weights
(copied by name), identical inputs and noise.  Covers train_step (losses, KL per group, every
parameter gradient, the Adamax update, BN moving statistics, spectral-norm state), the inference
forward with IWAE terms, and ancestral sampling.

Tolerances (stated per north_star): f32 path vs the fp64 oracle 1e-3 relative on losses and 5e-3 of
the gradient scale per tensor; bf16 path 3e-2 on losses, 0.2 of the gradient scale per tensor (median below 5e-2, at most 2 % of the tensors
exempt; bf16 keeps 8 significant bits through ~60 layers), gradient cosine > 0.9993; f16 path 5e-3 on losses, 0.1 per
tensor (measured: median 3.5e-3, worst 5.6e-2), cosine > 0.9999."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(n_encoder_channels=16, n_decoder_channels=16, res_cells_per_group=1, n_preprocess_blocks=2,
           n_preprocess_cells=2, n_latent_per_group=20, n_groups_per_scale=[2, 2], n_postprocess_blocks=2,
           n_post_process_cells=2, sr_lambda=0.01, scale_factor=2, total_epochs=10, n_total_iterations=1000,
           step_based_warmup=True)
B = 4


def build_pair(dev, dtype, cfg=CFG):
    from oracle.nvae_oracle import OracleConfig, OracleNVAE, synthetic_batch
    from nvae_tf_amd.models import NVAE
    ocfg = OracleConfig(**cfg)
    orc = OracleNVAE(ocfg, dtype=torch.float64, seed=5)
    model = NVAE(cfg["n_encoder_channels"], cfg["n_decoder_channels"], cfg["res_cells_per_group"],
                 cfg["n_preprocess_blocks"], cfg["n_preprocess_cells"], cfg["n_latent_per_group"],
                 len(cfg["n_groups_per_scale"]), cfg["n_groups_per_scale"], cfg["n_postprocess_blocks"],
                 cfg["n_post_process_cells"], cfg["sr_lambda"], cfg["scale_factor"], cfg["total_epochs"],
                 cfg["n_total_iterations"], cfg["step_based_warmup"], [B, 32, 32, 1], device=dev, dtype=dtype)
    assert model.n_trainable() == orc.n_trainable()
    # perturb BN affine parameters / biases so that they matter, then copy oracle -> product by name
    g = torch.Generator().manual_seed(99)
    with torch.no_grad():
        for k, v in orc.s.params.items():
            if k.endswith(".gamma"):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.1)
            elif k.endswith((".beta", ".b", ".b1", ".b2")):
                v.add_(torch.randn(v.shape, generator=g, dtype=torch.float64) * 0.05)
    model.ps.load_named(orc.s.params, orc.s.state)
    x = synthetic_batch(B, seed=3)
    eg = torch.Generator().manual_seed(8)
    eps = [torch.randn(s, generator=eg, dtype=torch.float64) for s in orc.eps_shapes(B)]
    assert [tuple(e.shape) for e in eps] == [tuple(s) for s in model.eps_shapes(B)]
    return orc, model, x, eps


def rel(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def oracle_16bit(orc, snap, x, eps, jitter=None, dt=torch.bfloat16):
    """One training-mode forward + backward of the oracle with its stored activations and conv weights rounded to the
    16-bit type `dt` (OracleNVAE.act_round), from the parameter / state snapshot `snap`.  jitter: None = round to nearest;
    a torch.Generator = every value is first nudged by up to half an ulp (which neighbour it lands on becomes
    arbitrary: the noise that f32 atomics / summation orders put under the kernels' rounding).  -> (loss dict, grads)."""
    def rnd(t):
        if jitter is not None:
            t = t * (1.0 + (torch.rand(t.shape, generator=jitter, dtype=t.dtype) - 0.5) * 2.0 ** -8)
        return t.to(dt).to(t.dtype)
    orc.s.params = {k: v.clone().requires_grad_(True) for k, v in snap[0].items()}
    orc.s.state = {k: v.clone() for k, v in snap[1].items()}
    orc.act_round = rnd
    try:
        orc.spectral_norm_step()
        out = orc.loss(x, eps, training=True)
        names = list(orc.s.params)
        gs = torch.autograd.grad(out["loss"], [orc.s.params[k] for k in names], allow_unused=True)
    finally:
        orc.act_round = None
    return out, {k: (g if g is not None else torch.zeros_like(orc.s.params[k])) for k, g in zip(names, gs)}


def bf16_spread(orc, snap, x, eps, exact, runs=6, dt=torch.bfloat16):
    """How far a CORRECT bf16 implementation can sit from the fp64 result, per gradient tensor: the oracle itself with
    16-bit storage (oracle_16bit: round-to-nearest once, then `runs - 1` jittered runs), each compared with the exact
    gradient.  Returns {name: max relative error over the runs}.  Tensors for which this is O(1) are ill-conditioned
    at this input (a ReLU kink of an SE hidden unit, a BatchNorm beta whose gradient nearly cancels): the test widens
    THEIR bound, by name, instead of exempting a blanket share of all tensors."""
    spread = {k: 0.0 for k in exact}
    gen = torch.Generator().manual_seed(1234)
    for r in range(runs):
        _, gs = oracle_16bit(orc, snap, x, eps, gen if r else None, dt)
        for k, g in gs.items():
            spread[k] = max(spread[k], rel(g, exact[k]))
    return spread


@pytest.mark.parametrize("dtype,ltol,gtol", [(torch.float32, 1e-3, 5e-3), (torch.bfloat16, 3e-2, 0.2),
                                             (torch.float16, 5e-3, 0.1)], ids=["f32", "bf16", "f16"])
def test_train_step_parity(lib, dev, dtype, ltol, gtol):
    orc, model, x, eps = build_pair(dev, dtype)
    orc.steps = model.steps = 100          # beta = 1/3 -> KL balancing active
    snap = ({k: v.detach().clone() for k, v in orc.s.params.items()}, {k: v.clone() for k, v in orc.s.state.items()})
    out_o = orc.train_step(x, eps, decay_steps=1000)
    out = model.train_step(x.float(), [e.float() for e in eps])
    torch.cuda.synchronize()
    assert rel(out["reconstruction_loss"], out_o["reconstruction_loss"]) < ltol
    assert rel(out["kl_per_group"], out_o["kl_per_group"]) < ltol * 3
    assert rel(out["kl_loss"], out_o["kl_loss"]) < ltol * 3
    assert abs(float(out["bn_loss"]) - float(out_o["bn_loss"])) < 1e-5
    assert abs(float(out["loss"]) - float(out_o["loss"])) / abs(float(out_o["loss"])) < ltol
    assert rel(model.coeff, out_o["kl_coeff"]) < ltol * 3
    # every parameter gradient (the f16 path holds loss_scale * gradient until Adamax divides the scale out)
    from nvae_tf_amd import _lib as L
    ls = float(model.hyper[L.HY_LSCALE])
    assert ls == (2.0 ** -8 if dtype == torch.float16 else 1.0)
    worst = []
    for k, g_o in out_o["grads"].items():
        worst.append((rel(model.ps.get_grad(k) / ls, g_o), k))
    worst.sort(reverse=True)
    print("worst gradient errors:", worst[:8])
    valid = sorted(e for e, k in worst if float(out_o["grads"][k].abs().max()) > 1e-6)
    print(f"per-tensor gradient error over {len(valid)} tensors with a real gradient: median {valid[len(valid) // 2]:.2e} "
          f"p90 {valid[len(valid) * 9 // 10]:.2e} p98 {valid[len(valid) * 98 // 100]:.2e} max {valid[-1]:.2e}")
    bad = [(e, k) for e, k in worst if e > gtol and float(out_o["grads"][k].abs().max()) > 1e-6]
    if dtype != torch.bfloat16:
        assert not bad, bad[:10]
    else:
        # bf16.  Round 2 exempted a blanket 2 % of the tensors after one run in four showed post.cell0.se.b1 / se.w1 /
        # bn3.beta at 0.4-1.0 of their scale.  Now every tensor has its own bound: max(gtol, 2 x the spread a correct
        # bf16 implementation has at THIS input, measured on the oracle itself: bf16_spread above).  The tensors that get
        # a widened bound are listed; they are the ones the emulation itself moves by more than gtol / 2 (profiles/
        # r03_bf16_spread.txt: SE parameters of cells where a hidden unit sits at its ReLU kink, and the BatchNorm in
        # front of that SE).  Everything else - and the distribution as a whole - keeps the tight bound.
        orc.steps = 100
        spread = bf16_spread(orc, snap, x, eps, out_o["grads"], runs=8)
        widened = sorted(((v, k) for k, v in spread.items() if 2 * v > gtol), reverse=True)
        print("tensors whose own bf16 spread exceeds gtol / 2 (bound widened to 2 x spread):", widened[:12])
        assert len(widened) <= 8, widened        # measured: 5 (post.cell0.se.b1 / se.w1 / bn3.beta, pre.cell0.se.w1 / se.b1)
        over = [(e, k, spread[k]) for e, k in bad if e > max(gtol, 2 * spread[k])]
        assert not over, over[:10]
        assert valid[len(valid) // 2] < 5e-2 and valid[len(valid) * 9 // 10] < 9e-2
    # direction of the whole gradient
    go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
    gp = torch.cat([model.ps.get_grad(k).double().cpu().reshape(-1) for k in out_o["grads"]]) / ls
    cos = float((go * gp).sum() / (go.norm() * gp.norm()))
    print("gradient cosine", cos)
    assert cos > {torch.float32: 0.99999, torch.bfloat16: 0.9993, torch.float16: 0.9999}[dtype]   # bf16 measured: 0.99970-0.99974
    # Adamax update, BN moving statistics, spectral-norm state.  Adamax divides by max|g|, so an
    # element whose true gradient is 0 (e.g. a conv bias feeding a BatchNorm) moves by +-lr on
    # rounding noise alone in ANY f32 implementation: compare only elements with a real gradient.
    if dtype == torch.float32:
        for k, p_o in orc.s.params.items():
            mask = out_o["grads"][k].abs() > 1e-4        # f32 noise ~4e-7 on g -> < 0.4 % of lr on the update
            d = (model.ps.get(k).double().cpu() - p_o.detach()).abs()
            assert float((d * mask).max()) < 2e-5, k
        for k, s_o in orc.s.state.items():
            assert rel(model.ps.get_state(k), s_o) < 5e-3, k


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2), (torch.float16, 6e-3)],
                         ids=["f32", "bf16", "f16"])
def test_inference_and_sampling_parity(lib, dev, dtype, tol):
    orc, model, x, eps = build_pair(dev, dtype)
    # make the moving statistics non-trivial on both sides
    g = torch.Generator().manual_seed(5)
    for k in orc.s.state:
        if k.endswith(".rm"):
            orc.s.state[k] = torch.randn(orc.s.state[k].shape, generator=g, dtype=torch.float64) * 0.1
        elif k.endswith(".rv"):
            orc.s.state[k] = torch.rand(orc.s.state[k].shape, generator=g, dtype=torch.float64) + 0.5
    model.ps.load_named(orc.s.params, orc.s.state)
    logits_o, zp_o, lp_o, lq_o, _ = orc.call(x, eps, training=False, nll=True)
    logits, zp, lp, lq = model(x.float(), nll=True, eps_list=[e.float() for e in eps])
    torch.cuda.synchronize()
    assert rel(logits, logits_o) < tol
    assert rel(lp, lp_o) < tol and rel(lq, lq_o) < tol
    for a, b in zip(zp, zp_o):
        assert rel(a.enc_mu, b.enc_mu) < tol and rel(a.enc_sigma, b.enc_sigma) < tol
        assert rel(a.dec_mu, b.dec_mu) < tol and rel(a.dec_sigma, b.dec_sigma) < tol
    rec = model.calculate_recon_loss(x.float(), logits, crop_output=True)
    assert rel(rec, orc.calculate_recon_loss(x, logits_o, crop_output=True)) < tol
    # ancestral sampling with temperature (models.py:137-178)
    img_o = orc.sample(B, 0.7, eps)
    img, last_s, z1, z2 = model.sample(B, 0.7, eps_list=[e.float() for e in eps])
    assert img.shape == img_o.shape and rel(img, img_o) < tol
    assert z1.shape == eps[-1].shape and float((z1.float() - z2.float()).abs().max()) > 0


def _params_close(a, b, q95=1e-4):
    # f32 atomics make zero-gradient elements take +-lr noise steps under Adamax (see above), so two runs
    # agree on all but those elements
    d = (a - b).abs()
    assert float(torch.quantile(d[:1 << 20], 0.95)) < q95
    assert float(d.max()) < 2e-2


def _state_close(a, b):
    # A conv bias in front of a BatchNorm has a zero true gradient, so Adamax random-walks it by +-lr per step
    # on rounding noise; the BatchNorm removes it from the activations but its MOVING MEAN follows it.  The
    # moving statistics of two runs therefore agree only to a few lr.
    _params_close(a, b, q95=1e-2)


def test_graph_replay_matches_eager(lib, dev):
    """The hipGraph-captured step must produce the same numbers as the eager launch sequence, from step 0,
    with NO re-synchronisation after capture: capture_train_step leaves parameters, Adamax slots, BN / SN
    state and the noise counter exactly as it found them."""
    _, m_eager, x, eps = build_pair(dev, torch.float32)
    _, m_graph, _, _ = build_pair(dev, torch.float32)
    xs = x.float()
    m_graph.steps = m_eager.steps = 50
    before = [t.clone() for t in (m_graph.ps.params, m_graph.ps.state, m_graph.ps.adam_m, m_graph.ps.adam_u,
                                  m_graph.rng_counter)]
    m_graph.capture_train_step(xs.shape)
    torch.cuda.synchronize()
    after = (m_graph.ps.params, m_graph.ps.state, m_graph.ps.adam_m, m_graph.ps.adam_u, m_graph.rng_counter)
    for a, b in zip(before, after):
        assert torch.equal(a, b)                   # bit-exact: capture is side-effect free
    assert m_graph.steps == 50 and m_graph.opt_iterations == 0
    for _ in range(3):
        o1 = m_eager.train_step(xs)
        o2 = m_graph.train_step_graphed(xs)
    torch.cuda.synchronize()
    assert abs(float(o1["loss"]) - float(o2["loss"])) / abs(float(o1["loss"])) < 1e-4
    assert torch.equal(m_graph.rng_counter, m_eager.rng_counter)
    _params_close(m_graph.ps.params, m_eager.ps.params)
    _state_close(m_graph.ps.state, m_eager.ps.state)


def test_checkpoint_resume_continues_exactly(lib, dev, tmp_path):
    """train -> checkpoint -> (new process state) load -> capture -> train equals the uninterrupted run
    (reference: load_weights + initial_epoch, train.py:46-55,133-135): parameters, Adamax slots, BN / SN
    state, step counters (cosine LR, beta) and the in-graph noise stream all continue."""
    from nvae_tf_amd.train import load_checkpoint, save_checkpoint
    _, a, x, _ = build_pair(dev, torch.float32)
    _, b, _, _ = build_pair(dev, torch.float32)
    xs = x.float()
    a.steps = b.steps = 40
    a.capture_train_step(xs.shape)
    for _ in range(2):
        a.train_step_graphed(xs)
    path = str(tmp_path / "ck" / "epoch_1.pt")
    save_checkpoint(a, path, epoch=1)
    for _ in range(2):
        out_a = a.train_step_graphed(xs)
    # the resumed run: a fresh model (different state until loaded), checkpoint, THEN capture
    for _ in range(3):
        b.train_step(xs)                          # scramble b's state, counters and RNG
    assert load_checkpoint(b, path) == 1
    assert b.steps == 42 and b.opt_iterations == 2
    b.capture_train_step(xs.shape)
    for _ in range(2):
        out_b = b.train_step_graphed(xs)
    torch.cuda.synchronize()
    assert b.steps == a.steps and b.opt_iterations == a.opt_iterations
    assert torch.equal(a.rng_counter, b.rng_counter)
    assert abs(float(out_a["loss"]) - float(out_b["loss"])) / abs(float(out_a["loss"])) < 1e-4
    _params_close(a.ps.params, b.ps.params)
    _state_close(a.ps.state, b.ps.state)
    _params_close(a.ps.adam_u, b.ps.adam_u)


@pytest.fixture
def deterministic(lib):
    """NVAE_DETERMINISTIC=1 for one test (nvae_set_deterministic: must be on before any model of the test is built)."""
    lib.nvae_set_deterministic(1)
    yield
    lib.nvae_set_deterministic(0)


def _bits_equal(a, b):
    assert torch.equal(a.ps.params, b.ps.params) and torch.equal(a.ps.state, b.ps.state)
    assert torch.equal(a.ps.adam_m, b.ps.adam_m) and torch.equal(a.ps.adam_u, b.ps.adam_u)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_deterministic_mode_is_bit_reproducible(lib, dev, dtype, deterministic, tmp_path):
    """NVAE_DETERMINISTIC: every sum across workgroups has one adder per address or a fixed order (one
    statistics-slab row per producing workgroup, slab-combined or unsplit weight gradients, ordered regulariser sum).
    (a) two models from the same state run the same graphed steps -> every parameter, BatchNorm / spectral-norm state and
    Adamax slot equal BIT FOR BIT, also against eager steps; (b) train -> checkpoint -> load into a scrambled model ->
    capture -> train equals the uninterrupted run with torch.equal (the default mode meets it to the Adamax-noise
    tolerance only: test_checkpoint_resume_continues_exactly); (c) the gradients agree with the default mode's to rounding."""
    from nvae_tf_amd.train import load_checkpoint, save_checkpoint
    assert lib.nvae_get_deterministic() == 1
    _, a, x, _ = build_pair(dev, dtype)
    _, b, _, _ = build_pair(dev, dtype)
    _, c, _, _ = build_pair(dev, dtype)
    xs = x.to(dtype)
    for m in (a, b, c):
        m.steps = 40
    a.capture_train_step(xs.shape)
    b.capture_train_step(xs.shape)
    losses = [[], [], []]
    for _ in range(3):
        losses[0].append(float(a.train_step_graphed(xs)["loss"]))
        losses[1].append(float(b.train_step_graphed(xs)["loss"]))
        losses[2].append(float(c.train_step(xs)["loss"]))            # eager: same launches, other timing
    torch.cuda.synchronize()
    assert losses[0] == losses[1] == losses[2], losses
    _bits_equal(a, b)
    _bits_equal(a, c)
    # (b) exact resume
    path = str(tmp_path / "ck" / "epoch_1.pt")
    save_checkpoint(a, path, epoch=1)
    for _ in range(2):
        out_a = a.train_step_graphed(xs)
    _, d, _, _ = build_pair(dev, dtype)
    for _ in range(2):
        d.train_step(xs)                          # scramble d's state, counters and RNG
    assert load_checkpoint(d, path) == 1
    d.capture_train_step(xs.shape)
    for _ in range(2):
        out_d = d.train_step_graphed(xs)
    torch.cuda.synchronize()
    assert float(out_a["loss"]) == float(out_d["loss"])
    _bits_equal(a, d)
    # (c) against the default mode: same arithmetic, other summation orders
    lib.nvae_set_deterministic(0)
    _, e, _, _ = build_pair(dev, dtype)
    e.steps = 40
    e.train_step(xs)
    lib.nvae_set_deterministic(1)
    _, f, _, _ = build_pair(dev, dtype)
    f.steps = 40
    f.train_step(xs)
    torch.cuda.synchronize()
    ge, gf = e.ps.grads.double(), f.ps.grads.double()
    cos = float((ge * gf).sum() / (ge.norm() * gf.norm()))
    assert cos > (0.999999 if dtype == torch.float32 else 0.999), cos


def test_deterministic_mode_c2_architecture(lib, dev, deterministic):
    """The benchmarked model (C2, batch 128, bf16) in deterministic mode: the halo / whole-image / image-split kernels, the
    batched weight gradients and the side stream are all in play; two runs of three graph-replayed steps are bit-equal."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    x = bench.synthetic_batch(128, 1, dev).to(torch.bfloat16)
    runs = []
    for _ in range(2):
        m = bench.make_model(dev, torch.bfloat16, 128)
        m.capture_train_step(x.shape, warmup=1)
        m._static_x.copy_(x)
        ls = [float(m.train_step_graphed(None)["loss"]) for _ in range(3)]
        torch.cuda.synchronize()
        runs.append((ls, m.ps.params.clone(), m.ps.adam_u.clone(), m.ps.state.clone()))
        del m
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    for i in (1, 2, 3):
        assert torch.equal(runs[0][i], runs[1][i])


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 2e-2), (torch.float16, 3e-3)],
                         ids=["f32", "bf16", "f16"])
def test_iwae_nll_parity(lib, dev, dtype, tol):
    """evaluate.py:111-123: the assembled k = 10 importance-weighted bound (cropped reconstruction term,
    sum over groups of log p(z) - log q(z|x), logsumexp over the k samples) on identical noise vs
    OracleNVAE.neg_log_likelihood.  Tolerance 1e-3 relative (f32), i.e. ~0.5 nats on a ~550-nat bound."""
    from nvae_tf_amd.evaluate import batch_neg_log_likelihood, neg_log_likelihood
    orc, model, x, _ = build_pair(dev, dtype)
    g = torch.Generator().manual_seed(5)
    for k in orc.s.state:
        if k.endswith(".rm"):
            orc.s.state[k] = torch.randn(orc.s.state[k].shape, generator=g, dtype=torch.float64) * 0.1
        elif k.endswith(".rv"):
            orc.s.state[k] = torch.rand(orc.s.state[k].shape, generator=g, dtype=torch.float64) + 0.5
    model.ps.load_named(orc.s.params, orc.s.state)
    eg = torch.Generator().manual_seed(21)
    k = 10 if dtype == torch.float32 else 4         # (the 16-bit runs repeat the check on a shorter list: suite time)
    eps_lists = [[torch.randn(s, generator=eg, dtype=torch.float64) for s in orc.eps_shapes(B)] for _ in range(k)]
    ref = float(orc.neg_log_likelihood(x, eps_lists))
    got = float(batch_neg_log_likelihood(model, x.float(), eps_lists=[[e.float() for e in l] for l in eps_lists]))
    print(f"IWAE NLL (k={k}):", got, "oracle", ref)
    assert abs(got - ref) / abs(ref) < tol
    if dtype == torch.float32:
        assert abs(got - ref) < 0.5                 # north_star: NLL within +-0.5 nats
    # the data-set level wrapper (mean +- std across batches) on two batches
    if dtype == torch.float32:
        m = neg_log_likelihood(model, [(x.float(), None), (x.float(), None)],
                               eps_lists=[[[e.float() for e in l] for l in eps_lists]] * 2)
        assert abs(m.mean - got) < 1e-3 * abs(got) and m.stddev < 1e-3 * abs(got)
    # a k = 1 bound is the negative single-sample ELBO with the cropped reconstruction term
    one = float(batch_neg_log_likelihood(model, x.float(), eps_lists=[[e.float() for e in eps_lists[0]]]))
    assert abs(one - float(orc.neg_log_likelihood(x, eps_lists[:1]))) / abs(one) < tol


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, 4e-2)], ids=["f32", "bf16"])
def test_sample_with_z_parity(lib, dev, dtype, tol):
    """models.py:181-189: decode from a fixed last-group z and its decoder state s."""
    orc, model, x, eps = build_pair(dev, dtype)
    img_o, last_s_o, last_z_o = orc.sample(B, 0.8, eps, return_last=True)
    # self-consistency of the restatement: sample == sample_with_z(last z, last s)
    assert rel(orc.sample_with_z(last_z_o, last_s_o), img_o) < 1e-12
    img, last_s, _, _ = model.sample(B, 0.8, eps_list=[e.float() for e in eps])
    assert rel(last_s, last_s_o) < tol              # the state sample() hands out is the oracle's
    out = model.sample_with_z(last_z_o.float(), last_s_o.float())
    torch.cuda.synchronize()
    assert out.shape == img_o.shape and rel(out, img_o) < tol
    # and from the product's own last_s
    out2 = model.sample_with_z(last_z_o.float(), last_s)
    assert rel(out2, img_o) < tol


def test_training_reduces_loss(lib, dev):
    """A few dozen bf16 steps on one synthetic batch must drive the ELBO down (sanity of the whole
    forward/backward/optimizer loop; no oracle involved)."""
    _, model, x, _ = build_pair(dev, torch.bfloat16)
    model.steps = 400   # beta = 1
    model.base_lr = 2e-3
    first = last = None
    for i in range(40):
        out = model.train_step(x.float())
        v = float(out["loss"])
        assert math.isfinite(v)
        first = v if first is None else first
        last = v
    print("loss", first, "->", last)
    assert last < first - 20


def test_tf_literal_train_step_parity(lib, dev):
    """the reference-literal mode (moving-statistics BN, no spectral norm while training),
    including the backward pass through the frozen BatchNorm layers."""
    orc, model, x, eps = build_pair(dev, torch.float32)
    g = torch.Generator().manual_seed(5)
    for k in orc.s.state:
        if k.endswith(".rm"):
            orc.s.state[k] = torch.randn(orc.s.state[k].shape, generator=g, dtype=torch.float64) * 0.1
        elif k.endswith(".rv"):
            orc.s.state[k] = torch.rand(orc.s.state[k].shape, generator=g, dtype=torch.float64) + 0.5
    model.ps.load_named(orc.s.params, orc.s.state)
    state_before = model.ps.state.clone()
    model.tf_literal = True
    orc.steps = model.steps = 100
    out_o = orc.train_step(x, eps, decay_steps=1000, tf_literal=True)
    out = model.train_step(x.float(), [e.float() for e in eps])
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(out_o["loss"])) / abs(float(out_o["loss"])) < 1e-3
    assert rel(out["kl_per_group"], out_o["kl_per_group"]) < 3e-3
    go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
    gp = torch.cat([model.ps.get_grad(k).double().cpu().reshape(-1) for k in out_o["grads"]])
    assert float((go * gp).sum() / (go.norm() * gp.norm())) > 0.99999
    for k in ("enc.g0.c0.bn1.gamma", "dec.g1.c0.bn3.beta", "post.cell0.bn2.gamma", "enc.g1.c0.conv1.w"):
        assert rel(model.ps.get_grad(k), out_o["grads"][k]) < 5e-3, k
    assert torch.equal(model.ps.state, state_before)      # neither moving statistics nor SN state move


def test_c1_architecture_parity(lib, dev):
    """BASELINE.json configs[0] architecture (32 channels, groups [1,1], 1 cell, 2 x 3 pre/post cells;
    17 243 405 parameters) at batch 4, f32 HIP path vs the fp64 oracle."""
    cfg = dict(CFG, n_encoder_channels=32, n_decoder_channels=32, n_preprocess_cells=3, n_post_process_cells=3,
               n_groups_per_scale=[1, 1])
    orc, model, x, eps = build_pair(dev, torch.float32, cfg)
    assert model.n_trainable() == 17243405
    orc.steps = model.steps = 100
    out_o = orc.train_step(x, eps, decay_steps=1000)
    out = model.train_step(x.float(), [e.float() for e in eps])
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(out_o["loss"])) / abs(float(out_o["loss"])) < 1e-3
    assert rel(out["reconstruction_loss"], out_o["reconstruction_loss"]) < 1e-3
    assert rel(out["kl_per_group"], out_o["kl_per_group"]) < 3e-3
    go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
    gp = torch.cat([model.ps.get_grad(k).double().cpu().reshape(-1) for k in out_o["grads"]])
    assert float((go * gp).sum() / (go.norm() * gp.norm())) > 0.99999


@pytest.mark.parametrize("batch", [1, 3, 37])
def test_odd_batch_sizes_run_and_match_eager_graph(lib, dev, batch):
    """Ragged unit counts everywhere (persistent depthwise rings, strip splits, slab fusions): a bf16
    training step at awkward batch sizes stays finite, and its hipGraph replay reproduces the eager step."""
    from nvae_tf_amd.models import NVAE
    from oracle.nvae_oracle import synthetic_batch
    c = CFG
    def make():
        return NVAE(c["n_encoder_channels"], c["n_decoder_channels"], c["res_cells_per_group"], c["n_preprocess_blocks"],
                    c["n_preprocess_cells"], c["n_latent_per_group"], len(c["n_groups_per_scale"]), c["n_groups_per_scale"],
                    c["n_postprocess_blocks"], c["n_post_process_cells"], c["sr_lambda"], c["scale_factor"], c["total_epochs"],
                    c["n_total_iterations"], c["step_based_warmup"], [batch, 32, 32, 1], device=dev, dtype=torch.bfloat16, seed=4)
    x = synthetic_batch(batch, seed=6).float()
    a, b = make(), make()
    a.steps = b.steps = 100
    out_a = a.train_step(x)
    torch.cuda.synchronize()
    assert math.isfinite(float(out_a["loss"])) and bool(torch.isfinite(a.ps.grads).all())
    assert out_a["kl_per_group"].shape == (a.n_groups, batch)
    b.capture_train_step(x.shape, warmup=1)         # side-effect free: no re-sync needed
    out_b = b.train_step_graphed(x)
    torch.cuda.synchronize()
    assert abs(float(out_a["loss"]) - float(out_b["loss"])) / abs(float(out_a["loss"])) < 2e-3


@pytest.mark.parametrize("batch", [2, 6])
def test_c2_architecture_parity(lib, dev, batch):
    """BASELINE.json configs[1] architecture at full width and depth (groups [5,10], 2 cells per group,
    62 225 021 parameters, 15 latent groups): f32 HIP path vs the fp64 oracle - losses, all 15 per-group KLs,
    balancing coefficients, per-tensor gradients and the direction of the whole 62 M-element gradient.

    Conditioning (measured, tests/diag/diag_c2b.py and the PyTorch-CPU f32 run of the oracle itself): at this random
    initialisation the 330-layer network amplifies f32 rounding by ~1e4, more with the batch size - the gradient norm
    is 6.9e6 at batch 2, 2.2e7 at 4, 1.3e8 at 8.  At batch 8 the f32 PyTorch run of the ORACLE is 4.6e-4 off the
    fp64 loss (and moves by +-3 nats = 1.5e-3 under 1-ulp perturbations of the weights or a different thread count),
    up to 2e-3 off single KL groups, and its gradient has cosine 0.9958 with the fp64 one; no f32 implementation can
    meet 1e-3 / 0.9999 there.  The strict bounds are therefore asserted at batch 2, where the problem is two
    orders of magnitude better conditioned, and a larger batch (6: 96 samples per channel in the 4x4 BatchNorms; 8 in
    tests/diag/diag_c2b.py - the fp64 oracle needs 80 s there, more than the suite can afford) checks the same quantities
    against bounds a few times the f32 oracle's own distance.  (The oracle passes took minutes on the GPU boxes while torch used every
    core the box SHOWS instead of the 16 it grants; tests/conftest.py caps the threads and both arms take seconds.)"""
    cfg = dict(CFG, n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=2, n_preprocess_cells=3,
               n_post_process_cells=3, n_groups_per_scale=[5, 10])
    global B
    old_b, B = B, batch
    try:
        orc, model, x, eps = build_pair(dev, torch.float32, cfg)
    finally:
        B = old_b
    strict = batch == 2
    assert model.n_trainable() == 62225021 and model.n_groups == 15
    orc.steps = model.steps = 100
    snap = ({k: v.detach().clone() for k, v in orc.s.params.items()}, {k: v.clone() for k, v in orc.s.state.items()})
    out_o = orc.train_step(x, eps, decay_steps=1000)
    out = model.train_step(x.float(), [e.float() for e in eps])
    torch.cuda.synchronize()
    assert abs(float(out["loss"]) - float(out_o["loss"])) / abs(float(out_o["loss"])) < (1e-3 if strict else 5e-3)
    assert rel(out["reconstruction_loss"], out_o["reconstruction_loss"]) < (1e-3 if strict else 1e-2)
    assert out["kl_per_group"].shape == (15, batch)
    for gi in range(15):                           # every one of the 15 KL terms on its own scale
        assert rel(out["kl_per_group"][gi], out_o["kl_per_group"][gi]) < (5e-3 if strict else 4e-2), gi
    assert rel(model.coeff, out_o["kl_coeff"]) < (5e-3 if strict else 4e-2)
    errs = sorted(((rel(model.ps.get_grad(k), g_o), k) for k, g_o in out_o["grads"].items()
                   if float(g_o.abs().max()) > 1e-6), reverse=True)
    med, p95 = errs[len(errs) // 2][0], errs[len(errs) // 20][0]
    go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
    gp = torch.cat([model.ps.get_grad(k).double().cpu().reshape(-1) for k in out_o["grads"]])
    assert go.numel() == 62225021
    cos = float((go * gp).sum() / (go.norm() * gp.norm()))
    print(f"C2 batch {batch}: gradient cosine {cos:.6f}, |g| {float(go.norm()):.3e}, per-tensor error median {med:.2e} "
          f"95th percentile {p95:.2e}, worst {errs[:4]}")
    if strict:
        assert cos > 0.9997                        # measured 0.999878 in every run (0.99975 before the two-level f32 accumulation)
        assert med < 2.5e-2 and p95 < 6e-2         # measured 1.5e-2 / 2.4e-2; a wrong layer shows up as O(1) errors
        # the few outliers are SE hidden units whose ReLU sits at its kink for one of the two images (a rounding
        # flips the unit: O(1) change of that row of w1 / b1) and biases feeding a BatchNorm (true gradient 0)
        assert sum(e > 0.25 for e, _ in errs) <= len(errs) // 100 and errs[0][0] < 2.0, errs[:12]
    else:
        assert cos > 0.993                         # measured 0.9971; f32 PyTorch vs fp64 PyTorch: 0.9958 (see the docstring)
        assert med < 0.15                          # measured 6.6e-2


def test_c2_architecture_parity_bf16_trained(lib, dev):
    """The benchmarked architecture in the benchmarked dtype.  At its RANDOM initialisation this 330-layer network is too
    ill-conditioned for any bf16 gradient comparison: measured at batch 2, the HIP bf16 gradient has cosine 0.25 with the
    fp64 one and the ORACLE ITSELF with bf16 storage (oracle_16bit) -0.09, losses 2.0e-2 / 1.2e-2 off, the worst KL
    group 0.33 / 0.23 (round 3).  After a little training it is not: the HIP f32 path trains the model for 300
    graph-replayed steps at batch 128 (synthetic data), then ONE training-mode forward + backward on a fresh batch of 4 is
    compared with the fp64 oracle at those weights, in f32 and in bf16.  Measured (tests/diag/diag_c2_bf16_trained.py,
    profiles/r03_c2_trained.txt): f32 loss 2.0e-7, worst KL group 7.7e-7, gradient cosine 1.00000, per-tensor median
    1.6e-6; bf16 loss 6.5e-4, worst KL group 1.4e-2, cosine 0.99959, median 3.0e-2, 90th percentile 6.4e-2 - next to
    the oracle with bf16 storage, the yardstick of what a correct bf16 implementation reaches: 1.1e-3, 3.7e-3, 0.99974,
    2.6e-2, 5.4e-2 (batch 2 after 200 steps: HIP 0.99483, emulation 0.99617).  Bounds = 2-3 x the measured values."""
    from nvae_tf_amd.models import NVAE
    from oracle.nvae_oracle import synthetic_batch
    cfg = dict(CFG, n_encoder_channels=32, n_decoder_channels=32, res_cells_per_group=2, n_preprocess_cells=3,
               n_post_process_cells=3, n_groups_per_scale=[5, 10])
    global B
    old_b, B = B, 4
    try:
        orc, _, x, eps = build_pair(dev, torch.float32, cfg)
    finally:
        B = old_b

    def make(dtype, batch, iters=1000):
        return NVAE(cfg["n_encoder_channels"], cfg["n_decoder_channels"], cfg["res_cells_per_group"],
                    cfg["n_preprocess_blocks"], cfg["n_preprocess_cells"], cfg["n_latent_per_group"], 2,
                    cfg["n_groups_per_scale"], cfg["n_postprocess_blocks"], cfg["n_post_process_cells"], cfg["sr_lambda"],
                    cfg["scale_factor"], cfg["total_epochs"], iters, True, [batch, 32, 32, 1], device=dev, dtype=dtype)
    trainer = make(torch.float32, 128, iters=20000)
    trainer.ps.load_named(orc.s.params, orc.s.state)
    xb = synthetic_batch(128, seed=11).float().to(dev)
    trainer.capture_train_step(xb.shape, warmup=1)
    trainer._static_x.copy_(xb)
    for _ in range(300):
        out = trainer.train_step_graphed(None)
    torch.cuda.synchronize()
    assert math.isfinite(float(out["loss"]))
    snap = ({k: v.detach().double().cpu() for k, v in trainer.ps.named().items()},
            {k: v.detach().double().cpu() for k, v in trainer.ps.named_state().items()})
    del trainer
    torch.cuda.empty_cache()
    orc.s.params = {k: v.clone().requires_grad_(True) for k, v in snap[0].items()}
    orc.s.state = {k: v.clone() for k, v in snap[1].items()}
    orc.steps = 300
    out_o = orc.train_step(x, eps, decay_steps=1000)
    go = torch.cat([out_o["grads"][k].reshape(-1) for k in out_o["grads"]])
    names = [k for k in out_o["grads"] if float(out_o["grads"][k].abs().max()) > 1e-6]
    lo_ = float(out_o["loss"].detach())
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        m = make(dtype, 4)
        m.ps.load_named(snap[0], snap[1])
        m.steps = 300
        out = m.train_step(x.float(), [e.float() for e in eps])
        torch.cuda.synchronize()
        gq = torch.cat([m.ps.get_grad(k).double().cpu().reshape(-1) for k in out_o["grads"]])
        es = sorted(rel(m.ps.get_grad(k), out_o["grads"][k]) for k in names)
        kl = max(rel(out["kl_per_group"][gi], out_o["kl_per_group"][gi]) for gi in range(15))
        res[dtype] = (abs(float(out["loss"]) - lo_) / abs(lo_), kl, float((go * gq).sum() / (go.norm() * gq.norm())),
                      es[len(es) // 2], es[len(es) * 9 // 10])
        del m
    print("C2 architecture after 300 f32 steps, batch 4, vs fp64 (loss, worst KL group, gradient cosine, per-tensor median, p90):", res)
    f32, b16 = res[torch.float32], res[torch.bfloat16]
    assert f32[0] < 1e-5 and f32[1] < 1e-4 and f32[2] > 0.999999 and f32[3] < 1e-4
    assert b16[0] < 2e-3 and b16[1] < 4e-2 and b16[2] > 0.999 and b16[3] < 6e-2 and b16[4] < 0.13


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16], ids=["f16", "bf16"])
def test_grad_range_normalisation_is_transparent(lib, dev, dtype):
    """The device-side renormalisation of the activation gradients (ops.GradScale: float16 on deep hierarchies) multiplies
    by powers of two and divides them out of the f32 parameter gradients again: where nothing over- or underflows the
    training step with it must equal the step without it up to the roundings of the merges - every parameter gradient
    of the shrunken parity model, losses, and the Adamax update."""
    orc, m_off, x, eps = build_pair(dev, dtype)
    _, m_on, _, _ = build_pair(dev, dtype)
    m_on.grad_rescale, m_off.grad_rescale = True, False
    orc.steps = 100
    exact = orc.train_step(x, eps, decay_steps=1000)["grads"]       # (only to tell real gradients from exact zeros)
    outs = []
    for m in (m_off, m_on):
        m.steps = 100
        outs.append(m.train_step(x.float(), [e.float() for e in eps]))
    torch.cuda.synchronize()
    assert abs(float(outs[0]["loss"]) - float(outs[1]["loss"])) < 1e-3 * abs(float(outs[0]["loss"]))
    # (conv biases in front of a BatchNorm and betas in front of another BatchNorm have a true gradient of 0: pure noise)
    worst = sorted(((rel(m_on.ps.get_grad(k), m_off.ps.get_grad(k)), k) for k in m_on.ps.slots
                    if float(exact[k].abs().max()) > 1e-6), reverse=True)
    print("largest differences with / without range normalisation:", worst[:5])
    # (the two runs also differ by the order of their f32 atomics, like any two runs)
    # measured: f16 worst tensor 1.4e-2 .. 1.9e-2, median 3.3e-3; bf16 worst 0.31 (pre.cell0.se.w1 / se.b1, the tensors
    # any two bf16 runs disagree on: bf16_spread 0.63 / 0.52), 95th percentile below 0.1 - the size of the difference
    # between any two runs of the respective dtype
    p95 = worst[len(worst) // 20][0]
    assert (worst[0][0] < 5e-2) if dtype == torch.float16 else (worst[0][0] < 1.5 and p95 < 0.15), worst[:5]
    assert worst[len(worst) // 2][0] < (1e-2 if dtype == torch.float16 else 4e-2)
    # (the updated parameters are not compared: Adamax moves an element whose true gradient is 0 by +-lr on noise alone)
    assert bool(torch.isfinite(m_on.ps.params).all()) and float((m_on.ps.params - m_off.ps.params).abs().max()) < 2.1e-3
