"""`torch.ops.nvae.*` (nvae_tf_amd/torch_ops.py): the dispatcher registrations of the hot-path kernel families against
plain PyTorch references of the same ops (fp64 on the GPU), forward and every gradient, plus `torch.library.opcheck`
(schema, fake implementation, autograd registration).  Tolerances: f32 2e-4 of the output scale, bf16 2.5e-2."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 2.5e-2, torch.float16: 3e-3}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]
CHECKS = ("test_schema", "test_autograd_registration", "test_faketensor")


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.fixture(scope="module")
def ops(lib):
    import nvae_tf_amd.torch_ops      # noqa: F401  (registers the library)
    return torch.ops.nvae


def _leaf(shape, dev, dtype=torch.float32, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dev, dtype).requires_grad_(True)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("k,B,H,cin,cout", [(1, 4, 8, 32, 64), (3, 4, 8, 32, 48), (5, 2, 16, 64, 192)])
def test_conv2d_same(ops, dev, dtype, k, B, H, cin, cout):
    x = _leaf((B, H, H, cin), dev, dtype, seed=1)
    w = _leaf((k, k, cin, cout), dev, scale=0.1, seed=2)
    b = _leaf((cout,), dev, seed=3)
    y = ops.conv2d_same(x, w, b)
    dy = torch.randn(y.shape, device=dev).to(dtype)
    y.backward(dy)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().to(dtype).double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True)
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(3, 2, 0, 1), br, padding=k // 2).permute(0, 2, 3, 1)
    yr.backward(dy.double())
    t = TOL[dtype]
    assert rel(y, yr) < t and rel(x.grad, xr.grad) < t and rel(w.grad, wr.grad) < t and rel(b.grad, br.grad) < t
    assert y.dtype == dtype and w.grad.dtype == torch.float32


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("stride,up", [(2, 1), (1, 2)], ids=["stride2", "upsample2"])
def test_conv2d_same_rescale(ops, dev, dtype, stride, up):
    """The Rescaler geometries (common.py:150-162): stride 2 with TF-'same' padding (bottom / right only on even sizes,
    SURVEY Q6) and a 3x3 conv behind a nearest upsample by 2 that is never materialised."""
    B, H, cin, cout, k = 3, 8, 32, 64, 3
    x = _leaf((B, H, H, cin), dev, dtype, seed=21)
    w = _leaf((k, k, cin, cout), dev, scale=0.1, seed=22)
    b = _leaf((cout,), dev, seed=23)
    y = ops.conv2d_same(x, w, b, stride, up)
    assert tuple(y.shape) == (B, H * up // stride, H * up // stride, cout)
    dy = torch.randn(y.shape, device=dev).to(dtype)
    y.backward(dy)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().to(dtype).double().requires_grad_(True)
    br = b.detach().double().requires_grad_(True)
    xin = xr.permute(0, 3, 1, 2)
    if up == 2:
        xin = xin.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        xin = F.pad(xin, (1, 1, 1, 1))
    else:
        xin = F.pad(xin, (0, 1, 0, 1))                 # TF 'same', stride 2, even size: total pad 1, all of it at the end
    yr = F.conv2d(xin, wr.permute(3, 2, 0, 1), br, stride=stride).permute(0, 2, 3, 1)
    yr.backward(dy.double())
    t = TOL[dtype]
    assert rel(y, yr) < t and rel(x.grad, xr.grad) < 2 * t and rel(w.grad, wr.grad) < t and rel(b.grad, br.grad) < t


@pytest.mark.parametrize("residual", [False, True], ids=["group0", "residual"])
def test_gauss_sample_kl(ops, dev, residual):
    """Sampler + KL of one latent group (common.py:76-102, models.py:197-201) against the closed forms in torch fp64, with a
    per-image upstream gradient on the KL output."""
    B, H, Lc = 5, 4, 20
    enc = _leaf((B, H, H, 2 * Lc), dev, seed=31)
    dec = _leaf((B, H, H, 2 * Lc), dev, seed=32) if residual else None
    eps = torch.randn(B, H, H, Lc, device=dev)
    z, kl = ops.gauss_sample_kl(enc, dec, eps)
    dz = torch.randn(z.shape, device=dev)
    wk = torch.rand(B, device=dev)
    ((z * dz).sum() + (kl * wk).sum()).backward()
    sc = lambda t: 5.0 * torch.tanh(t / 5.0)
    e = enc.detach().double().requires_grad_(True)
    d = dec.detach().double().requires_grad_(True) if residual else None
    if residual:
        mp, sp = sc(d[..., :Lc]), torch.exp(sc(d[..., Lc:])) + 1e-2
        mq, sq = sc(e[..., :Lc] + d[..., :Lc]), torch.exp(sc(e[..., Lc:] + d[..., Lc:])) + 1e-2
    else:
        mp, sp = torch.zeros_like(e[..., :Lc]), torch.ones_like(e[..., :Lc])
        mq, sq = sc(e[..., :Lc]), torch.exp(sc(e[..., Lc:])) + 1e-2
    zr = mq + sq * eps.double()
    t1, t2 = (mq - mp) / sp, sq / sp
    klr = (0.5 * (t1 * t1 + t2 * t2) - 0.5 - torch.log(t2)).sum((1, 2, 3))
    ((zr * dz.double()).sum() + (klr * wk.double()).sum()).backward()
    assert rel(z, zr) < 1e-5 and rel(kl, klr) < 1e-5
    assert rel(enc.grad, e.grad) < 1e-4
    if residual:
        assert rel(dec.grad, d.grad) < 1e-4


def test_kl_balance_bn_absmax_sn_adamax(ops, dev):
    """The remaining families of SURVEY 8b: KL balancing coefficients (models.py:203-213), the BatchNorm-gamma regulariser
    with its subgradient (models.py:252-267), one spectral-norm power iteration (TFA), one Keras Adamax step."""
    from nvae_tf_amd.models import NVAE
    # --- kl_balance: coefficient_g proportional to mean_b|KL_g| / alpha_g... = the product's own host formula
    G, B = 6, 8
    kl_all = torch.rand(G, B, device=dev) * 10 + 0.1
    alphas = NVAE.calculate_kl_alphas(2, [2, 4]).to(dev)
    coeff = ops.kl_balance(kl_all, alphas)
    m = kl_all.double().abs().mean(1) + 0.01                      # models.py:210
    ref = m / alphas.double()
    ref = ref / ref.mean()                                        # models.py:211-213: normalised to mean 1
    assert rel(coeff, ref) < 1e-5
    # --- bn_gamma_absmax: lambda * sum_l max|gamma_l|, gradient lambda * sign at each layer's arg-max
    params = torch.randn(96, device=dev, requires_grad=True)
    table = torch.tensor([[0, 32], [40, 24], [64, 32]], dtype=torch.int32, device=dev)
    loss, _ = ops.bn_gamma_absmax(params, table, 0.01)
    (3.0 * loss).backward()
    pr = params.detach().double().requires_grad_(True)
    lr_ = 0.01 * sum(pr[o:o + c].abs().max() for o, c in table.tolist())
    (3.0 * lr_).backward()
    assert abs(float(loss) - float(lr_)) < 1e-6 and rel(params.grad, pr.grad) < 1e-6
    # --- spectral_norm_step
    w = torch.randn(3, 3, 16, 24, device=dev)
    u = torch.randn(24, device=dev) * 0.02
    sigma, u2 = ops.spectral_norm_step(w, u)
    w2 = w.double().reshape(-1, 24)
    v = u.double().reshape(1, -1) @ w2.t()
    v = v / v.norm()
    un = v @ w2
    un = un / un.norm()
    assert abs(float(sigma) - float(v @ w2 @ un.t())) / float(sigma) < 1e-5 and rel(u2, un.reshape(-1)) < 1e-5
    # --- adamax_step against torch.optim.Adamax (Keras formulation: eps added to u)
    n = 4096
    p, g = torch.randn(n, device=dev), torch.randn(n, device=dev)
    m_, u_ = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pr_, mr, ur = p.double().clone(), torch.zeros(n, device=dev, dtype=torch.float64), torch.zeros(n, device=dev, dtype=torch.float64)
    for t in (1, 2, 3):
        lr_t = 1e-3 / (1 - 0.9 ** t)
        ops.adamax_step(p, g, m_, u_, lr_t, 0.9, 0.999, 1e-7)
        mr = 0.9 * mr + 0.1 * g.double()
        ur = torch.maximum(0.999 * ur, g.double().abs())
        pr_ = pr_ - lr_t * mr / (ur + 1e-7)
    assert rel(p, pr_) < 1e-6 and rel(m_, mr) < 1e-6 and rel(u_, ur) < 1e-6
    # --- schemas / fake implementations / autograd registrations
    enc = torch.randn(2, 4, 4, 40, device=dev, requires_grad=True)
    eps = torch.randn(2, 4, 4, 20, device=dev)
    torch.library.opcheck(ops.gauss_sample_kl.default, (enc, None, eps), test_utils=CHECKS)
    torch.library.opcheck(ops.kl_balance.default, (kl_all, alphas), test_utils=("test_schema", "test_faketensor"))
    torch.library.opcheck(ops.bn_gamma_absmax.default, (params.detach().requires_grad_(True), table, 0.01), test_utils=CHECKS)
    torch.library.opcheck(ops.spectral_norm_step.default, (w, u), test_utils=("test_schema", "test_faketensor"))
    x = torch.randn(2, 8, 8, 32, device=dev, requires_grad=True)
    wc = (torch.randn(3, 3, 32, 32, device=dev) * 0.1).requires_grad_(True)
    torch.library.opcheck(ops.conv2d_same.default, (x, wc, None, 2, 1), test_utils=CHECKS)
    torch.library.opcheck(ops.conv2d_same.default, (x, wc, None, 1, 2), test_utils=CHECKS)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,H,C_", [(8, 4, 128), (3, 8, 64), (2, 12, 72)])
def test_dwconv5(ops, dev, dtype, B, H, C_):
    x = _leaf((B, H, H, C_), dev, dtype, seed=4)
    w = _leaf((5, 5, C_), dev, scale=0.2, seed=5)
    b = _leaf((C_,), dev, seed=6)
    y = ops.dwconv5(x, w, b)
    dy = torch.randn(y.shape, device=dev).to(dtype)
    y.backward(dy)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr.permute(0, 3, 1, 2), wr.permute(2, 0, 1).unsqueeze(1), br, padding=2, groups=C_).permute(0, 2, 3, 1)
    yr.backward(dy.double())
    t = TOL[dtype]
    assert rel(y, yr) < t and rel(x.grad, xr.grad) < t and rel(w.grad, wr.grad) < t and rel(b.grad, br.grad) < t


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("act", [0, 1], ids=["none", "swish"])
def test_bn_act(ops, dev, dtype, act):
    B, H, C_ = 16, 4, 96
    x = _leaf((B, H, H, C_), dev, dtype, seed=7)
    gamma = (torch.rand(C_, device=dev) + 0.5).requires_grad_(True)
    beta = _leaf((C_,), dev, scale=0.3, seed=8)
    y, mean, invstd = ops.bn_act(x, gamma, beta, act, 1e-5)
    dy = torch.randn(y.shape, device=dev).to(dtype)
    y.backward(dy)
    xr, gr, br = (t.detach().double().requires_grad_(True) for t in (x, gamma, beta))
    m = xr.mean((0, 1, 2)); v = xr.var((0, 1, 2), unbiased=False)
    z = (xr - m) / torch.sqrt(v + 1e-5) * gr + br
    yr = z * torch.sigmoid(z) if act else z
    yr.backward(dy.double())
    t = TOL[dtype]
    assert rel(y, yr) < t and rel(mean, m) < 1e-4 and rel(invstd, 1 / torch.sqrt(v + 1e-5)) < 1e-4
    assert rel(x.grad, xr.grad) < 2 * t and rel(gamma.grad, gr.grad) < 2 * t and rel(beta.grad, br.grad) < 2 * t


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_se_residual(ops, dev, dtype):
    B, H, C_, Hd = 6, 4, 128, 8
    x = _leaf((B, H, H, C_), dev, dtype, seed=9)
    skip = _leaf((B, H, H, C_), dev, dtype, seed=10)
    w1 = _leaf((C_, Hd), dev, scale=0.2, seed=11); b1 = _leaf((Hd,), dev, scale=0.1, seed=12)
    w2 = _leaf((Hd, C_), dev, scale=0.2, seed=13); b2 = _leaf((C_,), dev, scale=0.1, seed=14)
    y = ops.se_residual(x, skip, w1, b1, w2, b2, 1.0, 0.1)[0]
    dy = torch.randn(y.shape, device=dev).to(dtype)
    y.backward(dy)
    r = [t.detach().double().requires_grad_(True) for t in (x, skip, w1, b1, w2, b2)]
    gate = torch.sigmoid(torch.relu(r[0].mean((1, 2)) @ r[2] + r[3]) @ r[4] + r[5])
    yr = r[1] + 0.1 * r[0] * gate[:, None, None, :]
    yr.backward(dy.double())
    t = TOL[dtype]
    assert rel(y, yr) < t
    for a, b in zip((x, skip, w1, b1, w2, b2), r):
        assert rel(a.grad, b.grad) < 2 * t


def test_bernoulli_nll(ops, dev):
    B, H = 5, 32
    logits = _leaf((B, H, H, 1), dev, seed=15)
    x = (torch.rand(B, H, H, 1, device=dev) < 0.2).to(torch.bfloat16)
    nll = ops.bernoulli_nll(logits, x)
    wgt = torch.rand(B, device=dev)
    (nll * wgt).sum().backward()
    lr = logits.detach().double().requires_grad_(True)
    ref = F.binary_cross_entropy_with_logits(lr, x.double(), reduction="none").sum((1, 2, 3))
    (ref * wgt.double()).sum().backward()
    assert rel(nll, ref) < 1e-5 and rel(logits.grad, lr.grad) < 8e-3       # (the gradient leaves the kernel in bf16, like x)


def test_opcheck_and_stock_autograd_loop(ops, dev):
    """The registrations are usable the way SURVEY 8b asked: from a stock torch module / optimizer loop."""
    x = torch.randn(4, 8, 8, 32, device=dev, requires_grad=True)
    w = (torch.randn(3, 3, 32, 32, device=dev) * 0.1).requires_grad_(True)
    b = torch.zeros(32, device=dev, requires_grad=True)
    torch.library.opcheck(ops.conv2d_same.default, (x, w, b), test_utils=CHECKS)
    dw = (torch.randn(5, 5, 32, device=dev) * 0.2).requires_grad_(True)
    torch.library.opcheck(ops.dwconv5.default, (x, dw, b), test_utils=CHECKS)
    gamma = torch.ones(32, device=dev, requires_grad=True)
    torch.library.opcheck(ops.bn_act.default, (x, gamma, b, 1, 1e-5), test_utils=CHECKS)

    class Cell(torch.nn.Module):          # a depthwise-separable residual cell out of the registered ops
        def __init__(self):
            super().__init__()
            p = lambda *s, sc=0.1: torch.nn.Parameter(torch.randn(*s, device=dev) * sc)
            self.w1, self.b1 = p(1, 1, 32, 64), p(64, sc=0.0)
            self.dw, self.db = p(5, 5, 64), p(64, sc=0.0)
            self.w2, self.b2 = p(1, 1, 64, 32), p(32, sc=0.0)
            self.g1, self.be1 = torch.nn.Parameter(torch.ones(64, device=dev)), p(64, sc=0.0)
            self.sw1, self.sb1, self.sw2, self.sb2 = p(32, 8), p(8, sc=0.0), p(8, 32), p(32, sc=0.0)

        def forward(self, x):
            h = ops.conv2d_same(x, self.w1, self.b1)
            h = ops.bn_act(h, self.g1, self.be1, 1, 1e-5)[0]
            h = ops.dwconv5(h, self.dw, self.db)
            h = ops.conv2d_same(h, self.w2, self.b2)
            return ops.se_residual(h, x, self.sw1, self.sb1, self.sw2, self.sb2, 1.0, 0.1)[0]

    cell = Cell()
    opt = torch.optim.Adamax(cell.parameters(), lr=1e-2)
    data = torch.randn(8, 8, 8, 32, device=dev)
    target = torch.randn(8, 8, 8, 32, device=dev)
    losses = []
    for _ in range(40):
        opt.zero_grad()
        loss = ((cell(data) - target) ** 2).mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(b < a for a, b in zip(losses, losses[1:])) and losses[-1] < 0.95 * losses[0], losses
