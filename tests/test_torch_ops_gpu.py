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
