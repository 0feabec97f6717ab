"""GPU parity tests of the individual kernels, called through the C ABI (ctypes), against fp64
PyTorch-CPU references of the same op.  Tolerances: f32 kernels 2e-4 relative to the output scale
(f32 MFMA is an exact fmaf chain, the slack covers summation order); bf16 kernels 2e-2; f16 kernels (11 significant
bits, the BASELINE.json configs[4] dtype) 3e-3."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 2.5e-2, torch.float16: 3e-3}
DTYPES = [torch.float32, torch.bfloat16, torch.float16]


def rel_err(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


def q(t, dtype):
    """Quantise a reference tensor to what the kernel will actually read."""
    return t.to(dtype).double()


def make_ctx(ps, dtype, training=True, record=True):
    from nvae_tf_amd.ops import Ctx
    ps.begin_step()
    ps.prepare_weights(spectral_norm=False)
    return Ctx(ps, dtype, training=training, record=record)


def ref_conv(x, w, b, stride, pad, up, out_hw):
    """fp64 reference: nearest-upsample, explicit (top,left) padding, HWIO weights, NHWC."""
    if up > 1:
        x = x.repeat_interleave(up, dim=1).repeat_interleave(up, dim=2)
    kh, kw = w.shape[0], w.shape[1]
    Ho, Wo = out_hw
    H, W = x.shape[1], x.shape[2]
    pb = (Ho - 1) * stride + kh - H - pad[0]
    pr = (Wo - 1) * stride + kw - W - pad[1]
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pad[1], max(pr, 0), pad[0], max(pb, 0))) if min(pad) >= 0 else xn
    if min(pad) < 0:   # negative pad = crop from the top/left (SkipScaler shifts)
        xn = xn[:, :, -pad[0]:, -pad[1]:]
        pb = (Ho - 1) * stride + kh - xn.shape[2]
        pr = (Wo - 1) * stride + kw - xn.shape[3]
        xn = F.pad(xn, (0, max(pr, 0), 0, max(pb, 0)))
    y = F.conv2d(xn.contiguous(), w.permute(3, 2, 0, 1).contiguous(), b, stride=stride)
    return y.permute(0, 2, 3, 1)[:, :Ho, :Wo, :]


CONV_CASES = [
    # k, cin, cout, stride, up, H, B, pad(None=same), bias, residual
    dict(k=3, cin=16, cout=24, stride=1, up=1, H=8, B=2),
    dict(k=5, cin=48, cout=48, stride=1, up=1, H=6, B=3),
    dict(k=3, cin=16, cout=32, stride=2, up=1, H=8, B=2),
    dict(k=3, cin=32, cout=16, stride=1, up=2, H=4, B=2),
    dict(k=1, cin=64, cout=200, stride=1, up=1, H=16, B=2),
    dict(k=1, cin=16, cout=8, stride=2, up=1, H=8, B=2, pad=(-1, -1)),
    dict(k=3, cin=24, cout=40, stride=1, up=1, H=4, B=5, residual=True),
    dict(k=5, cin=192, cout=192, stride=1, up=1, H=8, B=2, bias=False),
    dict(k=5, cin=128, cout=192, stride=1, up=1, H=32, B=16),     # halo-tile kernel (16x16 patches), 5x5
    dict(k=1, cin=64, cout=512, stride=1, up=1, H=32, B=16, family=14),   # short K, 512 tiles: 128x128, two workgroups per CU
    dict(k=3, cin=128, cout=200, stride=1, up=1, H=16, B=64),     # halo-tile kernel, 3x3, ragged N
    dict(k=5, cin=128, cout=192, stride=1, up=1, H=32, B=16, bias=False),   # + halo weight-gradient kernel
    dict(k=3, cin=64, cout=384, stride=1, up=1, H=16, B=64, bias=False),    # halo wgrad, 3x3, two n-tiles
    dict(k=3, cin=1, cout=16, stride=1, up=1, H=8, B=2),          # direct path (stem)
    dict(k=3, cin=16, cout=1, stride=1, up=1, H=8, B=2),          # direct dgrad/wgrad (logit head)
    dict(k=1, cin=20, cout=32, stride=1, up=1, H=4, B=4),         # latent half of the combiner
    # whole-image 3x3 kernel k_conv_img (16-bit; f32 runs the generic kernel): 4x4 and 8x8 images, 128 / 256 channels,
    # a ragged last M-tile (B*H*H not a multiple of 128), a ragged last N-tile, residual, no bias
    dict(k=3, cin=256, cout=256, stride=1, up=1, H=4, B=16, img=True),
    dict(k=3, cin=128, cout=128, stride=1, up=1, H=8, B=5, residual=True, img=True),
    dict(k=3, cin=128, cout=72, stride=1, up=1, H=4, B=18, bias=False, img=True),
    dict(k=3, cin=256, cout=64, stride=1, up=1, H=8, B=4, img=True),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "k{k}_ci{cin}_co{cout}_s{stride}_u{up}_H{H}".format(**c))
def test_conv_fwd_bwd(lib, dev, dtype, case):
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    k, cin, cout, stride, up, H, B = (case[n] for n in ("k", "cin", "cout", "stride", "up", "H", "B"))
    use_bias = case.get("bias", True)
    g = torch.Generator().manual_seed(1234)
    ps = ParamStore(seed=3)
    conv = ps.conv("c", k, cin, cout, bias=use_bias)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    if use_bias:
        ps.view(conv.b).copy_(torch.randn(cout, generator=g))
    x = torch.randn(B, H, H, cin, generator=g)
    Hu = H * up
    pad = case.get("pad")
    if pad is None:
        pad = (ops.same_pad(Hu, k, stride)[0],) * 2
        Ho = -(-Hu // stride)
    else:
        Ho = (H + 1) // 2
    res = torch.randn(B, Ho, Ho, cout, generator=g) if case.get("residual") else None
    dy = torch.randn(B, Ho, Ho, cout, generator=g)

    # ---- reference (on inputs quantised as the kernel sees them)
    w64 = q(ps.get("c.w").cpu(), dtype).requires_grad_(True)
    b64 = ps.view(conv.b).cpu().double().requires_grad_(True) if use_bias else None
    x64 = q(x, dtype).requires_grad_(True)
    y_ref = ref_conv(x64, w64, b64, stride, pad, up, (Ho, Ho))
    if res is not None:
        y_ref = y_ref + q(res, dtype)
    dy64 = q(dy, dtype)
    grads = torch.autograd.grad(y_ref, [x64, w64] + ([b64] if use_bias else []), dy64)

    # ---- kernels
    ctx = make_ctx(ps, dtype)
    xv = Var(x.to(dev, dtype))
    rv = Var(res.to(dev, dtype)) if res is not None else None
    kw = dict(stride=stride, up=up, residual=rv, bias=use_bias)
    if case.get("pad") is not None:
        kw.update(pad=pad, out_hw=(Ho, Ho))
    if case.get("img"):
        import ctypes as C_
        from nvae_tf_amd import _lib as L_
        gg = L_.ConvGeom(B, H, H, cin, H, H, cout, 3, 3, 1, 1, 1, 1, 0, cin, cout, cout)
        assert lib.nvae_conv_img_ok(L_.dtype_code(dtype), C_.byref(gg)) == (0 if dtype == torch.float32 else 1)
    if case.get("family"):
        import ctypes as C_
        from nvae_tf_amd import _lib as L_
        gg = L_.ConvGeom(B, H, H, cin, H, H, cout, k, k, 1, (k - 1) // 2, (k - 1) // 2, 1, 0, cin, cout, cout)
        assert lib.nvae_conv_gemm_family(L_.dtype_code(dtype), C_.byref(gg)) == case["family"]
    y = ops.conv2d(ctx, xv, conv, **kw)
    y.g = dy.to(dev, dtype)
    ctx.backward()
    torch.cuda.synchronize()
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < tol
    assert rel_err(xv.g, grads[0]) < tol
    assert rel_err(ps.get_grad("c.w"), grads[1]) < tol
    if use_bias:
        assert rel_err(ps.get_grad("c.b"), grads[2]) < tol
    if rv is not None:
        assert rel_err(rv.g, dy64) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(32, 16, 384), (16, 32, 192)], ids=["16x16_384", "32x32_192"])
def test_conv_halo_forms_bit_identical(lib, dev, dtype, shape):
    """The two 16-bit forms of the dense 5x5 halo kernel - eight ping-pong waves of 64 x 96 (production) and four
    software-pipelined waves of 128 x 96 with AGPR-pinned accumulators (nvae_conv_halo4_enable(1)) - accumulate every output
    element in the same order: outputs equal bit for bit in the three forms the step launches (plain, forward with the
    statistics epilogue, data gradient with the BatchNorm-backward epilogue); the f32 slabs differ by the order of their
    atomics only.  The 8-wave form is the one every other test checks against torch / the oracle."""
    import ctypes as C_
    from nvae_tf_amd import _lib as L_
    B, hw, ci = shape
    code = L_.dtype_code(dtype)
    torch.manual_seed(3)
    x = torch.randn(B, hw, hw, ci, device=dev).to(dtype)
    w = (torch.randn(ci, 25 * ci, device=dev) / (25 * ci) ** 0.5).to(dtype)
    g = L_.ConvGeom(B, hw, hw, ci, hw, hw, ci, 5, 5, 1, 2, 2, 1, 0, ci, ci, ci)
    rows = lib.nvae_conv_gemm_stats_rows(code, C_.byref(g))
    coef = torch.rand(4, ci, device=dev) + 0.5
    outs = {}
    try:
        for form in (0, 1):
            lib.nvae_conv_halo4_enable(form)
            y0, y1, dx = (torch.full((B, hw, hw, ci), float("nan"), device=dev, dtype=dtype) for _ in range(3))
            slab, part = torch.zeros(rows, 2, ci, device=dev), torch.zeros(rows, 2, ci, device=dev)
            dgb, k0k1 = torch.zeros(2, ci, device=dev), torch.zeros(2, ci, device=dev)
            f = L_.BnBwdFuse(L_.ptr(x), ci, L_.ACT_SWISH, 0, L_.ptr(coef[0]), L_.ptr(coef[1]), L_.ptr(coef[2]), L_.ptr(coef[3]),
                             L_.ptr(part), None, L_.ptr(dgb[0]), L_.ptr(dgb[1]), L_.ptr(k0k1))
            L_.call("nvae_conv_gemm", code, C_.byref(g), L_.ptr(x), L_.ptr(w), 25 * ci, None, None, L_.ptr(y0), 0, None)
            L_.call("nvae_conv_gemm", code, C_.byref(g), L_.ptr(x), L_.ptr(w), 25 * ci, None, None, L_.ptr(y1), 0, L_.ptr(slab))
            L_.call("nvae_conv_gemm_bnbwd", code, C_.byref(g), L_.ptr(x), L_.ptr(w), 25 * ci, None, None, L_.ptr(dx), C_.byref(f))
            torch.cuda.synchronize()
            outs[form] = (y0, y1, dx, slab, part)
    finally:
        lib.nvae_conv_halo4_enable(0)
    a, b = outs[0], outs[1]
    assert bool(torch.isfinite(b[0].float()).all()) and bool(torch.isfinite(b[2].float()).all())
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert torch.equal(a[0], a[1])                     # the statistics epilogue does not touch the output
    for p_, q_ in ((a[3], b[3]), (a[4], b[4])):
        assert float((p_ - q_).abs().max() / p_.abs().max()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(32, 4, 256, 40, 3), (16, 8, 128, 128, 3), (24, 4, 1536, 256, 1), (3, 8, 768, 72, 1)],
                         ids=lambda s: "B{}_H{}_{}-{}_k{}".format(*s))
def test_conv_split_k(lib, dev, dtype, shape):
    """Split-K of the generic implicit GEMM (nvae_conv_set_workspace): S = 2, 3, 5 K-slices per tile against the unsplit
    launch and an fp64 reference; the slice-ordered sum makes the result independent of the arrival order (bit-equal
    over repeated launches) and the arrival counters are back at zero after every launch.  The first shape is split by
    the launcher's own choice (narrow sampler conv), the others only when forced."""
    import ctypes as C_
    from nvae_tf_amd import _lib as L_
    from nvae_tf_amd import ops
    B, H, ci, co, k = shape
    L_.ensure_workspace(dev)
    slab, counters = L_._workspace["ws"]
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, H, ci, generator=g)
    w = torch.randn(k, k, ci, co, generator=g) / (k * k * ci) ** 0.5
    pad = (k - 1) // 2
    y_ref = ref_conv(q(x, dtype), q(w, dtype), None, 1, (pad, pad), 1, (H, H))
    xd = x.to(dev, dtype)
    wT = w.permute(3, 0, 1, 2).reshape(co, k * k * ci).contiguous().to(dev, dtype)
    geom = L_.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, pad, pad, 1, 0, ci, co, co)
    code = L_.dtype_code(dtype)
    lib.nvae_conv_img_enable(0)
    try:
        outs = {}
        for S in (1, 0, 2, 3, 5):                 # 1 = never split, 0 = the launcher's choice
            lib.nvae_conv_gemm_force_split(S)
            runs = []
            for _ in range(3):
                y = torch.full((B, H, H, co), float("nan"), device=dev, dtype=dtype)
                L_.call("nvae_conv_gemm", code, C_.byref(geom), L_.ptr(xd), L_.ptr(wT), k * k * ci, None, None, L_.ptr(y), 0, None)
                runs.append(y)
            torch.cuda.synchronize()
            assert int(counters.abs().sum()) == 0
            assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
            assert rel_err(runs[0], y_ref) < TOL[dtype], S
            outs[S] = runs[0]
        # a split changes the f32 summation order only
        for S in (0, 2, 3, 5):
            assert rel_err(outs[S], outs[1].double().cpu()) < (1e-5 if dtype == torch.float32 else TOL[dtype])
    finally:
        lib.nvae_conv_gemm_force_split(0)
        lib.nvae_conv_img_enable(1)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_conv_channel_slices(lib, dev, dtype):
    """Concat-free DecoderSampleCombiner (row slices + accumulate) and SkipScaler (output slices)."""
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    from nvae_tf_amd.decoder import DecoderSampleCombiner
    from nvae_tf_amd.preprocess import SkipScaler
    g = torch.Generator().manual_seed(7)
    ps = ParamStore(seed=5)
    comb = DecoderSampleCombiner(ps, "comb", 32, 20, 48)
    skips = {40: SkipScaler(ps, "skip40", 16, 40), 64: SkipScaler(ps, "skip64", 16, 64)}
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    B, H = 3, 4
    x, z = torch.randn(B, H, H, 32, generator=g), torch.randn(B, H, H, 20, generator=g)
    dy = torch.randn(B, H, H, 48, generator=g)
    w = q(ps.get("comb.conv.w").cpu(), dtype).requires_grad_(True)
    x64, z64 = q(x, dtype).requires_grad_(True), q(z, dtype).requires_grad_(True)
    y_ref = ref_conv(torch.cat((x64, z64), 3), w, ps.get("comb.conv.b").cpu().double(), 1, (0, 0), 1, (H, H))
    gr = torch.autograd.grad(y_ref, [x64, z64, w], q(dy, dtype))
    ctx = make_ctx(ps, dtype)
    xv, zv = Var(x.to(dev, dtype)), Var(z.to(dev, dtype))
    y = comb(ctx, xv, zv)
    y.g = dy.to(dev, dtype)
    ctx.backward()
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < tol
    assert rel_err(xv.g, gr[0]) < tol and rel_err(zv.g, gr[1]) < tol
    assert rel_err(ps.get_grad("comb.conv.w"), gr[2]) < tol

    # SkipScaler: 40 channels = unaligned slices (scalar kernels), 64 = aligned (MFMA kernels)
    H = 8
    for nch, skip in skips.items():
        name = f"skip{nch}"
        x = torch.randn(B, H, H, 16, generator=g)
        dy = torch.randn(B, H // 2, H // 2, nch, generator=g)
        x64 = q(x, dtype).requires_grad_(True)
        o = x64 * torch.sigmoid(x64)
        if dtype != torch.float32:
            o = o + (o.to(dtype).double() - o).detach()   # the kernel stores swish(x) in the 16-bit activation type
        views = [o, o[:, 1:, 1:, :], o[:, :, 1:, :], o[:, 1:, :, :]]
        ws = [q(ps.get(f"{name}.conv{i + 1}.w").cpu(), dtype).requires_grad_(True) for i in range(4)]
        bs = [ps.get(f"{name}.conv{i + 1}.b").cpu().double() for i in range(4)]
        parts = [ref_conv(v, w_, b_, 2, (0, 0), 1, (H // 2, H // 2)) for v, w_, b_ in zip(views, ws, bs)]
        y_ref = torch.cat(parts, 3)
        gr = torch.autograd.grad(y_ref, [x64] + ws, q(dy, dtype))
        ctx = make_ctx(ps, dtype)
        xv = Var(x.to(dev, dtype))
        y = skip(ctx, xv)
        y.g = dy.to(dev, dtype)
        ctx.backward()
        assert rel_err(y.t, y_ref) < tol, nch
        assert rel_err(xv.g, gr[0]) < 2 * tol, nch
        for i in range(4):
            assert rel_err(ps.get_grad(f"{name}.conv{i + 1}.w"), gr[1 + i]) < tol, (nch, i)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("act", [0, 1], ids=["none", "swish"])
@pytest.mark.parametrize("shape", [(4, 4, 4, 32), (3, 8, 8, 192), (2, 4, 4, 1536), (5, 16, 16, 64)])
def test_bn_act(lib, dev, dtype, act, shape):
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    g = torch.Generator().manual_seed(11)
    C_ = shape[3]
    ps = ParamStore(seed=1)
    bn = ps.bn("bn", C_)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    ps.get("bn.gamma").copy_(torch.rand(C_, generator=g) + 0.5)
    ps.get("bn.beta").copy_(torch.randn(C_, generator=g) * 0.3)
    x = torch.randn(shape, generator=g) * 1.5 + 0.4
    dy = torch.randn(shape, generator=g)
    x64 = q(x, dtype).requires_grad_(True)
    gam = ps.get("bn.gamma").cpu().double().requires_grad_(True)
    bet = ps.get("bn.beta").cpu().double().requires_grad_(True)
    mean, var = x64.mean((0, 1, 2)), x64.var((0, 1, 2), unbiased=False)
    pre = (x64 - mean) * torch.rsqrt(var + 1e-5) * gam + bet
    y_ref = pre * torch.sigmoid(pre) if act else pre
    gr = torch.autograd.grad(y_ref, [x64, gam, bet], q(dy, dtype))
    ctx = make_ctx(ps, dtype)
    xv = Var(x.to(dev, dtype))
    y = ops.bn_act(ctx, xv, bn, act)
    y.g = dy.to(dev, dtype)
    ctx.backward()
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < tol
    assert rel_err(xv.g, gr[0]) < 2 * tol
    assert rel_err(ps.get_grad("bn.gamma"), gr[1]) < tol
    assert rel_err(ps.get_grad("bn.beta"), gr[2]) < tol
    # Keras moving statistics: moving = 0.05*moving + 0.95*batch
    assert rel_err(ps.get_state("bn.rm"), 0.95 * mean) < 1e-4
    assert rel_err(ps.get_state("bn.rv"), 0.05 + 0.95 * var) < 1e-4
    # inference mode uses the moving statistics
    ctx = make_ctx(ps, dtype, training=False, record=False)
    y2 = ops.bn_act(ctx, Var(x.to(dev, dtype), False), bn, act)
    rm, rv = ps.get_state("bn.rm").cpu().double(), ps.get_state("bn.rv").cpu().double()
    pre = (x64 - rm) * torch.rsqrt(rv + 1e-5) * gam + bet
    assert rel_err(y2.t, pre * torch.sigmoid(pre) if act else pre) < tol


@pytest.mark.parametrize("shape", [(128, 8, 8, 128), (64, 4, 4, 1536), (16, 32, 32, 40)])
def test_bn_fused_finalize_matches_two_launch(lib, dev, shape):
    """"Last arriver finalizes" (csrc/bn_fin.h) against the separate reduce + finalize launches, at
    shapes with up to 128 row splits, re-using one counter buffer across launches."""
    from nvae_tf_amd._lib import call, ptr
    g = torch.Generator().manual_seed(3)
    Cc = shape[3]
    rows = shape[0] * shape[1] * shape[2]
    x = (torch.randn(shape, generator=g) * 1.3 + 0.2).to(dev, torch.bfloat16)
    dy = torch.randn(shape, generator=g).to(dev, torch.bfloat16)
    gamma = (torch.rand(Cc, generator=g) + 0.5).to(dev)
    beta = torch.randn(Cc, generator=g).to(dev)
    S = lib.nvae_reduce_splits(rows, Cc)
    counters = torch.zeros(256, dtype=torch.int32, device=dev)

    def run(fused):
        rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
        coef = torch.empty(4, Cc, device=dev)
        sc, sh, mean, istd = (ptr(coef) + i * Cc * 4 for i in range(4))
        part = torch.empty(S, 2, Cc, device=dev)
        dg, db, k = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.empty(2, Cc, device=dev)
        if fused:
            call("nvae_bn_stats_fin", 1, ptr(x), rows, Cc, ptr(part), ptr(counters), ptr(gamma), ptr(beta), ptr(rm),
                 ptr(rv), 0.05, 1e-5, sc, sh, mean, istd)
            call("nvae_bn_bwd_reduce_fin", 1, ptr(x), ptr(dy), rows, Cc, sc, sh, mean, istd, 1, ptr(part),
                 ptr(counters), ptr(dg), ptr(db), ptr(k), 0)
        else:
            call("nvae_bn_stats", 1, ptr(x), rows, Cc, ptr(part))
            call("nvae_bn_finalize", 1, ptr(part), rows, Cc, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.05, 1e-5, sc,
                 sh, mean, istd)
            call("nvae_bn_bwd_reduce", 1, ptr(x), ptr(dy), rows, Cc, sc, sh, 1, ptr(part))
            call("nvae_bn_bwd_finalize", 1, ptr(part), rows, Cc, sc, mean, istd, ptr(dg), ptr(db), ptr(k), 0)
        torch.cuda.synchronize()
        return [t.clone() for t in (coef, rm, rv, dg, db, k)]

    ref = run(False)
    for _ in range(3):
        got = run(True)
        assert int(counters.abs().sum()) == 0
        for a, b in zip(got, ref):
            assert rel_err(a, b) < 1e-5


@pytest.mark.parametrize("fused", [True, False], ids=["stats_fin", "stats+finalize"])
def test_bn_stats_f32_survive_large_means(lib, dev, fused):
    """f32 tensors whose channel means dwarf their spread (the depthwise-conv outputs in front of bn3 reach
    |mean| / std = 15-19 at initialisation; here 200): the f32 strip reduce accumulates x and x^2 in f64, so the
    single-pass variance keeps ~7 digits where f32 sums would keep 2-3."""
    from nvae_tf_amd._lib import call, ptr
    g = torch.Generator().manual_seed(11)
    B, H, W_, Cc = 64, 4, 4, 192
    rows = B * H * W_
    x = (torch.randn(B, H, W_, Cc, generator=g) * 0.05 + torch.linspace(-10, 10, Cc)).float()
    x64 = x.double()
    mean_ref, var_ref = x64.mean((0, 1, 2)), x64.var((0, 1, 2), unbiased=False)
    xd = x.to(dev)
    S = lib.nvae_reduce_splits(rows, Cc)
    part = torch.empty(S, 2, Cc, dtype=torch.float64, device=dev)
    gamma, beta = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    rm, rv = torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    coef = torch.empty(4, Cc, device=dev)
    sc, sh, mean, istd = (ptr(coef) + i * Cc * 4 for i in range(4))
    if fused:
        counters = torch.zeros(256, dtype=torch.int32, device=dev)
        call("nvae_bn_stats_fin", 0, ptr(xd), rows, Cc, ptr(part), ptr(counters), ptr(gamma), ptr(beta), ptr(rm),
             ptr(rv), 0.05, 1e-5, sc, sh, mean, istd)
    else:
        call("nvae_bn_stats", 0, ptr(xd), rows, Cc, ptr(part))
        call("nvae_bn_finalize", 0, ptr(part), rows, Cc, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), 0.05, 1e-5, sc, sh,
             mean, istd)
    torch.cuda.synchronize()
    got_var = 1.0 / coef[3].double().cpu() ** 2 - 1e-5
    assert float(((coef[2].double().cpu() - mean_ref).abs() / mean_ref.abs().clamp_min(1e-3)).max()) < 1e-6
    assert float(((got_var - var_ref).abs() / var_ref).max()) < 1e-4        # f32 sums: ~(200^2) * 6e-8 = 2.4e-3
    assert rel_err(rv, 0.05 + 0.95 * var_ref) < 1e-5


FUSED_CHAIN_CASES = [
    # B, H, c_in, c_mid, c_out, k1, k2 : conv1 (stats epilogue) -> BN + Swish (apply_fin) -> conv2, whose data
    # gradient reduces the BN backward sums (conv_gemm_bnbwd) -> bwd_apply_fin.  One case per conv tile family.
    dict(B=3, H=4, ci=32, cm=64, co=32, k1=1, k2=3),          # 64x64 (2,2) tiles
    dict(B=32, H=4, ci=64, cm=256, co=256, k1=3, k2=3),       # 32x64 small-M tiles (K >= 512)
    dict(B=40, H=8, ci=64, cm=128, co=128, k1=3, k2=3),       # 64x64 (2,4) 128-deep ring
    dict(B=16, H=16, ci=64, cm=384, co=64, k1=1, k2=1),       # 128x192 / 128x64 tiles
    dict(B=16, H=16, ci=128, cm=192, co=192, k1=5, k2=5),     # 16 halo tiles < 64: the generic kernel at a 5x5 geometry
    dict(B=64, H=32, ci=64, cm=192, co=64, k1=3, k2=3),       # 256x192 tiles
    # whole-image 3x3 kernel k_conv_img for both convs (16-bit): statistics epilogue, operand prologue (+ the activated
    # tensor written for the weight gradient), BatchNorm-backward epilogue; with prologue+fin the in-kernel finalize
    # sends both convs back to the generic kernel on a slab sized for the 128-row tiles
    dict(B=32, H=4, ci=128, cm=256, co=128, k1=3, k2=3),
    dict(B=9, H=8, ci=128, cm=128, co=256, k1=3, k2=3),
    # halo-tile kernel k_conv_halo (>= 64 tiles of 16x16 pixels x 192 channels): forward statistics epilogue, operand
    # prologue next to the ping-pong wave schedule (16-bit), BatchNorm-backward dgrad epilogue; 5x5 and 3x3
    dict(B=64, H=16, ci=128, cm=192, co=192, k1=5, k2=5, halo=True),
    dict(B=64, H=16, ci=128, cm=192, co=192, k1=3, k2=3, halo=True),
    # short-K 1x1 layers on many pixels: the two-workgroups-per-CU family (14) with the operand prologue (first case: the
    # second conv's forward) and with the BatchNorm-backward epilogue (second case: its data gradient)
    dict(B=16, H=32, ci=32, cm=128, co=512, k1=1, k2=1, fam14="fwd"),
    dict(B=16, H=32, ci=32, cm=512, co=128, k1=1, k2=1, fam14="dgrad"),
    # the same 5x5 chain on the four-wave software-pipelined form of the halo kernel (nvae_conv_halo4_enable)
    dict(B=64, H=16, ci=128, cm=192, co=192, k1=5, k2=5, halo=True, halo4=True),
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", FUSED_CHAIN_CASES,
                         ids=lambda c: "B{B}_H{H}_{ci}-{cm}-{co}_k{k1}{k2}".format(**c) + ("_4wave" if c.get("halo4") else ""))
@pytest.mark.parametrize("mode", ["prologue+fin", "prologue", "materialised"])
def test_fused_bn_chain(lib, dev, dtype, case, mode, monkeypatch):
    """conv -> BN(+Swish) -> conv against torch autograd (fp64) at every conv tile configuration.
    prologue+fin: the first conv's epilogue emits the statistics and its last workgroups finalize the BatchNorm
    (nvae_conv_gemm_ex fin), the second conv normalises + activates its operand in LDS and writes the activated
    tensor for its own weight gradient (nvae_conv_gemm_ex pre): no BatchNorm launch at all in the forward pass.
    prologue: statistics slab only, one nvae_bn_finalize_s launch.  materialised: round-1 sequence
    (finalize-in-apply pass).  Backward: BatchNorm sums in the second conv's data-gradient epilogue."""
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    monkeypatch.setattr(ops, "CONV_PRE", "all")        # (the product enables the prologue only where it measured faster)
    monkeypatch.setattr(ops, "STATS_FIN", True)
    lib.nvae_conv_halo4_enable(1 if case.get("halo4") else 0)
    B, H, ci, cm, co, k1, k2 = (case[n] for n in ("B", "H", "ci", "cm", "co", "k1", "k2"))
    if case.get("fam14"):
        import ctypes as C_
        from nvae_tf_amd import _lib as L_
        # forward GEMM of the second conv (K = cm, N = co) resp. its data gradient (K = co, N = cm)
        kk, nn = (cm, co) if case["fam14"] == "fwd" else (co, cm)
        gg = L_.ConvGeom(B, H, H, kk, H, H, nn, 1, 1, 1, 0, 0, 1, 0, kk, nn, nn)
        assert lib.nvae_conv_gemm_family(L_.dtype_code(dtype), C_.byref(gg)) == 14
    g = torch.Generator().manual_seed(77)
    ps = ParamStore(seed=5)
    c1 = ps.conv("c1", k1, ci, cm, bias=False)
    bn = ps.bn("bn", cm)
    c2 = ps.conv("c2", k2, cm, co)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    ps.get("bn.gamma").copy_(torch.rand(cm, generator=g) + 0.5)
    ps.get("bn.beta").copy_(torch.randn(cm, generator=g) * 0.3)
    x = torch.randn(B, H, H, ci, generator=g)
    dy = torch.randn(B, H, H, co, generator=g)
    # ---- reference
    x64 = q(x, dtype).requires_grad_(True)
    w1 = q(ps.get("c1.w").cpu(), dtype).requires_grad_(True)
    w2 = q(ps.get("c2.w").cpu(), dtype).requires_grad_(True)
    gam = ps.get("bn.gamma").cpu().double().requires_grad_(True)
    bet = ps.get("bn.beta").cpu().double().requires_grad_(True)
    p1, p2 = (k1 - 1) // 2, (k2 - 1) // 2
    h = ref_conv(x64, w1, None, 1, (p1, p1), 1, (H, H))
    hq = q(h.detach().float(), dtype) + (h - h.detach())          # the kernel stores h in `dtype`
    mean, var = h.mean((0, 1, 2)), h.var((0, 1, 2), unbiased=False)   # statistics come from the f32 accumulators
    pre = (hq - mean) * torch.rsqrt(var + 1e-5) * gam + bet
    a = pre * torch.sigmoid(pre)
    aq = q(a.detach().float(), dtype) + (a - a.detach())
    y_ref = ref_conv(aq, w2, ps.get("c2.b").cpu().double(), 1, (p2, p2), 1, (H, H))
    gr = torch.autograd.grad(y_ref, [x64, w1, w2, gam, bet], q(dy, dtype))
    # ---- kernels
    ctx = make_ctx(ps, dtype)
    xv = Var(x.to(dev, dtype))
    hv = ops.conv2d(ctx, xv, c1, bias=False, want_stats=True, stats_bn=bn if mode == "prologue+fin" else None)
    assert hv.stats is not None and (hv.fin is not None) == (mode == "prologue+fin")
    if case.get("halo"):
        # the halo kernel really is the one selected: its M-tile is a 16x16 patch = 256 rows, 64 tiles per slab row
        assert hv.stats[1] == -(-(B * (H // 16) ** 2) // 64)
        import ctypes as C_
        from nvae_tf_amd import _lib as L_
        for (cin_, cout_, kk) in ((ci, cm, k1), (cm, co, k2)):
            gg = L_.ConvGeom(B, H, H, cin_, H, H, cout_, kk, kk, 1, (kk - 1) // 2, (kk - 1) // 2, 1, 0, cin_, cout_, cout_)
            assert lib.nvae_conv_gemm_pre_max_cin(L_.dtype_code(dtype), C_.byref(gg)) == 512      # PRE_MAXC_HALO
    av = ops.bn_act(ctx, hv, bn, 1, lazy=mode != "materialised")
    assert (av.pre.mat is None) == (mode != "materialised")
    y = ops.conv2d(ctx, av, c2)
    assert (av.pre.mat is None) == (mode != "materialised")     # the prologue path never materialised it
    y.g = dy.to(dev, dtype)
    assert av.bn_src is not None and av.uses == 1
    ctx.backward()
    torch.cuda.synchronize()
    assert av.bn_src["fused"]                       # the data gradient of c2 carried the BatchNorm sums
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < 2 * tol
    assert rel_err(xv.g, gr[0]) < 4 * tol
    assert rel_err(ps.get_grad("c1.w"), gr[1]) < 4 * tol and rel_err(ps.get_grad("c2.w"), gr[2]) < 2 * tol
    assert rel_err(ps.get_grad("bn.gamma"), gr[3]) < 4 * tol and rel_err(ps.get_grad("bn.beta"), gr[4]) < 4 * tol
    assert rel_err(ps.get_state("bn.rm"), 0.95 * mean) < 1e-3


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape", [(6, 4, 4, 256), (3, 8, 8, 72), (2, 16, 16, 64), (130, 4, 4, 128), (2, 32, 32, 32)])
@pytest.mark.parametrize("lazy", [False, True], ids=["materialised", "lazy"])
@pytest.mark.parametrize("se_path", ["fused", "strips", "split4"])
def test_fused_bn_se_chain(lib, dev, dtype, shape, lazy, se_path, monkeypatch, request):
    """BN -> SE + residual -> BN: the SE kernel emits the next BatchNorm's statistics, its backward apply
    reduces the previous BatchNorm's backward sums; both BatchNorms use the finalize-in-apply passes.
    lazy: the first BatchNorm is applied inside the fused SE kernels (forward and backward) from its
    coefficient table and its output is never materialised (nor rounded to the activation dtype).
    (130 images: more than the 128 rows of the fused kernels' statistics slab -> two images per workgroup;
    72 channels: not a power of two -> the unfused launch sequence.)  se_path: the one-launch kernels or the
    strip-structured three-kernel path (the default below ~100 images), both at every shape."""
    from nvae_tf_amd import ops
    monkeypatch.setattr(ops, "SE_FUSED_MIN_B", 10 ** 9 if se_path == "strips" else 0)
    lib.nvae_se_force_split(4 if se_path == "split4" else 1)      # (image-split form of the fused kernels where legal)
    request.addfinalizer(lambda: lib.nvae_se_force_split(-1))
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    g = torch.Generator().manual_seed(31)
    C_ = shape[3]
    ps = ParamStore(seed=6)
    bn1, se, bn2 = ps.bn("bn1", C_), ps.se("se", C_), ps.bn("bn2", C_)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    for n in ("bn1", "bn2"):
        ps.get(n + ".gamma").copy_(torch.rand(C_, generator=g) + 0.5)
        ps.get(n + ".beta").copy_(torch.randn(C_, generator=g) * 0.3)
    x, skip, dy = (torch.randn(shape, generator=g) for _ in range(3))
    x64, s64 = q(x, dtype).requires_grad_(True), q(skip, dtype).requires_grad_(True)
    P = {n: ps.get(n).cpu().double().requires_grad_(True) for n in
         ("bn1.gamma", "bn1.beta", "bn2.gamma", "bn2.beta", "se.w1", "se.b1", "se.w2", "se.b2")}

    def bn_ref(t, ga, be):
        m, v = t.mean((0, 1, 2)), t.var((0, 1, 2), unbiased=False)
        return (t - m) * torch.rsqrt(v + 1e-5) * ga + be
    a = bn_ref(x64, P["bn1.gamma"], P["bn1.beta"])
    really_lazy = lazy and C_ & (C_ - 1) == 0 and se_path != "strips"
    aq = a if really_lazy else q(a.detach().float(), dtype) + (a - a.detach())
    gate = torch.sigmoid(torch.relu(aq.mean((1, 2)) @ P["se.w1"] + P["se.b1"]) @ P["se.w2"] + P["se.b2"])
    r = 0.1 * s64 + aq * gate[:, None, None, :]
    m2, v2 = r.mean((0, 1, 2)), r.var((0, 1, 2), unbiased=False)      # from the f32 values, before rounding
    rq = q(r.detach().float(), dtype) + (r - r.detach())
    pre = (rq - m2) * torch.rsqrt(v2 + 1e-5) * P["bn2.gamma"] + P["bn2.beta"]
    y_ref = pre * torch.sigmoid(pre)
    names = list(P)
    gr = torch.autograd.grad(y_ref, [x64, s64] + [P[n] for n in names], q(dy, dtype))
    ctx = make_ctx(ps, dtype)
    xv, sv = Var(x.to(dev, dtype)), Var(skip.to(dev, dtype))
    av = ops.bn_act(ctx, xv, bn1, 0, lazy=lazy)
    rv = ops.se_residual(ctx, av, se, sv, 0.1, 1.0, stats_bn=bn2 if lazy else None)
    assert (av.pre.mat is None) == really_lazy
    assert rv.stats is not None
    y = ops.bn_act(ctx, rv, bn2, 1)
    y.g = dy.to(dev, dtype)
    ctx.backward()
    torch.cuda.synchronize()
    assert av.bn_src["fused"]
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < 2 * tol
    assert rel_err(xv.g, gr[0]) < 4 * tol and rel_err(sv.g, gr[1]) < 4 * tol
    for i, n in enumerate(names):
        # bn1's beta / gamma gradients are sums that BN2's mean subtraction nearly cancels: in bf16 their
        # rounding noise is O(30 %) of the (tiny) true value; the f32 run of the same kernels is exact
        lim = {torch.bfloat16: 0.6, torch.float16: 0.1}.get(dtype, 8 * tol) if n.startswith("bn1.") else 8 * tol
        assert rel_err(ps.get_grad(n), gr[2 + i]) < lim, n


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("shape,ss,bs", [((3, 8, 8, 128), 0.1, 1.0), ((4, 4, 4, 256), 1.0, 0.1), ((2, 32, 32, 32), 1.0, 0.1),
                                          ((3, 16, 16, 64), 1.0, 0.1)])      # (the last: 8 register-resident chunks per thread)
@pytest.mark.parametrize("se_path", ["fused", "strips", "split2", "split8"])
def test_se_residual(lib, dev, dtype, shape, ss, bs, se_path, monkeypatch, request):
    from nvae_tf_amd import ops
    # the one-launch kernels (whole images per workgroup), the strip-structured three-kernel path, and the image-split
    # form of the one-launch kernels (S workgroups per image with an in-kernel hand-off; where S is not legal for a
    # shape the launcher falls back to whole images): all at every shape here
    monkeypatch.setattr(ops, "SE_FUSED_MIN_B", 10 ** 9 if se_path == "strips" else 0)
    lib.nvae_se_force_split(int(se_path[5:]) if se_path.startswith("split") else 1)
    request.addfinalizer(lambda: lib.nvae_se_force_split(-1))
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    g = torch.Generator().manual_seed(13)
    C_ = shape[3]
    ps = ParamStore(seed=2)
    se = ps.se("se", C_)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    ps.get("se.b1").copy_(torch.randn(se.hidden, generator=g) * 0.1 + 0.1)
    ps.get("se.b2").copy_(torch.randn(C_, generator=g) * 0.1)
    x, skip, dy = (torch.randn(shape, generator=g) for _ in range(3))
    x64, s64 = q(x, dtype).requires_grad_(True), q(skip, dtype).requires_grad_(True)
    P = {n: ps.get("se." + n).cpu().double().requires_grad_(True) for n in ("w1", "b1", "w2", "b2")}
    p = x64.mean((1, 2))
    h = torch.relu(p @ P["w1"] + P["b1"])
    gate = torch.sigmoid(h @ P["w2"] + P["b2"])
    y_ref = ss * s64 + bs * x64 * gate[:, None, None, :]
    gr = torch.autograd.grad(y_ref, [x64, s64, P["w1"], P["b1"], P["w2"], P["b2"]], q(dy, dtype))
    ctx = make_ctx(ps, dtype)
    xv, sv = Var(x.to(dev, dtype)), Var(skip.to(dev, dtype))
    y = ops.se_residual(ctx, xv, se, sv, ss, bs)
    y.g = dy.to(dev, dtype)
    ctx.backward()
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < tol
    assert rel_err(xv.g, gr[0]) < tol and rel_err(sv.g, gr[1]) < tol
    for i, n in enumerate(("w1", "b1", "w2", "b2")):
        assert rel_err(ps.get_grad("se." + n), gr[2 + i]) < 4 * tol, n
    if se_path.startswith("split"):
        from nvae_tf_amd import _lib as L_
        torch.cuda.synchronize()
        assert int(L_._workspace["se"][1].abs().sum()) == 0        # per-image hand-off counters back at rest
        # the slice-ordered sums make the forward result independent of which slice arrives last
        y2 = ops.se_residual(make_ctx(ps, dtype, record=False), Var(x.to(dev, dtype)), se, Var(skip.to(dev, dtype)), ss, bs)
        assert torch.equal(y.t, y2.t)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("B,H,W_,C_", [(3, 8, 8, 96), (5, 4, 4, 1536), (2, 16, 16, 192), (2, 6, 10, 72), (3, 3, 2, 8),
                                       (40, 32, 32, 64)],
                         ids=["8x8", "4x4", "16x16", "ragged", "tiny", "many-tiles"])
def test_dwconv5(lib, dev, dtype, B, H, W_, C_):
    """Depthwise 5x5 forward / data gradient / weight + bias gradient at the tower shapes (4x4, 8x8),
    multi-tile images, tiles and channel strips that are only partly inside the tensor."""
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    g = torch.Generator().manual_seed(17)
    ps = ParamStore(seed=2)
    dw = ps.dw("dw", C_)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    ps.get("dw.b").copy_(torch.randn(C_, generator=g))
    x, dy = torch.randn(B, H, W_, C_, generator=g), torch.randn(B, H, W_, C_, generator=g)
    x64 = q(x, dtype).requires_grad_(True)
    w64 = ps.get("dw.w").cpu().double().requires_grad_(True)
    b64 = ps.get("dw.b").cpu().double().requires_grad_(True)
    xn = F.pad(x64.permute(0, 3, 1, 2), (2, 2, 2, 2))
    y_ref = F.conv2d(xn, w64.permute(2, 0, 1).unsqueeze(1).contiguous(), b64, groups=C_).permute(0, 2, 3, 1)
    gr = torch.autograd.grad(y_ref, [x64, w64, b64], q(dy, dtype))
    ctx = make_ctx(ps, dtype)
    xv = Var(x.to(dev, dtype))
    y = ops.dwconv5(ctx, xv, dw)
    y.g = dy.to(dev, dtype)
    ctx.backward()
    tol = TOL[dtype]
    assert rel_err(y.t, y_ref) < tol and rel_err(xv.g, gr[0]) < tol
    assert rel_err(ps.get_grad("dw.w"), gr[1]) < tol and rel_err(ps.get_grad("dw.b"), gr[2]) < tol
    from nvae_tf_amd._lib import call, ptr
    # forward with fused BatchNorm statistics (bf16 ring kernel): column sums / sums of squares of the output
    rows = lib.nvae_dwconv5_stats_rows(ctx.dt, B, H, W_, C_)
    assert (rows > 0) == (dtype != torch.float32)
    if rows:
        slab = torch.zeros((rows, 2, C_), device=dev)          # accumulated into with atomics: must be zero
        y2 = torch.empty_like(y.t)
        call("nvae_dwconv5_stats", ctx.dt, ptr(xv.t), ptr(ps.view(dw.w)), ptr(ps.view(dw.b)), ptr(y2), B, H, W_, C_,
             ptr(slab))
        assert torch.equal(y2, y.t)
        yr = y_ref.detach().reshape(-1, C_)
        assert rel_err(slab[:, 0].sum(0), yr.sum(0)) < 1e-4 and rel_err(slab[:, 1].sum(0), (yr * yr).sum(0)) < 1e-4
    # accumulate into an existing data gradient (flip = 1, accumulate = 1)
    acc = xv.g.clone()
    call("nvae_dwconv5", ctx.dt, ptr(y.g), ptr(ps.view(dw.w)), None, ptr(acc), B, H, W_, C_, 1, 1)
    assert rel_err(acc, 2 * gr[0]) < 2 * tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("bnbwd", [False, True], ids=["bn-bwd-unfused", "bn-bwd-in-dgrad"])
@pytest.mark.parametrize("B,H,C_,act", [(5, 4, 1536, 1), (3, 8, 96, 1), (2, 16, 192, 0), (40, 32, 64, 1)],
                         ids=["4x4", "8x8", "16x16", "many-tiles"])
def test_dwconv5_bn_prologue(lib, dev, monkeypatch, dtype, B, H, C_, act, bnbwd):
    """BatchNorm(+Swish) -> depthwise 5x5 (decoder.py:125-131) with the BatchNorm applied inside the depthwise kernels
    (nvae_dwconv5_pre / nvae_dwconv5_wgrad_pre: never materialised) against the same chain with a materialised
    BatchNorm output: same rounding points, so outputs, statistics, every gradient and the published coefficient table /
    moving statistics agree to f32 summation-order noise; and against the fp64 chain within the dtype's tolerance.
    bn-bwd-in-dgrad (round 3, nvae_dwconv5_bnbwd): the depthwise data-gradient kernel also reduces the BatchNorm's backward
    sums - from its f32 accumulators, where the unfused reduce reads the 16-bit-rounded gradient - so the two arms are
    compared with the fp64 chain (dx, dgamma, dbeta) instead of with each other."""
    from nvae_tf_amd import ops
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    monkeypatch.setattr(ops, "DW_BNBWD", bnbwd)
    g = torch.Generator().manual_seed(23)
    x = torch.randn(B, H, H, C_, generator=g) * 1.5 + 0.3
    dy = torch.randn(B, H, H, C_, generator=g)
    res = {}
    for mode in ("pre", "materialised"):
        monkeypatch.setattr(ops, "DW_PRE", mode == "pre")
        ps = ParamStore(seed=2)
        bn, dw = ps.bn("bn", C_), ps.dw("dw", C_)
        ps.finalize(dev, dtype, zero_pool_floats=1 << 18)
        gg = torch.Generator().manual_seed(5)
        ps.get("bn.gamma").copy_(torch.rand(C_, generator=gg) + 0.5); ps.get("bn.beta").copy_(torch.randn(C_, generator=gg) * 0.3)
        ps.get("dw.b").copy_(torch.randn(C_, generator=gg))
        ctx = make_ctx(ps, dtype)
        xv = Var(x.to(dev, dtype))
        h = ops.bn_act(ctx, xv, bn, act)
        y = ops.dwconv5(ctx, h, dw, want_stats=True)
        assert (h.pre.mat is None) == (mode == "pre")
        y.g = dy.to(dev, dtype)
        ctx.backward()
        torch.cuda.synchronize()
        res[mode] = dict(y=y.t.float(), stats=y.stats[0].float().sum(0), dx=xv.g.float(), dw=ps.get_grad("dw.w").clone(),
                         db=ps.get_grad("dw.b").clone(), dgamma=ps.get_grad("bn.gamma").clone(),
                         dbeta=ps.get_grad("bn.beta").clone(), rm=ps.get_state("bn.rm").clone(), rv=ps.get_state("bn.rv").clone())
        if mode == "pre":
            w64, b64 = ps.get("dw.w").cpu().double(), ps.get("dw.b").cpu().double()
            g64, be64 = ps.get("bn.gamma").cpu().double(), ps.get("bn.beta").cpu().double()
    a, b = res["pre"], res["materialised"]
    for k in a:
        if bnbwd and k in ("dx", "dgamma", "dbeta"):
            continue                 # (the lazy arm fuses the sums, the materialised arm cannot: compared with fp64 below)
        assert rel_err(a[k], b[k]) < 1e-5, k
    # fp64 chain
    x64 = q(x, dtype).requires_grad_(True)
    g64 = g64.requires_grad_(True); be64 = be64.requires_grad_(True)
    m = x64.mean((0, 1, 2)); v = x64.var((0, 1, 2), unbiased=False)
    z = (x64 - m) / torch.sqrt(v + 1e-5) * g64 + be64
    hz = z * torch.sigmoid(z) if act else z
    yr = F.conv2d(F.pad(hz.permute(0, 3, 1, 2), (2, 2, 2, 2)), w64.permute(2, 0, 1).unsqueeze(1).contiguous(), b64,
                  groups=C_).permute(0, 2, 3, 1)
    gx, gg_, gb_ = torch.autograd.grad(yr, [x64, g64, be64], q(dy, dtype))
    tol = TOL[dtype]
    assert rel_err(a["y"], yr) < 2 * tol and rel_err(a["dx"], gx) < 4 * tol
    for arm in (a, b):
        assert rel_err(arm["dx"], gx) < 4 * tol and rel_err(arm["dgamma"], gg_) < 4 * tol and rel_err(arm["dbeta"], gb_) < 4 * tol


def _softclamp5(x):
    return 5.0 * torch.tanh(x / 5.0)


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("group0", [True, False])
def test_sampler_kl(lib, dev, dtype, group0):
    from nvae_tf_amd import ops, _lib as L
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    g = torch.Generator().manual_seed(19)
    B, H, Lc = 5, 4, 20
    ps = ParamStore(seed=2)
    ps.bn("dummy", 8)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    enc = torch.randn(B, H, H, 2 * Lc, generator=g) * 2
    dec = torch.randn(B, H, H, 2 * Lc, generator=g) * 2
    eps = torch.randn(B, H, H, Lc, generator=g)
    dz = torch.randn(B, H, H, Lc, generator=g)
    beta, coeff, inv_b = 0.7, 1.3, 1.0 / B
    e64 = enc.double().requires_grad_(True)
    d64 = dec.double().requires_grad_(True)
    a, b_ = e64[..., :Lc], e64[..., Lc:]
    if group0:
        mq, sq = _softclamp5(a), torch.exp(_softclamp5(b_)) + 1e-2
        mp, sp = torch.zeros_like(mq), torch.ones_like(sq)
    else:
        m, s = d64[..., :Lc], d64[..., Lc:]
        mp, sp = _softclamp5(m), torch.exp(_softclamp5(s)) + 1e-2
        mq, sq = _softclamp5(a + m), torch.exp(_softclamp5(b_ + s)) + 1e-2
    z_ref = mq + eps.double() * sq
    t1, t2 = (mq - mp) / sp, sq / sp
    kl_ref = (0.5 * (t1 * t1 + t2 * t2) - 0.5 - torch.log(t2)).sum((1, 2, 3))
    lq_ref = (-0.5 * ((z_ref - mq) / sq) ** 2 - 0.5 * math.log(2 * math.pi) - torch.log(sq)).sum((1, 2, 3))
    lp_ref = (-0.5 * ((z_ref - mp) / sp) ** 2 - 0.5 * math.log(2 * math.pi) - torch.log(sp)).sum((1, 2, 3))
    total = (z_ref * q(dz, dtype)).sum() + beta * coeff * inv_b * kl_ref.sum()
    gr = torch.autograd.grad(total, [e64] + ([] if group0 else [d64]))
    ctx = make_ctx(ps, dtype)
    hyper = torch.zeros(L.HY_SIZE, device=dev)
    hyper[L.HY_BETA] = beta
    cf = torch.full((1,), coeff, device=dev)
    kl = torch.zeros(B, device=dev)
    lq, lp = torch.zeros(B, device=dev), torch.zeros(B, device=dev)
    ms = torch.zeros(4, B, H, H, Lc, device=dev)
    ev = Var(enc.to(dev))
    dv = None if group0 else Var(dec.to(dev))
    z = ops.sampler(ctx, ev, dv, eps.to(dev), kl, cf, hyper, inv_b, lq, lp, ms)
    z.g = dz.to(dev, dtype)
    ctx.backward()
    tol = TOL[dtype]
    assert rel_err(z.t, z_ref) < tol
    assert rel_err(kl, kl_ref) < 1e-4 and rel_err(lq, lq_ref) < 1e-4 and rel_err(lp, lp_ref) < 1e-4
    assert rel_err(ms[0], mq) < 1e-5 and rel_err(ms[1], sq) < 1e-5 and rel_err(ms[3], sp) < 1e-5
    assert rel_err(ev.g, gr[0]) < tol
    if not group0:
        assert rel_err(dv.g, gr[1]) < tol


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16", "f16"])
def test_bernoulli_and_loss(lib, dev, dtype):
    from nvae_tf_amd import ops, _lib as L
    from nvae_tf_amd.ops import Var
    from nvae_tf_amd.params import ParamStore
    g = torch.Generator().manual_seed(23)
    B = 6
    ps = ParamStore(seed=2)
    ps.bn("dummy", 8)
    ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
    logits = torch.randn(B, 32, 32, 1, generator=g) * 3
    x = (torch.rand(B, 32, 32, 1, generator=g) < 0.2).float()
    l64 = logits.double().requires_grad_(True)
    rec_ref = (F.softplus(l64) - x.double() * l64).sum((1, 2, 3))
    gr = torch.autograd.grad(rec_ref.mean(), l64)[0]
    ctx = make_ctx(ps, dtype)
    lv = Var(logits.to(dev))
    rec = torch.zeros(B, device=dev)
    ops.bernoulli_nll(ctx, lv, x.to(dev, dtype), rec, 1.0 / B)
    ctx.backward()
    assert rel_err(rec, rec_ref) < 1e-5
    assert rel_err(lv.g, gr) < TOL[dtype]
    ctx = make_ctx(ps, dtype, record=False)
    ops.bernoulli_nll(ctx, lv, x.to(dev, dtype), rec, 1.0 / B, crop=True)
    crop_ref = (F.softplus(l64) - x.double() * l64)[:, 2:30, 2:30].sum((1, 2, 3))
    assert rel_err(rec, crop_ref) < 1e-5
    # KL balancing (models.py:204-218) through nvae_kl_absmean + nvae_loss_finalize
    G = 7
    kl_all = torch.rand(G, B, generator=g) * 10
    alphas = torch.tensor([1., 1, 1, 1, 8, 8, 8])
    beta = 0.35
    c = kl_all.double().abs().mean(1) + 0.01
    c = c / alphas.double() * c.sum()
    c = c / c.mean()
    kl_ref = beta * (kl_all.double() * c[:, None]).sum(0)
    am = torch.zeros(G, device=dev); coeff = torch.zeros(G, device=dev)
    klb = torch.zeros(B, device=dev); res = torch.zeros(L.RES_SIZE, device=dev)
    hyper = torch.zeros(L.HY_SIZE, device=dev); hyper[L.HY_BETA] = beta; hyper[L.HY_BALANCE] = 1
    bn = torch.full((1,), 0.88, device=dev)
    L.call("nvae_kl_absmean", L.ptr(kl_all.to(dev)), G, B, L.ptr(am))
    kd = kl_all.to(dev)
    L.call("nvae_loss_finalize", L.ptr(kd), L.ptr(am), L.ptr(alphas.to(dev)), G, B, L.ptr(rec), L.ptr(bn),
           L.ptr(hyper), L.ptr(coeff), L.ptr(klb), L.ptr(res))
    assert rel_err(coeff, c) < 1e-5 and rel_err(klb, kl_ref) < 1e-5
    want = float((crop_ref + kl_ref).mean()) + 0.88
    assert abs(float(res[L.RES_LOSS]) - want) / abs(want) < 1e-5


def test_spectral_norm_and_weight_prep(lib, dev):
    from nvae_tf_amd.params import ParamStore
    for dtype in DTYPES:
        ps = ParamStore(seed=4)
        convs = [ps.conv("a", 3, 16, 24), ps.conv("b", 1, 276, 64), ps.conv("c", 5, 8, 8, bias=False),
                 ps.conv("d", 3, 1, 16)]
        ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
        ref = {}
        for c in convs:
            w = ps.get(c.name + ".w").cpu().double()
            u = ps.get_state(c.name + ".u").cpu().double().reshape(1, -1)
            w2 = w.reshape(-1, c.cout)
            v = u @ w2.t(); v = v / v.norm()
            un = v @ w2; un = un / un.norm()
            sigma = (v @ w2 @ un.t()).item()
            ref[c.name] = (w / sigma, un.reshape(-1))
        ps.begin_step()
        ps.prepare_weights(spectral_norm=True)
        torch.cuda.synchronize()
        for c in convs:
            wn, un = ref[c.name]
            assert rel_err(ps.get(c.name + ".w"), wn) < 1e-5, c.name
            assert rel_err(ps.get_state(c.name + ".u"), un) < 1e-5, c.name
            K = c.k * c.k * c.cin
            wf = ps.wcopies[c.wf_off:c.wf_off + c.cout * c.wf_ld].reshape(c.cout, c.wf_ld)[:, :K]
            assert rel_err(wf, wn.reshape(K, c.cout).t()) < TOL[dtype], c.name
            if c.wd_off >= 0:
                wd = ps.wcopies[c.wd_off:c.wd_off + c.cin * c.wd_ld].reshape(c.cin, c.k, c.k, c.cout)
                want = wn.flip(0, 1).permute(2, 0, 1, 3)
                assert rel_err(wd, want) < TOL[dtype], c.name


def test_dynamic_loss_scale_kernels(lib, dev):
    """nvae_grad_guard / nvae_adamax / nvae_loss_scale_update (the f16 path's device-side dynamic loss scaling): a clean
    step is taken on gradient / scale and counted; a step with a non-finite gradient element is skipped and halves the
    scale; NVAE_LS_GROWTH_STEPS clean steps double it; the seed kernels multiply by the scale."""
    from nvae_tf_amd import _lib as L
    g = torch.Generator().manual_seed(31)
    n = 2048
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g) * 4.0          # gradient of the SCALED loss
    m, u = torch.zeros(n), torch.zeros(n)
    hyper = torch.zeros(L.HY_SIZE, device=dev)
    hyper[L.HY_LR], hyper[L.HY_LSCALE], hyper[L.HY_GSCALE] = 1e-2, 4.0, 0.25
    pd, gd, md, ud = (t.to(dev) for t in (p, gr, m, u))

    def step():
        L.call("nvae_grad_guard", L.ptr(gd), n, L.ptr(hyper))
        L.call("nvae_adamax", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(ud), n, L.ptr(hyper), 0.9, 0.999, 1e-7)
        L.call("nvae_loss_scale_update", L.ptr(hyper), 2.0 ** -24, 2.0 ** 16)
    step()
    g1 = gr.double() / 4.0
    m_ref = 0.1 * g1; u_ref = g1.abs(); p_ref = p.double() - 1e-2 * m_ref / (u_ref + 1e-7)
    assert rel_err(pd, p_ref) < 1e-6 and rel_err(md, m_ref) < 1e-6 and rel_err(ud, u_ref) < 1e-6
    assert hyper.tolist()[3:7] == [0.25, 4.0, 1.0, 0.0]
    # overflow: nothing moves, the scale halves, the flag is cleared for the next step
    before = (pd.clone(), md.clone(), ud.clone())
    gd[777] = float("inf")
    step()
    assert torch.equal(pd, before[0]) and torch.equal(md, before[1]) and torch.equal(ud, before[2])
    assert hyper.tolist()[3:7] == [0.5, 2.0, 0.0, 0.0]
    gd[777] = float("nan")
    step()
    assert torch.equal(pd, before[0]) and hyper.tolist()[3:7] == [1.0, 1.0, 0.0, 0.0]
    # growth after 200 clean steps
    gd[777] = 0.5
    for _ in range(200):
        step()
    assert hyper.tolist()[3:7] == [0.5, 2.0, 0.0, 0.0] and not torch.equal(pd, before[0])
    # the backward seeds carry the scale
    logits = torch.randn(2, 4, 4, 8, generator=g).to(dev)
    x = (torch.rand(2, 4, 4, 8, generator=g) < 0.3).float().to(dev)
    d1, d2 = torch.empty_like(logits), torch.empty_like(logits)
    L.call("nvae_bernoulli_bwd", L.F32, L.ptr(logits), L.ptr(x), L.ptr(d1), logits.numel(), 0.5, None)
    L.call("nvae_bernoulli_bwd", L.F32, L.ptr(logits), L.ptr(x), L.ptr(d2), logits.numel(), 0.5, L.ptr(hyper))
    assert rel_err(d2, d1.double() * 2.0) < 1e-6


def test_adamax_unary_randn(lib, dev):
    from nvae_tf_amd import _lib as L
    g = torch.Generator().manual_seed(29)
    n = 4096 + 8
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, u = torch.randn(n, generator=g) * 0.1, torch.rand(n, generator=g)
    lr_t = 3e-3
    m_ref = 0.9 * m.double() + 0.1 * gr.double()
    u_ref = torch.maximum(0.999 * u.double(), gr.double().abs())
    p_ref = p.double() - lr_t * m_ref / (u_ref + 1e-7)
    hyper = torch.zeros(L.HY_SIZE, device=dev); hyper[L.HY_LR] = lr_t
    pd, gd, md, ud = (t.to(dev) for t in (p, gr, m, u))
    L.call("nvae_adamax", L.ptr(pd), L.ptr(gd), L.ptr(md), L.ptr(ud), n, L.ptr(hyper), 0.9, 0.999, 1e-7)
    assert rel_err(pd, p_ref) < 1e-6 and rel_err(md, m_ref) < 1e-6 and rel_err(ud, u_ref) < 1e-6
    for dtype in DTYPES:
        x = torch.randn(1024, generator=g) * 2
        xd = x.to(dev, dtype); y = torch.empty_like(xd); dx = torch.empty_like(xd)
        dy = torch.randn(1024, generator=g).to(dev, dtype)
        for op, f in ((L.OP_SWISH, lambda t: t * torch.sigmoid(t)), (L.OP_ELU, F.elu)):
            x64 = xd.double().cpu().requires_grad_(True)
            L.call("nvae_unary_fwd", L.dtype_code(dtype), op, L.ptr(xd), L.ptr(y), 1024, 0.0, 0.0)
            L.call("nvae_unary_bwd", L.dtype_code(dtype), op, L.ptr(xd), L.ptr(dy), L.ptr(dx), 1024, 0)
            ref = f(x64)
            assert rel_err(y, ref) < TOL[dtype]
            assert rel_err(dx, torch.autograd.grad(ref, x64, dy.double().cpu())[0]) < TOL[dtype]
    out = torch.empty(1 << 20, device=dev)
    ctr = torch.zeros(1, dtype=torch.int64, device=dev)
    L.call("nvae_randn", L.ptr(out), out.numel(), 1234, L.ptr(ctr))
    assert abs(float(out.mean())) < 5e-3 and abs(float(out.std()) - 1) < 5e-3
    assert int(ctr[0]) == (1 << 20) // 4
    out2 = torch.empty(1 << 20, device=dev)
    L.call("nvae_randn", L.ptr(out2), out2.numel(), 1234, L.ptr(ctr))
    assert float((out - out2).abs().max()) > 1.0   # the counter advanced: fresh noise


def test_bad_arguments_fail_loudly(lib, dev):
    from nvae_tf_amd import _lib as L
    x = torch.zeros(16, device=dev)
    with pytest.raises(RuntimeError, match="multiple of 8"):
        L.call("nvae_unary_fwd", L.F32, L.OP_ELU, L.ptr(x), L.ptr(x), 12, 0.0, 0.0)
    g = L.ConvGeom(1, 4, 4, 3, 4, 4, 8, 3, 3, 1, 1, 1, 1, 0, 3, 8, 8)
    with pytest.raises(RuntimeError, match="conv_direct"):
        L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(x), 32, None, None, L.ptr(x), 0, None)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_grad_range_normalisation_kernels(lib, dev, dtype):
    """nvae_grad_amax / _rescale / _merge / _unscale (float16 on deep hierarchies): powers of two only, so every step is
    exact and checked with torch.equal against the same arithmetic in torch."""
    import math
    from nvae_tf_amd._lib import call, ptr
    n = 8192
    g = torch.Generator().manual_seed(3)
    a = (torch.randn(n, generator=g) * 3e-3).to(dev, dtype)
    b = (torch.randn(n, generator=g) * 40.0).to(dev, dtype)
    scales, amax = torch.zeros(8, device=dev), torch.zeros(8, device=dev)
    dt = {torch.bfloat16: 1, torch.float16: 2}[dtype]
    # a: tag 0 -> 1, b: tag 0 -> 2, both renormalised so that max |.| lands in (2^5, 2^6]
    for t, idx in ((a, 1), (b, 2)):
        ref, m = t.clone(), float(t.float().abs().max())
        call("nvae_grad_amax", dt, ptr(t), n, ptr(amax) + 4 * idx)
        assert float(amax[idx]) == m
        call("nvae_grad_rescale", dt, ptr(t), n, ptr(amax) + 4 * idx, ptr(scales), 0, idx, 6.0)
        k = math.floor(6.0 - math.log2(m))
        assert float(scales[idx]) == k and 2.0 ** 5 < float(t.float().abs().max()) <= 2.0 ** 6
        assert torch.equal(t, (ref.float() * 2.0 ** k).to(dtype))
    # merge on the smaller exponent: the result carries min(k_a, k_b) under a fresh tag and cannot overflow
    ka, kb = float(scales[1]), float(scales[2])
    want = (a.float() * 2.0 ** (min(ka, kb) - ka) + b.float() * 2.0 ** (min(ka, kb) - kb)).to(dtype)
    call("nvae_grad_merge", dt, ptr(a), ptr(b), n, ptr(scales), 1, 2, 3)
    assert float(scales[3]) == min(ka, kb) and torch.equal(a, want) and bool(torch.isfinite(a).all())
    # parameter gradients: ranges tagged 1 / 3 are divided by their factors, an untagged range is left alone
    grads = torch.randn(64, generator=g).to(dev)
    ref = grads.clone()
    table = torch.tensor([0, 4, 1, 0, 6, 2, 3, 0], dtype=torch.int32, device=dev)       # floats [0,16) tag 1, [24,32) tag 3
    call("nvae_grad_unscale", ptr(grads), ptr(table), 2, ptr(scales))
    ref[0:16] *= 2.0 ** -ka
    ref[24:32] *= 2.0 ** -min(ka, kb)
    assert torch.equal(grads, ref)
