"""Where does the f32 HIP path lose more bits than PyTorch-CPU f32?  Per op: error of the HIP f32 kernel and of the
PyTorch-CPU f32 op against the same fp64 reference, in units of 2^-24 of the output scale (rms and max)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nvae_tf_amd import _lib as L, ops
from nvae_tf_amd.ops import Ctx, Var
from nvae_tf_amd.params import ParamStore

dev = torch.device("cuda:0")
L.load()
ULP = 2.0 ** -24


def err(a, ref):
    d = (a.double().cpu() - ref).abs()
    s = float(ref.abs().max())
    return float(d.pow(2).mean().sqrt()) / s / ULP, float(d.max()) / s / ULP


def conv_case(B, H, cin, cout, k):
    ps = ParamStore(seed=2)
    conv = ps.conv("c", k, cin, cout)
    ps.finalize(dev, torch.float32, zero_pool_floats=1 << 16)
    ps.begin_step(); ps.prepare_weights(False)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, H, H, cin, generator=g)
    dy = torch.randn(B, H, H, cout, generator=g)
    w = ps.get("c.w").cpu(); b = ps.get("c.b").cpu()
    ps.grads.zero_()
    ctx = Ctx(ps, torch.float32, True, True)
    xv = Var(x.to(dev))
    y = ops.conv2d(ctx, xv, conv)
    y.g = dy.to(dev)
    ctx.backward()
    torch.cuda.synchronize()
    res = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        xt = x.to(dt).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
        wt = w.to(dt).permute(3, 2, 0, 1).contiguous().requires_grad_(True)
        yt = F.conv2d(xt, wt, b.to(dt), padding=k // 2)
        yt.backward(dy.to(dt).permute(0, 3, 1, 2))
        res[name] = (yt.detach().permute(0, 2, 3, 1), xt.grad.permute(0, 2, 3, 1), wt.grad.permute(2, 3, 1, 0))
    r = res["f64"]
    hip = (y.t, xv.g, ps.get_grad("c.w"))
    for i, what in enumerate(("fwd", "dgrad", "wgrad")):
        eh, et = err(hip[i], r[i].double()), err(res["f32"][i], r[i].double())
        print(f"conv B{B} {H}x{H} {k}x{k} {cin}->{cout} {what:5s}: HIP rms {eh[0]:6.2f} max {eh[1]:7.1f} | torch-f32 rms {et[0]:6.2f} max {et[1]:7.1f}  [2^-24 of scale]", flush=True)


def bn_case(B, H, C_, act, mean):
    ps = ParamStore(seed=3)
    bn = ps.bn("bn", C_)
    ps.finalize(dev, torch.float32, zero_pool_floats=1 << 16)
    ps.begin_step(); ps.prepare_weights(False)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, H, H, C_, generator=g) + mean
    dy = torch.randn(B, H, H, C_, generator=g)
    ps.grads.zero_()
    ctx = Ctx(ps, torch.float32, True, True)
    xv = Var(x.to(dev))
    y = ops.bn_act(ctx, xv, bn, act)
    yt_ = y.t
    y.g = dy.to(dev)
    ctx.backward()
    torch.cuda.synchronize()
    res = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        xt = x.to(dt).requires_grad_(True)
        m = xt.mean((0, 1, 2)); v = xt.var((0, 1, 2), unbiased=False)
        z = (xt - m) / torch.sqrt(v + 1e-5)
        yt = z * torch.sigmoid(z) if act else z
        yt.backward(dy.to(dt))
        res[name] = (yt.detach(), xt.grad)
    for i, what in enumerate(("fwd", "bwd")):
        eh, et = err((yt_, xv.g)[i], res["f64"][i]), err(res["f32"][i], res["f64"][i])
        print(f"bn{'+swish' if act else '      '} B{B} {H}x{H}x{C_} mean {mean}: {what}: HIP rms {eh[0]:6.2f} max {eh[1]:7.1f} | torch-f32 rms {et[0]:6.2f} max {et[1]:7.1f}", flush=True)


for shp in ((8, 4, 256, 256, 3), (8, 4, 1536, 256, 1), (8, 16, 384, 384, 5), (8, 32, 192, 192, 5), (8, 8, 128, 768, 1)):
    conv_case(*shp)
for shp in ((8, 4, 1536, 1, 0.0), (8, 4, 1536, 0, 0.0), (8, 4, 1536, 1, 3.0), (8, 16, 384, 1, 0.0)):
    bn_case(*shp)
