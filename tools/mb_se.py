"""Fused SE + residual at the small-batch configurations' shapes: whole-image kernels (S = 1) vs the image-split form
(S slices per image, in-kernel hand-off), forward and backward, graph-captured chains of 10 launches.
usage: python tools/mb_se.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import _lib as L
dev = torch.device("cuda:0")
lib = L.load()
L.ensure_workspace(dev)
SHAPES = [(32, 32, 128), (32, 16, 256), (32, 8, 512), (32, 64, 64), (64, 16, 256), (64, 32, 128), (32, 4, 256), (128, 4, 256)]
dt, code = torch.bfloat16, L.BF16


def chain(fn, n=10):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    return g


for (B, H, Cc) in SHAPES:
    HW, Hd = H * H, max(Cc // 16, 4)
    x, sk, dy = (torch.randn(B, H, H, Cc, device=dev).to(dt) for _ in range(3))
    y, gx, gs = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    coef = torch.zeros(4, Cc, device=dev); coef[0] = 1; coef[3] = 1
    gam, bet, rm, rv = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev), torch.ones(Cc, device=dev)
    bn = L.BnIn(None, 0, 0.05, 1e-5, L.ptr(gam), L.ptr(bet), L.ptr(rm), L.ptr(rv), *[L.ptr(coef[i]) for i in range(4)])
    w1, b1 = torch.randn(Cc, Hd, device=dev) * 0.1, torch.zeros(Hd, device=dev)
    w2, b2 = torch.randn(Hd, Cc, device=dev) * 0.1, torch.zeros(Cc, device=dev)
    pooled, gate, hidden = torch.empty(B, Cc, device=dev), torch.empty(B, Cc, device=dev), torch.empty(B, Hd, device=dev)
    scratch = torch.empty(B, Cc + Hd, device=dev)
    st, part = torch.zeros(8, 2, Cc, device=dev), torch.zeros(8, 2, Cc, device=dev)
    res = {}
    graphs = {}
    for S in (1, 2, 4, 8, 16):
        lib.nvae_se_force_split(S)
        fwd = lambda: L.call("nvae_se_fused_fwd", code, L.ptr(x), C.byref(bn), L.ptr(sk), L.ptr(y), B, HW, Cc, Hd, L.ptr(w1),
                             L.ptr(b1), L.ptr(w2), L.ptr(b2), 0.1, 1.0, L.ptr(pooled), L.ptr(gate), L.ptr(hidden), L.ptr(st))
        bwd = lambda: L.call("nvae_se_fused_bwd", code, L.ptr(x), L.ptr(coef[0]), L.ptr(coef[1]), L.ACT_NONE, L.ptr(dy), L.ptr(gate),
                             L.ptr(hidden), L.ptr(gx), L.ptr(gs), B, HW, Cc, Hd, L.ptr(w1), L.ptr(w2), 0.1, 1.0, 0, 0,
                             L.ptr(scratch), L.ptr(part))
        graphs[(S, "fwd")] = chain(fwd); graphs[(S, "bwd")] = chain(bwd)
    lib.nvae_se_force_split(-1)
    out = {k: [] for k in graphs}
    for _ in range(5):
        for k, g in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            out[k].append(e0.elapsed_time(e1) * 100)
    med = {k: sorted(v)[2] for k, v in out.items()}
    mb = x.numel() * 2 / 1e6
    print(f"B{B} {H}x{H}x{Cc} ({mb:.1f} MB/tensor): " + "  ".join(
        f"S={S}: fwd {med[(S, 'fwd')]:.1f} bwd {med[(S, 'bwd')]:.1f}" for S in (1, 2, 4, 8, 16)), flush=True)
