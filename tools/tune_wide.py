"""Tile-family sweep for the short-K, wide-N 1x1 convs (the 'expand' convs of the depthwise-separable cell and their
mirror images), WITH the epilogues they really run (statistics slab, operand prologue): plain nvae_conv_gemm timings
pick 128-row tiles for these shapes, but their time is almost all epilogue.  usage: python tools/tune_wide.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import _lib as L
from nvae_tf_amd.ops import same_pad

dev = "cuda:0"
lib = L.load()
SHAPES = [(128, 8, 128, 768, 1), (128, 8, 768, 128, 1), (128, 4, 256, 1536, 1), (128, 4, 1536, 256, 1), (128, 16, 64, 384, 1),
          (128, 16, 384, 64, 1), (128, 32, 32, 192, 1), (128, 32, 192, 32, 1)]
for (B, H, ci, co, k) in SHAPES:
    p = same_pad(H, k, 1)[0]
    g = L.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, p, p, 1, 0, ci, co, co)
    x = torch.randn(B, H, H, ci, device=dev).bfloat16()
    w = (torch.randn(co, k * k * ci, device=dev) * 0.05).bfloat16()
    out = torch.empty(B, H, H, co, device=dev, dtype=torch.bfloat16)
    coef = torch.rand(4, ci, device=dev) + 0.5
    sc, sh = L.ptr(coef), L.ptr(coef) + ci * 4
    bnin = L.BnIn(None, 0, 0.05, 1e-5, None, None, None, None, sc, sh, sc, sh)
    line = f"B{B} {H}x{H} {ci}->{co}:"
    for fam in (0, 1, 2, 3, 4, 7):
        lib.nvae_conv_gemm_force_tile(fam)
        S = lib.nvae_conv_gemm_stats_rows(L.BF16, C.byref(g)) if fam == 0 else 8
        slab = torch.zeros(max(S, 8) * 64, 2, co, device=dev)          # generous: forced tiles have other row counts
        arms = {
            "plain": lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None),
            "stats": lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, L.ptr(slab)),
            "pre+stats": lambda: L.call("nvae_conv_gemm_ex", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0,
                                        L.ptr(slab), C.byref(L.ConvPre(bnin, 0, None, ci)), None),
        }
        res = {}
        for name, fn in arms.items():
            fn(); torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(20):
                    fn()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1000 / 20)
            res[name] = sorted(ts)[2]
        line += f"  fam{fam}: " + "/".join(f"{res[n]:.1f}" for n in arms)
    lib.nvae_conv_gemm_force_tile(0)
    print(line + "   (plain/stats/pre+stats us)", flush=True)
