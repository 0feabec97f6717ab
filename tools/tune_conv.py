"""Tile-family sweep of the generic implicit-GEMM kernel on the tower shapes (forward and data-gradient
geometries), graph-captured chains of 20 launches, interleaved rounds.  usage: python tools/tune_conv.py"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import _lib as L
from nvae_tf_amd.ops import same_pad

dev = "cuda:0"
lib = L.load()
NAMES = {0: "auto", 1: "256x192", 2: "128x192", 3: "128x128", 4: "128x64", 5: "32x64/128deep", 6: "64x64/128deep", 7: "64x64"}
SHAPES = [  # B, H, Cin, Cout, k
    (128, 4, 256, 256, 3), (128, 4, 256, 1536, 1), (128, 4, 1536, 256, 1), (128, 4, 256, 256, 1), (128, 4, 256, 40, 3),
    (128, 8, 128, 128, 3), (128, 8, 128, 768, 1), (128, 8, 768, 128, 1), (128, 8, 128, 128, 1),
    (128, 16, 64, 384, 1), (128, 16, 384, 64, 1), (128, 32, 32, 192, 1), (128, 32, 192, 32, 1), (128, 16, 64, 64, 3), (128, 32, 32, 32, 3),
]
for (B, H, ci, co, k) in SHAPES:
    p = same_pad(H, k, 1)[0]
    g = L.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, p, p, 1, 0, ci, co, co)
    x = torch.randn(B, H, H, ci, device=dev).bfloat16()
    w = (torch.randn(co, k * k * ci, device=dev) * 0.05).bfloat16()
    out = torch.empty(B, H, H, co, device=dev, dtype=torch.bfloat16)
    graphs = {}
    for t in NAMES:
        lib.nvae_conv_gemm_force_tile(t)
        fn = lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None)
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fn()
        graphs[t] = gr
    lib.nvae_conv_gemm_force_tile(0)
    res = {t: [] for t in NAMES}
    for rnd in range(5):
        for t, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[t].append(e0.elapsed_time(e1) * 1000 / 20)
    med = {t: sorted(v)[len(v) // 2] for t, v in res.items()}
    best = min((v, t) for t, v in med.items() if t)
    print(f"B{B} {H}x{H} {k}x{k} {ci}->{co}: " + "  ".join(f"{NAMES[t]} {v:.1f}" for t, v in med.items()) + f"   best {NAMES[best[1]]}", flush=True)
