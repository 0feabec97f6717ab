"""A/B of nvae_conv_gemm against nvae_conv_gemm_ex arms (operand prologue, side store, in-kernel finalize) on
the tower shapes, interleaved rounds in one process (graph-captured chains of 20 launches)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import _lib as L
from nvae_tf_amd.ops import same_pad

dev = "cuda:0"
lib = L.load()
SHAPES = [  # B, H, Cin, Cout, k
    (128, 4, 256, 256, 3), (128, 4, 256, 1536, 1), (128, 4, 1536, 256, 1), (128, 8, 128, 768, 1), (128, 8, 128, 128, 3),
    (128, 16, 64, 384, 1), (128, 16, 384, 384, 5),
]
def geom(B, H, ci, co, k):
    p = same_pad(H, k, 1)[0]
    return L.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, p, p, 1, 0, ci, co, co)

for (B, H, ci, co, k) in SHAPES:
    g = geom(B, H, ci, co, k)
    x = torch.randn(B, H, H, ci, device=dev).bfloat16()
    w = (torch.randn(co, k * k * ci, device=dev) * 0.05).bfloat16()
    out = torch.empty(B, H, H, co, device=dev, dtype=torch.float32)     # (large enough for either output dtype)
    act = torch.empty_like(x)
    S = lib.nvae_conv_gemm_stats_rows(L.BF16, C.byref(g))
    slab = torch.zeros(S, 2, co, device=dev)
    coef = torch.rand(4, ci, device=dev) + 0.5
    coef2 = torch.empty(4, co, device=dev)
    gamma, beta, rm, rv = (torch.ones(co, device=dev) for _ in range(4))
    counters = torch.zeros(256, dtype=torch.int32, device=dev)
    sc, sh = L.ptr(coef), L.ptr(coef) + ci * 4
    bnin = L.BnIn(None, 0, 0.05, 1e-5, None, None, None, None, sc, sh, sc, sh)
    fin = L.BnFin(L.ptr(counters), L.ptr(gamma), L.ptr(beta), L.ptr(rm), L.ptr(rv), 0.05, 1e-5, L.ptr(coef2),
                  L.ptr(coef2) + co * 4, L.ptr(coef2) + 2 * co * 4, L.ptr(coef2) + 3 * co * 4)
    arms = {
        "plain": lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None),
        "stats": lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, L.ptr(slab)),
        "stats+fin": lambda: L.call("nvae_conv_gemm_ex", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, L.ptr(slab), None, C.byref(fin)),
        "pre(none)": lambda: L.call("nvae_conv_gemm_ex", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None, C.byref(L.ConvPre(bnin, 0, None, ci)), None),
        "pre(swish)": lambda: L.call("nvae_conv_gemm_ex", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None, C.byref(L.ConvPre(bnin, 1, None, ci)), None),
        "pre+store": lambda: L.call("nvae_conv_gemm_ex", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None, C.byref(L.ConvPre(bnin, 1, L.ptr(act), ci)), None),
        "pre+store+stats+fin": lambda: L.call("nvae_conv_gemm_ex", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, L.ptr(slab), C.byref(L.ConvPre(bnin, 1, L.ptr(act), ci)), C.byref(fin)),
    }
    graphs = {}
    for name, fn in arms.items():
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fn()
        graphs[name] = gr
    res = {n: [] for n in arms}
    for rnd in range(7):
        for name, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 1000 / 20)
    fn = arms["plain"]
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); torch.cuda.synchronize()
    res["plain(eager)"] = [e0.elapsed_time(e1) * 1000 / 20]
    print(f"B{B} {H}x{H} {k}x{k} {ci}->{co}: " + "  ".join(f"{n} {sorted(v)[len(v)//2]:.1f}" for n, v in res.items()), flush=True)
