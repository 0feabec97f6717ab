"""The dense 5x5 halo kernel in its two 16-bit forms (nvae_conv_halo4_enable): eight ping-pong waves of 64 x 96 against four
software-pipelined waves of 128 x 96.  Every launch form the step uses (plain, forward + BN statistics, operand prologue,
data gradient + BN-backward sums) at both shapes: results compared bit for bit, graph-captured chains of 10 launches timed
in interleaved rounds.  --clock: in-kernel clock and MFMA-pipe share of the main loop; --phases: where a launch's time goes
(stamps per workgroup).   usage: python tools/mb_halo.py [--f16] [--clock] [--phases]"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd import _lib as L

dev = "cuda:0"
lib = L.load()
dt = torch.float16 if "--f16" in sys.argv else torch.bfloat16
code = L.dtype_code(dt)
B = 128
keep = []


def chain(fn, n=10):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    return gr


def time_graphs(graphs, rounds=7, n=10):
    res = {k: [] for k in graphs}
    for _ in range(rounds):
        for k, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1000 / n)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


for hw, ci in ((16, 384), (32, 192)):
    co = ci
    torch.manual_seed(hw)
    x = torch.randn(B, hw, hw, ci, device=dev).to(dt)
    w = (torch.randn(co, 25 * ci, device=dev) / (25 * ci) ** 0.5).to(dt)
    g = L.ConvGeom(B, hw, hw, ci, hw, hw, co, 5, 5, 1, 2, 2, 1, 0, ci, co, co)
    rows = lib.nvae_conv_gemm_stats_rows(code, C.byref(g))
    coef = torch.rand(4, ci, device=dev) + 0.5
    flops = 2.0 * B * hw * hw * 25 * ci * co
    graphs, outs = {}, {}
    for h4 in (0, 1):
        lib.nvae_conv_halo4_enable(h4)
        y0 = torch.full((B, hw, hw, co), float("nan"), device=dev, dtype=dt)
        y1, y2, dx = torch.full_like(y0, float("nan")), torch.full_like(y0, float("nan")), torch.full_like(y0, float("nan"))
        slab = torch.zeros(rows, 2, co, device=dev)
        part = torch.zeros(rows, 2, ci, device=dev)
        act = torch.full_like(x, float("nan"))
        dgb, k0k1 = torch.zeros(2, ci, device=dev), torch.zeros(2, ci, device=dev)
        f = L.BnBwdFuse(L.ptr(x), ci, L.ACT_SWISH, 0, L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]),
                        L.ptr(part), None, L.ptr(dgb[0]), L.ptr(dgb[1]), L.ptr(k0k1))
        keep += [y0, y1, y2, dx, slab, part, act, dgb, k0k1, f]
        plain = lambda y0=y0: L.call("nvae_conv_gemm", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(y0), 0, None)
        stats = lambda y1=y1, slab=slab: L.call("nvae_conv_gemm", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(y1), 0, L.ptr(slab))
        dgrad = lambda dx=dx, f=f: L.call("nvae_conv_gemm_bnbwd", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(dx), C.byref(f))
        for nm, fn in (("plain", plain), ("fwd+stats", stats), ("dgrad+bnbwd", dgrad)):
            fn(); torch.cuda.synchronize()
        outs[h4] = (y0.clone(), y1.clone(), slab.clone(), dx.clone(), part.clone())
        for nm, fn in (("plain", plain), ("fwd+stats", stats), ("dgrad+bnbwd", dgrad)):
            graphs[(nm, h4)] = chain(fn)
    lib.nvae_conv_halo4_enable(0)
    a, b = outs[0], outs[1]
    assert bool(torch.isfinite(b[0].float()).all()) and bool(torch.isfinite(b[3].float()).all())
    same = [bool(torch.equal(p, q)) for p, q in zip(a, b)]
    # (slabs are sums of f32 atomics: equal up to their order)
    close = [float((p.float() - q.float()).abs().max() / (p.float().abs().max() + 1e-30)) for p, q in zip(a, b)]
    print(f"== {hw}x{hw} {ci}->{co}: 4-wave vs 8-wave  y {same[0]}  y(stats) {same[1]}  slab {close[2]:.1e}  dx {same[3]}  partials {close[4]:.1e}", flush=True)
    assert same[0] and same[1] and same[3] and close[2] < 1e-5 and close[4] < 1e-5
    med = time_graphs(graphs)
    for (nm, h4), v in sorted(med.items()):
        print(f"   {nm:>12s} {'4-wave' if h4 else '8-wave'}: {v:7.1f} us  {flops / v / 1e6:7.0f} TFLOP/s", flush=True)

if "--clock" in sys.argv:
    # in-kernel clock of the main loop (guide: 'DVFS give-back' item 6) after 2 s of back-to-back launches on random data
    import time, numpy as np
    hw, ci = 16, 384
    x = torch.randn(B, hw, hw, ci, device=dev).to(dt)
    w = (torch.randn(ci, 25 * ci, device=dev) / (25 * ci) ** 0.5).to(dt)
    g = L.ConvGeom(B, hw, hw, ci, hw, hw, ci, 5, 5, 1, 2, 2, 1, 0, ci, ci, ci)
    y = torch.empty(B, hw, hw, ci, device=dev, dtype=dt)
    fn = lambda: L.call("nvae_conv_gemm", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(y), 0, None)
    mfma_cycles = 150 * 96 * 16            # per SIMD: 150 steps x 96 MFMAs of 16 cycles (either form)
    for form, label in ((0, "8-wave"), (1, "4-wave")):
        lib.nvae_conv_halo4_enable(form)
        gr = chain(fn, 50)
        t0 = time.time()
        while time.time() - t0 < 2.0:
            gr.replay()
        torch.cuda.synchronize()
        lib.nvae_conv_halo4_enable(form | 16)
        for _ in range(20):
            fn()
        buf = (C.c_ulonglong * 512)()
        assert lib.nvae_conv_halo_stamps(buf, 512) == 0
        a = np.array(buf[:], dtype=np.float64).reshape(256, 2)
        cyc, real = np.median(a[:, 0]), np.median(a[:, 1])
        print(f"   {label}: loop {cyc:.0f} cycles, {real / 100:.1f} us, clock {cyc / real * 100:.0f} MHz, MFMA pipe busy {mfma_cycles / cyc * 100:.1f} % of loop cycles "
              f"= {2 * B * hw * hw * 25.0 * ci * ci / (real / 100) / 1e6:.0f} TFLOP/s in-loop", flush=True)
    lib.nvae_conv_halo4_enable(0)

if "--phases" in sys.argv:
    # where a launch's time goes: absolute 100 MHz stamps per workgroup at entry, loop start, loop end, exit
    import numpy as np
    for hw, ci in ((16, 384), (32, 192)):
        x = torch.randn(B, hw, hw, ci, device=dev).to(dt)
        w = (torch.randn(ci, 25 * ci, device=dev) / (25 * ci) ** 0.5).to(dt)
        g = L.ConvGeom(B, hw, hw, ci, hw, hw, ci, 5, 5, 1, 2, 2, 1, 0, ci, ci, ci)
        y = torch.empty(B, hw, hw, ci, device=dev, dtype=dt)
        rows = lib.nvae_conv_gemm_stats_rows(code, C.byref(g))
        slab = torch.zeros(rows, 2, ci, device=dev)
        part = torch.zeros(rows, 2, ci, device=dev)
        coef = torch.rand(4, ci, device=dev) + 0.5
        dgb, k0k1 = torch.zeros(2, ci, device=dev), torch.zeros(2, ci, device=dev)
        f = L.BnBwdFuse(L.ptr(x), ci, L.ACT_SWISH, 0, L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]),
                        L.ptr(part), None, L.ptr(dgb[0]), L.ptr(dgb[1]), L.ptr(k0k1))
        forms = {"plain": lambda: L.call("nvae_conv_gemm", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(y), 0, None),
                 "fwd+stats": lambda: L.call("nvae_conv_gemm", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(y), 0, L.ptr(slab)),
                 "dgrad+bnbwd": lambda: L.call("nvae_conv_gemm_bnbwd", code, C.byref(g), L.ptr(x), L.ptr(w), 25 * ci, None, None, L.ptr(y), C.byref(f))}
        for nm, fn in forms.items():
            lib.nvae_conv_halo4_enable(0)
            gr = chain(fn, 20)
            wall = time_graphs({"a": gr}, n=20)["a"]
            lib.nvae_conv_halo4_enable(32)
            for _ in range(10):
                fn()
            n_wg = min(B * (hw // 16) ** 2 * (ci // 192), 512)
            buf = (C.c_ulonglong * 2048)()
            assert lib.nvae_conv_halo_stamps(buf, 2048) == 0
            lib.nvae_conv_halo4_enable(0)
            a = np.array(buf[:], dtype=np.float64).reshape(512, 4)[:n_wg] / 100.0      # us
            t0 = a[:, 0].min()
            print(f"   {hw}x{hw} {nm:>12s}: wall {wall:6.1f} us | first entry -> last exit {a[:, 3].max() - t0:6.1f} | per workgroup (median): prologue {np.median(a[:, 1] - a[:, 0]):5.1f}  "
                  f"loop {np.median(a[:, 2] - a[:, 1]):6.1f}  epilogue {np.median(a[:, 3] - a[:, 2]):5.1f} | entry spread {a[:, 0].max() - t0:5.1f}  last loop end {a[:, 2].max() - t0:6.1f}", flush=True)
