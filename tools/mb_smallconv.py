"""Small-M implicit GEMMs of the 4x4 / 8x8 towers: tile family x ring depth x split-K sweep (nvae_conv_gemm_force_tile /
nvae_conv_gemm_force_split), graph-captured chains of 20 launches, interleaved rounds, every variant checked against an
fp32 torch reference and for run-to-run bit equality.   usage: python tools/mb_smallconv.py [--quick]"""
import ctypes as C, sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from nvae_tf_amd import _lib as L
from nvae_tf_amd.ops import same_pad

dev = "cuda:0"
lib = L.load()
ws = torch.zeros(64 << 20, dtype=torch.uint8, device=dev)
cnt = torch.zeros(8192, dtype=torch.int32, device=dev)
lib.nvae_conv_set_workspace(L.ptr(ws), ws.numel(), L.ptr(cnt), cnt.numel())

FAM = {-1: "whole-image", 0: "auto", 4: "128x64", 5: "32x64/128d/3st", 6: "64x64/128d/3st", 7: "64x64/4w", 8: "32x64/128d/4st", 9: "32x64/128d/6st",
       10: "64x64/128d/4st", 11: "32x64/4w/64d/6st", 12: "64x128/128d/3st", 13: "64x128/64d/4st", 3: "128x128"}
SHAPES = [  # B, H, Cin, Cout, k
    (128, 4, 256, 256, 3), (128, 4, 1536, 256, 1), (128, 8, 128, 128, 3), (128, 8, 768, 128, 1),
    (128, 4, 256, 1536, 1), (128, 8, 128, 768, 1), (128, 4, 256, 40, 3), (128, 4, 256, 256, 1),
]
VARIANTS = [(-1, 1), (0, 1), (5, 1), (8, 1), (9, 1), (11, 1), (6, 1), (10, 1), (7, 1), (4, 1),
            (5, 2), (5, 3), (6, 2), (6, 3), (6, 4), (7, 2), (7, 4), (4, 2), (4, 4), (4, 8),
            (12, 1), (12, 2), (12, 4), (12, 8), (13, 2), (13, 4), (13, 8)]
if "--quick" in sys.argv:
    SHAPES, VARIANTS = [SHAPES[0], SHAPES[2], SHAPES[6]], [(-1, 1), (0, 1), (5, 2), (5, 3), (12, 4)]


def chain(fn, n=20):
    fn(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    return gr


def time_graphs(graphs, rounds=5, n=20):
    res = {k: [] for k in graphs}
    for _ in range(rounds):
        for k, gr in graphs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
            res[k].append(e0.elapsed_time(e1) * 1000 / n)
    return {k: sorted(v)[len(v) // 2] for k, v in res.items()}


# launch floor: a chain of trivial kernels
tiny = torch.zeros(64, device=dev, dtype=torch.bfloat16)
floor = time_graphs({"add": chain(lambda: L.call("nvae_add", L.BF16, L.ptr(tiny), L.ptr(tiny), 64, 0))})
print(f"launch floor (20 dependent trivial kernels in a graph): {floor['add']:.2f} us per launch", flush=True)

out_rows = []
for (B, H, ci, co, k) in SHAPES:
    p = same_pad(H, k, 1)[0]
    g = L.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, p, p, 1, 0, ci, co, co)
    torch.manual_seed(1)
    x = torch.randn(B, H, H, ci, device=dev).bfloat16()
    w = (torch.randn(co, k * k * ci, device=dev) / (k * k * ci) ** 0.5).bfloat16()
    # fp32 reference: w is [co][(kh,kw,ci)]
    wref = w.float().view(co, k, k, ci).permute(0, 3, 1, 2).contiguous()
    # (reference on the CPU: torch's GPU convolution would go through MIOpen)
    ref = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wref.cpu(), padding=p).permute(0, 2, 3, 1).contiguous().to(dev)
    scale = float(ref.abs().max())
    graphs, errs, keep = {}, {}, []      # (a captured graph does not own the tensors its kernels write)
    for (t, S) in VARIANTS:
        lib.nvae_conv_gemm_force_tile(max(t, 0)); lib.nvae_conv_gemm_force_split(S); lib.nvae_conv_img_enable(int(t < 0))
        print(f"   .. {H}x{H} k{k} {ci}->{co}: {FAM[t]} S={S}", flush=True)
        out = torch.full((B, H, H, co), float("nan"), device=dev, dtype=torch.bfloat16)
        fn = lambda out=out: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None)
        keep.append(out)
        fn(); torch.cuda.synchronize()
        o1 = out.clone()
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        errs[(t, S)] = (float((out.float() - ref).abs().max()) / scale, bool(torch.equal(o1, out)))
        graphs[(t, S)] = chain(fn)
        assert int(cnt.abs().sum()) == 0, "split-K counters not back at zero"
    lib.nvae_conv_gemm_force_tile(0); lib.nvae_conv_gemm_force_split(0); lib.nvae_conv_img_enable(1)
    med = time_graphs(graphs)
    M, K = B * H * H, k * k * ci
    print(f"== B{B} {H}x{H} {k}x{k} {ci}->{co}  (M={M} K={K} N={co}, {2.0 * M * K * co / 1e9:.2f} GFLOP)", flush=True)
    for (t, S), v in sorted(med.items(), key=lambda kv: kv[1]):
        e, same = errs[(t, S)]
        print(f"   {FAM[t]:>18s} S={S}: {v:6.2f} us   err {e:.1e} {'bit-stable' if same else 'NOT bit-stable'}", flush=True)
        out_rows.append({"shape": [B, H, ci, co, k], "tile": FAM[t], "S": S, "us": round(v, 2), "err": e, "stable": same})

if "--quick" in sys.argv:
    sys.exit(0)
lib.nvae_conv_img_enable(0)
# fixed cost vs K: the 3x3 conv at 4x4 with growing Cin (auto tile, no split)
for ci in (32, 64, 128, 256, 512):
    B, H, co, k = 128, 4, 256, 3
    g = L.ConvGeom(B, H, H, ci, H, H, co, k, k, 1, 1, 1, 1, 0, ci, co, co)
    x = torch.randn(B, H, H, ci, device=dev).bfloat16()
    w = (torch.randn(co, k * k * ci, device=dev) * 0.05).bfloat16()
    out = torch.empty(B, H, H, co, device=dev, dtype=torch.bfloat16)
    graphs = {}
    for t in (5, 9):
        lib.nvae_conv_gemm_force_tile(t)
        graphs[t] = chain(lambda: L.call("nvae_conv_gemm", L.BF16, C.byref(g), L.ptr(x), L.ptr(w), k * k * ci, None, None, L.ptr(out), 0, None))
    lib.nvae_conv_gemm_force_tile(0)
    med = time_graphs(graphs)
    print(f"K-scaling 4x4 3x3 {ci}->256 (K={9 * ci}): " + "  ".join(f"{FAM[t]} {v:.2f} us" for t, v in med.items()), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
with open("gpurun_out/mb_smallconv.json", "w") as fh:
    json.dump(out_rows, fh)
