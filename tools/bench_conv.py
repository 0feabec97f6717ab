"""Per-shape timing of the conv kernels (forward / data-gradient / weight-gradient) through the same
ops.conv2d path the model uses.  usage: python tools/bench_conv.py [f32]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from nvae_tf_amd import ops
from nvae_tf_amd.ops import Ctx, Var
from nvae_tf_amd.params import ParamStore

SHAPES = [  # B, H, cin, cout, k, up
    (128, 4, 256, 256, 3, 1), (128, 8, 128, 128, 3, 1), (128, 4, 256, 1536, 1, 1), (128, 4, 1536, 256, 1, 1),
    (128, 8, 128, 768, 1, 1), (128, 8, 768, 128, 1, 1), (128, 4, 256, 40, 3, 1), (128, 4, 256, 256, 1, 1),
    (128, 16, 64, 384, 1, 1), (128, 16, 384, 64, 1, 1), (128, 32, 32, 192, 1, 1), (128, 32, 192, 32, 1, 1),
    (128, 16, 384, 384, 5, 1), (128, 32, 192, 192, 5, 1), (128, 8, 128, 64, 3, 2), (128, 16, 64, 32, 3, 2),
    (128, 32, 32, 32, 3, 1), (128, 16, 64, 64, 3, 1),
]


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def main():
    dtype = torch.float32 if "f32" in sys.argv else torch.bfloat16
    dev = torch.device("cuda:0")
    print(f"{'shape':40s} {'fwd us':>9s} {'TF':>7s} {'bwd(dgrad+wgrad) us':>20s} {'TF':>7s}")
    ft = [a for a in sys.argv[1:] if a.startswith("--force-tile=")]      # k_conv_gemm2 tile family (nvae_conv_gemm_force_tile)
    if ft:
        from nvae_tf_amd import _lib as L
        L.load().nvae_conv_gemm_force_tile(int(ft[0].split("=")[1]))
    only = [a for a in sys.argv[1:] if a.startswith("--only=")]
    shapes = SHAPES
    if only:
        idx = [int(v) for v in only[0].split("=")[1].split(",")]
        shapes = [SHAPES[i] for i in idx]
    for (B, H, cin, cout, k, up) in shapes:
        ps = ParamStore(seed=1)
        conv = ps.conv("c", k, cin, cout)
        ps.finalize(dev, dtype, zero_pool_floats=1 << 16)
        ps.begin_step(); ps.prepare_weights(False)
        x = Var(torch.randn(B, H, H, cin, device=dev).to(dtype))
        dy = torch.randn(B, H * up, H * up, cout, device=dev).to(dtype)
        flops = 2.0 * B * (H * up) ** 2 * k * k * cin * cout

        def fwd():
            ctx = Ctx(ps, dtype, True, False)
            ops.conv2d(ctx, x, conv, up=up)
        ctx = Ctx(ps, dtype, True, True)
        y = ops.conv2d(ctx, x, conv, up=up)
        y.g = dy
        bwd_fn = ctx.tape[0]

        def bwd():
            x.g = None
            bwd_fn()

        def wgrad_only():
            x.needs_grad = False
            bwd_fn()
            x.needs_grad = True
        tf_, tb, tw = timeit(fwd), timeit(bwd), timeit(wgrad_only)
        name = f"B{B} {H}x{H} {cin}->{cout} k{k} up{up}"
        print(f"{name:40s} {tf_:9.1f} {flops / tf_ / 1e6:7.1f} {tb:20.1f} {2 * flops / tb / 1e6:7.1f}   wgrad {tw:8.1f} us {flops / tw / 1e6:7.1f} TF")


if __name__ == "__main__":
    main()
