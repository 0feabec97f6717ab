"""Micro-benchmark of the depthwise 5x5 kernels at the decoder-tower shapes.  usage: python tools/bench_dw.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from nvae_tf_amd._lib import call, ptr

dev = torch.device("cuda:0")
for name, (B, H, W, C) in {"mnist 4x4": (128, 4, 4, 1536), "mnist 8x8": (128, 8, 8, 768), "cifar 16x16": (64, 16, 16, 1536),
                           "celeba 32x32": (32, 32, 32, 384), "celeba 8x8": (32, 8, 8, 1536)}.items():
    x = torch.randn(B, H, W, C, device=dev).bfloat16()
    dy = torch.randn(B, H, W, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    w = torch.randn(25, C, device=dev)
    b = torch.randn(C, device=dev)
    dw, db = torch.zeros(25, C, device=dev), torch.zeros(C, device=dev)
    fns = {"fwd": lambda: call("nvae_dwconv5", 1, ptr(x), ptr(w), ptr(b), ptr(y), B, H, W, C, 0, 0),
           "wgrad": lambda: call("nvae_dwconv5_wgrad", 1, ptr(x), ptr(dy), ptr(dw), ptr(db), B, H, W, C)}
    out = []
    for k, fn in fns.items():
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 50
        byts = x.numel() * 2 * 2
        out.append(f"{k} {us:7.1f} us = {byts / us / 1e6:5.2f} TB/s")
    print(f"{name:14s} B{B} {H}x{W}x{C}: " + " | ".join(out))
