"""Headline benchmark: NVAE training throughput (images/s) on MNIST-shaped synthetic data.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = BASELINE.json configs[1]: MNIST 28x28 zero-padded to 32x32, paper defaults (groups [5,10],
2 residual cells per group, 20 latents per group), batch 128 PER GPU, bf16 activations / MFMA with
f32 accumulation, f32 master weights, f32 BN statistics / KL / reconstruction.  One step = spectral
norm power iteration + forward + ELBO + backward + (gradient all-reduce) + Adamax, replayed from
hipGraphs.  Weak scaling: per-GPU batch fixed.  Prints ONE JSON line on rank 0."""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

TRAIN_FLOP_PER_IMG = 41.70e9     # 6 * 6949.4 M MAC (C2/C3)
PEAK_BF16_TFLOPS = 2500.0        # MI355X_MICROARCH.md: dense bf16 MFMA peak
BATCH_PER_GPU = 128


def make_model(device, dtype, batch):
    from nvae_tf_amd.models import NVAE
    iters = 400 * (60000 // batch)
    return NVAE(32, 32, 2, 2, 3, 20, 2, [5, 10], 2, 3, 0.01, 2, 400, iters, True, [batch, 32, 32, 1],
                device=device, dtype=dtype, seed=1)


def synthetic_batch(batch, seed, device):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(batch, 32, 32, 1)
    x[:, 2:30, 2:30, :] = (torch.rand(batch, 28, 28, 1, generator=g) < 0.19).float()
    return x.to(device)


def time_dominant_kernel(model, batch, iters=20):
    """Average launch duration of the dominant kernel (k_conv_halo, the dense 5x5 implicit GEMM of Postprocess) at
    its two shapes and in the two forms the step launches it: forward with the BatchNorm-statistics epilogue
    (ConvBNSwish, postprocess.py:100-108) and data gradient with the BatchNorm-backward epilogue
    (nvae_conv_gemm_bnbwd).  HIP events on the stream the kernel is launched on."""
    from nvae_tf_amd import _lib as L
    ps = model.ps
    dev = model.device
    code = L.dtype_code(model.dtype)
    out = []
    for name, hw in (("post.cell1.conv5", 16), ("post.cell4.conv5", 32)):
        conv = next(c for c in ps.convs if c.name == name)
        ci, co = conv.cin, conv.cout
        x = torch.randn(batch, hw, hw, ci, device=dev).to(model.dtype)
        y = torch.empty(batch, hw, hw, co, device=dev, dtype=model.dtype)
        g = L.ConvGeom(batch, hw, hw, ci, hw, hw, co, 5, 5, 1, 2, 2, 1, 0, ci, co, co)
        gd = L.ConvGeom(batch, hw, hw, co, hw, hw, ci, 5, 5, 1, 2, 2, 1, 1, co, ci, ci)
        wT = L.ptr(ps.wcopies) + conv.wf_off * ps.wcopies.element_size()
        wD = L.ptr(ps.wcopies) + conv.wd_off * ps.wcopies.element_size()
        rows = L.load().nvae_conv_gemm_stats_rows(code, C.byref(g))
        slab = torch.zeros(rows, 2, co, device=dev)
        rows_d = L.load().nvae_conv_gemm_stats_rows(code, C.byref(gd))
        part = torch.zeros(rows_d, 2, ci, device=dev)
        coef = torch.ones(4, ci, device=dev)
        dgb, k0k1 = torch.zeros(2, ci, device=dev), torch.zeros(2, ci, device=dev)
        dx = torch.empty(batch, hw, hw, ci, device=dev, dtype=model.dtype)
        f = L.BnBwdFuse(L.ptr(x), ci, L.ACT_SWISH, 0, L.ptr(coef[0]), L.ptr(coef[1]), L.ptr(coef[2]), L.ptr(coef[3]),
                        L.ptr(part), None, L.ptr(dgb[0]), L.ptr(dgb[1]), L.ptr(k0k1))

        def fwd():
            L.call("nvae_conv_gemm", code, C.byref(g), L.ptr(x), wT, conv.wf_ld, None, None, L.ptr(y), 0, L.ptr(slab))

        def dgrad():
            L.call("nvae_conv_gemm_bnbwd", code, C.byref(gd), L.ptr(y), wD, conv.wd_ld, None, None, L.ptr(dx), C.byref(f))
        flops = 2.0 * batch * hw * hw * 25 * ci * co
        for kind, launch in (("forward + BN statistics", fwd), ("data gradient + BN backward sums", dgrad)):
            for _ in range(3):
                launch()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                launch()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / iters
            out.append({"shape": f"B{batch}x{hw}x{hw} 5x5 {ci}->{co}", "launch": kind, "ms": ms,
                        "tflops": flops / ms / 1e9})
    return out


PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 1/16 of the bf16 rate
PEAK_HBM_GBPS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def time_hbm_kernels(model, batch, iters=30):
    """The HBM-bound kernels of the residual cells in the forms the step launches them
    (round 3: the lazy-BatchNorm / fused variants, not the round-1 kernels) at this workload's two tower shapes:
    algorithmic bytes (elements x bytes x tensor passes) / launch duration, HIP events on the launch stream."""
    from nvae_tf_amd import _lib as L
    dev, dt = model.device, model.dtype
    es = 2 if dt != torch.float32 else 4
    code = L.dtype_code(dt)
    lib = L.load()
    out = []
    for hw, cw, cb in ((4, 1536, 256), (8, 768, 128)):       # expanded width (decoder cell interior), cell width
        rows = batch * hw * hw
        cases = []

        def bn_tensors(ch):
            sl = torch.zeros(8, 2, ch, device=dev)
            sl[0, 1] = rows                                   # sum x = 0, sum x^2 = rows: mean 0, variance 1
            coef = torch.zeros(4, ch, device=dev); coef[0] = 1; coef[3] = 1
            return sl, coef, torch.ones(ch, device=dev), torch.zeros(ch, device=dev), torch.zeros(ch, device=dev), torch.ones(ch, device=dev)
        for ch in (cw, cb):
            x = torch.randn(batch, hw, hw, ch, device=dev).to(dt)
            dy = torch.randn(batch, hw, hw, ch, device=dev).to(dt)
            y = torch.empty_like(x)
            n = x.numel()
            sl, coef, gam, bet, rm, rv = bn_tensors(ch)
            cp = [L.ptr(coef[i]) for i in range(4)]
            dgb = torch.zeros(2, ch, device=dev)
            cases.append((f"k_bn_apply_fin (BN + Swish, statistics slab -> coefficients in-kernel) C={ch}", 2 * n * es,
                          lambda x=x, y=y, sl=sl, gam=gam, bet=bet, rm=rm, rv=rv, cp=cp, ch=ch: L.call(
                              "nvae_bn_apply_fin", code, L.ptr(x), L.ptr(y), rows, ch, L.ptr(sl), 1, L.ptr(gam), L.ptr(bet),
                              L.ptr(rm), L.ptr(rv), 0.05, 1e-5, *cp, L.ACT_SWISH)))
            cases.append((f"k_bn_bwd_apply_fin (BN + Swish backward, sums -> dgamma/dbeta/k0,k1 in-kernel) C={ch}", 3 * n * es,
                          lambda x=x, y=y, dy=dy, sl=sl, cp=cp, dgb=dgb, ch=ch: L.call(
                              "nvae_bn_bwd_apply_fin", code, L.ptr(x), L.ptr(dy), L.ptr(y), rows, ch, L.ptr(sl), 1, *cp,
                              L.ptr(dgb[0]), L.ptr(dgb[1]), L.ACT_SWISH, 0, 0)))
        # depthwise 5x5 on the expanded tensor, BN2 + Swish applied to the halo tile in LDS
        x = torch.randn(batch, hw, hw, cw, device=dev).to(dt)
        dy = torch.randn(batch, hw, hw, cw, device=dev).to(dt)
        y = torch.empty_like(x)
        n = x.numel()
        sl, coef, gam, bet, rm, rv = bn_tensors(cw)
        bn_in = L.BnIn(None, 0, 0.05, 1e-5, L.ptr(gam), L.ptr(bet), L.ptr(rm), L.ptr(rv), *[L.ptr(coef[i]) for i in range(4)])
        w, b = torch.randn(25, cw, device=dev), torch.randn(cw, device=dev)
        dw, db = torch.zeros(25, cw, device=dev), torch.zeros(cw, device=dev)
        srows = lib.nvae_dwconv5_stats_rows(code, batch, hw, hw, cw)
        dslab = torch.zeros(max(srows, 1), 2, cw, device=dev)
        if es == 2:
            cases.append((f"k_dw5_fwd_ring<pre> (BN2 + Swish -> depthwise 5x5 -> BN3 statistics) C={cw}", 2 * n * es,
                          lambda: L.call("nvae_dwconv5_pre", code, L.ptr(x), C.byref(bn_in), L.ACT_SWISH, L.ptr(w), L.ptr(b),
                                         L.ptr(y), batch, hw, hw, cw, L.ptr(dslab) if srows else None)))
            cases.append((f"k_dw5_wgrad_ring<pre> (depthwise weight gradient on act(BN2(x))) C={cw}", 2 * n * es,
                          lambda: L.call("nvae_dwconv5_wgrad_pre", code, L.ptr(x), L.ptr(coef[0]), L.ptr(coef[1]), L.ACT_SWISH,
                                         L.ptr(dy), L.ptr(dw), L.ptr(db), batch, hw, hw, cw)))
        cases.append((f"k_dw5_fwd_ring (depthwise 5x5 data gradient) C={cw}", 2 * n * es,
                      lambda: L.call("nvae_dwconv5", code, L.ptr(dy), L.ptr(w), None, L.ptr(y), batch, hw, hw, cw, 1, 0)))
        # fused SE + residual on the cell-width tensor, BN4 applied on load, statistics of the next BN emitted
        ch, hd = cb, max(cb // 16, 4)
        xs = torch.randn(batch, hw, hw, ch, device=dev).to(dt)
        sk = torch.randn(batch, hw, hw, ch, device=dev).to(dt)
        dys = torch.randn(batch, hw, hw, ch, device=dev).to(dt)
        ys, gx, gs = torch.empty_like(xs), torch.empty_like(xs), torch.empty_like(xs)
        ns = xs.numel()
        sl2, coef2, gam2, bet2, rm2, rv2 = bn_tensors(ch)
        bn2 = L.BnIn(None, 0, 0.05, 1e-5, L.ptr(gam2), L.ptr(bet2), L.ptr(rm2), L.ptr(rv2), *[L.ptr(coef2[i]) for i in range(4)])
        w1, b1 = torch.randn(ch, hd, device=dev) * 0.1, torch.zeros(hd, device=dev)
        w2, b2 = torch.randn(hd, ch, device=dev) * 0.1, torch.zeros(ch, device=dev)
        pooled, gate = torch.empty(batch, ch, device=dev), torch.empty(batch, ch, device=dev)
        hidden = torch.empty(batch, hd, device=dev)
        se_rows = lib.nvae_se_fused_rows(batch, hw * hw, ch)
        st, part = torch.zeros(se_rows, 2, ch, device=dev), torch.zeros(se_rows, 2, ch, device=dev)
        scratch = torch.empty(batch, ch + hd, device=dev)
        if ch & (ch - 1) == 0:
            cases.append((f"k_se_fused_fwd (BN4 on load + pool + FC + gate + residual + next BN statistics) C={ch}", 3 * ns * es,
                          lambda: L.call("nvae_se_fused_fwd", code, L.ptr(xs), C.byref(bn2), L.ptr(sk), L.ptr(ys), batch, hw * hw,
                                         ch, hd, L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), 0.1, 1.0, L.ptr(pooled),
                                         L.ptr(gate), L.ptr(hidden), L.ptr(st))))
            cases.append((f"k_se_fused_bwd (SE + residual backward + BN4 backward sums) C={ch}", 4 * ns * es,
                          lambda: L.call("nvae_se_fused_bwd", code, L.ptr(xs), L.ptr(coef2[0]), L.ptr(coef2[1]), L.ACT_NONE,
                                         L.ptr(dys), L.ptr(gate), L.ptr(hidden), L.ptr(gx), L.ptr(gs), batch, hw * hw, ch, hd,
                                         L.ptr(w1), L.ptr(w2), 0.1, 1.0, 0, 0, L.ptr(scratch), L.ptr(part))))
        for name, nbytes, fn in cases:
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            gbps = nbytes / us / 1e3
            out.append({"kernel": name, "shape": f"B{batch}x{hw}x{hw}", "us": round(us, 2),
                        "algorithmic_bytes": nbytes, "GB/s": round(gbps, 1), "frac_of_hbm_peak": round(gbps / PEAK_HBM_GBPS, 4)})
    return out


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(seconds_budget=24.0, workload="mnist_c2"):
    """The CPU oracle (PyTorch-CPU restatement of the reference arithmetic, f32, eager) timed on the
    host cores: same model (C2), a bounded sample of the workload (batch 8).  Two legs: all the
    cores this process may use (at most 16: the box's CPU share for one GPU), and 8 threads for comparability
    with the 8-vCPU build container.  `--workload mnist_c1` times BASELINE.json configs[0] (the reference's
    CPU-runnable case: groups [1,1], 1 cell) at its full batch of 32 instead."""
    from oracle.nvae_oracle import OracleConfig, OracleNVAE, synthetic_batch as sb
    if workload == "mnist_c1":
        b, ocfg = 32, OracleConfig(n_groups_per_scale=[1, 1], res_cells_per_group=1)
    else:
        b, ocfg = 8, OracleConfig(n_groups_per_scale=[5, 10], res_cells_per_group=2)
    orc = OracleNVAE(ocfg, dtype=torch.float32, seed=1)
    x = sb(b, seed=1, dtype=torch.float32)
    g = torch.Generator().manual_seed(2)
    eps = [torch.randn(s, generator=g) for s in orc.eps_shapes(b)]

    def leg(threads, budget):
        torch.set_num_threads(threads)
        orc.train_step(x, eps)   # warm-up
        t0 = time.time()
        n = 0
        while n < 2 or (time.time() - t0 < budget and n < 10):
            orc.train_step(x, eps)
            n += 1
        return b * n / (time.time() - t0), n
    cores = min(os.cpu_count() or 1, 16)
    v_all, n_all = leg(cores, seconds_budget / 2)
    v_8, _ = leg(min(8, cores), seconds_budget / 2)
    return {"value": v_all, "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": _cpu_model(),
            "value_8_threads": v_8,
            "sample": f"{n_all} full train steps (SN + fwd + ELBO + bwd + Adamax) of the same model at batch {b}, f32, "
                      f"PyTorch-CPU restatement of the reference (proxy for TF-CPU)"}


class LaunchCensus:
    """Counts the C-ABI calls of one captured training step by entry point (the number of kernel launches of a
    step is a property of the host code, so it is measured here, not read from a profile)."""

    def __init__(self):
        from nvae_tf_amd import _lib as L
        from nvae_tf_amd import ops, models
        self.counts, self.mods, self.orig = {}, (L, ops, models.L), L.call
        self.convs, self.dw_elems = [], 0

    def __enter__(self):
        def counting(name, *a):
            self.counts[name] = self.counts.get(name, 0) + 1
            if name in ("nvae_conv_gemm", "nvae_conv_gemm_ex", "nvae_conv_gemm_bnbwd", "nvae_conv_direct"):
                g = a[1]._obj                                # the NvaeConvGeom passed by reference
                self.convs.append((name, g.B, g.Hin, g.Cin, g.Hout, g.Wout, g.Cout, g.KH, g.KW))
            elif name in ("nvae_dwconv5", "nvae_dwconv5_pre", "nvae_dwconv5_stats"):
                self.dw_elems += (a[5] * a[6] * a[7] * a[8]) if name != "nvae_dwconv5_pre" else (a[7] * a[8] * a[9] * a[10])
            return self.orig(name, *a)
        for m in self.mods:
            m.call = counting
        return self

    def __exit__(self, *exc):
        for m in self.mods:
            m.call = self.orig


def fwd_macs_per_image(model, x):
    """Forward multiply-accumulates per image, counted from the geometry of every convolution launch of one
    inference pass of the module tree (dense convs: B*Ho*Wo*KH*KW*Cin*Cout; depthwise 5x5: 25 per element) - the
    same replay SURVEY 8d did by hand for the MNIST configurations (6 949.4 M at C2), applied to whatever model runs."""
    with LaunchCensus() as c:
        model(x)
    torch.cuda.synchronize()
    B = x.shape[0]
    dense = sum(b * ho * wo * kh * kw * ci * co for (_, b, _, ci, ho, wo, co, kh, kw) in c.convs)
    return {"dense_conv": dense / B, "depthwise": 25.0 * c.dw_elems / B}


def side_workload(args):
    """Throughput of one of the other BASELINE.json configurations on one GPU (no roofline / CPU legs)."""
    from nvae_tf_amd import configs
    device = torch.device("cuda:0")
    torch.cuda.set_device(device)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    c = configs.CONFIGS[args.workload]
    batch = args.batch if args.batch != BATCH_PER_GPU else c["batch"]
    model = configs.build(args.workload, batch=batch, device=device, dtype=dtype,
                           loss_scale=None if args.loss_scale in (None, "dynamic") else float(args.loss_scale))
    H, W, Cc = c["input_hwc"]
    if Cc == 1:
        x = synthetic_batch(batch, 1, device)
    else:
        from nvae_tf_amd.datasets import synthetic_rgb
        x = torch.from_numpy(synthetic_rgb(batch, H, 1)[0]).float().div_(255.0).to(device)
    if args.no_graph:
        step = lambda: model.train_step(x)
    else:
        model.capture_train_step(x.shape, warmup=1)
        model._static_x.copy_(x.to(model._static_x.dtype))
        step = lambda: model.train_step_graphed(None)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    macs = fwd_macs_per_image(model, x)
    flop_img = 6.0 * (macs["dense_conv"] + macs["depthwise"])      # training = forward + dgrad + wgrad
    value = batch * args.steps / dt
    res = {"metric": "train_images_per_sec", "value": value, "unit": "images/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": args.dtype, "data": "synthetic",
           "config": {"workload": args.workload, "global_batch": batch, "groups": c["n_groups_per_scale"],
                      "parameters": model.n_trainable(), "hip_graph": not args.no_graph},
           "loss_nats": float(out["loss"]),
           "fwd_mac_per_image": macs, "train_flop_per_image": flop_img,
           "e2e_mfma_frac": value * flop_img / 1e12 / (PEAK_BF16_TFLOPS if args.dtype != "f32" else 157.3),
           "roofline": None}
    if args.workload == "mnist_c1" and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(workload="mnist_c1")     # BASELINE.json configs[0] at its own batch of 32
    print(json.dumps(res))


def time_dp_configured(args, device, dtype, x, plain_step):
    """The step as ONE data-parallel rank runs it, on one GPU: GradReducer(force=True) over a single-rank RCCL group
    (every collective is issued and runs through RCCL's kernels on its own stream), backward cut into the DP
    segments, each its own hipGraph, bucketed asynchronous all-reduce after each.  What it cannot show is the
    xGMI transfer time itself; what it does show is everything a rank pays before any byte moves.  Plain and
    DP-configured steps are timed in interleaved rounds in this one process (cdna guide rule 24); medians reported."""
    import torch.distributed as dist
    from nvae_tf_amd import parallel
    own_group = not dist.is_initialized()
    if own_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 400))
        with parallel.stdout_to_stderr():
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            dist.all_reduce(torch.zeros(1, device=device)); torch.cuda.synchronize()
    try:
        m = make_model(device, dtype, args.batch)
        m.reducer = parallel.GradReducer(force=True, bucket_bytes=int(os.environ.get("NVAE_DP_BUCKET_MB", "64")) << 20)
        m.capture_train_step(x.shape, warmup=1)
        m._static_x.copy_(x.to(dtype))
        dp_step = lambda: m.train_step_graphed(None)
        # host time inside the reducer's calls (a collective that blocks the host shows up here)
        host = {"calls": 0, "s": 0.0}
        for name in ("allreduce_mean_", "start_allreduce_", "finish_allreduce_", "allreduce_grads_"):
            def wrap(fn):
                def timed_call(*a, **k):
                    t0 = time.perf_counter()
                    r = fn(*a, **k)
                    host["s"] += time.perf_counter() - t0
                    host["calls"] += 1
                    return r
                return timed_call
            setattr(m.reducer, name, wrap(getattr(m.reducer, name)))

        def timed(fn, n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                out = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3, out
        for _ in range(args.warmup):
            dp_step()
        rounds, n = 5, max(args.steps // 2, 4)
        plain, dp = [], []
        host["calls"], host["s"] = 0, 0.0
        for _ in range(rounds):
            plain.append(timed(plain_step, n)[0])
            t, out = timed(dp_step, n)
            dp.append(t)
        host_ms = host["s"] / (rounds * n) * 1e3
        med = lambda v: sorted(v)[len(v) // 2]
        segs = [m.grad_range(k) for k in range(m.n_segments())] if m._dp_segments() else [(0, int(m.ps.grads.numel()))]
        return {"ms_per_step": med(dp), "plain_ms_per_step": med(plain), "dp_overhead_ms": med(dp) - med(plain),
                "ratio": med(dp) / med(plain), "rounds_ms": {"plain": [round(v, 3) for v in plain], "dp": [round(v, 3) for v in dp]},
                "host_ms_inside_reducer_calls_per_step": round(host_ms, 3), "reducer_calls_per_step": host["calls"] / (rounds * n),
                "backward_segments": len(segs),
                "segment_gradient_mbytes": [round((hi - lo) * 4 / 1e6, 1) for lo, hi in segs],
                "allreduce_bytes_per_step": int(m.ps.grads.numel()) * 4, "loss_nats": float(out["loss"]),
                "collectives": "single-rank RCCL (ReduceOp.AVG), 64 MB buckets, asynchronous per segment"}
    finally:
        if own_group:
            dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH_PER_GPU, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32"])
    ap.add_argument("--loss-scale", type=str, default=None, help="f16: 'dynamic' (default) or a static factor")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-dp", action="store_true",
                    help="one GPU: also time the step exactly as a data-parallel rank runs it (five backward segment "
                         "graphs, single-rank RCCL all-reduce of every gradient bucket and of the KL statistic) and report "
                         "dp_overhead_ms = that step - the plain single-GPU step")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--workload", default="mnist_c2", choices=["mnist_c2", "mnist_c1", "cifar10", "celeba64"],
                    help="mnist_c2 (default) is the configuration BASELINE.json's metric is quoted on; the others "
                         "are the parity-test configurations (nvae_tf_amd/configs.py), timed for DESIGN.md only")
    args = ap.parse_args()
    if args.workload != "mnist_c2":
        return side_workload(args)

    from nvae_tf_amd import parallel
    rank, world, local = parallel.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback for the hot path)"
    device = torch.device(f"cuda:{local % max(torch.cuda.device_count(), 1)}")
    torch.cuda.set_device(device)
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype]
    import torch.distributed as dist

    model = make_model(device, dtype, args.batch)
    if world > 1:
        model.reducer = parallel.GradReducer()
        # identical replicas: broadcast rank 0's parameters and state
        dist.broadcast(model.ps.params, 0)
        dist.broadcast(model.ps.state, 0)
    x = synthetic_batch(args.batch, 1 + rank, device)

    use_graph = not args.no_graph
    census = LaunchCensus()
    if use_graph:
        try:
            model.capture_train_step(x.shape, warmup=1)
            model._static_x.copy_(x.to(dtype))
            with census:                  # one more (eager) step, counted; timed steps are graph replays of the same calls
                model.train_step(x)
        except Exception as e:      # keep the run alive (and say so in the JSON) rather than lose the measurement
            print(f"[bench] hipGraph capture failed on rank {rank}: {e!r}; falling back to eager launches", file=sys.stderr)
            use_graph = False
        if world > 1:               # all ranks must take the same path (the collectives must match)
            flag = torch.tensor([1 if use_graph else 0], device=device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            use_graph = bool(int(flag[0]))
    step = (lambda: model.train_step_graphed(None)) if use_graph else (lambda: model.train_step(x))

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    loss = float(out["loss"])

    dp_cfg = None
    if args.force_dp and world == 1:
        dp_cfg = time_dp_configured(args, device, dtype, x, step)
    if rank == 0:
        value = args.batch * world * args.steps / dt
        kern = time_dominant_kernel(model, args.batch)
        # weighted by what the step launches: per dense 5x5 layer one forward (statistics epilogue) and one data
        # gradient (BatchNorm-backward epilogue) - three layers at each of the two shapes
        avg_ms = sum(k["ms"] for k in kern) / len(kern)
        macs = fwd_macs_per_image(model, x)
        flops_per_launch = 2.0 * args.batch * 943.7184e6   # both 5x5 shapes: 943.7 M MAC per image
        achieved = flops_per_launch / avg_ms / 1e9
        traffic, traffic_src = None, None     # HBM-side bytes per launch: NOT live, from the PMC passes under profiles/
        prof_dir = os.path.join(ROOT, "profiles")
        pmc = sorted(f for f in os.listdir(prof_dir) if f.endswith("_pmc_dominant.json")) if os.path.isdir(prof_dir) else []
        if pmc:
            with open(os.path.join(prof_dir, pmc[-1])) as fh:
                traffic = json.load(fh).get("traffic_bytes_avg")
            traffic_src = "profiles/" + pmc[-1]
        # the family that dominates the step's TIME (small-M implicit GEMMs of the 4x4 / 8x8 towers): launches per
        # step counted live, kernel time per step from this round's committed rocprofv3 summary
        fam = sorted(f for f in os.listdir(prof_dir) if f.endswith("_bench_families.json")) if os.path.isdir(prof_dir) else []
        fam_json = None
        if fam:
            with open(os.path.join(prof_dir, fam[-1])) as fh:
                fam_json = json.load(fh)
        conv_calls = sum(v for k, v in census.counts.items() if k.startswith("nvae_conv_gemm"))
        peak_tf = PEAK_F32_TFLOPS if args.dtype == "f32" else PEAK_BF16_TFLOPS      # dense MFMA peak of the arithmetic type
        n_calls = sum(census.counts.values())
        res = {
            "metric": "train_images_per_sec", "value": value, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "MNIST 28x28 (zero-padded 32x32) NVAE paper defaults: groups [5,10], "
                                   "2 cells/group, 20 latents/group; full train step "
                                   "(SN + fwd + ELBO + bwd + Adamax)",
                       "global_batch": args.batch * world, "batch_per_gpu": args.batch,
                       "parallelism": f"dp{world}", "hip_graph": use_graph,
                       "deterministic": bool(__import__("nvae_tf_amd._lib", fromlist=["load"]).load().nvae_get_deterministic())},
            "loss_nats": loss,
            "fwd_mac_per_image": macs,      # counted from this run's launches; SURVEY 8d: 6 949.4 M (dense + depthwise)
            "e2e_mfma_frac": value / world * TRAIN_FLOP_PER_IMG / 1e12 / peak_tf,
            "roofline": {"bound": "mfma", "kernel": f"k_conv_halo<{args.dtype},192,5> (dense 5x5 implicit GEMM of Postprocess, fwd + dgrad)",
                         "achieved": achieved, "peak": peak_tf, "unit": "TFLOP/s",
                         "frac": achieved / peak_tf, "traffic": traffic,
                         "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, rocprofv3 --pmc)",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes": 0.5 * (57704448 + 102506496),
                         "avg_launch_ms": avg_ms, "shapes": kern,
                         "share_of_step": "this kernel is ~13-15 % of the step's time (12 launches); see time_dominant",
                         "time_dominant": {
                             "family": "k_conv_gemm2: small-M implicit GEMMs of the 4x4 / 8x8 towers (fwd + dgrad)",
                             "c_abi_calls_per_step": n_calls, "conv_gemm_calls_per_step": conv_calls,
                             "profile": fam_json, "profile_source": ("profiles/" + fam[-1]) if fam else None}},
            "hbm_kernels": time_hbm_kernels(model, args.batch),
        }
        if dp_cfg is not None:
            res["dp_configured"] = dp_cfg
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(workload="mnist_c1" if args.workload == "mnist_c1" else "mnist_c2")
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
