"""CPU tests of the host side: the C-ABI library loads and exports every entry point the header
declares, the module tree reproduces the reference's parameter totals and schedules, and the
data-parallel reducer works over gloo with world_size 2."""
import math
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "nvae_hip.h")).read()
    declared = set(re.findall(r"\b(nvae_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 40
    from nvae_tf_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    m = re.search(r"#define NVAE_ABI_VERSION (\d+)", hdr)
    assert lib.nvae_abi_version() == int(m.group(1)) == _lib.ABI_VERSION
    assert lib.nvae_reduce_splits(131072, 192) >= 1


def test_stale_library_is_refused(monkeypatch):
    """A libnvae_hip.so built from older sources (it is git-ignored and travels out of band) must not load
    silently against newer Python signatures."""
    from nvae_tf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(RuntimeError, match="stale"):
        _lib.load()


def test_deterministic_switch_and_slab_rows(lib, monkeypatch):
    """NVAE_DETERMINISTIC=1 is read when the library is loaded; with it the slab-row queries return one row per producing
    workgroup instead of one per 64 (host logic only: no kernel runs)."""
    import ctypes as C
    from nvae_tf_amd import _lib
    g = _lib.ConvGeom(128, 32, 32, 32, 32, 32, 192, 1, 1, 1, 0, 0, 1, 0, 32, 192, 192)     # 512 M-tiles of 256 rows
    try:
        assert lib.nvae_get_deterministic() == 0
        r0 = lib.nvae_conv_gemm_stats_rows(_lib.BF16, C.byref(g))
        s0 = lib.nvae_se_fused_rows(128, 16, 256)
        monkeypatch.setattr(_lib, "_lib", None)
        monkeypatch.setenv("NVAE_DETERMINISTIC", "1")
        lib2 = _lib.load()
        assert lib2.nvae_get_deterministic() == 1
        r1 = lib2.nvae_conv_gemm_stats_rows(_lib.BF16, C.byref(g))
        s1 = lib2.nvae_se_fused_rows(128, 16, 256)
        assert r0 == 8 and r1 == 512 and s1 > s0 >= 1
    finally:
        lib.nvae_set_deterministic(0)


def make(groups, cells, dtype=torch.bfloat16, **kw):
    from nvae_tf_amd.models import NVAE
    return NVAE(32, 32, cells, 2, 3, 20, len(groups), groups, 2, 3, 0.01, 2, 400, 1000, True, [1, 32, 32, 1],
                device="cpu", dtype=dtype, **kw)


def test_module_tree_matches_reference_parameter_totals(lib):
    for groups, cells, want in (([5, 10], 1, 40128893), ([1, 1], 1, 17243405), ([5, 10], 2, 62225021)):
        m = make(groups, cells)
        assert m.n_trainable() == want
    m = make([5, 10], 1)
    assert len(m.ps.bn_loss_layers) == 88 and len(m.ps.convs) == 163
    assert m.eps_shapes(3) == [(3, 4, 4, 20)] * 10 + [(3, 8, 8, 20)] * 5
    # names line up with the oracle's (weights are exchanged by name)
    from oracle.nvae_oracle import OracleConfig, OracleNVAE
    orc = OracleNVAE(OracleConfig(), dtype=torch.float32)
    assert set(orc.s.params) == set(m.ps.slots) and set(orc.s.state) == set(m.ps.sslots)
    for k, v in orc.s.params.items():
        assert tuple(v.shape) == m.ps.slots[k].shape, k


def test_schedules_and_alphas(lib):
    m = make([5, 10], 1)
    assert m.alphas.tolist() == [1.0] * 10 + [8.0] * 5
    assert m.calculate_kl_alphas(2, [1, 1]).tolist() == [1.0, 4.0]
    m.steps = 0
    assert m.beta() == 0
    m.steps = 150
    assert abs(m.beta() - 0.5) < 1e-12
    m.steps = 10 ** 6
    assert m.beta() == 1
    assert abs(m.learning_rate(0) - 1e-3) < 1e-15 and abs(m.learning_rate(500) - 5e-4) < 1e-12
    assert m.learning_rate(5000) < 1e-12
    from nvae_tf_amd.ops import same_pad
    assert same_pad(32, 3, 2) == (0, 1) and same_pad(8, 5, 1) == (2, 2) and same_pad(31, 1, 2) == (0, 0)


@pytest.mark.parametrize("nseg,last_mb", [(2, 120), (3, 120), (4, 56), (5, 30)])
def test_backward_segments_tile_the_gradient_buffer(lib, monkeypatch, nseg, last_mb):
    """Data-parallel backward segmentation (models._make_segments, NVAE_DP_SEGMENTS = 2..5; default 4) at the C2
    architecture: the flat gradient ranges of the segments tile [0, P) in backward order, tape ranges tile the tape, and
    the all-reduce that is left exposed after the last backward kernel (the last segment) is bounded (<= 56 MB for the
    default, 30 MB with five segments)."""
    from nvae_tf_amd import models
    monkeypatch.setattr(models, "DP_SEGMENTS", nseg)
    m = make([5, 10], 2)
    enc = m.encoder
    assert len(enc.group_param_off) == len(enc.groups)
    enc.group_tape_idx = [100 + 7 * i for i in range(len(enc.groups))]      # stand-in for a recorded forward pass
    enc_mark, dec_mark, end = 100 + 7 * len(enc.groups) + 3, 2000, 2600
    segs = m._make_segments(enc_mark, dec_mark, end)
    assert len(segs) == nseg
    if nseg >= 3:
        assert segs[0][:2] == (dec_mark, end) and segs[1][:2] == (enc_mark, dec_mark)
    hi_t, hi_p = end, m.param_marks[4]
    for lo_t, t_hi, lo_p, p_hi in segs:
        assert t_hi == hi_t and p_hi == hi_p and lo_t < t_hi and lo_p < p_hi
        hi_t, hi_p = lo_t, lo_p
    assert hi_t == 0 and hi_p == 0
    last = segs[-1]
    assert (last[3] - last[2]) * 4 <= last_mb * 2 ** 20, (last[3] - last[2]) * 4
    # a cut sits on an encoder group boundary
    if nseg >= 4:
        assert all(s[2] in enc.group_param_off for s in segs[2:-1])


def test_missing_library_fails_loudly(monkeypatch):
    from nvae_tf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libnvae_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def _reducer_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nvae_tf_amd.parallel import GradReducer
    red = GradReducer(bucket_bytes=4096)     # many buckets
    n = 10_000 + 3
    flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
    red.allreduce_grads_(flat)
    am = torch.tensor([1.0, 2.0, 3.0]) * (rank + 1)
    red.allreduce_mean_(am)
    buckets = red.buckets(n)
    q.put((rank, flat[:5].tolist(), float(flat.sum()), am.tolist(), buckets[0].stop, buckets[-1].start,
           sum(b.stop - b.start for b in buckets)))
    dist.destroy_process_group()


def test_grad_reducer_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_reducer_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n = 10_003
    for rank, head, tot, am, first_stop, last_start, covered in res:
        assert head == [0.0, 1.5, 3.0, 4.5, 6.0]                 # mean of 1x and 2x
        assert abs(tot - 1.5 * n * (n - 1) / 2) / tot < 1e-6
        assert am == [1.5, 3.0, 4.5]
        assert first_stop == n and last_start == 0 and covered == n   # end of the buffer first


def test_torch_library_registration():
    """`torch.ops.nvae.*` exist, their fake implementations give the output shapes, and there is no CPU kernel behind
    them (the dispatcher raises instead of falling back)."""
    import nvae_tf_amd.torch_ops      # noqa: F401
    ops = torch.ops.nvae
    x = torch.empty(2, 8, 8, 32, device="meta", dtype=torch.bfloat16)
    w = torch.empty(3, 3, 32, 64, device="meta")
    assert ops.conv2d_same(x, w, None).shape == (2, 8, 8, 64) and ops.conv2d_same(x, w, None).dtype == torch.bfloat16
    assert ops.dwconv5(x, torch.empty(5, 5, 32, device="meta"), torch.empty(32, device="meta")).shape == x.shape
    y, mean, invstd = ops.bn_act(x, torch.empty(32, device="meta"), torch.empty(32, device="meta"), 1, 1e-5)
    assert y.shape == x.shape and mean.shape == (32,) and invstd.shape == (32,)
    out = ops.se_residual(x, x, torch.empty(32, 4, device="meta"), torch.empty(4, device="meta"),
                          torch.empty(4, 32, device="meta"), torch.empty(32, device="meta"), 1.0, 0.1)
    assert out[0].shape == x.shape and out[2].shape == (2, 32) and out[3].shape == (2, 4)
    assert ops.bernoulli_nll(torch.empty(2, 32, 32, 1, device="meta"), torch.empty(2, 32, 32, 1, device="meta")).shape == (2,)
    with pytest.raises(NotImplementedError):
        ops.conv2d_same(torch.zeros(2, 8, 8, 32), torch.zeros(3, 3, 32, 64), None)


def test_weight_preparation_parts_tile_the_conv_list(lib):
    """ParamStore.prepare_weights(part=k): the four per-module descriptor tables (preprocess | encoder | decoder |
    postprocess) cover every conv exactly once, their workgroup numbering restarts at 0 in each part and is otherwise
    the global numbering, and float16 models default to the device-side dynamic loss scale."""
    import ctypes as C
    from nvae_tf_amd import _lib as L
    m = make([5, 10], 2)
    ps = m.ps
    assert ps.n_prep_parts() == 4 and ps.prep_marks[0] == 0 and ps.prep_marks[-1] == len(ps.convs)
    full = (L.ConvDesc * len(ps.convs)).from_buffer_copy(ps.descs.numpy().tobytes())
    reb = (L.ConvDesc * len(ps.convs)).from_buffer_copy(ps.descs_parts.numpy().tobytes())
    n_total = blocks_total = 0
    for k, (off, n, blocks) in enumerate(ps._parts):
        i0 = off // C.sizeof(L.ConvDesc)
        assert i0 == ps.prep_marks[k] and n == ps.prep_marks[k + 1] - ps.prep_marks[k] and n > 0
        assert reb[i0].blk_off == 0
        for i in range(i0, i0 + n):
            assert reb[i].blk_off == full[i].blk_off - full[i0].blk_off
            assert (reb[i].w_off, reb[i].t_off, reb[i].p_off, reb[i].idx) == (full[i].w_off, full[i].t_off, full[i].p_off, full[i].idx)
        last = i0 + n - 1
        assert blocks == reb[last].blk_off + (full[last].K + 15) // 16
        n_total += n; blocks_total += blocks
    assert n_total == len(ps.convs) and blocks_total == ps.sn_blocks
    assert not m.dynamic_loss_scale and m.loss_scale == 1.0
    m16 = make([1, 1], 1, dtype=torch.float16)
    assert m16.dynamic_loss_scale and float(m16.hyper[L.HY_LSCALE]) == 2.0 ** -8 and float(m16.hyper[L.HY_GSCALE]) == 256.0
    assert not make([1, 1], 1, dtype=torch.float16, loss_scale=4.0).dynamic_loss_scale
