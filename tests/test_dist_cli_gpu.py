"""GPU: (1) the data-parallel step with a real RCCL process group (single rank - one GPU per box):
hipGraph capture next to an initialised NCCL communicator, ReduceOp.AVG, async bucketed all-reduce
between graph replays; (2) the train.py command line end to end (train / test / sample modes)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(n_encoder_channels=16, n_decoder_channels=16, res_cells_per_group=1, n_preprocess_blocks=2,
           n_preprocess_cells=2, n_latent_per_group=20, n_groups_per_scale=[1, 1], n_postprocess_blocks=2,
           n_post_process_cells=2)


def _model(dev):
    from nvae_tf_amd.models import NVAE
    c = CFG
    return NVAE(c["n_encoder_channels"], c["n_decoder_channels"], c["res_cells_per_group"], c["n_preprocess_blocks"],
                c["n_preprocess_cells"], c["n_latent_per_group"], 2, c["n_groups_per_scale"], c["n_postprocess_blocks"],
                c["n_post_process_cells"], 0.01, 2, 10, 1000, True, [8, 32, 32, 1], device=dev, dtype=torch.float32, seed=3)


def test_rccl_single_rank_graph_step(lib, dev):
    import torch.distributed as dist
    from nvae_tf_amd.parallel import GradReducer
    from oracle.nvae_oracle import synthetic_batch
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29600 + os.getpid() % 300)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = synthetic_batch(8, seed=2).float()
        ref, dp = _model(dev), _model(dev)
        dp.ps.params.copy_(ref.ps.params); dp.ps.state.copy_(ref.ps.state)
        dp.reducer = GradReducer(bucket_bytes=1 << 20, force=True)       # several buckets
        dist.broadcast(dp.ps.params, 0)
        dp.capture_train_step(x.shape, warmup=1)
        dp.ps.params.copy_(ref.ps.params); dp.ps.state.copy_(ref.ps.state)
        dp.ps.adam_m.zero_(); dp.ps.adam_u.zero_(); dp.rng_counter.zero_(); ref.rng_counter.zero_()
        dp.steps = ref.steps = 50
        dp.opt_iterations = ref.opt_iterations = 0
        for _ in range(3):
            o_ref = ref.train_step(x)
            o_dp = dp.train_step_graphed(x)
        torch.cuda.synchronize()
        assert abs(float(o_ref["loss"]) - float(o_dp["loss"])) / abs(float(o_ref["loss"])) < 1e-4
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_forced_dp_step_equals_plain_step_c2(lib, dev, dtype):
    """The benchmarked model at the benchmarked size (C2, batch 128): the data-parallel step as ONE rank runs it - segmented
    backward graphs, single-rank RCCL collectives on every bucket and on the KL statistic - is the plain single-GPU step.
    f32: the two runs differ only by the order of their f32 atomics - same loss to 1e-4 over three graph-replayed steps,
    parameters equal except where a gradient at noise level changes sign (Adamax moves by lr * sign(g) on its first steps).
    bf16: DP also switches the depthwise BatchNorm prologue off (one more bf16 rounding of an activation), and the 15-group
    model at a random initialisation amplifies any rounding difference step over step (measured 1.4e-3 at step 1, 1.6e-2 at
    step 2 - the same spread as two bf16 runs of ONE configuration): the first step's loss within 5e-3."""
    import sys
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from nvae_tf_amd.parallel import GradReducer
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(29400 + os.getpid() % 300)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        x = bench.synthetic_batch(128, 1, dev)
        plain, dp = bench.make_model(dev, dtype, 128), bench.make_model(dev, dtype, 128)
        dp.reducer = GradReducer(force=True)
        assert torch.equal(plain.ps.params, dp.ps.params)
        losses = []
        for m in (plain, dp):
            m.capture_train_step(x.shape, warmup=1)
            m._static_x.copy_(x.to(dtype))
            losses.append([float(m.train_step_graphed(None)["loss"]) for _ in range(3)])
        torch.cuda.synchronize()
        assert dp.n_segments() >= 2 and isinstance(dp._plan[1], list)
        d = (plain.ps.params - dp.ps.params).abs()
        q95 = float(torch.quantile(d[:: max(d.numel() // 1_000_000, 1)], 0.95))
        print("plain / forced-DP losses:", losses, "param diff max / q95:", float(d.max()), q95)
        if dtype == torch.float32:
            for a, b in zip(*losses):
                assert abs(a - b) / abs(a) < 1e-4, losses
            assert float(d.max()) < 9e-3 and q95 < 1e-4
        else:
            assert abs(losses[0][0] - losses[1][0]) / abs(losses[0][0]) < 5e-3, losses
            assert float(d.max()) < 9e-3 and q95 < 4e-3
    finally:
        dist.destroy_process_group()


def _dp_worker(rank, world, port, q):
    """One of two data-parallel ranks sharing the box's single GPU (gloo carries the collectives; the
    RCCL path needs one GPU per rank and is covered single-rank above and by the driver's N>1 runs)."""
    import torch.distributed as dist
    from nvae_tf_amd.parallel import GradReducer
    from oracle.nvae_oracle import synthetic_batch
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        x = synthetic_batch(8, seed=10 + rank).float()
        ref, dp = _model(dev), _model(dev)           # same seed -> identical replicas on every rank
        for m in (ref, dp):
            m.steps = 50
        # reference: plain backward, collectives issued by hand on whole buffers
        ref._set_hyper()
        ctx = ref._seg_forward(ref._as_input(x), None)
        dist.all_reduce(ref.am); ref.am.div_(world)
        ref._seg_backward(ctx, 8)
        torch.cuda.synchronize()
        g = ref.ps.grads.clone()
        dist.all_reduce(g); g.div_(world)
        # product path: segmented backward, bucketed async all-reduce per segment
        dp.reducer = GradReducer(bucket_bytes=1 << 18)
        assert dp._dp_segments()
        # (both replicas draw their noise from Philox counter 0 with the same seed)
        dp.train_step(x, update=False)
        torch.cuda.synchronize()
        err_eager = float((dp.ps.grads - g).abs().max() / g.abs().max())
        m = dp.param_marks
        assert m[0] == 0 and m[4] == dp.ps._p_cursor and m[1] < m[2] < m[3] < m[4]
        # graphed path: parameters stay identical across ranks after optimizer steps
        # (capture is side-effect free: the replicas, built from one seed, are still identical - no re-broadcast)
        dp.capture_train_step(x.shape, warmup=1)
        for _ in range(5):
            out = dp.train_step_graphed(x)
        torch.cuda.synchronize()
        mine = dp.ps.params.clone()
        other = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(other, mine)
        drift = float((other[0] - other[1]).abs().max())
        # the non-trainable state is rank-local (every rank normalises its own shard): the BatchNorm moving statistics
        # differ between the ranks until sync_state() replaces them by their mean
        st = [torch.empty_like(dp.ps.state) for _ in range(world)]
        dist.all_gather(st, dp.ps.state.clone())
        assert float((st[0] - st[1]).abs().max()) > 0.0
        want = (st[0] + st[1]) / world
        dp.sync_state()
        dist.all_gather(st, dp.ps.state.clone())
        assert float((st[0] - st[1]).abs().max()) == 0.0 and float((st[0] - want).abs().max()) < 1e-6
        dp.sync_replicas()
        dist.all_gather(other, dp.ps.params.clone())
        assert float((other[0] - other[1]).abs().max()) == 0.0
        # local gradients differ between ranks (different data), so matching parameters prove the exchange
        q.put((rank, err_eager, drift, float(out["loss"])))
        dist.destroy_process_group()
    except Exception as e:      # surface the failure in the parent instead of hanging it
        import traceback
        q.put((rank, "error", traceback.format_exc(), str(e)))


def test_dp_two_ranks_one_gpu_gloo(lib, dev):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29900 + os.getpid() % 90
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=420) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[1] != "error", r[2]
        _, err_eager, drift, loss = r
        assert err_eager < 2e-5, r          # same sums, different f32 association (buckets / atomics)
        # identical averaged gradients + deterministic spectral norm (fixed summation order, no atomics) + elementwise
        # Adamax: the replicas stay BIT-identical over 5 graphed steps without any re-broadcast
        assert drift == 0.0, r
        assert loss == loss


def test_train_cli_modes(lib, dev, tmp_path, capsys):
    from nvae_tf_amd import train
    common = ["--synthetic", "--batch_size", "16", "--n_encoder_channels", "16", "--n_decoder_channels", "16",
              "--n_groups_per_scale", "1", "1", "--n_preprocess_cells", "2", "--n_postprocess_cells", "2",
              "--model_save_dir", str(tmp_path / "models"), "--sample_dir", str(tmp_path / "results"),
              "--tensorboard_log_dir", str(tmp_path / "logs"), "--dtype", "bf16", "--debug", "--step_based_warmup"]
    train.main(train.parse_args(["--mode", "train", "--epochs", "2", "--model_save_frequency", "1",
                                 "--sample_frequency", "1", "--patience", "5"] + common))
    assert (tmp_path / "models" / "epoch_final.pt").exists() and (tmp_path / "models" / "epoch_1.pt").exists()
    from nvae_tf_amd.util import read_events
    ev = read_events(str(next((tmp_path / "logs").glob("events.out.tfevents.*"))))
    assert {t for _, t, _ in ev} >= {"epoch_loss", "epoch_reconstruction_loss", "epoch_kl_loss", "epoch_bn_loss"}
    img_ev = read_events(str(next((tmp_path / "logs" / "images").glob("events.out.tfevents.*"))))
    # per logged epoch: 4 temperatures x 3 samples (evaluate.py:15-21) + 3 test reconstructions (evaluate.py:24-45)
    assert len(img_ev) == 2 * (4 * 3 + 3)
    assert {t.split("/")[0] for _, t, _ in img_ev} == {"t=0.7", "t=0.8", "t=0.9", "t=1.0", "test_reconstruction"}
    assert len(list((tmp_path / "results" / "epoch_0").iterdir())) == 16
    train.main(train.parse_args(["--mode", "test", "--epochs", "2", "--resume_from", "1", "--binary_eval"] + common))
    out = capsys.readouterr().out
    assert "Negative log likelihood" in out
    train.main(train.parse_args(["--mode", "sample", "--epochs", "2", "--resume_from", "1", "--n_samples", "16"] + common))
    for t in ("t_0.7", "t_0.8", "t_0.9", "t_1.0"):
        assert len(list((tmp_path / "results" / t).iterdir())) == 16


def test_train_cli_cifar10_synthetic(lib, dev, tmp_path, capsys):
    """--dataset cifar10 end to end on synthetic RGB data: train (graph replay), checkpoint, IWAE NLL, samples."""
    from nvae_tf_amd import train
    common = ["--dataset", "cifar10", "--synthetic", "--batch_size", "8", "--n_encoder_channels", "16",
              "--n_decoder_channels", "16", "--n_groups_per_scale", "3", "--n_preprocess_blocks", "1",
              "--n_postprocess_blocks", "1", "--n_preprocess_cells", "2", "--n_postprocess_cells", "2",
              "--model_save_dir", str(tmp_path / "models"), "--sample_dir", str(tmp_path / "results"),
              "--tensorboard_log_dir", str(tmp_path / "logs"), "--dtype", "bf16", "--debug", "--step_based_warmup"]
    train.main(train.parse_args(["--mode", "train", "--epochs", "1", "--model_save_frequency", "1",
                                 "--sample_frequency", "1"] + common))
    assert (tmp_path / "models" / "epoch_final.pt").exists()
    train.main(train.parse_args(["--mode", "test", "--epochs", "1"] + common))
    assert "Negative log likelihood" in capsys.readouterr().out
