"""Command-line driver mirroring the reference's train.py: the same flags and defaults
(train.py:145-297), modes train / test / sample, Adamax + cosine LR, periodic checkpoints.

Differences, all deliberate (SURVEY "Quirks"): --n_groups_per_scale is parsed as ints (Q13); resume
restores the step counter as epochs * batches_per_epoch (Q14); data come from local files or
--synthetic; --dtype selects bf16 (default) or f32 compute; launched under torchrun the batch is
sharded over ranks (one process per GPU, RCCL gradient all-reduce)."""
from __future__ import annotations

import os
import random
import time
from argparse import ArgumentParser

import numpy as np
import torch


LOSS_SCALE_FLOOR = 2.0 ** -24     # lower bound of the dynamic loss scale (models.NVAE._seg_update)


def checkpoint_path(model_save_dir, epoch):
    return os.path.join(model_save_dir, f"epoch_{epoch}.pt")


def save_checkpoint(model, path, epoch):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    ps = model.ps
    torch.save({"params": ps.params.cpu(), "state": ps.state.cpu(), "adam_m": ps.adam_m.cpu(),
                "adam_u": ps.adam_u.cpu(), "steps": model.steps, "opt_iterations": model.opt_iterations,
                "epoch": epoch, "rng_counter": model.rng_counter.cpu(), "hyper": model.hyper.cpu()}, path)


def load_checkpoint(model, path):
    ck = torch.load(path, map_location="cpu", weights_only=True)
    ps = model.ps
    ps.params.copy_(ck["params"]); ps.state.copy_(ck["state"])
    ps.adam_m.copy_(ck["adam_m"]); ps.adam_u.copy_(ck["adam_u"])
    model.steps, model.opt_iterations = int(ck["steps"]), int(ck["opt_iterations"])
    model.rng_counter.copy_(ck["rng_counter"])
    if "hyper" in ck:                         # the loss-scale state of the f16 path lives in hyper[3:]
        model.hyper.copy_(ck["hyper"])
    return int(ck["epoch"])


def train(args, model, train_data, test_data, rank=0, world=1):
    from .util import EventWriter, sample_to_dir
    best, bad_epochs, best_state = float("inf"), 0, None
    use_graph = not args.no_graph
    captured = False
    # callbacks.TensorBoard(update_freq="epoch") + the image logger of train.py:20-45, written as real
    # event files (util.EventWriter) so `tensorboard --logdir` works without TensorFlow on the box
    tb = EventWriter(args.tensorboard_log_dir) if (args.tensorboard_log_dir and rank == 0) else None
    tb_img = EventWriter(os.path.join(args.tensorboard_log_dir, "images")) if tb is not None else None
    for epoch in range(args.resume_from, args.epochs):
        model.on_epoch_begin(epoch)
        if epoch == args.resume_from:
            model.sync_replicas()   # data-parallel start-up / resume: one broadcast; replicas stay bit-identical afterwards
        t0, seen, logs = time.time(), 0, {"loss": [], "reconstruction_loss": [], "kl_loss": [], "bn_loss": []}
        for i, (images, _) in enumerate(train_data):
            images = images[rank::world] if world > 1 else images       # shard the batch over ranks
            if use_graph and images.shape[0] == args.batch_size // world:
                if not captured:
                    model.capture_train_step(images.shape)
                    captured = True
                out = model.train_step_graphed(images)
            else:
                out = model.train_step(images)
            seen += images.shape[0] * world
            if args.verbose or args.debug or i % 50 == 0:
                for k in logs:
                    logs[k].append(float(out[k].float().mean()))
        dt = time.time() - t0
        means = {k: float(np.mean(v)) for k, v in logs.items()}
        model.sync_state()         # BatchNorm moving statistics are rank-local: average them before anyone reads them
        if model.dynamic_loss_scale and rank == 0:
            # a step whose gradient overflows is skipped ON THE DEVICE (nvae_adamax); the host-side schedule (cosine
            # learning rate, Adamax bias correction, KL warm-up) still advances on it, so say so when it happens a lot
            ls, good = model.loss_scale_report()
            print(f"epoch {epoch}: dynamic loss scale 2^{np.log2(ls):.0f}, {good} clean steps since it last changed")
            if ls <= LOSS_SCALE_FLOOR:
                print("WARNING: the loss scale sits at its floor - every step of this epoch overflowed in float16 and "
                      "was skipped; nothing is being trained (use --dtype bf16: same footprint and speed, f32's "
                      "exponent range)")
        if rank == 0:
            print(f"epoch {epoch}: loss {means['loss']:.3f}  {seen / dt:.1f} images/s  beta {model.beta():.3f}")
            if tb is not None:
                for k, v in means.items():
                    tb.add_scalar("epoch_" + k, v, epoch)
                tb.add_scalar("images_per_sec", seen / dt, epoch)
            if epoch % args.sample_frequency == 0:
                sample_to_dir(model, 16, 16, 1.0, os.path.join(args.sample_dir, f"epoch_{epoch}"))
                if tb_img is not None:               # train.py:23-26 of the reference
                    from .evaluate import save_reconstructions_to_tensorboard, save_samples_to_tensorboard
                    save_samples_to_tensorboard(epoch, model, tb_img)
                    save_reconstructions_to_tensorboard(epoch, model, test_data, tb_img)
            if epoch % args.model_save_frequency == 0:
                save_checkpoint(model, checkpoint_path(args.model_save_dir, epoch), epoch)
        if args.patience:          # EarlyStopping(patience, restore_best_weights=True) on the training loss (train.py:35-38)
            cur = means["loss"]
            if world > 1:          # every rank must take the same decision, or the survivors hang in the next collective
                import torch.distributed as dist
                t = torch.tensor([cur], dtype=torch.float64, device=model.device if dist.get_backend() == "nccl" else "cpu")
                dist.all_reduce(t)
                cur = float(t[0]) / world
            if cur < best:
                best, bad_epochs = cur, 0
                best_state = (model.ps.params.clone(), model.ps.state.clone())
            else:
                bad_epochs += 1
            if bad_epochs > args.patience:
                if best_state is not None:
                    model.ps.params.copy_(best_state[0]); model.ps.state.copy_(best_state[1])
                break
    if tb is not None:
        tb.close(); tb_img.close()
    if rank == 0:
        save_checkpoint(model, checkpoint_path(args.model_save_dir, "final"), args.epochs)


def test(args, model, test_data):
    from .evaluate import evaluate_model
    evaluation = evaluate_model(epoch=args.resume_from, model=model, test_data=test_data, n_attempts=10)
    print(f"Negative log likelihood: {evaluation.nll}")
    print(evaluation)


def sample(args, model):
    from .util import sample_to_dir
    for t in [0.7, 0.8, 0.9, 1]:                      # train.py:76-80
        output_dir = os.path.join(args.sample_dir, f"t_{t:.1f}")
        os.makedirs(output_dir, exist_ok=True)
        sample_to_dir(model, args.batch_size, args.n_samples, t, output_dir)


def main(args):
    from . import parallel
    from .datasets import load_celeba64, load_cifar10, load_mnist
    from .models import NVAE
    print(f"Args: {args}")
    rank, world, local = parallel.init_from_env()
    if args.cpu or not torch.cuda.is_available():
        raise SystemExit("the NVAE hot path runs on an MI355X through libnvae_hip.so; there is no CPU path "
                         "(the CPU oracle under oracle/ is test infrastructure only)")
    local = local % max(torch.cuda.device_count(), 1)     # several gloo ranks may share a GPU (NVAE_DIST_BACKEND)
    torch.cuda.set_device(local)
    torch.manual_seed(args.seed); random.seed(args.seed); np.random.seed(args.seed)
    if args.dataset == "mnist":
        train_data, test_data = load_mnist(args.batch_size, binary=args.mode == "train" or args.binary_eval,
                                           data_dir=args.data_dir, synthetic=args.synthetic)
        hwc = [32, 32, 1]
    elif args.dataset == "cifar10":
        train_data, test_data = load_cifar10(args.batch_size, data_dir=args.data_dir, synthetic=args.synthetic)
        hwc = [32, 32, 3]
    else:
        train_data, test_data = load_celeba64(args.batch_size, data_dir=args.data_dir, synthetic=args.synthetic)
        hwc = [64, 64, 3]
    if args.debug:
        train_data, test_data = train_data.take(4), test_data.take(4)
    batches_per_epoch = len(train_data)
    model = NVAE(n_encoder_channels=args.n_encoder_channels, n_decoder_channels=args.n_decoder_channels,
                 res_cells_per_group=args.res_cells_per_group, n_preprocess_blocks=args.n_preprocess_blocks,
                 n_preprocess_cells=args.n_preprocess_cells, n_latent_per_group=args.n_latent_per_group,
                 n_latent_scales=len(args.n_groups_per_scale), n_groups_per_scale=args.n_groups_per_scale,
                 n_postprocess_blocks=args.n_postprocess_blocks, n_post_process_cells=args.n_postprocess_cells,
                 sr_lambda=args.sr_lambda, scale_factor=args.scale_factor, total_epochs=args.epochs,
                 n_total_iterations=batches_per_epoch * args.epochs, step_based_warmup=args.step_based_warmup,
                 input_shape=[args.batch_size // world] + hwc, device=f"cuda:{local}",
                 num_mixture_dec=args.num_mixture_dec,
                 dtype={"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[args.dtype],
                 loss_scale=(None if args.loss_scale is None else args.loss_scale if args.loss_scale == "dynamic"
                             else float(args.loss_scale)), seed=args.seed + rank,
                 lr_decay_steps=args.epochs * batches_per_epoch)
    model.tf_literal = args.tf_literal
    if world > 1:
        import torch.distributed as dist
        model.reducer = parallel.GradReducer()
        dist.broadcast(model.ps.params, 0); dist.broadcast(model.ps.state, 0)
    if args.resume_from > 0:
        load_checkpoint(model, checkpoint_path(args.model_save_dir, args.resume_from))
        model.steps = args.resume_from * batches_per_epoch      # Q14 fixed (reference: * batch_size)
    if args.mode == "train":
        train(args, model, train_data, test_data, rank, world)
    elif args.mode == "test":
        test(args, model, test_data)
    elif args.mode == "sample":
        sample(args, model)


def parse_args(argv=None):
    p = ArgumentParser()
    p.add_argument("--epochs", type=int, default=400, help="Number of epochs to train")
    p.add_argument("--batch_size", default=144, type=int)
    p.add_argument("--mode", type=str, choices=["train", "test", "sample"])
    p.add_argument("--n_encoder_channels", type=int, default=32)
    p.add_argument("--n_decoder_channels", type=int, default=32)
    p.add_argument("--res_cells_per_group", type=int, default=1)
    p.add_argument("--n_preprocess_blocks", type=int, default=2)
    p.add_argument("--n_preprocess_cells", type=int, default=3)
    p.add_argument("--n_postprocess_blocks", type=int, default=2)
    p.add_argument("--n_postprocess_cells", type=int, default=3)
    p.add_argument("--n_latent_per_group", type=int, default=20)
    p.add_argument("--n_groups_per_scale", nargs="+", type=int, default=[5, 10])
    p.add_argument("--sr_lambda", type=float, default=0.01, help="Spectral regularisation strength")
    p.add_argument("--scale_factor", type=int, default=2)
    p.add_argument("--dataset", type=str, choices=["mnist", "cifar10", "celeba64"], default="mnist",
                   help="mnist is the reference's only data set; cifar10 / celeba64 use the RGB "
                        "mixture-of-logistics head (nvae_tf_amd/configs.py lists the paper's shapes)")
    p.add_argument("--num_mixture_dec", type=int, default=10, help="Logistic mixtures of the RGB output head")
    p.add_argument("--cpu", action="store_true", help="(reference flag) not supported: no CPU path")
    p.add_argument("--debug", action="store_true", help="Use only the first four batches of data")
    p.add_argument("--n_samples", type=int, default=10)
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--model_save_dir", type=str, default="models")
    p.add_argument("--sample_dir", type=str, default="results")
    p.add_argument("--resume_from", type=int, default=0, help="Epoch to resume training from")
    p.add_argument("--tensorboard_log_dir", type=str, default="logs")
    p.add_argument("--sample_frequency", type=int, default=5)
    p.add_argument("--evaluate_frequency", type=int, default=10)
    p.add_argument("--log_frequency", type=int, default=1)
    p.add_argument("--binary_eval", action="store_true", help="Evaluate on binary data")
    p.add_argument("--patience", type=int)
    p.add_argument("--model_save_frequency", type=int, default=10)
    p.add_argument("--step_based_warmup", action="store_true")
    p.add_argument("--workers", type=int, default=1)
    p.add_argument("--multiprocessing", action="store_true")
    p.add_argument("--seed", type=int, default=1)
    # additions of this build
    p.add_argument("--dtype", choices=["bf16", "f16", "f32"], default="bf16", help="activation dtype of the HIP path")
    p.add_argument("--loss_scale", type=str, default=None,
                   help="f16 activations: 'dynamic' (default for f16: skip + halve on overflow, double after 200 clean "
                        "steps, on the device) or a static factor; undone inside Adamax")
    p.add_argument("--data_dir", type=str, default=None, help="directory with mnist.npz or the IDX files")
    p.add_argument("--synthetic", action="store_true", help="random MNIST-shaped data (no files needed)")
    p.add_argument("--no_graph", action="store_true", help="launch eagerly instead of replaying hipGraphs")
    p.add_argument("--tf_literal", action="store_true",
                   help="reference-literal semantics (SURVEY Q1): BatchNorm with moving statistics and no "
                        "spectral normalisation during training")
    return p.parse_args(argv)


if __name__ == "__main__":
    main(parse_args())
