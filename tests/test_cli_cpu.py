"""CPU tests of the caller-side pieces: CLI surface (flag names/defaults of train.py:145-297), local
MNIST reader / synthetic data, PNG writer, image tiling."""
import struct
import zlib

import numpy as np
import torch


def test_cli_flags_match_reference_defaults():
    from nvae_tf_amd.train import parse_args
    a = parse_args(["--mode", "train"])
    assert (a.epochs, a.batch_size, a.n_encoder_channels, a.n_decoder_channels) == (400, 144, 32, 32)
    assert (a.res_cells_per_group, a.n_preprocess_blocks, a.n_preprocess_cells) == (1, 2, 3)
    assert (a.n_postprocess_blocks, a.n_postprocess_cells, a.n_latent_per_group) == (2, 3, 20)
    assert a.n_groups_per_scale == [5, 10] and a.sr_lambda == 0.01 and a.scale_factor == 2 and a.seed == 1
    assert a.model_save_frequency == 10 and a.sample_frequency == 5 and a.resume_from == 0
    b = parse_args(["--mode", "sample", "--n_groups_per_scale", "2", "3"])
    assert b.n_groups_per_scale == [2, 3]            # ints, not strings


def test_mnist_reader_and_synthetic(tmp_path):
    from nvae_tf_amd.datasets import load_mnist
    img = np.zeros((5, 28, 28), np.uint8)
    img[:, 3, 4] = 7
    img[:, 10, 10] = 255
    lab = np.arange(5, dtype=np.uint8)
    for stem, arr in (("train-images-idx3-ubyte", img), ("t10k-images-idx3-ubyte", img)):
        with open(tmp_path / stem, "wb") as fh:
            fh.write(struct.pack(">IIII", 0x0803, 5, 28, 28) + arr.tobytes())
    for stem in ("train-labels-idx1-ubyte", "t10k-labels-idx1-ubyte"):
        with open(tmp_path / stem, "wb") as fh:
            fh.write(struct.pack(">II", 0x0801, 5) + lab.tobytes())
    tr, te = load_mnist(2, binary=True, data_dir=str(tmp_path))
    assert len(tr) == 3
    x, y = next(iter(tr))
    assert x.shape == (2, 32, 32, 1) and y.tolist() == [0, 1]
    assert float(x[0, 5, 6, 0]) == 1.0 and float(x[0, 12, 12, 0]) == 1.0 and float(x.sum()) == 4.0   # pixel > 0, padded by 2
    tr2, _ = load_mnist(2, binary=False, data_dir=str(tmp_path))
    x2, _ = next(iter(tr2))
    assert abs(float(x2[0, 5, 6, 0]) - 7 / 255) < 1e-7
    s, _ = load_mnist(8, synthetic=True, synthetic_sizes=(64, 16))
    xs, _ = next(iter(s))
    assert xs.shape == (8, 32, 32, 1) and float(xs[:, :2].sum()) == 0 and 0.1 < float(xs[:, 2:30, 2:30].mean()) < 0.3


def test_png_and_tiling(tmp_path):
    from nvae_tf_amd.util import encode_png, save_images_to_dir, tile_images
    im = (np.arange(32 * 32).reshape(32, 32, 1) % 256).astype(np.uint8)
    png = encode_png(im)
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    # decode the IDAT back
    pos, idat = 8, b""
    while pos < len(png):
        n, = struct.unpack(">I", png[pos:pos + 4])
        tag = png[pos + 4:pos + 8]
        if tag == b"IDAT":
            idat += png[pos + 8:pos + 8 + n]
        pos += 12 + n
    raw = zlib.decompress(idat)
    rows = [raw[r * 33 + 1:(r + 1) * 33] for r in range(32)]
    assert np.array_equal(np.frombuffer(b"".join(rows), np.uint8).reshape(32, 32), im[:, :, 0])
    imgs = torch.arange(9 * 4 * 4).float().reshape(9, 4, 4, 1)
    t = tile_images(imgs)
    # the reference's perm [2,0,3,1,4] (util.py:17-18) interleaves: pixel (h, w) of image (i, j) lands
    # at [h*n + i, w*n + j]; reproduced as is
    assert t.shape == (12, 12, 1) and float(t[2 * 3 + 1, 3 * 3 + 2, 0]) == float(imgs[1 * 3 + 2, 2, 3, 0])
    save_images_to_dir(torch.rand(3, 8, 8, 1), str(tmp_path / "o"))
    assert len(list((tmp_path / "o").iterdir())) == 3


def test_rgb_readers_and_configs(tmp_path):
    """CIFAR-10 binary reader (planar records -> NHWC), synthetic RGB data, the named configurations of
    BASELINE.json configs[3]/[4] and their CLI flags."""
    from nvae_tf_amd import configs
    from nvae_tf_amd.datasets import load_celeba64, load_cifar10, synthetic_rgb
    from nvae_tf_amd.train import parse_args
    rec = np.zeros((4, 3073), np.uint8)
    rec[:, 0] = [3, 1, 4, 1]
    planes = np.arange(4 * 3072, dtype=np.uint32).reshape(4, 3, 32, 32) % 251
    rec[:, 1:] = planes.reshape(4, -1)
    for i in range(1, 6):
        rec.tofile(tmp_path / f"data_batch_{i}.bin")
    rec[:2].tofile(tmp_path / "test_batch.bin")
    tr, te = load_cifar10(8, data_dir=str(tmp_path))
    assert len(tr) == 3 and len(te) == 1
    x, y = next(iter(tr))
    assert x.shape == (8, 32, 32, 3) and y[:4].tolist() == [3, 1, 4, 1]
    assert abs(float(x[1, 5, 7, 2]) - float(planes[1, 2, 5, 7]) / 255.0) < 1e-7
    img, lab = synthetic_rgb(6, 64, seed=3)
    assert img.shape == (6, 64, 64, 3) and img.dtype == np.uint8 and img.min() == 0 and img.max() == 255
    trc, tec = load_celeba64(4, synthetic=True, synthetic_sizes=(8, 4))
    xb, _ = next(iter(trc))
    assert xb.shape == (4, 64, 64, 3) and 0.0 <= float(xb.min()) and float(xb.max()) <= 1.0
    assert sum(configs.CONFIGS["cifar10"]["n_groups_per_scale"]) == 30
    assert sum(configs.CONFIGS["celeba64"]["n_groups_per_scale"]) == 40
    assert configs.CONFIGS["mnist_c2"]["n_groups_per_scale"] == [5, 10] and configs.CONFIGS["mnist_c2"]["batch"] == 128
    a = parse_args(["--mode", "train", "--dataset", "cifar10", "--num_mixture_dec", "5"])
    assert a.dataset == "cifar10" and a.num_mixture_dec == 5


def test_oracle_dmol_known_answers():
    """Known answers of the mixture-of-logistics specification: a single very wide logistic gives
    every 8-bit bin probability ~1/256 per sub-pixel in the interior; the probabilities of all 256 bins sum to one;
    the gradient matches finite differences."""
    from oracle.nvae_oracle import dmol_log_prob, dmol_sample
    M = 2
    l = torch.zeros(1, 1, 1, 10 * M, dtype=torch.float64)
    l[..., 0] = 5.0                    # mixture 0 dominates
    blk = l[..., M:].reshape(1, 1, 1, 3, 3 * M)
    blk[..., 0, 0] = 0.3; blk[..., 1, 0] = -0.2; blk[..., 2, 0] = 0.1        # means
    blk[..., :, M] = -2.0                                                    # log-scales
    total = torch.zeros((), dtype=torch.float64)
    levels = torch.arange(256, dtype=torch.float64) / 255.0
    x = torch.zeros(256, 1, 1, 3, dtype=torch.float64)
    x[:, 0, 0, 0] = levels
    x[:, 0, 0, 1:] = 0.5
    lp = dmol_log_prob(x, l.expand(256, 1, 1, -1), M)
    # marginal over the red sub-pixel: divide out the (constant) green/blue factors by normalising
    p_red = torch.exp(lp - torch.logsumexp(lp, 0))
    assert abs(float(p_red.sum()) - 1.0) < 1e-12
    xr = torch.full((1, 1, 1, 3), 100.0 / 255.0, dtype=torch.float64)
    lg = (torch.randn(1, 1, 1, 10 * M, generator=torch.Generator().manual_seed(1), dtype=torch.float64)).requires_grad_(True)
    (g,) = torch.autograd.grad(dmol_log_prob(xr, lg, M).sum(), lg)
    e = torch.zeros_like(lg); e[..., 7] = 1e-6
    fd = (dmol_log_prob(xr, lg + e, M) - dmol_log_prob(xr, lg - e, M)).sum() / 2e-6
    assert abs(float(fd) - float(g[..., 7])) < 1e-6
    # sampling at a tiny temperature returns the dominant mixture's (coefficient-shifted) means
    u_mix = torch.full((1, 1, 1, M), 0.5, dtype=torch.float64)
    u_pix = torch.full((1, 1, 1, 3), 0.5, dtype=torch.float64)
    s = dmol_sample(l, M, u_mix, u_pix, t=1.0)
    assert torch.allclose(s[0, 0, 0], torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64) / 2 + 0.5)


def test_tensorboard_event_writer_roundtrip(tmp_path):
    """Event files written without TensorFlow: record framing, masked CRC32-C, protobuf fields."""
    from nvae_tf_amd.util import EventWriter, crc32c, read_events
    assert crc32c(b"123456789") == 0xE3069283          # the CRC-32C check value
    w = EventWriter(str(tmp_path))
    w.add_scalar("epoch_loss", 123.5, 3)
    w.add_scalar("epoch_kl_loss", -0.25, 4)
    w.add_image("grid", torch.rand(8, 8, 1), 4)
    w.close()
    ev = read_events(w.path)
    assert (3, "epoch_loss", 123.5) in ev and (4, "epoch_kl_loss", -0.25) in ev
    assert any(t == "grid" and v is None and s == 4 for s, t, v in ev)
