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
