"""Input pipeline mirroring the reference's datasets.py, without tensorflow_datasets: MNIST is read
from local IDX / NPZ files (no network here), zero-padded 28 -> 32 (datasets.py:12) and binarised the
way the reference effectively does (Bernoulli(probs = raw 0..255 pixel) == pixel > 0).
A synthetic generator of the same shape backs the benchmarks."""
from __future__ import annotations

import gzip
import os
import struct
from typing import Iterator, List, Optional, Tuple

import numpy as np
import torch


def _read_idx(path: str) -> np.ndarray:
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rb") as fh:
        magic, = struct.unpack(">I", fh.read(4))
        ndim = magic & 0xff
        dims = struct.unpack(">" + "I" * ndim, fh.read(4 * ndim))
        return np.frombuffer(fh.read(), dtype=np.uint8).reshape(dims)


def _find(data_dir: str, stems: List[str]) -> Optional[str]:
    for s in stems:
        for ext in ("", ".gz"):
            p = os.path.join(data_dir, s + ext)
            if os.path.exists(p):
                return p
    return None


class Batches:
    """Re-iterable list of (images [B,32,32,1] f32, labels) batches; len() = batches per epoch."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, batch_size: int, binary: bool):
        self.images, self.labels, self.batch_size, self.binary = images, labels, batch_size, binary

    def __len__(self):
        return (len(self.images) + self.batch_size - 1) // self.batch_size

    def take(self, n):
        return Batches(self.images[:n * self.batch_size], self.labels[:n * self.batch_size], self.batch_size, self.binary)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        for i in range(0, len(self.images), self.batch_size):    # no shuffle, like the reference
            raw = torch.from_numpy(self.images[i:i + self.batch_size].astype(np.float32))
            img = torch.zeros(raw.shape[0], 32, 32, 1)
            img[:, 2:30, 2:30, 0] = raw
            img = (img > 0).float() if self.binary else img / 255.0
            yield img, torch.from_numpy(self.labels[i:i + self.batch_size].astype(np.int64))


def synthetic_mnist(n: int, seed: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """uint8 [n,28,28] with ~19 % foreground (the MNIST foreground fraction after `> 0`)."""
    g = np.random.default_rng(seed)
    img = (g.random((n, 28, 28)) < 0.19).astype(np.uint8) * 255
    return img, g.integers(0, 10, n).astype(np.uint8)


def load_mnist(batch_size: int, binary: bool = True, data_dir: Optional[str] = None, synthetic: bool = False,
               synthetic_sizes=(60000, 10000)):
    """datasets.py:6-20 -> (train batches, test batches)."""
    data_dir = data_dir or os.environ.get("MNIST_DIR", "data/mnist")
    if not synthetic:
        npz = os.path.join(data_dir, "mnist.npz")
        if os.path.exists(npz):
            d = np.load(npz)
            tr, trl, te, tel = d["x_train"], d["y_train"], d["x_test"], d["y_test"]
        else:
            p = [_find(data_dir, [s]) for s in ("train-images-idx3-ubyte", "train-labels-idx1-ubyte",
                                                "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")]
            if any(x is None for x in p):
                raise FileNotFoundError(
                    f"MNIST not found under {data_dir} (mnist.npz or the four IDX files); there is no network "
                    "here - pass --synthetic for random MNIST-shaped data")
            tr, trl, te, tel = (_read_idx(x) for x in p)
    else:
        tr, trl = synthetic_mnist(synthetic_sizes[0], 1)
        te, tel = synthetic_mnist(synthetic_sizes[1], 2)
    return Batches(tr, trl, batch_size, binary), Batches(te, tel, batch_size, binary)


# ---- RGB data (BASELINE.json configs[3] / configs[4]; the reference has only stubs: datasets.py:23-25) ----
class RgbBatches:
    """Re-iterable (images [B,H,W,3] f32 = k/255, labels) batches from a uint8 [N,H,W,3] array."""

    def __init__(self, images: np.ndarray, labels: np.ndarray, batch_size: int):
        assert images.dtype == np.uint8 and images.ndim == 4 and images.shape[3] == 3
        self.images, self.labels, self.batch_size = images, labels, batch_size

    def __len__(self):
        return (len(self.images) + self.batch_size - 1) // self.batch_size

    def take(self, n):
        return RgbBatches(self.images[:n * self.batch_size], self.labels[:n * self.batch_size], self.batch_size)

    def __iter__(self) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
        for i in range(0, len(self.images), self.batch_size):
            img = torch.from_numpy(self.images[i:i + self.batch_size].astype(np.float32)) / 255.0
            yield img, torch.from_numpy(self.labels[i:i + self.batch_size].astype(np.int64))


def synthetic_rgb(n: int, hw: int, seed: int = 1) -> Tuple[np.ndarray, np.ndarray]:
    """uint8 [n,hw,hw,3]: smooth random colour fields plus noise, clipped so that 0 and 255 occur."""
    g = np.random.default_rng(seed)
    low = g.random((n, max(hw // 8, 1), max(hw // 8, 1), 3))
    img = np.repeat(np.repeat(low, 8, axis=1), 8, axis=2)[:, :hw, :hw]
    img = (img - 0.5) * 2.2 + 0.5 + 0.05 * g.standard_normal(img.shape)
    return np.round(np.clip(img, 0, 1) * 255).astype(np.uint8), g.integers(0, 10, n).astype(np.uint8)


def _read_cifar_bin(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """CIFAR-10 binary version: records of 1 label byte + 3072 pixel bytes (planar R, G, B)."""
    raw = np.fromfile(path, dtype=np.uint8)
    if raw.size % 3073:
        raise ValueError(f"{path}: size {raw.size} is not a multiple of 3073")
    rec = raw.reshape(-1, 3073)
    return rec[:, 1:].reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1).copy(), rec[:, 0].copy()


def load_cifar10(batch_size: int, data_dir: Optional[str] = None, synthetic: bool = False,
                 synthetic_sizes=(50000, 10000)):
    """(train, test) RGB batches from <data_dir>/cifar10.npz (x_train, y_train, x_test, y_test) or the
    binary distribution (data_batch_[1-5].bin, test_batch.bin).  Python-pickle batches are not read."""
    data_dir = data_dir or os.environ.get("CIFAR10_DIR", "data/cifar10")
    if synthetic:
        (tr, trl), (te, tel) = synthetic_rgb(synthetic_sizes[0], 32, 1), synthetic_rgb(synthetic_sizes[1], 32, 2)
        return RgbBatches(tr, trl, batch_size), RgbBatches(te, tel, batch_size)
    npz = os.path.join(data_dir, "cifar10.npz")
    if os.path.exists(npz):
        d = np.load(npz)
        return (RgbBatches(d["x_train"], d["y_train"].reshape(-1), batch_size),
                RgbBatches(d["x_test"], d["y_test"].reshape(-1), batch_size))
    parts = [_find(data_dir, [f"data_batch_{i}.bin"]) for i in range(1, 6)]
    test = _find(data_dir, ["test_batch.bin"])
    if test is None or any(x is None for x in parts):
        raise FileNotFoundError(f"CIFAR-10 not found under {data_dir} (cifar10.npz or the *.bin files); there is "
                                "no network here - pass --synthetic for random data of the same shape")
    tr = [_read_cifar_bin(x) for x in parts]
    te, tel = _read_cifar_bin(test)
    return (RgbBatches(np.concatenate([a for a, _ in tr]), np.concatenate([b for _, b in tr]), batch_size),
            RgbBatches(te, tel, batch_size))


def load_celeba64(batch_size: int, data_dir: Optional[str] = None, synthetic: bool = False,
                  synthetic_sizes=(4096, 512)):
    """(train, test) 64x64 RGB batches from <data_dir>/celeba64.npz (x_train, x_test uint8 [N,64,64,3]);
    BASELINE.json's configuration is "CelebA-64 synthetic", which `synthetic=True` provides."""
    data_dir = data_dir or os.environ.get("CELEBA64_DIR", "data/celeba64")
    if synthetic:
        (tr, trl), (te, tel) = synthetic_rgb(synthetic_sizes[0], 64, 1), synthetic_rgb(synthetic_sizes[1], 64, 2)
        return RgbBatches(tr, trl, batch_size), RgbBatches(te, tel, batch_size)
    npz = os.path.join(data_dir, "celeba64.npz")
    if not os.path.exists(npz):
        raise FileNotFoundError(f"{npz} not found; there is no network here - pass --synthetic")
    d = np.load(npz)
    z = lambda a: np.zeros(len(a), np.uint8)
    return RgbBatches(d["x_train"], z(d["x_train"]), batch_size), RgbBatches(d["x_test"], z(d["x_test"]), batch_size)
