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


def load_celeba():   # datasets.py:23-25 is an empty stub in the reference too
    raise NotImplementedError("the reference has no CelebA loader (datasets.py:23-25)")
