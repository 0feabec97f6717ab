"""Helpers mirroring the reference's util.py: image tiling, sample -> PNG writers, metric records.
(softclamp5 / calculate_log_p live in the sampler kernel: csrc/loss.hip.)"""
from __future__ import annotations

import os
import struct
import uuid
import zlib
from dataclasses import dataclass
from typing import List

import numpy as np
import torch


def tile_images(images: torch.Tensor) -> torch.Tensor:
    """util.py:12-19: the first n*n images as an n x n grid, [n*H, n*W, C]."""
    n = int(np.floor(np.sqrt(images.shape[0])))
    _, h, w, c = images.shape
    g = images[:n * n].reshape(n, n, h, w, c).permute(2, 0, 3, 1, 4)
    return g.reshape(n * h, n * w, c)


def encode_png(img: np.ndarray) -> bytes:
    """Minimal PNG encoder (uint8 [H, W, 1|3]); replaces tf.io.encode_png (util.py:35)."""
    h, w, c = img.shape
    assert img.dtype == np.uint8 and c in (1, 3)
    raw = b"".join(b"\x00" + img[r].tobytes() for r in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0 if c == 1 else 2, 0, 0, 0)
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def save_images_to_dir(images: torch.Tensor, dir: str):
    """util.py:31-36."""
    if images.is_floating_point():
        images = (images.float() * 255).to(torch.uint8)
    arr = images.cpu().numpy()
    os.makedirs(dir, exist_ok=True)
    for im in arr:
        with open(os.path.join(dir, f"{uuid.uuid4()}.png"), "wb") as fh:
            fh.write(encode_png(im))


def sample_to_dir(model, batch_size, sample_size, temperature, output_dir, binary=False):
    """util.py:22-28."""
    for _ in range(max(sample_size // batch_size, 1)):
        images, *_ = model.sample(n_samples=batch_size, greyscale=not binary, temperature=temperature)
        save_images_to_dir(images, output_dir)


@dataclass
class Metric:               # util.py:53-60
    mean: float
    stddev: float

    @staticmethod
    def from_list(values):
        return Metric(mean=float(np.mean(values)), stddev=float(np.std(values)))


@dataclass
class ModelEvaluation:      # util.py:72-75 (FID / PPL / precision-recall are out of scope)
    nll: Metric
    sample_metrics: List = None
