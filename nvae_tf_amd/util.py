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


# ------------------------------------------------------------------------------------------
# TensorBoard event files without TensorFlow (train.py:20-45 logs scalars per epoch and image grids
# through tf.summary / callbacks.TensorBoard).  A TFRecord is  len:u64 | crc(len):u32 | data | crc(data):u32
# with masked CRC32-C; data is an `Event` protobuf, hand-encoded below (fields: wall_time=1 double,
# step=2 varint, file_version=3 string, summary=5 {value=1 {tag=1, simple_value=2 float, image=4
# {height=1, width=2, colorspace=3, encoded_image_string=4}}}).
# ------------------------------------------------------------------------------------------
def _crc32c_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TAB = _crc32c_table()


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c = _CRC_TAB[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _field(num: int, wire: int, payload: bytes) -> bytes:
    return _varint((num << 3) | wire) + payload


def _bytes_field(num: int, data: bytes) -> bytes:
    return _field(num, 2, _varint(len(data)) + data)


class EventWriter:
    """Minimal TensorBoard writer: scalars and PNG images into <logdir>/events.out.tfevents.*"""

    def __init__(self, logdir: str):
        import socket
        import time
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, f"events.out.tfevents.{int(time.time())}.{socket.gethostname()}.nvae")
        self._fh = open(self.path, "wb")
        self._event(0, _bytes_field(3, b"brain.Event:2"))

    def _event(self, step: int, body: bytes):
        import time
        data = _field(1, 1, struct.pack("<d", time.time())) + _field(2, 0, _varint(int(step))) + body
        hdr = struct.pack("<Q", len(data))
        self._fh.write(hdr + struct.pack("<I", _masked_crc(hdr)) + data + struct.pack("<I", _masked_crc(data)))
        self._fh.flush()

    def add_scalar(self, tag: str, value: float, step: int):
        val = _bytes_field(1, tag.encode()) + _field(2, 5, struct.pack("<f", float(value)))
        self._event(step, _bytes_field(5, _bytes_field(1, val)))

    def add_image(self, tag: str, image: torch.Tensor, step: int):
        """image: [H, W, 1|3] float in [0, 1] or uint8."""
        if image.is_floating_point():
            image = (image.float().clamp(0, 1) * 255).to(torch.uint8)
        arr = image.cpu().numpy()
        h, w, c = arr.shape
        img = _field(1, 0, _varint(h)) + _field(2, 0, _varint(w)) + _field(3, 0, _varint(c)) + _bytes_field(4, encode_png(arr))
        val = _bytes_field(1, tag.encode()) + _bytes_field(4, img)
        self._event(step, _bytes_field(5, _bytes_field(1, val)))

    def close(self):
        self._fh.close()


def read_events(path: str):
    """Parse an event file back (checks both CRCs); yields (step, tag, value-or-None).  Test helper."""
    out = []
    with open(path, "rb") as fh:
        blob = fh.read()
    pos = 0

    def fields(buf):
        i = 0
        while i < len(buf):
            key = 0; shift = 0
            while True:
                b = buf[i]; i += 1
                key |= (b & 0x7F) << shift; shift += 7
                if not b & 0x80:
                    break
            num, wire = key >> 3, key & 7
            if wire == 0:
                v = 0; shift = 0
                while True:
                    b = buf[i]; i += 1
                    v |= (b & 0x7F) << shift; shift += 7
                    if not b & 0x80:
                        break
                yield num, v
            elif wire == 1:
                yield num, buf[i:i + 8]; i += 8
            elif wire == 5:
                yield num, buf[i:i + 4]; i += 4
            else:
                ln = 0; shift = 0
                while True:
                    b = buf[i]; i += 1
                    ln |= (b & 0x7F) << shift; shift += 7
                    if not b & 0x80:
                        break
                yield num, buf[i:i + ln]; i += ln
    while pos < len(blob):
        hdr = blob[pos:pos + 8]
        (n,) = struct.unpack("<Q", hdr)
        assert struct.unpack("<I", blob[pos + 8:pos + 12])[0] == _masked_crc(hdr)
        data = blob[pos + 12:pos + 12 + n]
        assert struct.unpack("<I", blob[pos + 12 + n:pos + 16 + n])[0] == _masked_crc(data)
        pos += 16 + n
        step, summary = 0, None
        for num, v in fields(data):
            if num == 2:
                step = v
            elif num == 5:
                summary = v
        if summary is None:
            continue
        for num, v in fields(summary):
            if num != 1:
                continue
            tag, val = None, None
            for n2, v2 in fields(v):
                if n2 == 1:
                    tag = v2.decode()
                elif n2 == 2:
                    val = struct.unpack("<f", v2)[0]
            out.append((step, tag, val))
    return out

