"""Flat parameter storage.

All trainable parameters live in ONE f32 device buffer (`params`), their gradients in a matching
buffer (`grads`), Adamax slots likewise -- so the optimizer is one kernel launch, the gradient
all-reduce works on contiguous bucket views, and gradient zeroing is one memset.  Non-trainable
state (BN moving statistics, spectral-norm `u`) lives in `state`.  Names follow the convention
documented in DESIGN.md and shared with the oracle so weights can be exchanged by name."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib as L

ALIGN = 8   # floats


@dataclass
class Slot:
    off: int
    shape: Tuple[int, ...]
    logical: Optional[int] = None     # visible size of the last axis when the storage is padded

    @property
    def numel(self) -> int:
        return int(np.prod(self.shape))


class ConvParam:
    def __init__(self, name, k, cin, cout, w: Slot, b: Optional[Slot], u: Optional[Slot]):
        self.name, self.k, self.cin, self.cout = name, k, cin, cout
        self.w, self.b, self.u = w, b, u
        self.wf_off = self.wd_off = -1
        self.wf_ld = self.wd_ld = 0


class DwParam:
    def __init__(self, name, c, w: Slot, b: Slot):
        self.name, self.c, self.w, self.b = name, c, w, b


class BnParam:
    def __init__(self, name, c, gamma: Slot, beta: Slot, rm: Slot, rv: Slot):
        self.name, self.c, self.gamma, self.beta, self.rm, self.rv = name, c, gamma, beta, rm, rv


class SeParam:
    def __init__(self, name, c, hidden, w1, b1, w2, b2):
        self.name, self.c, self.hidden = name, c, hidden
        self.w1, self.b1, self.w2, self.b2 = w1, b1, w2, b2


class ParamStore:
    """Two-phase: declare parameters (host), then `finalize(device, dtype)` allocates and initialises."""

    def __init__(self, seed: int = 1):
        self.gen = torch.Generator().manual_seed(seed)
        self._p_cursor = 0
        self._s_cursor = 0
        self.slots: Dict[str, Slot] = {}        # trainable, by name
        self.sslots: Dict[str, Slot] = {}       # state, by name
        self._init: List[Tuple[Slot, torch.Tensor, bool]] = []
        self.convs: List[ConvParam] = []
        self.bn_loss_layers: List[BnParam] = []
        self.n_trainable = 0
        self.params = self.grads = self.state = None

    # ---- declaration -----------------------------------------------------------------------
    def _alloc(self, name, shape, init: torch.Tensor, state=False, logical: Optional[int] = None) -> Slot:
        n = int(np.prod(shape))
        if state:
            s = Slot(self._s_cursor, tuple(shape))
            self._s_cursor += (n + ALIGN - 1) // ALIGN * ALIGN
            self.sslots[name] = s
        else:
            s = Slot(self._p_cursor, tuple(shape))
            self._p_cursor += (n + ALIGN - 1) // ALIGN * ALIGN
            self.slots[name] = s
            self.n_trainable += n if logical is None else n // shape[-1] * logical
        s.logical = logical if (logical is not None and logical != shape[-1]) else None
        self._init.append((s, init.reshape(-1).to(torch.float32), state))
        return s

    def _glorot(self, shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=self.gen, dtype=torch.float64) * 2 - 1) * lim

    def conv(self, name, k, cin, cout, bias=True, sn=True, pad_cout: int = 1) -> ConvParam:
        """pad_cout > 1: the kernels see ceil(cout / pad_cout) * pad_cout output channels (the MFMA
        gradient kernels want a multiple of 8); the extra channels have zero weights, get zero
        gradients and stay zero, and are hidden from get()/load_named()/n_trainable."""
        cp = (cout + pad_cout - 1) // pad_cout * pad_cout

        def padded(t):
            if cp == cout:
                return t
            out = torch.zeros(t.shape[:-1] + (cp,), dtype=t.dtype)
            out[..., :cout] = t
            return out

        w = self._alloc(name + ".w", (k, k, cin, cp),
                        padded(self._glorot((k, k, cin, cout), k * k * cin, k * k * cout)), logical=cout)
        b = self._alloc(name + ".b", (cp,), torch.zeros(cp), logical=cout) if bias else None
        u = None
        if sn:
            ui = torch.randn(cout, generator=self.gen, dtype=torch.float64).clamp(-2, 2) * 0.02
            u = self._alloc(name + ".u", (cp,), padded(ui), state=True, logical=cout)
        c = ConvParam(name, k, cin, cp, w, b, u)
        c.cout_logical = cout
        self.convs.append(c)
        return c

    def dw(self, name, c) -> DwParam:
        w = self._alloc(name + ".w", (5, 5, c), self._glorot((5, 5, c), 25 * c, 25))
        b = self._alloc(name + ".b", (c,), torch.zeros(c))
        return DwParam(name, c, w, b)

    def bn(self, name, c, in_bn_loss=False) -> BnParam:
        p = BnParam(name, c,
                    self._alloc(name + ".gamma", (c,), torch.ones(c)),
                    self._alloc(name + ".beta", (c,), torch.zeros(c)),
                    self._alloc(name + ".rm", (c,), torch.zeros(c), state=True),
                    self._alloc(name + ".rv", (c,), torch.ones(c), state=True))
        if in_bn_loss:
            self.bn_loss_layers.append(p)
        return p

    def se(self, name, c) -> SeParam:
        h = int(max(c / 16, 4))   # common.py:125
        return SeParam(name, c, h,
                       self._alloc(name + ".w1", (c, h), self._glorot((c, h), c, h)),
                       self._alloc(name + ".b1", (h,), torch.zeros(h)),
                       self._alloc(name + ".w2", (h, c), self._glorot((h, c), h, c)),
                       self._alloc(name + ".b2", (c,), torch.zeros(c)))

    def tensor(self, name, init: torch.Tensor) -> Slot:
        return self._alloc(name, tuple(init.shape), init)

    # ---- materialisation ---------------------------------------------------------------------
    def finalize(self, device, dtype: torch.dtype, zero_pool_floats: int = 1 << 22):
        P = max(self._p_cursor, ALIGN)
        S = max(self._s_cursor, ALIGN)
        host_p = torch.zeros(P, dtype=torch.float32)
        host_s = torch.zeros(S, dtype=torch.float32)
        for slot, init, state in self._init:
            (host_s if state else host_p)[slot.off:slot.off + slot.numel] = init
        self._init.clear()
        self.device, self.dtype = device, dtype
        L.ensure_workspace(device)
        self.params = host_p.to(device)
        self.state = host_s.to(device)
        self.grads = torch.zeros(P, dtype=torch.float32, device=device)
        self.adam_m = torch.zeros(P, dtype=torch.float32, device=device)
        self.adam_u = torch.zeros(P, dtype=torch.float32, device=device)
        self.zero_pool = torch.zeros(zero_pool_floats, dtype=torch.float32, device=device)
        self.zero_reserved = 0
        # ---- compute copies + descriptors for spectral norm / weight prep
        ve = 4 if dtype == torch.float32 else 8
        cursor = t_cursor = blk = p_cursor = 0
        descs = (L.ConvDesc * len(self.convs))()
        for i, c in enumerate(self.convs):
            K = c.k * c.k * c.cin
            # forward copy for every conv (row slices of a 1x1 kernel may be MFMA-able even when
            # the whole Cin is not, e.g. the 256 + 20 DecoderSampleCombiner)
            c.wf_ld = (K + 7) // 8 * 8
            c.wf_off = cursor
            cursor += (c.cout * c.wf_ld + 15) // 16 * 16
            if c.cout % ve == 0:
                c.wd_ld = c.k * c.k * c.cout
                c.wd_off = cursor
                cursor += (c.cin * c.wd_ld + 15) // 16 * 16
            d = descs[i]
            d.w_off, d.wf_off, d.wd_off = c.w.off, c.wf_off, c.wd_off
            d.u_off = c.u.off if c.u is not None else 0
            d.t_off, d.K, d.Cout, d.Cin, d.taps = t_cursor, K, c.cout, c.cin, c.k * c.k
            d.wf_ld, d.wd_ld, d.idx, d.blk_off, d.p_off = c.wf_ld, c.wd_ld, i, blk, p_cursor
            t_cursor += (K + 7) // 8 * 8
            blk += (K + 15) // 16
            p_cursor += (K + 15) // 16 * c.cout
        self.n_convs = len(self.convs)
        self.sn_blocks = blk
        self.wcopies = torch.zeros(max(cursor, 16), dtype=dtype, device=device)
        raw = bytes(descs) or bytes(8)
        self.descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        # per-module descriptor tables (prepare_weights(part=k)): the same descriptors with the workgroup numbering
        # rebased to the part's first conv, so that a part is an ordinary launch of the multi-tensor kernels
        self._parts = []
        marks = getattr(self, "prep_marks", None)
        if marks and len(self.convs):
            assert marks[0] == 0 and marks[-1] == len(self.convs) and list(marks) == sorted(marks)
            reb = (L.ConvDesc * len(self.convs))()
            C.memmove(reb, descs, C.sizeof(descs))
            for i0, i1 in zip(marks[:-1], marks[1:]):
                base = descs[i0].blk_off if i0 < len(self.convs) else blk
                end = descs[i1].blk_off if i1 < len(self.convs) else blk
                for i in range(i0, i1):
                    reb[i].blk_off = descs[i].blk_off - base
                self._parts.append((i0 * C.sizeof(L.ConvDesc), i1 - i0, end - base))
            self.descs_parts = torch.frombuffer(bytearray(bytes(reb)), dtype=torch.uint8).to(device)
        self.sn_t = torch.zeros(max(t_cursor, 8), dtype=torch.float32, device=device)
        self.sn_inv_sigma = torch.ones(max(self.n_convs, 1), dtype=torch.float32, device=device)
        # spectral-norm scratch: per-16-row partial column sums, and w2 laid out like `state` (no zeroing needed)
        self.sn_colpart = torch.empty(max(p_cursor, 8), dtype=torch.float32, device=device)
        self.sn_w2 = torch.empty(S, dtype=torch.float32, device=device)
        # BN-loss table
        tab = [[b.gamma.off, b.c] for b in self.bn_loss_layers] or [[0, 0]]
        self.bn_table = torch.tensor(tab, dtype=torch.int32, device=device)
        self.bn_argmax = torch.zeros(len(tab), dtype=torch.int32, device=device)
        return self

    # ---- views ---------------------------------------------------------------------------------
    def view(self, slot: Slot) -> torch.Tensor:
        return self.params[slot.off:slot.off + slot.numel]

    def gview(self, slot: Slot) -> torch.Tensor:
        return self.grads[slot.off:slot.off + slot.numel]

    def sview(self, slot: Slot) -> torch.Tensor:
        return self.state[slot.off:slot.off + slot.numel]

    @staticmethod
    def _logical(t: torch.Tensor, s: Slot) -> torch.Tensor:
        t = t.reshape(s.shape)
        return t if s.logical is None else t[..., :s.logical]

    def get(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        return self._logical(self.view(s), s)

    def get_grad(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        return self._logical(self.gview(s), s)

    def get_state(self, name: str) -> torch.Tensor:
        s = self.sslots[name]
        return self._logical(self.sview(s), s)

    def load_named(self, params: Dict[str, torch.Tensor], state: Dict[str, torch.Tensor]):
        """Copy weights/state in by name (used by the parity tests and checkpoint loading)."""
        missing = set(self.slots) - set(params)
        extra = set(params) - set(self.slots)
        if missing or extra:
            raise KeyError(f"parameter name mismatch: missing {sorted(missing)[:5]}, extra {sorted(extra)[:5]}")
        for k, v in params.items():
            self.get(k).copy_(v.detach().to(torch.float32))
        for k, v in state.items():
            self.get_state(k).copy_(v.detach().to(torch.float32))

    def named(self) -> Dict[str, torch.Tensor]:
        return {k: self.get(k) for k in self.slots}

    def named_state(self) -> Dict[str, torch.Tensor]:
        return {k: self.get_state(k) for k in self.sslots}

    # ---- per-step device work -----------------------------------------------------------------
    def begin_step(self, zero_grads: bool = True):
        """Zero the gradient buffer and the scratch pool (both are atomics targets).  zero_grads=False: the caller
        zeroes `grads` itself, off the critical path (NVAE._prepare_weights_staged)."""
        if zero_grads:
            self.grads.zero_()
        self.zero_pool.zero_()

    def n_prep_parts(self) -> int:
        return len(self._parts)

    def prepare_weights(self, spectral_norm: bool, part=None):
        """Spectral-norm power iteration (training) and refresh of the MFMA compute copies.  part=k: only the convs
        of module k (prep_marks, set by the model before finalize) - the training step prepares the later modules'
        weights on its side stream while the earlier modules already run."""
        if self.n_convs == 0:
            return
        dt = L.dtype_code(self.dtype)
        if part is None:
            descs, n, blocks = L.ptr(self.descs), self.n_convs, self.sn_blocks
        else:
            off, n, blocks = self._parts[part]
            if n == 0:
                return
            descs = L.ptr(self.descs_parts) + off
        inv = None
        if spectral_norm:
            L.call("nvae_sn_power_iter", L.ptr(self.params), descs, n, blocks,
                   L.ptr(self.state), L.ptr(self.sn_t), L.ptr(self.sn_colpart), L.ptr(self.sn_w2),
                   L.ptr(self.sn_inv_sigma))
            inv = L.ptr(self.sn_inv_sigma)
        L.call("nvae_weight_prep", dt, L.ptr(self.params), descs, n, blocks, inv, L.ptr(self.wcopies))
