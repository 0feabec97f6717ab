"""Data parallelism: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI).

The path shards by samples: every rank holds a full replica and its own slice of the
global batch; the only exchange steps are (i) the [G]-float KL-balance statistic and (ii) the
gradient all-reduce.  Gradients live in ONE flat f32 buffer (params.ParamStore), so the reducer
works on contiguous bucket views, issued from the END of the buffer first: the flat layout is
construction order (pre, enc, dec, post) and backward produces gradients in the reverse order.
The reducer is backend-agnostic (gloo on CPU tensors in the unit tests)."""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, bucket_bytes: int = 64 << 20,
                 force: bool = False):
        assert dist.is_initialized()
        self.group = group
        self.world = dist.get_world_size(group)
        self.force = force      # issue the collectives even for a single rank (exercises RCCL in tests)
        self.bucket_elems = max(bucket_bytes // 4, 1)
        backend = dist.get_backend(group)
        self.native_avg = backend == "nccl"   # RCCL supports ReduceOp.AVG; gloo does not

    def buckets(self, n: int) -> List[slice]:
        """Contiguous slices covering [0, n), last-produced gradients (end of buffer) first."""
        out, hi = [], n
        while hi > 0:
            lo = max(hi - self.bucket_elems, 0)
            out.append(slice(lo, hi))
            hi = lo
        return out

    def _allreduce_mean(self, t: torch.Tensor, async_op: bool):
        if self.native_avg:
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op), False
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op), True

    def allreduce_mean_(self, t: torch.Tensor) -> torch.Tensor:
        if self.world == 1 and not self.force:
            return t
        _, need_div = self._allreduce_mean(t, False)
        if need_div:
            t.div_(self.world)
        return t

    def start_allreduce_(self, flat: torch.Tensor, lo: int, hi: int, works: list) -> None:
        """Launch (without waiting) the bucketed mean all-reduce of flat[lo:hi].  With RCCL the
        collectives run on the communicator's own stream, ordered after everything already issued on
        the current stream, so kernels launched afterwards (the rest of backward) overlap with them.
        Finish with `finish_allreduce_`."""
        if (self.world == 1 and not self.force) or hi <= lo:
            return
        seg = flat[lo:hi]
        for s in self.buckets(seg.numel()):
            w, nd = self._allreduce_mean(seg[s], True)
            works.append((w, seg[s] if nd else None))

    def finish_allreduce_(self, works: list) -> None:
        """Make the current stream wait for the collectives started by `start_allreduce_`."""
        for w, div in works:
            w.wait()
            if div is not None:
                div.div_(self.world)
        works.clear()

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.world > 1 or self.force:
            dist.broadcast(t, src, group=self.group)
        return t

    def allreduce_grads_(self, flat: torch.Tensor) -> torch.Tensor:
        """Average the flat gradient buffer across ranks, bucket by bucket (all in flight at once)."""
        if self.world == 1 and not self.force:
            return flat
        works, need_div = [], False
        for s in self.buckets(flat.numel()):
            w, nd = self._allreduce_mean(flat[s], True)
            works.append(w)
            need_div = need_div or nd
        for w in works:
            w.wait()
        if need_div:
            flat.div_(self.world)
        return flat


class stdout_to_stderr:
    """RCCL prints a version banner on stdout when its first communicator comes up; programs whose stdout is a
    protocol (bench.py: one JSON line) wrap the initialisation and the first collective in this."""

    def __enter__(self):
        import os, sys
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import os, sys
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)


def init_from_env(device_index: Optional[int] = None):
    """Initialise torch.distributed from torchrun's environment.  Returns (rank, world, local_rank)."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # NVAE_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the multi-rank paths on a
        # one-GPU box; RCCL itself needs one device per rank)
        backend = os.environ.get("NVAE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            n_dev = max(torch.cuda.device_count(), 1)
            torch.cuda.set_device((local if device_index is None else device_index) % n_dev)
        with stdout_to_stderr():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            if backend == "nccl":      # bring the communicator up now (and its banner with it)
                dist.all_reduce(torch.zeros(1, device="cuda"))
                torch.cuda.synchronize()
    return rank, world, local
