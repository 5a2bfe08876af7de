"""Data parallelism: one process per GPU, replicated weights, RCCL all-reduce of the flat
encoder / decoder gradient buffers.

Replaces the reference's single-process ``torch.nn.DataParallel(model, device_ids=[2, 3])``
(main_DataParallel.py:609).  Semantics kept from DataParallel: the global batch is split on
dim 0, BatchNorm statistics are per replica (no SyncBN), every loss term is a batch mean so
the average of the rank gradients equals the full-batch gradient.  Each optimiser step
needs exactly one exchange: sum the rank gradients, divide by the world size.

The all-reduce and the optimiser update that consumes it run on a side HIP stream, so that the
(tiny, 28.5 MB) exchange overlaps main-stream work that does not read the weights being updated:
the encoder exchange + update hide behind the D-step's first two decoder passes, the decoder
exchange + update behind the NEXT iteration's ``encode(real)``; the main stream waits on an event,
never on the host.  ``SIV_DP_OVERLAP=0`` runs everything in order on the main stream instead.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .backend import get_backend


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), \
        int(os.environ.get("WORLD_SIZE", "1"))


def _forced() -> bool:
    """SIV_FORCE_COLLECTIVE=1 runs the rendezvous and the all-reduce path even at world size 1,
    so that the RCCL code path can be rehearsed on a one-GPU machine."""
    return os.environ.get("SIV_FORCE_COLLECTIVE") == "1" and "MASTER_ADDR" in os.environ


def init_process_group(backend: str = None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT.
    'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests."""
    rank, local_rank, world = env_world()
    if (world > 1 or _forced()) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


class GradSync:
    """Keeps the replicas of a data-parallel job in step: initial broadcast of parameters and buffers,
    averaging of a flat gradient buffer over all ranks, and the optimiser step that follows it."""

    def __init__(self, device=None):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.side = None
        self.collective = self.world > 1 or (_forced() and dist.is_initialized())
        self.overlap = os.environ.get("SIV_DP_OVERLAP", "1") != "0"
        if self.collective and device is not None and torch.device(device).type == "cuda":
            self.side = torch.cuda.Stream(device=device)
        self._pending = None           # event recorded on the side stream behind an exchange + step

    @property
    def pending(self) -> bool:
        """True while an exchange + update issued on the side stream has not been joined by ``wait()``."""
        return self._pending is not None

    def shard(self, batch: torch.Tensor) -> torch.Tensor:
        """Rank r's slice [r*B/W, (r+1)*B/W) of a global batch."""
        if self.world == 1:
            return batch
        b = batch.shape[0]
        if b % self.world:
            raise ValueError(f"global batch {b} is not divisible by world size {self.world}")
        per = b // self.world
        return batch[self.rank * per:(self.rank + 1) * per]

    def broadcast_(self, module: torch.nn.Module, src: int = 0) -> None:
        """Every rank takes rank ``src``'s parameters and buffers (what nn.DataParallel's per-forward
        ``replicate`` guarantees in the reference, main_DataParallel.py:609): one collective for the
        floating-point state, one for the integer buffers."""
        if not self.collective:
            return
        with torch.no_grad():
            for pick in (lambda t: t.is_floating_point(), lambda t: not t.is_floating_point()):
                ts = [t for t in list(module.parameters()) + list(module.buffers()) if pick(t)]
                if not ts:
                    continue
                flat = torch.cat([t.detach().reshape(-1) for t in ts])
                dist.broadcast(flat, src=src)
                off = 0
                for t in ts:
                    n = t.numel()
                    t.copy_(flat[off:off + n].view(t.shape))
                    off += n

    def _exchange(self, flat_g: torch.Tensor) -> None:
        dist.all_reduce(flat_g, op=dist.ReduceOp.SUM)
        get_backend().scale_(flat_g, 1.0 / self.world)

    def average_(self, flat_g: torch.Tensor) -> None:
        """Sum over ranks / world size, complete (in stream order) when the call returns."""
        if not self.collective:
            return
        if self.side is not None:
            main = torch.cuda.current_stream(flat_g.device)
            self.side.wait_stream(main)
            with torch.cuda.stream(self.side):
                self._exchange(flat_g)
            main.wait_stream(self.side)
        else:
            self._exchange(flat_g)

    def average_and_step(self, opt) -> None:
        """Average ``opt.flat_g`` over the ranks, then ``opt.step()``.  On a GPU job both run on the side
        stream, behind everything the main stream has queued so far, and the main stream carries on: work
        that does not read the parameters of ``opt`` (the D-step's first two decoder passes after the
        encoder update) overlaps the exchange and the update.  ``wait()`` must be called before the
        updated parameters, or the gradient buffer, are touched again."""
        if not self.collective or self.side is None or not self.overlap:
            self.average_(opt.flat_g)
            opt.step()
            return
        self.wait()                    # one exchange in flight at a time (the side stream is ordered anyway)
        main = torch.cuda.current_stream(opt.flat_g.device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            self._exchange(opt.flat_g)
            opt.step()
            self._pending = self.side.record_event()

    def finish_step(self, opt, stats: torch.Tensor) -> torch.Tensor:
        """End of an iteration: the logged statistics become means over the global batch, ``opt.flat_g`` is
        averaged and ``opt.step()`` runs.  On a GPU job all three go to the side stream -- statistics first,
        and the main stream waits for those only; the exchange + update stay in flight (``pending``) until
        the next ``wait()``, which the caller places in front of the next use of ``opt``'s parameters."""
        if not self.collective or self.side is None or not self.overlap:
            self.average_(opt.flat_g)
            opt.step()
            return self.average_stats_(stats)
        self.wait()
        main = torch.cuda.current_stream(stats.device)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
            get_backend().scale_(stats, 1.0 / self.world)
            have_stats = self.side.record_event()
            self._exchange(opt.flat_g)
            opt.step()
            self._pending = self.side.record_event()
        stats.record_stream(self.side)
        main.wait_event(have_stats)
        return stats

    def wait(self) -> None:
        if self._pending is not None:
            torch.cuda.current_stream().wait_event(self._pending)
            self._pending = None

    def average_stats_(self, stats: torch.Tensor) -> torch.Tensor:
        """Batch-mean statistics of a rank -> means over the global batch (for logging)."""
        if self.collective:
            dist.all_reduce(stats, op=dist.ReduceOp.SUM)
            stats /= self.world
        return stats
