"""Random draws of the training path.

The reference draws from torch's global generator in three places: the latent noise
(utils/my_trainer.py:238), the ``eps`` of the reparameterisation (models/models.py:266)
and the Bernoulli masks inside nn.Dropout (models/models.py:95,122,140).  Here every
draw goes through one object so that

* production runs use the device Philox generator of libsiv, keyed by
  (seed, rank, running call index) -- no host round trip, no mask tensor on the host;
* parity tests can replay explicit tensors into the very same call sites.
"""
from __future__ import annotations

import contextlib

import torch

from .backend import get_backend


class PhiloxKeep:
    """A dropout keep mask that is never stored: the operator that consumes it draws element ``i`` (memory
    order of the masked tensor) from Philox(i // 4, stream_id [+ epoch * 2^20], seed), the bits
    ``HipBackend.dropout_keep`` writes for the same key.  Forward and backward of the operator both read the
    epoch counter when they run, so it must not advance between them."""

    __slots__ = ("seed", "stream_id", "epoch")

    def __init__(self, seed: int, stream_id: int, epoch=None):
        self.seed, self.stream_id, self.epoch = int(seed), int(stream_id), epoch

    def materialise(self, shape, p: float, device) -> torch.Tensor:
        return get_backend().dropout_keep(shape, device, p, self.seed, self.stream_id, self.epoch)


class DeviceRng:
    """Philox-4x32-10 on the device; call ``i`` of rank ``r`` uses stream id (r << 40) | i, with bit 39 set
    for the draws of a captured HIP graph: a graph replays the same call indices every time (only the epoch
    counter moves), so its ids live in a namespace of their own that no eager draw can land in."""

    EPOCH_STRIDE = 1 << 20      # stream ids of epoch e start at e * 2^20 (SIV_EPOCH_STRIDE of the kernels)
    GRAPH_NS = 1 << 39          # stream ids of draws captured in a HIP graph carry this bit; eager draws never do

    def __init__(self, seed: int = 0, rank: int = 0):
        self.seed = int(seed)
        self.rank = int(rank)
        self.calls = 0
        self.ns = 0             # 0, or GRAPH_NS while a HIP graph is being captured
        self.epoch = None       # device int64 counter, present once use_device_epoch() was called

    def _sid(self) -> int:
        sid = (self.rank << 40) | self.ns | self.calls
        self.calls += 1
        return sid

    def tell(self) -> int:
        """Index of the next draw."""
        return self.calls

    def seek(self, index: int) -> None:
        """The next draw gets call index ``index`` (draws are keyed by their index, not by a running state)."""
        self.calls = int(index)

    def use_device_epoch(self, device) -> torch.Tensor:
        """From now on every draw adds ``epoch * 2^20`` to its stream id, ``epoch`` being a counter in
        device memory read when the kernel runs: draws captured in a HIP graph then differ from replay to
        replay (``advance_epoch`` inside the captured sequence)."""
        if self.epoch is None:
            self.epoch = torch.zeros(1, dtype=torch.int64, device=device)
        return self.epoch

    def advance_epoch(self) -> None:
        if self.epoch is not None:
            get_backend().counter_add(self.epoch, 1)

    def keep_mask(self, shape, p: float, device) -> torch.Tensor:
        return get_backend().dropout_keep(shape, device, p, self.seed, self._sid(), self.epoch)

    def keep_mask_key(self, shape, p: float, device) -> PhiloxKeep:
        """The key of the mask ``keep_mask`` would return (same stream id consumed), for operators that
        draw the mask themselves."""
        return PhiloxKeep(self.seed, self._sid(), self.epoch)

    def eps(self, shape, device) -> torch.Tensor:
        return get_backend().randn(shape, device, self.seed, self._sid(), self.epoch)


class ReplayRng:
    """Hands out explicit tensors from ``source`` (anything with ``mask(shape, p)`` and
    ``eps(shape)`` returning CPU float tensors), moved to the requested device."""

    def __init__(self, source):
        self.source = source

    def tell(self):
        """Index of the next draw, or None when the source is a plain sequential generator."""
        return getattr(self.source, "i", None)

    def seek(self, index: int) -> None:
        if not hasattr(self.source, "i"):
            raise TypeError("this draw source cannot be repositioned")
        self.source.i = int(index)

    def keep_mask(self, shape, p: float, device) -> torch.Tensor:
        m = self.source.mask(tuple(shape), p)
        return m.to(torch.uint8).to(device).contiguous(memory_format=torch.channels_last_3d)

    keep_mask_key = keep_mask       # replayed masks are explicit tensors

    def eps(self, shape, device) -> torch.Tensor:
        return self.source.eps(tuple(shape)).to(device)


class PairedDraws:
    """Draw indices for two (or more) network passes that the reference runs one after the other and that are
    executed here interleaved, as one batch: member ``g`` makes ``stride`` draws in the reference, so its draws
    own the call indices ``base + g * stride ...``.  ``with paired.member(g):`` positions the generator at member
    g's next index and remembers where it stopped; ``close()`` checks that every member made exactly ``stride``
    draws and leaves the generator behind the last member, where the reference would be."""

    def __init__(self, stride: int, members: int = 2):
        self.rng = get_rng()
        base = self.rng.tell() if hasattr(self.rng, "tell") else None
        if base is None:
            raise TypeError("paired passes need a random source whose draws are addressed by call index")
        self.base, self.stride, self.members = base, int(stride), int(members)
        self.pos = [base + g * self.stride for g in range(self.members)]

    @contextlib.contextmanager
    def member(self, g: int):
        self.rng.seek(self.pos[g])
        try:
            yield
        finally:
            self.pos[g] = self.rng.tell()

    def close(self) -> None:
        for g, p in enumerate(self.pos):
            if p != self.base + (g + 1) * self.stride:
                raise RuntimeError(f"pass {g} of a paired call made {p - self.base - g * self.stride} random draws, "
                                   f"{self.stride} expected")
        self.rng.seek(self.base + self.members * self.stride)


def seekable() -> bool:
    """True when the current generator addresses its draws by call index (``PairedDraws`` works)."""
    r = get_rng()
    return hasattr(r, "tell") and r.tell() is not None


_current = DeviceRng(0)


def get_rng():
    return _current


def set_rng(r) -> None:
    global _current
    _current = r


def manual_seed(seed: int, rank: int = 0) -> None:
    set_rng(DeviceRng(seed, rank))


@contextlib.contextmanager
def use_rng(r):
    global _current
    prev = _current
    _current = r
    try:
        yield r
    finally:
        _current = prev
