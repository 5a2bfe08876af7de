"""Storage format of the tensors autograd keeps between forward and backward.

The reference trains in fp32 throughout (models/models.py, utils/my_trainer.py:242-325).  The
160x192x160 configuration asks for "bf16 activations": with ``set_storage("bf16")`` every large fp32
tensor that an encoder or decoder pass saves for its backward pass (the inputs of the convolutions,
kept for their weight gradients, and the convolution outputs, kept for the BatchNorm backward) is
held as bf16 (round-to-nearest-even, ``siv_f32_to_bf16``) and expanded again (``siv_bf16_to_f32``,
exact) when the backward pass asks for it.  All arithmetic stays fp32 and the forward pass is
bit-identical to the fp32 mode; only the gradients see the rounded copies (2^-9 relative per element).
Saved-activation memory halves; each saved tensor costs one extra read + half write, and one half read
+ write in backward.

Not stored as bf16: parameters and input volumes (leaf tensors), anything that is not fp32 (dropout masks), small
tensors (statistics, latents), and the operands of the loss functions, which are saved outside the
encoder / decoder passes (their gradients are differences of nearly equal numbers).
"""
import contextlib
import functools
import os

import torch

from . import backend as _backend

MIN_NUMEL = 1 << 16          # below this a tensor is a statistic, a latent or a weight of a thin layer
_storage = "fp32"
packed = {"tensors": 0, "bytes_fp32": 0}      # running totals of what was stored as bf16 (observability, tests)


def set_storage(kind: str) -> None:
    """``"fp32"`` (default: autograd keeps what the forward pass produced) or ``"bf16"``."""
    global _storage
    if kind not in ("fp32", "bf16"):
        raise ValueError(f"activation storage must be 'fp32' or 'bf16', not {kind!r}")
    _storage = kind


def get_storage() -> str:
    return _storage


@contextlib.contextmanager
def storage(kind: str):
    """``with activations.storage("bf16"): ...`` -- scoped :func:`set_storage`."""
    old = _storage
    set_storage(kind)
    try:
        yield
    finally:
        set_storage(old)


class _Packed:
    """A saved tensor held as bf16."""
    __slots__ = ("h",)

    def __init__(self, h):
        self.h = h


_depth = 0      # > 0 inside an encoder / decoder forward pass (``scoped``)
NATIVE = os.environ.get("SIV_BF16_NATIVE", "1") != "0"     # 0: every kept tensor goes through the conversion hooks


def native() -> bool:
    """True where operators that can write the bf16 copy of what they keep THEMSELVES should do so: bf16
    storage is on, an encoder / decoder pass is running and autograd is recording.  (BatchNorm + activation
    emits the copies of its input and output in its own pass and reads the bf16 input in its backward pass;
    everything else goes through the saved-tensor hooks below.)"""
    return _storage == "bf16" and _depth > 0 and torch.is_grad_enabled() and NATIVE


def count_native(*tensors) -> None:
    for t in tensors:
        packed["tensors"] += 1
        packed["bytes_fp32"] += 4 * t.numel()


def _eligible(t) -> bool:
    if not torch.is_tensor(t) or t.dtype != torch.float32 or t.numel() < MIN_NUMEL:
        return False
    if t.is_leaf:                                          # a parameter, or an input volume (data, not an activation)
        return False
    if getattr(t, "_siv_grad_buf", None) is not None:      # a parameter view of a flat optimiser buffer
        return False
    dense = t.is_contiguous() or (t.dim() == 5 and t.is_contiguous(memory_format=torch.channels_last_3d))
    return dense and t.data_ptr() % 16 == 0


def _pack(t):
    if not _eligible(t):
        return t
    packed["tensors"] += 1
    packed["bytes_fp32"] += 4 * t.numel()
    return _Packed(_backend.get_backend().to_bf16(t.detach()))


def _unpack(obj):
    return _backend.get_backend().from_bf16(obj.h) if isinstance(obj, _Packed) else obj


def scoped(forward):
    """Decorator for an encoder / decoder ``forward``: tensors saved inside follow the storage mode."""
    @functools.wraps(forward)
    def wrapper(*args, **kwargs):
        global _depth
        if _storage == "fp32" or not torch.is_grad_enabled():
            return forward(*args, **kwargs)
        _depth += 1
        try:
            with torch.autograd.graph.saved_tensors_hooks(_pack, _unpack):
                return forward(*args, **kwargs)
        finally:
            _depth -= 1
    return wrapper
