"""Fused Adam over flat parameter / gradient buffers.

The reference builds ``optim.Adam(model.encoder.parameters(), lr=2e-4)`` and the same for
the decoder (utils/my_trainer.py:183-184).  Here every parameter of a group is re-pointed
into one contiguous fp32 buffer (3x3x3 convolution weights in channels-last memory, the
layout the kernels read), its ``.grad`` into a parallel buffer, and one ``siv_adam_step``
launch updates the whole group.  The flat gradient buffer is also what the data-parallel
all-reduce sends.

Parameters the forward pass never touches (the shortcut projections of stride-2 blocks,
``encoder.conv``) keep a zero gradient; Adam without weight decay leaves them and their
moments unchanged, which is what torch's "skip parameters whose grad is None" does.
"""
from __future__ import annotations

from typing import Iterable, List

import torch

from .backend import get_backend

_ALIGN = 64  # floats; keeps every view 256-byte aligned for the vectorised kernels


def _view_of(flat: torch.Tensor, off: int, p: torch.Tensor) -> torch.Tensor:
    n = p.numel()
    seg = flat[off:off + n]
    if p.dim() == 5 and tuple(p.shape[2:]) == (3, 3, 3):
        co, ci = p.shape[0], p.shape[1]
        return seg.view(co, 3, 3, 3, ci).permute(0, 4, 1, 2, 3)      # logical (co,ci,3,3,3), NDHWC memory
    return seg.view(p.shape)


class FlatAdam:
    """Adam(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0) on one flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8):
        self.params: List[torch.nn.Parameter] = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam got an empty parameter list")
        dev = self.params[0].device
        self._lr, self.betas, self.eps = float(lr), tuple(betas), float(eps)
        self.base_lr = self._lr
        self.step_count = 0
        self.dev_lr = self.dev_step = None     # device-resident learning rate / step count (graph capture)
        self.offsets = []
        off = 0
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("all parameters of a FlatAdam group must be fp32 on one device")
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.v = torch.zeros(off, device=dev, dtype=torch.float32)
        self._gviews = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                view = _view_of(self.flat_p, o, p)
                view.copy_(p.data)
                p.data = view
                gv = _view_of(self.flat_g, o, p)
                p.grad = gv
                p._siv_grad_buf = gv        # siv.ops adds weight gradients straight into this view
                self._gviews.append(gv)
        B = get_backend()
        if dev.type == "cuda" and hasattr(B, "register_weights"):
            B.register_weights(self.flat_p)       # packed convolution weights are kept between steps

    @property
    def lr(self) -> float:
        return self._lr

    @lr.setter
    def lr(self, value: float) -> None:
        self._lr = float(value)
        if self.dev_lr is not None:
            self.dev_lr.fill_(self._lr)

    def use_device_state(self) -> None:
        """Keep the learning rate and the step count in device memory and let the kernels read them there:
        a ``step()`` captured in a HIP graph then stays correct on every replay."""
        if self.dev_lr is None:
            dev = self.flat_p.device
            self.dev_lr = torch.full((1,), self._lr, dtype=torch.float32, device=dev)
            self.dev_step = torch.full((1,), self.step_count, dtype=torch.int64, device=dev)

    def zero_grad(self) -> None:
        self.flat_g.zero_()
        for p, gv in zip(self.params, self._gviews):
            if p.grad is not gv:
                p.grad = gv

    def _gather_stray_grads(self) -> None:
        # autograd accumulates in place into the views; a caller that replaced .grad
        # (e.g. zero_grad(set_to_none=True) on the module) is folded back in here
        for p, gv in zip(self.params, self._gviews):
            g = p.grad
            if g is None:
                gv.zero_()
            elif g.data_ptr() != gv.data_ptr():
                with torch.no_grad():
                    gv.copy_(g)
            p.grad = gv

    def step(self) -> None:
        self._gather_stray_grads()
        self.step_count += 1
        B = get_backend()
        if self.dev_step is not None:
            B.counter_add(self.dev_step, 1)
            B.adam_step_at(self.flat_p, self.flat_g, self.m, self.v, self.dev_lr, self.betas[0], self.betas[1],
                           self.eps, self.dev_step)
        else:
            B.adam_step(self.flat_p, self.flat_g, self.m, self.v, self.lr, self.betas[0],
                        self.betas[1], self.eps, self.step_count)
        if hasattr(B, "weights_changed"):
            B.weights_changed(self.flat_p.device, self.flat_p)    # only this group's packed weights are stale


class MultiStepLR:
    """optim.lr_scheduler.MultiStepLR(milestones, gamma) stepped once per epoch
    (utils/my_trainer.py:185-186, 502-503)."""

    def __init__(self, opt: FlatAdam, milestones=(350,), gamma: float = 0.1):
        self.opt, self.milestones, self.gamma = opt, tuple(sorted(milestones)), gamma
        self.epoch = 0

    def step(self) -> None:
        self.epoch += 1
        k = sum(1 for m in self.milestones if m <= self.epoch)
        self.opt.lr = self.opt.base_lr * (self.gamma ** k)
