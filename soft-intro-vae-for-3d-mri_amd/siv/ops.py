"""Differentiable operators of the training path: torch.autograd.Function wrappers whose
forward and backward both run in libsiv.  autograd supplies only the bookkeeping (which
backward passes run is decided by ``requires_grad`` / ``detach`` exactly as in the
reference trainer)."""
from __future__ import annotations

import os
from typing import Optional

import torch
from torch.autograd import Function

from . import activations
from .backend import get_backend

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
# Weight gradients that are accumulated straight into an optimiser's flat gradient buffer have no
# consumer inside the backward pass, so they can run on a side stream for as long as the pass lasts:
# the matrix-core bound weight gradient of layer L then shares the chip with the HBM-bound
# BatchNorm backward of layer L-1 (and fills the CUs a small deep-layer launch leaves idle).  The
# side stream is joined by a callback at the end of the backward pass, i.e. before ``backward()``
# returns to the caller.
OVERLAP_WGRAD = os.environ.get("SIV_OVERLAP_WGRAD", "1") != "0"
# Parameters managed by siv.optim.FlatAdam carry ``_siv_grad_buf``, the view of the optimiser's flat
# gradient buffer that is their ``.grad``.  The backward passes below then ADD each weight / bias /
# BatchNorm-affine gradient into that view inside the kernel (``accumulate`` = 1 of the C ABI) and hand
# autograd ``None`` for it: no temporary gradient tensor and no separate accumulation kernel per
# parameter and pass.  This is the contract of ``loss.backward()`` (gradients end up in ``.grad``);
# ``torch.autograd.grad`` on such parameters would get ``None`` -- set DIRECT_GRAD = False for that.
DIRECT_GRAD = True
# the fused encoder stem keeps its dropout / activation decisions as two bits per element for its backward
# pass (0: the backward passes redraw or re-read the mask and always form the convolution output's gradient)
STEM_BITS = os.environ.get("SIV_STEM_BITS", "1") != "0"


def _grad_buf(p):
    if not DIRECT_GRAD or p is None:
        return None
    buf = getattr(p, "_siv_grad_buf", None)
    return buf if (buf is not None and p.grad is buf) else None


_join_queued = {}        # device -> id of the backward pass (autograd graph task) whose end joins the side stream


def _join_side(device, main):
    # ``main`` is the stream the backward nodes ran on (the engine's callback may run on another thread)
    main.wait_stream(get_backend().side_stream(device))


def _side_ready(B, dy):
    """Marks the point on the main stream where ``dy`` (and everything before it) is complete; the
    side-stream launch waits for this event only, so main-stream work enqueued after it -- the layer's
    own input gradient -- is ahead of the weight gradient in the hardware queues, not behind it."""
    if not (OVERLAP_WGRAD and dy.is_cuda and hasattr(B, "side_stream")):
        return None
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(dy.device))
    return ev


def _wgrad_on_side_stream(B, x, dy, ready, wgrad_fn):
    """Launch a weight gradient that accumulates into a flat gradient buffer on the side stream."""
    if ready is None:
        wgrad_fn()
        return
    dev = dy.device
    main = torch.cuda.current_stream(dev)
    side = B.side_stream(dev)
    side.wait_event(ready)                       # dy, x and the zeroed gradient buffer are ready
    with torch.cuda.stream(side):
        wgrad_fn()
    for t in (x, dy):
        t.record_stream(side)                    # their memory must outlive the side-stream launch
    # one join per backward pass, keyed by the pass itself: a pass that died in an exception must not
    # leave the next one without its join
    task = torch._C._current_graph_task_id()
    if _join_queued.get(dev) != task:
        _join_queued[dev] = task
        torch.autograd.Variable._execution_engine.queue_callback(lambda: _join_side(dev, main))


class _Conv3dK3(Function):
    @staticmethod
    def forward(ctx, x, w, bias, ups, bn=None):
        # bf16 activation storage: the operator that produced x may have written its bf16 copy in the same
        # pass -- that copy is what the weight gradient gets (expanded only if a weight gradient is asked for)
        xh = getattr(x, "_siv_bf16", None) if activations.get_storage() == "bf16" else None
        ctx.save_for_backward(x if xh is None else xh, w)
        ctx.ups = ups
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)       # no zero tensors for the unused (mean, var) outputs
        ctx.w_buf, ctx.b_buf = _grad_buf(w), _grad_buf(bias)
        if bn is None:
            return get_backend().conv3d_k3_fwd(x, w, bias, ups)
        # training-mode BatchNorm follows: its batch statistics come out of the same call
        y, mean, var = get_backend().conv3d_k3_fwd_stats(
            x, w, bias, ups, bn.running_mean, bn.running_var, bn.num_batches_tracked,
            BN_MOMENTUM if bn.momentum is None else bn.momentum)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, *unused):
        x, w = ctx.saved_tensors
        B = get_backend()
        dx = dw = db = None
        if dy is None:
            return None, None, None, None, None
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        want_w = ctx.needs_input_grad[1] or want_b
        direct = ctx.needs_input_grad[1] and ctx.w_buf is not None and (not want_b or ctx.b_buf is not None)
        if want_w and x.dtype == torch.bfloat16:
            x = B.from_bf16(x)          # before the event the side-stream weight gradient waits for
        ready = _side_ready(B, dy) if direct else None
        if ctx.needs_input_grad[0]:
            dx = B.conv3d_k3_dgrad(dy, w, ctx.ups)
        if direct:
            _wgrad_on_side_stream(B, x, dy, ready,
                                  lambda: B.conv3d_k3_wgrad(x, dy, ctx.ups, want_b, ctx.w_buf, ctx.b_buf))
        if want_w and not direct:
            dw, db = B.conv3d_k3_wgrad(x, dy, ctx.ups, want_b)
            if not ctx.needs_input_grad[1]:
                dw = None
        return dx, dw, db, None, None


def conv3d_k3(x, w, bias=None, ups: int = 1):
    """nn.Conv3d(k=3, s=1, p=1), optionally fused with a preceding nearest x2 upsample."""
    return _Conv3dK3.apply(x, w, bias, ups, None)


class _ConvTranspose3dK3(Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        return get_backend().conv_transpose3d_k3_fwd(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B = get_backend()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = B.conv_transpose3d_k3_dgrad(dy, w)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            dw, db = B.conv_transpose3d_k3_wgrad(x, dy, want_b)
            if not ctx.needs_input_grad[1]:
                dw = None
        return dx, dw, db


def conv_transpose3d_k3(x, w, bias=None):
    """nn.ConvTranspose3d(kernel_size=3, stride=1, padding=1) (models/model.py:202-205): ``w`` has the
    module's own (Cin, Cout, 3, 3, 3) shape."""
    return _ConvTranspose3dK3.apply(x, w, bias)


class _Conv1x1(Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.w_buf, ctx.b_buf = _grad_buf(w), _grad_buf(bias)
        return get_backend().conv1x1_fwd(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B = get_backend()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = B.conv1x1_dgrad(dy, w)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            if ctx.needs_input_grad[1] and ctx.w_buf is not None and (not want_b or ctx.b_buf is not None):
                B.conv1x1_wgrad(x, dy, want_b, ctx.w_buf, ctx.b_buf)
            else:
                dw, db = B.conv1x1_wgrad(x, dy, want_b)
                if not ctx.needs_input_grad[1]:
                    dw = None
        return dx, dw, db


def conv1x1(x, w, bias=None):
    """nn.Conv3d(kernel_size=1)."""
    return _Conv1x1.apply(x, w, bias)


class _Linear(Function):
    @staticmethod
    def forward(ctx, x, w, bias):
        ctx.save_for_backward(x, w)
        ctx.has_bias = bias is not None
        ctx.w_buf, ctx.b_buf = _grad_buf(w), _grad_buf(bias)
        return get_backend().linear_fwd(x, w, bias)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        B = get_backend()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = B.linear_dgrad(dy, w)
        want_b = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] or want_b:
            if ctx.needs_input_grad[1] and ctx.w_buf is not None and (not want_b or ctx.b_buf is not None):
                B.linear_wgrad(x, dy, want_b, ctx.w_buf, ctx.b_buf)
            else:
                dw, db = B.linear_wgrad(x, dy, want_b)
                if not ctx.needs_input_grad[1]:
                    dw = None
        return dx, dw, db


def linear(x, w, bias=None):
    """nn.Linear on a (B, K) input."""
    return _Linear.apply(x, w, bias)


class _AddAct(Function):
    @staticmethod
    def forward(ctx, a, b, slope):
        ctx.save_for_backward(a, b)
        ctx.slope = slope
        return get_backend().add_act_fwd(a, b, slope)

    @staticmethod
    def backward(ctx, dout):
        a, b = ctx.saved_tensors
        d = get_backend().add_act_bwd(dout, a, b, ctx.slope)
        return (d if ctx.needs_input_grad[0] else None), (d if ctx.needs_input_grad[1] else None), None


def add_act(a, b, slope: float):
    """LeakyReLU(slope)(a + b) -- a skip connection that is not preceded by a BatchNorm."""
    return _AddAct.apply(a, b, slope)


class _BnAct(Function):
    @staticmethod
    def forward(ctx, y, gamma, beta, res, keep, bn, p, slope, pool, train, stats=None, groups=1, out=None,
                emit_bf16=False):
        B = get_backend()
        if train and stats is not None:
            mean, var = stats          # already computed (and the running statistics updated)
        elif train and groups > 1:
            # the batch is `groups` passes of the reference run as one: statistics (groups, C), one running
            # update per part in part order; the kernels read the grouping off the shape of mean / var
            mean, var = B.bn_stats(y, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                   BN_MOMENTUM if bn.momentum is None else bn.momentum, groups)
        elif train:
            mean, var = B.bn_stats(y, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                   BN_MOMENTUM if bn.momentum is None else bn.momentum)
        else:
            mean, var = bn.running_mean, bn.running_var
        # `out` (an Into holder): the result is written into that tensor, e.g. a batch slice of a larger buffer
        dst = {"out": out.t} if out is not None else {}
        oh = None
        if emit_bf16:
            # bf16 activation storage: this pass also writes the bf16 copies of y (kept for backward in place
            # of y) and of the output (kept by the convolution that consumes it, for its weight gradient)
            out, yh, oh = B.bn_act_fwd(y, mean, var, gamma, beta, res, keep, p, slope, pool, bn.eps, emit_bf16=True,
                                       **dst)
            activations.count_native(y, out)
            y = yh
            ctx.mark_non_differentiable(oh)
            ctx.set_materialize_grads(False)        # no zero "gradient" tensor for the bf16 copy
        else:
            out = B.bn_act_fwd(y, mean, var, gamma, beta, res, keep, p, slope, pool, bn.eps, **dst)
        ctx.save_for_backward(y, mean, var, gamma, beta, res, keep)
        ctx.cfg = (p, slope, pool, bn.eps, train)
        ctx.g_buf, ctx.b_buf = _grad_buf(gamma), _grad_buf(beta)
        return out, oh

    @staticmethod
    def backward(ctx, dout, d_oh=None):
        if dout is None:
            return (None,) * 14
        y, mean, var, gamma, beta, res, keep = ctx.saved_tensors        # y: fp32, or its bf16 copy
        p, slope, pool, eps, train = ctx.cfg
        if not train:
            raise NotImplementedError("backward through evaluation-mode BatchNorm is not part of the "
                                      "training path (the reference validates under torch.no_grad())")
        want_aff = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        want_res = res is not None and ctx.needs_input_grad[3]
        if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and ctx.g_buf is not None and ctx.b_buf is not None:
            dy, _, _, dres = get_backend().bn_act_bwd(dout, y, mean, var, gamma, beta, res, keep, p, slope,
                                                      pool, eps, True, want_res, ctx.g_buf, ctx.b_buf)
            dg = db = None
        else:
            dy, dg, db, dres = get_backend().bn_act_bwd(dout, y, mean, var, gamma, beta, res, keep, p, slope,
                                                        pool, eps, want_aff, want_res)
        if not ctx.needs_input_grad[0]:
            dy = None
        return (dy, dg if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None,
                dres, None, None, None, None, None, None, None, None, None, None)


class Into:
    """Holder that hands a destination tensor to an operator without making it an autograd input."""
    __slots__ = ("t",)

    def __init__(self, t):
        self.t = t


class _Merge(Function):
    """The members of a paired pass as one batch.  When every member already lives in its batch slice of
    ``into.t`` (the operators that produced them were given those slices as their destination) nothing is
    copied; backward hands each member its slice of the gradient."""

    @staticmethod
    def forward(ctx, into, *hs):
        big = into.t
        per = hs[0].shape[0]
        ctx.per = per
        for g, h in enumerate(hs):
            dst = big[g * per:(g + 1) * per]
            if h.data_ptr() != dst.data_ptr() or h.stride() != dst.stride():
                dst.copy_(h)
        return big

    @staticmethod
    def backward(ctx, grad):
        per = ctx.per
        n = grad.shape[0] // per
        return (None,) + tuple(grad[g * per:(g + 1) * per] for g in range(n))


def merge_members(hs, into=None):
    """Concatenation of the member tensors ``hs`` on the batch axis; ``into``: the buffer they were written
    into slice by slice (then no copy is made)."""
    if into is None:
        return torch.cat(list(hs))
    return _Merge.apply(Into(into), *hs)


class _Split(Function):
    """Batch slices of a merged tensor, one per member; backward assembles the members' gradients in one
    buffer (a missing gradient counts as zero)."""

    @staticmethod
    def forward(ctx, x, members):
        per = x.shape[0] // members
        ctx.cfg = (per, members)
        base = x.detach()
        return tuple(base[g * per:(g + 1) * per] for g in range(members))

    @staticmethod
    def backward(ctx, *grads):
        per, members = ctx.cfg
        live = [g for g in grads if g is not None]
        if not live:
            return None, None
        ref = live[0]
        full = torch.empty((per * members,) + tuple(ref.shape[1:]), dtype=ref.dtype, device=ref.device,
                           memory_format=torch.channels_last_3d if ref.dim() == 5 else torch.contiguous_format)
        for g, gr in enumerate(grads):
            dst = full[g * per:(g + 1) * per]
            if gr is None:
                dst.zero_()
            else:
                dst.copy_(gr)
        return full, None


def split_members(x, members: int):
    """The ``members`` equal batch slices of ``x`` (views; no copy in the forward pass)."""
    return _Split.apply(x, members)


def bn_act(y, bn, slope: float, pool: int = 1, residual=None, keep=None, p: float = 0.0,
           train: bool = True, stats=None, groups: int = 1, out=None):
    """BatchNorm3d (parameters/buffers of module ``bn``) -> [+residual] -> LeakyReLU(slope)
    -> AvgPool3d(pool) -> dropout(keep, p) as one fused pass.  ``stats`` = (mean, biased var) when
    the producer of ``y`` already computed the batch statistics.  ``groups`` > 1 (training mode): the batch is
    that many passes of the reference concatenated; each part is normalised with its own batch statistics and
    updates the running statistics once, in order."""
    if pool not in (1, 2):
        raise ValueError(f"only stride/pool 1 or 2 is supported (got {pool})")
    emit = (train and activations.native() and y.dtype == torch.float32 and y.numel() >= activations.MIN_NUMEL
            and hasattr(get_backend(), "bn_bf16_supported") and get_backend().bn_bf16_supported(y))
    o, oh = _BnAct.apply(y, bn.weight, bn.bias, residual, keep, bn, p, slope, pool, train, stats,
                         groups if train else 1, Into(out) if out is not None else None, emit)
    if oh is not None:
        o._siv_bf16 = oh          # found by the convolution that takes `o` as its input
    return o


def conv_bn_act(x, conv, bn, slope: float, ups: int = 1, pool: int = 1, residual=None, keep=None,
                p: float = 0.0, train: bool = True, groups: int = 1, out=None):
    """conv3d_k3 (parameters of module ``conv``) followed by ``bn_act``.  In training mode the
    convolution call also produces the BatchNorm batch statistics (one set); with ``groups`` > 1 the
    statistics are per part of the batch and come from a pass of their own."""
    if train and groups > 1:
        y = _Conv3dK3.apply(x, conv.weight, conv.bias, ups, None)
        return bn_act(y, bn, slope, pool, residual, keep, p, True, None, groups, out)
    if train:
        y, mean, var = _Conv3dK3.apply(x, conv.weight, conv.bias, ups, bn)
        return bn_act(y, bn, slope, pool, residual, keep, p, True, (mean, var), 1, out)
    y = _Conv3dK3.apply(x, conv.weight, conv.bias, ups, None)
    return bn_act(y, bn, slope, pool, residual, keep, p, False, None, 1, out)


class _Conv1BnAct(Function):
    """Conv3d(1, C, 3, 1, 1) -> BatchNorm3d -> LeakyReLU -> Dropout (the encoder stem) with the convolution
    output recomputed by every pass instead of stored; backward hands the gradient of that output to the
    ordinary weight / input gradient kernels of the convolution -- or, where the input needs no gradient
    (data, detached reconstructions), forms the parameter gradients in one pass over ``dout`` without it.
    The training-mode forward pass leaves its dropout / activation decisions as two bits per element for
    the backward pass (backends with ``stem_bits``)."""

    @staticmethod
    def forward(ctx, x, w, bias, gamma, beta, keep, bn, p, slope, train, emit_bf16=False):
        B = get_backend()
        want_bits = bool(STEM_BITS and train and getattr(B, "stem_bits", False) and any(ctx.needs_input_grad))
        opts = {}
        if emit_bf16:
            opts["emit_bf16"] = True
        if want_bits:
            opts["emit_bits"] = True
        res = B.conv1_bnact_fwd(x, w, bias, gamma, beta, keep, p, slope, bn.eps, train,
                                bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                BN_MOMENTUM if bn.momentum is None else bn.momentum, **opts)
        out, mean, var = res[:3]
        oh = res[3] if emit_bf16 else None      # bf16 copy of the output, for the convolution that consumes it
        ctx.bits = res[-1] if want_bits else None
        if oh is not None:
            activations.count_native(out)
            ctx.mark_non_differentiable(oh)
            ctx.set_materialize_grads(False)        # no zero "gradient" tensor for the bf16 copy
        ctx.mask_key = None if keep is None or torch.is_tensor(keep) else keep   # a mask drawn in-kernel
        if want_bits:                            # the bits carry the mask; p = 0 where nothing was dropped
            p = float(p) if keep is not None else 0.0
            ctx.mask_key, keep = None, None
        ctx.save_for_backward(x, w, bias, mean, var, gamma, beta, keep if ctx.mask_key is None else None)
        ctx.cfg = (p, slope, bn.eps, train)
        ctx.bufs = (_grad_buf(w), _grad_buf(bias), _grad_buf(gamma), _grad_buf(beta))
        return out, oh

    @staticmethod
    def backward(ctx, dout, d_oh=None):
        if dout is None:
            return (None,) * 11
        x, w, bias, mean, var, gamma, beta, keep = ctx.saved_tensors
        if ctx.mask_key is not None:
            keep = ctx.mask_key
        p, slope, eps, train = ctx.cfg
        w_buf, cb_buf, g_buf, b_buf = ctx.bufs
        if not train:
            raise NotImplementedError("backward through evaluation-mode BatchNorm is not part of the "
                                      "training path (the reference validates under torch.no_grad())")
        B = get_backend()
        dx = dw = dcb = dg = db = None
        want_aff = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        want_cb = bias is not None and ctx.needs_input_grad[2]
        bits = ctx.bits
        if bits is not None and not ctx.needs_input_grad[0]:
            # parameters only: no gradient of the convolution output, one pass over dout
            need = ctx.needs_input_grad
            if (need[1] and need[3] and need[4] and (bias is None or need[2]) and w_buf is not None and
                    g_buf is not None and b_buf is not None and (bias is None or cb_buf is not None)):
                B.conv1_bnact_wgrad(dout, x, w, bias, mean, var, gamma, bits, p, slope, eps, w_buf, cb_buf, g_buf, b_buf)
                return (None,) * 11
            dw, dcb, dg, db = B.conv1_bnact_wgrad(dout, x, w, bias, mean, var, gamma, bits, p, slope, eps,
                                                  want_bias=want_cb, want_affine=want_aff)
            return (None, dw if need[1] else None, dcb if want_cb else None, dg if need[3] else None,
                    db if need[4] else None, None, None, None, None, None, None)
        kw = {"bits": bits} if bits is not None else {}
        if ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and g_buf is not None and b_buf is not None:
            dy, _, _ = B.conv1_bnact_bwd(dout, x, w, bias, mean, var, gamma, beta, keep, p, slope, eps, True,
                                         g_buf, b_buf, **kw)
        else:
            dy, dg, db = B.conv1_bnact_bwd(dout, x, w, bias, mean, var, gamma, beta, keep, p, slope, eps, want_aff,
                                           **kw)
        if ctx.needs_input_grad[1] or want_cb:
            direct = ctx.needs_input_grad[1] and w_buf is not None and (not want_cb or cb_buf is not None)
            if direct:
                _wgrad_on_side_stream(B, x, dy, _side_ready(B, dy),
                                      lambda: B.conv3d_k3_wgrad(x, dy, 1, want_cb, w_buf, cb_buf))
            else:
                dw, dcb = B.conv3d_k3_wgrad(x, dy, 1, want_cb)
                if not ctx.needs_input_grad[1]:
                    dw = None
        if ctx.needs_input_grad[0]:
            dx = B.conv3d_k3_dgrad(dy, w, 1)
        return (dx, dw, dcb, dg if ctx.needs_input_grad[3] else None, db if ctx.needs_input_grad[4] else None,
                None, None, None, None, None, None)


def conv1_bn_act_supported(x, conv) -> bool:
    B = get_backend()
    return (hasattr(B, "conv1_bnact_supported") and conv.in_channels == 1 and
            B.conv1_bnact_supported(x, conv.weight))


def conv1_bn_act(x, conv, bn, slope: float, keep=None, p: float = 0.0, train: bool = True):
    """``dropout(LeakyReLU(bn(conv(x))))`` for a single-channel input as one operator (see
    ``conv1_bn_act_supported``); the fallback is ``conv_bn_act``.  ``keep`` is a uint8 mask tensor or a
    ``rng.PhiloxKeep`` key, for which the kernels draw the mask themselves."""
    emit = train and activations.native() and x.numel() * conv.out_channels >= activations.MIN_NUMEL
    o, oh = _Conv1BnAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, keep, bn, p, slope, train, emit)
    if oh is not None:
        o._siv_bf16 = oh
    return o


class _BnActConvTo1(Function):
    """BatchNorm3d -> LeakyReLU -> Conv3d(C, 1, 3, 1, 1) with the first two applied while the convolution
    loads its operand: the activated C-channel tensor is neither written nor read, in forward or in the
    weight gradient.  Backward = the weight gradient (same prologue) and one fused call for the convolution's
    input gradient and BatchNorm's own backward."""

    @staticmethod
    def forward(ctx, y, gamma, beta, w, bias, bn, slope, train, stats, emit_bf16=False):
        B = get_backend()
        mean, var = stats if train else (bn.running_mean, bn.running_var)
        if emit_bf16:
            # bf16 activation storage: the forward pass writes the bf16 copy of y, which is all the backward
            # passes read
            out, yh = B.bnact_conv_to1_fwd(y, mean, var, gamma, beta, bn.eps, slope, w, bias, emit_bf16=True)
            activations.count_native(y)
            y = yh
        else:
            out = B.bnact_conv_to1_fwd(y, mean, var, gamma, beta, bn.eps, slope, w, bias)
        ctx.save_for_backward(y, mean, var, gamma, beta, w)
        ctx.cfg = (slope, bn.eps, train, bias is not None)
        ctx.bufs = (_grad_buf(gamma), _grad_buf(beta), _grad_buf(w), _grad_buf(bias))
        return out

    @staticmethod
    def backward(ctx, dout):
        y, mean, var, gamma, beta, w = ctx.saved_tensors
        slope, eps, train, has_bias = ctx.cfg
        g_buf, b_buf, w_buf, cb_buf = ctx.bufs
        if not train:
            raise NotImplementedError("backward through evaluation-mode BatchNorm is not part of the "
                                      "training path (the reference validates under torch.no_grad())")
        B = get_backend()
        dy = dg = db = dw = dcb = None
        want_cb = has_bias and ctx.needs_input_grad[4]
        if ctx.needs_input_grad[3] or want_cb:
            direct = ctx.needs_input_grad[3] and w_buf is not None and (not want_cb or cb_buf is not None)
            if direct:
                _wgrad_on_side_stream(B, y, dout, _side_ready(B, dout), lambda: B.bnact_conv_to1_wgrad(
                    y, mean, var, gamma, beta, eps, slope, dout, want_cb, w_buf, cb_buf))
            else:
                dw, dcb = B.bnact_conv_to1_wgrad(y, mean, var, gamma, beta, eps, slope, dout, want_cb)
                if not ctx.needs_input_grad[3]:
                    dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            # the gradient of the activation (never stored either) is recomputed inside both passes
            want_aff = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
            if ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and g_buf is not None and b_buf is not None:
                dy, _, _ = B.bnact_conv_to1_dgrad(dout, w, y, mean, var, gamma, beta, eps, slope, True, g_buf, b_buf)
            else:
                dy, dg, db = B.bnact_conv_to1_dgrad(dout, w, y, mean, var, gamma, beta, eps, slope, want_aff)
            if not ctx.needs_input_grad[0]:
                dy = None
        return (dy, dg if ctx.needs_input_grad[1] else None, db if ctx.needs_input_grad[2] else None, dw, dcb,
                None, None, None, None, None)


def bn_act_conv_to1_supported(y) -> bool:
    B = get_backend()
    return hasattr(B, "bnact_conv_to1_supported") and B.bnact_conv_to1_supported(y)


def bn_act_conv_to1(y, bn, slope: float, conv, train: bool = True, stats=None):
    """``conv(LeakyReLU(bn(y)))`` for a ``Conv3d(C, 1, 3, 1, 1)`` module ``conv`` as one fused operator
    (``bn_act_conv_to1_supported(y)`` tells whether the shape has a kernel); ``stats`` as in ``bn_act``."""
    if train and stats is None:
        B = get_backend()
        stats = B.bn_stats(y, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                           BN_MOMENTUM if bn.momentum is None else bn.momentum)
    emit = train and activations.native() and y.numel() >= activations.MIN_NUMEL
    return _BnActConvTo1.apply(y, bn.weight, bn.bias, conv.weight, conv.bias, bn, slope, train, stats, emit)


def conv_with_stats(x, conv, bn, ups: int = 1, train: bool = True):
    """The convolution half of ``conv_bn_act``: (raw output, (mean, var) or None)."""
    if train:
        y, mean, var = _Conv3dK3.apply(x, conv.weight, conv.bias, ups, bn)
        return y, (mean, var)
    return _Conv3dK3.apply(x, conv.weight, conv.bias, ups, None), None


class _ActDropout(Function):
    @staticmethod
    def forward(ctx, x, keep, p, slope):
        ctx.save_for_backward(x, keep)
        ctx.cfg = (p, slope)
        return get_backend().act_dropout_fwd(x, keep, p, slope)

    @staticmethod
    def backward(ctx, dout):
        x, keep = ctx.saved_tensors
        p, slope = ctx.cfg
        return get_backend().act_dropout_bwd(dout, x, keep, p, slope), None, None, None


def act_dropout(x, slope: float, keep=None, p: float = 0.0):
    """LeakyReLU(slope) (slope 0 = ReLU) followed by dropout with an explicit keep mask."""
    return _ActDropout.apply(x, keep, p, slope)


class _Reparam(Function):
    @staticmethod
    def forward(ctx, mu, logvar, eps, eps_const):
        ctx.save_for_backward(logvar, eps)
        ctx.eps_const = eps_const
        return get_backend().reparam_fwd(mu, logvar, eps, eps_const)

    @staticmethod
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dmu, dlv = get_backend().reparam_bwd(dz, logvar, eps, ctx.eps_const)
        return dmu, dlv, None, None


def reparameterize(mu, logvar, eps: Optional[torch.Tensor], eps_const: float = 0.0):
    """z = mu + eps * exp(0.5 * logvar); ``eps=None`` uses the constant ``eps_const``."""
    return _Reparam.apply(mu, logvar, eps, eps_const)


class _KlPerSample(Function):
    @staticmethod
    def forward(ctx, mu, logvar):
        ctx.save_for_backward(mu, logvar)
        return get_backend().kl_fwd(mu, logvar)

    @staticmethod
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        return get_backend().kl_bwd(g, mu, logvar)


def kl_per_sample(mu, logvar):
    """(B,) vector of -0.5 * sum(1 + logvar - mu^2 - exp(logvar)) per sample."""
    return _KlPerSample.apply(mu, logvar)


class _SsePerSample(Function):
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return get_backend().sse_fwd(x, y)

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        return get_backend().sse_bwd(g, x, y, ctx.needs_input_grad[0], ctx.needs_input_grad[1])


def sse_per_sample(x, y):
    """(B,) vector of sum((x - y)^2) per sample."""
    return _SsePerSample.apply(x, y)


class _SintroLossE(Function):
    @staticmethod
    def forward(ctx, scale, br, bn, bk, *vecs):
        ctx.save_for_backward(*vecs)
        ctx.cfg = (scale, br, bn, bk)
        out = get_backend().sintro_loss_e_fwd(vecs, scale, br, bn, bk)
        loss, stats = out[0], out[1:]
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return loss, stats

    @staticmethod
    def backward(ctx, g, g_stats):
        vecs = ctx.saved_tensors
        d = get_backend().sintro_loss_e_bwd(g.reshape(1), vecs, *ctx.cfg)
        return (None, None, None, None) + tuple(d[i] if ctx.needs_input_grad[4 + i] else None for i in range(6))


def sintro_loss_e(sse_real, kl_real, sse_fake, kl_fake, sse_rec, kl_rec, scale, beta_rec, beta_neg, beta_kl):
    """The encoder objective of utils/my_trainer.py:260-284 from six per-sample vectors (one kernel):
    returns (lossE, stats) with stats = [mean(kl_real), exp_elbo_rec, exp_elbo_fake] (no gradient)."""
    loss, stats = _SintroLossE.apply(scale, beta_rec, beta_neg, beta_kl, sse_real, kl_real, sse_fake, kl_fake,
                                     sse_rec, kl_rec)
    return loss, stats.detach()


class _SintroLossD(Function):
    @staticmethod
    def forward(ctx, scale, br, bk, gr, *vecs):
        ctx.cfg = (vecs[0].numel(), scale, br, bk, gr)
        out = get_backend().sintro_loss_d_fwd(vecs, scale, br, bk, gr)
        loss, stats = out[0], out[1:]
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)
        return loss, stats

    @staticmethod
    def backward(ctx, g, g_stats):
        d = get_backend().sintro_loss_d_bwd(g.reshape(1), *ctx.cfg)
        return (None, None, None, None) + tuple(d[i] if ctx.needs_input_grad[4 + i] else None for i in range(5))


def sintro_loss_d(sse_real, kl_rec, kl_fake, sse_rec_rec, sse_fake_rec, scale, beta_rec, beta_kl, gamma_r):
    """The decoder objective of utils/my_trainer.py:301-321 from five per-sample vectors (one kernel):
    returns (lossD, stats) with stats = [mean(kl_fake), mean(kl_rec), mean(sse_real)] (no gradient)."""
    loss, stats = _SintroLossD.apply(scale, beta_rec, beta_kl, gamma_r, sse_real, kl_rec, kl_fake, sse_rec_rec,
                                     sse_fake_rec)
    return loss, stats.detach()
