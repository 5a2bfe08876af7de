"""Tensor-level view of the libsiv C ABI.

``HipBackend`` turns torch tensors (logical NCDHW, ``channels_last_3d`` memory = NDHWC)
into the raw pointers/extents the C entry points take, allocates outputs with torch's
caching allocator and launches on torch's current HIP stream.  It is the only backend
this package ships: there is no CPU path.  ``set_backend`` exists so that the test-suite
can substitute a checker when exercising the host logic on a machine without a GPU.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

CL3D = torch.channels_last_3d


def cl(t: torch.Tensor) -> torch.Tensor:
    """NDHWC memory for a logical NCDHW tensor (no copy when already so); tensors of any other
    rank (the (B, z) latent vectors of the FC-latent model) are made row-major."""
    return t.contiguous(memory_format=CL3D) if t.dim() == 5 else t.contiguous()


def like(t: torch.Tensor) -> torch.Tensor:
    """Uninitialised tensor with the shape of ``t`` and the memory layout ``cl`` gives it."""
    return torch.empty_like(t, memory_format=CL3D if t.dim() == 5 else torch.contiguous_format)


def empty_cl(shape, like: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    return torch.empty(tuple(shape), device=like.device, dtype=dtype, memory_format=CL3D)


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class HipBackend:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._ws = {}
        self._side = {}
        self._pack_cache = {}          # device -> donated cache memory
        self._stable = {}              # weight pointer -> torch version counter when last used

    # ---- plumbing ------------------------------------------------------------
    def _stream(self, t: torch.Tensor) -> int:
        return torch.cuda.current_stream(t.device).cuda_stream

    def _workspace(self, t: torch.Tensor) -> Tuple[int, int]:
        # one workspace per (device, stream): calls on different streams may overlap
        key = (t.device, torch.cuda.current_stream(t.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = int(self.lib.siv_workspace_bytes())
            ws = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
            self._ws[key] = ws
        return ws.data_ptr(), ws.numel()

    def side_stream(self, device) -> "torch.cuda.Stream":
        """A second stream per device on which an independent kernel (the weight gradient of a
        layer, beside its input gradient) can run concurrently with the main stream."""
        st = self._side.get(device)
        if st is None:
            # default priority: the priority range of this device is (0, -1), there is nothing below it
            st = torch.cuda.Stream(device=device)
            self._side[device] = st
        return st

    @staticmethod
    def _check_dev(*ts):
        for t in ts:
            if t is not None and not t.is_cuda:
                raise _lib.SivError("libsiv kernels need tensors on a HIP device (got a CPU tensor); "
                                    "there is no CPU fallback")
            if t is not None and t.dtype not in (torch.float32, torch.uint8, torch.int64):
                raise _lib.SivError(f"unsupported dtype {t.dtype}")

    # ---- packed-weight cache ---------------------------------------------------
    PACK_CACHE_BYTES = 768 << 20

    def register_weights(self, flat: torch.Tensor) -> None:
        """Declare ``flat`` (a flat fp32 parameter buffer) as weight memory that changes only through
        ``weights_changed()`` or through torch in-place operations (tracked by its version counter): the
        packed / Winograd-transformed form of convolution weights inside it is then kept between calls."""
        import weakref
        self._check_dev(flat)
        dev = flat.device
        with torch.cuda.device(dev):
            if dev not in self._pack_cache:
                mem = torch.empty(self.PACK_CACHE_BYTES, dtype=torch.uint8, device=dev)
                self._pack_cache[dev] = mem
                _lib.check(self.lib.siv_pack_cache_configure(mem.data_ptr(), mem.numel()), "siv_pack_cache_configure")
            _lib.check(self.lib.siv_pack_cache_add_weights(flat.data_ptr(), flat.numel() * flat.element_size()),
                       "siv_pack_cache_add_weights")
        # once the buffer dies its address range may be handed to anything: stop caching inside it
        weakref.finalize(flat, self._forget_weights, flat.data_ptr(), dev)

    def _forget_weights(self, ptr: int, dev) -> None:
        try:
            with torch.cuda.device(dev):
                self.lib.siv_pack_cache_remove_weights(ptr)
        except Exception:       # interpreter shutdown
            pass

    def weights_changed(self, device=None, flat: Optional[torch.Tensor] = None) -> None:
        """The weights were written behind torch's back (an optimiser step, a graph replay): forget the cached
        packings -- all of them, or with ``flat`` (a registered flat parameter buffer) only those of the
        weights inside it, so that an encoder update leaves the decoder's packings valid and vice versa."""
        if not self._pack_cache:
            return
        if flat is not None:
            if flat.device in self._pack_cache:
                with torch.cuda.device(flat.device):
                    self.lib.siv_pack_cache_invalidate_range(flat.data_ptr(), flat.numel() * flat.element_size())
            return
        if device is not None:
            device = torch.device(device)
            if device.type == "cuda" and device.index is None:      # "cuda" = the current device
                device = torch.device("cuda", torch.cuda.current_device())
        for dev in ([device] if device is not None else list(self._pack_cache)):
            if dev in self._pack_cache:
                with torch.cuda.device(dev):
                    self.lib.siv_pack_cache_invalidate()

    def _track_weight_version(self, w: torch.Tensor) -> None:
        # torch-level in-place writes into a cached weight (initialisation, load_state_dict, ...) show in
        # the tensor's version counter; the optimiser's raw-pointer update reports itself (weights_changed)
        if not self._pack_cache:
            return
        k, v = w.data_ptr(), w._version
        seen = self._stable.get(k)
        if seen != v:
            self._stable[k] = v
            if seen is not None:
                self.weights_changed(w.device)

    def set_conv_path(self, path: int) -> None:
        _lib.check(self.lib.siv_set_conv_path(path), "siv_set_conv_path")

    # ---- 3x3x3 convolution ------------------------------------------------------
    def conv3d_k3_fwd(self, x, w, bias, ups: int = 1):
        self._check_dev(x, w, bias)
        self._track_weight_version(w)
        x, w = cl(x), cl(w)
        N, Cin, Di, Hi, Wi = x.shape
        Cout = w.shape[0]
        assert w.shape[1] == Cin and tuple(w.shape[2:]) == (3, 3, 3), (tuple(w.shape), Cin)
        D, H, W = Di * ups, Hi * ups, Wi * ups
        y = empty_cl((N, Cout, D, H, W), x)
        ws, wsb = self._workspace(x)
        _lib.check(self.lib.siv_conv3d_k3_fwd(_p(x), _p(w), _p(bias), _p(y), N, D, H, W, Cin, Cout, ups,
                                              ws, wsb, self._stream(x)), "siv_conv3d_k3_fwd")
        return y

    def conv3d_k3_fwd_stats(self, x, w, bias, ups, running_mean, running_var, nbt, momentum: float):
        """Convolution + training-mode BatchNorm statistics of its output: (y, mean, biased var)."""
        self._check_dev(x, w, bias, running_mean, running_var, nbt)
        self._track_weight_version(w)
        x, w = cl(x), cl(w)
        N, Cin, Di, Hi, Wi = x.shape
        Cout = w.shape[0]
        assert w.shape[1] == Cin and tuple(w.shape[2:]) == (3, 3, 3), (tuple(w.shape), Cin)
        D, H, W = Di * ups, Hi * ups, Wi * ups
        y = empty_cl((N, Cout, D, H, W), x)
        mean = torch.empty(Cout, device=x.device, dtype=torch.float32)
        var = torch.empty(Cout, device=x.device, dtype=torch.float32)
        ws, wsb = self._workspace(x)
        _lib.check(self.lib.siv_conv3d_k3_fwd_bnstats(_p(x), _p(w), _p(bias), _p(y), N, D, H, W, Cin, Cout, ups,
                                                      _p(mean), _p(var), _p(running_mean), _p(running_var),
                                                      _p(nbt), float(momentum), ws, wsb, self._stream(x)),
                   "siv_conv3d_k3_fwd_bnstats")
        return y, mean, var

    def conv3d_k3_dgrad(self, dy, w, ups: int = 1):
        self._check_dev(dy, w)
        self._track_weight_version(w)
        dy, w = cl(dy), cl(w)
        N, Cout, D, H, W = dy.shape
        Cin = w.shape[1]
        ws, wsb = self._workspace(dy)
        st = self._stream(dy)
        if ups == 2 and self.lib.siv_conv3d_k3_dgrad_up_supported(Cin, Cout):
            dxs = empty_cl((N, Cin, D // 2, H // 2, W // 2), dy)
            _lib.check(self.lib.siv_conv3d_k3_dgrad_up(_p(dy), _p(w), _p(dxs), N, D, H, W, Cin, Cout, ws, wsb, st),
                       "siv_conv3d_k3_dgrad_up")
            return dxs
        dx = empty_cl((N, Cin, D, H, W), dy)
        _lib.check(self.lib.siv_conv3d_k3_dgrad(_p(dy), _p(w), _p(dx), N, D, H, W, Cin, Cout, ws, wsb, st),
                   "siv_conv3d_k3_dgrad")
        if ups == 1:
            return dx
        assert ups == 2
        dxs = empty_cl((N, Cin, D // 2, H // 2, W // 2), dy)
        _lib.check(self.lib.siv_upsample2_bwd(_p(dx), _p(dxs), N, D, H, W, Cin, st), "siv_upsample2_bwd")
        return dxs

    @staticmethod
    def _grad_targets(shape_w, dy, want_bias, dw_out, db_out, cl_w=True):
        """Fresh gradient tensors, or the caller's accumulation buffers (``accumulate`` = 1)."""
        if dw_out is None:
            dw = empty_cl(shape_w, dy) if cl_w else torch.empty(shape_w, device=dy.device, dtype=torch.float32)
            db = torch.empty(shape_w[0], device=dy.device, dtype=torch.float32) if want_bias else None
            return dw, db, 0
        assert tuple(dw_out.shape) == tuple(shape_w) and dw_out.dtype == torch.float32, (dw_out.shape, shape_w)
        assert (cl(dw_out) if cl_w else dw_out.contiguous()).data_ptr() == dw_out.data_ptr(), "gradient buffer layout"
        assert not want_bias or db_out is not None
        return dw_out, (db_out if want_bias else None), 1

    def conv3d_k3_wgrad(self, x, dy, ups: int = 1, want_bias: bool = False, dw_out=None, db_out=None):
        """Weight (and bias) gradient; with ``dw_out`` / ``db_out`` the result is ADDED into those buffers."""
        self._check_dev(x, dy, dw_out, db_out)
        x, dy = cl(x), cl(dy)
        N, Cout, D, H, W = dy.shape
        Cin = x.shape[1]
        dw, db, acc = self._grad_targets((Cout, Cin, 3, 3, 3), dy, want_bias, dw_out, db_out)
        ws, wsb = self._workspace(dy)
        _lib.check(self.lib.siv_conv3d_k3_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, D, H, W, Cin, Cout, ups, acc,
                                                ws, wsb, self._stream(dy)), "siv_conv3d_k3_wgrad")
        return dw, db

    # ---- the encoder stem Conv3d(1, C) -> BatchNorm -> LeakyReLU -> Dropout as one operator ------------
    def conv1_bnact_supported(self, x, w) -> bool:
        return bool(x.is_cuda and x.dim() == 5 and x.shape[1] == 1 and
                    self.lib.siv_conv1_bnact_supported(int(w.shape[0])))

    @staticmethod
    def _mask_args(keep):
        """(mask tensor or None, (seed, stream id, epoch counter)) from a uint8 tensor, a key object with
        ``seed`` / ``stream_id`` / ``epoch`` (mask drawn by the kernels), or None."""
        if keep is None:
            return None, (0, 0, None)
        if torch.is_tensor(keep):
            return cl(keep), (0, 0, None)
        m = 2 ** 64 - 1
        return None, (keep.seed & m, keep.stream_id & m, keep.epoch)

    stem_bits = True        # conv1_bnact_fwd(emit_bits=True), conv1_bnact_bwd(bits=...), conv1_bnact_wgrad exist

    def conv1_bnact_fwd(self, x, w, bias, gamma, beta, keep, p, slope, eps, train: bool, running_mean,
                        running_var, nbt, momentum: float, emit_bf16: bool = False, emit_bits: bool = False):
        """Returns (out, mean, var); in evaluation mode mean / var are the running statistics.  ``emit_bf16``:
        (out, mean, var, bf16 copy of out), the copy written by the same pass.  ``emit_bits``: one more
        element, the uint8 tensor with the dropout / activation decisions of the pass (two bits per output
        element) that ``conv1_bnact_bwd(bits=...)`` and ``conv1_bnact_wgrad`` take."""
        masked = keep is not None       # the C entry point reads "no mask tensor, p > 0" as "draw the mask"
        keep, mk = self._mask_args(keep)
        self._check_dev(x, w, bias, gamma, beta, keep, mk[2], running_mean, running_var, nbt)
        x, w = x.contiguous(), cl(w)
        N, _, D, H, W = x.shape
        C = w.shape[0]
        out = empty_cl((N, C, D, H, W), x)
        if train:
            mean = torch.empty(C, device=x.device, dtype=torch.float32)
            var = torch.empty(C, device=x.device, dtype=torch.float32)
            rm, rv, nb = running_mean, running_var, nbt
        else:
            mean, var, rm, rv, nb = running_mean, running_var, None, None, None
        ws, wsb = self._workspace(x)
        oh = torch.empty_like(out, dtype=torch.bfloat16) if emit_bf16 else None
        bits = (torch.empty(self.lib.siv_conv1_bnact_bits_bytes(N, D, H, W, C), dtype=torch.uint8, device=x.device)
                if emit_bits else None)
        _lib.check(self.lib.siv_conv1_bnact_fwd_bits(_p(x), _p(w), _p(bias), _p(gamma), _p(beta), _p(keep), mk[0],
                                                     mk[1], _p(mk[2]), float(p) if masked else 0.0, float(slope),
                                                     float(eps), int(train), _p(mean), _p(var), _p(rm),
                                                     _p(rv), _p(nb), float(momentum), _p(out), _p(oh), _p(bits),
                                                     N, D, H, W, C, ws, wsb, self._stream(x)), "siv_conv1_bnact_fwd")
        res = (out, mean, var, oh) if emit_bf16 else (out, mean, var)
        return res + (bits,) if emit_bits else res

    def conv1_bnact_bwd(self, dout, x, w, bias, mean, var, gamma, beta, keep, p, slope, eps,
                        want_affine: bool, dgamma_out=None, dbeta_out=None, bits=None):
        """Gradient of the convolution output (and of gamma / beta: returned, or ADDED into the buffers).
        ``bits`` (from the forward pass) replaces ``keep``; ``p`` is then the dropout probability of that pass."""
        masked = keep is not None
        keep, mk = self._mask_args(keep)
        self._check_dev(dout, x, w, bias, mean, var, gamma, beta, keep, mk[2], dgamma_out, dbeta_out, bits)
        dout, x, w = cl(dout), x.contiguous(), cl(w)
        N, C, D, H, W = dout.shape
        dy = like(dout)
        acc = 0
        if want_affine and dgamma_out is not None:
            assert dbeta_out is not None and dgamma_out.is_contiguous() and dbeta_out.is_contiguous()
            dgamma, dbeta, acc = dgamma_out, dbeta_out, 1
        else:
            dgamma = torch.empty(C, device=x.device, dtype=torch.float32) if want_affine else None
            dbeta = torch.empty(C, device=x.device, dtype=torch.float32) if want_affine else None
        ws, wsb = self._workspace(dout)
        if bits is not None:
            _lib.check(self.lib.siv_conv1_bnact_bwd_bits(_p(dout), _p(x), _p(w), _p(bias), _p(mean), _p(var),
                                                         _p(gamma), _p(beta), _p(bits), float(p), float(slope),
                                                         float(eps), _p(dy), _p(dgamma), _p(dbeta), acc, N, D, H, W,
                                                         C, ws, wsb, self._stream(dout)), "siv_conv1_bnact_bwd_bits")
            return dy, dgamma, dbeta
        _lib.check(self.lib.siv_conv1_bnact_bwd(_p(dout), _p(x), _p(w), _p(bias), _p(mean), _p(var), _p(gamma),
                                                _p(beta), _p(keep), mk[0], mk[1], _p(mk[2]),
                                                float(p) if masked else 0.0, float(slope), float(eps), _p(dy),
                                                _p(dgamma), _p(dbeta), acc, N, D, H, W, C, ws, wsb,
                                                self._stream(dout)), "siv_conv1_bnact_bwd")
        return dy, dgamma, dbeta

    def conv1_bnact_wgrad(self, dout, x, w, bias, mean, var, gamma, bits, p, slope, eps,
                          dw_out=None, dbias_out=None, dgamma_out=None, dbeta_out=None, want_bias: bool = True,
                          want_affine: bool = True):
        """Weight-only backward of the stem: (dw, dbias, dgamma, dbeta) straight from ``dout``, the input and
        the bits of the forward pass, without the gradient of the convolution output.  Either all four ``*_out``
        buffers are given (the results are ADDED into them) or none (fresh tensors are returned; ``None`` for
        what was not asked for)."""
        self._check_dev(dout, x, w, bias, mean, var, gamma, bits, dw_out, dbias_out, dgamma_out, dbeta_out)
        dout, x, w = cl(dout), x.contiguous(), cl(w)
        N, C, D, H, W = dout.shape
        want_bias = want_bias and bias is not None
        if dw_out is not None:
            assert dw_out.is_contiguous(memory_format=torch.channels_last_3d) or dw_out.is_contiguous()
            dw, db, dg, dbt, acc = dw_out, dbias_out if want_bias else None, dgamma_out, dbeta_out, 1
        else:
            dw = torch.empty_like(w)
            new = lambda: torch.empty(C, device=x.device, dtype=torch.float32)
            db = new() if want_bias else None
            dg, dbt = (new(), new()) if want_affine else (None, None)
            acc = 0
        ws, wsb = self._workspace(dout)
        _lib.check(self.lib.siv_conv1_bnact_wgrad_bits(_p(dout), _p(x), _p(w), _p(bias), _p(mean), _p(var), _p(gamma),
                                                       _p(bits), float(p), float(slope), float(eps), _p(dw), _p(db),
                                                       _p(dg), _p(dbt), acc, N, D, H, W, C, ws, wsb,
                                                       self._stream(dout)), "siv_conv1_bnact_wgrad_bits")
        return dw, db, dg, dbt

    # ---- BatchNorm + LeakyReLU folded into the Cin -> 1 convolution behind them ---------------------
    def bnact_conv_to1_supported(self, x) -> bool:
        """True when ``bnact_conv_to1_fwd`` has a kernel for the raw convolution output ``x``."""
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 5):
            return False
        N, C, D, H, W = x.shape
        return bool(self.lib.siv_bnact_conv3d_k3_to1_supported(N, D, H, W, C))

    def bnact_conv_to1_fwd(self, x, mean, var, gamma, beta, eps: float, slope: float, w, bias,
                           emit_bf16: bool = False):
        """``emit_bf16``: returns (y, bf16 copy of x), the copy written by the same pass; the backward calls
        below take that copy in place of ``x``."""
        self._check_dev(x, mean, var, gamma, beta, w, bias)
        x, w = cl(x), cl(w)
        N, C, D, H, W = x.shape
        assert tuple(w.shape) == (1, C, 3, 3, 3), tuple(w.shape)
        y = empty_cl((N, 1, D, H, W), x)
        xh = torch.empty_like(x, dtype=torch.bfloat16) if emit_bf16 else None
        _lib.check(self.lib.siv_bnact_conv3d_k3_to1_fwd_bf16(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), eps, slope,
                                                             _p(w), _p(bias), _p(y), _p(xh), N, D, H, W, C,
                                                             self._stream(x)),
                   "siv_bnact_conv3d_k3_to1_fwd")
        return (y, xh) if emit_bf16 else y

    def bnact_conv_to1_dgrad(self, dy, w, x, mean, var, gamma, beta, eps: float, slope: float,
                             want_affine: bool, dgamma_out=None, dbeta_out=None):
        """Gradient of the raw tensor ``x`` (and of gamma / beta: returned, or ADDED into the given buffers)."""
        x_bf16 = x.dtype == torch.bfloat16
        self._check_dev(dy, w, None if x_bf16 else x, mean, var, gamma, beta, dgamma_out, dbeta_out)
        dy, w, x = cl(dy), cl(w), cl(x)
        N, C, D, H, W = x.shape
        dx = empty_cl(tuple(x.shape), dy)
        acc = 0
        if want_affine and dgamma_out is not None:
            assert dbeta_out is not None and dgamma_out.is_contiguous() and dbeta_out.is_contiguous()
            dgamma, dbeta, acc = dgamma_out, dbeta_out, 1
        else:
            dgamma = torch.empty(C, device=x.device, dtype=torch.float32) if want_affine else None
            dbeta = torch.empty(C, device=x.device, dtype=torch.float32) if want_affine else None
        ws, wsb = self._workspace(x)
        fn = self.lib.siv_bnact_conv3d_k3_to1_dgrad_bf16 if x_bf16 else self.lib.siv_bnact_conv3d_k3_to1_dgrad
        _lib.check(fn(_p(dy), _p(w), _p(x), _p(mean), _p(var), _p(gamma),
                      _p(beta), eps, slope, _p(dx), _p(dgamma), _p(dbeta), acc,
                      N, D, H, W, C, ws, wsb, self._stream(x)),
                   "siv_bnact_conv3d_k3_to1_dgrad")
        return dx, dgamma, dbeta

    def bnact_conv_to1_wgrad(self, x, mean, var, gamma, beta, eps: float, slope: float, dy,
                             want_bias: bool = False, dw_out=None, db_out=None):
        x_bf16 = x.dtype == torch.bfloat16
        self._check_dev(None if x_bf16 else x, mean, var, gamma, beta, dy, dw_out, db_out)
        x, dy = cl(x), cl(dy)
        N, C, D, H, W = x.shape
        dw, db, acc = self._grad_targets((1, C, 3, 3, 3), dy, want_bias, dw_out, db_out)
        ws, wsb = self._workspace(dy)
        fn = self.lib.siv_bnact_conv3d_k3_to1_wgrad_bf16 if x_bf16 else self.lib.siv_bnact_conv3d_k3_to1_wgrad
        _lib.check(fn(_p(x), _p(mean), _p(var), _p(gamma), _p(beta), eps, slope,
                      _p(dy), _p(dw), _p(db), N, D, H, W, C, acc, ws, wsb,
                      self._stream(dy)),
                   "siv_bnact_conv3d_k3_to1_wgrad")
        return dw, db

    def last_conv_kernel(self) -> str:
        """Kernel family the last 3x3x3 convolution call of this thread ran on."""
        return self.lib.siv_last_conv_kernel().decode()

    # ---- 3x3x3 transposed convolution (nn.ConvTranspose3d k3/s1/p1) ----------------
    def conv_transpose3d_k3_fwd(self, x, w, bias):
        """x (N,Cin,D,H,W), w (Cin,Cout,3,3,3) as nn.ConvTranspose3d stores it, bias (Cout) or None."""
        self._check_dev(x, w, bias)
        self._track_weight_version(w)
        x, w = cl(x), cl(w)
        N, Cin, D, H, W = x.shape
        assert w.shape[0] == Cin and tuple(w.shape[2:]) == (3, 3, 3), (tuple(w.shape), Cin)
        Cout = w.shape[1]
        y = empty_cl((N, Cout, D, H, W), x)
        ws, wsb = self._workspace(x)
        _lib.check(self.lib.siv_conv_transpose3d_k3_fwd(_p(x), _p(w), _p(bias), _p(y), N, D, H, W, Cin, Cout,
                                                        ws, wsb, self._stream(x)), "siv_conv_transpose3d_k3_fwd")
        return y

    def conv_transpose3d_k3_dgrad(self, dy, w):
        self._check_dev(dy, w)
        self._track_weight_version(w)
        dy, w = cl(dy), cl(w)
        N, Cout, D, H, W = dy.shape
        Cin = w.shape[0]
        assert w.shape[1] == Cout, (tuple(w.shape), Cout)
        dx = empty_cl((N, Cin, D, H, W), dy)
        ws, wsb = self._workspace(dy)
        _lib.check(self.lib.siv_conv_transpose3d_k3_dgrad(_p(dy), _p(w), _p(dx), N, D, H, W, Cin, Cout,
                                                          ws, wsb, self._stream(dy)), "siv_conv_transpose3d_k3_dgrad")
        return dx

    def conv_transpose3d_k3_wgrad(self, x, dy, want_bias: bool = False):
        self._check_dev(x, dy)
        x, dy = cl(x), cl(dy)
        N, Cout, D, H, W = dy.shape
        Cin = x.shape[1]
        dw = empty_cl((Cin, Cout, 3, 3, 3), dy)
        db = torch.empty(Cout, device=dy.device, dtype=torch.float32) if want_bias else None
        ws, wsb = self._workspace(dy)
        _lib.check(self.lib.siv_conv_transpose3d_k3_wgrad(_p(x), _p(dy), _p(dw), _p(db), N, D, H, W, Cin, Cout, 0,
                                                          ws, wsb, self._stream(dy)), "siv_conv_transpose3d_k3_wgrad")
        return dw, db

    # ---- 1x1x1 convolution -------------------------------------------------------
    def conv1x1_fwd(self, x, w, bias):
        self._check_dev(x, w, bias)
        x = cl(x)
        w = w.contiguous()
        N, Cin, D, H, W = x.shape
        Cout = w.shape[0]
        y = empty_cl((N, Cout, D, H, W), x)
        _lib.check(self.lib.siv_conv1x1_fwd(_p(x), _p(w), _p(bias), _p(y), N * D * H * W, Cin, Cout,
                                            self._stream(x)), "siv_conv1x1_fwd")
        return y

    def conv1x1_dgrad(self, dy, w):
        self._check_dev(dy, w)
        dy = cl(dy)
        w = w.contiguous()
        N, Cout, D, H, W = dy.shape
        Cin = w.shape[1]
        dx = empty_cl((N, Cin, D, H, W), dy)
        _lib.check(self.lib.siv_conv1x1_dgrad(_p(dy), _p(w), _p(dx), N * D * H * W, Cin, Cout,
                                              self._stream(dy)), "siv_conv1x1_dgrad")
        return dx

    def conv1x1_wgrad(self, x, dy, want_bias: bool = False, dw_out=None, db_out=None):
        self._check_dev(x, dy, dw_out, db_out)
        x, dy = cl(x), cl(dy)
        N, Cout, D, H, W = dy.shape
        Cin = x.shape[1]
        dw, db, acc = self._grad_targets((Cout, Cin, 1, 1, 1), dy, want_bias, dw_out, db_out, cl_w=False)
        ws, wsb = self._workspace(dy)
        _lib.check(self.lib.siv_conv1x1_wgrad(_p(x), _p(dy), _p(dw), _p(db), N * D * H * W, Cin, Cout, acc,
                                              ws, wsb, self._stream(dy)), "siv_conv1x1_wgrad")
        return dw, db

    # ---- fully connected layers ------------------------------------------------------
    def linear_fwd(self, x, w, bias):
        self._check_dev(x, w, bias)
        x, w = x.contiguous(), w.contiguous()
        B, K = x.shape
        O = w.shape[0]
        assert w.shape[1] == K, (tuple(w.shape), K)
        y = torch.empty((B, O), device=x.device, dtype=torch.float32)
        ws, wsb = self._workspace(x)
        _lib.check(self.lib.siv_linear_fwd(_p(x), _p(w), _p(bias), _p(y), B, K, O, ws, wsb, self._stream(x)),
                   "siv_linear_fwd")
        return y

    def linear_dgrad(self, dy, w):
        self._check_dev(dy, w)
        dy, w = dy.contiguous(), w.contiguous()
        B, O = dy.shape
        K = w.shape[1]
        dx = torch.empty((B, K), device=dy.device, dtype=torch.float32)
        ws, wsb = self._workspace(dy)
        _lib.check(self.lib.siv_linear_dgrad(_p(dy), _p(w), _p(dx), B, K, O, ws, wsb, self._stream(dy)),
                   "siv_linear_dgrad")
        return dx

    def linear_wgrad(self, x, dy, want_bias: bool = False, dw_out=None, db_out=None):
        self._check_dev(x, dy, dw_out, db_out)
        x, dy = x.contiguous(), dy.contiguous()
        B, K = x.shape
        O = dy.shape[1]
        dw, db, acc = self._grad_targets((O, K), dy, want_bias, dw_out, db_out, cl_w=False)
        _lib.check(self.lib.siv_linear_wgrad(_p(x), _p(dy), _p(dw), _p(db), B, K, O, acc, self._stream(dy)),
                   "siv_linear_wgrad")
        return dw, db

    def add_act_fwd(self, a, b, slope):
        self._check_dev(a, b)
        a, b = cl(a), cl(b)
        assert a.shape == b.shape, (a.shape, b.shape)
        out = like(a)
        _lib.check(self.lib.siv_add_act_fwd(_p(a), _p(b), float(slope), _p(out), a.numel(), self._stream(a)),
                   "siv_add_act_fwd")
        return out

    def add_act_bwd(self, dout, a, b, slope):
        self._check_dev(dout, a, b)
        dout, a, b = cl(dout), cl(a), cl(b)
        d = like(a)
        _lib.check(self.lib.siv_add_act_bwd(_p(dout), _p(a), _p(b), float(slope), _p(d), a.numel(),
                                            self._stream(a)), "siv_add_act_bwd")
        return d

    # ---- input pipeline ------------------------------------------------------------------
    def preprocess(self, voxels, cut_range: float = 4.0):
        """(B, D, H, W) or (B, 1, D, H, W) raw fp32 volumes -> (B, 1, D, H, W) normalised."""
        self._check_dev(voxels)
        v = voxels.contiguous()
        if v.dim() == 5:
            assert v.shape[1] == 1, tuple(v.shape)
            v = v[:, 0]
        B, D, H, W = v.shape
        out = torch.empty((B, 1, D, H, W), device=v.device, dtype=torch.float32)
        ws, wsb = self._workspace(v)
        _lib.check(self.lib.siv_preprocess(_p(v), _p(out), B, D * H * W, float(cut_range), ws, wsb,
                                           self._stream(v)), "siv_preprocess")
        return out

    def flip_shift(self, voxels, params):
        """params: int32 (B, 6) device tensor = flipD, flipH, flipW, shiftD, shiftH, shiftW."""
        self._check_dev(voxels)
        v = voxels.contiguous()
        shape = v.shape
        B, (D, H, W) = shape[0], shape[-3:]
        assert v.numel() == B * D * H * W, tuple(shape)
        assert params.is_cuda and params.dtype == torch.int32 and tuple(params.shape) == (B, 6)
        out = torch.empty_like(v)
        _lib.check(self.lib.siv_flip_shift(_p(v), _p(params.contiguous()), _p(out), B, D, H, W,
                                           self._stream(v)), "siv_flip_shift")
        return out

    def affine_resample(self, voxels, mats, pad="minimum"):
        """mats: float32 (B, 3, 4) device tensor, output index -> input index; pad: "minimum" or a number."""
        self._check_dev(voxels)
        v = voxels.contiguous()
        shape = v.shape
        B, (D, H, W) = shape[0], shape[-3:]
        assert v.numel() == B * D * H * W, tuple(shape)
        assert mats.is_cuda and mats.dtype == torch.float32 and mats.numel() == B * 12, tuple(mats.shape)
        out = torch.empty_like(v)
        mode, value = (1, 0.0) if pad == "minimum" else (0, float(pad))
        ws, wsb = self._workspace(v)
        _lib.check(self.lib.siv_affine_resample(_p(v), _p(mats.contiguous()), _p(out), B, D, H, W, mode, value,
                                                ws, wsb, self._stream(v)), "siv_affine_resample")
        return out

    # ---- batch norm + activation tail --------------------------------------------
    def bn_stats(self, y, running_mean, running_var, nbt, momentum: float, groups: int = 1):
        """Batch statistics (mean, biased variance) of ``y``; with ``groups`` > 1 one pair per consecutive part of
        the batch, shape (groups, C), and one running-statistics update per part, in part order."""
        self._check_dev(y, running_mean, running_var, nbt)
        y = cl(y)
        N, C, D, H, W = y.shape
        if N % groups:
            raise _lib.SivError(f"bn_stats: batch {N} is not a multiple of {groups} groups")
        shape = (C,) if groups == 1 else (groups, C)
        mean = torch.empty(shape, device=y.device, dtype=torch.float32)
        var = torch.empty(shape, device=y.device, dtype=torch.float32)
        ws, wsb = self._workspace(y)
        _lib.check(self.lib.siv_bn_stats_grouped(_p(y), (N // groups) * D * H * W, C, groups, _p(mean), _p(var),
                                                 _p(running_mean), _p(running_var), _p(nbt), float(momentum), ws,
                                                 wsb, self._stream(y)), "siv_bn_stats")
        return mean, var

    @staticmethod
    def _groups_of(mean) -> int:
        return 1 if mean.dim() == 1 else int(mean.shape[0])

    def bn_bf16_supported(self, y) -> bool:
        """Can ``bn_act_fwd(..., emit_bf16=True)`` / a bf16 ``y`` in ``bn_act_bwd`` be used for this tensor?"""
        return bool(y.is_cuda and y.dim() == 5 and self.lib.siv_bn_act_bf16_supported(int(y.shape[1]))
                    and y.shape[0] * y.shape[2] * y.shape[3] * y.shape[4] < (1 << 31))

    def bn_act_fwd(self, y, mean, var, gamma, beta, res, keep, p, slope, pool, eps, out=None, emit_bf16=False):
        """``mean`` / ``var`` of shape (groups, C) select the grouped operator (one set of statistics per
        consecutive part of the batch).  ``out``: write the result there (a dense channels-last tensor of the
        output shape, e.g. a batch slice of a larger buffer) instead of into a new tensor.  ``emit_bf16``:
        returns (out, bf16 copy of y, bf16 copy of out), both copies written by the same pass."""
        self._check_dev(y, mean, var, gamma, beta, res, keep)
        groups = self._groups_of(mean)
        mean, var = mean.contiguous(), var.contiguous()
        y = cl(y)
        res = cl(res) if res is not None else None
        N, C, D, H, W = y.shape
        oshape = (N, C, D // pool, H // pool, W // pool)
        if out is None:
            out = empty_cl(oshape, y)
        elif (tuple(out.shape) != oshape or out.dtype != torch.float32 or out.device != y.device
              or not out.is_contiguous(memory_format=CL3D)):
            raise _lib.SivError("bn_act_fwd: `out` must be a dense channels-last fp32 tensor of the output shape")
        if keep is not None:
            keep = cl(keep)
            assert keep.dtype == torch.uint8 and keep.shape == out.shape
        if emit_bf16:
            yh = torch.empty_like(y, dtype=torch.bfloat16)
            oh = torch.empty_like(out, dtype=torch.bfloat16)
            _lib.check(self.lib.siv_bn_act_fwd_bf16(_p(y), _p(mean), _p(var), _p(gamma), _p(beta), _p(res), _p(keep),
                                                    float(p), float(slope), int(pool), float(eps), _p(out), _p(yh),
                                                    _p(oh), N, D, H, W, C, groups, self._stream(y)),
                       "siv_bn_act_fwd_bf16")
            return out, yh, oh
        _lib.check(self.lib.siv_bn_act_fwd_grouped(_p(y), _p(mean), _p(var), _p(gamma), _p(beta), _p(res), _p(keep),
                                                   float(p), float(slope), int(pool), float(eps), _p(out),
                                                   N, D, H, W, C, groups, self._stream(y)), "siv_bn_act_fwd")
        return out

    def bn_act_bwd(self, dout, y, mean, var, gamma, beta, res, keep, p, slope, pool, eps,
                   want_affine: bool, want_res: bool, dgamma_out=None, dbeta_out=None):
        """With ``dgamma_out`` / ``dbeta_out`` the affine gradients are ADDED into those buffers."""
        y_bf16 = y.dtype == torch.bfloat16              # the bf16 copy bn_act_fwd(emit_bf16=True) made
        self._check_dev(dout, None if y_bf16 else y, mean, var, gamma, beta, res, keep, dgamma_out, dbeta_out)
        groups = self._groups_of(mean)
        mean, var = mean.contiguous(), var.contiguous()
        dout, y = cl(dout), cl(y)
        res = cl(res) if res is not None else None
        keep = cl(keep) if keep is not None else None
        N, C, D, H, W = y.shape
        dy = empty_cl(tuple(y.shape), dout)
        acc = 0
        if want_affine and dgamma_out is not None:
            assert dbeta_out is not None and dgamma_out.is_contiguous() and dbeta_out.is_contiguous()
            dgamma, dbeta, acc = dgamma_out, dbeta_out, 1
        else:
            dgamma = torch.empty(C, device=y.device, dtype=torch.float32) if want_affine else None
            dbeta = torch.empty(C, device=y.device, dtype=torch.float32) if want_affine else None
        dres = empty_cl(tuple(y.shape), dout) if (want_res and res is not None) else None
        ws, wsb = self._workspace(y)
        fn = self.lib.siv_bn_act_bwd_bf16 if y_bf16 else self.lib.siv_bn_act_bwd_grouped
        _lib.check(fn(_p(dout), _p(y), _p(mean), _p(var), _p(gamma), _p(beta), _p(res),
                      _p(keep), float(p), float(slope), int(pool), float(eps),
                      _p(dy), _p(dgamma), _p(dbeta), _p(dres), acc, N, D, H, W, C,
                      groups, ws, wsb, self._stream(y)), "siv_bn_act_bwd")
        return dy, dgamma, dbeta, dres

    def act_dropout_fwd(self, x, keep, p, slope):
        self._check_dev(x, keep)
        x = cl(x)
        keep = cl(keep) if keep is not None else None
        out = like(x)
        _lib.check(self.lib.siv_act_dropout_fwd(_p(x), _p(keep), float(p), float(slope), _p(out), x.numel(),
                                                self._stream(x)), "siv_act_dropout_fwd")
        return out

    def act_dropout_bwd(self, dout, x, keep, p, slope):
        self._check_dev(dout, x, keep)
        dout, x = cl(dout), cl(x)
        keep = cl(keep) if keep is not None else None
        dx = like(x)
        _lib.check(self.lib.siv_act_dropout_bwd(_p(dout), _p(x), _p(keep), float(p), float(slope), _p(dx),
                                                x.numel(), self._stream(x)), "siv_act_dropout_bwd")
        return dx

    # ---- latent sampling and losses ------------------------------------------------
    def reparam_fwd(self, mu, logvar, eps, eps_const):
        self._check_dev(mu, logvar, eps)
        mu, logvar = cl(mu), cl(logvar)
        eps = cl(eps) if eps is not None else None
        z = like(mu)
        _lib.check(self.lib.siv_reparam_fwd(_p(mu), _p(logvar), _p(eps), float(eps_const), _p(z), mu.numel(),
                                            self._stream(mu)), "siv_reparam_fwd")
        return z

    def reparam_bwd(self, dz, logvar, eps, eps_const):
        self._check_dev(dz, logvar, eps)
        dz, logvar = cl(dz), cl(logvar)
        eps = cl(eps) if eps is not None else None
        dmu = like(dz)
        dlv = like(dz)
        _lib.check(self.lib.siv_reparam_bwd(_p(dz), _p(logvar), _p(eps), float(eps_const), _p(dmu), _p(dlv),
                                            dz.numel(), self._stream(dz)), "siv_reparam_bwd")
        return dmu, dlv

    def kl_fwd(self, mu, logvar):
        self._check_dev(mu, logvar)
        mu, logvar = cl(mu), cl(logvar)
        B = mu.shape[0]
        out = torch.empty(B, device=mu.device, dtype=torch.float32)
        _lib.check(self.lib.siv_kl_fwd(_p(mu), _p(logvar), _p(out), B, mu.numel() // B, self._stream(mu)),
                   "siv_kl_fwd")
        return out

    def kl_bwd(self, g, mu, logvar):
        self._check_dev(g, mu, logvar)
        mu, logvar, g = cl(mu), cl(logvar), g.contiguous()
        B = mu.shape[0]
        dmu = like(mu)
        dlv = like(mu)
        _lib.check(self.lib.siv_kl_bwd(_p(g), _p(mu), _p(logvar), _p(dmu), _p(dlv), B, mu.numel() // B,
                                       self._stream(mu)), "siv_kl_bwd")
        return dmu, dlv

    def sse_fwd(self, x, y):
        self._check_dev(x, y)
        x, y = cl(x), cl(y)
        assert x.shape == y.shape, (x.shape, y.shape)
        B = x.shape[0]
        out = torch.empty(B, device=x.device, dtype=torch.float32)
        ws, wsb = self._workspace(x)
        _lib.check(self.lib.siv_sse_fwd(_p(x), _p(y), _p(out), B, x.numel() // B, ws, wsb, self._stream(x)),
                   "siv_sse_fwd")
        return out

    def sse_bwd(self, g, x, y, want_x: bool, want_y: bool):
        self._check_dev(g, x, y)
        x, y, g = cl(x), cl(y), g.contiguous()
        B = x.shape[0]
        dx = like(x) if want_x else None
        dy = like(x) if want_y else None
        _lib.check(self.lib.siv_sse_bwd(_p(g), _p(x), _p(y), _p(dx), _p(dy), B, x.numel() // B,
                                        self._stream(x)), "siv_sse_bwd")
        return dx, dy

    # ---- the two objectives of one iteration -----------------------------------------
    def sintro_loss_e_fwd(self, vecs, scale, br, bn, bk):
        """vecs = (sse_real, kl_real, sse_fake, kl_fake, sse_rec, kl_rec), each (B,) -> tensor
        [lossE, mean(kl_real), exp_elbo_rec, exp_elbo_fake]."""
        vecs = [v.contiguous() for v in vecs]
        self._check_dev(*vecs)
        B = vecs[0].numel()
        out = torch.empty(4, device=vecs[0].device, dtype=torch.float32)
        _lib.check(self.lib.siv_sintro_loss_e_fwd(*[_p(v) for v in vecs], B, float(scale), float(br), float(bn),
                                                  float(bk), _p(out), self._stream(out)), "siv_sintro_loss_e_fwd")
        return out

    def sintro_loss_e_bwd(self, g, vecs, scale, br, bn, bk):
        vecs = [v.contiguous() for v in vecs]
        B = vecs[0].numel()
        d = torch.empty((6, B), device=g.device, dtype=torch.float32)
        _lib.check(self.lib.siv_sintro_loss_e_bwd(_p(g.contiguous()), _p(vecs[2]), _p(vecs[3]), _p(vecs[4]), _p(vecs[5]),
                                                  B, float(scale), float(br), float(bn), float(bk), _p(d),
                                                  self._stream(d)), "siv_sintro_loss_e_bwd")
        return d

    def sintro_loss_d_fwd(self, vecs, scale, br, bk, gr):
        """vecs = (sse_real, kl_rec, kl_fake, sse_rec_rec, sse_fake_rec) -> tensor
        [lossD, mean(kl_fake), mean(kl_rec), mean(sse_real)]."""
        vecs = [v.contiguous() for v in vecs]
        self._check_dev(*vecs)
        B = vecs[0].numel()
        out = torch.empty(4, device=vecs[0].device, dtype=torch.float32)
        _lib.check(self.lib.siv_sintro_loss_d_fwd(*[_p(v) for v in vecs], B, float(scale), float(br), float(bk),
                                                  float(gr), _p(out), self._stream(out)), "siv_sintro_loss_d_fwd")
        return out

    def sintro_loss_d_bwd(self, g, B, scale, br, bk, gr):
        d = torch.empty((5, B), device=g.device, dtype=torch.float32)
        _lib.check(self.lib.siv_sintro_loss_d_bwd(_p(g.contiguous()), B, float(scale), float(br), float(bk), float(gr),
                                                  _p(d), self._stream(d)), "siv_sintro_loss_d_bwd")
        return d

    # ---- optimiser / rng / helpers ---------------------------------------------------
    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, step: int) -> None:
        self._check_dev(p, g, m, v)
        for t in (p, g, m, v):
            assert t.is_contiguous() and t.dim() == 1 and t.numel() == p.numel()
        _lib.check(self.lib.siv_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), float(lr), float(beta1),
                                          float(beta2), float(eps), int(step), self._stream(p)),
                   "siv_adam_step")

    def adam_step_at(self, p, g, m, v, lr_dev, beta1, beta2, eps, step_dev) -> None:
        """Adam with learning rate (float32 tensor) and step count (int64 tensor) in device memory."""
        self._check_dev(p, g, m, v, step_dev)
        assert lr_dev.is_cuda and lr_dev.dtype == torch.float32 and step_dev.dtype == torch.int64
        _lib.check(self.lib.siv_adam_step_at(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(lr_dev), float(beta1),
                                             float(beta2), float(eps), _p(step_dev), self._stream(p)),
                   "siv_adam_step_at")

    def counter_add(self, counter, inc: int) -> None:
        assert counter.is_cuda and counter.dtype == torch.int64 and counter.numel() == 1
        _lib.check(self.lib.siv_counter_add(_p(counter), int(inc), self._stream(counter)), "siv_counter_add")

    def randn(self, shape, device, seed: int, stream_id: int, epoch=None):
        out = torch.empty(tuple(shape), device=device, dtype=torch.float32)
        self._check_dev(out)
        _lib.check(self.lib.siv_randn_at(_p(out), out.numel(), seed & (2 ** 64 - 1), stream_id & (2 ** 64 - 1),
                                         _p(epoch), self._stream(out)), "siv_randn")
        return out

    def dropout_keep(self, shape, device, p: float, seed: int, stream_id: int, epoch=None):
        out = torch.empty(tuple(shape), device=device, dtype=torch.uint8, memory_format=CL3D)
        self._check_dev(out)
        _lib.check(self.lib.siv_dropout_keep_at(_p(out), out.numel(), float(p), seed & (2 ** 64 - 1),
                                                stream_id & (2 ** 64 - 1), _p(epoch), self._stream(out)),
                   "siv_dropout_keep")
        return out

    def scale_(self, x, a: float) -> None:
        """x *= a in place."""
        self._check_dev(x)
        assert x.is_contiguous()
        _lib.check(self.lib.siv_scale(float(a), _p(x), x.numel(), self._stream(x)), "siv_scale")

    def to_bf16(self, x):
        """bf16 copy (round-to-nearest-even) of a dense fp32 tensor, same shape and strides."""
        self._check_dev(x)
        h = torch.empty_like(x, dtype=torch.bfloat16)        # preserve_format keeps a dense tensor's strides
        assert h.stride() == x.stride()
        _lib.check(self.lib.siv_f32_to_bf16(_p(x), _p(h), x.numel(), self._stream(x)), "siv_f32_to_bf16")
        return h

    def from_bf16(self, h):
        """fp32 copy (exact) of a dense bf16 tensor, same shape and strides."""
        if not h.is_cuda or h.dtype != torch.bfloat16:
            raise _lib.SivError("from_bf16 needs a bf16 tensor on a HIP device")
        x = torch.empty_like(h, dtype=torch.float32)
        assert h.stride() == x.stride()
        _lib.check(self.lib.siv_bf16_to_f32(_p(h), _p(x), x.numel(), self._stream(x)), "siv_bf16_to_f32")
        return x

    def axpby(self, a: float, x, b: float, y) -> None:
        self._check_dev(x, y)
        assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
        _lib.check(self.lib.siv_axpby(float(a), _p(x), float(b), _p(y), x.numel(), self._stream(x)),
                   "siv_axpby")


_backend = None


def get_backend():
    """The HIP backend, created on first use.  Raises when libsiv.so is missing."""
    global _backend
    if _backend is None:
        _backend = HipBackend()
    return _backend


def set_backend(b) -> None:
    """Install another backend object (test-suite hook); ``None`` restores the default."""
    global _backend
    _backend = b
