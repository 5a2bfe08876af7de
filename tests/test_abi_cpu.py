"""The C-ABI library loads without a GPU and exports every symbol include/siv.h declares; argument
validation (which happens before any HIP call) behaves as documented.  No compute calls here."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "siv.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(siv_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    assert len(syms) >= 37
    for must in ("siv_conv3d_k3_fwd", "siv_conv3d_k3_dgrad", "siv_conv3d_k3_wgrad", "siv_bn_act_fwd",
                 "siv_bn_act_bwd", "siv_kl_fwd", "siv_sse_fwd", "siv_reparam_fwd", "siv_adam_step"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from siv import _lib
    lib = _lib.load()
    for name in declared_symbols():
        assert hasattr(lib, name), f"libsiv.so does not export {name}"
    # the ctypes table binds everything that takes device pointers
    helpers = {"siv_last_error", "siv_version", "siv_workspace_bytes", "siv_last_conv_kernel", "siv_pack_cache_hits",
               "siv_conv1_bnact_bits_bytes"}
    assert set(declared_symbols()) - helpers == set(_lib.SIGNATURES)
    assert lib.siv_version() >= 100
    assert lib.siv_workspace_bytes() >= (64 << 20)
    assert lib.siv_conv1_bnact_bits_bytes(2, 3, 4, 5, 64) == 2 * 3 * 4 * 5 * 16


def test_argument_validation_needs_no_gpu():
    from siv import _lib
    lib = _lib.load()
    assert lib.siv_set_conv_path(7) == -1 and b"path" in lib.siv_last_error()
    assert lib.siv_set_conv_path(0) == 0
    rc = lib.siv_conv3d_k3_fwd(None, None, None, None, 1, 4, 4, 4, 8, 8, 1, None, 0, None)
    assert rc == -1 and b"null pointer" in lib.siv_last_error()
    one = ctypes.c_void_p(256)      # never dereferenced: shape checks come first
    rc = lib.siv_conv3d_k3_fwd(one, one, None, one, 1, 4, 4, 4, 8, 8, 3, None, 0, None)
    assert rc == -1 and b"ups" in lib.siv_last_error()
    rc = lib.siv_conv3d_k3_fwd(one, one, None, one, 1, 5, 4, 4, 8, 8, 2, None, 0, None)
    assert rc == -1 and b"even" in lib.siv_last_error()
    rc = lib.siv_bn_act_fwd(one, one, one, one, one, None, None, 0.0, 0.2, 3, 1e-5, one, 1, 4, 4, 4, 8, None)
    assert rc == -1 and b"pool" in lib.siv_last_error()
    rc = lib.siv_bn_stats(one, 100, 8, one, one, None, None, None, 0.1, None, 0, None)
    assert rc == -2 and b"workspace" in lib.siv_last_error()
    rc = lib.siv_adam_step(one, one, one, one, 10, 1e-3, 0.9, 0.999, 1e-8, 0, None)
    assert rc == -1


def test_product_refuses_cpu_tensors():
    import torch
    import siv.backend as sb
    sb.set_backend(None)
    b = sb.get_backend()
    with pytest.raises(Exception, match="no CPU fallback"):
        b.conv3d_k3_fwd(torch.zeros(1, 8, 4, 4, 4), torch.zeros(64, 8, 3, 3, 3), None, 1)
