"""Shared comparison helpers of the parity tests (test infrastructure)."""
import numpy as np
import torch


def to_np(a):
    return a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float32)


def check_max(a, b, what, tol=1e-4, floor=0.0):
    """max|a - b| <= tol * max|b| + floor."""
    a, b = to_np(a), to_np(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what
    err = float(np.abs(a - b).max())
    lim = tol * float(np.abs(b).max()) + floor
    assert err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e} (max|ref| {np.abs(b).max():.3e})"
    return err / (float(np.abs(b).max()) + 1e-30)


def check_params_after_adam(mine, ref, grad_ref, lr_steps, what):
    """Adam's first steps move every element by ~lr*sign(g): where |g| is far above rounding noise the
    update must agree tightly; elsewhere the sign of a noise-level gradient decides and only
    |delta| <= 2*lr per step can be asserted.  ``grad_ref`` None = the parameter received no gradient
    (Adam skips it) and must be unchanged."""
    mine, ref = to_np(mine), to_np(ref)
    if grad_ref is None:
        np.testing.assert_allclose(mine, ref, rtol=1e-5, atol=2e-6, err_msg=what)
        return
    grad_ref = to_np(grad_ref)
    solid = np.abs(grad_ref) > 2e-2 * np.abs(grad_ref).max()
    np.testing.assert_allclose(mine[solid], ref[solid], rtol=1e-5, atol=5e-6, err_msg=what + " (solid)")
    np.testing.assert_allclose(mine, ref, rtol=1e-5, atol=2 * lr_steps + 2e-6, err_msg=what)
