"""Drop-in for the reference's ``models/lossf.py``: plain-VAE loss helpers.  The per-sample
reductions run in libsiv; only the final mean over the (B,) vector is a torch scalar op."""
from .. import ops


def mse_loss(out, x):
    """models/lossf.py:5-12: batch mean of the per-sample SUM of squared errors."""
    return ops.sse_per_sample(x, out).mean(dim=0)


def kld_loss(mu, logvar):
    """models/lossf.py:14-18."""
    return ops.kl_per_sample(mu, logvar).mean(dim=0)


def normal_loss(x_hat, mu, logvar, x, msew=1, kldw=10):
    """models/lossf.py:20-24 -> (loss, mse, kld)."""
    mse = mse_loss(x_hat, x) * msew
    kld = kld_loss(mu, logvar) * kldw
    return mse + kld, mse, kld
