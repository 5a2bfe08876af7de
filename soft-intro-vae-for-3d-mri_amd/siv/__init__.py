"""siv -- MI355X-native Soft-IntroVAE training path for 3D brain MRI.

Python host side of ``libsiv.so`` (hand-written HIP kernels for gfx950 behind a C ABI).
The module layout mirrors the reference repository so that a caller switches by
changing imports only::

    import siv.models.models as models          # was: import models.models as models
    import siv.utils.my_trainer as trainer      # was: import utils.my_trainer as trainer

PyTorch is used for device memory, streams, autograd bookkeeping and
``torch.distributed``; every tensor operation on the training path runs in libsiv.
"""
from . import backend, ops, rng  # noqa: F401

__all__ = ["backend", "ops", "rng"]
__version__ = "0.1.0"
