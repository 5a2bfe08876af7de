"""Drop-in for the reference's ``utils/data_load.py`` with the per-sample arithmetic moved to
the device.

The reference normalises every volume on the host inside ``BrainDataset.__getitem__``
(utils/data_load.py:18-30).  At the throughput this package trains at, that host work (and
the pageable host-to-device copy that follows it) would be the bottleneck, so here

* ``BrainDataset`` keeps the reference's constructor and indexing contract but returns the RAW
  volume ``(1, D, H, W)`` float32 (``transform``, if given, is still applied on the host as in
  the reference);
* ``DeviceLoader`` wraps any loader of raw batches: it stages each batch in pinned memory,
  copies it on a side stream one batch ahead of its use, and applies ``_preprocess`` (and,
  optionally, the flip/shift and random-affine augmentations) in libsiv on the device.  It supports ``len()`` and
  ``iter()`` and yields ``(batch_on_device, labels)``, which is what ``train_soft_intro_vae``
  consumes (its ``batch.to(device)`` is then a no-op).
"""
from __future__ import annotations

import numpy as np
import torch
from torch.utils.data import Dataset

from ..backend import get_backend


class BrainDataset(Dataset):
    """utils/data_load.py:7-32, minus the host-side ``_preprocess`` (see ``DeviceLoader``)."""

    def __init__(self, voxels, labels, transform=None, phase=None):
        self.voxels = voxels
        self.labels = labels
        self.transform = transform
        self.phase = phase

    def __len__(self):
        return len(self.voxels)

    def __getitem__(self, index):
        voxel = self.voxels[index]
        label = self.labels[index]
        if self.transform:
            voxel = self.transform(voxel, self.phase)
        return np.asarray(voxel)[np.newaxis, ].astype("f"), label

    def __call__(self, index):
        return self.__getitem__(index)


def preprocess(voxels: torch.Tensor, cut_range: float = 4.0) -> torch.Tensor:
    """``BrainDataset._preprocess`` (utils/data_load.py:25-30) for a batch of raw device
    volumes ``(B, D, H, W)`` / ``(B, 1, D, H, W)`` -> ``(B, 1, D, H, W)``."""
    return get_backend().preprocess(voxels, cut_range)


class FlipShift:
    """Random flip / integer shift per volume.  ``p_flip`` is the per-axis flip probability,
    ``max_shift`` the largest shift (voxels) per axis.  The draws come from the host generator
    ``torch.Generator`` passed in (or a fresh one seeded from ``seed``), so a run is
    reproducible independently of the device RNG stream."""

    def __init__(self, p_flip=(0.0, 0.0, 0.5), max_shift=(4, 4, 4), seed=0, generator=None):
        self.p_flip = tuple(float(p) for p in p_flip)
        self.max_shift = tuple(int(s) for s in max_shift)
        self.gen = generator if generator is not None else torch.Generator().manual_seed(seed)

    def draw(self, batch: int) -> torch.Tensor:
        flips = torch.stack([(torch.rand(batch, generator=self.gen) < p).to(torch.int32) for p in self.p_flip], 1)
        shifts = torch.stack([torch.randint(-s, s + 1, (batch,), generator=self.gen, dtype=torch.int32)
                              for s in self.max_shift], 1)
        return torch.cat([flips, shifts], 1).contiguous()

    def __call__(self, voxels: torch.Tensor, params: torch.Tensor = None) -> torch.Tensor:
        if params is None:
            params = self.draw(voxels.shape[0])
        return get_backend().flip_shift(voxels, params.to(voxels.device, non_blocking=True))


def affine_matrix(scales, degrees, dims) -> np.ndarray:
    """3x4 matrix (float64) mapping an output voxel index to the input index it samples:
    ``p_in = c + R S (p_out - c)`` with ``c`` the image centre ``(dims - 1) / 2``, ``S`` the
    per-axis scaling and ``R = R2 R1 R0`` the rotations by ``degrees[k]`` about array axis k."""
    s = np.diag(np.asarray(scales, dtype=np.float64))
    r = np.eye(3)
    for ax, deg in enumerate(degrees):
        t = np.deg2rad(float(deg))
        i, j = [k for k in range(3) if k != ax]
        m = np.eye(3)
        m[i, i], m[i, j], m[j, i], m[j, j] = np.cos(t), -np.sin(t), np.sin(t), np.cos(t)
        r = m @ r
    lin = r @ s
    c = (np.asarray(dims, dtype=np.float64) - 1.0) / 2.0
    return np.concatenate([lin, (c - lin @ c)[:, None]], 1)


class RandomAffine:
    """Device form of ``tio.transforms.RandomAffine`` as the reference configures it
    (aug-z-1200main.py:106-119: ``RandomAffine(degrees=10)`` inside ``OneOf(..., p=0.35)``):
    per volume, with probability ``p``, a rotation by ``U(-degrees, degrees)`` about each axis and
    a scaling by ``U(1 - scales, 1 + scales)`` per axis (torchio's defaults: ``scales=0.1``,
    ``isotropic=False``, ``center='image'``, no translation), linear interpolation, padding with
    the image minimum.  torchio and SimpleITK are not installed where this was written, so the
    resampling is defined here (see ``siv_affine_resample``) rather than checked against them:
    it differs from SimpleITK's in the one-voxel shell at the volume boundary, where this one
    blends toward the pad value.  The draws come from a host ``torch.Generator``."""

    def __init__(self, scales=0.1, degrees=10.0, p=1.0, pad="minimum", seed=0, generator=None):
        self.scales = (1.0 - float(scales), 1.0 + float(scales)) if np.isscalar(scales) else tuple(scales)
        self.degrees = (-float(degrees), float(degrees)) if np.isscalar(degrees) else tuple(degrees)
        self.p = float(p)
        self.pad = pad
        self.gen = generator if generator is not None else torch.Generator().manual_seed(seed)

    def draw(self, batch: int, dims) -> torch.Tensor:
        """(batch, 3, 4) float32 matrices; identity for the volumes the transform skips."""
        u = torch.rand(batch, 7, generator=self.gen, dtype=torch.float64).numpy()
        mats = np.empty((batch, 3, 4), dtype=np.float64)
        for b in range(batch):
            if u[b, 0] < self.p:
                sc = self.scales[0] + (self.scales[1] - self.scales[0]) * u[b, 1:4]
                dg = self.degrees[0] + (self.degrees[1] - self.degrees[0]) * u[b, 4:7]
            else:
                sc, dg = np.ones(3), np.zeros(3)
            mats[b] = affine_matrix(sc, dg, dims)
        return torch.from_numpy(mats.astype(np.float32))

    def __call__(self, voxels: torch.Tensor, mats: torch.Tensor = None) -> torch.Tensor:
        if mats is None:
            mats = self.draw(voxels.shape[0], voxels.shape[-3:])
        return get_backend().affine_resample(voxels, mats.to(voxels.device, non_blocking=True), self.pad)


class DeviceLoader:
    """Pinned, double-buffered host-to-device feed + on-device ``_preprocess``.  ``augment`` is a
    callable on the raw device batch (``FlipShift``, ``RandomAffine``) or a sequence of them,
    applied in order before ``_preprocess`` as the reference's ``transform`` is."""

    def __init__(self, loader, device, cut_range: float = 4.0, augment=None):
        self.loader = loader
        self.device = torch.device(device)
        self.cut_range = cut_range
        self.augment = augment
        self._copy_stream = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        """Start the asynchronous copy of one raw host batch; returns (device tensor, event)."""
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream(self.device)
        host = batch if batch.is_pinned() else batch.contiguous().pin_memory()
        with torch.cuda.stream(self._copy_stream):
            dev = host.to(self.device, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._copy_stream)
        return dev, ev, host      # the pinned buffer must outlive the copy

    def __iter__(self):
        it = iter(self.loader)
        try:
            batch, labels = next(it)
        except StopIteration:
            return
        pending = (self._stage(torch.as_tensor(batch, dtype=torch.float32)), labels)
        while pending is not None:
            (dev, ev, _host), labels = pending
            try:
                nb, nl = next(it)
                pending = (self._stage(torch.as_tensor(nb, dtype=torch.float32)), nl)   # copy i+1 during step i
            except StopIteration:
                pending = None
            torch.cuda.current_stream(self.device).wait_event(ev)
            dev.record_stream(torch.cuda.current_stream(self.device))
            if self.augment is not None:
                for aug in (self.augment if isinstance(self.augment, (list, tuple)) else (self.augment,)):
                    dev = aug(dev)
            yield preprocess(dev, self.cut_range), labels
