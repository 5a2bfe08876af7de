"""Host side of the input pipeline (CPU): the Dataset keeps the reference's contract and does no
arithmetic on the host."""
import numpy as np

import siv.utils.data_load as DL


def test_brain_dataset_contract():
    vols = [np.full((3, 4, 5), i, dtype=np.float64) for i in range(3)]
    ds = DL.BrainDataset(vols, [10, 11, 12])
    assert len(ds) == 3
    v, y = ds[1]
    assert v.shape == (1, 3, 4, 5) and v.dtype == np.float32 and y == 11
    np.testing.assert_array_equal(v[0], vols[1])         # raw: normalisation happens on the device
    calls = []
    ds2 = DL.BrainDataset(vols, [0, 0, 0], transform=lambda v, ph: calls.append(ph) or v + 1, phase="train")
    v2, _ = ds2(2)
    assert calls == ["train"] and float(v2[0, 0, 0, 0]) == 3.0


def test_flip_shift_draws_are_reproducible():
    a = DL.FlipShift(p_flip=(0.5, 0.5, 0.5), max_shift=(2, 3, 4), seed=5).draw(16)
    b = DL.FlipShift(p_flip=(0.5, 0.5, 0.5), max_shift=(2, 3, 4), seed=5).draw(16)
    assert a.dtype.is_floating_point is False and tuple(a.shape) == (16, 6)
    assert (a == b).all()
    assert int(a[:, :3].min()) >= 0 and int(a[:, :3].max()) <= 1
    assert int(a[:, 3].abs().max()) <= 2 and int(a[:, 5].abs().max()) <= 4
