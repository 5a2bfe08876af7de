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


def test_affine_matrix_and_random_affine_draws():
    dims = (8, 12, 10)
    c = (np.array(dims) - 1) / 2
    np.testing.assert_allclose(DL.affine_matrix((1, 1, 1), (0, 0, 0), dims), np.eye(3, 4), atol=1e-15)
    a = DL.affine_matrix((1, 1, 1), (10, -5, 3), dims)
    np.testing.assert_allclose(a[:, :3] @ a[:, :3].T, np.eye(3), atol=1e-12)             # pure rotation
    np.testing.assert_allclose(a[:, :3] @ c + a[:, 3], c, atol=1e-12)                    # about the centre
    r0 = DL.affine_matrix((1, 1, 1), (90, 0, 0), dims)[:, :3]                            # about axis 0
    np.testing.assert_allclose(r0 @ np.array([1.0, 0, 0]), [1, 0, 0], atol=1e-12)
    np.testing.assert_allclose(r0 @ np.array([0, 1.0, 0]), [0, 0, 1], atol=1e-12)
    s = DL.affine_matrix((0.9, 1.0, 1.1), (0, 0, 0), dims)
    np.testing.assert_allclose(np.diag(s[:, :3]), (0.9, 1.0, 1.1))
    m1 = DL.RandomAffine(degrees=10, p=0.35, seed=3).draw(64, dims)
    m2 = DL.RandomAffine(degrees=10, p=0.35, seed=3).draw(64, dims)
    assert m1.dtype.is_floating_point and tuple(m1.shape) == (64, 3, 4) and (m1 == m2).all()
    ident = (m1 == np.eye(3, 4, dtype=np.float32)).all(2).all(1)
    assert 0.45 < float(ident.float().mean()) < 0.85                                    # ~65 % skipped
    lin = m1[~ident][:, :, :3].double().numpy()
    sv = np.linalg.svd(lin, compute_uv=False)                                           # the three scalings
    assert sv.min() >= 0.9 - 1e-6 and sv.max() <= 1.1 + 1e-6
