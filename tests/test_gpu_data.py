"""Input pipeline on a real MI355X: siv_preprocess / siv_flip_shift through the C ABI against the
reference fixture (g10) and the oracle; DeviceLoader end to end.

Tolerance: 1e-4 of the output range (outputs are in [0, 1]).  The clip threshold is 4*std: numpy
computes std in the input dtype, the kernel in float64; the two thresholds differ by ~1e-7
relative, which moves the clipped maximum -- and with it every output -- by the same amount."""
import os

import numpy as np
import pytest
import torch

from oracle import siv_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def B():
    import siv.backend as sb
    sb.set_backend(None)
    return sb.get_backend()


@pytest.mark.parametrize("name", ["brain_f32", "signed_f32", "f64", "i16"])
def test_preprocess_matches_reference_fixture(B, name):
    g = np.load(os.path.join(GOLDEN, "g10_preprocess.npz"), allow_pickle=False)
    raw = torch.from_numpy(g["in." + name].astype(np.float32))
    out = B.preprocess(raw[None].cuda()).cpu().numpy()
    assert out.shape == (1,) + g["out." + name].shape
    np.testing.assert_allclose(out[0], g["out." + name], rtol=0, atol=1e-4)


def test_preprocess_batch_matches_oracle_and_properties(B):
    rs = np.random.RandomState(3)
    vols = np.stack([np.abs(rs.randn(80, 96, 80)).astype(np.float32) * (i + 1) - 0.3 * i for i in range(3)])
    out = B.preprocess(torch.from_numpy(vols).cuda()).cpu().numpy()
    assert out.shape == (3, 1, 80, 96, 80)
    for i in range(3):                                   # volumes are normalised independently
        np.testing.assert_allclose(out[i], O.preprocess(vols[i]), rtol=0, atol=1e-4)
        assert out[i].min() == 0.0 and out[i].max() == 1.0
    # idempotent up to the clip: a volume already in [0,1] whose 4*std exceeds 1 is unchanged
    u = rs.rand(1, 16, 16, 16).astype(np.float32)
    u[0, 0, 0, 0], u[0, 0, 0, 1] = 0.0, 1.0
    np.testing.assert_allclose(B.preprocess(torch.from_numpy(u).cuda()).cpu().numpy()[0], u, atol=1e-6)
    # a constant volume gives NaN, as numpy's 0/0 does in the reference
    c = B.preprocess(torch.full((1, 4, 4, 4), 2.0).cuda())
    assert torch.isnan(c).all()


def test_flip_shift_matches_definition(B):
    rs = np.random.RandomState(4)
    vols = rs.rand(5, 6, 7, 9).astype(np.float32)
    params = np.array([[0, 0, 0, 0, 0, 0], [1, 0, 1, 0, 0, 0], [0, 1, 0, 2, -1, 3], [1, 1, 1, -2, 3, -4],
                       [0, 0, 1, 6, 0, 0]], dtype=np.int32)
    out = B.flip_shift(torch.from_numpy(vols).cuda(), torch.from_numpy(params).cuda()).cpu().numpy()
    for i in range(5):
        np.testing.assert_array_equal(out[i], O.flip_shift(vols[i], params[i, :3], params[i, 3:]))
    # flipping twice restores the volume exactly
    p = torch.tensor([[1, 1, 1, 0, 0, 0]] * 5, dtype=torch.int32).cuda()
    twice = B.flip_shift(B.flip_shift(torch.from_numpy(vols).cuda(), p), p).cpu().numpy()
    np.testing.assert_array_equal(twice, vols)


def test_input_errors(B):
    import siv._lib as L
    x = torch.zeros(2, 8, device="cuda")
    with pytest.raises(L.SivError):
        L.check(B.lib.siv_preprocess(x.data_ptr(), x.data_ptr(), 0, 8, 4.0, x.data_ptr(), 64, None), "siv_preprocess")
    with pytest.raises(L.SivError):      # workspace too small
        L.check(B.lib.siv_preprocess(x.data_ptr(), x.data_ptr(), 2, 8, 4.0, x.data_ptr(), 64, None), "siv_preprocess")
    with pytest.raises(L.SivError):      # in place
        L.check(B.lib.siv_flip_shift(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1, 2, 2, 2, None), "siv_flip_shift")


def test_device_loader_feeds_preprocessed_batches(B):
    import siv.utils.data_load as DL
    rs = np.random.RandomState(7)
    vols = [np.abs(rs.randn(16, 24, 16)) * 100 for _ in range(5)]
    ds = DL.BrainDataset(vols, list(range(5)))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)
    dl = DL.DeviceLoader(loader, "cuda")
    assert len(dl) == 3
    seen = []
    for batch, labels in dl:
        assert batch.is_cuda and batch.shape[1:] == (1, 16, 24, 16)
        for j, lab in enumerate(labels.tolist()):
            np.testing.assert_allclose(batch[j].cpu().numpy(), O.preprocess(vols[lab].astype(np.float32)),
                                       rtol=0, atol=1e-4)
            seen.append(lab)
    assert seen == [0, 1, 2, 3, 4]
    # with augmentation: same draws -> same result as flip_shift followed by preprocess
    aug = DL.FlipShift(p_flip=(0.5, 0.5, 0.5), max_shift=(2, 2, 2), seed=9)
    expect = DL.FlipShift(p_flip=(0.5, 0.5, 0.5), max_shift=(2, 2, 2), seed=9)
    for batch, labels in DL.DeviceLoader(loader, "cuda", augment=aug):
        prm = expect.draw(batch.shape[0]).numpy()
        for j, lab in enumerate(labels.tolist()):
            ref = O.preprocess(O.flip_shift(vols[lab].astype(np.float32), prm[j, :3], prm[j, 3:]))
            np.testing.assert_allclose(batch[j].cpu().numpy(), ref, rtol=0, atol=1e-4)


def test_affine_resample_matches_definition(B):
    """siv_affine_resample against the oracle's definition.  Tolerance: the kernel computes the
    sample position in fp32 (error ~1e-5 voxel at index ~100) and the data vary by O(1) per voxel,
    so 2e-4 absolute on data in [0, 1) (+ the pad value's magnitude)."""
    import siv.utils.data_load as DL
    rs = np.random.RandomState(21)
    vols = rs.rand(4, 40, 48, 36).astype(np.float32)
    vols[1] -= 0.5
    dims = vols.shape[1:]
    mats = np.stack([np.eye(3, 4),
                     DL.affine_matrix((1.0, 1.0, 1.0), (10.0, 0.0, 0.0), dims),
                     DL.affine_matrix((0.9, 1.1, 1.05), (-7.0, 4.0, 9.5), dims),
                     DL.affine_matrix((1.1, 0.95, 0.9), (3.0, -10.0, 1.0), dims)]).astype(np.float32)
    x, m = torch.from_numpy(vols).cuda(), torch.from_numpy(mats).cuda()
    for pad in ("minimum", 0.0, -1.25):
        out = B.affine_resample(x, m, pad).cpu().numpy()
        for i in range(4):
            np.testing.assert_allclose(out[i], O.affine_resample(vols[i], mats[i], pad), rtol=0, atol=2e-4)
        np.testing.assert_array_equal(out[0], vols[0])                  # the identity is exact
    # RandomAffine: same draws -> same volumes; skipped volumes come back unchanged
    aug = DL.RandomAffine(degrees=10, p=0.5, seed=9)
    mm = aug.draw(4, dims)
    got = aug(x, mm).cpu().numpy()
    for i in range(4):
        np.testing.assert_allclose(got[i], O.affine_resample(vols[i], mm[i].numpy()), rtol=0, atol=2e-4)
    # bad arguments are reported, not executed
    import siv._lib as L
    with pytest.raises(RuntimeError):
        L.check(B.lib.siv_affine_resample(x.data_ptr(), m.data_ptr(), x.data_ptr(), 1, 2, 2, 2, 0, 0.0, None, 0, None),
                "siv_affine_resample")


def test_device_loader_with_affine_then_preprocess(B):
    import siv.utils.data_load as DL
    rs = np.random.RandomState(5)
    vols = [np.abs(rs.randn(16, 20, 12)).astype(np.float32) for _ in range(4)]
    ds = DL.BrainDataset(vols, list(range(4)))
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False)
    aug = DL.RandomAffine(degrees=10, p=1.0, seed=2)
    ref_aug = DL.RandomAffine(degrees=10, p=1.0, seed=2)
    for batch, labels in DL.DeviceLoader(loader, "cuda:0", augment=[aug]):
        mats = ref_aug.draw(2, (16, 20, 12)).numpy()
        for j, lab in enumerate(labels.tolist()):
            ref = O.preprocess(O.affine_resample(vols[lab], mats[j]).astype(np.float32))
            np.testing.assert_allclose(batch[j].cpu().numpy(), ref, rtol=0, atol=5e-4)
