"""Per-operator parity: every libsiv entry point, called through the C ABI (ctypes) on a real
MI355X, against the CPU oracle on the same seeded inputs.

Tolerance (the bar this build is held to): 1e-4 relative fp32, measured as
max|hip - oracle| <= 1e-4 * max|oracle| per output tensor.  The backward of the BN/activation
tail contains LeakyReLU/ReLU kinks: an element whose pre-activation is within rounding of zero
may legitimately take the other branch, so those comparisons allow a handful of outlier
elements (reported) instead of loosening the tolerance."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import siv_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def B():
    import siv.backend as sb
    sb.set_backend(None)
    b = sb.get_backend()
    assert b.name == "hip"
    yield b
    b.set_conv_path(0)


def dev(t):
    return None if t is None else t.to("cuda")


def relerr(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def check(a, b, what, tol=TOL, max_outliers=0):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.isfinite(a).all(), what
    scale = float(b.abs().max()) + 1e-30
    bad = ((a - b).abs() > tol * scale)
    nbad = int(bad.sum())
    assert nbad <= max_outliers, f"{what}: {nbad} elements beyond {tol} of max (rel err {relerr(a, b):.3e})"


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


# (Cin, Cout, dims, N, ups, path) -- path 0 = automatic dispatch, 1 = generic kernels, 3 = Winograd
CONV_CASES = [
    (1, 4, (6, 7, 5), 1, 1, 0), (4, 4, (6, 7, 5), 3, 1, 0), (4, 8, (8, 12, 8), 1, 1, 0),
    (8, 4, (8, 12, 8), 3, 1, 0), (4, 1, (6, 7, 5), 3, 1, 0), (3, 5, (5, 4, 3), 2, 1, 0),
    (4, 8, (8, 12, 8), 2, 2, 0), (6, 6, (4, 6, 4), 2, 2, 0),
    # single-channel-side kernels (encoder stem / decoder output layer)
    (1, 8, (8, 12, 8), 2, 1, 0), (1, 64, (9, 10, 11), 2, 1, 0), (1, 12, (6, 7, 5), 1, 1, 0),
    (8, 1, (8, 12, 8), 2, 1, 0), (64, 1, (9, 10, 11), 2, 1, 0), (16, 1, (6, 7, 5), 3, 1, 0),
    (32, 1, (8, 12, 8), 1, 1, 0), (24, 1, (6, 7, 5), 2, 1, 0), (1, 32, (5, 9, 8), 2, 1, 0), (1, 24, (6, 7, 5), 1, 1, 0),
    # ... long enough in z for the column-walking 64 -> 1 kernel (forward, and 1 -> 64 input gradient)
    (64, 1, (17, 9, 18), 2, 1, 0), (64, 1, (16, 8, 16), 1, 1, 0), (1, 64, (33, 12, 20), 1, 1, 0),
    # ... 12 x 16 columns with ragged last tiles in H and W and an odd plane count; 13 rows stay on 8 x 16 columns
    (64, 1, (19, 44, 36), 2, 1, 0), (64, 1, (16, 13, 16), 1, 1, 0), (1, 64, (24, 24, 32), 3, 1, 0),
    # fp32 matrix-core kernels
    (8, 64, (8, 12, 8), 1, 1, 0), (64, 64, (8, 12, 8), 2, 1, 0), (64, 64, (5, 7, 9), 1, 1, 0),
    (16, 128, (10, 12, 10), 2, 1, 0), (128, 64, (4, 8, 8), 1, 1, 0), (64, 128, (3, 5, 17), 1, 1, 0),
    (64, 64, (8, 12, 8), 2, 2, 0), (128, 64, (8, 8, 16), 1, 2, 0), (256, 256, (2, 4, 8), 1, 1, 0),
    (64, 64, (16, 16, 16), 1, 1, 0),
    # channel counts that are not multiples of the 8-channel chunk / 32-64-channel tiles
    # (vae_main.py's 12-24-32-48 and 600z_main.py's 16-32-64-128 widths)
    (12, 24, (8, 12, 8), 2, 1, 0), (16, 16, (9, 10, 11), 1, 1, 0), (24, 12, (8, 12, 8), 1, 2, 0),
    (48, 32, (4, 8, 8), 2, 1, 0), (32, 48, (5, 6, 5), 3, 1, 0), (128, 16, (4, 8, 8), 1, 1, 0),
    (4, 8, (8, 8, 8), 1, 1, 0), (20, 100, (4, 4, 8), 1, 1, 0),
    # ... narrow layers (Cin, Cout <= 32) whose weight gradient splits the voxels over the four waves, with two
    # taps per 32-column tile when Cin <= 16: several bricks per block, ragged bricks, 10-wide bricks
    (16, 16, (16, 24, 40), 2, 1, 0), (16, 32, (8, 12, 20), 3, 1, 0), (32, 32, (9, 10, 11), 2, 1, 0),
    (32, 16, (8, 16, 16), 1, 1, 0), (8, 24, (7, 9, 13), 2, 1, 0), (12, 8, (6, 10, 10), 1, 1, 0),
    # ... and behind an upsample (collapsed-tap weight gradient in its narrow modes; 10-wide low-resolution bricks)
    (16, 16, (8, 16, 16), 2, 2, 0), (8, 32, (12, 8, 40), 1, 2, 0), (32, 16, (8, 8, 40), 2, 2, 0),
    (16, 32, (16, 24, 40), 1, 2, 0),
    # ... whose forward runs on the collapsed-tap / product-form kernels with a partial 64-channel output tile
    (32, 16, (16, 24, 40), 2, 2, 0), (8, 48, (6, 10, 14), 2, 2, 0), (64, 32, (8, 8, 16), 1, 2, 6), (16, 80, (4, 24, 40), 1, 2, 5),
    # ... and whose input gradient runs as the stride-2 convolution with a partial 64-channel tile of dx
    # (16-tap and box forms, 10-wide bricks, a whole tile plus a partial one)
    (16, 8, (6, 10, 14), 2, 2, 0), (32, 32, (16, 24, 40), 1, 2, 0), (80, 16, (4, 24, 40), 1, 2, 0), (48, 8, (2, 2, 2), 1, 2, 6),
    # widths 10 / 20 (the two deepest levels of an 80x96x80 input): 10-wide bricks in 3-wave blocks
    (8, 64, (10, 24, 20), 1, 1, 0), (16, 64, (10, 24, 20), 1, 2, 0), (8, 128, (5, 12, 10), 2, 1, 0),
    (64, 64, (10, 12, 10), 1, 1, 0), (8, 64, (4, 6, 10), 1, 2, 0),
    # Winograd F(2,3)-along-W kernels (path 3; falls back to the direct kernels where unsupported)
    (8, 64, (8, 12, 8), 1, 1, 3), (64, 64, (5, 7, 9), 1, 1, 3), (64, 128, (8, 8, 16), 2, 1, 3),
    (64, 64, (8, 12, 8), 2, 2, 3), (16, 64, (10, 24, 20), 1, 1, 3), (128, 64, (4, 8, 8), 1, 2, 3),
    (72, 64, (4, 6, 10), 1, 1, 3), (64, 64, (16, 16, 16), 1, 1, 3),
    # upsampled input through the collapsed-tap kernel (automatic path): odd low-resolution extents,
    # partial bricks, 10- and 20-wide low-resolution volumes, two output-channel tiles
    (8, 64, (6, 10, 14), 2, 2, 0), (64, 128, (4, 24, 40), 1, 2, 0), (24, 64, (2, 2, 2), 3, 2, 0),
    (16, 64, (6, 26, 18), 1, 2, 0),
    # ... whose input gradient runs as a stride-2 4x4x4 convolution when Cin % 64 == 0
    (64, 8, (6, 10, 14), 2, 2, 0), (64, 16, (10, 24, 20), 1, 2, 0), (64, 64, (2, 2, 2), 1, 2, 0),
    (128, 24, (12, 20, 18), 1, 2, 0),
    # path 5: Winograd with fp32 products formed from bf16 triples on the bf16 matrix cores
    (8, 64, (8, 12, 8), 1, 1, 5), (64, 64, (5, 7, 9), 2, 1, 5), (64, 128, (8, 8, 16), 2, 1, 5),
    (16, 64, (10, 24, 20), 1, 1, 5), (72, 64, (4, 6, 10), 1, 1, 5), (64, 64, (16, 16, 16), 1, 1, 5),
    (128, 64, (8, 8, 16), 1, 2, 5), (8, 64, (6, 10, 14), 2, 2, 5), (64, 128, (4, 24, 40), 1, 2, 5),
    (64, 64, (2, 2, 2), 1, 2, 5), (64, 16, (10, 24, 20), 1, 2, 5),
    # path 6: two-dimensional Winograd F(2x2,3x3)
    (8, 64, (8, 12, 8), 1, 1, 6), (64, 64, (5, 7, 9), 2, 1, 6), (64, 128, (8, 8, 16), 2, 1, 6),
    (16, 64, (10, 24, 20), 1, 1, 6), (72, 64, (4, 6, 10), 1, 1, 6), (64, 64, (16, 16, 16), 1, 1, 6),
    # ... and the Winograd F(3x3,2x2) weight gradient (multiples of 64 channels): ragged planes, several
    # channel tiles, columns cut along z (D = 40 with one column)
    (128, 64, (9, 10, 12), 1, 1, 6), (128, 128, (6, 12, 20), 2, 1, 6), (64, 64, (17, 8, 24), 3, 1, 6),
    (256, 256, (2, 4, 8), 1, 1, 6), (64, 64, (40, 8, 8), 1, 1, 6), (64, 192, (3, 9, 9), 2, 1, 6),
    # ... with 6 x 10-voxel columns (10- and 20-wide planes), partial columns included
    (64, 64, (10, 12, 10), 2, 1, 6), (128, 64, (5, 13, 20), 1, 1, 6), (64, 128, (20, 7, 30), 1, 1, 6),
    # ... and its form for an upsampled input (3x3 live positions, z planes paired): odd low-resolution
    # extents, partial columns, two channel tiles, columns cut along z
    (64, 64, (8, 12, 8), 2, 2, 6), (128, 64, (8, 8, 16), 1, 2, 6), (64, 128, (4, 24, 40), 1, 2, 6),
    (64, 64, (2, 2, 2), 1, 2, 6), (64, 64, (20, 8, 8), 1, 2, 6), (64, 64, (6, 10, 14), 3, 2, 6),
    # ... and the product form of the upsampled forward (Cin % 8 == 0, Cout % 64 == 0): odd low-resolution
    # extents, 10- and 20-wide low-resolution volumes, three samples
    (8, 64, (6, 10, 14), 2, 2, 6), (16, 64, (6, 26, 18), 1, 2, 6), (24, 64, (2, 2, 2), 3, 2, 6),
    (64, 128, (4, 24, 40), 1, 2, 6), (16, 64, (10, 24, 20), 1, 2, 6),
    # ... with more bricks than CUs, so that a block of the 2-D Winograd kernel chains several bricks into one
    # chunk stream: uneven brick counts per block (375, 315 bricks on 256 blocks), brick counts that are not a
    # multiple of 8 (no XCD remap), four and eight chunks, 6 x 10-voxel brick planes (264 bricks), two
    # output-channel tiles (144 bricks x 2 on 128 x 2 blocks); forward and input gradient
    (32, 64, (20, 24, 40), 5, 1, 6), (64, 64, (12, 24, 40), 7, 1, 6), (64, 64, (12, 24, 20), 11, 1, 6),
    (64, 128, (16, 24, 24), 4, 1, 6),
    # box-form input gradient behind an upsample with a whole cache line (32 channels) per staged voxel
    # (Cout % 32 == 0, low-resolution width not a multiple of 10): three channel groups with a whole and a
    # partial 64-channel tile of dx and partial bricks along every axis; several whole bricks per sample on
    # the automatic path; one channel group
    (80, 96, (12, 20, 18), 2, 2, 6), (64, 64, (16, 32, 32), 2, 2, 0), (16, 32, (10, 16, 16), 1, 2, 6),
    # fewer (brick, output tile) units than CUs: the 2-D Winograd kernel splits the input channels four ways
    # (768 / 384 items chained three per block) and a second kernel adds the partial outputs; forward and
    # input gradient, four and two output-channel tiles, 16 and 32 chunks
    (128, 256, (10, 12, 10), 8, 1, 6), (256, 128, (10, 12, 10), 8, 1, 0),
    # ... and two ways where the last of a few rounds of blocks would be at most 60 % full (320 units -> 640 items)
    (128, 128, (20, 24, 20), 4, 1, 6),
    # four-plane box form of the upsampled input gradient with its step loop cut in two (520 one-brick samples:
    # a second round of blocks would be 2 % full) -- the four-part cut is taken by the small 10-wide cases above
    (16, 64, (4, 4, 20), 520, 2, 6),
    # the same wide shapes through the generic kernels (cross-check of the dispatcher)
    (64, 64, (5, 7, 9), 1, 1, 1), (8, 64, (4, 6, 4), 1, 2, 1),
]


@pytest.mark.parametrize("cin,cout,dims,n,ups,path", CONV_CASES)
def test_conv3d_k3_fwd_dgrad_wgrad(B, cin, cout, dims, n, ups, path):
    gen = torch.Generator().manual_seed(cin * 1000 + cout * 10 + dims[0] + ups)
    D, H, W = dims
    x = rnd(gen, n, cin, D // ups, H // ups, W // ups)
    w = rnd(gen, cout, cin, 3, 3, 3, scale=(2.0 / (27 * cin)) ** 0.5)
    b = rnd(gen, cout)
    dy = rnd(gen, n, cout, D, H, W)
    B.set_conv_path(path)
    try:
        y = B.conv3d_k3_fwd(dev(x), dev(w), dev(b), ups)
        y_nb = B.conv3d_k3_fwd(dev(x), dev(w), None, ups)
        dx = B.conv3d_k3_dgrad(dev(dy), dev(w), ups)
        dw, db = B.conv3d_k3_wgrad(dev(x), dev(dy), ups, True)
        dw2, db2 = B.conv3d_k3_wgrad(dev(x), dev(dy), ups, False)
    finally:
        B.set_conv_path(0)
    assert db2 is None
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = O.conv3d_k3(xr, wr, br, ups)
    gx, gw, gb = torch.autograd.grad(yr, [xr, wr, br], dy)
    check(y, yr, "fwd")
    check(y_nb, O.conv3d_k3(x, w, None, ups), "fwd (no bias)")
    check(dx, gx, "dgrad")
    check(dw, gw, "wgrad")
    check(dw2, gw, "wgrad (no bias)")
    check(db, gb, "bias grad")
    assert y.is_contiguous(memory_format=torch.channels_last_3d)


@pytest.mark.parametrize("path,ups", [(0, 1), (6, 1), (0, 2), (6, 2)])
def test_conv_wgrad_accumulates(B, path, ups):
    """accumulate = 1 adds to dw / dbias instead of overwriting them (C-ABI call)."""
    import siv._lib as L
    from siv.backend import cl, _p
    gen = torch.Generator().manual_seed(11)
    x = cl(dev(rnd(gen, 2, 64, 6 // ups, 8 // ups, 8 // ups)))
    dy = cl(dev(rnd(gen, 2, 64, 6, 8, 8)))
    base = cl(dev(rnd(gen, 64, 64, 3, 3, 3)))
    bias0 = dev(rnd(gen, 64))
    B.set_conv_path(path)
    try:
        dw_ref, db_ref = B.conv3d_k3_wgrad(x, dy, ups, True)
        dw, db = base.clone(memory_format=torch.preserve_format), bias0.clone()
        ws, wsb = B._workspace(dy)
        L.check(B.lib.siv_conv3d_k3_wgrad(_p(x), _p(dy), _p(dw), _p(db), 2, 6, 8, 8, 64, 64, ups, 1, ws, wsb,
                                          B._stream(dy)), "siv_conv3d_k3_wgrad")
    finally:
        B.set_conv_path(0)
    check(dw, base + dw_ref, "wgrad accumulate")
    check(db, bias0 + db_ref, "bias grad accumulate")


def test_split_contraction_falls_back_without_workspace_room(B):
    """The deep-level launches that cut their channel contraction into parts keep the partial outputs in the
    tail of the workspace; with a workspace that only holds the packed weights they run unsplit.  Both forms
    are checked against each other (summation order differs: tolerance of the other convolution tests) for
    the forward, the forward with statistics and the input gradient behind an upsample."""
    import siv._lib as L
    from siv.backend import cl, _p
    gen = torch.Generator().manual_seed(5)
    N, D, H, W, C = 8, 10, 12, 10, 256
    x = cl(dev(rnd(gen, N, C, D, H, W)))
    w = cl(dev(rnd(gen, C, C, 3, 3, 3) * 0.05))
    b = dev(rnd(gen, C))
    small = torch.empty(20 << 20, dtype=torch.uint8, device="cuda")      # 12.6 MB of packed weights fit, 39 MB more do not
    y_ref = B.conv3d_k3_fwd(x, w, b, 1)
    assert B.last_conv_kernel() == "wino2d"
    y = torch.empty_like(y_ref)
    L.check(B.lib.siv_conv3d_k3_fwd(_p(x), _p(w), _p(b), _p(y), N, D, H, W, C, C, 1, small.data_ptr(), small.numel(),
                                    B._stream(x)), "siv_conv3d_k3_fwd")
    check(y, y_ref, "forward, unsplit against split")
    y2, mean_ref, var_ref = B.conv3d_k3_fwd_stats(x, w, b, 1, None, None, None, 0.1)
    mean, var = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    L.check(B.lib.siv_conv3d_k3_fwd_bnstats(_p(x), _p(w), _p(b), _p(y), N, D, H, W, C, C, 1, _p(mean), _p(var), None, None,
                                            None, 0.1, small.data_ptr(), small.numel(), B._stream(x)),
            "siv_conv3d_k3_fwd_bnstats")
    check(y, y2, "forward with statistics, unsplit against split")
    check(mean, mean_ref, "mean")
    check(var, var_ref, "var")
    # input gradient behind an upsample: 256 -> 128 @20x24x20 (four parts of the step loop)
    dy = cl(dev(rnd(gen, N, 128, 2 * D, 2 * H, 2 * W)))
    wu = cl(dev(rnd(gen, 128, C, 3, 3, 3) * 0.05))
    dx_ref = B.conv3d_k3_dgrad(dy, wu, 2)
    dx = torch.empty_like(dx_ref)
    L.check(B.lib.siv_conv3d_k3_dgrad_up(_p(dy), _p(wu), _p(dx), N, 2 * D, 2 * H, 2 * W, C, 128, small.data_ptr(),
                                         small.numel(), B._stream(dy)), "siv_conv3d_k3_dgrad_up")
    check(dx, dx_ref, "upsampled input gradient, unsplit against split")


def test_conv_rejects_bad_arguments(B):
    import siv._lib as L
    x = torch.zeros(1, 4, 3, 3, 3, device="cuda")
    w = torch.zeros(4, 4, 3, 3, 3, device="cuda")
    with pytest.raises(L.SivError):
        B.conv3d_k3_fwd(x, w, None, 3)                      # ups must be 1 or 2
    with pytest.raises(L.SivError):
        B.conv3d_k3_fwd(x.cpu(), w.cpu(), None, 1)          # no CPU fallback
    lib = L.load()
    assert lib.siv_conv3d_k3_fwd(None, None, None, None, 1, 1, 1, 1, 1, 1, 1, None, 0, None) != 0
    assert b"null" in lib.siv_last_error()
    B.set_conv_path(2)
    try:
        with pytest.raises(L.SivError):
            B.conv3d_k3_fwd(torch.zeros(1, 3, 4, 4, 4, device="cuda"),
                            torch.zeros(5, 3, 3, 3, 3, device="cuda"), None, 1)
    finally:
        B.set_conv_path(0)


@pytest.mark.parametrize("cin,cout", [(8, 1), (1, 8), (256, 1), (6, 10), (64, 2)])
def test_conv1x1(B, cin, cout):
    gen = torch.Generator().manual_seed(cin + cout)
    x = rnd(gen, 3, cin, 4, 6, 5)
    w = rnd(gen, cout, cin, 1, 1, 1)
    b = rnd(gen, cout)
    dy = rnd(gen, 3, cout, 4, 6, 5)
    y = B.conv1x1_fwd(dev(x), dev(w), dev(b))
    dx = B.conv1x1_dgrad(dev(dy), dev(w))
    dw, db = B.conv1x1_wgrad(dev(x), dev(dy), True)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv3d(xr, wr, br)
    gx, gw, gb = torch.autograd.grad(yr, [xr, wr, br], dy)
    check(y, yr, "fwd")
    check(dx, gx, "dgrad")
    check(dw, gw, "wgrad")
    check(db, gb, "bias grad")


def test_upsample2_bwd(B):
    import siv._lib as L
    gen = torch.Generator().manual_seed(3)
    dy = rnd(gen, 2, 6, 4, 6, 8)
    dx = torch.empty(2, 6, 2, 3, 4, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    g = dev(dy).contiguous(memory_format=torch.channels_last_3d)
    L.check(B.lib.siv_upsample2_bwd(g.data_ptr(), dx.data_ptr(), 2, 4, 6, 8, 6, 0), "upsample2_bwd")
    torch.cuda.synchronize()
    check(dx, F.avg_pool3d(dy, 2) * 8, "upsample2_bwd")


@pytest.mark.parametrize("c,dims,n", [(4, (6, 7, 5), 3), (64, (8, 12, 8), 2), (12, (4, 4, 4), 1),
                                      (256, (2, 3, 2), 3), (1, (5, 5, 5), 2), (300, (2, 2, 2), 2),
                                      # more than 1024 rows of per-block partial sums: folded onto 256 rows first
                                      # (1152 rows of 6 columns; 1040 rows of 16 columns through the 16-byte kernels)
                                      (3, (32, 48, 48), 1), (8, (64, 130, 128), 1)])
def test_bn_stats(B, c, dims, n):
    gen = torch.Generator().manual_seed(c)
    y = rnd(gen, n, c, *dims) * (1 + torch.arange(c).view(1, c, 1, 1, 1) % 5) + \
        10.0 * torch.randn(c, generator=gen).view(1, c, 1, 1, 1)       # mean >> std in places
    rm, rv = rnd(gen, c), torch.rand(c, generator=gen) + 0.5
    nbt = torch.tensor(7, dtype=torch.int64)
    rm_d, rv_d, nbt_d = dev(rm.clone()), dev(rv.clone()), dev(nbt.clone())
    mean, var = B.bn_stats(dev(y), rm_d, rv_d, nbt_d, 0.1)
    F.batch_norm(y, rm, rv, None, None, True, 0.1, 1e-5)
    check(mean, y.mean(dim=(0, 2, 3, 4)), "mean")
    check(var, y.var(dim=(0, 2, 3, 4), unbiased=False), "var")
    check(rm_d, rm, "running_mean")
    check(rv_d, rv, "running_var")
    assert int(nbt_d) == 8
    mean2, var2 = B.bn_stats(dev(y), None, None, None, 0.1)      # statistics only
    check(mean2, mean, "mean (no running)", tol=0)


BN_CASES = [  # c, dims, n, slope, pool, residual, dropout p
    (4, (6, 8, 4), 3, 0.2, 1, False, 0.0), (4, (6, 8, 4), 3, 0.2, 2, False, 0.0),
    (8, (4, 6, 4), 2, 0.2, 1, True, 0.0), (8, (4, 6, 4), 2, 0.0, 1, True, 0.0),
    (64, (8, 12, 8), 2, 0.2, 1, False, 0.35), (64, (8, 12, 8), 1, 0.2, 2, False, 0.0),
    (12, (4, 4, 4), 2, 0.0, 2, False, 0.0), (256, (2, 3, 2), 3, 0.2, 1, True, 0.25),
    # more than 1024 rows of per-block backward sums (folded onto 256 rows before the per-channel kernel)
    (8, (64, 130, 128), 1, 0.2, 1, False, 0.0), (3, (32, 48, 48), 1, 0.2, 1, True, 0.0),
]


@pytest.mark.parametrize("c,dims,n,slope,pool,use_res,p", BN_CASES)
def test_bn_act_fwd_bwd(B, c, dims, n, slope, pool, use_res, p):
    gen = torch.Generator().manual_seed(c + pool + int(slope * 10))
    y = rnd(gen, n, c, *dims) * 2 + 0.5
    gamma, beta = 1 + 0.3 * rnd(gen, c), 0.2 * rnd(gen, c)
    res = rnd(gen, n, c, *dims) if use_res else None
    odims = tuple(d // pool for d in dims)
    keep = (torch.rand(n, c, *odims, generator=gen) >= p).float() if p > 0 else None
    dout = rnd(gen, n, c, *odims)
    rm, rv = torch.zeros(c), torch.ones(c)
    yr, gr, br = y.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rr = res.clone().requires_grad_(True) if use_res else None
    ref = O.bn_act(yr, gr, br, rm.clone(), rv.clone(), slope, pool, rr, keep, p, True)
    grads = torch.autograd.grad(ref, [yr, gr, br] + ([rr] if use_res else []), dout)

    mean, var = B.bn_stats(dev(y), None, None, None, 0.1)
    keep_d = dev(keep.to(torch.uint8)) if keep is not None else None
    out = B.bn_act_fwd(dev(y), mean, var, dev(gamma), dev(beta), dev(res), keep_d, p, slope, pool, 1e-5)
    check(out, ref, "fwd")
    dy, dg, db, dres = B.bn_act_bwd(dev(dout), dev(y), mean, var, dev(gamma), dev(beta), dev(res), keep_d,
                                    p, slope, pool, 1e-5, True, use_res)
    check(dy, grads[0], "dy", max_outliers=3)
    check(dg, grads[1], "dgamma", tol=2e-4)
    check(db, grads[2], "dbeta", tol=2e-4)
    if use_res:
        check(dres, grads[3], "dres", max_outliers=3)
    dy2, dg2, db2, dres2 = B.bn_act_bwd(dev(dout), dev(y), mean, var, dev(gamma), dev(beta), dev(res), keep_d,
                                        p, slope, pool, 1e-5, False, False)
    assert dg2 is None and db2 is None and dres2 is None
    check(dy2, dy, "dy (no affine grads)", tol=0)
    # evaluation mode: running statistics in place of batch statistics
    rm2, rv2 = rnd(gen, c), torch.rand(c, generator=gen) + 0.5
    out_e = B.bn_act_fwd(dev(y), dev(rm2), dev(rv2), dev(gamma), dev(beta), dev(res), None, 0.0, slope, pool, 1e-5)
    check(out_e, O.bn_act(y, gamma, beta, rm2, rv2, slope, pool, res, None, 0.0, False), "eval fwd")


@pytest.mark.parametrize("c,dims,n,groups,pool,use_res,p", [
    (64, (8, 12, 8), 4, 2, 1, False, 0.25), (256, (10, 12, 10), 4, 2, 1, True, 0.0), (128, (20, 24, 20), 4, 2, 2, False, 0.0),
    (8, (4, 6, 4), 6, 3, 2, True, 0.35), (6, (4, 6, 4), 4, 2, 1, True, 0.0), (12, (4, 4, 4), 2, 2, 2, False, 0.0),
    (8, (64, 130, 128), 2, 2, 1, False, 0.0),          # partial rows folded per group
])
def test_bn_grouped_equals_one_call_per_part(B, c, dims, n, groups, pool, use_res, p):
    """The grouped BatchNorm operators (a batch made of `groups` passes of the reference: statistics per part,
    running statistics updated part after part, affine gradients summed over the parts) against the same
    operators called once per part -- bit for bit, same kernels per part -- and the statistics against torch."""
    gen = torch.Generator().manual_seed(c + pool + groups)
    y = rnd(gen, n, c, *dims) * 2 + torch.arange(n).view(n, 1, 1, 1, 1).float()      # parts differ in mean
    gamma, beta = 1 + 0.3 * rnd(gen, c), 0.2 * rnd(gen, c)
    res = rnd(gen, n, c, *dims) if use_res else None
    odims = tuple(d // pool for d in dims)
    keep = (torch.rand(n, c, *odims, generator=gen) >= p).to(torch.uint8) if p > 0 else None
    dout = rnd(gen, n, c, *odims)
    yd, gd, bd, rd, kd, dd = dev(y), dev(gamma), dev(beta), dev(res), dev(keep), dev(dout)
    per = n // groups
    part = lambda t, g: None if t is None else t[g * per:(g + 1) * per]
    # statistics
    rm, rv, nbt = dev(rnd(gen, c)), dev(torch.rand(c, generator=gen) + 0.5), dev(torch.tensor(3))
    rm1, rv1, nbt1 = rm.clone(), rv.clone(), nbt.clone()
    mean, var = B.bn_stats(yd, rm, rv, nbt, 0.1, groups)
    assert mean.shape == (groups, c) and int(nbt) == 3 + groups
    for g in range(groups):
        m1, v1 = B.bn_stats(part(yd, g), rm1, rv1, nbt1, 0.1)
        assert torch.equal(mean[g], m1) and torch.equal(var[g], v1), g
        check(mean[g], part(y, g).double().mean(dim=(0, 2, 3, 4)).float(), "mean")
        check(var[g], part(y, g).double().var(dim=(0, 2, 3, 4), unbiased=False).float(), "var")
    assert torch.equal(rm, rm1) and torch.equal(rv, rv1) and int(nbt1) == int(nbt)
    # forward
    out = B.bn_act_fwd(yd, mean, var, gd, bd, rd, kd, p, 0.2, pool, 1e-5)
    for g in range(groups):
        o1 = B.bn_act_fwd(part(yd, g), mean[g], var[g], gd, bd, part(rd, g), part(kd, g), p, 0.2, pool, 1e-5)
        assert torch.equal(part(out, g), o1), g
    # backward, affine gradients accumulated into buffers that already hold something
    acc_g, acc_b = dev(rnd(gen, c)), dev(rnd(gen, c))
    acc_g0, acc_g1, acc_b1 = acc_g.clone(), acc_g.clone(), acc_b.clone()
    dy, _, _, dres = B.bn_act_bwd(dd, yd, mean, var, gd, bd, rd, kd, p, 0.2, pool, 1e-5, True, use_res, acc_g, acc_b)
    for g in range(groups):
        dy1, _, _, dres1 = B.bn_act_bwd(part(dd, g), part(yd, g), mean[g], var[g], gd, bd, part(rd, g), part(kd, g),
                                        p, 0.2, pool, 1e-5, True, use_res, acc_g1, acc_b1)
        assert torch.equal(part(dy, g), dy1), g
        if use_res:
            assert torch.equal(part(dres, g), dres1), g
    assert torch.equal(acc_g, acc_g1) and torch.equal(acc_b, acc_b1)
    dy2, dg2, db2, _ = B.bn_act_bwd(dd, yd, mean, var, gd, bd, rd, kd, p, 0.2, pool, 1e-5, True, False)
    assert torch.equal(dy2, dy)
    check(dg2, acc_g - acc_g0, "dgamma returned vs accumulated", tol=1e-4)
    import siv._lib as L
    with pytest.raises(L.SivError):                               # batch not a multiple of the groups
        B.bn_stats(torch.zeros(groups + 1, c, 2, 2, 2, device="cuda"), None, None, None, 0.1, groups)


def test_act_dropout(B):
    gen = torch.Generator().manual_seed(11)
    x = rnd(gen, 2, 1, 6, 7, 5)
    keep = (torch.rand(2, 1, 6, 7, 5, generator=gen) >= 0.35).float()
    dout = rnd(gen, 2, 1, 6, 7, 5)
    for slope in (0.0, 0.2):
        xr = x.clone().requires_grad_(True)
        ref = O.act(xr, slope) * keep * (1 / (1 - 0.35))
        gx, = torch.autograd.grad(ref, xr, dout)
        out = B.act_dropout_fwd(dev(x), dev(keep.to(torch.uint8)), 0.35, slope)
        dx = B.act_dropout_bwd(dev(dout), dev(x), dev(keep.to(torch.uint8)), 0.35, slope)
        check(out, ref, "fwd")
        check(dx, gx, "bwd")
        check(B.act_dropout_fwd(dev(x), None, 0.0, slope), O.act(x, slope), "fwd (no dropout)")


def test_reparam_kl_sse(B):
    gen = torch.Generator().manual_seed(12)
    mu, lv = rnd(gen, 4, 1, 2, 3, 2) * 3, rnd(gen, 4, 1, 2, 3, 2) * 2
    lv[0, 0, 0, 0, 0], lv[1, 0, 0, 0, 0] = 30.0, -30.0                 # extreme log-variances
    eps = rnd(gen, 4, 1, 2, 3, 2)
    dz = rnd(gen, 4, 1, 2, 3, 2)
    mr, lr_ = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    zr = mr + eps * torch.exp(0.5 * lr_)
    gm, gl = torch.autograd.grad(zr, [mr, lr_], dz)
    z = B.reparam_fwd(dev(mu), dev(lv), dev(eps), 0.0)
    dmu, dlv = B.reparam_bwd(dev(dz), dev(lv), dev(eps), 0.0)
    check(z, zr, "z")
    check(dmu, gm, "dmu")
    check(dlv, gl, "dlogvar")
    check(B.reparam_fwd(dev(mu), dev(lv), None, 0.1), mu + 0.1 * torch.exp(0.5 * lv), "z (eps=0.1)")
    # KL
    g = rnd(gen, 4)
    mr, lr_ = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    klr = O.calc_kl(lr_, mr, "none")
    gm, gl = torch.autograd.grad(klr, [mr, lr_], g)
    kl = B.kl_fwd(dev(mu), dev(lv))
    dmu, dlv = B.kl_bwd(dev(g), dev(mu), dev(lv))
    check(kl, klr, "kl")
    check(dmu, gm, "kl dmu")
    check(dlv, gl, "kl dlogvar")
    # SSE, including a size that is not a multiple of anything
    for shape in ((4, 1, 6, 5, 7), (3, 1, 16, 24, 16)):
        x, y = torch.rand(*shape, generator=gen), torch.rand(*shape, generator=gen)
        g = rnd(gen, shape[0])
        xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
        ser = O.calc_reconstruction_loss(xr, yr, "none")
        gx, gy = torch.autograd.grad(ser, [xr, yr], g)
        se = B.sse_fwd(dev(x), dev(y))
        dx, dy = B.sse_bwd(dev(g), dev(x), dev(y), True, True)
        check(se, ser, "sse")
        check(dx, gx, "sse dx")
        check(dy, gy, "sse dy")
        dx2, dy2 = B.sse_bwd(dev(g), dev(x), dev(y), False, True)
        assert dx2 is None
        check(dy2, gy, "sse dy only")
    assert float(B.sse_fwd(dev(x), dev(x)).abs().max()) == 0.0
    zero = torch.zeros(2, 1, 2, 3, 2, device="cuda")
    assert float(B.kl_fwd(zero, zero).abs().max()) == 0.0


def test_adam_matches_torch(B):
    gen = torch.Generator().manual_seed(13)
    n = 1000 + 37
    p0 = rnd(gen, n)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=2e-4)
    p, m, v = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for step in range(1, 6):
        g = rnd(gen, n) * (10.0 ** (step - 3))
        g[::7] = 0.0
        ref.grad = g.clone()
        opt.step()
        B.adam_step(p, dev(g), m, v, 2e-4, 0.9, 0.999, 1e-8, step)
        assert float((p.cpu() - ref.detach()).abs().max()) <= 1e-6, step
    zeros = torch.zeros(n, device="cuda")                    # zero gradient => parameter untouched
    p2, m2, v2 = dev(p0.clone()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    B.adam_step(p2, zeros, m2, v2, 2e-4, 0.9, 0.999, 1e-8, 1)
    assert torch.equal(p2.cpu(), p0)


def test_device_rng_statistics(B):
    n = 1 << 20
    a = B.randn((n,), torch.device("cuda"), 77, 5).cpu()
    b = B.randn((n,), torch.device("cuda"), 77, 5).cpu()
    c = B.randn((n,), torch.device("cuda"), 77, 6).cpu()
    assert torch.equal(a, b) and not torch.equal(a, c)           # counter based: reproducible, stream-keyed
    assert abs(float(a.mean())) < 5e-3 and abs(float(a.std()) - 1) < 5e-3
    assert abs(float((a ** 4).mean()) - 3.0) < 0.1               # normal kurtosis
    assert torch.isfinite(a).all()
    odd = B.randn((3, 1, 5, 7, 3), torch.device("cuda"), 1, 0)
    assert odd.shape == (3, 1, 5, 7, 3) and torch.isfinite(odd).all()
    for p in (0.25, 0.35):
        k = B.dropout_keep((2, 8, 16, 16, 16), torch.device("cuda"), p, 77, 9).cpu()
        assert k.dtype == torch.uint8 and set(k.unique().tolist()) <= {0, 1}
        assert abs(float(k.float().mean()) - (1 - p)) < 3e-3


def test_axpby(B):
    gen = torch.Generator().manual_seed(14)
    x, y = rnd(gen, 1001), rnd(gen, 1001)
    yd = dev(y.clone())
    B.axpby(0.5, dev(x), 2.0, yd)
    check(yd, 0.5 * x + 2.0 * y, "axpby")
    B.scale_(yd, 0.25)                                           # in-place scaling
    check(yd, 0.25 * (0.5 * x + 2.0 * y), "scale")
    import siv._lib as L
    with pytest.raises(L.SivError):                              # the out-of-place kernel rejects aliased buffers
        B.axpby(0.25, yd, 0.0, yd)


@pytest.mark.parametrize("b,k,o", [(2, 1200, 32), (8, 19200, 1200), (8, 600, 19200), (3, 37, 5), (11, 130, 64),
                                   (17, 68, 21), (4, 19200, 1200), (1, 600, 19200), (5, 4, 3),
                                   (3, 20, 7), (2, 19204, 16), (2, 76, 33)])  # last K slab shorter than 16
def test_linear_fwd_dgrad_wgrad(B, b, k, o):
    """nn.Linear of the FC-latent model (models/mymodel.py:125,151) incl. the benchmark shapes
    (B=8: 19200 -> 1200 and 600 -> 19200), ragged shapes on the VALU kernels (K not a multiple of 4) and on
    the matrix-pipe kernels (partial column groups, slabs and a batch above one 16-row tile)."""
    gen = torch.Generator().manual_seed(b * 7 + k + o)
    x, w, bias = rnd(gen, b, k), rnd(gen, o, k, scale=k ** -0.5), rnd(gen, o)
    dy = rnd(gen, b, o)
    y = B.linear_fwd(dev(x), dev(w), dev(bias))
    check(y, F.linear(x.double(), w.double(), bias.double()).float(), "linear fwd")
    check(B.linear_fwd(dev(x), dev(w), None), F.linear(x.double(), w.double()).float(), "linear fwd (no bias)")
    check(B.linear_dgrad(dev(dy), dev(w)), (dy.double() @ w.double()).float(), "linear dgrad")
    dw, db = B.linear_wgrad(dev(x), dev(dy), True)
    check(dw, (dy.double().t() @ x.double()).float(), "linear wgrad")
    check(db, dy.double().sum(0).float(), "linear bias grad")
    # adjoint identities tie the three kernels together
    a = float((y.double().cpu() - bias.double()) .mul(dy.double()).sum())
    bb = float((x.double() * B.linear_dgrad(dev(dy), dev(w)).double().cpu()).sum())
    assert abs(a - bb) <= 1e-5 * float(y.double().norm() * dy.double().norm())


def test_linear_rejects_bad_arguments(B):
    import siv._lib as L
    x = torch.zeros(2, 8, device="cuda")
    with pytest.raises(L.SivError):
        L.check(B.lib.siv_linear_fwd(x.data_ptr(), None, None, x.data_ptr(), 2, 8, 4, x.data_ptr(), 16, None), "siv_linear_fwd")
    with pytest.raises(L.SivError):
        L.check(B.lib.siv_linear_fwd(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 0, 8, 4, x.data_ptr(), 16, None), "siv_linear_fwd")
    with pytest.raises(L.SivError):      # workspace too small for the split-row partial sums
        L.check(B.lib.siv_linear_dgrad(x.data_ptr(), x.data_ptr(), x.data_ptr(), 2, 8, 4, x.data_ptr(), 16, None),
                "siv_linear_dgrad")


@pytest.mark.parametrize("shape", [(2, 8, 5, 6, 5), (3, 7), (1, 4, 10, 12, 10)])
def test_add_act(B, shape):
    gen = torch.Generator().manual_seed(len(shape))
    a, b, dout = rnd(gen, *shape), rnd(gen, *shape), rnd(gen, *shape)
    out = B.add_act_fwd(dev(a), dev(b), 0.2)
    ref = F.leaky_relu(a + b, 0.2)
    check(out, ref, "add_act fwd")
    d = B.add_act_bwd(dev(dout), dev(a), dev(b), 0.2)
    s = a + b
    check(d, dout * torch.where(s > 0, torch.ones_like(s), torch.full_like(s, 0.2)), "add_act bwd", max_outliers=2)


@pytest.mark.parametrize("cin,cout,dims,n,ups,path,use_bias", [
    (64, 64, (8, 12, 8), 2, 1, 3, False), (8, 64, (5, 7, 9), 3, 1, 3, True), (64, 128, (8, 8, 16), 1, 2, 3, True),
    (16, 64, (10, 24, 20), 1, 1, 3, False),            # statistics from the Winograd epilogue
    (64, 128, (8, 8, 16), 1, 2, 0, True), (16, 64, (10, 24, 20), 1, 2, 0, False),
    (8, 64, (6, 10, 14), 2, 2, 0, True),               # ... and from the collapsed-tap (upsampled) kernel
    (64, 64, (8, 12, 8), 2, 1, 5, False), (8, 64, (5, 7, 9), 3, 1, 5, True),   # ... and the split-bf16 kernels
    (64, 128, (8, 8, 16), 1, 2, 5, True), (16, 64, (10, 24, 20), 1, 2, 5, False),
    (64, 64, (8, 12, 8), 2, 1, 6, False), (8, 64, (5, 7, 9), 3, 1, 6, True), (16, 128, (10, 24, 20), 1, 1, 6, True),
    # ... from chained bricks (375 bricks on 256 blocks: a block emits the rows of all its bricks)
    (32, 64, (20, 24, 40), 5, 1, 6, True),
    # ... and from the kernel that adds the four partial outputs of a split channel contraction
    (256, 256, (10, 12, 10), 8, 1, 0, True), (128, 128, (10, 12, 10), 8, 1, 6, False),
    (128, 128, (20, 24, 20), 4, 1, 0, True),           # ... two partial outputs
    # ... and the product form of the upsampled kernel (forced by path 6; one-plane bricks, 20-wide variant)
    (64, 128, (8, 8, 16), 1, 2, 6, True), (16, 64, (10, 24, 20), 1, 2, 6, False), (8, 64, (6, 10, 14), 2, 2, 6, True),
    # ... with a partial output-channel tile (16, 32, 80 channels), both forms, and under the split-bf16 path
    (32, 16, (16, 24, 40), 2, 2, 0, True), (16, 32, (8, 12, 20), 1, 2, 6, False), (16, 80, (4, 24, 40), 1, 2, 5, True),
    (4, 8, (6, 7, 5), 2, 1, 0, True), (1, 8, (8, 12, 8), 2, 1, 0, True), (64, 64, (10, 12, 10), 1, 1, 0, False),
    # ... and the single-channel stem on the matrix pipe (ragged bricks, several bricks per block, narrow widths)
    (1, 64, (9, 10, 11), 2, 1, 0, True), (1, 64, (16, 24, 40), 3, 1, 0, True), (1, 16, (7, 9, 12), 2, 1, 0, False),
    (1, 12, (6, 7, 5), 1, 1, 0, True),
])
def test_conv_fwd_with_bn_statistics(B, cin, cout, dims, n, ups, path, use_bias):
    """siv_conv3d_k3_fwd_bnstats == convolution followed by siv_bn_stats (incl. running statistics),
    both where the convolution kernel emits the sums itself and where it falls back."""
    gen = torch.Generator().manual_seed(cin + cout + dims[2] + ups)
    D, H, W = dims
    x = rnd(gen, n, cin, D // ups, H // ups, W // ups) + 0.3
    w = rnd(gen, cout, cin, 3, 3, 3, scale=(2.0 / (27 * cin)) ** 0.5)
    b = rnd(gen, cout) * 3 if use_bias else None          # a bias far from zero: the sums are bias-free
    rm0, rv0 = rnd(gen, cout), torch.rand(cout, generator=gen) + 0.5
    rm, rv, nbt = dev(rm0.clone()), dev(rv0.clone()), torch.tensor(4, device="cuda")
    B.set_conv_path(path)
    try:
        y, mean, var = B.conv3d_k3_fwd_stats(dev(x), dev(w), dev(b), ups, rm, rv, nbt, 0.1)
    finally:
        B.set_conv_path(0)
    yr = O.conv3d_k3(x, w, b, ups)
    check(y, yr, "conv output")
    m = yr.numel() // cout
    mr = yr.double().mean(dim=(0, 2, 3, 4))
    vr = yr.double().var(dim=(0, 2, 3, 4), unbiased=False)
    check(mean, mr.float(), "mean")
    check(var, vr.float(), "biased variance")
    check(rm, (0.9 * rm0.double() + 0.1 * mr).float(), "running mean")
    check(rv, (0.9 * rv0.double() + 0.1 * vr * m / (m - 1)).float(), "running var")
    assert int(nbt) == 5


@pytest.mark.parametrize("C,dims,n,p", [(64, (9, 10, 11), 2, 0.2), (16, (16, 24, 20), 1, 0.0), (64, (8, 12, 8), 3, 0.35),
                                        (12, (6, 7, 5), 2, 0.5)])
def test_conv1_bnact_matches_separate_ops(B, C, dims, n, p):
    """The encoder stem Conv3d(1, C) -> BatchNorm -> LeakyReLU -> Dropout (models/models.py:91-96) as one
    operator that never stores the convolution output: forward, batch / running statistics and the
    backward down to the convolution output against float64 autograd of the composition; the autograd
    operator (incl. the stem's weight, bias and input gradients) against the unfused operators."""
    import siv.ops as ops
    gen = torch.Generator().manual_seed(C + dims[0] + n)
    D, H, W = dims
    slope, eps = 0.2, 1e-5
    x = torch.rand(n, 1, D, H, W, generator=gen)
    w, b = rnd(gen, C, 1, 3, 3, 3, scale=0.4), rnd(gen, C) * 2
    gamma, beta = 1 + 0.3 * rnd(gen, C), 0.2 * rnd(gen, C)
    keep = (torch.rand(n, C, D, H, W, generator=gen) >= p).to(torch.uint8) if p > 0 else None
    dout = rnd(gen, n, C, D, H, W)
    rm0, rv0 = rnd(gen, C), torch.rand(C, generator=gen) + 0.5
    # float64 reference
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    gd, btd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yd = F.conv3d(x.double(), wd, bd, padding=1)
    yd.retain_grad()
    mu = yd.mean(dim=(0, 2, 3, 4), keepdim=True)
    vr = yd.var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
    act = F.leaky_relu((yd - mu) / torch.sqrt(vr + eps) * gd.view(1, -1, 1, 1, 1) + btd.view(1, -1, 1, 1, 1), slope)
    ref = act * keep.double() / (1 - p) if keep is not None else act
    ref.backward(dout.double())
    rm, rv, nbt = dev(rm0.clone()), dev(rv0.clone()), torch.tensor(7, device="cuda")
    assert B.conv1_bnact_supported(dev(x), dev(w))
    out, mean, var = B.conv1_bnact_fwd(dev(x), dev(w), dev(b), dev(gamma), dev(beta), dev(keep), p, slope, eps, True,
                                       rm, rv, nbt, 0.1)
    check(out, ref.detach().float(), "fused forward")
    check(mean, mu.flatten().float(), "batch mean")
    check(var, vr.flatten().float(), "batch variance")
    m = yd.numel() // C
    check(rm, (0.9 * rm0.double() + 0.1 * mu.flatten()).float(), "running mean")
    check(rv, (0.9 * rv0.double() + 0.1 * vr.flatten() * m / (m - 1)).float(), "running variance")
    assert int(nbt) == 8
    dy, dg, db = B.conv1_bnact_bwd(dev(dout), dev(x), dev(w), dev(b), mean, var, dev(gamma), dev(beta), dev(keep), p,
                                   slope, eps, True)
    check(dy, yd.grad.float(), "gradient of the convolution output", max_outliers=2)
    check(dg, gd.grad.float(), "dgamma")
    check(db, btd.grad.float(), "dbeta")
    # evaluation mode: running statistics, nothing updated
    oe, _, _ = B.conv1_bnact_fwd(dev(x), dev(w), dev(b), dev(gamma), dev(beta), None, 0.0, slope, eps, False,
                                 dev(rm0), dev(rv0), None, 0.1)
    ye = F.leaky_relu((yd.detach() - rm0.double().view(1, -1, 1, 1, 1)) / torch.sqrt(rv0.double().view(1, -1, 1, 1, 1) + eps)
                      * gamma.double().view(1, -1, 1, 1, 1) + beta.double().view(1, -1, 1, 1, 1), slope)
    check(oe, ye.float(), "evaluation-mode forward")

    # the autograd operator against the unfused operators
    outs = []
    for fused in (True, False):
        bn = torch.nn.BatchNorm3d(C).to("cuda").train()
        conv = torch.nn.Conv3d(1, C, 3, 1, 1).to("cuda")
        with torch.no_grad():
            bn.weight.copy_(dev(gamma)); bn.bias.copy_(dev(beta))
            conv.weight.copy_(dev(w)); conv.bias.copy_(dev(b))
        xg = dev(x).requires_grad_(True)
        fn = ops.conv1_bn_act if fused else ops.conv_bn_act
        o = fn(xg, conv, bn, slope, keep=dev(keep), p=p, train=True)
        o.backward(dev(dout))
        outs.append((o.detach(), xg.grad, conv.weight.grad, conv.bias.grad, bn.weight.grad, bn.bias.grad,
                     bn.running_mean.clone(), bn.running_var.clone()))
    for a_, b_, what in zip(outs[0], outs[1], ("out", "dx", "dw", "dbias", "dgamma", "dbeta", "running mean",
                                               "running var")):
        if what == "dbias":          # BatchNorm cancels the convolution bias: both are rounding noise around zero
            assert float(a_.abs().max()) <= 1e-3 * float(outs[0][2].abs().max()) + 1e-6
            continue
        check(a_, b_, "operator: " + what, tol=5e-5, max_outliers=2)
    check(outs[0][2], wd.grad.float(), "dw against float64")


@pytest.mark.parametrize("C,dims,n,p,epoch", [(64, (9, 10, 11), 2, 0.25, None), (64, (16, 12, 8), 1, 0.25, 3),
                                              (12, (6, 7, 5), 2, 0.5, 0), (16, (5, 9, 6), 3, 0.1, 70000)])
def test_conv1_bnact_draws_the_mask_it_would_have_read(B, C, dims, n, p, epoch):
    """Dropout of the fused stem with the mask drawn inside the kernels from (seed, stream id, epoch
    counter) instead of read from a tensor: forward output and all backward results are bit-identical to
    the same calls fed the mask ``dropout_keep`` materialises for that key, with and without the device
    epoch counter, through the raw operator and through autograd."""
    import siv.ops as ops
    from siv.rng import PhiloxKeep
    gen = torch.Generator().manual_seed(C * 3 + n)
    D, H, W = dims
    slope, eps = 0.2, 1e-5
    x = dev(torch.rand(n, 1, D, H, W, generator=gen))
    w, b = dev(rnd(gen, C, 1, 3, 3, 3, scale=0.4)), dev(rnd(gen, C))
    gamma, beta = dev(1 + 0.3 * rnd(gen, C)), dev(0.2 * rnd(gen, C))
    dout = dev(rnd(gen, n, C, D, H, W))
    ep = None if epoch is None else torch.tensor([epoch], dtype=torch.int64, device="cuda")
    key = PhiloxKeep(seed=0x9E3779B97F4A7C15, stream_id=(2 << 40) | 17, epoch=ep)
    mask = key.materialise((n, C, D, H, W), p, "cuda")
    frac = float(mask.float().mean())
    assert abs(frac - (1 - p)) < 0.03, frac
    res = []
    for keep in (mask, key):
        rm, rv, nbt = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.tensor(0, device="cuda")
        out, mean, var = B.conv1_bnact_fwd(x, w, b, gamma, beta, keep, p, slope, eps, True, rm, rv, nbt, 0.1)
        dy, dg, db = B.conv1_bnact_bwd(dout, x, w, b, mean, var, gamma, beta, keep, p, slope, eps, True)
        res.append((out, dy, dg, db))
    for a_, b_, what in zip(res[0], res[1], ("out", "dy", "dgamma", "dbeta")):
        assert torch.equal(a_, b_), what
    assert float((res[0][0] == 0).float().mean()) > 0.5 * p        # the mask did drop elements
    if epoch is not None:           # another epoch, another mask
        ep.add_(1)
        assert not torch.equal(key.materialise((n, C, D, H, W), p, "cuda"), mask)
        ep.sub_(1)
    outs = []
    for keep in (mask, key):
        bn = torch.nn.BatchNorm3d(C).to("cuda").train()
        conv = torch.nn.Conv3d(1, C, 3, 1, 1).to("cuda")
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta); conv.weight.copy_(w); conv.bias.copy_(b)
        xg = x.clone().requires_grad_(True)
        o = ops.conv1_bn_act(xg, conv, bn, slope, keep=keep, p=p, train=True)
        o.backward(dout)
        torch.cuda.synchronize()
        outs.append((o.detach(), xg.grad, conv.weight.grad, bn.weight.grad, bn.bias.grad))
    for a_, b_, what in zip(outs[0], outs[1], ("out", "dx", "dw", "dgamma", "dbeta")):
        assert torch.equal(a_, b_), "operator: " + what


@pytest.mark.parametrize("C,dims,n,p,draw,smooth", [
    (64, (9, 10, 11), 2, 0.2, False, False), (16, (16, 24, 20), 1, 0.0, False, False),
    (64, (8, 12, 8), 3, 0.35, True, False), (12, (6, 7, 5), 2, 0.5, False, False),
    (64, (20, 24, 36), 2, 0.35, True, False),
    # smooth volumes under zero-sum (edge detector) filters: the case in which the filter's response to the
    # input's tap correlation is a small difference of large terms
    (64, (40, 48, 40), 2, 0.35, True, True), (64, (24, 24, 40), 1, 0.0, False, True)])
def test_conv1_bnact_bits_and_weight_only_backward(B, C, dims, n, p, draw, smooth):
    """The stem's forward pass with its dropout / activation decisions left as two bits per element
    (siv_conv1_bnact_fwd_bits), the backward that takes the mask from them (siv_conv1_bnact_bwd_bits: bit-identical
    to the mask-tensor form) and the weight-only backward that never forms the gradient of the convolution
    output (siv_conv1_bnact_wgrad_bits): dw, dgamma, dbeta against float64 autograd of the composition
    (models/models.py:91-96) and against the two-step path (bwd, then the convolution's weight gradient);
    the autograd operator with an input that needs no gradient, with plain .grad tensors and with the
    optimiser's flat buffers, against the unfused operators."""
    import siv.ops as ops
    from siv.optim import FlatAdam
    from siv.rng import PhiloxKeep
    gen = torch.Generator().manual_seed(2 * C + dims[0] + n)
    D, H, W = dims
    slope, eps = 0.2, 1e-5
    x = torch.rand(n, 1, D, H, W, generator=gen)
    w, b = rnd(gen, C, 1, 3, 3, 3, scale=0.4), rnd(gen, C) * 2
    if smooth:
        zz, yy, xx = torch.meshgrid(torch.arange(D), torch.arange(H), torch.arange(W), indexing="ij")
        base = 0.5 + 0.25 * torch.sin(zz / 7.0) * torch.cos(yy / 9.0) + 0.2 * torch.sin(xx / 11.0 + 0.3)
        x = (base.float()[None, None] * (0.8 + 0.2 * torch.rand(n, 1, 1, 1, 1, generator=gen)) + 0.01 * x).clamp(0, 1)
        w = w - w.mean(dim=(1, 2, 3, 4), keepdim=True)
    gamma, beta = 1 + 0.3 * rnd(gen, C), 0.2 * rnd(gen, C)
    dout = rnd(gen, n, C, D, H, W)
    if p > 0 and draw:
        key = PhiloxKeep(seed=0x51ED270B, stream_id=(3 << 40) | 5, epoch=None)
        keep = key.materialise((n, C, D, H, W), p, "cuda").cpu()
        keep_arg = key
    else:
        keep = (torch.rand(n, C, D, H, W, generator=gen) >= p).to(torch.uint8) if p > 0 else None
        keep_arg = dev(keep)
    # float64 reference
    wd, bd = w.double().requires_grad_(True), b.double().requires_grad_(True)
    gd, btd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yd = F.conv3d(x.double(), wd, bd, padding=1)
    mu = yd.mean(dim=(0, 2, 3, 4), keepdim=True)
    vr = yd.var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
    zd = (yd - mu) / torch.sqrt(vr + eps) * gd.view(1, -1, 1, 1, 1) + btd.view(1, -1, 1, 1, 1)
    act = F.leaky_relu(zd, slope)
    ref = act * keep.double() / (1 - p) if keep is not None else act
    ref.backward(dout.double())

    def stats():
        return torch.zeros(C, device="cuda"), torch.ones(C, device="cuda"), torch.tensor(0, device="cuda")
    xs, ws, bs, gs, bts, ds = dev(x), dev(w), dev(b), dev(gamma), dev(beta), dev(dout)
    out0, mean0, var0 = B.conv1_bnact_fwd(xs, ws, bs, gs, bts, keep_arg, p, slope, eps, True, *stats(), 0.1)
    out, mean, var, bits = B.conv1_bnact_fwd(xs, ws, bs, gs, bts, keep_arg, p, slope, eps, True, *stats(), 0.1,
                                             emit_bits=True)
    assert torch.equal(out, out0) and torch.equal(mean, mean0) and torch.equal(var, var0)
    # layout: one byte per channel quad of a voxel, bit j = kept, bit 4 + j = BatchNorm output > 0
    assert bits.dtype == torch.uint8 and bits.numel() == n * D * H * W * C // 4
    bq = bits.view(n, D, H, W, C // 4).cpu().long()
    kept = torch.stack([(bq >> j) & 1 for j in range(4)], dim=-1).reshape(n, D, H, W, C).permute(0, 4, 1, 2, 3)
    posb = torch.stack([(bq >> (4 + j)) & 1 for j in range(4)], dim=-1).reshape(n, D, H, W, C).permute(0, 4, 1, 2, 3)
    assert torch.equal(kept, keep.long() if keep is not None else torch.ones_like(kept))
    near = zd.detach().abs() < 1e-4 * float(zd.detach().abs().max())        # sign undecided within rounding
    assert bool(((posb == 1) == (zd.detach() > 0))[~near].all())
    # the mask from the bits: same results as from the mask tensor / the generator, bit for bit
    dy0, dg0, db0 = B.conv1_bnact_bwd(ds, xs, ws, bs, mean, var, gs, bts, keep_arg, p, slope, eps, True)
    dy1, dg1, db1 = B.conv1_bnact_bwd(ds, xs, ws, bs, mean, var, gs, bts, None, p, slope, eps, True, bits=bits)
    assert torch.equal(dy0, dy1) and torch.equal(dg0, dg1) and torch.equal(db0, db1)
    # weight-only backward
    dw, dcb, dg, db = B.conv1_bnact_wgrad(ds, xs, ws, bs, mean, var, gs, bits, p, slope, eps)
    # (smooth volumes put many BatchNorm outputs within rounding of zero: an element whose sign differs between
    # fp32 and float64 moves a sum by its whole gradient, so float64 is the looser of the two comparisons there;
    # the two-step path shares the decisions and must agree closely)
    tol64, out64 = (1e-3, 2) if smooth else (TOL, 0)
    check(dw, wd.grad.float(), "dw against float64", tol=tol64, max_outliers=out64)
    check(dg, gd.grad.float(), "dgamma against float64", tol=tol64, max_outliers=out64)
    check(db, btd.grad.float(), "dbeta against float64", tol=tol64, max_outliers=out64)
    dw2, _ = B.conv3d_k3_wgrad(xs, dy0, 1, True)
    check(dw, dw2, "dw against the two-step path", tol=2e-5)
    check(dg, dg0, "dgamma against the two-step path", tol=2e-5)
    check(db, db0, "dbeta against the two-step path", tol=1e-6)
    assert float(dcb.abs().max()) <= 1e-3 * float(dw.abs().max()) + 1e-6      # BatchNorm cancels the convolution bias
    # accumulating form
    acc = [torch.full_like(t, 0.5) for t in (dw, dcb, dg, db)]
    B.conv1_bnact_wgrad(ds, xs, ws, bs, mean, var, gs, bits, p, slope, eps, *acc)
    for a_, r_, what in zip(acc, (dw, dcb, dg, db), ("dw", "dbias", "dgamma", "dbeta")):
        check(a_, 0.5 + r_, "accumulated " + what, tol=1e-6)

    # the autograd operator on an input that needs no gradient
    outs = []
    for mode in ("fused", "fused-flat", "unfused"):
        bn = torch.nn.BatchNorm3d(C).to("cuda").train()
        conv = torch.nn.Conv3d(1, C, 3, 1, 1).to("cuda")
        with torch.no_grad():
            bn.weight.copy_(gs); bn.bias.copy_(bts); conv.weight.copy_(ws); conv.bias.copy_(bs)
        if mode == "fused-flat":
            opt = FlatAdam(list(conv.parameters()) + list(bn.parameters()), lr=1e-3)
        fn = ops.conv_bn_act if mode == "unfused" else ops.conv1_bn_act
        o = fn(xs, conv, bn, slope, keep=keep_arg if mode != "unfused" else dev(keep), p=p, train=True)
        o.backward(ds)
        torch.cuda.synchronize()
        outs.append((o.detach(), conv.weight.grad.clone(), conv.bias.grad.clone(), bn.weight.grad.clone(),
                     bn.bias.grad.clone()))
    for k, what in enumerate(("out", "dw", "dbias", "dgamma", "dbeta")):
        assert torch.equal(outs[0][k], outs[1][k]), "flat buffers: " + what
        if what == "dbias":
            assert float(outs[0][k].abs().max()) <= 1e-3 * float(outs[0][1].abs().max()) + 1e-6
            continue
        check(outs[0][k], outs[2][k], "operator: " + what, tol=5e-5, max_outliers=2)


@pytest.mark.parametrize("dims,n", [((16, 24, 32), 2), ((19, 44, 36), 1), ((16, 12, 16), 3)])
def test_bnact_conv_to1_matches_separate_ops(B, dims, n):
    """BatchNorm + LeakyReLU folded into the 64 -> 1 output convolution (models/models.py:76-78,137):
    forward and weight gradient against the oracle's conv(leaky_relu(batch_norm(x))) in float64, the
    operator through autograd against the unfused operators (bn_act + conv3d_k3) on the same inputs, the
    shapes without a kernel reported as unsupported."""
    import siv.ops as ops
    gen = torch.Generator().manual_seed(dims[1] + n)
    D, H, W = dims
    C, slope, eps = 64, 0.2, 1e-5
    x = rnd(gen, n, C, D, H, W) * 1.5 + 0.2
    gamma, beta = 1 + 0.3 * rnd(gen, C), 0.2 * rnd(gen, C)
    w, b = rnd(gen, 1, C, 3, 3, 3, scale=(2.0 / (27 * C)) ** 0.5), rnd(gen, 1)
    dy = rnd(gen, n, 1, D, H, W)
    mean = x.double().mean(dim=(0, 2, 3, 4))
    var = x.double().var(dim=(0, 2, 3, 4), unbiased=False)
    assert B.bnact_conv_to1_supported(dev(x))
    assert not B.bnact_conv_to1_supported(dev(x[:, :32])) and not B.bnact_conv_to1_supported(dev(x[:, :, :8]))
    # float64 reference of the composition
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    mg = xd.mean(dim=(0, 2, 3, 4), keepdim=True)             # batch statistics inside the graph
    vg = xd.var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
    a = F.leaky_relu((xd - mg) / torch.sqrt(vg + eps) * gamma.double().view(1, -1, 1, 1, 1)
                     + beta.double().view(1, -1, 1, 1, 1), slope)
    yr = F.conv3d(a, wd, b.double(), padding=1)
    yr.backward(dy.double())
    args = (dev(x), dev(mean.float()), dev(var.float()), dev(gamma), dev(beta), eps, slope)
    y = B.bnact_conv_to1_fwd(*args, dev(w), dev(b))
    check(y, yr.detach().float(), "fused forward")
    dw, db = B.bnact_conv_to1_wgrad(*args, dev(dy), True)
    check(dw, wd.grad.float(), "fused weight gradient")
    check(db, dy.double().sum().reshape(1).float(), "bias gradient")
    acc = dev(torch.ones_like(w)).contiguous(memory_format=torch.channels_last_3d)
    B.bnact_conv_to1_wgrad(*args, dev(dy), False, acc, None)
    check(acc, 1 + wd.grad.float(), "fused weight gradient, accumulated")
    # input gradient of the convolution + BatchNorm backward in one call: dx, dgamma, dbeta
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    x2 = x.double().requires_grad_(True)
    m2, v2 = x2.mean(dim=(0, 2, 3, 4), keepdim=True), x2.var(dim=(0, 2, 3, 4), unbiased=False, keepdim=True)
    F.conv3d(F.leaky_relu((x2 - m2) / torch.sqrt(v2 + eps) * gd.view(1, -1, 1, 1, 1) + bd.view(1, -1, 1, 1, 1), slope),
             w.double(), None, padding=1).backward(dy.double())
    dx, dgm, dbt = B.bnact_conv_to1_dgrad(dev(dy), dev(w), *args, True)
    check(dx, x2.grad.float(), "fused input gradient")
    check(dgm, gd.grad.float(), "fused dgamma")
    check(dbt, bd.grad.float(), "fused dbeta")
    ag, ab = torch.ones(C, device="cuda"), torch.full((C,), 2.0, device="cuda")
    B.bnact_conv_to1_dgrad(dev(dy), dev(w), *args, True, ag, ab)
    check(ag, 1 + gd.grad.float(), "dgamma, accumulated")
    check(ab, 2 + bd.grad.float(), "dbeta, accumulated")

    # the autograd operator against the unfused operators: same statistics, same parameters
    class _Bn(torch.nn.BatchNorm3d):
        pass
    outs = []
    for fused in (True, False):
        bn = _Bn(C).to("cuda").train()
        conv = torch.nn.Conv3d(C, 1, 3, 1, 1).to("cuda")
        with torch.no_grad():
            bn.weight.copy_(dev(gamma)); bn.bias.copy_(dev(beta))
            conv.weight.copy_(dev(w)); conv.bias.copy_(dev(b))
        conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last_3d)
        xg = dev(x).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
        if fused:
            out = ops.bn_act_conv_to1(xg, bn, slope, conv, True, None)
        else:
            out = ops.conv3d_k3(ops.bn_act(xg, bn, slope, train=True), conv.weight, conv.bias)
        out.backward(dev(dy))
        outs.append((out.detach(), xg.grad, bn.weight.grad, bn.bias.grad, conv.weight.grad, conv.bias.grad,
                     bn.running_mean.clone(), bn.running_var.clone()))
    for a_, b_, what in zip(outs[0], outs[1], ("out", "dx", "dgamma", "dbeta", "dw", "dbias", "running mean",
                                               "running var")):
        check(a_, b_, "operator: " + what, tol=2e-5)
    check(outs[0][1], xd.grad.float(), "dx against float64")
    import siv._lib as L
    with pytest.raises(L.SivError):
        B.bnact_conv_to1_fwd(dev(x[:, :, :8].contiguous()), *args[1:], dev(w), dev(b))


# nn.ConvTranspose3d(k=3, s=1, p=1): the shapes of models/model.py:202-205 (Decoder_lucky: 8 -> 16 -> 32 -> 64 -> 1
# on 32 x 24 x 32 volumes ... scaled down in extent for the CPU reference), an odd shape for the generic path and
# one large enough for the Winograd kernels
@pytest.mark.parametrize("cin,cout,dims,n,path", [
    (8, 16, (8, 12, 8), 2, 0), (16, 32, (8, 12, 8), 2, 0), (32, 64, (8, 12, 8), 1, 0), (64, 1, (9, 10, 11), 2, 0),
    (3, 5, (5, 4, 3), 2, 0), (64, 64, (16, 16, 16), 2, 6), (64, 128, (8, 8, 16), 1, 6), (1, 8, (6, 7, 5), 2, 0),
])
def test_conv_transpose3d_k3(B, cin, cout, dims, n, path):
    gen = torch.Generator().manual_seed(cin * 100 + cout + path)
    x = rnd(gen, n, cin, *dims)
    w = rnd(gen, cin, cout, 3, 3, 3, scale=(2.0 / (27 * cin)) ** 0.5)     # nn.ConvTranspose3d's own layout
    b = rnd(gen, cout)
    dy = rnd(gen, n, cout, *dims)
    B.set_conv_path(path)
    try:
        y = B.conv_transpose3d_k3_fwd(dev(x), dev(w), dev(b))
        y_nb = B.conv_transpose3d_k3_fwd(dev(x), dev(w), None)
        dx = B.conv_transpose3d_k3_dgrad(dev(dy), dev(w))
        dw, db = B.conv_transpose3d_k3_wgrad(dev(x), dev(dy), True)
    finally:
        B.set_conv_path(0)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv_transpose3d(xr, wr, br, stride=1, padding=1)
    gx, gw, gb = torch.autograd.grad(yr, [xr, wr, br], dy)
    check(y, yr, "fwd")
    check(y_nb, F.conv_transpose3d(x, w, None, stride=1, padding=1), "fwd (no bias)")
    check(dx, gx, "dgrad")
    check(dw, gw, "wgrad")
    check(db, gb, "bias grad")


def test_conv_transpose3d_autograd(B):
    """ops.conv_transpose3d_k3 as a differentiable operator, two layers deep with a sigmoid on top as in
    Decoder_lucky (models/model.py:202-216)."""
    import siv.ops as ops
    gen = torch.Generator().manual_seed(21)
    x = rnd(gen, 2, 8, 6, 8, 6)
    w1, b1 = rnd(gen, 8, 16, 3, 3, 3, scale=0.1), rnd(gen, 16)
    w2, b2 = rnd(gen, 16, 1, 3, 3, 3, scale=0.1), rnd(gen, 1)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    yr = torch.sigmoid(F.conv_transpose3d(F.conv_transpose3d(ref_in[0], ref_in[1], ref_in[2], 1, 1), ref_in[3], ref_in[4], 1, 1))
    gr = torch.autograd.grad(yr.square().sum(), ref_in)
    mine = [dev(t).requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y = torch.sigmoid(ops.conv_transpose3d_k3(ops.conv_transpose3d_k3(mine[0], mine[1], mine[2]), mine[3], mine[4]))
    gm = torch.autograd.grad(y.square().sum(), mine)
    check(y, yr, "output")
    for a, b_, nme in zip(gm, gr, ("dx", "dw1", "db1", "dw2", "db2")):
        check(a, b_, nme)


@pytest.mark.parametrize("nb", [8, 3, 70])
def test_sintro_objectives(B, nb):
    """siv_sintro_loss_e / _d (the scalar algebra of utils/my_trainer.py:260-284 and :301-321) and their
    gradients against the reference's torch expressions, incl. exp-ELBO terms that underflow."""
    import siv.ops as ops
    gen = torch.Generator().manual_seed(nb)
    s, br, bn, bk, gr = 8.0 / 614400, 1.0, 1024.0, 0.75, 1e-8
    sse = [torch.rand(nb, generator=gen) * 3e5 for _ in range(5)]
    kl = [torch.rand(nb, generator=gen) * 40 for _ in range(3)]          # small KLs: exp(-2 s bn kl) ~ e^-1
    kl[2][0] = 5e4                                                       # ... and one that underflows to 0
    ve = [sse[0], kl[0], sse[1], kl[1], sse[2], kl[2]]
    ref_in = [v.clone().requires_grad_(True) for v in ve]
    a, k, sf, kf, sr, kr = ref_in
    ef = (-2 * s * (br * sf + bn * kf)).exp().mean()
    er = (-2 * s * (br * sr + bn * kr)).exp().mean()
    lossE = 10 * (s * (br * a.mean() + bk * k.mean()) + 0.5 * (ef + er))
    gref = torch.autograd.grad(lossE, ref_in)
    mine_in = [dev(v).requires_grad_(True) for v in ve]
    loss, st = ops.sintro_loss_e(*mine_in, s, br, bn, bk)
    gm = torch.autograd.grad(loss, mine_in)
    np.testing.assert_allclose(float(loss), float(lossE), rtol=1e-5)
    np.testing.assert_allclose(st.cpu().numpy(), [float(k.mean()), float(er), float(ef)], rtol=1e-5)
    for x, y, n in zip(gm, gref, ("sse_real", "kl_real", "sse_fake", "kl_fake", "sse_rec", "kl_rec")):
        np.testing.assert_allclose(x.cpu().numpy(), y.numpy(), rtol=1e-5, atol=1e-30, err_msg=n)
    vd = [sse[0], kl[0], kl[1], sse[3], sse[4]]
    ref_in = [v.clone().requires_grad_(True) for v in vd]
    a, kr, kf, rr, fr = ref_in
    lossD = 10 * (s * (br * a.mean() + 0.5 * bk * (kr.mean() + kf.mean()) + gr * 0.5 * br * (rr.mean() + fr.mean())))
    gref = torch.autograd.grad(lossD, ref_in)
    mine_in = [dev(v).requires_grad_(True) for v in vd]
    loss, st = ops.sintro_loss_d(*mine_in, s, br, bk, gr)
    gm = torch.autograd.grad(loss, mine_in)
    np.testing.assert_allclose(float(loss), float(lossD), rtol=1e-5)
    np.testing.assert_allclose(st.cpu().numpy(), [float(kf.mean()), float(kr.mean()), float(a.mean())], rtol=1e-5)
    for x, y in zip(gm, gref):
        np.testing.assert_allclose(x.cpu().numpy(), y.numpy(), rtol=1e-5, atol=1e-30)


def test_packed_weight_cache(B):
    """Weights registered by FlatAdam keep their packed form between calls: the second convolution with the
    same weights is served from the cache; an optimiser step, and a torch-level in-place write, drop it."""
    from siv.optim import FlatAdam
    gen = torch.Generator().manual_seed(31)
    conv = torch.nn.Conv3d(64, 64, 3, padding=1, bias=False)
    with torch.no_grad():
        conv.weight.copy_(rnd(gen, 64, 64, 3, 3, 3, scale=0.05))
    conv = conv.cuda()
    opt = FlatAdam(conv.parameters(), lr=0.1)
    x, dy = rnd(gen, 2, 64, 8, 8, 8), rnd(gen, 2, 64, 8, 8, 8)
    hits = lambda: B.lib.siv_pack_cache_hits()

    def run():
        return B.conv3d_k3_fwd(dev(x), conv.weight, None, 1), B.conv3d_k3_dgrad(dev(dy), conv.weight, 1)

    for path in (0, 6):                       # direct matrix-core kernel, two-dimensional Winograd kernel
        B.set_conv_path(path)
        try:
            B.weights_changed()
            h0 = hits()
            y1, d1 = run()
            assert hits() == h0                                   # first use packs
            y2, d2 = run()
            assert hits() == h0 + 2 and torch.equal(y1, y2) and torch.equal(d1, d2)
            w_cpu = conv.weight.detach().cpu()
            check(y2, O.conv3d_k3(x, w_cpu, None, 1), "fwd from cache")
            # an optimiser step changes the weights behind torch's back and must drop the packings
            opt.zero_grad()
            conv.weight.grad.copy_(torch.sign(rnd(gen, 64, 64, 3, 3, 3)).cuda())
            opt.step()
            h1 = hits()
            y3, d3 = run()
            assert hits() == h1
            w_cpu = conv.weight.detach().cpu()
            assert float((w_cpu - y1.new_zeros(1).cpu()).abs().max()) > 0
            check(y3, O.conv3d_k3(x, w_cpu, None, 1), "fwd after step")
            assert not torch.equal(y3, y2)
            # a torch in-place write is seen through the tensor's version counter
            with torch.no_grad():
                conv.weight.mul_(2.0)
            y4, _ = run()
            check(y4, O.conv3d_k3(x, 2 * w_cpu, None, 1), "fwd after in-place write")
            # weights outside any registered buffer are never cached
            h2 = hits()
            w_loose = dev(rnd(gen, 64, 64, 3, 3, 3, scale=0.05))
            B.conv3d_k3_fwd(dev(x), w_loose, None, 1)
            B.conv3d_k3_fwd(dev(x), w_loose, None, 1)
            assert hits() == h2
        finally:
            B.set_conv_path(0)


@pytest.mark.parametrize("path,ups", [(6, 1), (0, 2), (3, 1)])
def test_fused_bn_statistics_with_large_channel_means(B, path, ups):
    """The kernels that emit the BatchNorm sums from their epilogue accumulate UNSHIFTED fp32 sums per brick
    (combined in fp64); siv_bn_stats shifts by a sample first.  With every channel mean several standard
    deviations away from zero (a nearly constant input through all-positive weights; the zero padding at
    the volume border is what keeps the ratio near 5) the two still agree at 1e-4 of the largest variance."""
    gen = torch.Generator().manual_seed(path + ups)
    D, H, W = 16, 16, 16
    x = rnd(gen, 2, 64, D // ups, H // ups, W // ups) * 0.05 + 1.0        # nearly constant input: output mean >> std
    w = rnd(gen, 64, 64, 3, 3, 3, scale=0.02) + 0.01
    B.set_conv_path(path)
    try:
        y, mean, var = B.conv3d_k3_fwd_stats(dev(x), dev(w), None, ups, None, None, None, 0.1)
    finally:
        B.set_conv_path(0)
    m2, v2 = B.bn_stats(y, None, None, None, 0.1)
    yd = y.double()
    ratio = float((yd.mean(dim=(0, 2, 3, 4)).abs() / yd.std(dim=(0, 2, 3, 4))).max())
    assert ratio > 4, ratio
    check(mean, m2, "mean")
    check(var, yd.var(dim=(0, 2, 3, 4), unbiased=False).float(), "variance (fused) vs float64")
    check(var, v2, "variance (fused) vs siv_bn_stats")
