"""bf16 storage of the tensors kept for the backward pass (siv.activations, siv_f32_to_bf16 /
siv_bf16_to_f32) on a real MI355X.

The reference trains in fp32 only, so this mode has no reference counterpart: it is checked against
(a) torch's own fp32 -> bf16 rounding, bit for bit, (b) the fp32 mode of the same model on the same
device (the forward pass must be bit-identical) and (c) the fp32 CPU oracle.  Tolerance of the
gradients, stated: 6e-2 of each tensor's maximum against the fp32 oracle.  Every saved activation
carries a relative rounding error of up to 2^-9 = 2e-3, a pre-activation that close to zero can change
sides in the LeakyReLU mask the backward pass recomputes, and the E-step gradient is a difference of
three encoder passes weighted 1 : 1024-scaled exponentials; measured on this model: decoder gradients
1e-4..3e-2 of max (growing with the distance from the output), encoder gradients 2e-2..4.6e-2, against
1e-5..2e-2 in the fp32 mode.  That is the price of the mode, not an implementation error: the forward
pass is bit-identical and the conversions are exact to the last bit.

By default the HBM-bound kernels write and read the bf16 copies themselves (BatchNorm forward / backward,
the fused stem, the fused output layer); their entry points are checked here against their fp32 twins:
fp32 outputs bit-identical, bf16 copies equal to the rounded fp32 tensors bit for bit, backward passes
from bf16 inputs within bf16 rounding of the fp32 backward.  SIV_BF16_NATIVE=0 selects conversion kernels
around the fp32 entry points instead; the model-level tests run both."""
import numpy as np
import pytest
import torch

from oracle import siv_oracle as O

pytestmark = pytest.mark.gpu
WIDE = (64, [[64, 1, 2], [128, 1, 2]])
BN_CANCELLED = ("encoder.blocks.0.0.bias", "decoder.blocks.0.0.bias", "decoder.blocks.0.0.weight")


@pytest.fixture(autouse=True)
def hip_backend():
    import siv.activations as act
    import siv.backend as sb
    import siv.rng as srng
    sb.set_backend(None)
    assert sb.get_backend().name == "hip"
    yield
    act.set_storage("fp32")
    srng.set_rng(srng.DeviceRng(0))


@pytest.mark.parametrize("n", [8, 1 << 16, (1 << 20) + 8, (1 << 18) + 5, 7 * 4096 + 3])
def test_conversion_matches_torch_rounding(n):
    import siv.backend as sb
    B = sb.get_backend()
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=gen) * torch.exp(8 * torch.randn(n, generator=gen))
    x[:8] = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1.00390625, 1.01171875, 3.3895314e38])
    xd = x.cuda()
    h = B.to_bf16(xd)
    ref = x.to(torch.bfloat16)
    a, b = h.cpu().view(torch.int16), ref.view(torch.int16)
    nan = torch.isnan(ref)
    assert torch.equal(a[~nan], b[~nan])            # ties to even (1.00390625 -> 1.0, 1.01171875 -> 1.015625), overflow to inf
    assert torch.isnan(h.cpu()[nan]).all()
    back = B.from_bf16(h).cpu()
    assert torch.equal(back[~nan], ref.float()[~nan])


def test_conversion_keeps_channels_last_strides():
    import siv.backend as sb
    B = sb.get_backend()
    x = torch.randn(2, 8, 4, 6, 10, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    h = B.to_bf16(x)
    assert h.stride() == x.stride() and torch.equal(h, x.to(torch.bfloat16))
    assert torch.equal(B.from_bf16(h), x.to(torch.bfloat16).float())


def _make(seed_w, seed_d, dims, Bn):
    import siv.models.models as models
    torch.manual_seed(seed_w)
    net = models.SoftIntroVAE(*WIDE)
    gen = torch.Generator().manual_seed(seed_d)
    with torch.no_grad():
        for m in net.modules():
            if type(m) == torch.nn.Conv3d:
                torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    real = torch.rand(Bn, 1, *dims, generator=gen)
    noise = torch.randn(Bn, 1, *(d // 4 for d in dims), generator=gen)
    return net, real, noise


def _step(net, real, noise, hp, draws):
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    opt_e, opt_d = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, hp.beta_rec, hp.beta_neg, hp.beta_kl, hp.gamma_r, hp.scale)
    grads = {}
    oe, od = opt_e.step, opt_d.step
    opt_e.step = lambda: (grads.update({"encoder." + k: p.grad.detach().clone().cpu()
                                        for k, p in net.encoder.named_parameters()}), oe())
    opt_d.step = lambda: (grads.update({"decoder." + k: p.grad.detach().clone().cpu()
                                        for k, p in net.decoder.named_parameters()}), od())
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with srng.use_rng(srng.ReplayRng(draws)):
        st = dict(zip(trainer.STAT_NAMES, step(real.cuda(), noise.cuda()).cpu().tolist()))
    return st, grads, torch.cuda.max_memory_allocated() - base


@pytest.mark.parametrize("native", [True, False])
def test_bf16_storage_step(native, monkeypatch):
    """One full E+D iteration of a 64-128 channel model at 32^3, batch 2, with the saved activations held
    as bf16 -- by the kernels that emit / read the copies themselves (native) and by the conversion kernels
    around the fp32 entry points: forward bit-identical to the fp32 mode, gradients within the stated 6e-2
    of the fp32 oracle, less memory at the peak."""
    import siv.activations as act
    monkeypatch.setattr(act, "NATIVE", native)
    dims, Bn = (32, 32, 32), 2
    hp = O.Hyper(dims=dims)
    net, real, noise = _make(25, 26, dims, Bn)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    ref = O.soft_intro_step(sd, O.Cfg(*WIDE), real, noise, O.Draws(seed=27), hp, {}, {})
    warm, _, _ = _make(25, 26, dims, Bn)                    # first use allocates workspace and pack cache
    _step(warm.to("cuda").train(), real, noise, hp, O.Draws(seed=27))
    del warm
    net = net.to("cuda").train()
    st32, g32, peak32 = _step(net, real, noise, hp, O.Draws(seed=27))
    net2, _, _ = _make(25, 26, dims, Bn)
    net2 = net2.to("cuda").train()
    before = dict(act.packed)
    with act.storage("bf16"):
        st16, g16, peak16 = _step(net2, real, noise, hp, O.Draws(seed=27))
    assert act.get_storage() == "fp32"
    n_packed = act.packed["tensors"] - before["tensors"]
    assert n_packed >= 40, n_packed                       # 13 encoder/decoder passes x the large saved tensors
    # everything the E-step reports is forward arithmetic on the initial weights: identical bits
    for k in ("lossE", "kl_real", "exp_elbo_rec", "exp_elbo_fake"):
        assert st16[k] == st32[k], (k, st16[k], st32[k])
    # the D-step runs after an encoder update that used the rounded activations; Adam's first update is
    # lr * sign(g), so every element whose noise-level gradient changes sign moves the other way: 1e-2
    for mine, r in (("lossE", "lossE"), ("lossD", "lossD"), ("kl_real", "kl_real"), ("fake_kl", "fake_kl"),
                    ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
        np.testing.assert_allclose(st16[mine], float(ref[r]), rtol=1e-2, err_msg=mine)
    worst = worst32 = 0.0
    for k, gr in g16.items():
        rg = ref["gradE" if k.startswith("encoder.") else "gradD"][k]
        if rg is None:
            assert float(gr.abs().max()) == 0.0, k
            continue
        if k.endswith(BN_CANCELLED):
            continue
        m = float(rg.abs().max()) + 1e-30
        err = float((gr - rg).abs().max()) / m
        assert err <= 6e-2, f"{k}: bf16-storage gradient {err:.2e} of max away from the fp32 oracle"
        worst, worst32 = max(worst, err), max(worst32, float((g32[k] - rg).abs().max()) / m)
    assert worst > worst32                                # the mode really was on (and costs accuracy)
    assert peak16 < 0.8 * peak32, (peak16, peak32)
    print(f"bf16 storage: {n_packed} tensors packed, worst gradient error {worst:.2e} of max "
          f"(fp32 mode {worst32:.2e}), peak memory {peak16 / 2**20:.0f} MiB against {peak32 / 2**20:.0f} MiB")


@pytest.mark.parametrize("c,dims,n,groups,pool,use_res,p", [
    (64, (8, 12, 8), 2, 1, 1, False, 0.25), (64, (8, 12, 8), 2, 1, 2, False, 0.0), (256, (10, 12, 10), 4, 2, 1, True, 0.0),
    (128, (20, 24, 20), 2, 2, 2, False, 0.0), (8, (4, 6, 4), 2, 1, 1, True, 0.35),
])
def test_bn_act_bf16_entry_points(c, dims, n, groups, pool, use_res, p):
    """siv_bn_act_fwd_bf16 / siv_bn_act_bwd_bf16: the forward pass that also writes the bf16 copies of its input
    and output (bit-identical to torch's rounding, output unchanged), and the backward pass that reads the
    convolution output from its bf16 copy (bit-identical to the fp32-reading kernels fed the expanded copy)."""
    import siv.backend as sb
    B = sb.get_backend()
    gen = torch.Generator().manual_seed(c + pool + groups)
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last_3d)
    y = cl(torch.randn(n, c, *dims, generator=gen) * 2 + 0.5)
    gamma, beta = (1 + 0.3 * torch.randn(c, generator=gen)).cuda(), (0.2 * torch.randn(c, generator=gen)).cuda()
    res = cl(torch.randn(n, c, *dims, generator=gen)) if use_res else None
    odims = tuple(d // pool for d in dims)
    keep = cl((torch.rand(n, c, *odims, generator=gen) >= p).to(torch.uint8)) if p > 0 else None
    dout = cl(torch.randn(n, c, *odims, generator=gen))
    assert B.bn_bf16_supported(y)
    mean, var = B.bn_stats(y, None, None, None, 0.1, groups)
    out = B.bn_act_fwd(y, mean, var, gamma, beta, res, keep, p, 0.2, pool, 1e-5)
    out2, yh, oh = B.bn_act_fwd(y, mean, var, gamma, beta, res, keep, p, 0.2, pool, 1e-5, emit_bf16=True)
    assert torch.equal(out, out2)
    assert yh.dtype == torch.bfloat16 and torch.equal(yh, y.to(torch.bfloat16)) and yh.stride() == y.stride()
    assert torch.equal(oh, out.to(torch.bfloat16))
    dy, dg, db, dres = B.bn_act_bwd(dout, yh, mean, var, gamma, beta, res, keep, p, 0.2, pool, 1e-5, True, use_res)
    dy1, dg1, db1, dres1 = B.bn_act_bwd(dout, yh.float(), mean, var, gamma, beta, res, keep, p, 0.2, pool, 1e-5, True,
                                        use_res)
    assert torch.equal(dy, dy1) and torch.equal(dg, dg1) and torch.equal(db, db1)
    if use_res:
        assert torch.equal(dres, dres1)
    assert not B.bn_bf16_supported(torch.zeros(1, 6, 2, 2, 2, device="cuda"))
    import siv._lib as L
    y6 = torch.zeros(2, 6, 2, 2, 2, device="cuda").contiguous(memory_format=torch.channels_last_3d)
    m6, v6 = B.bn_stats(y6, None, None, None, 0.1)
    g6 = torch.ones(6, device="cuda")
    with pytest.raises(L.SivError):                        # no float4 kernel for 6 channels: no bf16 copies
        B.bn_act_fwd(y6, m6, v6, g6, g6, None, None, 0.0, 0.2, 1, 1e-5, emit_bf16=True)


def test_fused_stem_and_output_layer_bf16_entry_points():
    """The fused encoder stem writes the bf16 copy of its output beside it (siv_conv1_bnact_fwd_bf16); the fused
    decoder end (BatchNorm + LeakyReLU inside the 64 -> 1 convolution) writes the bf16 copy of the raw tensor it
    normalises (siv_bnact_conv3d_k3_to1_fwd_bf16) and both of its backward operators read that copy
    (..._dgrad_bf16, ..._wgrad_bf16).  Copies: bit-identical to torch's rounding, every voxel written once;
    backward from the copy: bit-identical to the fp32-reading kernels fed the expanded copy."""
    import siv.backend as sb
    B = sb.get_backend()
    gen = torch.Generator().manual_seed(77)
    cl = lambda t: t.cuda().contiguous(memory_format=torch.channels_last_3d)
    # ---- stem: Conv3d(1, 64, 3) -> BatchNorm -> LeakyReLU -> Dropout
    n, dims, C = 2, (12, 20, 18), 64
    x = torch.rand(n, 1, *dims, generator=gen).cuda()
    w = cl(torch.randn(C, 1, 3, 3, 3, generator=gen) * 0.3)
    bias, gamma, beta = (torch.randn(C, generator=gen) * 0.1).cuda(), (1 + 0.2 * torch.randn(C, generator=gen)).cuda(), \
        (0.1 * torch.randn(C, generator=gen)).cuda()
    keep = cl((torch.rand(n, C, *dims, generator=gen) >= 0.35).to(torch.uint8))
    rm, rv = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    out, mean, var = B.conv1_bnact_fwd(x, w, bias, gamma, beta, keep, 0.35, 0.2, 1e-5, True, rm.clone(), rv.clone(),
                                       None, 0.1)
    out2, mean2, var2, oh = B.conv1_bnact_fwd(x, w, bias, gamma, beta, keep, 0.35, 0.2, 1e-5, True, rm.clone(),
                                              rv.clone(), None, 0.1, emit_bf16=True)
    assert torch.equal(out, out2) and torch.equal(mean, mean2) and torch.equal(var, var2)
    assert oh.dtype == torch.bfloat16 and oh.stride() == out.stride() and torch.equal(oh, out.to(torch.bfloat16))
    # ---- decoder end: BatchNorm -> LeakyReLU -> Conv3d(64, 1, 3) on a raw 64-channel tensor
    n, dims = 2, (18, 24, 35)                      # a partial 16-wide column, several z segments
    y = cl(torch.randn(n, C, *dims, generator=gen) * 1.5 + 0.2)
    w1 = cl(torch.randn(1, C, 3, 3, 3, generator=gen) * 0.05)
    b1 = torch.randn(1, generator=gen).cuda()
    dout = cl(torch.randn(n, 1, *dims, generator=gen))
    assert B.bnact_conv_to1_supported(y)
    m, v = B.bn_stats(y, None, None, None, 0.1)
    o1 = B.bnact_conv_to1_fwd(y, m, v, gamma, beta, 1e-5, 0.2, w1, b1)
    o2, yh = B.bnact_conv_to1_fwd(y, m, v, gamma, beta, 1e-5, 0.2, w1, b1, emit_bf16=True)
    assert torch.equal(o1, o2)
    assert yh.dtype == torch.bfloat16 and yh.stride() == y.stride() and torch.equal(yh, y.to(torch.bfloat16))
    yx = yh.float()
    dx, dg, db = B.bnact_conv_to1_dgrad(dout, w1, yh, m, v, gamma, beta, 1e-5, 0.2, True)
    dx1, dg1, db1 = B.bnact_conv_to1_dgrad(dout, w1, yx, m, v, gamma, beta, 1e-5, 0.2, True)
    assert torch.equal(dx, dx1) and torch.equal(dg, dg1) and torch.equal(db, db1)
    dw, dcb = B.bnact_conv_to1_wgrad(yh, m, v, gamma, beta, 1e-5, 0.2, dout, True)
    dw1, dcb1 = B.bnact_conv_to1_wgrad(yx, m, v, gamma, beta, 1e-5, 0.2, dout, True)
    assert torch.equal(dw, dw1) and torch.equal(dcb, dcb1)
