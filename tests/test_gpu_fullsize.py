"""Oracle parity AT THE SIZE THE BENCHMARK RUNS (80x96x80, z=1200 model, 64..256 channels) on the
automatic kernel dispatch: the size-gated kernels (two-dimensional Winograd forward / input gradient /
weight gradient, product-form / box-form upsampled kernels, Winograd weight gradient behind an upsample,
column-walking 64 -> 1 kernel) are compared with the CPU oracle itself, not only through properties.

Each of the ten distinct 3x3x3 convolution shapes of the z=1200 model runs forward (with the fused
BatchNorm statistics, as in training), input gradient and weight gradient through the C ABI on the
volumes the model gives it; the batch is chosen so that the dispatcher takes the same kernel as at the
benchmark's batch 8 (asserted through siv_last_conv_kernel).  Then one full E-step + D-step of
SoftIntroVAE(64,[[64,1,2],[128,1,2],[256,2,2]]) at 80x96x80, batch 1, against O.soft_intro_step with
replayed random draws: losses 1e-4 (2e-3 for the D-step's KL terms, see the test), every gradient 2e-3 of its maximum (LeakyReLU kinks, see
tests/test_gpu_model.py), post-Adam weights, BatchNorm running statistics.

Tolerance of the per-kernel comparisons: 1e-4 of the tensor maximum (fp32)."""
import numpy as np
import pytest
import torch

from oracle import siv_oracle as O
from tests.parity import check_max, check_params_after_adam

pytestmark = pytest.mark.gpu

V0, V1, V2, V3 = (80, 96, 80), (40, 48, 40), (20, 24, 20), (10, 12, 10)
# cin, cout, output dims, batch, ups, expected kernel family (fwd, dgrad, wgrad)
FULL_CONV_CASES = [
    (1, 64, V0, 1, 1, ("thin", "thin", "thin")),                  # encoder stem (models/models.py:92)
    (64, 1, V0, 1, 1, ("thin", "thin", "thin")),                  # decoder output layer (:137)
    (64, 64, V0, 1, 1, ("wino2d", "wino2d", "wino2d")),           # encoder block 1, conv 1 (:17)
    (64, 64, V0, 1, 2, ("up", "up", "up_wino")),                  # decoder last block, conv 2 (:57-59)
    (64, 64, V1, 2, 1, ("wino2d", "wino2d", "wino2d")),
    (64, 128, V1, 2, 1, ("wino2d", "wino2d", "wino2d")),
    (128, 64, V1, 2, 2, ("up", "up", "up_wino")),
    (128, 128, V2, 8, 1, ("wino2d", "wino2d", "wino2d")),
    (128, 256, V2, 8, 1, ("wino2d", "wino2d", "wino2d")),
    (256, 128, V2, 8, 2, ("up", "up", "up_wino")),
    (256, 256, V3, 8, 1, ("wino2d", "wino2d", "wino2d")),
]


@pytest.fixture(scope="module")
def B():
    import siv.backend as sb
    sb.set_backend(None)
    b = sb.get_backend()
    assert b.name == "hip"
    b.set_conv_path(0)
    return b


@pytest.mark.parametrize("cin,cout,dims,n,ups,kern", FULL_CONV_CASES)
def test_benchmark_conv_shapes_match_oracle(B, cin, cout, dims, n, ups, kern):
    gen = torch.Generator().manual_seed(cin * 7 + cout + ups + dims[0])
    D, H, W = dims
    x = torch.randn(n, cin, D // ups, H // ups, W // ups, generator=gen) + 0.25
    w = torch.randn(cout, cin, 3, 3, 3, generator=gen) * (2.0 / (27 * cin)) ** 0.5
    thin = cin == 1 or cout == 1
    bias = torch.randn(cout, generator=gen) if thin else None        # the wide convolutions are bias-free
    dy = torch.randn(n, cout, D, H, W, generator=gen)
    rm, rv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    bd = bias.cuda() if bias is not None else None
    y, mean, var = B.conv3d_k3_fwd_stats(xd, wd, bd, ups, rm, rv, None, 0.1)
    k_f = B.last_conv_kernel()
    dx = B.conv3d_k3_dgrad(dyd, wd, ups)
    k_d = B.last_conv_kernel()
    dw, db = B.conv3d_k3_wgrad(xd, dyd, ups, bias is not None)
    k_w = B.last_conv_kernel()
    assert (k_f, k_d, k_w) == kern, (k_f, k_d, k_w)

    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = O.conv3d_k3(xr, wr, bias, ups)
    gx, gw = torch.autograd.grad(yr, [xr, wr], dy)
    e = {"fwd": check_max(y, yr, "fwd"), "dgrad": check_max(dx, gx, "dgrad"), "wgrad": check_max(dw, gw, "wgrad")}
    if bias is not None:
        check_max(db, dy.sum(dim=(0, 2, 3, 4)), "bias grad")
    yd64 = yr.detach().double()
    check_max(mean, yd64.mean(dim=(0, 2, 3, 4)).float(), "BN mean", floor=1e-6)
    check_max(var, yd64.var(dim=(0, 2, 3, 4), unbiased=False).float(), "BN biased variance")
    m = yr.numel() // cout
    check_max(rv, (0.9 + 0.1 * yd64.var(dim=(0, 2, 3, 4), unbiased=False) * m / (m - 1)).float(), "running var")
    print(f"{cin}->{cout} @{dims} ups={ups} n={n} [{k_f},{k_d},{k_w}]: "
          + ", ".join(f"{k} {v:.1e}" for k, v in e.items()))


Z1200 = (64, [[64, 1, 2], [128, 1, 2], [256, 2, 2]])       # z-1200main.py:158


def test_z1200_step_at_full_size_matches_oracle():
    """One E-step + D-step of the benchmark model at the benchmark volume size, batch 1."""
    import siv.models.models as models
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    torch.manual_seed(82)
    net = models.SoftIntroVAE(*Z1200)
    torch.manual_seed(77)
    net.apply(trainer.init_weights_he)
    gen = torch.Generator().manual_seed(103)
    with torch.no_grad():          # BN affine parameters away from (1, 0) so their gradients matter
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sd = {k: v.detach().clone() for k, v in sd0.items()}
    real = torch.rand(1, 1, *V0, generator=gen)
    noise = torch.randn(1, 1, *V3, generator=gen)
    hp = O.Hyper(dims=V0)
    ref = O.soft_intro_step(sd, O.Cfg(*Z1200), real, noise, O.Draws(seed=41), hp, {}, {})

    net = net.to("cuda").train()
    opt_e = FlatAdam(net.encoder.parameters(), lr=hp.lr)
    opt_d = FlatAdam(net.decoder.parameters(), lr=hp.lr)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, hp.beta_rec, hp.beta_neg, hp.beta_kl, hp.gamma_r, hp.scale)
    grads = {}
    oe, od = opt_e.step, opt_d.step
    opt_e.step = lambda: (grads.update({"encoder." + k: p.grad.detach().clone().cpu()
                                        for k, p in net.encoder.named_parameters()}), oe())
    opt_d.step = lambda: (grads.update({"decoder." + k: p.grad.detach().clone().cpu()
                                        for k, p in net.decoder.named_parameters()}), od())
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=41))):
        st = dict(zip(trainer.STAT_NAMES, step(real.cuda(), noise.cuda()).cpu().tolist()))
    # E-step scalars at 1e-4.  The D-step runs on the encoder AFTER its Adam update, which moves every
    # element by +-lr whatever the size of its gradient: noise-level gradient elements take either sign, and
    # at batch 1 the KL of a decoded volume (~1e4, dominated by exp(logvar)) answers to that with ~1e-3.
    print({k: (st[k], float(ref[r])) for k, r in (("lossE", "lossE"), ("lossD", "lossD"), ("kl_real", "kl_real"),
                                                  ("fake_kl", "fake_kl"), ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D"))})
    for mine, r, rt in (("lossE", "lossE", 1e-4), ("kl_real", "kl_real", 1e-4), ("lossD", "lossD", 2e-3),
                        ("fake_kl", "fake_kl", 2e-3), ("rec_kl", "rec_kl", 2e-3), ("loss_rec", "loss_rec_D", 1e-4)):
        np.testing.assert_allclose(st[mine], float(ref[r]), rtol=rt, err_msg=mine)
    for mine in ("exp_elbo_rec", "exp_elbo_fake"):
        b = float(ref[mine])
        if b > 1e-44:
            np.testing.assert_allclose(np.log(max(st[mine], 1e-45)), np.log(b), rtol=1e-4, err_msg=mine)
        else:
            assert st[mine] < 1e-40, mine
    cancelled = ("encoder.blocks.0.0.bias", "decoder.blocks.0.0.bias", "decoder.blocks.0.0.weight")
    worst = 0.0
    for k, gr in grads.items():
        rg = ref["gradE" if k.startswith("encoder.") else "gradD"][k]
        if rg is None:
            assert float(gr.abs().max()) == 0.0, k
        elif not k.endswith(cancelled):       # BN cancels these gradients to rounding noise
            worst = max(worst, check_max(gr, rg, k, tol=2e-3, floor=3e-7))
    print(f"z=1200 full-size step: worst gradient error {worst:.2e} of max")
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(sd[k]), k
        elif k.endswith(("running_mean", "running_var")):
            check_max(v, sd[k], k, tol=1e-4, floor=5e-4 if k.endswith("blocks.0.1.running_mean") else 1e-5)
        elif not k.endswith(cancelled):
            rg = ref["gradE" if k.startswith("encoder.") else "gradD"][k]
            check_params_after_adam(v, sd[k], rg, hp.lr, k)
