"""Model- and step-level parity on a real MI355X: the product path (libsiv through the C ABI)
against (a) the committed reference-generated fixtures and (b) the CPU oracle run on the spot
for configurations that exercise the matrix-core and single-channel kernels, plus
size-independent properties at the full benchmark size (80x96x80, 64-128-256 channels).

Tolerances: 1e-4 relative (of the tensor maximum) for everything continuous -- activations,
losses, BN statistics.  End-to-end gradients cross ~10^6 LeakyReLU/ReLU kinks; one
pre-activation landing on the other side of zero in one implementation shifts all upstream
gradients by 1e-4..1e-3 of their maximum (see tests/test_host_logic.py), so those use 1e-3
(narrow fixture model) and 2e-3 (wide on-the-spot model); every kernel on the path is pinned
at 1e-4 by tests/test_gpu_ops.py."""
import os

import numpy as np
import pytest
import torch

from oracle import siv_oracle as O
from tests.parity import check_params_after_adam

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
BN_CANCELLED = ("encoder.blocks.0.0.bias", "decoder.blocks.0.0.bias", "decoder.blocks.0.0.weight")


@pytest.fixture(autouse=True)
def hip_backend():
    import siv.backend as sb
    import siv.rng as srng
    sb.set_backend(None)
    assert sb.get_backend().name == "hip"
    yield
    srng.set_rng(srng.DeviceRng(0))


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sd_from(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}


def check(a, b, what, tol=1e-4, floor=0.0):
    a = a.detach().float().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a, dtype=np.float32)
    b = b.detach().float().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b, dtype=np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), what
    err = float(np.abs(a - b).max())
    lim = tol * float(np.abs(b).max()) + floor
    assert err <= lim, f"{what}: max abs err {err:.3e} > {lim:.3e} (max|ref| {np.abs(b).max():.3e})"


def test_g3_forward_matches_reference_fixture():
    import siv.models.models as models
    import siv.rng as srng
    g = load("g3_model.npz")
    net = models.SoftIntroVAE(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    net.load_state_dict(sd_from(g, "sd0."))
    net = net.to("cuda")
    x, z_in = torch.from_numpy(g["x"]).cuda(), torch.from_numpy(g["z_in"]).cuda()
    net.train()
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))):
        mu, lv = net.encode(x)
        xd = net.decode(z_in)
        fmu, flv, fz, fx = net.forward(x)
    for a, k in ((mu, "enc.mu"), (lv, "enc.logvar"), (xd, "dec.x"), (fmu, "fwd.mu"),
                 (flv, "fwd.logvar"), (fz, "fwd.z"), (fx, "fwd.x")):
        check(a, g[k], k)
    ref1 = sd_from(g, "sd1.")
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref1[k]), k
        else:
            check(v, ref1[k], k, floor=1e-6)
    net.eval()
    with torch.no_grad():
        mu, lv = net.encode(x)
        z = net.reparameterize(mu, lv, True)
        xe = net.decode(z)
    for a, k in ((mu, "eval.mu"), (lv, "eval.logvar"), (z, "eval.z"), (xe, "eval.x")):
        check(a, g[k], k)


sd_after = []          # state_dict after each iteration of the last _run_steps call


def _run_steps(net, g_inputs, hp, draws, n_steps, lr=2e-4):
    del sd_after[:]
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    opt_e = FlatAdam(net.encoder.parameters(), lr=lr)
    opt_d = FlatAdam(net.decoder.parameters(), lr=lr)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, hp.beta_rec, hp.beta_neg, hp.beta_kl, hp.gamma_r, hp.scale)
    grads, out = {}, []
    oe, od = opt_e.step, opt_d.step
    opt_e.step = lambda: (grads.update({"encoder." + k: p.grad.detach().clone().cpu()
                                        for k, p in net.encoder.named_parameters()}), oe())
    opt_d.step = lambda: (grads.update({"decoder." + k: p.grad.detach().clone().cpu()
                                        for k, p in net.decoder.named_parameters()}), od())
    with srng.use_rng(srng.ReplayRng(draws)):
        for s in range(n_steps):
            real, noise = g_inputs(s)
            st = dict(zip(trainer.STAT_NAMES, step(real.cuda(), noise.cuda()).cpu().tolist()))
            out.append((st, dict(grads)))
            sd_after.append({k: v.detach().clone().cpu() for k, v in net.state_dict().items()})
    return out


def test_g4_step_matches_reference_fixture():
    import siv.models.models as models
    g = load("g4_step.npz")
    net = models.SoftIntroVAE(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    net.load_state_dict(sd_from(g, "sd0."))
    net = net.to("cuda").train()
    hp = O.Hyper(beta_rec=float(g["hp.beta_rec"]), beta_neg=float(g["hp.beta_neg"]),
                 beta_kl=float(g["hp.beta_kl"]), gamma_r=float(g["hp.gamma_r"]), scale=float(g["hp.scale"]))
    draws = O.Draws(seed=int(g["draw_seed"]))
    res = _run_steps(net, lambda s: (torch.from_numpy(g[f"s{s}.real"]), torch.from_numpy(g[f"s{s}.noise"])),
                     hp, draws, 2)
    assert draws.i == int(g["n_draws"])
    for s, (st, grads) in enumerate(res):
        srt = 1e-4 if s == 0 else 1e-3     # iteration 2 starts from weights that differ by Adam sign noise
        for mine, ref in (("lossE", "lossE"), ("lossD", "lossD"), ("kl_real", "kl_real"),
                          ("fake_kl", "fake_kl"), ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
            np.testing.assert_allclose(st[mine], float(g[f"s{s}.{ref}"]), rtol=srt, err_msg=f"s{s}.{mine}")
        for mine in ("exp_elbo_rec", "exp_elbo_fake"):
            b = float(g[f"s{s}.{mine}"])
            if b > 1e-44:
                np.testing.assert_allclose(np.log(max(st[mine], 1e-45)), np.log(b), rtol=srt, err_msg=mine)
        if s == 0:
            for k, gr in grads.items():
                which = "gradE" if k.startswith("encoder.") else "gradD"
                key = f"s0.{which}.{k}"
                if key + ".none" in g.files:
                    assert float(gr.abs().max()) == 0.0, key
                elif not k.endswith(BN_CANCELLED):
                    check(gr, g[key], key, tol=1e-3, floor=3e-7)
    ref_sd = sd_from(g, "s1.sd.")
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref_sd[k]), k
        elif k.endswith(("running_mean", "running_var")):
            check(v, ref_sd[k], k, tol=2e-3, floor=2e-4)
    # weights after the first iteration's two Adam updates (on the device: siv_adam_step over the flat buffers)
    ref_sd0 = sd_from(g, "s0.sd.")
    for k, v in sd_after[0].items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref_sd0[k]), k
        elif k.endswith(("running_mean", "running_var")):
            check(v, ref_sd0[k], k, tol=1e-4, floor=1e-4)
        elif k.endswith(BN_CANCELLED):
            np.testing.assert_allclose(v.numpy(), ref_sd0[k].numpy(), rtol=1e-5, atol=2 * 2e-4 + 2e-6, err_msg=k)
        else:
            gk = f"s0.{'gradE' if k.startswith('encoder.') else 'gradD'}.{k}"
            check_params_after_adam(v, ref_sd0[k], g[gk] if gk in g.files else None, 2e-4, k)


def test_g6_plain_vae_matches_reference_fixture():
    import siv.models.vaemodel as vaemodel
    import siv.models.lossf as lossf
    import siv.rng as srng
    from siv.optim import FlatAdam
    g = load("g6_vae.npz")
    net = vaemodel.ResNetVAE(4, [[4, 1, 2], [6, 1, 2], [8, 2, 2], [12, 2, 2]])
    net.load_state_dict(sd_from(g, "sd0."))
    net = net.to("cuda").train()
    opt = FlatAdam(net.parameters(), 1e-3)
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))):
        x = torch.from_numpy(g["s0.x"]).cuda()
        opt.zero_grad()
        x_re, mu, logvar = net.forward(x)
        loss, mse, kl = lossf.normal_loss(x_re, mu, logvar, x, 1, 20)
        loss.backward()
        check(torch.stack([loss, mse, kl]), g["s0.loss"], "losses")
        check(mu, g["s0.mu"], "mu")
        check(x_re[:, :, 16], g["s0.x_re_slice"], "x_re slice")
        for k, p in net.named_parameters():
            key = f"s0.grad.{k}"
            if key + ".none" in g.files:
                assert float(p.grad.abs().max()) == 0.0
            elif not k.endswith(BN_CANCELLED):
                check(p.grad, g[key], key, tol=1e-3, floor=3e-7)
        opt.step()


WIDE = (64, [[64, 1, 2], [128, 1, 2]])


def test_wide_model_step_matches_oracle():
    """A 64-128 channel model (fp32 matrix-core conv kernels, single-channel stem/tail kernels,
    x2 upsample fused into the gather) through one full E+D iteration against the CPU oracle
    computed on the spot (a few seconds of CPU)."""
    import siv.models.models as models
    dims, Bn = (16, 16, 16), 2
    torch.manual_seed(5)
    net = models.SoftIntroVAE(*WIDE)
    gen = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in net.modules():
            if type(m) == torch.nn.Conv3d:
                torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    real = torch.rand(Bn, 1, *dims, generator=gen)
    noise = torch.randn(Bn, 1, 4, 4, 4, generator=gen)
    cfg, hp = O.Cfg(*WIDE), O.Hyper(dims=dims)
    ref = O.soft_intro_step(sd, cfg, real, noise, O.Draws(seed=17), hp, {}, {})
    net = net.to("cuda").train()
    (st, grads), = _run_steps(net, lambda s: (real, noise), hp, O.Draws(seed=17), 1)
    for mine, r in (("lossE", "lossE"), ("lossD", "lossD"), ("kl_real", "kl_real"), ("fake_kl", "fake_kl"),
                    ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
        np.testing.assert_allclose(st[mine], float(ref[r]), rtol=1e-4, err_msg=mine)
    worst = 0.0
    for k, gr in grads.items():
        rg = ref["gradE" if k.startswith("encoder.") else "gradD"][k]
        if rg is None:
            assert float(gr.abs().max()) == 0.0, k
        elif not k.endswith(BN_CANCELLED):
            check(gr, rg, k, tol=2e-3, floor=3e-7)
            worst = max(worst, float((gr - rg).abs().max() / (rg.abs().max() + 1e-30)))
    print(f"wide-model step: worst gradient error {worst:.2e} of max")
    for k, v in net.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            # the stem's running mean contains its BN-cancelled conv bias, which Adam moves by
            # +-lr in a noise-determined direction between the E-step and D-step passes
            check(v, sd[k], k, tol=1e-4, floor=5e-4 if k.endswith("blocks.0.1.running_mean") else 1e-5)


def test_split_bf16_path_step_matches_fp32_path():
    """Conv path 5 (fp32 products formed from bf16 triples on the bf16 matrix cores) against the
    exact-fp32 kernels of the same structure (path 3: Winograd wherever supported) through one full
    E+D iteration of a 64-128 channel model: every loss term to 1e-5 relative, gradient norms per
    parameter to 2 % (single gradient elements are LeakyReLU-sign-flip noise at this size for ANY
    change of rounding, the two fp32 kernels included)."""
    import siv.backend as sb
    import siv.models.models as models
    dims, Bn = (16, 24, 16), 2
    gen = torch.Generator().manual_seed(16)
    real = torch.rand(Bn, 1, *dims, generator=gen)
    noise = torch.randn(Bn, 1, 4, 6, 4, generator=gen)
    hp = O.Hyper(dims=dims)
    res = {}
    for path in (3, 5):
        torch.manual_seed(15)
        net = models.SoftIntroVAE(*WIDE)
        with torch.no_grad():
            for m in net.modules():
                if type(m) == torch.nn.Conv3d:
                    torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
        net = net.to("cuda").train()
        sb.get_backend().set_conv_path(path)
        try:
            (st, grads), = _run_steps(net, lambda s: (real, noise), hp, O.Draws(seed=23), 1)
        finally:
            sb.get_backend().set_conv_path(0)
        res[path] = (st, grads)
    for k in ("lossE", "lossD", "kl_real", "fake_kl", "rec_kl", "loss_rec"):
        np.testing.assert_allclose(res[5][0][k], res[3][0][k], rtol=1e-5, err_msg=k)
    for k, g3 in res[3][1].items():
        if k.endswith(BN_CANCELLED) or float(g3.norm()) == 0.0:
            continue
        np.testing.assert_allclose(float(res[5][1][k].norm()), float(g3.norm()), rtol=2e-2, err_msg=k)


def test_large_volume_kernels_step_matches_direct_kernels():
    """One full E+D iteration of the 64-128 channel model at 48x48x48, batch 2 -- large enough for the
    automatic dispatch (path 0) to pick every large-volume kernel: Winograd F(2x2,3x3) forward / input
    gradient, Winograd weight gradients (plain and upsampled), the box-form upsampled input gradient and
    the column-walking 64 -> 1 kernel -- against the same iteration on the direct matrix-core kernels
    (path 4).  Loss terms to 1e-4 relative, gradient norms per parameter to 2 % (exact fp32 arithmetic
    on both sides; only the order of the additions differs)."""
    import siv.backend as sb
    import siv.models.models as models
    dims, Bn = (48, 48, 48), 2
    gen = torch.Generator().manual_seed(26)
    real = torch.rand(Bn, 1, *dims, generator=gen)
    noise = torch.randn(Bn, 1, 12, 12, 12, generator=gen)
    hp = O.Hyper(dims=dims)
    res = {}
    for path in (4, 0):
        torch.manual_seed(25)
        net = models.SoftIntroVAE(*WIDE)
        with torch.no_grad():
            for m in net.modules():
                if type(m) == torch.nn.Conv3d:
                    torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
        net = net.to("cuda").train()
        sb.get_backend().set_conv_path(path)
        try:
            (st, grads), = _run_steps(net, lambda s: (real, noise), hp, O.Draws(seed=27), 1)
        finally:
            sb.get_backend().set_conv_path(0)
        res[path] = (st, grads)
    for k in ("lossE", "lossD", "kl_real", "fake_kl", "rec_kl", "loss_rec"):
        np.testing.assert_allclose(res[0][0][k], res[4][0][k], rtol=1e-4, err_msg=k)
    for k, g4 in res[4][1].items():
        if k.endswith(BN_CANCELLED) or float(g4.norm()) == 0.0:
            continue
        np.testing.assert_allclose(float(res[0][1][k].norm()), float(g4.norm()), rtol=2e-2, err_msg=k)


def test_eval_forward_wide_matches_oracle():
    import siv.models.models as models
    torch.manual_seed(8)
    net = models.SoftIntroVAE(*WIDE)
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.running_mean.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
                m.running_var.copy_(0.5 + torch.rand(m.bias.shape, generator=gen))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.rand(1, 1, 8, 16, 24, generator=gen)
    cfg = O.Cfg(*WIDE)
    mu_r, lv_r = O.encode(x, sd, cfg, None, False)
    z_r = O.reparameterize(mu_r, lv_r, None, True)
    x_r = O.decode(z_r, sd, cfg, None, False)
    net = net.to("cuda").eval()
    with torch.no_grad():
        mu, lv = net.encode(x.cuda())
        z = net.reparameterize(mu, lv, True)
        xe = net.decode(z)
    check(mu, mu_r, "mu")
    check(lv, lv_r, "logvar")
    check(z, z_r, "z")
    check(xe, x_r, "x_re")


# ---------------------------------------------------------------------------------------
# full benchmark size: size-independent properties (no oracle needed)
# ---------------------------------------------------------------------------------------
FULL = (80, 96, 80)


def _dot(a, b):
    return float((a.double() * b.double()).sum())


@pytest.mark.parametrize("cin,cout,dims,n,ups", [
    (64, 64, FULL, 1, 1),            # 60 % of the model's FLOPs
    (64, 64, FULL, 1, 2),            # decoder's last block: x2 upsample fused
    (1, 64, FULL, 2, 1),             # encoder stem
    (64, 1, FULL, 2, 1),             # decoder output layer
    (128, 128, (20, 24, 20), 2, 1), (256, 256, (10, 12, 10), 8, 1), (256, 128, (20, 24, 20), 1, 2),
    # the 160x192x160 volumes of the augmented configuration: 1.26 GB tensors, > 2^28 elements
    (64, 64, (160, 192, 160), 1, 1), (1, 64, (160, 192, 160), 1, 1), (64, 1, (160, 192, 160), 1, 1),
    (64, 64, (80, 96, 80), 16, 1),     # batch 16: > 2^31 bytes per tensor
])
def test_full_size_conv_adjoint_identities(cin, cout, dims, n, ups):
    """<conv(x,w), dy> == <x, dgrad(dy,w)> == <w, wgrad(x,dy)>: the three kernels of a layer must be
    exact adjoints of one another (bilinearity), at the sizes the benchmark runs."""
    import siv.backend as sb
    B = sb.get_backend()
    gen = torch.Generator(device="cuda").manual_seed(cin + cout + ups)
    D, H, W = dims
    x = torch.randn(n, cin, D // ups, H // ups, W // ups, device="cuda", generator=gen).contiguous(
        memory_format=torch.channels_last_3d)
    w = (torch.randn(cout, cin, 3, 3, 3, device="cuda", generator=gen) * (2.0 / (27 * cin)) ** 0.5).contiguous(
        memory_format=torch.channels_last_3d)
    dy = torch.randn(n, cout, D, H, W, device="cuda", generator=gen).contiguous(memory_format=torch.channels_last_3d)
    y = B.conv3d_k3_fwd(x, w, None, ups)
    dx = B.conv3d_k3_dgrad(dy, w, ups)
    dw, _ = B.conv3d_k3_wgrad(x, dy, ups, False)
    a, b, c = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    scale = float(y.double().norm() * dy.double().norm())
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)
    # linearity: conv(2x) == 2 conv(x) exactly in fp32 (power-of-two scaling)
    y2 = B.conv3d_k3_fwd(x * 2, w, None, ups)
    assert torch.equal(y2, y * 2)
    # a delta input reads the kernel back: y[:, :, 1+kd', 1+kh', 1+kw'] = w[:, ci, 2-kd', ...]
    if ups == 1:
        xd = torch.zeros_like(x[:1])
        xd[0, 0, 2, 2, 2] = 1.0
        yd = B.conv3d_k3_fwd(xd, w, None, 1)
        # (to fp32 rounding: the Winograd kernels reconstruct each tap from sums of transformed taps)
        got, want = yd[0, :, 1:4, 1:4, 1:4].cpu(), w[:, 0].flip(1, 2, 3).cpu()
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())


def test_full_size_bn_tail_properties():
    import siv.backend as sb
    B = sb.get_backend()
    gen = torch.Generator(device="cuda").manual_seed(3)
    y = (torch.randn(2, 64, *FULL, device="cuda", generator=gen) * 3 + 1).contiguous(
        memory_format=torch.channels_last_3d)
    ones, zeros = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    mean, var = B.bn_stats(y, None, None, None, 0.1)
    out = B.bn_act_fwd(y, mean, var, ones, zeros, None, None, 0.0, 1.0, 1, 1e-5)    # slope 1 = identity act
    m2, v2 = B.bn_stats(out, None, None, None, 0.1)
    assert float(m2.abs().max()) < 1e-4 and float((v2 - 1).abs().max()) < 1e-4       # normalised
    pooled = B.bn_act_fwd(y, mean, var, ones, zeros, None, None, 0.0, 1.0, 2, 1e-5)
    assert pooled.shape == (2, 64, 40, 48, 40)
    ref = torch.nn.functional.avg_pool3d(out, 2)
    assert float((pooled - ref).abs().max()) < 1e-5
    # gradient of sum(out) w.r.t. y is zero through training-mode BN (mean subtraction)
    dout = torch.ones_like(out)
    dy, dg, db, _ = B.bn_act_bwd(dout, y, mean, var, ones, zeros, None, None, 0.0, 1.0, 1, 1e-5, True, False)
    assert float(dy.abs().max()) < 1e-5
    assert abs(float(db[0]) - 2 * 80 * 96 * 80) < 1.0


# ---------------------------------------------------------------------------------------
# FC-latent model (models/mymodel.py + utils/trainer_fc.py)
# ---------------------------------------------------------------------------------------
def test_g7_fc_model_matches_reference_fixture():
    """Forward (train + eval) and the first E+D iteration of the narrow FC-latent fixture at the
    full 80x96x80 extent.  Gradients are compared through float64 truth (tests/fc_truth.py)."""
    import siv.models.mymodel as mymodel
    import siv.rng as srng
    from tests.test_oracle_golden import fc_inputs
    from tests.fc_truth import assert_grad_as_good_as_reference, fc_step0_fp64
    g = load("g7_fc.npz")
    cfgt, x, zz, steps = fc_inputs(g)
    net = mymodel.SoftIntroVAE(*cfgt)
    net.load_state_dict(sd_from(g, "sd0."))
    net = net.to("cuda").train()
    draws = O.Draws(seed=int(g["draw_seed"]))
    with srng.use_rng(srng.ReplayRng(draws)):
        fmu, flv, fz, fx = net.forward(x.cuda())
        xd = net.decode(zz.cuda())
    for a, k in ((fmu, "fwd.mu"), (flv, "fwd.logvar"), (fz, "fwd.z"), (fx[:, :, 40], "fwd.x_slice"),
                 (xd[:, :, 40], "dec.x_slice")):
        check(a, g[k], k)
    ref1 = sd_from(g, "sd1.")
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref1[k]), k
        else:
            check(v, ref1[k], k, floor=1e-6)
    net.eval()
    with torch.no_grad():
        emu, elv = net.encode(x.cuda())
        ex = net.decode(zz.cuda())
    for a, k in ((emu, "eval.mu"), (elv, "eval.logvar"), (ex[:, :, 40], "eval.x_slice")):
        check(a, g[k], k)
    net.train()
    hp = O.Hyper(beta_rec=float(g["hp.beta_rec"]), beta_neg=float(g["hp.beta_neg"]),
                 beta_kl=float(g["hp.beta_kl"]), gamma_r=float(g["hp.gamma_r"]), scale=float(g["hp.scale"]))
    (st, grads), = _run_steps(net, lambda s: steps[s], hp, draws, 1)
    # E-step scalars at 1e-4; the D-step runs on the encoder AFTER its first Adam update, which is
    # sign-like (+-lr per element) and so turns noise-level gradient differences into weight
    # differences: its scalars are pinned at 1e-3
    for mine, ref, rt in (("lossE", "lossE", 1e-4), ("kl_real", "kl_real", 1e-4), ("lossD", "lossD", 1e-3),
                          ("fake_kl", "fake_kl", 1e-3), ("rec_kl", "rec_kl", 1e-3),
                          ("loss_rec", "loss_rec_D", 1e-3)):
        np.testing.assert_allclose(st[mine], float(g[f"s0.{ref}"]), rtol=rt, err_msg=mine)
    for mine in ("exp_elbo_rec", "exp_elbo_fake"):
        np.testing.assert_allclose(np.log(st[mine]), np.log(float(g[f"s0.{mine}"])), rtol=1e-4, err_msg=mine)
    truth = fc_step0_fp64()
    for k, gr in grads.items():
        which = "gradE" if k.startswith("encoder.") else "gradD"
        key = f"s0.{which}.{k}"
        if key + ".none" in g.files:
            assert float(gr.abs().max()) == 0.0, key
        elif float(np.abs(truth[f"{which}.{k}"]).max()) > 1e-4:     # skips BN-cancelled conv biases
            assert_grad_as_good_as_reference(gr.numpy(), g[key], truth[f"{which}.{k}"], key)


FC_WIDE = (8, 16, 32, 32, 48)


def test_fc_wide_forward_matches_oracle():
    """8-16-32-32 channel FC-latent model (matrix-core conv kernels with 8/16/32-wide tiles, pooled
    BN tails, fused upsample, both Linear layers) forward in training mode against the oracle."""
    import siv.models.mymodel as mymodel
    import siv.rng as srng
    torch.manual_seed(11)
    net = mymodel.SoftIntroVAE(*FC_WIDE)
    gen = torch.Generator().manual_seed(12)
    with torch.no_grad():
        for m in net.modules():
            if type(m) == torch.nn.Conv3d:
                torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    x = torch.rand(2, 1, 80, 96, 80, generator=gen)
    cfg = O.FcCfg(*FC_WIDE)
    mu_r, lv_r, z_r, x_r = O.forward(x, sd, cfg, O.Draws(seed=3), True)
    net = net.to("cuda").train()
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=3))):
        mu, lv, z, xr = net.forward(x.cuda())
    check(mu, mu_r, "mu")
    check(lv, lv_r, "logvar")
    check(z, z_r, "z")
    check(xr, x_r, "x_re", tol=2e-4)
    for k, v in net.state_dict().items():
        if k.endswith(("running_mean", "running_var")):
            check(v, sd[k], k, floor=1e-6)


def test_trainer_fc_runs_on_device(tmp_path):
    """siv.utils.trainer_fc end to end on the GPU with the device generator: finite losses, the
    reference's list lengths, log files and checkpoint names."""
    import siv.models.mymodel as mymodel
    import siv.utils.trainer_fc as trainer_fc
    torch.manual_seed(1)
    net = mymodel.SoftIntroVAE(4, 8, 8, 8, 16)
    gen = torch.Generator().manual_seed(2)
    loader = [(torch.rand(2, 1, 80, 96, 80, generator=gen), torch.zeros(2)) for _ in range(2)]
    res = trainer_fc.train_soft_intro_vae(net, loader, loader[:1], 1, 2e-4, torch.device("cuda"),
                                          str(tmp_path) + "/", 1.0, 1024.0, 0.75)
    assert all(len(r) == 2 and np.isfinite(r).all() for r in res)
    assert (tmp_path / "prams" / "S-IntroVAE_4184_epoch0.pth").exists()
    assert (tmp_path / "loss.txt").exists() and (tmp_path / "kl_losses.txt").exists()
    sd = torch.load(tmp_path / "prams" / "S-IntroVAE_4184_epoch0.pth", weights_only=True)
    assert list(sd.keys()) == list(net.state_dict().keys())


def _fc_init(net, gen):
    with torch.no_grad():
        for m in net.modules():
            if type(m) == torch.nn.Conv3d:
                torch.nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))


class _Draws64(O.Draws):
    def _emit(self, shape, make):
        return super()._emit(shape, make).double()


def test_fc_wide_step_backward_matches_oracle():
    """One E+D iteration of the 8-16-32-32 channel FC-latent model (batch 2): losses against the fp32
    oracle, every gradient "as close to the float64 value of the same formulas as the fp32 oracle is"
    (tests/fc_truth.py explains why an element-wise fp32 comparison is not meaningful at batch 2)."""
    import siv.models.mymodel as mymodel
    from tests.fc_truth import assert_grad_as_good_as_reference
    torch.manual_seed(31)
    net = mymodel.SoftIntroVAE(*FC_WIDE)
    gen = torch.Generator().manual_seed(32)
    _fc_init(net, gen)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    real = torch.rand(2, 1, 80, 96, 80, generator=gen)
    noise = torch.randn(2, FC_WIDE[4], generator=gen)
    cfg, hp = O.FcCfg(*FC_WIDE), O.Hyper()
    sd32 = {k: v.clone() for k, v in sd.items()}
    ref = O.soft_intro_step(sd32, cfg, real, noise, O.Draws(seed=33), hp, {}, {})
    sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd.items()}
    truth = O.soft_intro_step(sd64, cfg, real.double(), noise.double(), _Draws64(seed=33), hp, {}, {})
    net = net.to("cuda").train()
    (st, grads), = _run_steps(net, lambda s: (real, noise), hp, O.Draws(seed=33), 1)
    for mine, r, rt in (("lossE", "lossE", 1e-4), ("kl_real", "kl_real", 1e-4), ("lossD", "lossD", 1e-3),
                        ("fake_kl", "fake_kl", 1e-3), ("rec_kl", "rec_kl", 1e-3), ("loss_rec", "loss_rec_D", 1e-3)):
        np.testing.assert_allclose(st[mine], float(ref[r]), rtol=rt, err_msg=mine)
    for k, gr in grads.items():
        which = "gradE" if k.startswith("encoder.") else "gradD"
        rg, tg = ref[which][k], truth[which][k]
        if rg is None:
            assert float(gr.abs().max()) == 0.0, k
        elif float(tg.abs().max()) > 1e-4:                 # skips BN-cancelled convolution biases
            assert_grad_as_good_as_reference(gr.numpy(), rg.numpy(), tg.numpy(), k)


def test_fc_z600_benchmark_model_step():
    """BASELINE configs[1]: mymodel.SoftIntroVAE(16, 32, 64, 128, 600) at 80x96x80, batch 4 -- training-mode
    forward against the oracle (1e-4 / 2e-4), then one full E+D iteration: E-step scalars at 1e-4, D-step
    scalars at 1e-3 (it runs on the encoder after its first, sign-like Adam update), per-parameter gradient
    norms at 2 % (element-wise comparison: test_fc_wide_step_backward_matches_oracle)."""
    import siv.models.mymodel as mymodel
    import siv.rng as srng
    Z600 = (16, 32, 64, 128, 600)
    torch.manual_seed(41)
    net = mymodel.SoftIntroVAE(*Z600)
    gen = torch.Generator().manual_seed(42)
    _fc_init(net, gen)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    real = torch.rand(4, 1, 80, 96, 80, generator=gen)
    noise = torch.randn(4, 600, generator=gen)
    cfg, hp = O.FcCfg(*Z600), O.Hyper()
    sdf = {k: v.clone() for k, v in sd.items()}
    mu_r, lv_r, z_r, x_r = O.forward(real, sdf, cfg, O.Draws(seed=43), True)
    net = net.to("cuda").train()
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=43))):
        mu, lv, z, xr = net.forward(real.cuda())
    check(mu, mu_r, "mu")
    check(lv, lv_r, "logvar")
    check(z, z_r, "z")
    check(xr, x_r, "x_re", tol=2e-4)
    net.load_state_dict(sd)                       # undo the running-statistics update of the forward above
    ref = O.soft_intro_step({k: v.clone() for k, v in sd.items()}, cfg, real, noise, O.Draws(seed=44), hp, {}, {})
    (st, grads), = _run_steps(net, lambda s: (real, noise), hp, O.Draws(seed=44), 1)
    for mine, r, rt in (("lossE", "lossE", 1e-4), ("kl_real", "kl_real", 1e-4), ("lossD", "lossD", 1e-3),
                        ("fake_kl", "fake_kl", 1e-3), ("rec_kl", "rec_kl", 1e-3), ("loss_rec", "loss_rec_D", 1e-3)):
        np.testing.assert_allclose(st[mine], float(ref[r]), rtol=rt, err_msg=mine)
    mods = dict(net.named_modules())
    for k, gr in grads.items():
        rg = ref["gradE" if k.startswith("encoder.") else "gradD"][k]
        # every convolution of this model is biased and, except the last one, followed by a BatchNorm that
        # cancels the gradient of that bias to rounding noise
        cancelled = (k.endswith(".bias") and isinstance(mods[k.rsplit(".", 1)[0]], torch.nn.Conv3d)
                     and not k.startswith("decoder.last_block"))
        if rg is None:
            assert float(gr.abs().max()) == 0.0, k
        elif not cancelled:
            np.testing.assert_allclose(float(gr.norm()), float(rg.norm()), rtol=2e-2, err_msg=k)


def test_graphed_step_matches_eager_step():
    """The iteration captured in a HIP graph and replayed twice against the same two iterations launched
    kernel by kernel (same weights, same inputs, same random stream ids incl. the per-replay epoch): loss
    statistics, weights after each iteration (both Adam step counts live in device memory), BatchNorm
    running statistics.  Also: the capture leaves model, optimiser and generator untouched."""
    import siv.models.models as models
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    dims, Bn = (16, 24, 16), 2
    gen = torch.Generator().manual_seed(51)
    batches = [(torch.rand(Bn, 1, *dims, generator=gen).cuda(), torch.randn(Bn, 1, 4, 6, 4, generator=gen).cuda())
               for _ in range(2)]

    def build():
        torch.manual_seed(50)
        net = models.SoftIntroVAE(*WIDE)
        net.apply(trainer.init_weights_he)
        net = net.to("cuda").train()
        oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
        return net, oe, od, trainer.SoftIntroStep(net, oe, od)

    srng.manual_seed(9, 0)
    net_g, oe_g, od_g, step_g = build()
    sd_before = {k: v.detach().clone() for k, v in net_g.state_dict().items()}
    g = trainer.GraphedSoftIntroStep(step_g, batches[0][0].shape, batches[0][1].shape, "cuda")
    for k, v in net_g.state_dict().items():
        assert torch.equal(v, sd_before[k]), "capture changed " + k
    assert oe_g.step_count == 0 and int(oe_g.dev_step) == 0 and srng.get_rng().calls == 0
    out_g, sd_g = [], []
    for real, noise in batches:
        out_g.append(g(real, noise).clone())
        sd_g.append({k: v.detach().clone() for k, v in net_g.state_dict().items()})
    assert oe_g.step_count == 2 and int(oe_g.dev_step) == 2 and int(srng.get_rng().epoch) == 2
    # eager draws (the per-iteration latent noise) can never meet a draw baked into the graph: the graph's
    # stream ids carry DeviceRng.GRAPH_NS, eager ids stay below it however many iterations have run
    rg = srng.get_rng()
    assert rg.ns == 0 and g.sid_base > 0
    for calls, epoch in ((0, 0), (g.sid_base, 1), (3 * g.sid_base + 5, 3 * g.sid_base + 5), (1 << 19, 1 << 18)):
        rg_calls, rg.calls = rg.calls, calls
        eager_id = rg._sid() + epoch * srng.DeviceRng.EPOCH_STRIDE
        rg.calls = rg_calls
        assert eager_id < srng.DeviceRng.GRAPH_NS
    import siv.backend as sb
    span = 64
    e_eager = sb.get_backend().randn((4, 16), "cuda", 9, g.sid_base + 3, None)
    e_graph = sb.get_backend().randn((4, 16), "cuda", 9, srng.DeviceRng.GRAPH_NS | (g.sid_base + 3), None)
    assert not torch.equal(e_eager, e_graph) and span > 0

    net_e, oe_e, od_e, step_e = build()
    r = srng.DeviceRng(9, 0)
    srng.set_rng(r)
    r.ns = srng.DeviceRng.GRAPH_NS                                       # the graph's stream-id namespace
    for i, (real, noise) in enumerate(batches):
        r.calls = g.sid_base + i * srng.DeviceRng.EPOCH_STRIDE          # the ids the graph uses in replay i
        out = step_e(real, noise)
        np.testing.assert_allclose(out_g[i].cpu().numpy(), out.cpu().numpy(), rtol=1e-6, err_msg=f"stats of step {i}")
        for k, v in net_e.state_dict().items():
            if v.is_floating_point():
                np.testing.assert_allclose(sd_g[i][k].cpu().numpy(), v.detach().cpu().numpy(), rtol=1e-6, atol=1e-7,
                                           err_msg=f"step {i}: {k}")
            else:
                assert int(sd_g[i][k]) == int(v), k


def test_side_stream_weight_gradients_match_serial_backward():
    """Weight gradients that accumulate into the optimiser's flat buffers run on a side stream that is joined
    when backward() ends (siv.ops.OVERLAP_WGRAD).  Two iterations with and without it from the same state:
    every kernel is deterministic and each buffer element is still written by the same launches in the same
    order (all weight gradients share the one side stream), so statistics, gradients and weights are
    bit-identical -- and the gradient buffer is complete the moment backward() returns."""
    import siv.models.models as models
    import siv.ops as ops
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    dims, Bn = (16, 24, 16), 2
    gen = torch.Generator().manual_seed(61)
    batches = [(torch.rand(Bn, 1, *dims, generator=gen).cuda(), torch.randn(Bn, 1, 4, 6, 4, generator=gen).cuda())
               for _ in range(2)]
    results = []
    saved = ops.OVERLAP_WGRAD
    try:
        for overlap in (True, False):
            ops.OVERLAP_WGRAD = overlap
            torch.manual_seed(60)
            net = models.SoftIntroVAE(*WIDE)
            net.apply(trainer.init_weights_he)
            net = net.to("cuda").train()
            oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
            step = trainer.SoftIntroStep(net, oe, od)
            srng.set_rng(srng.DeviceRng(13, 0))
            grads = []
            e_step, d_step = oe.step, od.step
            oe.step = lambda: (grads.append(oe.flat_g.clone()), e_step())      # right after lossE.backward()
            od.step = lambda: (grads.append(od.flat_g.clone()), d_step())
            stats = [step(real, noise).clone() for real, noise in batches]
            results.append((stats, grads, oe.flat_p.clone(), od.flat_p.clone()))
    finally:
        ops.OVERLAP_WGRAD = saved
    (s1, g1, pe1, pd1), (s0, g0, pe0, pd0) = results
    assert len(g1) == len(g0) == 4
    for a, b in zip(s1 + g1 + [pe1, pd1], s0 + g0 + [pe0, pd0]):
        assert torch.equal(a, b)
    assert float(g1[0].abs().max()) > 0 and float(g1[1].abs().max()) > 0


def test_data_parallel_side_stream_path_matches_serial_path(monkeypatch):
    """The multi-GPU path of the iteration on ONE GPU (a one-rank RCCL job forced through the collective code:
    SIV_FORCE_COLLECTIVE=1): encoder exchange + Adam on the side stream beside the D-step's first two decoder
    passes, statistics all-reduce + decoder exchange + Adam on the side stream at the end of the iteration and
    joined by the NEXT iteration behind its encoder pass (which is issued first, with the random call indices
    of the reference order).  Three iterations against the same three with SIV_DP_OVERLAP=0 (every exchange
    and update in order on the main stream): statistics, weights, Adam moments and BatchNorm buffers are
    bit-identical."""
    import socket
    import torch.distributed as dist
    import siv.models.models as models
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    from siv.parallel import GradSync
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    for k, v in dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                     SIV_FORCE_COLLECTIVE="1").items():
        monkeypatch.setenv(k, v)
    dims, Bn = (16, 24, 16), 2
    gen = torch.Generator().manual_seed(61)
    batches = [(torch.rand(Bn, 1, *dims, generator=gen).cuda(), torch.randn(Bn, 1, 4, 6, 4, generator=gen).cuda())
               for _ in range(3)]
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        def run(overlap):
            monkeypatch.setenv("SIV_DP_OVERLAP", "1" if overlap else "0")
            torch.manual_seed(60)
            net = models.SoftIntroVAE(*WIDE)
            net.apply(trainer.init_weights_he)
            net = net.to("cuda").train()
            sync = GradSync(torch.device("cuda", 0))
            assert sync.collective and sync.side is not None and sync.overlap == overlap
            sync.broadcast_(net)
            oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
            step = trainer.SoftIntroStep(net, oe, od, sync=sync)
            srng.manual_seed(11, 0)
            outs = []
            for i, (real, noise) in enumerate(batches):
                outs.append(step(real, noise).clone())
                assert sync.pending == overlap        # the decoder update is still in flight / already done
            step.flush()
            torch.cuda.synchronize()
            return outs, {k: v.detach().clone() for k, v in net.state_dict().items()}, \
                (oe.flat_p.clone(), od.flat_p.clone(), oe.m.clone(), od.m.clone(), oe.v.clone(), od.v.clone()), \
                srng.get_rng().calls
        o1, sd1, opt1, calls1 = run(True)
        o0, sd0, opt0, calls0 = run(False)
    finally:
        dist.destroy_process_group()
    assert calls1 == calls0
    for a, b in zip(o1, o0):
        assert torch.equal(a, b), (a, b)
    for k in sd0:
        assert torch.equal(sd1[k], sd0[k]), k
    for a, b in zip(opt1, opt0):
        assert torch.equal(a, b)
