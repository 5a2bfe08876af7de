"""Host-logic tests (CPU, no GPU): the product's model / autograd wiring / trainer /
flat optimiser driven through a TEST-ONLY CPU backend (tests/cpu_backend.py) and compared
with fixtures produced by the imported reference.  Kernel numerics are NOT tested here
(see tests/test_gpu_*.py); these tests pin everything above the C ABI."""
import os
import tempfile

import numpy as np
import pytest
import torch

from oracle import siv_oracle as O
from tests.cpu_backend import CpuCheckerBackend

import siv.backend as sb
import siv.rng as srng
import siv.models.models as models
import siv.models.vaemodel as vaemodel
import siv.models.lossf as lossf
import siv.utils.my_trainer as trainer
from siv.optim import FlatAdam

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Parameters whose gradient is analytically zero because the following BatchNorm removes
# their effect (a bias, or the scale of a 1-input-channel 1x1x1 conv): what any
# implementation computes for them is cancellation noise of size ~1e-6.
BN_CANCELLED = ("encoder.blocks.0.0.bias", "decoder.blocks.0.0.bias", "decoder.blocks.0.0.weight")


# End-to-end gradients pass through ~10^6 LeakyReLU/ReLU kinks.  Two fp32 implementations
# that differ only in rounding can put one pre-activation on opposite sides of zero; that
# single flip changes every upstream gradient at the 1e-4..1e-3 level (measured: one flip
# among 73k activations of the last decoder BN moved the decoder gradients of fixture g4's
# first D-step by up to 3.4e-4 of their maximum).  The kernels themselves are pinned at
# 1e-4 by the per-operator tests; the end-to-end comparisons therefore use 1e-3 of the
# tensor's maximum for gradients, and tight tolerances for everything that is continuous
# (forward outputs, losses, statistics).
E2E_GRAD_RTOL_OF_MAX = 1e-3


def grad_atol(key, ref, step=0):
    if key.endswith(BN_CANCELLED):
        return 2e-5
    # the second iteration starts from weights that differ by Adam's sign-of-noise updates
    return E2E_GRAD_RTOL_OF_MAX * (1 if step == 0 else 5) * float(np.abs(ref).max()) + 3e-7


def check_params_after_adam(mine, ref, grad_ref, lr_steps, what):
    """Adam's first steps move every element by ~lr*sign(g): where |g| is far above rounding
    noise the update must agree tightly; elsewhere the sign of a noise-level gradient decides
    and only |delta| <= 2*lr per step can be asserted."""
    mine, ref = np.asarray(mine), np.asarray(ref)
    if grad_ref is None:
        np.testing.assert_allclose(mine, ref, rtol=1e-5, atol=2e-6, err_msg=what)
        return
    solid = np.abs(grad_ref) > 2e-2 * np.abs(grad_ref).max()
    np.testing.assert_allclose(mine[solid], ref[solid], rtol=1e-5, atol=5e-6, err_msg=what + " (solid)")
    np.testing.assert_allclose(mine, ref, rtol=1e-5, atol=2 * lr_steps + 2e-6, err_msg=what)


@pytest.fixture(autouse=True)
def cpu_backend():
    sb.set_backend(CpuCheckerBackend())
    yield
    sb.set_backend(None)
    srng.set_rng(srng.DeviceRng(0))


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sd_from(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}


def close(a, b, rtol=1e-5, atol=1e-6, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol, err_msg=what)


def test_state_dict_layout_matches_reference():
    g = load("g3_model.npz")
    net = models.SoftIntroVAE(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    ref = sd_from(g, "sd0.")
    mine = net.state_dict()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    net.load_state_dict(ref, strict=True)


def test_model_forward_matches_reference():
    g = load("g3_model.npz")
    net = models.SoftIntroVAE(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    net.load_state_dict(sd_from(g, "sd0."))
    x, z_in = torch.from_numpy(g["x"]), torch.from_numpy(g["z_in"])
    net.train()
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))) as r:
        mu, lv = net.encode(x)
        xd = net.decode(z_in)
        fmu, flv, fz, fx = net.forward(x)
        assert r.source.i == int(g["n_draws"])
    for a, k in ((mu, "enc.mu"), (lv, "enc.logvar"), (xd, "dec.x"), (fmu, "fwd.mu"),
                 (flv, "fwd.logvar"), (fz, "fwd.z"), (fx, "fwd.x")):
        close(a, g[k], rtol=1e-4, atol=1e-5, what=k)
    ref1 = sd_from(g, "sd1.")
    for k, v in net.state_dict().items():
        close(v, ref1[k], rtol=1e-5, atol=1e-6, what=k)
    net.eval()
    with torch.no_grad():
        mu, lv = net.encode(x)
        z = net.reparameterize(mu, lv, True)
        xe = net.decode(z)
    for a, k in ((mu, "eval.mu"), (lv, "eval.logvar"), (z, "eval.z"), (xe, "eval.x")):
        close(a, g[k], rtol=1e-4, atol=1e-5, what=k)


def test_soft_intro_step_matches_reference():
    """Losses, every gradient, post-Adam parameters and BN running statistics of two
    consecutive E+D iterations (fixture g4, produced by the reference)."""
    g = load("g4_step.npz")
    net = models.SoftIntroVAE(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    net.load_state_dict(sd_from(g, "sd0."))
    net.train()
    opt_e = FlatAdam(net.encoder.parameters(), lr=2e-4)
    opt_d = FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, float(g["hp.beta_rec"]), float(g["hp.beta_neg"]),
                                 float(g["hp.beta_kl"]), float(g["hp.gamma_r"]), float(g["hp.scale"]))
    names = dict(net.named_parameters())
    grads = {}

    def snap(which, module, prefix):
        grads[which] = {prefix + k: p.grad.detach().clone() for k, p in module.named_parameters()}

    orig_e, orig_d = opt_e.step, opt_d.step
    opt_e.step = lambda: (snap("gradE", net.encoder, "encoder."), orig_e())
    opt_d.step = lambda: (snap("gradD", net.decoder, "decoder."), orig_d())
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))) as r:
        for s in range(2):
            real = torch.from_numpy(g[f"s{s}.real"])
            noise = torch.from_numpy(g[f"s{s}.noise"])
            st = dict(zip(trainer.STAT_NAMES, step(real, noise).tolist()))
            # the second iteration starts from weights that already differ by Adam's
            # sign-of-noise updates (<= lr per element): scalars then agree to ~1e-4..1e-3
            srt = 1e-4 if s == 0 else 1e-3
            for mine, ref in (("lossE", "lossE"), ("lossD", "lossD"), ("kl_real", "kl_real"),
                              ("fake_kl", "fake_kl"), ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
                np.testing.assert_allclose(st[mine], float(g[f"s{s}.{ref}"]), rtol=srt, err_msg=mine)
            for mine in ("exp_elbo_rec", "exp_elbo_fake"):
                # exp(-2s(beta_rec*mse + beta_neg*kl)) ~ e^-55: compare the exponent at 1e-4
                a, b = st[mine], float(g[f"s{s}.{mine}"])
                if b > 1e-44:
                    np.testing.assert_allclose(np.log(max(a, 1e-45)), np.log(b), rtol=srt, err_msg=mine)
                else:
                    assert a < 1e-40, mine
            if s == 1:
                # Gradients of the second iteration are not comparable element-wise: in the
                # reference's own arithmetic (the oracle), moving 0.2 % of the weights by the
                # +-2*lr that Adam's sign-of-noise updates produce changes these gradients by
                # 20-50 % of their maximum (tiny batch, exp(logvar) in the KL path).  The
                # losses above and the optimiser bookkeeping below are what can be pinned.
                assert opt_e.step_count == 2 and opt_d.step_count == 2
                for k, v in net.state_dict().items():
                    if k.endswith("num_batches_tracked"):
                        assert int(v) == int(g[f"s1.sd.{k}"]), k
                continue
            for which in ("gradE", "gradD"):
                for k, gr in grads[which].items():
                    key = f"s{s}.{which}.{k}"
                    if key + ".none" in g.files:     # unused parameter: reference grad is None
                        assert float(gr.abs().max()) == 0.0, key
                        continue
                    ref = g[key]
                    close(gr, ref, rtol=1e-4, atol=grad_atol(key, ref, s), what=key)
            ref_sd = sd_from(g, f"s{s}.sd.")
            for k, v in net.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    assert int(v) == int(ref_sd[k]), k
                    continue
                if k.endswith(("running_mean", "running_var")):
                    close(v, ref_sd[k], rtol=1e-4, atol=1e-4, what=k)     # includes BN-cancelled biases
                    continue
                gk = [q for q in (f"s{s}.{w}.{k}" for w in ("gradE", "gradD")) if q in g.files]
                gref = None if (not gk or k.endswith(BN_CANCELLED)) else g[gk[0]]
                if k.endswith(BN_CANCELLED):
                    close(v, ref_sd[k], rtol=1e-5, atol=2 * (s + 1) * 2e-4 + 2e-6, what=k)
                else:
                    check_params_after_adam(v.detach().numpy(), ref_sd[k].numpy(), gref, (s + 1) * 2e-4, k)
        assert r.source.i == int(g["n_draws"])
    assert set(names) == set(k for k in ref_sd if not k.endswith(("running_mean", "running_var",
                                                                   "num_batches_tracked")))


def test_plain_vae_step_matches_reference():
    g = load("g6_vae.npz")
    net = vaemodel.ResNetVAE(4, [[4, 1, 2], [6, 1, 2], [8, 2, 2], [12, 2, 2]])
    ref0 = sd_from(g, "sd0.")
    assert list(net.state_dict().keys()) == list(ref0.keys())
    net.load_state_dict(ref0)
    net.train()
    opt = FlatAdam(net.parameters(), 1e-3)
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))):
        for s in range(2):
            x = torch.from_numpy(g[f"s{s}.x"])
            opt.zero_grad()
            x_re, mu, logvar = net.forward(x)
            loss, mse, kl = lossf.normal_loss(x_re, mu, logvar, x, 1, 20)
            loss.backward()
            srt = 1e-4 if s == 0 else 1e-3      # see test_soft_intro_step_matches_reference
            close(torch.stack([loss, mse, kl]), g[f"s{s}.loss"], rtol=srt)
            if s == 0:
                close(mu, g[f"s{s}.mu"], rtol=1e-4, atol=1e-5)
                close(x_re[:, :, 16], g[f"s{s}.x_re_slice"], rtol=1e-4, atol=1e-5)
                for k, p in net.named_parameters():
                    key = f"s{s}.grad.{k}"
                    if key + ".none" in g.files:
                        assert float(p.grad.abs().max()) == 0.0
                    elif not k.endswith(BN_CANCELLED):
                        ref = g[key]
                        close(p.grad, ref, rtol=1e-4, atol=grad_atol(key, ref), what=key)
            opt.step()
            if s == 0:
                ref_sd = sd_from(g, f"s{s}.sd.")
                for k, v in net.state_dict().items():
                    if k.endswith("num_batches_tracked"):
                        assert int(v) == int(ref_sd[k]), k
                    elif k.endswith(("running_mean", "running_var")):
                        close(v, ref_sd[k], rtol=1e-4, atol=1e-4, what=k)
                    elif k.endswith(BN_CANCELLED):
                        close(v, ref_sd[k], rtol=1e-5, atol=2 * 1e-3 + 2e-6, what=k)
                    else:
                        gk = f"s{s}.grad.{k}"
                        check_params_after_adam(v.detach().numpy(), ref_sd[k].numpy(),
                                                g[gk] if gk in g.files else None, 1e-3, k)
    assert opt.step_count == 2


def test_loss_helpers_match_reference():
    g = load("g5_losses.npz")
    mu, lv = torch.from_numpy(g["mu"]), torch.from_numpy(g["logvar"])
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for r in ("mean", "sum", "none"):
        close(trainer.calc_kl(lv, mu, reduce=r), g["kl." + r], what=r)
    for r in ("mean", "None", "none"):
        close(trainer.calc_reconstruction_loss(x, y, loss_type="mse", reduction=r), g["rec." + r], what=r)
    loss, mse, kld = lossf.normal_loss(y, mu, lv, x, 1, 10)
    close(torch.stack([loss, mse, kld]), g["normal_loss"])
    close(lossf.kld_loss(mu, lv), g["lossf.kld"])
    close(lossf.mse_loss(y, x), g["lossf.mse"])


class ListLoader:
    def __init__(self, batches):
        self.batches = batches

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


def test_train_soft_intro_vae_matches_reference_trainer():
    """The product trainer against the REAL reference ``train_soft_intro_vae`` (fixture g8:
    2 epochs x 2 batches + validation, B=8, 80x96x80, narrow model): returned lists, the
    four text logs, initial and final weights."""
    g = load("g8_train.npz")
    cfg = (int(g["cfg_in_ch"]), g["cfg_blocks"].tolist())
    torch.manual_seed(82)
    net = models.SoftIntroVAE(*cfg)
    gen = torch.Generator().manual_seed(int(g["data_seed"]))
    train = [(torch.rand(8, 1, 80, 96, 80, generator=gen), torch.zeros(8)) for _ in range(2)]
    val = [(torch.rand(8, 1, 80, 96, 80, generator=gen), torch.zeros(8))]
    captured = {}
    orig_init = FlatAdam.__init__

    def capture_then_init(self, params, *a, **k):
        if "sd" not in captured:
            captured["sd"] = {kk: v.detach().clone() for kk, v in net.state_dict().items()}
        orig_init(self, params, *a, **k)

    FlatAdam.__init__ = capture_then_init
    try:
        with tempfile.TemporaryDirectory() as d, \
                srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))) as r:
            res = trainer.train_soft_intro_vae(net, ListLoader(train), ListLoader(val), 2, 2e-4,
                                               torch.device("cpu"), d + "/", 1.0, 1024.0, 0.75)
            assert r.source.i == int(g["n_draws"])
            logs = {n: open(os.path.join(d, n)).read() for n in
                    ("loss.txt", "kl_losses.txt", "train_losses.txt", "val_losses.txt")}
            assert os.path.exists(os.path.join(d, "prams", "S-IntroVAE_3898_epoch1.pth"))
    finally:
        FlatAdam.__init__ = orig_init
    ref0 = sd_from(g, "sd0.")
    for k, v in captured["sd"].items():            # same construction order + same init as the reference
        close(v, ref0[k], rtol=0, atol=0, what="init " + k)
    for name, lst in zip(("train_lossE", "train_lossD", "val_lossE", "val_lossD"), res):
        np.testing.assert_allclose(np.array(lst), g["ret." + name], rtol=2e-4, err_msg=name)
    for n, txt in logs.items():
        ref_lines = str(g["txt." + n]).strip().split("\n")
        my_lines = txt.strip().split("\n")
        assert len(ref_lines) == len(my_lines), n
        for a, b in zip(my_lines, ref_lines):
            ka, va = a.rsplit("=", 1)
            kb, vb = b.rsplit("=", 1)
            assert ka == kb, (n, a, b)
            np.testing.assert_allclose(float(va), float(vb), rtol=5e-3, err_msg=f"{n}: {a} vs {b}")
    ref_final = sd_from(g, "sd_final.")
    for k, v in net.state_dict().items():
        if k.endswith("num_batches_tracked"):
            assert int(v) == int(ref_final[k]), k
        else:
            close(v, ref_final[k], rtol=2e-3, atol=2e-3, what="final " + k)


def test_bf16_activation_storage_hooks():
    """siv.activations on the CPU checker: with storage("bf16") the large fp32 tensors saved inside the
    encoder / decoder passes are held as bf16; parameters, masks and small tensors are not; the forward
    pass is unchanged and the gradients move by no more than bf16 rounding of the saved tensors explains (5e-2 of max at 8 channels:
    besides the 2^-9 rounding itself, a pre-activation within 2^-9 of zero can change sides in the recomputed
    LeakyReLU mask)."""
    import siv.activations as act
    cfg = (8, [[8, 1, 2], [16, 1, 2]])
    dims, bn = (16, 32, 32), 2                     # 8 ch x 16384 voxels x 2: above the size threshold
    hp = O.Hyper(dims=dims)
    res = {}
    for kind in ("fp32", "bf16"):
        torch.manual_seed(3)
        net = models.SoftIntroVAE(*cfg)
        net.apply(trainer.init_weights_he)
        net.train()
        gen = torch.Generator().manual_seed(4)
        real = torch.rand(bn, 1, *dims, generator=gen)
        noise = torch.randn(*net.latent_shape(bn, dims), generator=gen)
        oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
        step = trainer.SoftIntroStep(net, oe, od, hp.beta_rec, hp.beta_neg, hp.beta_kl, hp.gamma_r, hp.scale)
        grads = {}
        e_step = oe.step
        oe.step = lambda: (grads.update({k: p.grad.detach().clone() for k, p in net.encoder.named_parameters()}),
                           e_step())
        before = act.packed["tensors"]
        with act.storage(kind), srng.use_rng(srng.ReplayRng(O.Draws(seed=5))):
            st = dict(zip(trainer.STAT_NAMES, step(real, noise).tolist()))
        res[kind] = (st, grads, act.packed["tensors"] - before)
    assert res["fp32"][2] == 0 and res["bf16"][2] > 10
    assert res["bf16"][0]["lossE"] == res["fp32"][0]["lossE"]
    moved = 0.0
    for k, g in res["fp32"][1].items():
        if ("encoder." + k).endswith(BN_CANCELLED):
            continue
        err = float((res["bf16"][1][k] - g).abs().max() / (g.abs().max() + 1e-30))
        assert err <= 5e-2, (k, err)
        moved = max(moved, err)
    assert moved > 0.0
    with pytest.raises(ValueError):
        act.set_storage("fp16")
