"""Pins the CPU oracle (oracle/siv_oracle.py) against fixtures produced by the
imported reference (tools/make_goldens.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import siv_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL, ATOL = 1e-5, 1e-6      # same library, same machine class: only op-ordering noise is expected


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def sd_from(g, prefix):
    return {k[len(prefix):]: torch.from_numpy(g[k].copy()) for k in g.files if k.startswith(prefix)}


def close(a, b, rtol=RTOL, atol=ATOL, what=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, np.asarray(b), rtol=rtol, atol=atol, err_msg=what)


@pytest.mark.parametrize("name", ["bb_s1", "bb_s2", "bb_s1p", "ub_s1", "ub_s2", "vbb_s2", "vub_s2"])
def test_g2_blocks(name):
    g = load("g2_blocks.npz")
    cin, cout, s = (int(v) for v in g[name + ".meta"])
    sd = sd_from(g, name + ".sd0.")
    x = torch.from_numpy(g[name + ".x"])
    slope = 0.0 if name.startswith("v") else 0.2
    fn = O.upsample_block if "ub" in name else O.building_block
    y = fn(x, sd, "", cin, cout, s, slope, True)
    close(y, g[name + ".y"], what="train output")
    sd1 = sd_from(g, name + ".sd1.")
    for k in sd1:
        close(sd[k], sd1[k], what=k)
    y_eval = fn(x, sd, "", cin, cout, s, slope, False)
    close(y_eval, g[name + ".y_eval"], what="eval output")


def test_g3_model():
    g = load("g3_model.npz")
    cfg = O.Cfg(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    sd = sd_from(g, "sd0.")
    x, z_in = torch.from_numpy(g["x"]), torch.from_numpy(g["z_in"])
    draws = O.Draws(seed=int(g["draw_seed"]))
    mu, lv = O.encode(x, sd, cfg, draws, True)
    xd = O.decode(z_in, sd, cfg, draws, True)
    fmu, flv, fz, fx = O.forward(x, sd, cfg, draws, True)
    assert draws.i == int(g["n_draws"])
    for a, k in ((mu, "enc.mu"), (lv, "enc.logvar"), (xd, "dec.x"), (fmu, "fwd.mu"),
                 (flv, "fwd.logvar"), (fz, "fwd.z"), (fx, "fwd.x")):
        close(a, g[k], what=k)
    sd1 = sd_from(g, "sd1.")
    for k in sd1:
        close(sd[k], sd1[k], what=k)
    mu, lv = O.encode(x, sd, cfg, None, False)
    z = O.reparameterize(mu, lv, None, True)
    xe = O.decode(z, sd, cfg, None, False)
    for a, k in ((mu, "eval.mu"), (lv, "eval.logvar"), (z, "eval.z"), (xe, "eval.x")):
        close(a, g[k], what=k)


def test_g4_step():
    g = load("g4_step.npz")
    cfg = O.Cfg(4, [[4, 1, 2], [8, 1, 2], [8, 2, 2]])
    sd = sd_from(g, "sd0.")
    hp = O.Hyper(beta_rec=float(g["hp.beta_rec"]), beta_neg=float(g["hp.beta_neg"]),
                 beta_kl=float(g["hp.beta_kl"]), gamma_r=float(g["hp.gamma_r"]),
                 scale=float(g["hp.scale"]))
    draws = O.Draws(seed=int(g["draw_seed"]))
    ae, ad = {}, {}
    for step in range(2):
        real = torch.from_numpy(g[f"s{step}.real"])
        noise = torch.from_numpy(g[f"s{step}.noise"])
        r = O.soft_intro_step(sd, cfg, real, noise, draws, hp, ae, ad)
        for k in ("lossE", "lossD", "loss_rec_E", "loss_rec_D", "kl_real", "rec_kl", "fake_kl",
                  "fake_kl_e", "rec_kl_e"):
            close(r[k], g[f"s{step}.{k}"], rtol=2e-5, what=f"s{step}.{k}")
        for k in ("exp_elbo_fake", "exp_elbo_rec"):
            close(r[k], g[f"s{step}.{k}"], rtol=1e-3, atol=1e-30, what=k)
        for which in ("gradE", "gradD"):
            for pk, gr in r[which].items():
                key = f"s{step}.{which}.{pk}"
                if gr is None:
                    assert key + ".none" in g.files, key
                else:
                    ref = g[key]
                    # gradients that are analytically ~0 (a conv scale/bias that the
                    # following BatchNorm cancels) are rounding noise: absolute floor
                    tol = 2e-5 * float(np.abs(ref).max()) + 3e-7
                    close(gr, ref, rtol=1e-4, atol=tol, what=key)
        ref_sd = sd_from(g, f"s{step}.sd.")
        for k in ref_sd:
            # Adam normalises the step by |g|: a parameter whose gradient is pure rounding
            # noise still moves by up to lr per step, in a noise-determined direction
            gk = [f"s{step}.{w}.{k}" for w in ("gradE", "gradD")]
            noise = any(q in g.files and float(np.abs(g[q]).max()) < 1e-5 for q in gk)
            close(sd[k], ref_sd[k], rtol=1e-5, atol=(2 * (step + 1) * hp.lr if noise else 2e-6),
                  what=f"s{step}.sd.{k}")
    assert draws.i == int(g["n_draws"])


def test_g5_losses():
    g = load("g5_losses.npz")
    mu, lv = torch.from_numpy(g["mu"]), torch.from_numpy(g["logvar"])
    x, y = torch.from_numpy(g["x"]), torch.from_numpy(g["y"])
    for r in ("mean", "sum", "none"):
        close(O.calc_kl(lv, mu, r), g["kl." + r], what=r)
    for r in ("mean", "None", "none"):
        close(O.calc_reconstruction_loss(x, y, r), g["rec." + r], what=r)
    loss, mse, kld = O.normal_loss(y, mu, lv, x, 1, 10)
    close(torch.stack([loss, mse, kld]), g["normal_loss"])
    close(O.calc_kl(lv, mu, "mean"), g["lossf.kld"])
    close(O.calc_reconstruction_loss(x, y, "mean"), g["lossf.mse"])


def test_g6_vae():
    g = load("g6_vae.npz")
    cfg = O.Cfg(4, [[4, 1, 2], [6, 1, 2], [8, 2, 2], [12, 2, 2]], variant="vae")
    sd = sd_from(g, "sd0.")
    draws = O.Draws(seed=int(g["draw_seed"]))
    adam = {}
    for step in range(2):
        x = torch.from_numpy(g[f"s{step}.x"])
        r = O.vae_step(sd, cfg, x, draws, 1, 20, 1e-3, adam)
        close(torch.stack([r["loss"], r["mse"], r["kld"]]), g[f"s{step}.loss"], rtol=2e-5)
        close(r["mu"], g[f"s{step}.mu"], rtol=1e-4, atol=1e-5)
        close(r["x_re"][:, :, 16], g[f"s{step}.x_re_slice"], rtol=1e-4, atol=1e-5)
        for pk, gr in r["grad"].items():
            key = f"s{step}.grad.{pk}"
            if gr is None:
                assert key + ".none" in g.files, key
            else:
                ref = g[key]
                close(gr, ref, rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()) + 3e-7, what=key)
        ref_sd = sd_from(g, f"s{step}.sd.")
        for k in ref_sd:
            close(sd[k], ref_sd[k], rtol=1e-5, atol=5e-6, what=f"s{step}.sd.{k}")


def fc_inputs(g):
    """The inputs tools/make_goldens.py::g7_fc drew, regenerated from the stored seed."""
    cfg = tuple(int(v) for v in g["cfg"])
    B = int(g["batch"])
    gen = torch.Generator().manual_seed(int(g["data_seed"]))
    x = torch.rand(B, 1, 80, 96, 80, generator=gen)
    zz = torch.randn(B, cfg[4], generator=gen)
    steps = [(torch.rand(B, 1, 80, 96, 80, generator=gen), torch.randn(B, cfg[4], generator=gen))
             for _ in range(2)]
    return cfg, x, zz, steps


def test_g7_fc_model_and_step():
    """models/mymodel.py (FC-latent) forward in train and eval mode, then two E+D iterations."""
    g = load("g7_fc.npz")
    cfgt, x, zz, steps = fc_inputs(g)
    cfg = O.FcCfg(*cfgt)
    sd = sd_from(g, "sd0.")
    draws = O.Draws(seed=int(g["draw_seed"]))
    fmu, flv, fz, fx = O.forward(x, sd, cfg, draws, True)
    xd = O.decode(zz, sd, cfg, draws, True)
    for a, k in ((fmu, "fwd.mu"), (flv, "fwd.logvar"), (fz, "fwd.z"), (fx[:, :, 40], "fwd.x_slice"),
                 (xd[:, :, 40], "dec.x_slice")):
        close(a, g[k], rtol=1e-4, atol=1e-5, what=k)
    close(fx.double().sum(), g["fwd.x_sum"], rtol=1e-5)
    close(xd.double().sum(), g["dec.x_sum"], rtol=1e-5)
    sd1 = sd_from(g, "sd1.")
    for k in sd1:
        close(sd[k], sd1[k], what=k)
    emu, elv = O.encode(x, sd, cfg, None, False)
    ex = O.decode(zz, sd, cfg, None, False)
    for a, k in ((emu, "eval.mu"), (elv, "eval.logvar"), (ex[:, :, 40], "eval.x_slice")):
        close(a, g[k], rtol=1e-4, atol=1e-5, what=k)
    hp = O.Hyper(beta_rec=float(g["hp.beta_rec"]), beta_neg=float(g["hp.beta_neg"]),
                 beta_kl=float(g["hp.beta_kl"]), gamma_r=float(g["hp.gamma_r"]), scale=float(g["hp.scale"]))
    ae, ad = {}, {}
    for step, (real, noise) in enumerate(steps):
        r = O.soft_intro_step(sd, cfg, real, noise, draws, hp, ae, ad)
        for k in ("lossE", "lossD", "loss_rec_E", "loss_rec_D", "kl_real", "rec_kl", "fake_kl",
                  "fake_kl_e", "rec_kl_e"):
            close(r[k], g[f"s{step}.{k}"], rtol=2e-5, what=f"s{step}.{k}")
        for k in ("exp_elbo_fake", "exp_elbo_rec"):
            close(r[k], g[f"s{step}.{k}"], rtol=1e-3, atol=1e-30, what=k)
        for which in ("gradE", "gradD"):
            for pk, gr in r[which].items():
                key = f"s{step}.{which}.{pk}"
                if gr is None:
                    assert key + ".none" in g.files, key      # encoder.block8.* is never called
                else:
                    ref = g[key]
                    close(gr, ref, rtol=1e-4, atol=2e-5 * float(np.abs(ref).max()) + 3e-7, what=key)
        ref_sd = sd_from(g, f"s{step}.sd.")
        for k in ref_sd:
            # every conv bias here is cancelled by the BatchNorm that follows: its gradient is
            # rounding noise that Adam turns into a +-lr move
            gk = [f"s{step}.{w}.{k}" for w in ("gradE", "gradD")]
            noise_g = any(q in g.files and float(np.abs(g[q]).max()) < 1e-5 for q in gk)
            close(sd[k], ref_sd[k], rtol=1e-5,
                  atol=(2 * (step + 1) * hp.lr if noise_g else 2e-6), what=f"s{step}.sd.{k}")
    assert draws.i == int(g["n_draws"])


@pytest.mark.parametrize("name", ["brain_f32", "signed_f32", "f64", "i16"])
def test_g10_preprocess(name):
    """utils/data_load.py BrainDataset._preprocess (fixture from the imported reference)."""
    g = load("g10_preprocess.npz")
    out = O.preprocess(g["in." + name])
    assert out.dtype == np.float32 and out.shape == g["out." + name].shape
    np.testing.assert_array_equal(out, g["out." + name])


def test_flip_shift_definition():
    v = np.arange(2 * 3 * 4, dtype=np.float32).reshape(2, 3, 4)
    np.testing.assert_array_equal(O.flip_shift(v, (0, 0, 0), (0, 0, 0)), v)
    np.testing.assert_array_equal(O.flip_shift(v, (0, 0, 1), (0, 0, 0)), v[:, :, ::-1])
    s = O.flip_shift(v, (0, 0, 0), (0, 1, -2))
    assert (s[:, 0] == 0).all() and (s[:, :, 2:] == 0).all()
    np.testing.assert_array_equal(s[:, 1:, :2], v[:, :2, 2:])
    assert (O.flip_shift(v, (1, 1, 1), (2, 0, 0)) == 0).all()


def test_affine_resample_matches_scipy():
    """The trilinear resampling definition against an independent implementation
    (scipy.ndimage.affine_transform, order 1, grid-constant padding)."""
    from scipy import ndimage
    import siv.utils.data_load as DL
    rs = np.random.RandomState(11)
    v = rs.rand(9, 12, 10)
    np.testing.assert_allclose(O.affine_resample(v, np.eye(3, 4)), v, atol=1e-12)      # identity
    for scales, degrees, pad in (((1.0, 1.0, 1.0), (10.0, 0.0, 0.0), 0.0),
                                 ((0.9, 1.1, 1.05), (-7.0, 4.0, 9.5), -2.5),
                                 ((1.1, 1.1, 0.9), (3.0, -10.0, 1.0), "minimum")):
        a = DL.affine_matrix(scales, degrees, v.shape)
        cval = v.min() if isinstance(pad, str) else pad
        ref = ndimage.affine_transform(v, a[:, :3], offset=a[:, 3], order=1, mode="grid-constant", cval=cval)
        np.testing.assert_allclose(O.affine_resample(v, a, pad), ref, atol=1e-10)
    # an integer shift reproduces the flip/shift definition (zero fill)
    sh = np.concatenate([np.eye(3), -np.array([[2.0], [0.0], [-1.0]])], 1)
    np.testing.assert_allclose(O.affine_resample(v, sh, 0.0), O.flip_shift(v, (0, 0, 0), (2, 0, -1)), atol=1e-12)
