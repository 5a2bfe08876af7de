"""Host-logic tests (CPU, no GPU) of the FC-latent path: siv.models.mymodel and
siv.utils.trainer_fc driven through the TEST-ONLY CPU backend and compared with fixtures
produced by the imported reference (models/mymodel.py, utils/trainer_fc.py)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import siv_oracle as O
from tests.cpu_backend import CpuCheckerBackend
from tests.test_host_logic import (ListLoader, check_params_after_adam, close, load, sd_from)
from tests.test_oracle_golden import fc_inputs
from tests.fc_truth import assert_grad_as_good_as_reference, fc_step0_fp64

import siv.backend as sb
import siv.rng as srng
import siv.models.mymodel as mymodel
import siv.utils.my_trainer as trainer
import siv.utils.trainer_fc as trainer_fc
from siv.optim import FlatAdam


@pytest.fixture(autouse=True)
def cpu_backend():
    sb.set_backend(CpuCheckerBackend())
    yield
    sb.set_backend(None)
    srng.set_rng(srng.DeviceRng(0))


def bn_cancelled_biases(net):
    """Biases of convolutions that feed a BatchNorm: their gradient is analytically zero."""
    out = set()
    for name, seq in net.named_modules():
        if isinstance(seq, nn.Sequential):
            mods = list(seq)
            for i, m in enumerate(mods[:-1]):
                if isinstance(m, nn.Conv3d) and isinstance(mods[i + 1], nn.BatchNorm3d):
                    out.add(f"{name}.{i}.bias")
    return out


def test_fc_state_dict_layout_matches_reference():
    g = load("g7_fc.npz")
    net = mymodel.SoftIntroVAE(*(int(v) for v in g["cfg"]))
    ref = sd_from(g, "sd0.")
    mine = net.state_dict()
    assert list(mine.keys()) == list(ref.keys())
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    net.load_state_dict(ref, strict=True)
    assert net.z_ch == int(g["cfg"][4])
    plain = mymodel.ResNetVAE(*(int(v) for v in g["cfg"]))
    assert list(plain.state_dict().keys()) == list(ref.keys())


def test_fc_forward_and_step_match_reference():
    g = load("g7_fc.npz")
    cfgt, x, zz, steps = fc_inputs(g)
    net = mymodel.SoftIntroVAE(*cfgt)
    net.load_state_dict(sd_from(g, "sd0."))
    net.train()
    cancelled = bn_cancelled_biases(net)
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=int(g["draw_seed"])))) as r:
        fmu, flv, fz, fx = net.forward(x)
        xd = net.decode(zz)
        for a, k in ((fmu, "fwd.mu"), (flv, "fwd.logvar"), (fz, "fwd.z"), (fx[:, :, 40], "fwd.x_slice"),
                     (xd[:, :, 40], "dec.x_slice")):
            # 1e-4 relative, with the floor scaled to the tensor (outputs pass 20 conv+BN layers)
            close(a, g[k], rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(g[k]).max())), what=k)
        ref1 = sd_from(g, "sd1.")
        for k, v in net.state_dict().items():
            close(v, ref1[k], rtol=1e-5, atol=1e-6, what=k)
        net.eval()
        with torch.no_grad():
            emu, elv = net.encode(x)
            ex = net.decode(zz)
        for a, k in ((emu, "eval.mu"), (elv, "eval.logvar"), (ex[:, :, 40], "eval.x_slice")):
            close(a, g[k], rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(g[k]).max())), what=k)
        net.train()

        opt_e = FlatAdam(net.encoder.parameters(), lr=2e-4)
        opt_d = FlatAdam(net.decoder.parameters(), lr=2e-4)
        step = trainer.SoftIntroStep(net, opt_e, opt_d, float(g["hp.beta_rec"]), float(g["hp.beta_neg"]),
                                     float(g["hp.beta_kl"]), float(g["hp.gamma_r"]), float(g["hp.scale"]))
        grads = {}

        def snap(which, module, prefix):
            grads[which] = {prefix + k: p.grad.detach().clone() for k, p in module.named_parameters()}

        orig_e, orig_d = opt_e.step, opt_d.step
        opt_e.step = lambda: (snap("gradE", net.encoder, "encoder."), orig_e())
        opt_d.step = lambda: (snap("gradD", net.decoder, "decoder."), orig_d())
        for s, (real, noise) in enumerate(steps):
            st = dict(zip(trainer.STAT_NAMES, step(real, noise).tolist()))
            # second iteration: starts from weights that differ by Adam's first, sign-like update
            # wherever a noise-level gradient changed sign; KL terms move by a few 1e-3
            srt = 1e-4 if s == 0 else 1e-2
            for mine, ref in (("lossE", "lossE"), ("lossD", "lossD"), ("kl_real", "kl_real"),
                              ("fake_kl", "fake_kl"), ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
                np.testing.assert_allclose(st[mine], float(g[f"s{s}.{ref}"]), rtol=srt, err_msg=mine)
            for mine in ("exp_elbo_rec", "exp_elbo_fake"):
                np.testing.assert_allclose(np.log(st[mine]), np.log(float(g[f"s{s}.{mine}"])), rtol=srt,
                                           err_msg=mine)
            if s == 1:
                assert opt_e.step_count == 2 and opt_d.step_count == 2
                continue
            for which in ("gradE", "gradD"):
                for k, gr in grads[which].items():
                    key = f"s{s}.{which}.{k}"
                    if key + ".none" in g.files:      # encoder.block8.*: built, never called
                        assert float(gr.abs().max()) == 0.0, key
                        continue
                    ref = g[key]
                    if k in cancelled:
                        # analytically zero; both sides hold summation noise over 1e5..1e6 voxels
                        wref = g[key[:-len("bias")] + "weight"]
                        bound = 1e-3 * float(np.abs(wref).max()) + 1e-5
                        assert float(gr.abs().max()) <= bound and float(np.abs(ref).max()) <= bound, key
                        continue
                    assert_grad_as_good_as_reference(gr.numpy(), ref, fc_step0_fp64()[f"{which}.{k}"], key)
            ref_sd = sd_from(g, f"s{s}.sd.")
            for k, v in net.state_dict().items():
                if k.endswith("num_batches_tracked"):
                    assert int(v) == int(ref_sd[k]), k
                elif k.endswith(("running_mean", "running_var")):
                    close(v, ref_sd[k], rtol=1e-4, atol=1e-4, what=k)
                elif k in cancelled:
                    close(v, ref_sd[k], rtol=1e-5, atol=2 * 2e-4 + 2e-6, what=k)
                else:
                    gk = [q for q in (f"s{s}.{w}.{k}" for w in ("gradE", "gradD")) if q in g.files]
                    check_params_after_adam(v.detach().numpy(), ref_sd[k].numpy(),
                                            g[gk[0]] if gk else None, 2e-4, k)
        assert r.source.i == int(g["n_draws"])


def test_trainer_fc_matches_reference_trainer():
    """siv.utils.trainer_fc against the REAL reference utils/trainer_fc.py (fixture g9:
    2 epochs x 2 batches + validation, B=8, narrow FC-latent model)."""
    g = load("g9_train_fc.npz")
    cfg = tuple(int(v) for v in g["cfg"])
    torch.manual_seed(84)
    net = mymodel.SoftIntroVAE(*cfg)
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
            res = trainer_fc.train_soft_intro_vae(net, ListLoader(train), ListLoader(val), 2, 2e-4,
                                                  torch.device("cpu"), d + "/", 1.0, 1024.0, 0.75)
            assert r.source.i == int(g["n_draws"])
            logs = {n: open(os.path.join(d, n)).read() for n in ("loss.txt", "kl_losses.txt")}
            assert sorted(os.listdir(os.path.join(d, "prams"))) == [str(f) for f in g["files"]]
            assert not os.path.exists(os.path.join(d, "train_losses.txt"))
    finally:
        FlatAdam.__init__ = orig_init
    ref0 = sd_from(g, "sd0.")
    for k, v in captured["sd"].items():
        close(v, ref0[k], rtol=0, atol=0, what="init " + k)
    for name, lst in zip(("train_lossE", "train_lossD", "val_lossE", "val_lossD"), res):
        # four iterations of fp32-conditioned gradients (see tests/fc_truth.py) separate the runs
        np.testing.assert_allclose(np.array(lst), g["ret." + name], rtol=1e-3, err_msg=name)
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
