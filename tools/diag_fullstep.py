"""Diagnostic: one E+D iteration of the z=1200 model at 80x96x80, batch 1, three ways -- the fp32 CPU oracle,
the same oracle in float64 (truth) and libsiv on the GPU -- and, per scalar and per parameter gradient, the
distance of each fp32 implementation from the truth.  Shows whether a libsiv-vs-oracle difference is of the
size two correct fp32 implementations have between them.  Usage: python tools/diag_fullstep.py [batch]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

from oracle import siv_oracle as O
import siv.models.models as models
import siv.rng as srng
import siv.utils.my_trainer as trainer
from siv.optim import FlatAdam

Z1200 = (64, [[64, 1, 2], [128, 1, 2], [256, 2, 2]])
V0, V3 = (80, 96, 80), (10, 12, 10)
Bn = int(sys.argv[1]) if len(sys.argv) > 1 else 1


class Draws64(O.Draws):
    def _emit(self, shape, make):
        return super()._emit(shape, make).double()


torch.manual_seed(82)
net = models.SoftIntroVAE(*Z1200)
torch.manual_seed(77)
net.apply(trainer.init_weights_he)
gen = torch.Generator().manual_seed(103)
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm3d):
            m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=gen))
            m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=gen))
sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
real = torch.rand(Bn, 1, *V0, generator=gen)
noise = torch.randn(Bn, 1, *V3, generator=gen)
hp = O.Hyper(dims=V0)
t0 = time.time()
sd32 = {k: v.clone() for k, v in sd0.items()}
ref = O.soft_intro_step(sd32, O.Cfg(*Z1200), real, noise, O.Draws(seed=41), hp, {}, {})
print(f"fp32 oracle {time.time() - t0:.1f} s", flush=True)
t0 = time.time()
sd64 = {k: (v.double() if v.is_floating_point() else v.clone()) for k, v in sd0.items()}
tru = O.soft_intro_step(sd64, O.Cfg(*Z1200), real.double(), noise.double(), Draws64(seed=41), hp, {}, {})
print(f"fp64 oracle {time.time() - t0:.1f} s", flush=True)

import siv.backend as sb

PATHS = [int(v) for v in os.environ.get("DIAG_PATHS", "0,4").split(",")]
NAMES = {0: "automatic dispatch (2-D Winograd forward / input gradient, Winograd weight gradients, collapsed taps "
            "behind an upsample)",
         4: "direct matrix-core kernels only (27 taps, the oracle's order of additions per output)"}
for path in PATHS:
    sb.get_backend().set_conv_path(path)
    net = models.SoftIntroVAE(*Z1200)
    net.load_state_dict(sd0)
    net = net.to("cuda").train()
    opt_e = FlatAdam(net.encoder.parameters(), lr=hp.lr)
    opt_d = FlatAdam(net.decoder.parameters(), lr=hp.lr)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, hp.beta_rec, hp.beta_neg, hp.beta_kl, hp.gamma_r, hp.scale)
    grads = {}
    oe, od = opt_e.step, opt_d.step
    opt_e.step = lambda: (grads.update({"encoder." + k: p.grad.detach().clone().cpu() for k, p in net.encoder.named_parameters()}), oe())
    opt_d.step = lambda: (grads.update({"decoder." + k: p.grad.detach().clone().cpu() for k, p in net.decoder.named_parameters()}), od())
    with srng.use_rng(srng.ReplayRng(O.Draws(seed=41))):
        st = dict(zip(trainer.STAT_NAMES, step(real.cuda(), noise.cuda()).cpu().tolist()))
    print(f"\n==== conv path {path}: {NAMES.get(path, '')}; batch {Bn} ====")
    print("scalar            libsiv          fp32 oracle     fp64 oracle     |siv-64|/64  |o32-64|/64")
    for mine, r in (("lossE", "lossE"), ("kl_real", "kl_real"), ("lossD", "lossD"), ("fake_kl", "fake_kl"),
                    ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
        a, b, c = st[mine], float(ref[r]), float(tru[r])
        print(f"{mine:10s} {a:16.6f} {b:16.6f} {c:16.6f}   {abs(a - c) / abs(c):.2e}   {abs(b - c) / abs(c):.2e}")
    print("gradient (max err / max|truth|)                  libsiv     fp32 oracle")
    worst = {"encoder": [0.0, 0.0], "decoder": [0.0, 0.0]}
    ratio = 0.0
    for k, gr in grads.items():
        w = "gradE" if k.startswith("encoder.") else "gradD"
        rg, tg = ref[w][k], tru[w][k]
        if rg is None:
            continue
        s = float(tg.abs().max()) + 1e-300
        e_siv, e_o32 = float((gr.double() - tg).abs().max()) / s, float((rg.double() - tg).abs().max()) / s
        # the bias of a convolution in front of a BatchNorm (and the decoder's 1x1x1 stem weight) has a true
        # gradient of zero: those rows show rounding noise over ~0 and are left out of the summary
        cancelled = k.endswith(("encoder.blocks.0.0.bias", "decoder.blocks.0.0.bias", "decoder.blocks.0.0.weight"))
        if not cancelled:
            ww = worst[k.split(".")[0]]
            ww[0], ww[1] = max(ww[0], e_siv), max(ww[1], e_o32)
            ratio = max(ratio, e_siv / e_o32)
        print(f"{k:46s} {e_siv:.2e}   {e_o32:.2e}   max|g| {s:.2e}{'   (true gradient = 0)' if cancelled else ''}", flush=True)
    for half, (a, b) in worst.items():
        print(f"worst {half} gradient: libsiv {a:.2e}, fp32 oracle {b:.2e} (of the tensor's maximum, against float64)")
    print(f"largest ratio libsiv / fp32 oracle over the parameters: {ratio:.2f}")
    del net, opt_e, opt_d, step
sb.get_backend().set_conv_path(0)
