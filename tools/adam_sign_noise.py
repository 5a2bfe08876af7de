#!/usr/bin/env python3
"""Why quantities computed after an Adam update carry a looser tolerance in the step tests: the first Adam
update moves every weight by lr * g / (|g| + eps), i.e. by +-lr whatever the size of g, so a weight whose
true gradient is ~0 moves in a direction that rounding decides.  Runs the 64-128 channel model of
tests/test_gpu_model.py::test_wide_model_step_matches_oracle on the device and on the CPU oracle and prints how
many encoder gradient elements differ in sign, how large those are, and what that does to the D-step scalars."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch
import oracle.siv_oracle as O
import tests.test_gpu_model as T

def main():
    import siv.models.models as models
    dims, Bn = (16, 16, 16), 2
    torch.manual_seed(5)
    net = models.SoftIntroVAE(*T.WIDE)
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
    cfg, hp = O.Cfg(*T.WIDE), O.Hyper(dims=dims)
    ref = O.soft_intro_step(sd, cfg, real, noise, O.Draws(seed=17), hp, {}, {})
    net = net.to("cuda").train()
    (st, grads), = T._run_steps(net, lambda s: (real, noise), hp, O.Draws(seed=17), 1)
    for cancelled in (True, False):          # convolution biases in front of a BatchNorm: their true gradient is 0
        flips = total = 0
        worst, where = 0.0, ""
        for k, g in grads.items():
            rg = ref["gradE"].get(k) if k.startswith("encoder.") else None
            if rg is None or k.endswith(T.BN_CANCELLED) != cancelled:
                continue
            f = (torch.sign(g) != torch.sign(rg)) & ((g != 0) | (rg != 0))
            flips += int(f.sum()); total += g.numel()
            if f.any() and float(rg[f].abs().max() / rg.abs().max()) > worst:
                worst, where = float(rg[f].abs().max() / rg.abs().max()), k
        print(f"encoder gradient elements ({'BatchNorm-cancelled biases' if cancelled else 'all others'}): {total}, sign "
              f"differs from the oracle's in {flips}; the largest of those is {worst:.1e} of its tensor's maximum ({where})")
    for mine, r in (("lossE", "lossE"), ("kl_real", "kl_real"), ("lossD", "lossD"), ("fake_kl", "fake_kl"),
                    ("rec_kl", "rec_kl"), ("loss_rec", "loss_rec_D")):
        a, b = st[mine], float(ref[r])
        print(f"  {mine:9s} device {a:.6f}  oracle {b:.6f}  rel {abs(a - b) / abs(b):.1e}")

if __name__ == "__main__":
    main()
