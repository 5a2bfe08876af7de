#!/usr/bin/env python3
"""Iteration time of the FC-latent Soft-IntroVAE (models/mymodel.py) on one MI355X:
80x96x80 fp32, SoftIntroVAE(16, 32, 64, 128, 600), batch 4 (the z=600 configuration) -- or any
widths / batch given on the command line.  Not the headline benchmark (see bench.py)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--widths", type=int, nargs=5, default=[16, 32, 64, 128, 600])
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graph", action="store_true", help="capture the iteration in a HIP graph and replay it")
    a = ap.parse_args()
    import siv.models.mymodel as mymodel
    import siv.rng as srng
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    dev = torch.device("cuda")
    torch.manual_seed(82)
    net = mymodel.SoftIntroVAE(*a.widths)
    torch.manual_seed(77)
    net.apply(trainer.init_weights_he)
    net = net.to(dev).train()
    opt_e = FlatAdam(net.encoder.parameters(), lr=2e-4)
    opt_d = FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, 1.0, 1024.0, 0.75, 1e-8, None, None)
    srng.manual_seed(77, 0)
    gen = torch.Generator().manual_seed(103)
    batches = [torch.rand(a.batch, 1, 80, 96, 80, generator=gen).to(dev) for _ in range(2)]
    if a.graph:
        step = trainer.GraphedSoftIntroStep(step, batches[0].shape, (a.batch, net.z_ch), dev)
    last = None
    for i in range(a.warmup + a.steps):
        if i == a.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        last = step(batches[i % 2], srng.get_rng().eps((a.batch, net.z_ch), dev))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    st = dict(zip(trainer.STAT_NAMES, last.cpu().tolist()))
    print(json.dumps({"workload": f"mymodel.SoftIntroVAE{tuple(a.widths)} 80x96x80 fp32 batch {a.batch}",
                      "launch": "HIP graph replay" if a.graph else "kernel by kernel",
                      "ms_per_step": round(dt * 1e3, 2), "volumes_per_s": round(a.batch / dt, 3),
                      "losses": {k: st[k] for k in ("lossE", "lossD", "loss_rec", "kl_real")}}))


if __name__ == "__main__":
    main()
