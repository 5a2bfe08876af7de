#!/usr/bin/env python3
"""Iteration time and peak device memory of the conv-latent Soft-IntroVAE on 160x192x160 volumes on
one MI355X: models.SoftIntroVAE(64, [[64,1,2],[64,1,2],[128,1,2],[256,2,2]]) -- the z=1200 model
with one more stride-2 stage in front, so that the latent stays 10x12x10 (this stage is not in
the reference, whose scripts only build 80x96x80 models) -- fp32 or bf16 storage of the saved activations
(--activations), optional device
flip/shift augmentation + normalisation in front of the step.  Not the headline benchmark."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402

DIMS = (160, 192, 160)
MODEL = (64, [[64, 1, 2], [64, 1, 2], [128, 1, 2], [256, 2, 2]])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=2)      # the second iteration still grows the allocator's pool
    ap.add_argument("--augment", action="store_true")
    ap.add_argument("--activations", choices=("fp32", "bf16"), default="fp32",
                    help="storage of the tensors kept for backward (siv.activations)")
    ap.add_argument("--dims", type=int, nargs=3, default=list(DIMS))
    a = ap.parse_args()
    import siv.activations as act
    import siv.models.models as models
    import siv.rng as srng
    import siv.utils.data_load as DL
    import siv.utils.my_trainer as trainer
    from siv.optim import FlatAdam
    dev = torch.device("cuda")
    dims = tuple(a.dims)
    act.set_storage(a.activations)
    torch.manual_seed(82)
    net = models.soft_intro_vae_z1200(full_resolution=True)
    assert net.encoder.block_setting == MODEL[1]
    torch.manual_seed(77)
    net.apply(trainer.init_weights_he)
    net = net.to(dev).train()
    opt_e = FlatAdam(net.encoder.parameters(), lr=2e-4)
    opt_d = FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, opt_e, opt_d, 1.0, 1024.0, 0.75, 1e-8, None, None)
    srng.manual_seed(77, 0)
    gen = torch.Generator().manual_seed(103)
    raw = [torch.rand(a.batch, 1, *dims, generator=gen).to(dev) for _ in range(2)]
    aug = DL.FlipShift(seed=1) if a.augment else None
    last = None
    for i in range(a.warmup + a.steps):
        if i == a.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        x = raw[i % 2]
        if aug is not None:
            x = DL.preprocess(aug(x))
        last = step(x, srng.get_rng().eps(net.latent_shape(a.batch, dims), dev))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    st = dict(zip(trainer.STAT_NAMES, last.cpu().tolist()))
    print(json.dumps({"workload": f"models.SoftIntroVAE{MODEL} {'x'.join(map(str, dims))} fp32 arithmetic, "
                                  f"{a.activations} saved activations, batch {a.batch}"
                                  + (" + device flip/shift + normalisation" if a.augment else ""),
                      "ms_per_step": round(dt * 1e3, 1), "volumes_per_s": round(a.batch / dt, 3),
                      "peak_device_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1),
                      "losses": {k: st[k] for k in ("lossE", "lossD", "loss_rec", "kl_real")}}))


if __name__ == "__main__":
    main()
