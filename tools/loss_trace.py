#!/usr/bin/env python3
"""Losses of the first K iterations of the benchmark workload under two convolution paths (same
weights, data and device RNG streams): how far the exact-fp32 kernel families drift apart."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
import siv.backend as sb  # noqa: E402
import siv.rng as srng  # noqa: E402
import siv.utils.my_trainer as trainer  # noqa: E402
from siv.optim import FlatAdam  # noqa: E402
from siv.parallel import GradSync  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
BATCH = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda")
B = sb.get_backend()
out = {}
for path in (4, 0):
    B.set_conv_path(path)
    net = bench.build_model(dev)
    oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
    step = trainer.SoftIntroStep(net, oe, od, 1.0, 1024.0, 0.75, 1e-8, None, GradSync(dev))
    srng.manual_seed(77, 0)
    gen = torch.Generator().manual_seed(103)
    batches = [torch.rand(BATCH, 1, *bench.DIMS, generator=gen).to(dev) for _ in range(2)]
    rows = []
    for i in range(K):
        real = batches[i % 2]
        noise = srng.get_rng().eps(net.latent_shape(real.size(0), bench.DIMS), dev)
        rows.append(dict(zip(trainer.STAT_NAMES, step(real, noise).cpu().tolist())))
    out[path] = rows
B.set_conv_path(0)
for i in range(K):
    a, b = out[4][i], out[0][i]
    print(i, " ".join(f"{k}={a[k]:.6g}/{b[k]:.6g}({abs(a[k] - b[k]) / max(abs(a[k]), 1e-30):.1e})"
                      for k in ("lossE", "lossD", "loss_rec", "kl_real")), flush=True)
