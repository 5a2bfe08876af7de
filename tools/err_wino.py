#!/usr/bin/env python3
"""Rounding error of the forward convolution kernels against float64 (CPU), 64 -> 64 and 64 -> 128 on volumes with
8 x 8 brick planes: direct kernel (conv path 4), Winograd F(2x2,3x3) (automatic path) and F(2,3) x F(4,3)
(automatic path with SIV_W2_W4=1).  Prints max and rms error as fractions of the output's maximum."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import siv.backend as sb  # noqa: E402

B = sb.get_backend()
g = torch.Generator().manual_seed(1)
for cin, cout, dims, n in ((64, 64, (24, 32, 40), 2), (64, 128, (16, 24, 24), 2), (128, 128, (16, 24, 24), 2)):
    for kind in ("randn", "smooth"):
        x = torch.randn(n, cin, *dims, generator=g)
        if kind == "smooth":      # activations after BatchNorm + LeakyReLU of a smooth field: positive mean, correlated
            x = F.leaky_relu(F.avg_pool3d(F.pad(x, (2,) * 6, mode="replicate"), 5, 1) * 3 + 0.5, 0.2)
        w = torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5
        ref = F.conv3d(x.double(), w.double(), None, padding=1)
        m = float(ref.abs().max())
        xs, ws = x.cuda(), w.cuda()
        for name, path, w4 in (("direct", 4, "0"), ("F(2x2)", 6, "0"), ("F(2x4)", 6, "1")):
            os.environ["SIV_W2_W4"] = w4
            B.set_conv_path(path)
            y = B.conv3d_k3_fwd(xs, ws, None, 1).cpu().double()
            B.set_conv_path(0)
            e = y - ref
            print(f"{cin:4d}->{cout:4d} {dims} {kind:6s} {name:7s} max {float(e.abs().max()) / m:.2e}  rms {float(e.pow(2).mean().sqrt()) / m:.2e}")
os.environ["SIV_W2_W4"] = "0"
