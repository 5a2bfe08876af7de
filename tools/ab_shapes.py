#!/usr/bin/env python3
"""Forward time of the 2-D Winograd kernel at a few synthetic shapes (which property of the 10x12x10 level
costs its chunks 17 k cycles where the 80x96x80 level takes 14 k: the 6x10 brick planes or the 256 channels)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch
import siv.backend as sb
B = sb.get_backend(); B.set_conv_path(6)
dev = torch.device("cuda")
wx = torch.randn(4096, 4096, device=dev)
for _ in range(20): wx @ wx
for name, (cin, cout, dims, n) in {"c256 tpr4 16x16x16 b32": (256, 256, (16, 16, 16), 32), "c256 tpr5 10x12x10 b32": (256, 256, (10, 12, 10), 32),
                                   "c64 tpr5 10x12x10 b128": (64, 64, (10, 12, 10), 128), "c64 tpr4 16x16x16 b128": (64, 64, (16, 16, 16), 128),
                                   "c128 tpr5 20x24x20 b8": (128, 128, (20, 24, 20), 8), "c128 tpr4 24x24x24 b8": (128, 128, (24, 24, 24), 8)}.items():
    x = torch.randn(n, cin, *dims, device=dev).contiguous(memory_format=torch.channels_last_3d)
    w = torch.randn(cout, cin, 3, 3, 3, device=dev).contiguous(memory_format=torch.channels_last_3d)
    for _ in range(3): B.conv3d_k3_fwd(x, w, None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): B.conv3d_k3_fwd(x, w, None, 1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    fl = 2.0 * 27 * cin * cout * dims[0] * dims[1] * dims[2] * n
    print(f"{name:28s} {ms:7.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s algorithmic  = {fl * 12 / 27 / ms / 1e9 / 157.3:.3f} of peak issued")
