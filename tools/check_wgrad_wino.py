#!/usr/bin/env python3
"""Winograd F(3x3,2x2) weight-gradient kernel (conv path 6) against an fp64 reference and against the
direct matrix-core kernel (conv path 4): error at small shapes (ragged ones included), then timing at
the benchmark shapes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import siv.backend as sb  # noqa: E402

B = sb.get_backend()
g = torch.Generator().manual_seed(5)
dev = torch.device("cuda")


def ref_wgrad(x, dy):
    xd = x.double().cpu().requires_grad_(False)
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, 3, dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv3d(xd, w, None, 1, 1)
    y.backward(dy.double().cpu())
    return w.grad


ok = True
for (n, cin, cout, dims) in ((2, 64, 64, (8, 8, 8)), (1, 64, 64, (5, 6, 7)), (2, 64, 128, (4, 16, 8)),
                             (1, 128, 64, (9, 10, 12)), (3, 64, 64, (17, 8, 24)), (2, 128, 128, (6, 12, 20))):
    x = torch.randn(n, cin, *dims, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(n, cout, *dims, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    ref = ref_wgrad(x, dy)
    res = {}
    for path in (4, 6):
        B.set_conv_path(path)
        dw, _ = B.conv3d_k3_wgrad(x, dy, 1, False)
        res[path] = dw.double().cpu()
    B.set_conv_path(0)
    scale = ref.abs().max().item()
    e4 = (res[4] - ref).abs().max().item() / scale
    e6 = (res[6] - ref).abs().max().item() / scale
    flag = "ok" if e6 < 1e-5 else "FAIL"
    ok = ok and e6 < 1e-5
    print(f"n={n} {cin:3d}->{cout:3d} {dims}: direct {e4:.2e}  winograd {e6:.2e}  {flag}", flush=True)
if not ok:
    sys.exit(1)

# upsampled input (x at low resolution)
for (n, cin, cout, dims) in ((2, 64, 64, (8, 8, 8)), (1, 64, 64, (6, 10, 14)), (2, 64, 128, (4, 16, 8)),
                             (1, 128, 64, (10, 12, 20)), (3, 64, 64, (18, 8, 24)), (1, 64, 64, (40, 8, 8))):
    D, H, W = dims
    x = torch.randn(n, cin, D // 2, H // 2, W // 2, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(n, cout, *dims, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
    xu = torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")
    ref = ref_wgrad(xu, dy)
    res = {}
    for path in (4, 0, 6):
        B.set_conv_path(path)
        res[path] = B.conv3d_k3_wgrad(x, dy, 2, False)[0].double().cpu()
    B.set_conv_path(0)
    scale = ref.abs().max().item()
    e = {k: (v - ref).abs().max().item() / scale for k, v in res.items()}
    flag = "ok" if e[6] < 1e-5 else "FAIL"
    ok = ok and e[6] < 1e-5
    print(f"up n={n} {cin:3d}->{cout:3d} {dims}: direct {e[4]:.2e} collapsed {e[0]:.2e} winograd {e[6]:.2e}  {flag}", flush=True)
if not ok:
    sys.exit(1)

# accumulate flag
x = torch.randn(2, 64, 8, 8, 8, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
dy = torch.randn(2, 64, 8, 8, 8, generator=g).to(dev).contiguous(memory_format=torch.channels_last_3d)
B.set_conv_path(6)
dw1, _ = B.conv3d_k3_wgrad(x, dy, 1, False)
print("finite", bool(torch.isfinite(dw1).all()))
B.set_conv_path(0)

SHAPES = (("v0_64_64", 64, 64, (80, 96, 80)), ("v1_64_64", 64, 64, (40, 48, 40)), ("v1_64_128", 64, 128, (40, 48, 40)),
          ("v1_128_128", 128, 128, (40, 48, 40)), ("v2_128_128", 128, 128, (20, 24, 20)),
          ("v2_128_256", 128, 256, (20, 24, 20)), ("v2_256_256", 256, 256, (20, 24, 20)),
          ("v3_256_256", 256, 256, (10, 12, 10)))
wx = torch.randn(4096, 4096, device=dev)
for _ in range(20):
    wx @ wx
torch.cuda.synchronize()
UPS = (("v0_64_64_up", 64, 64, (80, 96, 80)), ("v1_128_64_up", 128, 64, (40, 48, 40)),
       ("v2_256_128_up", 256, 128, (20, 24, 20)))
for name, cin, cout, (D, H, W) in UPS:
    x = torch.randn(8, cin, D // 2, H // 2, W // 2, device=dev).contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(8, cout, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    flop = 2.0 * 27 * cin * cout * D * H * W * 8
    line = f"{name:14s}"
    outs = {}
    for path in (2, 6):
        B.set_conv_path(path)
        outs[path] = B.conv3d_k3_wgrad(x, dy, 2, False)[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            B.conv3d_k3_wgrad(x, dy, 2, False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += f"  path {path}: {ms:7.3f} ms {flop / ms / 1e9:6.1f} TF/s"
    B.set_conv_path(0)
    d = ((outs[2] - outs[6]).abs().max() / outs[2].abs().max()).item()
    print(line + f"  max diff {d:.2e}", flush=True)
for name, cin, cout, (D, H, W) in SHAPES:
    x = torch.randn(8, cin, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    dy = torch.randn(8, cout, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
    flop = 2.0 * 27 * cin * cout * D * H * W * 8
    line = f"{name:12s}"
    outs = {}
    for path in (4, 6):
        B.set_conv_path(path)
        outs[path] = B.conv3d_k3_wgrad(x, dy, 1, False)[0]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            B.conv3d_k3_wgrad(x, dy, 1, False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        line += f"  path {path}: {ms:7.3f} ms {flop / ms / 1e9:6.1f} TF/s"
    B.set_conv_path(0)
    d = ((outs[4] - outs[6]).abs().max() / outs[4].abs().max()).item()
    print(line + f"  max diff {d:.2e}", flush=True)
