#!/usr/bin/env python3
"""Micro-benchmark of the convolution kernels at the benchmark shapes (HIP-event timing on the
launch stream).  Used for kernel tuning and as the program profiled with rocprofv3 --pmc."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import siv.backend as sb  # noqa: E402

SHAPES = {  # name: (Cin, Cout, (D,H,W) output, ups)
    "v0_64_64": (64, 64, (80, 96, 80), 1), "v0_64_64_up": (64, 64, (80, 96, 80), 2),
    "v1_64_64": (64, 64, (40, 48, 40), 1), "v1_64_128": (64, 128, (40, 48, 40), 1),
    "v1_128_64_up": (128, 64, (40, 48, 40), 2), "v2_128_128": (128, 128, (20, 24, 20), 1),
    "v2_128_256": (128, 256, (20, 24, 20), 1), "v2_256_128_up": (256, 128, (20, 24, 20), 2),
    "v3_256_256": (256, 256, (10, 12, 10), 1), "v0_1_64": (1, 64, (80, 96, 80), 1),
    "v0_64_1": (64, 1, (80, 96, 80), 1),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=list(SHAPES))
    ap.add_argument("--ops", nargs="*", default=["fwd", "dgrad", "wgrad"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--path", type=int, default=0, help="siv_set_conv_path value (3 = Winograd)")
    a = ap.parse_args()
    B = sb.get_backend()
    B.set_conv_path(a.path)
    dev = torch.device("cuda")
    # bring the clocks up before the first measurement
    wx = torch.randn(4096, 4096, device=dev)
    for _ in range(20):
        wx @ wx
    torch.cuda.synchronize()
    for name in a.shapes:
        cin, cout, (D, H, W), ups = SHAPES[name]
        n = a.batch
        x = torch.randn(n, cin, D // ups, H // ups, W // ups, device=dev).contiguous(memory_format=torch.channels_last_3d)
        w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last_3d)
        dy = torch.randn(n, cout, D, H, W, device=dev).contiguous(memory_format=torch.channels_last_3d)
        flop = 2.0 * 27 * cin * cout * D * H * W * n
        fns = {"fwd": lambda: B.conv3d_k3_fwd(x, w, None, ups), "dgrad": lambda: B.conv3d_k3_dgrad(dy, w, ups),
               "wgrad": lambda: B.conv3d_k3_wgrad(x, dy, ups, False)}
        for op in a.ops:
            fns[op]()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fns[op]()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.reps
            print(f"{name:16s} {op:6s} {ms:8.3f} ms  {flop / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
