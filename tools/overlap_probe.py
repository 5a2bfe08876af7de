#!/usr/bin/env python3
"""Does an HBM-bound elementwise kernel run BESIDE a matrix-core convolution kernel, or do they
time-share the CUs?  Times a Winograd weight gradient (64->64 @80x96x80, batch 8) and a train of
in-place scale kernels (few VGPRs) over a tensor of the same size, alone, back to back on one stream
and concurrently on two streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import siv.backend as sb  # noqa: E402

B = sb.get_backend()
dev = torch.device("cuda")
cl = torch.channels_last_3d
x = torch.randn(8, 64, 80, 96, 80, device=dev).contiguous(memory_format=cl)
dy = torch.randn(8, 64, 80, 96, 80, device=dev).contiguous(memory_format=cl)
w = (torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05).contiguous(memory_format=cl)
t = torch.randn(8, 64, 80, 96, 80, device=dev)
NSCALE = int(sys.argv[1]) if len(sys.argv) > 1 else 6
which = sys.argv[2] if len(sys.argv) > 2 else "wgrad"
side = torch.cuda.Stream()
conv = {"wgrad": lambda: B.conv3d_k3_wgrad(x, dy, 1, False), "fwd": lambda: B.conv3d_k3_fwd(x, w, None, 1),
        "dgrad": lambda: B.conv3d_k3_dgrad(dy, w, 1)}[which]


def elem():
    for _ in range(NSCALE):
        B.scale_(t, 1.0001)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        conv()
    elem()
    main.wait_stream(side)


with torch.cuda.stream(side):          # workspace of the side stream allocated before timing
    conv()
torch.cuda.synchronize()
a, b = timed(conv), timed(elem)
c = timed(lambda: (conv(), elem()))
d = timed(both)
print(f"{which} alone {a:.3f} ms, {NSCALE} scale kernels alone {b:.3f} ms, back to back {c:.3f} ms, "
      f"on two streams {d:.3f} ms (ideal {max(a, b):.3f})")
