#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound kernels (BatchNorm tail, single-channel convs) at the
benchmark shapes: HIP-event time per call and the algorithmic bytes it moves."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import siv.backend as sb  # noqa: E402

CL = torch.channels_last_3d


def timeit(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", nargs="*", default=None)
    a = ap.parse_args()
    B = sb.get_backend()
    dev = torch.device("cuda")
    n = a.batch
    shapes = [(64, (80, 96, 80)), (64, (40, 48, 40)), (128, (40, 48, 40)), (128, (20, 24, 20)),
              (256, (20, 24, 20)), (256, (10, 12, 10))]

    def report(name, ms, nbytes):
        print(f"{name:44s} {ms * 1e3:9.1f} us  {nbytes / ms / 1e9:7.2f} TB/s", flush=True)

    for C, (D, H, W) in shapes:
        tag = f"C{C}@{D}x{H}x{W}"
        if a.only and not any(o in tag for o in a.only):
            continue
        y = torch.randn(n, C, D, H, W, device=dev).contiguous(memory_format=CL)
        res = torch.randn_like(y)
        g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
        T = y.numel() * 4
        mean, var = B.bn_stats(y, None, None, None, 0.1)
        report(f"{tag} bn_stats", timeit(lambda: B.bn_stats(y, None, None, None, 0.1), a.reps), T)
        report(f"{tag} bn_act_fwd", timeit(lambda: B.bn_act_fwd(y, mean, var, g, b, None, None, 0.0, 0.2, 1, 1e-5), a.reps), 2 * T)
        report(f"{tag} bn_act_fwd+res", timeit(lambda: B.bn_act_fwd(y, mean, var, g, b, res, None, 0.0, 0.2, 1, 1e-5), a.reps), 3 * T)
        report(f"{tag} bn_act_fwd pool2", timeit(lambda: B.bn_act_fwd(y, mean, var, g, b, None, None, 0.0, 0.2, 2, 1e-5), a.reps), T + T // 8)
        dout = torch.randn_like(y)
        report(f"{tag} bn_act_bwd", timeit(lambda: B.bn_act_bwd(dout, y, mean, var, g, b, None, None, 0.0, 0.2, 1, 1e-5, True, False), a.reps), 5 * T)
        report(f"{tag} bn_act_bwd+res", timeit(lambda: B.bn_act_bwd(dout, y, mean, var, g, b, res, None, 0.0, 0.2, 1, 1e-5, True, True), a.reps), 8 * T)
        dp = torch.randn(n, C, D // 2, H // 2, W // 2, device=dev).contiguous(memory_format=CL)
        report(f"{tag} bn_act_bwd pool2", timeit(lambda: B.bn_act_bwd(dp, y, mean, var, g, b, None, None, 0.0, 0.2, 2, 1e-5, True, False), a.reps), 3 * T + 2 * (T // 8))
        if C == 64 and D == 80:
            keep = (torch.rand(n, C, D, H, W, device=dev) > 0.35).to(torch.uint8).contiguous(memory_format=CL)
            report(f"{tag} bn_act_fwd+keep", timeit(lambda: B.bn_act_fwd(y, mean, var, g, b, None, keep, 0.35, 0.2, 1, 1e-5), a.reps), 2 * T + T // 4)
            report(f"{tag} bn_act_bwd+keep", timeit(lambda: B.bn_act_bwd(dout, y, mean, var, g, b, None, keep, 0.35, 0.2, 1, 1e-5, True, False), a.reps), 5 * T + T // 2)
            x1 = torch.randn(n, 1, D, H, W, device=dev).contiguous(memory_format=CL)
            w1 = (torch.randn(C, 1, 3, 3, 3, device=dev) * 0.2).contiguous(memory_format=CL)
            wc = (torch.randn(1, C, 3, 3, 3, device=dev) * 0.05).contiguous(memory_format=CL)
            d1 = torch.randn(n, 1, D, H, W, device=dev).contiguous(memory_format=CL)
            report(f"{tag} conv 1->C fwd", timeit(lambda: B.conv3d_k3_fwd(x1, w1, None, 1), a.reps), T)
            report(f"{tag} conv C->1 fwd", timeit(lambda: B.conv3d_k3_fwd(y, wc, None, 1), a.reps), T)
            report(f"{tag} conv 1->C dgrad (C->1)", timeit(lambda: B.conv3d_k3_dgrad(dout, w1, 1), a.reps), T)
            report(f"{tag} conv C->1 dgrad (1->C)", timeit(lambda: B.conv3d_k3_dgrad(d1, wc, 1), a.reps), T)
            report(f"{tag} conv 1->C wgrad", timeit(lambda: B.conv3d_k3_wgrad(x1, dout, 1, True), a.reps), T)
            report(f"{tag} conv C->1 wgrad", timeit(lambda: B.conv3d_k3_wgrad(y, d1, 1, True), a.reps), T)
            ad = torch.randn_like(y)
            report(f"{tag} act_dropout fwd", timeit(lambda: B.act_dropout_fwd(y, None, 0.0, 0.0), a.reps), 2 * T)
            report(f"{tag} sse_fwd", timeit(lambda: B.sse_fwd(y, ad), a.reps), 2 * T)


if __name__ == "__main__":
    main()
