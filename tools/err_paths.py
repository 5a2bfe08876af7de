import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")): sys.path.insert(0, p)
import torch
import siv.backend as sb
B = sb.get_backend()
g = torch.Generator().manual_seed(1)
for (cin, cout, dims) in ((64, 64, (16, 16, 16)), (64, 64, (8, 8, 8)), (64, 128, (8, 8, 8)), (128, 128, (8, 8, 8)),
                          (128, 128, (4, 4, 4)), (128, 64, (4, 4, 4)), (64, 64, (4, 4, 4))):
    x = torch.randn(2, cin, *dims, generator=g).cuda(); w = (torch.randn(cout, cin, 3, 3, 3, generator=g) * (2.0 / (27 * cin)) ** 0.5).cuda()
    dy = torch.randn(2, cout, *dims, generator=g).cuda()
    out = {}
    for path in (4, 5):
        B.set_conv_path(path)
        out[path] = (B.conv3d_k3_fwd(x, w, None, 1), B.conv3d_k3_dgrad(dy, w, 1))
    B.set_conv_path(0)
    for i, nm in enumerate(("fwd", "dgrad")):
        a, b = out[4][i], out[5][i]
        print(f"{cin:4d}->{cout:4d} {dims} {nm:6s} max|p5-p4|/max = {((a - b).abs().max() / a.abs().max()).item():.3e}")
