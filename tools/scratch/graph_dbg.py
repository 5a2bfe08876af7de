import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "soft-intro-vae-for-3d-mri_amd")):
    sys.path.insert(0, p)
import torch
import siv.models.models as models
import siv.rng as srng
import siv.utils.my_trainer as trainer
from siv.optim import FlatAdam
WIDE = (64, [[64, 1, 2], [128, 1, 2]])
dims, Bn = (16, 24, 16), 2
gen = torch.Generator().manual_seed(51)
batches = [(torch.rand(Bn, 1, *dims, generator=gen).cuda(), torch.randn(Bn, 1, 4, 6, 4, generator=gen).cuda()) for _ in range(2)]
def build():
    torch.manual_seed(50)
    net = models.SoftIntroVAE(*WIDE)
    net.apply(trainer.init_weights_he)
    net = net.to("cuda").train()
    oe, od = FlatAdam(net.encoder.parameters(), lr=2e-4), FlatAdam(net.decoder.parameters(), lr=2e-4)
    return net, oe, od, trainer.SoftIntroStep(net, oe, od)
import siv.backend as sb
if os.environ.get("FULL_INV") == "1":
    B = sb.get_backend()
    orig = B.weights_changed
    B.weights_changed = lambda device=None, flat=None: orig(device if device is not None else (flat.device if flat is not None else None))
srng.manual_seed(9, 0)
net_g, oe_g, od_g, step_g = build()
g = trainer.GraphedSoftIntroStep(step_g, batches[0][0].shape, batches[0][1].shape, "cuda")
print("sid_base", g.sid_base)
outs = [g(r, n).clone() for r, n in batches]
print("graph", [o.cpu().tolist()[:4] for o in outs])
for ns in (srng.DeviceRng.GRAPH_NS, 0):
    net_e, oe_e, od_e, step_e = build()
    r = srng.DeviceRng(9, 0); r.ns = ns
    srng.set_rng(r)
    res = []
    for i, (real, noise) in enumerate(batches):
        r.calls = g.sid_base + i * srng.DeviceRng.EPOCH_STRIDE
        res.append(step_e(real, noise).cpu().tolist()[:4])
    print("eager ns", ns, res)
