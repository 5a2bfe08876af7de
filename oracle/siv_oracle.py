"""CPU oracle for the Soft-IntroVAE 3D-MRI training path.  TEST INFRASTRUCTURE ONLY.

This file is a restatement, in plain functional PyTorch fp32 on the CPU, of the
arithmetic the reference performs on its hot path.  It is imported only by
``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of
``bench.py`` -- always as the checker, never by the product package.

Parity pinning: the reference ships no tests, fixtures or golden vectors of its
own.  The oracle is pinned instead against outputs of the reference itself,
imported in the build container by ``tools/make_goldens.py`` (explicit dropout
masks / eps so both sides consume the same random tensors); the resulting
``tests/golden/*.npz`` fixtures are checked by ``tests/test_oracle_golden.py``.

Every function cites the reference lines it follows (paths relative to the
reference repository root).

State is a flat ``dict[str, Tensor]`` with exactly the reference's
``state_dict`` keys, so a reference checkpoint can be fed in unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]

BN_EPS = 1e-5        # nn.BatchNorm3d default, models/models.py:18
BN_MOMENTUM = 0.1    # nn.BatchNorm3d default

# bound at import so that a caller may patch torch.randn / torch.rand around a
# reference run without the draw generator recursing into the patch
_randn, _rand = torch.randn, torch.rand


# --------------------------------------------------------------------------
# random-draw plumbing: every stochastic op consumes one explicit tensor
# --------------------------------------------------------------------------
class Draws:
    """Source of the explicit random tensors the stochastic ops consume, in call
    order.  ``mask`` returns the 0/1 keep-mask of one nn.Dropout call, ``eps``
    the tensor one ``torch.randn_like`` call returns in the reference.

    Tensors are derived deterministically from (seed, call index) with the CPU
    generator, so a golden fixture stores only the seed; ``log`` keeps what was
    handed out so a test can replay the very same tensors into another
    implementation (``Draws(replay=log)``)."""

    def __init__(self, seed: int = 0, replay: Optional[Sequence[Tensor]] = None):
        self.seed = seed
        self.replay = list(replay) if replay is not None else None
        self.i = 0
        self.log: List[Tensor] = []

    def _gen(self) -> torch.Generator:
        return torch.Generator().manual_seed(self.seed * 1000003 + self.i)

    def _emit(self, shape, make) -> Tensor:
        if self.replay is not None:
            t = self.replay[self.i]
            assert tuple(t.shape) == tuple(shape), (self.i, tuple(t.shape), tuple(shape))
        else:
            t = make(self._gen())
        self.i += 1
        self.log.append(t)
        return t

    def mask(self, shape, p: float) -> Tensor:
        return self._emit(shape, lambda g: (_rand(tuple(shape), generator=g) >= p).float())

    def eps(self, shape) -> Tensor:
        return self._emit(shape, lambda g: _randn(tuple(shape), generator=g))


def dropout(x: Tensor, p: float, draws: Optional[Draws], train: bool) -> Tensor:
    """nn.Dropout(p): y = x * keep / (1 - p) in training, identity in eval."""
    if not train or p == 0.0:
        return x
    keep = draws.mask(x.shape, p)
    return x * keep * (1.0 / (1.0 - p))


# --------------------------------------------------------------------------
# model configuration
# --------------------------------------------------------------------------
class Cfg:
    """Architecture description shared by models/models.py (variant 'sintro':
    LeakyReLU(0.2) + dropouts) and models/vaemodel.py (variant 'vae': ReLU,
    no dropout)."""

    def __init__(self, in_ch: int, block_setting, variant: str = "sintro"):
        self.in_ch = in_ch
        self.block_setting = [list(b) for b in block_setting]
        self.variant = variant
        if variant == "sintro":
            self.slope = 0.2            # models/models.py:15,19,94,121
            self.p_enc_stem = 0.35      # models/models.py:95
            self.p_dec_stem = 0.25      # models/models.py:122
            self.p_dec_out = 0.35       # models/models.py:140
        elif variant == "vae":
            self.slope = 0.0            # nn.ReLU, models/vaemodel.py:14,18,90,113
            self.p_enc_stem = self.p_dec_stem = self.p_dec_out = 0.0
        else:
            raise ValueError(variant)

    def enc_blocks(self) -> List[Tuple[int, int, int]]:
        """(in_ch, out_ch, stride) per BuildingBlock -- models/models.py:97-102."""
        out, c_in = [], self.in_ch
        for c, n, s in self.block_setting:
            for i in range(n):
                out.append((c_in, c, s if i == 0 else 1))
                c_in = c
        return out

    def dec_blocks(self) -> List[Tuple[int, int, int]]:
        """(in_ch, out_ch, stride) per UpsampleBuildingkBlock -- models/models.py:124-135."""
        bs = self.block_setting
        out, c_in = [], bs[-1][0]
        for i in range(len(bs)):
            nc = self.in_ch if i == len(bs) - 1 else bs[::-1][i + 1][0]
            c, n, s = bs[::-1][i]
            for j in range(n):
                stride = s if j == n - 1 else 1
                cc = nc if j == n - 1 else c
                out.append((c_in, cc, stride))
                c_in = cc
        return out

    @property
    def inner_ch(self) -> int:
        return self.block_setting[-1][0]


class FcCfg:
    """The FC-latent model of models/mymodel.py: SoftIntroVAE(first_ch, second_ch, third_ch,
    forth_ch, z_ch).  LeakyReLU(0.2) everywhere, no dropout, biased convolutions, a Linear
    head/stem around a forth_ch x 5 x 6 x 5 bottleneck (input must be 80 x 96 x 80)."""

    slope = 0.2

    def __init__(self, first_ch: int, second_ch: int, third_ch: int, forth_ch: int, z_ch: int):
        self.ch = (first_ch, second_ch, third_ch, forth_ch)
        self.z_ch = z_ch


# --------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------
def act(x: Tensor, slope: float) -> Tensor:
    return F.leaky_relu(x, slope) if slope != 0.0 else F.relu(x)


def batchnorm(x: Tensor, sd: State, prefix: str, train: bool) -> Tensor:
    """nn.BatchNorm3d: batch statistics + running-stat update (unbiased var,
    momentum 0.1) in training, running statistics in eval."""
    nbt = sd.get(prefix + "num_batches_tracked")
    if train and nbt is not None:
        nbt += 1
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"],
                        sd[prefix + "weight"], sd[prefix + "bias"],
                        training=train, momentum=BN_MOMENTUM, eps=BN_EPS)


def building_block(x: Tensor, sd: State, p: str, cin: int, cout: int, stride: int,
                   slope: float, train: bool) -> Tensor:
    """BuildingBlock.forward -- models/models.py:8-43 (vaemodel.py:8-42).
    conv3 -> BN -> act -> AvgPool(stride) -> conv3 -> BN -> [+shortcut(x) if stride==1] -> act."""
    h = F.conv3d(x, sd[p + "block.0.weight"], None, 1, 1)
    h = batchnorm(h, sd, p + "block.1.", train)
    h = act(h, slope)
    if stride != 1:
        h = F.avg_pool3d(h, stride)
    h = F.conv3d(h, sd[p + "block.4.weight"], None, 1, 1)
    h = batchnorm(h, sd, p + "block.5.", train)
    if stride == 1:                                   # self.res, models/models.py:11,39-41
        sc = x
        if cin != cout:                               # 1x1x1 projection, models/models.py:28-35
            sc = F.conv3d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
        h = h + sc
    return act(h, slope)


def upsample_block(x: Tensor, sd: State, p: str, cin: int, cout: int, stride: int,
                   slope: float, train: bool) -> Tensor:
    """UpsampleBuildingkBlock.forward -- models/models.py:46-80.
    conv3(in->in) -> BN -> act -> nearest Upsample(stride) -> conv3(in->out) -> BN -> [+shortcut] -> act."""
    h = F.conv3d(x, sd[p + "block.0.weight"], None, 1, 1)
    h = batchnorm(h, sd, p + "block.1.", train)
    h = act(h, slope)
    if stride != 1:
        h = F.interpolate(h, scale_factor=stride, mode="nearest")
    h = F.conv3d(h, sd[p + "block.4.weight"], None, 1, 1)
    h = batchnorm(h, sd, p + "block.5.", train)
    if stride == 1:
        sc = x
        if cin != cout:
            sc = F.conv3d(x, sd[p + "shortcut.weight"], sd[p + "shortcut.bias"])
        h = h + sc
    return act(h, slope)


def _cbl(x: Tensor, sd: State, p: str, i: int, train: bool, slope: Optional[float] = 0.2) -> Tensor:
    """Entries i, i+1, i+2 of one of models/mymodel.py's Sequentials: biased conv3 -> BN ->
    LeakyReLU (``slope=None`` stops after the BN, as block7 / decoder block1 / block3 do)."""
    h = F.conv3d(x, sd[f"{p}{i}.weight"], sd[f"{p}{i}.bias"], 1, 1)
    h = batchnorm(h, sd, f"{p}{i + 1}.", train)
    return h if slope is None else act(h, slope)


def fc_encode(x: Tensor, sd: State, cfg: FcCfg, train: bool, prefix: str = "encoder.") -> Tuple[Tensor, Tensor]:
    """ResNetVAEencoder.forward -- models/mymodel.py:127-143."""
    for b in ("block1.", "block2.", "block3."):                 # :128-133
        x = _cbl(_cbl(x, sd, prefix + b, 0, train), sd, prefix + b, 3, train)
        x = F.avg_pool3d(x, 2)
    x = _cbl(x, sd, prefix + "block4short.", 0, train)          # :134
    h = _cbl(x, sd, prefix + "block5.", 0, train)               # :135
    x = act(x + h, cfg.slope)                                   # :136
    x = _cbl(x, sd, prefix + "block6.", 0, train)               # :137 (conv, BN, act, pool, conv, BN, act)
    x = F.avg_pool3d(x, 2)
    x = _cbl(x, sd, prefix + "block6.", 4, train)
    h = _cbl(x, sd, prefix + "block7.", 0, train)               # :138
    h = _cbl(h, sd, prefix + "block7.", 3, train, slope=None)
    x = act(x + h, cfg.slope)                                   # :139
    x = x.reshape(x.size(0), -1)                                # :140
    x = F.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])
    mu, logvar = x.chunk(2, dim=1)                              # :142
    return mu, logvar


def fc_decode(z: Tensor, sd: State, cfg: FcCfg, train: bool, prefix: str = "decoder.") -> Tensor:
    """ResNetDecoder.forward -- models/mymodel.py:217-230."""
    def up(y, b):                                               # conv, BN, act, Upsample(2), conv, BN, act
        y = _cbl(y, sd, prefix + b, 0, train)
        y = F.interpolate(y, scale_factor=2, mode="nearest")
        return _cbl(y, sd, prefix + b, 4, train)

    def res(y, b):                                              # LeakyReLU(y + [conv, BN, act, conv, BN](y))
        h = _cbl(y, sd, prefix + b, 0, train)
        h = _cbl(h, sd, prefix + b, 3, train, slope=None)
        return act(y + h, cfg.slope)

    y = z.reshape(z.size(0), -1)
    y = F.relu(F.linear(y, sd[prefix + "dfc.0.weight"], sd[prefix + "dfc.0.bias"]))   # :219
    y = y.reshape(y.size(0), cfg.ch[3], 5, 6, 5)                # :220
    y = res(y, "block1.")                                       # :221-222
    y = up(y, "block2u.")
    y = res(y, "block3.")                                       # :224-225
    y = up(y, "block4u.")
    y = up(y, "block5u.")
    y = up(y, "block6u.")
    return F.relu(F.conv3d(y, sd[prefix + "last_block.0.weight"], sd[prefix + "last_block.0.bias"], 1, 1))


def encode(x: Tensor, sd: State, cfg, draws: Optional[Draws], train: bool,
           prefix: str = "encoder.") -> Tuple[Tensor, Tensor]:
    """VAEResNetEncoder.forward -- models/models.py:83-108, 213-223 (``FcCfg`` selects
    models/mymodel.py's encoder instead)."""
    if isinstance(cfg, FcCfg):
        return fc_encode(x, sd, cfg, train, prefix)
    p = prefix + "blocks.0."
    h = F.conv3d(x, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1)
    h = batchnorm(h, sd, p + "1.", train)
    h = act(h, cfg.slope)
    h = dropout(h, cfg.p_enc_stem, draws, train)
    for i, (cin, cout, s) in enumerate(cfg.enc_blocks()):
        h = building_block(h, sd, f"{prefix}blocks.{i + 1}.0.", cin, cout, s, cfg.slope, train)
    mu = F.conv3d(h, sd[prefix + "mu.weight"], sd[prefix + "mu.bias"])
    logvar = F.conv3d(h, sd[prefix + "var.weight"], sd[prefix + "var.bias"])
    return mu, logvar


def decode(z: Tensor, sd: State, cfg, draws: Optional[Draws], train: bool,
           prefix: str = "decoder.") -> Tensor:
    """ResNetDecoder.forward -- models/models.py:110-145 (``FcCfg``: models/mymodel.py's)."""
    if isinstance(cfg, FcCfg):
        return fc_decode(z, sd, cfg, train, prefix)
    p = prefix + "blocks.0."
    h = F.conv3d(z, sd[p + "0.weight"], sd[p + "0.bias"])
    h = batchnorm(h, sd, p + "1.", train)
    h = act(h, cfg.slope)
    h = dropout(h, cfg.p_dec_stem, draws, train)
    blocks = cfg.dec_blocks()
    for i, (cin, cout, s) in enumerate(blocks):
        h = upsample_block(h, sd, f"{prefix}blocks.{i + 1}.0.", cin, cout, s, cfg.slope, train)
    p = f"{prefix}blocks.{len(blocks) + 1}."
    h = F.conv3d(h, sd[p + "0.weight"], sd[p + "0.bias"], 1, 1)
    h = F.relu(h)                                      # models/models.py:139
    h = dropout(h, cfg.p_dec_out, draws, train)        # models/models.py:140
    return h


def reparameterize(mu: Tensor, logvar: Tensor, draws: Optional[Draws], val_flag: bool = False) -> Tensor:
    """SoftIntroVAE.reparameterize -- models/models.py:263-271.  eps = 0.1 when val_flag."""
    std = torch.exp(0.5 * logvar)
    if val_flag:
        return mu + 0.1 * std
    eps = draws.eps(std.shape)
    return mu + eps * std


def forward(x: Tensor, sd: State, cfg: Cfg, draws: Optional[Draws], train: bool,
            val_flag: bool = False):
    """SoftIntroVAE.forward -> (mu, logvar, z, x_re) -- models/models.py:273-277.
    Draw order: encoder dropout, reparam eps, decoder dropouts."""
    mu, logvar = encode(x, sd, cfg, draws, train)
    z = reparameterize(mu, logvar, draws, val_flag)
    return mu, logvar, z, decode(z, sd, cfg, draws, train)


# --------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------
def calc_kl(logvar: Tensor, mu: Tensor, reduce: str = "mean") -> Tensor:
    """utils/my_trainer.py:38-48 (== models/lossf.py:14-18 for 'mean')."""
    b = mu.size(0)
    mu = mu.reshape(b, -1)
    logvar = logvar.reshape(b, -1)
    kl = -0.5 * torch.sum(1 + logvar - mu ** 2 - logvar.exp(), dim=1)
    if reduce == "mean":
        return kl.mean(dim=0)
    if reduce == "sum":
        return kl.sum()
    return kl


def calc_reconstruction_loss(x: Tensor, recon_x: Tensor, reduction: str = "None") -> Tensor:
    """utils/my_trainer.py:62-78: per-sample SUM of squared error; 'mean' averages over the batch."""
    b = x.size(0)
    se = ((x.reshape(b, -1) - recon_x.reshape(b, -1)) ** 2).sum(dim=1)
    return se.mean(dim=0) if reduction == "mean" else se


def normal_loss(x_hat: Tensor, mu: Tensor, logvar: Tensor, x: Tensor, msew=1, kldw=10):
    """models/lossf.py:20-24."""
    mse = calc_reconstruction_loss(x, x_hat, "mean") * msew
    kld = calc_kl(logvar, mu, "mean") * kldw
    return mse + kld, mse, kld


# --------------------------------------------------------------------------
# optimiser
# --------------------------------------------------------------------------
def adam_step(params: Iterable[Tensor], grads: Iterable[Optional[Tensor]], state: Dict[int, dict],
              lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults as constructed at utils/my_trainer.py:183-184
    (no weight decay, no amsgrad); parameters whose grad is None are skipped
    and their step counter does not advance."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        st = state.setdefault(i, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
        st["step"] += 1
        t = st["step"]
        st["m"].mul_(beta1).add_(g, alpha=1 - beta1)
        st["v"].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        bc1 = 1 - beta1 ** t
        bc2 = 1 - beta2 ** t
        denom = (st["v"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st["m"], denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# the Soft-IntroVAE E-step / D-step
# --------------------------------------------------------------------------
def _is_param(k: str) -> bool:
    return not (k.endswith("running_mean") or k.endswith("running_var")
                or k.endswith("num_batches_tracked"))


def param_keys(sd: State, prefix: str) -> List[str]:
    return [k for k in sd if k.startswith(prefix) and _is_param(k)]


class Hyper:
    def __init__(self, beta_rec=1.0, beta_neg=1024.0, beta_kl=0.75, gamma_r=1e-8,
                 scale=None, lr=2e-4, dims=(80, 96, 80)):
        self.beta_rec, self.beta_neg, self.beta_kl = beta_rec, beta_neg, beta_kl
        self.gamma_r = gamma_r                                   # utils/my_trainer.py:193
        d, h, w = dims
        self.scale = scale if scale is not None else 8.0 / (d * h * w)  # utils/my_trainer.py:194
        self.lr = lr                                             # utils/my_trainer.py:183-184


def soft_intro_step(sd: State, cfg: Cfg, real: Tensor, noise: Tensor, draws: Draws, hp: Hyper,
                    adam_e: Dict[int, dict], adam_d: Dict[int, dict]) -> Dict[str, object]:
    """One training iteration of train_soft_intro_vae -- utils/my_trainer.py:242-325.

    ``sd`` is updated in place (parameters after both Adam steps, BN running
    statistics after all 13 half-network passes).  Returns every scalar the
    reference logs plus the gradients seen by each optimiser.

    Draw order (one entry of ``draws`` each): E-step: decode(noise) [2 dropouts],
    encode(real) [1], eps(z), decode(z) [2], forward(rec) [1, eps, 2],
    forward(fake) [1, eps, 2]; D-step: decode(noise) [2], decode(z) [2],
    encode(rec) [1], eps, encode(fake) [1], eps, decode(z_rec) [2], decode(z_fake) [2]."""
    ek, dk = param_keys(sd, "encoder."), param_keys(sd, "decoder.")
    s, br, bn, bk = hp.scale, hp.beta_rec, hp.beta_neg, hp.beta_kl

    # ================= Update E (my_trainer.py:242-289) =================
    for k in ek:
        sd[k] = sd[k].detach().requires_grad_(True)
    for k in dk:
        sd[k] = sd[k].detach().requires_grad_(False)
    fake = decode(noise, sd, cfg, draws, True)                               # :248
    real_mu, real_logvar = encode(real, sd, cfg, draws, True)                # :250
    z = reparameterize(real_mu, real_logvar, draws)                          # :251
    rec = decode(z, sd, cfg, draws, True)                                    # :252
    loss_rec = calc_reconstruction_loss(real, rec, "mean")                   # :260
    lossE_real_kl = calc_kl(real_logvar, real_mu, "mean")                    # :261
    rec_mu, rec_logvar, z_rec, rec_rec = forward(rec.detach(), sd, cfg, draws, True)      # :266
    fake_mu, fake_logvar, z_fake, rec_fake = forward(fake.detach(), sd, cfg, draws, True)  # :267
    fake_kl_e = calc_kl(fake_logvar, fake_mu, "none")                        # :270
    rec_kl_e = calc_kl(rec_logvar, rec_mu, "none")                           # :271
    loss_fake_rec = calc_reconstruction_loss(fake, rec_fake, "none")         # :274
    loss_rec_rec = calc_reconstruction_loss(rec, rec_rec, "none")            # :275 (rec NOT detached)
    exp_elbo_fake = (-2 * s * (br * loss_fake_rec + bn * fake_kl_e)).exp().mean()   # :278
    exp_elbo_rec = (-2 * s * (br * loss_rec_rec + bn * rec_kl_e)).exp().mean()      # :279
    lossE = s * (br * loss_rec + bk * lossE_real_kl) + 0.5 * (exp_elbo_fake + exp_elbo_rec)  # :282
    lossE = lossE * 10                                                       # :284
    gE = torch.autograd.grad(lossE, [sd[k] for k in ek], allow_unused=True)  # :286-287
    out = {
        "lossE": lossE.detach().clone(), "loss_rec_E": loss_rec.detach().clone(),
        "kl_real": lossE_real_kl.detach().clone(),
        "exp_elbo_fake": exp_elbo_fake.detach().clone(), "exp_elbo_rec": exp_elbo_rec.detach().clone(),
        "fake_kl_e": fake_kl_e.detach().clone(), "rec_kl_e": rec_kl_e.detach().clone(),
        "gradE": {k: (g.clone() if g is not None else None) for k, g in zip(ek, gE)},
    }
    z_det = z.detach()
    with torch.no_grad():
        for k in ek:
            sd[k] = sd[k].detach()
        adam_step([sd[k] for k in ek], gE, adam_e, hp.lr)                    # :288

    # ================= Update D (my_trainer.py:291-325) =================
    for k in ek:
        sd[k] = sd[k].detach().requires_grad_(False)
    for k in dk:
        sd[k] = sd[k].detach().requires_grad_(True)
    fake = decode(noise, sd, cfg, draws, True)                               # :297
    rec = decode(z_det, sd, cfg, draws, True)                                # :298
    loss_rec = calc_reconstruction_loss(real, rec, "mean")                   # :301
    rec_mu, rec_logvar = encode(rec, sd, cfg, draws, True)                   # :304
    z_rec = reparameterize(rec_mu, rec_logvar, draws)                        # :305
    fake_mu, fake_logvar = encode(fake, sd, cfg, draws, True)                # :307
    z_fake = reparameterize(fake_mu, fake_logvar, draws)                     # :308
    rec_rec = decode(z_rec.detach(), sd, cfg, draws, True)                   # :310
    rec_fake = decode(z_fake.detach(), sd, cfg, draws, True)                 # :311
    loss_rec_rec = calc_reconstruction_loss(rec.detach(), rec_rec, "mean")   # :313
    loss_fake_rec = calc_reconstruction_loss(fake.detach(), rec_fake, "mean")  # :314
    rec_kl = calc_kl(rec_logvar, rec_mu, "mean")                             # :316
    fake_kl = calc_kl(fake_logvar, fake_mu, "mean")                          # :317
    lossD = s * (br * loss_rec + 0.5 * bk * (rec_kl + fake_kl)
                 + hp.gamma_r * 0.5 * br * (loss_rec_rec + loss_fake_rec))   # :319
    lossD = lossD * 10                                                       # :321
    gD = torch.autograd.grad(lossD, [sd[k] for k in dk], allow_unused=True)  # :322-323
    out.update({
        "lossD": lossD.detach().clone(), "loss_rec_D": loss_rec.detach().clone(),
        "rec_kl": rec_kl.detach().clone(), "fake_kl": fake_kl.detach().clone(),
        "gradD": {k: (g.clone() if g is not None else None) for k, g in zip(dk, gD)},
    })
    with torch.no_grad():
        for k in dk:
            sd[k] = sd[k].detach()
        adam_step([sd[k] for k in dk], gD, adam_d, hp.lr)                    # :324
    for k in ek + dk:
        sd[k] = sd[k].detach()
    return out


def vae_step(sd: State, cfg: Cfg, x: Tensor, draws: Draws, mse_w: float, kl_w: float,
             lr: float, adam: Dict[int, dict]) -> Dict[str, object]:
    """One iteration of train_ResNetVAE -- utils/my_trainer.py:588-598, with
    models/vaemodel.py:226-230 (forward -> x_re, mu, logvar) and
    models/lossf.py:20-24."""
    keys = param_keys(sd, "encoder.") + param_keys(sd, "decoder.")
    for k in keys:
        sd[k] = sd[k].detach().requires_grad_(True)
    mu, logvar, _, x_re = forward(x, sd, cfg, draws, True)
    loss, mse, kld = normal_loss(x_re, mu, logvar, x, mse_w, kl_w)
    g = torch.autograd.grad(loss, [sd[k] for k in keys], allow_unused=True)
    with torch.no_grad():
        for k in keys:
            sd[k] = sd[k].detach()
        adam_step([sd[k] for k in keys], g, adam, lr)
    return {"loss": loss.detach(), "mse": mse.detach(), "kld": kld.detach(),
            "grad": {k: gi for k, gi in zip(keys, g)}, "x_re": x_re.detach(),
            "mu": mu.detach(), "logvar": logvar.detach()}


# --------------------------------------------------------------------------
# per-op restatements used by the kernel parity tests
# --------------------------------------------------------------------------
def conv3d_k3(x: Tensor, w: Tensor, b: Optional[Tensor] = None, upsample: int = 1) -> Tensor:
    """nn.Conv3d(k=3, s=1, p=1) (models/models.py:17), optionally preceded by
    nn.Upsample(scale_factor=upsample) nearest (models/models.py:58)."""
    if upsample != 1:
        x = F.interpolate(x, scale_factor=upsample, mode="nearest")
    return F.conv3d(x, w, b, 1, 1)


def bn_act(y: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor,
           slope: float, pool: int = 1, residual: Optional[Tensor] = None,
           keep: Optional[Tensor] = None, p: float = 0.0, train: bool = True) -> Tensor:
    """BN -> [+residual] -> act -> AvgPool(pool) -> dropout(keep, p): the
    elementwise tail that follows each conv in models/models.py:16-22,39-43,91-96."""
    h = F.batch_norm(y, running_mean, running_var, gamma, beta, train, BN_MOMENTUM, BN_EPS)
    if residual is not None:
        h = h + residual
    h = act(h, slope)
    if pool != 1:
        h = F.avg_pool3d(h, pool)
    if keep is not None:
        h = h * keep * (1.0 / (1.0 - p))
    return h


# --------------------------------------------------------------------------
# input pipeline
# --------------------------------------------------------------------------
def preprocess(voxel, cut_range: float = 4.0):
    """BrainDataset._preprocess + normalize -- utils/data_load.py:25-30, 36-37 (numpy, as the
    reference): clip to [0, cut_range * std], min-max to [0, 1], add the channel axis, float32."""
    import numpy as np
    voxel = np.clip(voxel, 0, cut_range * np.std(voxel))
    lo, hi = np.min(voxel), np.max(voxel)
    voxel = (voxel - lo) / (hi - lo)
    return voxel[np.newaxis, ].astype("f")


def flip_shift(vol, flips, shifts):
    """Definition of the flip/shift augmentation (no reference counterpart runs here: the
    reference's augmentation is torchio's RandomAffine, aug-z-1200main.py:106-119, and torchio
    is not installed -- parity unpinned).  vol: (D, H, W) numpy; each flagged axis is reversed,
    then the volume is shifted by `shifts` voxels toward larger indices with zero fill."""
    import numpy as np
    out = vol
    for ax, f in enumerate(flips):
        if f:
            out = np.flip(out, axis=ax)
    res = np.zeros_like(out)
    src, dst = [], []
    for ax, sft in enumerate(shifts):
        n = out.shape[ax]
        if abs(sft) >= n:
            return res
        src.append(slice(max(0, -sft), min(n, n - sft)))
        dst.append(slice(max(0, sft), min(n, n + sft)))
    res[tuple(dst)] = out[tuple(src)]
    return res


def affine_resample(vol, mat, pad="minimum"):
    """Definition of the trilinear affine resampling the device augmentation performs (the
    reference's augmentation is torchio's RandomAffine(degrees=10), aug-z-1200main.py:106-119,
    with SimpleITK doing the resampling; neither is installed here -- parity with them is
    unpinned; this restatement is pinned against scipy.ndimage.affine_transform(order=1,
    mode='grid-constant') in the tests).  vol: (D, H, W) numpy; mat: (3, 4), output index ->
    input index; each of the 8 neighbours outside the volume contributes `pad` (a number, or
    "minimum" for vol.min()).  Computed in float64."""
    import numpy as np
    v = np.asarray(vol, dtype=np.float64)
    cval = float(v.min()) if isinstance(pad, str) else float(pad)
    a = np.asarray(mat, dtype=np.float64).reshape(3, 4)
    dims = v.shape
    grid = np.stack(np.meshgrid(*[np.arange(n, dtype=np.float64) for n in dims], indexing="ij"), 0)
    pts = np.tensordot(a[:, :3], grid, axes=1) + a[:, 3].reshape(3, 1, 1, 1)
    base = np.floor(pts)
    frac = pts - base
    base = base.astype(np.int64)
    out = np.zeros(dims, dtype=np.float64)
    for c in range(8):
        off = ((c >> 2) & 1, (c >> 1) & 1, c & 1)
        wgt = np.ones(dims, dtype=np.float64)
        ok = np.ones(dims, dtype=bool)
        idx = []
        for ax in range(3):
            wgt = wgt * (frac[ax] if off[ax] else 1.0 - frac[ax])
            i = base[ax] + off[ax]
            ok &= (i >= 0) & (i < dims[ax])
            idx.append(np.clip(i, 0, dims[ax] - 1))
        out += wgt * np.where(ok, v[idx[0], idx[1], idx[2]], cval)
    return out
