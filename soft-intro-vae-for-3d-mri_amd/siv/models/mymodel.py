"""Drop-in for the reference's ``models/mymodel.py``: the FC-latent Soft-IntroVAE
(``SoftIntroVAE(first_ch, second_ch, third_ch, forth_ch, z_ch)``, z-1200main.py's 1200-d
latent when built as (32, 64, 128, 128, 600) / used with ``utils/trainer_fc.py``).

Attribute names, ``nn.Sequential`` positions and ``state_dict`` keys are the reference's
(``encoder.block1.0.weight`` ... ``encoder.fc.weight``, ``decoder.dfc.0.weight`` ...
``decoder.last_block.0.bias``), including the encoder's ``block8`` that the reference
builds but never calls (models/mymodel.py:108-116).  The modules hold parameters only; the
forward passes below drive the fused HIP operators:

* conv -> BN -> LeakyReLU [-> AvgPool3d(2)]        = conv3d_k3 + bn_act(pool)
* nearest Upsample(2) -> conv                      = conv3d_k3(ups=2)
* conv -> BN, then ``LeakyReLU(x + h)``            = conv3d_k3 + bn_act(residual=x)
* ``LeakyReLU(x + h)`` after an activated block    = add_act
* ``view(B, -1)`` / ``view(B, C, 5, 6, 5)``         = channel-major reshapes (the reference's
  NCDHW flatten order; the fc weights keep the reference's column order)
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import activations, ops
from ._core import sample_eps

SLOPE = 0.2          # every nn.LeakyReLU(0.2) of models/mymodel.py
LATENT_DIMS = (5, 6, 5)   # models/mymodel.py:125,151,220 (80x96x80 after four AvgPool3d(2))


def _cbl(cin, cout):
    """conv3(bias) + BN + LeakyReLU, the repeating triple of models/mymodel.py:55-116."""
    return [nn.Conv3d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True),
            nn.BatchNorm3d(cout), nn.LeakyReLU(SLOPE, inplace=True)]


def _conv_bn_act(x, conv, bn, train, pool=1, ups=1, residual=None):
    return ops.conv_bn_act(x, conv, bn, SLOPE, ups=ups, pool=pool, residual=residual, train=train)


class BaseEncoder(nn.Module):
    pass


class BaseDecoder(nn.Module):
    pass


class ResNetVAEencoder(nn.Module):
    """models/mymodel.py:51-143."""

    def __init__(self, first_ch, second_ch, third_ch, forth_ch, z_ch):
        super().__init__()
        self.forth_ch = forth_ch
        self.block1 = nn.Sequential(*_cbl(1, first_ch), *_cbl(first_ch, first_ch))
        self.block2 = nn.Sequential(*_cbl(first_ch, first_ch), *_cbl(first_ch, second_ch))
        self.block3 = nn.Sequential(*_cbl(second_ch, second_ch), *_cbl(second_ch, third_ch))
        self.block4short = nn.Sequential(*_cbl(third_ch, third_ch))
        self.block5 = nn.Sequential(*_cbl(third_ch, third_ch))
        self.block6 = nn.Sequential(*_cbl(third_ch, third_ch), nn.AvgPool3d(kernel_size=2),
                                    *_cbl(third_ch, forth_ch))
        self.block7 = nn.Sequential(*_cbl(forth_ch, forth_ch), *_cbl(forth_ch, forth_ch)[:2])
        self.block8 = nn.Sequential(*_cbl(third_ch, third_ch), *_cbl(third_ch, forth_ch))   # never called
        self.pool1 = nn.AvgPool3d(kernel_size=2)
        self.pool2 = nn.AvgPool3d(kernel_size=2)
        self.pool3 = nn.AvgPool3d(kernel_size=2)
        self.pool4 = nn.AvgPool3d(kernel_size=2)
        self.Leakyrelu5 = nn.LeakyReLU(SLOPE, inplace=True)
        self.Leakyrelu7 = nn.LeakyReLU(SLOPE, inplace=True)
        self.fc = nn.Linear(forth_ch * 5 * 6 * 5, z_ch * 2)

    @activations.scoped
    def forward(self, x):
        t = self.training
        for blk in (self.block1, self.block2, self.block3):         # :128-133, pool fused into the tail
            x = _conv_bn_act(x, blk[0], blk[1], t)
            x = _conv_bn_act(x, blk[3], blk[4], t, pool=2)
        x = _conv_bn_act(x, self.block4short[0], self.block4short[1], t)
        h = _conv_bn_act(x, self.block5[0], self.block5[1], t)
        x = ops.add_act(x, h, SLOPE)                                 # :136
        b6 = self.block6
        x = _conv_bn_act(x, b6[0], b6[1], t, pool=2)
        x = _conv_bn_act(x, b6[4], b6[5], t)
        b7 = self.block7
        h = _conv_bn_act(x, b7[0], b7[1], t)
        x = _conv_bn_act(h, b7[3], b7[4], t, residual=x)             # :138-139
        x = x.contiguous().view(x.size(0), -1)                       # :140, (c, d, h, w) order
        x = ops.linear(x, self.fc.weight, self.fc.bias)
        mu, logvar = x.chunk(2, dim=1)
        return mu, logvar


class ResNetDecoder(nn.Module):
    """models/mymodel.py:146-230."""

    def __init__(self, first_ch, second_ch, third_ch, forth_ch, z_ch):
        super().__init__()
        self.forth_ch = forth_ch
        self.dfc = nn.Sequential(nn.Linear(z_ch, forth_ch * 150), nn.ReLU(True))
        ups = lambda: nn.Upsample(scale_factor=2, mode="nearest")
        self.block1 = nn.Sequential(*_cbl(forth_ch, forth_ch), *_cbl(forth_ch, forth_ch)[:2])
        self.block2u = nn.Sequential(*_cbl(forth_ch, forth_ch), ups(), *_cbl(forth_ch, third_ch))
        self.block3 = nn.Sequential(*_cbl(third_ch, third_ch), *_cbl(third_ch, third_ch)[:2])
        self.block4u = nn.Sequential(*_cbl(third_ch, third_ch), ups(), *_cbl(third_ch, second_ch))
        self.block5u = nn.Sequential(*_cbl(second_ch, second_ch), ups(), *_cbl(second_ch, first_ch))
        self.block6u = nn.Sequential(*_cbl(first_ch, first_ch), ups(), *_cbl(first_ch, first_ch))
        self.last_block = nn.Sequential(
            nn.Conv3d(first_ch, 1, kernel_size=3, stride=1, padding=1, bias=True), nn.ReLU())
        self.dLeakyrelu1 = nn.LeakyReLU(SLOPE, inplace=True)
        self.dLeakyrelu2 = nn.LeakyReLU(SLOPE, inplace=True)

    @staticmethod
    def _res_block(y, blk, t):
        h = _conv_bn_act(y, blk[0], blk[1], t)
        return _conv_bn_act(h, blk[3], blk[4], t, residual=y)        # LeakyReLU(y + h), :221-225

    @staticmethod
    def _up_block(y, blk, t):
        y = _conv_bn_act(y, blk[0], blk[1], t)
        return _conv_bn_act(y, blk[4], blk[5], t, ups=2)

    @activations.scoped
    def forward(self, z):
        t = self.training
        y = z.reshape(z.size(0), -1)
        fc = self.dfc[0]
        y = ops.act_dropout(ops.linear(y, fc.weight, fc.bias), 0.0)  # nn.ReLU(True)
        y = y.view(y.size(0), self.forth_ch, *LATENT_DIMS)
        y = self._res_block(y, self.block1, t)
        y = self._up_block(y, self.block2u, t)
        y = self._res_block(y, self.block3, t)
        y = self._up_block(y, self.block4u, t)
        y = self._up_block(y, self.block5u, t)
        y = self._up_block(y, self.block6u, t)
        c = self.last_block[0]
        return ops.act_dropout(ops.conv3d_k3(y, c.weight, c.bias), 0.0)


class BaseVAE(nn.Module):
    """models/mymodel.py:29-48."""

    def __init__(self) -> None:
        super().__init__()
        self.encoder = BaseEncoder()
        self.decoder = BaseDecoder()

    def encode(self, x):
        return self.encoder(x)

    def decode(self, vec):
        return self.decoder(vec)

    def reparameterize(self, mu, logvar) -> torch.Tensor:
        return ops.reparameterize(mu, logvar, sample_eps(mu))

    def forward(self, x):
        mu, logvar = self.encode(x)
        vec = self.reparameterize(mu, logvar)
        return self.decode(vec), vec, mu, logvar


class ResNetVAE(BaseVAE):
    """models/mymodel.py:233-248: forward returns (x_re, mu, logvar)."""

    def __init__(self, first_ch, second_ch, third_ch, forth_ch, z_ch) -> None:
        super().__init__()
        self.encoder = ResNetVAEencoder(first_ch, second_ch, third_ch, forth_ch, z_ch)
        self.decoder = ResNetDecoder(first_ch, second_ch, third_ch, forth_ch, z_ch)

    def reparamenterize(self, mu, logvar):
        return ops.reparameterize(mu, logvar, sample_eps(mu))

    def forward(self, x):
        mu, logvar = self.encoder(x)
        z = self.reparamenterize(mu, logvar)
        return self.decoder(z), mu, logvar


class SoftIntroVAE(nn.Module):
    """models/mymodel.py:256-290."""

    def __init__(self, first_ch, second_ch, third_ch, forth_ch, z_ch) -> None:
        super().__init__()
        self.encoder = ResNetVAEencoder(first_ch, second_ch, third_ch, forth_ch, z_ch)
        self.decoder = ResNetDecoder(first_ch, second_ch, third_ch, forth_ch, z_ch)
        self.z_ch = z_ch

    def reparameterize(self, mu, logvar):
        return ops.reparameterize(mu, logvar, sample_eps(mu))

    def forward(self, x):
        mu, logvar = self.encoder(x)
        z = self.reparameterize(mu, logvar)
        return mu, logvar, z, self.decoder(z)

    def encode(self, x, o_cond=None):
        return self.encoder(x)

    def decode(self, z, y_cond=None):
        return self.decoder(z)

    def latent_shape(self, batch: int, dims=None):
        """Shape of the latent noise, ``(b_size, model.z_ch)`` of utils/trainer_fc.py:218."""
        return (batch, self.z_ch)
