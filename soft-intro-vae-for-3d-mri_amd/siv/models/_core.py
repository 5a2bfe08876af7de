"""3D ResNet encoder/decoder stacks whose forward and backward run in libsiv.

Module tree, attribute names and ``state_dict`` keys are those of the reference
(models/models.py and models/vaemodel.py): ``nn.Conv3d`` / ``nn.BatchNorm3d`` instances
hold the parameters and buffers at the same positions inside the same ``nn.Sequential``
containers, so reference checkpoints load both ways and ``model.apply(fn)`` with the
reference's type checks (utils/my_trainer.py:511-519) behaves identically.  Their own
``forward`` is never called: each block's ``forward`` drives the fused HIP operators
(conv -> [BN + residual + activation + pool + dropout] as two launches per layer).

A *variant* selects the activation slope and dropout rates:
``SINTRO`` = models/models.py (LeakyReLU(0.2), Dropout .35/.25/.35),
``PLAIN``  = models/vaemodel.py (ReLU, no dropout).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from .. import activations, ops, rng


@dataclass(frozen=True)
class Variant:
    slope: float
    p_enc_stem: float
    p_dec_stem: float
    p_dec_out: float


SINTRO = Variant(0.2, 0.35, 0.25, 0.35)   # models/models.py:15,19,94-95,121-122,139-140
PLAIN = Variant(0.0, 0.0, 0.0, 0.0)       # models/vaemodel.py:14,18,90,113,129

# Paired passes.  The Soft-IntroVAE iteration sends two different batches through the same weights four times
# (utils/my_trainer.py:266-267, 297-298, 304-311).  A pair runs the layers whose volumes are at most this many
# voxels per sample (20x24x20 and below: launches that do not fill the chip at batch 8) as ONE batch with
# BatchNorm statistics per member; the larger layers, the fused single-channel stem and the output layer run per
# member as before.
PAIR_MAX_VOXELS = 20 * 24 * 20


def _act_module(v: Variant) -> nn.Module:
    return nn.LeakyReLU(v.slope, inplace=True) if v.slope != 0.0 else nn.ReLU(inplace=True)


def _keep(shape, p: float, training: bool, device, stored: bool = True):
    """The dropout mask of one call site; ``stored=False`` asks for the key of the mask instead of the
    tensor (consumers that draw it themselves)."""
    if not training or p == 0.0:
        return None
    r = rng.get_rng()
    return r.keep_mask(shape, p, device) if stored else r.keep_mask_key(shape, p, device)


def _shortcut(in_ch: int, out_ch: int):
    # models/models.py:28-35: identity, or a biased 1x1x1 projection when the widths differ
    return nn.Conv3d(in_ch, out_ch, kernel_size=1, stride=1, padding=0) if in_ch != out_ch else None


class _ResBlockBase(nn.Module):
    """Shared by BuildingBlock / UpsampleBuildingkBlock: ``self.block`` is the reference's
    six-entry Sequential (conv, BN, act, pool|upsample, conv, BN)."""

    def _res(self, x):
        if not self.res:
            return None
        sc = self._modules.get("shortcut")
        return x if sc is None else ops.conv1x1(x, sc.weight, sc.bias)


class BuildingBlock(_ResBlockBase):
    """models/models.py:8-43.  conv3 -> BN -> act -> AvgPool(stride) -> conv3 -> BN ->
    [+ shortcut(x) when stride == 1] -> act.  Stride-2 blocks never use their shortcut but
    still own its parameters when in_ch != out_ch, as in the reference."""

    def __init__(self, in_ch, out_ch, stride, bias=False, variant: Variant = SINTRO):
        super().__init__()
        self.res = stride == 1
        self.stride = stride
        self.variant = variant
        sc = _shortcut(in_ch, out_ch)
        if sc is not None:
            self.shortcut = sc
        self.block = nn.Sequential(
            nn.Conv3d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=bias),
            nn.BatchNorm3d(out_ch),
            _act_module(variant),
            nn.AvgPool3d(kernel_size=stride),
            nn.Conv3d(out_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=bias),
            nn.BatchNorm3d(out_ch),
        )

    def forward(self, x, groups: int = 1, out=None):
        """``groups`` > 1: ``x`` is that many passes of the reference concatenated on the batch axis
        (BatchNorm statistics per part).  ``out``: destination of the block's output (a dense tensor of its
        shape, e.g. a batch slice of the buffer a paired pass merges its members in)."""
        c1, b1, _, _, c2, b2 = self.block
        s = self.variant.slope
        h = ops.conv_bn_act(x, c1, b1, s, pool=self.stride, train=self.training, groups=groups)
        return ops.conv_bn_act(h, c2, b2, s, residual=self._res(x), train=self.training, groups=groups, out=out)


class UpsampleBuildingkBlock(_ResBlockBase):
    """models/models.py:46-80.  conv3(in->in) -> BN -> act -> nearest Upsample(stride) ->
    conv3(in->out) -> BN -> [+ shortcut(x)] -> act.  The upsample is folded into the second
    convolution's input gather; no upsampled tensor is materialised."""

    def __init__(self, in_ch, out_ch, stride, bias=False, variant: Variant = SINTRO):
        super().__init__()
        self.res = stride == 1
        self.stride = stride
        self.variant = variant
        sc = _shortcut(in_ch, out_ch)
        if sc is not None:
            self.shortcut = sc
        self.block = nn.Sequential(
            nn.Conv3d(in_ch, in_ch, kernel_size=3, stride=1, padding=1, bias=bias),
            nn.BatchNorm3d(in_ch),
            _act_module(variant),
            nn.Upsample(scale_factor=stride),
            nn.Conv3d(in_ch, out_ch, kernel_size=3, stride=1, padding=1, bias=bias),
            nn.BatchNorm3d(out_ch),
        )

    def forward(self, x, groups: int = 1):
        c1, b1, _, _, c2, b2 = self.block
        s = self.variant.slope
        if self.stride not in (1, 2):
            raise ValueError(f"only stride 1 or 2 is supported (got {self.stride})")
        h = ops.conv_bn_act(x, c1, b1, s, train=self.training, groups=groups)
        return ops.conv_bn_act(h, c2, b2, s, ups=self.stride, residual=self._res(x), train=self.training,
                               groups=groups)

    def forward_unnormalised(self, x):
        """The block up to and including its second convolution: (raw output, batch statistics or None).
        For a consumer that applies this block's closing BatchNorm + activation itself (no residual)."""
        c1, b1, _, _, c2, b2 = self.block
        if self.res or self.stride not in (1, 2):
            raise ValueError("only for blocks without a residual connection")
        h = ops.conv_bn_act(x, c1, b1, self.variant.slope, train=self.training)
        return ops.conv_with_stats(h, c2, b2, ups=self.stride, train=self.training)


class ResNetEncoder(nn.Module):
    """models/models.py:83-108: stem conv3(1->in_ch)+BN+act(+Dropout), BuildingBlock stack,
    and the (unused by the VAE heads) ``conv`` 1x1x1 projection."""

    def __init__(self, in_ch, block_setting, variant: Variant = SINTRO):
        super().__init__()
        self.block_setting = block_setting
        self.in_ch = in_ch
        self.variant = variant
        stem = [nn.Conv3d(1, in_ch, kernel_size=3, stride=1, padding=1, bias=True),
                nn.BatchNorm3d(in_ch), _act_module(variant)]
        if variant.p_enc_stem > 0:
            stem.append(nn.Dropout(p=variant.p_enc_stem))
        blocks = [nn.Sequential(*stem)]
        for c, n, s in block_setting:
            for i in range(n):
                blocks.append(nn.Sequential(BuildingBlock(in_ch, c, s if i == 0 else 1, variant=variant)))
                in_ch = c
        self.inner_ch = in_ch
        self.blocks = nn.Sequential(*blocks)
        self.conv = nn.Sequential(nn.Conv3d(in_ch, 1, kernel_size=1, stride=1, bias=True))

    def _stem(self, x):
        v = self.variant
        conv, bn = self.blocks[0][0], self.blocks[0][1]
        fused = ops.conv1_bn_act_supported(x, conv)  # the stem's convolution output and mask are never stored
        keep = _keep((x.shape[0], conv.out_channels) + tuple(x.shape[2:]), v.p_enc_stem, self.training, x.device,
                     stored=not fused)
        if fused:
            return ops.conv1_bn_act(x, conv, bn, v.slope, keep=keep, p=v.p_enc_stem, train=self.training)
        return ops.conv_bn_act(x, conv, bn, v.slope, keep=keep, p=v.p_enc_stem, train=self.training)

    def features(self, x):
        h = self._stem(x)
        for blk in list(self.blocks)[1:]:
            h = blk[0](h)
        return h

    def n_draws(self) -> int:
        """Random draws of one pass (the stem's dropout mask)."""
        return int(self.training and self.variant.p_enc_stem > 0)

    def features_pair(self, xs, cur: "rng.PairedDraws"):
        """``features`` of every member of ``xs`` (same shapes), returned concatenated on the batch axis: the
        stem and the blocks on large volumes run per member, the blocks from ``PAIR_MAX_VOXELS`` voxels per
        sample down run once on the merged batch with BatchNorm statistics per member."""
        blocks = [b[0] for b in list(self.blocks)[1:]]
        vox = 1
        for d in xs[0].shape[2:]:
            vox *= int(d)
        first_merged = len(blocks)
        for i, blk in enumerate(blocks):
            if vox <= PAIR_MAX_VOXELS:
                first_merged = i
                break
            vox //= blk.stride ** 3
        hs, into = [], None
        for g, x in enumerate(xs):
            with cur.member(g):
                h = self._stem(x)
                for blk in blocks[:max(first_merged - 1, 0)]:
                    h = blk(h)
                if first_merged > 0:
                    # the last per-member block writes straight into its slice of the merged batch
                    blk = blocks[first_merged - 1]
                    if into is None:
                        n, d = h.shape[0], [int(v) // blk.stride for v in h.shape[2:]]
                        into = torch.empty((n * len(xs), blk.block[4].out_channels, *d), device=h.device,
                                           dtype=torch.float32, memory_format=torch.channels_last_3d)
                    h = blk(h, out=None if into is None else into[g * h.shape[0]:(g + 1) * h.shape[0]])
            hs.append(h)
        h = ops.merge_members(hs, into)
        for blk in blocks[first_merged:]:
            h = blk(h, groups=len(xs))
        return h

    @activations.scoped
    def forward(self, x):
        c = self.conv[0]
        return ops.conv1x1(self.features(x), c.weight, c.bias)


class VAEResNetEncoder(ResNetEncoder):
    """models/models.py:213-223: two 1x1x1 heads ``mu`` and ``var`` (the latter is log-variance)."""

    def __init__(self, in_ch, block_setting, variant: Variant = SINTRO):
        super().__init__(in_ch, block_setting, variant)
        self.mu = nn.Conv3d(self.inner_ch, 1, kernel_size=1, stride=1, bias=True)
        self.var = nn.Conv3d(self.inner_ch, 1, kernel_size=1, stride=1, bias=True)

    @activations.scoped
    def forward(self, x):
        h = self.features(x)
        return ops.conv1x1(h, self.mu.weight, self.mu.bias), ops.conv1x1(h, self.var.weight, self.var.bias)

    @activations.scoped
    def forward_pair(self, xs, cur):
        """(mu, logvar) of all members of ``xs`` concatenated on the batch axis (see ``features_pair``)."""
        h = self.features_pair(xs, cur)
        return ops.conv1x1(h, self.mu.weight, self.mu.bias), ops.conv1x1(h, self.var.weight, self.var.bias)


class ResNetDecoder(nn.Module):
    """models/models.py:110-145: 1x1x1 stem (1->last)+BN+act(+Dropout), mirrored upsample blocks,
    conv3(in_ch->1)+ReLU(+Dropout)."""

    def __init__(self, encoder: ResNetEncoder, blocks=None):
        super().__init__()
        v = self.variant = encoder.variant
        setting = encoder.block_setting
        last = setting[-1][0]
        if blocks is None:
            stem = [nn.Conv3d(1, last, 1, 1, bias=True), nn.BatchNorm3d(last), _act_module(v)]
            if v.p_dec_stem > 0:
                stem.append(nn.Dropout(p=v.p_dec_stem))
            blocks = [nn.Sequential(*stem)]
        in_ch = last
        rev = setting[::-1]
        for i, (c, n, s) in enumerate(rev):
            nc = encoder.in_ch if i == len(rev) - 1 else rev[i + 1][0]
            for j in range(n):
                is_last = j == n - 1
                out_ch = nc if is_last else c
                blocks.append(nn.Sequential(
                    UpsampleBuildingkBlock(in_ch, out_ch, s if is_last else 1, variant=v)))
                in_ch = out_ch
        tail = [nn.Conv3d(in_ch, 1, kernel_size=3, stride=1, padding=1, bias=True), nn.ReLU()]
        if v.p_dec_out > 0:
            tail.append(nn.Dropout(p=v.p_dec_out))
        blocks.append(nn.Sequential(*tail))
        self.blocks = nn.Sequential(*blocks)

    @activations.scoped
    def forward(self, z):
        v = self.variant
        mods = list(self.blocks)
        conv, bn = mods[0][0], mods[0][1]
        h = ops.conv1x1(z, conv.weight, conv.bias)
        keep = _keep(h.shape, v.p_dec_stem, self.training, h.device)
        h = ops.bn_act(h, bn, v.slope, keep=keep, p=v.p_dec_stem, train=self.training)
        for blk in mods[1:-2]:
            h = blk[0](h)
        return self._last_block_and_output(h, mods)

    def n_draws(self) -> int:
        """Random draws of one pass (stem and output dropout masks)."""
        v = self.variant
        return int(self.training and v.p_dec_stem > 0) + int(self.training and v.p_dec_out > 0)

    @activations.scoped
    def forward_pair(self, z, members: int, cur: "rng.PairedDraws"):
        """``forward`` of ``members`` latents concatenated in ``z``: the 1x1x1 stem and the blocks whose output
        has at most ``PAIR_MAX_VOXELS`` voxels per sample run once on the merged batch (BatchNorm statistics
        and dropout masks per member), the rest, the last block and the output layer per member.  Returns the
        list of the members' outputs."""
        v = self.variant
        mods = list(self.blocks)
        conv, bn = mods[0][0], mods[0][1]
        if z.shape[0] % members:
            raise ValueError("the merged latent batch is not a multiple of the number of members")
        per = z.shape[0] // members
        h = ops.conv1x1(z, conv.weight, conv.bias)
        keeps = []
        for g in range(members):
            with cur.member(g):
                keeps.append(_keep((per,) + tuple(h.shape[1:]), v.p_dec_stem, self.training, h.device))
        keep = torch.cat(keeps) if keeps[0] is not None else None
        h = ops.bn_act(h, bn, v.slope, keep=keep, p=v.p_dec_stem, train=self.training, groups=members)
        body = [m[0] for m in mods[1:-2]]
        vox = 1
        for d in z.shape[2:]:
            vox *= int(d)
        n_merged = 0
        for blk in body:
            vox *= blk.stride ** 3
            if vox > PAIR_MAX_VOXELS:
                break
            n_merged += 1
        for blk in body[:n_merged]:
            h = blk(h, groups=members)
        outs = []
        parts = ops.split_members(h, members)
        for g in range(members):
            hg = parts[g]
            for blk in body[n_merged:]:
                hg = blk(hg)
            with cur.member(g):
                outs.append(self._last_block_and_output(hg, mods))
        return outs

    def _last_block_and_output(self, h, mods):
        v = self.variant
        last, conv = mods[-2][0], mods[-1][0]
        y = stats = None
        if len(mods) > 2 and isinstance(last, UpsampleBuildingkBlock) and not last.res and conv.out_channels == 1:
            # the last block's closing BatchNorm + activation can run inside the output convolution
            y, stats = last.forward_unnormalised(h)
            if not ops.bn_act_conv_to1_supported(y):
                h = ops.bn_act(y, last.block[5], v.slope, train=self.training, stats=stats)
                y = None
        elif len(mods) > 2:
            h = last(h)
        if y is not None:
            h = ops.bn_act_conv_to1(y, last.block[5], v.slope, conv, self.training, stats)
        else:
            h = ops.conv3d_k3(h, conv.weight, conv.bias)
        keep = _keep(h.shape, v.p_dec_out, self.training, h.device)
        return ops.act_dropout(h, 0.0, keep=keep, p=v.p_dec_out)   # nn.ReLU then nn.Dropout


def sample_eps(like: torch.Tensor) -> torch.Tensor:
    """The ``torch.randn_like(std)`` of models/models.py:204,234,266."""
    return rng.get_rng().eps(like.shape, like.device)


class ResNetVAE(nn.Module):
    """models/models.py:226-241 / models/vaemodel.py:215-230 (plus the BaseVAE helpers
    :191-210): forward returns (x_re, mu, logvar)."""

    variant = SINTRO

    def __init__(self, in_ch, block_setting) -> None:
        super().__init__()
        self.encoder = VAEResNetEncoder(in_ch, block_setting, self.variant)
        self.decoder = ResNetDecoder(self.encoder)

    def encode(self, x):
        return self.encoder(x)

    def decode(self, vec):
        return self.decoder(vec)

    def reparameterize(self, mu, logvar):
        return ops.reparameterize(mu, logvar, sample_eps(mu))

    reparamenterize = reparameterize   # the reference's spelling, models/models.py:232

    def forward(self, x):
        mu, logvar = self.encoder(x)
        z = self.reparameterize(mu, logvar)
        return self.decoder(z), mu, logvar


class SoftIntroVAE(nn.Module):
    """models/models.py:257-285."""

    variant = SINTRO

    def __init__(self, in_ch, block_setting) -> None:
        super().__init__()
        self.encoder = VAEResNetEncoder(in_ch, block_setting, self.variant)
        self.decoder = ResNetDecoder(self.encoder)

    def reparameterize(self, mu, logvar, val_flag=False):
        if val_flag is False:
            return ops.reparameterize(mu, logvar, sample_eps(mu))
        return ops.reparameterize(mu, logvar, None, 0.1)      # eps = 0.1, models/models.py:269

    def forward(self, x):
        mu, logvar = self.encoder(x)
        z = self.reparameterize(mu, logvar)
        return mu, logvar, z, self.decoder(z)

    def encode(self, x, o_cond=None):
        return self.encoder(x)

    def decode(self, z, y_cond=None):
        return self.decoder(z)

    # ---- two passes through the same weights as one batch ------------------------------------------------
    # The training iteration calls the network on two different batches in a row four times
    # (utils/my_trainer.py:266-267, 297-298, 304-311).  The *_pair methods compute exactly what the two calls
    # compute -- BatchNorm batch statistics and running-statistics updates per call, dropout masks and eps drawn
    # with the call indices the two calls would use -- but run the small deep layers once on the merged batch.
    # They return the merged (2B) latent tensors; member k owns rows [k*B, (k+1)*B).

    def pair_ready(self) -> bool:
        """Paired passes need a random source that addresses draws by call index, and training mode."""
        return self.training and rng.seekable()

    def decode_pair(self, za, zb=None):
        """[decode(za), decode(zb)]; with ``zb`` None, ``za`` is the two latents already concatenated."""
        z = za if zb is None else torch.cat([za, zb])
        cur = rng.PairedDraws(self.decoder.n_draws())
        outs = self.decoder.forward_pair(z, 2, cur)
        cur.close()
        return outs

    def encode_reparam_pair(self, xa, xb):
        """``mu, logvar = encode(x); z = reparameterize(mu, logvar)`` for ``xa`` then ``xb`` (:304-308):
        returns merged (mu, logvar, z)."""
        cur = rng.PairedDraws(self.encoder.n_draws() + 1)
        mu, logvar = self.encoder.forward_pair([xa, xb], cur)
        z = self._reparam_pair(mu, logvar, cur)
        cur.close()
        return mu, logvar, z

    def forward_pair(self, xa, xb):
        """``forward(xa)`` then ``forward(xb)`` (:266-267): returns merged (mu, logvar, z) and the list of
        the two reconstructions."""
        cur = rng.PairedDraws(self.encoder.n_draws() + 1 + self.decoder.n_draws())
        mu, logvar = self.encoder.forward_pair([xa, xb], cur)
        z = self._reparam_pair(mu, logvar, cur)
        outs = self.decoder.forward_pair(z, 2, cur)
        cur.close()
        return mu, logvar, z, outs

    def _reparam_pair(self, mu, logvar, cur):
        per = mu.shape[0] // 2
        eps = []
        for g in range(2):
            with cur.member(g):
                eps.append(rng.get_rng().eps((per,) + tuple(mu.shape[1:]), mu.device))
        return ops.reparameterize(mu, logvar, torch.cat(eps))

    def latent_shape(self, batch: int, dims):
        """Shape of the conv latent for input extents ``dims`` (the reference hard-codes
        (B,1,10,12,10) for 80x96x80, utils/my_trainer.py:238)."""
        f = 1
        for _, _, s in self.encoder.block_setting:
            f *= s
        return (batch, 1) + tuple(d // f for d in dims)
