"""Drop-in for the reference's ``utils/my_trainer.py`` (the conv-latent trainers).

``train_soft_intro_vae`` and ``train_ResNetVAE`` keep the reference signatures, return
values, log files, checkpoint names and error behaviour; the arithmetic of every
iteration runs in libsiv.  ``SoftIntroStep`` is the E-step / D-step of one iteration
(utils/my_trainer.py:242-325), shared with bench.py.

Deliberate differences from the reference, all host-side:
* the volume extents are read from the batch instead of being hard-coded to 80x96x80
  (``scale = 8/(D*H*W)``, latent noise shape, RMSE normaliser) -- identical for 80x96x80;
* the ten per-iteration ``.item()`` host syncs are one stacked device-to-host copy;
* slice images (``save_image``, matplotlib + skimage) are not produced;
* output sub-directories are created instead of being required to pre-exist;
* with WORLD_SIZE > 1 (one process per GPU) the trainer does what main_DataParallel.py's nn.DataParallel
  does inside one process: every rank starts from rank 0's parameters and buffers, each batch the loader
  yields is the GLOBAL batch and rank r trains on its slice of it (the batch size must be a multiple of
  the world size), the encoder / decoder gradients are averaged over ranks before each optimiser step,
  logged statistics are means over the global batch, BatchNorm statistics stay per replica, and only
  rank 0 writes checkpoints and logs.
"""
from __future__ import annotations

import csv
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from .. import ops, rng
from ..models import lossf
from ..optim import FlatAdam, MultiStepLR
from ..parallel import GradSync


def write_csv(epoch, train_loss, val_loss, path):
    with open(path, "a") as f:
        csv.writer(f).writerow([epoch, train_loss, val_loss])


def calc_kl(logvar, mu, reduce="mean"):
    """utils/my_trainer.py:38-48 -- note the (logvar, mu) argument order."""
    kl = ops.kl_per_sample(mu, logvar)
    if reduce == "mean":
        return kl.mean(dim=0)
    if reduce == "sum":
        return kl.sum()
    return kl


def calc_reconstruction_loss(x, recon_x, loss_type="mse", reduction="None"):
    """utils/my_trainer.py:62-78: per-sample sum of squared errors; only
    ``reduction == 'mean'`` averages over the batch; ``loss_type`` is ignored there too."""
    se = ops.sse_per_sample(x, recon_x)
    return se.mean(dim=0) if reduction == "mean" else se


def load_model(model, pretrained, device):
    """utils/my_trainer.py:130-132 (``strict=False``)."""
    model.load_state_dict(torch.load(pretrained, map_location=device, weights_only=True), strict=False)


def init_weights_he(m):
    """utils/my_trainer.py:511-514."""
    if type(m) == nn.Conv3d or type(m) == nn.ConvTranspose3d:
        nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")


def init_weights_he_relu(m):
    """utils/my_trainer.py:516-519."""
    if type(m) == nn.Conv3d or type(m) == nn.ConvTranspose3d:
        nn.init.kaiming_normal_(m.weight, nonlinearity="relu")


# ------------------------------------------------------------------ text logs (:522-554)
def _write_pairs(path, names, cols):
    with open(path, "w") as f:
        for row in zip(*cols):
            for n, v in zip(names, row):
                f.write(f"{n}{v}\n")


def write_fig(path, trainE, valE, trainD, valD):
    _write_pairs(path, ("trainE=", "valE===", "trainD=", "valD==="), (trainE, valE, trainD, valD))


def write_kl_losses(path, kls_real, kls_fake, kls_rec, rec_errs):
    _write_pairs(path, ("kls_real==", "kls_fake==", "kls_rec===", "rec_errs=="),
                 (kls_real, kls_fake, kls_rec, rec_errs))


def write_kl_losses_onlyvae(path, mse_losses, kl_losses):
    _write_pairs(path, ("mse_loss==", "kl_loss==="), (mse_losses, kl_losses))


def write_figres(path, train, val):
    _write_pairs(path, ("train=", "val==="), (train, val))


# ------------------------------------------------------------------ one iteration
class SoftIntroStep:
    """E-step then D-step of one batch, utils/my_trainer.py:242-325."""

    def __init__(self, model, opt_e: FlatAdam, opt_d: FlatAdam, beta_rec=1.0, beta_neg=1024.0,
                 beta_kl=0.75, gamma_r=1e-8, scale=None, sync: GradSync = None):
        self.model, self.opt_e, self.opt_d = model, opt_e, opt_d
        self.beta_rec, self.beta_neg, self.beta_kl, self.gamma_r = beta_rec, beta_neg, beta_kl, gamma_r
        self.scale = scale
        self.sync = sync if sync is not None else GradSync(None)
        self._dec_draws = None        # random draws one decoder pass makes (known after the first one)
        # SIV_PAIR=0: every pass in a launch sequence of its own, as the reference runs them
        self.pair = os.environ.get("SIV_PAIR", "1") != "0"

    def flush(self) -> None:
        """Join a decoder update that a data-parallel job still has in flight (call before anything but the
        next iteration reads the decoder's parameters: validation, checkpoints)."""
        self.sync.wait()

    def _first_two_passes(self, model, real_batch, noise_batch):
        """``fake = decode(noise)`` then ``encode(real)`` (:248-250).  While the previous iteration's decoder
        exchange + update is still in flight on the side stream, the encoder pass -- which reads no decoder
        parameter -- is issued first and the decoder pass behind the join.  The device generator keys every
        draw by its call index, so the two passes keep the indices of the reference order."""
        r = rng.get_rng()
        if self.sync.pending and self._dec_draws is not None and isinstance(r, rng.DeviceRng):
            base = r.calls
            r.calls = base + self._dec_draws                 # the decoder pass's draws come first in call order
            real_mu, real_logvar = model.encode(real_batch)
            after = r.calls
            r.calls = base
            self.sync.wait()
            fake = model.decode(noise_batch)
            if r.calls != base + self._dec_draws:
                raise RuntimeError("a decoder pass changed its number of random draws between iterations")
            r.calls = after
            return fake, real_mu, real_logvar
        self.sync.wait()
        c0 = getattr(r, "calls", None)
        fake = model.decode(noise_batch)
        if c0 is not None:
            self._dec_draws = r.calls - c0
        real_mu, real_logvar = model.encode(real_batch)
        return fake, real_mu, real_logvar

    def __call__(self, real_batch, noise_batch):
        model = self.model
        br, bn, bk, gr = self.beta_rec, self.beta_neg, self.beta_kl, self.gamma_r
        scale = self.scale if self.scale is not None else 8.0 / float(np.prod(real_batch.shape[2:]))

        # ================ Update E (:242-289) ================
        for p in model.encoder.parameters():
            p.requires_grad = True
        for p in model.decoder.parameters():
            p.requires_grad = False
        fake, real_mu, real_logvar = self._first_two_passes(model, real_batch, noise_batch)
        z = model.reparameterize(real_mu, real_logvar)
        rec = model.decode(z)
        # passes that share their weights and differ only in the input run as one batch where that fills the
        # chip better (BatchNorm statistics, running updates and random draws per pass, as in the reference)
        pair = self.pair and hasattr(model, "pair_ready") and model.pair_ready()
        nb = real_batch.shape[0]
        if pair:
            mu2, logvar2, _, (rec_rec, rec_fake) = model.forward_pair(rec.detach(), fake.detach())     # :266-267
            kl2 = ops.kl_per_sample(mu2, logvar2)
            kl_rec, kl_fake = kl2[:nb], kl2[nb:]
        else:
            rec_mu, rec_logvar, z_rec, rec_rec = model.forward(rec.detach())
            fake_mu, fake_logvar, z_fake, rec_fake = model.forward(fake.detach())
            kl_rec, kl_fake = ops.kl_per_sample(rec_mu, rec_logvar), ops.kl_per_sample(fake_mu, fake_logvar)
        # per-sample terms (:260-275); `rec` is NOT detached in the last one (:275)
        lossE, se = ops.sintro_loss_e(
            ops.sse_per_sample(real_batch, rec), ops.kl_per_sample(real_mu, real_logvar),
            ops.sse_per_sample(fake, rec_fake), kl_fake,
            ops.sse_per_sample(rec, rec_rec), kl_rec, scale, br, bn, bk)
        self.opt_e.zero_grad()
        lossE.backward()
        self.sync.average_and_step(self.opt_e)       # multi-GPU: exchange + update run beside the next two decodes
        stats_e = (lossE.detach(), se[0], se[1], se[2])      # lossE, kl_real, exp_elbo_rec, exp_elbo_fake
        z = z.detach()
        del fake, rec, rec_rec, rec_fake, lossE

        # ================ Update D (:291-325) ================
        for p in model.encoder.parameters():
            p.requires_grad = False
        for p in model.decoder.parameters():
            p.requires_grad = True
        if pair:
            fake, rec = model.decode_pair(noise_batch, z)                                   # :297-298
            self.sync.wait()                         # the encoder update is complete from here on
            mu2, logvar2, z2 = model.encode_reparam_pair(rec, fake)                          # :304-308
            rec_rec, rec_fake = model.decode_pair(z2.detach())                               # :310-311
            kl2 = ops.kl_per_sample(mu2, logvar2)
            kl_rec, kl_fake = kl2[:nb], kl2[nb:]
        else:
            fake = model.decode(noise_batch)
            rec = model.decode(z)
            self.sync.wait()                         # the encoder update is complete from here on
            rec_mu, rec_logvar = model.encode(rec)
            z_rec = model.reparameterize(rec_mu, rec_logvar)
            fake_mu, fake_logvar = model.encode(fake)
            z_fake = model.reparameterize(fake_mu, fake_logvar)
            rec_rec = model.decode(z_rec.detach())
            rec_fake = model.decode(z_fake.detach())
            kl_rec, kl_fake = ops.kl_per_sample(rec_mu, rec_logvar), ops.kl_per_sample(fake_mu, fake_logvar)
        lossD, sd = ops.sintro_loss_d(
            ops.sse_per_sample(real_batch, rec), kl_rec, kl_fake, ops.sse_per_sample(rec.detach(), rec_rec),
            ops.sse_per_sample(fake.detach(), rec_fake), scale, br, bk, gr)
        self.opt_d.zero_grad()
        lossD.backward()
        fake_kl, rec_kl, loss_rec = sd[0], sd[1], sd[2]
        # one device->host copy serves every statistic the trainer logs; on a data-parallel job the decoder
        # exchange + update stay in flight behind it and are joined by the next iteration (or flush())
        return self.sync.finish_step(
            self.opt_d,
            torch.stack([stats_e[0], lossD.detach(), stats_e[1], fake_kl, rec_kl, loss_rec, stats_e[2], stats_e[3]]))


class GraphedSoftIntroStep:
    """One ``SoftIntroStep`` iteration captured in a HIP graph and replayed.

    An iteration is ~1250 kernel launches; issued one by one from Python they cost ~60 ms of host time.
    Captured once (all forward passes, both backward passes, both Adam updates) and replayed, the host cost
    is one graph launch (0.4 ms).  The device time does not change -- the iteration is GPU-bound even at
    batch 1 -- so this is for callers that want the host thread back, not a speed-up.  What has to
    change from step to step lives in device memory: the inputs (copied into static buffers), the random
    stream (``DeviceRng`` epoch counter, advanced inside the graph), Adam's step count and learning rate
    (``FlatAdam.use_device_state``).  Requires the device generator, fixed batch shapes and a single
    rank (the RCCL exchange is not captured).  The model, optimiser and generator state are left exactly
    as they were before the capture (its warm-up iterations are undone)."""

    def __init__(self, step: SoftIntroStep, real_shape, noise_shape, device, warmup: int = 2):
        r = rng.get_rng()
        if not isinstance(r, rng.DeviceRng):
            raise ValueError("graph capture needs the device random generator (siv.rng.DeviceRng)")
        if step.sync.collective:
            raise ValueError("graph capture is single-rank: the gradient exchange is not captured")
        if warmup < 1:
            raise ValueError("graph capture needs at least one warm-up iteration (lazy allocations, and the "
                             "packed-weight cache must be empty when the capture starts)")
        self.step, self.rng = step, r
        device = torch.device(device)
        r.use_device_epoch(device)
        step.opt_e.use_device_state()
        step.opt_d.use_device_state()
        self.real = torch.rand(tuple(real_shape), device=device)
        self.noise = torch.randn(tuple(noise_shape), device=device)
        model, opts = step.model, (step.opt_e, step.opt_d)
        saved_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        saved_opt = [(o.m.clone(), o.v.clone(), o.step_count) for o in opts]
        saved_calls = r.calls
        r.ns = rng.DeviceRng.GRAPH_NS                     # the graph's draws: a stream-id namespace of their own
        cur = torch.cuda.current_stream(device)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                     # warm-up off the default stream: lazy allocations
            for _ in range(warmup):
                step(self.real, self.noise)
        cur.wait_stream(side)
        self.sid_base = r.calls                           # stream ids baked into the graph start here (+ GRAPH_NS)
        from ..backend import get_backend
        B = get_backend()
        if hasattr(B, "weights_changed"):
            # every packed weight is packed INSIDE the graph: a packing that was cached while the capture ran
            # would be missing from the replays, which update the weights themselves
            B.weights_changed(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = step(self.real, self.noise)
            r.advance_epoch()
        # undo warm-up and capture: nothing was trained yet
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(saved_sd[k])
            for o, (m, v, n) in zip(opts, saved_opt):
                o.m.copy_(m)
                o.v.copy_(v)
                o.step_count = n
                o.dev_step.fill_(n)
            r.epoch.zero_()
        r.calls = saved_calls
        r.ns = 0
        if hasattr(B, "weights_changed"):
            B.weights_changed(device)                     # the weights were just restored
        self.replays = 0

    def __call__(self, real_batch, noise_batch):
        self.real.copy_(real_batch)
        self.noise.copy_(noise_batch)
        self.graph.replay()
        self.replays += 1
        for o in (self.step.opt_e, self.step.opt_d):
            o.step_count += 1
        from ..backend import get_backend
        B = get_backend()
        if hasattr(B, "weights_changed"):                 # the replay updated the weights behind the host's back
            B.weights_changed(self.real.device)
        return self.out


STAT_NAMES = ("lossE", "lossD", "kl_real", "fake_kl", "rec_kl", "loss_rec", "exp_elbo_rec", "exp_elbo_fake")


def _fix_seed(seed: int) -> None:
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def _noise(model, n, dims, device):
    """Latent noise of the reference's ``torch.randn(size=(n,1,10,12,10))`` (:238,:379,:388)."""
    return rng.get_rng().eps(model.latent_shape(n, dims), device)


#  ===============  train Soft Intro VAE function  =================
def train_soft_intro_vae(
    model,
    train_loader,
    val_loader,
    epochs,
    lr=0.001,
    device=torch.device("cpu"),
    path="./output_SoftIntroVAE/",
    beta_rec=1.0,
    beta_neg=1024.0,
    beta_kl=0.75,
    pretrained_path=None,
):
    """utils/my_trainer.py:147-508.  Returns (train_lossE_list, train_lossD_list,
    val_lossE_list, val_lossD_list) with the reference's bookkeeping quirks: running loss
    sums are never reset between epochs (:205,:348) and each list is appended to twice per
    epoch (:350-351,:492-493).  ``lr`` is accepted and ignored (Adam lr is 2e-4, :183-184);
    a NaN loss raises SystemError (:327-328)."""
    seed = 77
    os.makedirs(path, exist_ok=True)
    for sub in ("prams",):
        os.makedirs(os.path.join(path, sub), exist_ok=True)
    with open(path + "train_result.csv", "w") as f:
        csv.writer(f).writerow(["epoch", "train_lossE", "train_lossD", "val_lossE", "val_lossD"])
    sync = GradSync(device)
    if seed != -1:
        _fix_seed(seed)
        if isinstance(rng.get_rng(), rng.DeviceRng):
            rng.manual_seed(seed, sync.rank)
        print("random seed: ", seed)

    if pretrained_path is not None:
        load_model(model, pretrained_path, device)
    model.apply(init_weights_he)                                         # :202 (after the load: quirk kept)
    sync.broadcast_(model)                                               # every rank starts from rank 0's replica
    optimizer_e = FlatAdam(model.encoder.parameters(), lr=2e-4)           # :183
    optimizer_d = FlatAdam(model.decoder.parameters(), lr=2e-4)           # :184
    e_scheduler = MultiStepLR(optimizer_e, milestones=(350,), gamma=0.1)  # :185
    d_scheduler = MultiStepLR(optimizer_d, milestones=(350,), gamma=0.1)  # :186
    gamma_r = 1e-8                                                       # :193
    step = SoftIntroStep(model, optimizer_e, optimizer_d, beta_rec, beta_neg, beta_kl, gamma_r, None, sync)
    start_time = time.time()

    train_lossE_list, train_lossD_list, val_lossE_list, val_lossD_list = [], [], [], []
    train_lossE, train_lossD, val_lossE, val_lossD = 0.0, 0.0, 0.0, 0.0
    kls_real, kls_fake, kls_rec, rec_errs = [], [], [], []
    train_losses, val_losses = [], []
    train_mse_losses, train_kl_losses, val_mse_losses, val_kl_losses = [], [], [], []
    print(f"beta_rec = {beta_rec},  beta_neg = {beta_neg},  beta_kl = {beta_kl}")
    print(f"training epoch:{epochs}")
    for epoch in range(0, epochs):
        loop_start_time = time.time()
        model.train()
        batch_kls_real, batch_kls_fake, batch_kls_rec, batch_rec_errs = [], [], [], []
        train_run_mse = train_run_kl = train_run_loss = 0.0
        nvox = 1
        st = None
        for batch, labels in train_loader:
            real_batch = sync.shard(batch.to(device))                     # rank r trains on its slice of the batch
            noise_batch = _noise(model, real_batch.size(0), real_batch.shape[2:], device)   # :238
            nvox = int(np.prod(real_batch.shape[2:]))
            st = dict(zip(STAT_NAMES, step(real_batch, noise_batch).cpu().tolist()))
            train_lossE += st["lossE"]
            train_lossD += st["lossD"]
            if np.isnan(st["lossD"]) or np.isnan(st["lossE"]):
                raise SystemError                                         # :327-328
            batch_kls_real.append(st["kl_real"])
            batch_kls_fake.append(st["fake_kl"])
            batch_kls_rec.append(st["rec_kl"])
            batch_rec_errs.append(st["loss_rec"])
            train_run_mse += st["loss_rec"]
            train_run_kl += st["rec_kl"]
            train_run_loss += st["loss_rec"] + st["rec_kl"]

        step.flush()
        train_lossE /= len(train_loader)                                  # :348 (never reset: quirk kept)
        train_lossD /= len(train_loader)
        train_lossE_list.append(train_lossE)
        train_lossD_list.append(train_lossD)
        train_run_mse = np.sqrt(train_run_mse / (len(train_loader) * nvox))
        train_run_kl /= len(train_loader)
        train_run_loss /= len(train_loader)
        train_losses.append(train_run_loss)
        train_mse_losses.append(train_run_mse)
        train_kl_losses.append(train_run_kl)
        if st is not None:
            print(f"Epoch[{epoch + 1}/{epochs}] Rec:{st['loss_rec']:.3f}, Kl_E:{st['kl_real']:.3f}, "
                  f"expELBO_R:{st['exp_elbo_rec']:.4e}, expELBO_F:{st['exp_elbo_fake']:.4e}, "
                  f"Kl_F:{st['rec_kl']:.3f}, KL_R:{st['fake_kl']:.3f}, "
                  f"DIFF_Kl_F:{(-st['kl_real'] + st['fake_kl']):.3f}")

        vm, vk, vl, val_lossE, val_lossD = validate_soft_intro_vae(
            model, train_loader, val_loader, device, beta_rec, beta_neg, beta_kl, gamma_r, val_lossE, val_lossD)
        val_lossE_list.append(val_lossE)
        val_lossD_list.append(val_lossD)
        val_losses.append(vl)
        val_mse_losses.append(vm)
        val_kl_losses.append(vk)

        kls_real.append(np.mean(batch_kls_real))
        kls_fake.append(np.mean(batch_kls_fake))
        kls_rec.append(np.mean(batch_kls_rec))
        rec_errs.append(np.mean(batch_rec_errs))

        if sync.rank == 0:
            savename = path + f"prams/S-IntroVAE_3898_epoch{epoch}.pth"   # :476-479
            torch.save({k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}, savename)
        now_time = time.time()
        print(f"Epoch[{epoch + 1}/{epochs}] train_lossE:{train_lossE:.3f}  train_lossD:{train_lossD:.3f}  "
              f"val_lossE:{val_lossE:.3f}  val_lossD:{val_lossD:.3f}\n"
              f"Epoch[{epoch + 1}/{epochs}] Train[RMSE:{train_run_mse:.5f} kl:{train_run_kl:.1f} "
              f"loss:{train_run_loss:.1f}]  Val[RMSE:{vm:.5f} kl:{vk:.2f} loss:{vl:.1f}]  "
              f"1epoch:{(now_time - loop_start_time):.0f}s  total:{(now_time - start_time) / 60:.0f}min")

        train_lossE_list.append(train_lossE)                              # :492-495 (second append: quirk kept)
        train_lossD_list.append(train_lossD)
        val_lossE_list.append(val_lossE)
        val_lossD_list.append(val_lossD)
        if sync.rank == 0:
            write_fig(path + "/loss.txt", train_lossE_list, val_lossE_list, train_lossD_list, val_lossD_list)
            write_kl_losses(path + "/kl_losses.txt", kls_real, kls_fake, kls_rec, rec_errs)
            write_kl_losses_onlyvae(path + "/train_losses.txt", train_mse_losses, train_kl_losses)
            write_kl_losses_onlyvae(path + "/val_losses.txt", val_mse_losses, val_kl_losses)
        e_scheduler.step()
        d_scheduler.step()

    print("Finished S-IntroVAE Traininig !!")
    model.to("cpu")                                                      # :507
    return train_lossE_list, train_lossD_list, val_lossE_list, val_lossD_list


def validate_soft_intro_vae(model, train_loader, val_loader, device, beta_rec, beta_neg, beta_kl, gamma_r,
                            val_lossE=0.0, val_lossD=0.0):
    """The evaluation block of the epoch loop, utils/my_trainer.py:367-463: running BN
    statistics, no dropout, eps = 0.1.  The forward passes whose only purpose in the
    reference is to feed ``save_image`` are still executed where they consume random
    draws, so the draw sequence stays the reference's."""
    model.eval()
    val_run_loss = val_run_mse = val_run_kl = 0.0
    br, bn, bk, gr = beta_rec, beta_neg, beta_kl, gamma_r
    nvox = 1
    with torch.no_grad():
        image, _ = next(iter(train_loader))                               # :370-373
        image = image.to(device)
        model.forward(image)
        _noise(model, 8, image.shape[2:], device)                         # :379 (8 fake samples for the grid)
        for batch, labels in val_loader:
            real_batch = batch.to(device)
            noise_batch = _noise(model, real_batch.size(0), real_batch.shape[2:], device)   # :388
            nvox = int(np.prod(real_batch.shape[2:]))
            scale = 8.0 / nvox
            fake = model.decode(noise_batch)
            real_mu, real_logvar = model.encode(real_batch)
            z = model.reparameterize(real_mu, real_logvar, True)
            rec = model.decode(z)
            loss_rec = calc_reconstruction_loss(real_batch, rec, reduction="mean")
            lossE_real_kl = calc_kl(real_logvar, real_mu, reduce="mean")
            rec_mu, rec_logvar, z_rec, rec_rec = model.forward(rec.detach())
            fake_mu, fake_logvar, z_fake, rec_fake = model.forward(fake.detach())
            fake_kl_e = calc_kl(fake_logvar, fake_mu, reduce="none")
            rec_kl_e = calc_kl(rec_logvar, rec_mu, reduce="none")
            loss_fake_rec = calc_reconstruction_loss(fake, rec_fake, reduction="none")
            loss_rec_rec = calc_reconstruction_loss(rec, rec_rec, reduction="none")
            exp_elbo_fake = (-2 * scale * (br * loss_fake_rec + bn * fake_kl_e)).exp().mean()
            exp_elbo_rec = (-2 * scale * (br * loss_rec_rec + bn * rec_kl_e)).exp().mean()
            lossE = scale * (br * loss_rec + bk * lossE_real_kl) + 0.5 * (exp_elbo_fake + exp_elbo_rec)
            # ---- decoder part (:417-435)
            loss_rec = calc_reconstruction_loss(real_batch, rec.detach(), reduction="mean")
            rec_mu, rec_logvar = model.encode(rec)
            z_rec = model.reparameterize(rec_mu, rec_logvar, True)
            fake_mu, fake_logvar = model.encode(fake)
            z_fake = model.reparameterize(fake_mu, fake_logvar, True)
            rec_rec = model.decode(z_rec)
            rec_fake = model.decode(z_fake)
            loss_rec_rec = calc_reconstruction_loss(rec.detach(), rec_rec, reduction="mean")
            loss_fake_rec = calc_reconstruction_loss(fake.detach(), rec_fake, reduction="mean")
            rec_kl = calc_kl(rec_logvar, rec_mu, reduce="mean")
            fake_kl = calc_kl(fake_logvar, fake_mu, reduce="mean")
            lossD = scale * (loss_rec * br + 0.5 * bk * (rec_kl + fake_kl)
                             + gr * 0.5 * br * (loss_rec_rec + loss_fake_rec))
            vals = torch.stack([lossE, lossD, loss_rec, rec_kl]).cpu().tolist()
            val_lossE += vals[0]
            val_lossD += vals[1]
            val_run_mse += vals[2]
            val_run_kl += vals[3]
            val_run_loss += vals[2] + vals[3]
        val_lossE /= len(val_loader)
        val_lossD /= len(val_loader)
        val_run_mse = np.sqrt(val_run_mse / (len(val_loader) * nvox))
        val_run_kl /= len(val_loader)
        val_run_loss /= len(val_loader)
        image, _ = next(iter(val_loader))                                 # :455-458
        model.forward(image.to(device))
    return val_run_mse, val_run_kl, val_run_loss, val_lossE, val_lossD


# trainer for the plain ResNet VAE
def train_ResNetVAE(
    net,
    train_loader,
    val_loader,
    epochs=1,
    lr=0.001,
    mse_w=1,
    kl_w=20,
    device=torch.device("cpu"),
    path="./output_ResNetVAE/",
):
    """utils/my_trainer.py:557-652.  Returns (train_losses, val_losses).  Validation uses
    ``normal_loss``'s default weights (1, 10), not (mse_w, kl_w) -- :616, quirk kept."""
    os.makedirs(path, exist_ok=True)
    with open(path + "train_result.csv", "w") as f:
        csv.writer(f).writerow(["epoch", "train_loss", "val_loss"])
    sync = GradSync(device)
    net = net.to(device)
    print(f"mse_weight={mse_w}  kl_weight={kl_w}")
    net = net.apply(init_weights_he_relu)
    sync.broadcast_(net)
    optimizer = FlatAdam(net.parameters(), lr)
    train_losses, val_losses = [], []
    train_mse_losses, train_kl_losses, val_mse_losses, val_kl_losses = [], [], [], []
    start_time = time.time()
    for epoch in range(epochs):
        loop_start_time = time.time()
        train_run_mse = train_run_kl = train_run_loss = 0.0
        net.train()
        for inputs, labels in train_loader:
            inputs = sync.shard(inputs.to(device))
            optimizer.zero_grad()
            x_re, mu, logvar = net.forward(inputs)
            loss, mse, kl = lossf.normal_loss(x_re, mu, logvar, inputs, mse_w, kl_w)
            loss.backward()
            sync.average_(optimizer.flat_g)
            optimizer.step()
            l, m, k = sync.average_stats_(torch.stack([loss.detach(), mse.detach(), kl.detach()])).cpu().tolist()
            train_run_loss += l
            train_run_mse += m
            train_run_kl += k
        train_run_loss /= len(train_loader)
        train_run_mse /= len(train_loader)
        train_run_kl /= len(train_loader)
        train_losses.append(train_run_loss)
        train_mse_losses.append(train_run_mse)
        train_kl_losses.append(train_run_kl)

        net.eval()
        val_run_loss = val_run_mse = val_run_kl = 0.0
        with torch.no_grad():
            for inputs, labels in val_loader:
                inputs = inputs.to(device)
                x_re, mu, logvar = net.forward(inputs)
                loss, mse, kl = lossf.normal_loss(x_re, mu, logvar, inputs)
                l, m, k = torch.stack([loss, mse, kl]).cpu().tolist()
                val_run_loss += l
                val_run_mse += m
                val_run_kl += k
        val_run_loss /= len(val_loader)
        val_run_mse /= len(val_loader)
        val_run_kl /= len(val_loader)
        val_losses.append(val_run_loss)
        val_mse_losses.append(val_run_mse)
        val_kl_losses.append(val_run_kl)
        if epoch % 10 == 0 and sync.rank == 0:
            torch.save({k: v.detach().to("cpu").contiguous() for k, v in net.state_dict().items()},
                       path + f"ResNetVAE_3898epoch{epoch}.pth")
        now_time = time.time()
        print(f"Epoch[{epoch + 1}/{epochs}] Train[mse:{train_run_mse:.1f} kl:{train_run_kl:.1f} "
              f"loss:{train_run_loss:.1f}]  Val[mse:{val_run_mse:.1f} kl:{val_run_kl:.2f} "
              f"loss:{val_run_loss:.1f}]  1epoch:{(now_time - loop_start_time):.0f}s  "
              f"total:{(now_time - start_time) / 60:.0f}min")

    write_figres(path + "/loss.txt", train_losses, val_losses)
    write_kl_losses_onlyvae(path + "/train_losses.txt", train_mse_losses, train_kl_losses)
    write_kl_losses_onlyvae(path + "/val_losses.txt", val_mse_losses, val_kl_losses)
    if epochs != 0:
        net = net.to("cpu")
        if sync.rank == 0:
            torch.save(net.state_dict(), path + "resnetvae_weight.pth")
    print("finished ResNetVAE traininig.")
    return train_losses, val_losses
