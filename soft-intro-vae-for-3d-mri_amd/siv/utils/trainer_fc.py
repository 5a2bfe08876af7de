"""Drop-in for the reference's ``utils/trainer_fc.py``: the Soft-IntroVAE trainer for the
FC-latent model (``models/mymodel.py``).  The iteration is the E-step / D-step of
``my_trainer.SoftIntroStep`` (utils/trainer_fc.py:221-299 is the same algebra as
utils/my_trainer.py:242-325); what differs from ``my_trainer.train_soft_intro_vae`` is
kept as the reference has it:

* the model is moved to ``device`` inside the function (:159);
* latent noise is ``(b_size, model.z_ch)`` (:218,:340,:350);
* validation samples z stochastically (``reparameterize`` has no val flag, :355,:381,:384)
  and accumulates ``lossE * 10`` / ``lossD * 10`` (:374,:396);
* only ``loss.txt`` and ``kl_losses.txt`` are written (:445-446); checkpoints are
  ``prams/S-IntroVAE_4184_epoch{e}.pth`` (:429);
* the two MultiStepLR schedulers step ONCE, after the last epoch (:448-449).

Host-side differences from the reference are those listed in ``my_trainer`` (no slice
images or matplotlib result plots, one device-to-host copy per iteration, output
directories created, gradient averaging over ranks when WORLD_SIZE > 1).
"""
from __future__ import annotations

import csv
import os
import time

import numpy as np
import torch

from .. import rng
from ..optim import FlatAdam, MultiStepLR
from ..parallel import GradSync
from .my_trainer import (STAT_NAMES, SoftIntroStep, _fix_seed, calc_kl, calc_reconstruction_loss,  # noqa: F401
                         init_weights_he, load_model, write_csv, write_fig, write_kl_losses)


def _noise(model, n, device):
    return rng.get_rng().eps((n, model.z_ch), device)


def train_soft_intro_vae(
    model=None,
    train_loader=None,
    val_loader=None,
    epochs=500,
    lr=2e-4,
    device=torch.device("cpu"),
    path="./output_SoftIntroVAE/",
    beta_rec=1.0,
    beta_neg=1024.0,
    beta_kl=0.75,
    pretrained_path=None,
):
    """utils/trainer_fc.py:128-454.  Returns (train_lossE_list, train_lossD_list,
    val_lossE_list, val_lossD_list); running sums are never reset between epochs and every
    list is appended to twice per epoch, as in the reference.  ``lr`` is ignored (Adam lr is
    2e-4, :165-166); a NaN loss raises SystemError (:302-303)."""
    seed = 77
    os.makedirs(os.path.join(path, "prams"), exist_ok=True)
    with open(path + "train_result.csv", "w") as f:
        csv.writer(f).writerow(["epoch", "train_lossE", "train_lossD", "val_lossE", "val_lossD"])
    sync = GradSync(device)
    if seed != -1:
        _fix_seed(seed)
        if isinstance(rng.get_rng(), rng.DeviceRng):
            rng.manual_seed(seed, sync.rank)
        print("random seed: ", seed)

    model.to(device)                                                     # :159
    if pretrained_path is not None:
        load_model(model, pretrained_path, device)
    model.apply(init_weights_he)                                         # :187
    sync.broadcast_(model)                                               # every rank starts from rank 0's replica
    optimizer_e = FlatAdam(model.encoder.parameters(), lr=2e-4)           # :165
    optimizer_d = FlatAdam(model.decoder.parameters(), lr=2e-4)           # :166
    e_scheduler = MultiStepLR(optimizer_e, milestones=(350,), gamma=0.1)
    d_scheduler = MultiStepLR(optimizer_d, milestones=(350,), gamma=0.1)
    gamma_r = 1e-8                                                       # :177
    step = SoftIntroStep(model, optimizer_e, optimizer_d, beta_rec, beta_neg, beta_kl, gamma_r, None, sync)
    br, bn, bk, gr = beta_rec, beta_neg, beta_kl, gamma_r
    start_time = time.time()

    train_lossE_list, train_lossD_list, val_lossE_list, val_lossD_list = [], [], [], []
    train_lossE, train_lossD, val_lossE, val_lossD = 0.0, 0.0, 0.0, 0.0
    kls_real, kls_fake, kls_rec, rec_errs = [], [], [], []
    print(f"beta_rec = {beta_rec},  beta_neg = {beta_neg},  beta_kl = {beta_kl}")
    print(f"training epoch:{epochs}")
    for epoch in range(0, epochs):
        loop_start_time = time.time()
        model.train()
        batch_kls_real, batch_kls_fake, batch_kls_rec, batch_rec_errs = [], [], [], []
        st = None
        for batch, labels in train_loader:
            real_batch = sync.shard(batch.to(device))                     # rank r trains on its slice of the batch
            noise_batch = _noise(model, real_batch.size(0), device)       # :218
            st = dict(zip(STAT_NAMES, step(real_batch, noise_batch).cpu().tolist()))
            train_lossE += st["lossE"]
            train_lossD += st["lossD"]
            if np.isnan(st["lossD"]) or np.isnan(st["lossE"]):
                raise SystemError                                         # :302-303
            batch_kls_real.append(st["kl_real"])
            batch_kls_fake.append(st["fake_kl"])
            batch_kls_rec.append(st["rec_kl"])
            batch_rec_errs.append(st["loss_rec"])

        step.flush()
        train_lossE /= len(train_loader)                                  # :317 (never reset: quirk kept)
        train_lossD /= len(train_loader)
        train_lossE_list.append(train_lossE)
        train_lossD_list.append(train_lossD)
        if st is not None:
            print(f"Epoch [{epoch + 1}/{epochs}]  Rec:{st['loss_rec']:.4f}, Kl_E:{st['kl_real']:.4f}, "
                  f"expELBO_R:{st['exp_elbo_rec']:.4e}, expELBO_F:{st['exp_elbo_fake']:.4e}, "
                  f"Kl_F:{st['rec_kl']:.4f}, KL_R:{st['fake_kl']:.4f}, "
                  f"DIFF_Kl_F:{(-st['kl_real'] + st['fake_kl']):.4f}")

        model.eval()
        with torch.no_grad():
            image, _ = next(iter(train_loader))                           # :331-334 (consumes one eps draw)
            model.forward(image.to(device))
            model.decode(_noise(model, 8, device))                        # :340-341
            for batch, labels in val_loader:
                noise_batch = _noise(model, batch.size(0), device)        # :350
                real_batch = batch.to(device)
                scale = 8.0 / float(np.prod(real_batch.shape[2:]))
                fake = model.decode(noise_batch)
                real_mu, real_logvar = model.encode(real_batch)
                z = model.reparameterize(real_mu, real_logvar)
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
                lossE = lossE * 10                                        # :374
                # ---- decoder part (:378-397)
                loss_rec = calc_reconstruction_loss(real_batch, rec.detach(), reduction="mean")
                rec_mu, rec_logvar = model.encode(rec)
                z_rec = model.reparameterize(rec_mu, rec_logvar)
                fake_mu, fake_logvar = model.encode(fake)
                z_fake = model.reparameterize(fake_mu, fake_logvar)
                rec_rec = model.decode(z_rec)
                rec_fake = model.decode(z_fake)
                loss_rec_rec = calc_reconstruction_loss(rec.detach(), rec_rec, reduction="mean")
                loss_fake_rec = calc_reconstruction_loss(fake.detach(), rec_fake, reduction="mean")
                rec_kl = calc_kl(rec_logvar, rec_mu, reduce="mean")
                fake_kl = calc_kl(fake_logvar, fake_mu, reduce="mean")
                lossD = scale * (loss_rec * br + 0.5 * bk * (rec_kl + fake_kl)
                                 + gr * 0.5 * br * (loss_rec_rec + loss_fake_rec))
                lossD = lossD * 10                                        # :396
                vals = torch.stack([lossE, lossD]).cpu().tolist()
                val_lossE += vals[0]
                val_lossD += vals[1]
            val_lossE /= len(val_loader)
            val_lossD /= len(val_loader)
            val_lossE_list.append(val_lossE)
            val_lossD_list.append(val_lossD)
            image, _ = next(iter(val_loader))                             # :405-408
            model.forward(image.to(device))

        kls_real.append(np.mean(batch_kls_real))
        kls_fake.append(np.mean(batch_kls_fake))
        kls_rec.append(np.mean(batch_kls_rec))
        rec_errs.append(np.mean(batch_rec_errs))

        if sync.rank == 0:
            savename = path + f"prams/S-IntroVAE_4184_epoch{epoch}.pth"   # :429-431
            torch.save({k: v.detach().to("cpu").contiguous() for k, v in model.state_dict().items()}, savename)
        now_time = time.time()
        print(f"Epoch [{epoch + 1}/{epochs}]  train_lossE:{train_lossE:.3f}  train_lossD:{train_lossD:.3f}  "
              f"val_lossE:{val_lossE:.3f}  val_lossD:{val_lossD:.3f}, "
              f"1epoch:{(now_time - loop_start_time):.1f}s  total time:{(now_time - start_time) / 60:.1f}min")

        train_lossE_list.append(train_lossE)                              # :440-443 (second append: quirk kept)
        train_lossD_list.append(train_lossD)
        val_lossE_list.append(val_lossE)
        val_lossD_list.append(val_lossD)
        if sync.rank == 0:
            write_fig(path + "/loss.txt", train_lossE_list, val_lossE_list, train_lossD_list, val_lossD_list)
            write_kl_losses(path + "/kl_losses.txt", kls_real, kls_fake, kls_rec, rec_errs)

    e_scheduler.step()                                                   # :448-449: once, after the loop
    d_scheduler.step()
    print("Finished S-IntroVAE Traininig !!")
    model.to("cpu")                                                      # :453
    return train_lossE_list, train_lossD_list, val_lossE_list, val_lossD_list
