"""TEST-ONLY stand-in for siv.backend.HipBackend that computes every operator on the CPU
with plain torch / oracle code.  It lets the CPU test-suite exercise the HOST logic of the
product (autograd wiring, trainer bookkeeping, flat optimiser, gradient averaging) against
the reference goldens on a machine without a GPU.  It is never importable from the product
package and nothing in ``siv`` falls back to it."""
import torch
import torch.nn.functional as F
from torch.nn import grad as nngrad

from oracle import siv_oracle as O


class CpuCheckerBackend:
    name = "cpu-checker"

    def set_conv_path(self, path):
        pass

    def conv3d_k3_fwd(self, x, w, bias, ups=1):
        return O.conv3d_k3(x, w, bias, ups)

    def conv3d_k3_fwd_stats(self, x, w, bias, ups, running_mean, running_var, nbt, momentum):
        y = O.conv3d_k3(x, w, bias, ups)
        mean, var = self.bn_stats(y, running_mean, running_var, nbt, momentum)
        return y, mean, var

    def conv3d_k3_dgrad(self, dy, w, ups=1):
        N, Cout, D, H, W = dy.shape
        dx = nngrad.conv3d_input((N, w.shape[1], D, H, W), w, dy, stride=1, padding=1)
        if ups != 1:
            dx = F.avg_pool3d(dx, ups) * (ups ** 3)
        return dx

    @staticmethod
    def _into(dw, db, dw_out, db_out):
        """Mirror of the HIP backend's accumulate-into-buffer mode."""
        if dw_out is None:
            return dw, db
        with torch.no_grad():
            dw_out += dw
            if db is not None:
                db_out += db
        return dw_out, (db_out if db is not None else None)

    def conv3d_k3_wgrad(self, x, dy, ups=1, want_bias=False, dw_out=None, db_out=None):
        if ups != 1:
            x = F.interpolate(x, scale_factor=ups, mode="nearest")
        dw = nngrad.conv3d_weight(x, (dy.shape[1], x.shape[1], 3, 3, 3), dy, stride=1, padding=1)
        return self._into(dw, (dy.sum(dim=(0, 2, 3, 4)) if want_bias else None), dw_out, db_out)

    def conv_transpose3d_k3_fwd(self, x, w, bias):
        return F.conv_transpose3d(x, w, bias, stride=1, padding=1)

    def conv_transpose3d_k3_dgrad(self, dy, w):
        return F.conv3d(dy, w, None, 1, 1)

    def conv_transpose3d_k3_wgrad(self, x, dy, want_bias=False):
        dw = nngrad.conv3d_weight(dy, (x.shape[1], dy.shape[1], 3, 3, 3), x, stride=1, padding=1)
        return dw, (dy.sum(dim=(0, 2, 3, 4)) if want_bias else None)

    def conv1x1_fwd(self, x, w, bias):
        return F.conv3d(x, w, bias)

    def conv1x1_dgrad(self, dy, w):
        return nngrad.conv3d_input((dy.shape[0], w.shape[1]) + tuple(dy.shape[2:]), w, dy)

    def conv1x1_wgrad(self, x, dy, want_bias=False, dw_out=None, db_out=None):
        dw = nngrad.conv3d_weight(x, (dy.shape[1], x.shape[1], 1, 1, 1), dy)
        return self._into(dw, (dy.sum(dim=(0, 2, 3, 4)) if want_bias else None), dw_out, db_out)

    def linear_fwd(self, x, w, bias):
        return F.linear(x, w, bias)

    def linear_dgrad(self, dy, w):
        return dy @ w

    def linear_wgrad(self, x, dy, want_bias=False, dw_out=None, db_out=None):
        return self._into(dy.t() @ x, (dy.sum(0) if want_bias else None), dw_out, db_out)

    def add_act_fwd(self, a, b, slope):
        return O.act(a + b, slope)

    def add_act_bwd(self, dout, a, b, slope):
        s = a + b
        return dout * torch.where(s > 0, torch.ones_like(s), torch.full_like(s, slope))

    def bn_stats(self, y, running_mean, running_var, nbt, momentum, groups=1):
        if groups > 1:          # one statistics pair and one running update per consecutive part of the batch
            parts = [self.bn_stats(yg, running_mean, running_var, nbt, momentum) for yg in y.chunk(groups)]
            return torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts])
        mean = y.mean(dim=(0, 2, 3, 4))
        var = y.var(dim=(0, 2, 3, 4), unbiased=False)
        if running_mean is not None:
            m = y.numel() // y.shape[1]
            running_mean.mul_(1 - momentum).add_(momentum * mean)
            running_var.mul_(1 - momentum).add_(momentum * var * (m / max(m - 1, 1)))
        if nbt is not None:
            nbt += 1
        return mean, var

    @staticmethod
    def _tail(y, mean, var, gamma, beta, res, keep, p, slope, pool, eps):
        sh = (1, -1, 1, 1, 1)
        h = (y - mean.view(sh)) / torch.sqrt(var.view(sh) + eps) * gamma.view(sh) + beta.view(sh)
        if res is not None:
            h = h + res
        h = O.act(h, slope)
        if pool != 1:
            h = F.avg_pool3d(h, pool)
        if keep is not None:
            h = h * keep.float() * (1.0 / (1.0 - p))
        return h

    def bn_act_fwd(self, y, mean, var, gamma, beta, res, keep, p, slope, pool, eps, out=None):
        if out is not None:     # destination given by the caller
            # written behind autograd's back, as a kernel writes through a raw pointer (the destination is a
            # slice of a buffer whose other slices are outputs of earlier calls)
            out.data.copy_(self.bn_act_fwd(y, mean, var, gamma, beta, res, keep, p, slope, pool, eps))
            return out
        if mean.dim() == 2:     # grouped: statistics (groups, C)
            G = mean.shape[0]
            ch = lambda t: t.chunk(G) if t is not None else [None] * G
            return torch.cat([self._tail(yg, mean[g], var[g], gamma, beta, rg, kg, p, slope, pool, eps)
                              for g, (yg, rg, kg) in enumerate(zip(ch(y), ch(res), ch(keep)))])
        return self._tail(y, mean, var, gamma, beta, res, keep, p, slope, pool, eps)

    def bn_act_bwd(self, dout, y, mean, var, gamma, beta, res, keep, p, slope, pool, eps, want_affine, want_res,
                   dgamma_out=None, dbeta_out=None):
        if mean.dim() == 2:
            G = mean.shape[0]
            ch = lambda t: t.chunk(G) if t is not None else [None] * G
            parts = [self.bn_act_bwd(dg, yg, mean[g], var[g], gamma, beta, rg, kg, p, slope, pool, eps, want_affine,
                                     want_res)
                     for g, (dg, yg, rg, kg) in enumerate(zip(ch(dout), ch(y), ch(res), ch(keep)))]
            dy = torch.cat([q[0] for q in parts])
            dres = torch.cat([q[3] for q in parts]) if parts[0][3] is not None else None
            dg_ = sum(q[1] for q in parts) if want_affine else None
            db_ = sum(q[2] for q in parts) if want_affine else None
            if want_affine and dgamma_out is not None:
                with torch.no_grad():
                    dgamma_out += dg_
                    dbeta_out += db_
                return dy, dgamma_out, dbeta_out, dres
            return dy, dg_, db_, dres
        # differentiate THROUGH the batch statistics, as training-mode BatchNorm does
        with torch.enable_grad():
            yy = y.detach().requires_grad_(True)
            g = gamma.detach().requires_grad_(True)
            b = beta.detach().requires_grad_(True)
            r = res.detach().requires_grad_(True) if res is not None else None
            m = yy.mean(dim=(0, 2, 3, 4))
            v = yy.var(dim=(0, 2, 3, 4), unbiased=False)
            out = self._tail(yy, m, v, g, b, r, keep, p, slope, pool, eps)
            ins = [yy, g, b] + ([r] if r is not None else [])
            grads = torch.autograd.grad(out, ins, dout)
        dres = grads[3] if (r is not None and want_res) else None
        if want_affine and dgamma_out is not None:
            with torch.no_grad():
                dgamma_out += grads[1]
                dbeta_out += grads[2]
            return grads[0], dgamma_out, dbeta_out, dres
        return grads[0], (grads[1] if want_affine else None), (grads[2] if want_affine else None), dres

    def act_dropout_fwd(self, x, keep, p, slope):
        h = O.act(x, slope)
        return h * keep.float() * (1.0 / (1.0 - p)) if keep is not None else h

    def act_dropout_bwd(self, dout, x, keep, p, slope):
        g = dout * keep.float() * (1.0 / (1.0 - p)) if keep is not None else dout
        return g * torch.where(x > 0, torch.ones_like(x), torch.full_like(x, slope))

    def reparam_fwd(self, mu, logvar, eps, eps_const):
        e = eps if eps is not None else eps_const
        return mu + e * torch.exp(0.5 * logvar)

    def reparam_bwd(self, dz, logvar, eps, eps_const):
        e = eps if eps is not None else eps_const
        return dz, dz * e * 0.5 * torch.exp(0.5 * logvar)

    def kl_fwd(self, mu, logvar):
        return O.calc_kl(logvar, mu, "none")

    def kl_bwd(self, g, mu, logvar):
        gb = g.view(-1, *([1] * (mu.dim() - 1)))
        return gb * mu, gb * 0.5 * (logvar.exp() - 1)

    def sse_fwd(self, x, y):
        return O.calc_reconstruction_loss(x, y, "none")

    def sse_bwd(self, g, x, y, want_x, want_y):
        d = 2 * g.view(-1, *([1] * (x.dim() - 1))) * (x - y)
        return (d if want_x else None), (-d if want_y else None)

    def sintro_loss_e_fwd(self, vecs, s, br, bn, bk):
        a, k, sf, kf, sr, kr = vecs
        ef = (-2 * s * (br * sf + bn * kf)).exp().mean()
        er = (-2 * s * (br * sr + bn * kr)).exp().mean()
        return torch.stack([10 * (s * (br * a.mean() + bk * k.mean()) + 0.5 * (ef + er)), k.mean(), er, ef])

    def sintro_loss_e_bwd(self, g, vecs, s, br, bn, bk):
        with torch.enable_grad():
            vs = [v.detach().requires_grad_(True) for v in vecs]
            loss = self.sintro_loss_e_fwd(vs, s, br, bn, bk)[0]
            gr = torch.autograd.grad(loss, vs, g.reshape(()))
        return torch.stack(gr)

    def sintro_loss_d_fwd(self, vecs, s, br, bk, gr):
        a, kr, kf, rr, fr = vecs
        loss = 10 * (s * (br * a.mean() + 0.5 * bk * (kr.mean() + kf.mean()) + gr * 0.5 * br * (rr.mean() + fr.mean())))
        return torch.stack([loss, kf.mean(), kr.mean(), a.mean()])

    def sintro_loss_d_bwd(self, g, B, s, br, bk, gr):
        c = 10 * g.reshape(()) * s / B
        one = torch.ones(B)
        return torch.stack([c * br * one, c * 0.5 * bk * one, c * 0.5 * bk * one, c * gr * 0.5 * br * one,
                            c * gr * 0.5 * br * one])

    def adam_step(self, p, g, m, v, lr, beta1, beta2, eps, step):
        st = {0: {"step": step - 1, "m": m, "v": v}}
        with torch.no_grad():
            O.adam_step([p], [g], st, lr, beta1, beta2, eps)

    def randn(self, shape, device, seed, stream_id):
        gen = torch.Generator().manual_seed((seed * 7919 + stream_id) % (2 ** 63))
        return torch.randn(tuple(shape), generator=gen)

    def dropout_keep(self, shape, device, p, seed, stream_id):
        gen = torch.Generator().manual_seed((seed * 7919 + stream_id) % (2 ** 63))
        return (torch.rand(tuple(shape), generator=gen) >= p).to(torch.uint8)

    def to_bf16(self, x):
        return x.to(torch.bfloat16)          # torch rounds to nearest even, like siv_f32_to_bf16

    def from_bf16(self, h):
        return h.to(torch.float32)

    def scale_(self, x, a):
        with torch.no_grad():
            x.mul_(a)

    def axpby(self, a, x, b, y):
        with torch.no_grad():
            if b == 0.0:
                y.copy_(a * x)
            else:
                y.mul_(b).add_(x, alpha=a)
