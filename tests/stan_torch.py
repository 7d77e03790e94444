"""Second, independent oracle for the log-posterior: a statement-by-statement transcription of the reference's Stan
programs into torch (fp64), differentiated by autograd -- SURVEY section 7 step 2(iv).

TEST INFRASTRUCTURE.  Written from the model TEXT (bayes_drt/stan_model_files/*_modelcode.txt), not from oracle/bdrt_oracle.c:
no hand-derived gradient, no shared code.  It pins the two model families no stored Stan result covers -- the package form of
the Series outlier models (S2) and the single parallel distribution (S6) -- and, as a control, S1.

Stan semantics used: parameters in declaration order; `<lower=0>` => theta = exp(y) and, when the
Jacobian is on (sampling), lp += y; `~` statements drop additive constants that do not depend on parameters.
"""
import numpy as np
import torch

torch.set_default_dtype(torch.float64)


class _Unpack:
    """Reads parameters off the unconstrained vector in declaration order, applying the lower=0 transform."""

    def __init__(self, y, jacobian):
        self.y, self.o, self.jac, self.lp = y, 0, jacobian, torch.zeros((), dtype=torch.float64)

    def take(self, n=None, lower0=True):
        k = 1 if n is None else n
        v = self.y[self.o:self.o + k]
        self.o += k
        if lower0:
            if self.jac:
                self.lp = self.lp + v.sum()
            v = torch.exp(v)
        return v[0] if n is None else v


def _std_normal(v):
    return (-0.5 * v * v).sum()


def _normal(y, mu, sigma):                 # y ~ normal(mu, sigma) with sigma a parameter: -log(sigma) stays
    return (-torch.log(sigma) - 0.5 * ((y - mu) / sigma) ** 2).sum()


def _inv_gamma(y, alpha, beta):            # alpha, beta are data: their normalising terms drop
    return (-(alpha + 1.0) * torch.log(y) - beta / y).sum()


def _exponential(y, lam):
    return (-lam * y).sum()


def _dups(ups):
    K = ups.shape[0]
    return torch.stack([0.5 * (ups[k + 1] - 0.5 * (ups[k] + ups[k + 2])) / ups[k + 1] for k in range(K - 2)])


def _t(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float64))


def series_outliers_lp(y, dat, jacobian, pos):
    """Series_outliers_modelcode.txt / Series_pos_outliers_modelcode.txt (they differ in `vector<lower=0>[K] x`).
    dat: N (= number of frequencies), K, A [2N x K], Z [2N], freq [N], L0, L1, L2, sigma_min, ups_alpha, ups_beta,
    induc_scale, sigma_out_lambda, sigma_out_alpha, sigma_out_beta."""
    N, K = int(dat['N']), int(dat['K'])
    A, Z, freq = _t(dat['A']), _t(dat['Z']), _t(dat['freq'])
    L0, L1, L2 = _t(dat['L0']), _t(dat['L1']), _t(dat['L2'])
    Rinf_vec = torch.cat([torch.ones(N), torch.zeros(N)])
    induc_vec = torch.cat([torch.zeros(N), 2 * np.pi * freq])
    u = _Unpack(y, jacobian)
    Rinf_raw, induc_raw = u.take(), u.take()
    x = u.take(K, lower0=pos)
    sigma_res_raw, alpha_prop_raw, alpha_re_raw, alpha_im_raw = u.take(), u.take(), u.take(), u.take()
    sigma_out_raw, sigma_out_scale = u.take(N), u.take(N)
    ups_raw = u.take(K)
    d0, d1, d2 = u.take(), u.take(), u.take()
    assert u.o == y.shape[0]
    # transformed parameters
    Rinf = Rinf_raw * 100
    induc = induc_raw * dat['induc_scale']
    q = torch.sqrt(d0 * (L0 @ x) ** 2 + d1 * (L1 @ x) ** 2 + d2 * (L2 @ x) ** 2)
    sigma_res, alpha_prop, alpha_re, alpha_im = sigma_res_raw * 0.05, alpha_prop_raw * 0.05, alpha_re_raw * 0.05, alpha_im_raw * 0.05
    sigma_out = sigma_out_raw * sigma_out_scale * 0.05
    Z_hat = A @ x + Rinf * Rinf_vec + induc * induc_vec
    Z_hat_re = torch.cat([Z_hat[:N], Z_hat[:N]])
    Z_hat_im = torch.cat([Z_hat[N:], Z_hat[N:]])
    sigma_tot = torch.sqrt(dat['sigma_min'] ** 2 + sigma_res ** 2 + (alpha_prop * Z_hat) ** 2 + (alpha_re * Z_hat_re) ** 2
                           + (alpha_im * Z_hat_im) ** 2 + torch.cat([sigma_out, sigma_out]) ** 2)
    ups = ups_raw * 0.15
    dups = _dups(ups)
    # model
    lp = u.lp
    lp = lp + _inv_gamma(d0, 5.0, 5.0) + _inv_gamma(d1, 5.0, 5.0) + _inv_gamma(d2, 5.0, 5.0)
    lp = lp + _inv_gamma(ups_raw, dat['ups_alpha'], dat['ups_beta'])
    lp = lp + _std_normal(Rinf_raw) + _std_normal(induc_raw)
    lp = lp + _normal(q, 0.0, ups)
    lp = lp + _std_normal(dups)
    lp = lp + _normal(Z, Z_hat, sigma_tot)
    lp = lp + _std_normal(sigma_res_raw) + _std_normal(alpha_prop_raw) + _std_normal(alpha_re_raw) + _std_normal(alpha_im_raw)
    lp = lp + _exponential(sigma_out_raw, dat['sigma_out_lambda'])
    lp = lp + _inv_gamma(sigma_out_scale, dat['sigma_out_alpha'], dat['sigma_out_beta'])
    return lp, dict(Z_hat=Z_hat, sigma_tot=sigma_tot, q=q, sigma_out=sigma_out)


def parallel_lp(y, dat, jacobian):
    """Parallel_modelcode.txt: one parallel (admittance) distribution.  dat: N (= stacked length), K, A [N x K], Z [N],
    freq [N/2], L0, L1, L2, sigma_min, ups_alpha, ups_beta, induc_scale."""
    N, K = int(dat['N']), int(dat['K'])
    h = N // 2
    A, Z, freq = _t(dat['A']), _t(dat['Z']), _t(dat['freq'])
    L0, L1, L2 = _t(dat['L0']), _t(dat['L1']), _t(dat['L2'])
    Rinf_vec = torch.cat([torch.ones(h), torch.zeros(h)])
    induc_vec = torch.cat([torch.zeros(h), 2 * np.pi * freq])
    u = _Unpack(y, jacobian)
    Rinf_raw, induc_raw = u.take(), u.take()
    x = u.take(K)
    sigma_res_raw, alpha_prop_raw, alpha_re_raw, alpha_im_raw = u.take(), u.take(), u.take(), u.take()
    ups_raw = u.take(K)
    d0, d1, d2 = u.take(), u.take(), u.take()
    assert u.o == y.shape[0]
    Rinf = Rinf_raw * 100
    induc = induc_raw * dat['induc_scale']
    q = torch.sqrt(d0 * (L0 @ x) ** 2 + d1 * (L1 @ x) ** 2 + d2 * (L2 @ x) ** 2)
    sigma_res, alpha_prop, alpha_re, alpha_im = sigma_res_raw * 0.05, alpha_prop_raw * 0.05, alpha_re_raw * 0.05, alpha_im_raw * 0.05
    Y_hat = A @ x
    Y_re, Y_im = Y_hat[:h], Y_hat[h:]
    Z_hat_p = torch.cat([Y_re / (Y_re ** 2 + Y_im ** 2), -Y_im / (Y_re ** 2 + Y_im ** 2)])
    Z_hat = Z_hat_p + Rinf * Rinf_vec + induc * induc_vec
    Z_hat_re = torch.cat([Z_hat[:h], Z_hat[:h]])
    Z_hat_im = torch.cat([Z_hat[h:], Z_hat[h:]])
    sigma_tot = torch.sqrt(dat['sigma_min'] ** 2 + sigma_res ** 2 + (alpha_prop * Z_hat) ** 2 + (alpha_re * Z_hat_re) ** 2
                           + (alpha_im * Z_hat_im) ** 2)
    ups = ups_raw * 0.15
    dups = _dups(ups)
    lp = u.lp
    lp = lp + _inv_gamma(d0, 5.0, 5.0) + _inv_gamma(d1, 5.0, 5.0) + _inv_gamma(d2, 5.0, 5.0)
    lp = lp + _inv_gamma(ups_raw, dat['ups_alpha'], dat['ups_beta'])
    lp = lp + _std_normal(Rinf_raw) + _std_normal(induc_raw)
    lp = lp + _normal(q, 0.0, ups)
    lp = lp + _std_normal(dups)
    lp = lp + _normal(Z, Z_hat, sigma_tot)
    lp = lp + _std_normal(sigma_res_raw) + _std_normal(alpha_prop_raw) + _std_normal(alpha_re_raw) + _std_normal(alpha_im_raw)
    return lp, dict(Z_hat=Z_hat, sigma_tot=sigma_tot, q=q)


def series_lp(y, dat, jacobian, pos):
    """Series_modelcode.txt / Series_pos_modelcode.txt (control: this family IS pinned by stored Stan results)."""
    N, K = int(dat['N']), int(dat['K'])
    h = N // 2
    A, Z, freq = _t(dat['A']), _t(dat['Z']), _t(dat['freq'])
    L0, L1, L2 = _t(dat['L0']), _t(dat['L1']), _t(dat['L2'])
    Rinf_vec = torch.cat([torch.ones(h), torch.zeros(h)])
    induc_vec = torch.cat([torch.zeros(h), 2 * np.pi * freq])
    u = _Unpack(y, jacobian)
    Rinf_raw, induc_raw = u.take(), u.take()
    x = u.take(K, lower0=pos)
    sigma_res_raw, alpha_prop_raw, alpha_re_raw, alpha_im_raw = u.take(), u.take(), u.take(), u.take()
    ups_raw = u.take(K)
    d0, d1, d2 = u.take(), u.take(), u.take()
    Rinf = Rinf_raw * 100
    induc = induc_raw * dat['induc_scale']
    q = torch.sqrt(d0 * (L0 @ x) ** 2 + d1 * (L1 @ x) ** 2 + d2 * (L2 @ x) ** 2)
    sigma_res, alpha_prop, alpha_re, alpha_im = sigma_res_raw * 0.05, alpha_prop_raw * 0.05, alpha_re_raw * 0.05, alpha_im_raw * 0.05
    Z_hat = A @ x + Rinf * Rinf_vec + induc * induc_vec
    Z_hat_re = torch.cat([Z_hat[:h], Z_hat[:h]])
    Z_hat_im = torch.cat([Z_hat[h:], Z_hat[h:]])
    sigma_tot = torch.sqrt(dat['sigma_min'] ** 2 + sigma_res ** 2 + (alpha_prop * Z_hat) ** 2 + (alpha_re * Z_hat_re) ** 2
                           + (alpha_im * Z_hat_im) ** 2)
    ups = ups_raw * 0.15
    dups = _dups(ups)
    lp = u.lp
    lp = lp + _inv_gamma(d0, 5.0, 5.0) + _inv_gamma(d1, 5.0, 5.0) + _inv_gamma(d2, 5.0, 5.0)
    lp = lp + _inv_gamma(ups_raw, dat['ups_alpha'], dat['ups_beta'])
    lp = lp + _std_normal(Rinf_raw) + _std_normal(induc_raw)
    lp = lp + _normal(q, 0.0, ups)
    lp = lp + _std_normal(dups)
    lp = lp + _normal(Z, Z_hat, sigma_tot)
    lp = lp + _std_normal(sigma_res_raw) + _std_normal(alpha_prop_raw) + _std_normal(alpha_re_raw) + _std_normal(alpha_im_raw)
    return lp, dict(Z_hat=Z_hat, sigma_tot=sigma_tot, q=q)


def lp_and_grad(fn, theta, *args):
    """(lp, d lp / d theta, transformed parameters) of one of the functions above at the unconstrained point theta."""
    y = torch.tensor(np.asarray(theta, dtype=np.float64), requires_grad=True)
    lp, tp = fn(y, *args)
    (g,) = torch.autograd.grad(lp, y)
    return float(lp.detach()), g.numpy().copy(), {k: v.detach().numpy() for k, v in tp.items()}
