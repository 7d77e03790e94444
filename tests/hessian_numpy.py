"""Closed-form Hessian of the Series / Series_pos log-posterior (optimize mode: no Jacobian) on the unconstrained scale --
an independent numpy statement of what csrc/bdrt_newton.hip::newton_hessian_kernel computes, written from the Stan text
(bayes_drt/stan_model_files/Series_pos_modelcode.txt:24-69) for the tests; checked against central differences of the ORACLE's
gradient (tests/test_oracle_hessian.py).  Test infrastructure, not the product.

Layout of the unconstrained vector y (Stan declaration order, include/bdrt.h), D = 2 K + 9:
  0 Rinf_raw, 1 induc_raw, 2 .. 2+K x, then sigma_res_raw, alpha_prop_raw, alpha_re_raw, alpha_im_raw, ups_raw[K], d0, d1, d2.
Every parameter but x is <lower=0> (y = log raw); x is <lower=0> for Series_pos, free for Series."""
import numpy as np


def series_hessian(y, A, L, Z, w, sigma_min, ups_alpha, ups_beta, induc_scale=1.0, pos=True):
    """(lp, grad [D], H [D, D]) at the unconstrained point y.  A [2 Nf, K] stacked (re; im), L = (L0, L1, L2) mode-scaled [K, K],
    Z [2 Nf] stacked, w [Nf] = 2 pi f."""
    A = np.asarray(A, dtype=float); nf = len(w); K = A.shape[1]
    D = 2 * K + 9
    o_x, o_e, o_u, o_d = 2, 2 + K, 6 + K, 6 + 2 * K
    y = np.asarray(y, dtype=float)
    is_exp = np.ones(D, dtype=bool)
    if not pos:
        is_exp[o_x:o_x + K] = False
    r = np.where(is_exp, np.exp(y), y)                                     # raw (constrained) parameters
    # physical variables phi = c * r
    c = np.ones(D)
    c[0], c[1] = 100.0, induc_scale
    c[o_e:o_e + 4] = 0.05
    c[o_u:o_u + K] = 0.15
    phi = c * r
    Rinf, induc = phi[0], phi[1]
    x = phi[o_x:o_x + K]
    sres, ap, ar, ai = phi[o_e:o_e + 4]
    u = phi[o_u:o_u + K]
    d = phi[o_d:o_d + 3]
    g = np.zeros(D); H = np.zeros((D, D))                                  # w.r.t. phi first
    lp = 0.0
    # ---- priors stated on the raw scale (added after the chain rule to raw, below)
    # ---- q ~ normal(0, ups):  sum_k -log u_k - 1/2 q_k^2 / u_k^2,  q_k^2 = sum_i d_i (L_i x)_k^2
    v = [Li @ x for Li in L]
    iu2 = 1.0 / u ** 2
    q2 = sum(d[i] * v[i] ** 2 for i in range(3))
    lp += np.sum(-np.log(u) - 0.5 * q2 * iu2)
    sx, su, sd = slice(o_x, o_x + K), slice(o_u, o_u + K), slice(o_d, o_d + 3)
    for i in range(3):
        g[sx] += -d[i] * (L[i].T @ (v[i] * iu2))
        g[o_d + i] += -0.5 * np.sum(v[i] ** 2 * iu2)
        H[sx, sx] += -d[i] * (L[i].T @ (iu2[:, None] * L[i]))
        Hxu = 2.0 * d[i] * L[i].T * (v[i] / u ** 3)[None, :]              # [m, k] = 2 d_i L_i[k, m] v_ik / u_k^3
        H[sx, su] += Hxu; H[su, sx] += Hxu.T
        hxd = -(L[i].T @ (v[i] * iu2))
        H[sx, o_d + i] += hxd; H[o_d + i, sx] += hxd
        hud = v[i] ** 2 / u ** 3
        H[su, o_d + i] += hud; H[o_d + i, su] += hud
    g[su] += -1.0 / u + q2 / u ** 3
    H[su, su] += np.diag(1.0 / u ** 2 - 3.0 * q2 / u ** 4)
    # ---- dups ~ std_normal:  D_k = 1/2 - 1/4 (u_k + u_{k+2}) / u_{k+1},  k = 0 .. K-3
    for k in range(K - 2):
        a, b, cc = o_u + k, o_u + k + 1, o_u + k + 2
        s2 = u[k] + u[k + 2]
        Dk = 0.5 - 0.25 * s2 / u[k + 1]
        lp += -0.5 * Dk * Dk
        gD = {a: -0.25 / u[k + 1], cc: -0.25 / u[k + 1], b: 0.25 * s2 / u[k + 1] ** 2}
        hD = {(a, b): 0.25 / u[k + 1] ** 2, (cc, b): 0.25 / u[k + 1] ** 2, (b, b): -0.5 * s2 / u[k + 1] ** 3}
        for i_, gi in gD.items():
            g[i_] += -Dk * gi
            for j_, gj in gD.items():
                H[i_, j_] += -gi * gj
        for (i_, j_), hv in hD.items():
            H[i_, j_] += -Dk * hv
            if i_ != j_:
                H[j_, i_] += -Dk * hv
    # ---- likelihood: per frequency a function of xi = (zr, zi, sres, ap, ar, ai)
    Zh = A @ x
    zr = Zh[:nf] + Rinf; zi = Zh[nf:] + induc * w
    c0 = sigma_min ** 2 + sres ** 2
    S_re = c0 + (ap ** 2 + ar ** 2) * zr ** 2 + ai ** 2 * zi ** 2
    S_im = c0 + ar ** 2 * zr ** 2 + (ap ** 2 + ai ** 2) * zi ** 2
    e_re, e_im = Z[:nf] - zr, Z[nf:] - zi
    lp += np.sum(-0.5 * np.log(S_re) - 0.5 * e_re ** 2 / S_re - 0.5 * np.log(S_im) - 0.5 * e_im ** 2 / S_im)
    G6 = np.zeros((nf, 6)); H6 = np.zeros((nf, 6, 6))
    for (e, S, part) in ((e_re, S_re, 0), (e_im, S_im, 1)):
        f_e = -e / S; f_S = -0.5 / S + 0.5 * e ** 2 / S ** 2
        f_ee = -1.0 / S; f_eS = e / S ** 2; f_SS = 0.5 / S ** 2 - e ** 2 / S ** 3
        de = np.zeros((nf, 6)); de[:, part] = -1.0
        dS = np.zeros((nf, 6)); d2S = np.zeros((nf, 6, 6))
        if part == 0:
            dS[:, 0] = 2 * (ap ** 2 + ar ** 2) * zr; dS[:, 1] = 2 * ai ** 2 * zi
            dS[:, 3] = 2 * ap * zr ** 2; dS[:, 4] = 2 * ar * zr ** 2; dS[:, 5] = 2 * ai * zi ** 2
            d2S[:, 0, 0] = 2 * (ap ** 2 + ar ** 2); d2S[:, 1, 1] = 2 * ai ** 2
            d2S[:, 3, 3] = 2 * zr ** 2; d2S[:, 4, 4] = 2 * zr ** 2; d2S[:, 5, 5] = 2 * zi ** 2
            d2S[:, 0, 3] = d2S[:, 3, 0] = 4 * ap * zr; d2S[:, 0, 4] = d2S[:, 4, 0] = 4 * ar * zr; d2S[:, 1, 5] = d2S[:, 5, 1] = 4 * ai * zi
        else:
            dS[:, 0] = 2 * ar ** 2 * zr; dS[:, 1] = 2 * (ap ** 2 + ai ** 2) * zi
            dS[:, 3] = 2 * ap * zi ** 2; dS[:, 4] = 2 * ar * zr ** 2; dS[:, 5] = 2 * ai * zi ** 2
            d2S[:, 0, 0] = 2 * ar ** 2; d2S[:, 1, 1] = 2 * (ap ** 2 + ai ** 2)
            d2S[:, 3, 3] = 2 * zi ** 2; d2S[:, 4, 4] = 2 * zr ** 2; d2S[:, 5, 5] = 2 * zi ** 2
            d2S[:, 1, 3] = d2S[:, 3, 1] = 4 * ap * zi; d2S[:, 0, 4] = d2S[:, 4, 0] = 4 * ar * zr; d2S[:, 1, 5] = d2S[:, 5, 1] = 4 * ai * zi
        dS[:, 2] = 2 * sres; d2S[:, 2, 2] = 2.0
        G6 += f_e[:, None] * de + f_S[:, None] * dS
        H6 += (f_ee[:, None, None] * de[:, :, None] * de[:, None, :] + f_eS[:, None, None] * (de[:, :, None] * dS[:, None, :] + dS[:, :, None] * de[:, None, :])
               + f_SS[:, None, None] * dS[:, :, None] * dS[:, None, :] + f_S[:, None, None] * d2S)
    # chain to phi: zr = A_re x + Rinf, zi = A_im x + induc w.   B = d(zr, zi)/d(Rinf, induc, x)
    Are, Aim = A[:nf], A[nf:]
    Br = np.zeros((nf, D)); Bi = np.zeros((nf, D))
    Br[:, 0] = 1.0; Br[:, sx] = Are
    Bi[:, 1] = w; Bi[:, sx] = Aim
    g += Br.T @ G6[:, 0] + Bi.T @ G6[:, 1]
    H += Br.T @ (H6[:, 0, 0][:, None] * Br) + Bi.T @ (H6[:, 1, 1][:, None] * Bi) + Br.T @ (H6[:, 0, 1][:, None] * Bi) + Bi.T @ (H6[:, 0, 1][:, None] * Br)
    for a in range(4):
        ja = o_e + a
        g[ja] += np.sum(G6[:, 2 + a])
        col = Br.T @ H6[:, 0, 2 + a] + Bi.T @ H6[:, 1, 2 + a]
        H[:, ja] += col; H[ja, :] += col
        for b in range(4):
            H[ja, o_e + b] += np.sum(H6[:, 2 + a, 2 + b])
    # ---- chain rule phi -> raw (diagonal scaling), priors on the raw scale, raw -> unconstrained
    g_raw = c * g
    H_raw = c[:, None] * H * c[None, :]
    lp += -0.5 * r[0] ** 2 - 0.5 * r[1] ** 2 - 0.5 * np.sum(r[o_e:o_e + 4] ** 2)
    for j in (0, 1, o_e, o_e + 1, o_e + 2, o_e + 3):
        g_raw[j] += -r[j]; H_raw[j, j] += -1.0
    rd = r[sd]
    lp += np.sum(-6.0 * np.log(rd) - 5.0 / rd)
    g_raw[sd] += -6.0 / rd + 5.0 / rd ** 2
    H_raw[sd, sd] += np.diag(6.0 / rd ** 2 - 10.0 / rd ** 3)
    ru = r[su]
    lp += np.sum(-(ups_alpha + 1.0) * np.log(ru) - ups_beta / ru)
    g_raw[su] += -(ups_alpha + 1.0) / ru + ups_beta / ru ** 2
    H_raw[su, su] += np.diag((ups_alpha + 1.0) / ru ** 2 - 2.0 * ups_beta / ru ** 3)
    # the ~ statements drop constants; Stan's normal(0, ups) on q keeps -log(ups) = -log(0.15) - log(ups_raw): the constant is dropped
    t = np.where(is_exp, r, 1.0)
    g_y = t * g_raw
    H_y = t[:, None] * H_raw * t[None, :] + np.diag(np.where(is_exp, g_y, 0.0))
    return lp, g_y, H_y
