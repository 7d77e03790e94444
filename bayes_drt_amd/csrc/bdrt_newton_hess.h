// bdrt_newton_hess.h -- closed-form Hessian of the Series / Series_pos log-posterior (optimize mode: no Jacobian term) on the
// unconstrained scale, for the Levenberg-Marquardt iteration of bdrt_newton.hip (StanModel.optimizing, reference
// bayes_drt/inversion.py:1216; density: stan_model_files/Series_pos_modelcode.txt:24-69).
//
// Rounds 1-5 took the Hessian from central differences of the gradient: 2 D evaluations per Newton round (662 at D = 331, 167 841
// per K = 161 fit).  The density has a short closed form:
//   likelihood   per frequency a function of xi = (Z_hat_re, Z_hat_im, sigma_res, alpha_prop, alpha_re, alpha_im): a 6 x 6 block
//                H6_n from the first / second derivatives of the two variances; Z_hat = A x + Rinf + i w induc is linear, so the
//                x-x part is  A^T C A  with C block-diagonal 2 x 2 per frequency -- the one dense product (4 Nf K^2 MACs);
//   q ~ N(0, ups)  x-x: -sum_i d_i L_i^T diag(1/u^2) L_i (band 12), x-ups: band 6, x-d_i and ups-d_i: vectors, ups-ups: diagonal;
//   dups ~ N(0,1)  ups-ups: pentadiagonal;   priors on the raw scale: diagonal;
//   chain rule     phi = c r (100 Rinf_raw, 0.05 error raws, 0.15 ups_raw), r = exp(y) for every <lower=0> parameter:
//                  H_y = s_i s_j H_phi + delta_ij (t_i^2 h_prior_i + [exp] g_y,i),  s = c t, t = r or 1.
// Two kernels per round: `newton_hess_prep_kernel` (one workgroup per fit: forward quantities, the per-frequency blocks, every vector
// and band of the border) and `newton_hess_fill_kernel` (16 rows of one fit per workgroup: the dense product from LDS-staged columns
// of A, everything else by index class from the prepared vectors; writes the block lower triangle of H and of M = -H + lam I).  tests/hessian_numpy.py states the
// same formulas in numpy (on the log scale); both are held to central differences of the ORACLE's gradient (tests/test_oracle_hessian.py,
// tests/test_gpu_hessian.py).  Families beyond the single series distribution keep the finite-difference Hessian.
//
// Coordinates of the iteration.  Stan's `optimizing` works on y = log x for the <lower=0> coefficients.  A spectrum's gamma has compact
// support, so most coefficients of the MAP sit AT the bound: on the log scale they run to -infinity at one unit per Newton step (a
// boundary optimum is approached linearly, never reached), one after the other along the edge of the support -- 185 of the 250 rounds of a
// K = 161 fit (profiles/r06/README.md).  Once the damping has dropped to 1e-4 (the iteration is in its Newton regime) a fit therefore
// switches its coefficients to the LINEAR scale z = x with a floor (1e-14 of the largest coefficient): H_z = H_phi, g_z = g_y / x, a
// coefficient at the floor whose gradient points below it is held there for the round (projected Newton: its row and column leave the
// system), the trial points are y = log(max(x + s, floor)).  Same stationary point (the convergence test stays |g_y|_inf < tol, and
// g_y = x g_x vanishes at the floor); 105 rounds instead of 249 at K = 161, 77 instead of 128 at K = 101, 69 instead of 84 at K = 81.
#pragma once

namespace bdrt {

// per-fit workspace (doubles): offsets from the problem's sizes
struct HessLayout {
    int nf, K, D;
    int sc, dex, tz, act, cf, hz, hss, bR, bI, bS, h0, iu2, cu3, hxd, hud, uu, pxx, total;
    __host__ __device__ HessLayout(int nf_, int K_) : nf(nf_), K(K_), D(2 * K_ + 9)
    {
        int o = 0;
        sc = o; o += D;             // s_i = c_i t_i
        dex = o; o += D;            // diagonal extra: t_i^2 * (raw-scale prior curvature) + [exp coordinate] g_y,i
        tz = o; o += D;             // d y_i / d z_i of the iteration's coordinates z: 1, or 1 / x_k for a coefficient on the linear scale
        act = o; o += D;            // 1.0: coefficient held at its floor this round (linear scale, gradient pointing below the floor)
        cf = o; o += 3 * nf;        // c11, c12, c22
        hz = o; o += 8 * nf;        // H6[n][0][2 + a] (a < 4), then H6[n][1][2 + a]
        hss = o; o += 16;
        bR = o; o += K; bI = o; o += K; bS = o; o += 4 * K;
        h0 = o; o += 16;            // h00, h01, h11, h0s[4], h1s[4]; [11]: floor of the coefficients on the linear scale
        iu2 = o; o += K;
        cu3 = o; o += 3 * K;        // 2 d_i v_ik / u_k^3
        hxd = o; o += 3 * K;        // -(L_i^T (v_i / u^2))_m
        hud = o; o += 3 * K;        // v_ik^2 / u_k^3
        uu = o; o += 3 * K;         // ups-ups band in phi space: main, first, second diagonal
        pxx = o; o += (2 * MAXBW + 1) * K;   // x-x band of the q-prior: pxx[dm][m] = entry (m, m + dm), dm = 0 .. 2 MAXBW
        total = (o + 7) & ~7;
    }
};

struct HessArgs {
    const DevProblem *P;
    const double *x, *g;        // [n_fits][D] current points / gradients (unconstrained)
    const int *spec;            // [n_fits] spectrum of each fit
    double *ws;                 // [n_fits][layout.total]
    double *H, *M;              // [n_fits][Dp][Dp]
    int D, Dp;
    int lin;                    // the coefficients x of a Series_pos fit are iterated on the LINEAR scale (see hess_prep)
    long long *prof;            // BDRT_NEWTON_PROF=1: cycle stamps of workgroup 0 (slots 16 .. 31 prep, 32 .. 39 fill); nullptr otherwise
};

// element (n, m) of half `part` (0 re, 1 im) of A: from the Toeplitz generators or the plain copy
__device__ __forceinline__ double hess_A(const DevBlock &B, int nf, int part, int n, int m)
{
    if (B.tg) return B.tg[(size_t)part * (nf + B.K - 1) + n - m + B.K - 1];
    return B.Ad[((size_t)part * nf + n) * B.K + m];
}

// whether the closed form applies to a problem (single series distribution, no outlier model, banded Toeplitz L, not streamed, and
// the fill kernel's LDS tile fits)
__host__ inline bool hess_analytic_ok(const DevProblem &P)
{
    if (!(P.nblocks == 1 && !P.blk[0].is_parallel && P.outlier_mode == 0 && !P.use_x_sum && P.blk[0].toep && !P.big &&
          (P.blk[0].tg || P.blk[0].Ad) && P.nf <= 256)) return false;
    const size_t nf = P.nf, K = P.blk[0].K;
    const size_t fill = 35 * nf + 40 + 2 * (nf + K) + 16 * ((2 * K + 9 + 15) & ~(size_t)15);
    return fill * sizeof(double) <= 150 * 1024;
}

constexpr int HP_NT = 512;
static_assert(2 * MAXBW + 1 == 13, "the LDS copy of the band coefficients is laid out as [3][13]");

// ---- prep: one workgroup per fit -------------------------------------------------------------------------------------------
// LDS (doubles): x[K], u[K], v[3][K], w[3][K] (v_i / u^2), 16 scalars, T[3][13], the Toeplitz generators of A [2][nf + K - 1] (when A
// has them), then c11 / c12 / c22 [3][nf] and the cross terms [8][nf] of the per-frequency blocks.
__host__ __device__ inline size_t hess_prep_lds_doubles(int nf, int K) { return (size_t)9 * K + 16 + 40 + 2 * (size_t)(nf + K) + 14 * (size_t)nf + 16 * (HP_NT / 64); }

__device__ inline void hess_prep(const HessArgs &a, int f, double *lds)
{
    const DevProblem &P = *a.P;
    const DevBlock &B = P.blk[0];
    const int tid = threadIdx.x, nf = P.nf, K = B.K, D = a.D;
    const HessLayout L(nf, K);
    const double *y = a.x + (size_t)f * D, *gy = a.g + (size_t)f * D;
    double *W = a.ws + (size_t)f * L.total;
    const int o_x = B.o_x, o_e = P.o_err, o_u = B.o_ups, o_d = B.o_d;
    double *xs = lds, *us = xs + K, *vs = us + K, *wv = vs + 3 * K, *iu2s = wv + 3 * K, *sca = iu2s + K;   // sca: Rinf, induc, sres, ap, ar, ai, d0, d1, d2
    double *Ts = sca + 16, *tgs = Ts + 40, *cfs = tgs + 2 * (nf + K), *hzs = cfs + 3 * nf, *wfs = hzs + 8 * nf, *wcs = wfs + nf, *red = wcs + 2 * nf;
    long long tp_ = (a.prof && blockIdx.x == 0 && tid == 0) ? clock64() : 0;
#define HP_STAMP(k) do { if (a.prof && blockIdx.x == 0 && tid == 0) { const long long n_ = clock64(); a.prof[k] += n_ - tp_; tp_ = n_; } } while (0)
    const int glen = nf + K - 1;
    // (every field of the problem that the loops below read, once, into registers: through the references a store to W could
    //  alias them, and each use would be a dependent pair of global loads)
    const double *const tgp = B.tg, *const Adp = B.Ad, *const Atp = B.At, *const wp = P.w;
    const double sigma_min = P.sigma_min, ups_alpha = P.ups_alpha, ups_beta = P.ups_beta, induc_scale = P.induc_scale;
    const bool pos = B.is_pos != 0, toepA = tgp != nullptr, lin = a.lin != 0;
    if (tid < 3 * (2 * MAXBW + 1)) Ts[tid] = B.T[tid / (2 * MAXBW + 1)][tid % (2 * MAXBW + 1)];
    if (toepA) for (int e = tid; e < 2 * glen; e += HP_NT) tgs[e] = tgp[e];
    for (int n = tid; n < nf; n += HP_NT) wfs[n] = wp[n];
    // ---- constrained values, scalings
    for (int j = tid; j < D; j += HP_NT) {
        const bool isx = j >= o_x && j < o_x + K;
        const bool is_exp = !isx || pos;
        const double r = is_exp ? exp(y[j]) : y[j];
        double c = 1.0, hp = 0.0;                                          // scale raw -> phi, raw-scale prior curvature
        if (j == 0) { c = 100.0; hp = -1.0; }
        else if (j == 1) { c = induc_scale; hp = -1.0; }
        else if (j >= o_e && j < o_e + 4) { c = 0.05; hp = -1.0; }
        else if (j >= o_u && j < o_u + K) { c = 0.15; hp = (ups_alpha + 1.0) / (r * r) - 2.0 * ups_beta / (r * r * r); }
        else if (j >= o_d && j < o_d + 3) { hp = 6.0 / (r * r) - 10.0 / (r * r * r); }
        // (lin: the coefficients on the linear scale -- no chain rule through the exponential, no gradient term on the diagonal)
        const bool linx = lin && isx && pos;
        const double t = (is_exp && !linx) ? r : 1.0;
        W[L.sc + j] = c * t;
        W[L.dex + j] = t * t * hp + ((is_exp && !linx) ? gy[j] : 0.0);
        W[L.tz + j] = linx ? 1.0 / r : 1.0;
        const double phi = c * r;
        if (isx) xs[j - o_x] = phi;
        else if (j >= o_u && j < o_u + K) us[j - o_u] = phi;
        else if (j < 2) sca[j] = phi;
        else if (j >= o_e && j < o_e + 4) sca[2 + j - o_e] = phi;
        else if (j >= o_d && j < o_d + 3) sca[6 + j - o_d] = phi;
    }
    __syncthreads();
    // the floor of the linear scale (1e-14 of the largest coefficient) and the coefficients held at it this round
    {
        double mx = 0.0;
        for (int k = tid; k < K; k += HP_NT) mx = fmax(mx, xs[k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = red[0];
        for (int w_ = 1; w_ < HP_NT / 64; ++w_) mx = fmax(mx, red[w_]);
        const double floor_x = 1e-14 * fmax(mx, 1e-300);
        if (tid == 0) W[L.h0 + 11] = floor_x;
        for (int j = tid; j < D; j += HP_NT) {
            const bool isx = j >= o_x && j < o_x + K;
            double av = 0.0;
            if (lin && pos && isx) { const double xk = xs[j - o_x]; av = (xk <= 2.0 * floor_x && gy[j] < 0.0) ? 1.0 : 0.0; }   // (g_x = g_y / x: same sign)
            W[L.act + j] = av;
        }
        __syncthreads();
    }
    HP_STAMP(16);
    const double Rinf = sca[0], induc = sca[1], sres = sca[2], ap = sca[3], ar = sca[4], ai = sca[5];
    const double d3[3] = {sca[6], sca[7], sca[8]};
    // ---- v_i = L_i x (13-tap convolutions), w_i = v_i / u^2
    for (int k = tid; k < K; k += HP_NT) {
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll
        for (int d = 0; d < 2 * MAXBW + 1; ++d) {
            const int m = k + d - MAXBW;
            const double xm = (m >= 0 && m < K) ? xs[m] : 0.0;
            v0 = fma(Ts[d], xm, v0); v1 = fma(Ts[13 + d], xm, v1); v2 = fma(Ts[26 + d], xm, v2);
        }
        const double u = us[k], iu2 = 1.0 / (u * u);
        vs[k] = v0; vs[K + k] = v1; vs[2 * K + k] = v2;
        wv[k] = v0 * iu2; wv[K + k] = v1 * iu2; wv[2 * K + k] = v2 * iu2;
        iu2s[k] = iu2;
        W[L.iu2 + k] = iu2;
    }
    HP_STAMP(17);
    // ---- Z_hat = A x + offsets (thread per stacked row): parked in red .. (2 nf doubles; hzs is free until the blocks are written)
    double *zh = hzs;
    for (int r = tid; r < 2 * nf; r += HP_NT) {
        const int part = r >= nf, n = r - part * nf;
        double s0 = 0.0, s1 = 0.0;
        if (toepA) {
            const double *gq = tgs + part * glen + n + K - 1;
            int m = 0;
            for (; m + 1 < K; m += 2) { s0 = fma(gq[-m], xs[m], s0); s1 = fma(gq[-m - 1], xs[m + 1], s1); }
            if (m < K) s0 = fma(gq[-m], xs[m], s0);
        } else if (Atp) {
#pragma unroll 8
            for (int m = 0; m < K; ++m) s0 = fma(Atp[(size_t)m * 2 * nf + r], xs[m], s0);
        } else {
#pragma unroll 8
            for (int m = 0; m < K; ++m) s0 = fma(Adp[(size_t)r * K + m], xs[m], s0);
        }
        zh[r] = s0 + s1 + (part ? induc * wfs[n] : Rinf);
    }
    __syncthreads();
    HP_STAMP(18);
    // ---- q-prior pieces, the ups-ups band and the x-x band of the prior (phi space)
    for (int k = tid; k < K; k += HP_NT) {
        const double u = us[k], iu = 1.0 / u, iu2 = iu * iu, iu3 = iu2 * iu;
        double q2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double v = vs[i * K + k];
            q2 += d3[i] * v * v;
            W[L.cu3 + i * K + k] = 2.0 * d3[i] * v * iu3;
            W[L.hud + i * K + k] = v * v * iu3;
            // -(L_i^T (v_i / u^2))_k = -sum_j L_i[j][k] w_ij,  L_i[j][k] = T_i[k - j + MAXBW]
            double s = 0.0;
#pragma unroll
            for (int d = 0; d < 2 * MAXBW + 1; ++d) {
                const int j = k - (d - MAXBW);
                if (j >= 0 && j < K) s = fma(Ts[13 * i + d], wv[i * K + j], s);
            }
            W[L.hxd + i * K + k] = -s;
        }
        // dups_j = 1/2 - 1/4 (u_j + u_{j+2}) / u_{j+1}, j = 0 .. K-3: element k is the left end of dups_k, the centre of
        // dups_{k-1} and the right end of dups_{k-2}
        double h0 = iu2 - 3.0 * q2 * iu2 * iu2, h1 = 0.0, h2 = 0.0;       // (k,k), (k,k+1), (k,k+2)
        auto dk = [&](int j, double &Dj, double &ga, double &gb, double &hab, double &hbb) {
            const double ub = us[j + 1], s2 = us[j] + us[j + 2], ib = 1.0 / ub;
            Dj = 0.5 - 0.25 * s2 * ib; ga = -0.25 * ib; gb = 0.25 * s2 * ib * ib; hab = 0.25 * ib * ib; hbb = -0.5 * s2 * ib * ib * ib;
        };
        double Dj, ga, gb, hab, hbb;
        if (k + 2 < K) {                       // dups_k: a = k, b = k+1, c = k+2
            dk(k, Dj, ga, gb, hab, hbb);
            h0 += -ga * ga;
            h1 += -ga * gb - Dj * hab;
            h2 += -ga * ga;                    // (g_a = g_c)
        }
        if (k >= 1 && k + 1 < K) {             // dups_{k-1}: b = k, c = k+1
            dk(k - 1, Dj, ga, gb, hab, hbb);
            h0 += -gb * gb - Dj * hbb;
            h1 += -gb * ga - Dj * hab;
        }
        if (k >= 2) {                          // dups_{k-2}: c = k
            dk(k - 2, Dj, ga, gb, hab, hbb);
            h0 += -ga * ga;
        }
        W[L.uu + k] = h0; W[L.uu + K + k] = h1; W[L.uu + 2 * K + k] = h2;
    }
    HP_STAMP(19);
    // x-x band: entry (m, m + dm) = -sum_i d_i sum_j T_i[m - j + MAXBW] T_i[m + dm - j + MAXBW] / u_j^2,  j = m + dm - MAXBW .. m + MAXBW
    for (int e = tid; e < (2 * MAXBW + 1) * K; e += HP_NT) {
        const int dm = e / K, m = e - dm * K;
        double s = 0.0;
        if (m + dm < K) {
            // j = m + dm - MAXBW .. m + MAXBW: 13 - dm terms; indices clamped, out-of-range terms weighted zero
            double w_[2 * MAXBW + 1];
#pragma unroll
            for (int t = 0; t < 2 * MAXBW + 1; ++t) {
                const int j = m + dm - MAXBW + t;
                const bool in = t < 2 * MAXBW + 1 - dm && j >= 0 && j < K;
                w_[t] = in ? iu2s[j < 0 ? 0 : (j >= K ? K - 1 : j)] : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 2 * MAXBW + 1; ++t) {
                // p = m - j + MAXBW = 2 MAXBW - dm - t, q = p + dm = 2 MAXBW - t   (p >= 0 exactly when the term exists)
                const int q = 2 * MAXBW - t, pp = q - dm;
                const int pc = pp < 0 ? 0 : pp;
                s += (d3[0] * Ts[pc] * Ts[q] + d3[1] * Ts[13 + pc] * Ts[13 + q] + d3[2] * Ts[26 + pc] * Ts[26 + q]) * w_[t];
            }
        }
        W[L.pxx + e] = -s;
    }
    HP_STAMP(20);
    // ---- the per-frequency 6 x 6 blocks
    const double *Zm = P.Z + (size_t)a.spec[f] * 2 * nf;
    double h6r[8];                              // this thread's frequency: cross terms kept until Z_hat has been read by everybody
    double c11 = 0.0, c12 = 0.0, c22 = 0.0, hs[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) hs[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) h6r[i] = 0.0;
    const int n0 = tid;                         // (nf <= 256 < HP_NT)
    if (n0 < nf) {
        const double zr = zh[n0], zi = zh[nf + n0];
        const double c0 = sigma_min * sigma_min + sres * sres;
        double H6[6][6];
#pragma unroll
        for (int i = 0; i < 6; ++i)
#pragma unroll
            for (int j = 0; j < 6; ++j) H6[i][j] = 0.0;
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            double dS[6], d2d[6], S, e;
            double x03 = 0.0, x04 = 0.0, x15 = 0.0, x13 = 0.0;                 // off-diagonal second derivatives of S
            if (part == 0) {
                S = c0 + (ap * ap + ar * ar) * zr * zr + ai * ai * zi * zi; e = Zm[n0] - zr;
                dS[0] = 2 * (ap * ap + ar * ar) * zr; dS[1] = 2 * ai * ai * zi; dS[3] = 2 * ap * zr * zr; dS[4] = 2 * ar * zr * zr; dS[5] = 2 * ai * zi * zi;
                d2d[0] = 2 * (ap * ap + ar * ar); d2d[1] = 2 * ai * ai; d2d[3] = 2 * zr * zr; d2d[4] = 2 * zr * zr; d2d[5] = 2 * zi * zi;
                x03 = 4 * ap * zr; x04 = 4 * ar * zr; x15 = 4 * ai * zi;
            } else {
                S = c0 + ar * ar * zr * zr + (ap * ap + ai * ai) * zi * zi; e = Zm[nf + n0] - zi;
                dS[0] = 2 * ar * ar * zr; dS[1] = 2 * (ap * ap + ai * ai) * zi; dS[3] = 2 * ap * zi * zi; dS[4] = 2 * ar * zr * zr; dS[5] = 2 * ai * zi * zi;
                d2d[0] = 2 * ar * ar; d2d[1] = 2 * (ap * ap + ai * ai); d2d[3] = 2 * zi * zi; d2d[4] = 2 * zr * zr; d2d[5] = 2 * zi * zi;
                x13 = 4 * ap * zi; x04 = 4 * ar * zr; x15 = 4 * ai * zi;
            }
            dS[2] = 2 * sres; d2d[2] = 2.0;
            const double iS = 1.0 / S;
            const double f_S = -0.5 * iS + 0.5 * e * e * iS * iS, f_ee = -iS, f_eS = e * iS * iS, f_SS = 0.5 * iS * iS - e * e * iS * iS * iS;
            // de = -1 at index `part`
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    double h = f_SS * dS[i] * dS[j];
                    if (i == j) h += f_S * d2d[i];
                    if (i == part) h += -f_eS * dS[j];
                    if (j == part) h += -f_eS * dS[i];
                    if (i == part && j == part) h += f_ee;
                    H6[i][j] += h;
                }
            H6[0][3] += f_S * x03; H6[3][0] += f_S * x03;
            H6[0][4] += f_S * x04; H6[4][0] += f_S * x04;
            H6[1][5] += f_S * x15; H6[5][1] += f_S * x15;
            H6[1][3] += f_S * x13; H6[3][1] += f_S * x13;
        }
        c11 = H6[0][0]; c12 = H6[0][1]; c22 = H6[1][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) { h6r[q] = H6[0][2 + q]; h6r[4 + q] = H6[1][2 + q]; }
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = 0; q < 4; ++q) hs[4 * p + q] = H6[2 + p][2 + q];
    }
    __syncthreads();                            // everybody has read Z_hat: the hz region takes the cross terms
    HP_STAMP(21);
    if (n0 < nf) {
        cfs[n0] = c11; cfs[nf + n0] = c12; cfs[2 * nf + n0] = c22;
        wcs[n0] = wfs[n0] * c12; wcs[nf + n0] = wfs[n0] * c22;
        W[L.cf + n0] = c11; W[L.cf + nf + n0] = c12; W[L.cf + 2 * nf + n0] = c22;
#pragma unroll
        for (int q = 0; q < 8; ++q) hzs[q * nf + n0] = h6r[q];
    }
    // the sixteen error-structure sums: block reduction through LDS
    {
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            double t = hs[i];
            t = sum32(t); t += __shfl_xor(t, 32);
            if (lane == 0) red[i * (HP_NT / 64) + wave] = t;
        }
    }
    __syncthreads();
    if (tid < 16) {
        double t = 0.0;
        for (int w_ = 0; w_ < HP_NT / 64; ++w_) t += red[tid * (HP_NT / 64) + w_];
        W[L.hss + tid] = t;
    }
    HP_STAMP(22);
    // ---- border vectors and border scalars, from the LDS copies.  Work item (q, m): vector q of  R-x, I-x, s_0..3-x  at basis function m,
    //      sum_n alpha_q[n] A_re[n][m] + beta_q[n] A_im[n][m]
    for (int e = tid; e < 6 * K; e += HP_NT) {
        const int q = e / K, m = e - q * K;
        const double *al = q == 0 ? cfs : (q == 1 ? wcs : hzs + (size_t)(q - 2) * nf);
        const double *be = q == 0 ? cfs + nf : (q == 1 ? wcs + nf : hzs + (size_t)(4 + q - 2) * nf);
        // (toepA is uniform: two copies of the loop instead of a select per element; eight terms requested per trip -- the loop is bound
        //  by the latency of its LDS reads, not by their number)
        double s0 = 0.0, s1 = 0.0;
        if (toepA) {
            const double *g0 = tgs + K - 1 - m, *g1 = tgs + glen + K - 1 - m;
            int n = 0;
            for (; n + 7 < nf; n += 8) {
                double a_[8], b_[8], p_[8], q_[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { a_[u] = al[n + u]; b_[u] = be[n + u]; p_[u] = g0[n + u]; q_[u] = g1[n + u]; }
#pragma unroll
                for (int u = 0; u < 8; u += 2) { s0 = fma(a_[u], p_[u], fma(b_[u], q_[u], s0)); s1 = fma(a_[u + 1], p_[u + 1], fma(b_[u + 1], q_[u + 1], s1)); }
            }
            for (; n < nf; ++n) s0 = fma(al[n], g0[n], fma(be[n], g1[n], s0));
        } else {
#pragma unroll 4
            for (int n = 0; n < nf; ++n) s0 = fma(al[n], Adp[(size_t)n * K + m], fma(be[n], Adp[((size_t)nf + n) * K + m], s0));
        }
        W[(q == 0 ? L.bR : (q == 1 ? L.bI : L.bS + (q - 2) * K)) + m] = s0 + s1;
    }
    HP_STAMP(23);
    // the eleven border scalars: sums over the frequencies of cf_0, w cf_1, w^2 cf_2, hz_0..3, w hz_4..7 -- a segment of 32 lanes per sum,
    // strided terms and a butterfly.  (One thread per sum, eleven lanes of one wave each on its own branch of the selection, was the tail of
    // the kernel: 20 of its 49 us, invisible in the stamps of thread 0.)
    {
        const int q = tid >> 5, l = tid & 31;
        if (q < 11) {
            const double *src = q < 3 ? cfs + (size_t)q * nf : hzs + (size_t)(q - 3) * nf;
            const int pw = q < 3 ? q : (q < 7 ? 0 : 1);
            double t = 0.0;
            for (int n = l; n < nf; n += 32) {
                const double wn = wfs[n], v = src[n];
                t += pw == 0 ? v : (pw == 1 ? v * wn : v * wn * wn);
            }
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o);
            if (l == 0) W[L.h0 + q] = t;
        }
    }
    HP_STAMP(24);
#undef HP_STAMP
}

// ---- fill: rows r0 .. r0+15 of fit f ----------------------------------------------------------------------------------------
// LDS: ArT[nf][16], AiT[nf][16] (this tile's columns of A), cfs[3][nf], T[3][13], the generators of A, the tile [16][Dp] in phi space.
// Returns true when a non-finite entry was seen.
__host__ __device__ inline size_t hess_fill_lds_doubles(int nf, int K) { return (size_t)35 * nf + 40 + 2 * (size_t)(nf + K) + 16 * (size_t)((2 * K + 9 + 15) & ~15); }
__device__ inline bool hess_fill(const HessArgs &a, int f, int r0, double lam, double *lds)
{
    const DevProblem &P = *a.P;
    const DevBlock &B = P.blk[0];
    const int tid = threadIdx.x, nf = P.nf, K = B.K, D = a.D, Dp = a.Dp;
    const HessLayout L(nf, K);
    const double *__restrict__ W = a.ws + (size_t)f * L.total;
    double *__restrict__ H = a.H + (size_t)f * Dp * Dp, *__restrict__ M = a.M + (size_t)f * Dp * Dp;
    const int o_x = B.o_x, o_e = P.o_err, o_u = B.o_ups, o_d = B.o_d;
    long long tp_ = (a.prof && blockIdx.x == 8 && blockIdx.y == 0 && tid == 0) ? clock64() : 0;
#define HF_STAMP(k) do { if (a.prof && blockIdx.x == 8 && blockIdx.y == 0 && tid == 0) { const long long n_ = clock64(); a.prof[k] += n_ - tp_; tp_ = n_; } } while (0)
    double *ArT = lds, *AiT = ArT + (size_t)nf * 16, *cfs = AiT + (size_t)nf * 16, *Ts = cfs + 3 * nf;
    if (tid < 3 * (2 * MAXBW + 1)) Ts[tid] = B.T[tid / (2 * MAXBW + 1)][tid % (2 * MAXBW + 1)];
    // (the problem's pointers once, into registers: behind the references every use is a dependent pair of global loads, because the
    //  stores to H and M could alias them)
    const double *const tgp = B.tg, *const Adp = B.Ad;
    const int glen = nf + K - 1;
    double *tgs = Ts + 40;                                     // the Toeplitz generators of A [2][glen], when A has them
    double *tile = tgs + 2 * (nf + K);                         // [16][Dp]: the tile's entries in phi space, assembled in three passes
    const bool toepA = tgp != nullptr;
    if (toepA) { for (int e = tid; e < 2 * glen; e += HP_NT) tgs[e] = tgp[e]; __syncthreads(); }
    const bool has_x = r0 < o_x + K && r0 + 16 > o_x;          // some row of the tile is an x row
    if (has_x) {
        for (int e = tid; e < nf * 16; e += HP_NT) {
            const int n = e >> 4, rr = e & 15, m = r0 + rr - o_x;
            const bool in = m >= 0 && m < K;
            const int mm = in ? m : 0;
            double vr, vi;
            if (toepA) { vr = tgs[n - mm + K - 1]; vi = tgs[glen + n - mm + K - 1]; }
            else { vr = Adp[(size_t)n * K + mm]; vi = Adp[((size_t)nf + n) * K + mm]; }
            ArT[e] = in ? vr : 0.0;
            AiT[e] = in ? vi : 0.0;
        }
        for (int e = tid; e < 3 * nf; e += HP_NT) cfs[e] = W[L.cf + e];
    }
    __syncthreads();
    HF_STAMP(32);
    // Only the block lower triangle is written (columns j < r0 + 16): the factorisation reads the lower triangle and the diagonal blocks.
    // A thread owns the column pair (j, j + NH), NH = half the tile's columns rounded up to 64: the 32 broadcast reads of the tile's A
    // columns per frequency serve 64 multiply-adds instead of 32 (the product is bound by the LDS bandwidth of those reads).
    bool bad = false;
    const int ncol = r0 + 16 < Dp ? r0 + 16 : Dp;
    const int NH = ((ncol + 1) / 2 + 63) & ~63;                 // 64, 128 or 192 (Dp <= 384 here; wider: 256 and one group)
    // The product is bound by the LATENCY of its LDS reads while few waves run it, so the frequencies are split over NG groups of NH
    // threads (all eight waves busy); the groups' partial sums meet in the tile in a fixed order.
    const int NG = HP_NT / NH > 0 ? HP_NT / NH : 1;
    for (int e = tid; e < 16 * Dp; e += HP_NT) tile[e] = 0.0;
    __syncthreads();
    {
        const bool owner = tid < NG * NH;
        const int grp = owner ? tid / NH : 0, j0 = owner ? tid - grp * NH : 0;
        const int n_lo = (int)((long)nf * grp / NG), n_hi = (int)((long)nf * (grp + 1) / NG);
        const int jc[2] = {j0, j0 + NH};
        double acc[2][16];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) acc[h][rr] = 0.0;
        const bool jx0 = owner && jc[0] >= o_x && jc[0] < o_x + K && jc[0] < ncol, jx1 = owner && jc[1] >= o_x && jc[1] < o_x + K && jc[1] < ncol;
        if (has_x && (jx0 || jx1)) {
            const int m0 = jx0 ? jc[0] - o_x : 0, m1 = jx1 ? jc[1] - o_x : 0;
            const double z0 = jx0 ? 1.0 : 0.0, z1 = jx1 ? 1.0 : 0.0;
            // (two copies of the loop, one per address space of A's source: a select between an LDS and a global pointer makes every
            //  read a FLAT load with a full wait behind it)
            auto product = [&](auto ael) {
                for (int n = n_lo; n < n_hi; ++n) {
                    const double are0 = ael(0, n, m0), aim0 = ael(1, n, m0), are1 = ael(0, n, m1), aim1 = ael(1, n, m1);
                    const double k11 = cfs[n], k12 = cfs[nf + n], k22 = cfs[2 * nf + n];
                    double a_[16], b_[16];                     // the tile's 2 x 16 values of this frequency, all requested before the first use
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) { a_[rr] = ArT[n * 16 + rr]; b_[rr] = AiT[n * 16 + rr]; }
                    const double p0 = z0 * (k11 * are0 + k12 * aim0), q0 = z0 * (k12 * are0 + k22 * aim0);
                    const double p1 = z1 * (k11 * are1 + k12 * aim1), q1 = z1 * (k12 * are1 + k22 * aim1);
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) __asm__ volatile("" : "+v"(a_[rr]), "+v"(b_[rr]));
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) {
                        acc[0][rr] = fma(a_[rr], p0, fma(b_[rr], q0, acc[0][rr]));
                        acc[1][rr] = fma(a_[rr], p1, fma(b_[rr], q1, acc[1][rr]));
                    }
                }
            };
            if (toepA) product([&](int part, int n, int m) -> double { return tgs[part * glen + n - m + K - 1]; });
            else product([&](int part, int n, int m) -> double { return Adp[((size_t)part * nf + n) * K + m]; });
        }
        // pass A: the groups' partial sums into the LDS tile, one group after the other (a fixed order of additions: the batched
        // iteration reproduces single fits bit for bit)
        for (int gq = 0; gq < NG; ++gq) {
            if (grp == gq && has_x) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (!(h == 0 ? jx0 : jx1)) continue;
#pragma unroll
                    for (int rr = 0; rr < 16; ++rr) tile[rr * Dp + jc[h]] += acc[h][rr];
                }
            }
            __syncthreads();
        }
    }
    __syncthreads();
    HF_STAMP(33);
    // pass B: everything but the dense product -- bands and border vectors -- added row by row, a wave per row (the row's class is
    // wave-uniform: no divergent maze of index classes), a lane per entry of the row's lists.  Only columns j < ncol exist in the tile.
    {
        const int lane = tid & 63, wave = tid >> 6;
        for (int rr = wave; rr < 16; rr += HP_NT / 64) {
            const int i = r0 + rr;
            if (i >= D) continue;
            double *trow = tile + rr * Dp;
            auto put = [&](int j, double v) { if (j < ncol) trow[j] += v; };
            if (i >= o_x && i < o_x + K) {                                   // x row m
                const int m = i - o_x;
                if (lane <= 4 * MAXBW) {                                     // x-x band of the prior
                    const int mp = m - 2 * MAXBW + lane;
                    if (mp >= 0 && mp < K) { const int dm = mp > m ? mp - m : m - mp, lo = mp > m ? m : mp; put(o_x + mp, W[L.pxx + dm * K + lo]); }
                } else if (lane < 4 * MAXBW + 1 + 2 * MAXBW + 1) {           // x-ups band: sum_q T_q[m - k + MAXBW] cu3[q][k]
                    const int t = lane - (4 * MAXBW + 1), k = m - MAXBW + t;
                    if (k >= 0 && k < K) {
                        const int d = m - k + MAXBW;
                        put(o_u + k, Ts[d] * W[L.cu3 + k] + Ts[13 + d] * W[L.cu3 + K + k] + Ts[26 + d] * W[L.cu3 + 2 * K + k]);
                    }
                } else if (lane >= 39 && lane < 48) {                        // borders: Rinf, induc, the error parameters, d
                    const int t = lane - 39;
                    if (t == 0) put(0, W[L.bR + m]);
                    else if (t == 1) put(1, W[L.bI + m]);
                    else if (t < 6) put(o_e + t - 2, W[L.bS + (t - 2) * K + m]);
                    else put(o_d + t - 6, W[L.hxd + (t - 6) * K + m]);
                }
            } else if (i >= o_u && i < o_u + K) {                            // ups row k
                const int k = i - o_u;
                if (lane <= 2 * MAXBW) {                                     // ups-x band
                    const int m = k - MAXBW + lane;
                    if (m >= 0 && m < K) {
                        const int d = m - k + MAXBW;
                        put(o_x + m, Ts[d] * W[L.cu3 + k] + Ts[13 + d] * W[L.cu3 + K + k] + Ts[26 + d] * W[L.cu3 + 2 * K + k]);
                    }
                } else if (lane < 2 * MAXBW + 1 + 5) {                       // ups-ups band
                    const int kp = k - 2 + lane - (2 * MAXBW + 1);
                    if (kp >= 0 && kp < K) { const int dm = kp > k ? kp - k : k - kp, lo = kp > k ? k : kp; put(o_u + kp, W[L.uu + dm * K + lo]); }
                } else if (lane < 2 * MAXBW + 1 + 8) {
                    const int q = lane - (2 * MAXBW + 1 + 5);
                    put(o_d + q, W[L.hud + q * K + k]);
                }
            } else {
                // Rinf, induc, an error parameter or a d row: vectors over x (and ups for d), then the few scalar entries
                const int cls = i < 2 ? i : (i < o_e + 4 ? 2 + (i - o_e) : 6 + (i - o_d));      // 0 R, 1 I, 2..5 s_a, 6..8 d_q
                const double *vx = cls == 0 ? W + L.bR : (cls == 1 ? W + L.bI : (cls < 6 ? W + L.bS + (cls - 2) * K : W + L.hxd + (cls - 6) * K));
                for (int m = lane; m < K; m += 64) put(o_x + m, vx[m]);
                if (cls >= 6) for (int k = lane; k < K; k += 64) put(o_u + k, W[L.hud + (cls - 6) * K + k]);
                if (lane < 6) {                                               // columns Rinf, induc, s_0..3
                    const int cj = lane;
                    double v = 0.0;
                    if (cls < 6) {
                        const int a_ = cls < cj ? cls : cj, b_ = cls < cj ? cj : cls;      // ordered pair of {R, I, s_0..3}
                        if (a_ == 0) v = b_ == 0 ? W[L.h0 + 0] : (b_ == 1 ? W[L.h0 + 1] : W[L.h0 + 3 + b_ - 2]);
                        else if (a_ == 1) v = b_ == 1 ? W[L.h0 + 2] : W[L.h0 + 7 + b_ - 2];
                        else v = W[L.hss + 4 * (a_ - 2) + (b_ - 2)];
                    }
                    put(cj < 2 ? cj : o_e + cj - 2, v);
                }
            }
        }
    }
    __syncthreads();
    // pass C: to the unconstrained scale, H and M = -H + lam I (identity on the padding); a thread per column, coalesced rows
    for (int j = tid; j < ncol; j += HP_NT) {
        const double scj = j < D ? W[L.sc + j] : 0.0;
        const bool actj = j < D && W[L.act + j] != 0.0;
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {
            const int i = r0 + rr;
            double m_ = i == j ? 1.0 : 0.0;
            if (i < D && j < D) {
                double hv = W[L.sc + i] * scj * tile[rr * Dp + j] + (i == j ? W[L.dex + i] : 0.0);
                if (actj || W[L.act + i] != 0.0) hv = i == j ? -1.0 : 0.0;       // a coefficient held at its floor leaves the system
                if (!isfinite(hv)) bad = true;
                m_ = -hv + (i == j ? lam : 0.0);
                H[(size_t)i * Dp + j] = hv;
            }
            M[(size_t)i * Dp + j] = m_;
        }
    }
    HF_STAMP(34);
#undef HF_STAMP
    return bad;
}

}  // namespace bdrt
