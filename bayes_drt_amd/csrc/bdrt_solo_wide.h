// bdrt_solo_wide.h -- log-posterior + gradient of ONE chain by the whole workgroup, general block model.
//
// The reference's own call shape is 2-4 chains (inversion.py:1079).  For the headline family that is the kernel of
// bdrt_solo.h; models with several distributions, parallel (admittance) blocks or the outlier error model (BASELINE
// config 5, D = 818) used to run one chain per workgroup on the 16-chain kernel, whose evaluator computes a 16-column
// MFMA tile for one live column: ~60 us per leapfrog.  Here the evaluation is spread over the 512 threads the way
// bdrt_solo.h does it -- A x and A^T g as Toeplitz products from the blocks' generators (every block of a log-uniform
// grid is Toeplitz: the integrand depends on ln(omega_n tau_m) only), banded L as 13-tap convolutions -- block after block,
// with the formulas of bdrt_tile_hw.h (Series / Parallel / Series-Parallel / Series-2Parallel model code, +- pos,
// +- outliers).  The chain's vectors stay in HBM (they do not fit in LDS at D = 818); everything after the evaluation is
// the cooperative stage of bdrt_nuts_wide.h.
//
// (Included by bdrt_nuts.hip inside namespace bdrt, after bdrt_solo.h and bdrt_nuts_wide.h.)
#pragma once

constexpr int W1_MAXB = 3;                     // blocks (distributions) per model

struct Wide1Geom {
    SoloGeom g;                                // product partition of one block (all blocks share nf and K)
    int nb;
    int o_xr, o_xs, o_us, o_w, o_gen, o_zp, o_y, o_gz, o_rop, o_gl, o_red, o_scv, total;   // LDS offsets (doubles)
};

__host__ __device__ inline Wide1Geom wide1_geometry(int nf, int K, int D, int nb)
{
    Wide1Geom G;
    G.g = solo_geometry(nf, K, D);             // (forward products on the waves behind the K-threads, beside the prior chain: bdrt_solo.h)
    G.nb = nb;
    const SoloGeom &g = G.g;
    int o = 0;
    G.o_xr = o; o += g.XL;                     // x of the current block, raw (prior chain)
    G.o_xs = o; o += g.XL;                     // x times x_scale (operand of A)
    G.o_us = o; o += g.XL;                     // ups with its halo of two
    G.o_w = o; o += 3 * g.XL;                  // w_i = -d_i v_i / ups^2
    G.o_gen = o; o += nb * 2 * 4 * g.GQ;       // generators of every block, swizzled
    const int zp = g.NP * 4 * g.RG, gp = g.NPB * 4 * g.MG;
    G.o_zp = o; o += zp > gp ? zp : gp;        // partial sums of the current product
    G.o_y = o; o += nb * 2 * g.NFP;            // Y_b = A_b x_b of every block
    G.o_gz = o; o += 2 * g.NFP;                // d lp / d Z_hat
    G.o_rop = o; o += 2 * g.NFP;               // operand of A_b^T
    G.o_gl = o; o += nb * g.XL;                // sum_i L_i^T w_i of every block
    G.o_red = o; o += SOLO_NW * 32;            // wave partials: [wave][32]
    G.o_scv = o; o += 32;                      // constrained scalars, x_sum term, lp
    G.total = o;
    return G;
}

// can this problem take the one-chain evaluator?  (host)
inline bool wide1_capable(const DevProblem &P)
{
    if (P.nblocks < 1 || P.nblocks > W1_MAXB || P.nf > 128 || P.D > 2 * WIDE_NT) return false;
    const int K = P.blk[0].K;
    if (K < 2 * MAXBW + 3 || K > 192) return false;
    for (int b = 0; b < P.nblocks; ++b)
        if (P.blk[b].K != K || !P.blk[b].toep || (P.blk[b].tg == nullptr && (P.blk[b].Ad == nullptr || P.blk[b].At == nullptr))) return false;
    // + the cooperative stage's scratch, lp and chain state of nuts_wide1_kernel (wide1_lds_bytes with no resident rows)
    return (size_t)wide1_geometry(P.nf, K, P.D, P.nblocks).total * sizeof(double) + 16384 <= 160 * 1024;
}

struct Wide1Regs {
    double zre, zim, wn;      // measured spectrum of row n = tid (tid < nf)
    int fpart, frg;           // forward product: m-part and 4-row group of this thread
    int bpart, bmg;           // backward product: n-part and 4-column group of this thread
};

__device__ __forceinline__ Wide1Regs wide1_setup(const DevProblem &P, const Wide1Geom &G, int spec, int tid)
{
    const SoloGeom &g = G.g;
    Wide1Regs er;
    const int n = tid < g.nf ? tid : 0;
    const double *Zm = P.Z + (size_t)spec * 2 * g.nf;
    er.zre = Zm[n]; er.zim = Zm[g.nf + n]; er.wn = P.w[n];
    const int pt = tid - g.FP0;
    er.fpart = pt >= 0 ? pt / g.RG : g.NP; er.frg = pt - er.fpart * g.RG;
    er.bpart = tid / g.MG; er.bmg = tid - er.bpart * g.MG;
    return er;
}

// one-time LDS set-up: zero halos / pads, the generators of every block in swizzled order
__device__ __forceinline__ void wide1_init(const DevProblem &P, const Wide1Geom &G, double *lds, int tid)
{
    const SoloGeom &g = G.g;
    for (int i = tid; i < G.o_gen; i += SOLO_NT)                                     // x rows, w rows: zero halos; ups pads: finite, never used
        lds[i] = (i >= G.o_us && i < G.o_us + g.XL) ? 1.0 : 0.0;
    for (int i = tid; i < G.total - G.o_zp; i += SOLO_NT) lds[G.o_zp + i] = 0.0;
    const int glen = g.nf + g.K - 1;
    for (int i = tid; i < G.nb * 2 * 4 * g.GQ; i += SOLO_NT) {
        const int blk = i / (2 * 4 * g.GQ), i2 = i - blk * 2 * 4 * g.GQ;
        const int h = i2 / (4 * g.GQ), r = i2 - h * 4 * g.GQ, rho = r / g.GQ, q = r - rho * g.GQ;
        const int e = 4 * q + rho;                                                   // logical index: n - m + S
        const int src = e - g.S + g.K - 1;
        const double *tg = P.blk[blk].tg;                                            // [2][nf + K - 1]: c_h[n - m + K - 1]
        lds[G.o_gen + i] = (tg && src >= 0 && src < glen) ? tg[(size_t)h * glen + src] : 0.0;
    }
}

// dense counterparts of solo_toeplitz4 for a block whose A is not Toeplitz (read from L2):
//   forward  acc[i] += sum_t At[(m0 + t) * N2 + r0 + i] * v[t]   (rows r0 .. r0+3 of one half, `rows_left` of them exist; columns < K)
//   backward acc[i] += sum_t Ad[(r0 + t) * K + m0 + i] * v[t]    (columns m0 .. m0+3 < K; `rows_left` rows of the half exist)
__device__ __forceinline__ void wide1_dense_fwd(const double *__restrict__ At, int N2, int r0, int rows_left, int m0, int K,
                                                const double *v, int len, double (&acc)[4])
{
    const bool e0 = rows_left > 0, e1 = rows_left > 1, e2 = rows_left > 2, e3 = rows_left > 3;
#pragma unroll 4
    for (int t = 0; t < len; ++t) {
        const int m = m0 + t;
        if (m >= K) break;
        const double *a = At + (size_t)m * N2 + r0;
        const double x = v[t];
        acc[0] = fma(e0 ? a[0] : 0.0, x, acc[0]); acc[1] = fma(e1 ? a[1] : 0.0, x, acc[1]);
        acc[2] = fma(e2 ? a[2] : 0.0, x, acc[2]); acc[3] = fma(e3 ? a[3] : 0.0, x, acc[3]);
    }
}
__device__ __forceinline__ void wide1_dense_bwd(const double *__restrict__ Ad, int K, int r0, int rows_left, int m0,
                                                const double *v, int len, double (&acc)[4])
{
    const bool e0 = m0 < K, e1 = m0 + 1 < K, e2 = m0 + 2 < K, e3 = m0 + 3 < K;
#pragma unroll 4
    for (int t = 0; t < len; ++t) {
        if (t >= rows_left) break;
        const double *a = Ad + (size_t)(r0 + t) * K + m0;
        const double x = v[t];
        acc[0] = fma(e0 ? a[0] : 0.0, x, acc[0]); acc[1] = fma(e1 ? a[1] : 0.0, x, acc[1]);
        acc[2] = fma(e2 ? a[2] : 0.0, x, acc[2]); acc[3] = fma(e3 ? a[3] : 0.0, x, acc[3]);
    }
}

// log-posterior + gradient at theta (global row TH) -> gradient to the global row GR, lp to *lp_out (LDS or global).
// All 512 threads call; ends with a __syncthreads().
__device__ inline void wide1_eval(const DevProblem &P, const Wide1Geom &G, double *lds, const double *__restrict__ TH,
                                  double *__restrict__ GR, double *lp_out, const Wide1Regs &er, int jacobian, int tid,
                                  long long *prof = nullptr)
{
    long long tprev = (prof && tid == 0) ? clock64() : 0;
#define BDRT_W1_PROF(slot) do { if (prof && tid == 0) { const long long t_ = clock64(); prof[slot] += t_ - tprev; tprev = t_; } } while (0)
    const SoloGeom &g = G.g;
    const int lane = tid & 63, wave = tid >> 6;
    const int nf = g.nf, K = g.K, nb = G.nb;
    const double jac = jacobian ? 1.0 : 0.0;
    double *xr = lds + G.o_xr, *xs = lds + G.o_xs, *us = lds + G.o_us, *wr = lds + G.o_w;
    double *zp = lds + G.o_zp, *yb = lds + G.o_y, *gz = lds + G.o_gz, *rop = lds + G.o_rop, *gls = lds + G.o_gl;
    double *ered = lds + G.o_red, *scv = lds + G.o_scv;
    double lp = 0.0;                                       // this thread's share of lp

    // ---- scalars: thread SOLO_SCAL0 + i owns scalar i: the six global ones, then d_0..d_2 of every block ------------------------
    const int sidx = tid - SOLO_SCAL0, nsc = 6 + 3 * nb;
    int sj = -1;
    double sraw = 0.0;
    if (sidx >= 0 && sidx < nsc) {
        sj = sidx < 2 ? sidx : (sidx < 6 ? P.o_err + (sidx - 2) : P.blk[(sidx - 6) / 3].o_d + (sidx - 6) % 3);
        const double st = TH[sj];
        sraw = lean_exp(st);
        scv[sidx] = sraw;
        lp += (sidx < 6 ? -0.5 * sraw * sraw : -6.0 * st - 5.0 * lean_rcp(sraw)) + jac * st;
    }
    double xv[W1_MAXB], gupv[W1_MAXB];                     // x_k and d lp / d theta_ups_k of this thread's k, per block
    double xsum_p = 0.0;
#pragma unroll
    for (int b = 0; b < W1_MAXB; ++b) { xv[b] = 0.0; gupv[b] = 0.0; }

    // ================================================= forward, block by block ===================================================
    // Phases (one barrier each): [constrain block 0] -> for every block: [products + prior chain] -> [Y_b sums + L^T w of the
    // block + constrain of the NEXT block]; the last block's sums and L^T w ride on the likelihood phase.
    double tu = 0.0, uu = 0.0;                             // theta_ups / ups of this thread's k in the block being constrained
    auto constrain = [&](int b) {
        const DevBlock &B = P.blk[b];
        if (tid < K) {
            const double tx = TH[B.o_x + tid];
            tu = TH[B.o_ups + tid];
            const double x = B.is_pos ? lean_exp(tx) : tx;
            if (B.is_pos) lp += jac * tx;
            uu = 0.15 * lean_exp(tu);
            xv[b] = x;
            xsum_p += x;
            xr[MAXBW + tid] = x;
            xs[MAXBW + tid] = x * B.x_scale;               // xp = xp_raw * xp_scale (1 for series blocks)
            us[2 + tid] = uu;
        }
    };
    auto conv_lt = [&](int b) {                            // sum_i L_i^T w_i of block b (threads CONV0 + k)
        const DevBlock &B = P.blk[b];
        const int kc = tid - SOLO_CONV0;
        if (kc >= 0 && kc < K) {
            double gl = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double *wi = wr + i * g.XL;
#pragma unroll
                for (int d = 0; d < 2 * MAXBW + 1; ++d) gl = fma(B.T[i][d], wi[kc + 2 * MAXBW - d], gl);   // w_i[kc + MAXBW - d]
            }
            gls[b * g.XL + kc] = gl;
        }
    };
    auto sum_partials = [&](int n, double &yr, double &yi) {       // Y_b[n] from the partial sums of the current product
        const int st = 4 * g.RG;
        double r_[16], i_[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {                     // all loads first: one LDS round trip, not one per part
            const int pp = p < g.NP ? p : 0;
            r_[p] = zp[pp * st + n]; i_[p] = zp[pp * st + 4 * g.RGb + n];
        }
        yr = 0.0; yi = 0.0;
#pragma unroll
        for (int p = 0; p < 16; ++p) if (p < g.NP) { yr += r_[p]; yi += i_[p]; }
        for (int p = 16; p < g.NP; ++p) { yr += zp[p * st + n]; yi += zp[p * st + 4 * g.RGb + n]; }
    };
    constrain(0);
    __syncthreads();
    BDRT_W1_PROF(2);
#pragma unroll
    for (int b = 0; b < W1_MAXB; ++b) {
        if (b >= nb) break;
        const DevBlock &B = P.blk[b];
        const double *gen = lds + G.o_gen + b * 2 * 4 * g.GQ;
        // ---- forward product partials; prior chain x -> L x -> w
        {
            const int part = er.fpart, rg = er.frg;
            if (part < g.NP) {
                const int h = rg >= g.RGb, n0 = 4 * (rg - h * g.RGb), m0 = part * g.ML;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                if (B.tg) solo_toeplitz4<1>(gen + h * 4 * g.GQ, g.GQ, n0 - m0 + g.S, xs + MAXBW + m0, g.ML, acc);
                else wide1_dense_fwd(B.At, 2 * nf, h * nf + n0, nf - n0, m0, K, xs + MAXBW + m0, g.ML, acc);
                double *o = zp + part * (4 * g.RG) + 4 * rg;
                o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
            }
        }
        double sv0 = 0.0, sv1 = 0.0, sv2 = 0.0;
        if (tid < K) {
            const int k = tid;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
#pragma unroll
            for (int d = 0; d < 2 * MAXBW + 1; ++d) {
                const double xw = xr[k + d];               // x[k - MAXBW + d]
                v0 = fma(B.T[0][d], xw, v0); v1 = fma(B.T[1][d], xw, v1); v2 = fma(B.T[2][d], xw, v2);
            }
            const double d0 = scv[6 + 3 * b], d1 = scv[7 + 3 * b], d2 = scv[8 + 3 * b];
            const double um2 = us[k], um1 = us[k + 1], up1 = us[k + 3], up2 = us[k + 4];
            const double iu = lean_rcp(uu), iu2 = iu * iu;
            const double q2 = d0 * v0 * v0 + d1 * v1 * v1 + d2 * v2 * v2;
            const double ir = 0.15 * iu;                   // 1 / ups_raw
            lp += -(tu + LOG_015) - 0.5 * q2 * iu2 - (P.ups_alpha + 1.0) * tu - P.ups_beta * ir + jac * tu;
            sv0 = v0 * v0 * iu2; sv1 = v1 * v1 * iu2; sv2 = v2 * v2 * iu2;
            double gu = -iu + q2 * iu2 * iu;
            if (k >= 1 && k + 1 < K) {
                const double du = 0.5 * (uu - 0.5 * (um1 + up1)) * iu;
                lp += -0.5 * du * du;
                gu += -du * 0.25 * (um1 + up1) * iu2;
            }
            if (k >= 2) {
                const double i0 = lean_rcp(um1);
                const double du = 0.5 * (um1 - 0.5 * (um2 + uu)) * i0;
                gu += du * 0.25 * i0;
            }
            if (k + 2 < K) {
                const double i0 = lean_rcp(up1);
                const double du = 0.5 * (up1 - 0.5 * (uu + up2)) * i0;
                gu += du * 0.25 * i0;
            }
            gupv[b] = uu * gu - (P.ups_alpha + 1.0) + P.ups_beta * ir + jac;
            wr[MAXBW + k] = -d0 * v0 * iu2;
            wr[g.XL + MAXBW + k] = -d1 * v1 * iu2;
            wr[2 * g.XL + MAXBW + k] = -d2 * v2 * iu2;
        }
        if (wave < 3) {                                    // the K-threads live in waves 0..2 (K <= 192)
            // one butterfly for the three sums (sum32_by_lane: lane j of each half-wave ends with sum j), the halves meet in lanes 0..2
            const double q3[4] = {sv0, sv1, sv2, 0.0};
            double t3 = sum32_by_lane<4>(q3, lane);
            t3 += __shfl_xor(t3, 32);
            if (lane < 3) ered[wave * 32 + 8 + 3 * b + lane] = t3;
        }
        __syncthreads();
        BDRT_W1_PROF(3);
        if (b + 1 < nb) {
            // ---- Y_b (threads n < nf: both halves), L^T w of this block, and the next block's constrained parameters
            if (tid < nf) {
                double yr, yi;
                sum_partials(tid, yr, yi);
                yb[b * 2 * g.NFP + tid] = yr; yb[b * 2 * g.NFP + g.NFP + tid] = yi;
            }
            conv_lt(b);
            constrain(b + 1);
            __syncthreads();
            BDRT_W1_PROF(4);
        }
    }

    // ================================================= x_sum prior, likelihood ===================================================
    {
        const double wx = solo_wave_sum(xsum_p);
        if (lane == 0) ered[wave * 32 + 20] = wx;
    }
    conv_lt(nb - 1);                                       // (the last block's L^T w: threads CONV0 + k, beside the likelihood threads)
    {
        double sR = 0, sL = 0, sH = 0, sHz2 = 0, sHzr2 = 0, sHzi2 = 0;
        if (tid < nf) {
            const int n = tid;
            {   // Y of the last block: this thread sums its partials itself
                double yr, yi;
                sum_partials(n, yr, yi);
                yb[(nb - 1) * 2 * g.NFP + n] = yr; yb[(nb - 1) * 2 * g.NFP + g.NFP + n] = yi;
            }
            double zr = 0.0, zi = 0.0;
#pragma unroll
            for (int b = 0; b < W1_MAXB; ++b) {
                if (b >= nb) break;
                const double yr = yb[b * 2 * g.NFP + n], yi = yb[b * 2 * g.NFP + g.NFP + n];
                if (!P.blk[b].is_parallel) { zr += yr; zi += yi; }
                else {
                    const double idn = lean_rcp(yr * yr + yi * yi);
                    zr += yr * idn;                        // Z_hat_p = conj(Y) / |Y|^2 (Parallel_modelcode.txt:47)
                    zi += -yi * idn;
                }
            }
            const double Rinf = 100.0 * scv[0], induc = scv[1] * P.induc_scale;
            const double s_res = 0.05 * scv[2], a_p = 0.05 * scv[3], a_r = 0.05 * scv[4], a_i = 0.05 * scv[5];
            zr += Rinf; zi += induc * er.wn;
            double so_re = 0.0, so_im = 0.0, r0 = 0.0, r1 = 0.0, t0 = 0.0, t1 = 0.0;
            if (P.outlier_mode) {
                t0 = TH[P.o_so + n]; t1 = TH[P.o_so + nf + n];
                r0 = lean_exp(t0); r1 = lean_exp(t1);
                if (P.outlier_mode == 1) so_re = so_im = 0.05 * r0 * r1;
                else { so_re = 0.05 * r0; so_im = 0.05 * r1; }
            }
            const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
            const double ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
            const double common = ar2 * zr * zr + ai2 * zi * zi;
            const double s2_re = c0 + ap2 * zr * zr + common + so_re * so_re;
            const double s2_im = c0 + ap2 * zi * zi + common + so_im * so_im;
            const double e_re = er.zre - zr, e_im = er.zim - zi;
            const double prod = s2_re * s2_im, ip = lean_rcp(prod);
            const double w_re = s2_im * ip, w_im = s2_re * ip;
            lp += -0.5 * lean_log(prod) - 0.5 * e_re * e_re * w_re - 0.5 * e_im * e_im * w_im;
            const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;
            const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
            const double gzr = e_re * w_re + 2.0 * zr * (h_re * (ap2 + ar2) + h_im * ar2);
            const double gzi = e_im * w_im + 2.0 * zi * (h_im * (ap2 + ai2) + h_re * ai2);
            gz[n] = gzr; gz[g.NFP + n] = gzi;
            sR = gzr; sL = gzi * er.wn; sH = h_re + h_im; sHz2 = h_re * zr * zr + h_im * zi * zi;
            sHzr2 = (h_re + h_im) * zr * zr; sHzi2 = (h_re + h_im) * zi * zi;
            if (P.outlier_mode == 1) {
                const double dso = 2.0 * so_re * (h_re + h_im);
                GR[P.o_so + n] = r0 * (0.05 * r1 * dso - P.so_lambda) + jac;
                GR[P.o_so + nf + n] = 0.05 * r0 * r1 * dso - (P.so_alpha + 1.0) + P.so_beta * lean_rcp(r1) + jac;
                lp += -P.so_lambda * r0 - (P.so_alpha + 1.0) * t1 - P.so_beta * lean_rcp(r1) + jac * (t0 + t1);
            } else if (P.outlier_mode == 2) {
                GR[P.o_so + n] = r0 * (0.05 * 2.0 * so_re * h_re - P.so_lambda) + jac;
                GR[P.o_so + nf + n] = r1 * (0.05 * 2.0 * so_im * h_im - P.so_lambda) + jac;
                lp += -P.so_lambda * (r0 + r1) + jac * (t0 + t1);
            }
        }
        if (wave < 2) {                                    // nf <= 128: the likelihood threads are waves 0 and 1
            // the six sums in one butterfly: lanes 0..5 -> slots 0..5
            const double q6[8] = {sR, sL, sH, sHz2, sHzr2, sHzi2, 0.0, 0.0};
            double t6 = sum32_by_lane<8>(q6, lane);
            t6 += __shfl_xor(t6, 32);
            if (lane < 6) ered[wave * 32 + lane] = t6;
        }
    }
    __syncthreads();
    BDRT_W1_PROF(5);
    // x_sum ~ std_normal(), real<lower=0> x_sum_raw (Series-Parallel model code): the term enters every d lp / d x, so every
    // thread adds up the eight wave partials itself
    double xs_term = 0.0, lpx = 0.0;
    bool rej = false;
    if (P.use_x_sum) {
        double xs_raw = 0.0;
#pragma unroll
        for (int w = 0; w < SOLO_NW; ++w) xs_raw += ered[w * 32 + 20];
        const double xsn = xs_raw * P.x_sum_invscale;
        lpx = -0.5 * xsn * xsn;
        rej = xs_raw < 0.0;
        xs_term = -xs_raw * P.x_sum_invscale * P.x_sum_invscale;
    }

    // ---- scalar gradients, the waves' shares of lp: everything they need is complete here; in front of the backward pass they run beside
    //      its first operand phase instead of behind the last chain rule with two barriers of their own --------------------------------
    if (sj >= 0) {
        double gsc;
        if (sidx < 6) {
            const double t = ered[sidx] + ered[32 + sidx];
            double dl;
            if (sidx == 0) dl = 100.0 * t;
            else if (sidx == 1) dl = P.induc_scale * t;
            else dl = 0.05 * 2.0 * (0.05 * sraw) * t;
            gsc = sraw * (dl - sraw) + jac;
        } else {
            const int q = 8 + (sidx - 6);                  // slot 8 + 3 b + i
            const double sv = ered[q] + ered[32 + q] + ered[64 + q];
            gsc = -0.5 * sraw * sv - 6.0 + 5.0 * lean_rcp(sraw) + jac;
        }
        GR[sj] = gsc;
    }
    {
        const double wl = solo_wave_sum(lp);
        if (lane == 0) ered[wave * 32 + 21] = wl;
    }

    // ================================================= backward, block by block ==================================================
#pragma unroll
    for (int b = 0; b < W1_MAXB; ++b) {
        if (b >= nb) break;
        const DevBlock &B = P.blk[b];
        const double *gen = lds + G.o_gen + b * 2 * 4 * g.GQ;
        // operand of A_b^T: g_Zhat, or J^T g_Zhat through Z_hat_p = conj(Y)/|Y|^2 (times xp_scale) for a parallel block
        if (tid < nf) {
            const int n = tid;
            double rr = gz[n], ri = gz[g.NFP + n];
            if (B.is_parallel) {
                const double yr = yb[b * 2 * g.NFP + n], yi = yb[b * 2 * g.NFP + g.NFP + n];
                const double dn = yr * yr + yi * yi, id2 = lean_rcp(dn * dn);
                const double dd = (yi * yi - yr * yr) * id2, doff = 2.0 * yr * yi * id2;
                const double gr_ = rr, gi_ = ri;
                rr = (gr_ * dd + gi_ * doff) * B.x_scale;
                ri = (-gr_ * doff + gi_ * dd) * B.x_scale;
            }
            rop[n] = rr; rop[g.NFP + n] = ri;
        }
        __syncthreads();
        BDRT_W1_PROF(6);
        {
            const int part = er.bpart, mg = er.bmg;
            if (part < g.NPB) {
                const int half = g.NPB / 2, h = part >= half, pp = part - h * half, n0 = pp * g.NL, m0 = 4 * mg;
                double acc[4] = {0.0, 0.0, 0.0, 0.0};
                if (B.tg) solo_toeplitz4<-1>(gen + h * 4 * g.GQ, g.GQ, n0 - m0 + g.S, rop + h * g.NFP + n0, g.NL, acc);
                else wide1_dense_bwd(B.Ad, K, h * nf + n0, nf - n0, m0, rop + h * g.NFP + n0, g.NL, acc);
                double *o = zp + part * (4 * g.MG) + 4 * mg;
                o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
            }
        }
        __syncthreads();
        BDRT_W1_PROF(7);
        if (tid < K) {
            double graw = gls[b * g.XL + tid] + xs_term;
            const int st = 4 * g.MG;
            for (int p = 0; p < g.NPB; ++p) graw += zp[p * st + tid];
            GR[B.o_x + tid] = B.is_pos ? xv[b] * graw + jac : graw;
            GR[B.o_ups + tid] = gupv[b];
        }
        __syncthreads();                                   // (the next block's partials reuse zp)
        BDRT_W1_PROF(8);
    }

    // ---- lp (the scalar gradients and the waves' shares of lp were formed in front of the backward pass) ---------------------------
    if (tid == 0) {
        double s = lpx;
#pragma unroll
        for (int w = 0; w < SOLO_NW; ++w) s += ered[w * 32 + 21];
        *lp_out = rej ? -INFINITY : s;
    }
    __syncthreads();
    BDRT_W1_PROF(9);
#undef BDRT_W1_PROF
}
