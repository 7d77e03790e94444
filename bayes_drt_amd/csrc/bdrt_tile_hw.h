// bdrt_tile_hw.h -- the half-wave-per-chain evaluator (bdrt_tile_s1.h) for EVERY model family on log-uniform grids:
// several distributions (Series-Parallel, Series-2Parallel: *_modelcode.txt with xs / xp*), parallel (admittance) blocks,
// the x_sum prior of the mixed models, the outlier error models.  Same formulas as the generic block evaluator
// (bdrt_device.h), same mapping and building blocks as the S1 evaluator:
//   * chain c of the workgroup <-> half-wave; reductions are DPP butterflies, per-chain scalars live in the lanes that own
//     them (lane j: scalar j; lanes 6 + 3 b + i: d_i of block b);
//   * per block b: x_b -> MFMA operand tile and private row; A_b x_b by MFMA; the prior chain x_b -> L x_b -> w -> L^T w on
//     the private row between the GEMM and the barrier that publishes A_b x_b; Z_hat accumulates in registers (a parallel
//     block contributes conj(Y)/|Y|^2 and parks Y in its own LDS tile for the Jacobian of the backward pass);
//   * backward per block: g_Zhat (through the Jacobian for a parallel block) -> operand tile, A_b^T by MFMA, chain rule.
//     L^T w of a block waits in the gradient row itself (stored in the prior phase, read back by the same lane).
// Four workgroup barriers per block (generic evaluator: ~24 per block).  LDS: X/G tile + A x tile + one Y tile per parallel
// block + the private rows = 98.5 KiB + 22.5 KiB per parallel block.
#pragma once
#include "bdrt_tile_s1.h"

namespace bdrt {

__host__ __device__ inline size_t hw_lds_doubles(const DevProblem &P)
{
    return (size_t)NC * (P.XR + 16 * P.blk[0].tilesA * (1 + P.npar)) + (size_t)2 * NC * RW;
}

// the longest basis of the problem's blocks (what picks the instantiation: bdrt_tile_s1.h::s1_ku)
__host__ __device__ inline int hw_kmax(const DevProblem &P)
{
    int k = 0;
    for (int b = 0; b < P.nblocks; ++b) k = P.blk[b].K > k ? P.blk[b].K : k;
    return k;
}

// All threads of the workgroup must call.  Ends with a __syncthreads().  KU: basis functions per lane (every block's K <= 32 KU),
// as in logp_grad_tile_s1.
// PROFT: the profiling instantiation (slots 4..8, 25..31 of the sampler's phase profile: wave-summed cycles of the sub-phases).
template <int KU = 6, bool PROFT = false>
__device__ inline void logp_grad_tile_hw(const DevProblem &P, const TileIO &io, double *smem)
{
    int tid = threadIdx.x;
    __asm__ volatile("" : "+v"(tid));                      // see bdrt_tile_s1.h: keeps index arithmetic out of the caller's loop
    constexpr int LPC = 32, UKV = KU, UNV = 4, WINV = UKV + NTAP - 1, NWV = 8, GPFV = 7;      // K <= 32 KU, Nf <= 128
    const int lane = tid & 63, wave = tid >> 6;
    const int c = tid / LPC;
    const int l32 = tid % LPC, hb = lane & 32;
    const int nf = P.nf, N2 = 2 * nf, nblocks = P.nblocks;
    const int TA = 16 * P.blk[0].tilesA;                   // rows of an A x tile
    const bool valid = c < io.nvalid;
    const int cc = valid ? c : 0;
    const double jac = io.jacobian ? 1.0 : 0.0;
    long long thw = (PROFT && io.prof) ? clock64() : 0;
#define BDRT_HW_PROF(slot) do { if (PROFT && io.prof) { const long long t_ = clock64(); \
        if (lane == 0) atomicAdd((unsigned long long *)&io.prof[slot], (unsigned long long)(t_ - thw)); thw = t_; } } while (0)

    double *Xs = smem;
    double *Zh = Xs + (size_t)P.XR * NC;
    double *Yp = Zh + (size_t)TA * NC;                                        // [npar][TA][16]
    double *xrow = Yp + (size_t)P.npar * TA * NC + (size_t)c * (2 * RW);      // the chain's private row
    double *wrow = xrow + RW;

    const double *th = io.theta + (long)cc * io.t_sc + (io.t_off ? io.t_off[cc] : 0);
    auto TH = [&](int j) -> double { return th[(long)j * io.t_sj]; };
    double *gr = (io.grad && valid) ? io.grad + (long)cc * io.g_sc : nullptr;
    auto GW = [&](int j, double v) { if (gr) gr[(long)j * io.g_sj] = v; };
    auto GR = [&](int j) -> double { return gr ? gr[(long)j * io.g_sj] : 0.0; };
    double *pr = (io.params && valid) ? io.params + (size_t)cc * P.D : nullptr;
    auto PW = [&](int j, double v) { if (pr) pr[j] = v; };

    // ---- scalars: lanes 0..5 the six global ones, lanes 6 + 3 b + i the penalty strengths d_i of block b ---------------------
    double sraw = 0.0, st = 0.0;
    {
        int j = -1;
        if (l32 < 2) j = l32;
        else if (l32 < 6) j = P.o_err + (l32 - 2);
        else if (l32 < 6 + 3 * nblocks) j = P.blk[(l32 - 6) / 3].o_d + (l32 - 6) % 3;
        if (j >= 0) { st = TH(j); sraw = lean_exp(st); PW(j, sraw); }
    }
    double lp = 0.0;
    if (l32 < 6) lp += -0.5 * sraw * sraw + jac * st;                      // std_normal on the raws, log transform
    else if (l32 < 6 + 3 * nblocks) lp += -6.0 * st - 5.0 * lean_rcp(sraw) + jac * st;   // d ~ inv_gamma(5, 5)
    double gsc = 0.0;

    double zre_a[UNV], zim_a[UNV];                                         // Z_hat without the offsets, n = l32 + 32 v
#pragma unroll
    for (int v = 0; v < UNV; ++v) { zre_a[v] = 0.0; zim_a[v] = 0.0; }
    double xsum_p = 0.0;                                                   // this lane's share of sum_b sum_k x_raw

    // ================================================= forward, block by block ============================================
#pragma unroll 1
    for (int b = 0; b < nblocks; ++b) {
        const DevBlock &B = P.blk[b];
        const int K = B.K, KP = 8 * B.kpairs;
        // ---- P1 (M1): x_b into the operand tile (times xp_scale for a parallel block) and the private row (raw) ---------------
        {
            double tx_[UKV], tu_[UKV], ex_[UKV];
#pragma unroll
            for (int u = 0; u < UKV; ++u) {                                // (loads from a clamped index, THEN the selects: a load
                const int k = l32 + LPC * u, kk = k < K ? k : 0;           //  under `k < K` is a branch around a memory round trip of
                tx_[u] = TH(B.o_x + kk);                                   //  its own -- six in a row per block, bdrt_tile_s1.h)
                tu_[u] = TH(B.o_ups + kk);
            }
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const bool in = l32 + LPC * u < K;
                tx_[u] = in ? tx_[u] : 0.0;
                tu_[u] = in ? tu_[u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < UKV; ++u) ex_[u] = lean_exp(tx_[u]);
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const int k = l32 + LPC * u;
                double xr = 0.0;
                if (k < K) {
                    xr = B.is_pos ? ex_[u] : tx_[u];
                    if (B.is_pos) lp += jac * tx_[u];
                    xsum_p += xr;
                    PW(B.o_x + k, xr);
                }
                if (k < KP) Xs[swz(k, c)] = xr * B.x_scale;                // xp = xp_raw * xp_scale (1 for series blocks)
                xrow[MAXBW + k] = xr;
                wrow[MAXBW + k] = tu_[u];
            }
            if (l32 < MAXBW) {
                xrow[l32] = 0.0; xrow[MAXBW + LPC * UKV + l32] = 0.0;
                wrow[l32] = 0.0; wrow[MAXBW + LPC * UKV + l32] = 0.0;
            }
        }
        const double d0 = __shfl(sraw, hb | (6 + 3 * b)), d1 = __shfl(sraw, hb | (7 + 3 * b)), d2 = __shfl(sraw, hb | (8 + 3 * b));
        BDRT_HW_PROF(4);
        __syncthreads();                                                   // B1: X_b of all 16 chains in the operand tile
        BDRT_HW_PROF(5);
        gemm_sw<NWV, GPFV>(B.Af, B.tilesA, B.kpairs, Xs, Zh, wave, lane); // Zh = A_b x_b (pad rows: exact zeros)
        BDRT_HW_PROF(6);

        // ---- P2 (M2): v_i = L_i x, q / ups / dups priors, w_i, sum_i L_i^T w_i on the private row ------------------------------
        {
            const int kb = UKV * l32;
            double xw[WINV], tuc[UKV];
#pragma unroll
            for (int j = 0; j < WINV; ++j) xw[j] = xrow[kb + j];
#pragma unroll
            for (int u = 0; u < UKV; ++u) tuc[u] = wrow[MAXBW + kb + u];
            wave_sync();
            double v0_[UKV], v1_[UKV], v2_[UKV];
#pragma unroll
            for (int u = 0; u < UKV; ++u) { v0_[u] = 0.0; v1_[u] = 0.0; v2_[u] = 0.0; }
            double tn0 = B.T[0][0], tn1 = B.T[1][0], tn2 = B.T[2][0];
#pragma unroll
            for (int d = 0; d < NTAP; ++d) {
                const double t0 = tn0, t1 = tn1, t2 = tn2;
                if (d + 1 < NTAP) { tn0 = B.T[0][d + 1]; tn1 = B.T[1][d + 1]; tn2 = B.T[2][d + 1]; }
                __asm__ volatile("" ::: "memory");
#pragma unroll
                for (int u = 0; u < UKV; ++u) { v0_[u] = fma(t0, xw[u + d], v0_[u]); v1_[u] = fma(t1, xw[u + d], v1_[u]); v2_[u] = fma(t2, xw[u + d], v2_[u]); }
            }
            double ue[UKV + 4], ie[UKV + 2];
#pragma unroll
            for (int u = 0; u < UKV; ++u) { ue[u + 2] = 0.15 * lean_exp(tuc[u]); ie[u + 1] = lean_rcp(ue[u + 2]); }
            {
                const int lo = hb | ((l32 + LPC - 1) & (LPC - 1)), hi = hb | ((l32 + 1) & (LPC - 1));
                ue[0] = __shfl(ue[UKV], lo); ue[1] = __shfl(ue[UKV + 1], lo);
                ue[UKV + 2] = __shfl(ue[2], hi); ue[UKV + 3] = __shfl(ue[3], hi);
                ie[0] = __shfl(ie[UKV], lo); ie[UKV + 1] = __shfl(ie[1], hi);
            }
            double sv0 = 0, sv1 = 0, sv2 = 0;
            double w0_[UKV], w1_[UKV], w2_[UKV], gup[UKV];
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const int k = kb + u;
                w0_[u] = 0.0; w1_[u] = 0.0; w2_[u] = 0.0; gup[u] = 0.0;
                if (k < K) {
                    const double um2 = ue[u], um1 = ue[u + 1], uu = ue[u + 2], up1 = ue[u + 3], up2 = ue[u + 4];
                    const double t = tuc[u];
                    const double iu = ie[u + 1], iu2 = iu * iu;
                    const double v0 = v0_[u], v1 = v1_[u], v2 = v2_[u];
                    const double q2 = d0 * v0 * v0 + d1 * v1 * v1 + d2 * v2 * v2;
                    const double ir = 0.15 * iu;
                    lp += -(t + LOG_015) - 0.5 * q2 * iu2 - (P.ups_alpha + 1.0) * t - P.ups_beta * ir + jac * t;
                    sv0 += v0 * v0 * iu2; sv1 += v1 * v1 * iu2; sv2 += v2 * v2 * iu2;
                    double gu = -iu + q2 * iu2 * iu;
                    if (k >= 1 && k + 1 < K) {
                        const double du = 0.5 * (uu - 0.5 * (um1 + up1)) * iu;
                        lp += -0.5 * du * du;
                        gu += -du * 0.25 * (um1 + up1) * iu2;
                    }
                    if (k >= 2) {
                        const double i0 = ie[u];
                        const double du = 0.5 * (um1 - 0.5 * (um2 + uu)) * i0;
                        gu += du * 0.25 * i0;
                    }
                    if (k + 2 < K) {
                        const double i0 = ie[u + 2];
                        const double du = 0.5 * (up1 - 0.5 * (uu + up2)) * i0;
                        gu += du * 0.25 * i0;
                    }
                    gup[u] = uu * gu - (P.ups_alpha + 1.0) + P.ups_beta * ir + jac;
                    PW(B.o_ups + k, uu * (1.0 / 0.15));
                    w0_[u] = -d0 * v0 * iu2; w1_[u] = -d1 * v1 * iu2; w2_[u] = -d2 * v2 * iu2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            double gl_[UKV];
#pragma unroll
            for (int u = 0; u < UKV; ++u) gl_[u] = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int u = 0; u < UKV; ++u) wrow[MAXBW + kb + u] = i == 0 ? w0_[u] : (i == 1 ? w1_[u] : w2_[u]);
                wave_sync();
                double ww[WINV];
#pragma unroll
                for (int j = 0; j < WINV; ++j) ww[j] = wrow[kb + j];
                wave_sync();
                double tn = B.T[i][0];
#pragma unroll
                for (int d = 0; d < NTAP; ++d) {
                    const double t = tn;
                    if (d + 1 < NTAP) tn = B.T[i][d + 1];
                    __asm__ volatile("" ::: "memory");
#pragma unroll
                    for (int u = 0; u < UKV; ++u) gl_[u] = fma(t, ww[u + 2 * MAXBW - d], gl_[u]);
                }
            }
            // back to M1: d lp / d theta_ups is final, L^T w waits in the gradient row for the backward pass of this block
#pragma unroll
            for (int u = 0; u < UKV; ++u) { xrow[MAXBW + kb + u] = gup[u]; wrow[MAXBW + kb + u] = gl_[u]; }
            wave_sync();
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const int k = l32 + LPC * u;
                if (k < K) { GW(B.o_ups + k, xrow[MAXBW + k]); GW(B.o_x + k, wrow[MAXBW + k]); }
            }
            wave_sync();
            // one butterfly for the three sums: lane l ends with sum l & 3; the owner of d_i (lane 6 + 3 b + i) fetches sum i from lane i
            const double q3[4] = {sv0, sv1, sv2, 0.0};
            const double tot3 = sum32_by_lane<4>(q3, l32);
            const int iown = l32 - 6 - 3 * b;
            const double sv = __shfl(tot3, hb | (iown & 3));
            if (l32 >= 6 + 3 * b && l32 < 9 + 3 * b) {                     // d_i gradients of this block: lane 6 + 3 b + i
                const int i = iown;
                GW(B.o_d + i, -0.5 * sraw * sv - 6.0 + 5.0 * lean_rcp(sraw) + jac);
            }
        }
        BDRT_HW_PROF(7);
        __syncthreads();                                                   // B2: A_b x_b of all chains in Zh
        BDRT_HW_PROF(8);
        // ---- accumulate Z_hat; a parallel block contributes conj(Y)/|Y|^2 and parks Y ----------------------------------------
        double *Y = Yp + (size_t)B.yp_slot * TA * NC;
        double yr_[UNV], yi_[UNV];
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v, nn = n < nf ? n : 0;
            yr_[v] = Zh[swz(nn, c)]; yi_[v] = Zh[swz(nf + nn, c)];
        }
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v;
            if (n >= nf) continue;
            const double yr = yr_[v], yi = yi_[v];
            if (!B.is_parallel) { zre_a[v] += yr; zim_a[v] += yi; }
            else {
                Y[swz(n, c)] = yr; Y[swz(nf + n, c)] = yi;
                const double idn = lean_rcp(yr * yr + yi * yi);
                zre_a[v] += yr * idn;                                      // Z_hat_p (Parallel_modelcode.txt:47)
                zim_a[v] += -yi * idn;
            }
        }
        // (the next block's B1 separates these reads from its GEMM's writes to Zh)
        BDRT_HW_PROF(25);
    }

    // ================================================= x_sum prior, likelihood ==============================================
    double xs_term = 0.0;
    bool reject = false;
    if (P.use_x_sum) {
        const double xs_raw = hsum<LPC>(xsum_p);
        const double xs = xs_raw * P.x_sum_invscale;
        if (l32 == 0) lp += -0.5 * xs * xs;                                // x_sum ~ std_normal()
        reject = xs_raw < 0.0;                                             // real<lower=0> x_sum_raw
        xs_term = -xs_raw * P.x_sum_invscale * P.x_sum_invscale;
    }
    const double Rinf = 100.0 * __shfl(sraw, hb | 0), induc = __shfl(sraw, hb | 1) * P.induc_scale;
    const double s_res = 0.05 * __shfl(sraw, hb | 2), a_p = 0.05 * __shfl(sraw, hb | 3), a_r = 0.05 * __shfl(sraw, hb | 4),
                 a_i = 0.05 * __shfl(sraw, hb | 5);
    double gzr_[UNV], gzi_[UNV];
    {
        const int sp = io.spec ? io.spec[cc] : 0;
        const double *Zm = P.Z + (size_t)sp * N2;
        const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
        const double ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
        double sR = 0, sL = 0, sH = 0, sHz2 = 0, sHzr2 = 0, sHzi2 = 0;
        // (the phase's memory reads -- weights, measured spectrum, outlier parameters -- in one batch from a clamped row)
        double wn_[UNV], zmr_[UNV], zmi_[UNV], to0_[UNV], to1_[UNV];
        const int omode = P.outlier_mode;
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v, nn = n < nf ? n : 0;
            wn_[v] = P.w[nn]; zmr_[v] = Zm[nn]; zmi_[v] = Zm[nf + nn];
            to0_[v] = 0.0; to1_[v] = 0.0;
            if (omode) { to0_[v] = TH(P.o_so + nn); to1_[v] = TH(P.o_so + nf + nn); }
        }
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v;
            gzr_[v] = 0.0; gzi_[v] = 0.0;
            if (n >= nf) continue;
            const double wn = wn_[v];
            const double zr = zre_a[v] + Rinf;
            const double zi = zim_a[v] + induc * wn;
            double so_re = 0.0, so_im = 0.0, r0 = 0.0, r1 = 0.0, t0 = 0.0, t1 = 0.0;
            if (P.outlier_mode) {
                t0 = to0_[v]; t1 = to1_[v];
                r0 = lean_exp(t0); r1 = lean_exp(t1);
                PW(P.o_so + n, r0); PW(P.o_so + nf + n, r1);
                if (P.outlier_mode == 1) so_re = so_im = 0.05 * r0 * r1;
                else { so_re = 0.05 * r0; so_im = 0.05 * r1; }
            }
            const double common = ar2 * zr * zr + ai2 * zi * zi;
            const double s2_re = c0 + ap2 * zr * zr + common + so_re * so_re;
            const double s2_im = c0 + ap2 * zi * zi + common + so_im * so_im;
            const double e_re = zmr_[v] - zr, e_im = zmi_[v] - zi;
            const double prod = s2_re * s2_im, ip = lean_rcp(prod);
            const double w_re = s2_im * ip, w_im = s2_re * ip;
            lp += -0.5 * lean_log(prod) - 0.5 * e_re * e_re * w_re - 0.5 * e_im * e_im * w_im;
            const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;
            const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
            const double gzr = e_re * w_re + 2.0 * zr * (h_re * (ap2 + ar2) + h_im * ar2);
            const double gzi = e_im * w_im + 2.0 * zi * (h_im * (ap2 + ai2) + h_re * ai2);
            gzr_[v] = gzr; gzi_[v] = gzi;
            sR += gzr;
            sL += gzi * wn;
            sH += h_re + h_im;
            sHz2 += h_re * zr * zr + h_im * zi * zi;
            sHzr2 += (h_re + h_im) * zr * zr;
            sHzi2 += (h_re + h_im) * zi * zi;
            if (P.outlier_mode == 1) {
                const double dso = 2.0 * so_re * (h_re + h_im);
                GW(P.o_so + n, r0 * (0.05 * r1 * dso - P.so_lambda) + jac);
                GW(P.o_so + nf + n, 0.05 * r0 * r1 * dso - (P.so_alpha + 1.0) + P.so_beta * lean_rcp(r1) + jac);
                lp += -P.so_lambda * r0 - (P.so_alpha + 1.0) * t1 - P.so_beta * lean_rcp(r1) + jac * (t0 + t1);
            } else if (P.outlier_mode == 2) {
                GW(P.o_so + n, r0 * (0.05 * 2.0 * so_re * h_re - P.so_lambda) + jac);
                GW(P.o_so + nf + n, r1 * (0.05 * 2.0 * so_im * h_im - P.so_lambda) + jac);
                lp += -P.so_lambda * (r0 + r1) + jac * (t0 + t1);
            }
            if (io.Z_hat && valid) { io.Z_hat[(size_t)c * N2 + n] = zr; io.Z_hat[(size_t)c * N2 + nf + n] = zi; }
            if (io.sigma_tot && valid) {
                io.sigma_tot[(size_t)c * N2 + n] = sqrt(s2_re);
                io.sigma_tot[(size_t)c * N2 + nf + n] = sqrt(s2_im);
            }
        }
        // the six sums in one butterfly (sum32_by_lane: lane j ends with sum j, which is where its scalar gradient is wanted); the factors
        // 0.05 * 2 * s_res etc. are 0.005 x the lane's own raw value (s_res = 0.05 raw_2, alpha_* = 0.05 raw_3..5)
        const double q6[8] = {sR, sL, sH, sHz2, sHzr2, sHzi2, 0.0, 0.0};
        const double tot6 = sum32_by_lane<8>(q6, l32);
        if (l32 < 6) {
            const double dl = (l32 == 0 ? 100.0 : (l32 == 1 ? P.induc_scale : 0.005 * sraw)) * tot6;
            const int j = l32 < 2 ? l32 : P.o_err + (l32 - 2);
            gsc = sraw * (dl - sraw) + jac;
            GW(j, gsc);
        }
    }

    BDRT_HW_PROF(26);
    // ================================================= backward, block by block ===========================================
#pragma unroll 1
    for (int b = 0; b < nblocks; ++b) {
        const DevBlock &B = P.blk[b];
        const int K = B.K;
        const double *Y = Yp + (size_t)B.yp_slot * TA * NC;
        // operand of A_b^T: g_Zhat, or J^T g_Zhat through Z_hat_p = conj(Y)/|Y|^2 (times xp_scale) for a parallel block
        double yr_[UNV], yi_[UNV];
#pragma unroll
        for (int v = 0; v < UNV; ++v) { yr_[v] = 1.0; yi_[v] = 0.0; }
        if (B.is_parallel) {
#pragma unroll
            for (int v = 0; v < UNV; ++v) {
                const int n = l32 + LPC * v, nn = n < nf ? n : 0;
                yr_[v] = Y[swz(nn, c)]; yi_[v] = Y[swz(nf + nn, c)];
            }
        }
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v;
            if (n >= nf) continue;
            double rr = gzr_[v], ri = gzi_[v];
            if (B.is_parallel) {
                const double yr = yr_[v], yi = yi_[v];
                const double dn = yr * yr + yi * yi, id2 = lean_rcp(dn * dn);
                const double dd = (yi * yi - yr * yr) * id2, doff = 2.0 * yr * yi * id2;
                rr = (gzr_[v] * dd + gzi_[v] * doff) * B.x_scale;
                ri = (-gzr_[v] * doff + gzi_[v] * dd) * B.x_scale;
            }
            Zh[swz(n, c)] = rr;
            Zh[swz(nf + n, c)] = ri;
        }
        BDRT_HW_PROF(27);
        __syncthreads();                                                   // B3: operand of all chains in Zh
        BDRT_HW_PROF(28);
        gemm_sw<NWV, GPFV>(B.BkA, B.tilesK, B.rpairsA, Zh, Xs, wave, lane);   // Xs = A_b^T (.)
        BDRT_HW_PROF(29);
        __syncthreads();                                                   // B4
        BDRT_HW_PROF(30);
        {
            double tx_[UKV], gl_[UKV], ex_[UKV], ag_[UKV];
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const int k = l32 + LPC * u, kk = k < K ? k : 0;
                tx_[u] = TH(B.o_x + kk);
                gl_[u] = GR(B.o_x + kk);                                   // L^T w of this block, parked by the prior phase
                ag_[u] = Xs[swz(kk, c)];
            }
#pragma unroll
            for (int u = 0; u < UKV; ++u) tx_[u] = (l32 + LPC * u < K && B.is_pos) ? tx_[u] : 0.0;
#pragma unroll
            for (int u = 0; u < UKV; ++u) ex_[u] = lean_exp(tx_[u]);
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const int k = l32 + LPC * u;
                if (k < K) {
                    const double graw = ag_[u] + gl_[u] + xs_term;
                    GW(B.o_x + k, B.is_pos ? ex_[u] * graw + jac : graw);
                }
            }
        }
        // (the next block's B3 comes after every wave has left this epilogue: Xs and Zh are free again by then)
        BDRT_HW_PROF(31);
    }
    lp = hsum<LPC>(lp);
    const bool rej = __shfl((int)reject, hb) != 0;
    if (l32 == 0 && io.lp && valid) io.lp[c] = rej ? -INFINITY : lp;
    __syncthreads();
#undef BDRT_HW_PROF
}

}  // namespace bdrt
