/*
 * bdrt.h -- C ABI of libbdrt.so, the MI355X (gfx950) hot path of bayes-drt.
 *
 * The reference (jdhuang-csm/bayes-drt) has no FFI: its hot path is reached from Python through three
 * internal seams of bayes_drt.inversion.Inverter:
 *   (1) construct_A / construct_L / construct_M          bayes_drt/matrices.py:120, :268, :366
 *   (2) StanModel.optimizing / StanModel.sampling        bayes_drt/inversion.py:1216, :1218-1221 (pystan 2.19.1.1)
 *   (3) cvxopt.solvers.qp via Inverter._convex_opt       bayes_drt/inversion.py:1043-1067
 * Each entry point below replaces one of those calls; the ctypes binding a maintainer would add is shown
 * in INTEGRATION.md and implemented in bayes_drt_amd/_lib.py.
 *
 * Conventions: all arrays are C-contiguous fp64 unless stated; the caller owns every buffer passed in;
 * the library copies inputs to HBM at bdrt_problem_create and owns device memory until bdrt_problem_destroy.
 * Functions return 0 on success, a negative code on error (text via bdrt_last_error()).  A non-finite or
 * rejected log-density is NOT an error: lp = -inf for that row.
 * Host-pointer entry points move data over PCIe; the *_dev entry points take device pointers (HBM-resident
 * inputs) and a hipStream_t (passed as void*).
 */
#ifndef BDRT_H
#define BDRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDRT_MAX_BLOCKS 3

/* integrand ids: bayes_drt/matrices.py:27-117 get_A_func */
enum {
    BDRT_KERNEL_DRT = 0,              /* matrices.py:45-52 */
    BDRT_KERNEL_DDT_BLOCK_PLANAR = 1, /* matrices.py:59-70 */
    BDRT_KERNEL_DDT_BLOCK_SPHER = 2,  /* matrices.py:72-80 */
    BDRT_KERNEL_DDT_TRANS_PLANAR = 3  /* matrices.py:83-92 */
};

/* basis function ids: bayes_drt/matrices.py:8-24 get_basis_func */
enum {
    BDRT_BASIS_GAUSSIAN = 0,          /* matrices.py:12-13 (the only one Inverter accepts, inversion.py:38-39) */
    BDRT_BASIS_COLE_COLE = 1,         /* matrices.py:15-17, 0 < epsilon < 1 */
    BDRT_BASIS_ZIC = 2                /* matrices.py:19-21, epsilon unused */
};

/* ---- (1) matrix construction ------------------------------------------------------------------ */

/* replaces construct_A (matrices.py:120-265).  out: [nf x k] row-major.  part 0=real 1=imag.
 * dist_series: 1 integrate Z_D, 0 integrate 1/Z_D (DDT).  toeplitz=1: first column + first row only
 * (matrices.py:213-242), returns -2 if r[0]!=c[0] (matrices.py:239-241). */
int bdrt_build_A(const double *freq, int nf, const double *tau, int k, double eps, int kernel_id, int part,
                 int dist_series, int use_ct, double k_ct, int toeplitz, double *out);
/* construct_A with another basis function (the `basis` argument of matrices.py:120); bdrt_build_A = gaussian */
int bdrt_build_A_basis(const double *freq, int nf, const double *tau, int k, double eps, int kernel_id, int part,
                       int dist_series, int use_ct, double k_ct, int toeplitz, int basis_id, double *out);
/* replaces construct_L (matrices.py:268-325) in the collocated form Inverter calls it (frequencies = 1/(2 pi tau),
 * inversion.py:2302-2307); coef4 weights derivative orders 0..3.  out: [k x k] */
int bdrt_build_L(const double *tau, int k, double eps, const double *coef4, double *out);
/* construct_L as the function is written: any `frequencies` against any `tau`, out [nf x k]; freq == NULL: collocated
 * (nf == k).  basis_id: gaussian, or Zic with order 0 only (matrices.py:316-318); Cole-Cole has no derivative there. */
int bdrt_build_L_rect(const double *freq, int nf, const double *tau, int k, double eps, const double *coef4, int basis_id,
                      double *out);
/* replaces construct_M (matrices.py:366-411); coef3 weights orders 0..2.  out: [k x k] */
int bdrt_build_M(const double *tau, int k, double eps, const double *coef3, int toeplitz, double *out);

/* ---- (2) the Stan model: data block, log-posterior + gradient, optimizing, sampling -------------- */

/* The Stan `data` block of the model families in bayes_drt/stan_model_files/ (SURVEY.md 8(a) S1-S6),
 * as assembled by Inverter._prep_stan_data (inversion.py:1684-2122), written as "blocks":
 *   Series(_pos)            1 series block                         Series_modelcode.txt
 *   Parallel                1 parallel block, use_x_sum=0          Parallel_modelcode.txt
 *   Series-Parallel(_pos)   series + 1 parallel, use_x_sum=1       Series-Parallel_modelcode.txt
 *   Series-2Parallel(_pos)  series + 2 parallel, use_x_sum=1       Series-2Parallel_modelcode.txt
 *   *_outliers              outlier_mode 1 (Series: raw[nf],scale[nf]) or 2 (stacked raw[2nf])
 * Parameter order = Stan declaration order: Rinf_raw, induc_raw, x blocks, sigma_res_raw, alpha_prop_raw,
 * alpha_re_raw, alpha_im_raw, [sigma_out_raw(, sigma_out_scale)], ups blocks, (d0,d1,d2) blocks. */
typedef struct {
    int nf;                                /* measured frequencies (Stan N/2)                          */
    int nblocks;
    int K[BDRT_MAX_BLOCKS];
    int is_parallel[BDRT_MAX_BLOCKS];
    int nonneg[BDRT_MAX_BLOCKS];           /* vector<lower=0> x (parallel blocks are always lower=0)   */
    double x_scale[BDRT_MAX_BLOCKS];       /* xp_scale                                                 */
    const double *A[BDRT_MAX_BLOCKS];      /* [2nf x K] stacked [A_re ; A_im]                          */
    const double *L0[BDRT_MAX_BLOCKS];     /* [K x K], mode-scaled as in inversion.py:1725-1737        */
    const double *L1[BDRT_MAX_BLOCKS];
    const double *L2[BDRT_MAX_BLOCKS];
    const double *freq;                    /* [nf]                                                     */
    int n_spectra;                         /* spectra sharing the grids (multi-spectrum batch)         */
    const double *Z;                       /* [n_spectra x 2nf] stacked [Z' ; Z''] per spectrum        */
    double sigma_min, ups_alpha, ups_beta, induc_scale;
    int outlier_mode;
    double so_lambda, so_alpha, so_beta;
    int use_x_sum;
    double x_sum_invscale;
} bdrt_dat;

typedef struct bdrt_problem bdrt_problem;

bdrt_problem *bdrt_problem_create(const bdrt_dat *dat);   /* NULL on error */
void bdrt_problem_destroy(bdrt_problem *p);
int bdrt_num_params(const bdrt_problem *p);               /* D: length of the unconstrained vector */
/* diagnostics: which 16-column tile evaluator bdrt_logp_grad / the sampler use for this problem.  0 dense L (MFMA), 1 banded
 * Toeplitz L (generic tile), 2 one-block fast tile with the A fragments streamed from L2, 3 general half-wave tile (several
 * blocks), 4 one-block fast tile with the A operands from an LDS-resident Toeplitz generator table (A_re, A_im exactly
 * Toeplitz -- equal log spacing of the frequencies and of tau --, nf, K >= 32; env BDRT_STREAM_A=1 at bdrt_problem_create forces 2), 5 a problem beyond the
 * LDS budget of all of these (more than 128 frequencies, ~200 basis functions per distribution ...): the streamed evaluator, one
 * workgroup per point, vectors in an HBM workspace -- slow but working, the reference takes any grid (inversion.py:2127-2209);
 * env BDRT_BIG=1 at bdrt_problem_create forces it */
int bdrt_problem_evaluator(const bdrt_problem *p);
/* is_pos[D]: 1 where the Stan parameter is declared <lower=0> (log transform) */
int bdrt_param_is_pos(const bdrt_problem *p, unsigned char *is_pos);
/* replace the measured spectra of an existing problem (same grids): Z [n_spectra x 2nf] */
int bdrt_problem_set_Z(bdrt_problem *p, const double *Z, int n_spectra);

/* log_prob + gradient on the unconstrained scale for B points (Stan's log_prob_grad, evaluated by pystan
 * inside optimizing/sampling: inversion.py:1216-1221).  theta [B x D]; spec[B] selects the spectrum of each
 * row (NULL: all 0); jacobian 1 = sampling, 0 = optimizing; lp [B]; grad [B x D]. */
int bdrt_logp_grad(bdrt_problem *p, const double *theta, const int *spec, int B, int jacobian, double *lp,
                   double *grad);
/* same with device pointers (inputs already in HBM); stream = hipStream_t */
int bdrt_logp_grad_dev(bdrt_problem *p, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                       double *d_grad, void *stream);
/* constrained parameters + transformed parameters (Stan `transformed parameters` block) for B points:
 * params [B x D] (constrained), Z_hat [B x 2nf], sigma_tot [B x 2nf]; any output may be NULL */
int bdrt_transformed(bdrt_problem *p, const double *theta, const int *spec, int B, double *params, double *Z_hat,
                     double *sigma_tot);

/* replaces StanModel.optimizing (inversion.py:1216): L-BFGS on the unconstrained scale, no Jacobian.
 * One optimisation per row of init (n_fits rows, spectrum spec[i]); every log_prob+grad is evaluated on the GPU. */
typedef struct {
    int max_iter;           /* Stan `iter` (inversion.py:1077: 50000): cap on L-BFGS iterations          */
    int history;            /* 5                                                                         */
    double init_alpha;      /* 1e-3                                                                      */
    double tol_obj;         /* 1e-12                                                                     */
    double tol_rel_obj;     /* 1e4  (x machine eps)                                                      */
    double tol_grad;        /* 1e-8                                                                      */
    double tol_rel_grad;    /* 1e7  (x machine eps)                                                      */
    double tol_param;       /* 1e-8                                                                      */
    /* second-order polish (not in Stan; bdrt_newton.h): after at most lbfgs_before_newton L-BFGS iterations a
     * damped Newton iteration with the full Hessian (2D batched gradient evaluations per step) runs until
     * |grad|_inf < newton_tol.  newton_max_iter = 0 gives the plain Stan-style L-BFGS.                   */
    int newton_max_iter;    /* 2000                                                                      */
    int lbfgs_before_newton;/* 0 (measured: L-BFGS iterations before the Newton iteration only add time)    */
    double newton_tol;      /* 1e-8                                                                      */
} bdrt_opt_options;
typedef struct {
    int iterations;         /* L-BFGS iterations                                                         */
    int n_evals;            /* log_prob+grad evaluations, all phases                                     */
    int return_code;        /* 0 converged, 1 iteration cap, 2 damping exhausted, <0 failure             */
    double lp;
    double grad_norm;       /* 2-norm                                                                    */
    int newton_iterations;
    double grad_inf;        /* max-norm at the returned point                                            */
} bdrt_opt_report;
void bdrt_opt_defaults(bdrt_opt_options *o);
int bdrt_optimize(bdrt_problem *p, const double *init_theta, const int *spec, int n_fits, const bdrt_opt_options *opts,
                  double *theta_out, bdrt_opt_report *reports);

/* replaces StanModel.sampling (inversion.py:1218-1221): NUTS, diagonal metric, Stan-2.19 style adaptation.
 * The whole transition loop runs on the GPU (one workgroup per 16 chains); the host only relaunches. */
typedef struct {
    double adapt_delta;     /* 0.9  (inversion.py:1221)                  */
    double adapt_t0;        /* 10   (inversion.py:1221)                  */
    double adapt_gamma;     /* 0.05                                      */
    double adapt_kappa;     /* 0.75                                      */
    int max_treedepth;      /* 10                                        */
    int init_buffer;        /* 75                                        */
    int term_buffer;        /* 50                                        */
    int base_window;        /* 25                                        */
    double init_radius;     /* 2: U(-2,2) on the unconstrained scale     */
    double max_deltaH;      /* 1000                                      */
    double stepsize0;       /* 1                                         */
} bdrt_nuts_control;
typedef struct {
    int64_t n_leapfrog;     /* log_prob+grad evaluations (leapfrogs)     */
    int n_divergent;        /* post-warm-up                              */
    int n_max_treedepth;    /* post-warm-up iterations that hit the cap  */
    double stepsize;        /* adapted step size                         */
    double mean_accept;     /* mean accept_stat post-warm-up             */
} bdrt_chain_diag;
/* bdrt_sampler_create checks the control block the way Stan's services do: adapt_delta in (0,1); adapt_gamma, adapt_kappa,
 * adapt_t0, stepsize0, max_deltaH > 0; init_radius >= 0; window sizes >= 0; max_treedepth in [1,10]; NaN fails every check. */
void bdrt_nuts_defaults(bdrt_nuts_control *c);

typedef struct bdrt_sampler bdrt_sampler;
/* n_units chains; unit u samples spectrum spec[u] (NULL: 0) with RNG stream (seed, chain_id[u]) (NULL: u);
 * init_theta [n_units x D] or NULL (random U(-r,r), retried until finite).
 * Size limits (NULL + bdrt_last_error beyond them): D <= 864 parameters for problems on the LDS-resident evaluators
 * (bdrt_problem_evaluator 0..4: twenty-seven elements per lane of a chain's half-wave), D <= 8192 for problems on the streamed
 * evaluator (code 5: every row in HBM, up to sixteen elements per thread of the cooperative stage).  Stan has no limit
 * (reference bayes_drt/inversion.py:2127-2209); evaluation and MAP have none here either. */
bdrt_sampler *bdrt_sampler_create(bdrt_problem *p, int n_units, const int *spec, const int *chain_id, int warmup,
                                  int n_draws, uint64_t seed, const double *init_theta, const bdrt_nuts_control *ctrl);
void bdrt_sampler_destroy(bdrt_sampler *s);
/* advance every unfinished chain by at most `rounds` leapfrogs (one launch, asynchronous on `stream`);
 * *all_done is set when every chain has produced warmup+n_draws iterations (requires a sync: pass NULL to skip) */
int bdrt_sampler_advance(bdrt_sampler *s, int rounds, int *all_done);
int bdrt_sampler_sync(bdrt_sampler *s);
/* run to completion */
int bdrt_sampler_run(bdrt_sampler *s);
/* draws [n_units x n_draws x D] unconstrained; diag [n_units]; either may be NULL */
int bdrt_sampler_results(bdrt_sampler *s, double *draws_unconstrained, double *lp, bdrt_chain_diag *diag);
/* run to completion.  A run that starts on the 16-chains-per-workgroup kernel (more than four chains per CU) hands its last
 * live chains to the one-chain-per-workgroup kernel once that finishes them sooner (BDRT_TAIL_MIGRATION=0 forbids it);
 * bdrt_sampler_tail_units tells how many chains were handed over (0: none). */
int bdrt_sampler_tail_units(bdrt_sampler *s);
/* A run with more than 16 units per CU re-packs its live chains into fewer 16-chain workgroups whenever 1/16 of the
 * workgroups can be dropped (finished chains), so that the MFMA tiles stay full until fewer than 16 live chains per CU are left;
 * the chains continue bit for bit.  Number of such re-packings so far (BDRT_COMPACTION=0 disables them). */
int bdrt_sampler_compactions(bdrt_sampler *s);
/* which kernel advances the chains now: 0 sixteen chains per workgroup (bdrt_nuts.hip), 1 one chain per workgroup with its
 * state in LDS (bdrt_solo.h: few chains of the single-DRT family), 2 one chain per workgroup, general block model
 * (bdrt_solo_wide.h: few chains of any other model on log-uniform grids), 3 one chain per WAVEFRONT (bdrt_wave.h: the single-DRT
 * family while more than 2.5 and at most 8 chains per CU are running: chosen per launch by the number of live chains), 4 one
 * chain per workgroup on the streamed evaluator of a problem beyond the LDS budget (bdrt_big.h) */
int bdrt_sampler_kind(bdrt_sampler *s);
/* total leapfrogs executed so far, summed over chains (device counter) */
int64_t bdrt_sampler_total_leapfrogs(bdrt_sampler *s);
/* HIP-event time (ms) and launch count of the NUTS kernel accumulated since creation / last reset */
int bdrt_sampler_kernel_time(bdrt_sampler *s, double *ms_total, int64_t *n_launches, int reset);
/* phase profile of the NUTS kernel (development aid): cycles32[0..9] = phases of the log-posterior tile, [10..16] =
 * NUTS stages, summed over workgroups since the last call; enable != 0 switches the in-kernel clock reads on */
int bdrt_sampler_phase_profile(bdrt_sampler *s, int enable, long long *cycles32);
/* convenience: create + run + results.  draws [n_units x n_draws x D] */
int bdrt_sample(bdrt_problem *p, int n_units, const int *spec, const int *chain_id, int warmup, int n_draws,
                uint64_t seed, const double *init_theta, const bdrt_nuts_control *ctrl, double *draws, double *lp,
                bdrt_chain_diag *diag);

/* ---- (3) ridge: Gram matrices and the box-constrained QP (replaces _convex_opt, inversion.py:1043-1067) -- */
/* P = WA_re^T WA_re + WA_im^T WA_im + L2mat ; q = -(WA_re^T WZ_re + WA_im^T WZ_im) + L1vec  (inversion.py:1045-1052)
 * WA [2nf x n] stacked; WZ [2nf]; P [n x n]; q [n]  */
int bdrt_gram(const double *WA, const double *WZ, int nrows, int n, const double *L2mat, const double *L1vec, double *P,
              double *q);
/* min 1/2 x^T P x + q^T x  s.t. x >= lo (lo[i] = -inf allowed): primal-dual interior point with cvxopt-like
 * tolerances (abstol 1e-7, reltol 1e-6, feastol 1e-7).  Returns iterations (>=0) or <0. */
int bdrt_qp_box(const double *P, const double *q, const double *lo, int n, double *x, double *primal_objective);
/* The same solver on the GPU for a batch of nb problems that share n and lo (one workgroup per problem, KKT matrix
 * factored in LDS): P [nb][n][n], q [nb][n], x [nb][n], primal_objective [nb] or NULL, iterations [nb] or NULL.
 * This is what Inverter.ridge_fit / ridge_ReImCV call (reference inversion.py:1043-1067 inside the loops at :560-740 and
 * :902-945); bdrt_qp_box above is the host implementation of the identical algorithm, kept as the checker.
 * Returns 0, or <0 (-3: a KKT matrix was not positive definite, -4: iteration limit). */
int bdrt_qp_box_batch(const double *P, const double *q, const double *lo, int n, int nb, double *x,
                      double *primal_objective, int *iterations);

/* The whole hierarchical ridge fit of Inverter.ridge_fit (reference inversion.py:518-740) for a batch of nb independent
 * fits in ONE launch (one workgroup per fit): the hyper-lambda outer loop -- lambda update (:947-983), penalty matrix
 * (:695-700), the QP of _convex_opt (:1043-1067), convergence test (:730-736) -- runs on the device.  ridge_ReImCV
 * (:902-945) is one call with nb = 2 x len(lambdas).
 *   unknowns: n (series: [R_inf, L/1e-4, x[K]], off = 2; parallel: x[K], off = 0)
 *   G [ng][n][n], qbase [ng][n]: Gram matrices WA^T WA and q = -WA^T WZ + L1_vec of the ng distinct data parts
 *     (Re-Im CV: real part, imaginary part); gsel[nb] picks one per fit
 *   base [3][n][n]: padded M0, M1, M2 (penalty 1 = 'integral') or L_o^T L_o (penalty 0 = 'discrete');
 *     Ls [3][K][n]: the padded L_o themselves (discrete only)
 *   lambda0[nb]; lam0s[nb][3], betas[nb][3]: (2a-2)/(2b) resp. (2a-1)/(2b) and 2a of the reference's prior terms (:608-628)
 *   per iteration, from the previous coefficients x:
 *     discrete:  lambda_o = 1 / ((L_o x)^2 / (beta_o - 1) + 1 / lam0_o)            (:947-954), or with hl_fbeta > 0
 *                lambda_o = lambda0 / ((L_o x)^2 / (max (L_o x)^2 * hl_fbeta) + 1)  (:956-964); 1 for the off leading unknowns
 *     integral:  c = factor_o x (100, 10, 1), C_j = sum_{r != j} c_r sqrt(lambda_r) M_rj c_j, d = c^2 diag(M) + 2b,
 *                lambda = (C^2 - sign(C) C sqrt(4 d (2a-2) + C^2) + 2 d (2a-2)) / (2 d^2), floored at 1e-15  (:973-983)
 *     P = G + sum_o reg_ord[o] sqrt(lambda_o) base_o sqrt(lambda_o);  x = argmin 1/2 x'Px + q'x, x >= lo
 *     stop when mean |(x - x_prev) / x_prev| < xtol (entry 1 -- the inductance -- excluded for the fits on data part g
 *     when bit g of zero_delta1 is set: the reference zeroes it when the inductance is not fitted or only the real part is,
 *     :733-734, so in a Re-Im cross-validation the real-part fits exclude it and the imaginary-part fits do not)
 *   hyper_lambda = 0: one QP with lambda = lambda0 (ordinary ridge).
 * Outputs: coef [nb][n], lam [nb][3][n], cost = 1/2 x'Px + q'x, fun = the QP's primal objective, iters (outer iterations),
 * flags (bit 0 converged, bit 2 a QP reached its iteration limit); optional per-iteration history (all four or none):
 * hist_coef [nb][max_iter][n], hist_lam [nb][max_iter][3][n], hist_fun / hist_cost [nb][max_iter]. */
typedef struct {
    int n, K, off;
    int penalty;            /* 0 discrete, 1 integral */
    int max_iter;
    int hyper_lambda;
    int zero_delta1;        /* bit g: fits with gsel == g leave entry 1 out of the convergence test (ng <= 31) */
    double xtol;
    double hl_fbeta;        /* <= 0: analytic discrete update */
    double reg_ord[3];
} bdrt_ridge_options;
int bdrt_ridge(const bdrt_ridge_options *opt, int nb, int ng, const double *G, const double *qbase, const int *gsel,
               const double *base, const double *Ls, const double *lo, const double *lambda0, const double *lam0s,
               const double *betas, const double *x0, double *coef, double *lam, double *cost, double *fun, int *iters,
               int *flags, double *hist_coef, double *hist_lam, double *hist_fun, double *hist_cost);

/* ---- (4) posterior post-processing on the device ------------------------------------
 * Replaces the numpy reductions applied to the HMC draws right after `sampling`:
 *   np.percentile(samples, q, axis=0)            reference bayes_drt/inversion.py:2560 (coef_percentile), :2702
 *                                                (predict_Z from Z_hat), :3068/:3085 (predict_Rp), :3096-3113 (predict_sigma)
 *   np.percentile(x_samples @ A.T + offsets, ..) reference inversion.py:2716-2735 (predict_Z on new frequencies)
 *
 * out[nq x ncols] (row-major) = percentile q[t] (0..100, numpy's default 'linear' rule, same lerp formula) over the
 * `rows` samples of every column of Y, where Y = X (Phi == NULL, ncols = K) or Y = X Phi^T + bias (Phi is [M x K]
 * row-major, bias [M] or NULL, ncols = M).  X is [rows x K] with row stride ldx >= K (doubles).  A column that
 * contains a NaN yields NaN (numpy behaviour).  Columns of up to 16384 samples are sorted in LDS, longer ones in an HBM
 * scratch buffer (np.percentile has no row limit either). */
int bdrt_percentiles(const double *X, int rows, int K, long ldx, const double *Phi, int M, const double *bias,
                     const double *q, int nq, double *out);
/* The same on the draws a sampler holds on the device (unconstrained parameters theta): samples = all draws of units
 * [unit_lo, unit_hi), columns [col0, col0 + ncols) of theta.  The draws are not copied to the host. */
int bdrt_sampler_percentiles(bdrt_sampler *s, int unit_lo, int unit_hi, int col0, int ncols, const double *Phi, int M,
                             const double *bias, const double *q, int nq, double *out);

/* Per-spectrum posterior summary on the CONSTRAINED scale (what the reference reduces fit['x'], fit['Rinf'], ... to:
 * np.mean(..., axis=0) inversion.py:2517-2519 and np.percentile(..., q, axis=0) :2560): samples = all draws of units
 * [unit_lo, unit_hi); <lower=0> parameters are exp(theta).  mean [D] (may be NULL), pct [nq x D].  This is what a
 * multi-GPU run gathers instead of raw draws. */
int bdrt_sampler_summary(bdrt_sampler *s, int unit_lo, int unit_hi, const double *q, int nq, double *mean, double *pct);
/* the same reduction on host-resident draws X [rows x K] (row stride ldx): is_pos[K] flags the exp() columns (NULL: none) */
int bdrt_summary(const double *X, int rows, int K, long ldx, const unsigned char *is_pos, const double *q, int nq,
                 double *mean, double *pct);
/* device pointer of the draws [n_units x n_draws x D] (unconstrained), valid until bdrt_sampler_destroy: lets a
 * collective library (RCCL) gather draws without a host round trip.  Synchronises the sampler's stream. */
const double *bdrt_sampler_draws_dev(bdrt_sampler *s);

/* ---- misc ----------------------------------------------------------------------------------------- */
const char *bdrt_last_error(void);
int bdrt_device_count(void);
int bdrt_set_device(int dev);
const char *bdrt_version(void);

#ifdef __cplusplus
}
#endif
#endif
