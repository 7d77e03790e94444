/*
 * bdrt_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * Plain-C restatement of the reference's hot path, used only by tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg as the checker / reported CPU baseline.  The shipped path is the
 * HIP library under bayes_drt_amd/csrc; nothing there includes, links or calls this file.
 *
 * Parity pinning:
 *   - matrices:   against tests/golden/mat_*.npz, ddt_*.npz  (outputs of bayes_drt/matrices.py run here)
 *   - forward:    against the 37 stored Stan `optimizing` results tests/golden/kat_*.npz
 *                 (code_EchemActa/map_results/obj_*.pkl) -- transformed parameters to <=1e-12
 *   - gradient:   central finite differences of this file's own log-density + stationarity at the
 *                 stored MAPs.  The log-density VALUE and the L-BFGS / NUTS iterate paths are not
 *                 stored by any reference artefact: "parity unpinned" for those (Stan 2.19.1 is a
 *                 third-party dependency absent from /root/reference: pystan==2.19.1.1, setup.py:21).
 */
#ifndef BDRT_ORACLE_H
#define BDRT_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAX_BLOCKS 3

/* kernel ids for orc_build_A (reference: bayes_drt/matrices.py:27-117 get_A_func) */
enum {
    ORC_KERNEL_DRT = 0,            /* matrices.py:45-52 */
    ORC_KERNEL_DDT_BLOCK_PLANAR = 1,   /* matrices.py:59-70  1/(x tanh x)          */
    ORC_KERNEL_DDT_BLOCK_SPHER = 2,    /* matrices.py:72-80  tanh x/(x - tanh x)   */
    ORC_KERNEL_DDT_TRANS_PLANAR = 3    /* matrices.py:83-92  tanh x / x            */
};

/* One distribution block of a Stan model (series or parallel). */
typedef struct {
    int nf;                       /* number of measured frequencies (Stan N/2)                     */
    int nblocks;                  /* 1..3: [series] + parallel blocks, in Stan declaration order    */
    int K[ORC_MAX_BLOCKS];
    int is_parallel[ORC_MAX_BLOCKS];
    int nonneg[ORC_MAX_BLOCKS];   /* vector<lower=0> x ; parallel blocks are always lower=0        */
    double x_scale[ORC_MAX_BLOCKS];       /* xp_scale (parallel blocks), 1 otherwise               */
    const double *A[ORC_MAX_BLOCKS];      /* [2nf x K] row-major, rows = [A_re ; A_im]             */
    const double *L0[ORC_MAX_BLOCKS];     /* [K x K] row-major, already mode-scaled (P4)           */
    const double *L1[ORC_MAX_BLOCKS];
    const double *L2[ORC_MAX_BLOCKS];
    const double *Z;              /* [2nf] stacked [Z' ; Z'']                                      */
    const double *freq;           /* [nf]                                                          */
    double sigma_min, ups_alpha, ups_beta, induc_scale;
    int outlier_mode;             /* 0 none | 1 Series_*outliers (raw[nf],scale[nf]) | 2 stacked raw[2nf] */
    double so_lambda, so_alpha, so_beta;
    int use_x_sum;                /* Series-Parallel / Series-2Parallel: x_sum ~ std_normal()      */
    double x_sum_invscale;
} orc_model;

int orc_num_params(const orc_model *m);
/* offsets of the parameter groups in the unconstrained vector; returns D */
int orc_layout(const orc_model *m, int *o_x, int *o_err, int *o_so, int *o_ups, int *o_d, unsigned char *is_pos);

/* log-density (up to the constants Stan's `~` drops) and gradient w.r.t. the unconstrained
 * parameters.  jacobian=1: sampling (log|J| added), 0: optimizing.  Returns 0, or 1 if the point
 * is rejected by a declared constraint (lp=-inf, grad=0). */
int orc_logp_grad(const orc_model *m, const double *theta, int jacobian, double *lp, double *grad);

/* transformed parameters for known-answer tests; any output pointer may be NULL */
int orc_forward(const orc_model *m, const double *theta, double *Z_hat, double *sigma_tot, double *q_all,
                double *ups_all, double *dups_all, double *x_sum);

/* constrain: unconstrained theta -> Stan parameter values (exp for lower=0) */
void orc_constrain(const orc_model *m, const double *theta, double *params);
void orc_unconstrain(const orc_model *m, const double *params, double *theta);

/* A[n,m] = trapz_{y in linspace(-20,20,1000)} integrand(y; w_n, t_m)   (matrices.py:120-265)
 * part: 0 real, 1 imag.  dist_series: 1 -> Z_D, 0 -> 1/Z_D (DDT only).  use_ct/k_ct: matrices.py:61-65.
 * toeplitz=1 follows matrices.py:213-242 (first column, first row, toeplitz(c,r)); returns -1 if
 * r[0]!=c[0] (matrices.py:239-241). */
int orc_build_A(const double *freq, int nf, const double *tau, int k, double eps, int kernel, int part,
                int dist_series, int use_ct, double k_ct, int toeplitz, double *out);
/* the same with another basis function of get_basis_func (matrices.py:8-24) */
#define ORC_BASIS_GAUSSIAN 0
#define ORC_BASIS_COLE_COLE 1
#define ORC_BASIS_ZIC 2
int orc_build_A_basis(const double *freq, int nf, const double *tau, int k, double eps, int kernel, int part,
                      int dist_series, int use_ct, double k_ct, int toeplitz, int basis, double *out);
/* construct_L for any frequencies against any tau ([nf x k]); basis Zic: order 0 only (matrices.py:316-318) */
void orc_build_L_rect(const double *freq, int nf, const double *tau, int k, double eps, const double *coef4, int basis,
                      double *out);
/* L[n,m] = sum_j coef[j] * d^j/dy^j exp(-(eps y)^2) at y = ln(1/(w_n t_m)), w_n = 2 pi (1/(2 pi t_n))
 * (matrices.py:268-325); coef[4] weights derivative orders 0..3 */
void orc_build_L(const double *tau, int k, double eps, const double *coef4, double *out);
/* closed-form penalty matrices (matrices.py:328-411); coef[3] weights orders 0..2; toeplitz: :396-405 */
void orc_build_M(const double *tau, int k, double eps, const double *coef3, int toeplitz, double *out);

#ifdef __cplusplus
}
#endif
#endif
