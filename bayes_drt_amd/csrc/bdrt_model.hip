// bdrt_model.hip -- problem object (Stan `data` block in HBM, matrices packed into MFMA fragment order),
// the batched log-posterior + gradient kernel, and their C ABI (include/bdrt.h section (2)).
//
// Replaces: pystan's log_prob/grad_log_prob evaluation of bayes_drt/stan_model_files/*_modelcode.txt
// (reference bayes_drt/inversion.py:1216-1221).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "bdrt_host.h"
#include "bdrt_big.h"

namespace bdrt {

static thread_local std::string g_last_error;
std::atomic<int> g_process_device{-1};

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// element (tile t, pair p, lane l, h) = M[16 t + (l & 15)][8 p + 4 h + (l >> 4)], zero outside [R x C]
template <class F>
static std::vector<double> pack_fragments(int tiles, int pairs, F elem)
{
    std::vector<double> out((size_t)tiles * pairs * 128, 0.0);
    for (int t = 0; t < tiles; ++t)
        for (int p = 0; p < pairs; ++p)
            for (int l = 0; l < 64; ++l)
                for (int h = 0; h < 2; ++h)
                    out[(((size_t)t * pairs + p) * 64 + l) * 2 + h] = elem(16 * t + (l & 15), 8 * p + 4 * h + (l >> 4));
    return out;
}

template <class T>
static int upload(Problem &P, const std::vector<T> &h, const T **dptr)
{
    void *d = nullptr;
    BDRT_HIP(hipMalloc(&d, h.size() * sizeof(T) + 16));
    BDRT_HIP(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    P.allocs.push_back(d);
    *dptr = (const T *)d;
    return 0;
}

int Problem::sync_dev()
{
    if (!d_dev) BDRT_HIP(hipMalloc((void **)&d_dev, sizeof(DevProblem)));
    BDRT_HIP(hipMemcpy(d_dev, &dev, sizeof(DevProblem), hipMemcpyHostToDevice));
    return 0;
}

int Problem::ensure_scratch(size_t rows)
{
    if (rows <= scratch_rows) return 0;
    size_t n = rows < 64 ? 64 : rows;
    if (d_theta) { hipFree(d_theta); hipFree(d_grad); hipFree(d_lp); hipFree(d_spec); }
    BDRT_HIP(hipMalloc((void **)&d_theta, n * dev.D * sizeof(double)));
    BDRT_HIP(hipMalloc((void **)&d_grad, n * dev.D * sizeof(double)));
    BDRT_HIP(hipMalloc((void **)&d_lp, n * sizeof(double)));
    BDRT_HIP(hipMalloc((void **)&d_spec, n * sizeof(int)));
    scratch_rows = n;
    return 0;
}

// the streamed evaluator of problems beyond the LDS budget (bdrt_big.h): grid-stride over the points, one workspace per workgroup
__global__ __launch_bounds__(BIG_NT) void big_eval_kernel(const DevProblem *__restrict__ Pp, double *ws, const double *theta, const int *spec,
                                                           int B, int jacobian, double *lp, double *grad, double *params, double *Zhat, double *sig)
{
    __shared__ double red[9 * 8];
    const DevProblem &P = *Pp;
    const size_t wsz = big_ws_doubles(P);
    for (int b = blockIdx.x; b < B; b += gridDim.x)
        big_eval(P, ws + (size_t)blockIdx.x * wsz, theta + (size_t)b * P.D, grad ? grad + (size_t)b * P.D : nullptr, lp ? lp + b : nullptr,
                 spec ? spec[b] : 0, jacobian, red, threadIdx.x, params ? params + (size_t)b * P.D : nullptr,
                 Zhat ? Zhat + (size_t)b * 2 * P.nf : nullptr, sig ? sig + (size_t)b * 2 * P.nf : nullptr);
}


int Problem::ensure_bigws(size_t doubles)
{
    if (doubles <= bigws_doubles) return 0;
    // (growing: hipFree waits for the device, so no launch that was handed the old buffer is still running)
    if (d_bigws) { hipFree(d_bigws); d_bigws = nullptr; bigws_doubles = 0; }
    BDRT_HIP(hipMalloc((void **)&d_bigws, doubles * sizeof(double)));
    bigws_doubles = doubles;
    return 0;
}

static int build_problem(Problem &P, const bdrt_dat *dat)
{
    if (!dat || dat->nf <= 0 || dat->nblocks < 1 || dat->nblocks > MAXB) {
        set_error("bdrt_problem_create: invalid nf/nblocks");
        return -1;
    }
    if (dat->n_spectra < 1 || !dat->Z || !dat->freq) {
        set_error("bdrt_problem_create: Z/freq missing or n_spectra < 1");
        return -1;
    }
    // The matrices and the frequency list must be finite: the structure detection below (Toeplitz generators, banded L) reads
    // a few rows and columns and would silently drop a stray NaN / inf elsewhere.  (Z may hold NaN: the log-posterior is then
    // not finite and every caller reports that, see bdrt.h.)
    {
        auto finite = [](const double *v, size_t n) { for (size_t i = 0; i < n; ++i) if (!std::isfinite(v[i])) return false; return true; };
        bool ok = finite(dat->freq, (size_t)dat->nf);
        for (int b = 0; b < dat->nblocks && ok; ++b) {
            const size_t K = (size_t)dat->K[b];
            if (dat->K[b] < 1 || !dat->A[b] || !dat->L0[b] || !dat->L1[b] || !dat->L2[b]) { set_error("bdrt_problem_create: block %d incomplete", b); return -1; }
            ok = finite(dat->A[b], 2 * (size_t)dat->nf * K) && finite(dat->L0[b], K * K) && finite(dat->L1[b], K * K) && finite(dat->L2[b], K * K);
        }
        if (!ok) { set_error("bdrt_problem_create: non-finite entry in freq / A / L0 / L1 / L2"); return -1; }
    }
    bind_process_device();
    BDRT_HIP(hipGetDevice(&P.device));
    DevProblem &D = P.dev;
    memset(&D, 0, sizeof(D));
    const int nf = dat->nf, N2 = 2 * nf;
    D.nf = nf;
    D.nblocks = dat->nblocks;
    D.outlier_mode = dat->outlier_mode;
    D.use_x_sum = dat->use_x_sum;
    D.sigma_min = dat->sigma_min;
    D.ups_alpha = dat->ups_alpha;
    D.ups_beta = dat->ups_beta;
    D.induc_scale = dat->induc_scale;
    D.so_lambda = dat->so_lambda;
    D.so_alpha = dat->so_alpha;
    D.so_beta = dat->so_beta;
    D.x_sum_invscale = dat->x_sum_invscale;

    // parameter layout = Stan declaration order (see include/bdrt.h)
    int o = 2;
    for (int b = 0; b < dat->nblocks; ++b) { P.o_x[b] = o; o += dat->K[b]; }
    D.o_err = o; o += 4;
    D.o_so = o;
    if (dat->outlier_mode) o += 2 * nf;
    for (int b = 0; b < dat->nblocks; ++b) { P.o_ups[b] = o; o += dat->K[b]; }
    for (int b = 0; b < dat->nblocks; ++b) { P.o_d[b] = o; o += 3; }
    D.D = o;
    P.is_pos.assign((size_t)o, 1);
    for (int b = 0; b < dat->nblocks; ++b)
        if (!dat->is_parallel[b] && !dat->nonneg[b])
            for (int k = 0; k < dat->K[b]; ++k) P.is_pos[P.o_x[b] + k] = 0;

    int XR = 0, LR = 0, npar = 0;
    int toep_ok[MAXB] = {0, 0, 0};
    const int rpairsA = cdiv(N2, 8), tilesA = cdiv(N2, 16);
    for (int b = 0; b < dat->nblocks; ++b) {
        const int K = dat->K[b];
        if (K < 3 || !dat->A[b] || !dat->L0[b] || !dat->L1[b] || !dat->L2[b]) {
            set_error("bdrt_problem_create: block %d: K < 3 or missing matrix", b);
            return -1;
        }
        DevBlock &B = D.blk[b];
        B.K = K;
        B.is_parallel = dat->is_parallel[b] ? 1 : 0;
        B.is_pos = (dat->is_parallel[b] || dat->nonneg[b]) ? 1 : 0;
        B.x_scale = dat->is_parallel[b] ? dat->x_scale[b] : 1.0;
        B.tilesA = tilesA;
        B.tilesL = cdiv(3 * K, 16);
        B.tilesK = cdiv(K, 16);
        B.kpairs = cdiv(K, 8);
        B.rpairsA = rpairsA;
        B.rpairsL = cdiv(3 * K, 8);
        B.o_x = P.o_x[b]; B.o_ups = P.o_ups[b]; B.o_d = P.o_d[b];
        B.yp_slot = B.is_parallel ? npar++ : 0;
        const double *A = dat->A[b];
        const double *Ls[3] = {dat->L0[b], dat->L1[b], dat->L2[b]};
        auto Aelem = [&](int r, int k) -> double { return (r < N2 && k < K) ? A[(size_t)r * K + k] : 0.0; };
        auto Lelem = [&](int r, int k) -> double {
            return (r < 3 * K && k < K) ? Ls[r / K][(size_t)(r % K) * K + k] : 0.0;
        };
        const int ra8 = 8 * rpairsA;
        auto Telem = [&](int k, int r) -> double { return r < ra8 ? Aelem(r, k) : Lelem(r - ra8, k); };
        int rc;
        if ((rc = upload(P, pack_fragments(B.tilesA, B.kpairs, Aelem), &B.Af))) return rc;
        if ((rc = upload(P, pack_fragments(B.tilesL, B.kpairs, Lelem), &B.Lf))) return rc;
        if ((rc = upload(P, pack_fragments(B.tilesK, B.rpairsA + B.rpairsL, Telem), &B.Bk))) return rc;
        auto TAelem = [&](int k, int r) -> double { return Aelem(r, k); };
        if ((rc = upload(P, pack_fragments(B.tilesK, B.rpairsA, TAelem), &B.BkA))) return rc;
        // exact Toeplitz structure of the two halves of A (what construct_A builds from one column and one row on
        // log-uniform grids, reference matrices.py:236-242): generators for the one-chain-per-workgroup path (bdrt_solo.h)
        {
            const int glen = nf + K - 1;
            std::vector<double> tgv((size_t)2 * glen, 0.0);
            bool tz = true;
            for (int h = 0; h < 2 && tz; ++h) {
                const double *Ah = A + (size_t)h * nf * K;
                for (int n = 0; n < nf; ++n) tgv[(size_t)h * glen + n + K - 1] = Ah[(size_t)n * K];
                for (int m = 0; m < K; ++m) tgv[(size_t)h * glen + K - 1 - m] = Ah[m];
                for (int n = 0; n < nf && tz; ++n)
                    for (int m = 0; m < K; ++m)
                        if (Ah[(size_t)n * K + m] != tgv[(size_t)h * glen + n - m + K - 1]) { tz = false; break; }
            }
            B.tg = nullptr;
            if (tz && (rc = upload(P, tgv, &B.tg))) return rc;
            B.Ad = nullptr; B.At = nullptr; B.Ld = nullptr; B.Lt = nullptr;
            if (!tz) {
                std::vector<double> ad((size_t)N2 * K), at((size_t)K * N2);
                for (int r = 0; r < N2; ++r)
                    for (int m = 0; m < K; ++m) { ad[(size_t)r * K + m] = A[(size_t)r * K + m]; at[(size_t)m * N2 + r] = A[(size_t)r * K + m]; }
                if ((rc = upload(P, ad, &B.Ad)) || (rc = upload(P, at, &B.At))) return rc;
            }
        }
        // banded-Toeplitz detection of L0, L1, L2 (log-uniform tau grids): every entry outside the band
        // is below 1e-19 of the largest entry and every diagonal is constant to 1e-12 relative.  The dense MFMA path
        // remains for any other grid.
        {
            bool ok = K <= NG * UK && K > 2 * MAXBW + 2;
            for (int i = 0; i < 3 && ok; ++i) {
                const double *L = Ls[i];
                double mx = 0.0;
                for (size_t e = 0; e < (size_t)K * K; ++e) mx = std::max(mx, std::fabs(L[e]));
                for (int d = -(K - 1); d <= K - 1 && ok; ++d) {
                    double lo = INFINITY, hi = -INFINITY;
                    for (int r = std::max(0, -d); r < std::min(K, K - d); ++r) {
                        const double v = L[(size_t)r * K + r + d];
                        lo = std::min(lo, v); hi = std::max(hi, v);
                    }
                    if (std::abs(d) > MAXBW) { if (std::max(std::fabs(lo), std::fabs(hi)) > 1e-19 * mx) ok = false; }
                    else {
                        if (hi - lo > 1e-12 * mx) ok = false;
                        B.T[i][d + MAXBW] = L[(size_t)(K / 2) * K + K / 2 + d];
                    }
                }
            }
            toep_ok[b] = ok ? 1 : 0;
        }
        XR = std::max(XR, std::max(8 * B.kpairs, 16 * B.tilesK));
        LR = std::max(LR, std::max(16 * B.tilesL, 16 * B.tilesA));
    }
    D.XR = XR;
    D.ZR = 8 * rpairsA;
    D.LR = std::max(LR, MIN_LR);
    D.npar = npar;
    // cache exp(theta_x) in LDS when the workgroup's LDS budget allows it (it always does for one 81x161 block)
    {
        int rows = 0;
        // one buffer shared by the blocks (the evaluator regenerates a block's x when it needs it again)
        for (int b = 0; b < dat->nblocks; ++b) { D.xc_off[b] = 0; rows = std::max(rows, 8 * D.blk[b].kpairs + 2 * MAXBW); }
        D.XCR = rows;
        // structured path: Lr only holds A x (16*tilesA rows) and ONE halo-padded w buffer; dense path: [L0;L1;L2] x
        const int LR0 = D.LR;
        bool all = !getenv("BDRT_DENSE_L");
        int LRs = MIN_LR;
        for (int b = 0; b < dat->nblocks; ++b) {
            all = all && toep_ok[b];
            LRs = std::max(LRs, std::max(16 * D.blk[b].tilesA, ((dat->K[b] + 2 * MAXBW) + 15) / 16 * 16));
        }
        if (all) {
            // all three w buffers when they fit (saves four barriers per block and evaluation), else one
            int LR3 = LRs;
            for (int b = 0; b < dat->nblocks; ++b) LR3 = std::max(LR3, (3 * (dat->K[b] + 2 * MAXBW) + 15) / 16 * 16);
            D.LR = LR3; D.w3 = 1;
            if (lds_doubles(D) * sizeof(double) + 8192 > 160 * 1024) { D.LR = LRs; D.w3 = 0; }
            if (lds_doubles(D) * sizeof(double) + 8192 > 160 * 1024) all = false;
        }
        if (!all) {
            D.LR = LR0; D.w3 = 0;
            if (lds_doubles(D) * sizeof(double) + 8192 > 160 * 1024) { D.XCR = 0; for (int b = 0; b < MAXB; ++b) D.xc_off[b] = 0; }
        }
        for (int b = 0; b < dat->nblocks; ++b) D.blk[b].toep = all ? 1 : 0;
        D.toep_all = all ? 1 : 0;
        // headline family on a log-uniform grid: half-wave-per-chain evaluator (bdrt_tile_s1.h)
        D.fast_s1 = (all && dat->nblocks == 1 && !D.blk[0].is_parallel && !D.use_x_sum &&
                     D.blk[0].x_scale == 1.0 && nf <= 128 && D.blk[0].K <= 32 * UK &&
                     !getenv("BDRT_GENERIC_TILE")) ? 1 : 0;
        // ... and with A_re, A_im exactly Toeplitz (equal log spacing of frequencies and tau) on the shapes of the reference's
        // default grids (ten points per decade over eight decades: nf = 80..82; K = 80..82 or 160..162), which tile as whole
        // blocks of 80 plus at most two rows: GEMM operands from an LDS-resident generator table (bdrt_tile_s1.h::toep_gemm)
        if (D.fast_s1 && D.blk[0].tg && nf / 16 == 5 && nf % 16 <= 2 && (D.blk[0].K / 16) % 5 == 0 && D.blk[0].K >= 80 && D.blk[0].K % 16 <= 2 &&
            !getenv("BDRT_STREAM_A")) {
            D.toepA = 1;
            D.tlen = (8 + nf + D.blk[0].K - 1 + 8 + 1) & ~1;
            // (the sampler's theta rows, bdrt_nuts.hip::nuts_lds_bytes; with outlier parameters its state stays in HBM)
            const size_t nj = D.outlier_mode ? 0 : s1_nj(D.D);
            if ((s1_lds_doubles(D) + (size_t)NC * 32 * nj) * sizeof(double) + SAMPLER_LDS_RESERVE > 160 * 1024) { D.toepA = 0; D.tlen = 0; }
        }
        // ... and on every other shape (partial tiles, any reduction length: bdrt_tile_s1.h::toep_gemm_gen; BDRT_TOEP_GEN=1: on the
        // default shapes too, for the tests).  The imaginary rows of A x start at nf rounded up to four.
        if (D.fast_s1 && D.blk[0].tg && nf >= 32 && D.blk[0].K >= 32 && !getenv("BDRT_STREAM_A") &&
            (!D.toepA || (getenv("BDRT_TOEP_GEN") && atoi(getenv("BDRT_TOEP_GEN")) != 0)) &&
            !(getenv("BDRT_TOEP_GEN") && atoi(getenv("BDRT_TOEP_GEN")) == 0)) {
            const int toepA0 = D.toepA, tlen0 = D.tlen;
            D.toepA = 2;
            D.tlen = (16 + nf + D.blk[0].K - 1 + 16 + 1) & ~1;
            D.zrows = std::max(2 * ((nf + 3) & ~3), 16 * D.blk[0].tilesA);     // (never below what the other S1 instantiations lay out)
            const size_t nj = D.outlier_mode ? 0 : s1_nj(D.D);
            std::vector<unsigned> steps(TOEP_STEP_WORDS);
            bool ok = true;
            for (int dir = 0; dir < 2 && ok; ++dir)
                for (int w = 0; w < 8 && ok; ++w)
                    ok = toep_gen_steps(dir == 0, nf, D.blk[0].K, D.tlen, w, steps.data() + (size_t)(dir * 8 + w) * 2 * TOEP_STEPS) >= 0;
            if (!ok || (s1_lds_doubles(D) + (size_t)NC * 32 * nj) * sizeof(double) + SAMPLER_LDS_RESERVE > 160 * 1024) { D.toepA = toepA0; D.tlen = tlen0; D.zrows = 0; }
            else if (int rc = upload(P, steps, &D.tsteps)) return rc;
        }
        // every other family (several distributions, parallel blocks, x_sum prior): the general half-wave evaluator
        bool hw = all && !D.fast_s1 && nf <= 128 && !getenv("BDRT_GENERIC_TILE") &&
                  (hw_lds_doubles(D) + 64) * sizeof(double) + 4096 <= 160 * 1024;
        for (int b = 0; b < dat->nblocks; ++b) hw = hw && D.blk[b].K <= 32 * UK;
        D.fast_hw = hw ? 1 : 0;
    }
    if (getenv("BDRT_VERBOSE"))
        fprintf(stderr, "bdrt_problem_create: nf=%d blocks=%d D=%d structured_L=%d (per block %d %d %d) fast_s1=%d fast_hw=%d LDS rows X=%d Z=%d x%d L=%d XC=%d -> %zu B\n",
                nf, dat->nblocks, D.D, D.toep_all, toep_ok[0], dat->nblocks > 1 ? toep_ok[1] : -1, dat->nblocks > 2 ? toep_ok[2] : -1,
                D.fast_s1, D.fast_hw, D.XR, D.ZR, 1 + D.npar, D.LR, D.XCR, lds_doubles(D) * sizeof(double));
    if (const char *e = getenv("BDRT_DEBUG_SKIP")) D.dbg = atoi(e);
    // a half-wave evaluator whose tile does not fit while the generic tile does: the generic tile, not the streamed evaluator
    if (D.fast_s1 && s1_lds_doubles(D) * sizeof(double) + SAMPLER_LDS_RESERVE > 160 * 1024 &&
        lds_doubles(D) * sizeof(double) + SAMPLER_LDS_RESERVE <= 160 * 1024) { D.fast_s1 = 0; D.toepA = 0; D.tlen = 0; D.zrows = 0; }
    if (D.fast_hw && hw_lds_doubles(D) * sizeof(double) + SAMPLER_LDS_RESERVE > 160 * 1024 &&
        lds_doubles(D) * sizeof(double) + SAMPLER_LDS_RESERVE <= 160 * 1024) D.fast_hw = 0;
    P.lds_bytes = std::max(std::max(lds_doubles(D), D.fast_hw ? hw_lds_doubles(D) : (size_t)0), D.fast_s1 ? s1_lds_doubles(D) : (size_t)0) * sizeof(double);
    D.big = 0;
    if (P.lds_bytes + SAMPLER_LDS_RESERVE > 160 * 1024 || getenv("BDRT_BIG")) {
        // Beyond the LDS budget of the tile evaluators (the reference takes any grid: inversion.py:2127-2209): the streamed
        // evaluator of bdrt_big.h -- plain copies of the matrices and their transposes in HBM, vectors in a per-point workspace
        D.big = 1;
        D.fast_s1 = 0; D.fast_hw = 0; D.toepA = 0; D.tlen = 0; D.zrows = 0;
        P.lds_bytes = 4096;
        for (int b = 0; b < dat->nblocks; ++b) {
            DevBlock &B = D.blk[b];
            const int K = B.K;
            const double *A = dat->A[b];
            const double *Ls[3] = {dat->L0[b], dat->L1[b], dat->L2[b]};
            int rc;
            if (!B.Ad) {
                std::vector<double> ad((size_t)N2 * K), at((size_t)K * N2);
                for (int r = 0; r < N2; ++r)
                    for (int m = 0; m < K; ++m) { ad[(size_t)r * K + m] = A[(size_t)r * K + m]; at[(size_t)m * N2 + r] = A[(size_t)r * K + m]; }
                if ((rc = upload(P, ad, &B.Ad)) || (rc = upload(P, at, &B.At))) return rc;
            }
            std::vector<double> ld((size_t)3 * K * K), lt((size_t)3 * K * K);
            for (int i = 0; i < 3; ++i)
                for (int r = 0; r < K; ++r)
                    for (int m = 0; m < K; ++m) {
                        ld[((size_t)i * K + r) * K + m] = Ls[i][(size_t)r * K + m];
                        lt[(size_t)m * 3 * K + (size_t)i * K + r] = Ls[i][(size_t)r * K + m];
                    }
            if ((rc = upload(P, ld, &B.Ld)) || (rc = upload(P, lt, &B.Lt))) return rc;
        }
    }
    std::vector<double> w(nf);
    for (int n = 0; n < nf; ++n) w[n] = 2.0 * M_PI * dat->freq[n];
    int rc;
    if ((rc = upload(P, w, &D.w))) return rc;
    BDRT_HIP(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
    D.n_spectra = 0;
    return 0;
}

static int set_Z(Problem &P, const double *Z, int n_spectra)
{
    const size_t need = (size_t)n_spectra * 2 * P.dev.nf;
    if (need > P.z_capacity) {
        if (P.d_Z) hipFree(P.d_Z);
        BDRT_HIP(hipMalloc((void **)&P.d_Z, need * sizeof(double)));
        P.z_capacity = need;
    }
    BDRT_HIP(hipMemcpy(P.d_Z, Z, need * sizeof(double), hipMemcpyHostToDevice));
    P.dev.Z = P.d_Z;
    P.dev.n_spectra = n_spectra;
    return P.sync_dev();
}

// MODE 0: dense L path, 1: structured L path (generic tile), 2: fast S1 tile, 3: general half-wave tile, 4: S1 tile with the A operands from the LDS table, 6: the same on any shape (toepA == 2)
template <int MODE, int KU = 6>
__global__ __launch_bounds__(NT) void logp_grad_kernel(const DevProblem *__restrict__ Pp, const double *theta, const int *spec, int B,
                                                       int jacobian, double *lp, double *grad, double *params,
                                                       double *Zhat, double *sig)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    // A workgroup walks the tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the launcher asks for a few workgroups per CU at most):
    // the generator table and the problem's scalars are set up once per workgroup, not once per sixteen points.
    if (MODE == 4 || MODE == 6) s1_toep_init(P, smem);
    for (int c0 = blockIdx.x * NC; c0 < B; c0 += gridDim.x * NC) {
        TileIO io;
        io.theta = theta + (size_t)c0 * P.D;
        io.t_sc = P.D; io.t_sj = 1; io.t_off = nullptr;
        io.grad = grad ? grad + (size_t)c0 * P.D : nullptr;
        io.g_sc = P.D; io.g_sj = 1;
        io.lp = lp ? lp + c0 : nullptr;
        io.spec = spec ? spec + c0 : nullptr;
        io.nvalid = min(NC, B - c0);
        io.jacobian = jacobian;
        io.Z_hat = Zhat ? Zhat + (size_t)c0 * 2 * P.nf : nullptr;
        io.sigma_tot = sig ? sig + (size_t)c0 * 2 * P.nf : nullptr;
        io.params = params ? params + (size_t)c0 * P.D : nullptr;
        io.prof = nullptr;
        if (MODE == 4) logp_grad_tile_s1<false, 32, NoHook, NoHook, 1, KU>(P, io, smem);
        else if (MODE == 6) logp_grad_tile_s1<false, 32, NoHook, NoHook, 2, KU>(P, io, smem);
        else if (MODE == 3) logp_grad_tile_hw<KU>(P, io, smem);
        else if (MODE == 2) logp_grad_tile_s1<false, 32, NoHook, NoHook, 0, KU>(P, io, smem);
        else if (MODE == 1) logp_grad_tile<true>(P, io, smem);
        else logp_grad_tile<false>(P, io, smem);
    }
}

// the S1 evaluator with a whole wavefront per chain: 1024 threads = 16 waves = 4 per SIMD (<= 128 VGPRs)
__global__ __launch_bounds__(1024) void logp_grad_kernel_wide(const DevProblem *__restrict__ Pp, const double *theta, const int *spec,
                                                              int B, int jacobian, double *lp, double *grad, double *params,
                                                              double *Zhat, double *sig)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int c0 = blockIdx.x * NC;
    TileIO io;
    io.theta = theta + (size_t)c0 * P.D;
    io.t_sc = P.D; io.t_sj = 1; io.t_off = nullptr;
    io.grad = grad ? grad + (size_t)c0 * P.D : nullptr;
    io.g_sc = P.D; io.g_sj = 1;
    io.lp = lp ? lp + c0 : nullptr;
    io.spec = spec ? spec + c0 : nullptr;
    io.nvalid = min(NC, B - c0);
    io.jacobian = jacobian;
    io.Z_hat = Zhat ? Zhat + (size_t)c0 * 2 * P.nf : nullptr;
    io.sigma_tot = sig ? sig + (size_t)c0 * 2 * P.nf : nullptr;
    io.params = params ? params + (size_t)c0 * P.D : nullptr;
    io.prof = nullptr;
    logp_grad_tile_s1<false, 64>(P, io, smem);
}

int launch_logp_grad(Problem *p, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                     double *d_grad, double *d_params, double *d_Zhat, double *d_sig, hipStream_t stream)
{
    if (B <= 0) return 0;
    BDRT_HIP(hipSetDevice(p->device));
    if (p->dev.big) {
        const int grid = std::min(B, 1024);
        std::lock_guard<std::mutex> lock(p->bigws_mu);
        // the workspace is the problem's, not the stream's: this launch starts when the previous one (whatever its stream) is done
        if (!p->bigws_done) BDRT_HIP(hipEventCreateWithFlags(&p->bigws_done, hipEventDisableTiming));
        else BDRT_HIP(hipStreamWaitEvent(stream, p->bigws_done, 0));
        if (int rc = p->ensure_bigws((size_t)grid * big_ws_doubles(p->dev))) return rc;
        hipLaunchKernelGGL(big_eval_kernel, dim3(grid), dim3(BIG_NT), 0, stream, (const DevProblem *)p->d_dev, p->d_bigws, d_theta, d_spec, B,
                           jacobian, d_lp, d_grad, d_params, d_Zhat, d_sig);
        BDRT_HIP(hipGetLastError());
        BDRT_HIP(hipEventRecord(p->bigws_done, stream));
        return 0;
    }
    static LdsAttrCache attr_cache;
    BDRT_HIP(attr_cache.ensure(p->lds_bytes, [&]() {
        const void *fns[17] = {(const void *)logp_grad_kernel<0>, (const void *)logp_grad_kernel<1>, (const void *)logp_grad_kernel<2>,
                               (const void *)logp_grad_kernel_wide, (const void *)logp_grad_kernel<3>, (const void *)logp_grad_kernel<4>,
                               (const void *)logp_grad_kernel<6>, (const void *)logp_grad_kernel<2, 2>, (const void *)logp_grad_kernel<2, 3>,
                               (const void *)logp_grad_kernel<2, 4>, (const void *)logp_grad_kernel<4, 3>, (const void *)logp_grad_kernel<6, 2>,
                               (const void *)logp_grad_kernel<6, 3>, (const void *)logp_grad_kernel<6, 4>, (const void *)logp_grad_kernel<3, 2>,
                               (const void *)logp_grad_kernel<3, 3>, (const void *)logp_grad_kernel<3, 4>};
        hipError_t e = hipSuccess;
        for (int i = 0; i < 17 && e == hipSuccess; ++i)
            e = hipFuncSetAttribute(fns[i], hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes);
        return e;
    }));
    // (BDRT_EVAL_WGS_PER_CU: workgroups per CU a large batch is spread over, default 4; 0: one workgroup per tile as before)
    static const int wgs_per_cu = getenv("BDRT_EVAL_WGS_PER_CU") ? atoi(getenv("BDRT_EVAL_WGS_PER_CU")) : 4;
    static int cu_of[64] = {0};                                   // (per device, asked once)
    int &n_cu = cu_of[p->device & 63];
    if (n_cu == 0) {
        hipDeviceProp_t prop;
        n_cu = (hipGetDeviceProperties(&prop, p->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    static const bool wide = getenv("BDRT_S1_WIDE") != nullptr;
    const int grid = (wide || wgs_per_cu <= 0) ? cdiv(B, NC) : std::min(cdiv(B, NC), wgs_per_cu * n_cu);
    if (p->dev.fast_s1 && wide && p->dev.nf <= 128)
        hipLaunchKernelGGL(logp_grad_kernel_wide, dim3(grid), dim3(1024), p->lds_bytes, stream, (const DevProblem *)p->d_dev,
                           d_theta, d_spec, B, jacobian, d_lp, d_grad, d_params, d_Zhat, d_sig);
    else if (p->dev.fast_hw) {
        static const bool ku6 = getenv("BDRT_S1_KU") && atoi(getenv("BDRT_S1_KU")) == 6;
        const int ku = ku6 ? 6 : s1_ku(hw_kmax(p->dev));
#define BDRT_HW_LAUNCH(KU_)                                                                                                             \
        hipLaunchKernelGGL((logp_grad_kernel<3, KU_>), dim3(grid), dim3(NT), p->lds_bytes, stream, (const DevProblem *)p->d_dev, d_theta, \
                           d_spec, B, jacobian, d_lp, d_grad, d_params, d_Zhat, d_sig)
        if (ku == 2) BDRT_HW_LAUNCH(2); else if (ku == 3) BDRT_HW_LAUNCH(3); else if (ku == 4) BDRT_HW_LAUNCH(4); else BDRT_HW_LAUNCH(6);
#undef BDRT_HW_LAUNCH
    }
    else if (p->dev.fast_s1) {
        // the instantiation by the basis length (BDRT_S1_KU=6: the one for K <= 192 whatever K is)
        static const bool ku6 = getenv("BDRT_S1_KU") && atoi(getenv("BDRT_S1_KU")) == 6;
        const int ku = ku6 ? 6 : s1_ku(p->dev.blk[0].K), ta = p->dev.toepA;
#define BDRT_S1_LAUNCH(MODE_, KU_)                                                                                                         \
        hipLaunchKernelGGL((logp_grad_kernel<MODE_, KU_>), dim3(grid), dim3(NT), p->lds_bytes, stream, (const DevProblem *)p->d_dev, d_theta, \
                           d_spec, B, jacobian, d_lp, d_grad, d_params, d_Zhat, d_sig)
        if (ta == 2) { if (ku == 2) BDRT_S1_LAUNCH(6, 2); else if (ku == 3) BDRT_S1_LAUNCH(6, 3); else if (ku == 4) BDRT_S1_LAUNCH(6, 4); else BDRT_S1_LAUNCH(6, 6); }
        else if (ta == 1) { if (ku == 3) BDRT_S1_LAUNCH(4, 3); else BDRT_S1_LAUNCH(4, 6); }
        else { if (ku == 2) BDRT_S1_LAUNCH(2, 2); else if (ku == 3) BDRT_S1_LAUNCH(2, 3); else if (ku == 4) BDRT_S1_LAUNCH(2, 4); else BDRT_S1_LAUNCH(2, 6); }
#undef BDRT_S1_LAUNCH
    }
    else if (p->dev.toep_all)
        hipLaunchKernelGGL(logp_grad_kernel<1>, dim3(grid), dim3(NT), p->lds_bytes, stream, (const DevProblem *)p->d_dev,
                           d_theta, d_spec, B, jacobian, d_lp, d_grad, d_params, d_Zhat, d_sig);
    else
        hipLaunchKernelGGL(logp_grad_kernel<0>, dim3(grid), dim3(NT), p->lds_bytes, stream, (const DevProblem *)p->d_dev,
                           d_theta, d_spec, B, jacobian, d_lp, d_grad, d_params, d_Zhat, d_sig);
    BDRT_HIP(hipGetLastError());
    return 0;
}

// debug / tests only: the lean elementary functions of bdrt_device.h on n values (out: [3][n] = exp, log, reciprocal)
__global__ void lean_math_kernel(const double *x, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = lean_exp(x[i]); out[n + i] = lean_log(x[i]); out[2 * (size_t)n + i] = lean_rcp(x[i]);
}

// debug / tests only: sum32_by_lane<8> and <4> on one wavefront: in [64][8], out [2][64] (lane l of each half: the total of quantity l & 7 / l & 3)
__global__ void sum_by_lane_kernel(const double *in, double *out)
{
    const int l = threadIdx.x;
    double q8[8], q4[4];
    for (int j = 0; j < 8; ++j) q8[j] = in[l * 8 + j];
    for (int j = 0; j < 4; ++j) q4[j] = in[l * 8 + j];
    out[l] = sum32_by_lane<8>(q8, l & 31);
    out[64 + l] = sum32_by_lane<4>(q4, l & 31);
}

}  // namespace bdrt

using namespace bdrt;

extern "C" {

const char *bdrt_last_error(void) { return g_last_error.c_str(); }
const char *bdrt_version(void) { return "bdrt-mi355x 0.1 (gfx950)"; }

int bdrt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int bdrt_set_device(int dev)
{
    BDRT_HIP(hipSetDevice(dev));
    g_process_device.store(dev);
    return 0;
}

bdrt_problem *bdrt_problem_create(const bdrt_dat *dat)
{
    bdrt_problem *p = new bdrt_problem();
    int rc = build_problem(p->impl, dat);
    if (rc == 0) rc = set_Z(p->impl, dat->Z, dat->n_spectra);
    if (rc != 0) {
        bdrt_problem_destroy(p);
        return nullptr;
    }
    return p;
}

void bdrt_problem_destroy(bdrt_problem *p)
{
    if (!p) return;
    Problem &P = p->impl;
    for (void *d : P.allocs) hipFree(d);
    if (P.d_Z) hipFree(P.d_Z);
    if (P.d_dev) hipFree(P.d_dev);
    if (P.d_theta) { hipFree(P.d_theta); hipFree(P.d_grad); hipFree(P.d_lp); hipFree(P.d_spec); }
    if (P.d_bigws) hipFree(P.d_bigws);
    if (P.bigws_done) hipEventDestroy(P.bigws_done);
    if (P.stream) hipStreamDestroy(P.stream);
    delete p;
}

int bdrt_num_params(const bdrt_problem *p) { return p ? p->impl.dev.D : -1; }
int bdrt_problem_evaluator(const bdrt_problem *p)
{
    if (!p) return -1;
    const bdrt::DevProblem &D = p->impl.dev;
    return D.big ? 5 : (D.fast_hw ? 3 : (D.fast_s1 ? (D.toepA ? 4 : 2) : (D.toep_all ? 1 : 0)));
}

int bdrt_param_is_pos(const bdrt_problem *p, unsigned char *is_pos)
{
    if (!p || !is_pos) return -1;
    memcpy(is_pos, p->impl.is_pos.data(), p->impl.is_pos.size());
    return 0;
}

int bdrt_problem_set_Z(bdrt_problem *p, const double *Z, int n_spectra)
{
    if (!p || !Z || n_spectra < 1) { set_error("bdrt_problem_set_Z: bad arguments"); return -1; }
    BDRT_HIP(hipStreamSynchronize(p->impl.stream));
    return set_Z(p->impl, Z, n_spectra);
}

static int check_spec(Problem &P, const int *spec, int B)
{
    if (!spec) return 0;
    for (int i = 0; i < B; ++i)
        if (spec[i] < 0 || spec[i] >= P.dev.n_spectra) {
            set_error("spectrum index %d out of range [0,%d) at row %d", spec[i], P.dev.n_spectra, i);
            return -1;
        }
    return 0;
}

// debug only: device buffer [workgroups][8 waves][16] that the S1 tile of the logp kernel fills with clock64() stamps
int bdrt_debug_set_tile_trace(void *d_buf)
{
    BDRT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), &d_buf, sizeof(void *)));
    return 0;
}

int bdrt_debug_sum_by_lane(const double *in, double *out)
{
    if (!in || !out) { set_error("bdrt_debug_sum_by_lane: bad arguments"); return -1; }
    bind_process_device();
    double *di = nullptr, *dout = nullptr;
    BDRT_HIP(hipMalloc((void **)&di, 512 * sizeof(double)));
    BDRT_HIP(hipMalloc((void **)&dout, 128 * sizeof(double)));
    BDRT_HIP(hipMemcpy(di, in, 512 * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sum_by_lane_kernel, dim3(1), dim3(64), 0, 0, di, dout);
    BDRT_HIP(hipMemcpy(out, dout, 128 * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(di); hipFree(dout);
    return 0;
}

int bdrt_debug_lean_math(const double *x, int n, double *out)
{
    if (!x || !out || n < 1) { set_error("bdrt_debug_lean_math: bad arguments"); return -1; }
    bind_process_device();
    double *dx = nullptr, *dout = nullptr;
    BDRT_HIP(hipMalloc((void **)&dx, (size_t)n * sizeof(double)));
    BDRT_HIP(hipMalloc((void **)&dout, (size_t)3 * n * sizeof(double)));
    BDRT_HIP(hipMemcpy(dx, x, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(lean_math_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, n, dout);
    BDRT_HIP(hipMemcpy(out, dout, (size_t)3 * n * sizeof(double), hipMemcpyDeviceToHost));
    hipFree(dx); hipFree(dout);
    return 0;
}

int bdrt_logp_grad_dev(bdrt_problem *p, const double *d_theta, const int *d_spec, int B, int jacobian, double *d_lp,
                       double *d_grad, void *stream)
{
    if (!p || !d_theta) { set_error("bdrt_logp_grad_dev: null argument"); return -1; }
    hipStream_t st = stream ? (hipStream_t)stream : p->impl.stream;
    // a handful of points: one workgroup per point (a 16-column tile with one live column costs the same 24 us as a full one)
    const int few = launch_logp_grad_few(&p->impl, d_theta, d_spec, B, jacobian, d_lp, d_grad, st);
    if (few <= 0) return few;
    return launch_logp_grad(&p->impl, d_theta, d_spec, B, jacobian, d_lp, d_grad, nullptr, nullptr, nullptr, st);
}

int bdrt_logp_grad(bdrt_problem *p, const double *theta, const int *spec, int B, int jacobian, double *lp, double *grad)
{
    if (!p || !theta || B < 0) { set_error("bdrt_logp_grad: bad arguments"); return -1; }
    if (B == 0) return 0;
    Problem &P = p->impl;
    int rc;
    if ((rc = check_spec(P, spec, B))) return rc;
    if ((rc = P.ensure_scratch((size_t)B))) return rc;
    const size_t nb = (size_t)B * P.dev.D * sizeof(double);
    BDRT_HIP(hipMemcpyAsync(P.d_theta, theta, nb, hipMemcpyHostToDevice, P.stream));
    if (spec) BDRT_HIP(hipMemcpyAsync(P.d_spec, spec, (size_t)B * sizeof(int), hipMemcpyHostToDevice, P.stream));
    rc = launch_logp_grad_few(&P, P.d_theta, spec ? P.d_spec : nullptr, B, jacobian, P.d_lp, grad ? P.d_grad : nullptr, P.stream);
    if (rc < 0) return rc;
    if (rc > 0 && (rc = launch_logp_grad(&P, P.d_theta, spec ? P.d_spec : nullptr, B, jacobian, P.d_lp, grad ? P.d_grad : nullptr,
                                         nullptr, nullptr, nullptr, P.stream)))
        return rc;
    if (lp) BDRT_HIP(hipMemcpyAsync(lp, P.d_lp, (size_t)B * sizeof(double), hipMemcpyDeviceToHost, P.stream));
    if (grad) BDRT_HIP(hipMemcpyAsync(grad, P.d_grad, nb, hipMemcpyDeviceToHost, P.stream));
    BDRT_HIP(hipStreamSynchronize(P.stream));
    return 0;
}

int bdrt_transformed(bdrt_problem *p, const double *theta, const int *spec, int B, double *params, double *Z_hat,
                     double *sigma_tot)
{
    if (!p || !theta || B < 0) { set_error("bdrt_transformed: bad arguments"); return -1; }
    if (B == 0) return 0;
    Problem &P = p->impl;
    int rc;
    if ((rc = check_spec(P, spec, B))) return rc;
    if ((rc = P.ensure_scratch((size_t)B))) return rc;
    const size_t nb = (size_t)B * P.dev.D * sizeof(double), nz = (size_t)B * 2 * P.dev.nf * sizeof(double);
    double *d_par = nullptr, *d_zh = nullptr, *d_sg = nullptr;
    BDRT_HIP(hipMalloc((void **)&d_par, nb));
    BDRT_HIP(hipMalloc((void **)&d_zh, nz));
    BDRT_HIP(hipMalloc((void **)&d_sg, nz));
    BDRT_HIP(hipMemcpyAsync(P.d_theta, theta, nb, hipMemcpyHostToDevice, P.stream));
    if (spec) BDRT_HIP(hipMemcpyAsync(P.d_spec, spec, (size_t)B * sizeof(int), hipMemcpyHostToDevice, P.stream));
    rc = launch_logp_grad(&P, P.d_theta, spec ? P.d_spec : nullptr, B, 0, P.d_lp, nullptr, d_par, d_zh, d_sg, P.stream);
    if (rc == 0) {
        if (params) hipMemcpyAsync(params, d_par, nb, hipMemcpyDeviceToHost, P.stream);
        if (Z_hat) hipMemcpyAsync(Z_hat, d_zh, nz, hipMemcpyDeviceToHost, P.stream);
        if (sigma_tot) hipMemcpyAsync(sigma_tot, d_sg, nz, hipMemcpyDeviceToHost, P.stream);
        if (hipStreamSynchronize(P.stream) != hipSuccess) { set_error("bdrt_transformed: sync failed"); rc = -10; }
    }
    hipFree(d_par); hipFree(d_zh); hipFree(d_sg);
    return rc;
}

}  // extern "C"
