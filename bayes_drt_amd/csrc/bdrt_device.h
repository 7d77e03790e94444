// bdrt_device.h -- device-side model description and the tile evaluator of the log-posterior + gradient.
//
// One workgroup (512 threads = 8 wave64) evaluates NC = 16 points ("chains") at once so that every dense
// contraction of the Stan model is an fp64 MFMA GEMM with N = 16:
//     forward   Y[R x 16]  = M[R x K] . X[K x 16]        (A x, L0 x, L1 x, L2 x)
//     backward  G[K x 16]  = M^T[K x R] . Rm[R x 16]     (A^T g_Zhat - sum_i d_i L_i^T (v_i / ups^2))
// with v_mfma_f64_16x16x4_f64.  The matrices are pre-packed into MFMA fragment order (bdrt_model.hip) so a
// wave's operand load is one contiguous 1 KiB global_load_dwordx4; they stay L2-resident (0.83 MB for the
// 81 x 161 benchmark shape).  X / Y / Rm live in LDS as [row][16 chains] so that every MFMA B-operand read is a
// conflict-free ds_read_b64 and the element-wise model code in between (likelihood, priors, chain rule) is a
// (row, chain) map over the workgroup with wave-shuffle + LDS reductions for the per-chain scalars.
//
// Model being evaluated: bayes_drt/stan_model_files/*_modelcode.txt (reference), all families written as
// "blocks" (see include/bdrt.h).  Gradient: hand-derived reverse pass.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bdrt {

constexpr int NC = 16;        // chains per workgroup = MFMA N
// LDS the 16-chain sampler kernel keeps beside the evaluator's tile region (chain states, lp, spectrum ids ...): a problem is
// accepted only if tile + this fits 160 KiB, so that whatever can be evaluated and optimised can also be sampled
constexpr size_t SAMPLER_LDS_RESERVE = 6144;
constexpr int NT = 512;       // threads per workgroup
constexpr int NW = NT / 64;   // waves per workgroup
constexpr int NG = NT / NC;   // row groups in the element-wise phases
constexpr int MAXB = 3;
constexpr int NRED = 8;       // max quantities per block reduction
constexpr int MAXBW = 6;      // half bandwidth of the structured (banded Toeplitz) L path: at the reference's default shape parameter (epsilon = 1 / mean spacing of ln tau) the diagonals beyond +-6 are below 1e-19 of the largest entry (e^-49); wider bands (smaller epsilon) take the dense MFMA path
constexpr int MIN_LR = 192;    // the NUTS kernel borrows the Lr buffer (>= 8*22*16 doubles) for its reductions
constexpr double LOG_015 = -1.8971199848858813;   // log(0.15): ups = 0.15*ups_raw

struct DevBlock {
    int K;            // basis functions
    int is_parallel, is_pos;
    int tilesA;       // ceil(2nf/16)
    int tilesL;       // ceil(3K/16)
    int tilesK;       // ceil(K/16)
    int kpairs;       // ceil(K/8): forward reduction in pairs of k-steps (8 columns per pair)
    int rpairsA;      // ceil(2nf/8)
    int rpairsL;      // ceil(3K/8)
    int o_x, o_ups, o_d;
    int yp_slot;      // index of the Y_hat buffer for parallel blocks
    double x_scale;
    const double *Af; // [tilesA][kpairs][64][2]   A rows x K cols, fragment order
    const double *Lf; // [tilesL][kpairs][64][2]   (L0;L1;L2) rows x K cols
    const double *Bk; // [tilesK][rpairsA+rpairsL][64][2]  transposed: K rows x (2nf pad | 3K pad) reduction
    // structured path: when L0, L1, L2 are banded Toeplitz (log-uniform tau grid, the
    // reference's recommended basis) the three L products are 2*MAXBW+1-tap convolutions on the VALU and only A uses MFMA
    int toep;         // 1: use T[][] instead of Lf / the L part of Bk
    const double *BkA;// [tilesK][rpairsA][64][2]  transposed A only
    double T[3][2 * MAXBW + 1];   // T[i][d + MAXBW] = L_i[k][k + d]
    // A_re and A_im exactly Toeplitz (log-uniform frequency and tau grids, reference matrices.py:197-205): their generators
    // [2][nf + K - 1], tg[b][n - m + K - 1] = A_b[n][m]; nullptr otherwise.  Used by the one-chain-per-workgroup path.
    const double *tg;
    // otherwise (measured spectra on their own frequency grid): plain copies of A for the one-chain-per-workgroup path,
    // Ad [2 nf][K] row-major (A^T r reads four consecutive columns) and At [K][2 nf] (A x reads four consecutive rows)
    const double *Ad, *At;
    // problems beyond the LDS budget (bdrt_big.h; Ad / At are kept for them too): the stack [L0; L1; L2] row-major [3K][K]
    // and its transpose [K][3K]
    const double *Ld, *Lt;
};

struct DevProblem {
    int nf, nblocks, D;
    int o_err, o_so;
    int outlier_mode, use_x_sum;
    int n_spectra;
    int XR, ZR, LR, npar;     // LDS row counts
    int toep_all;             // every block takes the structured (banded Toeplitz) L path
    int w3;                   // structured path: Lr holds all three w_i buffers (else one buffer, i after i: +4 barriers per block)
    int fast_s1;              // single series block, structured: bdrt_tile_s1.h evaluates it
    int fast_hw;              // any other family on log-uniform grids that fits: bdrt_tile_hw.h evaluates it
    int toepA;                // fast_s1 and A_re, A_im exactly Toeplitz with nf % 16 <= 2, K % 16 <= 2: the S1 tile's two GEMMs take their
                              // A operands from a [2][tlen] table in LDS instead of streaming packed fragments from L2 (bdrt_tile_s1.h)
                              // 2: the same for any shape with nf, K >= 32 (toep_gemm_gen: partial tiles, any reduction length)
    int tlen;                 // length of one part of that table: 8 (toepA == 2: 16) leading zeros, the nf + K - 1 generators, trailing zeros
    const unsigned *tsteps;   // toepA == 2: the waves' step lists through the two GEMMs, [2][8][TOEP_STEPS][2] (bdrt_tile_s1.h::toep_gen_steps)
    int zrows;                // toepA == 2: rows of the S1 tile's A x region (the imaginary rows start at nf rounded up to four; else 16 tilesA)
    int XCR;                  // rows of the x cache (0: exp(theta_x) is recomputed where needed)
    int xc_off[MAXB];         // first cache row of each block
    int big;                  // beyond the LDS budget of every tile / one-chain evaluator: bdrt_big.h evaluates it (workspace in HBM)
    int dbg;                  // timing ablation only (env BDRT_DEBUG_SKIP): 1 skip forward GEMMs, 2 skip backward GEMM
    double sigma_min, ups_alpha, ups_beta, induc_scale;
    double so_lambda, so_alpha, so_beta, x_sum_invscale;
    const double *Z;          // [n_spectra][2nf]
    const double *w;          // [nf] 2*pi*f
    DevBlock blk[MAXB];
};

__host__ __device__ inline size_t lds_doubles(const DevProblem &P)
{
    return (size_t)NC * ((size_t)P.XR + (size_t)P.ZR * (1 + P.npar) + (size_t)P.LR + (size_t)P.XCR + NW * NRED + 32);
}

typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ---- lean elementary functions for the evaluators' inner loops -----------------------------------------------------------------------
// On this machine every VALU instruction of the sampler's round is issue time on the SIMD's one fp64 pipe, and
// the device library's exp / log / IEEE division spend a third of theirs on range and special-case handling that the evaluator's
// arguments never need.  Accuracy (tests/test_gpu_lean_math.py, against numpy on 1e6 arguments per function): <= 1.5 ulp.
// Arguments outside the stated domains give a non-finite or inaccurate value, never a trap; the evaluators' callers treat a
// non-finite log-posterior the same way whatever produced it.

// 1 / b for finite, normal b != 0 (|b| in [1e-300, 1e300]): hardware estimate + two Newton steps; 5 instructions against the 11
// of the IEEE-exact quotient.  b = 0, inf: NaN (the quotient: inf, 0).
__device__ __forceinline__ double lean_rcp(double b)
{
    double r = __builtin_amdgcn_rcp(b);
    double e = __builtin_fma(-b, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-b, r, 1.0);
    return __builtin_fma(r, e, r);
}

// exp(x) for finite x: n = rint(x / ln 2), r = x - n ln 2 in two steps (Cody-Waite), e^r by the degree-13 Taylor polynomial
// (|r| <= 0.3466: truncation 6e-18 relative), 2^n by v_ldexp_f64, which overflows to inf and underflows towards 0 by itself.
// 19 instructions against 28.  x = +-inf: NaN.
// a * b + C with the constant C in a scalar register pair.  Left to itself the compiler writes a Horner step as `v_fmac_f64` (the
// addend is the destination) and materialises every coefficient in a VGPR pair first -- two `v_mov_b32` per step, ~25 VALU
// instructions per exponential on top of its 19 (tools/isa_blocks.py: 1.7 k static `v_mov` in the headline kernel); as the third
// operand of `v_fma_f64` the coefficient comes from two `s_mov_b32`, which issue beside the other wave's VALU instructions.
__device__ __forceinline__ double fma_sc(double a, double b, double C)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double d;
    __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(C));
    return d;
#else
    return __builtin_fma(a, b, C);
#endif
}

__device__ __forceinline__ double lean_exp(double x)
{
    const double n = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(-n, 6.93147180369123816490e-01, x);
    r = __builtin_fma(-n, 1.90821492927058770002e-10, r);
    double p = __builtin_fma(r, 1.6059043836821614599e-10, 2.0876756987868098979e-09);     // 1/13!, 1/12!
    p = fma_sc(p, r, 2.5052108385441718775e-08);   // 1/11!
    p = fma_sc(p, r, 2.7557319223985890653e-07);   // 1/10!
    p = fma_sc(p, r, 2.7557319223985892511e-06);   // 1/9!
    p = fma_sc(p, r, 2.4801587301587301566e-05);   // 1/8!
    p = fma_sc(p, r, 1.9841269841269841253e-04);   // 1/7!
    p = fma_sc(p, r, 1.3888888888888889419e-03);   // 1/6!
    p = fma_sc(p, r, 8.3333333333333332177e-03);   // 1/5!
    p = fma_sc(p, r, 4.1666666666666664354e-02);   // 1/4!
    p = fma_sc(p, r, 1.6666666666666665741e-01);   // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)n);
}

// log(x) for finite, normal x > 0: x = 2^e m with m in [sqrt(1/2), sqrt(2)), f = m - 1, s = f / (2 + f),
// log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the classic degree-7 polynomial in s^2 (fdlibm's e_log.c coefficients, error
// of the approximation < 2^-58.45), log x = e ln2_hi + (log(1 + f) + e ln2_lo).  ~35 instructions against 58.
// Anything else -- zero, a denormal, a negative number, inf, NaN -- gives NaN (one v_cmp_class and a select): a point whose
// sigma^2 product has left the normal range is then rejected by the logarithm itself, not by what a neighbouring
// reciprocal happens to return.
__device__ __forceinline__ double lean_log(double x)
{
    const bool in_domain = __builtin_amdgcn_class(x, 0x100);       // positive normal
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752440;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f * lean_rcp(2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * fma_sc(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * fma_sc(w, fma_sc(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01),
                                           2.857142874366239149e-01), 6.666666666666735130e-01);
    const double R = t1 + t2, hfsq = 0.5 * f * f, de = (double)e;
    const double r = __builtin_fma(de, 6.93147180369123816490e-01, f - (hfsq - __builtin_fma(s, hfsq + R, de * 1.90821492927058770002e-10)));
    return in_domain ? r : __builtin_nan("");
}

// Sum over the 32 lanes of a half-wave, result in every lane.  Four of the five butterfly steps are DPP moves on the VALU
// (quad xor 1, quad xor 2, row_half_mirror, row_mirror: any pairing of equal partial sums will do), only the step across
// the two 16-lane rows needs the LDS crossbar (ds_swizzle, no address register).  The plain __shfl_xor butterfly is five
// ds_bpermute round trips (~150 cycles each) on the critical path of every per-chain reduction.
template <int CTRL>
__device__ __forceinline__ double dpp_perm(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swizzle_xor16(double x)
{
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(x), 0x401F);      // bit mode: and 0x1F, or 0, xor 0x10
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(x), 0x401F);
    return __hiloint2double(hi, lo);
}
// the value of lane (l ^ 4) of the same row of 16: two DPP moves per half (shift left by 4 into the banks whose lanes have bit 2
// clear, shift right into the others)
__device__ __forceinline__ double dpp_xor4(double x)
{
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x104, 0xF, 0x5, false);       // row_shl:4 -> banks 0, 2
    lo = __builtin_amdgcn_update_dpp(lo, __double2loint(x), 0x114, 0xF, 0xA, false);          // row_shr:4 -> banks 1, 3
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x104, 0xF, 0x5, false);
    hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(x), 0x114, 0xF, 0xA, false);
    return __hiloint2double(hi, lo);
}

// Sums of N = 4 or 8 quantities over the 32 lanes of a half-wave in ONE butterfly: at the step across lane bit b every lane
// keeps the half of its quantities whose index has bit b equal to its own lane bit and receives the partner's partial sums
// of those -- 4 + 2 + 1 exchanges for eight quantities instead of 8 x 5.  Lane l ends with the total of quantity (l & (N - 1)),
// the same value in all lanes that share it.  (57 instructions for eight sums against 8 x 17.)
template <int N>
__device__ __forceinline__ double sum32_by_lane(const double (&v)[N], int lane)
{
    static_assert(N == 4 || N == 8, "four or eight quantities");
    const bool b0 = lane & 1, b1 = lane & 2, b2 = lane & 4;
    double a[N / 2];
#pragma unroll
    for (int i = 0; i < N / 2; ++i) a[i] = (b0 ? v[2 * i + 1] : v[2 * i]) + dpp_perm<0xB1>(b0 ? v[2 * i] : v[2 * i + 1]);      // index bit 0 = lane bit 0
    double b[N / 4];
#pragma unroll
    for (int i = 0; i < N / 4; ++i) b[i] = (b1 ? a[2 * i + 1] : a[2 * i]) + dpp_perm<0x4E>(b1 ? a[2 * i] : a[2 * i + 1]);      // index bit 1 = lane bit 1
    double c;
    if (N == 8) c = (b2 ? b[N / 4 - 1] : b[0]) + dpp_xor4(b2 ? b[0] : b[N / 4 - 1]);                                           // index bit 2 = lane bit 2
    else c = b[0] + dpp_xor4(b[0]);
    c += dpp_perm<0x128>(c);       // row_ror:8 = lane ^ 8
    c += swizzle_xor16(c);
    return c;
}

__device__ __forceinline__ double sum32(double x)
{
    x += dpp_perm<0xB1>(x);        // quad_perm [1,0,3,2]
    x += dpp_perm<0x4E>(x);        // quad_perm [2,3,0,1]
    x += dpp_perm<0x141>(x);       // row_half_mirror
    x += dpp_perm<0x140>(x);       // row_mirror
    x += swizzle_xor16(x);
    return x;
}

// Y[(16 t + i)][c] = sum_k M[16 t + i][k] X[k][c] for the tiles of this wave.
// Fragment order (host packing in bdrt_model.hip::pack_forward): element (tile t, pair p, lane l, h) =
// M[16 t + (l & 15)][8 p + 4 h + (l >> 4)].
constexpr int PF = 7;   // operand pairs per chunk (two register buffers of PF fragments)

// acc += sum over `pairs` operand pairs of M-fragments (global/L2, stride 64 double2 per pair) times LDS rows.
// Chunks of PF pairs, two register buffers.  Per chunk: all 2*PF LDS B-operands first, the first MFMA pair, THEN the
// global loads of the next chunk, then the other MFMAs back to back (sched_barriers pin that order).  Left to itself the
// compiler emitted `ds_read -> s_waitcnt lgkmcnt(0) -> v_mfma` per instruction (~180 cycles per MFMA instead of 64) and
// waited with vmcnt(0) for the prefetch it had just issued (measured with tools/tile_trace.py on the S1 evaluator, which
// uses the same scheme).  The last chunk may be partial: its loads are clamped, its MFMAs guarded (uniform branches).
__device__ __forceinline__ void mfma_stream(const double2 *__restrict__ mp_, int pairs, const double *xb, d4 &acc0,
                                            d4 &acc1)
{
    typedef double dv2 __attribute__((ext_vector_type(2)));
    // global (not flat) loads: a flat load also counts on lgkmcnt and would serialise with the LDS operand reads
    typedef const __attribute__((address_space(1))) dv2 *gp2;
    gp2 mp = (gp2) reinterpret_cast<const dv2 *>(mp_);
    const int nchunk = (pairs + PF - 1) / PF;
    auto load = [&](dv2 (&buf)[PF], int ch) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            int p = ch * PF + i;
            p = p < pairs ? p : pairs - 1;
            buf[i] = mp[(size_t)p * 64];
        }
    };
    auto step = [&](const dv2 (&cur)[PF], dv2 (&nxt)[PF], int ch, bool prefetch) {
        const int p0 = ch * PF;
        const double *x0 = xb + (size_t)(8 * p0) * NC;
        double bx[PF], by[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) {
            const int pi = p0 + i < pairs ? i : 0;                  // rows of a partial chunk's missing pairs: re-read pair 0
            bx[i] = x0[(8 * pi) * NC];
            by[i] = x0[(8 * pi + 4) * NC];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = mfma_f64(cur[0].x, bx[0], acc0);
        acc1 = mfma_f64(cur[0].y, by[0], acc1);
        __builtin_amdgcn_sched_barrier(0);
        if (prefetch) load(nxt, ch + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (p0 + PF <= pairs) {
#pragma unroll
            for (int i = 1; i < PF; ++i) {
                acc0 = mfma_f64(cur[i].x, bx[i], acc0);
                acc1 = mfma_f64(cur[i].y, by[i], acc1);
            }
        } else {
#pragma unroll
            for (int i = 1; i < PF; ++i) {
                if (p0 + i < pairs) {
                    acc0 = mfma_f64(cur[i].x, bx[i], acc0);
                    acc1 = mfma_f64(cur[i].y, by[i], acc1);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    dv2 a[PF], b[PF];
    if (nchunk > 0) load(a, 0);
    for (int ch = 0; ch < nchunk; ch += 2) {
        step(a, b, ch, ch + 1 < nchunk);
        if (ch + 1 < nchunk) step(b, a, ch + 1, ch + 2 < nchunk);
    }
}

__device__ __forceinline__ void gemm_forward(const double *__restrict__ Mp, int ntiles, int kpairs,
                                             const double *Xs, double *Ys, int wave, int lane)
{
    const int col = lane & 15, kq = lane >> 4;
    for (int t = wave; t < ntiles; t += NW) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double2 *mp = reinterpret_cast<const double2 *>(Mp) + ((size_t)t * kpairs) * 64 + lane;
        mfma_stream(mp, kpairs, Xs + kq * NC + col, acc0, acc1);
        // C/D layout of v_mfma_f64_16x16x4_f64: row = (lane >> 4) + 4 r, col = lane & 15
#pragma unroll
        for (int r = 0; r < 4; ++r) Ys[(16 * t + kq + 4 * r) * NC + col] = acc0[r] + acc1[r];
    }
}

// G[(16 t + i)][c] = sum_r M[r][16 t + i] Rm[r][c]; reduction rows = [A part (Ra) | L part (Rl)].
// Fragment order (pack_backward): element (tile t, pair p, lane l, h) = Mt[16 t + (l & 15)][8 p + 4 h + (l >> 4)],
// Mt = transposed stacked matrix with the A part padded to 8*rpairsA rows.
__device__ __forceinline__ void gemm_backward(const double *__restrict__ Mp, int ntiles, int rpairsA, int rpairsL,
                                              const double *Ra, const double *Rl, double *Gs, int wave, int lane)
{
    const int col = lane & 15, kq = lane >> 4;
    const int rp = rpairsA + rpairsL;
    for (int t = wave; t < ntiles; t += NW) {
        d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
        const double2 *mp = reinterpret_cast<const double2 *>(Mp) + ((size_t)t * rp) * 64 + lane;
        mfma_stream(mp, rpairsA, Ra + kq * NC + col, acc0, acc1);
        mfma_stream(mp + (size_t)rpairsA * 64, rpairsL, Rl + kq * NC + col, acc0, acc1);
#pragma unroll
        for (int r = 0; r < 4; ++r) Gs[(16 * t + kq + 4 * r) * NC + col] = acc0[r] + acc1[r];
    }
}

// Sum NQ per-thread partials over the row groups of each chain.  Deterministic order:
// lanes (c, c+16, c+32, c+48) by xor-shuffle, then the 8 waves in index order.  Result in out[q*NC + c].
template <int NQ>
__device__ __forceinline__ void chain_reduce(double (&v)[NQ], double *red, double *out, int tid)
{
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double x = v[q];
        x += __shfl_xor(x, 16);
        x += __shfl_xor(x, 32);
        if (lane < NC) red[(wave * NRED + q) * NC + lane] = x;
    }
    __syncthreads();
    if (tid < NQ * NC) {
        const int q = tid / NC, c = tid % NC;
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) s += red[(w * NRED + q) * NC + c];
        out[q * NC + c] = s;
    }
    __syncthreads();
}

// per-chain scalar slots in LDS
enum { S_RINF = 0, S_INDUC, S_SRES, S_AP, S_AR, S_AI, S_D0 /* 9 slots */, S_XSUM = 15, S_LP = 16, S_TMP = 17 /* 8 slots */,
       S_REJ = 25, S_RAW = 26 /* 6 slots: Rinf_raw, induc_raw, sigma_res_raw, alpha_*_raw */, S_NSLOT = 32 };

struct TileIO {
    const double *theta;   // unconstrained parameters
    long t_sc, t_sj;       // strides: chain, parameter
    const int *t_off;      // optional per-chain extra offset (doubles) of the theta row: the sampler's ping-pong rows; nullptr: none
    double *grad;          // may be nullptr
    long g_sc, g_sj;
    double *lp;            // [16] (global or LDS), stride 1; may be nullptr
    const int *spec;       // spectrum index per chain (nullptr: 0)
    int nvalid;            // chains in this tile that exist
    int jacobian;
    double *Z_hat;         // optional [chain][2nf] outputs (transformed parameters)
    double *sigma_tot;
    double *params;        // optional [chain][D] constrained parameters
    long long *prof;       // optional cycle counters per phase (thread 0 of the workgroup), slots 0..9
};

constexpr int UK = 6;      // elements per thread handled per unrolled batch in the K-loops  (K <= 192 -> one batch)
constexpr int UN = 3;      // ... in the Nf-loops (Nf <= 96 -> one batch)

// Evaluate lp and gradient for the 16 chains of this workgroup.  smem: lds_doubles(P) doubles.
// All threads of the workgroup must call.  Ends with a __syncthreads().
// Element-wise phases are written as "issue all loads of a batch, then compute": the strided theta / Z reads are
// L2 hits whose latency would otherwise be paid once per element.
template <bool TOEP>
__device__ inline void logp_grad_tile(const DevProblem &P, const TileIO &io, double *smem)
{
    int tid = threadIdx.x;
    // opaque to the optimiser: inside the sampler's round loop everything derived from the thread index is loop
    // invariant, gets hoisted by the hundreds (addresses, offsets) and is then spilled and reloaded every round
    __asm__ volatile("" : "+v"(tid));
    const int lane = tid & 63, wave = tid >> 6;
    const int c = tid & (NC - 1), g = tid >> 4;          // chain column, row group
    const int nf = P.nf, N2 = 2 * nf;
    const bool valid = c < io.nvalid;
    const int cc = valid ? c : 0;                         // padded columns replay chain 0 (never written back)
    const double jac = io.jacobian ? 1.0 : 0.0;

    double *Xs = smem;
    double *Zh = Xs + (size_t)P.XR * NC;
    double *Yp = Zh + (size_t)P.ZR * NC;
    double *Lr = Yp + (size_t)P.ZR * NC * P.npar;
    double *XC = Lr + (size_t)P.LR * NC;
    double *red = XC + (size_t)P.XCR * NC;
    double *sc = red + NW * NRED * NC;
    const bool cache_x = P.XCR > 0;

    const double *th = io.theta + (long)cc * io.t_sc + (io.t_off ? io.t_off[cc] : 0);
    auto TH = [&](int j) -> double { return th[(long)j * io.t_sj]; };
    double *gr = io.grad ? io.grad + (long)cc * io.g_sc : nullptr;
    auto GW = [&](int j, double v) { if (gr && valid) gr[(long)j * io.g_sj] = v; };
    double *pr = io.params ? io.params + (size_t)cc * P.D : nullptr;
    auto PW = [&](int j, double v) { if (pr && valid) pr[j] = v; };

    long long tprev = (io.prof && tid == 0) ? clock64() : 0;
#define BDRT_TILE_PROF(slot) do { if (io.prof && tid == 0) { const long long t_ = clock64(); io.prof[slot] += t_ - tprev; tprev = t_; } } while (0)
    // ---- phase 0: per-chain scalars, one exp per thread ------------------------------------------------
    {
        const int which = tid >> 4;                       // 0..5: Rinf_raw, induc_raw, 4 error raws; 6..: d strengths
        const int nsc = 6 + 3 * P.nblocks;
        if (which < nsc) {
            int j;
            if (which < 2) j = which;
            else if (which < 6) j = P.o_err + (which - 2);
            else j = P.blk[(which - 6) / 3].o_d + (which - 6) % 3;
            const double raw = exp(TH(j));
            if (which == 0) { sc[S_RINF * NC + c] = 100.0 * raw; sc[(S_RAW + 0) * NC + c] = raw; }
            else if (which == 1) { sc[S_INDUC * NC + c] = raw * P.induc_scale; sc[(S_RAW + 1) * NC + c] = raw; }
            else if (which < 6) { sc[(S_SRES + which - 2) * NC + c] = 0.05 * raw; sc[(S_RAW + which) * NC + c] = raw; }
            else sc[(S_D0 + which - 6) * NC + c] = raw;
        }
        if (tid < NC) { sc[S_LP * NC + c] = 0.0; sc[S_XSUM * NC + c] = 0.0; sc[S_REJ * NC + c] = 0.0; }
    }
    for (int i = tid; i < P.ZR * NC; i += NT) Zh[i] = 0.0;
    __syncthreads();
    BDRT_TILE_PROF(0);

    // ---- phase 1: Z_hat = sum_b (A_b x_b  or  1/(A_b x_b)) + offsets --------------------------------
    for (int b = 0; b < P.nblocks; ++b) {
        const DevBlock &B = P.blk[b];
        const int KP = 8 * B.kpairs;
        double *xc = XC + (size_t)P.xc_off[b] * NC;
        double xs2[2] = {0.0, 0.0};                       // sum x_raw, sum theta_x (log-Jacobian)
        for (int k0 = g; k0 < KP; k0 += NG * UK) {
            double t_[UK];
#pragma unroll
            for (int u = 0; u < UK; ++u) { const int k = k0 + NG * u; t_[u] = k < B.K ? TH(B.o_x + k) : 0.0; }
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                const int k = k0 + NG * u;
                if (k < KP) {
                    double xr = 0.0;
                    if (k < B.K) {
                        xr = B.is_pos ? exp(t_[u]) : t_[u];
                        xs2[0] += xr;
                        if (B.is_pos) xs2[1] += t_[u];
                        PW(B.o_x + k, xr);
                    }
                    Xs[k * NC + c] = xr * B.x_scale;      // xp = xp_raw * xp_scale (1 for series blocks)
                    if (cache_x) xc[(MAXBW + k) * NC + c] = xr;
                }
            }
        }
        if (cache_x && g < MAXBW) {                       // zero borders of the cached x (convolution halo)
            xc[g * NC + c] = 0.0;
            xc[(MAXBW + KP + g) * NC + c] = 0.0;
        }
        if (P.use_x_sum || (io.jacobian && B.is_pos)) {
            chain_reduce<2>(xs2, red, sc + S_TMP * NC, tid);
            if (tid < NC) {
                sc[S_XSUM * NC + c] += sc[(S_TMP + 0) * NC + c];
                if (io.jacobian) sc[S_LP * NC + c] += sc[(S_TMP + 1) * NC + c];
            }
        }
        __syncthreads();
        BDRT_TILE_PROF(1);
        if (!(P.dbg & 1)) gemm_forward(B.Af, B.tilesA, B.kpairs, Xs, Lr, wave, lane);     // T = A_b x_b  (in the Lr buffer)
        __syncthreads();
        BDRT_TILE_PROF(2);
        for (int n = g; n < nf; n += NG) {
            const double yr = Lr[n * NC + c], yi = Lr[(nf + n) * NC + c];
            if (!B.is_parallel) {
                Zh[n * NC + c] += yr;
                Zh[(nf + n) * NC + c] += yi;
            } else {
                double *Y = Yp + (size_t)B.yp_slot * P.ZR * NC;
                Y[n * NC + c] = yr;
                Y[(nf + n) * NC + c] = yi;
                const double idn = 1.0 / (yr * yr + yi * yi);
                Zh[n * NC + c] += yr * idn;               // Z_hat_p (Parallel_modelcode.txt:47)
                Zh[(nf + n) * NC + c] += -yi * idn;
            }
        }
        __syncthreads();
        BDRT_TILE_PROF(3);
    }

    // ---- phase 2: likelihood Z ~ normal(Z_hat, sigma_tot); g_Zhat in place --------------------------
    {
        const double Rinf = sc[S_RINF * NC + c], induc = sc[S_INDUC * NC + c];
        const double s_res = sc[S_SRES * NC + c], a_p = sc[S_AP * NC + c], a_r = sc[S_AR * NC + c],
                     a_i = sc[S_AI * NC + c];
        const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
        const double ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
        const int sp = io.spec ? io.spec[cc] : 0;
        const double *Zm = P.Z + (size_t)sp * N2;
        double acc[7] = {0, 0, 0, 0, 0, 0, 0};   // lp, S_R, S_L, S_h, S_hz2, S_hzr2, S_hzi2
        for (int n0 = g; n0 < nf; n0 += NG * UN) {
            double zre_[UN], zim_[UN], wn_[UN], t0_[UN], t1_[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int n = n0 + NG * u, nn = n < nf ? n : 0;
                zre_[u] = Zm[nn]; zim_[u] = Zm[nf + nn]; wn_[u] = P.w[nn];
                t0_[u] = 0.0; t1_[u] = 0.0;
                if (P.outlier_mode) { t0_[u] = TH(P.o_so + nn); t1_[u] = TH(P.o_so + nf + nn); }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int n = n0 + NG * u;
                if (n >= nf) continue;
                const double wn = wn_[u];
                const double zr = Zh[n * NC + c] + Rinf;
                const double zi = Zh[(nf + n) * NC + c] + induc * wn;
                double so_re = 0.0, so_im = 0.0, r0 = 0.0, r1 = 0.0;
                if (P.outlier_mode) {
                    r0 = exp(t0_[u]);
                    r1 = exp(t1_[u]);
                    PW(P.o_so + n, r0); PW(P.o_so + nf + n, r1);
                    if (P.outlier_mode == 1) so_re = so_im = 0.05 * r0 * r1;
                    else { so_re = 0.05 * r0; so_im = 0.05 * r1; }
                }
                const double common = ar2 * zr * zr + ai2 * zi * zi;
                const double s2_re = c0 + ap2 * zr * zr + common + so_re * so_re;
                const double s2_im = c0 + ap2 * zi * zi + common + so_im * so_im;
                const double e_re = zre_[u] - zr, e_im = zim_[u] - zi;
                // one reciprocal and one logarithm for the (re, im) pair
                const double prod = s2_re * s2_im, ip = 1.0 / prod;
                const double w_re = s2_im * ip, w_im = s2_re * ip;
                acc[0] += -0.5 * log(prod) - 0.5 * e_re * e_re * w_re - 0.5 * e_im * e_im * w_im;
                const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;
                const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
                const double gzr = e_re * w_re + 2.0 * zr * (h_re * (ap2 + ar2) + h_im * ar2);
                const double gzi = e_im * w_im + 2.0 * zi * (h_im * (ap2 + ai2) + h_re * ai2);
                Zh[n * NC + c] = gzr;
                Zh[(nf + n) * NC + c] = gzi;
                acc[1] += gzr;
                acc[2] += gzi * wn;
                acc[3] += h_re + h_im;
                acc[4] += h_re * zr * zr + h_im * zi * zi;
                acc[5] += (h_re + h_im) * zr * zr;
                acc[6] += (h_re + h_im) * zi * zi;
                if (io.Z_hat && valid) { io.Z_hat[(size_t)c * N2 + n] = zr; io.Z_hat[(size_t)c * N2 + nf + n] = zi; }
                if (io.sigma_tot && valid) {
                    io.sigma_tot[(size_t)c * N2 + n] = sqrt(s2_re);
                    io.sigma_tot[(size_t)c * N2 + nf + n] = sqrt(s2_im);
                }
                if (P.outlier_mode == 1) {
                    // sigma_out = raw .* scale * 0.05 ; raw ~ exponential(lambda) ; scale ~ inv_gamma(alpha, beta)
                    const double dso = 2.0 * so_re * (h_re + h_im);
                    GW(P.o_so + n, r0 * (0.05 * r1 * dso - P.so_lambda) + jac);
                    GW(P.o_so + nf + n, 0.05 * r0 * r1 * dso - (P.so_alpha + 1.0) + P.so_beta / r1 + jac);
                    acc[0] += -P.so_lambda * r0 - (P.so_alpha + 1.0) * t1_[u] - P.so_beta / r1 + jac * (t0_[u] + t1_[u]);
                } else if (P.outlier_mode == 2) {
                    GW(P.o_so + n, r0 * (0.05 * 2.0 * so_re * h_re - P.so_lambda) + jac);
                    GW(P.o_so + nf + n, r1 * (0.05 * 2.0 * so_im * h_im - P.so_lambda) + jac);
                    acc[0] += -P.so_lambda * (r0 + r1) + jac * (t0_[u] + t1_[u]);
                }
            }
        }
        chain_reduce<7>(acc, red, sc + S_TMP * NC, tid);
        if (tid < 6 * NC) {
            // one thread per (scalar parameter, chain): Rinf_raw, induc_raw ~ std_normal(); error raws ~ std_normal()
            const int which = tid >> 4;
            const double *S = sc + S_TMP * NC;
            const double raw = sc[(S_RAW + which) * NC + c];
            const int j = which < 2 ? which : P.o_err + (which - 2);
            double dl;                                        // d lp / d(constrained raw), likelihood part
            if (which == 0) dl = 100.0 * S[1 * NC + c];
            else if (which == 1) dl = P.induc_scale * S[2 * NC + c];
            else dl = 0.05 * 2.0 * (0.05 * raw) * S[(3 + which - 2) * NC + c];
            GW(j, raw * (dl - raw) + jac);
            PW(j, raw);
        }
        if (tid < NC) {
            double lp = sc[(S_TMP + 0) * NC + c];
            double tsum = 0.0;
#pragma unroll
            for (int q = 0; q < 6; ++q) { const double raw = sc[(S_RAW + q) * NC + c]; lp += -0.5 * raw * raw; }
            if (io.jacobian) {
                tsum = TH(0) + TH(1);
                for (int q = 0; q < 4; ++q) tsum += TH(P.o_err + q);
                lp += tsum;
            }
            if (P.use_x_sum) {
                const double xs_raw = sc[S_XSUM * NC + c];
                const double xs = xs_raw * P.x_sum_invscale;
                lp += -0.5 * xs * xs;                     // x_sum ~ std_normal()
                if (xs_raw < 0.0) sc[S_REJ * NC + c] = 1.0;   // real<lower=0> x_sum_raw
            }
            sc[S_LP * NC + c] += lp;
        }
        __syncthreads();
        BDRT_TILE_PROF(4);
    }

    // ---- phase 3: per block: q-prior, ups prior, back-propagation ------------------------------------
    for (int b = 0; b < P.nblocks; ++b) {
        const DevBlock &B = P.blk[b];
        const int K = B.K, KP = 8 * B.kpairs;
        double *xc = XC + (size_t)P.xc_off[b] * NC;
        if (cache_x && P.nblocks > 1) {
            // the blocks share ONE x-cache buffer (LDS budget): bring this block's raw x back (phase 1 left the last block's)
            for (int k0 = g; k0 < KP; k0 += NG * UK) {
                double t_[UK];
#pragma unroll
                for (int u = 0; u < UK; ++u) { const int k = k0 + NG * u; t_[u] = k < K ? TH(B.o_x + k) : 0.0; }
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int k = k0 + NG * u;
                    if (k < KP) xc[(MAXBW + k) * NC + c] = k < K ? (B.is_pos ? exp(t_[u]) : t_[u]) : 0.0;
                }
            }
            if (g < MAXBW) { xc[g * NC + c] = 0.0; xc[(MAXBW + KP + g) * NC + c] = 0.0; }
            __syncthreads();
        }
        if (cache_x) {
            if (!TOEP) for (int k = g; k < KP; k += NG) Xs[k * NC + c] = xc[(MAXBW + k) * NC + c];
        } else {
            for (int k0 = g; k0 < KP; k0 += NG * UK) {
                double t_[UK];
#pragma unroll
                for (int u = 0; u < UK; ++u) { const int k = k0 + NG * u; t_[u] = k < K ? TH(B.o_x + k) : 0.0; }
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int k = k0 + NG * u;
                    if (k < KP) Xs[k * NC + c] = k < K ? (B.is_pos ? exp(t_[u]) : t_[u]) : 0.0;   // raw x: q uses L * x_raw
                }
            }
        }
        double cv_[3][TOEP ? UK : 1];                      // structured path: v_i[k] of this thread's k's
        double wreg_[3][TOEP ? UK : 1];                    // structured path: w_i[k] = -d_i v_i / ups^2 (one LDS buffer, i after i)
#pragma unroll
        for (int u = 0; u < (TOEP ? UK : 1); ++u) { wreg_[0][u] = 0.0; wreg_[1][u] = 0.0; wreg_[2][u] = 0.0; }
        if (TOEP) {
            // v_i = L_i x as a (2*MAXBW+1)-tap convolution of the cached x.  Tap-outer loop: only the three
            // coefficients of one tap are live (scalar loads), the 18 accumulators stay in registers.
#pragma unroll
            for (int u = 0; u < UK; ++u) { cv_[0][u] = 0.0; cv_[1][u] = 0.0; cv_[2][u] = 0.0; }
#pragma unroll 1
            for (int d = 0; d <= 2 * MAXBW; ++d) {
                const double t0 = B.T[0][d], t1 = B.T[1][d], t2 = B.T[2][d];
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int k = g + NG * u;
                    const double xv = k < K ? xc[(k + d) * NC + c] : 0.0;           // x[k + d - MAXBW]
                    cv_[0][u] += t0 * xv; cv_[1][u] += t1 * xv; cv_[2][u] += t2 * xv;
                }
            }
            BDRT_TILE_PROF(5);
        } else {
            __syncthreads();
            BDRT_TILE_PROF(5);
            if (!(P.dbg & 1)) gemm_forward(B.Lf, B.tilesL, B.kpairs, Xs, Lr, wave, lane);     // v = [L0 x; L1 x; L2 x]
            __syncthreads();
        }
        BDRT_TILE_PROF(6);
        {
            const double d0 = sc[(S_D0 + 3 * b + 0) * NC + c], d1 = sc[(S_D0 + 3 * b + 1) * NC + c],
                         d2 = sc[(S_D0 + 3 * b + 2) * NC + c];
            double acc[4] = {0, 0, 0, 0};                 // lp, Sv0, Sv1, Sv2
            // ups = ups_raw*0.15 for every k first (the dups prior couples neighbours)
            for (int k0 = g; k0 < K; k0 += NG * UK) {
                double t_[UK];
#pragma unroll
                for (int u = 0; u < UK; ++u) { const int k = k0 + NG * u; t_[u] = k < K ? TH(B.o_ups + k) : 0.0; }
#pragma unroll
                for (int u = 0; u < UK; ++u) { const int k = k0 + NG * u; if (k < K) Xs[k * NC + c] = 0.15 * exp(t_[u]); }
            }
            __syncthreads();
            for (int k0 = g; k0 < K; k0 += NG * UK) {
                double t_[UK];
#pragma unroll
                for (int u = 0; u < UK; ++u) { const int k = k0 + NG * u; t_[u] = k < K ? TH(B.o_ups + k) : 0.0; }
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int k = k0 + NG * u;
                    if (k >= K) continue;
                    const double uu = Xs[k * NC + c], t = t_[u];
                    const double iu = 1.0 / uu, iu2 = iu * iu;
                    double v0, v1, v2;
                    if (TOEP) { v0 = cv_[0][TOEP ? u : 0]; v1 = cv_[1][TOEP ? u : 0]; v2 = cv_[2][TOEP ? u : 0]; }     // (K <= NG*UK: k0 == g)
                    else { v0 = Lr[k * NC + c]; v1 = Lr[(K + k) * NC + c]; v2 = Lr[(2 * K + k) * NC + c]; }
                    const double q2 = d0 * v0 * v0 + d1 * v1 * v1 + d2 * v2 * v2;
                    const double ir = 0.15 * iu;          // 1 / ups_raw
                    // q ~ normal(0, ups) evaluated on q^2; ups_raw ~ inv_gamma(alpha, beta); log transform
                    acc[0] += -(t + LOG_015) - 0.5 * q2 * iu2 - (P.ups_alpha + 1.0) * t - P.ups_beta * ir + jac * t;
                    acc[1] += v0 * v0 * iu2; acc[2] += v1 * v1 * iu2; acc[3] += v2 * v2 * iu2;
                    double gu = -iu + q2 * iu2 * iu;
                    // dups[k] = 0.5*(ups[k+1] - 0.5*(ups[k]+ups[k+2]))/ups[k+1] ~ std_normal()
                    const double um1 = k >= 1 ? Xs[(k - 1) * NC + c] : 1.0, up1 = k + 1 < K ? Xs[(k + 1) * NC + c] : 1.0;
                    if (k >= 1 && k + 1 < K) {
                        const double du = 0.5 * (uu - 0.5 * (um1 + up1)) * iu;
                        acc[0] += -0.5 * du * du;
                        gu += -du * 0.25 * (um1 + up1) * iu2;
                    }
                    if (k >= 2) {                             // k is the right neighbour of centre k-1
                        const double um2 = Xs[(k - 2) * NC + c], i0 = 1.0 / um1;
                        const double du = 0.5 * (um1 - 0.5 * (um2 + uu)) * i0;
                        gu += du * 0.25 * i0;
                    }
                    if (k + 2 < K) {                          // k is the left neighbour of centre k+1
                        const double up2 = Xs[(k + 2) * NC + c], i0 = 1.0 / up1;
                        const double du = 0.5 * (up1 - 0.5 * (uu + up2)) * i0;
                        gu += du * 0.25 * i0;
                    }
                    GW(B.o_ups + k, uu * gu - (P.ups_alpha + 1.0) + P.ups_beta * ir + jac);
                    PW(B.o_ups + k, uu * (1.0 / 0.15));
                    if (TOEP) {
                        wreg_[0][TOEP ? u : 0] = -d0 * v0 * iu2;
                        wreg_[1][TOEP ? u : 0] = -d1 * v1 * iu2;
                        wreg_[2][TOEP ? u : 0] = -d2 * v2 * iu2;
                        Lr[(MAXBW + k) * NC + c] = wreg_[0][TOEP ? u : 0];        // w_0 goes out right away
                        if (P.w3) {
                            const int WS = K + 2 * MAXBW;
                            Lr[(WS + MAXBW + k) * NC + c] = wreg_[1][TOEP ? u : 0];
                            Lr[(2 * WS + MAXBW + k) * NC + c] = wreg_[2][TOEP ? u : 0];
                        }
                    } else {
                        Lr[k * NC + c] = -d0 * v0 * iu2;
                        Lr[(K + k) * NC + c] = -d1 * v1 * iu2;
                        Lr[(2 * K + k) * NC + c] = -d2 * v2 * iu2;
                    }
                }
            }
            if (TOEP && g < MAXBW) {                    // zero halo of the w buffer(s)
                const int WS = K + 2 * MAXBW;
                for (int i = 0; i < (P.w3 ? 3 : 1); ++i) {
                    Lr[(i * WS + g) * NC + c] = 0.0;
                    Lr[(i * WS + MAXBW + K + g) * NC + c] = 0.0;
                }
            }
            // rows 3K .. 8*rpairsL of the reduction must be exactly zero
            if (!TOEP) for (int r = 3 * K + g; r < 8 * B.rpairsL; r += NG) Lr[r * NC + c] = 0.0;
            if (B.is_parallel) {
                // chain through Z_hat_p = conj(Y)/|Y|^2 : g_Y = J^T g_Z
                double *Y = Yp + (size_t)B.yp_slot * P.ZR * NC;
                for (int n = g; n < nf; n += NG) {
                    const double yr = Y[n * NC + c], yi = Y[(nf + n) * NC + c];
                    const double dn = yr * yr + yi * yi, id2 = 1.0 / (dn * dn);
                    const double dd = (yi * yi - yr * yr) * id2, doff = 2.0 * yr * yi * id2;
                    const double gzr = Zh[n * NC + c], gzi = Zh[(nf + n) * NC + c];
                    Y[n * NC + c] = (gzr * dd + gzi * doff) * B.x_scale;      // xp = xp_raw * xp_scale
                    Y[(nf + n) * NC + c] = (-gzr * doff + gzi * dd) * B.x_scale;
                }
                for (int r = N2 + g; r < P.ZR; r += NG) Y[r * NC + c] = 0.0;
            }
            chain_reduce<4>(acc, red, sc + S_TMP * NC, tid);
            if (tid < 3 * NC) {                           // d ~ inv_gamma(5,5), log transform: one thread per (i, chain)
                const int i = tid >> 4;
                const double dv = sc[(S_D0 + 3 * b + i) * NC + c];
                const double idv = 1.0 / dv;
                GW(B.o_d + i, -0.5 * dv * sc[(S_TMP + 1 + i) * NC + c] - 6.0 + 5.0 * idv + jac);
                PW(B.o_d + i, dv);
                sc[(S_TMP + 4 + i) * NC + c] = -5.0 * idv + (jac - 6.0) * TH(B.o_d + i);
            }
            __syncthreads();
            if (tid < NC)
                sc[S_LP * NC + c] += sc[(S_TMP + 0) * NC + c] + sc[(S_TMP + 4) * NC + c] + sc[(S_TMP + 5) * NC + c] +
                                     sc[(S_TMP + 6) * NC + c];
        }
        __syncthreads();
        const double *Ra = B.is_parallel ? Yp + (size_t)B.yp_slot * P.ZR * NC : Zh;
        BDRT_TILE_PROF(7);
        double gl_[TOEP ? UK : 1];                        // structured path: (sum_i L_i^T w_i)[k] of this thread's k's
#pragma unroll
        for (int u = 0; u < (TOEP ? UK : 1); ++u) gl_[u] = 0.0;
        if (TOEP) {
            if (!(P.dbg & 2)) gemm_backward(B.BkA, B.tilesK, B.rpairsA, 0, Ra, Lr, Xs, wave, lane);   // A^T g only
            if (P.w3) {
                const int WS = K + 2 * MAXBW;
#pragma unroll 1
                for (int d = 0; d <= 2 * MAXBW; ++d) {
                    // (L^T w)[k] = sum_d' L[k - d'][k] w[k - d'] with d' = d - MAXBW
                    const double t0 = B.T[0][d], t1 = B.T[1][d], t2 = B.T[2][d];
#pragma unroll
                    for (int u = 0; u < UK; ++u) {
                        const int k = g + NG * u;
                        if (k < K) {
                            const int r = 2 * MAXBW + k - d;                      // = MAXBW + (k - d')
                            gl_[u] += t0 * Lr[r * NC + c] + t1 * Lr[(WS + r) * NC + c] + t2 * Lr[(2 * WS + r) * NC + c];
                        }
                    }
                }
            } else
            // sum_i L_i^T w_i through ONE LDS buffer: w_0 is already there; w_1, w_2 follow (two barriers each)
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                if (i > 0) {
                    __syncthreads();                                          // everybody is done reading w_(i-1)
#pragma unroll
                    for (int u = 0; u < UK; ++u) { const int k = g + NG * u; if (k < K) Lr[(MAXBW + k) * NC + c] = wreg_[i][u]; }
                    __syncthreads();
                }
#pragma unroll 1
                for (int d = 0; d <= 2 * MAXBW; ++d) {
                    // (L^T w)[k] = sum_d' L[k - d'][k] w[k - d'] with d' = d - MAXBW
                    const double t = B.T[i][d];
#pragma unroll
                    for (int u = 0; u < UK; ++u) {
                        const int k = g + NG * u;
                        if (k < K) gl_[u] += t * Lr[(2 * MAXBW + k - d) * NC + c];        // row MAXBW + (k - d')
                    }
                }
            }
        } else {
            if (!(P.dbg & 2)) gemm_backward(B.Bk, B.tilesK, B.rpairsA, B.rpairsL, Ra, Lr, Xs, wave, lane);
        }
        __syncthreads();
        BDRT_TILE_PROF(8);
        if (gr) {
            const double xs_term = P.use_x_sum ? -sc[S_XSUM * NC + c] * P.x_sum_invscale * P.x_sum_invscale : 0.0;
            for (int k0 = g; k0 < K; k0 += NG * UK) {
                double x_[UK];
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int k = k0 + NG * u;
                    x_[u] = 1.0;
                    if (B.is_pos && k < K) x_[u] = cache_x ? xc[(MAXBW + k) * NC + c] : TH(B.o_x + k);
                }
#pragma unroll
                for (int u = 0; u < UK; ++u) {
                    const int k = k0 + NG * u;
                    if (k >= K) continue;
                    const double graw = Xs[k * NC + c] + xs_term + (TOEP ? gl_[TOEP ? u : 0] : 0.0);
                    if (B.is_pos) GW(B.o_x + k, (cache_x ? x_[u] : exp(x_[u])) * graw + jac);
                    else GW(B.o_x + k, graw);
                }
            }
        }
        __syncthreads();
    }

    BDRT_TILE_PROF(9);
    if (tid < NC && io.lp && valid) {
        double lp = sc[S_LP * NC + c];
        if (sc[S_REJ * NC + c] != 0.0) lp = -INFINITY;
        io.lp[c] = lp;
    }
    __syncthreads();
}

}  // namespace bdrt
