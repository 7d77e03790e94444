// bdrt_wave.h -- ONE chain per WAVEFRONT: the sampler of the headline family whose round time is proportional to the live chains.
//
// What the measurements of rounds 2-3 say about this machine: an fp64 MFMA occupies the SIMD's VALU for its
// 64 cycles and moves 2048 flop -- exactly the rate of v_fma_f64 --, so for the dense products of ONE chain a matrix tile buys
// nothing over the vector unit, and what the 16-chain tile kernel and the 512-thread one-chain kernel really pay for is
// lock-step: workgroup barriers, LDS round trips between waves, and an instruction stream whose length does not depend on
// how many of its columns / threads carry a live chain.  Here a chain is a workgroup of ONE wave:
//   * no workgroup barrier anywhere (a wave's LDS traffic is in program order), per-chain sums are DPP butterflies;
//   * theta, momentum, gradient and inverse metric stay in registers for the whole launch (lane l owns x[l + 64 u],
//     ups[l + 64 u] and one of the nine scalar parameters: a K-vector's neighbour coupling and the prior chain need no
//     re-mapping);
//   * A x and A^T g use the Toeplitz structure of A on log-uniform grids (reference matrices.py:197-205, 236-242) as
//     register-blocked VALU products over a 4-way interleaved generator table in LDS: forward 4 rows x both parts per lane
//     (the x operand is shared), backward 6 columns of one part per lane; L0, L1, L2 are the 13-tap convolutions of the
//     structured path;
//   * up to eight chains share a CU (two waves per SIMD, <= 20 KB of LDS each): the hardware interleaves independent chains,
//     a chain that closes a subtree or waits for a checkpoint row delays nobody else, and a launch with c live chains per CU
//     costs what c chains cost.
// Same model code (bayes_drt/stan_model_files/Series(_pos)_modelcode.txt:37-69), same transition logic (bdrt_nuts_device.h),
// same Philox indexing and the same state layout in HBM as the one-chain-per-workgroup kernel (bdrt_solo.h): chains move
// between the kernels at launch boundaries, results agree draw by draw up to summation order.
#pragma once
#include "bdrt_device.h"
#include "bdrt_nuts_device.h"
#include "bdrt_solo.h"

namespace bdrt {

constexpr int WV_NT = 64;

struct WaveGeom {
    int nf, K, D;
    int KS, NS;               // K / nf elements per lane
    int RGb, NP, ML, R4;      // forward product: 4-row groups, m-parts, m per part (multiple of 8), 4 * RGb
    int MG, KP, NLP;          // backward product: 6-column groups per part, 6 * MG, nf padded to a multiple of 8
    int S, GQ;                // generator: logical index n - m + S, quarter length of one part's swizzled table
    int KZ;                   // x is kept zero on [K, KZ): what the forward product reads beyond the basis
    int nb;                   // blocks (distributions) of the model: a generator table each
    int XL;                   // halo-padded K-row
    int DSS;                  // row stride of the D-vectors (as bdrt_solo.h)
    int o_xs, o_us, o_w, o_gen, o_zp, o_gz, o_tc, o_state, o_hot, total0;      // LDS offsets (doubles); total0: without hot rows
};

constexpr int WV_TCL = 3 * 16 + 8;            // doubles of one block's band-coefficient table in LDS (one batch of over-read behind it)
constexpr int WV_MAXPARTS = 8;            // m-parts of the forward product at most

__host__ __device__ inline WaveGeom wave_geometry(int nf, int K, int D, int nb = 1)
{
    WaveGeom g;
    g.nf = nf; g.K = K; g.D = D; g.nb = nb;
    g.KS = (K + 63) / 64; g.NS = (nf + 63) / 64;
    g.RGb = (nf + 3) / 4; g.R4 = 4 * g.RGb;
    int np = WV_NT / g.RGb;
    if (np > WV_MAXPARTS) np = WV_MAXPARTS;
    g.ML = ((K + np - 1) / np + 7) & ~7;
    g.NP = (K + g.ML - 1) / g.ML;
    g.MG = (K + 5) / 6; g.KP = 6 * g.MG;
    g.NLP = (nf + 7) & ~7;
    // index ranges (one block of over-read by the operand prefetch included)
    const int mx = g.NP * g.ML > g.KP ? g.NP * g.ML : g.KP;
    g.S = mx + 16;
    const int nmax = g.R4 > g.NLP + 16 ? g.R4 : g.NLP + 16;
    g.GQ = (g.S + nmax + 8) / 4 + 2;
    g.KZ = g.NP * g.ML + 8;                                  // (the pipelined products request one iteration -- eight steps -- beyond the end)
    const int kmax = K > g.KZ ? K : g.KZ;
    g.XL = (2 * MAXBW + kmax + 8) & ~1;
    g.DSS = (D + 7) & ~7;
    int o = 0;
    g.o_xs = o; o += g.XL;
    g.o_us = o; o += g.XL;
    g.o_w = o; o += 3 * g.XL;
    g.o_gen = o; o += nb * 2 * 4 * g.GQ;
    const int zp = g.NP * 2 * g.R4, gk = 2 * g.KP;
    g.o_zp = o; o += ((zp > gk ? zp : gk) + 1) & ~1;
    g.o_gz = o; o += 2 * g.NLP + 16;
    g.o_tc = o; o += nb * WV_TCL;          // band coefficients of L0, L1, L2, rows of 16 (taps 13..15 zero): the one-wave-per-SIMD instantiations read them here
    g.o_state = o; o += (int)((sizeof(ChainState) + 15) / 16) * 2 + 2;
    g.o_hot = o;
    g.total0 = o;
    return g;
}

// rows of the chain that stay in LDS for the launch when there is room (most used first); everything else is read and written
// where it is in HBM, each lane its own elements
constexpr int WV_HOT_MAX = 12;
__host__ __device__ constexpr int wave_hot_rank(int v)
{
    return v == SV_CKP ? 0 : v == SV_CKP + 1 ? 1 : v == SV_CKC + 1 ? 2 : v == SV_CKP + 2 ? 3 : v == SV_CKC + 2 ? 4
         : v == SV_THQ ? 5 : v == SV_GQ ? 6 : v == SV_CKP + 3 ? 7 : v == SV_CKC + 3 ? 8 : v == SV_RHO ? 9 : v == SV_THS ? 10
         : v == SV_GS ? 11 : 99;
}
__host__ __device__ inline size_t wave_lds_bytes(const WaveGeom &g, int nhot) { return ((size_t)g.total0 + (size_t)nhot * g.DSS) * sizeof(double) + 64; }

// can this problem take the one-chain-per-wave path?  (host)
// (the outlier error models -- two further parameters per frequency, bdrt_tile_hw.h -- take the same kernel with 2 NS more slots per lane
//  and one wave per SIMD: wave_chains_per_cu)
// (and the models of several distributions -- Series-Parallel, Series-2Parallel, any mix of series and parallel blocks with one basis
//  length on a log-uniform grid, with or without the x_sum prior and the outlier models: wave_eval_nb)
inline bool wave_capable(const DevProblem &P)
{
    const int K = P.blk[0].K;
    if (P.nblocks < 1 || P.nblocks > 3 || K < 2 * MAXBW + 3 || K > 192 || P.nf > 128) return false;
    if (P.nblocks > 1 && K <= 64) return false;            // (instantiated for two or three slots of basis functions per lane: bdrt_wave_nb.hip)
    for (int b = 0; b < P.nblocks; ++b)
        if (P.blk[b].K != K || P.blk[b].tg == nullptr || !P.blk[b].toep) return false;
    if (P.nblocks == 1 && !P.fast_s1) return false;
    return P.D == P.nblocks * (2 * K + 3) + 6 + (P.outlier_mode ? 2 * P.nf : 0);
}
// chains a CU keeps resident: two waves per SIMD for the headline family, one with more parameters per lane than that
inline int wave_chains_per_cu(const DevProblem &P) { return (P.outlier_mode || P.nblocks > 1) ? 4 : 8; }

typedef const __attribute__((address_space(4))) double *wv_cptr;      // uniform read-only data: scalar loads

// ---- wave-level helpers --------------------------------------------------------------------------------------------------
// LDS traffic of the one wave in program order: compiler-level ordering only (the hardware executes a wave's DS instructions in order)
__device__ __forceinline__ void wv_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// the value of lane `src` (uniform), as a scalar
__device__ __forceinline__ double wv_bcast(double x, int src)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), src), __builtin_amdgcn_readlane(__double2loint(x), src));
}
__device__ __forceinline__ double wv_xor32(double x)
{
    return __shfl_xor(x, 32);
}
__device__ __forceinline__ double wv_sum(double x)
{
    x = sum32(x);
    return wv_bcast(x, 0) + wv_bcast(x, 32);
}
// two sums in one butterfly (the first exchange keeps a in even lanes, b in odd lanes)
__device__ __forceinline__ void wv_sum2(double &a, double &b, int lane)
{
    const bool odd = lane & 1;
    double t = (odd ? b : a) + dpp_perm<0xB1>(odd ? a : b);
    t += dpp_perm<0x4E>(t);        // every later exchange keeps the lane's parity: xor 2, 4, 8, 16
    t += dpp_xor4(t);
    t += dpp_perm<0x128>(t);
    t += swizzle_xor16(t);
    a = wv_bcast(t, 0) + wv_bcast(t, 32);
    b = wv_bcast(t, 1) + wv_bcast(t, 33);
}

// ---- Toeplitz products ---------------------------------------------------------------------------------------------------------
// gen: one part's generator in LDS, swizzled [4][GQ]: logical element e at gen[(e & 3) * GQ + (e >> 2)].  Four per-lane
// pointers make element E + c (c a compile-time constant) the address q[c & 3] + (c >> 2): lanes whose E differ by a
// multiple of four read consecutive addresses, any E is allowed.
// (Measured and dropped in round 4: the generator handed up each row of 16 lanes by DPP with only the row's first lane reading
//  LDS -- R x R blocks, LDS traffic a third.  The DPP moves are VALU instructions, the LDS reads they replace are not: with two
//  waves per SIMD the kernel is bound by VALU issue, and with one it gained nothing either: 5.7 k / 7.6 k cycles against 5.4 k / 6.1 k.)
struct ToepPtr { const double *q[4]; };
__device__ __forceinline__ ToepPtr toep_ptrs(const double *gen, int GQ, int E)
{
    ToepPtr t;
    const int rho = E & 3, Q = E >> 2;
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int c = rho + r; t.q[r] = gen + (c & 3) * GQ + Q + (c >> 2); }
    return t;
}
#define BDRT_TE(t, c, o) (t).q[(c) & 3][((c) >> 2) + (o)]

// Both products run as a software pipeline of four-step blocks, two blocks per loop iteration: the operands of block j + 1 (the
// four new generator elements of each part, four vector elements) are requested before the FMAs of block j.
// The last iteration requests one block beyond the end: in-bounds LDS (margins of wave_geometry), never used.
// (The optimiser folds the two requests of an iteration into one batch of ds_read2_b64 / ds_read_b128 at the top of the loop and
//  waits for it; pinning the order with empty asm statements that consume the operands where they must have arrived kept the
//  pipeline but cost more than it hid -- 6.2 k / 6.7 k cycles against 5.4 k / 6.1 k per product, worse with two waves per SIMD.)

// forward: rows n0 .. n0 + 3 of BOTH parts, columns m0 .. m0 + len - 1 (len a multiple of 8):
//   ar[i] += sum_t genA[E + i - t] v[t],  ai[i] likewise with genB = genA + 4 GQ;  E = n0 - m0 + S
__device__ __forceinline__ void wave_toep_fwd(const double *genA, int GQ, int E, const double *v, int len, double (&ar)[4], double (&ai)[4])
{
    const ToepPtr A = toep_ptrs(genA, GQ, E), Bq = toep_ptrs(genA + 4 * GQ, GQ, E);
    double a0 = BDRT_TE(A, 0, 0), a1 = BDRT_TE(A, 1, 0), a2 = BDRT_TE(A, 2, 0), a3 = BDRT_TE(A, 3, 0);
    double b0 = BDRT_TE(Bq, 0, 0), b1 = BDRT_TE(Bq, 1, 0), b2 = BDRT_TE(Bq, 2, 0), b3 = BDRT_TE(Bq, 3, 0);
    // block: steps t .. t + 3 with x = (xa, xb) and the new elements E - 1 - t .. E - 4 - t of both parts; slides the windows
#define BDRT_WV_FWD_BLOCK(xa, xb, na, nb, nc, nd, ma, mb, mc, md)                                                                   \
    ar[0] = fma(a0, xa.x, ar[0]); ar[1] = fma(a1, xa.x, ar[1]); ar[2] = fma(a2, xa.x, ar[2]); ar[3] = fma(a3, xa.x, ar[3]);        \
    ai[0] = fma(b0, xa.x, ai[0]); ai[1] = fma(b1, xa.x, ai[1]); ai[2] = fma(b2, xa.x, ai[2]); ai[3] = fma(b3, xa.x, ai[3]);        \
    ar[0] = fma(na, xa.y, ar[0]); ar[1] = fma(a0, xa.y, ar[1]); ar[2] = fma(a1, xa.y, ar[2]); ar[3] = fma(a2, xa.y, ar[3]);        \
    ai[0] = fma(ma, xa.y, ai[0]); ai[1] = fma(b0, xa.y, ai[1]); ai[2] = fma(b1, xa.y, ai[2]); ai[3] = fma(b2, xa.y, ai[3]);        \
    ar[0] = fma(nb, xb.x, ar[0]); ar[1] = fma(na, xb.x, ar[1]); ar[2] = fma(a0, xb.x, ar[2]); ar[3] = fma(a1, xb.x, ar[3]);        \
    ai[0] = fma(mb, xb.x, ai[0]); ai[1] = fma(ma, xb.x, ai[1]); ai[2] = fma(b0, xb.x, ai[2]); ai[3] = fma(b1, xb.x, ai[3]);        \
    ar[0] = fma(nc, xb.y, ar[0]); ar[1] = fma(nb, xb.y, ar[1]); ar[2] = fma(na, xb.y, ar[2]); ar[3] = fma(a0, xb.y, ar[3]);        \
    ai[0] = fma(mc, xb.y, ai[0]); ai[1] = fma(mb, xb.y, ai[1]); ai[2] = fma(ma, xb.y, ai[2]); ai[3] = fma(b0, xb.y, ai[3]);        \
    a3 = na; a2 = nb; a1 = nc; a0 = nd;                                                                                             \
    b3 = ma; b2 = mb; b1 = mc; b0 = md;
    double2 xa = *reinterpret_cast<const double2 *>(v), xb = *reinterpret_cast<const double2 *>(v + 2);
    double na = BDRT_TE(A, -1, 0), nb = BDRT_TE(A, -2, 0), nc = BDRT_TE(A, -3, 0), nd = BDRT_TE(A, -4, 0);
    double ma = BDRT_TE(Bq, -1, 0), mb = BDRT_TE(Bq, -2, 0), mc = BDRT_TE(Bq, -3, 0), md = BDRT_TE(Bq, -4, 0);
#pragma unroll 1
    for (int t = 0; t < len; t += 8) {
        const int o = -(t >> 2);
        const double2 ya = *reinterpret_cast<const double2 *>(v + t + 4), yb = *reinterpret_cast<const double2 *>(v + t + 6);
        const double pa = BDRT_TE(A, -1, o - 1), pb = BDRT_TE(A, -2, o - 1), pc = BDRT_TE(A, -3, o - 1), pd = BDRT_TE(A, -4, o - 1);
        const double qa = BDRT_TE(Bq, -1, o - 1), qb = BDRT_TE(Bq, -2, o - 1), qc = BDRT_TE(Bq, -3, o - 1), qd = BDRT_TE(Bq, -4, o - 1);
        BDRT_WV_FWD_BLOCK(xa, xb, na, nb, nc, nd, ma, mb, mc, md)
        xa = *reinterpret_cast<const double2 *>(v + t + 8); xb = *reinterpret_cast<const double2 *>(v + t + 10);
        na = BDRT_TE(A, -1, o - 2); nb = BDRT_TE(A, -2, o - 2); nc = BDRT_TE(A, -3, o - 2); nd = BDRT_TE(A, -4, o - 2);
        ma = BDRT_TE(Bq, -1, o - 2); mb = BDRT_TE(Bq, -2, o - 2); mc = BDRT_TE(Bq, -3, o - 2); md = BDRT_TE(Bq, -4, o - 2);
        BDRT_WV_FWD_BLOCK(ya, yb, pa, pb, pc, pd, qa, qb, qc, qd)
    }
#undef BDRT_WV_FWD_BLOCK
}

// backward: columns m0 .. m0 + 5 of ONE part, rows 0 .. len - 1 (len a multiple of 8): acc[i] += sum_t gen[E + t - i] v[t], E = S - m0
__device__ __forceinline__ void wave_toep_bwd(const double *gen, int GQ, int E, const double *v, int len, double (&acc)[6])
{
    const ToepPtr A = toep_ptrs(gen, GQ, E);
    double w0 = BDRT_TE(A, 0, 0), w1 = BDRT_TE(A, -1, 0), w2 = BDRT_TE(A, -2, 0), w3 = BDRT_TE(A, -3, 0), w4 = BDRT_TE(A, -4, 0), w5 = BDRT_TE(A, -5, 0);
    // block: steps t .. t + 3 with g = (xa, xb) and the new elements E + t + 1 .. E + t + 4; slides the window
#define BDRT_WV_BWD_BLOCK(xa, xb, n1, n2, n3, n4)                                                                                   \
    acc[0] = fma(w0, xa.x, acc[0]); acc[1] = fma(w1, xa.x, acc[1]); acc[2] = fma(w2, xa.x, acc[2]);                                \
    acc[3] = fma(w3, xa.x, acc[3]); acc[4] = fma(w4, xa.x, acc[4]); acc[5] = fma(w5, xa.x, acc[5]);                                \
    acc[0] = fma(n1, xa.y, acc[0]); acc[1] = fma(w0, xa.y, acc[1]); acc[2] = fma(w1, xa.y, acc[2]);                                \
    acc[3] = fma(w2, xa.y, acc[3]); acc[4] = fma(w3, xa.y, acc[4]); acc[5] = fma(w4, xa.y, acc[5]);                                \
    acc[0] = fma(n2, xb.x, acc[0]); acc[1] = fma(n1, xb.x, acc[1]); acc[2] = fma(w0, xb.x, acc[2]);                                \
    acc[3] = fma(w1, xb.x, acc[3]); acc[4] = fma(w2, xb.x, acc[4]); acc[5] = fma(w3, xb.x, acc[5]);                                \
    acc[0] = fma(n3, xb.y, acc[0]); acc[1] = fma(n2, xb.y, acc[1]); acc[2] = fma(n1, xb.y, acc[2]);                                \
    acc[3] = fma(w0, xb.y, acc[3]); acc[4] = fma(w1, xb.y, acc[4]); acc[5] = fma(w2, xb.y, acc[5]);                                \
    w5 = w1; w4 = w0; w3 = n1; w2 = n2; w1 = n3; w0 = n4;
    double2 xa = *reinterpret_cast<const double2 *>(v), xb = *reinterpret_cast<const double2 *>(v + 2);
    double n1 = BDRT_TE(A, 1, 0), n2 = BDRT_TE(A, 2, 0), n3 = BDRT_TE(A, 3, 0), n4 = BDRT_TE(A, 4, 0);
#pragma unroll 1
    for (int t = 0; t < len; t += 8) {
        const int o = t >> 2;
        const double2 ya = *reinterpret_cast<const double2 *>(v + t + 4), yb = *reinterpret_cast<const double2 *>(v + t + 6);
        const double p1 = BDRT_TE(A, 1, o + 1), p2 = BDRT_TE(A, 2, o + 1), p3 = BDRT_TE(A, 3, o + 1), p4 = BDRT_TE(A, 4, o + 1);
        BDRT_WV_BWD_BLOCK(xa, xb, n1, n2, n3, n4)
        xa = *reinterpret_cast<const double2 *>(v + t + 8); xb = *reinterpret_cast<const double2 *>(v + t + 10);
        n1 = BDRT_TE(A, 1, o + 2); n2 = BDRT_TE(A, 2, o + 2); n3 = BDRT_TE(A, 3, o + 2); n4 = BDRT_TE(A, 4, o + 2);
        BDRT_WV_BWD_BLOCK(ya, yb, p1, p2, p3, p4)
    }
#undef BDRT_WV_BWD_BLOCK
}

// ---- the same products for ONE wave per SIMD (the OCC = 1 instantiations of wave_eval / nuts_wave_kernel) ------------------------------------------
// With a second wave on the SIMD the LDS latency of a block's operands is the other wave's issue time; alone, the loop above exposes it
// once per iteration (the scheduler batches the twelve reads at the top and the first FMA waits for them: ~230 of ~500 cycles).  Here
// the operands of a whole iteration (eight steps) are requested one iteration ahead into a second register set; the loop is unrolled by
// two so that the sets swap roles without moves.
struct ToepFwdOps { double2 xa, xb, ya, yb; double n[4], m[4], p[4], q[4]; };
// the generator elements of the first iteration and the initial windows: they do not depend on the point, so the evaluator requests them
// before it computes the constrained parameters (their latency runs under the exponentials)
struct ToepFwdPre { ToepPtr A, Bq; double a[4], b[4]; ToepFwdOps w; };
__device__ __forceinline__ void toep_fwd_load_gen(const ToepPtr &A, const ToepPtr &Bq, ToepFwdOps &w, int t)
{
    const int o = -(t >> 2);
    w.n[0] = BDRT_TE(A, -1, o); w.n[1] = BDRT_TE(A, -2, o); w.n[2] = BDRT_TE(A, -3, o); w.n[3] = BDRT_TE(A, -4, o);
    w.m[0] = BDRT_TE(Bq, -1, o); w.m[1] = BDRT_TE(Bq, -2, o); w.m[2] = BDRT_TE(Bq, -3, o); w.m[3] = BDRT_TE(Bq, -4, o);
    w.p[0] = BDRT_TE(A, -1, o - 1); w.p[1] = BDRT_TE(A, -2, o - 1); w.p[2] = BDRT_TE(A, -3, o - 1); w.p[3] = BDRT_TE(A, -4, o - 1);
    w.q[0] = BDRT_TE(Bq, -1, o - 1); w.q[1] = BDRT_TE(Bq, -2, o - 1); w.q[2] = BDRT_TE(Bq, -3, o - 1); w.q[3] = BDRT_TE(Bq, -4, o - 1);
}
__device__ __forceinline__ ToepFwdPre wave_toep_fwd_pre(const double *genA, int GQ, int E)
{
    ToepFwdPre r;
    r.A = toep_ptrs(genA, GQ, E); r.Bq = toep_ptrs(genA + 4 * GQ, GQ, E);
    r.a[0] = BDRT_TE(r.A, 0, 0); r.a[1] = BDRT_TE(r.A, 1, 0); r.a[2] = BDRT_TE(r.A, 2, 0); r.a[3] = BDRT_TE(r.A, 3, 0);
    r.b[0] = BDRT_TE(r.Bq, 0, 0); r.b[1] = BDRT_TE(r.Bq, 1, 0); r.b[2] = BDRT_TE(r.Bq, 2, 0); r.b[3] = BDRT_TE(r.Bq, 3, 0);
    toep_fwd_load_gen(r.A, r.Bq, r.w, 0);
    return r;
}
__device__ __forceinline__ void wave_toep_fwd_pipe(const ToepFwdPre &pre, const double *v, int len, double (&ar)[4], double (&ai)[4])
{
    const ToepPtr &A = pre.A, &Bq = pre.Bq;
    double a0 = pre.a[0], a1 = pre.a[1], a2 = pre.a[2], a3 = pre.a[3];
    double b0 = pre.b[0], b1 = pre.b[1], b2 = pre.b[2], b3 = pre.b[3];
    auto loadx = [&](ToepFwdOps &w, int t) {
        w.xa = *reinterpret_cast<const double2 *>(v + t); w.xb = *reinterpret_cast<const double2 *>(v + t + 2);
        w.ya = *reinterpret_cast<const double2 *>(v + t + 4); w.yb = *reinterpret_cast<const double2 *>(v + t + 6);
    };
    auto load = [&](ToepFwdOps &w, int t) { loadx(w, t); toep_fwd_load_gen(A, Bq, w, t); };
    auto block = [&](const double2 &xa, const double2 &xb, const double (&n)[4], const double (&m)[4]) {
        ar[0] = fma(a0, xa.x, ar[0]); ar[1] = fma(a1, xa.x, ar[1]); ar[2] = fma(a2, xa.x, ar[2]); ar[3] = fma(a3, xa.x, ar[3]);
        ai[0] = fma(b0, xa.x, ai[0]); ai[1] = fma(b1, xa.x, ai[1]); ai[2] = fma(b2, xa.x, ai[2]); ai[3] = fma(b3, xa.x, ai[3]);
        ar[0] = fma(n[0], xa.y, ar[0]); ar[1] = fma(a0, xa.y, ar[1]); ar[2] = fma(a1, xa.y, ar[2]); ar[3] = fma(a2, xa.y, ar[3]);
        ai[0] = fma(m[0], xa.y, ai[0]); ai[1] = fma(b0, xa.y, ai[1]); ai[2] = fma(b1, xa.y, ai[2]); ai[3] = fma(b2, xa.y, ai[3]);
        ar[0] = fma(n[1], xb.x, ar[0]); ar[1] = fma(n[0], xb.x, ar[1]); ar[2] = fma(a0, xb.x, ar[2]); ar[3] = fma(a1, xb.x, ar[3]);
        ai[0] = fma(m[1], xb.x, ai[0]); ai[1] = fma(m[0], xb.x, ai[1]); ai[2] = fma(b0, xb.x, ai[2]); ai[3] = fma(b1, xb.x, ai[3]);
        ar[0] = fma(n[2], xb.y, ar[0]); ar[1] = fma(n[1], xb.y, ar[1]); ar[2] = fma(n[0], xb.y, ar[2]); ar[3] = fma(a0, xb.y, ar[3]);
        ai[0] = fma(m[2], xb.y, ai[0]); ai[1] = fma(m[1], xb.y, ai[1]); ai[2] = fma(m[0], xb.y, ai[2]); ai[3] = fma(b0, xb.y, ai[3]);
        a3 = n[0]; a2 = n[1]; a1 = n[2]; a0 = n[3];
        b3 = m[0]; b2 = m[1]; b1 = m[2]; b0 = m[3];
    };
    auto compute = [&](const ToepFwdOps &w) { block(w.xa, w.xb, w.n, w.m); block(w.ya, w.yb, w.p, w.q); };
    ToepFwdOps w0 = pre.w, w1;
    loadx(w0, 0);
    int t = 0;
#pragma unroll 1
    for (; t + 16 <= len; t += 16) {
        load(w1, t + 8);
        compute(w0);
        load(w0, t + 16);
        compute(w1);
    }
    if (t < len) compute(w0);
}
struct ToepBwdOps { double2 xa, xb, ya, yb; double n[4], p[4]; };
struct ToepBwdPre { ToepPtr A; double w[6]; ToepBwdOps q; };
__device__ __forceinline__ void toep_bwd_load_gen(const ToepPtr &A, ToepBwdOps &w, int t)
{
    const int o = t >> 2;
    w.n[0] = BDRT_TE(A, 1, o); w.n[1] = BDRT_TE(A, 2, o); w.n[2] = BDRT_TE(A, 3, o); w.n[3] = BDRT_TE(A, 4, o);
    w.p[0] = BDRT_TE(A, 1, o + 1); w.p[1] = BDRT_TE(A, 2, o + 1); w.p[2] = BDRT_TE(A, 3, o + 1); w.p[3] = BDRT_TE(A, 4, o + 1);
}
__device__ __forceinline__ ToepBwdPre wave_toep_bwd_pre(const double *gen, int GQ, int E)
{
    ToepBwdPre r;
    r.A = toep_ptrs(gen, GQ, E);
    r.w[0] = BDRT_TE(r.A, 0, 0); r.w[1] = BDRT_TE(r.A, -1, 0); r.w[2] = BDRT_TE(r.A, -2, 0); r.w[3] = BDRT_TE(r.A, -3, 0); r.w[4] = BDRT_TE(r.A, -4, 0); r.w[5] = BDRT_TE(r.A, -5, 0);
    toep_bwd_load_gen(r.A, r.q, 0);
    return r;
}
__device__ __forceinline__ void wave_toep_bwd_pipe(const ToepBwdPre &pre, const double *v, int len, double (&acc)[6])
{
    const ToepPtr &A = pre.A;
    double w0 = pre.w[0], w1 = pre.w[1], w2 = pre.w[2], w3 = pre.w[3], w4 = pre.w[4], w5 = pre.w[5];
    auto loadx = [&](ToepBwdOps &w, int t) {
        w.xa = *reinterpret_cast<const double2 *>(v + t); w.xb = *reinterpret_cast<const double2 *>(v + t + 2);
        w.ya = *reinterpret_cast<const double2 *>(v + t + 4); w.yb = *reinterpret_cast<const double2 *>(v + t + 6);
    };
    auto load = [&](ToepBwdOps &w, int t) { loadx(w, t); toep_bwd_load_gen(A, w, t); };
    auto block = [&](const double2 &xa, const double2 &xb, const double (&n)[4]) {
        acc[0] = fma(w0, xa.x, acc[0]); acc[1] = fma(w1, xa.x, acc[1]); acc[2] = fma(w2, xa.x, acc[2]);
        acc[3] = fma(w3, xa.x, acc[3]); acc[4] = fma(w4, xa.x, acc[4]); acc[5] = fma(w5, xa.x, acc[5]);
        acc[0] = fma(n[0], xa.y, acc[0]); acc[1] = fma(w0, xa.y, acc[1]); acc[2] = fma(w1, xa.y, acc[2]);
        acc[3] = fma(w2, xa.y, acc[3]); acc[4] = fma(w3, xa.y, acc[4]); acc[5] = fma(w4, xa.y, acc[5]);
        acc[0] = fma(n[1], xb.x, acc[0]); acc[1] = fma(n[0], xb.x, acc[1]); acc[2] = fma(w0, xb.x, acc[2]);
        acc[3] = fma(w1, xb.x, acc[3]); acc[4] = fma(w2, xb.x, acc[4]); acc[5] = fma(w3, xb.x, acc[5]);
        acc[0] = fma(n[2], xb.y, acc[0]); acc[1] = fma(n[1], xb.y, acc[1]); acc[2] = fma(n[0], xb.y, acc[2]);
        acc[3] = fma(w0, xb.y, acc[3]); acc[4] = fma(w1, xb.y, acc[4]); acc[5] = fma(w2, xb.y, acc[5]);
        w5 = w1; w4 = w0; w3 = n[0]; w2 = n[1]; w1 = n[2]; w0 = n[3];
    };
    auto compute = [&](const ToepBwdOps &w) { block(w.xa, w.xb, w.n); block(w.ya, w.yb, w.p); };
    ToepBwdOps q0 = pre.q, q1;
    loadx(q0, 0);
    int t = 0;
#pragma unroll 1
    for (; t + 16 <= len; t += 16) {
        load(q1, t + 8);
        compute(q0);
        load(q0, t + 16);
        compute(q1);
    }
    if (t < len) compute(q0);
}

// per-launch constants of the evaluator held in registers: the measured spectrum at this lane's rows, its m-part of the forward
// product.  (Re-reading the spectrum from L2 per evaluation instead -- twelve registers less across the products -- was measured:
// the scheduler spends the freedom on longer live ranges elsewhere, 224 instead of 198 registers for the evaluator alone.)
template <int NS>
struct WaveEvalRegs {
    double zre[NS], zim[NS], wn[NS];
    int fpart;                // forward product: this lane's m-part (lane / RGb)
};

template <int NS>
__device__ __forceinline__ WaveEvalRegs<NS> wave_eval_setup(const DevProblem &P, const WaveGeom &g, int spec, int lane)
{
    WaveEvalRegs<NS> er;
    const double *Zm = P.Z + (size_t)spec * 2 * g.nf;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int n = lane + 64 * s < g.nf ? lane + 64 * s : 0;
        er.zre[s] = Zm[n]; er.zim[s] = Zm[g.nf + n]; er.wn[s] = P.w[n];
    }
    er.fpart = lane / g.RGb;
    return er;
}

// one-time LDS set-up of the evaluator: zero halos / pads, the two generators in swizzled order
__device__ __forceinline__ void wave_eval_init(const DevProblem &P, const WaveGeom &g, double *lds, int lane)
{
    for (int i = lane; i < g.o_gen; i += WV_NT) lds[i] = (i >= g.o_us && i < g.o_us + g.XL) ? 1.0 : 0.0;
    for (int i = lane; i < 2 * g.NLP + 16; i += WV_NT) lds[g.o_gz + i] = 0.0;
    for (int i = lane; i < g.nb * WV_TCL; i += WV_NT) {
        const int blk = i / WV_TCL, r = i - blk * WV_TCL, row = r >> 4, d = r & 15;
        lds[g.o_tc + i] = (row < 3 && d < 2 * MAXBW + 1) ? P.blk[blk].T[row][d] : 0.0;
    }
    const int glen = g.nf + g.K - 1;
    for (int i = lane; i < g.nb * 2 * 4 * g.GQ; i += WV_NT) {
        const int blk = i / (2 * 4 * g.GQ), i2 = i - blk * 2 * 4 * g.GQ;
        const int b = i2 / (4 * g.GQ), r = i2 - b * 4 * g.GQ, rho = r / g.GQ, q = r - rho * g.GQ;
        const int e = 4 * q + rho;                                                   // logical index: n - m + S
        const int src = e - g.S + g.K - 1;
        const double *tg = P.blk[blk].tg;                                            // [2][nf + K - 1]: c_b[n - m + K - 1]
        lds[g.o_gen + i] = (src >= 0 && src < glen) ? tg[(size_t)b * glen + src] : 0.0;
    }
}

// D-index of this lane's slot u: slots 0 .. KS-1 x[l + 64 u], KS .. 2KS-1 ups[l + 64 u], slot 2 KS the nine scalars (lanes 0..8:
// Rinf_raw, induc_raw, sigma_res_raw, alpha_prop_raw, alpha_re_raw, alpha_im_raw, d0, d1, d2); -1: no element
// With the outlier error model (OM): slots 2 KS + 1 .. 2 KS + NS the first outlier parameter of rows n = l + 64 s, the NS after them the second.
template <int KS, int NS = 0, bool OM = false>
__device__ __forceinline__ int wave_slot_index(const DevProblem &P, int K, int u, int lane)
{
    const DevBlock &B = P.blk[0];
    if (u < KS) return lane + 64 * u < K ? B.o_x + lane + 64 * u : -1;
    if (u < 2 * KS) return lane + 64 * (u - KS) < K ? B.o_ups + lane + 64 * (u - KS) : -1;
    if (u == 2 * KS) return lane < 2 ? lane : (lane < 6 ? P.o_err + (lane - 2) : (lane < 9 ? B.o_d + (lane - 6) : -1));
    if constexpr (OM) {
        const int s = u - (2 * KS + 1), h = s >= NS ? 1 : 0, n = lane + 64 * (s - h * NS);
        return n < P.nf ? P.o_so + h * P.nf + n : -1;
    }
    return -1;
}
template <int KS, int NS, bool OM> constexpr int wave_slots() { return 2 * KS + 1 + (OM ? 2 * NS : 0); }

// log-posterior + gradient of the chain at theta (registers, slot order above) -> gradient (registers), returns lp (uniform).
// Formulas: bdrt_solo.h / bdrt_tile_s1.h (same model code, other thread mapping).  `jac`: 1.0 with the Jacobian of the
// lower = 0 transforms (sampling), 0.0 without (optimisation).
// OCC: waves per SIMD the instantiation is scheduled for.  2 (default): the kernel of a full CU (eight chains), bound by VALU issue.  1: at most four
// chains on the CU -- nothing hides a wave's own LDS / scalar-load latency, so the products request their operands an iteration ahead
// (wave_toep_*_pipe), the band coefficients come from LDS instead of scalar loads, the neighbour terms of the prior are branch-free.
template <int KS, int NS, bool OM = false, int OCC = 2>
__device__ __forceinline__ double wave_eval(const DevProblem &P, const WaveGeom &g, double *lds, const double (&th)[wave_slots<KS, NS, OM>()],
                                            double (&gr)[wave_slots<KS, NS, OM>()], const WaveEvalRegs<NS> &er, const double jac, int lane_,
                                            long long *prof = nullptr)
{
    int lane = lane_;
    __asm__ volatile("" : "+v"(lane));                     // per-lane LDS addresses are recomputed per evaluation, not hoisted and spilled
    long long tprev = prof ? clock64() : 0;
#define BDRT_WV_PROF(slot) do { if (prof) { const long long t_ = clock64(); if (lane == 0) prof[slot] += t_ - tprev; tprev = t_; } } while (0)
    const DevBlock &B = P.blk[0];
    const int nf = g.nf, K = g.K;
    double *xs = lds + g.o_xs, *us = lds + g.o_us, *wr = lds + g.o_w;
    const double *gen = lds + g.o_gen;
    double *zp = lds + g.o_zp, *gz = lds + g.o_gz;
    double lp = 0.0;                                       // this lane's share of lp
    [[maybe_unused]] ToepFwdPre fpre;
    if constexpr (OCC == 1) {
        // (lanes behind the last part: the last part's addresses, loaded and not used)
        const int fpart_c = er.fpart < g.NP ? er.fpart : g.NP - 1;
        fpre = wave_toep_fwd_pre(gen, g.GQ, 4 * (lane - er.fpart * g.RGb) - fpart_c * g.ML + g.S);
    }

    // ---- E0: constrained parameters ----------------------------------------------------------------------------------
    double x[KS], uu[KS];
    bool kv[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        const int k = lane + 64 * u;
        kv[u] = k < K;
        const double tx = th[u], tu = th[KS + u];
        const double ex = lean_exp(tx);
        x[u] = kv[u] ? (B.is_pos ? ex : tx) : 0.0;
        uu[u] = kv[u] ? 0.15 * lean_exp(tu) : 1.0;
        lp += (kv[u] && B.is_pos) ? jac * tx : 0.0;
        if (k < g.KZ || kv[u]) xs[MAXBW + k] = x[u];      // zeros on [K, KZ): the forward product reads that far (g_Zhat aliases the row)
        if (kv[u]) us[2 + k] = uu[u];
    }
    double sc[9];
    const double st = th[2 * KS];
    const double sraw = lean_exp(st);
    lp += lane < 9 ? (lane < 6 ? -0.5 * sraw * sraw : -6.0 * st - 5.0 * lean_rcp(sraw)) + jac * st : 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) sc[i] = wv_bcast(sraw, i);
    wv_sync();
    BDRT_WV_PROF(0);

    // ---- E1: forward product partials; prior chain x -> L x -> w ---------------------------------------------------------
    {
        const int part = er.fpart, rg = lane - part * g.RGb;
        if (part < g.NP) {
            const int n0 = 4 * rg, m0 = part * g.ML;
            double ar[4] = {0.0, 0.0, 0.0, 0.0}, ai[4] = {0.0, 0.0, 0.0, 0.0};
            if constexpr (OCC == 1) wave_toep_fwd_pipe(fpre, xs + MAXBW + m0, g.ML, ar, ai);
            else wave_toep_fwd(gen, g.GQ, n0 - m0 + g.S, xs + MAXBW + m0, g.ML, ar, ai);
            double *o = zp + (size_t)(2 * part) * g.R4 + n0;
            *reinterpret_cast<double2 *>(o) = make_double2(ar[0], ar[1]); *reinterpret_cast<double2 *>(o + 2) = make_double2(ar[2], ar[3]);
            o += g.R4;
            *reinterpret_cast<double2 *>(o) = make_double2(ai[0], ai[1]); *reinterpret_cast<double2 *>(o + 2) = make_double2(ai[2], ai[3]);
        }
    }
    BDRT_WV_PROF(1);
    double gup[KS];
    double sv0 = 0.0, sv1 = 0.0, sv2 = 0.0;
    {
        const double d0 = sc[6], d1 = sc[7], d2 = sc[8];
        double v0[KS], v1[KS], v2[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) { v0[u] = 0.0; v1[u] = 0.0; v2[u] = 0.0; }
        {
            // The band coefficients come through an opaque scalar pointer and the taps in a ROLLED loop of CB-tap batches: left
            // visible / unrolled, the 39 coefficients are loaded once at the top of the kernel and kept (and spilled) for both
            // convolutions, and every LDS read of the phase is hoisted in front of the first FMA (hundreds of live registers).
            constexpr int NT13 = 2 * MAXBW + 1, CB = 4;
            wv_cptr Tc = (wv_cptr)&B.T[0][0];
            __asm__ volatile("" : "+s"(Tc));
            const double *xl = xs + (kv[0] ? lane : 0);                                            // (slots without an element read slot 0's taps)
            double xa[CB][KS], xb[CB][KS];
            auto ld = [&](double (&xv)[CB][KS], int d0) {
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int u = 0; u < KS; ++u) xv[j][u] = xl[(kv[u] ? 64 * u : 0) + d0 + j];      // x[k - MAXBW + d] (beyond the band: in-bounds, unused)
            };
            auto mac = [&](const double (&xv)[CB][KS], int d0) {
#pragma unroll
                for (int j = 0; j < CB; ++j) {
                    const int d = d0 + j, dd = d < NT13 ? d : NT13 - 1;
                    const bool in = d < NT13;
                    const double t0 = in ? Tc[dd] : 0.0, t1 = in ? Tc[NT13 + dd] : 0.0, t2 = in ? Tc[2 * NT13 + dd] : 0.0;
#pragma unroll
                    for (int u = 0; u < KS; ++u) {
                        v0[u] = fma(t0, xv[j][u], v0[u]); v1[u] = fma(t1, xv[j][u], v1[u]); v2[u] = fma(t2, xv[j][u], v2[u]);
                    }
                }
            };
            if constexpr (OCC == 1) {
            // one wave per SIMD: the coefficients come from the LDS table with the taps' operands (a scalar load returns out of order: its
            // s_waitcnt drains the LDS queue as well -- with no second wave to fill the gap, once per batch)
            const double *tcl = lds + g.o_tc;
            double ca[3][CB], cb[3][CB];
            auto ldc = [&](double (&cv)[3][CB], int d0) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < CB; ++j) cv[i][j] = tcl[16 * i + d0 + j];
            };
            auto macc = [&](const double (&xv)[CB][KS], const double (&cv)[3][CB]) {
#pragma unroll
                for (int j = 0; j < CB; ++j)
#pragma unroll
                    for (int u = 0; u < KS; ++u) {
                        v0[u] = fma(cv[0][j], xv[j][u], v0[u]); v1[u] = fma(cv[1][j], xv[j][u], v1[u]); v2[u] = fma(cv[2][j], xv[j][u], v2[u]);
                    }
            };
            ld(xa, 0); ldc(ca, 0);
#pragma unroll 1
            for (int d0 = 0; d0 < NT13; d0 += 2 * CB) {
                ld(xb, d0 + CB); ldc(cb, d0 + CB);
                macc(xa, ca);
                ld(xa, d0 + 2 * CB); ldc(ca, d0 + 2 * CB);
                macc(xb, cb);
            }
            } else {
            ld(xa, 0);
#pragma unroll 1
            for (int d0 = 0; d0 < NT13; d0 += 2 * CB) {
                ld(xb, d0 + CB);
                mac(xa, d0);
                ld(xa, d0 + 2 * CB);
                mac(xb, d0 + CB);
            }
            }
        }
        BDRT_WV_PROF(12);
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int k = lane + 64 * u;
            const int kc = kv[u] ? k : 0;
            const double um2 = us[kc], um1 = us[kc + 1], up1 = us[kc + 3], up2 = us[kc + 4];     // ups[k-2], [k-1], [k+1], [k+2]
            const double uk = uu[u], tu = th[KS + u];
            const double iu = lean_rcp(uk), iu2 = iu * iu;
            const double q2 = d0 * v0[u] * v0[u] + d1 * v1[u] * v1[u] + d2 * v2[u] * v2[u];
            const double ir = 0.15 * iu;                       // 1 / ups_raw
            double lpk = -(tu + LOG_015) - 0.5 * q2 * iu2 - (P.ups_alpha + 1.0) * tu - P.ups_beta * ir + jac * tu;
            double gu = -iu + q2 * iu2 * iu;
            // neighbour terms of the ups prior: centre k, and k as the right / left neighbour of the centres k - 1 / k + 1.  Every accumulation
            // is an explicit fma of the same operands in both schedules (OCC 1 and 2 give the same bits: chains change schedule between
            // launches when the number of live chains crosses four per CU).
            {
                const double s1 = um1 + up1;
                const double duc = 0.5 * (uk - 0.5 * s1) * iu;
                const bool cc = k >= 1 && k + 1 < K, cl = k >= 2 && kv[u], cr = k + 2 < K;
                if constexpr (OCC == 1) {
                    // one wave per SIMD: branch-free, so that the lane's slots interleave (4.8 k -> 3.5 k cycles for this phase; the 40
                    // registers it costs are what made it slower with two waves per SIMD).  The rows' pads hold 1.0: every reciprocal is finite.
                    lpk = fma(cc ? -0.5 * duc : 0.0, duc, lpk);
                    gu = fma(cc ? -duc * 0.25 * s1 : 0.0, iu2, gu);
                    const double il = lean_rcp(um1), ir1 = lean_rcp(up1);
                    const double dul = 0.5 * (um1 - 0.5 * (um2 + uk)) * il;
                    gu = fma(cl ? dul * 0.25 : 0.0, il, gu);
                    const double dur = 0.5 * (up1 - 0.5 * (uk + up2)) * ir1;
                    gu = fma(cr ? dur * 0.25 : 0.0, ir1, gu);
                } else {
                    if (cc) {
                        lpk = fma(-0.5 * duc, duc, lpk);
                        gu = fma(-duc * 0.25 * s1, iu2, gu);
                    }
                    if (cl) {
                        const double il = lean_rcp(um1);
                        const double dul = 0.5 * (um1 - 0.5 * (um2 + uk)) * il;
                        gu = fma(dul * 0.25, il, gu);
                    }
                    if (cr) {
                        const double ir1 = lean_rcp(up1);
                        const double dur = 0.5 * (up1 - 0.5 * (uk + up2)) * ir1;
                        gu = fma(dur * 0.25, ir1, gu);
                    }
                }
            }
            lp += kv[u] ? lpk : 0.0;
            sv0 += kv[u] ? v0[u] * v0[u] * iu2 : 0.0; sv1 += kv[u] ? v1[u] * v1[u] * iu2 : 0.0; sv2 += kv[u] ? v2[u] * v2[u] * iu2 : 0.0;
            gup[u] = uk * gu - (P.ups_alpha + 1.0) + P.ups_beta * ir + jac;
            if (kv[u]) {                                       // (the pads behind K stay zero)
                wr[MAXBW + k] = -d0 * v0[u] * iu2;
                wr[g.XL + MAXBW + k] = -d1 * v1[u] * iu2;
                wr[2 * g.XL + MAXBW + k] = -d2 * v2[u] * iu2;
            }
        }
    }
    double S0, S1, S2;
    {
        const double q[4] = {sv0, sv1, sv2, 0.0};
        double t = sum32_by_lane<4>(q, lane);
        t += wv_xor32(t);
        S0 = wv_bcast(t, 0); S1 = wv_bcast(t, 1); S2 = wv_bcast(t, 2);
    }
    wv_sync();
    BDRT_WV_PROF(2);

    // ---- E2: likelihood (rows n = l + 64 s), backward convolutions (own k) ----------------------------------------------------
    double T[7];
    {
        double sR = 0, sL = 0, sH = 0, sHz2 = 0, sHzr2 = 0, sHzi2 = 0;
        const double Rinf = 100.0 * sc[0], induc = sc[1] * P.induc_scale;
        const double s_res = 0.05 * sc[2], a_p = 0.05 * sc[3], a_r = 0.05 * sc[4], a_i = 0.05 * sc[5];
        const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
        const double ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = lane + 64 * s;
            const bool nv = n < nf;
            const int nn = nv ? n : 0;
            double zr = 0.0, zi = 0.0;
#pragma unroll 1
            for (int p = 0; p < g.NP; ++p) { zr += zp[(size_t)(2 * p) * g.R4 + nn]; zi += zp[(size_t)(2 * p + 1) * g.R4 + nn]; }
            zr += Rinf; zi += induc * er.wn[s];
            const double common = ar2 * zr * zr + ai2 * zi * zi;
            // outlier error model (formulas of bdrt_tile_hw.h): sigma_out enters the variances; its two parameters per row are this lane's
            [[maybe_unused]] double so_re = 0.0, so_im = 0.0, r0 = 0.0, r1 = 0.0, t0 = 0.0, t1 = 0.0;
            if constexpr (OM) {
                t0 = th[2 * KS + 1 + s]; t1 = th[2 * KS + 1 + NS + s];
                r0 = lean_exp(t0); r1 = lean_exp(t1);
                if (P.outlier_mode == 1) so_re = so_im = 0.05 * r0 * r1;
                else { so_re = 0.05 * r0; so_im = 0.05 * r1; }
            }
            double s2_re = c0 + ap2 * zr * zr + common, s2_im = c0 + ap2 * zi * zi + common;
            if constexpr (OM) { s2_re += so_re * so_re; s2_im += so_im * so_im; }
            const double e_re = er.zre[s] - zr, e_im = er.zim[s] - zi;
            const double prod = s2_re * s2_im, ip = lean_rcp(prod);
            const double w_re = s2_im * ip, w_im = s2_re * ip;
            const double lpn = -0.5 * lean_log(prod) - 0.5 * e_re * e_re * w_re - 0.5 * e_im * e_im * w_im;
            const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;
            const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
            const double gzr = e_re * w_re + 2.0 * zr * (h_re * (ap2 + ar2) + h_im * ar2);
            const double gzi = e_im * w_im + 2.0 * zi * (h_im * (ap2 + ai2) + h_re * ai2);
            if constexpr (OM) {
                double g0 = 0.0, g1 = 0.0, lpo = 0.0;
                if (P.outlier_mode == 1) {
                    const double dso = 2.0 * so_re * (h_re + h_im), ir1 = lean_rcp(r1);
                    g0 = r0 * (0.05 * r1 * dso - P.so_lambda) + jac;
                    g1 = 0.05 * r0 * r1 * dso - (P.so_alpha + 1.0) + P.so_beta * ir1 + jac;
                    lpo = -P.so_lambda * r0 - (P.so_alpha + 1.0) * t1 - P.so_beta * ir1 + jac * (t0 + t1);
                } else {
                    g0 = r0 * (0.05 * 2.0 * so_re * h_re - P.so_lambda) + jac;
                    g1 = r1 * (0.05 * 2.0 * so_im * h_im - P.so_lambda) + jac;
                    lpo = -P.so_lambda * (r0 + r1) + jac * (t0 + t1);
                }
                gr[2 * KS + 1 + s] = nv ? g0 : 0.0; gr[2 * KS + 1 + NS + s] = nv ? g1 : 0.0;
                lp += nv ? lpo : 0.0;
            }
            if (nv) {                                          // (zeros from the set-up on [nf, NLP): the backward product's padding)
                gz[n] = gzr; gz[g.NLP + n] = gzi;
                lp += lpn;
                sR += gzr; sL += gzi * er.wn[s]; sH += h_re + h_im; sHz2 += h_re * zr * zr + h_im * zi * zi;
                sHzr2 += (h_re + h_im) * zr * zr; sHzi2 += (h_re + h_im) * zi * zi;
            }
        }
        BDRT_WV_PROF(6);
        // the six sums and lp in one butterfly: lane j of each half-wave ends with sum j
        const double q[8] = {sR, sL, sH, sHz2, sHzr2, sHzi2, lp, 0.0};
        double t = sum32_by_lane<8>(q, lane);
        t += wv_xor32(t);
#pragma unroll
        for (int i = 0; i < 7; ++i) T[i] = wv_bcast(t, i);
        BDRT_WV_PROF(7);
    }
    [[maybe_unused]] ToepBwdPre bpre;
    if constexpr (OCC == 1) {
        // the backward product's first generator elements: requested here, used behind the backward convolutions
        const int bmg_c = (lane & 31) < g.MG ? (lane & 31) : g.MG - 1;
        bpre = wave_toep_bwd_pre(gen + (lane >> 5) * 4 * g.GQ, g.GQ, g.S - 6 * bmg_c);
    }
    double gl[KS];
    {
#pragma unroll
        for (int u = 0; u < KS; ++u) gl[u] = 0.0;
        constexpr int NT13 = 2 * MAXBW + 1, CA = 7, CBB = NT13 - CA;
        wv_cptr Tc = (wv_cptr)&B.T[0][0];
        __asm__ volatile("" : "+s"(Tc));
        // per band row: taps 0..6, then 7..12; the other half (the next row's first) is requested before each half's FMAs
        const double *wl = wr + 2 * MAXBW;
        int ko[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) ko[u] = kv[u] ? lane + 64 * u : 0;
        double wa[CA][KS], wb[CBB][KS];
#pragma unroll
        for (int j = 0; j < CA; ++j)
#pragma unroll
            for (int u = 0; u < KS; ++u) wa[j][u] = wl[ko[u] - j];                                  // w_i[k + MAXBW - d]
#pragma unroll 1
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < CBB; ++j)
#pragma unroll
                for (int u = 0; u < KS; ++u) wb[j][u] = wl[ko[u] - (CA + j)];
            [[maybe_unused]] double cr[16];           // OCC 1: the row's coefficients from the LDS table (see the forward convolutions)
            if constexpr (OCC == 1) {
#pragma unroll
                for (int j = 0; j < 16; ++j) cr[j] = (lds + g.o_tc)[16 * i + j];
            }
#pragma unroll
            for (int j = 0; j < CA; ++j) {
                double c;
                if constexpr (OCC == 1) c = cr[j]; else c = Tc[i * NT13 + j];
#pragma unroll
                for (int u = 0; u < KS; ++u) gl[u] = fma(c, wa[j][u], gl[u]);
            }
            wl += g.XL;                                                                            // (row 3: the generator table, unused)
#pragma unroll
            for (int j = 0; j < CA; ++j)
#pragma unroll
                for (int u = 0; u < KS; ++u) wa[j][u] = wl[ko[u] - j];
#pragma unroll
            for (int j = 0; j < CBB; ++j) {
                double c;
                if constexpr (OCC == 1) c = cr[CA + j]; else c = Tc[i * NT13 + CA + j];
#pragma unroll
                for (int u = 0; u < KS; ++u) gl[u] = fma(c, wb[j][u], gl[u]);
            }
        }
    }
    wv_sync();
    BDRT_WV_PROF(3);

    // ---- E3: backward product; scalar gradients ---------------------------------------------------------------------------------
    double *gk = zp;                                       // [2][KP] (the forward partials are dead)
    {
        const int b = lane >> 5, mg = lane & 31;
        if (mg < g.MG) {
            double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            if constexpr (OCC == 1) wave_toep_bwd_pipe(bpre, gz + b * g.NLP, g.NLP, acc);
            else wave_toep_bwd(gen + b * 4 * g.GQ, g.GQ, g.S - 6 * mg, gz + b * g.NLP, g.NLP, acc);
            double *o = gk + b * g.KP + 6 * mg;
            *reinterpret_cast<double2 *>(o) = make_double2(acc[0], acc[1]);
            *reinterpret_cast<double2 *>(o + 2) = make_double2(acc[2], acc[3]);
            *reinterpret_cast<double2 *>(o + 4) = make_double2(acc[4], acc[5]);
        }
    }
    {
        // d lp / d(raw), likelihood part, of Rinf_raw, induc_raw, sigma_res_raw, alpha_prop/re/im_raw; the d strengths
        double gsc = 0.0;
        if (lane < 6) {
            const double t = lane == 0 ? T[0] : lane == 1 ? T[1] : lane == 2 ? T[2] : lane == 3 ? T[3] : lane == 4 ? T[4] : T[5];
            double dl;
            if (lane == 0) dl = 100.0 * t;
            else if (lane == 1) dl = P.induc_scale * t;
            else dl = 0.05 * 2.0 * (0.05 * sraw) * t;
            gsc = sraw * (dl - sraw) + jac;
        } else if (lane < 9) {
            const double svs = lane == 6 ? S0 : lane == 7 ? S1 : S2;
            gsc = -0.5 * sraw * svs - 6.0 + 5.0 * lean_rcp(sraw) + jac;
        }
        gr[2 * KS] = gsc;
    }
    wv_sync();
    BDRT_WV_PROF(4);

    // ---- E4: chain rule through x = exp(theta_x) -------------------------------------------------------------------------------------
#pragma unroll
    for (int u = 0; u < KS; ++u) {
        const int k = lane + 64 * u;
        const int kk = kv[u] ? k : 0;
        const double graw = gl[u] + gk[kk] + gk[g.KP + kk];
        gr[u] = kv[u] ? (B.is_pos ? x[u] * graw + jac : graw) : 0.0;
        gr[KS + u] = kv[u] ? gup[u] : 0.0;
    }
    BDRT_WV_PROF(5);
#undef BDRT_WV_PROF
    return T[6];
}

// ---- several distributions ------------------------------------------------------------------------------------------------------------
// NB blocks of one basis length (series or parallel, each with its own generator table, band coefficients, sign constraint and scale),
// the x_sum prior of the mixed models, the outlier error models: the formulas of bdrt_tile_hw.h / bdrt_solo_wide.h on the lane mapping of
// wave_eval.  Slots of a lane: per block KS x-slots and KS ups-slots, then the scalars (lanes 0..5 the global ones, lane 6 + 3 b + i the
// penalty strength d_i of block b), then the outlier parameters.  One wave per SIMD (512 registers).
template <int KS, int NS, bool OM, int NB> constexpr int wave_slots_nb() { return 2 * KS * NB + 1 + (OM ? 2 * NS : 0); }

template <int KS, int NS, bool OM, int NB>
__device__ __forceinline__ int wave_slot_index_nb(const DevProblem &P, int K, int u, int lane)
{
    if (u < 2 * KS * NB) {
        const int b = u / (2 * KS), r = u - b * 2 * KS;
        const DevBlock &B = P.blk[b];
        if (r < KS) return lane + 64 * r < K ? B.o_x + lane + 64 * r : -1;
        return lane + 64 * (r - KS) < K ? B.o_ups + lane + 64 * (r - KS) : -1;
    }
    if (u == 2 * KS * NB)
        return lane < 2 ? lane : (lane < 6 ? P.o_err + (lane - 2) : (lane < 6 + 3 * NB ? P.blk[(lane - 6) / 3].o_d + (lane - 6) % 3 : -1));
    if constexpr (OM) {
        const int s = u - (2 * KS * NB + 1), h = s >= NS ? 1 : 0, n = lane + 64 * (s - h * NS);
        return n < P.nf ? P.o_so + h * P.nf + n : -1;
    }
    return -1;
}

template <int KS, int NS, bool OM, int NB>
__device__ __forceinline__ double wave_eval_nb(const DevProblem &P, const WaveGeom &g, double *lds, const double (&th)[wave_slots_nb<KS, NS, OM, NB>()],
                                               double (&gr)[wave_slots_nb<KS, NS, OM, NB>()], const WaveEvalRegs<NS> &er, const double jac, int lane_)
{
    int lane = lane_;
    __asm__ volatile("" : "+v"(lane));
    constexpr int SC = 2 * KS * NB, SO = SC + 1;
    const int nf = g.nf, K = g.K;
    double *xs = lds + g.o_xs, *us = lds + g.o_us, *wr = lds + g.o_w;
    double *zp = lds + g.o_zp, *gz = lds + g.o_gz;
    double lp = 0.0;

    // ---- scalars --------------------------------------------------------------------------------------------------------------
    const double st = th[SC];
    const double sraw = lean_exp(st);
    lp += lane < 6 + 3 * NB ? (lane < 6 ? -0.5 * sraw * sraw : -6.0 * st - 5.0 * lean_rcp(sraw)) + jac * st : 0.0;
    double sc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) sc[i] = wv_bcast(sraw, i);

    double xk[NB][KS], gup[NB][KS], gl[NB][KS], yr[NB][NS], yi[NB][NS], Ssum[NB][3];
    bool kv[KS];
#pragma unroll
    for (int u = 0; u < KS; ++u) kv[u] = lane + 64 * u < K;
    double xsum_p = 0.0;

    // ================================================= forward, block by block ===========================================================
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const DevBlock &B = P.blk[b];
        const double *gen = lds + g.o_gen + b * 2 * 4 * g.GQ;
        double uu[KS];
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int k = lane + 64 * u;
            const double tx = th[b * 2 * KS + u], tu = th[b * 2 * KS + KS + u];
            const double ex = lean_exp(tx);
            xk[b][u] = kv[u] ? (B.is_pos ? ex : tx) : 0.0;
            uu[u] = kv[u] ? 0.15 * lean_exp(tu) : 1.0;
            lp += (kv[u] && B.is_pos) ? jac * tx : 0.0;
            xsum_p += xk[b][u];
            if (k < g.KZ || kv[u]) xs[MAXBW + k] = xk[b][u];
            if (kv[u]) us[2 + k] = uu[u];
        }
        wv_sync();
        // forward product partials (A_b x_raw; the block's scale is applied to the sums)
        {
            const int part = er.fpart, rg = lane - part * g.RGb;
            if (part < g.NP) {
                const int n0 = 4 * rg, m0 = part * g.ML;
                double ar[4] = {0.0, 0.0, 0.0, 0.0}, ai[4] = {0.0, 0.0, 0.0, 0.0};
                wave_toep_fwd_pipe(wave_toep_fwd_pre(gen, g.GQ, n0 - m0 + g.S), xs + MAXBW + m0, g.ML, ar, ai);      // (one wave per SIMD: see wave_eval, OCC = 1)
                double *o = zp + (size_t)(2 * part) * g.R4 + n0;
                *reinterpret_cast<double2 *>(o) = make_double2(ar[0], ar[1]); *reinterpret_cast<double2 *>(o + 2) = make_double2(ar[2], ar[3]);
                o += g.R4;
                *reinterpret_cast<double2 *>(o) = make_double2(ai[0], ai[1]); *reinterpret_cast<double2 *>(o + 2) = make_double2(ai[2], ai[3]);
            }
        }
        // prior chain x -> L x -> w (as wave_eval)
        double sv0 = 0.0, sv1 = 0.0, sv2 = 0.0;
        {
            const double d0 = wv_bcast(sraw, 6 + 3 * b), d1 = wv_bcast(sraw, 7 + 3 * b), d2 = wv_bcast(sraw, 8 + 3 * b);
            double v0[KS], v1[KS], v2[KS];
#pragma unroll
            for (int u = 0; u < KS; ++u) { v0[u] = 0.0; v1[u] = 0.0; v2[u] = 0.0; }
            {
                constexpr int NT13 = 2 * MAXBW + 1, CB = 4;
                wv_cptr Tc = (wv_cptr)&B.T[0][0];
                __asm__ volatile("" : "+s"(Tc));
                const double *xl = xs + (kv[0] ? lane : 0);
                double xa[CB][KS], xb[CB][KS];
                auto ld = [&](double (&xv)[CB][KS], int dd0) {
#pragma unroll
                    for (int j = 0; j < CB; ++j)
#pragma unroll
                        for (int u = 0; u < KS; ++u) xv[j][u] = xl[(kv[u] ? 64 * u : 0) + dd0 + j];
                };
                // (the band coefficients from the block's LDS table, rows of 16 with zeros behind tap 12: a scalar load's s_waitcnt would drain
                //  the LDS queue once per batch -- see wave_eval, OCC = 1)
                (void)Tc;
                const double *tcl = lds + g.o_tc + b * WV_TCL;
                double ca[3][CB], cb[3][CB];
                auto ldc = [&](double (&cv)[3][CB], int dd0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < CB; ++j) cv[i][j] = tcl[16 * i + dd0 + j];
                };
                auto mac = [&](const double (&xv)[CB][KS], const double (&cv)[3][CB]) {
#pragma unroll
                    for (int j = 0; j < CB; ++j)
#pragma unroll
                        for (int u = 0; u < KS; ++u) {
                            v0[u] = fma(cv[0][j], xv[j][u], v0[u]); v1[u] = fma(cv[1][j], xv[j][u], v1[u]); v2[u] = fma(cv[2][j], xv[j][u], v2[u]);
                        }
                };
                ld(xa, 0); ldc(ca, 0);
#pragma unroll 1
                for (int dd0 = 0; dd0 < NT13; dd0 += 2 * CB) {
                    ld(xb, dd0 + CB); ldc(cb, dd0 + CB);
                    mac(xa, ca);
                    ld(xa, dd0 + 2 * CB); ldc(ca, dd0 + 2 * CB);
                    mac(xb, cb);
                }
            }
#pragma unroll
            for (int u = 0; u < KS; ++u) {
                const int k = lane + 64 * u;
                const int kc = kv[u] ? k : 0;
                const double um2 = us[kc], um1 = us[kc + 1], up1 = us[kc + 3], up2 = us[kc + 4];
                const double uk = uu[u], tu = th[b * 2 * KS + KS + u];
                const double iu = lean_rcp(uk), iu2 = iu * iu;
                const double q2 = d0 * v0[u] * v0[u] + d1 * v1[u] * v1[u] + d2 * v2[u] * v2[u];
                const double ir = 0.15 * iu;
                double lpk = -(tu + LOG_015) - 0.5 * q2 * iu2 - (P.ups_alpha + 1.0) * tu - P.ups_beta * ir + jac * tu;
                double gu = -iu + q2 * iu2 * iu;
                if (k >= 1 && k + 1 < K) {
                    const double du = 0.5 * (uk - 0.5 * (um1 + up1)) * iu;
                    lpk += -0.5 * du * du;
                    gu += -du * 0.25 * (um1 + up1) * iu2;
                }
                if (k >= 2 && kv[u]) {
                    const double i0 = lean_rcp(um1);
                    const double du = 0.5 * (um1 - 0.5 * (um2 + uk)) * i0;
                    gu += du * 0.25 * i0;
                }
                if (k + 2 < K) {
                    const double i0 = lean_rcp(up1);
                    const double du = 0.5 * (up1 - 0.5 * (uk + up2)) * i0;
                    gu += du * 0.25 * i0;
                }
                lp += kv[u] ? lpk : 0.0;
                sv0 += kv[u] ? v0[u] * v0[u] * iu2 : 0.0; sv1 += kv[u] ? v1[u] * v1[u] * iu2 : 0.0; sv2 += kv[u] ? v2[u] * v2[u] * iu2 : 0.0;
                gup[b][u] = uk * gu - (P.ups_alpha + 1.0) + P.ups_beta * ir + jac;
                if (kv[u]) {
                    wr[MAXBW + k] = -d0 * v0[u] * iu2;
                    wr[g.XL + MAXBW + k] = -d1 * v1[u] * iu2;
                    wr[2 * g.XL + MAXBW + k] = -d2 * v2[u] * iu2;
                }
            }
        }
        {
            const double q[4] = {sv0, sv1, sv2, 0.0};
            double t = sum32_by_lane<4>(q, lane);
            t += wv_xor32(t);
            Ssum[b][0] = wv_bcast(t, 0); Ssum[b][1] = wv_bcast(t, 1); Ssum[b][2] = wv_bcast(t, 2);
        }
        wv_sync();
        // Y_b at this lane's rows; sum_i L_i^T w_i at this lane's k
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = lane + 64 * s, nn = n < nf ? n : 0;
            double zr = 0.0, zi = 0.0;
#pragma unroll 1
            for (int p = 0; p < g.NP; ++p) { zr += zp[(size_t)(2 * p) * g.R4 + nn]; zi += zp[(size_t)(2 * p + 1) * g.R4 + nn]; }
            yr[b][s] = zr * B.x_scale; yi[b][s] = zi * B.x_scale;
        }
        {
#pragma unroll
            for (int u = 0; u < KS; ++u) gl[b][u] = 0.0;
            constexpr int NT13 = 2 * MAXBW + 1;
            const double *tcl = lds + g.o_tc + b * WV_TCL;
#pragma unroll 1
            for (int i = 0; i < 3; ++i) {
                const double *wl = wr + i * g.XL + 2 * MAXBW;
                double cr[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) cr[j] = tcl[16 * i + j];
#pragma unroll
                for (int j = 0; j < NT13; ++j) {
                    const double c = cr[j];
#pragma unroll
                    for (int u = 0; u < KS; ++u) gl[b][u] = fma(c, wl[(kv[u] ? lane + 64 * u : 0) - j], gl[b][u]);
                }
            }
        }
        wv_sync();                                         // (the next block reuses the x, ups, w rows and the partial sums)
    }

    // ================================================= x_sum prior, likelihood ===========================================================
    double xs_term = 0.0;
    bool rej = false;
    if (P.use_x_sum) {
        const double xs_raw = wv_sum(xsum_p);
        const double xsn = xs_raw * P.x_sum_invscale;
        lp += lane == 0 ? -0.5 * xsn * xsn : 0.0;
        rej = xs_raw < 0.0;
        xs_term = -xs_raw * P.x_sum_invscale * P.x_sum_invscale;
    }
    double T[7], gzr_[NS], gzi_[NS];
    {
        double sR = 0, sL = 0, sH = 0, sHz2 = 0, sHzr2 = 0, sHzi2 = 0;
        const double Rinf = 100.0 * sc[0], induc = sc[1] * P.induc_scale;
        const double s_res = 0.05 * sc[2], a_p = 0.05 * sc[3], a_r = 0.05 * sc[4], a_i = 0.05 * sc[5];
        const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
        const double ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = lane + 64 * s;
            const bool nv = n < nf;
            double zr = Rinf, zi = induc * er.wn[s];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (!P.blk[b].is_parallel) { zr += yr[b][s]; zi += yi[b][s]; }
                else {
                    const double idn = lean_rcp(yr[b][s] * yr[b][s] + yi[b][s] * yi[b][s]);
                    zr += yr[b][s] * idn;                   // Z_hat_p = conj(Y) / |Y|^2 (Parallel_modelcode.txt:47)
                    zi += -yi[b][s] * idn;
                }
            }
            const double common = ar2 * zr * zr + ai2 * zi * zi;
            [[maybe_unused]] double so_re = 0.0, so_im = 0.0, r0 = 0.0, r1 = 0.0, t0 = 0.0, t1 = 0.0;
            if constexpr (OM) {
                t0 = th[SO + s]; t1 = th[SO + NS + s];
                r0 = lean_exp(t0); r1 = lean_exp(t1);
                if (P.outlier_mode == 1) so_re = so_im = 0.05 * r0 * r1;
                else { so_re = 0.05 * r0; so_im = 0.05 * r1; }
            }
            double s2_re = c0 + ap2 * zr * zr + common, s2_im = c0 + ap2 * zi * zi + common;
            if constexpr (OM) { s2_re += so_re * so_re; s2_im += so_im * so_im; }
            const double e_re = er.zre[s] - zr, e_im = er.zim[s] - zi;
            const double prod = s2_re * s2_im, ip = lean_rcp(prod);
            const double w_re = s2_im * ip, w_im = s2_re * ip;
            const double lpn = -0.5 * lean_log(prod) - 0.5 * e_re * e_re * w_re - 0.5 * e_im * e_im * w_im;
            const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;
            const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
            const double gzr = e_re * w_re + 2.0 * zr * (h_re * (ap2 + ar2) + h_im * ar2);
            const double gzi = e_im * w_im + 2.0 * zi * (h_im * (ap2 + ai2) + h_re * ai2);
            gzr_[s] = nv ? gzr : 0.0; gzi_[s] = nv ? gzi : 0.0;
            if constexpr (OM) {
                double g0 = 0.0, g1 = 0.0, lpo = 0.0;
                if (P.outlier_mode == 1) {
                    const double dso = 2.0 * so_re * (h_re + h_im), ir1 = lean_rcp(r1);
                    g0 = r0 * (0.05 * r1 * dso - P.so_lambda) + jac;
                    g1 = 0.05 * r0 * r1 * dso - (P.so_alpha + 1.0) + P.so_beta * ir1 + jac;
                    lpo = -P.so_lambda * r0 - (P.so_alpha + 1.0) * t1 - P.so_beta * ir1 + jac * (t0 + t1);
                } else {
                    g0 = r0 * (0.05 * 2.0 * so_re * h_re - P.so_lambda) + jac;
                    g1 = r1 * (0.05 * 2.0 * so_im * h_im - P.so_lambda) + jac;
                    lpo = -P.so_lambda * (r0 + r1) + jac * (t0 + t1);
                }
                gr[SO + s] = nv ? g0 : 0.0; gr[SO + NS + s] = nv ? g1 : 0.0;
                lp += nv ? lpo : 0.0;
            }
            if (nv) {
                lp += lpn;
                sR += gzr; sL += gzi * er.wn[s]; sH += h_re + h_im; sHz2 += h_re * zr * zr + h_im * zi * zi;
                sHzr2 += (h_re + h_im) * zr * zr; sHzi2 += (h_re + h_im) * zi * zi;
            }
        }
        const double q[8] = {sR, sL, sH, sHz2, sHzr2, sHzi2, lp, 0.0};
        double t = sum32_by_lane<8>(q, lane);
        t += wv_xor32(t);
#pragma unroll
        for (int i = 0; i < 7; ++i) T[i] = wv_bcast(t, i);
    }

    // ================================================= backward, block by block ==========================================================
    double *gk = zp;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const DevBlock &B = P.blk[b];
        const double *gen = lds + g.o_gen + b * 2 * 4 * g.GQ;
        // operand of A_b^T: g_Zhat, or J^T g_Zhat through Z_hat_p = conj(Y)/|Y|^2 (times the block's scale) for a parallel block
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int n = lane + 64 * s;
            double rr = gzr_[s], ri = gzi_[s];
            if (B.is_parallel) {
                const double a = yr[b][s], c = yi[b][s];
                const double dn = a * a + c * c, id2 = lean_rcp(dn * dn);
                const double dd = (c * c - a * a) * id2, doff = 2.0 * a * c * id2;
                rr = gzr_[s] * dd + gzi_[s] * doff;
                ri = -gzr_[s] * doff + gzi_[s] * dd;
            }
            if (n < nf) { gz[n] = rr * B.x_scale; gz[g.NLP + n] = ri * B.x_scale; }
        }
        wv_sync();
        {
            const int h = lane >> 5, mg = lane & 31;
            if (mg < g.MG) {
                double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
                wave_toep_bwd_pipe(wave_toep_bwd_pre(gen + h * 4 * g.GQ, g.GQ, g.S - 6 * mg), gz + h * g.NLP, g.NLP, acc);
                double *o = gk + h * g.KP + 6 * mg;
                *reinterpret_cast<double2 *>(o) = make_double2(acc[0], acc[1]);
                *reinterpret_cast<double2 *>(o + 2) = make_double2(acc[2], acc[3]);
                *reinterpret_cast<double2 *>(o + 4) = make_double2(acc[4], acc[5]);
            }
        }
        wv_sync();
#pragma unroll
        for (int u = 0; u < KS; ++u) {
            const int k = lane + 64 * u;
            const int kk = kv[u] ? k : 0;
            const double graw = gl[b][u] + gk[kk] + gk[g.KP + kk] + xs_term;
            gr[b * 2 * KS + u] = kv[u] ? (B.is_pos ? xk[b][u] * graw + jac : graw) : 0.0;
            gr[b * 2 * KS + KS + u] = kv[u] ? gup[b][u] : 0.0;
        }
        wv_sync();                                         // (the next block's operand and product reuse gz and the partial sums)
    }
    {
        double gsc = 0.0;
        if (lane < 6) {
            const double t = lane == 0 ? T[0] : lane == 1 ? T[1] : lane == 2 ? T[2] : lane == 3 ? T[3] : lane == 4 ? T[4] : T[5];
            double dl;
            if (lane == 0) dl = 100.0 * t;
            else if (lane == 1) dl = P.induc_scale * t;
            else dl = 0.05 * 2.0 * (0.05 * sraw) * t;
            gsc = sraw * (dl - sraw) + jac;
        } else if (lane < 6 + 3 * NB) {
            double svs = 0.0;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 3; ++i) svs = lane == 6 + 3 * b + i ? Ssum[b][i] : svs;
            gsc = -0.5 * sraw * svs - 6.0 + 5.0 * lean_rcp(sraw) + jac;
        }
        gr[SC] = gsc;
    }
    return rej ? -INFINITY : T[6];
}

}  // namespace bdrt
