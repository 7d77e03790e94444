// bdrt_tile_s1.h -- fast evaluator for the headline model family S1 (Series / Series_pos, one DRT block, no outlier
// parameters) on log-uniform grids: Series_modelcode.txt / Series_pos_modelcode.txt.
//
// Differences from the generic tile (bdrt_device.h):
//   * chain c of the workgroup is owned by ONE half-wave (wave c/2, lanes 32*(c%2)..+31): every per-chain reduction is
//     a 5-step xor shuffle, every per-chain scalar lives in registers, the ups neighbour coupling is a lane shuffle.
//     The only workgroup barriers left are the four around the two MFMA GEMMs (X ready, A.x ready, g_Zhat ready,
//     A^T g ready) instead of ~24;
//   * theta / gradient rows are read and written coalesced (lane = element index), which is also the layout of the
//     device-resident NUTS state, so the sampler and the evaluator share the thread mapping;
//   * L0, L1, L2 products are 17-tap convolutions on the chain's own LDS column (no cross-wave dependency at all);
//   * LDS tiles are [row][16] with the column XOR-swizzled by (row & 15): conflict-free for the MFMA operand reads
//     (16 lanes = one row), 2-way for the column walks of the element-wise code.
// Numerics are the same formulas as the generic tile; parity is asserted against the oracle and against the generic
// tile by the GPU tests.
#pragma once
#include "bdrt_device.h"
#include <type_traits>

namespace bdrt {

// debug only (tools/tile_trace.py): per-wave timestamps of the phase boundaries, [workgroup][wave][16]
static __device__ long long *g_tile_trace = nullptr;

__device__ __forceinline__ int swz(int row, int col) { return row * NC + (col ^ (row & 15)); }

// sum over the LPC lanes (32: half-wave, 64: wave) that own one chain
template <int LPC>
__device__ __forceinline__ double hsum(double x)
{
    x = sum32(x);
    if (LPC == 64) x += __shfl_xor(x, 32);
    return x;
}

// Orders the LDS traffic of one wave: lanes of a half-wave exchange data through the chain's private LDS row without a
// workgroup barrier.  The hardware executes a wave's LDS instructions in order; this only stops the compiler from moving
// a lane's loads/stores across the exchange point (to a single thread they look independent).
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Os[(16 t + i)][c] = sum_k M[16 t + i][k] Bs[k][c] for the tiles t = wave, wave + 8, ... of this wave (swizzled LDS tiles).
// The packed fragments (bdrt_model.hip::pack_fragments) stream from L2 through two register buffers of GPF operand pairs:
// while the 2*GPF MFMAs of one chunk issue (~GPF * 128 cycles) the next chunk -- of this tile or of the wave's next tile --
// is in flight.  The steady-state loop has no conditional loads, so the compiler's s_waitcnt vmcnt counts stay exact
// (a conditional load forces vmcnt(0) at the merge point, which serialises the stream).
template <int NWV, int GPF>
__device__ __forceinline__ void gemm_sw(const double *__restrict__ Mp, int ntiles, int pairs, const double *Bs, double *Os,
                                        int wave, int lane)
{
    typedef double dv2 __attribute__((ext_vector_type(2)));
    // global (not flat) loads: a flat load also counts on lgkmcnt and would serialise with the LDS operand reads
    typedef const __attribute__((address_space(1))) dv2 *gp2;
    const int col = lane & 15, kq = lane >> 4;
    wave = __builtin_amdgcn_readfirstlane(wave);          // uniform: scalar loop control and address arithmetic
    if (wave >= ntiles) return;
    const int nt = (ntiles - wave + NWV - 1) / NWV;
    const int nchunk = pairs / GPF, rem = pairs - nchunk * GPF;
    const int items = nt * nchunk;                       // chunks of this wave, tile-major
    gp2 base = (gp2) reinterpret_cast<const dv2 *>(Mp) + lane;
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    int l_ti = 0, l_ch = 0;                              // position of the load stream
    int c_ti = 0, c_ch = 0;                              // position of the compute stream
    // LDS address of the B operand of operand pair p: rows 8p + kq and 8p + kq + 4, column col swizzled by (row & 15),
    // which only depends on the parity of p: four lane-constant bases + p * 128 doubles
    const double *bX[2], *bY[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        bX[e] = Bs + kq * NC + (col ^ (8 * e + kq));
        bY[e] = Bs + (kq + 4) * NC + (col ^ (8 * e + kq + 4));
    }
    auto load = [&](dv2 (&buf)[GPF]) {
        gp2 p = base + ((size_t)(wave + l_ti * NWV) * pairs + (size_t)l_ch * GPF) * 64;
#pragma unroll
        for (int i = 0; i < GPF; ++i) buf[i] = p[(size_t)i * 64];
        if (++l_ch == nchunk) { l_ch = 0; ++l_ti; }
    };
    // one chunk: B operands of the chunk from LDS, first MFMA pair, THEN the global loads of the next chunk, then the other
    // MFMAs.  (The compiler waits with vmcnt(0) before the first MFMA that reads `cur`; issuing the next chunk's loads
    // after that point keeps them in flight behind ~800 cycles of MFMA work instead of being waited for immediately.)
    auto step = [&](const dv2 (&cur)[GPF], dv2 (&nxt)[GPF], bool prefetch) {
        const int p0 = c_ch * GPF;
        const bool odd = p0 & 1;
        const double *x0 = (odd ? bX[1] : bX[0]) + p0 * (8 * NC), *x1 = (odd ? bX[0] : bX[1]) + p0 * (8 * NC);
        const double *y0 = (odd ? bY[1] : bY[0]) + p0 * (8 * NC), *y1 = (odd ? bY[0] : bY[1]) + p0 * (8 * NC);
        double bx[GPF], by[GPF];
#pragma unroll
        for (int i = 0; i < GPF; ++i) {
            bx[i] = ((i & 1) ? x1 : x0)[i * (8 * NC)];
            by[i] = ((i & 1) ? y1 : y0)[i * (8 * NC)];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = mfma_f64(cur[0].x, bx[0], acc0);
        acc1 = mfma_f64(cur[0].y, by[0], acc1);
        __builtin_amdgcn_sched_barrier(0);
        if (prefetch) load(nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 1; i < GPF; ++i) {
            acc0 = mfma_f64(cur[i].x, bx[i], acc0);
            acc1 = mfma_f64(cur[i].y, by[i], acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (++c_ch == nchunk) {
            const int t = wave + c_ti * NWV;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) { const int row = 16 * t + kq + 4 * rr; Os[swz(row, col)] = acc0[rr] + acc1[rr]; }
            acc0 = d4{0.0, 0.0, 0.0, 0.0}; acc1 = d4{0.0, 0.0, 0.0, 0.0};
            c_ch = 0; ++c_ti;
        }
    };
    dv2 a[GPF], b[GPF];
    if (items > 0) load(a);
    for (int it = 0; it < items; it += 2) {
        step(a, b, it + 1 < items);
        if (it + 1 < items) step(b, a, it + 2 < items);
    }
    // operand pairs beyond the last full chunk (pairs % GPF; none for the 81 x 161 shape) and tiles narrower than a chunk
    if (rem > 0) {
        for (int ti = 0; ti < nt; ++ti) {
            const int t = wave + ti * NWV;
            d4 r0a = {0.0, 0.0, 0.0, 0.0}, r1a = {0.0, 0.0, 0.0, 0.0};
            for (int p = nchunk * GPF; p < pairs; ++p) {
                const dv2 v = base[((size_t)t * pairs + p) * 64];
                const int r0 = 8 * p + kq, r1 = r0 + 4;
                r0a = mfma_f64(v.x, Bs[swz(r0, col)], r0a);
                r1a = mfma_f64(v.y, Bs[swz(r1, col)], r1a);
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = 16 * t + kq + 4 * rr;
                const double prev = nchunk > 0 ? Os[swz(row, col)] : 0.0;
                Os[swz(row, col)] = prev + r0a[rr] + r1a[rr];
            }
        }
    }
}

constexpr int RW = 32 * UK + 2 * MAXBW;   // length of a chain's private LDS row (k = 0..191 plus the convolution halo)
constexpr int NTAP = 2 * MAXBW + 1;
constexpr int WIN = UK + NTAP - 1;        // 22 values feed the 17-tap convolution of six consecutive k

// ---- GEMMs with the A operand from an LDS-resident generator table (DevProblem::toepA) -----------------------------------------
// A_re and A_im are exactly Toeplitz (equal log spacing of the frequencies and of tau): A_part[n][k] = tg[part][n - k + K - 1].
// The table Tt[part * tlen + 8 + d] = tg[part][d] (zeros in front and behind) replaces the packed fragment stream of gemm_sw:
//   * nothing streams from L2 during the GEMMs (473 KB per GEMM and workgroup before; the stream ran the MFMA pipe at 70 %:
//     126 MFMAs of a SIMD took 11.4 k cycles instead of 8.1 k, tools/tile_trace.py);
//   * any wave can compute any (tile, k-range): the two parts tile separately (81 rows = 5 tiles + 1 row each, the odd rows are
//     dot products on the VALU while their operands are in registers anyway), 161 = 10 tiles + 1 row likewise, and 10 tiles
//     spread over the four SIMDs as 2.5 each -- eight whole tiles, two split in halves of the reduction range.  The two
//     halves of a tile accumulate into zeroed cells with LDS atomics: two contributions commute exactly (0 + a + b = 0 + b + a),
//     so the result does not depend on their order.
// Lane l of a wave holds row i = l & 15, reduction index kq = l >> 4 of the A operand, column l & 15 of the B operand.
// rows of the A x / g region Zh; leading zeros of a part of the generator table; first row of the imaginary part in Zh
// the instantiation for K basis functions; the sampler's elements per lane for D parameters (D = 2 K + 9 without outlier
// parameters: s1_ku(K) <= the KU that goes with s1_nj(D), bdrt_nuts.hip)
__host__ __device__ inline int s1_ku(int K) { return K <= 64 ? 2 : (K <= 96 ? 3 : (K <= 128 ? 4 : 6)); }
__host__ __device__ inline int s1_nj(int D) { return D <= 32 * 4 ? 4 : (D <= 32 * 6 ? 6 : (D <= 32 * 7 ? 7 : (D <= 32 * 11 ? 11 : 16))); }
__host__ __device__ inline int s1_zrows(const DevProblem &P) { return P.toepA == 2 ? P.zrows : 16 * P.blk[0].tilesA; }
template <int TA> __device__ __forceinline__ constexpr int toep_pad() { return TA == 2 ? 16 : 8; }
template <int TA> __device__ __forceinline__ int toep_im_row(int nf) { return TA == 2 ? (nf + 3) & ~3 : nf; }

__device__ __forceinline__ const double *s1_toep_table(const DevProblem &P, const double *smem)
{
    return smem + (size_t)NC * (P.XR + s1_zrows(P)) + (size_t)2 * NC * RW;
}

// FWD: Os = Zh, rows part * nf + n of (A_part x)[n], Bs = Xs (x, zero from row K on); else Os = Xs, rows k of A^T g, Bs = Zh (g).
// (nf, K, tlen come from the caller's registers: read from the DevProblem here they are serial scalar-memory round trips right
// behind the barrier, with every wave of the workgroup waiting and the MFMA pipe idle.)
//
// What shapes this routine (tools/ubench/f64_overlap.hip, toep_loop.hip, toep_gemm_probe): an fp64 MFMA occupies the SIMD's
// VALU for its 64 cycles -- no VALU instruction of either wave issues meanwhile -- so every VALU instruction between two
// MFMAs is serial time (a first version with 5 pointer increments per 8 MFMAs and ~60 address instructions per piece ran at
// 82 cycles per MFMA for a wave alone and lost ~900 cycles per piece).  Hence:
//   * blocks of five quads (1 quad = 16 reduction indices = 4 MFMAs; 80 rows of g / 80 columns of A = one block) with every
//     operand address an immediate offset from five pointers that are set once per block;
//   * the lane-dependent parts of all addresses are computed once per call, a block's pointers are those plus a scalar;
//   * operands are read one quad ahead, behind the first MFMA of the quad before, so that they return in the shadow of the
//     other three -- across blocks, across the real / imaginary rows of g, and across the pieces of a wave: one pipeline per call.
// Requires nf / 16 == 5 and K / 16 a multiple of 5 (bdrt_problem_create sets toepA only then).
#ifndef BDRT_TOEP_STAMP
#define BDRT_TOEP_STAMP(slot)          // tools/ubench/toep_gemm_probe defines it: cycle stamps inside the routine
#endif
template <bool FWD>
__device__ __forceinline__ void toep_gemm(int nf, int K, int tlen, const double *Tt, const double *Bs, double *Os, int wave, int lane)
{
    typedef const __attribute__((address_space(3))) double *lds_cptr;
    const int i = lane & 15, kq = lane >> 4, col = i;
    wave = __builtin_amdgcn_readfirstlane(wave);
    constexpr int TPP = 5;
    const int r4 = nf & 3;
    const int T = FWD ? 2 * TPP : (K >> 4);
    const int base = T & ~7, R = T - base;                 // whole rounds of eight tiles, the tiles beyond
    const int nqt = FWD ? (K >> 4) : 2 * TPP;              // quads of a whole tile's reduction
    const bool halves = R > 0 && R <= 4 && nqt % 10 == 0;  // the tiles beyond, each shared by two waves
    const int npiece = base + (halves ? 2 * R : R);
    if (wave >= npiece) return;
    BDRT_TOEP_STAMP(0);

    // ---- lane constants ----
    // A operand: table position falls by 4 per chunk going forward (the pointer sits 76 below the block's first position, the
    // lowest one its five quads reach), rises going backward
    const lds_cptr aL = (lds_cptr)(Tt + (FWD ? i - kq - 76 : kq - i));
    // B operand rows 4 j + kq (+ 16 per quad: the swizzle repeats); backward: what moves them to the imaginary rows nf + ...
    lds_cptr bL0 = (lds_cptr)(Bs + swz(kq, col)), bL1 = (lds_cptr)(Bs + swz(4 + kq, col)), bL2 = (lds_cptr)(Bs + swz(8 + kq, col)),
             bL3 = (lds_cptr)(Bs + swz(12 + kq, col));
    const int dI0 = FWD ? 0 : swz(nf + kq, col) - swz(kq, col), dI1 = FWD ? 0 : swz(nf + 4 + kq, col) - swz(4 + kq, col),
              dI2 = FWD ? 0 : swz(nf + 8 + kq, col) - swz(8 + kq, col), dI3 = FWD ? 0 : swz(nf + 12 + kq, col) - swz(12 + kq, col);
    // backward: the chunk for the rows of g beyond the chunks of four (kq < r4 real, kq < 2 r4 imaginary, the other lanes a zero row)
    const int mpart = kq >= r4 ? 1 : 0, mn = nf - r4 + kq - mpart * r4;
    const lds_cptr mL = (lds_cptr)(Tt + mpart * tlen + 8 + mn - i + (K - 1));
    double Bm = 0.0;
    if (!FWD && r4) Bm = Bs[swz(kq < 2 * r4 ? mpart * nf + mn : 2 * nf + kq, col)];

    // ---- the piece in flight (uniform) ----
    int t = 0, half = -1, orow = 0, nblk = 0, nrem = 0, aoff = 0, boff = 0;
    bool mixed = false;
    lds_cptr a, b0, b1, b2, b3;
    // operand buffers: (A2, B2) holds the first quad of a block, the other four alternate between (A0, B0) and (A1, B1) -- the same
    // registers in every block, whatever follows it
    double A0[4], B0[4], A1[4], B1[4], A2[4], B2[4], Am = 0.0;
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};

    auto describe = [&](int sidx) {
        t = sidx; half = -1;
        if (sidx >= base && halves) { t = base + ((sidx - base) >> 1); half = (sidx - base) & 1; }
        if (FWD) {
            const int part = t >= TPP ? 1 : 0, n0 = 16 * (t - part * TPP);
            orow = part * nf + n0;
            const int nch = (K + 3) >> 2, nq = nch >> 2;
            const int q0 = half == 1 ? nq / 2 : 0, q1 = half == 0 ? nq / 2 : nq;
            nblk = (q1 - q0) / 5; nrem = half == 0 ? 0 : nch & 3;
            aoff = part * tlen + 8 + n0 + (K - 1) - 16 * q0;      // position of (row i = 0, kq = 0) of the first chunk
            boff = 16 * NC * q0;
            mixed = false;
        } else {
            orow = 16 * t;
            nblk = half < 0 ? 2 : 1; nrem = 0;
            aoff = 8 - 16 * t + (K - 1); boff = 0;
            mixed = r4 != 0 && half != 0;
        }
    };
    // pointers of block blk of the piece
    auto set_ptrs = [&](int blk) {
        if (FWD) {
            a = aL + (aoff - 80 * blk);
            const int o = boff + 80 * NC * blk;
            b0 = bL0 + o; b1 = bL1 + o; b2 = bL2 + o; b3 = bL3 + o;
        } else {
            const bool im = half == 1 || blk == 1;
            a = aL + (aoff + (im ? tlen : 0));
            b0 = bL0 + (im ? dI0 : 0); b1 = bL1 + (im ? dI1 : 0); b2 = bL2 + (im ? dI2 : 0); b3 = bL3 + (im ? dI3 : 0);
        }
    };
    // quad r (0..4) of the block
    auto ld = [&](double (&A)[4], double (&B)[4], auto Rr) {
        constexpr int r = decltype(Rr)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) A[j] = FWD ? a[76 - 16 * r - 4 * j] : a[16 * r + 4 * j];
        B[0] = b0[16 * NC * r]; B[1] = b1[16 * NC * r]; B[2] = b2[16 * NC * r]; B[3] = b3[16 * NC * r];
    };
    typedef std::integral_constant<int, 0> R0;
    typedef std::integral_constant<int, 1> R1;
    typedef std::integral_constant<int, 2> R2;
    typedef std::integral_constant<int, 3> R3;
    typedef std::integral_constant<int, 4> R4;
    auto mm = [&](const double (&A)[4], const double (&B)[4], auto &&next) {
        acc0 = mfma_f64(A[0], B[0], acc0);
        __builtin_amdgcn_sched_barrier(0);
        next();
        __builtin_amdgcn_sched_barrier(0);
        acc1 = mfma_f64(A[1], B[1], acc1);
        acc0 = mfma_f64(A[2], B[2], acc0);
        acc1 = mfma_f64(A[3], B[3], acc1);
        __builtin_amdgcn_sched_barrier(0);
    };
    // five quads; `after` runs behind the first MFMA of the last quad and requests what follows into (A2, B2)
    auto block = [&](auto &&after) {
        mm(A2, B2, [&]() { ld(A0, B0, R1()); });
        mm(A0, B0, [&]() { ld(A1, B1, R2()); });
        mm(A1, B1, [&]() { ld(A0, B0, R3()); });
        mm(A0, B0, [&]() { ld(A1, B1, R4()); });
        mm(A1, B1, after);
    };
    // opens piece sidx: its first quad into (A2, B2), the odd chunk's A operand
    auto open = [&](int sidx) {
        describe(sidx);
        set_ptrs(0);
        ld(A2, B2, R0());
        if (!FWD && mixed) Am = mL[-16 * t];
    };

    open(wave);
    BDRT_TOEP_STAMP(1);
    for (int sidx = wave; sidx < npiece; sidx += 8) {
        const bool more = sidx + 8 < npiece;
        const int o_row = orow, o_half = half, o_nrem = nrem;
        const bool o_mixed = mixed;
        const double o_Am = Am;
        const int nb = nblk;
        for (int blk = 0; blk < nb; ++blk) {
            const bool last = blk + 1 == nb;
            block([&]() {
                if (!last) { set_ptrs(blk + 1); ld(A2, B2, R0()); }
                else if (FWD && o_nrem) { set_ptrs(nb); ld(A2, B2, R0()); }      // the chunks beyond the last quad
                else if (more) open(sidx + 8);
            });
        }
        if (FWD && o_nrem) {
            double ra[3] = {A2[0], A2[1], A2[2]}, rb[3] = {B2[0], B2[1], B2[2]};
            acc0 = mfma_f64(ra[0], rb[0], acc0);
            if (o_nrem > 1) acc1 = mfma_f64(ra[1], rb[1], acc1);
            if (o_nrem > 2) acc0 = mfma_f64(ra[2], rb[2], acc0);
            __builtin_amdgcn_sched_barrier(0);
            if (more) open(sidx + 8);               // (behind these MFMAs: a short bubble in front of the next piece)
        }
        if (!FWD && o_mixed) acc1 = mfma_f64(o_Am, Bm, acc1);
        __builtin_amdgcn_sched_barrier(0);
        BDRT_TOEP_STAMP(2 + 3 * (sidx >> 3));
        const d4 s = acc0 + acc1;
        BDRT_TOEP_STAMP(3 + 3 * (sidx >> 3));
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            double *o = Os + swz(o_row + kq + 4 * rr, col);
            if (o_half < 0) *o = s[rr];
            else __hip_atomic_fetch_add(o, s[rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        acc0 = d4{0.0, 0.0, 0.0, 0.0}; acc1 = d4{0.0, 0.0, 0.0, 0.0};
        BDRT_TOEP_STAMP(4 + 3 * (sidx >> 3));
    }
}

// the output cells of the tiles that two waves share: zero before the GEMM (by all NT threads, any time after the previous
// reader of those rows and before the barrier in front of the GEMM)
template <bool FWD>
__device__ __forceinline__ void toep_zero_split(int nf, int K, double *Os, int tid)
{
    const int TPP = nf >> 4, T = FWD ? 2 * TPP : (K >> 4);
    const int base = T & ~7, R = T - base;
    if (R == 0 || R > 4 || (FWD ? (K >> 4) : 2 * TPP) % 10 != 0) return;       // (toep_gemm's `halves`)
    for (int e = tid; e < R * 256; e += NT) {
        const int t = base + (e >> 8), r = (e >> 4) & 15;
        const int part = (FWD && t >= TPP) ? 1 : 0;
        const int row = FWD ? part * nf + 16 * (t - part * TPP) + r : 16 * t + r;
        Os[row * NC + (e & 15)] = 0.0;
    }
}

// ---- the same for any shape (DevProblem::toepA == 2: nf, K >= 16) ---------------------------------------------------------------
// What differs from toep_gemm:
//   * a part's rows beyond its whole tiles are one more (partial) tile when there are more than two of them (two or fewer are the
//     VALU dot products of the caller, as there); the rows of a partial tile beyond nf (forward) / K (backward) are not stored;
//   * the imaginary rows of Zh start at nf rounded up to four, with zeros in between and behind: the backward reduction is
//     ceil(nf / 4) whole chunks per part, whatever nf;
//   * a reduction is cut into blocks of at most five quads of equal length (immediate offsets from one pointer set per block,
//     as there) plus up to three single chunks, a step of their own; whatever a step is, what follows it is requested by the same
//     instructions behind the first MFMA of its last quad;
//   * T tiles on eight waves: rounds of eight whole tiles; the R tiles beyond them as halves of the reduction range (R <= 4),
//     four whole tiles and the others as halves (R = 5, 6), whole (R = 7).  Halves add up in zeroed cells as there.
struct ToepSched { int base, R, mode, npiece; };     // mode 0: whole tiles only, 1: the R tiles in halves, 2: four whole + (R - 4) in halves
__host__ __device__ inline ToepSched toep_sched(int T, bool can_halve)
{
    ToepSched sc;
    sc.base = T & ~7; sc.R = T - sc.base;
    sc.mode = (sc.R == 0 || !can_halve || sc.R == 7) ? 0 : (sc.R <= 4 ? 1 : 2);
    sc.npiece = sc.base + (sc.mode == 0 ? sc.R : (sc.mode == 1 ? 2 * sc.R : 2 * sc.R - 4));
    return sc;
}
// tiles of m rows (of one part going forward): two or fewer rows beyond the whole tiles are not a tile
__host__ __device__ inline int toep_gen_tiles(int m) { return (m & 15) <= 2 ? (m >> 4) : ((m + 15) >> 4); }
__host__ __device__ inline int toep_gen_odd(int m) { return (m & 15) <= 2 ? (m & 15) : 0; }

// The runs of wave `wave` through one GEMM, worked out on the host at bdrt_problem_create (the schedule depends on the shape only;
// walking it on the device cost ~200 scalar instructions per step in front of the step's first MFMA).  A run = the reduction range
// of one part for one tile (or half of it): whole quads and, behind them, the range's single chunks.  Two words per run:
//   w0 = table offset of the A operand (16 bits) | first quad << 16 (B operand rows, 8 bits) | kind << 24 | imaginary part << 30
//        kind 0: a whole tile's run (forward: one per tile; backward: two, real and imaginary -- or one of them when two waves share
//        the tile), 1 / 2: first / second half of the quads of a forward tile that two waves share (the second with the single chunks)
//   w1 = first output row (12 bits) | rows of the tile that exist << 12 (5 bits) | shared with another wave << 21 | valid << 24
// A wave's whole tiles come first, then at most one shared one.  The list ends with a run that is not valid and repeats the first
// one's addresses (requested, never used).
constexpr int TOEP_STEPS = 8;                         // runs per wave and GEMM, with the end mark
constexpr int TOEP_STEP_WORDS = 2 * 8 * TOEP_STEPS * 2;    // both GEMMs, eight waves
inline int toep_gen_steps(bool fwd, int nf, int K, int tlen, int wave, unsigned *out)
{
    const int PAD = 16;
    const int nfi = (nf + 3) & ~3;
    const int TPP = fwd ? toep_gen_tiles(nf) : 0;
    const int T = fwd ? 2 * TPP : toep_gen_tiles(K);
    const int nch = fwd ? (K + 3) >> 2 : nfi >> 2;           // chunks of four reduction indices (per part going backward)
    const int nq = nch >> 2;
    const int qsplit = (nq + 1) >> 1;
    const int lim = fwd ? nf : K;
    if (nq < 2 || nq > 12) return -1;
    const ToepSched sc = toep_sched(T, fwd ? nq >= 4 : true);           // (a run has two quads or more)
    int n = 0;
    bool shared_seen = false;
    auto emit = [&](int part, int n0, int q0, int kind, int shared) {
        const int aoff = fwd ? part * tlen + PAD + n0 + (K - 1) - 16 * q0 : part * tlen + PAD - n0 + (K - 1);
        const int orow = fwd ? part * nfi + n0 : n0;
        int room = lim - n0; room = room > 16 ? 16 : room;
        if (aoff < 0 || aoff > 65535 || q0 > 255 || orow > 4095 || n >= TOEP_STEPS - 1) return -1;
        out[2 * n] = (unsigned)aoff | (unsigned)q0 << 16 | (unsigned)kind << 24 | (unsigned)(fwd ? 0 : part) << 30;
        out[2 * n + 1] = (unsigned)orow | (unsigned)room << 12 | (unsigned)shared << 21 | 1u << 24;
        ++n;
        return 0;
    };
    for (int s = wave; s < sc.npiece; s += 8) {
        int t = s, half = -1;
        if (s >= sc.base) {
            const int j = s - sc.base;
            if (sc.mode == 1) { t = sc.base + (j >> 1); half = j & 1; }
            else if (sc.mode == 2 && j >= 4) { t = sc.base + 4 + ((j - 4) >> 1); half = (j - 4) & 1; }
        }
        if (shared_seen) return -1;                           // (whole tiles first, at most one shared tile)
        if (half >= 0) shared_seen = true;
        if (fwd) {
            const int part = t >= TPP ? 1 : 0, n0 = 16 * (t - part * TPP);
            if (emit(part, n0, half == 1 ? qsplit : 0, half + 1, half >= 0 ? 1 : 0)) return -1;
        } else if (half < 0) {
            if (emit(0, 16 * t, 0, 0, 0) || emit(1, 16 * t, 0, 0, 0)) return -1;
        } else if (emit(half, 16 * t, 0, 0, 1)) return -1;
    }
    const int runs = n;
    if (runs == 0) { out[0] = (unsigned)(PAD + K - 1); out[1] = 0; }
    else { out[2 * n] = out[0]; out[2 * n + 1] = out[1] & ~(1u << 24); }
    ++n;
    for (; n < TOEP_STEPS; ++n) { out[2 * n] = out[2 * runs]; out[2 * n + 1] = out[2 * runs + 1]; }
    return runs;
}

template <int I, int N, class F>
__device__ __forceinline__ void toep_static_for(F &&f)
{
    if constexpr (I < N) { f(std::integral_constant<int, I>()); toep_static_for<I + 1, N>(f); }
}

// Tt: the generator table, with the run lists behind it (s1_toep_init).  A run is straight-line code for its number of quads (the
// same for every whole tile of a call: one switch in front of the loops) -- no block loop inside a run, one place in the code for
// every load, so that the operand buffers and the accumulators stay where they are from run to run.  (A version that walked
// blocks of two to five quads through one chain of code with entry points moved sixteen to twenty-four registers per step, and
// a VALU instruction costs a whole MFMA slot while the SIMD's other wave streams MFMAs.)
template <bool FWD>
__device__ __forceinline__ void toep_gemm_gen(int nf, int K, int tlen, const double *Tt, const double *Bs, double *Os, int wave, int lane)
{
    typedef const __attribute__((address_space(3))) double *lds_cptr;
    typedef const __attribute__((address_space(3))) unsigned *lds_uptr;
    constexpr int BIAS = FWD ? 16 * 12 + 12 : 0;          // keeps the immediate offsets of the A operand non-negative up to quad 12
    const int i = lane & 15, kq = lane >> 4, col = i;
    wave = __builtin_amdgcn_readfirstlane(wave);
    const int nfi = (nf + 3) & ~3;
    const int nch = FWD ? (K + 3) >> 2 : nfi >> 2;
    const int nq = nch >> 2, nr = nch & 3;
    // lane j holds run j of this wave
    const lds_uptr stp = (lds_uptr)(Tt + 2 * tlen) + ((FWD ? 0 : 8) + wave) * (2 * TOEP_STEPS) + 2 * (lane & (TOEP_STEPS - 1));
    const unsigned w0v = stp[0], w1v = stp[1];

    // ---- lane constants (see toep_gemm) ----
    const lds_cptr aL = (lds_cptr)(Tt + (FWD ? i - kq - BIAS : kq - i));
    lds_cptr bL0 = (lds_cptr)(Bs + swz(kq, col)), bL1 = (lds_cptr)(Bs + swz(4 + kq, col)), bL2 = (lds_cptr)(Bs + swz(8 + kq, col)),
             bL3 = (lds_cptr)(Bs + swz(12 + kq, col));
    const int dI0 = FWD ? 0 : swz(nfi + kq, col) - swz(kq, col), dI1 = FWD ? 0 : swz(nfi + 4 + kq, col) - swz(4 + kq, col),
              dI2 = FWD ? 0 : swz(nfi + 8 + kq, col) - swz(8 + kq, col), dI3 = FWD ? 0 : swz(nfi + 12 + kq, col) - swz(12 + kq, col);

    lds_cptr a, b0, b1, b2, b3;
    double A0[4], B0[4], A1[4], B1[4], A2[4], B2[4];
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    // quad r behind the run's pointers
    auto ld = [&](double (&A)[4], double (&B)[4], auto Rr) {
        constexpr int r = decltype(Rr)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) A[j] = FWD ? a[BIAS - 16 * r - 4 * j] : a[16 * r + 4 * j];
        B[0] = b0[16 * NC * r]; B[1] = b1[16 * NC * r]; B[2] = b2[16 * NC * r]; B[3] = b3[16 * NC * r];
    };
    // the run's pointers and its first quad into (A2, B2)
    auto prefetch = [&](unsigned w0) {
        const int aoff = (int)(w0 & 0xffffu), o = 16 * NC * (int)((w0 >> 16) & 0xffu);
        a = aL + aoff;
        if (FWD) { b0 = bL0 + o; b1 = bL1 + o; b2 = bL2 + o; b3 = bL3 + o; }
        else {
            const bool im = (w0 >> 30) != 0;
            b0 = bL0 + (im ? dI0 : 0); b1 = bL1 + (im ? dI1 : 0); b2 = bL2 + (im ? dI2 : 0); b3 = bL3 + (im ? dI3 : 0);
        }
        ld(A2, B2, std::integral_constant<int, 0>());
    };
    // a run of L quads and m single chunks; FIRST: the tile's first run (nothing accumulated yet); nw0: the run that follows.
    // Quad r + 1 is requested behind the first MFMA of quad r; behind the first MFMA of the last quad: the single chunks into the
    // buffer that quad does not use (whether there are any or not: no load under a condition) and the next run's first quad.
    auto run = [&](auto Lc, auto Fc, unsigned nw0, int m) {
        constexpr int L = decltype(Lc)::value;
        constexpr bool FIRST = decltype(Fc)::value;
        toep_static_for<0, L>([&](auto Rc) {
            constexpr int r = decltype(Rc)::value;
            auto quad = [&](const double (&A)[4], const double (&B)[4]) {
                const d4 zero = {0.0, 0.0, 0.0, 0.0};
                acc0 = mfma_f64(A[0], B[0], (FIRST && r == 0) ? zero : acc0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (((r + 1) & 1) != 0) ld(A0, B0, std::integral_constant<int, r + 1>());
                else ld(A1, B1, std::integral_constant<int, r + 1>());
                if constexpr (r == L - 1) prefetch(nw0);
                __builtin_amdgcn_sched_barrier(0);
                acc1 = mfma_f64(A[1], B[1], (FIRST && r == 0) ? zero : acc1);
                acc0 = mfma_f64(A[2], B[2], acc0);
                acc1 = mfma_f64(A[3], B[3], acc1);
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (r == 0) quad(A2, B2);
            else if constexpr ((r & 1) != 0) quad(A0, B0);
            else quad(A1, B1);
        });
        auto single = [&](const double (&A)[4], const double (&B)[4]) {
            if (m > 0) acc0 = mfma_f64(A[0], B[0], acc0);
            if (m > 1) acc1 = mfma_f64(A[1], B[1], acc1);
            if (m > 2) acc0 = mfma_f64(A[2], B[2], acc0);
            __builtin_amdgcn_sched_barrier(0);
        };
        if constexpr ((L & 1) != 0) single(A0, B0); else single(A1, B1);
    };
    auto store = [&](unsigned c1) {
        const d4 sum = acc0 + acc1;
        const int orow = (int)(c1 & 0xfffu), room = (int)((c1 >> 12) & 0x1fu);
        const bool shared = ((c1 >> 21) & 1u) != 0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            if (kq + 4 * rr < room) {
                double *o = Os + swz(orow + kq + 4 * rr, col);
                if (!shared) *o = sum[rr];
                else __hip_atomic_fetch_add(o, sum[rr], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    };
    typedef std::true_type Tr;
    typedef std::false_type Fa;

    unsigned c0 = __builtin_amdgcn_readlane(w0v, 0), c1 = __builtin_amdgcn_readlane(w1v, 0);
    if (!((c1 >> 24) & 1u)) return;
    prefetch(c0);
    // the loops for whole tiles of L quads, then the shared tile if the wave has one
    auto body = [&](auto Lc) {
        constexpr int L = decltype(Lc)::value;
        int st = 1;
        if (FWD) {
            while (((c1 >> 24) & 1u) && ((c0 >> 24) & 3u) == 0) {
                const unsigned n0w = __builtin_amdgcn_readlane(w0v, st), n1w = __builtin_amdgcn_readlane(w1v, st);
                ++st;
                run(Lc, Tr(), n0w, nr);
                store(c1);
                c0 = n0w; c1 = n1w;
            }
            if ((c1 >> 24) & 1u) {
                const unsigned n0w = __builtin_amdgcn_readlane(w0v, st);
                if (((c0 >> 24) & 3u) == 1) run(std::integral_constant<int, (L + 1) / 2 >= 2 ? (L + 1) / 2 : 2>(), Tr(), n0w, 0);
                else run(std::integral_constant<int, L / 2 >= 2 ? L / 2 : 2>(), Tr(), n0w, nr);
                store(c1);
            }
        } else {
            while (((c1 >> 24) & 1u) && !((c1 >> 21) & 1u)) {
                const unsigned m0w = __builtin_amdgcn_readlane(w0v, st);
                const unsigned n0w = __builtin_amdgcn_readlane(w0v, st + 1), n1w = __builtin_amdgcn_readlane(w1v, st + 1);
                st += 2;
                run(Lc, Tr(), m0w, nr);                    // real part; requests the imaginary part's first quad
                run(Lc, Fa(), n0w, nr);
                store(c1);
                c0 = n0w; c1 = n1w;
            }
            if ((c1 >> 24) & 1u) {
                const unsigned n0w = __builtin_amdgcn_readlane(w0v, st);
                run(Lc, Tr(), n0w, nr);
                store(c1);
            }
        }
    };
    switch (nq) {
    case 2: body(std::integral_constant<int, 2>()); break;
    case 3: body(std::integral_constant<int, 3>()); break;
    case 4: body(std::integral_constant<int, 4>()); break;
    case 5: body(std::integral_constant<int, 5>()); break;
    case 6: body(std::integral_constant<int, 6>()); break;
    case 7: body(std::integral_constant<int, 7>()); break;
    case 8: body(std::integral_constant<int, 8>()); break;
    case 9: if constexpr (FWD) body(std::integral_constant<int, 9>()); break;
    case 10: if constexpr (FWD) body(std::integral_constant<int, 10>()); break;
    case 11: if constexpr (FWD) body(std::integral_constant<int, 11>()); break;
    case 12: if constexpr (FWD) body(std::integral_constant<int, 12>()); break;
    default: break;
    }
}

// the output cells of the tiles that two waves share (toep_gemm_gen's schedule): zero before the GEMM, as toep_zero_split
template <bool FWD>
__device__ __forceinline__ void toep_zero_split_gen(int nf, int K, double *Os, int tid)
{
    const int nfi = (nf + 3) & ~3;
    const int TPP = FWD ? toep_gen_tiles(nf) : 0;
    const int T = FWD ? 2 * TPP : toep_gen_tiles(K);
    const ToepSched sc = toep_sched(T, FWD ? (((K + 3) >> 2) >> 2) >= 4 : true);
    if (sc.mode == 0) return;
    const int first = sc.mode == 1 ? sc.base : sc.base + 4, cnt = sc.mode == 1 ? sc.R : sc.R - 4;
    for (int e = tid; e < cnt * 256; e += NT) {
        const int t = first + (e >> 8), r = (e >> 4) & 15;
        const int part = (FWD && t >= TPP) ? 1 : 0;
        const int n = FWD ? 16 * (t - part * TPP) + r : 16 * t + r;
        if (n < (FWD ? nf : K)) Os[(FWD ? part * nfi + n : n) * NC + (e & 15)] = 0.0;
    }
}

// once per kernel, all NT threads, ends with a barrier: the generator table, and zeros in the rows of Zh behind the 2 nf
// rows of g that the backward GEMM's last chunk multiplies for its idle lanes
__device__ __forceinline__ void s1_toep_init(const DevProblem &P, double *smem)
{
    const DevBlock &B = P.blk[0];
    double *Tt = const_cast<double *>(s1_toep_table(P, smem));
    double *Zh = smem + (size_t)P.XR * NC;
    const int glen = P.nf + B.K - 1, tlen = P.tlen, pad = P.toepA == 2 ? 16 : 8;
    for (int e = threadIdx.x; e < 2 * tlen; e += NT) {
        const int part = e >= tlen ? 1 : 0, j = e - part * tlen - pad;
        Tt[e] = (j >= 0 && j < glen) ? B.tg[(size_t)part * glen + j] : 0.0;
    }
    if (P.toepA == 2) {
        unsigned *st = (unsigned *)(Tt + 2 * tlen);
        for (int e = threadIdx.x; e < TOEP_STEP_WORDS; e += NT) st[e] = P.tsteps[e];
    }
    // (toepA == 2: all of Zh -- the rows between the parts and behind them stay zero for good)
    for (int e = (P.toepA == 2 ? 0 : 2 * P.nf * NC) + threadIdx.x; e < s1_zrows(P) * NC; e += NT) Zh[e] = 0.0;
    __syncthreads();
}

// the chain's private row as seen by the sampler after a LDSIO evaluation: d lp / d theta in parameter order
__device__ __forceinline__ double *s1_grad_row(const DevProblem &P, double *smem, int c);


// LDS: Xs [XR rows][16] | Zh [16*tilesA rows][16] | private rows [16 chains][2 RW] | (toepA) generator table [2][tlen] | (toepA == 2) step lists
__host__ __device__ inline size_t s1_lds_doubles(const DevProblem &P)
{
    return (size_t)NC * (P.XR + s1_zrows(P)) + (size_t)2 * NC * RW + (P.toepA ? (size_t)2 * P.tlen : 0) + (P.toepA == 2 ? (size_t)TOEP_STEP_WORDS / 2 : 0);
}

// Two thread mappings of a chain's K-vectors inside its half-wave:
//   M1  k = l32 + 32 u   (u < 6): coalesced global rows, the MFMA operand tile;
//   M2  k = 6 l32 + u    (u < 6): six consecutive k per lane, so a 17-tap convolution needs one 22-value LDS window per
//                                 lane instead of 102 reads, and the ups neighbours k-2..k+2 are mostly the lane's own registers.
// A vector changes mapping through the chain's private LDS row (written in one mapping, read in the other); only the
// owning half-wave touches that row, so no barrier is involved.  The whole prior chain x -> L x -> w -> L^T w is
// independent of A x, so it is issued between the forward GEMM and the barrier that publishes A x: the VALU work of one
// wave overlaps the MFMA work of the other wave on the same SIMD.
//
// LDSIO (the NUTS kernel): io.theta points to LDS rows (the sampler keeps theta there) and the gradient is not stored to
// io.grad but left, in parameter order, in the chain's private LDS row (s1_grad_row) for the sampler's next stage.
//
// All threads of the workgroup must call.  Ends with a __syncthreads().
struct NoHook { __device__ __forceinline__ void operator()() const {} };

// `before_backward` runs right before the backward GEMM (the last ~10 k cycles of the evaluation): the sampler uses it to
// issue the global loads of the state it needs next, so that their latency hides behind the MFMA work.
// `after_x_ready` runs right after the first barrier (X of all 16 chains published): the sampler reads the chains' activity
// flags there -- every wave has finished its previous round by then -- instead of voting at a barrier of its own.
// TA: DevProblem::toepA (0, 1: toep_gemm, 2: toep_gemm_gen) -- the caller has run s1_toep_init once in this kernel.
// KU: basis functions per lane of a half-wave, K <= 32 KU (2, 3, 4 or 6): every per-lane loop over k is unrolled KU (LPC = 64: KU / 2)
// times whatever K is, so a short basis pays for a long one unless the caller picks the instantiation by K (s1_ku)
template <bool LDSIO, int LPC = 32, class Hook = NoHook, class Hook1 = NoHook, int TA = 0, int KU = 6>
__device__ inline void logp_grad_tile_s1(const DevProblem &P, const TileIO &io, double *smem, Hook before_backward = Hook(),
                                         Hook1 after_x_ready = Hook1())
{
    int tid = threadIdx.x;
    // Opaque to the optimiser: inside the sampler's round loop everything derived from the thread index is loop invariant,
    // gets hoisted out of the loop by the hundreds (LDS addresses, swizzle offsets, row pointers) and is then spilled to
    // scratch and reloaded every round.  Recomputing a few integer ops per evaluation is far cheaper.
    __asm__ volatile("" : "+v"(tid));
    // LPC lanes own one chain: 32 (half-wave, 512 threads, 2 waves per SIMD) or 64 (wave, 1024 threads, 4 waves per SIMD,
    // half the per-lane work and registers)
    constexpr int UKV = 32 * KU / LPC, UNV = 128 / LPC, WINV = UKV + NTAP - 1, NWV = 16 * LPC / 64;     // K <= 32 KU, Nf <= 128
    constexpr int GPFV = LPC == 32 ? 7 : 3;
    const int lane = tid & 63, wave = tid >> 6;
    const int c = tid / LPC;                               // chain owned by this group of LPC lanes
    const int l32 = tid % LPC, hb = lane & (64 - LPC);     // lane within the group, first lane of the group in the wave
    const DevBlock &B = P.blk[0];
    const int nf = P.nf, N2 = 2 * nf, K = B.K, KP = 8 * B.kpairs;
    const int omode = LDSIO ? 0 : P.outlier_mode;           // (the sampler keeps its state in LDS only without outlier parameters: bdrt_nuts.hip use_s1)
    const int dbg = P.dbg;                                  // (read once: a scalar load behind each barrier otherwise)
    const bool valid = c < io.nvalid;
    const int cc = valid ? c : 0;
    const double jac = io.jacobian ? 1.0 : 0.0;

    double *Xs = smem;
    double *Zh = Xs + (size_t)P.XR * NC;
    double *xrow = Zh + (size_t)(TA == 2 ? P.zrows : 16 * B.tilesA) * NC + (size_t)c * (2 * RW);   // the chain's private row: 2 RW doubles
    double *wrow = xrow + RW;

    const double *th = io.theta + (long)cc * io.t_sc;
    typedef const __attribute__((address_space(3))) double *lds_cptr;
    auto TH = [&](int j) -> double { return LDSIO ? ((lds_cptr)th)[j] : th[(long)j * io.t_sj]; };
    double *gr = (io.grad && valid) ? io.grad + (long)cc * io.g_sc : nullptr;
    auto GW = [&](int j, double v) { if (!LDSIO && gr) gr[(long)j * io.g_sj] = v; };
    double gsc = 0.0;                                      // LDSIO: gradient of the scalar this lane owns (lanes 0..8)
    // (LDSIO = the sampler: it never asks for the constrained parameters, the model spectrum or sigma_tot)
    double *pr = (!LDSIO && io.params && valid) ? io.params + (size_t)cc * P.D : nullptr;
    auto PW = [&](int j, double v) { if (!LDSIO && pr) pr[j] = v; };

    long long tprev = (io.prof && tid == 0) ? clock64() : 0;
    long long *trc = g_tile_trace ? g_tile_trace + ((size_t)blockIdx.x * NWV + wave) * 16 : nullptr;
#define BDRT_S1_TRACE(slot) do { if (trc && (lane == 0)) trc[slot] = clock64(); } while (0)
    BDRT_S1_TRACE(0);
#define BDRT_S1_PROF(slot) do { if (io.prof && tid == 0) { const long long t_ = clock64(); io.prof[slot] += t_ - tprev; tprev = t_; } } while (0)
    // wait at the four barriers, summed over the workgroup's waves (slots 25..28 of the phase profile): the cost of the waves' skew;
    // slots 29..31: the waves' own time from the entry to B1, from B1 to B2 and from B2 to B3 (B3 .. the end: the rest of the tile)
    long long twe = io.prof ? clock64() : 0;
#define BDRT_S1_BARRIER(slot, wslot) do { if (io.prof) { const long long tb_ = clock64(); __syncthreads(); const long long te_ = clock64(); \
        if (lane == 0) { atomicAdd((unsigned long long *)&io.prof[slot], (unsigned long long)(te_ - tb_)); \
                         if (wslot) atomicAdd((unsigned long long *)&io.prof[wslot], (unsigned long long)(tb_ - twe)); } \
        twe = te_; } else __syncthreads(); } while (0)

    // ---- P1 (M1): parameters of this chain: scalars by lanes 0..8, x and ups rows coalesced -----------------------------
    double sraw = 0.0, st = 0.0;
    {
        int j = -1;
        if (l32 < 2) j = l32;
        else if (l32 < 6) j = P.o_err + (l32 - 2);
        else if (l32 < 9) j = B.o_d + (l32 - 6);
        if (j >= 0) { st = TH(j); sraw = lean_exp(st); PW(j, sraw); }
    }
    double lp = 0.0;
    int lp_lane = -1;                                      // the lane that ends up with the chain's total (-1: not summed yet)
    double x_[UKV];
    {
        double tx_[UKV], tu_[UKV];
        // (loads first, unconditionally, from a clamped index; THEN the selects: a load under `k < K` becomes a branch around an
        // LDS read with its own s_waitcnt -- six serial LDS round trips per phase, tools/isa_blocks.py)
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            const int k = l32 + LPC * u, kk = k < K ? k : 0;
            tx_[u] = TH(B.o_x + kk);
            tu_[u] = TH(B.o_ups + kk);
        }
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            const bool in = l32 + LPC * u < K;
            tx_[u] = in ? tx_[u] : 0.0;
            tu_[u] = in ? tu_[u] : 0.0;
        }
        // all exponentials of the lane in one straight-line block (tx = 0 beyond K): the polynomial's constants are then
        // materialised once for the six evaluations instead of once per predicated block
        double ex_[UKV];
#pragma unroll
        for (int u = 0; u < UKV; ++u) ex_[u] = lean_exp(tx_[u]);
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            const int k = l32 + LPC * u;
            double xr = 0.0;
            if (k < K) {
                xr = B.is_pos ? ex_[u] : tx_[u];
                if (B.is_pos) lp += jac * tx_[u];
                PW(B.o_x + k, xr);
            }
            x_[u] = xr;
            if (k < KP) Xs[swz(k, c)] = xr;
            xrow[MAXBW + k] = xr;                          // zero beyond K: the convolution halo
            wrow[MAXBW + k] = tu_[u];                      // theta_ups on its way to mapping M2
        }
        if (l32 < MAXBW) {
            xrow[l32] = 0.0; xrow[MAXBW + LPC * UKV + l32] = 0.0;
            wrow[l32] = 0.0; wrow[MAXBW + LPC * UKV + l32] = 0.0;
        }
    }
    static_assert(!TA || LPC == 32, "the Toeplitz-A GEMMs are written for 512 threads");
    const double *Tt = TA ? s1_toep_table(P, smem) : nullptr;
    const int tlen = TA ? P.tlen : 0;
    constexpr int TPAD = toep_pad<TA>();
    const int nfi = toep_im_row<TA>(nf);                   // first row of A_im x in Zh
    if (TA) {
        // the rows of A_re x and A_im x beyond the full tiles (nf % 16 <= 2 of each part): dot products while x is in registers
        const int r16 = TA == 2 ? toep_gen_odd(nf) : (nf & 15), n0 = nf - r16;
        for (int j = 0; j < r16; ++j) {
            double sr = 0.0, si = 0.0;
#pragma unroll
            for (int u = 0; u < UKV; ++u) {
                const int idx = max(TPAD + n0 + j + (K - 1) - (l32 + LPC * u), 0);     // (x_ is zero from K on)
                sr = fma(Tt[idx], x_[u], sr); si = fma(Tt[tlen + idx], x_[u], si);
            }
            sr = hsum<LPC>(sr); si = hsum<LPC>(si);
            if (l32 == 0) { Zh[swz(n0 + j, c)] = sr; Zh[swz(nfi + n0 + j, c)] = si; }
        }
        if (TA == 2) toep_zero_split_gen<true>(nf, K, Zh, tid); else toep_zero_split<true>(nf, K, Zh, tid);
    }
    const double d0 = __shfl(sraw, hb | 6), d1 = __shfl(sraw, hb | 7), d2 = __shfl(sraw, hb | 8);
    // priors of the 9 scalars (std_normal on the six raws, inv_gamma(5,5) on the d's) + log-Jacobian: lane j owns scalar j
    const double israw5 = 5.0 * lean_rcp(sraw);                        // (meaningful in lanes 6..8: 5 / d_i)
    if (l32 < 6) lp += -0.5 * sraw * sraw + jac * st;
    else if (l32 < 9) lp += -6.0 * st - israw5 + jac * st;
    // The prior chain x -> L x -> w -> L^T w (P2) needs neither A x nor the other chains: it runs BEFORE the first barrier.  In the
    // sampler the waves reach this evaluation at different times (their chains close sub-trees of different depth); a wave
    // that is early spends the wait on its P2 -- VALU work that fills the issue slots the late waves' memory-bound bookkeeping
    // leaves -- instead of idling at B1 and then sharing the fp64 pipe with the forward GEMM (an fp64 MFMA occupies the SIMD's VALU
    // for its 64 cycles: MFMA and VALU work add up, whatever the order; tools/ubench/f64_overlap.hip).
#ifndef BDRT_SPEC_EARLY
#define BDRT_SPEC_EARLY 0
#endif
#ifndef BDRT_EARLY_P2
#define BDRT_EARLY_P2 1
#endif
#ifndef BDRT_P2_OPAQUE_K
#define BDRT_P2_OPAQUE_K 1
#endif
    constexpr bool EARLY_P2 = BDRT_EARLY_P2 != 0;
    double zre_[UNV], zim_[UNV], wn_[UNV];
    auto load_spectrum = [&]() {
        const int sp = io.spec ? io.spec[cc] : 0;
        const double *Zm = P.Z + (size_t)sp * N2;
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v, nn = n < nf ? n : 0;
            zre_[v] = Zm[nn]; zim_[v] = Zm[nf + nn]; wn_[v] = P.w[nn];
        }
    };
#pragma unroll 1
    for (int step = 0; step < 2; ++step) {
    if (EARLY_P2 ? step == 1 : step == 0) {
        BDRT_S1_TRACE(1);
        BDRT_S1_BARRIER(25, 29);                                           // B1: X of all 16 chains in the operand tile
        after_x_ready();
        if constexpr (BDRT_SPEC_EARLY) load_spectrum();
        BDRT_S1_TRACE(2);
        BDRT_S1_PROF(1);
        BDRT_S1_TRACE(3);
        if (!(dbg & 1)) {
            if (TA == 2) toep_gemm_gen<true>(nf, K, tlen, Tt, Xs, Zh, wave, lane);
            else if (TA) toep_gemm<true>(nf, K, tlen, Tt, Xs, Zh, wave, lane);
            else gemm_sw<NWV, GPFV>(B.Af, B.tilesA, B.kpairs, Xs, Zh, wave, lane);                // Zh = A x  (pad rows come out as exact zeros)
        }
        continue;
    }

    // ---- P2 (M2): v_i = L_i x, q / ups / dups priors, w_i, sum_i L_i^T w_i -- all on this chain's private rows -------------
    BDRT_S1_TRACE(4);
    if (!(dbg & 4)) {
        const int kb = UKV * l32;                                        // first k of this lane
        double xw[WINV], tuc[UKV];
#pragma unroll
        for (int j = 0; j < WINV; ++j) xw[j] = xrow[kb + j];            // x[kb + j - MAXBW]
#pragma unroll
        for (int u = 0; u < UKV; ++u) tuc[u] = wrow[MAXBW + kb + u];
        wave_sync();                                                   // every lane holds its x window and theta_ups
        // v_i = L_i x: three independent 17-tap convolutions of the window (independent accumulators: the latency of one
        // FMA chain hides behind the other two)
        double v0_[UKV], v1_[UKV], v2_[UKV];
#pragma unroll
        for (int u = 0; u < UKV; ++u) { v0_[u] = 0.0; v1_[u] = 0.0; v2_[u] = 0.0; }
        double tn0 = B.T[0][0], tn1 = B.T[1][0], tn2 = B.T[2][0];
#pragma unroll
        for (int d = 0; d < NTAP; ++d) {
            const double t0 = tn0, t1 = tn1, t2 = tn2;
            if (d + 1 < NTAP) { tn0 = B.T[0][d + 1]; tn1 = B.T[1][d + 1]; tn2 = B.T[2][d + 1]; }
            __asm__ volatile("" ::: "memory");       // keeps the tap loads in program order (one tap ahead), not all 51 up front
#pragma unroll
            for (int u = 0; u < UKV; ++u) { v0_[u] = fma(t0, xw[u + d], v0_[u]); v1_[u] = fma(t1, xw[u + d], v1_[u]); v2_[u] = fma(t2, xw[u + d], v2_[u]); }
        }
        // ups of k-2 .. k+7 around the lane's six k: own registers plus two values from each neighbouring lane
        double ue[UKV + 4], ie[UKV + 2];                                 // ups of k-2..k+7, 1/ups of k-1..k+6
#pragma unroll
        for (int u = 0; u < UKV; ++u) { ue[u + 2] = 0.15 * lean_exp(tuc[u]); ie[u + 1] = lean_rcp(ue[u + 2]); }
        {
            const int lo = hb | ((l32 + LPC - 1) & (LPC - 1)), hi = hb | ((l32 + 1) & (LPC - 1));
            ue[0] = __shfl(ue[UKV], lo); ue[1] = __shfl(ue[UKV + 1], lo);
            ue[UKV + 2] = __shfl(ue[2], hi); ue[UKV + 3] = __shfl(ue[3], hi);
            ie[0] = __shfl(ie[UKV], lo); ie[UKV + 1] = __shfl(ie[1], hi);
        }
        BDRT_S1_PROF(5);
        double sv0 = 0, sv1 = 0, sv2 = 0;
        double w0_[UKV], w1_[UKV], w2_[UKV], gup[UKV];
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            // (k through an opaque copy per element: with k = kb + u visible the compiler forms the lane masks of all six elements'
            // range tests at once, up front -- ten SGPR pairs that do not fit and travel through v_writelane / v_readlane)
            int k = kb + u;
#if BDRT_P2_OPAQUE_K
            __asm__ volatile("" : "+v"(k));
#endif
            w0_[u] = 0.0; w1_[u] = 0.0; w2_[u] = 0.0; gup[u] = 0.0;
            if (k < K) {
                const double um2 = ue[u], um1 = ue[u + 1], uu = ue[u + 2], up1 = ue[u + 3], up2 = ue[u + 4];
                const double t = tuc[u];
                const double iu = ie[u + 1], iu2 = iu * iu;
                const double v0 = v0_[u], v1 = v1_[u], v2 = v2_[u];
                const double q2 = d0 * v0 * v0 + d1 * v1 * v1 + d2 * v2 * v2;
                const double ir = 0.15 * iu;                              // 1 / ups_raw
                // q ~ normal(0, ups) evaluated on q^2; ups_raw ~ inv_gamma(alpha, beta); log transform
                lp += -(t + LOG_015) - 0.5 * q2 * iu2 - (P.ups_alpha + 1.0) * t - P.ups_beta * ir + jac * t;
                sv0 += v0 * v0 * iu2; sv1 += v1 * v1 * iu2; sv2 += v2 * v2 * iu2;
                double gu = -iu + q2 * iu2 * iu;
                // dups[k] = 0.5*(ups[k+1] - 0.5*(ups[k]+ups[k+2]))/ups[k+1] ~ std_normal()
                if (k >= 1 && k + 1 < K) {
                    const double du = 0.5 * (uu - 0.5 * (um1 + up1)) * iu;
                    lp += -0.5 * du * du;
                    gu += -du * 0.25 * (um1 + up1) * iu2;
                }
                if (k >= 2) {                                             // k is the right neighbour of centre k-1
                    const double i0 = ie[u];
                    const double du = 0.5 * (um1 - 0.5 * (um2 + uu)) * i0;
                    gu += du * 0.25 * i0;
                }
                if (k + 2 < K) {                                          // k is the left neighbour of centre k+1
                    const double i0 = ie[u + 2];
                    const double du = 0.5 * (up1 - 0.5 * (uu + up2)) * i0;
                    gu += du * 0.25 * i0;
                }
                gup[u] = uu * gu - (P.ups_alpha + 1.0) + P.ups_beta * ir + jac;
                PW(B.o_ups + k, uu * (1.0 / 0.15));
                w0_[u] = -d0 * v0 * iu2; w1_[u] = -d1 * v1 * iu2; w2_[u] = -d2 * v2 * iu2;
            }
            __builtin_amdgcn_sched_barrier(0);     // one k at a time: interleaving all six only inflates the register peak
        }
        // (L_i^T w_i)[k] = sum_d T_i[d] w_i[k + MAXBW - d]: one window per i through the private row
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
            for (int j = 0; j < WINV; ++j) ww[j] = wrow[kb + j];         // w_i[kb + j - MAXBW]
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
#pragma unroll
        for (int u = 0; u < UKV; ++u) { xrow[MAXBW + kb + u] = gup[u]; wrow[MAXBW + kb + u] = gl_[u]; }   // back to M1 for the epilogue
        double sv;                                                       // lane 6 + i: the total of sv_i
        if constexpr (LPC == 32) {
            const double q[4] = {sv2, 0.0, sv0, sv1};                    // (lane & 3 of the lanes 6, 7, 8 = 2, 3, 0)
            sv = sum32_by_lane<4>(q, l32);
        } else {
            sv0 = hsum<LPC>(sv0); sv1 = hsum<LPC>(sv1); sv2 = hsum<LPC>(sv2);
            sv = l32 == 6 ? sv0 : (l32 == 7 ? sv1 : sv2);
        }
        if (l32 >= 6 && l32 < 9) {                                       // d_i gradients: lane 6+i
            gsc = -0.5 * sraw * sv - 6.0 + israw5 + jac;
            GW(B.o_d + (l32 - 6), gsc);
        }
    }
    }
    // measured spectrum of this chain: requested in front of the forward GEMM (BDRT_SPEC_EARLY: right behind the first barrier, so
    // that the round trip runs under the MFMAs; else in front of the second barrier, where only the wait hides it)
    if constexpr (!(BDRT_SPEC_EARLY)) load_spectrum();
    BDRT_S1_TRACE(5);
    BDRT_S1_BARRIER(26, 30);                                               // B2: A x of all chains in Zh
    BDRT_S1_TRACE(6);
    BDRT_S1_PROF(7);

    // ---- P3 (n = l32 + 32 v): likelihood Z ~ normal(Z_hat, sigma_tot); g_Zhat in place -------------------------------------
    // the six scalar parameters of this chain, broadcast from the lanes that own them (lane j holds scalar j in sraw)
    const double Rinf = 100.0 * __shfl(sraw, hb | 0), induc = __shfl(sraw, hb | 1) * P.induc_scale;
    const double s_res = 0.05 * __shfl(sraw, hb | 2), a_p = 0.05 * __shfl(sraw, hb | 3), a_r = 0.05 * __shfl(sraw, hb | 4),
                 a_i = 0.05 * __shfl(sraw, hb | 5);
    if (!(dbg & 8)) {
        const double c0 = P.sigma_min * P.sigma_min + s_res * s_res;
        const double ap2 = a_p * a_p, ar2 = a_r * a_r, ai2 = a_i * a_i;
        double sR = 0, sL = 0, sH = 0, sHz2 = 0, sHzr2 = 0, sHzi2 = 0;
        // TA: the rows k >= 16 (K / 16) of A^T g (K % 16 <= 2 of them) are dot products with g while it is in registers
        const int rk = TA == 2 ? toep_gen_odd(K) : (TA ? (K & 15) : 0);
        double lk0 = 0.0, lk1 = 0.0;
        if (TA == 2) toep_zero_split_gen<false>(nf, K, Xs, tid); else if (TA) toep_zero_split<false>(nf, K, Xs, tid);
        double pp_lo = 1.0, pp_hi = 1.0;
        // (the phase's LDS reads in one batch, from a clamped row: under `n < nf` each of them is a round trip of its own)
        double azr_[UNV], azi_[UNV], tk0r_[UNV], tk0i_[UNV], tk1r_[UNV], tk1i_[UNV];
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v, nn = n < nf ? n : 0;
            azr_[v] = Zh[swz(nn, c)]; azi_[v] = Zh[swz(nfi + nn, c)];
            tk0r_[v] = 0.0; tk0i_[v] = 0.0; tk1r_[v] = 0.0; tk1i_[v] = 0.0;
            if (TA && rk > 0) {
                tk0r_[v] = Tt[TPAD + nn + rk - 1]; tk0i_[v] = Tt[tlen + TPAD + nn + rk - 1];
                if (rk > 1) { tk1r_[v] = Tt[TPAD + nn + rk - 2]; tk1i_[v] = Tt[tlen + TPAD + nn + rk - 2]; }
            }
        }
#pragma unroll
        for (int v = 0; v < UNV; ++v) {
            const int n = l32 + LPC * v;
            if (n >= nf) continue;
            const double wn = wn_[v];
            const double zr = azr_[v] + Rinf;
            const double zi = azi_[v] + induc * wn;
            // outlier error model (Series_*_outliers_modelcode.txt): 2 Nf extra parameters, read where they are needed
            double so_re = 0.0, so_im = 0.0, r0 = 0.0, r1 = 0.0, t0 = 0.0, t1 = 0.0;
            if (omode) {
                t0 = TH(P.o_so + n); t1 = TH(P.o_so + nf + n);
                r0 = lean_exp(t0); r1 = lean_exp(t1);
                PW(P.o_so + n, r0); PW(P.o_so + nf + n, r1);
                if (omode == 1) so_re = so_im = 0.05 * r0 * r1;
                else { so_re = 0.05 * r0; so_im = 0.05 * r1; }
            }
            const double common = ar2 * zr * zr + ai2 * zi * zi;
            const double s2_re = c0 + ap2 * zr * zr + common + so_re * so_re;
            const double s2_im = c0 + ap2 * zi * zi + common + so_im * so_im;
            const double e_re = zre_[v] - zr, e_im = zim_[v] - zi;
            const double prod = s2_re * s2_im, ip = lean_rcp(prod);       // one reciprocal and one logarithm per (re, im) pair
            const double w_re = s2_im * ip, w_im = s2_re * ip;
            // (the logarithm of the PRODUCT of two (re, im) pairs: one lean_log -- 33 instructions -- per two frequencies of the
            // lane instead of one each; sigma_tot^8 stays far inside the double range for every point a trajectory visits)
            if (v < UNV / 2) pp_lo *= prod; else pp_hi *= prod;
            lp += -0.5 * e_re * e_re * w_re - 0.5 * e_im * e_im * w_im;
            const double h_re = -0.5 * w_re + 0.5 * e_re * e_re * w_re * w_re;
            const double h_im = -0.5 * w_im + 0.5 * e_im * e_im * w_im * w_im;
            const double gzr = e_re * w_re + 2.0 * zr * (h_re * (ap2 + ar2) + h_im * ar2);
            const double gzi = e_im * w_im + 2.0 * zi * (h_im * (ap2 + ai2) + h_re * ai2);
            Zh[swz(n, c)] = gzr;
            Zh[swz(nfi + n, c)] = gzi;
            if (TA && rk > 0) {            // A_part[n][K - rk + j] = tg[part][n + rk - 1 - j]
                lk0 = fma(tk0r_[v], gzr, fma(tk0i_[v], gzi, lk0));
                if (rk > 1) lk1 = fma(tk1r_[v], gzr, fma(tk1i_[v], gzi, lk1));
            }
            sR += gzr;
            sL += gzi * wn;
            sH += h_re + h_im;
            sHz2 += h_re * zr * zr + h_im * zi * zi;
            sHzr2 += (h_re + h_im) * zr * zr;
            sHzi2 += (h_re + h_im) * zi * zi;
            if (omode == 1) {
                // sigma_out = raw .* scale * 0.05 ; raw ~ exponential(lambda) ; scale ~ inv_gamma(alpha, beta)
                const double dso = 2.0 * so_re * (h_re + h_im);
                GW(P.o_so + n, r0 * (0.05 * r1 * dso - P.so_lambda) + jac);
                GW(P.o_so + nf + n, 0.05 * r0 * r1 * dso - (P.so_alpha + 1.0) + P.so_beta / r1 + jac);
                lp += -P.so_lambda * r0 - (P.so_alpha + 1.0) * t1 - P.so_beta / r1 + jac * (t0 + t1);
            } else if (omode == 2) {
                GW(P.o_so + n, r0 * (0.05 * 2.0 * so_re * h_re - P.so_lambda) + jac);
                GW(P.o_so + nf + n, r1 * (0.05 * 2.0 * so_im * h_im - P.so_lambda) + jac);
                lp += -P.so_lambda * (r0 + r1) + jac * (t0 + t1);
            }
            if (!LDSIO && io.Z_hat && valid) { io.Z_hat[(size_t)c * N2 + n] = zr; io.Z_hat[(size_t)c * N2 + nf + n] = zi; }
            if (!LDSIO && io.sigma_tot && valid) {
                io.sigma_tot[(size_t)c * N2 + n] = sqrt(s2_re);
                io.sigma_tot[(size_t)c * N2 + nf + n] = sqrt(s2_im);
            }
        }
        lp += -0.5 * lean_log(pp_lo);
        if (nf > LPC * (UNV / 2)) lp += -0.5 * lean_log(pp_hi);
        // the six sums, the odd row of A^T g and the chain's log-posterior (complete by now) in one butterfly: lane j gets sum j
        double dl = 0.0;
        if constexpr (LPC == 32) {
            const double q[8] = {sR, sL, sH, sHz2, sHzr2, sHzi2, lk0, lp};
            const double tot = sum32_by_lane<8>(q, l32);
            // d lp / d(raw), likelihood part: 100 sR, induc_scale sL, 0.05 * 2 s_res sH, 0.05 * 2 alpha_prop sHz2, ... with
            // s_res = 0.05 raw_2, alpha_* = 0.05 raw_3..5 -- the lane's own raw value
            dl = (l32 == 0 ? 100.0 : (l32 == 1 ? P.induc_scale : 0.005 * sraw)) * tot;
            if (TA && rk > 0) {
                if (l32 == 6) Xs[swz(K - rk, c)] = tot;
                if (rk > 1) { lk1 = hsum<LPC>(lk1); if (l32 == 0) Xs[swz(K - rk + 1, c)] = lk1; }
            }
            lp = tot;                                                    // (lane 7: the log-posterior)
            lp_lane = 7;
        } else {
            sR = hsum<LPC>(sR); sL = hsum<LPC>(sL); sH = hsum<LPC>(sH); sHz2 = hsum<LPC>(sHz2); sHzr2 = hsum<LPC>(sHzr2); sHzi2 = hsum<LPC>(sHzi2);
            if (l32 == 0) dl = 100.0 * sR;
            else if (l32 == 1) dl = P.induc_scale * sL;
            else if (l32 == 2) dl = 0.05 * 2.0 * s_res * sH;
            else if (l32 == 3) dl = 0.05 * 2.0 * a_p * sHz2;
            else if (l32 == 4) dl = 0.05 * 2.0 * a_r * sHzr2;
            else dl = 0.05 * 2.0 * a_i * sHzi2;
        }
        if (l32 < 6) {
            // d lp / d(raw) of Rinf_raw, induc_raw, sigma_res_raw, alpha_prop/re/im_raw (lane j owns scalar j)
            const int j = l32 < 2 ? l32 : P.o_err + (l32 - 2);
            gsc = sraw * (dl - sraw) + jac;
            GW(j, gsc);
        }
    }
    BDRT_S1_TRACE(7);
    BDRT_S1_BARRIER(27, 31);                                               // B3: g_Zhat of all chains in Zh
    BDRT_S1_TRACE(8);
    BDRT_S1_PROF(4);
    before_backward();
    if (!(dbg & 2)) {
        if (TA == 2) toep_gemm_gen<false>(nf, K, tlen, Tt, Zh, Xs, wave, lane);
        else if (TA) toep_gemm<false>(nf, K, tlen, Tt, Zh, Xs, wave, lane);
        else gemm_sw<NWV, GPFV>(B.BkA, B.tilesK, B.rpairsA, Zh, Xs, wave, lane);                    // Xs = A^T g_Zhat
    }
    BDRT_S1_TRACE(9);
    BDRT_S1_BARRIER(28, 0);                                               // B4
    BDRT_S1_TRACE(10);
    BDRT_S1_PROF(8);

    // ---- epilogue (M1): chain rule through x = exp(theta_x); coalesced gradient rows ----------------------------------------
    double gx_[UKV], gu_[UKV];
    {
        // (all LDS reads of the phase in one batch: the rows of A^T g beyond K lie inside the tile, the private rows are 32 KU long)
        double ag_[UKV], gl2_[UKV];
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            const int k = l32 + LPC * u, kk = k < K ? k : 0;
            ag_[u] = Xs[swz(kk, c)]; gl2_[u] = wrow[MAXBW + k]; gu_[u] = xrow[MAXBW + k];
        }
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            const int k = l32 + LPC * u;
            const double graw = ag_[u] + gl2_[u];
            const double gx = B.is_pos ? x_[u] * graw + jac : graw;
            gx_[u] = k < K ? gx : 0.0;
            gu_[u] = k < K ? gu_[u] : 0.0;
            if (!LDSIO && k < K) { GW(B.o_x + k, gx_[u]); GW(B.o_ups + k, gu_[u]); }
        }
    }
    if (LDSIO) {
        wave_sync();                                                   // all transit values are in registers
#pragma unroll
        for (int u = 0; u < UKV; ++u) {
            const int k = l32 + LPC * u;
            if (k < K) { xrow[B.o_x + k] = gx_[u]; xrow[B.o_ups + k] = gu_[u]; }
        }
        if (l32 < 9) xrow[l32 < 2 ? l32 : (l32 < 6 ? P.o_err + (l32 - 2) : B.o_d + (l32 - 6))] = gsc;
    }
    if (lp_lane < 0) { lp = hsum<LPC>(lp); lp_lane = 0; }
    if (l32 == lp_lane && io.lp && valid) io.lp[c] = lp;
    BDRT_S1_PROF(9);
    BDRT_S1_TRACE(11);
    // LDSIO: what follows in the sampler only touches this chain's own rows; its end-of-round barrier closes the round
    if (LDSIO) wave_sync(); else __syncthreads();
}

__device__ __forceinline__ double *s1_grad_row(const DevProblem &P, double *smem, int c)
{
    return smem + (size_t)NC * (P.XR + s1_zrows(P)) + (size_t)c * (2 * RW);
}

}  // namespace bdrt
