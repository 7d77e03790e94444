// bdrt_nuts16.h -- the 16-chains-per-workgroup NUTS kernel (device-resident NUTS, replaces StanModel.sampling, reference
// bayes_drt/inversion.py:1218-1221).  A header so that its instantiations compile in several translation units
// (bdrt_nuts_k*.hip) next to the host side in bdrt_nuts.hip, and so that tools can instantiate one kernel alone.
#pragma once
#include <cmath>
#include <cstring>
#include <type_traits>

#include "bdrt_host.h"
#include "bdrt_lbfgs.h"
#include "bdrt_nuts_device.h"
#include "bdrt_solo.h"
#include "bdrt_wave.h"
#include "bdrt_nuts_args.h"
#include "bdrt_big.h"

namespace bdrt {

constexpr int MAXD = 10;            // checkpoint slots (>= max_treedepth)
constexpr int NQ_CHK = 2 * MAXD + 2;
static_assert(NW * NQ_CHK * NC <= MIN_LR * NC, "Lr buffer too small for the NUTS reduction scratch");

// state vectors per workgroup, each [D][16]
enum { V_TH = 0, V_P, V_G, V_THM, V_PM, V_GM, V_THP, V_PP, V_GP, V_THS, V_GS, V_THQ, V_GQ, V_RHO, V_MINV,
       V_WMEAN, V_WM2, V_ZN /* normals of the next transition's momentum, produced ahead of time */,
       V_TH2 /* second theta row of the wide-vector path (ping-pong with V_TH) */, V_CKC /* MAXD */, V_CKP = V_CKC + MAXD /* MAXD */, V_COUNT = V_CKP + MAXD };

// chain k of a workgroup <-> column (half-wave) of the 16-column tile: k = 0..7 -> columns 0, 2, .., 14 (one wave each),
// k = 8..15 -> columns 1, 3, .., 15
__host__ __device__ __forceinline__ int slot_col(int k) { return 2 * (k & 7) + (k >> 3); }
__host__ __device__ __forceinline__ int col_slot(int c) { return (c & 1) * 8 + (c >> 1); }

// sum over the 32 lanes of a half-wave (one chain), fixed order => deterministic; every lane gets the result
__device__ __forceinline__ double half_sum(double x) { return sum32(x); }

// compiler-level ordering of one wave's LDS traffic (lanes of a half-wave exchange data through LDS without a barrier;
// the hardware executes a wave's LDS instructions in order)
__device__ __forceinline__ void lds_wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


#include "bdrt_nuts_wide.h"
#include "bdrt_solo_wide.h"
#include "bdrt_lbfgs_dev.h"

// Thread mapping of the bookkeeping stages: chain c of the workgroup lives in ONE half-wave (wave c/2, lanes
// 32*(c%2)..+31); its D-vectors are contiguous rows, lane l handles elements l, l+32, ...  Per-chain dot products are
// 5-step xor-shuffle reductions, the per-chain scalar logic runs redundantly in the 32 lanes of the half-wave (state
// in LDS), and no stage between two tile evaluations needs a workgroup barrier.
// NJ = elements of a D-vector per lane (D <= 32*NJ): compile-time so that every pass over a chain's vectors is fully
// unrolled into a batch of independent loads followed by the arithmetic (one memory round trip per stage instead of one
// per element -- the state vectors of 2048+ chains live in HBM/MALL, not in L2).
// TA (MODE 2 only): DevProblem::toepA (1: the default shapes, 2: any shape), the S1 tile's GEMMs take the A operands from the generator table in LDS
#ifndef BDRT_DEEP_VARIANT
#define BDRT_DEEP_VARIANT 1
#endif
#ifndef BDRT_PRE1
#define BDRT_PRE1 0
#endif
#ifndef BDRT_GLDS
#define BDRT_GLDS 0
#endif
#ifndef BDRT_NUTS_ABLATE
#define BDRT_NUTS_ABLATE 0
#endif
#ifndef BDRT_RESIDENT
#define BDRT_RESIDENT 1
#endif
// cache hints of the checkpoint traffic (bits): 1 the single-use reads of waiting siblings non-temporal, 2 the proposal copies
// non-temporal stores, 4 the checkpoints of levels >= 3 non-temporal stores
#ifndef BDRT_NT
#define BDRT_NT 1
#endif
#ifndef BDRT_KARG_RELOAD
#define BDRT_KARG_RELOAD 1
#endif
#ifndef BDRT_NUTS_EARLY_STATE
#define BDRT_NUTS_EARLY_STATE 1
#endif
// PROF: the instantiation that fills the phase profile (NutsArgs::prof).  The measurement kernel is a kernel of its own: with a
// run-time `if (prof)` in front of every stamp the production kernel carries ~25 scalar branches per round, two copies of
// every barrier of the evaluator, and basic-block boundaries the scheduler cannot move loads across.
template <int NJ, int MODE, int TA = 0, bool PROF = false>   // MODE 0: dense L, 1: structured L (generic tile), 2: S1 tile + theta rows in LDS, 3: S1 tile, state in HBM, 4: general half-wave tile
__global__ __launch_bounds__(NT) void nuts_kernel(const DevProblem *__restrict__ Pp, NutsParams np, NutsArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem &P = *Pp;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int c = 2 * wave + (lane >> 5);
    int l32 = lane & 31;
    const int D = P.D, DS = a.ds;
    const int wg = blockIdx.x;
    // Unit u = wg * cpw + k sits in column slot_col(k): the first eight chains of a workgroup get one wave each (even
    // columns = lanes 0..31 of waves 0..7), the next eight the other half-waves.  With few chains per workgroup the per-chain
    // VALU work of a round is then spread over the four SIMDs instead of piling up on one.
    const int *su = a.slot_unit + (size_t)wg * NC;
    const int kslot = col_slot(c);                       // this column's slot within the workgroup
    const int unit = su[kslot];
    const bool valid = unit >= 0;

    // LDS carve-up: [tile region | (fast S1 path: theta rows of the 16 chains) | lp of the 16 chains | chain states | spectrum ids]
    const size_t tile_doubles = MODE == 4 ? hw_lds_doubles(P) : (MODE >= 2 ? s1_lds_doubles(P) : lds_doubles(P));
    constexpr int DSL = 32 * NJ;                    // LDS row stride of the theta rows
    double *thl = smem + tile_doubles;
    double *lpn = thl + (MODE == 2 ? (size_t)NC * DSL : 0);
    double *hvk = lpn + NC;             // wide-vector path: [NC][2] results of a chain's own pass handed to the cooperative phase
    ChainState *sts = reinterpret_cast<ChainState *>(hvk + 2 * NC);
    int *spec = reinterpret_cast<int *>(sts + NC);
    volatile int *slow = spec + NC;     // round stamp (round + 1) of the last round in which some chain did something long (see stage Z)
    int *thoff = spec + NC + 4;         // wide-vector path: offset of each chain's live theta row (0 or V_TH2 - V_TH rows)
    int *hvy = thoff + NC;              // wide-vector path: chains that the cooperative phase finishes (bdrt_nuts_wide.h)
    double *ublk = reinterpret_cast<double *>(hvy + NC);     // register path: [NC][16] uniforms of sixteen consecutive leaves (stage S1)
    constexpr bool SPEC = NJ > 16 || (NJ > 11 && MODE < 2);      // wide-vector path (the S1 evaluator leaves room for 16 elements per lane)
    const int TH2OFF = (V_TH2 - V_TH) * NC * DS;

    double *V = a.vecs + (size_t)wg * V_COUNT * NC * DS;
    auto row = [&](int v) -> double * { return V + ((size_t)v * NC + c) * DS; };   // this chain's row of vector v

    // (MODE 3 / 4: TA carries the basis functions per lane of the evaluator's instantiation, 0 = 6)
    static_assert(!TA || MODE >= 2, "the generator table belongs to the S1 tile");
    constexpr int KUX = TA == 0 ? 6 : TA;
    if (TA && MODE == 2) s1_toep_init(P, smem);
    if (tid < NC) {
        const int u = su[col_slot(tid)];
        if (u >= 0) chain_state_copy(sts[tid], a.states[u]);      // (member by member: a struct assignment goes through scratch)
        else { memset(&sts[tid], 0, sizeof(ChainState)); sts[tid].phase = PH_DONE; }
        spec[tid] = u >= 0 ? sts[tid].spec : 0;
        thoff[tid] = (SPEC && sts[tid].thsel) ? TH2OFF : 0;
        if (tid == 0) *slow = 0;
    }
    __syncthreads();
    ChainState &s = sts[c];
    const Philox rng = {np.seed_lo, np.seed_hi, (uint32_t)s.chain_id};

    TileIO io;
    io.theta = V + (size_t)V_TH * NC * DS; io.t_sc = DS; io.t_sj = 1;
    io.t_off = SPEC ? thoff : nullptr;
    io.grad = V + (size_t)V_G * NC * DS; io.g_sc = DS; io.g_sj = 1;
    if (MODE == 2) {
        // fast S1 path: theta lives in LDS for the whole launch (read by the tile, updated in place by the leapfrog), and
        // the tile leaves the gradient in the chain's private LDS row instead of storing it to HBM
        io.theta = thl; io.t_sc = DSL;
        io.grad = nullptr;
    }
    io.lp = lpn;
    io.spec = spec;
    io.nvalid = NC;            // padded columns carry a DONE state and finite vectors
    io.jacobian = 1;
    io.Z_hat = nullptr; io.sigma_tot = nullptr; io.params = nullptr;
    io.prof = (PROF && a.prof) ? a.prof + (size_t)wg * 32 : nullptr;
    long long tnp = 0;
#define BDRT_NUTS_PROF(slot) do { if (!BDRT_PROF_FINE && io.prof && tid == 0) { const long long t_ = clock64(); io.prof[slot] += t_ - tnp; tnp = t_; } } while (0)
    // per-wave stage times (slots 17..24, summed over the 8 waves): where each wave spends the round, incl. the barrier wait
    long long twv = 0;
#define BDRT_WAVE_PROF(slot) do { if (io.prof && lane == 0) { const long long t_ = clock64(); atomicAdd((unsigned long long *)&io.prof[slot], (unsigned long long)(t_ - twv)); twv = t_; } } while (0)

    // -DBDRT_PROF_FINE=1 (tools/build_variant.sh): the thread-0 slots 0..16 of the phase profile become wave-summed timers of the
    // register path's sub-stages (bench.py --phase-profile prints them with BDRT_BENCH_FINE=1)
#ifndef BDRT_PROF_FINE
#define BDRT_PROF_FINE 0
#endif
    long long tfn = 0;
    // (accumulated in LDS -- the wide-vector path's `hvk` cells, free on the register path -- and added to the global slots at the
    // end of the launch: a global atomic per stamp puts a memory operation in front of every stage's s_waitcnt)
    typedef __attribute__((address_space(3))) unsigned long long *lds_u64;
    const lds_u64 fine = (lds_u64)(unsigned long long *)hvk;
    if (BDRT_PROF_FINE && tid < 2 * NC) fine[tid] = 0ull;
#define BDRT_FINE(slot) do { if (BDRT_PROF_FINE && io.prof && lane == 0) { const long long t_ = clock64(); __hip_atomic_fetch_add(&fine[slot], (unsigned long long)(t_ - tfn), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); tfn = t_; } } while (0)
#define BDRT_FINE_COUNT(slot, cond) do { if (BDRT_PROF_FINE && io.prof) { const bool w_ = __builtin_amdgcn_ballot_w64(cond) != 0; if (lane == 0 && w_) __hip_atomic_fetch_add(&fine[slot], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } } while (0)
    const int ndbg = BDRT_NUTS_ABLATE ? P.dbg : 0;          // timing ablations (-DBDRT_NUTS_ABLATE=1 builds only; results wrong on purpose)
    unsigned long long my_leaps = 0;
    double *TH = row(V_TH) + (SPEC && s.thsel ? TH2OFF : 0), *Pm = row(V_P), *G = row(V_G), *MI = row(V_MINV);

    // half kick + drift of the first evaluation of a freshly created sampler (afterwards every loop body ends with the
    // kick/drift of the NEXT evaluation, reusing p, g, Minv from registers: stage A' below)
    if (!s.kicked) {
        const int ph = s.phase;
        const double e = ph == PH_EPS ? s.eps : (ph == PH_TREE ? s.dir * s.eps : 0.0);
        if (ph == PH_INIT || ph == PH_EPS || ph == PH_TREE) {
            for (int j = l32; j < D; j += 32) {
                const double p = Pm[j] + 0.5 * e * G[j];
                Pm[j] = p;
                TH[j] += e * MI[j] * p;
            }
        }
        s.kicked = 1;
    }
    if (MODE == 2) {
        double *thr = thl + (size_t)c * DSL;
        for (int j = l32; j < DSL; j += 32) thr[j] = j < D ? TH[j] : 0.0;
        TH = thr;                                           // every later theta access of this launch is an LDS access
        G = s1_grad_row(P, smem, c);                        // where the tile leaves d lp / d theta
    }
    // Is any chain of the workgroup still running?  Voted at a barrier per round boundary -- except on the fast path (MODE 2),
    // where between the end of the backward GEMM of one round and the first barrier of the next evaluation a wave touches only
    // its own chains' LDS (theta / gradient rows, its columns of the operand tile): there the round boundary has NO barrier.
    // Each chain leaves an activity flag, every thread reads the 16 flags right after the evaluator's first barrier, and a wave
    // that finishes its bookkeeping early goes on with the next point's exponentials while another one still closes a subtree.
    // (The verdict is one round late: a workgroup whose last chain has just finished runs one idle round.)
    constexpr bool FREE_RUN = MODE == 2 && !SPEC;
    int *actf = hvy;                    // (the wide-vector path's `hvy` slots are free on this path)
    int any_act;
    {
        const int ph = s.phase;
        const bool running = ph == PH_INIT || ph == PH_EPS || ph == PH_TREE;
        if (FREE_RUN && l32 == 0) actf[c] = running ? 1 : 0;
        any_act = __syncthreads_or(running);
    }

    // Register path: what the top of a round needs of the chain state -- phase, leaf index, signed step size -- is carried in
    // registers and refreshed from LDS at the end of a round only when the chain did anything but a plain continuing leaf.
    int ph_c = 0, leaf_c = 0;
    double e_c = 0.0;
    auto refresh_carried = [&]() {
        ph_c = s.phase; leaf_c = s.leaf;
        e_c = ph_c == PH_EPS ? s.eps : (ph_c == PH_TREE ? s.dir * s.eps : 0.0);
    };
    if constexpr (!SPEC) refresh_carried();
    // RESIDENT (MODE 2, <= 11 elements per lane): the chain's momentum and inverse metric stay in the registers of its half-wave
    // for the whole launch -- loaded here, written back behind the loop; the stages that set a new momentum (next start point,
    // next doubling) set the registers as well as the row.  Two row reads and one row write less per leapfrog (33 of ~100
    // vector-memory instructions per wave and round, 8 of 19 KB), no wait for them in front of the kick.
    constexpr bool RESIDENT = BDRT_RESIDENT && MODE == 2 && !SPEC && NJ <= 11;
    constexpr int NR = RESIDENT ? NJ : 1;
    double pr_[NR], mir_[NR];
    if constexpr (RESIDENT) {
#pragma unroll
        for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; pr_[m] = valid ? Pm[j] : 0.0; mir_[m] = valid ? MI[j] : 1.0; }
    }

    // The kernel's by-value arguments (48 dwords of NutsParams / NutsArgs) are read where they are used, from the kernarg segment,
    // through a pointer the optimiser cannot see through: kept in scalar registers across the round loop they -- with the masks and
    // addresses of the evaluator -- oversubscribe the 102 SGPRs, and an SGPR spilled to a VGPR lane costs a v_writelane / v_readlane
    // (VALU instructions: 700 of them in the round loop's listing, profiles/r05/isa_mix_nuts_kernel.txt) per use.
    struct KArgs { const DevProblem *Pp; NutsParams np; NutsArgs a; };
    typedef const __attribute__((address_space(4))) KArgs *kargs_ptr;
    kargs_ptr ka = (kargs_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const int n_rounds = a.rounds;
    for (int round = 0; round < n_rounds; ++round) {
#if BDRT_KARG_RELOAD
        __asm__ volatile("" : "+s"(ka));
        const NutsParams &np = *(const NutsParams *)&ka->np;
        const NutsArgs &a = *(const NutsArgs *)&ka->a;
#endif
        // keep per-lane address arithmetic inside the loop (see the note in bdrt_tile_s1.h): hoisted, it is spilled
        __asm__ volatile("" : "+v"(c), "+v"(l32));
        if (SPEC && tid < NC) hvy[tid] = 0;
        int ph0;
        double e;
        if constexpr (SPEC) {
            ph0 = s.phase;
            e = ph0 == PH_EPS ? s.eps : (ph0 == PH_TREE ? s.dir * s.eps : 0.0);
        } else { ph0 = ph_c; e = e_c; }
        const bool act = ph0 == PH_INIT || ph0 == PH_EPS || ph0 == PH_TREE;
        if (!any_act) break;
        if (io.prof && tid == 0) tnp = clock64();
        if (io.prof && lane == 0) twv = clock64();

        // ---- B: log-posterior + gradient at the new point (MFMA tile) ------------------------------------
        // state of this chain that the stages after the evaluation need: momentum, inverse metric, and (odd leaves) the
        // momentum of the previous leaf.  The fast path issues these loads from inside the evaluation, right before its
        // backward GEMM; otherwise stage C loads them.
        // Wide parameter vectors (SPEC: outlier error model, several distributions) do not fit in registers next to the
        // evaluator: they take the wide-vector path below, which streams the chain's rows from HBM.
        constexpr int NA = SPEC ? 1 : NJ;
        // elements per lane that lie below D whatever D is: MODE 2 picks the smallest NJ of {4, 6, 7, 11, 16} with D <= 32 NJ (s1_nj)
        constexpr int MSAFE = MODE != 2 ? 0 : (NJ == 6 ? 4 : (NJ == 7 ? 6 : (NJ == 11 ? 7 : (NJ == 16 ? 11 : 0))));
        double p_[NA], g_[NA], mi_[NA];
#pragma unroll
        for (int m = 0; m < NA; ++m) { p_[m] = 0.0; g_[m] = 0.0; mi_[m] = 1.0; }
        if constexpr (RESIDENT) {
#pragma unroll
            for (int m = 0; m < NJ; ++m) { p_[m] = pr_[m]; mi_[m] = mir_[m]; }
        }
        auto load_state = [&]() {
            if constexpr (!SPEC && !RESIDENT) {
                if (act) {
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; p_[m] = Pm[j]; mi_[m] = MI[j]; }
                }
            }
        };
        int any_next = 0;
        if (MODE == 2) {
            auto read_flags = [&]() {
                if constexpr (FREE_RUN) {
                    typedef int iv4 __attribute__((ext_vector_type(4)));
                    typedef const __attribute__((address_space(3))) iv4 *lds_i4;
                    int v = 0;
#pragma unroll
                    for (int q = 0; q < NC / 4; ++q) { const iv4 f = ((lds_i4)actf)[q]; v |= f.x | f.y | f.z | f.w; }
                    any_next = v;
                }
            };
            // (momentum and inverse metric are requested right before the backward GEMM: their HBM round trip runs under its MFMAs
            // instead of in front of stage C)
            constexpr bool EARLY = BDRT_NUTS_EARLY_STATE && NJ <= 11;      // (16 elements per lane: the rows would be spilled)
            // (Measured and dropped: touching one dword per line of the checkpoint rows that this leaf's merges of levels 1..3 will read,
            // so that they wait in L2 by stage D: 0.767 -> 0.751, conditional or not.)
            auto early_state = [&]() { if constexpr (EARLY) load_state(); };
            // (D = 2 K + 9 here: K <= 59 / 91 / 107 for 4 / 6 / 7 elements per lane)
            constexpr int KU = NJ <= 4 ? 2 : (NJ <= 6 ? 3 : (NJ <= 7 ? 4 : 6));
            logp_grad_tile_s1<true, 32, decltype(early_state), decltype(read_flags), TA, KU>(P, io, smem, early_state, read_flags);
            if constexpr (!EARLY) load_state();
        }
        else if (MODE == 3) { logp_grad_tile_s1<false, 32, NoHook, NoHook, 0, KUX>(P, io, smem); load_state(); }
        else if (MODE == 4) { logp_grad_tile_hw<KUX, PROF>(P, io, smem); load_state(); }
        else { logp_grad_tile<MODE == 1>(P, io, smem); load_state(); }
        if (io.prof && tid == 0) tnp = clock64();
        BDRT_WAVE_PROF(17);
        const long long t_post0 = (io.prof && SPEC) ? clock64() : 0;      // wide-vector path: post-evaluation time of this wave (slots 0..3)
        if constexpr (SPEC) {
            // ================= wide-vector path (see bdrt_nuts_wide.h) ======================================================
            // Phase P, per chain: a tree leaf that is not the last of its subtree and closes at most two sub-subtrees (7 of 8
            // non-final leaves; all of this is known from the leaf index before the evaluation) is finished by its own
            // half-wave in ONE pass over the chain's rows: second half kick, kinetic energy, the merges of levels 0 and 1 with
            // their U-turn dot products, the new checkpoint, and -- speculatively -- the half kick + drift of the NEXT leapfrog
            // (momentum in place, theta into the chain's other theta row, so that the evaluated point stays readable for the
            // proposal copy).  If the verdict is "continue" the two theta rows swap roles.  If not (divergence or U-turn: the
            // transition ends, nothing of this leaf's theta / momentum is looked at again, and the checkpoint written ahead
            // sits in a level that had no waiting sibling) the chain joins the others in phase H.
            const int dir_now = s.dir;
            const int leaf_now = s.leaf;
            (void)dir_now;
            constexpr int MB = NJ % 8 == 0 ? 8 : 9;
            static_assert(NJ % MB == 0, "chunk size must divide NJ");
            constexpr int NS2 = (32 * NJ + 63) / 64;            // 64-element slices of a row
            constexpr int LMAX = 4;                             // merge levels a chain's own pass can take (6 measured: phase H -4 k cycles, phase P +4 k)
            typedef double dv2 __attribute__((ext_vector_type(2)));
            auto ld2 = [](const double *q) -> dv2 { return *reinterpret_cast<const dv2 *>(q); };
            auto st2 = [](double *q, dv2 v) { *reinterpret_cast<dv2 *>(q) = v; };
            if (act) {
                const bool treeph = ph0 == PH_TREE;
                int nmf = 0;
                if (treeph) { while ((leaf_now >> nmf) & 1) ++nmf; }
                const bool lastf = treeph && leaf_now == s.nleaves - 1;
                if (treeph && !lastf && nmf <= LMAX) {
                    // waiting siblings of the levels this leaf closes (level 0: a single leaf, rho = its momentum); a level the
                    // chain does not close aliases the momentum row: loads stay unconditional (a branch around a load makes the
                    // compiler wait for everything in flight at the join)
                    const double *RLp[LMAX], *PLp[LMAX];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        RLp[l] = l < nmf ? row(l == 0 ? V_CKP : V_CKC + l) : Pm;
                        PLp[l] = l < nmf ? row(V_CKP + l) : Pm;
                    }
                    double *RHD = row(V_CKC + (nmf > 0 ? nmf : 1)), *FRD = row(V_CKP + nmf);
                    double *THN = row(V_TH) + (s.thsel ? 0 : TH2OFF);
                    double kin = 0.0, nonfin = 0.0;
                    double uma[LMAX], umb[LMAX];
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) { uma[l] = 0.0; umb[l] = 0.0; }
                    // Straight-line variants of the pass, chosen per WAVE by the number of levels NL either of its two chains
                    // closes; SB = 64-element slices per batch of loads (fewer when more rows are in flight).
                    // 16-byte accesses: lane l handles the element pairs 2 l + 64 sl (the pass is bound by the number of
                    // vector-memory instructions the texture addresser gets through, not by bytes: half as many this way).
                    auto pass = [&](auto nlc, auto sbc) {
                        constexpr int NL = decltype(nlc)::value, SBM = decltype(sbc)::value;
                        auto chunk = [&](int sb, auto sbn) {                 // slices sb .. sb + SB of the rows
                            constexpr int SB = decltype(sbn)::value;
                            dv2 a_[SB], b_[SB], c_[SB], t_[SB], lr_[NL > 0 ? NL : 1][SB], lp_[NL > 0 ? NL : 1][SB];
#pragma unroll
                            for (int ss = 0; ss < SB; ++ss) {
                                {
                                    const int j = 2 * l32 + 64 * (sb + ss);
                                    const int jc = j < DS ? j : 0;               // (the last slice of a row may be half empty)
                                    a_[ss] = ld2(Pm + jc); b_[ss] = ld2(G + jc); c_[ss] = ld2(MI + jc); t_[ss] = ld2(TH + jc);
#pragma unroll
                                    for (int l = 0; l < NL; ++l) {
                                        lp_[l][ss] = ld2(PLp[l] + jc);
                                        if (l > 0) lr_[l][ss] = ld2(RLp[l] + jc);
                                    }
                                }
                            }
#pragma unroll
                            for (int ss = 0; ss < SB; ++ss) {
                                {
                                    const int j = 2 * l32 + 64 * (sb + ss);
                                    dv2 pn2, tn2, rho2, first2;
#pragma unroll
                                    for (int hh = 0; hh < 2; ++hh) {
                                        const double p = a_[ss][hh] + 0.5 * e * b_[ss][hh];
                                        const double pn = p + 0.5 * e * b_[ss][hh];
                                        const bool live = j + hh < D;
                                        pn2[hh] = pn;
                                        tn2[hh] = t_[ss][hh] + e * c_[ss][hh] * pn;
                                        if (live) {
                                            kin += c_[ss][hh] * p * p;
                                            nonfin += isfinite(b_[ss][hh]) ? 0.0 : 1.0;
                                        }
                                        double rho = p, first = p;
#pragma unroll
                                        for (int l = 0; l < NL; ++l) {
                                            if (l < nmf) {
                                                const double plv = lp_[l][ss][hh];
                                                rho = (l == 0 ? plv : lr_[l][ss][hh]) + rho;
                                                first = plv;
                                                if (live) {
                                                    uma[l] += c_[ss][hh] * plv * rho;
                                                    umb[l] += c_[ss][hh] * p * rho;
                                                }
                                            }
                                        }
                                        rho2[hh] = rho; first2[hh] = first;
                                    }
                                    if (j < DS) {                                // (rows are padded to DS: whole pairs are stored)
                                        st2(Pm + j, pn2);
                                        st2(THN + j, tn2);
                                        if (NL > 0 && nmf > 0) st2(RHD + j, rho2);
                                        st2(FRD + j, first2);
                                    }
                                }
                            }
                        };
                        // (no bounds tests inside: a branch around loads costs the whole batch its overlap)
#pragma unroll 1
                        for (int sb = 0; sb + SBM <= NS2; sb += SBM) chunk(sb, std::integral_constant<int, SBM>{});
                        if constexpr (NS2 % SBM != 0) chunk(NS2 - NS2 % SBM, std::integral_constant<int, NS2 % SBM>{});
                    };
                    {
                        const bool w1 = __builtin_amdgcn_ballot_w64(nmf > 0) != 0, w2 = __builtin_amdgcn_ballot_w64(nmf > 1) != 0,
                                   w3 = __builtin_amdgcn_ballot_w64(nmf > 2) != 0;
                        if (w3) pass(std::integral_constant<int, LMAX>{}, std::integral_constant<int, 2>{});
                        else if (w2) pass(std::integral_constant<int, 2>{}, std::integral_constant<int, 4>{});
                        else if (w1) pass(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{});
                        else pass(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
                    }
                    kin = 0.5 * half_sum(kin);
                    nonfin = half_sum(nonfin);
                    bool ok = true;
#pragma unroll
                    for (int l = 0; l < LMAX; ++l) {
                        const double ua = half_sum(uma[l]), ub = half_sum(umb[l]);
                        if (l < nmf) ok = ok && (ua > 0.0) && (ub > 0.0);
                    }
                    // verdict (Stan base_nuts::build_tree: divergence test at the leaf, compute_criterion at every merge)
                    const double lp = lpn[c];
                    double h = -lp + kin;
                    if (isnan(h)) h = INFINITY;
                    const double H0 = s.H0;
                    const bool divergent = (h - H0) > np.max_deltaH;
                    if (!divergent && ok) {
                        if (l32 == 0) my_leaps += 1;
                        s.n_leap_iter = s.n_leap_iter + 1;
                        const double w = H0 - h;
                        s.sum_metro = s.sum_metro + (w > 0.0 ? 1.0 : BDRT_NUTS_EXP(w));
                        // uniform sampling inside the new subtree: keep leaf i with probability w_i / W_i
                        const double u = rng_uniform(rng, (uint32_t)leaf_now, RNG_LEAF, (uint32_t)s.depth, 0, (uint32_t)s.iter);
                        double lsw_new;
                        const bool joins = nuts_leaf_joins(s.lsw_sub, w, u, lsw_new);       // (bdrt_nuts_device.h)
                        const bool copyq = leaf_now == 0 || joins;
                        s.lsw_sub = lsw_new;
                        s.leaf = leaf_now + 1;
                        if (copyq) {
                            s.lpq = lp;
                            double *THQ = row(V_THQ), *GQ = row(V_GQ);
#pragma unroll 1
                            for (int mb = 0; mb < NJ; mb += MB) {
                                double a_[MB], b_[MB];
#pragma unroll
                                for (int mm = 0; mm < MB; ++mm) { const int j = l32 + 32 * (mb + mm); a_[mm] = TH[j]; b_[mm] = G[j]; }
#pragma unroll
                                for (int mm = 0; mm < MB; ++mm) __asm__ volatile("" : "+v"(a_[mm]), "+v"(b_[mm]));
#pragma unroll
                                for (int mm = 0; mm < MB; ++mm) {
                                    const int j = l32 + 32 * (mb + mm);
                                    if (j < D) { THQ[j] = a_[mm]; GQ[j] = b_[mm]; }
                                }
                            }
                        }
                        const int sel = s.thsel ^ 1;               // the drifted theta row becomes the live one
                        s.thsel = sel;
                        TH = row(V_TH) + (sel ? TH2OFF : 0);
                        thoff[c] = sel ? TH2OFF : 0;
                    } else {
                        hvy[c] = 2;
                        hvk[2 * c] = kin; hvk[2 * c + 1] = nonfin;
                    }
                } else {
                    hvy[c] = 1;
                }
            }
            BDRT_WAVE_PROF(18);
            __syncthreads();
            // Phase H: the chains with more to do, one after the other, each by all 512 threads
            {
                WideCtx wx;
                wx.P = Pp; wx.np = &np; wx.a = &a; wx.V = V; wx.smem = smem; wx.sts = sts; wx.lpn = lpn; wx.hvy = hvy; wx.hvk = hvk;
                wx.prof = io.prof; wx.D = D; wx.DS = DS; wx.TH2OFF = TH2OFF; wx.c0 = 0; wx.nvalid = 0; wx.slot_unit = su; wx.ncol = NC; wx.hot_base = nullptr; wx.hot_slot = nullptr;
                for (int hc = 0; hc < NC; ++hc) {
                    const int kind = hvy[hc];
                    if (kind) wide_coop_tail<(NJ * 32 + WIDE_NT - 1) / WIDE_NT>(wx, hc, kind == 2, my_leaps, tid);
                }
            }
            TH = row(V_TH) + (s.thsel ? TH2OFF : 0);
            BDRT_WAVE_PROF(20);
            if (io.prof) {
                int cls = hvy[c] ? 1 : 0;
                const int other = __shfl_xor(cls, 32);
                cls = cls > other ? cls : other;
                if (lane == 0) {
                    atomicAdd((unsigned long long *)&io.prof[2 * cls], (unsigned long long)(clock64() - t_post0));
                    atomicAdd((unsigned long long *)&io.prof[2 * cls + 1], 1ull);
                }
            }
            {
                // (also separates this round's reads of hvy from the reset at the top of the next round)
                const int ph = sts[c].phase;
                any_act = __syncthreads_or(ph == PH_INIT || ph == PH_EPS || ph == PH_TREE);
            }
            BDRT_WAVE_PROF(23);
            BDRT_NUTS_PROF(10);
        } else {
        // ================= register path: the chain's p, g, Minv stay in the registers of its half-wave ====================

        // ---- C: second half kick, kinetic energy, finiteness of the gradient -----------------------------
        // (p, g, Minv of this chain stay in registers from here to the end of stage D)
        if (BDRT_PROF_FINE && io.prof && lane == 0) tfn = clock64();
        double kin = 0.0, nonfin = 0.0;
        bool bad_g = false;
        const int leaf_now = leaf_c;
        // The part of the chain state that a plain tree leaf reads, requested in ONE batch here and used in stage S1 (read where
        // it is used, every member is an LDS round trip of its own behind the branches of the scalar logic: eight in a row).
        struct LeafHot { int nleaves, depth, iter, n_leap_iter; double H0, lsw_sub, sum_metro, lpq; } h;
        h.nleaves = s.nleaves; h.depth = s.depth; h.iter = s.iter; h.n_leap_iter = s.n_leap_iter;
        h.H0 = s.H0; h.lsw_sub = s.lsw_sub; h.sum_metro = s.sum_metro; h.lpq = 0.0;
        const double lp_now = lpn[c];
        // The leaf's index says which waiting sub-subtrees it closes (nm_pre trailing one bits: levels 0 .. nm_pre - 1), before
        // anything is evaluated.  Level 0 (the previous leaf's momentum) is requested here by every chain (even leaves alias the
        // chain's own momentum row: the load stays unconditional); a wave with a chain that closes two levels or more requests
        // level 1 as well, right behind the kick (below) -- both round trips run under the scalar logic of stage S1 instead of
        // one after the other behind it; the levels from 2 on come two per round trip in stage D.  (Round 3 requested level 1
        // with level 0 in every wave: spilled; levels 1 AND 2 here: 49 registers spilled.)
        constexpr bool PRE0 = NJ <= 11;                  // (16 elements per lane leave no registers for it)
        const bool treeph = act && ph0 == PH_TREE;
        const int nm_pre = treeph ? __builtin_ctz(~(unsigned)leaf_now) : 0;
        const bool deep = PRE0 && __builtin_amdgcn_ballot_w64(nm_pre >= 2) != 0;       // (wave-uniform)
        double pl0_[PRE0 ? NA : 1], alr_[PRE0 ? NA : 1], alp_[PRE0 ? NA : 1], blr_[PRE0 ? NA : 1], blp_[PRE0 ? NA : 1];
        if constexpr (PRE0) {
            const double *PL0 = (RESIDENT || nm_pre >= 1) ? row(V_CKP) : Pm;      // (RESIDENT: the same row whatever the leaf)
#pragma unroll
            for (int m = 0; m < NJ; ++m) pl0_[m] = (BDRT_NT & 1) ? __builtin_nontemporal_load(PL0 + l32 + 32 * m) : PL0[l32 + 32 * m];
        }
        // the rows of level l into a buffer (a level the chain does not close aliases its momentum row: no load under a condition)
        auto request_level = [&](int l, int nmx, double (&lr_)[PRE0 ? NA : 1], double (&lp_)[PRE0 ? NA : 1]) {
            if constexpr (PRE0) {
                const double *RA = l < nmx ? row(V_CKC + l) : Pm, *PA = l < nmx ? row(V_CKP + l) : Pm;
#pragma unroll
                for (int m = 0; m < NJ; ++m) {
                    const int j = l32 + 32 * m;
                    if (BDRT_NT & 1) { lr_[m] = __builtin_nontemporal_load(RA + j); lp_[m] = __builtin_nontemporal_load(PA + j); }
                    else { lr_[m] = RA[j]; lp_[m] = PA[j]; }
                }
            }
        };
        // No element of the passes below is guarded by `j < D`: every state row is DS = 32 NJ long and its elements from D on are
        // zero from bdrt_sampler_create on (the inverse metric's: finite), every statement maps zeros to zeros, so the lanes
        // beyond D add exact zeros to the sums and store zeros -- eleven `s_and_saveexec / s_cbranch_execz` pairs per pass
        // otherwise.  Exceptions: the gradient row in LDS (MODE 2: the evaluator's transit values lie behind element D), the
        // draws (rows of D) and the random re-initialisation.
        BDRT_FINE(0);
        // MODE 2: the gradient is NOT held in registers from here on -- it stays readable in the chain's LDS row until the next
        // evaluation, and the stages that want it again (proposal copy, end of a subtree, the next kick) read it there: 22
        // registers less across stages S1 and D, which is what leaves room for the rows of level 1.
        constexpr bool GLDS = MODE == 2 && BDRT_GLDS;
        auto load_grad = [&](double (&g)[NA]) {
#pragma unroll
            for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; g[m] = G[j]; }
            if constexpr (MODE == 2) {
                const int lim = D - l32;
#pragma unroll
                for (int m = MSAFE; m < NJ; ++m) g[m] = 32 * m < lim ? g[m] : 0.0;
            }
        };
        if (act) {
            double gc_[NA];
            load_grad(gc_);
#pragma unroll
            for (int m = 0; m < NJ; ++m) {
                const double p = p_[m] + 0.5 * e * gc_[m];
                p_[m] = p;                            // written to memory by stage A' / E at the end of the body
                kin += mi_[m] * p * p;
                bad_g = bad_g || !isfinite(gc_[m]);
                if constexpr (!GLDS) g_[m] = gc_[m];
            }
        }
        BDRT_FINE(1);
        // (behind the first use of p / Minv: in front of it the kick would wait for these rows too -- s_waitcnt counts in order)
        if constexpr (PRE0 && BDRT_PRE1) request_level(1, nm_pre, alr_, alp_);     // (under `if (deep)`: 150 bytes of spills)
        kin = 0.5 * half_sum(kin);
        // a non-finite gradient entry anywhere in the chain's half-wave: one ballot instead of a second butterfly
        {
            // (the half is picked through `c`, which is opaque per round: a mask made from `lane` is hoisted out of the round loop and spilled)
            const unsigned long long bl = __builtin_amdgcn_ballot_w64(bad_g);
            const unsigned half = (c & 1) ? (unsigned)(bl >> 32) : (unsigned)bl;
            nonfin = half ? 1.0 : 0.0;
        }
        BDRT_FINE(2);
        BDRT_NUTS_PROF(11);
        BDRT_WAVE_PROF(18);

        // ---- S1: per-chain scalar logic after the evaluation (redundant in the 32 lanes of the chain) --------
        bool copyq = false, cur2s = false, tree = false, last = false;
        bool upds = false, welf = false, wend = false;
        int nm = 0, endt = 0, next = 0, draw = -1, dir_now = 0;
        double wn = 0.0;
        if (act) {
            const double lp = lp_now;
            const bool finite_pt = isfinite(lp) && nonfin == 0.0;
            if (ph0 == PH_INIT) {
                if (finite_pt) {
                    s.lps = lp;
                    cur2s = true;
                    s.phase = PH_EPS; s.eps_dir = 0; s.eps_trials = 0;
                    next = 3;
                } else {
                    const int att = s.init_attempt + 1;
                    s.init_attempt = att;
                    if (att >= 100) s.phase = PH_FAILED;
                    else next = 4;
                }
            } else if (ph0 == PH_EPS) {
                // Stan base_hmc::init_stepsize
                if (l32 == 0) my_leaps += 1;
                next = nuts_stepsize_trial(s, np, lp, kin);
            } else {   // PH_TREE: one new leaf
                if (l32 == 0) my_leaps += 1;
                // The leaf's uniform (a Philox block: ~110 VALU instructions, the same in all 32 lanes) for SIXTEEN consecutive
                // leaves at once: lane i of the chain draws leaf 16 b + i into the chain's LDS cells when a block of sixteen
                // begins (a new subtree begins at leaf 0) and in the first round of a launch; same counters, same values.
                if ((leaf_now & 15) == 0 || round == 0) {
                    if (l32 < 16)
                        ublk[c * 16 + l32] = rng_uniform(rng, (uint32_t)((leaf_now & ~15) + l32), RNG_LEAF, (uint32_t)h.depth, 0, (uint32_t)h.iter);
                    lds_wave_sync();
                }
                const double u_leaf = ublk[c * 16 + (leaf_now & 15)];
                nuts_tree_leaf(h, np, rng, lp, kin, leaf_now, copyq, tree, nm, last, endt, &u_leaf);      // (bdrt_nuts_device.h)
                s.n_leap_iter = h.n_leap_iter; s.sum_metro = h.sum_metro;
                if (tree) s.lsw_sub = h.lsw_sub;
                if (copyq) s.lpq = h.lpq;
                if (last) *slow = round + 1;                        // closing a subtree (and maybe the transition): a long round
            }
        }
        BDRT_FINE(3);
        BDRT_NUTS_PROF(12);
        BDRT_WAVE_PROF(19);

        // ---- D: proposal copy, checkpoints, running rho, U-turn tests, subtree close ----------------------------
        if ((copyq || cur2s) && !(ndbg & 64)) {
            double *THQ = row(V_THQ), *GQ = row(V_GQ), *THS = row(V_THS), *GS = row(V_GS);
            {
                double th_[NJ];
#pragma unroll
                for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; th_[m] = TH[j]; }
                if constexpr (GLDS) load_grad(g_);
#pragma unroll
                for (int m = 0; m < NJ; ++m) {
                    const int j = l32 + 32 * m;
                    if (copyq) {
                        if (BDRT_NT & 2) { __builtin_nontemporal_store(th_[m], THQ + j); __builtin_nontemporal_store(g_[m], GQ + j); }
                        else { THQ[j] = th_[m]; GQ[j] = g_[m]; }
                    }
                    if (cur2s) { THS[j] = th_[m]; GS[j] = g_[m]; }
                }
            }
        }
        BDRT_FINE(4);
        BDRT_FINE_COUNT(14, copyq || cur2s);
        bool cont = false;                                  // a plain leaf: the trajectory goes on from the point just evaluated
        if (tree) {
            // Binary-counter bookkeeping of the new subtree (leaves arrive in time order): level l of the checkpoint rows
            // holds the completed left sub-subtree of 2^l leaves that still waits for its sibling -- rho (sum of momenta,
            // V_CKC + l) and the momentum of its first leaf (V_CKP + l; at level 0 the two coincide and only V_CKP is used).
            // A leaf with nm trailing one bits closes nm sub-subtrees: each merge is one generalised U-turn test
            // (Stan 2.19 base_nuts::build_tree: compute_criterion(p_sharp_left, p_sharp_right, rho_subtree)).  The sums
            // associate exactly like the recursion does.
            double rc_[NA], cpl_[NA];                  // rho / first momentum of the sub-subtree that ends at this leaf
            bool ok = true;
            {
#pragma unroll
                for (int m = 0; m < NJ; ++m) { rc_[m] = p_[m]; cpl_[m] = p_[m]; }
                auto merge = [&](const double (&lr_)[PRE0 ? NA : 1], const double (&lp_)[PRE0 ? NA : 1]) {
                    double a0 = 0.0, a1 = 0.0;
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const double rho = lr_[m] + rc_[m];
                        a0 += mi_[m] * lp_[m] * rho;
                        a1 += mi_[m] * p_[m] * rho;
                        rc_[m] = rho;
                        cpl_[m] = lp_[m];
                    }
                    a0 = half_sum(a0); a1 = half_sum(a1);
                    ok = ok && (a0 > 0.0) && (a1 > 0.0);
                };
                if constexpr (PRE0) {
                    if (nm > 0) merge(pl0_, pl0_);
                    BDRT_FINE(5);
                    BDRT_FINE_COUNT(15, nm > 1);
                    if (deep && !(ndbg & 16)) {
                        // level 1 is here (requested in stage C); the levels beyond it one per round trip
                        if (BDRT_PRE1 && nm > 1) merge(alr_, alp_);
                        for (int l = BDRT_PRE1 ? 2 : 1; __builtin_amdgcn_ballot_w64(l < nm) != 0; l += 1) {
                            request_level(l, nm, alr_, alp_);
                            if (l < nm) merge(alr_, alp_);
                        }
                    }
                } else {
                    for (int l = 0; l < nm; ++l) {
                        double lr_[NJ], lp_[NJ];
                        const double *RL = row(l == 0 ? V_CKP : V_CKC + l), *PL = row(V_CKP + l);
#pragma unroll
                        for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; lr_[m] = RL[j]; lp_[m] = PL[j]; }
                        double a0 = 0.0, a1 = 0.0;
#pragma unroll
                        for (int m = 0; m < NJ; ++m) {
                            const double rho = lr_[m] + rc_[m];
                            a0 += mi_[m] * lp_[m] * rho;
                            a1 += mi_[m] * p_[m] * rho;
                            rc_[m] = rho;
                            cpl_[m] = lp_[m];
                        }
                        a0 = half_sum(a0); a1 = half_sum(a1);
                        ok = ok && (a0 > 0.0) && (a1 > 0.0);
                    }
                }
                BDRT_FINE(6);
                if (ok && !last && !(ndbg & 32)) {
                    // the sub-subtree of 2^nm leaves that ends here becomes the waiting left sibling of level nm
                    double *PLn = row(V_CKP + nm), *RLn = row(V_CKC + nm);
#pragma unroll
                    for (int m = 0; m < NJ; ++m) PLn[l32 + 32 * m] = cpl_[m];
                    if (nm > 0) {
#pragma unroll
                        for (int m = 0; m < NJ; ++m) RLn[l32 + 32 * m] = rc_[m];
                    }
                }
            }
            BDRT_FINE(7);
            BDRT_FINE_COUNT(16, ok && last);
            if (!ok) {
                endt = 1;                                   // U-turn inside the new subtree: discard it, stop
            } else if (last) {
                // subtree complete and valid (Stan base_nuts::transition after build_tree): extend the trajectory
                dir_now = s.dir;
                double *RHO = row(V_RHO);
                double *THE = row(dir_now > 0 ? V_THP : V_THM), *PE = row(dir_now > 0 ? V_PP : V_PM);
                double *GE = row(dir_now > 0 ? V_GP : V_GM);
                const double *PO = row(dir_now > 0 ? V_PM : V_PP);     // momentum at the other end
                double t0 = 0.0, t1 = 0.0;
                {
                    double rt_[NJ], po_[NJ], th_[NJ];
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m, jj = j;
                        rt_[m] = RHO[jj]; po_[m] = PO[jj]; th_[m] = TH[jj];
                    }
                    if constexpr (GLDS) load_grad(g_);
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m;
                        const double rt = rt_[m] + rc_[m];
                        RHO[j] = rt;
                        THE[j] = th_[m]; PE[j] = p_[m]; GE[j] = g_[m];
                        t0 += mi_[m] * po_[m] * rt;
                        t1 += mi_[m] * p_[m] * rt;
                    }
                }
                t0 = half_sum(t0); t1 = half_sum(t1);
                const int depth = s.depth + 1;
                s.depth = depth;
                const double lsw = s.lsw, lsw_sub = s.lsw_sub;
                bool take;
                if (lsw_sub > lsw) take = true;
                else take = rng_uniform(rng, 0, RNG_TOP, (uint32_t)depth, 0, (uint32_t)s.iter) < BDRT_NUTS_EXP(lsw_sub - lsw);
                if (take) { upds = true; s.lps = s.lpq; }
                s.lsw = log_sum_exp2(lsw, lsw_sub);
                const bool keep_going = (t0 > 0.0) && (t1 > 0.0);
                if (!keep_going || depth >= np.max_depth) endt = 1;
                else {
                    // next doubling
                    s.dir = rng_uniform(rng, 0, RNG_DIRECTION, (uint32_t)depth, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
                    s.leaf = 0; s.nleaves = 1 << depth; s.lsw_sub = -INFINITY;
                    next = 2;
                }
            } else {
                s.leaf = leaf_now + 1;
                cont = true;
            }
        }
        BDRT_FINE(8);
        if (endt) {
            next = nuts_transition_end(s, np, endt, draw, welf, wend, wn);     // (bdrt_nuts_device.h)
            if (draw >= 0 && a.lp_draws && valid && l32 == 0) a.lp_draws[(size_t)unit * np.n_draws + draw] = s.lps;
        }
        BDRT_FINE(9);
        BDRT_NUTS_PROF(14);
        BDRT_WAVE_PROF(20);

        // ---- A' (common case): the trajectory continues from the point just evaluated: half kick + drift of the NEXT
        //      leapfrog with p, g, Minv still in registers (theta is the only vector read; p is written once per leapfrog).
        //      Same direction, same step size: the signed step of this round.
        if (cont) {
            double th_[NJ];
#pragma unroll
            for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; th_[m] = TH[j]; }
            if constexpr (GLDS) load_grad(g_);
#pragma unroll
            for (int m = 0; m < NJ; ++m) {
                const int j = l32 + 32 * m;
                const double p = p_[m] + 0.5 * e * g_[m];
                if constexpr (RESIDENT) p_[m] = p; else if (!(ndbg & 128)) Pm[j] = p;
                TH[j] = th_[m] + e * mi_[m] * p;
            }
        }

        BDRT_FINE(10);
        // ---- Z: momentum normals of the NEXT transition, ahead of time.  When some chain of the workgroup closes a subtree
        //      or a transition this round, every other wave would only wait for it at the round barrier; a wave whose own
        //      two chains are on the plain path uses that time to draw the 2*ceil(D/2) normals its chains need at their
        //      next start point (Philox counters depend on (seed, chain, iteration) only, so the values are the same
        //      whenever they are computed).  Takes the RNG (the largest part of a new start point) off the critical path.
        {
            const bool heavy = !act || ph0 != PH_TREE || last || endt != 0 || next != 0;
            const bool wave_heavy = __builtin_amdgcn_ballot_w64(heavy) != 0;
            if (!wave_heavy && *slow == round + 1 && s.z_iter != s.iter + 1) {
                double *ZN = row(V_ZN);
                const uint32_t it1 = (uint32_t)(s.iter + 1);
#pragma unroll
                for (int mp = 0; mp < (NJ + 1) / 2; ++mp) {
                    const int i = l32 + 32 * mp;
                    if (2 * i < D) {
                        double z0, z1;
                        rng_normal_pair(rng, (uint32_t)i, RNG_MOMENTUM, 0u, it1, z0, z1);
                        *reinterpret_cast<double2 *>(ZN + 2 * i) = make_double2(z0, z1);
                    }
                }
                s.z_iter = (int)it1;
            }
        }
        BDRT_FINE(11);
        BDRT_WAVE_PROF(21);
        // ---- E: sample update, metric adaptation, draw output, and the start of the next leapfrog when the trajectory does
        //      not simply continue (new transition, next doubling, step-size search, re-initialisation).  One batch of
        //      loads per case, everything else in registers, including the half kick + drift of the next evaluation.
        {
            if (upds || welf || wend || draw >= 0 || next) {
                const uint32_t iter = (uint32_t)s.iter, trial = (uint32_t)s.eps_trials, att = (uint32_t)s.init_attempt;
                double *THS = row(V_THS), *GS = row(V_GS);
                double ths_[NJ], gs_[NJ];
#pragma unroll
                for (int m = 0; m < NJ; ++m) { ths_[m] = 0.0; gs_[m] = 0.0; }
                if (upds || welf || wend || draw >= 0 || next == 1 || next == 3) {     // (a plain doubling needs none of it)
                    // current sample: the proposal of the tree if it was just accepted, else the stored sample
                    const double *ST = upds ? row(V_THQ) : THS, *SG = upds ? row(V_GQ) : GS;
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; ths_[m] = ST[j]; gs_[m] = SG[j]; }
                }
                if (upds) {
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; THS[j] = ths_[m]; GS[j] = gs_[m]; }
                }
                if (welf || wend) {
                    double *WM = row(V_WMEAN), *W2 = row(V_WM2);
                    double wm_[NJ], w2_[NJ];
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; wm_[m] = WM[j]; w2_[m] = W2[j]; }
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m;
                        double mean = wm_[m], m2 = w2_[m];
                        if (welf) {            // Welford (stan::math::welford_var_estimator)
                            const double delta = ths_[m] - mean;
                            mean += delta / wn;
                            m2 += (ths_[m] - mean) * delta;
                        }
                        if (wend) {            // var_adaptation::learn_variance
                            const double var = wn > 1.0 ? m2 / (wn - 1.0) : 0.0;
                            mi_[m] = (wn / (wn + 5.0)) * var + 1e-3 * (5.0 / (wn + 5.0));
                            MI[j] = mi_[m];
                            mean = 0.0; m2 = 0.0;
                        }
                        WM[j] = mean; W2[j] = m2;
                    }
                }
                if (draw >= 0 && valid) {
                    double *dr = a.draws + ((size_t)unit * np.n_draws + draw) * D;
#pragma unroll
                    for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; if (j < D) dr[j] = ths_[m]; }
                }
                if (next == 1 || next == 3) {
                    // fresh momentum p ~ N(0, M), M = diag(1/Minv).  Normals 2i and 2i+1 share one Philox block and one
                    // Box-Muller transform: a lane produces PAIRS and the chain's scratch row (its share of the idle tile
                    // LDS) turns them into the lane's own elements j = l32 + 32 m.
                    double *zrow = MODE == 2 ? s1_grad_row(P, smem, c) : smem + (size_t)c * (tile_doubles / NC);
                    const bool have_z = next == 1 && s.z_iter == (int)iter;      // stage Z of an earlier round did the work
                    double z_[NJ];
                    if (have_z) {
                        const double *ZN = row(V_ZN);
#pragma unroll
                        for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; z_[m] = ZN[j]; }
                    } else {
#pragma unroll
                        for (int mp = 0; mp < (NJ + 1) / 2; ++mp) {
                            const int i = l32 + 32 * mp;
                            if (2 * i < D) {
                                double z0, z1;
                                rng_normal_pair(rng, (uint32_t)i, next == 1 ? RNG_MOMENTUM : RNG_EPS_MOMENTUM, next == 1 ? 0u : trial, iter, z0, z1);
                                zrow[2 * i] = z0; zrow[2 * i + 1] = z1;
                            }
                        }
                        lds_wave_sync();
#pragma unroll
                        for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; z_[m] = j < D ? zrow[j] : 0.0; }
                        lds_wave_sync();
                    }
                    double kin0 = 0.0;
                    double pn_[NJ];
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m;
                        const double p = j < D ? z_[m] / sqrt(mi_[m]) : 0.0;
                        pn_[m] = p;
                        kin0 += mi_[m] * p * p;
                    }
                    // Hamiltonian at the start point; a new transition also resets the tree and draws its first direction
                    kin0 = half_sum(kin0);
                    s.H0 = -s.lps + 0.5 * kin0;
                    if (next == 1) {
                        s.lsw = 0.0; s.lsw_sub = -INFINITY; s.depth = 0; s.leaf = 0; s.nleaves = 1;
                        s.n_leap_iter = 0; s.sum_metro = 0.0;
                        s.dir = rng_uniform(rng, 0, RNG_DIRECTION, 0, 0, (uint32_t)s.iter) > 0.5 ? 1 : -1;
                    }
                    const double e1 = next == 1 ? s.dir * s.eps : s.eps;
                    double *rTHM = row(V_THM), *rTHP = row(V_THP), *rPM = row(V_PM), *rPP = row(V_PP), *rGM = row(V_GM),
                           *rGP = row(V_GP), *rRHO = row(V_RHO);
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m;
                        const double p = pn_[m];
                        if (next == 1) {
                            rTHM[j] = ths_[m]; rTHP[j] = ths_[m];
                            rPM[j] = p; rPP[j] = p;
                            rGM[j] = gs_[m]; rGP[j] = gs_[m];
                            rRHO[j] = p;
                        }
                        const double pk = p + 0.5 * e1 * gs_[m];
                        if constexpr (RESIDENT) p_[m] = pk; else Pm[j] = pk;
                        TH[j] = ths_[m] + e1 * mi_[m] * pk;
                    }
                } else if (next == 2) {
                    // continue from the trajectory end in the new direction
                    const int dir = s.dir;
                    const double e1 = dir * s.eps;
                    double et_[NJ], ep_[NJ], eg_[NJ];
                    if (dir == dir_now) {
                        // same direction again: that end of the trajectory is the point just evaluated (theta, p, grad are here)
                        if constexpr (GLDS) load_grad(g_);
#pragma unroll
                        for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; et_[m] = TH[j]; ep_[m] = p_[m]; eg_[m] = g_[m]; }
                    } else {
                        const double *ET = row(dir > 0 ? V_THP : V_THM), *EP = row(dir > 0 ? V_PP : V_PM), *EG = row(dir > 0 ? V_GP : V_GM);
#pragma unroll
                        for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; et_[m] = ET[j]; ep_[m] = EP[j]; eg_[m] = EG[j]; }
                    }
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m;
                        const double pk = ep_[m] + 0.5 * e1 * eg_[m];
                        if constexpr (RESIDENT) p_[m] = pk; else Pm[j] = pk;
                        TH[j] = et_[m] + e1 * mi_[m] * pk;
                    }
                } else if (next == 4) {
#pragma unroll
                    for (int m = 0; m < NJ; ++m) {
                        const int j = l32 + 32 * m;
                        if (j < D) {
                            TH[j] = np.init_radius * (2.0 * rng_uniform(rng, (uint32_t)j, RNG_INIT, 0, att, 0) - 1.0);
                            if constexpr (RESIDENT) p_[m] = 0.0; else Pm[j] = 0.0;
                        }
                    }
                }
            }
        }
        BDRT_FINE(12);
        BDRT_NUTS_PROF(15);
        BDRT_NUTS_PROF(16);

        BDRT_WAVE_PROF(22);
        if (io.prof && lane == 0 && next) atomicAdd((unsigned long long *)&io.prof[24], 1ull);
        if constexpr (RESIDENT) {
#pragma unroll
            for (int m = 0; m < NJ; ++m) { pr_[m] = p_[m]; mir_[m] = mi_[m]; }
        }
        {
            if (cont) leaf_c = leaf_now + 1; else refresh_carried();
            const int ph = ph_c;
            const bool running = ph == PH_INIT || ph == PH_EPS || ph == PH_TREE;
            if constexpr (FREE_RUN) {
                if (l32 == 0) actf[c] = running ? 1 : 0;
                any_act = any_next;                         // (the same 16 flags were read by every thread after this round's first barrier)
            } else {
                any_act = __syncthreads_or(running);
            }
        }
        BDRT_WAVE_PROF(23);
        BDRT_NUTS_PROF(10);
        }   // register path
    }

    // ---- write the chain states back -----------------------------------------------------------------------------
    __syncthreads();
    if (BDRT_PROF_FINE && io.prof && tid < 17) atomicAdd((unsigned long long *)&io.prof[tid], (unsigned long long)fine[tid]);
    if (MODE == 2) {
        double *THg = row(V_TH);
        for (int j = l32; j < D; j += 32) THg[j] = TH[j];
    }
    if constexpr (RESIDENT) {
        if (valid) {
#pragma unroll
            for (int m = 0; m < NJ; ++m) { const int j = l32 + 32 * m; Pm[j] = pr_[m]; }
        }
    }
    if (l32 == 0 && valid) chain_state_copy(a.states[unit], s);
    {
        unsigned long long x = my_leaps;      // non-zero only in lane 0 of each half-wave
        x += __shfl_xor(x, 32);
        if (lane == 0 && x) atomicAdd(a.leap_counter, x);
    }
    {
        const int ph = s.phase;
        const bool running = ph == PH_INIT || ph == PH_EPS || ph == PH_TREE;
        const int busy = __syncthreads_or(running);
        if (tid == 0 && !busy) atomicAdd(a.done_counter, 1);
        if (a.active_counter && l32 == 0 && running && valid) atomicAdd(a.active_counter, 1);
    }
}

// ---- the instantiations, in groups of similar compile time: bdrt_nuts_k<g>.hip defines group g, bdrt_nuts.hip (the host side)
//      declares them all `extern template`
#define BDRT_NUTS16_G0(X) X(11, 2, 1) X(11, 2, 2) X(11, 2, 0) X(4, 2, 2) X(4, 2, 0) X(6, 2, 1)
#define BDRT_NUTS16_G1(X) X(16, 2, 2) X(16, 2, 1) X(16, 2, 0) X(6, 2, 2) X(6, 2, 0) X(7, 2, 2) X(7, 2, 0)
#define BDRT_NUTS16_G2(X) X(11, 1, 0) X(11, 0, 0) X(16, 1, 0) X(16, 0, 0) X(27, 1, 0) X(27, 0, 0)
#define BDRT_NUTS16_G3(X) X(11, 3, 0) X(16, 3, 0) X(11, 3, 3) X(16, 3, 3) X(11, 3, 4) X(16, 3, 4)
#define BDRT_NUTS16_G4(X) X(11, 4, 0) X(16, 4, 0) X(27, 4, 0) X(11, 4, 3) X(16, 4, 3) X(27, 4, 3) X(11, 4, 4) X(16, 4, 4) X(27, 4, 4)
// (group 5: the profiling instantiations -- the headline family and BASELINE config 5's; other kernels leave the profile empty)
#define BDRT_NUTS16_G5(X) X(11, 2, 1) X(11, 2, 2) X(7, 2, 2) X(27, 4, 0)
#define BDRT_NUTS16_DEFINE(NJ_, MODE_, TA_) template __global__ void nuts_kernel<NJ_, MODE_, TA_, false>(const DevProblem *__restrict__, NutsParams, NutsArgs);
#define BDRT_NUTS16_DECLARE(NJ_, MODE_, TA_) extern template __global__ void nuts_kernel<NJ_, MODE_, TA_, false>(const DevProblem *__restrict__, NutsParams, NutsArgs);
#define BDRT_NUTS16_DEFINE_PROF(NJ_, MODE_, TA_) template __global__ void nuts_kernel<NJ_, MODE_, TA_, true>(const DevProblem *__restrict__, NutsParams, NutsArgs);
#define BDRT_NUTS16_DECLARE_PROF(NJ_, MODE_, TA_) extern template __global__ void nuts_kernel<NJ_, MODE_, TA_, true>(const DevProblem *__restrict__, NutsParams, NutsArgs);

}  // namespace bdrt
