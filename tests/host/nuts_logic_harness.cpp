// Host build of the sampler kernels' shared scalar logic (bayes_drt_amd/csrc/bdrt_nuts_device.h): prints what a chain's
// adaptation bookkeeping does over a warm-up, for tests/test_host_nuts_logic.py to compare with Stan 2.19's documented
// behaviour (windowed_adaptation, stepsize_adaptation) and with a direct evaluation of the dual-averaging recurrences.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define __host__
#define __device__
using std::isnan; using std::isfinite; using std::exp; using std::log; using std::log1p; using std::sqrt; using std::pow; using std::sin; using std::cos;
#include "../../bayes_drt_amd/csrc/bdrt_nuts_device.h"

using namespace bdrt;

int main(int argc, char **argv)
{
    const int warmup = argc > 1 ? atoi(argv[1]) : 1000, n_draws = argc > 2 ? atoi(argv[2]) : 5;
    NutsParams np;
    memset(&np, 0, sizeof(np));
    np.warmup = warmup; np.n_draws = n_draws; np.max_depth = 10;
    np.delta = 0.8; np.gamma = 0.05; np.t0 = 10.0; np.kappa = 0.75; np.stepsize0 = 1.0; np.max_deltaH = 1000.0;
    ChainState s;
    memset(&s, 0, sizeof(s));
    window_init(s, warmup, 75, 50, 25);
    s.phase = PH_TREE; s.eps = 1.0; s.da_mu = log(10.0 * 1.0);
    da_restart(s);
    printf("buffers %d %d %d\n", s.init_buffer, s.term_buffer, s.base_window);
    for (int it = 0; it < warmup + n_draws; ++it) {
        // a synthetic transition: 7 leapfrogs, acceptance statistic that depends on the iteration
        s.n_leap_iter = 7;
        const double acc = 0.5 + 0.45 * sin(0.37 * it);
        s.sum_metro = 7 * acc;
        s.depth = 3;
        int draw = -1; bool welf = false, wend = false; double wn = 0.0;
        const int next = nuts_transition_end(s, np, it % 97 == 13 ? 2 : 1, draw, welf, wend, wn);
        printf("it %d next %d draw %d welf %d wend %d wn %.0f eps %.17g xbar %.17g counter %d phase %d\n", it, next, draw, (int)welf,
               (int)wend, wn, s.eps, s.da_xbar, s.da_counter, s.phase);
        if (next == 3) {
            // the kernels run the step-size search here; the test only needs the restart of the dual averaging it ends with
            s.phase = PH_TREE; s.da_mu = log(10.0 * s.eps); da_restart(s);
        }
        s.sum_metro = 0.0;
    }
    printf("totals n_post %d n_div %d n_leap_total %lld\n", s.n_post, s.n_div, s.n_leap_total);
    return 0;
}
