#include "bdrt_host.h"
using namespace bdrt;
extern "C" {
void bdrt_nuts_defaults(bdrt_nuts_control *c)
{
    c->adapt_delta = 0.9; c->adapt_t0 = 10; c->adapt_gamma = 0.05; c->adapt_kappa = 0.75; c->max_treedepth = 10;
    c->init_buffer = 75; c->term_buffer = 50; c->base_window = 25; c->init_radius = 2; c->max_deltaH = 1000; c->stepsize0 = 1;
}
bdrt_sampler *bdrt_sampler_create(bdrt_problem *, int, const int *, const int *, int, int, uint64_t, const double *, const bdrt_nuts_control *) { set_error("not built yet"); return nullptr; }
void bdrt_sampler_destroy(bdrt_sampler *) {}
int bdrt_sampler_advance(bdrt_sampler *, int, int *) { return -99; }
int bdrt_sampler_sync(bdrt_sampler *) { return -99; }
int bdrt_sampler_run(bdrt_sampler *) { return -99; }
int bdrt_sampler_results(bdrt_sampler *, double *, double *, bdrt_chain_diag *) { return -99; }
int64_t bdrt_sampler_total_leapfrogs(bdrt_sampler *) { return -99; }
int bdrt_sampler_kernel_time(bdrt_sampler *, double *, int64_t *, int) { return -99; }
int bdrt_sample(bdrt_problem *, int, const int *, const int *, int, int, uint64_t, const double *, const bdrt_nuts_control *, double *, double *, bdrt_chain_diag *) { return -99; }
}
