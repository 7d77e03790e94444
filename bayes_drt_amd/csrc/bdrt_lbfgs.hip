#include "bdrt_host.h"
using namespace bdrt;
extern "C" {
void bdrt_opt_defaults(bdrt_opt_options *o)
{
    o->max_iter = 50000; o->history = 5; o->init_alpha = 1e-3; o->tol_obj = 1e-12; o->tol_rel_obj = 1e4;
    o->tol_grad = 1e-8; o->tol_rel_grad = 1e7; o->tol_param = 1e-8;
}
int bdrt_optimize(bdrt_problem *, const double *, const int *, int, const bdrt_opt_options *, double *, bdrt_opt_report *)
{ set_error("bdrt_optimize: not built yet"); return -99; }
}
