#include "bdrt_host.h"
using namespace bdrt;
extern "C" {
int bdrt_gram(const double *, const double *, int, int, const double *, const double *, double *, double *) { set_error("not built yet"); return -99; }
int bdrt_qp_box(const double *, const double *, const double *, int, double *, double *) { set_error("not built yet"); return -99; }
}
