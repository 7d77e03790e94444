// bdrt_nuts_k5.hip -- the profiling instantiations of the 16-chain NUTS kernel (bdrt_nuts16.h, PROF = true)
#include "bdrt_nuts16.h"
namespace bdrt {
BDRT_NUTS16_G5(BDRT_NUTS16_DEFINE_PROF)
}
