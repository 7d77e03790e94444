// bdrt_nuts_k2.hip -- instantiations of the 16-chain NUTS kernel (bdrt_nuts16.h), group 2
#include "bdrt_nuts16.h"
namespace bdrt {
BDRT_NUTS16_G2(BDRT_NUTS16_DEFINE)
}
