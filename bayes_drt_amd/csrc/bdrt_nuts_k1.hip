// bdrt_nuts_k1.hip -- instantiations of the 16-chain NUTS kernel (bdrt_nuts16.h), group 1
#include "bdrt_nuts16.h"
namespace bdrt {
BDRT_NUTS16_G1(BDRT_NUTS16_DEFINE)
}
