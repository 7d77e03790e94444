// bdrt_nuts_k0.hip -- instantiations of the 16-chain NUTS kernel (bdrt_nuts16.h), group 0
#include "bdrt_nuts16.h"
namespace bdrt {
BDRT_NUTS16_G0(BDRT_NUTS16_DEFINE)
}
