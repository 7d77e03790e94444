// bdrt_nuts_k3.hip -- instantiations of the 16-chain NUTS kernel (bdrt_nuts16.h), group 3
#include "bdrt_nuts16.h"
namespace bdrt {
BDRT_NUTS16_G3(BDRT_NUTS16_DEFINE)
}
