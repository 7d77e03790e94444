// bdrt_nuts_k4.hip -- instantiations of the 16-chain NUTS kernel (bdrt_nuts16.h), group 4
#include "bdrt_nuts16.h"
namespace bdrt {
BDRT_NUTS16_G4(BDRT_NUTS16_DEFINE)
}
