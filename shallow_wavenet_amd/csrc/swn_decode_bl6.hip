// placeholder: the register/LDS-resident BL6-class decode kernel is added next.
#include <hip/hip_runtime.h>
#include "swn_geom.hpp"

extern "C" int swn_decode_bl6_try(const swn_net_desc*, const float*, const float*, int, int, int,
                                  const float*, const void*, void*, float*, void*) {
    return SWN_E_UNSUPPORTED;
}
