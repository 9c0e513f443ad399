// Instantiations of the fused row-streaming pass for the Op9cv operator policy (see fused_kernel.h).
#include "fused_kernel.h"

namespace mgcmt {

void launch_fused_op9cv(hipStream_t s, const fused::FusedArgs& a, int multicolour, int nsweep, int flags, int k) {
  using namespace fused;
  if (multicolour) {
    if (nsweep == 1) launch_variant<Op9cv, kFourColour, 1>(s, a, flags, k);
    else launch_variant<Op9cv, kFourColour, 2>(s, a, flags, k);
  } else {
    if (nsweep == 1) launch_variant<Op9cv, kJacobi, 1>(s, a, flags, k);
    else launch_variant<Op9cv, kJacobi, 2>(s, a, flags, k);
  }
}

}  // namespace mgcmt
