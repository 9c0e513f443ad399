// Instantiations of the wide-window fused pass (fused_wide.h): constant 5-point operator, weighted Jacobi.
#include "fused_wide.h"

namespace mgcmt {

bool launch_fused_wide(hipStream_t s, const fused::FusedArgs& a, int nsweep, int flags, int k) {
  using namespace fused;
  if (nsweep == 1) return launch_wide_variant<1>(s, a, flags, k);
  if (nsweep == 2) return launch_wide_variant<2>(s, a, flags, k);
  return false;
}

}  // namespace mgcmt
