// ns_q2.hip -- specialised 3D Q2/Q1 streaming kernel (placeholder until the
// sweep kernel lands; the generic path is used meanwhile).
#include "kernels.hpp"

namespace adaflo_hip
{
  bool q2_supported(const adaflo_ctx *) { return false; }
  int  q2_prepare_state(adaflo_ctx *) { return ADAFLO_EUNSUPPORTED; }
  int  launch_ns_vmult_q2(adaflo_ctx *, int, double *, double *, const double *, const double *)
  {
    return ADAFLO_EUNSUPPORTED;
  }
} // namespace adaflo_hip
