// pick_resident_kernel instantiations: the resident workgroup behind the small-batch latency path (EPPK_RESIDENT=1; see
// eppk_kernels.hip.h).  Fused chains with a PREFIX scorer and at most 63 blocks per request (what pick_fast_kernel's NPL = 6 form serves).
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW>
static const void* resident_ptr(bool has_l, bool p_first) {
  if (has_l) return p_first ? (const void*)pick_resident_kernel<LW, true, true> : (const void*)pick_resident_kernel<LW, true, false>;
  return (const void*)pick_resident_kernel<LW, false, false>;
}
const void* pick_resident(int lw_bytes, bool has_l, bool p_first) {
  return lw_bytes == 2 ? resident_ptr<uint16_t>(has_l, p_first) : lw_bytes == 4 ? resident_ptr<uint32_t>(has_l, p_first) : resident_ptr<uint64_t>(has_l, p_first);
}

}  // namespace eppk
