// pick_quad_kernel<..., TAIL> instantiations: the four-requests-per-wavefront kernel whose last workgroup scores what the launch
// deferred itself (pick_fast_body as a grid of one) -- no second launch (see eppk_kernels.hip.h).  Unmasked picks only.
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW>
static const void* quad_tail_ptr(bool has_l, bool p_first) {
  if (has_l) return p_first ? (const void*)pick_quad_kernel<LW, true, true, false, false, true> : (const void*)pick_quad_kernel<LW, true, false, false, false, true>;
  return (const void*)pick_quad_kernel<LW, false, false, false, false, true>;
}
const void* pick_quad_tail_u16(bool has_l, bool p_first) { return quad_tail_ptr<uint16_t>(has_l, p_first); }
const void* pick_quad_tail_u32(bool has_l, bool p_first) { return quad_tail_ptr<uint32_t>(has_l, p_first); }
const void* pick_quad_tail_u64(bool has_l, bool p_first) { return quad_tail_ptr<uint64_t>(has_l, p_first); }

}  // namespace eppk
