// Work-list instantiations of pick_fast_kernel with ordered fallbacks AND candidate masks: the requests pick_quad_kernel<..., MASKED, TOPK>
// deferred.  One addressing variant (BIG: rows through wave-uniform 64-bit bases -- correct for an index of any size; these requests are
// the rare ones).
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW>
static const void* fast_wl_topk_masked(bool has_l, bool p_first) {
  if (has_l) return p_first ? (const void*)pick_fast_kernel<LW, 6, true, true, true, true, true, false, true, true>
                            : (const void*)pick_fast_kernel<LW, 6, true, true, false, true, true, false, true, true>;
  return (const void*)pick_fast_kernel<LW, 6, false, true, false, true, true, false, true, true>;
}
const void* pick_fast_wl_topk_masked_u16(bool has_l, bool p_first) { return fast_wl_topk_masked<uint16_t>(has_l, p_first); }
const void* pick_fast_wl_topk_masked_u32(bool has_l, bool p_first) { return fast_wl_topk_masked<uint32_t>(has_l, p_first); }
const void* pick_fast_wl_topk_masked_u64(bool has_l, bool p_first) { return fast_wl_topk_masked<uint64_t>(has_l, p_first); }

}  // namespace eppk
