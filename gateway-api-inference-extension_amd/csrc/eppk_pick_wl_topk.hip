// Work-list instantiations of pick_fast_kernel with ordered fallbacks: the requests pick_quad_kernel<..., TOPK> deferred.
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW, bool BIG>
static const void* fast_wl_topk_ptr(bool has_l, bool p_first) {
  if (has_l) return p_first ? (const void*)pick_fast_kernel<LW, 6, true, true, true, false, BIG, false, true, true>
                            : (const void*)pick_fast_kernel<LW, 6, true, true, false, false, BIG, false, true, true>;
  return (const void*)pick_fast_kernel<LW, 6, false, true, false, false, BIG, false, true, true>;
}
template <typename LW>
static const void* fast_wl_topk(bool has_l, bool p_first, bool big) {
  return big ? fast_wl_topk_ptr<LW, true>(has_l, p_first) : fast_wl_topk_ptr<LW, false>(has_l, p_first);
}
const void* pick_fast_wl_topk_u16(bool has_l, bool p_first, bool big) { return fast_wl_topk<uint16_t>(has_l, p_first, big); }
const void* pick_fast_wl_topk_u32(bool has_l, bool p_first, bool big) { return fast_wl_topk<uint32_t>(has_l, p_first, big); }
const void* pick_fast_wl_topk_u64(bool has_l, bool p_first, bool big) { return fast_wl_topk<uint64_t>(has_l, p_first, big); }

}  // namespace eppk
