// Work-list instantiations of pick_fast_kernel: the requests pick_quad_kernel deferred (see eppk_kernels.hip.h: KWork).
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW, bool BIG, bool MASKED>
static const void* fast_wl_ptr(bool has_l, bool p_first) {
  if (has_l) return p_first ? (const void*)pick_fast_kernel<LW, 6, true, true, true, MASKED, BIG, false, false, true>
                            : (const void*)pick_fast_kernel<LW, 6, true, true, false, MASKED, BIG, false, false, true>;
  return (const void*)pick_fast_kernel<LW, 6, false, true, false, MASKED, BIG, false, false, true>;
}
template <typename LW>
static const void* fast_wl(bool has_l, bool p_first, bool big, bool masked) {
  if (masked) return big ? fast_wl_ptr<LW, true, true>(has_l, p_first) : fast_wl_ptr<LW, false, true>(has_l, p_first);
  return big ? fast_wl_ptr<LW, true, false>(has_l, p_first) : fast_wl_ptr<LW, false, false>(has_l, p_first);
}
const void* pick_fast_wl_u16(bool has_l, bool p_first, bool big, bool masked) { return fast_wl<uint16_t>(has_l, p_first, big, masked); }
const void* pick_fast_wl_u32(bool has_l, bool p_first, bool big, bool masked) { return fast_wl<uint32_t>(has_l, p_first, big, masked); }
const void* pick_fast_wl_u64(bool has_l, bool p_first, bool big, bool masked) { return fast_wl<uint64_t>(has_l, p_first, big, masked); }

}  // namespace eppk
