// pick_quad_kernel instantiations (four requests per wavefront; see eppk_kernels.hip.h).
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW, bool MASKED, bool TOPK>
static const void* quad_ptr(bool has_l, bool p_first) {
  if (has_l) return p_first ? (const void*)pick_quad_kernel<LW, true, true, MASKED, TOPK> : (const void*)pick_quad_kernel<LW, true, false, MASKED, TOPK>;
  return (const void*)pick_quad_kernel<LW, false, false, MASKED, TOPK>;
}
template <typename LW>
static const void* quad_ptr(bool has_l, bool p_first, bool masked, bool topk) {
  if (masked) return topk ? quad_ptr<LW, true, true>(has_l, p_first) : quad_ptr<LW, true, false>(has_l, p_first);
  return topk ? quad_ptr<LW, false, true>(has_l, p_first) : quad_ptr<LW, false, false>(has_l, p_first);
}
const void* pick_quad_u16(bool has_l, bool p_first, bool masked, bool topk) { return quad_ptr<uint16_t>(has_l, p_first, masked, topk); }
const void* pick_quad_u32(bool has_l, bool p_first, bool masked, bool topk) { return quad_ptr<uint32_t>(has_l, p_first, masked, topk); }
const void* pick_quad_u64(bool has_l, bool p_first, bool masked, bool topk) { return quad_ptr<uint64_t>(has_l, p_first, masked, topk); }

}  // namespace eppk
