// pick_quad_kernel<..., TAIL> instantiations: the four-requests-per-wavefront kernel whose workgroups score what they deferred
// themselves (pick_fast_body over their own work-list segments, in a function that is never inlined) -- no second launch (see
// eppk_kernels.hip.h).  This unit: unmasked single picks, and the dispatchers over the other three units.
#include "eppk_kernels.hip.h"
#define EPPK_QUAD_TAIL_UNIT pick_quad_tail_plain
#define EPPK_QUAD_TAIL_MASKED false
#define EPPK_QUAD_TAIL_TOPK false
namespace eppk { const void* pick_quad_tail_plain(int lw_bytes, bool has_l, bool p_first); }
#include "eppk_pick_inst.hip.h"

namespace eppk {

static const void* quad_tail_any(int lw_bytes, bool has_l, bool p_first, bool masked, bool topk) {
  if (masked) return topk ? pick_quad_tail_topk_masked(lw_bytes, has_l, p_first) : pick_quad_tail_masked(lw_bytes, has_l, p_first);
  return topk ? pick_quad_tail_topk(lw_bytes, has_l, p_first) : pick_quad_tail_plain(lw_bytes, has_l, p_first);
}
const void* pick_quad_tail_u16(bool has_l, bool p_first, bool masked, bool topk) { return quad_tail_any(2, has_l, p_first, masked, topk); }
const void* pick_quad_tail_u32(bool has_l, bool p_first, bool masked, bool topk) { return quad_tail_any(4, has_l, p_first, masked, topk); }
const void* pick_quad_tail_u64(bool has_l, bool p_first, bool masked, bool topk) { return quad_tail_any(8, has_l, p_first, masked, topk); }

}  // namespace eppk
