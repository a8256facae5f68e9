// pick_resident_kernel<..., QUAD, TOPK>: the resident workgroup for small batches with ordered fallbacks (see eppk_pick_resident.hip);
// also the dispatcher over the three variant units.
#define EPPK_RESIDENT_QUAD true
#define EPPK_RESIDENT_TOPK true
#define EPPK_RESIDENT_FN pick_resident_quad_topk
#include "eppk_pick_resident.hip"

namespace eppk {
const void* pick_resident_quad_variant(int lw_bytes, bool has_l, bool p_first, bool masked, bool topk) {
  if (topk) return masked ? pick_resident_quad_topk_masked(lw_bytes, has_l, p_first) : pick_resident_quad_topk(lw_bytes, has_l, p_first);
  return masked ? pick_resident_quad_masked(lw_bytes, has_l, p_first) : pick_resident_quad(lw_bytes, has_l, p_first);
}
}  // namespace eppk
