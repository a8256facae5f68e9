// pick_resident_kernel<..., QUAD, LEARN>: small batches whose picks the index learns at once -- the resident workgroup answers, then
// applies the post-route update itself (see eppk_pick_resident.hip, eppk_kernels.hip.h: resident_learn_update); also the dispatcher
// over the two LEARN units.
#define EPPK_RESIDENT_QUAD true
#define EPPK_RESIDENT_LEARN true
#define EPPK_RESIDENT_FN pick_resident_quad_learn_plain
namespace eppk { const void* pick_resident_quad_learn_plain(int lw_bytes, bool has_l, bool p_first); }
#include "eppk_pick_resident.hip"

namespace eppk {
const void* pick_resident_quad_learn(int lw_bytes, bool has_l, bool p_first, bool masked) {
  return masked ? pick_resident_quad_learn_masked(lw_bytes, has_l, p_first) : pick_resident_quad_learn_plain(lw_bytes, has_l, p_first);
}
}  // namespace eppk
