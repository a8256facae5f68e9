// pick_resident_kernel instantiations: the resident workgroup behind the small-batch latency path (EPPK_RESIDENT=1; see
// eppk_kernels.hip.h).  Fused chains with a PREFIX scorer and at most 63 blocks per request (what pick_fast_kernel's NPL = 6 form serves).
// Two forms, one unit each (this file is also included by eppk_pick_resident_quad.hip with EPPK_RESIDENT_QUAD = true): pick_fast_kernel's
// body alone, and pick_quad_kernel's body with the fast body's work-list form behind it.
#ifndef EPPK_RESIDENT_QUAD
#define EPPK_RESIDENT_QUAD false
#define EPPK_RESIDENT_FN pick_resident
#endif
#ifndef EPPK_RESIDENT_MASKED      // the variants of the quad form (eppk_pick_resident_quad_{masked,topk,topk_masked}.hip)
#define EPPK_RESIDENT_MASKED false
#endif
#ifndef EPPK_RESIDENT_TOPK
#define EPPK_RESIDENT_TOPK false
#endif
#ifndef EPPK_RESIDENT_LEARN       // ... followed by the post-route index update (eppk_pick_resident_quad_learn[_masked].hip)
#define EPPK_RESIDENT_LEARN false
#endif
#include "eppk_kernels.hip.h"
#include "eppk_pick_inst.hip.h"

namespace eppk {

template <typename LW>
static const void* resident_ptr(bool has_l, bool p_first) {
  constexpr bool Q = EPPK_RESIDENT_QUAD, M = EPPK_RESIDENT_MASKED, T = EPPK_RESIDENT_TOPK, L = EPPK_RESIDENT_LEARN;
  if (has_l) return p_first ? (const void*)pick_resident_kernel<LW, true, true, Q, M, T, L> : (const void*)pick_resident_kernel<LW, true, false, Q, M, T, L>;
  return (const void*)pick_resident_kernel<LW, false, false, Q, M, T, L>;
}
const void* EPPK_RESIDENT_FN(int lw_bytes, bool has_l, bool p_first) {
  return lw_bytes == 2 ? resident_ptr<uint16_t>(has_l, p_first) : lw_bytes == 4 ? resident_ptr<uint32_t>(has_l, p_first) : resident_ptr<uint64_t>(has_l, p_first);
}

}  // namespace eppk
