// pick_quad_kernel<..., TAIL> instantiations (see eppk_pick_quad_tail.hip): MASKED = true, TOPK = true.
#include "eppk_kernels.hip.h"
#define EPPK_QUAD_TAIL_UNIT pick_quad_tail_topk_masked
#define EPPK_QUAD_TAIL_MASKED true
#define EPPK_QUAD_TAIL_TOPK true
#include "eppk_pick_inst.hip.h"
