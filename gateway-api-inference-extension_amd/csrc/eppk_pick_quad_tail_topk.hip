// pick_quad_kernel<..., TAIL> instantiations (see eppk_pick_quad_tail.hip): MASKED = false, TOPK = true.
#include "eppk_kernels.hip.h"
#define EPPK_QUAD_TAIL_UNIT pick_quad_tail_topk
#define EPPK_QUAD_TAIL_MASKED false
#define EPPK_QUAD_TAIL_TOPK true
#include "eppk_pick_inst.hip.h"
