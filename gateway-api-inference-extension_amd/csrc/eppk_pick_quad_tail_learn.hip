// pick_quad_kernel<..., TAIL, LEARN> instantiations (see eppk_pick_quad_tail.hip): single picks that also leave the learn words for the
// post-route index update (eppk_pick_learn_device, EPPK_PICK_LEARN); MASKED = false.
#include "eppk_kernels.hip.h"
#define EPPK_QUAD_TAIL_UNIT pick_quad_tail_learn
#define EPPK_QUAD_TAIL_MASKED false
#define EPPK_QUAD_TAIL_TOPK false
#define EPPK_QUAD_TAIL_LEARN 1
#include "eppk_pick_inst.hip.h"
