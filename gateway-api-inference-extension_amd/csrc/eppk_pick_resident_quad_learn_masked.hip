// pick_resident_kernel<..., QUAD, MASKED, LEARN>: small batches with candidate masks whose picks the index learns at once (see
// eppk_pick_resident_quad_learn.hip).
#define EPPK_RESIDENT_QUAD true
#define EPPK_RESIDENT_MASKED true
#define EPPK_RESIDENT_LEARN true
#define EPPK_RESIDENT_FN pick_resident_quad_learn_masked
#include "eppk_pick_resident.hip"
