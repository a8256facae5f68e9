// pick_resident_kernel<..., QUAD, MASKED, TOPK>: small batches with candidate masks AND ordered fallbacks (see eppk_pick_resident.hip).
#define EPPK_RESIDENT_QUAD true
#define EPPK_RESIDENT_MASKED true
#define EPPK_RESIDENT_TOPK true
#define EPPK_RESIDENT_FN pick_resident_quad_topk_masked
#include "eppk_pick_resident.hip"
