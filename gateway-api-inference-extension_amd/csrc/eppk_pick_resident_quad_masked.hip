// pick_resident_kernel<..., QUAD, MASKED>: the resident workgroup for small batches WITH candidate masks (see eppk_pick_resident.hip).
#define EPPK_RESIDENT_QUAD true
#define EPPK_RESIDENT_MASKED true
#define EPPK_RESIDENT_FN pick_resident_quad_masked
#include "eppk_pick_resident.hip"
