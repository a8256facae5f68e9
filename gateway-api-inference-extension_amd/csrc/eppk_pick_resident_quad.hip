// pick_resident_kernel<..., QUAD = true>: the resident workgroup with pick_quad_kernel's body (eppk_pick_resident.hip has the other form).
#define EPPK_RESIDENT_QUAD true
#define EPPK_RESIDENT_FN pick_resident_quad
#include "eppk_pick_resident.hip"
