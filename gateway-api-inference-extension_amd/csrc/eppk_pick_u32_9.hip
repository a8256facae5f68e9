// Pick-kernel instantiations for uint32_t lane words, 9 counter planes (see eppk_pick_inst.hip.h).
#include "eppk_kernels.hip.h"
#define EPPK_PICK_INST_LW uint32_t
#define EPPK_PICK_INST_NPL 9
#define EPPK_PICK_INST_NAME pick_kernel_u32_9
#include "eppk_pick_inst.hip.h"
