// Pick-kernel instantiations for uint64_t lane words, 6 counter planes (see eppk_pick_inst.hip.h).
#include "eppk_kernels.hip.h"
#define EPPK_PICK_INST_LW uint64_t
#define EPPK_PICK_INST_NPL 6
#define EPPK_PICK_INST_NAME pick_kernel_u64_6
#include "eppk_pick_inst.hip.h"
