// dwcpbx.hip compiled for fp16 tensors (cp_io.h): entry point dw_cpbx_try_f16
// hipcc-flags: -fno-slp-vectorize
#define DW_BF16 1
#define CFN_F16 1
#include "dwcpbx.hip"
