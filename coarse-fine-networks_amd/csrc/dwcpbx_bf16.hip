// dwcpbx.hip compiled for bf16 tensors (cp_io.h): entry point dw_cpbx_try_bf16
// hipcc-flags: -fno-slp-vectorize
#define DW_BF16 1
#include "dwcpbx.hip"
