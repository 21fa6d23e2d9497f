// dwflatb.hip compiled for fp16 tensors (cp_io.h): entry points dw_flatb_try_f16, dw_flatb_s2_try_f16
// hipcc-flags: -fno-slp-vectorize
#define DW_BF16 1
#define CFN_F16 1
#include "dwflatb.hip"
