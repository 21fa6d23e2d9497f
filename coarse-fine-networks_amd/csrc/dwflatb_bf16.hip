// dwflatb.hip compiled for bf16 tensors (cp_io.h): entry points dw_flatb_try_bf16, dw_flatb_s2_try_bf16
// hipcc-flags: -fno-slp-vectorize
#define DW_BF16 1
#include "dwflatb.hip"
