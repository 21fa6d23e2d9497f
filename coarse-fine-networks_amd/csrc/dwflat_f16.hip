// dwflat.hip compiled for fp16 tensors (cp_io.h): entry point dw_flat_fwd_try_f16
#define DW_BF16 1
#define CFN_F16 1
#include "dwflat.hip"
