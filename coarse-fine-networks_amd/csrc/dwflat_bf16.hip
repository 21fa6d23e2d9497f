// dwflat.hip compiled for bf16 tensors (cp_io.h): entry point dw_flat_fwd_try_bf16
#define DW_BF16 1
#include "dwflat.hip"
