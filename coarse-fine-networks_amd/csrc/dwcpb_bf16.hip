// bf16 build of the column-pair fused depthwise backward (stride 1; see cp_io.h)
// hipcc-flags: -fno-slp-vectorize
#define DW_BF16
#include "dwcpb.hip"
