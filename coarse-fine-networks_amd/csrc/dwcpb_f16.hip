// fp16 (IEEE half, h16.h) build of the column-pair fused depthwise backward (stride 1; see cp_io.h)
// hipcc-flags: -fno-slp-vectorize
#define DW_BF16
#define CFN_F16 1
#include "dwcpb.hip"
