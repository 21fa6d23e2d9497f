// fp16 (IEEE half, h16.h) build of the column-pair depthwise forward kernels (see cp_io.h)
#define DW_BF16
#define CFN_F16 1
#include "dwcp.hip"
