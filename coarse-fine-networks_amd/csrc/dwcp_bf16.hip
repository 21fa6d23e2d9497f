// bf16 build of the column-pair depthwise forward kernels (see cp_io.h)
#define DW_BF16
#include "dwcp.hip"
