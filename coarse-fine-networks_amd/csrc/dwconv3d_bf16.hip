// bf16-storage build of the depthwise 3x3x3 kernels: dwconv3d.hip compiled a second time with 2-byte tensor elements
// (fp32 arithmetic, fp32 LDS images, fp32 weights, fp64 reductions unchanged).  Entry points: cfn_dwconv3d_*_bf16.
// The argument structs are renamed so that the kernel symbols of the two builds differ.
#define DW_BF16 1
#define DwArgs DwArgsBf16
#define DwFusedArgs DwFusedArgsBf16
#define DwS2Args DwS2ArgsBf16
#define DwPlan DwPlanBf16
#include "dwconv3d.hip"
