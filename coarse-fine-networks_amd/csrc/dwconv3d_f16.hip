// fp16-storage build of the depthwise 3x3x3 kernels: dwconv3d.hip compiled a third time with 2-byte IEEE-half tensor elements
// (fp32 arithmetic, fp32 LDS images, fp32 weights, fp64 reductions unchanged; h16.h).  Entry points: cfn_dwconv3d_*_f16.
// The argument structs are renamed so that the kernel symbols of the builds differ.
#define DW_BF16 1
#define CFN_F16 1
#define DwArgs DwArgsF16
#define DwFusedArgs DwFusedArgsF16
#define DwS2Args DwS2ArgsF16
#define DwPlan DwPlanF16
#include "dwconv3d.hip"
