// streambf16.hip compiled for IEEE-half tensors (h16.h): entry points cfn_bn_add_relu_*_f16, cfn_pool_hw_*_f16
#define CFN_F16 1
#include "streambf16.hip"
