// pwbf16.hip compiled for IEEE-half tensors (h16.h): v_mfma_f32_32x32x16_f16, entry points cfn_pwconv_*_f16, cfn_subsample_hw_f16
#define CFN_F16 1
#include "pwbf16.hip"
