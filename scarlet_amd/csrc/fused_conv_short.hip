// fused_conv.hip once more with 512-thread workgroups: transforms with rows of up to 96
// elements (launch_fused_conv picks it; measurements there).
#define SMI_CONV_THREADS 512
#define SMI_CONV_SHORT_ROWS
#include "fused_conv.hip"
