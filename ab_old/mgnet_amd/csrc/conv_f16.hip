// fp16 instantiation of conv.hip (see h16.h): exports mgn_weight_layout_f16, mgn_weight_layout_batch_f16, mgn_conv_igemm_f16,
// mgn_conv_wgrad_f16
#define MGN_F16 1
#include "conv.hip"
