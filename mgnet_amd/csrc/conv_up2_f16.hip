// fp16 instantiation of conv_up2.hip (see h16.h): exports mgn_conv3x3_up2_win_f16
#define MGN_F16 1
#include "conv_up2.hip"
