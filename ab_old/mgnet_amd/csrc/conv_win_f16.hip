// fp16 instantiation of conv_win.hip (see h16.h): exports mgn_conv3x3_win_f16
#define MGN_F16 1
#include "conv_win.hip"
