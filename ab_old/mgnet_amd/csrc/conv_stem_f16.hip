// fp16 instantiation of conv_stem.hip (see h16.h): exports mgn_conv_stem7_f16
#define MGN_F16 1
#include "conv_stem.hip"
