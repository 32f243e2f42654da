// fp16 instantiation of headloss.hip (see h16.h)
#define MGN_F16 1
#include "headloss.hip"
