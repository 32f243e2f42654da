// fp16 instantiation of eltwise.hip (see h16.h)
#define MGN_F16 1
#include "eltwise.hip"
