// fp16 instantiation of prep.hip (see h16.h)
#define MGN_F16 1
#include "prep.hip"
