// fp16 instantiation of pool.hip (see h16.h)
#define MGN_F16 1
#include "pool.hip"
