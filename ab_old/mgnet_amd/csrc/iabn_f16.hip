// fp16 instantiation of iabn.hip (see h16.h); dtype 1 of its entry points then means IEEE fp16
#define MGN_F16 1
#include "iabn.hip"
