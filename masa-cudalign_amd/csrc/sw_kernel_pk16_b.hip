// packed strip kernel, instantiation part 1 of 6 (see the end of sw_kernel_pk16.inc)
#define PK16_PART 1
#include "sw_kernel_pk16.inc"
