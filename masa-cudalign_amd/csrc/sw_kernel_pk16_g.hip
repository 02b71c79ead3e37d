// packed 16-bit strip kernel, translation unit G: the mixed-height launch (1536- and 1408-row strips in one kernel)
#define PK16_PART 6
#include "sw_kernel_pk16.inc"
