// packed 16-bit strip kernel, translation unit H: the batch kernel with 1024-row strips (batches of tall partitions)
#define PK16_PART 7
#include "sw_kernel_pk16.inc"
