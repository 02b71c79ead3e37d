// Shared declarations between the HIP kernels (sw_kernel.hip) and the host runtime (runtime.cpp).
#ifndef MI355SW_KERNEL_H_
#define MI355SW_KERNEL_H_

#include <hip/hip_runtime.h>

namespace mi355sw {

enum { CHUNK = 64 };   // bus columns staged per hand-off (one per lane)

// M/libmasa/IManager.hpp:36-47
enum { INIT_WITH_ZEROES = 0, INIT_WITH_GAPS = 1, INIT_WITH_CUSTOM_DATA = 2, INIT_WITH_GAPS_OPENED = 3 };

struct KernelArgs {
    // problem
    const unsigned char* seq0;   // m codes (vertical)
    const unsigned char* seq1;   // n shift codes (code*4) or raw bytes, padded to a multiple of 64
    int m, n;
    int n_match_codes;           // PROFILE: codes < this can match; others never do
    int pad_code;                // code used for rows >= m
    // strip geometry
    int num_strips;              // strips in this launch
    int strip_row0;              // DP row of strip 0 of this launch
    int strip_index0;            // global ordinal of strip 0 (for special-row spacing)
    // buses (HBM)
    int2* bus;                   // n cells (H,F): row above on entry, emit row of the last strip on exit
    const int2* first_col;       // m+1 cells (H,E) incl. corner, or nullptr => INIT_WITH_ZEROES
    int2* last_col;              // m+1 cells (H,E) or nullptr
    int2* special_rows;          // slot k = k * special_pitch cells, or nullptr
    long long special_pitch;
    int special_interval_strips; // every K-th strip end is flushed (0 = none)
    int2* last_row;              // n cells (H,F) of DP row m, or nullptr
    int2* ckpt_rows;             // checkpoint rows for the exact-position pass: slot k = bus row below strip k*K-1
    long long ckpt_pitch;        //   (slot 0 = the first row, written by the host), or nullptr
    int ckpt_interval_strips;    // K
    // synchronisation / results
    int* progress;               // num_strips+1 ints; progress[0] = n (virtual strip above)
    int* ticket;                 // next strip to claim
    int* abort_flag;             // device word: the kernel sets it on an overflow report; strips claimed afterwards are skipped
    const int* host_abort;       // pinned host word: the host sets it != 0 to stop (mustContinue() == false) -- a store,
                                 // not a copy, because no copy may be queued while the persistent kernel runs
    int* error_flag;             // set by the kernel on a bounded-spin timeout
    const int* first_col_ready;  // pinned host counter: rows of first_col that are valid, or nullptr (all)
    int* strips_done_dev;        // device counter, ordered: value s means strips [0,s) complete
    int* strips_done_host;       // pinned host mirror (system scope)
    int* gbest;                  // running global best (T domain): lower bound that seeds every lane's threshold
    int4* strip_best;            // per strip {score, i, j, valid}
    int independent;             // seed pass: strips do not feed each other (progress is published into a dummy area)
    int prune;                   // block pruning on (packed SW kernel): skip slabs that cannot reach the running best
    int prune_rows, prune_cols;  // rows / columns from the partition origin to the end of the super-partition
    unsigned long long* pruned_slabs;   // device counter of skipped 64-step slabs
    int* dbg;                    // optional debug words (nullptr in production)
    long long* trace;            // optional per-strip timing {start,end,poll spins,first chunk} (nullptr in production)
};

// The argument block lives in device memory and is read through the constant address space with a
// readfirstlane'd pointer: every field is then provably wave-uniform (SGPRs, scalar branches).  Passing
// it by value and taking its address for the noinline strip function made hipcc spill it to scratch,
// treat every field as divergent and structurize the persistent loop so that lanes left it one by one.
#if defined(__HIPCC__)
typedef const __attribute__((address_space(4))) KernelArgs* UniformArgs;
__device__ __forceinline__ UniformArgs uniform_args(const KernelArgs* p) {
    const unsigned long long v = (unsigned long long) p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned) v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned) (v >> 32));
    return (UniformArgs) (((unsigned long long) hi << 32) | lo);
}
#endif

// `dargs` = device copy of the argument block (the launcher uploads `a` into it on `stream`)
hipError_t launch_strip_kernel(const KernelArgs& a, KernelArgs* dargs, int rows_per_lane, int grid, hipStream_t stream,
                               bool sw, bool profile, bool track);
// packed 16-bit SW kernel (sw_kernel_pk16.inc, instantiated by sw_kernel_pk16_{a,b,c}.hip): strip height = 128*rows_per_half
hipError_t launch_strip_kernel_pk16(const KernelArgs& a, KernelArgs* dargs, int rows_per_half, int grid, hipStream_t stream, bool track, bool sw);
hipError_t launch_fill_bus(int2* bus, int n, int init_type, int start_offset, hipStream_t stream);
hipError_t launch_fill_int(int* p, long long count, int value, hipStream_t stream);

}  // namespace mi355sw
#endif
