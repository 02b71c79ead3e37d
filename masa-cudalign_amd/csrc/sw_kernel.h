// Shared declarations between the HIP kernels (sw_kernel.hip) and the host runtime (runtime.cpp).
#ifndef MI355SW_KERNEL_H_
#define MI355SW_KERNEL_H_

#include <hip/hip_runtime.h>

#include <vector>

namespace mi355sw {

enum { CHUNK = 64 };   // bus columns staged per hand-off (one per lane)
enum { DET_UNSET = (int) 0x80000000 };   // KernelArgs::det_prefix entry that has not been written yet

// M/libmasa/IManager.hpp:36-47
enum { INIT_WITH_ZEROES = 0, INIT_WITH_GAPS = 1, INIT_WITH_CUSTOM_DATA = 2, INIT_WITH_GAPS_OPENED = 3 };

struct KernelArgs {
    // problem
    const unsigned char* seq0;   // m codes (vertical)
    const unsigned char* seq1;   // n shift codes (code*4) or raw bytes, padded to a multiple of 64
    int m, n;
    int n_match_codes;           // PROFILE: codes < this can match; others never do
    int pad_code;                // code used for rows >= m
    int seq0_shift;              // byte-compare kernels: seq0 bytes are compared as (byte << seq0_shift) with seq1 bytes
                                 // (coded sequences keep seq1 as code*4; raw sequences: 0)
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
    int* stop_word;              // device word: the first wavefront that sees the host's stop sets it, every strip looks at it
                                 // once per chunk and wherever it waits -- all strips in flight end together, like the blocks
                                 // of AbstractDiagonalAligner::alignPartition (mustContinue() per diagonal, :64), instead of one
                                 // after the other at their own polls of the host's word
    int* error_flag;             // set by the kernel on a bounded-spin timeout
    const int* first_col_ready;  // counter of first_col rows that are valid (system scope), or nullptr (all): pinned host
                                 // memory when the host feeds the column, this GPU's HBM (fine-grained) when the
                                 // neighbour GPU's kernel writes it over xGMI (column port)
    int* peer_ready;             // column port of the NEXT band (peer-mapped, system scope): rows of last_col written so
                                 // far are published here by complete_strip, or nullptr
    int* host_error;             // pinned host mirror of error_flag, written BEFORE the strip counter moves past the
                                 // failing strip (the host must not hand out rows of a strip that left its exact range)
    int fault_strip;             // test knob (MI355SW_FAULT_OVERFLOW_STRIP): this strip of the packed kernel reports an
                                 // overflow it did not have, so that the int32 rerun / replay paths can be exercised; -1 = off
    long long wait_ticks;        // budget (10 ns ticks of s_memrealtime) of waits on data another GPU or the host delivers
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
    // ---- running best shared along a chain of column bands (T domain like gbest; any value ever stored is the score
    //      of a real alignment, so a stale or out-of-order word is only a weaker pruning bound) ----
    // Every word has ONE writer kernel and plain system-scope stores, like the row counter of a column port:
    const int* chain_down_in;    // own inbound port +64: best known to the bands on the LEFT (the previous band's kernel stores it)
    int* chain_up_pub;           // own inbound port +128: best known HERE, for the previous band's kernel to read
    int* chain_down_pub;         // next band's port +64 (peer-mapped): best known here, pushed to the RIGHT
    const int* chain_up_in;      // next band's port +128 (peer-mapped): what the bands on the right know (remote load)
    const int* host_best_hint;   // pinned host word: a lower bound from outside (mi355sw_stream_best_hint), or nullptr
    int* host_best_report;       // pinned host word: the running best as of the last completed strip, or nullptr
    unsigned long long* wait_acc; // 10 ns ticks the wavefronts spent in claim_strip_common waiting for first-column rows
                                 // that somebody else delivers (host or neighbour GPU), summed over the wavefronts; or nullptr
    int mix_first;               // strips [0, mix_first) have the kernel's first height, the rest its second (mixed-height
                                 // launches, sw_strip_kernel_pk16_mixed); INT_MAX for the ordinary kernels
    int strip_row0_b;            // DP row of strip s >= mix_first = strip_row0_b + s * (second height)
    int* win;                    // pruning kernels: the strips' windows, 2 ints per strip + 2 for the row above -- [gap_lo, gap_hi) of output
                                 // columns strip s left unwritten at win[2(s+1)] (see WIN_RETIRED in sw_kernel_pk16.inc) -- or nullptr (no window)
    int band_c0, band_slope_q16, band_w;   // band mode (pruning kernels with a window; band_w > 0): only columns within band_w of
                                 // the line  column = band_c0 + row * band_slope_q16 / 65536  (row relative to the partition) are computed
    // ---- reproducible pruning (MI355SW_F_DETERMINISTIC_PRUNE; nullptr = off) ----
    // WHICH slabs a pruning run skips depends on what the running best was when a wavefront looked at it -- with gbest, on
    // timing: two runs of one input leave different lower bounds off the optimal paths in their special rows.  The reference
    // decides its pruning window on the host between two external diagonals (BlockPruningDiagonal::updatePruningWindow,
    // BlockPruningDiagonal.cpp:109-152): a function of the input.  In this mode strip s tests against
    //   det_prefix[max(0, s - det_lag)] = max(what the run started from, what strips 0 .. s - det_lag - 1 found)
    // -- final when strip s starts (completion is ordered, at most det_lag strips are ever in flight) -- and against what IT
    // has found so far, left to right.  det_news[s]: what strip s found (T domain), folded into the prefix by
    // complete_strip_common.  Entries not yet written hold DET_UNSET; a reader waits for them (it never has to, see above).
    int* det_prefix;             // [strips + 1]
    int* det_news;               // [strips]
    int det_lag;
    const int* gbest_in;         // where the strips READ the running best from: gbest itself, or a word that stays at -INF
                                 // when every strip record must be that strip's own exact best (block scores) instead
                                 // of "nothing below what is already known elsewhere"
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

// ---- strip hand-over shared by both kernel families (claim a strip / complete it in order) -------------------
// Separate noinline functions on purpose: with a lane-0-only `if` as the last statement of the persistent
// loop, hipcc's structurizer moved lane 0 out of the loop body and sent lanes 1..63 through the next iteration
// on their own; and a wait loop placed in the strip function itself changes the hot loop's register allocation
// (measured: -40 % on the 3 M x 3 M case).
#if defined(__HIPCC__)
namespace sync {
__device__ __forceinline__ int rfl(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int poll_dev(const int* p) { return rfl(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ int poll_system(const int* p) { return rfl(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)); }
}  // namespace sync

// Claim the next strip: ordered tickets => forward progress for any grid size.
// A band whose first column is delivered from outside (host-fed pinned column, or the neighbour GPU writing
// this GPU's column port) waits HERE until the rows of the claimed strip have arrived: first for the strip above
// to have produced its first columns (device-scope poll; a strip cannot move before that anyway, so only one or
// two wavefronts poll the outside counter at any time -- with every wavefront of a launch polling pinned host
// memory the neighbour band's PCIe stores crawled), then for the counter.  The budget is wall time
// (s_memrealtime, 100 MHz), not a spin count; when it runs out the strip is given up (abort + error 2) instead
// of being computed on rows that never came.
// the wait of a claimed strip for first-column rows that somebody else delivers (see claim_strip_common)
static __device__ __attribute__((noinline, unused)) void wait_first_column_common(const KernelArgs* ap, const int s_in, const int lane, const int strip_rows) {
    const UniformArgs a = uniform_args(ap);
    const int s = sync::rfl(s_in);
    if (a->first_col_ready != nullptr && s < a->num_strips) {
        const long long t0 = __builtin_amdgcn_s_memrealtime();
        const long long budget = a->wait_ticks;
        const int* prog_in = &a->progress[s];
        bool ok = true, stop = false;
        int it = 0;
        while (sync::poll_dev(prog_in) < 1) {
            if (sync::poll_dev(a->abort_flag) != 0 || sync::poll_dev(a->stop_word) != 0) { stop = true; break; }
            if ((++it & 255) == 0) {
                if (a->host_abort != nullptr && sync::poll_system(a->host_abort) != 0) { stop = true; break; }
                if (__builtin_amdgcn_s_memrealtime() - t0 > budget) { ok = false; break; }
            }
            __builtin_amdgcn_s_sleep(16);
        }
        long long need = (long long) a->strip_row0 + (long long) (s + 1) * strip_rows;
        if (need > a->m) need = a->m;
        while (ok && !stop && sync::poll_system(a->first_col_ready) < (int) need) {
            if (sync::poll_dev(a->abort_flag) != 0 || sync::poll_dev(a->stop_word) != 0) break;
            if ((++it & 63) == 0) {
                if (a->host_abort != nullptr && sync::poll_system(a->host_abort) != 0) break;
                if (__builtin_amdgcn_s_memrealtime() - t0 > budget) { ok = false; break; }
            }
            __builtin_amdgcn_s_sleep(32);
        }
        if (!ok && lane == 0) {
            atomicExch(a->error_flag, 2);
            __hip_atomic_store(a->abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0 && a->wait_acc != nullptr) atomicAdd(a->wait_acc, (unsigned long long) (__builtin_amdgcn_s_memrealtime() - t0));
        __builtin_amdgcn_wave_barrier();
    }
}

static __device__ __attribute__((noinline, unused)) int claim_strip_common(const KernelArgs* ap, const int lane, const int strip_rows) {
    const UniformArgs a = uniform_args(ap);
    int s = 0;
    if (lane == 0) s = atomicAdd(a->ticket, 1);
    s = sync::rfl(s);
    wait_first_column_common(ap, s, lane, strip_rows);
    return s;
}

// Relay of the running best between this kernel, its neighbours in the band chain and the host.  Called by lane 0
// inside the ordered (hence serialised) section of complete_strip_common: reads every inbound word, folds the
// maximum into this GPU's gbest and republishes it on every outbound word -- so a score found by any band reaches
// every other band hop by hop, in both directions, without any kernel ever waiting for it.
static __device__ __forceinline__ void relay_running_best(const UniformArgs a) {
    if (a->gbest == nullptr) return;
    if (a->chain_down_in == nullptr && a->chain_up_in == nullptr && a->chain_down_pub == nullptr && a->chain_up_pub == nullptr &&
        a->host_best_hint == nullptr && a->host_best_report == nullptr) return;
    const int v0 = __hip_atomic_load(a->gbest, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int v = v0;
    if (a->chain_down_in != nullptr) v = max(v, __hip_atomic_load(a->chain_down_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    if (a->chain_up_in != nullptr) v = max(v, __hip_atomic_load(a->chain_up_in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    if (a->host_best_hint != nullptr) v = max(v, __hip_atomic_load(a->host_best_hint, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
    if (v > v0) atomicMax(a->gbest, v);
    if (a->chain_up_pub != nullptr) __hip_atomic_store(a->chain_up_pub, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a->chain_down_pub != nullptr) __hip_atomic_store(a->chain_down_pub, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a->host_best_report != nullptr) __hip_atomic_store(a->host_best_report, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Ordered completion: strips_done == s+1 means strips 0..s are complete and their last-column / special-row /
// best records are visible (system scope) to the host -- and, through the column port, to the next band's GPU.
// `stopped_waves_leave`: the kernel's wavefronts leave the persistent loop when they find a stop at their claim (the
// one-partition kernels), so after a stop the counter may never reach s -- a wavefront that claimed ticket s and saw the
// stop is gone, while the holder of ticket s+1, whose poll came a moment earlier, swept its strip.  That wavefront must
// not sit out the spin limit here (10-20 s) nor turn the stop into a time-out: once a stop is set it leaves, publishing
// nothing but the error mirror.  (The batch kernel takes every ticket through here in order, stopped or not, and the
// host counts on that: there the counter always arrives.)
static __device__ __attribute__((noinline, unused)) void complete_strip_common(const KernelArgs* ap, const int s_in, const int lane, const int strip_rows,
                                                                                const bool stopped_waves_leave) {
    const UniformArgs a = uniform_args(ap);
    const int s = sync::rfl(s_in);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    int spins = 0;
    // a band fed from outside may legitimately stand still for as long as its first column takes to arrive
    const int spin_limit = a->first_col_ready != nullptr ? (1 << 30) : (1 << 24);
    bool gone = false;                 // stopped while waiting for a predecessor that will never count itself
    while (sync::poll_dev(a->strips_done_dev) != s && spins < spin_limit) {
        __builtin_amdgcn_s_sleep(8);
        spins++;
        if (stopped_waves_leave && (spins & 31) == 0 &&
            (sync::poll_dev(a->abort_flag) != 0 || sync::poll_dev(a->stop_word) != 0 ||
             (a->host_abort != nullptr && sync::poll_system(a->host_abort) != 0))) { gone = true; break; }
    }
    if (lane == 0) {
        // (never over a report that is already there: an overflow code 16 must reach the host as such)
        if (spins >= spin_limit) atomicCAS(a->error_flag, 0, 3);
        if (!gone) relay_running_best(a);
        const int err = __hip_atomic_load(a->error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (err != 0 && a->host_error != nullptr) {
            // low byte: the code; above it, for an overflow report of the packed kernel, its causes (error_flag[1])
            const int why = __hip_atomic_load(a->error_flag + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a->host_error, err | (why << 8), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (!gone && a->det_prefix != nullptr) {
            // (ordered section: prefix[s] is final; release before the counter below lets anybody count on prefix[s + 1])
            const int before = __hip_atomic_load(&a->det_prefix[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int mine = __hip_atomic_load(&a->det_news[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&a->det_prefix[s + 1], max(before, mine), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!gone) {
            if (a->peer_ready != nullptr && err == 0 &&
                __hip_atomic_load(a->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
                (a->host_abort == nullptr || __hip_atomic_load(a->host_abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0)) {
                long long rows = (long long) a->strip_row0 + (long long) (s + 1) * strip_rows;
                if (rows > a->m) rows = a->m;
                // the strip's last-column cells were stored into the neighbour's HBM by every lane before the
                // system-scope release fence above; this is the flag that follows them over xGMI
                __hip_atomic_store(a->peer_ready, (int) rows, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            if (a->strips_done_host != nullptr)
                __hip_atomic_store(a->strips_done_host, s + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(a->strips_done_dev, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_wave_barrier();
}
#endif

// Several partitions in ONE launch (mi355sw_align_partitions): every partition keeps its own argument block, buses,
// progress words and control block; the launch only shares the wavefronts.  Tickets are handed out in the order of
// `map` -- strip 0 of every partition, then strip 1 of every partition, ... -- so a strip's predecessor always holds
// an earlier ticket (claimed by a wavefront that is running): forward progress for any grid size, as with one partition.
struct BatchArgs {
    const KernelArgs* const* args;   // [partitions] device argument blocks
    const int2* map;                 // [total] ticket -> (partition, strip)
    int total;
    int* ticket;
};
// two strip heights in one launch: strips [0, a.mix_first) are 128*rows_per_half_a rows tall, the others 128*rows_per_half_b
hipError_t launch_strip_kernel_pk16_mixed(const KernelArgs& a, KernelArgs* dargs, int rows_per_half_a, int rows_per_half_b, int grid,
                                          hipStream_t stream, bool track, bool sw);
hipError_t launch_batch_kernel_pk16(const BatchArgs* dbatch, int rows_per_half, int grid, hipStream_t stream, bool track, bool sw);
hipError_t launch_batch_kernel_pk16_band(const BatchArgs* dbatch, int grid, hipStream_t stream);   // 512-row strips, NW, value-only, pruning kernels (band mode)

// `dargs` = device copy of the argument block (the launcher uploads `a` into it on `stream`)
hipError_t launch_strip_kernel(const KernelArgs& a, KernelArgs* dargs, int rows_per_lane, int grid, hipStream_t stream,
                               bool sw, bool profile, bool track);
int strip_kernel_waves_per_simd(int rows_per_lane);   // int32 family: wavefronts of one launch that share a SIMD (2 for 256/512-row strips; the packed kernels: 1)
// packed 16-bit SW kernel (sw_kernel_pk16.inc, instantiated by sw_kernel_pk16_{a,b,c}.hip): strip height = 128*rows_per_half
hipError_t launch_strip_kernel_pk16(const KernelArgs& a, KernelArgs* dargs, int rows_per_half, int grid, hipStream_t stream, bool track, bool sw);
// stage 4 (stage4.hip): Myers-Miller refinement of a crosspoint list, batched on the GPU
struct Stage4Crosspoint { int type, i, j, score; };          // M/common/Crosspoint.hpp
struct Stage4Stats { int steps; double kernel_ms; long long dp_cells, partitions; };
int stage4_refine(const unsigned char* d_seq0, long long len0, const unsigned char* d_seq1, long long len1, int seq0_shift,
                  hipStream_t stream, std::vector<Stage4Crosspoint>& list, int max_size, Stage4Stats* stats, hipError_t* hip_err);
hipError_t launch_fill_bus(int2* bus, int n, int init_type, int start_offset, hipStream_t stream);
hipError_t launch_fill_int(int* p, long long count, int value, hipStream_t stream);
hipError_t launch_fill_cells(int2* p, long long count, int h, int f, hipStream_t stream);

}  // namespace mi355sw
#endif
