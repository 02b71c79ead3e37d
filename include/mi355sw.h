/*
 * mi355sw.h -- C ABI of the MI355X-native Stage-1 SW/NW strip-wavefront engine.
 *
 * This is the drop-in boundary for the Stage-1 hot path of MASA-CUDAlign.  Every entry point
 * cites the reference interface it replaces ("M/" = masa-cudalign-4.0.2.1028/libs/masa-core/src/,
 * "X/" = masa-cudalign-4.0.2.1028/src/).  Plain C types only: no C++, no torch, no HIP types.
 * All functions return 0 on success or a negative MI355SW_E* code (never exit(); the reference's
 * cutilSafeCall aborts, X/cuda_util.h:34-42); mi355sw_last_error() gives the message.
 *
 * Threading: like MASA-Core (M/libmasa/IAligner.hpp:55-100) one thread drives a handle; manager
 * callbacks are invoked only from the thread that called mi355sw_align_partition().
 * mi355sw_progress() may be called from another thread (logger pthread, sw_stage1.cpp:113-128).
 */
#ifndef MI355SW_H_
#define MI355SW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI355SW_ABI_VERSION 8

/* M/libmasa/libmasaTypes.hpp:35-41  cell_t {int h; union{int f; int e;};} 8-byte aligned */
typedef struct { int32_t h; int32_t f; } mi355sw_cell;
/* M/libmasa/libmasaTypes.hpp:88-95  score_t */
typedef struct { int32_t i; int32_t j; int32_t score; } mi355sw_score;
/* M/libmasa/libmasaTypes.hpp:100-109 score_params_t */
typedef struct { int32_t match, mismatch, gap_open, gap_ext; } mi355sw_score_params;
/* M/libmasa/libmasaTypes.hpp:51-60 match_result_t */
typedef struct { int32_t found, k, score, type; } mi355sw_match_result;
/* M/libmasa/Partition.hpp: half-open [i0,i1) x [j0,j1) in sequence-relative coordinates */
typedef struct { int32_t i0, j0, i1, j1; } mi355sw_partition;

#define MI355SW_INF 999999999                 /* libmasaTypes.hpp:46 */
#define MI355SW_NEEDLEMAN_WUNSCH 0            /* IManager.hpp:31 */
#define MI355SW_SMITH_WATERMAN 1              /* IManager.hpp:33 */
#define MI355SW_INIT_WITH_ZEROES 0            /* IManager.hpp:38 */
#define MI355SW_INIT_WITH_GAPS 1              /* IManager.hpp:41 */
#define MI355SW_INIT_WITH_CUSTOM_DATA 2       /* IManager.hpp:47 */
#define MI355SW_INIT_WITH_GAPS_OPENED 3       /* IManager.hpp:44 */

#define MI355SW_OK 0
#define MI355SW_EINVAL (-1)     /* bad argument / call order                    */
#define MI355SW_EHIP (-2)       /* HIP runtime error                            */
#define MI355SW_ENOGPU (-3)     /* no usable gfx950 device                      */
#define MI355SW_ENOMEM (-4)
#define MI355SW_ETIMEOUT (-5)   /* a bounded in-kernel spin gave up             */
#define MI355SW_ESTATE (-6)
#define MI355SW_ETRACEBACK (-8)  /* stage 4: no column of a partition matches its score difference (the crosspoints do
                                    not lie on one optimal alignment of these sequences)                          */
#define MI355SW_ETOOLARGE (-9)   /* stage 4: a partition beyond the reference's own limit (131072 columns)         */
#define MI355SW_EBOUND (-10)     /* a pruning run ended BELOW the bound it started from (mi355sw_stream_params.initial_bound): the
                                    bound was not the score of an alignment that exists -- the optimum may have been pruned
                                    away with it, so the result is void.  The reference has no such check (a wrong best
                                    score loaded by Status::load, sw_stage1.cpp:210-217, silently prunes). */
#define MI355SW_EOVERFLOW16 (-7) /* packed 16-bit kernel left its exact range: rerun with force_int32
                                    (mi355sw_align_partition and mi355sw_process_block do that by themselves;
                                    the streaming form reports it from poll/end, rows handed out before are exact) */

typedef struct mi355sw_handle mi355sw_handle;

/* Extension parameters; replaces X/CUDAlignerParameters.cpp:33-110 (--gpu, --blocks).  The library reads NO environment
 * variable: every switch is a field here (ABI 7; the Python front maps its MI355SW_* variables onto them, engine.py). */
typedef struct {
    int32_t device;          /* HIP ordinal, -1 = current device (reference: --gpu)                 */
    int32_t rows_per_lane;   /* R in {4,8,12,16,24,32} (12/24/32: packed kernel only); strip height = 64*R (reference: THREADS_COUNT*ALPHA);
                                0 = choose from the partition size                                 */
    int32_t waves;           /* persistent wavefronts (reference: --blocks); 0 = one per SIMD      */
    int32_t flags;           /* MI355SW_F_*                                                          */
    int64_t max_special_bytes; /* HBM budget for device-resident special rows, 0 = 60 % of the HBM free at stream_begin */
    int32_t block_score_columns; /* > 0: mi355sw_align_partition also reports the best cell of every BLOCK of its grid
                                    through dispatch_score(score, bx, by) -- block (bx, by) = rows of strip `by` x columns
                                    [j0 + bx*W, j0 + (bx+1)*W), W = this value (Grid::setBlockHeight / setBlockWidth,
                                    M/libmasa/Grid.cpp:52-68).  Reference: CUDAligner::getBlockScores +
                                    AbstractDiagonalAligner::flushBlockScores (X/CUDAligner.cpp:441-452,
                                    AbstractDiagonalAligner.cpp:392-403); its only consumer is --dump-blocks
                                    (AlignerManager.cpp:418-423, BlocksFile.cpp).  Costs a second sweep of the
                                    partition, as a chain of W-column bands.  0 = off. */
    int32_t verbosity;       /* MI355SW_V_* bits: diagnostics on stderr (0 = none) */
    double wait_seconds;     /* wall-time budget of the in-kernel waits on data somebody else delivers (the host's first
                                column, the previous band's GPU); 0 = 3600 */
    int32_t fault_overflow_strip_plus1; /* TEST KNOB: strip (k - 1) of every packed launch reports an overflow it did not have, so
                                that the int32 rerun / replay paths can be exercised; 0 = off */
    int32_t stream_priority; /* 0 = the kernel stream gets the highest priority (default), 1 = normal, 2 = lowest */
    const char* trace_path;  /* per-strip timing records of every stream are written to this file (tools/trace_hops.py); NULL = off.
                                The string is copied. */
    int32_t batch_rows_per_lane; /* strip height of mi355sw_align_partitions' ONE launch, as rows_per_lane: 4 (256-row strips, the
                                default: many small partitions stopped early -- stage 3's walks), 8 (512) or 16 (1024: a batch
                                of tall partitions is throughput-bound and wants tall strips -- stage 2's sweeps from guessed
                                crosspoints).  0 = 4 */
    int32_t reserved32_;     /* zero */
    int64_t reserved_[3];    /* zero */
} mi355sw_config;
#define MI355SW_F_FORCE_GENERIC_COMPARE 1   /* raw byte compare kernels even if a profile fits */
#define MI355SW_F_FORCE_INT32 2             /* never use the packed 16-bit SW kernel */
#define MI355SW_F_NO_DIAGONAL_SEED 4        /* block pruning without the diagonal seed pass (see mi355sw_stats.seed_ms): the
                                               bound then only grows with what the sweep itself finds, as in the reference.
                                               For callers that want SEVERAL alignments (--max-alignments > 1): a strong
                                               first bound prunes the weaker ones away sooner than the reference's would. */
#define MI355SW_F_NO_SEED_PASS 8            /* no throw-away pass that warms the running best before the main launch, no probe */
#define MI355SW_F_NO_PRUNE_PROBE 16         /* a run that asks for pruning gets the pruning kernels whatever the probe of the
                                               pair says (see mi355sw_stream_begin) */
#define MI355SW_F_TWO_PHASE 32              /* value-only tracking + exact pass of the winning strip at every size (default:
                                               from 32 Mi rows) */
#define MI355SW_F_NO_MIXED 64               /* never two strip heights in one launch */
#define MI355SW_F_NO_SHARED_BEST 128        /* share_best streams keep their running best to themselves */
#define MI355SW_F_NO_BATCH 256              /* mi355sw_align_partitions runs its partitions one by one */
#define MI355SW_F_NO_WINDOW 1024            /* pruning runs without the pruning window: every strip walks the whole width and writes every
                                               skipped cell (A/B measurements, tests) */
#define MI355SW_F_STAIRCASE_SEED 2048       /* the diagonal seed as ONE chain of tiles down the diagonal (the round-4 form) instead of
                                               segments between anchors swept side by side (A/B measurements, tests) */
#define MI355SW_F_GENERATE_GAP_COLUMNS 4096  /* (ABI 7's opt-in; the DEFAULT since ABI 8 -- accepted and ignored) */
#define MI355SW_F_STREAM_GAP_COLUMNS 8192   /* mi355sw_align_partition takes a gap-initialised first column from the manager's stream cell
                                               by cell, like any other column, instead of recognising it from its first cells and making it
                                               on the device (InitialCellsReader is a function of the position).  The default became
                                               possible with the stop every strip sees at once (KernelArgs::stop_word): with all rows there
                                               from the start hundreds of strips are in flight when the manager says stop (A/B, tests) */
#define MI355SW_F_DETERMINISTIC_PRUNE 16384 /* reproducible special rows under block pruning: a strip tests against the bound as it stood a fixed number
                                               of strips above it (plus its own finds) instead of against the newest value any wavefront has
                                               published -- WHICH slabs go is then a function of the input, as in the reference, whose pruning
                                               window is set on the host between two diagonals (BlockPruningDiagonal.cpp:109-152).  Two runs, or an
                                               interrupted and resumed run and an uninterrupted one, leave the same special rows.  Costs nothing
                                               where the bound starts from the seed; a bound that grows with the sweep arrives one round of
                                               wavefronts later.  A stream on its own only (the bands of a chain share their finds as they arrive) */
#define MI355SW_F_NO_GOAL_SWEEP_HEIGHTS 32768 /* mi355sw_align_partition leaves the strip height of a sweep that looks goal-stopped (MASA-Core's stages 2 and 3,
                                               see AlignJob::begin) to the cost model like any other partition's (A/B measurements) */
#define MI355SW_F_NO_HOST_COUNTER 512       /* the kernel does not mirror its strip counter into host memory (measurements) */
#define MI355SW_V_MESSAGES 1                /* one line per noteworthy event (overflow reruns, the diagonal seed, ...) */
#define MI355SW_V_JOBS 2                    /* timing of every mi355sw_align_partition job */
#define MI355SW_V_SEED_TILES 4              /* every tile of the diagonal seed */
#define MI355SW_V_BATCH 8                   /* progress of mi355sw_align_partitions */
#define MI355SW_V_DEBUG_WORDS 16            /* kernels keep debug words, mi355sw_progress shows them */

/* aligner_capabilities_t, M/libmasa/capabilities.hpp:59-225 (same fields, int32 instead of bool) */
typedef struct {
    int32_t dispatch_last_cell, dispatch_last_row, dispatch_last_column;
    int32_t dispatch_special_row, dispatch_special_column;
    int32_t dispatch_scores, dispatch_block_scores, dispatch_best_score;
    int32_t customize_first_row, customize_first_column;
    int32_t process_partition, variable_penalties, block_pruning;
    int32_t needleman_wunsch, smith_waterman, fork_processes;
    int32_t maximum_seq0_len, maximum_seq1_len;
} mi355sw_capabilities;

/* The IManager callbacks the aligner drives, M/libmasa/IManager.hpp:98-313, as C function
 * pointers.  `user` is passed back verbatim.  Semantics as in the reference:
 *  - receive_first_row/column are sequential streams; the first call consumes the corner cell
 *    (M/libmasa/aligners/AbstractDiagonalAligner.cpp:84-86);
 *  - dispatch_row(i,...)/dispatch_column(j,...) are called in increasing position per row/column,
 *    the first call of a row/column carries the border cell with f = -INF
 *    (AbstractDiagonalAligner.cpp:290-298, :419-422); buffers are borrowed for the call;
 *  - must_continue()==0 makes mi355sw_align_partition return promptly. */
typedef struct {
    int32_t (*get_recurrence_type)(void* user);
    int32_t (*get_special_row_interval)(void* user);
    int32_t (*get_first_column_init_type)(void* user);
    int32_t (*get_first_row_init_type)(void* user);
    void (*get_super_partition)(void* user, mi355sw_partition* out);
    void (*receive_first_row)(void* user, mi355sw_cell* buffer, int32_t len);
    void (*receive_first_column)(void* user, mi355sw_cell* buffer, int32_t len);
    void (*dispatch_column)(void* user, int32_t j, const mi355sw_cell* buffer, int32_t len);
    void (*dispatch_row)(void* user, int32_t i, const mi355sw_cell* buffer, int32_t len);
    void (*dispatch_score)(void* user, mi355sw_score score, int32_t bx, int32_t by);
    int32_t (*must_continue)(void* user);
    int32_t (*must_dispatch_last_cell)(void* user);
    int32_t (*must_dispatch_last_row)(void* user);
    int32_t (*must_dispatch_last_column)(void* user);
    int32_t (*must_dispatch_special_rows)(void* user);
    int32_t (*must_dispatch_scores)(void* user);
    int32_t (*must_prune_blocks)(void* user);
    /* OPTIONAL (may be NULL; MASA-Core's IManager has no such call).  Partitions of >= 32 Mi rows track only the
     * best VALUE per strip in the main pass and locate the winning cell afterwards; a caller that checkpoints
     * (special rows on disk + resume) is told here, before every special row, the best value of the strips above it:
     * rows [row_lo, row_hi) hold a cell of that score.  Exact (i, j, score) records, where the engine has them, go
     * through dispatch_score at the same points. */
    void (*dispatch_strip_value)(void* user, int32_t row_lo, int32_t row_hi, int32_t score);
} mi355sw_manager;

/* Timing / accounting of the last mi355sw_align_partition (reference: Timer events + MCUPS line,
 * M/stage1/sw_stage1.cpp:440-448; getProcessedCells IAligner.hpp:377). */
typedef struct {
    int64_t cells;              /* m*n of the partition (GCUPS convention counts pruned cells)   */
    int64_t processed_cells;    /* cells actually computed                                       */
    double kernel_ms;           /* HIP-event time of the strip kernel launches on the engine's stream */
    double total_ms;            /* wall time of the call                                         */
    int32_t kernel_launches;
    int32_t strips, strip_rows, waves;
    int32_t profile_kernel;     /* 1 = 4-bit profile scoring, 0 = generic byte compare, 2 = packed 16-bit */
    int64_t algorithmic_bytes;  /* 17*n*ceil(m/S) + m + 8(n+1)*rows_flushed           */
    int64_t pruned_cells;       /* cells of skipped slabs (block pruning)                        */
    double wait_ms;             /* time the strip wavefronts spent waiting for first-column rows that somebody else
                                   delivers (the host, or the previous band's GPU through the column port), average
                                   per wavefront: what a band of a chain loses to its left neighbour              */
    /* (ABI 5) which kernel did the work -- the reference prints its launch geometry per run (sw_stage1.cpp:440-448,
     * CUDAligner::printInitialStatistics); a measurement must be able to name the code it measured */
    int32_t strips_first;       /* mixed-height launches: strips [0, strips_first) are strip_rows tall, the others
                                   strip_rows_second (one launch, sw_strip_kernel_pk16_mixed); otherwise = strips, 0 */
    int32_t strip_rows_second;
    int32_t restarts;           /* reruns on the int32 kernels after an overflow report of the packed one */
    int32_t reserved_;
    double seed_ms;             /* wall time of the seed pass (ABI 7: anchors from column stripes + the segments between them swept side
                                   by side in band mode; MI355SW_F_STAIRCASE_SEED or no anchors: a staircase of tiles) that gave a pruning run of a large matrix its first
                                   bound (0: none ran): a staircase of tiles along the diagonal, swept before the main
                                   launch; the score it finds is a real alignment's, so the bound is valid whatever it is */
    char kernel[64];            /* the kernel instantiation of the (last) main launch, as the profiler names it without
                                   its namespace: "sw_strip_kernel_pk16_mixed<12,11,true,true>", "sw_strip_kernel_pk16<16,
                                   false,false,true>" (rows per half, track, SW, prune), "sw_strip_kernel<8,true,true,true>"
                                   (rows per lane, SW, profile, track), "sw_batch_kernel_pk16<2,...>" */
} mi355sw_stats;

/* ---- life cycle: IAligner::initialize/finalize (IAligner.hpp:186,226; X/CUDAligner.cpp:137-174) ---- */
int mi355sw_create(const mi355sw_config* config, mi355sw_handle** out);
/* every field of `config` but `device` and `stream_priority` again, for the calls that follow (no stream may be active):
 * what a MASA extension does when its parameters change between stages (IAlignerParameters, X/CUDAlignerParameters.cpp) */
int mi355sw_configure(mi355sw_handle* h, const mi355sw_config* config);
void mi355sw_destroy(mi355sw_handle* h);
const char* mi355sw_last_error(mi355sw_handle* h);
int mi355sw_abi_version(void);
/* identity of the device code this library was built from: sha256 (16 hex digits) over the kernel sources and their
 * build recipe (csrc/build_id.py).  Measurements (rocprof counters) are keyed by it. */
const char* mi355sw_build_id(void);

/* Strip height of the partitions that follow (mi355sw_config.rows_per_lane; 0 = back to the cost model).  The cost model
 * minimises the time of a FULL sweep; a caller that knows its partitions will be stopped early by their goal (stage 2:
 * tall partitions, the goal a few hundred thousand rows down) asks for short strips instead -- the first strip's sweep
 * and one hop per strip down to the goal row are all that such a partition costs. */
int mi355sw_set_rows_per_lane(mi355sw_handle* h, int32_t rows_per_lane);

/* IAligner::getCapabilities (IAligner.hpp:159; X/CUDAligner.cpp:87-111) */
int mi355sw_get_capabilities(mi355sw_handle* h, mi355sw_capabilities* out);
/* IAligner::getScoreParameters (IAligner.hpp:177; X/CUDAligner.cpp:56-59): +1/-3/-3/-2 */
int mi355sw_get_score_parameters(mi355sw_handle* h, mi355sw_score_params* out);

/* IAligner::setSequences / unsetSequences (IAligner.hpp:197,204; X/CUDAligner.cpp:229-288).
 * Bytes are compared raw (upper-cased FASTA bytes, X/CUDAligner.cu:276-289); the sequences are
 * copied to HBM, the caller keeps ownership of its buffers. */
int mi355sw_set_sequences(mi355sw_handle* h, const char* seq0, const char* seq1, int32_t seq0_len, int32_t seq1_len);
int mi355sw_unset_sequences(mi355sw_handle* h);

/* IAligner::alignPartition (IAligner.hpp:216; AbstractDiagonalAligner.cpp:59-70) */
int mi355sw_align_partition(mi355sw_handle* h, const mi355sw_partition* partition,
                            const mi355sw_manager* manager, void* user);

/* Several independent partitions in ONE kernel launch -- the partitions stage 3 refines (M/stage3/sw_stage3.cpp:210-262
 * processes them one alignPartition call after the other; each is a tall, narrow NW partition that stops as soon as its
 * goal shows up: milliseconds of critical path on a handful of wavefronts, the rest of the GPU idle).  Semantics: as if
 * mi355sw_align_partition(h, &partitions[k], managers[k], users[k]) had been called for every k -- same hooks, same
 * order PER partition, all from the calling thread -- but the partitions run side by side, so the calls of different
 * managers interleave.  Every partition needs its own manager state (borders, goal, sinks).  Partitions the batch
 * cannot take (more than 14 common byte values, >= 32 Mi rows, block pruning or block scores wanted, an overflow report
 * of the packed kernel) are run one by one after the others.  mi355sw_get_stats: sums over the call. */
int mi355sw_align_partitions(mi355sw_handle* h, int32_t count, const mi355sw_partition* partitions,
                             const mi355sw_manager* const* managers, void* const* users);

/* AbstractBlockProcessor::processBlock (M/libmasa/processors/AbstractBlockProcessor.hpp:27-37,
 * semantics CPUBlockProcessor.cpp:95-112): row[k] = (H,F) of (i0-1,j0+k) in/out, col[0] = diagonal
 * H, col[k+1] = (H,E) of (i0+k,j0-1) in/out; returns the first strict maximum in row-major order. */
int mi355sw_process_block(mi355sw_handle* h, mi355sw_cell* row, mi355sw_cell* col,
                          int32_t i0, int32_t j0, int32_t i1, int32_t j1, int32_t recurrence_type,
                          mi355sw_score* best);

/* IAligner::matchLastColumn (IAligner.hpp:249; AlignerUtils::matchColumn, AlignerUtils.cpp:50-107) */
int mi355sw_match_last_column(mi355sw_handle* h, const mi355sw_cell* buffer, const mi355sw_cell* base,
                              int32_t len, int32_t goal_score, mi355sw_match_result* out);

/* IAligner::getProgressString (IAligner.hpp:366) -- async-safe */
int mi355sw_progress(mi355sw_handle* h, char* buf, size_t len);
/* IAligner::getProcessedCells (IAligner.hpp:377) */
long long mi355sw_processed_cells(mi355sw_handle* h);
int mi355sw_get_stats(mi355sw_handle* h, mi355sw_stats* out);

/* ---- streaming form used by the column-band (multi-GPU) driver --------------------------------
 * Replaces the reference's socket-fed border streams (M/common/io/SocketCells{Reader,Writer}.cpp,
 * BufferedCells*.cpp) : the strip kernel runs once for the whole band while the host feeds the
 * first column and drains the last column in row segments. */
typedef struct {
    int32_t recurrence_type;
    int32_t first_row_init_type, first_row_start_offset;   /* InitialCellsReader startOffset */
    const mi355sw_cell* first_row;      /* n+1 cells incl. corner when INIT_WITH_CUSTOM_DATA, else NULL */
    int32_t first_column_init_type, first_column_start_offset;
    int32_t stream_first_column;        /* 1: rows arrive through mi355sw_stream_feed_column() */
    const mi355sw_cell* first_column;   /* m+1 cells incl. corner when CUSTOM and not streamed */
    int32_t want_last_column;           /* keep (H,E) of column j1 for mi355sw_stream_read_column() */
    int32_t want_last_row;
    int32_t special_row_interval;       /* rows; 0 = none (rounded up to whole strips like
                                           AbstractDiagonalAligner::isSpecialRow :466-478) */
    int32_t track_best;                 /* mustDispatchScores() */
    int32_t force_int32;                /* 1: int32 kernel even where the packed 16-bit one applies */
    int32_t prune_blocks;               /* mustPruneBlocks(): skip 64-column slabs of a strip that cannot matter
                                           (AbstractBlockPruning::isBlockPrunable, AbstractBlockPruning.cpp:70-111).
                                           SMITH_WATERMAN: slabs that cannot reach the running best score.
                                           NEEDLEMAN_WUNSCH: the caller states that the alignment is GLOBAL -- its score is
                                           read from the LAST cell of the super-partition -- and slabs through which no
                                           path can reach a running lower bound of that cell are skipped (:92-104: gap terms
                                           + the lower bound `score - dec`); honoured only with track_best = 0.  Skipped
                                           cells read H = 0 (local) or -INF (global) with E = F = -INF, lower bounds of
                                           the true cells; every cell an optimal path can use stays exact. */
    int32_t prune_rows, prune_cols;     /* rows/columns left from the partition origin to the end of the
                                           SUPER-partition (max_i - i0, max_j - j0); 0 = the partition's own */
    int32_t first_column_port;          /* 1: the first column arrives in this handle's inbound column port, written by
                                           the previous band's GPU (first_column_init_type must be CUSTOM_DATA;
                                           first_column[0] = corner cell) */
    int32_t last_column_port;           /* 1: the last column is stored straight into the next band's column port
                                           (opened with mi355sw_port_open/attach) instead of being kept for
                                           mi355sw_stream_read_column(); excludes want_last_column */
    int32_t first_column_resume_rows;   /* restart of the SAME partition on this handle (int32 rerun after
                                           MI355SW_EOVERFLOW16): rows of the streamed first column that were fed
                                           before the restart are still in place and count as fed again */
    int32_t share_best;                 /* 1: the running best score of this stream is shared while the kernel runs --
                                           along a chain of column bands through the column ports (each band's kernel
                                           pushes its best to the next band's port and reads what the next band
                                           publishes, so a score found anywhere reaches every band in both directions),
                                           and with the host through mi355sw_stream_best_hint / _running_best.  Block
                                           pruning then works against the best of the WHOLE matrix, which the
                                           reference gives up when it forks (M/libmasa/libmasa.cpp:1318-1321). */
    int32_t have_initial_bound;         /* 1: `initial_bound` is valid */
    int32_t initial_bound;              /* prune_blocks only: what the pruning bound starts from instead of "nothing known" -- the
                                           score of a local alignment that exists (the best of a run that is being resumed,
                                           of another node: Status::load / AlignerPool::getBestNodeScore in the reference), or
                                           for a global alignment a lower bound of the last cell's score.  Without it, large
                                           matrices get one from the diagonal seed pass (mi355sw_stats.seed_ms).
                                           A value NO alignment reaches would prune the optimum away: a stream that covers
                                           its whole super-partition checks what it found against the bound it was given and
                                           mi355sw_stream_end returns MI355SW_EBOUND when it ended below it (chains of bands:
                                           the driver checks the chain's result, bands.py). */
} mi355sw_stream_params;

int mi355sw_stream_begin(mi355sw_handle* h, const mi355sw_partition* partition, const mi355sw_stream_params* p);
/* The seed pass on its own (a first value for the pruning bound: the score of an alignment the pass really finds along the
 * diagonal of the pair's alignment -- see mi355sw_stats.seed_ms), for callers that divide one matrix among several streams (a chain of column bands, one GPU
 * each: a band cannot make the seed of the whole matrix, and mi355sw_stream_begin leaves it out for streams with column ports
 * or a streamed first column).  `partition` = the WHOLE matrix, borders as the run will have them: zeroes for SMITH_WATERMAN,
 * gap penalties from the origin for NEEDLEMAN_WUNSCH (a global alignment).  *have_bound = 1 and *bound = the value for
 * mi355sw_stream_params.initial_bound of every band (the score of a local alignment that exists / a lower bound of the last
 * cell's score); *have_bound = 0 when there is none: an unrelated pair (local), a matrix below 64 Ki x 16 Ki, sequences the
 * packed kernel cannot take.  No stream may be active on the handle; mi355sw_stats.seed_ms of the next stream reports its time.
 * Reference: the bound a node starts from -- Status::load -> BestScoreList (sw_stage1.cpp:210-217) and the other nodes' best
 * score, AlignerPool::getBestNodeScore (M/common/AlignerPool.cpp:182-184). */
int mi355sw_seed_bound(mi355sw_handle* h, const mi355sw_partition* partition, int32_t recurrence_type, int32_t* have_bound, int32_t* bound);
/* cells[0..len) = (H,E) of rows [row, row+len) of column j0-1 ; rows must arrive in order */
int mi355sw_stream_feed_column(mi355sw_handle* h, int32_t row, const mi355sw_cell* cells, int32_t len);
/* number of DP rows whose strips are complete (monotonic); *finished = 1 when the kernel ended */
int mi355sw_stream_poll(mi355sw_handle* h, int32_t* rows_done, int32_t* finished);
/* (H,E) of rows [row,row+len) of the last column; only rows < rows_done are valid */
int mi355sw_stream_read_column(mi355sw_handle* h, int32_t row, mi355sw_cell* cells, int32_t len);
/* special row k (0-based) / last row: n cells (H,F) of columns j0..j1-1, valid once its strip is done */
int mi355sw_stream_read_special_row(mi355sw_handle* h, int32_t k, int32_t* dp_row, mi355sw_cell* cells, int32_t col, int32_t len);
int mi355sw_stream_read_last_row(mi355sw_handle* h, mi355sw_cell* cells, int32_t col, int32_t len);
int mi355sw_stream_abort(mi355sw_handle* h);
/* share_best streams: `score` is the score of an alignment that exists somewhere in the super-partition (found by
 * another band, another node, a previous run) -- for a GLOBAL alignment (NEEDLEMAN_WUNSCH with prune_blocks): a lower
 * bound of the score in the super-partition's last cell, e.g. the score of any global alignment of the two sequences
 * somebody has computed: the running kernel folds it into its pruning bound at its next strip hand-over.  Lower bounds
 * only -- a value no alignment reaches would prune the optimum away. */
int mi355sw_stream_best_hint(mi355sw_handle* h, int32_t score);
/* share_best streams: the best score (global alignments: the best lower bound of the last cell's score) the kernel knew
 * at its last strip hand-over -- its own cells and every hint it received; -MI355SW_INF before the first.  May be called
 * from any thread while the stream runs. */
int mi355sw_stream_running_best(mi355sw_handle* h, int32_t* score);
/* waits for the kernel; best = canonical (max score, min i, min j), sequence-relative 0-based cell */
int mi355sw_stream_end(mi355sw_handle* h, mi355sw_score* best, int32_t* n_special_rows);
/* per-strip best scores of the finished stream (for dispatch_score); returns count written */
int mi355sw_stream_strip_scores(mi355sw_handle* h, mi355sw_score* out, int32_t max_count);
/* ---- column ports: the boundary column of a band chain, GPU to GPU over xGMI -------------------------------
 * (and, for share_best streams, the running best score: two more words in the port's control block, one written by
 *  each side, next to the row counter)
 * Replaces the reference's socket chain between forked processes (M/libmasa/libmasa.cpp:540-642,
 * M/common/io/SocketCellsWriter.cpp, BufferedCellsWriter.cpp:57-66).  Band g+1 owns an inbound PORT in the HBM
 * of its own GPU: (H,E) cells of its first column plus a row counter (fine-grained memory).  Band g maps that
 * port (hipIpc between the rank processes, direct peer access inside one process); its strip kernel stores the
 * last-column cells of every finished strip into it and then publishes the row count with a system-scope
 * release store; band g+1's kernel polls the counter in its own HBM.  No host, no copy queue and no PCIe in the
 * loop.  A port is used for ONE run of the chain (create it before the handle is handed to the neighbour). */
typedef struct {
    unsigned char ipc[64];   /* hipIpcMemHandle_t */
    int64_t bytes;
    int32_t rows;            /* capacity in DP rows (cells 1..rows; cell 0 = corner, written by the owner) */
    int32_t device;          /* HIP ordinal of the owning GPU */
} mi355sw_port_handle;
/* owner side (band g+1): allocate the inbound port for `rows` rows, counter = 0; `out` may be sent to another process */
int mi355sw_port_create(mi355sw_handle* h, int32_t rows, mi355sw_port_handle* out);
/* writer side (band g), other process: map the neighbour's port as this handle's outbound port */
int mi355sw_port_open(mi355sw_handle* h, const mi355sw_port_handle* remote);
/* writer side, same process: `downstream`'s inbound port becomes `h`'s outbound port (peer access is enabled
 * when the two handles sit on different GPUs) */
int mi355sw_port_attach(mi355sw_handle* h, mi355sw_handle* downstream);
/* owner side: counter back to 0 for another run of the chain; the caller makes sure that no writer is active
 * (bands.py: the owner resets, THEN tells the writer to start) */
int mi355sw_port_reset(mi355sw_handle* h);
/* rows published so far in the inbound port (reads the counter from HBM; diagnostics, tests) */
int mi355sw_port_rows_ready(mi355sw_handle* h, int32_t* rows);
/* copies cells [row, row+len) of the inbound port's column to the host (diagnostics, tests) */
int mi355sw_port_read(mi355sw_handle* h, int32_t row, mi355sw_cell* cells, int32_t len);
/* device addresses of the inbound port for callers that move the column themselves (RCCL recv, hipMemcpyPeer):
 * cells[1 + row] and the int32 row counter, which they must store with system scope AFTER the cells */
int mi355sw_port_local_pointers(mi355sw_handle* h, void** cells, void** counter);
/* release the inbound port and unmap the outbound one */
int mi355sw_port_close(mi355sw_handle* h);

/* ---- stage 4: Myers-Miller refinement of the stage-3 crosspoints on the GPU ---------------------------------
 * Replaces MASA-Core's CPU stage 4 (M/stage4/sw_stage4.cpp:880-960, strategy STAGE_4_STRATEGY_OPTIMIZED =
 * ort_split_2 :293-380; four pthreads).  `in` = the crosspoints of M/common/CrosspointsFile (crosspoint_03.NN) in
 * increasing order, absolute 1-based DP coordinates of the sequences given to mi355sw_set_sequences; every
 * partition between two consecutive crosspoints is cut in the middle of its longer side, over and over, until none
 * is larger than `max_partition_size` (the reference's --stage-4 limit: 16).  `*out` (free it with mi355sw_free) is
 * the list MASA-Core writes to crosspoint_04.NN: same points, same order, same tie-breaks. */
typedef struct { int32_t type, i, j, score; } mi355sw_crosspoint;   /* M/common/Crosspoint.hpp:40-50 */
typedef struct { int32_t steps; double kernel_ms; int64_t dp_cells; int64_t partitions; } mi355sw_stage4_stats;
int mi355sw_stage4(mi355sw_handle* h, const mi355sw_crosspoint* in, int32_t count, int32_t max_partition_size,
                   mi355sw_crosspoint** out, int32_t* out_count, mi355sw_stage4_stats* stats);
void mi355sw_free(void* p);

/* ---- stage 5: the exact alignment of every partition between consecutive crosspoints ----------------------------
 * Replaces the per-partition full-matrix traceback of MASA-Core's stage 5 (M/stage5/sw_stage5.cpp: sw() :83-319, the
 * loop of stage5() :322-485).  Host code, like the reference's (no handle, no GPU): `crosspoints` = crosspoint_04.NN
 * (partitions of at most 16 x 16 after stage 4; the reference's own limit of 8192 per side is enforced), seq0 / seq1 =
 * the sequence data as the reference's Sequence::getData() gives it (modifiers applied).  Returns the gap events in
 * the order the reference's traceback emits them -- gaps0[k] = DP row i of an event for sequence 0's gap list
 * (_dot type 2, :64-80), gaps1[k] = DP column j of one for sequence 1's list (type 1); release both with mi355sw_free --
 * and the alignment's totals.  MI355SW_ETRACEBACK: partition *failed_at has no traceback (the crosspoints are not on
 * one optimal path). */
typedef struct { int64_t score, matches, mismatches, gap_open, gap_extensions; } mi355sw_stage5_totals;
int mi355sw_stage5(const char* seq0, int32_t seq0_len, const char* seq1, int32_t seq1_len, const mi355sw_crosspoint* crosspoints,
                   int32_t count, int32_t** gaps0, int64_t* n_gaps0, int32_t** gaps1, int64_t* n_gaps1,
                   mi355sw_stage5_totals* totals, int32_t* failed_at);

/* ---- stage 6: the text of alignment.NN.txt from the first block on -------------------------------------------------
 * Replaces printText of M/stage6/sw_stage6.cpp:60-262 after its three header lines (which the caller writes: they need the
 * sequences' descriptions and trim positions): blocks of 60 columns -- "Query:" line, match marks with the block's and the
 * running score, "Sbjct:" line -- and the summary.  seq0 / seq1 = the FORWARD data of the whole sequences
 * (Sequence::getForwardData()), (i0, j0) -> (i1, j1) = Alignment start / end (absolute 1-based positions; all four -1: "no
 * alignment produced"), gaps0 / gaps1 = the alignment's gap lists as (position, length) pairs in Alignment::finalize's
 * order.  `*text` (release it with mi355sw_free) is not NUL-counted in *text_len.  MI355SW_ETRACEBACK: the text re-scores
 * to something else than raw_score ("Stage6 error: Alignment score is different", :243-247); totals are filled either way. */
int mi355sw_stage6_text(const char* seq0, int32_t seq0_len, const char* seq1, int32_t seq1_len, int32_t i0, int32_t j0,
                        int32_t i1, int32_t j1, const int32_t* gaps0, int32_t n_gaps0, const int32_t* gaps1, int32_t n_gaps1,
                        int64_t raw_score, char** text, int64_t* text_len, mi355sw_stage5_totals* totals);

/* ---- crosspoint files as text ------------------------------------------------------------------------------------------
 * Replaces CrosspointsFile::save / write (M/common/CrosspointsFile.cpp:99-160): "START", one "type,i,j,score" line per
 * crosspoint, "END" -- the bytes of crosspoint_NN.II.  Host code (no handle, no GPU); `*text` is released with mi355sw_free
 * and not NUL-counted in *text_len.  (crosspoint_04 of BASELINE config 3 is 4.6 M lines.) */
int mi355sw_crosspoints_text(const mi355sw_crosspoint* points, int64_t count, char** text, int64_t* text_len);

/* device enumeration: X/cuda_util.cpp:191-287 (--list-gpus, GPU weights) */
int mi355sw_device_count(void);
int mi355sw_device_info(int32_t device, char* name, size_t name_len, int32_t* compute_units, int32_t* clock_mhz, int64_t* hbm_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MI355SW_H_ */
