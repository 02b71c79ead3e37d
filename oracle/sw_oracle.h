/*
 * oracle/sw_oracle.h -- TEST INFRASTRUCTURE ONLY (parity checker / cpu_baseline).
 *
 * Plain-C restatement of the reference's Stage-1 CPU path.  Nothing in the
 * product (masa-cudalign_amd/, include/) may include, link or call this;
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this file
 * against outputs of the reference's own MASA-Core CPU path (oracle/_ref,
 * built from /root/reference by oracle/build_ref.sh) and against the fixtures
 * under tests/golden/ that oracle/make_golden.py generated from it.
 *
 * All file:line citations are relative to
 *   /root/reference/masa-cudalign-4.0.2.1028/libs/masa-core/src/   ("M/")
 */
#ifndef SW_ORACLE_H_
#define SW_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* M/libmasa/libmasaTypes.hpp:35-41 (cell_t), :46 (INF), :88-95 (score_t) */
typedef struct { int h; int f; } oc_cell;   /* .f aliases .e: rows carry (H,F), columns (H,E) */
typedef struct { int i; int j; int score; } oc_score;
#define OC_INF 999999999

/* M/libmasa/IManager.hpp:31-47 */
#define OC_NEEDLEMAN_WUNSCH 0
#define OC_SMITH_WATERMAN   1
#define OC_INIT_WITH_ZEROES      0
#define OC_INIT_WITH_GAPS        1
#define OC_INIT_WITH_CUSTOM_DATA 2
#define OC_INIT_WITH_GAPS_OPENED 3

/* M/libmasa/processors/CPUBlockProcessor.cpp:36-44 (hard-coded scoring) */
#define OC_MATCH     1
#define OC_MISMATCH (-3)
#define OC_GAP_EXT   2
#define OC_GAP_OPEN  3

/* where the best score may lie (M/common/Job.hpp AT_*; sw_stage1.cpp:388-403) */
#define OC_BEST_NOWHERE   0
#define OC_BEST_ANYWHERE  1   /* every cell (SW local)            */
#define OC_BEST_LAST_CELL 2   /* only H[m][n] (global NW, "++")   */
#define OC_BEST_LAST_ROW  3
#define OC_BEST_LAST_COL  4
#define OC_BEST_LAST_ROW_OR_COL 5

/* CPUBlockProcessor::processBlock, M/libmasa/processors/CPUBlockProcessor.cpp:113-174 */
oc_score oracle_process_block(const unsigned char* seq0, const unsigned char* seq1,
        oc_cell* row, oc_cell* col, int i0, int j0, int i1, int j1, int recurrence);

/* InitialCellsReader::read, M/common/io/InitialCellsReader.cpp:84-108.
 * `position` is the absolute reader position (startOffset + cells already read). */
void oracle_initial_cells(int init_type, int position, oc_cell* buffer, int len);

/* AlignerUtils::matchColumn, M/libmasa/utils/AlignerUtils.cpp:50-107.
 * returns 1 found (k,score,type filled), 0 not found, <0 error type */
int oracle_match_column(const oc_cell* buffer, const oc_cell* base, int len, int goal,
        int* k, int* score, int* type);

typedef struct {
    const unsigned char* seq0; int m;    /* vertical   */
    const unsigned char* seq1; int n;    /* horizontal */
    int recurrence;                      /* OC_SMITH_WATERMAN / OC_NEEDLEMAN_WUNSCH */
    int first_row_type, first_col_type;  /* OC_INIT_* */
    int row_start_offset;                /* InitialCellsReader startOffset of the first row (band j0) */
    int col_start_offset;                /* same for the first column */
    const oc_cell* custom_first_row;     /* n+1 cells incl. corner, when CUSTOM */
    const oc_cell* custom_first_col;     /* m+1 cells incl. corner, when CUSTOM */
    int block_h, block_w;                /* grid geometry (AbstractBlockAligner blocks) */
    int special_row_interval;            /* 0 = none; rows (M/.../AbstractBlockAligner.cpp:418-439) */
    int want_last_row, want_last_col;
    int pruning;                         /* BlockPruningGenericN2 + AbstractBlockPruning::isBlockPrunable */
    int max_i, max_j;                    /* super-partition ends for the pruning bound (0 => m,n) */
    int best_mode;                       /* OC_BEST_* */
} oc_params;

typedef struct {
    oc_score best;                       /* 1-based DP coords (AlignerManager.cpp:411-415), -INF if none */
    long long blocks_total, blocks_pruned;
    int n_special_rows;
    int* special_row_ids;                /* DP row index i of each flushed row (malloc) */
    oc_cell* special_rows;               /* n_special_rows x (n+1) cells (malloc), cell 0 = first-column cell, f=-INF */
    oc_cell* last_row;                   /* n+1 cells (malloc) if want_last_row */
    oc_cell* last_col;                   /* m+1 cells (malloc) if want_last_col */
    oc_score* block_scores;              /* serial schedule only: best cell of block (bx, by) at [bx * grid_h + by], 0-based */
    int grid_w, grid_h;
} oc_result;

int oracle_stage1(const oc_params* p, oc_result* r);
void oracle_free_result(oc_result* r);

/* Same pass run as an anti-diagonal wavefront of blocks over `threads` pthreads
 * (no pruning, SW/NW, best ANYWHERE or LAST_CELL only) -- cpu_baseline "port". */
int oracle_stage1_mt(const oc_params* p, oc_result* r, int threads);

#ifdef __cplusplus
}
#endif
#endif
