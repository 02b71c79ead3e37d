// See Mi355Aligner.hpp.  Error policy follows the reference extension (cutilSafeCall, X/cuda_util.h:34-42):
// a failing engine call is fatal for the MASA process.
#include "Mi355Aligner.hpp"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

Mi355Aligner::Mi355Aligner(int device, int rowsPerLane, int waves) : handle(NULL) {
    memset(&config, 0, sizeof(config));
    config.device = device;
    config.rows_per_lane = rowsPerLane;
    config.waves = waves;
    mi355sw_score_params sp;
    mi355sw_get_score_parameters(NULL, &sp);
    score_params.match = sp.match;
    score_params.mismatch = sp.mismatch;
    score_params.gap_open = sp.gap_open;
    score_params.gap_ext = sp.gap_ext;
    params = new Mi355AlignerParameters();
    if (device >= 0) params->setGPU(device);
    // fork weights are asked for lazily (getForkWeights): enumerating the GPUs costs a child process
    weightsKnown = false;
    clearStatistics();
    progress[0] = 0;
}

Mi355Aligner::~Mi355Aligner() {
    if (handle) mi355sw_destroy(handle);
}

void Mi355Aligner::check(int rc, const char* what) {
    if (rc != MI355SW_OK) {
        fprintf(stderr, "Mi355Aligner: %s failed (%d): %s\n", what, rc, handle ? mi355sw_last_error(handle) : "");
        exit(1);
    }
}

// one forked instance per GPU, seq1 split in proportion to compute units x clock (X/CUDAligner.cpp:63-66); only when
// MASA-Core asks (--fork), and before it forks: the devices are enumerated in a throw-away child process
const int* Mi355Aligner::getForkWeights() {
    if (!weightsKnown) {
        int weights[64];
        const int gpus = Mi355AlignerParameters::deviceWeights(weights, 64);
        if (gpus > 0) setForkCount(gpus, weights);
        else setForkCount(1);
        weightsKnown = true;
    }
    return AbstractAligner::getForkWeights();
}

aligner_capabilities_t Mi355Aligner::getCapabilities() {
    aligner_capabilities_t c;
    mi355sw_capabilities k;
    memset(&k, 0, sizeof(k));
    if (handle) mi355sw_get_capabilities(handle, &k);
    else {
        k.dispatch_last_cell = k.dispatch_last_row = k.dispatch_last_column = k.dispatch_special_row = 1;
        k.dispatch_scores = k.dispatch_best_score = k.customize_first_row = k.customize_first_column = 1;
        k.process_partition = k.needleman_wunsch = k.smith_waterman = k.fork_processes = 1;
    }
    c.dispatch_last_cell = k.dispatch_last_cell;
    c.dispatch_last_row = k.dispatch_last_row;
    c.dispatch_last_column = k.dispatch_last_column;
    c.dispatch_special_row = k.dispatch_special_row;
    c.dispatch_special_column = k.dispatch_special_column;
    c.dispatch_scores = k.dispatch_scores;
    c.dispatch_block_scores = k.dispatch_block_scores;
    c.dispatch_best_score = k.dispatch_best_score;
    c.customize_first_row = k.customize_first_row;
    c.customize_first_column = k.customize_first_column;
    c.process_partition = k.process_partition;
    c.variable_penalties = k.variable_penalties;
    c.block_pruning = k.block_pruning;
    c.needleman_wunsch = k.needleman_wunsch;
    c.smith_waterman = k.smith_waterman;
    c.fork_processes = k.fork_processes;
    c.maximum_seq0_len = k.maximum_seq0_len;
    c.maximum_seq1_len = k.maximum_seq1_len;
    return c;
}

const score_params_t* Mi355Aligner::getScoreParameters() { return &score_params; }
IAlignerParameters* Mi355Aligner::getParameters() { return params; }

void Mi355Aligner::initialize() {
    if (handle) return;
    // GPU selection as in CUDAligner::initialize (X/CUDAligner.cpp:137-150): a forked instance takes the GPU of its
    // fork id (wrapping around), otherwise --gpu, otherwise the fastest device
    if (params->getForkId() != NOT_FORKED_INSTANCE) {
        const int gpus = mi355sw_device_count();
        int id = params->getForkId();
        if (gpus > 0 && id >= gpus) {
            fprintf(stderr, "INFO: Wrapping gpu ID (%d -> %d). (max.: %d).\n", id, id % gpus, gpus);
            id %= gpus;
        }
        params->setGPU(id);
    }
    if (params->getGPU() == MI355_DETECT_FASTEST_GPU) params->setGPU(Mi355AlignerParameters::fastestGPU());
    if (params->getGPU() >= mi355sw_device_count()) {     // --gpu is validated here, after any fork (see processArgument)
        fprintf(stderr, "Mi355Aligner: GPU index %d out of range, %d device(s) (see --list-gpus).\n", params->getGPU(), mi355sw_device_count());
        exit(2);
    }
    config.device = params->getGPU();
    if (params->getWaves() > 0) config.waves = params->getWaves();
    if (params->getStripRows() > 0) config.rows_per_lane = params->getStripRows() / 64;
    if (params->getNoDiagonalSeed()) config.flags |= MI355SW_F_NO_DIAGONAL_SEED;
    config.flags |= params->getEngineFlags();
    config.verbosity = params->getEngineVerbosity();
    if (params->getBlockColumns() > 0) {
        // the grid MASA-Core asks for (AlignerManager::dispatchScore -> BlocksFile::initialize(getGrid())) must be
        // known before the engine has picked anything: fixed strip height, one all three kernel families have
        const int sr = params->getStripRows();
        if (sr != 256 && sr != 512 && sr != 1024) {
            fprintf(stderr, "Mi355Aligner: --block-columns needs --strip-rows=256, 512 or 1024.\n");
            exit(2);
        }
        config.block_score_columns = params->getBlockColumns();
    }
    check(mi355sw_create(&config, &handle), "mi355sw_create");
}

void Mi355Aligner::finalize() {
    if (handle) { mi355sw_destroy(handle); handle = NULL; }
}

void Mi355Aligner::setSequences(const char* seq0, const char* seq1, int seq0_len, int seq1_len) {
    initialize();
    check(mi355sw_set_sequences(handle, seq0, seq1, seq0_len, seq1_len), "mi355sw_set_sequences");
}

void Mi355Aligner::unsetSequences() {
    if (handle) check(mi355sw_unset_sequences(handle), "mi355sw_unset_sequences");
}

void Mi355Aligner::alignPartition(Partition partition) {
    Grid* grid = createGrid(partition);   // AlignerManager asks for getGrid() when it keeps a blocks file (--dump-blocks)
    if (params->getBlockColumns() > 0) {
        grid->setBlockHeight(params->getStripRows());
        grid->setBlockWidth(params->getBlockColumns());
    }
    mi355sw_manager m;
    m.get_recurrence_type = cbRecurrence;
    m.get_special_row_interval = cbSpecialInterval;
    m.get_first_column_init_type = cbFirstColumnType;
    m.get_first_row_init_type = cbFirstRowType;
    m.get_super_partition = cbSuperPartition;
    m.receive_first_row = cbReceiveFirstRow;
    m.receive_first_column = cbReceiveFirstColumn;
    m.dispatch_column = cbDispatchColumn;
    m.dispatch_row = cbDispatchRow;
    m.dispatch_score = cbDispatchScore;
    m.must_continue = cbMustContinue;
    m.must_dispatch_last_cell = cbLastCell;
    m.must_dispatch_last_row = cbLastRow;
    m.must_dispatch_last_column = cbLastColumn;
    m.must_dispatch_special_rows = cbSpecialRows;
    m.must_dispatch_scores = cbScores;
    m.must_prune_blocks = cbPrune;
    m.dispatch_strip_value = NULL;      /* IManager has no value-only score call */
    mi355sw_partition p;
    p.i0 = partition.getI0(); p.j0 = partition.getJ0(); p.i1 = partition.getI1(); p.j1 = partition.getJ1();
    check(mi355sw_align_partition(handle, &p, &m, this), "mi355sw_align_partition");
    mi355sw_stats st;
    if (mi355sw_get_stats(handle, &st) == MI355SW_OK) {
        statCells += st.cells;
        statKernelMs += st.kernel_ms;
        statPruned += st.pruned_cells;
        statPartitions++;
    }
}

int Mi355Aligner::refineCrosspoints(const char* seq0, const char* seq1, int seq0_len, int seq1_len, const int* in_tijs, int count,
                                    int max_partition_size, int** out_tijs, int* out_count, double* kernel_ms) {
    setSequences(seq0, seq1, seq0_len, seq1_len);
    mi355sw_crosspoint* out = NULL;
    int32_t n = 0;
    mi355sw_stage4_stats st;
    memset(&st, 0, sizeof(st));
    check(mi355sw_stage4(handle, (const mi355sw_crosspoint*) in_tijs, count, max_partition_size, &out, &n, &st), "mi355sw_stage4");
    unsetSequences();
    *out_tijs = (int*) out;          /* {type, i, j, score} quadruples; release with mi355sw_free */
    *out_count = n;
    if (kernel_ms) *kernel_ms = st.kernel_ms;
    return st.steps;
}

#define SELF ((Mi355Aligner*) u)
int32_t Mi355Aligner::cbRecurrence(void* u) { return SELF->getRecurrenceType(); }
int32_t Mi355Aligner::cbSpecialInterval(void* u) { return SELF->getSpecialRowInterval(); }
int32_t Mi355Aligner::cbFirstColumnType(void* u) { return SELF->getFirstColumnInitType(); }
int32_t Mi355Aligner::cbFirstRowType(void* u) { return SELF->getFirstRowInitType(); }
void Mi355Aligner::cbSuperPartition(void* u, mi355sw_partition* out) {
    Partition p = SELF->getSuperPartition();
    out->i0 = p.getI0(); out->j0 = p.getJ0(); out->i1 = p.getI1(); out->j1 = p.getJ1();
}
/* mi355sw_cell and cell_t are layout-identical (libmasaTypes.hpp:35-41) */
void Mi355Aligner::cbReceiveFirstRow(void* u, mi355sw_cell* b, int32_t len) { SELF->receiveFirstRow((cell_t*) b, len); }
void Mi355Aligner::cbReceiveFirstColumn(void* u, mi355sw_cell* b, int32_t len) { SELF->receiveFirstColumn((cell_t*) b, len); }
void Mi355Aligner::cbDispatchColumn(void* u, int32_t j, const mi355sw_cell* b, int32_t len) { SELF->dispatchColumn(j, (const cell_t*) b, len); }
void Mi355Aligner::cbDispatchRow(void* u, int32_t i, const mi355sw_cell* b, int32_t len) { SELF->dispatchRow(i, (const cell_t*) b, len); }
void Mi355Aligner::cbDispatchScore(void* u, mi355sw_score s, int32_t bx, int32_t by) {
    score_t t;
    t.i = s.i; t.j = s.j; t.score = s.score;
    SELF->dispatchScore(t, bx, by);
}
int32_t Mi355Aligner::cbMustContinue(void* u) { return SELF->mustContinue() ? 1 : 0; }
int32_t Mi355Aligner::cbLastCell(void* u) { return SELF->mustDispatchLastCell() ? 1 : 0; }
int32_t Mi355Aligner::cbLastRow(void* u) { return SELF->mustDispatchLastRow() ? 1 : 0; }
int32_t Mi355Aligner::cbLastColumn(void* u) { return SELF->mustDispatchLastColumn() ? 1 : 0; }
int32_t Mi355Aligner::cbSpecialRows(void* u) { return SELF->mustDispatchSpecialRows() ? 1 : 0; }
int32_t Mi355Aligner::cbScores(void* u) { return SELF->mustDispatchScores() ? 1 : 0; }
// --prune-global: MASA-Core's stage 1 switches pruning off unless the alignment may end anywhere (sw_stage1.cpp:219-225) although
// the bound for a global alignment exists (AbstractBlockPruning.cpp:92-109).  The engine has it; it is offered for the
// partitions whose score is read from the last cell -- never for stages 2 and 3, whose partitions end at a goal
// (they want scores or a last column, and the engine ignores the request then).
int32_t Mi355Aligner::cbPrune(void* u) {
    if (SELF->mustPruneBlocks()) return 1;
    return (SELF->params->getPruneGlobal() && SELF->getRecurrenceType() == NEEDLEMAN_WUNSCH && SELF->mustDispatchLastCell()) ? 1 : 0;
}
#undef SELF

void Mi355Aligner::clearStatistics() { statCells = 0; statKernelMs = 0; statPartitions = 0; statPruned = 0; }
void Mi355Aligner::printInitialStatistics(FILE* file) {
    char name[128]; int cus = 0, mhz = 0; long long bytes = 0;
    if (mi355sw_device_info(config.device < 0 ? 0 : config.device, name, sizeof(name), &cus, &mhz, (int64_t*) &bytes) == MI355SW_OK)
        fprintf(file, "GPU: %s, %d CUs, %d MHz, %.1f GB\n", name, cus, mhz, bytes / 1e9);
}
void Mi355Aligner::printStageStatistics(FILE* file) {}
void Mi355Aligner::printFinalStatistics(FILE* file) {}
void Mi355Aligner::printStatistics(FILE* file) {
    fprintf(file, "\n===== MI355 ENGINE =====\nPartitions: %d\nCells: %lld\nPruned cells: %lld\nKernel time: %.3f ms\n", statPartitions, statCells, statPruned, statKernelMs);
    if (statKernelMs > 0) fprintf(file, "Kernel GCUPS: %.2f\n", statCells / statKernelMs / 1e6);
}
const char* Mi355Aligner::getProgressString() const {
    if (handle) mi355sw_progress(handle, progress, sizeof(progress));
    return progress;
}
long long Mi355Aligner::getProcessedCells() { return handle ? mi355sw_processed_cells(handle) : 0; }
