/*
 * oracle/ref_driver.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin driver that links the *real* MASA-Core objects compiled from
 * /root/reference (see oracle/build_ref.sh) and runs the reference's own
 * stage1()..stage6() on a FASTA pair.  It exists only to pin oracle/sw_oracle.c
 * and the HIP engine against the reference's own CPU path and to generate the
 * fixtures under tests/golden/.
 *
 * Two reference files on this path cannot be compiled in this image because
 * they include autoconf/awk generated headers (config.h / default.h):
 *   libs/masa-core/src/libmasa/libmasa.cpp            (CLI, getopt table)
 *   libs/masa-core/src/libmasa/aligners/AbstractBlockAligner.cpp
 * (and AbstractDiagonalAligner.*, configs/Configs.cpp which nothing here uses).
 * Everything else (82 sources: CPUBlockProcessor, Grid, pruning/*, AlignerManager,
 * BestScoreList, sra/*, io/*, stage1..6, Job ...) is the unmodified reference.
 *
 * What this file re-states (my code, following the cited reference lines):
 *   - SerialBlockAligner: the block schedule of AbstractBlockAligner
 *     (AbstractBlockAligner.cpp:276-327 alignPartition, :362-392 processBlock,
 *      :418-449 isSpecialRow/isSpecialColumn, :141-167 capabilities) with the serial
 *      scheduler of SURVEY.md section 8c.  The cell arithmetic is NOT restated here: it is the
 *      reference's CPUBlockProcessor::processBlock, the pruning is the reference's
 *      BlockPruningGenericN2, the manager is the reference's AlignerManager.
 *   - main(): the part of libmasa_entry_point (libmasa.cpp:762-1400) that builds
 *     the Job and calls the stages, with a minimal flag parser.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>

#include "libmasa/libmasa.hpp"
#include "libmasa/aligners/AbstractAligner.hpp"
#include "libmasa/processors/CPUBlockProcessor.hpp"
#include "libmasa/pruning/BlockPruningGenericN2.hpp"
#include "libmasa/parameters/BlockAlignerParameters.hpp"
#include "common/Common.hpp"
#include "common/Job.hpp"

#ifdef USE_MI355_ALIGNER
#include "Mi355Aligner.hpp"   /* masa-cudalign_amd/host: the product's IAligner adapter (drop-in test) */
#endif

int stage1(Job* job);
crosspoint_t stage2(Job* job, int id);
int stage3(Job* job, int id);
int stage4(Job* job, int id);
int stage5(Job* job, int id);
int stage6(Job* job, int id);

/* Same constants as AbstractBlockAligner.cpp:41-45 / CUDAligner.hpp:77-98 */
#define REF_MATCH     (1)
#define REF_MISMATCH  (-3)
#define REF_GAP_EXT   (2)
#define REF_GAP_OPEN  (3)

class SerialBlockAligner : public AbstractAligner {
public:
    SerialBlockAligner(int bh, int bw) : blockH(bh), blockW(bw), row(NULL), col(NULL),
            totalBlocks(0), prunedBlocks(0) {
        score_params.match = REF_MATCH;
        score_params.mismatch = REF_MISMATCH;
        score_params.gap_open = REF_GAP_OPEN;
        score_params.gap_ext = REF_GAP_EXT;
        params = new BlockAlignerParameters();
        processor = new CPUBlockProcessor();
        pruner = new BlockPruningGenericN2();
        setForkCount(1);
    }
    virtual ~SerialBlockAligner() {}

    /* AbstractBlockAligner.cpp:141-167 */
    virtual aligner_capabilities_t getCapabilities() {
        aligner_capabilities_t c;
        c.smith_waterman = SUPPORTED;
        c.needleman_wunsch = SUPPORTED;
        c.block_pruning = SUPPORTED;
        c.customize_first_column = SUPPORTED;
        c.customize_first_row = SUPPORTED;
        c.dispatch_last_cell = NOT_SUPPORTED;
        c.dispatch_last_column = SUPPORTED;
        c.dispatch_last_row = SUPPORTED;
        c.dispatch_special_column = SUPPORTED;
        c.dispatch_special_row = SUPPORTED;
        c.dispatch_block_scores = SUPPORTED;
        c.dispatch_scores = SUPPORTED;
        c.process_partition = SUPPORTED;
        c.variable_penalties = NOT_SUPPORTED;
        c.fork_processes = SUPPORTED;
        c.maximum_seq0_len = 0;
        c.maximum_seq1_len = 0;
        return c;
    }
    virtual const score_params_t* getScoreParameters() { return &score_params; }
    virtual IAlignerParameters* getParameters() { return params; }
    virtual void initialize() {}
    virtual void finalize() {}
    virtual void setSequences(const char* s0, const char* s1, int l0, int l1) {
        processor->setSequences(s0, s1, l0, l1);
    }
    virtual void unsetSequences() { processor->unsetSequences(); }
    virtual void clearStatistics() { totalBlocks = prunedBlocks = 0; }
    virtual void printInitialStatistics(FILE*) {}
    virtual void printStageStatistics(FILE*) {}
    virtual void printFinalStatistics(FILE*) {}
    virtual void printStatistics(FILE* f) {
        fprintf(f, "Pruned Blocks: %d / %d\n", prunedBlocks, totalBlocks);
    }
    virtual const char* getProgressString() const { return ""; }
    virtual long long getProcessedCells() { return 0; }

    /* AbstractBlockAligner.cpp:276-327 with the serial schedule of SURVEY.md 8c */
    virtual void alignPartition(Partition partition) {
        Grid* grid = createGrid(partition);
        grid->setBlockHeight(blockH);
        grid->setBlockWidth(blockW);
        initializeBlockPruning(pruner);

        const int gw = grid->getGridWidth();
        const int gh = grid->getGridHeight();
        row = new cell_t*[gw];
        for (int j = 0; j < gw; j++) row[j] = new cell_t[grid->getBlockWidth(j, 0)];
        col = new cell_t*[gh];
        for (int i = 0; i < gh; i++) col[i] = new cell_t[grid->getBlockHeight(0, i) + 1];
        std::vector<std::vector<score_t> > scores(gw, std::vector<score_t>(gh));
        for (int bx = 0; bx < gw; bx++) for (int by = 0; by < gh; by++) {
            scores[bx][by].i = -1; scores[bx][by].j = -1; scores[bx][by].score = -INF;
        }

        cell_t dummy;
        receiveFirstColumn(&dummy, 1);
        receiveFirstRow(&dummy, 1);

        for (int by = 0; by < gh; by++) {
            for (int bx = 0; bx < gw; bx++) {
                int i0, j0, i1, j1;
                grid->getBlockPosition(bx, by, &i0, &j0, &i1, &j1);
                if (by == 0) receiveFirstRow(row[bx], j1 - j0);
                if (bx == 0) {
                    col[by][0] = getFirstColumnTail();
                    receiveFirstColumn(col[by] + 1, i1 - i0);
                }
                if (bx == 0 && isSpecialRow(by)) {
                    cell_t c = col[by][i1 - i0]; c.f = -INF; dispatchRow(i1, &c, 1);
                }
                if (by == 0 && isSpecialColumn(bx)) {
                    cell_t c = row[bx][j1 - j0 - 1]; c.f = -INF; dispatchColumn(j1, &c, 1);
                }
                /* AbstractBlockAligner.cpp:362-392 */
                totalBlocks++;
                if (!pruner->isBlockPruned(bx, by)) {
                    scores[bx][by] = processor->processBlock(row[bx], col[by], i0, j0, i1, j1,
                            getRecurrenceType());
                    pruner->pruningUpdate(bx, by, scores[bx][by].score);
                } else {
                    prunedBlocks++;
                }
                if (isSpecialRow(by)) dispatchRow(i1, row[bx], j1 - j0);
                if (isSpecialColumn(bx)) dispatchColumn(j1, col[by] + 1, i1 - i0);
            }
        }
        /* AbstractBlockAligner.cpp:310-315 */
        for (int bx = 0; bx < gw; bx++)
            for (int by = 0; by < gh; by++)
                dispatchScore(scores[bx][by], bx, by);
        /* AbstractBlockAligner.cpp:317-323 */
        if (mustDispatchLastCell()) {
            score_t s;
            s.score = row[gw - 1][grid->getBlockWidth(gw - 1, gh - 1) - 1].h;
            s.i = partition.getI1() - 1;
            s.j = partition.getJ1() - 1;
            dispatchScore(s, gw - 1, gh - 1);
        }
        for (int j = 0; j < gw; j++) delete[] row[j];
        delete[] row;
        for (int i = 0; i < gh; i++) delete[] col[i];
        delete[] col;
    }

private:
    /* AbstractBlockAligner.cpp:418-439 */
    bool isSpecialRow(int by) {
        if (mustDispatchLastRow() && by == getGrid()->getGridHeight() - 1) return true;
        if (mustDispatchSpecialRows()) {
            const int bh = getGrid()->getBlockHeight(0, 0);
            int interval = (getSpecialRowInterval() + bh - 1) / bh;
            if (interval <= 0) interval = 1;
            return ((by + 1) % interval == 0);
        }
        return false;
    }
    /* AbstractBlockAligner.cpp:447-449 */
    bool isSpecialColumn(int bx) {
        return mustDispatchLastColumn() && bx == getGrid()->getGridWidth() - 1;
    }

    int blockH, blockW;
    cell_t** row;
    cell_t** col;
    score_params_t score_params;
    BlockAlignerParameters* params;
    CPUBlockProcessor* processor;
    BlockPruningGenericN2* pruner;
    int totalBlocks, prunedBlocks;
};

static int parse_edge(char c) {
    switch (c) {
    case '*': return AT_ANYWHERE;
    case '1': return AT_SEQUENCE_1;
    case '2': return AT_SEQUENCE_2;
    case '3': return AT_SEQUENCE_1_OR_2;
    case '+': return AT_SEQUENCE_1_AND_2;
    }
    fprintf(stderr, "bad edge flag %c\n", c);
    exit(2);
}

/* parse_sequence_flags, libmasa.cpp: none | 1 | 2 | both */
static void parse_seq_flags(const char* s, bool* flags) {
    flags[0] = (!strcmp(s, "1") || !strcmp(s, "both"));
    flags[1] = (!strcmp(s, "2") || !strcmp(s, "both"));
}

static long long parse_size(const char* s) {
    char* end;
    double v = strtod(s, &end);
    if (*end == 'K') v *= 1024.0;
    else if (*end == 'M') v *= 1024.0 * 1024.0;
    else if (*end == 'G') v *= 1024.0 * 1024.0 * 1024.0;
    return (long long) v;
}

/*
 * usage: ref_driver [options] seq0.fasta seq1.fasta
 *   --work-dir=DIR --stage-1 --edges=XY --disk-size=N[KMG] --no-flush
 *   --no-block-pruning --block=H,W --split=COUNT --part=STEP
 *   --flush-column=URL --load-column=URL --max-alignments=N
 *   --dump-blocks --trim=I0,I1,J0,J1 --clear-n --reverse=1|2|both --complement=1|2|both --reverse-complement=1|2|both
 */
int main(int argc, char** argv) {
    std::string work = "./work.tmp";
    bool stage1_only = false;
    int astart = AT_ANYWHERE, aend = AT_ANYWHERE;
    long long disk = 0, ram = 0;
    bool pruning = true;
    int bh = 1024, bw = 1024;
    int split_count = 0, split_step = 0;
    int max_alignments = 1;
    std::string flush_url, load_url, shared_dir;
    bool gpu_stage4 = false;
    bool dump_blocks = false;
    bool do_fork = false;
    std::vector<int> fork_weights;
    std::vector<const char*> files;
    std::vector<char*> extension_args;
    int trim_start[2] = {0, 0}, trim_end[2] = {0, 0};
    bool clear_n = false, reverse_seq[2] = {false, false}, complement_seq[2] = {false, false};

    for (int a = 1; a < argc; a++) {
        const char* s = argv[a];
        if (!strncmp(s, "--work-dir=", 11)) work = s + 11;
        else if (!strcmp(s, "--stage-1")) stage1_only = true;
        else if (!strncmp(s, "--edges=", 8)) { astart = parse_edge(s[8]); aend = parse_edge(s[9]); }
        else if (!strncmp(s, "--disk-size=", 12)) { if (disk != -1) disk = parse_size(s + 12); }
        else if (!strncmp(s, "--ram-size=", 11)) { if (ram != -1) ram = parse_size(s + 11); }
        else if (!strcmp(s, "--no-flush")) { disk = -1; ram = -1; }
        else if (!strcmp(s, "--no-block-pruning")) pruning = false;
        else if (!strncmp(s, "--block=", 8)) sscanf(s + 8, "%d,%d", &bh, &bw);
        else if (!strncmp(s, "--split=", 8)) split_count = atoi(s + 8);
        else if (!strncmp(s, "--part=", 7)) split_step = atoi(s + 7);
        else if (!strncmp(s, "--flush-column=", 15)) { flush_url = s + 15; pruning = false; }
        else if (!strncmp(s, "--load-column=", 14)) { load_url = s + 14; pruning = false; }
        else if (!strncmp(s, "--max-alignments=", 17)) max_alignments = atoi(s + 17);
        /* sequence modifiers, libmasa.cpp:986-1050 */
        else if (!strncmp(s, "--trim=", 7)) sscanf(s + 7, "%d,%d,%d,%d", &trim_start[0], &trim_end[0], &trim_start[1], &trim_end[1]);
        else if (!strcmp(s, "--clear-n")) clear_n = true;
        else if (!strncmp(s, "--reverse=", 10)) parse_seq_flags(s + 10, reverse_seq);
        else if (!strncmp(s, "--complement=", 13)) parse_seq_flags(s + 13, complement_seq);
        else if (!strncmp(s, "--reverse-complement=", 21)) { parse_seq_flags(s + 21, complement_seq); reverse_seq[0] = complement_seq[0]; reverse_seq[1] = complement_seq[1]; }
        else if (!strncmp(s, "--shared-dir=", 13)) shared_dir = s + 13;        /* libmasa.cpp --shared-dir: AlignerPool's message directory (Job.cpp:156-158) */
        else if (!strcmp(s, "--dump-blocks")) dump_blocks = true;              /* libmasa.cpp:1082: best score of every block -> <work>/pruning_dump.txt */
        else if (!strcmp(s, "--gpu-stage4")) gpu_stage4 = true;               /* product stage 4 instead of MASA-Core's */
        else if (!strcmp(s, "--fork")) do_fork = true;                       /* weights from IAligner::getForkWeights */
        else if (!strncmp(s, "--fork=", 7)) {                                /* --fork=W1,W2,... (libmasa.cpp:964-980) */
            do_fork = true;
            for (const char* q = s + 7; *q; ) { fork_weights.push_back(atoi(q)); q = strchr(q, ','); if (!q) break; q++; }
        }
        else if (s[0] == '-') extension_args.push_back(argv[a]);    /* libmasa.cpp hands unknown options to the extension */
        else files.push_back(s);
    }
    if (files.size() != 2) { fprintf(stderr, "need two fasta files\n"); return 2; }

#ifdef USE_MI355_ALIGNER
    Mi355Aligner* aligner = new Mi355Aligner(-1, 0, 0);
    (void) bh; (void) bw;
    for (size_t k = 0; k < extension_args.size(); k++) {
        char* av[3] = {argv[0], extension_args[k], NULL};
        optind = 2;                       /* AbstractAlignerParameters::callGetOpt re-reads argv[optind - 1] */
        if (aligner->getParameters()->processArgument(2, av) != 0) {
            fprintf(stderr, "option %s: %s\n", extension_args[k], aligner->getParameters()->getLastError());
            return 2;
        }
    }
#else
    SerialBlockAligner* aligner = new SerialBlockAligner(bh, bw);
    if (!extension_args.empty()) { fprintf(stderr, "unknown option %s\n", extension_args[0]); return 2; }
#endif

    /* libmasa.cpp:765-806 defaults */
    Job* job = new Job(2);
    job->configs = NULL;
    AlignmentParams* ap = job->getAlignmentParams();
    ap->setAlignmentMethod(ALIGNMENT_METHOD_LOCAL);
    const score_params_t* sp = aligner->getScoreParameters();
    ap->setAffineGapPenalties(-sp->gap_open, -sp->gap_ext);
    ap->setMatchMismatchScores(sp->match, sp->mismatch);
    job->disk_limit = disk;
    job->ram_limit = ram;
    job->block_pruning = pruning;
    job->dump_blocks = dump_blocks;
    job->setWorkPath(work);
    if (!shared_dir.empty()) job->setSharedPath(shared_dir);
    job->stage4_maximum_partition_size = 16;
    job->stage4_strategy = STAGE_4_STRATEGY_OPTIMIZED;
    job->stage6_output_format = 0;
    job->flush_column_url = flush_url;
    job->load_column_url = load_url;
    job->alignment_start = astart;
    job->alignment_end = aend;
    job->max_alignments = max_alignments;
    job->peer_listen_port = -1;
    job->predicted_traceback = false;
    job->setBufferLimit(1024 * 1024);
    job->aligner = aligner;

    /* libmasa.cpp:1269-1288 */
    for (int i = 0; i < 2; i++) {
        SequenceInfo* info = new SequenceInfo();
        info->setFilename(files[i]);
        SequenceModifiers* mod = new SequenceModifiers();
        mod->setClearN(clear_n);
        mod->setReverse(reverse_seq[i]);
        mod->setComplement(complement_seq[i]);
        mod->setTrimStart(trim_start[i]);
        mod->setTrimEnd(trim_end[i]);
        Sequence* seq = new Sequence(info, mod);
        job->addSequence(seq);
        ap->addSequence(seq);
    }

    /* libmasa.cpp:497-535 split_sequences with equal weights */
    if (split_count > 0) {
        int seq1_len = ap->getSequence(1)->getLen();
        int trim_j0 = (int) ((((long long) seq1_len) * (split_step - 1)) / split_count + 1);
        int trim_j1 = (int) ((((long long) seq1_len) * split_step) / split_count);
        char str[256];
        if (split_step > 1 && job->load_column_url == "") {
            sprintf(str, "file://%s/../STEP-%d-%d-%d.tmp", work.c_str(), split_step - 1, split_count, trim_j0 - 1);
            job->load_column_url = str;
        }
        if (split_step < split_count && job->flush_column_url == "") {
            sprintf(str, "file://%s/../STEP-%d-%d-%d.tmp", work.c_str(), split_step, split_count, trim_j1);
            job->flush_column_url = str;
        }
        ap->getSequence(1)->trim(trim_j0, trim_j1);
        job->block_pruning = false;
    }

    /* libmasa.cpp:1305-1325 + fork_multi_process (:540-642): one child per positive weight, chained through
     * socket://127.0.0.1:7000+id, seq1 trimmed in proportion to the weights, work dir FORK.NN (Job.cpp:127) */
    if (do_fork) {
        if (fork_weights.empty()) {
            const int* w = aligner->getForkWeights();
            for (int k = 0; w && w[k] != 0; k++) fork_weights.push_back(w[k]);
        }
        const int count = (int) fork_weights.size();
        if (count == 0) { fprintf(stderr, "No forked instances allowed.\n"); return 1; }
        job->block_pruning = false;
        std::vector<long long> prop(count + 1, 0);
        std::vector<int> prev(count + 1, -1);
        int firstId = -1, lastId = -1;
        for (int i = 0; i < count; i++) {
            prop[i + 1] = prop[i] + fork_weights[i];
            if (fork_weights[i] > 0) { if (firstId < 0) firstId = i; prev[i] = lastId; lastId = i; }
        }
        for (int i = 0; i < count; i++)
            printf("fork[%d%c]: %.2f%%\n", i, (i >= firstId && i <= lastId) ? '+' : ' ', (prop[i + 1] - prop[i]) * 100.0 / prop[count]);
        if (firstId < 0) { fprintf(stderr, "No forked instances with valid weight.\n"); return 1; }
        fflush(stdout);
        bool parent = true;
        int me = -1;
        for (int i = 0; i < count && parent; i++) {
            if (fork_weights[i] <= 0) continue;
            const pid_t pid = fork();
            if (pid == 0) {
                parent = false; me = i;
                aligner->getParameters()->setForkId(i);
                char str[128];
                if (i > firstId) { sprintf(str, "socket://127.0.0.1:%d", 7000 + prev[i]); job->load_column_url = str; }
                if (i < lastId) { sprintf(str, "socket://127.0.0.1:%d", 7000 + i); job->flush_column_url = str; }
            }
        }
        if (parent) {
            int status = 0, bad = 0;
            while (wait(&status) > 0) if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) bad = 1;
            return bad;
        }
        const int seq1_len = ap->getSequence(1)->getLen();
        const int trim_j0 = (int) (((long long) seq1_len * prop[me]) / prop[count] + 1);
        const int trim_j1 = (int) (((long long) seq1_len * prop[me + 1]) / prop[count]);
        ap->getSequence(1)->trim(trim_j0, trim_j1);
    }

    if (!job->initialize()) { fprintf(stderr, "job init failed\n"); return 1; }

    /* libmasa.cpp:1349-1385 */
    int count = stage1(job);
    if (!stage1_only) {
        for (int id = 0; id < count; id++) {
            stage2(job, id);
            stage3(job, id);
#ifdef USE_MI355_ALIGNER
            if (gpu_stage4) {
                /* what stage4() does (sw_stage4.cpp:880-960), with the refinement itself on the GPU */
                Timer t4; int ev = t4.createEvent("GPU_STAGE4"); t4.init();
                CrosspointsFile in3(job->getCrosspointFile(STAGE_3, id));
                in3.loadCrosspoints();
                std::vector<int> tijs;
                for (size_t k = 0; k < in3.size(); k++) { tijs.push_back(in3[k].type); tijs.push_back(in3[k].i); tijs.push_back(in3[k].j); tijs.push_back(in3[k].score); }
                Sequence* q0 = job->getAlignmentParams()->getSequence(0);
                Sequence* q1 = job->getAlignmentParams()->getSequence(1);
                int* out = NULL; int n = 0; double kms = 0;
                int steps = aligner->refineCrosspoints(q0->getData(false), q1->getData(false), q0->getInfo()->getSize(), q1->getInfo()->getSize(),
                                                       tijs.data(), (int) in3.size(), job->stage4_maximum_partition_size, &out, &n, &kms);
                CrosspointsFile out4(job->getCrosspointFile(STAGE_4, id));
                for (int k = 0; k < n; k++) { crosspoint_t c; c.type = out[4 * k]; c.i = out[4 * k + 1]; c.j = out[4 * k + 2]; c.score = out[4 * k + 3]; out4.push_back(c); }
                out4.save();
                mi355sw_free(out);
                float ms = t4.eventRecord(ev);
                FILE* st4 = job->fopenStatistics(STAGE_4, id);
                fprintf(st4, "GPU STAGE 4 (mi355sw_stage4): steps %d  crosspoints %d -> %d  kernels %.3f ms  total %.3f ms\n", steps, (int) in3.size(), n, kms, ms);
                fclose(st4);
            } else
#endif
            stage4(job, id);
            stage5(job, id);
            stage6(job, id);
        }
    }
    aligner->finalize();
    return 0;
}
