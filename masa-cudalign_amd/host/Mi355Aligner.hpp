// MASA extension adapter: an IAligner (M/libmasa/IAligner.hpp:149-387) that forwards every call to
// the C ABI of include/mi355sw.h.  This is the file a MASA-CUDAlign maintainer adds next to
// X/CUDAligner.{hpp,cpp}; X/main.cpp:39-41 then reads
//     return libmasa_entry_point(argc, argv, new Mi355Aligner(), header);
// It needs MASA-Core's headers, so it is compiled only where the reference tree is present
// (oracle/build_ref.sh links it into oracle/_ref/masa_mi355 for the end-to-end drop-in test).
#ifndef MI355ALIGNER_HPP_
#define MI355ALIGNER_HPP_

#include "libmasa/libmasa.hpp"
#include "libmasa/aligners/AbstractAligner.hpp"
#include "Mi355AlignerParameters.hpp"

#include "mi355sw.h"

class Mi355Aligner : public AbstractAligner {
public:
    Mi355Aligner(int device = -1, int rowsPerLane = 0, int waves = 0);
    virtual ~Mi355Aligner();

    /* IAligner */
    virtual aligner_capabilities_t getCapabilities();
    virtual const int* getForkWeights();          /* enumerates the GPUs on first use (a child process, bounded in time) */
    virtual const score_params_t* getScoreParameters();
    virtual IAlignerParameters* getParameters();
    virtual void initialize();
    virtual void finalize();
    virtual void setSequences(const char* seq0, const char* seq1, int seq0_len, int seq1_len);
    virtual void unsetSequences();
    virtual void alignPartition(Partition partition);
    virtual void clearStatistics();
    virtual void printInitialStatistics(FILE* file);
    virtual void printStageStatistics(FILE* file);
    virtual void printFinalStatistics(FILE* file);
    virtual void printStatistics(FILE* file);
    virtual const char* getProgressString() const;
    virtual long long getProcessedCells();

    /* Stage 4 on the GPU: what MASA-Core's stage4() (M/stage4/sw_stage4.cpp:880-960) does with the crosspoints of
     * crosspoint_03.NN, as one call.  `in`/`out` use the core's crosspoint_t (M/common/Crosspoint.hpp); the sequences
     * must be the ones of the job (setSequences is called here).  A maintainer's stage driver calls this instead of
     * stage4() when the aligner is a Mi355Aligner; oracle/ref_driver.cpp --gpu-stage4 shows the four lines. */
    int refineCrosspoints(const char* seq0, const char* seq1, int seq0_len, int seq1_len, const int* in_tijs, int count,
                          int max_partition_size, int** out_tijs, int* out_count, double* kernel_ms);

private:
    void check(int rc, const char* what);
    /* IManager trampolines (M/libmasa/IManager.hpp:98-313) */
    static int32_t cbRecurrence(void* u);
    static int32_t cbSpecialInterval(void* u);
    static int32_t cbFirstColumnType(void* u);
    static int32_t cbFirstRowType(void* u);
    static void cbSuperPartition(void* u, mi355sw_partition* out);
    static void cbReceiveFirstRow(void* u, mi355sw_cell* b, int32_t len);
    static void cbReceiveFirstColumn(void* u, mi355sw_cell* b, int32_t len);
    static void cbDispatchColumn(void* u, int32_t j, const mi355sw_cell* b, int32_t len);
    static void cbDispatchRow(void* u, int32_t i, const mi355sw_cell* b, int32_t len);
    static void cbDispatchScore(void* u, mi355sw_score s, int32_t bx, int32_t by);
    static int32_t cbMustContinue(void* u);
    static int32_t cbLastCell(void* u);
    static int32_t cbLastRow(void* u);
    static int32_t cbLastColumn(void* u);
    static int32_t cbSpecialRows(void* u);
    static int32_t cbScores(void* u);
    static int32_t cbPrune(void* u);

    mi355sw_handle* handle;
    mi355sw_config config;
    score_params_t score_params;
    Mi355AlignerParameters* params;
    bool weightsKnown;
    mutable char progress[256];
    long long statCells;
    long long statPruned;
    double statKernelMs;
    int statPartitions;
};

#endif
