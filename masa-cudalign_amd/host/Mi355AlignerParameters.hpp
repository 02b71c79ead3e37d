// Command-line parameters of the MI355X extension: the counterpart of X/CUDAlignerParameters.{hpp,cpp}
// (--gpu, --list-gpus, --blocks) for an engine whose unit of parallelism is a strip wavefront.
// MASA-Core hands every option it does not know to IAlignerParameters::processArgument
// (M/libmasa/IAlignerParameter.hpp, M/libmasa/parameters/AbstractAlignerParameters.cpp:58-67).
#ifndef MI355ALIGNERPARAMETERS_HPP_
#define MI355ALIGNERPARAMETERS_HPP_

#include "libmasa/parameters/AbstractAlignerParameters.hpp"

#define MI355_DETECT_FASTEST_GPU (-1)     /* X/CUDAlignerParameters.hpp: DETECT_FASTEST_GPU */
#define MI355_MAX_WAVES 4096

class Mi355AlignerParameters : public AbstractAlignerParameters {
public:
    Mi355AlignerParameters();
    virtual ~Mi355AlignerParameters();
    virtual void printUsage() const;
    virtual int processArgument(int argc, char** argv);

    int getGPU() const { return gpu; }
    void setGPU(int g) { gpu = g; }
    int getWaves() const { return waves; }            /* --blocks: strip wavefronts per launch, 0 = one per SIMD */
    int getStripRows() const { return stripRows; }    /* --strip-rows: 256..2048, 0 = cost model */
    int getBlockColumns() const { return blockColumns; }   /* --block-columns: width of the blocks whose scores are dispatched, 0 = none */
    int getPruneGlobal() const { return pruneGlobal; }     /* --prune-global: block pruning of partitions whose goal is the last cell */
    int getNoDiagonalSeed() const { return noDiagonalSeed; }   /* --no-diagonal-seed: MI355SW_F_NO_DIAGONAL_SEED */
    int getEngineFlags() const { return engineFlags; }         /* --engine-flags=N: MI355SW_F_* bits OR-ed into mi355sw_config.flags (0x... accepted) */
    int getEngineVerbosity() const { return engineVerbosity; } /* --engine-verbosity=N: MI355SW_V_* bits (1 = messages, 2 = one line per partition, ...) */
    static void printGPUDevices(FILE* file);          /* --list-gpus (X/cuda_util.cpp:191-230) */
    static int fastestGPU();                          /* X/cuda_util.cpp:238-287: largest CUs x clock */
    static int deviceWeights(int* weights, int max);  /* X/cuda_util.cpp:191-257: per-GPU weights, asked from a child process */

private:
    int gpu, waves, stripRows, blockColumns, pruneGlobal, noDiagonalSeed, engineFlags, engineVerbosity;
};

#endif
