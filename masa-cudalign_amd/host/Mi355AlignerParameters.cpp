// See Mi355AlignerParameters.hpp.
#include "Mi355AlignerParameters.hpp"

#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <poll.h>
#include <signal.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "mi355sw.h"

#define USAGE "\
--gpu=INDEX             Run on the MI355X with this HIP device index (see --list-gpus).\n\
                           Without it the device with the most compute units x clock is\n\
                           taken; a forked instance takes the device of its fork id.\n\
--list-gpus             Print the gfx950 devices this extension can use, then exit.\n\
--blocks=B              Keep B strip wavefronts resident (default: one per SIMD).\n\
--strip-rows=R          Height of a strip in DP rows: 256, 512, 768, 1024, 1536 or 2048\n\
                           (default: picked per partition by the engine's cost model).\n\
                           Special rows fall on multiples of the strip height; use 1024 or\n\
                           2048 to share a special-rows area with CUDAlign (8192 spacing).\n\
--block-columns=W       Also report the best score of every block of the grid\n\
                           (strip x W columns), as MASA-Core's --dump-blocks wants them;\n\
                           needs --strip-rows=256, 512 or 1024 and costs a second sweep.\n\
--prune-global          Block pruning for GLOBAL alignments as well (--alignment-edges=++):\n\
                           MASA-Core's stage 1 only asks for pruning when the alignment may end\n\
                           anywhere; with this option the engine also prunes whenever the score is\n\
                           read from the last cell, against a lower bound of that cell.\n\
--no-diagonal-seed      Pruning runs of large matrices (8 Mi x 8 Mi and more) first sweep a narrow band\n\
                           along the diagonal and start their pruning bound from the score found there;\n\
                           this option leaves that pass out (use it with --max-alignments > 1: a strong\n\
                           first bound prunes the weaker alignments away sooner).\n\
--engine-flags=N        MI355SW_F_* switches of the engine (include/mi355sw.h), OR-ed together, decimal or 0x...:\n\
                           32 two-phase best at every size, 1024 no pruning window, 2048 staircase seed,\n\
                           4096 gap-initialised first columns made on the device, ... (the engine reads no\n\
                           environment variable: this is where its switches come from).\n\
--engine-verbosity=N    Diagnostics of the engine on stderr (MI355SW_V_*): 1 messages, 2 one line per\n\
                           partition with its timing, 4 the seed's anchors and segments.\n\
"

#define ARG_GPU        0x1001
#define ARG_LIST_GPUS  0x1002
#define ARG_BLOCKS     0x1003
#define ARG_STRIP_ROWS 0x1004
#define ARG_BLOCK_COLUMNS 0x1005
#define ARG_PRUNE_GLOBAL 0x1006
#define ARG_NO_DIAGONAL_SEED 0x1007
#define ARG_ENGINE_FLAGS 0x1008
#define ARG_ENGINE_VERBOSITY 0x1009

static struct option long_options[] = {
    {"gpu",        required_argument, 0, ARG_GPU},
    {"list-gpus",  no_argument,       0, ARG_LIST_GPUS},
    {"blocks",     required_argument, 0, ARG_BLOCKS},
    {"strip-rows", required_argument, 0, ARG_STRIP_ROWS},
    {"block-columns", required_argument, 0, ARG_BLOCK_COLUMNS},
    {"prune-global", no_argument, 0, ARG_PRUNE_GLOBAL},
    {"no-diagonal-seed", no_argument, 0, ARG_NO_DIAGONAL_SEED},
    {"engine-flags", required_argument, 0, ARG_ENGINE_FLAGS},
    {"engine-verbosity", required_argument, 0, ARG_ENGINE_VERBOSITY},
    {0, 0, 0, 0}
};

Mi355AlignerParameters::Mi355AlignerParameters() : gpu(MI355_DETECT_FASTEST_GPU), waves(0), stripRows(0), blockColumns(0), pruneGlobal(0), noDiagonalSeed(0),
    engineFlags(0), engineVerbosity(0) {}
Mi355AlignerParameters::~Mi355AlignerParameters() {}

void Mi355AlignerParameters::printUsage() const {
    AbstractAlignerParameters::printFormattedUsage("MI355X Specific Options", USAGE);
}

void Mi355AlignerParameters::printGPUDevices(FILE* file) {
    const int n = mi355sw_device_count();
    fprintf(file, "Available GPUs: %d\n", n);
    for (int d = 0; d < n; d++) {
        char name[128]; int32_t cus = 0, mhz = 0; int64_t bytes = 0;
        if (mi355sw_device_info(d, name, sizeof(name), &cus, &mhz, &bytes) == MI355SW_OK)
            fprintf(file, "  %d: %s, %d CUs, %d MHz, %.1f GB%s\n", d, name, cus, mhz, bytes / 1e9, d == fastestGPU() ? "  [fastest]" : "");
    }
}

int Mi355AlignerParameters::fastestGPU() {
    const int n = mi355sw_device_count();
    int best = 0; long long bw = -1;
    for (int d = 0; d < n; d++) {
        char name[8]; int32_t cus = 0, mhz = 0; int64_t bytes = 0;
        if (mi355sw_device_info(d, name, sizeof(name), &cus, &mhz, &bytes) != MI355SW_OK) continue;
        const long long w = (long long) cus * mhz;
        if (w > bw) { bw = w; best = d; }
    }
    return best;
}

// Weight of every usable GPU (compute units x clock in MHz), for MASA-Core's --fork split of seq1
// (AbstractAligner::setForkCount; the reference: X/CUDAligner.cpp:63-66, X/cuda_util.cpp:191-257).  The aligner
// object is constructed BEFORE MASA-Core forks its per-GPU children, and a HIP runtime initialised in the parent
// cannot be used in a child: the devices are therefore enumerated in a throw-away child process that reports
// through a pipe and exits with its runtime.
int Mi355AlignerParameters::deviceWeights(int* weights, int max) {
    int fd[2];
    if (pipe(fd) != 0) return 0;
    const pid_t pid = fork();
    if (pid < 0) { close(fd[0]); close(fd[1]); return 0; }
    if (pid == 0) {
        close(fd[0]);
        const int n = mi355sw_device_count();
        for (int d = 0; d < n; d++) {
            char name[8]; int32_t cus = 0, mhz = 0; int64_t bytes = 0;
            int w = 0;
            if (mi355sw_device_info(d, name, sizeof(name), &cus, &mhz, &bytes) == MI355SW_OK) w = cus * (mhz > 0 ? mhz : 1);
            if (write(fd[1], &w, sizeof(w)) != (ssize_t) sizeof(w)) _exit(1);
        }
        close(fd[1]);
        _exit(0);
    }
    close(fd[1]);
    // bounded: a child that inherited a HIP runtime already initialised in this process (a second aligner, a library
    // user) may never answer -- ten seconds, then it is killed and the caller falls back to one instance
    int count = 0, w = 0;
    bool timed_out = false;
    for (;;) {
        struct pollfd pfd = {fd[0], POLLIN, 0};
        const int pr = poll(&pfd, 1, 10000);
        if (pr <= 0) { timed_out = true; break; }
        if (read(fd[0], &w, sizeof(w)) != (ssize_t) sizeof(w)) break;
        if (w > 0 && count < max) weights[count++] = w;        // a device that could not be queried gets no instance
    }
    close(fd[0]);
    if (timed_out) { kill(pid, SIGKILL); count = 0; }
    int status = 0;
    waitpid(pid, &status, 0);
    return count;
}

int Mi355AlignerParameters::processArgument(int argc, char** argv) {
    const int ret = AbstractAlignerParameters::callGetOpt(argc, argv, long_options);
    switch (ret) {
    case ARG_GPU:
        // Only parsed here.  The range is checked in Mi355Aligner::initialize(), i.e. after MASA-Core has forked its
        // --fork / --split children (libmasa.cpp:1204, :581): asking the HIP runtime for the device count here
        // would initialise it in the parent, and a runtime inherited across fork() is unusable in the children
        // (the reference only runs sscanf at this point too, X/CUDAlignerParameters.cpp:84-88).
        if (optarg == NULL || sscanf(optarg, "%d", &gpu) != 1 || gpu < 0) {
            setLastError("--gpu needs a non-negative device index (see --list-gpus).");
            return -1;
        }
        break;
    case ARG_LIST_GPUS:
        printGPUDevices(stdout);
        exit(1);
        break;
    case ARG_BLOCKS:
        if (optarg != NULL) sscanf(optarg, "%d", &waves);
        if (waves < 0 || waves > MI355_MAX_WAVES) {
            setLastError("Blocks count cannot be greater than 4096.");
            return -1;
        }
        break;
    case ARG_STRIP_ROWS:
        if (optarg != NULL) sscanf(optarg, "%d", &stripRows);
        if (stripRows != 256 && stripRows != 512 && stripRows != 768 && stripRows != 1024 && stripRows != 1536 && stripRows != 2048) {
            setLastError("Strip rows must be one of 256, 512, 768, 1024, 1536, 2048.");
            return -1;
        }
        break;
    case ARG_BLOCK_COLUMNS:
        if (optarg == NULL || sscanf(optarg, "%d", &blockColumns) != 1 || blockColumns < 1) {
            setLastError("--block-columns needs a positive number of columns.");
            return -1;
        }
        break;
    case ARG_PRUNE_GLOBAL:
        pruneGlobal = 1;
        break;
    case ARG_NO_DIAGONAL_SEED:
        noDiagonalSeed = 1;
        break;
    case ARG_ENGINE_FLAGS:              // the library reads no environment variable (ABI 7): its switches come from here
        engineFlags = (int) strtol(optarg, NULL, 0);
        break;
    case ARG_ENGINE_VERBOSITY:
        engineVerbosity = (int) strtol(optarg, NULL, 0);
        break;
    default:
        return ret;
    }
    return 0;
}
