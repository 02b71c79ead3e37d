// See Mi355AlignerParameters.hpp.
#include "Mi355AlignerParameters.hpp"

#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>

#include "mi355sw.h"

#define USAGE "\
--gpu=GPU               Selects the index of the GPU used for the computation. If  \n\
                           GPU is not informed, the fastest GPU is selected.   \n\
                           A list of available GPUs can be obtained with the   \n\
                           --list-gpus parameter. \n\
--list-gpus             Lists all available GPUs. \n\
--blocks=B              Run B strip wavefronts (default: one per SIMD of the GPU)\n\
--strip-rows=R          Rows of one strip: 256, 512, 768, 1024, 1536 or 2048\n\
                           (default: chosen per partition by the engine's cost model)\n\
"

#define ARG_GPU        0x1001
#define ARG_LIST_GPUS  0x1002
#define ARG_BLOCKS     0x1003
#define ARG_STRIP_ROWS 0x1004

static struct option long_options[] = {
    {"gpu",        required_argument, 0, ARG_GPU},
    {"list-gpus",  no_argument,       0, ARG_LIST_GPUS},
    {"blocks",     required_argument, 0, ARG_BLOCKS},
    {"strip-rows", required_argument, 0, ARG_STRIP_ROWS},
    {0, 0, 0, 0}
};

Mi355AlignerParameters::Mi355AlignerParameters() : gpu(MI355_DETECT_FASTEST_GPU), waves(0), stripRows(0) {}
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

int Mi355AlignerParameters::processArgument(int argc, char** argv) {
    const int ret = AbstractAlignerParameters::callGetOpt(argc, argv, long_options);
    switch (ret) {
    case ARG_GPU:
        if (optarg != NULL) sscanf(optarg, "%d", &gpu);
        if (gpu < 0 || gpu >= mi355sw_device_count()) {
            setLastError("GPU index out of range (see --list-gpus).");
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
    default:
        return ret;
    }
    return 0;
}
