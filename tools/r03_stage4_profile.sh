#!/bin/bash
# run ON THE GPU BOX: rocprofv3 kernel statistics and SQ counters of the whole native pipeline at 10 M x 10 M --
# what the stage-4 kernels (mm_half_kernel, mm_match_kernel) and the stage-2/3 strip launches cost next to stage 1.
# Summaries land in gpurun_out/r03/; copy them into profiles/.
set -x
out=gpurun_out/r03
mkdir -p $out
export TMPDIR=/tmp
CMD="python3 tools/native_pipeline_run.py 10000000 10000000"
rocprofv3 --kernel-trace --stats -d $out/p_stats -- $CMD > $out/pipeline_stats.log 2>&1
python3 tools/rocpd_summary.py stats $(find $out/p_stats -name '*.db' | head -1) $out/r03_native_pipeline_10M_kernel_stats.csv $out/r03_native_pipeline_10M_dispatches.csv > /dev/null
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $out/p_sq -- $CMD > $out/pipeline_sq.log 2>&1
python3 tools/rocpd_summary.py pmc $out/r03_native_pipeline_10M_sq_pmc.json "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -- $CMD" $(find $out/p_sq -name '*.db' | head -1) > /dev/null
rm -rf $out/p_stats $out/p_sq
head -12 $out/r03_native_pipeline_10M_kernel_stats.csv
