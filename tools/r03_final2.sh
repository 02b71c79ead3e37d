#!/bin/bash
# round 3, closing call: the whole GPU suite, the multi-rank rehearsals, the default bench line, the native pipeline at 10 M and C3
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03/final2_gpu_suite.log 2>&1
echo "gpu suite rc=$?"; tail -3 gpurun_out/r03/final2_gpu_suite.log
bash tools/r03_rehearse.sh 2>&1 | tail -30
timeout 600 python bench.py > gpurun_out/r03/final2_bench_default.json 2> gpurun_out/r03/final2_bench_default.err
echo "bench rc=$?"; cut -c1-300 gpurun_out/r03/final2_bench_default.json
timeout 200 python tools/native_pipeline_run.py 10000000 10000000 - gpurun_out/r03/native_10M_final.json 2>&1 | tail -1 | cut -c1-700
timeout 900 python tools/native_pipeline_run.py 48000000 46000000 25769803776 gpurun_out/r03/native_c3_final.json 5 2>&1 | tail -1 | cut -c1-900
