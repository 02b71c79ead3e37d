#!/bin/bash
# round 3, after a device-code change: the whole GPU suite, smoke(), the PMC passes for the new build id, the default bench line
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03/final_gpu_suite.log 2>&1
echo "gpu suite rc=$?"; tail -3 gpurun_out/r03/final_gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
bash tools/pmc_collect.sh r03 > gpurun_out/r03/final_pmc_collect.log 2>&1
echo "pmc rc=$?"; ls gpurun_out/prof_r03
timeout 600 python bench.py > gpurun_out/r03/final_bench_default.json 2> gpurun_out/r03/final_bench_default.err
echo "bench rc=$?"; cut -c1-700 gpurun_out/r03/final_bench_default.json
