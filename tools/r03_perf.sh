#!/bin/bash
# round 3 kernel iteration: parity suite (packed kernels), C2 bench, pruned related pair, tall shape
tag=${1:-x}
mkdir -p gpurun_out/r03
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_overflow.py tests/test_gpu_sra.py -q -x > gpurun_out/r03/perf_${tag}_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r03/perf_${tag}_tests.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-target-shape --no-cpu-baseline > gpurun_out/r03/perf_${tag}_bench_c2.json 2> gpurun_out/r03/perf_${tag}_bench_c2.err
echo "bench rc=$?"; python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r03/perf_${tag}_bench_c2.json').read().strip().splitlines()[-1])
print('C2 GCUPS %.1f ms/step %.1f kernel_ms %.1f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
timeout 200 python tools/prune_probe.py 4000000 3000000 2>&1 | tail -2
timeout 200 python tools/gpu_perf.py 16777216,4000000 4000000,3000000,0,0,0,1,2,0 2>&1 | tail -3
