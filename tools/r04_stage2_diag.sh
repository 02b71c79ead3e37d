#!/bin/bash
# round 4: where stage 2's time goes at C3 -- the hop of a stage-2-shaped sweep from the kernel's trace, then one pipeline
# run that leaves its work directory and stage 2 alone, twice, with the per-call timing lines of the engine
mkdir -p gpurun_out/r04
echo "== trace of a stage-2-shaped call (R = 8, then 4)"
MI355SW_TRACE=/tmp/trace8.bin timeout 120 python tools/stage2_probe.py 46000000 700000 700000 9786 8 2>&1 | tail -1
python tools/trace_hops.py /tmp/trace8.bin 1500
MI355SW_TRACE=/tmp/trace4.bin timeout 120 python tools/stage2_probe.py 46000000 700000 700000 9786 4 2>&1 | tail -1
python tools/trace_hops.py /tmp/trace4.bin 3000
echo "== trace of a local score pass, 3 M x 700 k, R = 8 and 24"
MI355SW_TRACE=/tmp/trace_sw8.bin timeout 120 python tools/gpu_perf.py 3000000,700000,8,0,0,0,1 2>&1 | tail -1
python tools/trace_hops.py /tmp/trace_sw8.bin 1500
MI355SW_TRACE=/tmp/trace_sw24.bin timeout 120 python tools/gpu_perf.py 3000000,700000,24,0,0,0,1 2>&1 | tail -1
python tools/trace_hops.py /tmp/trace_sw24.bin 1500
export MI355SW_WORK=/tmp/c3work
mkdir -p $MI355SW_WORK
ARGS="48000000 46000000 25769803776"
timeout 600 python tools/native_pipeline_run.py $ARGS gpurun_out/r04/c3_diag.json 5 > gpurun_out/r04/c3_diag.log 2>&1
python -c "import json;d=json.load(open('gpurun_out/r04/c3_diag.json'));print('pipeline',d['seconds'])"
for k in 1 2; do
  MI355SW_VERBOSE_JOBS=1 timeout 300 python tools/stage2_rerun.py $ARGS 5 1 > gpurun_out/r04/stage2_jobs_run$k.log 2>&1; grep "stage 2, run" gpurun_out/r04/stage2_jobs_run$k.log
done
MI355SW_SRA_SYNC=1 MI355SW_VERBOSE_JOBS=1 timeout 300 python tools/stage2_rerun.py $ARGS 5 1 > gpurun_out/r04/stage2_jobs_sync.log 2>&1; grep "stage 2, run" gpurun_out/r04/stage2_jobs_sync.log
rm -rf $MI355SW_WORK
