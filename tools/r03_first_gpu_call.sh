#!/bin/bash
# The GPU call round 2 could not make (its GPU minutes were spent when the native stages 2-3 were written):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/r03_first_gpu_call.sh'
# 1. the native pipeline's GPU cases, failures counting (MI355SW_NATIVE_PIPELINE=1);
# 2. seconds per stage natively at 10 M x 10 M (MASA-Core's own stages on the engine: 21 s, profiles/r02_dropin_pipeline_10Mx10M_gpu_stage4.json);
# 3. the N > 1 start-up of bench.py rehearsed on the one GPU (ports opened AND verified: bands.verify_p2p).
# Everything lands in gpurun_out/r03/; copy what is worth judging into profiles/.
mkdir -p gpurun_out/r03
export MI355SW_NATIVE_PIPELINE=1
timeout 900 python -m pytest tests/test_gpu_zz_native_pipeline.py -q > gpurun_out/r03/native_pipeline_tests.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r03/native_pipeline_tests.log
tail -5 gpurun_out/r03/native_pipeline_tests.log
timeout 700 python tools/native_pipeline_run.py 10000000 10000000 - gpurun_out/r03/native_pipeline_10Mx10M.json > gpurun_out/r03/native_pipeline_10Mx10M.log 2>&1
echo "native 10Mx10M rc=$?"
tail -2 gpurun_out/r03/native_pipeline_10Mx10M.log
MI355SW_BENCH_REHEARSAL=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 1 --warmup 0 --tall 1 > gpurun_out/r03/bench_rehearsal_n2_verify.log 2>&1
echo "rehearsal rc=$?"
tail -2 gpurun_out/r03/bench_rehearsal_n2_verify.log
