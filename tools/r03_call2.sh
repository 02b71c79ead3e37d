#!/bin/bash
# round 3, GPU call 2: the new chain-pruning tests, the headline kernel after the KernelArgs change, the pruning kernel
# after the chain-best code, and the N = 2 rehearsal that hung in call 1 (stacks of both ranks every 40 s)
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests/test_gpu_bands.py -q -x -s > gpurun_out/r03/call2_bands_tests.log 2>&1
echo "bands tests rc=$?"; tail -5 gpurun_out/r03/call2_bands_tests.log
timeout 300 python bench.py --steps 3 --warmup 1 --no-target-shape --no-cpu-baseline > gpurun_out/r03/call2_bench_c2.json 2> gpurun_out/r03/call2_bench_c2.err
echo "bench rc=$?"; cut -c1-400 gpurun_out/r03/call2_bench_c2.json
timeout 200 python tools/prune_probe.py 4000000 3000000 > gpurun_out/r03/call2_prune_probe.log 2>&1
echo "prune rc=$?"; cat gpurun_out/r03/call2_prune_probe.log | tail -4
MI355SW_BENCH_STACKS=40 MI355SW_BAND_DEBUG=1 MI355SW_DEBUG=1 MI355SW_BENCH_REHEARSAL=1 timeout 130 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
    bench.py --gpus 2 --steps 1 --warmup 0 --tall 1 > gpurun_out/r03/call2_rehearsal_n2.log 2>&1
echo "rehearsal rc=$?"; grep -v "^\[band" gpurun_out/r03/call2_rehearsal_n2.log | tail -5 | cut -c1-300
