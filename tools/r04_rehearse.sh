#!/bin/bash
# Round 4: N > 1 path of bench.py rehearsed on the ONE GPU (all ranks on cuda:0, gloo collectives), local and global
# (--nw) with block pruning over the chain; numbers mean nothing, the result line's fields and the checks do
mkdir -p gpurun_out/r04
run() {  # n tag extra...
    n=$1; tag=$2; shift 2
    MI355SW_BENCH_REHEARSAL=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus $n --steps 2 --warmup 1 --size 300000 "$@" > gpurun_out/r04/rehearsal_$tag.log 2>&1
    echo "rehearsal $tag rc=$?"
    grep '^{' gpurun_out/r04/rehearsal_$tag.log | tail -1 > gpurun_out/r04/rehearsal_$tag.json
    python3 -c "
import json
d=json.loads(open('gpurun_out/r04/rehearsal_$tag.json').read())
c=d['config']
print('value %.0f comm %s xgmi %s pruned %.3f best %s kernel %s single %s' % (d['value'], c['comm'], c['xgmi'], c['pruned_fraction'], d['best'], c['kernel_name'], c['same_shape_single_gpu'] and round(c['same_shape_single_gpu']['gcups'])))
for r in c['ranks']: print('  ', r['rank'], r['device'], r['pci_bus_id'], r['band_columns'], round(r['kernel_ms'],1), round(r['wait_for_left_neighbour_ms'],1), r['pruned_cells'], r['restarts'])
"
}
run 2 n2
run 4 n4_related --related
run 4 n4_nw --nw
run 4 n4_nw_related --nw --related
run 8 n8_nw_related --nw --related
MI355SW_BENCH_COMM=host run 4 n4_nw_related_host --nw --related
# the fall-back transports: as if the hipIpc check had failed (p2p-attach), and with the in-process chain refused too (host)
MI355SW_BENCH_FAIL_IPC=1 run 4 n4_attach
MI355SW_BENCH_FAIL_IPC=1 run 4 n4_attach_nw_related --nw --related
MI355SW_BENCH_FAIL_IPC=1 MI355SW_BENCH_NO_ATTACH=1 run 4 n4_host_after_both_failed --related
