#!/bin/bash
# round 4, last GPU call: the profiles of the round's final library (kernel trace + PMC passes, default bench line, the
# whole GPU suite) and the full-size runs whose figures the docs quote
mkdir -p gpurun_out/r04
export TMPDIR=/tmp
tools/pmc_collect.sh r04 > gpurun_out/pmc_r04.log 2>&1
( time timeout 900 python bench.py ) > gpurun_out/r04/bench_default.log 2>&1; echo "bench rc=$?"
timeout 1300 python -m pytest tests -m gpu -x -q --timeout=900 > gpurun_out/r04/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/r04/gpu_suite.log
timeout 600 python tools/native_pipeline_run.py 48000000 46000000 25769803776 gpurun_out/r04/native_c3_final.json 5 > gpurun_out/r04/native_c3_final.log 2>&1; echo "native rc=$?"
timeout 600 python tools/scale_run.py c3pruned gpurun_out/r04/scale_c3_final.json > gpurun_out/r04/scale_c3_final.log 2>&1; echo "c3pruned rc=$?"
timeout 600 python tools/nw_big.py 62250000 57000000 gpurun_out/r04/nw_c5q_final.json > gpurun_out/r04/nw_c5q_final.log 2>&1; echo "nw rc=$?"
DROPIN_EXTRA="--gpu-stage4" timeout 700 python tools/dropin_scale.py 48000000 46000000 24G gpurun_out/r04/dropin_c3_final.json > gpurun_out/r04/dropin_c3_final.log 2>&1; echo "dropin rc=$?"
timeout 900 python tools/chain_nw_run.py 62250000 57000000 8 gpurun_out/r04/chain_nw_final.json > gpurun_out/r04/chain_nw_final.log 2>&1; echo "chain rc=$?"
python - <<'PY'
import json
for f in ("native_c3_final", "scale_c3_final", "nw_c5q_final", "dropin_c3_final", "chain_nw_final"):
    try:
        d = json.load(open("gpurun_out/r04/%s.json" % f))
        print(f, json.dumps(d)[:700])
    except Exception as e:
        print(f, "missing", e)
PY
