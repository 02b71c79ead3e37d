#!/bin/bash
# tools/gpu_call.sh TAG STEP [STEP...]  -- ONE script for the GPU calls of a round (run ON THE GPU BOX through gpurun, from the
# repo root); replaces the per-round one-offs (r03_*.sh, r04_*.sh).  Everything lands under gpurun_out/TAG/.
#   new:FILES      pytest -m gpu -x of the listed test files (comma separated, without tests/ and .py)
#   suite          the whole GPU suite (pytest tests -m gpu), then __graft_entry__.smoke()
#   fastsuite[:N]  the GPU suite under pytest-xdist (N workers, default 6, one test FILE per worker at a time: module fixtures stay whole),
#                  then the tests that failed once more ALONE -- the box has 256 host cores and the suite's time is the oracle's
#                  and Python's, not the GPU's: 140 s instead of 500 s (profiles/r05_gpu_suite_xdist_rehearsal.log); tests that
#                  bound a run's TIME may fail next to five other processes on the GPU, hence the second, serial pass
#   bench          python3 bench.py (the driver's N = 1 command) -> bench.json
#   selflaunch:N   MI355SW_BENCH_REHEARSAL=1 python3 bench.py --gpus N --size 300000  (bench.py starts its own ranks; all on cuda:0)
#   launchcheck    bench.py --launch-check: 2 ranks (gloo on a one-GPU box), then the nccl (= RCCL) backend with one rank on the GPU
#   rehearse       the N > 1 path of bench.py on the one GPU through every transport and recurrence (self-launched)
#   ab:M,N[,R]     a related M x N pair: unpruned, pruned with the window, pruned without it (tools/window_probe.py)
#   c3             C3's stage 1 at full size, pruned, against the recorded unpruned run (tools/scale_run.py c3pruned) -> scale_c3pruned.json
#   native[:chain] BASELINE config 3 (48 M x 46 M) through the native pipeline, stages 1-6 (chain: MI355SW_STAGE2_SPECULATE=0, stage 2 as the plain chain) -> native_pipeline_c3[_spec].json
#   dropin[:OPTS]  ... through MASA-Core's own stages on the engine (tools/dropin_scale.py, --gpu-stage4 OPTS) -> dropin_pipeline_c3.json
#   pmc            rocprofv3 kernel statistics + PMC passes of the default bench command (tools/pmc_collect.sh TAG)
#   py:SCRIPT,ARGS python3 tools/SCRIPT ARGS... (comma separated)
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
for step in "$@"; do
    name=${step%%:*}; arg=""; [ "$step" != "$name" ] && arg=${step#*:}
    t0=$(date +%s)
    case $name in
    new)
        files=$(echo $arg | tr ',' '\n' | sed 's#^#tests/#; s#$#.py#' | tr '\n' ' ')
        timeout 1500 python3 -m pytest $files -x -q -m gpu -s > $out/new_tests.log 2>&1; rc=$?
        tail -15 $out/new_tests.log ;;
    suite)
        timeout 2400 python3 -m pytest tests -q -m gpu > $out/suite.log 2>&1; rc=$?
        tail -8 $out/suite.log
        timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log ;;
    fastsuite)
        timeout 1200 python3 -m pytest tests -q -m gpu -n ${arg:-6} --dist loadfile > $out/fastsuite.log 2>&1; rc=$?
        tail -5 $out/fastsuite.log
        if [ $rc -ne 0 ]; then
            timeout 1200 python3 -m pytest tests -q -m gpu --last-failed > $out/fastsuite_failed_alone.log 2>&1; rc=$?
            echo "-- the failed ones alone:"; tail -5 $out/fastsuite_failed_alone.log
        fi
        timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $out/smoke.log ;;
    bench)
        timeout 1700 python3 bench.py > $out/bench.json 2> $out/bench.err; rc=$?
        python3 - <<PY
import json
try:
    d = json.loads(open("$out/bench.json").read().strip().splitlines()[-1])
    print("bench value %.1f GCUPS, ms/step %.2f, roofline %.5f, kernel %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["kernel_name"]))
    for k in ("target_shape", "c3_shape", "c5_shape", "c3_full", "cpu_baseline"):
        if k in d:
            v = d[k]
            print(" ", k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list))}, v.get("check"))
except Exception as e:
    print("bench line unreadable:", e)
PY
        ;;
    selflaunch)
        MI355SW_BENCH_REHEARSAL=1 MI355SW_BENCH_TIMEOUT_S=600 python3 bench.py --gpus $arg --steps 2 --warmup 1 --size 300000 > $out/selflaunch_$arg.json 2> $out/selflaunch_$arg.err; rc=$?
        python3 -c "
import json
d=json.loads(open('$out/selflaunch_$arg.json').read().strip().splitlines()[-1]); c=d['config']
print('self-launched N=%d: value %.0f comm %s launcher %s xgmi %s note %s best %s' % (d['n_gpus'], d['value'], c['comm'], c['launcher'], c['xgmi'], c['comm_note'], d['best']))
for k in ('c4_half', 'c4_full', 'c5_full'):
    if k in d:
        v = d[k]; print(' ', k, {kk: (round(vv, 2) if isinstance(vv, float) else vv) for kk, vv in v.items() if not isinstance(vv, (dict, list))}, v.get('check'))" ;;
    launchcheck)  # --launch-check on the box: N ranks over gloo (one GPU: fewer devices than ranks), then RCCL itself with the one rank a one-GPU box allows
        python3 bench.py --gpus 2 --launch-check > $out/launch_check_n2.json 2> $out/launch_check_n2.err; rc=$?
        MI355SW_LAUNCH_CHECK_BACKEND=nccl timeout 600 python3 bench.py --gpus 1 --launch-check > $out/launch_check_nccl_one_rank.json 2> $out/launch_check_nccl_one_rank.err; r2=$?
        [ $r2 -ne 0 ] && rc=$r2
        cat $out/launch_check_n2.json $out/launch_check_nccl_one_rank.json; tail -3 $out/launch_check_nccl_one_rank.err ;;
    rehearse)
        rc=0
        run() { n=$1; t=$2; shift 2
            MI355SW_BENCH_REHEARSAL=1 MI355SW_BENCH_TIMEOUT_S=600 python3 bench.py --gpus $n --steps 2 --warmup 1 --size 300000 "$@" > $out/rehearsal_$t.json 2> $out/rehearsal_$t.err
            r=$?; [ $r -ne 0 ] && rc=$r
            python3 -c "
import json
d=json.loads(open('$out/rehearsal_$t.json').read().strip().splitlines()[-1]); c=d['config']
print('rehearsal $t rc=$r: value %.0f comm %s launcher %s pruned %.3f best %s kernel %s' % (d['value'], c['comm'], c['launcher'], c['pruned_fraction'], d['best'], c['kernel_name']))" || echo "rehearsal $t rc=$r: no line"; }
        run 2 n2
        run 4 n4_related --related
        run 4 n4_nw_related --nw --related
        run 8 n8_nw_related --nw --related
        MI355SW_BENCH_COMM=host run 4 n4_related_host --related
        MI355SW_BENCH_FAIL_IPC=1 run 4 n4_attach --related
        MI355SW_BENCH_FAIL_IPC=1 MI355SW_BENCH_NO_ATTACH=1 run 4 n4_host_after_both_failed --related ;;
    ab)
        timeout 1500 python3 tools/window_probe.py $(echo $arg | tr ',' ' ') > $out/ab_$(echo $arg | tr ',' 'x').log 2>&1; rc=$?
        cat $out/ab_$(echo $arg | tr ',' 'x').log ;;
    c3)
        timeout 1500 python3 tools/scale_run.py c3pruned $out/scale_c3pruned.json > $out/scale_c3.log 2>&1; rc=$?
        tail -5 $out/scale_c3.log ;;
    native)       # BASELINE config 3 through the native pipeline (stages 1-6); native:spec = stage 2 from guessed crosspoints
        sfx=""; [ "$arg" = "spec" ] && { export MI355SW_STAGE2_SPECULATE=1; sfx="_spec"; }; [ "$arg" = "chain" ] && { export MI355SW_STAGE2_SPECULATE=0; sfx="_chain"; }
        timeout 1500 python3 tools/native_pipeline_run.py 48000000 46000000 25769803776 $out/native_pipeline_c3$sfx.json 5 > $out/native_c3$sfx.log 2>&1; rc=$?
        unset MI355SW_STAGE2_SPECULATE
        python3 -c "
import json; d=json.load(open('$out/native_pipeline_c3$sfx.json'))
print('native C3$sfx: total %.1f s, stages %s, alignment %s crosspoint_04 %s' % (d['total_seconds'], {k: round(v, 2) for k, v in d['seconds'].items()}, d['alignment_sha256'][:8], d['crosspoint_04_sha256'][:8]))" || tail -5 $out/native_c3$sfx.log ;;
    dropin)       # ... and through MASA-Core's own stages on the engine (oracle/_ref/masa_mi355)
        DROPIN_LOG=$out/dropin_c3_full.log DROPIN_EXTRA="--gpu-stage4 $arg" timeout 1500 python3 tools/dropin_scale.py 48000000 46000000 24G $out/dropin_pipeline_c3.json > $out/dropin_c3.log 2>&1; rc=$?
        grep "job " $out/dropin_c3_full.log | awk 'NR % 200 == 1' | head -40 > $out/dropin_c3_jobs_sample.log; rm -f $out/dropin_c3_full.log
        python3 -c "
import json; d=json.load(open('$out/dropin_pipeline_c3.json'))
print('drop-in C3: wall %.1f s, alignment %s crosspoint_04 %s, stage totals %s' % (d['wall_s'], d['alignment_sha256'][:8], d['crosspoint_04_sha256'][:8], {k: v.get('TOTAL', v.get('Total')) for k, v in d.items() if k.startswith('stage') and isinstance(v, dict)}))" || tail -5 $out/dropin_c3.log ;;
    pmc)
        bash tools/pmc_collect.sh $tag > $out/pmc.log 2>&1; rc=$? ;;
    py)
        script=${arg%%,*}; rest=""; [ "$arg" != "$script" ] && rest=$(echo ${arg#*,} | tr ',' ' ')
        timeout 1700 python3 tools/$script $rest > $out/${script%.py}.log 2>&1; rc=$?
        tail -20 $out/${script%.py}.log ;;
    *) echo "unknown step $step"; rc=2 ;;
    esac
    echo "== step $step rc=$rc $(( $(date +%s) - t0 )) s"
done
