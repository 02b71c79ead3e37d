#!/bin/bash
# tools/pmc_collect.sh TAG  -- run ON THE GPU BOX (through gpurun) from the repo root.
# rocprofv3 kernel-trace statistics of the default bench command plus the PMC passes (each in its own run, no trace
# domains next to --pmc), summarised with tools/rocpd_summary.py into gpurun_out/prof_TAG/*.{csv,json}; copy those
# into profiles/ and run tools/pmc_index.py to refresh profiles/pmc_index.json for this kernel build.
set -x
tag=${1:-r02}
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --no-cpu-baseline --no-target-shape --no-shapes --no-c3-full"
rocprofv3 --kernel-trace --stats -d $out/stats -- $BENCH --steps 3 --warmup 1 > $out/bench_stats.json 2> $out/bench_stats.err
python3 tools/rocpd_summary.py stats $(find $out/stats -name '*.db' | head -1) $out/${tag}_pk16_kernel_stats.csv $out/${tag}_pk16_kernel_dispatches.csv > /dev/null
rocprofv3 --pmc FETCH_SIZE -d $out/fetch -- $BENCH --steps 1 --warmup 0 > $out/bench_fetch.json 2> $out/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $out/write -- $BENCH --steps 1 --warmup 0 > $out/bench_write.json 2> $out/bench_write.err
python3 tools/rocpd_summary.py pmc $out/${tag}_pk16_hbm_pmc.json "rocprofv3 --pmc FETCH_SIZE -- $BENCH --steps 1 --warmup 0 ; same with --pmc WRITE_SIZE (separate passes, no trace domains)" \
    $(find $out/fetch -name '*.db' | head -1) $(find $out/write -name '*.db' | head -1) > /dev/null
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $out/sq -- $BENCH --steps 1 --warmup 0 > $out/bench_sq.json 2> $out/bench_sq.err
python3 tools/rocpd_summary.py pmc $out/${tag}_pk16_sq_pmc.json "rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -- $BENCH --steps 1 --warmup 0" \
    $(find $out/sq -name '*.db' | head -1) > /dev/null
cp $out/bench_stats.json $out/${tag}_bench_line.json
rm -rf $out/stats $out/fetch $out/write $out/sq
ls -la $out
