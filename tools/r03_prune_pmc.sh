#!/bin/bash
# SQ counters of an unpruned and a pruned sweep of the same related pair (separate kernels: the names tell them apart)
export TMPDIR=/tmp
out=gpurun_out/prof_prune
mkdir -p $out
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d $out/sq -- python3 tools/prune_probe.py ${1:-16000000} ${2:-12000000} 32 01 > $out/probe.log 2>&1
tail -2 $out/probe.log
python3 tools/rocpd_summary.py pmc $out/r03_prune_sq_pmc.json "rocprofv3 --pmc SQ_* -- python3 tools/prune_probe.py ${1:-16000000} ${2:-12000000} 32 01 (unpruned pass, then pruned pass)" $(find $out/sq -name '*.db' | head -1) | tail -60
rm -rf $out/sq
