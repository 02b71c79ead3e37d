#!/bin/bash
# tools/dropin_heights_probe.sh OUTDIR [M N]: does the alignment MASA-Core's stages 2-6 produce on the engine depend on the strip height
# the engine picks for their partitions?  The same pair through oracle/_ref/masa_mi355 with the engine's own choice (goal-stopped
# sweeps: 512 / 256 rows), and with --strip-rows fixed at 256, 512, 1024 and 2048: digests of crosspoint_02/03/04 and alignment.00.txt.
out=$1; m=${2:-10000000}; n=${3:-10000000}
mkdir -p $out
for v in default 256 512 1024 2048; do
    extra="--gpu-stage4"; [ $v != default ] && extra="--gpu-stage4 --strip-rows=$v"
    DROPIN_EXTRA="$extra" python3 tools/dropin_scale.py $m $n 4G $out/heights_$v.json > $out/heights_$v.log 2>&1
    python3 -c "
import json; d=json.load(open('$out/heights_$v.json'))
print('$v', 'wall %.1f' % d['wall_s'], 'cp2', d.get('crosspoint_02_sha256','')[:8], 'cp3', d.get('crosspoint_03_sha256','')[:8], 'cp4', d.get('crosspoint_04_sha256','')[:8], 'text', d.get('alignment_sha256','')[:8], 'rescore ok', d.get('rescore_equals_best'), 'stage2 %.0f ms stage3 %.0f ms' % (d['stage2']['TOTAL'], d['stage3']['TOTAL']))"
done
