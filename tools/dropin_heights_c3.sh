#!/bin/bash
# tools/dropin_heights_c3.sh OUTDIR: BASELINE config 3 through MASA-Core's own stages on the engine with different strip-height policies for
# stages 2 and 3 -- which of them leave which crosspoint_03 / alignment.00.txt (round 6: 256-row strips for stage 3's narrow partitions changed
# the digests at this size and at no smaller one)
out=$1; mkdir -p $out
run() { tag=$1; shift
    DROPIN_KEEP_CROSSPOINTS=$out/cp_$tag DROPIN_EXTRA="--gpu-stage4 $*" python3 tools/dropin_scale.py 48000000 46000000 24G $out/c3_$tag.json > $out/c3_$tag.log 2>&1
    python3 -c "
import json; d=json.load(open('$out/c3_$tag.json'))
print('$tag', 'wall %.1f' % d['wall_s'], 'cp2', d.get('crosspoint_02_sha256','')[:8], 'cp3', d.get('crosspoint_03_sha256','')[:8], 'cp4', d.get('crosspoint_04_sha256','')[:8], 'text', d.get('alignment_sha256','')[:8], 'rescore ok', d.get('rescore_equals_best'), 'stage2 %.0f ms stage3 %.0f ms' % (d['stage2']['TOTAL'], d['stage3']['TOTAL']))"
}
run default
run model_heights --engine-flags=32768
run rows256 --strip-rows=256
python3 - <<PY
import os
def load(t):
    fn = os.path.join("$out", "cp_" + t, "crosspoint_03.00")
    return [l.strip() for l in open(fn)] if os.path.exists(fn) else []
a, b = load("default"), load("model_heights")
diff = [(k, x, y) for k, (x, y) in enumerate(zip(a, b)) if x != y]
print("crosspoint_03 default vs model heights: %d / %d lines, %d differ; first: %s" % (len(a), len(b), len(diff), diff[:6]))
PY
rm -rf $out/cp_*
