"""GPU: the native pipeline (masa-cudalign_amd/pipeline.py: stage1.py, stage2.py, stage3.py drive the ENGINE through
mi355sw_align_partition / mi355sw_match_last_column, stage 4 is mi355sw_stage4, stages 5-6 host code) on the
full-pipeline fixtures: best score, stage-2 crosspoints and alignment.00.txt as MASA-Core wrote them.

The same drivers are pinned on the CPU against MASA-Core byte for byte (tests/test_native_pipeline.py, on the aligner
double); the same engine calls are made by MASA-Core's own stages 2-3 in tests/test_gpu_dropin.py.  What is new here is
the combination -- the Python AlignerManager as the engine's callback table.

First green on an MI355X in round 3 (gpurun_out/r03 -> profiles/r03_native_pipeline_*): every case counts.  Each case
still runs in a CHILD process with a time limit (tests/native_pipeline_cases.py), so that a hang is a failure with the
child's output, not a stuck session.  The file sorts last."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _case(name, limit_s=180, extra=()):
    try:
        p = subprocess.run([sys.executable, os.path.join(HERE, "native_pipeline_cases.py"), name] + list(extra), stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=limit_s)
        rc, log = p.returncode, p.stdout.decode(errors="replace")
    except subprocess.TimeoutExpired as e:
        rc, log = -1, "timed out after %d s\n%s" % (limit_s, (e.stdout or b"").decode(errors="replace"))
    if rc != 0:
        pytest.fail("native pipeline case %s: exit code %d\n%s" % (name, rc, log[-3000:]))
    res = json.loads([ln for ln in log.splitlines() if ln.startswith("{")][-1])
    assert res["ok"] and all(res["checks"].values()), res
    return res


@pytest.mark.parametrize("name", ["b8192_3000x2700", "b8192_20000x9000"])
def test_native_pipeline_on_the_engine(name):
    """fixtures made with the engine's special-row spacing (8192 rows): the traceback coincides byte for byte"""
    res = _case(name)
    assert {"best", "crosspoints_2", "alignment_txt", "alignment_score"} <= set(res["checks"])


def test_native_pipeline_other_geometry_same_optimum():
    """against the fixture made with 128-row blocks: another spacing may pick another, equally optimal path -- score,
    start and end of the alignment are the same, and stage 5 re-scores the path to the best score"""
    res = _case("other_geometry")
    assert res["checks"]["start_and_end"]


def test_native_pipeline_with_pruning_biting():
    """60000 x 50000 with block pruning on in stage 1: the special rows are lower bounds off the optimal path, the
    traceback on top of them recovers the reference's crosspoints and text"""
    res = _case("pruned_60000x50000", limit_s=400)
    assert res["checks"]["pruned"] and res["checks"]["alignment_txt"]


def test_native_pipeline_global_alignment_with_pruning():
    """--alignment-edges=++ through all six stages with block pruning ON in stage 1 (BASELINE config 5 as worded: the
    reference holds the bound, AbstractBlockPruning.cpp:92-109, but its stage 1 never asks for it, sw_stage1.cpp:219-225):
    more than a quarter of the matrix is skipped, the special rows are lower bounds off the optimal paths -- and best
    score, stage-2 crosspoints, crosspoint_04 and alignment.00.txt are the files MASA-Core wrote WITHOUT pruning."""
    res = _case("global_pruned_60000x50000", limit_s=400)
    assert res["checks"]["pruned"] and res["checks"]["alignment_txt"] and res["checks"]["crosspoints_2"] and res["checks"]["crosspoints_4"]
    assert res["pruned_fraction"] > 0.25
    res = _case("global_unpruned_60000x50000", limit_s=400)
    assert res["checks"]["alignment_txt"] and res["pruned_fraction"] == 0


def test_native_pipeline_equals_the_dropin_at_4M():
    """4 000 000 x 4 000 000 related pair, pruning biting, special rows on disk: native stages 1-6 = MASA-Core's own
    stages on the same engine (crosspoint files and alignment.00.txt byte for byte), alignment re-scores to the best"""
    if not os.path.exists(os.path.join(os.path.dirname(HERE), "oracle", "_ref", "masa_mi355")):
        pytest.skip("oracle/_ref/masa_mi355 not prebuilt (needs /root/reference at build time)")
    res = _case("at_size", limit_s=900, extra=["4000000", "4000000"])
    assert res["checks"]["alignment_txt"] and res["checks"]["crosspoints_4"]
    print("native %.1f s (stages %s), drop-in %.1f s" % (res["native_seconds"], res["seconds"], res["dropin_seconds"]))
