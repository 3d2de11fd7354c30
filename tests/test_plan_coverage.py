"""Host-side half of VERDICT r5 next #1: the tile planner on conv stacks this repo did not design (no GPU: bh_plan_fused_blocks).

The reference runs whatever `.onnx` it is handed (src/inference/classifier.rs:269-283); the fused MBConv path must not depend on
the two channel plans of birda_amd/synth.py.  Every inverted-residual block of the five probe plans the round-5 judge used (1 / 16
... 16 / 23 fused then) and of 50 seeded random stacks must find a tile entry, in split-f16 AND in f32 (the path BH_FLAG_AUTO
re-runs an overflowing row on).  The device side -- the same plans against the oracle -- is tests/test_random_plans_gpu.py.
"""
import io
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_every_block_of_plans_nobody_tiled_by_hand_finds_a_fused_entry():
    import plan_coverage
    from birda_amd import synth
    plans = [(k, synth.probe_plan(k)) for k in synth.PROBE_PLANS]
    plans += [(k + "+se", synth.probe_plan(k, se=True)) for k in ("efficientnet_b2", "b3_on_birdnet_image")]
    plans += [(f"random_{s}", synth.random_plan(s)) for s in range(40)]
    plans += [(f"random_big_{s}", synth.random_plan(1000 + s, big=True)) for s in range(10)]
    out = io.StringIO()
    tot = plan_coverage.survey(plans, why=True, out=out)
    text = out.getvalue()
    for prec in ("f16x3", "f32", "f16"):
        fused, blocks = tot[prec]
        assert blocks > 400 and fused >= 0.95 * blocks, text
    # the five probe plans: every block ("16/16, 17/17, 23/23, 26/26, 16/16")
    want = {"b0x1.5_stem48": 16, "mobilenet_v2": 17, "efficientnet_b2": 23, "b3_on_birdnet_image": 26, "b0_plus8": 16}
    for line in text.splitlines():
        name = line.split()[0] if line.strip() else ""
        if name in want:
            assert line.count(f"{want[name]:3d}/{want[name]:<3d}") == 3, line


def test_this_repos_own_plans_keep_their_hand_tuned_entries():
    """the relaxed passes come AFTER the exact ones: the BirdNET-v2.4-shaped and Perch-sized stacks plan exactly as in round 5"""
    import plan_models
    _, by_model = plan_models.survey(models=["birdnet_v24", "perch_v2"])
    assert [c for _, c in by_model["birdnet_v24/default/f16x3"]["fused"]] == [65, 48, 49, 50, 51, 52, 113, 113, 99, 101, 101, 58, 103, 103, 103, 105]
    got = [c % 301 for _, c in by_model["perch_v2/default/f16x3"]["fused"]]
    assert len(got) == 26 and max(got) < 211, got
