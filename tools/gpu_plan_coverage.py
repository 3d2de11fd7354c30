"""Device side of tools/plan_coverage.py (VERDICT r5 next #1b): for stacks this repo did NOT design, what a block costs fused
against the same block layer by layer (BIRDA_HIP_FUSE=0: expand GEMM / depthwise / [pool, gate, scale] / project GEMM), per block,
from HIP events around every launch (bh_batch_context_layer_ms).

    python tools/gpu_plan_coverage.py [n_segments] [precision] [plan ...]      -> profiles/r6_plan_coverage.txt

Plans: the five probe plans of the round-5 verdict on the BirdNET front-end (+ the two EfficientNet ones with their gates), the
repo's own two headline plans for scale, and full-size random plans.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
PREC = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
ONLY = sys.argv[3:]
REPS = 3


def block_times(path, m, fused):
    if fused:
        os.environ.pop("BIRDA_HIP_FUSE", None)
    else:
        os.environ["BIRDA_HIP_FUSE"] = "0"
    clf = BirdClassifier(path, precision=PREC)
    nf = len(clf.fused_blocks())
    cfgs = clf.fused_blocks()
    ctx = clf.create_batch_context(N)
    base = synth.synth_segments(8, m.sample_count, m.sample_rate)
    x = torch.from_numpy(np.tile(base, (N // 8 + 1, 1))[:N]).cuda()
    logits = torch.empty((N, m.n_classes), device="cuda")
    idx = torch.empty((N, 5), dtype=torch.int32, device="cuda")
    conf = torch.empty((N, 5), device="cuda")
    for _ in range(2):
        clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    ctx.set_profiling(True)
    for _ in range(REPS):
        clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    ly = [ms / REPS * 1e3 for ms, _ in ctx.layer_ms()]
    out = logits[:4].cpu().numpy().copy()
    ctx.close()
    clf.close()
    os.environ.pop("BIRDA_HIP_FUSE", None)
    return ly, nf, cfgs, out


def blocks_of(m):
    """[(first layer, last layer, description)] of every inverted-residual block (its stem conv included when it has no expand conv)"""
    L = m.layers
    out = []
    for i, D in enumerate(L):
        if D.op != mf.OP_DWCONV:
            continue
        first = i
        prev = L[i - 1] if i else None
        if prev is not None and D.in_tensor == i and ((prev.op == mf.OP_PWCONV and prev.act != mf.ACT_NONE and prev.res_tensor == mf.NO_TENSOR) or (prev.op == mf.OP_CONV and prev.in_tensor == 0)):
            first = i - 1
        last = next(k for k in range(i + 1, len(L)) if L[k].op == mf.OP_PWCONV and L[k].in_h == D.out_h and L[k].act == mf.ACT_NONE and L[k].in_h * L[k].in_w > 1)
        E = L[first]
        cin = D.cin if first == i else E.cin
        out.append((first, last, f"{'stem ' if E.op == mf.OP_CONV else ''}{cin:4d} -> {D.cout:5d} -> {L[last].cout:4d} k{D.kh} s{D.sh} {D.in_h:3d}x{D.in_w:<3d}{' se' if last - i > 1 else ''}"))
    return out


def main():
    plans = [(k, synth.probe_plan(k)) for k in synth.PROBE_PLANS]
    plans += [(k + "+se", synth.probe_plan(k, se=True)) for k in ("efficientnet_b2", "b3_on_birdnet_image")]
    plans += [(f"random_big_{s}", synth.random_plan(1000 + s, big=True)) for s in range(4)]
    models = [("birdnet_v24 (this repo's plan)", synth.build_model("birdnet_v24"))]
    for name, plan in plans:
        if "classes" in plan and plan["classes"] > 1000:
            plan = dict(plan, classes=200, head=256)
        models.append((name, synth.build_model("custom", plan=plan)))
    tot_f = tot_l = 0.0
    for name, m in models:
        if ONLY and not any(o in name for o in ONLY):
            continue
        path = "/tmp/_cov.bhm"
        mf.write_model(path, m)
        lf, nf, cfgs, of = block_times(path, m, True)
        ll, _, _, ol = block_times(path, m, False)
        bl = blocks_of(m)
        scale = max(1.0, float(np.abs(ol).max()))
        print(f"== {name}: {nf} of {len(bl)} blocks fused, {N} segments a launch, {PREC}; fused vs layer by layer max |dlogit| {np.abs(of - ol).max() / scale:.1e} of the logit scale")
        sf = sl = 0.0
        n_base = 301
        for k, (a, b, desc) in enumerate(bl):
            tf, tl = sum(lf[a:b + 1]), sum(ll[a:b + 1])
            sf += tf
            sl += tl
            print(f"   {desc:46s} fused {tf:8.1f} us   layers {tl:8.1f} us   x{tl / max(tf, 1e-9):5.2f}" + (f"   entry {cfgs[k] % n_base}" if k < len(cfgs) else ""))
        print(f"   all blocks: fused {sf:9.1f} us, layer by layer {sl:9.1f} us per {N} segments: x{sl / max(sf, 1e-9):.2f}")
        tot_f += sf
        tot_l += sl
    print(f"TOTAL over the plans: fused {tot_f:.0f} us, layer by layer {tot_l:.0f} us: x{tot_l / max(tot_f, 1e-9):.2f}")


if __name__ == "__main__":
    main()
