#!/usr/bin/env python3
"""How much of a conv stack this repo did NOT design runs as fused launches?  Host logic only (bh_plan_fused_blocks: no GPU).

    python tools/plan_coverage.py [--random N] [--why]     # the five probe plans of VERDICT r5 + N seeded random plans

Per plan and precision: fused blocks / inverted-residual blocks.  --why prints, per unfused block, its shape.
(The device side of the same question -- us per block, fused against layer by layer -- is tools/gpu_plan_coverage.py.)
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PRECISIONS = {"f16x3": 1, "f32": 3, "f16": 2}


def fused_layers(path: str, flags: int):
    from birda_amd import _lib
    L = _lib.load()
    cfgs = (C.c_int32 * 1024)()
    layers = (C.c_int32 * 1024)()
    n = L.bh_plan_fused_blocks(path.encode(), flags, cfgs, layers, 1024)
    if n < 0:
        raise RuntimeError(L.bh_last_error().decode())
    out = {}
    for i in range(min(n, 1024)):
        out.setdefault(int(layers[i]), int(cfgs[i]))    # (twins follow their block under the same layer index)
    return out


def block_starts(m):
    """first layer of every inverted-residual block of a model: [stem conv | expand 1x1] -> depthwise ... -> project 1x1"""
    from birda_amd import modelfile as mf
    L = m.layers
    starts = []
    for i, D in enumerate(L):
        if D.op != mf.OP_DWCONV:
            continue
        prev = L[i - 1] if i else None
        # the expand conv (or the stem conv, when the block has none and sits right behind it) belongs to the block
        if prev is not None and D.in_tensor == i and ((prev.op == mf.OP_PWCONV and prev.act != mf.ACT_NONE and prev.res_tensor == mf.NO_TENSOR and prev.cout == D.cin)
                                                      or (prev.op == mf.OP_CONV and prev.in_tensor == 0)):
            starts.append((i - 1, D))
        else:
            starts.append((i, D))
    return starts


def survey(plans, why=False, out=sys.stdout):
    from birda_amd import modelfile as mf, synth
    tot = {p: [0, 0] for p in PRECISIONS}
    with tempfile.TemporaryDirectory() as d:
        for name, plan in plans:
            m = synth.build_model("custom", plan=plan)
            p = os.path.join(d, "m.bhm")
            mf.write_model(p, m)
            starts = block_starts(m)
            line = f"{name:28s} blocks {len(starts):3d}"
            missing = {}
            for prec, flag in PRECISIONS.items():
                got = fused_layers(p, flag)
                # (a stem block no entry serves runs its stem conv as a layer of its own and the rest as a no-expand block: fused)
                dwi = {id(D): k for k, D in enumerate(m.layers)}
                is_fused = lambda i, D: i in got or dwi[id(D)] in got
                nf = sum(1 for (i, D) in starts if is_fused(i, D))
                tot[prec][0] += nf
                tot[prec][1] += len(starts)
                line += f"   {prec} {nf:3d}/{len(starts):<3d}"
                missing[prec] = [(i, D) for (i, D) in starts if not is_fused(i, D)]
            print(line, file=out)
            if why:
                for prec, miss in missing.items():
                    for (i, D) in miss:
                        E = m.layers[i]
                        P = next(q for q in m.layers[i + 1:] if q.op == mf.OP_PWCONV and q.in_h == D.out_h and q.act == mf.ACT_NONE) if True else None
                        print(f"      {prec:6s} layer {i:3d}: {'stem ' if E.op == mf.OP_CONV else ''}{E.cin if E.op != mf.OP_DWCONV else D.cin:4d} -> {D.cout:5d} -> {P.cout:4d}  "
                              f"k{D.kh} s{D.sh}  {D.in_h}x{D.in_w} -> {D.out_h}x{D.out_w}  act {D.act}", file=out)
            os.remove(p)
    print("total: " + "   ".join(f"{p} {a}/{b} = {100.0 * a / max(b, 1):.1f} %" for p, (a, b) in tot.items()), file=out)
    return tot


def main():
    from birda_amd import synth
    n_rand = 40
    if "--random" in sys.argv:
        n_rand = int(sys.argv[sys.argv.index("--random") + 1])
    plans = [(k, synth.probe_plan(k)) for k in synth.PROBE_PLANS]
    plans += [(k + "+se", synth.probe_plan(k, se=True)) for k in ("efficientnet_b2", "b3_on_birdnet_image")]
    plans += [(f"random_{s}", synth.random_plan(s)) for s in range(n_rand)]
    plans += [(f"random_big_{s}", synth.random_plan(1000 + s, big=True)) for s in range(max(n_rand // 4, 4))]
    survey(plans, why="--why" in sys.argv)


if __name__ == "__main__":
    main()
