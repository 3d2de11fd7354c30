#!/usr/bin/env python3
"""Which tile configuration does the planner pick for every fused block of every synthetic model, in every precision and for
every templated activation?  Host logic only (bh_plan_fused_blocks: no GPU needed).  The union is the set of configurations the
product library has to ship; tests/test_abi_and_host.py holds mbconv_cfgs.inc to it.

    python tools/plan_models.py            # table + the reachable set
    python tools/plan_models.py --json     # {"reachable": [...], "by_model": {...}}
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MODELS = ["birdnet_v24", "perch_v2", "perch_v2_tiny", "birdnet_v30", "mini", "mini_b0", "mini_hg", "mini_se", "birdnet_v24_tiny"]
PRECISIONS = {"f32": 3, "f16x3": 1, "f16": 2}   # BH_FLAG_* (auto plans as f16x3)


def plan(path: str, flags: int):
    from birda_amd import _lib
    L = _lib.load()
    cfgs = (C.c_int32 * 256)()
    layers = (C.c_int32 * 256)()
    n = L.bh_plan_fused_blocks(path.encode(), flags, cfgs, layers, 256)
    if n < 0:
        raise RuntimeError(L.bh_last_error().decode())
    return [(int(layers[i]), int(cfgs[i])) for i in range(min(n, 256))]


def fusable_triples(m):
    """expand / stem -> depthwise -> project triples of a model (what COULD fuse), by first layer"""
    from birda_amd import modelfile as mf
    out = []
    L = m.layers
    for i in range(len(L) - 2):
        stem = L[i].op == mf.OP_CONV and L[i].in_tensor == 0
        if (L[i].op == mf.OP_PWCONV or stem) and L[i + 1].op == mf.OP_DWCONV and L[i + 2].op == mf.OP_PWCONV \
                and L[i + 1].in_tensor == i + 1 and L[i + 2].in_tensor == i + 2:
            out.append(i)
    fused3 = set(out) | {i + 1 for i in out} | {i + 2 for i in out}
    for i in range(len(L) - 1):   # depthwise -> project without an expand convolution
        if i not in fused3 and L[i].op == mf.OP_DWCONV and L[i + 1].op == mf.OP_PWCONV and L[i + 1].in_tensor == i + 1:
            out.append(i)
    return out


def survey(models=MODELS, acts=(None,)):
    from birda_amd import modelfile as mf, synth
    n_base = None
    by_model = {}
    reach = set()
    with tempfile.TemporaryDirectory() as d:
        for kind in models:
            for act in acts:
                m = synth.build_model(kind, act=act)
                p = os.path.join(d, f"{kind}_{act}.bhm")
                mf.write_model(p, m)
                triples = fusable_triples(m)
                for prec, flag in PRECISIONS.items():
                    both = plan(p, flag)
                    # (bh_plan_fused_blocks lists a block's small-launch twin -- one segment per workgroup, launches of <= 256
                    #  segments -- right behind the block: same layer index)
                    got, twins, seen = [], [], set()
                    for l, c in both:
                        (twins if l in seen else got).append((l, c))
                        seen.add(l)
                    key = f"{kind}/{'default' if act is None else act}/{prec}"
                    starts = {l for l, _ in got}
                    # (a candidate whose depthwise layer heads a fused two-layer block is not a triple: its first layer has two readers)
                    by_model[key] = {"fused": got, "twins": twins, "unfused_triples": sorted(t for t in set(triples) - starts if t + 1 not in starts)}
                    reach |= {c for _, c in both}
                os.remove(p)
    return reach, by_model


def main():
    from birda_amd import modelfile as mf
    acts = (None, mf.ACT_SWISH, mf.ACT_RELU6)
    reach, by_model = survey(acts=acts)
    if "--json" in sys.argv:
        print(json.dumps({"reachable": sorted(reach), "by_model": by_model}))
        return
    for k, v in by_model.items():
        print(f"{k:40s} fused {len(v['fused']):2d} {[c for _, c in v['fused']]}" + (f"  twins {[c for _, c in v['twins']]}" if v["twins"] else "") +
              (f"  NOT fused at layers {v['unfused_triples']}" if v["unfused_triples"] else ""))
    print("reachable configuration indices (activation copies folded):", sorted(reach))


if __name__ == "__main__":
    main()
