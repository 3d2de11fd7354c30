"""GPU box: byte-level fuzz of everything bh_*_create reads from disk, THROUGH the device path -- mutants of a BHM1 model, of its
`.onnx` twin and of its labels file through bh_classifier_create + a forward of two segments; of a BHC1 custom classifier through
bh_custom_classifier_create; of the reference's fixture geomodel (.onnx + labels) through bh_range_filter_create + a query.  Whatever
the bytes: an error code and a message, or a classifier that runs; never a crash, a hang or a device fault.  Child processes with
time limits; the header-heavy first kilobytes are mutated most.
    python tools/fuzz_create.py [n_mutants] [seed]"""
import os, random, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GEO = os.path.join(ROOT, "tests", "golden", "reference_fixtures")


def child(jobs):
    import numpy as np
    from birda_amd import synth
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import BirdClassifier, CustomClassifier, RangeFilter
    segs = None
    for job in jobs:
        kind, a, b = job.split("|")
        rc, what = 0, "ok"
        try:
            if kind == "model":
                clf = BirdClassifier(a, b or None, top_k=3, min_confidence=0.1)
                n = clf.sample_count() if callable(clf.sample_count) else clf.sample_count
                if not 0 < n <= 20_000_000: raise ValueError(f"sample_count {n}")
                ctx = clf.create_batch_context(2)
                x = synth.synth_segments(2, int(n), 48000)
                clf.predict_logits(ctx, x)
                ctx.close(); clf.close()
            elif kind == "custom":
                cc = CustomClassifier(a, b or None)
                cc.close() if hasattr(cc, "close") else None
            else:
                rf = RangeFilter(a, b, threshold=0.03)
                rf.predict(60.17, 24.94, 6, 1)
                rf.close()
        except BirdaHipError as e:
            rc, what = e.code, str(e)[:160]
        except Exception as e:
            rc, what = -99, type(e).__name__ + ": " + str(e)[:160]
        print("@" + kind, os.path.basename(a), os.path.basename(b) if b else "-", rc, "#", " ".join(what.split()), flush=True)


def mutate(rng, raw, header=1500):
    b = bytearray(raw)
    for _ in range(rng.choice((1, 1, 2, 3))):
        kind = rng.randrange(6)
        pos = rng.randrange(min(len(b), header)) if rng.random() < 0.75 else rng.randrange(len(b))
        if kind == 0: b[pos] ^= 1 << rng.randrange(8)
        elif kind == 1: b[pos] = rng.choice((0, 1, 0x7f, 0x80, 0xff, rng.randrange(256)))
        elif kind == 2: b = b[:rng.randrange(len(b))]
        elif kind == 3: b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
        elif kind == 4: b[pos:pos + 4] = struct.pack("<I", rng.choice((0, 1, 2, 0xffffffff, 0x7fffffff, 0x80000000, rng.randrange(1 << 32))))
        else:
            q = rng.randrange(len(b)); b[pos:pos + 8] = b[q:q + 8]
        if not b: b = bytearray(b"\x00")
    return bytes(b)


def main():
    import numpy as np
    from birda_amd import convert, modelfile as mf, onnx_io as ox, synth
    n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    d = tempfile.mkdtemp()
    m = synth.build_model("mini_se")
    bhm, lab = os.path.join(d, "base.bhm"), os.path.join(d, "base.txt")
    mf.write_model(bhm, m); synth.write_labels(lab, m.n_classes)
    onnx = os.path.join(d, "base.onnx")
    open(onnx, "wb").write(ox.dump(convert.graph_from_model(m, frontend_spelling="conv1d")))
    cc = os.path.join(d, "base.bhc"); ccl = os.path.join(d, "base_cc.txt")
    cm = synth.build_custom_classifier(m.embedding_dim or 64, 12, hidden=(32,))
    mf.write_custom_classifier(cc, cm); synth.write_labels(ccl, 12)
    geo, geol = os.path.join(GEO, "fixture-geomodel.onnx"), os.path.join(GEO, "fixture-geomodel-labels.txt")
    raw = {p: open(p, "rb").read() for p in (bhm, lab, onnx, cc, ccl, geo, geol)}
    jobs = []
    for i in range(n):
        t = rng.randrange(7)
        def w(src, ext, hdr=1500):
            p = os.path.join(d, f"m{i:05d}{ext}"); open(p, "wb").write(mutate(rng, raw[src], hdr)); return p
        if t == 0: jobs.append(f"model|{w(bhm, '.bhm')}|{lab}")
        elif t == 1: jobs.append(f"model|{w(onnx, '.onnx', 3000)}|{lab}")
        elif t == 2: jobs.append(f"model|{bhm}|{w(lab, '.txt', 1 << 30)}")
        elif t == 3: jobs.append(f"custom|{w(cc, '.bhc')}|{ccl}")
        elif t == 4: jobs.append(f"custom|{cc}|{w(ccl, '.txt', 1 << 30)}")
        elif t == 5: jobs.append(f"geo|{w(geo, '.onnx', 3000)}|{geol}")
        else: jobs.append(f"geo|{geo}|{w(geol, '.txt', 1 << 30)}")
    bad, codes = 0, {}
    for i in range(0, n, 20):
        batch = jobs[i:i + 20]
        try:
            rr = subprocess.run([sys.executable, __file__, "--child"] + batch, capture_output=True, text=True, timeout=400)
        except subprocess.TimeoutExpired:
            print("TIMEOUT in batch", i, batch); bad += 1; continue
        lines = [l[1:] for l in rr.stdout.split("\n") if l.startswith("@")]
        for l in lines:
            f = l.split()
            if len(f) < 4: continue
            key = f[0] + ":" + f[3]; codes[key] = codes.get(key, 0) + 1
            if f[3] in ("-4", "-7", "-99"): print(l[:300])
        if rr.returncode != 0 or len(lines) != len(batch):
            bad += 1
            print(f"CRASH rc {rr.returncode} after {len(lines)} of batch {i}: {batch[len(lines)] if len(lines) < len(batch) else '?'}\n{rr.stderr[-500:]}")
    print(f"{n} mutants, {bad} bad batches; kind:code {dict(sorted(codes.items()))}" + (f"; files kept in {d}" if bad else ""))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child": child(sys.argv[2:])
    else: main()
