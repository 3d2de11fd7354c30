"""GPU box: the argument surface of the call entry points -- segment counts, lengths, channels, rates, overlaps, capacities and start
tables at and beyond their edges (0, 1, just below / at / above the segment length, 2^31, 2^32 - 1, 2^63), null pointers where the
header allows none.  Whatever the arguments: an error code (and a message) or a result; never a crash, a hang or a device fault.
Child processes with a time limit.    python tools/fuzz_args.py [n_calls] [seed]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(seed, n):
    import ctypes as C
    import numpy as np
    from birda_amd import _lib, modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier, BhResult
    rng = random.Random(seed)
    d = tempfile.mkdtemp()
    m = synth.build_model("mini")
    mp, lp = os.path.join(d, "m.bhm"), os.path.join(d, "l.txt")
    mf.write_model(mp, m); synth.write_labels(lp, m.n_classes)
    clf = BirdClassifier(mp, lp, top_k=3, min_confidence=0.1)
    L = _lib.load()
    S = clf.sample_count()
    ctx = clf.create_batch_context(8)
    pcm = (np.random.default_rng(seed).uniform(-0.5, 0.5, 6 * S) * 32767).astype(np.int16)
    f32 = synth.synth_segments(8, S, 48000)
    res = (BhResult * 64)()
    starts = (C.c_uint64 * 64)()
    nseg = C.c_size_t()
    edge = [0, 1, 2, S - 1, S, S + 1, 2 * S, 6 * S, 1 << 31, (1 << 32) - 1, 1 << 40, (1 << 63)]
    codes = {}
    _choice = rng.choice
    def choice(seq):
        v = _choice(seq); print(f"  arg {v}", file=sys.stderr, flush=True); return v
    rng.choice = choice
    from birda_amd.classifier import RangeFilter
    GEO = os.path.join(ROOT, "tests", "golden", "reference_fixtures")
    rf = RangeFilter(os.path.join(GEO, "fixture-geomodel.onnx"), os.path.join(GEO, "fixture-geomodel-labels.txt"), threshold=0.03)
    nsp = rf.num_species()
    NC = clf.n_classes() if callable(clf.n_classes) else clf.n_classes
    odd = [0.0, -0.0, 1.0, -1.0, 90.0, -90.0, 180.0, 1e30, -1e30, float("nan"), float("inf"), -float("inf"), 60.17]
    for i in range(n):
        print(f"call {i}", file=sys.stderr, flush=True)
        which = rng.randrange(10)
        print(f" which {which}", file=sys.stderr, flush=True)
        st = rng.getstate()
        try:
            if which == 0:
                ch = rng.choice((0, 1, 2, 3, 65535))
                rc = L.bh_predict_pcm16(clf._h, ctx._h, pcm.ctypes.data, rng.choice([e for e in edge[:8] if e * max(ch, 1) <= pcm.size]), ch, rng.choice((0, 1, 8000, 44100, 48000, 47999, (1 << 32) - 1)),
                                        rng.choice(edge), res, rng.choice((0, 1, 64)), C.byref(nseg), starts)
            elif which == 1:
                ptrs = (C.c_void_p * 8)(*[f32[k].ctypes.data for k in range(8)])
                rc = L.bh_predict_batch_with_context(clf._h, ctx._h, ptrs, rng.choice((0, 1, 8, 9, 64, 1 << 40)) if rng.random() < 0.5 else 8, rng.choice(edge[:6]), res)
            elif which == 2:
                tab = (C.c_uint64 * 8)(*[rng.choice(edge) for _ in range(8)])
                ch = rng.choice((0, 1, 2))
                rc = L.bh_predict_pcm16_at(clf._h, ctx._h, pcm.ctypes.data, rng.choice([e for e in edge[:8] if e * max(ch, 1) <= pcm.size]), ch, rng.choice((0, 44100, 48000)), tab, rng.choice((0, 1, 8)), res)
            elif which == 3:
                out = np.zeros(4 * S, np.float32); got = C.c_size_t()
                rc = L.bh_resample(clf._h, f32[0].ctypes.data, rng.choice(edge[:6]), rng.choice((0, 1, 22050, 44100, 47999, 48000, (1 << 32) - 1)), rng.choice((0, 1, 32000, 48000)),
                                   out.ctypes.data, rng.choice((0, 1, out.size)), C.byref(got))
            elif which == 4:
                h = C.c_void_p()
                rc = L.bh_batch_context_create(clf._h, rng.choice((0, 1, 64, 1 << 40, (1 << 63))), C.byref(h))
                if rc == 0: L.bh_batch_context_destroy(h)
            elif which == 6:     # the geomodel query: coordinates and dates of every kind, capacities at the edges
                sc = np.zeros(nsp + 4, np.float32); ix = np.zeros(nsp + 4, np.uint32); kept = C.c_size_t()
                if rng.random() < 0.5:
                    rc = L.bh_range_filter_predict(rf._h, rng.choice(odd), rng.choice(odd), rng.choice((0, 1, 2, 12, 13, 255, (1 << 32) - 1)), rng.choice((0, 1, 28, 29, 30, 31, 32, (1 << 32) - 1)),
                                                   sc.ctypes.data, rng.choice((0, 1, nsp - 1, nsp, nsp + 4)), ix.ctypes.data, C.byref(kept))
                else:
                    rc = L.bh_range_filter_predict_week(rf._h, rng.choice(odd), rng.choice(odd), rng.choice(odd + [48.0, 49.0, 0.5]), sc.ctypes.data,
                                                        rng.choice((0, 1, nsp, nsp + 4)), ix.ctypes.data, C.byref(kept))
            elif which == 7:     # the filter tables: class counts that do not match, NaN / inf scores and thresholds
                k = rng.choice((0, 1, NC - 1, NC, NC + 1))
                tab = np.array([rng.choice(odd) for _ in range(max(k, 1))], np.float32)
                rc = L.bh_classifier_set_range_filter(clf._h, tab.ctypes.data, k, rng.choice(odd), rng.randrange(-1, 3), rng.randrange(-1, 3))
                if rc == 0: clf.predict_logits(ctx, f32); clf.predict_batch_with_context(ctx, [f32[0], f32[1]])
                L.bh_classifier_clear_filters(clf._h)
            elif which == 8:
                k = rng.choice((0, 1, NC - 1, NC, NC + 1))
                keep = np.array([rng.choice((0, 1, 255)) for _ in range(max(k, 1))], np.uint8)
                rc = L.bh_classifier_set_species_list(clf._h, keep.ctypes.data, k)
                if rc == 0: clf.predict_batch_with_context(ctx, [f32[0], f32[1]])
                L.bh_classifier_clear_filters(clf._h)
            elif which == 9:     # BSG calibration tables
                k = rng.choice((0, 1, NC - 1, NC, NC + 1))
                a = np.array([rng.choice(odd) for _ in range(max(k, 1))], np.float32); b = np.array([rng.choice(odd) for _ in range(max(k, 1))], np.float32)
                pr = np.array([rng.choice(odd) for _ in range(max(k, 1))], np.float32)
                rc = L.bh_classifier_set_bsg(clf._h, a.ctypes.data, b.ctypes.data, pr.ctypes.data if rng.random() < 0.5 else None, k)
                if rc == 0: clf.predict_batch_with_context(ctx, [f32[0], f32[1]])
                L.bh_classifier_clear_bsg(clf._h)
            else:
                n_out = C.c_size_t()
                rc = L.bh_resample_output_len(rng.choice(edge), rng.choice((0, 1, 44100, (1 << 32) - 1)), rng.choice((0, 1, 48000, (1 << 32) - 1)), C.byref(n_out))
        except Exception as e:
            rc = "py:" + type(e).__name__
        codes[(which, rc)] = codes.get((which, rc), 0) + 1
    # the classifier still works
    out = clf.predict_logits(ctx, f32)
    assert np.isfinite(out).all()
    print("DONE", dict(sorted(codes.items(), key=str)), flush=True)


def main():
    n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2000, int(sys.argv[2]) if len(sys.argv) > 2 else 1
    bad = 0
    for k in range(0, n, 500):
        try:
            rr = subprocess.run([sys.executable, __file__, "--child", str(seed * 1000 + k), "500"], capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            print("TIMEOUT", k); bad += 1; continue
        if rr.returncode != 0 or "DONE" not in rr.stdout:
            bad += 1; print(f"CRASH rc {rr.returncode} in block {k}\n{rr.stdout[-300:]}\n" + "\n".join(rr.stderr.strip().splitlines()[-14:]))
        else:
            print(rr.stdout.strip().splitlines()[-1][:1500])
    print(f"{n} calls, {bad} bad blocks")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child": child(int(sys.argv[2]), int(sys.argv[3]))
    else: main()
