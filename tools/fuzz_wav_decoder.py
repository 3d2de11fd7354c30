"""Byte-level fuzz of the host WAV decoder (bhh_decoder_open / _next_segment: the reference's StreamingDecoder, src/audio/decode.rs:54-
411, restated in host_pipeline.cpp) -- CPU only.  A recording is untrusted input: whatever the bytes, open and every next_segment
must RETURN (segments, end of stream, or an error with a message), never crash, hang or read outside the file.  Mutations of small
valid files in every sample format (PCM 16 / 24 / 32, float32, mono / stereo, WAVE_FORMAT_EXTENSIBLE, a LIST chunk before `data`):
byte flips and overwrites (mostly in the header), truncations, insertions, spliced ranges.  Child processes: a crash is a signal.
    python tools/fuzz_wav_decoder.py [n_mutants] [seed]
    FUZZ_PROCESS=1 python tools/fuzz_wav_decoder.py ...      (GPU box: through bhh_process_file, device and host front ends)"""
import os, random, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(paths):
    import ctypes as C
    import numpy as np
    from birda_amd import _lib
    L = _lib.load()
    buf = np.empty(9000, np.float32)
    for p in paths:
        h = C.c_void_p()
        rc = L.bhh_decoder_open(p.encode(), C.byref(h))
        segs = 0
        if rc == 0:
            L.bhh_decoder_sample_rate(h)
            d = C.c_double(); L.bhh_decoder_duration_hint(h, C.byref(d))
            start = C.c_size_t()
            while segs < 10000:
                rc = L.bhh_decoder_next_segment(h, 9000, 3000, buf.ctypes.data, C.byref(start))
                if rc <= 0:
                    break
                segs += 1
            L.bhh_decoder_close(h)
        print(os.path.basename(p), rc, segs, flush=True)


def child_process(paths):
    """GPU box: the same mutants through bhh_process_file with the DEVICE front end (the file is mapped, its samples are uploaded in
    the file's own layout and segmented on the device: the header's sizes decide what is read)"""
    import tempfile as tf
    from birda_amd import modelfile as mf, pipeline, synth
    from birda_amd._lib import BirdaHipError
    from birda_amd.classifier import BirdClassifier
    d = tf.mkdtemp()
    m = synth.build_model("mini")
    mp, lp = os.path.join(d, "m.bhm"), os.path.join(d, "l.txt")
    mf.write_model(mp, m); synth.write_labels(lp, m.n_classes)
    clf = BirdClassifier(mp, lp, top_k=3, min_confidence=0.5)
    # ... and all of them at once through bhh_process_files (packed uploads: a file's header decides where its frames sit in the pack)
    try:
        res, status = pipeline.process_files_packed(clf, list(paths), d, min_confidence=0.5, overlap=0.0)
        print("# packed:", sum(1 for s_ in status if s_ == 0), "of", len(paths), "processed", file=sys.stderr, flush=True)
    except Exception as e:
        print("# packed call failed:", type(e).__name__, str(e)[:200], file=sys.stderr, flush=True)
    for p in paths:
        for fe in ("device", "host"):
            try:
                r = pipeline.process_file(clf, p, d, min_confidence=0.5, overlap=0.0, front_end=fe)
                rc, segs = 0, r.segments
            except BirdaHipError as e:
                rc, segs = e.code, 0
                if rc not in (-2, -6): print("#", os.path.basename(p), fe, str(e)[:300], file=sys.stderr, flush=True)
            except Exception as e:          # (the host layer's own error type)
                rc, segs = getattr(e, "code", -1), 0
            print(os.path.basename(p) + ":" + fe, rc, segs, flush=True)


def wav(x, rate, ch, fmt, extensible=False, list_chunk=False):
    import numpy as np
    x = np.asarray(x, np.float64)
    if fmt == "f32": data, tag, bits = x.astype("<f4").tobytes(), 3, 32
    elif fmt == "s32": data, tag, bits = np.round(x * 2147483647.0).astype("<i4").tobytes(), 1, 32
    elif fmt == "s24": data, tag, bits = np.round(x * 8388607.0).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes(), 1, 24
    else: data, tag, bits = np.round(x * 32767.0).astype("<i2").tobytes(), 1, 16
    if extensible:
        guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt_body = struct.pack("<HHIIHH", 0xFFFE, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits) + struct.pack("<HHI", 22, bits, 0) + guid
    else:
        fmt_body = struct.pack("<HHIIHH", tag, ch, rate, rate * ch * bits // 8, ch * bits // 8, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body
    if list_chunk:
        chunks += b"LIST" + struct.pack("<I", 11) + b"INFOabcdefg" + b"\x00"     # (odd size: a pad byte follows)
    chunks += b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks


def main():
    import numpy as np
    n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 600, int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    r = np.random.default_rng(seed)
    d = tempfile.mkdtemp()
    bases = []
    for fmt in ("s16", "s24", "s32", "f32"):
        for ch in (1, 2):
            x = r.uniform(-0.9, 0.9, size=20000 * ch)
            bases.append(wav(x, 48000, ch, fmt, extensible=(ch == 2), list_chunk=(fmt in ("s24", "f32"))))
    paths = []
    for i in range(n):
        b = bytearray(rng.choice(bases))
        for _ in range(rng.choice((1, 1, 2, 3))):
            kind = rng.randrange(6)
            pos = rng.randrange(min(len(b), 120)) if rng.random() < 0.8 else rng.randrange(len(b))
            if kind == 0: b[pos] ^= 1 << rng.randrange(8)
            elif kind == 1: b[pos] = rng.choice((0, 1, 0x7f, 0x80, 0xff, rng.randrange(256)))
            elif kind == 2: b = b[:rng.randrange(len(b))]
            elif kind == 3: b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
            elif kind == 4: b[pos:pos + 4] = struct.pack("<I", rng.choice((0, 1, 2, 0xffffffff, 0x7fffffff, 0x80000000, rng.randrange(1 << 32))))
            else:
                q = rng.randrange(len(b)); b[pos:pos + 8] = b[q:q + 8]
            if not b: b = bytearray(b"R")
        p = os.path.join(d, f"w{i:05d}.wav")
        open(p, "wb").write(bytes(b)); paths.append(p)
    bad, codes, decoded = 0, {}, 0
    for i in range(0, n, 25):
        batch = paths[i:i + 25]
        try:
            rr = subprocess.run([sys.executable, __file__, "--child"] + batch, capture_output=True, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            print("TIMEOUT in batch", i); bad += 1; continue
        lines = [l.split() for l in rr.stdout.split("\n") if l.strip()]
        for l in rr.stderr.split("\n"):
            if l.startswith("#"): print(l)
        for l in lines:
            codes[l[1]] = codes.get(l[1], 0) + 1; decoded += int(l[2]) > 0
        if rr.returncode != 0 or len(lines) != len(batch) * (2 if os.environ.get("FUZZ_PROCESS") else 1):
            bad += 1
            print(f"CRASH rc {rr.returncode} after {len(lines)} of batch {i}: {batch[len(lines)] if len(lines) < len(batch) else '?'}\n{rr.stderr[-400:]}")
    print(f"{n} mutants, {bad} bad batches; last return codes {dict(sorted(codes.items()))}; {decoded} gave segments" + (f"; files kept in {d}" if bad else ""))
    if not bad:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child": (child_process if os.environ.get("FUZZ_PROCESS") else child)(sys.argv[2:])
    else: main()
