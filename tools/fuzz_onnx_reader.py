"""Byte-level fuzz of the library's model readers (FUZZ_BHM=1: also the flat BHM1 container through bh_plan_fused_blocks; bh_onnx_to_bhm: protobuf walk, conv-stack reader, front-end recovery with its
float64 evaluator) -- CPU only.  A model file is untrusted input: whatever the bytes, the call must RETURN (an error code and a
message), never crash or hang.  Mutations of a small valid file: byte flips, varint bumps, truncations, spliced ranges.  Each batch
runs in a child process so that a crash is seen as a signal, and every mutant has a time limit.
    python tools/fuzz_onnx_reader.py [n_mutants] [seed]"""
import os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(paths):
    from birda_amd import _lib
    L = _lib.load()
    import ctypes as C
    cfgs, layers = (C.c_int32 * 1024)(), (C.c_int32 * 1024)()
    for p in paths:
        if p.endswith(".bhm"):   # the flat container's reader + the planner (host logic: no GPU)
            rc = L.bh_plan_fused_blocks(p.encode(), 1, cfgs, layers, 1024)
            rc = min(rc, 0) if rc < 0 else 0
        else:
            rc = L.bh_onnx_to_bhm(p.encode(), (p + ".out.bhm").encode())
        print(os.path.basename(p), rc, flush=True)

def main():
    from birda_amd import convert, onnx_io as ox, synth
    n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    d = tempfile.mkdtemp()
    bases = []
    for k, sp in enumerate(("conv1d", "stft", "complex", "fused")):
        m = synth.build_model("custom", plan=synth.random_plan(k))
        bases.append((".onnx", ox.dump(convert.graph_from_model(m, frontend_spelling=sp))))
        if os.environ.get("FUZZ_BHM"):
            from birda_amd import modelfile as mf
            q = os.path.join(d, f"base{k}.bhm"); mf.write_model(q, m); bases.append((".bhm", open(q, "rb").read()))
    paths = []
    for i in range(n):
        ext, raw = rng.choice(bases)
        b = bytearray(raw)
        for _ in range(rng.choice((1, 1, 2, 4))):
            kind = rng.randrange(5)
            # (the first 3 KB hold most of the structure: node list, attributes, tensor headers; weights fill the rest)
            pos = rng.randrange(min(len(b), 3000)) if rng.random() < 0.7 else rng.randrange(len(b))
            if kind == 0: b[pos] ^= 1 << rng.randrange(8)
            elif kind == 1: b[pos] = rng.randrange(256)
            elif kind == 2: b = b[:pos]
            elif kind == 3: b[pos:pos] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
            else:
                q = rng.randrange(len(b)); b[pos:pos + 8] = b[q:q + 8]
            if not b: b = bytearray(b"\x00")
        p = os.path.join(d, f"m{i:05d}{ext}")
        open(p, "wb").write(bytes(b)); paths.append(p)
    bad = 0
    codes = {}
    for i in range(0, n, 20):
        batch = paths[i:i + 20]
        try:
            r = subprocess.run([sys.executable, __file__, "--child"] + batch, capture_output=True, text=True, timeout=600)
        except subprocess.TimeoutExpired:
            print("TIMEOUT in batch", i); bad += 1; continue
        done = len(r.stdout.strip().splitlines())
        for l in r.stdout.split("\n"):
            if l.strip(): codes[l.split()[1]] = codes.get(l.split()[1], 0) + 1
        if r.returncode != 0 or done != len(batch):
            bad += 1
            print(f"CRASH rc {r.returncode} after {done} of batch {i}: {batch[done] if done < len(batch) else '?'}\n{r.stderr[-400:]}")
    print(f"{n} mutants, {bad} bad batches; return codes {dict(sorted(codes.items()))}" + (f"; files kept in {d}" if bad else ""))
    if not bad:
        import shutil
        shutil.rmtree(d, ignore_errors=True)

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child": child(sys.argv[2:])
    else: main()
