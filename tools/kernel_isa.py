"""Static instruction counts of ONE kernel of a built object (no recompilation): vector / MFMA / scalar / LDS / memory / wait /
branch instructions, in the whole kernel and -- for the fused MBConv kernel -- in its three regions (set-up before the chunk loop,
the chunk loop, the epilogue), split at the loop's back edge.

    python tools/kernel_isa.py birda_amd/csrc/_build/kernels_mbconv_gelu.o "mbconv_kernel<3, 2, 16, 1, 5, 1, 4, 1, 1, 2, 4, 0, 1, 4, 0, 3, 0, 4, 0, 0>"
"""
import os, re, subprocess, sys, tempfile
llvm = "/opt/rocm/lib/llvm/bin"
src, want = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, src], check=True, capture_output=True)
data = open(fat, "rb").read()
starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), data)]
for k, a in enumerate(starts):
    piece, code = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"c{k}.o")
    open(piece, "wb").write(data[a:starts[k + 1] if k + 1 < len(starts) else len(data)])
    subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + piece,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + code], check=True, capture_output=True)
    syms = subprocess.run([f"{llvm}/llvm-readelf", "--symbols", "--wide", code], capture_output=True, text=True).stdout.split("\n")
    names = [l.split()[-1] for l in syms if " FUNC " in l]
    for mangled in names:
        dem = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").replace("bh::", "")
        if want not in dem:
            continue
        dis = subprocess.run([f"{llvm}/llvm-objdump", "-d", f"--disassemble-symbols={mangled}", code], capture_output=True, text=True).stdout
        ins = []
        for l in dis.splitlines():
            m = re.match(r"\s+([a-z_0-9]+)\s", l)
            if m and not l.strip().startswith("//"):
                ins.append((m.group(1), l))
        def cat(op):
            if op.startswith("v_mfma"): return "mfma"
            if op.startswith("v_"): return "valu"
            if op.startswith("ds_"): return "lds"
            if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
            if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
            if op.startswith(("s_cbranch", "s_branch")): return "branch"
            if op.startswith("s_barrier"): return "barrier"
            if op.startswith("s_"): return "salu"
            return "other"
        def count(seq):
            c = {}
            for op, _ in seq:
                c[cat(op)] = c.get(cat(op), 0) + 1
            return c
        tot = count(ins)
        print(dem[:150])
        print("  all      :", len(ins), " ".join(f"{k} {v}" for k, v in sorted(tot.items())))
        # the chunk loop: the LAST backward branch that spans MFMAs -- from its target to the branch
        addr = []
        for op, l in ins:
            m = re.search(r"//\s*([0-9A-Fa-f]+):", l)
            addr.append(int(m.group(1), 16) if m else -1)
        best = None
        for i, (op, l) in enumerate(ins):
            if op.startswith("s_cbranch") or op == "s_branch":
                m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", l)
                if not m: continue
                tgt = int(m.group(1), 16)
                base = addr[0]
                j = next((q for q, a_ in enumerate(addr) if a_ - base == tgt or a_ == tgt), None)
                if j is not None and j < i and any(o.startswith("v_mfma") for o, _ in ins[j:i]) and sum(1 for o, _ in ins[j:i] if o == "s_barrier") >= 2:
                    if best is None or (i - j) > (best[1] - best[0]):
                        best = (j, i)
        if best:
            j, i = best
            for lab, seq in (("set-up", ins[:j]), ("chunk loop", ins[j:i + 1]), ("epilogue", ins[i + 1:])):
                c = count(seq)
                print(f"  {lab:9s}:", len(seq), " ".join(f"{k} {v}" for k, v in sorted(c.items())))
        sys.exit(0)
print("kernel not found")
