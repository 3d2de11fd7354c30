"""Register / scratch / LDS use of the kernels in a built object or library, from the code objects' metadata notes
(no recompilation): python tools/kernel_resources.py birda_amd/csrc/_build/kernels_mbconv.o [name filter]"""
import os, re, subprocess, sys, tempfile
llvm = "/opt/rocm/lib/llvm/bin"
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp()
fat = os.path.join(tmp, "fat.bin")
subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, src], check=True, capture_output=True)
data = open(fat, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
starts = [m.start() for m in re.finditer(re.escape(magic), data)]
for k, a in enumerate(starts):
    piece, code = os.path.join(tmp, f"b{k}.bin"), os.path.join(tmp, f"c{k}.o")
    open(piece, "wb").write(data[a:starts[k + 1] if k + 1 < len(starts) else len(data)])
    subprocess.run([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + piece,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + code], check=True, capture_output=True)
    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", code], check=True, capture_output=True, text=True).stdout
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        blk = ".agpr_count:" + blk
        g = lambda key: (re.search(r"\." + key + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name.replace("(anonymous namespace)::", "")).replace("void ", "").replace("bh::", "")
        if flt and flt not in name:
            continue
        print(f"{name[:100]:100s} vgpr {g('vgpr_count'):>3s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} "
              f"threads {g('max_flat_workgroup_size')}")
