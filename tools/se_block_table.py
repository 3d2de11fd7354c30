"""The squeeze-excite blocks of the Perch-sized plan, one row each: pass A / gate / gated project GEMM against the gate-free block
(profiles/r5_c_perch_se_blocks.txt).  Input: the outputs of
    python tools/gpu_layer_times.py perch_v2 1000 f16x3 > se.txt;  python tools/gpu_layer_times.py perch_v2_nose 1000 f16x3 > nose.txt
    python tools/se_block_table.py se.txt nose.txt"""
import os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birda_amd import modelfile as mf, synth


def load(path):
    out = {}
    for l in open(path):
        m = re.match(r"layer\s+(\d+)\s+\w+\s+\d+->\s*\d+\s+\d+x\d+\s+k\d+\s+s\d+\s+([\d.]+)", l)
        if m:
            out[int(m[1])] = float(m[2])
    return out


def gated_kernel(K, N, rows, noexp=False):
    if noexp:
        return "fused"      # no expand convolution: depthwise x gate -> project in one launch, D computed again
    nt = -(-N // 16)
    if nt <= 3 and -(-K // 32) * nt * 2048 <= 65536 and rows >= 4096:
        return "thin x2" if K <= 32 else "thin"
    return "wide" if 4 <= nt <= 15 and rows >= 4096 else "staged"


se, nose = load(sys.argv[1]), load(sys.argv[2])
m, m0 = synth.build_model("perch_v2"), synth.build_model("perch_v2_nose")
L, blocks, i = m.layers, [], 0
while i < len(L):
    noexp = L[i].op == mf.OP_DWCONV
    d = i if noexp else i + 1
    if (d + 5 < len(L) and L[d].op == mf.OP_DWCONV and L[d + 1].op == mf.OP_GAP and L[d + 4].op == mf.OP_SCALE and L[d + 5].op == mf.OP_PWCONV):
        blocks.append((i, d, d + 1, d + 5)); i = d + 6
    else:
        i += 1
firsts = sorted(k for k in nose if k < len(m0.layers) - 3)      # gate-free: one fused launch per block, booked on its first layer
print("block  first-layer  image    Cexp->Cout dw k s |  pass A    gate   pass B  (kernel ) |  gate-free block")
ta = tg = tb = tn = 0.0
for b, (i0, d, g, p) in enumerate(blocks):
    D, P = L[d], L[p]
    a, gt, pb = se.get(i0, 0.0), se.get(g, 0.0), se.get(p, 0.0)
    nf = nose.get(firsts[b], 0.0) if b < len(firsts) else 0.0
    ta += a; tg += gt; tb += pb; tn += nf
    print(f"{b + 1:5d} {i0:12d}  {D.out_h}x{D.out_w:<6d} {P.cin:5d}->{P.cout:4d}     {D.kh} {D.sh} | {a:7.1f} {gt:7.1f} {pb:8.1f}  ({gated_kernel(P.cin, P.cout, 1000 * D.out_h * D.out_w, i0 == d):7s}) | {nf:8.1f}")
print(f"\nsum: pass A {ta / 1000:.2f} us per segment, gate {tg / 1000:.2f}, pass B {tb / 1000:.2f}; gate-free fused blocks {tn / 1000:.2f}")
