"""ONNX conv stack -> BHM1 (birda_amd/convert.py), or just list what a graph contains.

  python tools/onnx_to_bhm.py inspect model.onnx
  python tools/onnx_to_bhm.py convert model.onnx frontend.bhm out.bhm [spectrogram tensor name]

`frontend.bhm` carries the model family's front-end (sample rate, STFT / mel branches, mel matrices):
e.g. one written by birda_amd.synth.build_model("birdnet_v24") for the published v2.4 parameters."""
import os, sys
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from birda_amd import convert, onnx_io

if len(sys.argv) >= 3 and sys.argv[1] == "inspect":
    g = onnx_io.load(open(sys.argv[2], "rb").read())
    print(f"producer {g.producer!r} opset {g.opset}: {len(g.nodes)} nodes, {len(g.initializers)} initializers "
          f"({sum(a.size for a in g.initializers.values()) * 4 / 1e6:.1f} MB as f32)")
    print("inputs ", [(v.name, v.shape) for v in g.inputs])
    print("outputs", [(v.name, v.shape) for v in g.outputs])
    for op, n in Counter(n.op_type for n in g.nodes).most_common():
        print(f"  {op:24s} {n}")
elif len(sys.argv) >= 5 and sys.argv[1] == "convert":
    m = convert.convert_file(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    print(f"{sys.argv[4]}: {len(m.layers)} layers, {m.n_classes} classes, {m.macs_per_segment() / 1e6:.1f} M MACs per segment")
else:
    print(__doc__)
    sys.exit(2)
