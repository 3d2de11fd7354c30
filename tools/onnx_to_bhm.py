"""ONNX conv stack -> BHM1 (birda_amd/convert.py), or just list what a graph contains.

  python tools/onnx_to_bhm.py inspect model.onnx
  python tools/onnx_to_bhm.py convert model.onnx frontend.bhm out.bhm [spectrogram tensor name]
  python tools/onnx_to_bhm.py convert model.onnx --sample-rate 48000 out.bhm [--family 0]
  python tools/onnx_to_bhm.py frontend model.onnx --sample-rate 48000        (only report what the probing finds)

`frontend.bhm` carries the model family's front-end (sample rate, STFT / mel branches, mel matrices):
e.g. one written by birda_amd.synth.build_model("birdnet_v24") for the published v2.4 parameters.
With --sample-rate instead, the graph must start at the audio input and the front-end is read off it by probing
(birda_amd/frontend_recover.py; about half a minute for a BirdNET-sized front-end)."""
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
elif len(sys.argv) >= 5 and sys.argv[1] == "frontend" and sys.argv[3] == "--sample-rate":
    from birda_amd.frontend_recover import recover_frontend
    rec = recover_frontend(onnx_io.load(open(sys.argv[2], "rb").read()), int(sys.argv[4]))
    for k, v in rec.report.items():
        print(f"  {k}: {v}")
elif len(sys.argv) >= 6 and sys.argv[1] == "convert" and sys.argv[3] == "--sample-rate":
    fam = int(sys.argv[7]) if len(sys.argv) > 7 and sys.argv[6] == "--family" else 0
    m = convert.convert_file(sys.argv[2], None, sys.argv[5], sample_rate=int(sys.argv[4]), family=fam)
    print(f"{sys.argv[5]}: {len(m.layers)} layers, {m.n_classes} classes, {m.macs_per_segment() / 1e6:.1f} M MACs per segment")
    for b in m.branches:
        print(f"  branch: frame length {b.frame_length}, step {b.frame_step}, {b.n_mels} mels x {b.n_frames} frames, "
              f"mag_scale {b.mag_scale:.4f}, affine {b.out_scale:.4g} x + {b.out_shift:.4g}, flip {b.flags & 1}")
elif len(sys.argv) >= 5 and sys.argv[1] == "convert":
    m = convert.convert_file(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else None)
    print(f"{sys.argv[4]}: {len(m.layers)} layers, {m.n_classes} classes, {m.macs_per_segment() / 1e6:.1f} M MACs per segment")
else:
    print(__doc__)
    sys.exit(2)
