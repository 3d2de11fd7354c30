import sys, os, time, statistics, json, tempfile
sys.path.insert(0, os.getcwd())
import bench
from birda_amd import modelfile as mf, synth
from birda_amd.classifier import BirdClassifier
tmp = tempfile.mkdtemp()
m = synth.build_model("birdnet_v24"); path = os.path.join(tmp, "m.bhm"); mf.write_model(path, m)
labels = os.path.join(tmp, "bench_labels.txt"); synth.write_labels(labels, m.n_classes)
clf = BirdClassifier(path, labels, precision="auto")
legs = bench.host_legs(clf, m, path, "auto", tmp)
e2e = legs.pop("end_to_end")
print(json.dumps({k: v["value"] for k, v in legs.items()}))
print(json.dumps({k: (v.get("value") if isinstance(v, dict) else v) for k, v in e2e.items() if k != "what"}))
