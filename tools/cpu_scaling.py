"""How the oracle's OpenMP leg scales on this host (what `cpu_baseline.cores` should be): python tools/cpu_scaling.py"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import numpy as np
    from birda_amd import modelfile as mf, synth
    from oracle import oracle as O
    m = synth.build_model("birdnet_v24"); p = "/tmp/cpu_scaling.bhm"
    if not os.path.exists(p): mf.write_model(p, m)
    om = O.OracleModel(p)
    n = int(sys.argv[1])
    segs = np.tile(synth.synth_segments(16, m.sample_count, m.sample_rate), (n // 16 + 1, 1))[:n]
    om.forward(segs[:min(n, 16)])
    t = time.perf_counter(); om.forward(segs); dt = time.perf_counter() - t
    print(f"OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}: {n} segments in {dt:.2f} s = {n / dt:.1f} segments/s = {n / dt / int(os.environ['OMP_NUM_THREADS']):.2f} per thread")
else:
    import bench
    print("usable_cores:", bench.usable_cores())
    print(open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0] if os.path.exists("/proc/cpuinfo") else "")
    print("loadavg", open("/proc/loadavg").read().strip())
    for t in (8, 16, 32, 64, 128, 256):
        if t > (os.cpu_count() or 1): break
        subprocess.run([sys.executable, __file__, str(max(64, 2 * t))], env=dict(os.environ, OMP_NUM_THREADS=str(t), OMP_PROC_BIND="false"))
