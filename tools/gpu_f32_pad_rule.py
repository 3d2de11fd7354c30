"""The f32 planner's padding rule (kernels_mbconv.hip mb_plan: a block whose best entry issues more than 2.5 x its own MFMA steps stays
layer by layer) measured on stacks that trip it: forward time of N segments in f32 mode with the rule as shipped and switched off
(BIRDA_HIP_MB_F32_PAD, EXPERIMENTS build: LIBX=1 tools/ab.sh x python tools/gpu_f32_pad_rule.py ...).
    python tools/gpu_f32_pad_rule.py <seed> [n]        (one process per setting: the factor is read once)"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3:          # child: seed n factor
    import numpy as np, torch
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    seed, N = int(sys.argv[1]), int(sys.argv[2])
    m = synth.build_model("custom", plan=synth.random_plan(seed, big=seed >= 1000))
    path = f"/tmp/_pad{seed}.bhm"; mf.write_model(path, m)
    clf = BirdClassifier(path, precision="f32")
    nb = sum(1 for L in m.layers if L.op == mf.OP_DWCONV)
    ctx = clf.create_batch_context(N)
    base = synth.synth_segments(8, m.sample_count, m.sample_rate)
    x = torch.from_numpy(np.tile(base, (N // 8 + 1, 1))[:N]).cuda()
    logits = torch.empty((N, m.n_classes), device="cuda"); idx = torch.empty((N, 5), dtype=torch.int32, device="cuda"); conf = torch.empty((N, 5), device="cuda")
    for _ in range(3): clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    import time
    t = time.perf_counter()
    for _ in range(10): clf.forward_device(ctx, x.data_ptr(), N, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
    ctx.synchronize()
    dt = (time.perf_counter() - t) / 10
    print(f"seed {seed} factor {sys.argv[3]:>5s}: {len(clf.fused_blocks())} of {nb} blocks fused, {dt * 1e3:8.3f} ms per {N} segments", flush=True)
else:
    seed, N = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "256"
    for f in ("2.5", "1000"):
        env = dict(os.environ, BIRDA_HIP_MB_F32_PAD=f)
        subprocess.run([sys.executable, __file__, seed, N, f], env=env)
