#!/usr/bin/env python3
"""bench.py -- 3 s / 48 kHz segments per second through the MI355X hot path (BirdNET v2.4 shape).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run, one rank per GPU over RCCL.  One "step" = one pass of the hot
path (min/max -> STFT*mel -> conv stack -> logits -> sigmoid/top-k) over 1 000 synthetic
segments per GPU that are already resident in HBM (BASELINE.json configs[1]); with N > 1 each
rank owns its own 1 000-segment shard (weak scaling) and the only collective is the gather
of the top-k results to rank 0.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEGMENTS_PER_GPU = 1000
MEL_BYTES_PER_SEGMENT = 968_448          # 576 000 B read + 392 448 B written
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_HBM_GBPS = 8000.0


def cpu_baseline(model_path, sample_count, sample_rate):
    """The oracle (a port, not the reference: the reference's ORT path cannot run here) timed
    on this box's host cores over a bounded sample of the same synthetic workload."""
    import numpy as np
    from birda_amd import synth
    from oracle import oracle as O

    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    om = O.OracleModel(model_path)
    n = max(32, min(2 * cores, 256))
    segs = synth.synth_segments(min(n, 16), sample_count, sample_rate)
    segs = np.tile(segs, (n // segs.shape[0] + 1, 1))[:n]
    om.forward(segs[: min(cores, n)])  # touch code/pages once
    t = time.perf_counter()
    om.forward(segs)
    dt = time.perf_counter() - t
    return {"value": round(n / dt, 2), "unit": "segments/s", "cores": cores, "kind": "port",
            "sample": f"{n} synthetic 3 s/48 kHz segments, oracle/birda_oracle.c, OpenMP across segments, fp32"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--micro-batch", type=int, default=int(os.environ.get("BIRDA_HIP_MICRO_BATCH", "256")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from birda_amd import modelfile as mf, sharding, synth
    from birda_amd.classifier import BirdClassifier

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # seeded synthetic BirdNET-v2.4-shaped model (no real weights exist offline)
    tmp = tempfile.mkdtemp(prefix=f"birda_bench_r{rank}_")
    model_path = os.path.join(tmp, "birdnet_v24_synth.bhm")
    m = synth.build_model("birdnet_v24")
    mf.write_model(model_path, m)
    clf = BirdClassifier(model_path, None, top_k=5, min_confidence=0.1, device=local_rank)
    ctx = clf.create_batch_context(args.micro_batch)
    info = clf.info

    n_local = SEGMENTS_PER_GPU
    n_total = n_local * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    # segment i of the global list; 64 distinct seeds tiled keeps host prep short
    uniq = synth.synth_segments(64, m.sample_count, m.sample_rate, start=0)
    host = np.stack([uniq[(lo + j) % 64] for j in range(n_local)])
    x = torch.from_numpy(host).cuda()
    logits = torch.empty((n_local, m.n_classes), device="cuda")
    tk_idx = torch.empty((n_local, 5), dtype=torch.int32, device="cuda")
    tk_conf = torch.empty((n_local, 5), device="cuda")

    def step():
        clf.forward_device(ctx, x.data_ptr(), n_local, logits.data_ptr(), tk_idx.data_ptr(), tk_conf.data_ptr())
        if world > 1:
            ctx.synchronize()
            packed = torch.cat([tk_idx.to(torch.float32), tk_conf], 1)
            sharding.gather_results(packed, n_total, rank, world)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()

    ctx.set_profiling(True)
    stage_tot = {}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, (ms, n) in ctx.stage_ms().items():   # HIP events on the context stream
            a = stage_tot.setdefault(k, [0.0, 0])
            a[0] += ms
            a[1] += n
    sync_all()
    elapsed = time.perf_counter() - t0
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    value = n_total * args.steps / elapsed
    segs_done = n_local * args.steps
    pw_ms, pw_launches = stage_tot["pointwise"]
    mel_ms, mel_launches = stage_tot["mel"]
    pw_macs = sum(L.out_h * L.out_w * L.cin * L.cout for L in m.layers if L.op == mf.OP_PWCONV)
    pw_tflops = 2.0 * pw_macs * segs_done / (pw_ms * 1e-3) / 1e12
    mel_gbps = MEL_BYTES_PER_SEGMENT * segs_done / (mel_ms * 1e-3) / 1e9
    out = {
        "metric": "3s/48kHz segments/sec (BirdNET v2.4)", "value": round(value, 1), "unit": "segments/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: 1000 synthetic 3 s/48 kHz segments per GPU per step, HBM-resident, "
                               "seeded synthetic BirdNET-v2.4-shaped model (EfficientNet-B0-like, 6522 classes)",
                   "segments_per_gpu": n_local, "micro_batch": args.micro_batch,
                   "gflop_per_segment": round((2 * info.macs_per_segment + info.mel_flops_per_segment) / 1e9, 3)},
        "roofline": {"kernel": "pw_gemm_kernel (pointwise 1x1 conv, v_mfma_f32_16x16x4_f32)", "bound": "mfma",
                     "achieved": round(pw_tflops, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(pw_tflops / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                     "launches": pw_launches, "avg_launch_us": round(pw_ms * 1e3 / max(pw_launches, 1), 2)},
        "roofline_mel": {"kernel": "mel_kernel (folded STFT x mel, v_mfma_f32_16x16x4_f32)", "bound": "hbm",
                         "achieved": round(mel_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(mel_gbps / PEAK_HBM_GBPS, 4), "traffic": None,
                         "launches": mel_launches, "avg_launch_us": round(mel_ms * 1e3 / max(mel_launches, 1), 2),
                         "mfma_tflops": round(info.mel_flops_per_segment * segs_done / (mel_ms * 1e-3) / 1e12, 2)},
        "stage_us_per_segment": {k: round(v[0] * 1e3 / segs_done, 3) for k, v in stage_tot.items()},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model_path, m.sample_count, m.sample_rate)
    elif rank == 0:
        out["cpu_baseline"] = None
    ctx.close()
    clf.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
