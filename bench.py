#!/usr/bin/env python3
"""bench.py -- 3 s / 48 kHz segments per second through the MI355X hot path (BirdNET v2.4 shape).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run, one rank per GPU over RCCL.  One "step" = one pass of the hot
path (min/max -> STFT*mel -> stem -> fused MBConv blocks -> head -> logits -> sigmoid/top-k)
over 1 000 synthetic segments per GPU that are already resident in HBM (BASELINE.json
configs[1]); with N > 1 each rank owns its own 1 000-segment shard (weak scaling) and the only
collective is the gather of the top-k results to rank 0.  Prints ONE JSON line on rank 0.

`roofline` is the dominant kernel (largest total time in the timed region): one of the fused
MBConv kernels, priced in ALGORITHMIC flops (2 x MACs of the block's expand + depthwise +
project convolutions, no halo / padding work) against the f32 MFMA peak; its launch duration
comes from HIP events recorded on the context stream around every launch of the timed steps.
`roofline_mel` is the front-end kernel against the HBM roofline (968 448 B per
segment).  `cpu_baseline` is the oracle timed on this box's host cores over a bounded sample.
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
PEAK_F16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak
PEAK_HBM_GBPS = 8000.0


def cpu_baseline(model_path, sample_count, sample_rate, hip_logits=None):
    """The oracle (a port, not the reference: the reference's ORT path cannot run here) timed
    on this box's host cores over a bounded sample of the same synthetic workload.  As the checker it
    also gives BASELINE's second figure, max |dlogit| of the timed HIP path on the first segments."""
    import numpy as np
    from birda_amd import synth
    from oracle import oracle as O

    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    om = O.OracleModel(model_path)
    base = synth.synth_segments(min(cores, 16), sample_count, sample_rate)
    ref = om.forward(base[: min(cores, base.shape[0])])  # touch code / pages once; reference logits of segments 0..
    parity = None
    if hip_logits:
        parity = {}
        for name, got in hip_logits.items():
            k = min(len(ref), len(got))
            scale = float(max(1.0, np.abs(ref[:k]).max()))
            d = float(np.abs(got[:k] - ref[:k]).max())
            parity[name] = {"max_abs_dlogit": round(d, 6), "max_abs_logit": round(scale, 3), "relative": float(f"{d / scale:.3e}"),
                            "segments": k, "top1_agree": bool((got[:k].argmax(1) == ref[:k].argmax(1)).all())}
    n = int(min(1024, max(64, 4 * cores)))         # ~10-30 s of CPU work on 8 ... 256 cores
    segs = np.tile(base, (n // base.shape[0] + 1, 1))[:n]
    t = time.perf_counter()
    om.forward(segs)
    dt = time.perf_counter() - t
    # the true reference is ONNX Runtime (ORT_DYLIB_PATH, constants.rs:547) running the published birdnet.onnx.
    ort, onnx = os.environ.get("ORT_DYLIB_PATH", ""), os.environ.get("BIRDA_REFERENCE_ONNX", "")
    if ort and onnx and os.path.exists(ort) and os.path.exists(onnx):
        ref_note = ("ONNX Runtime and a reference model are present on this box, but this build carries no ORT binding yet "
                    "(DESIGN.md section 8, lead 4): compared against the CPU restatement")
    else:
        ref_note = "reference ORT path unavailable (no ORT_DYLIB_PATH / birdnet.onnx on this box): compared against the CPU restatement"
    return {"value": round(n / dt, 2), "unit": "segments/s", "cores": cores, "kind": "port", "reference": ref_note,
            "sample": f"{n} synthetic 3 s/48 kHz segments, oracle/birda_oracle.c, OpenMP across segments "
                      f"({cores} threads), fp32, {dt:.1f} s",
            "max_abs_dlogit_vs_oracle": parity}


def layers_have_fused_stem(m, layer_tot):
    """True when layer 0 (the stem conv) heads a fused block: it was launched and layers 1, 2 were not."""
    from birda_amd import modelfile as mf
    L = m.layers
    return (len(L) > 2 and L[0].op == mf.OP_CONV and L[1].op == mf.OP_DWCONV and L[2].op == mf.OP_PWCONV
            and layer_tot[0][1] > 0 and layer_tot[1][1] == 0 and layer_tot[2][1] == 0)


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (profiles/*_traffic.json,
    collected with tools/profile_round.sh on this same command: two rocprofv3 --pmc passes, FETCH_SIZE
    doubled for gfx950 as MI355X_MICROARCH.md prescribes).  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except (OSError, ValueError):
        return None
    for k, v in d.items():
        if k.startswith(kernel_prefix):
            return {"bytes_per_launch": v["hbm_bytes_per_launch"], "read": v["read_bytes_per_launch"],
                    "write": v["write_bytes_per_launch"], "source": os.path.basename(files[-1])}
    return None


def analyse(clf, m, info, fused, stage_tot, layer_tot, segs_done, steps, slices_per_step, precision):
    """Roofline objects from the HIP-event timings of one timed region."""
    from birda_amd import modelfile as mf

    def macs(L):
        px = L.out_h * L.out_w
        if L.op == mf.OP_CONV:
            return px * L.kh * L.kw * L.cin * L.cout
        if L.op == mf.OP_DWCONV:
            return px * L.kh * L.kw * L.cout
        if L.op in (mf.OP_PWCONV, mf.OP_DENSE):
            return px * L.cin * L.cout
        return 0

    # launches grouped by kernel: fused blocks of equal shape share one instantiation
    groups = {}
    layers = m.layers
    n_stem_blocks = 1 if (fused and layers_have_fused_stem(m, layer_tot)) else 0
    bi, i = 0, 0
    while fused and i + 2 < len(layers) and bi + n_stem_blocks < len(fused):
        E, D, P = layers[i], layers[i + 1], layers[i + 2]
        if (E.op == mf.OP_PWCONV and D.op == mf.OP_DWCONV and P.op == mf.OP_PWCONV and D.in_tensor == i + 1
                and P.in_tensor == i + 2 and layer_tot[i][1] > 0 and layer_tot[i + 1][1] == 0):
            key = (E.cin, E.cout, P.cout, D.kh, D.sh, E.in_h, E.in_w)
            g = groups.setdefault(key, {"ms": 0.0, "launches": 0, "macs": macs(E) + macs(D) + macs(P),
                                        "kernel": clf.fused_kernel_name(fused[bi + n_stem_blocks])})
            g["ms"] += layer_tot[i][0]
            g["launches"] += layer_tot[i][1]
            bi += 1
            i += 3
        else:
            i += 1
    out = {}
    mb_ms, mb_launches = stage_tot.get("mbconv", (0.0, 0))
    if groups:
        dom_key, dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
        # this kernel runs n_blocks equal-shaped blocks per slice; every block sees every segment once
        n_blocks = max(1, dom["launches"] // (steps * slices_per_step))
        # the instantiation's last template argument is its MFMA type: 0 = f32, 3 = split f16 (x3), 1 = f16
        dom_prec = int(dom["kernel"].rstrip(">").split(",")[-1])
        if dom_prec == 0:
            peak, note, insn = PEAK_F32_MFMA_TFLOPS, "dense f32 MFMA peak (runs at the vector rate)", "v_mfma_f32_16x16x4_f32"
        elif dom_prec == 3:
            peak, note, insn = (PEAK_F16_MFMA_TFLOPS / 3.0, "dense f16 MFMA peak / 3: three MFMAs per f32-grade product; the "
                                "kernel is bound by its vector work (GELU, depthwise taps), see DESIGN.md",
                                "3 x v_mfma_f32_16x16x32_f16 per product")
        else:
            peak, note, insn = PEAK_F16_MFMA_TFLOPS, "dense f16 MFMA peak", "v_mfma_f32_16x16x32_f16"
        total_flops = 2.0 * dom["macs"] * segs_done * n_blocks
        tflops = total_flops / (dom["ms"] * 1e-3) / 1e12
        out["roofline"] = {
            "kernel": "mbconv_kernel (fused expand 1x1 -> depthwise %dx%d s%d -> project 1x1, Cin %d -> %d -> %d at %dx%d, %s)"
                      % (dom_key[3], dom_key[3], dom_key[4], dom_key[0], dom_key[1], dom_key[2], dom_key[5], dom_key[6], insn),
            "bound": "mfma", "achieved": round(tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(tflops / peak, 4), "peak_note": note, "traffic": pmc_traffic(dom["kernel"]),
            "rocprof_name": "bh::mbconv_kernel<" + dom["kernel"][len("mbconv<"):],
            "launches": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
            "algorithmic_gflop_per_launch": round(total_flops / max(dom["launches"], 1) / 1e9, 3)}
        if mb_ms > 0:
            flops = 2.0 * sum(g["macs"] * max(1, g["launches"] // (steps * slices_per_step)) for g in groups.values()) * segs_done
            out["all_fused_blocks"] = {"achieved": round(flops / (mb_ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s (algorithmic)",
                                       "launches": mb_launches, "us_per_segment": round(mb_ms * 1e3 / segs_done, 3)}
    mel_ms, mel_launches = stage_tot["mel"]
    mel_gbps = MEL_BYTES_PER_SEGMENT * segs_done / (mel_ms * 1e-3) / 1e9
    out["roofline_mel"] = {"kernel": "mel_kernel (folded STFT x mel, %s)" % (
                               "split f16 x3 MFMA" if (precision != "f32" and os.environ.get("BIRDA_HIP_MEL_F32") != "1")
                               else "v_mfma_f32_16x16x4_f32"), "bound": "hbm",
                           "achieved": round(mel_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                           "frac": round(mel_gbps / PEAK_HBM_GBPS, 4), "traffic": pmc_traffic("bh::mel_kernel"),
                           "launches": mel_launches, "avg_launch_us": round(mel_ms * 1e3 / max(mel_launches, 1), 2),
                           "mfma_tflops": round(info.mel_flops_per_segment * segs_done / (mel_ms * 1e-3) / 1e12, 2)}
    out["stage_us_per_segment"] = {k: round(v[0] * 1e3 / segs_done, 3) for k, v in stage_tot.items()}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--micro-batch", type=int, default=int(os.environ.get("BIRDA_HIP_MICRO_BATCH", "1000")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default=os.environ.get("BIRDA_HIP_BENCH_PRECISION", "f16x3"),
                    choices=["f32", "f16x3", "f16"],
                    help="GEMM operands: f16x3 = f32 values split into f16 hi + lo, three f16 MFMAs per product, "
                         "f32 accumulate (same fp32 logit tolerance as f32); f32 = v_mfma_f32_16x16x4_f32 everywhere")
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
    # (dry-run aid for a 1-GPU box: BIRDA_BENCH_DRYRUN_ONE_DEVICE=1 puts every rank on cuda:0 and uses gloo,
    #  which exercises the launch / sharding / timing / reporting flow; the driver's runs use RCCL)
    dryrun = world > 1 and os.environ.get("BIRDA_BENCH_DRYRUN_ONE_DEVICE") == "1"
    if dryrun:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if dryrun:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    # seeded synthetic BirdNET-v2.4-shaped model (no real weights exist offline)
    tmp = tempfile.mkdtemp(prefix=f"birda_bench_r{rank}_")
    model_path = os.path.join(tmp, "birdnet_v24_synth.bhm")
    m = synth.build_model("birdnet_v24")
    mf.write_model(model_path, m)
    clf = BirdClassifier(model_path, None, top_k=5, min_confidence=0.1, device=local_rank, precision=args.precision)
    ctx = clf.create_batch_context(args.micro_batch)
    info = clf.info
    fused = clf.fused_blocks()

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

    # The result gather runs on torch's stream; the forward runs on the context's own HIP stream.  Order them with
    # events (stream-to-stream, no host synchronise inside a step) when torch can wrap the context stream.
    ctx_stream = None
    if world > 1:
        try:
            ctx_stream = torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda", local_rank))
        except Exception:   # noqa: BLE001 -- fall back to a host synchronise per step
            ctx_stream = None

    def step():
        clf.forward_device(ctx, x.data_ptr(), n_local, logits.data_ptr(), tk_idx.data_ptr(), tk_conf.data_ptr())
        if world > 1:
            cur = torch.cuda.current_stream()
            if ctx_stream is not None:
                done = torch.cuda.Event()
                done.record(ctx_stream)
                cur.wait_event(done)                      # the gather's inputs are complete
            else:
                ctx.synchronize()
            packed = torch.cat([tk_idx.to(torch.float32), tk_conf], 1)
            if ctx_stream is not None:
                read = torch.cuda.Event()
                read.record(cur)
                ctx_stream.wait_event(read)               # the next forward's top-k must not overwrite them earlier
            sharding.gather_results(packed.cpu() if dryrun else packed, n_total, rank, world)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()

    ctx.set_profiling(True)     # HIP events around every launch, on the context stream
    stage_tot = {}
    layer_tot = [[0.0, 0] for _ in range(int(info.n_layers))]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()           # nothing in the timed region waits for the events: they are read after it
    sync_all()
    elapsed = time.perf_counter() - t0
    for k, (ms, n) in ctx.stage_ms().items():
        stage_tot[k] = [ms, n]
    for i, (ms, n) in enumerate(ctx.layer_ms()):
        layer_tot[i] = [ms, n]
    ctx.set_profiling(False)
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if dryrun else "cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    hip_logits = {args.precision: logits[:16].cpu().numpy()} if rank == 0 else None   # segments 0..15 of the global list
    value = n_total * args.steps / elapsed
    segs_done = n_local * args.steps
    slices_per_step = max(1, -(-n_local // args.micro_batch))
    out = {
        "metric": "3s/48kHz segments/sec (BirdNET v2.4)", "value": round(value, 1), "unit": "segments/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "configs[1]: 1000 synthetic 3 s/48 kHz segments per GPU per step, HBM-resident, "
                               "seeded synthetic BirdNET-v2.4-shaped model (EfficientNet-B0-like, 6522 classes)",
                   "segments_per_gpu": n_local, "micro_batch": args.micro_batch,
                   "gflop_per_segment": round((2 * info.macs_per_segment + info.mel_flops_per_segment) / 1e9, 3),
                   "fused_blocks": len(fused), "precision": args.precision,
                   "gemm": {"f32": "v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains)",
                            "f16x3": "f32 operands split into f16 hi + lo, 3 x v_mfma_f32_16x16x32_f16 per product, f32 "
                                     "accumulate (|err| ~1e-7 of sum|a b|, same fp32 logit tolerance as the f32 MFMA path, "
                                     "tests/test_parity_gpu.py)",
                            "f16": "operands rounded to f16, v_mfma_f32_16x16x32_f16, f32 accumulate"}[args.precision]},
    }
    out.update(analyse(clf, m, info, fused, stage_tot, layer_tot, segs_done, args.steps, slices_per_step, args.precision))
    if rank == 0 and world == 1 and args.precision != "f32":
        # the exact-f32 MFMA path, measured beside it on the same inputs (a few steps; not the headline)
        ctx.close()
        clf.close()
        clf = BirdClassifier(model_path, None, top_k=5, min_confidence=0.1, device=local_rank, precision="f32")
        ctx = clf.create_batch_context(args.micro_batch)
        for _ in range(2):
            step()
        sync_all()
        ctx.set_profiling(True)
        st2, ly2 = {}, [[0.0, 0] for _ in range(int(info.n_layers))]
        k2 = max(3, min(args.steps, 5))
        t2 = time.perf_counter()
        for _ in range(k2):
            step()
        sync_all()
        e2 = time.perf_counter() - t2
        for k, (ms, n) in ctx.stage_ms().items():
            st2[k] = [ms, n]
        for i, (ms, n) in enumerate(ctx.layer_ms()):
            ly2[i] = [ms, n]
        ctx.set_profiling(False)
        f32 = {"value": round(n_local * k2 / e2, 1), "unit": "segments/s", "steps": k2,
               "gemm": "v_mfma_f32_16x16x4_f32 in every kernel"}
        f32.update(analyse(clf, m, info, clf.fused_blocks(), st2, ly2, n_local * k2, k2, slices_per_step, "f32"))
        out["f32_mfma_path"] = f32
        hip_logits["f32"] = logits[:16].cpu().numpy()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(model_path, m.sample_count, m.sample_rate, hip_logits)
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
