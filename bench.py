#!/usr/bin/env python3
"""bench.py -- 3 s / 48 kHz segments per second through the MI355X hot path (BirdNET v2.4 shape).

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver launches it
under torch.distributed.run, one rank per GPU over RCCL.  One "step" = one pass of the hot
path (min/max -> STFT*mel -> stem -> fused MBConv blocks -> head -> logits -> sigmoid/top-k)
over 1 000 synthetic segments per GPU that are already resident in HBM (BASELINE.json
configs[1]); with N > 1 each rank owns its own 1 000-segment shard (weak scaling) and the only
collective is the gather of the packed top-k rows to rank 0.  Prints ONE JSON line on rank 0.

`value` is the first timed region of exactly K steps (barrier + synchronise on both sides, max over ranks);
`repeats` holds four more K-step regions and the median of the five.
`roofline` is the dominant kernel (largest total time in the timed region): one of the fused
MBConv kernels, priced in ALGORITHMIC flops (2 x MACs of the block's expand + depthwise +
project convolutions, no halo / padding work) against the MFMA peak of the instruction it uses; its launch
duration comes from HIP events recorded on the context stream around every launch of the timed steps.
`roofline_mel` is the front-end kernel against the HBM roofline (968 448 B per
segment).  `h2d_inclusive` and `end_to_end` (rank 0, N = 1) are the host-buffer entry points and the whole
per-file pipeline (`bhh_process_file`: WAV in, CSV out), which pay the host copy and PCIe and are never `value`.
`cpu_baseline` is the oracle timed on this box's host cores over a bounded sample.

Other workloads: --config c3 (10 000 segments as 8 shards through the C ABI's bh_multi_*; on a 1-GPU box the shards
are logical devices on ordinal 0), c4 (Perch-sized model, 5 s / 32 kHz), c5 (22.05 / 44.1 / 48 kHz round-robin ->
device resampler -> v2.4-shaped model with f16 MFMA operands).
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEGMENTS_PER_GPU = 1000
PRE_WARM_S = 0.6    # untimed forwards before the --warmup steps: the shader clock reaches its sustained value (main()) ...
PRE_WARM_MAX_S = 3.0   # ... continued, a quarter of a second at a time, while the steps still get faster (a box whose first GPU process
                       # this is ramps for ~2 s: 144.8 k in the first region against 153 k on a box that had run anything before)


def pre_warm(one_step, sync):
    """Untimed forwards of the timed shape until the step time stops falling (rank-LOCAL work only: time-bounded loops must not hold
    a collective).  Returns the seconds spent."""
    t0, best = time.perf_counter(), None
    while True:
        tw, k = time.perf_counter(), 0
        while time.perf_counter() - tw < 0.25:
            one_step(); sync(); k += 1
        mean, total = (time.perf_counter() - tw) / k, time.perf_counter() - t0
        if total >= PRE_WARM_MAX_S or (total >= PRE_WARM_S and best is not None and mean > best * 0.995):
            return total
        best = mean if best is None else min(best, mean)
class ClockSampler:
    """Shader clock and package power DURING a timed region (the line said which clock the box idled at, never
    which mode it ran in -- one box flips between 153 k and 143 k with identical code): a side thread reads sysfs every 10 ms --
    /sys/class/drm/card*/device/pp_dpm_sclk (the level marked '*'), .../hwmon/*/power1_average (microwatts) and, where the driver
    offers it, .../hwmon/*/freq1_input -- and falls back to one `rocm-smi --showclocks --showpower --json` call mid-region when
    sysfs has none of them.  Host-side file reads only: nothing touches the device or the timed stream."""

    def __init__(self, device_index=0):
        import glob
        import threading
        self._stop = threading.Event()
        self._thread = None
        self.sclk, self.power, self.source, self.card = [], [], "unavailable", None
        # the sysfs node of THIS HIP device: by PCI address (a box shows every GPU of its host in sysfs, one of them granted -- the
        # first card is usually somebody else's idle GPU); without the address, the card whose clock is highest right now
        base = None
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            addr = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
            if os.path.exists(f"/sys/bus/pci/devices/{addr}/pp_dpm_sclk"):
                base = f"/sys/bus/pci/devices/{addr}"
                self.card = addr
        except Exception:   # noqa: BLE001
            base = None
        if base is None:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"))

            def mhz(pth):
                try:
                    return max(float(l.split(":")[1].lower().replace("mhz", "").replace("*", "")) for l in open(pth).read().splitlines() if l.rstrip().endswith("*"))
                except (OSError, ValueError, IndexError):
                    return -1.0
            if cards:
                base = os.path.dirname(max(cards, key=mhz))
                self.card = base
        self._dpm = base + "/pp_dpm_sclk" if base else None
        self._pw = (sorted(glob.glob(base + "/hwmon/hwmon*/power1_average")) + sorted(glob.glob(base + "/hwmon/hwmon*/power1_input"))) if base else []
        self._fq = sorted(glob.glob(base + "/hwmon/hwmon*/freq1_input")) if base else []
        self._threading = threading

    def _read_once(self):
        got = False
        try:
            if self._fq:
                self.sclk.append(int(open(self._fq[0]).read().strip()) / 1e6)
                got = True
            elif self._dpm:
                for line in open(self._dpm).read().splitlines():
                    if line.rstrip().endswith("*"):
                        self.sclk.append(float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", "")))
                        got = True
            if self._pw:
                self.power.append(int(open(self._pw[0]).read().strip()) / 1e6)
                got = True
        except (OSError, ValueError, IndexError):
            pass
        return got

    def _smi_once(self):
        import subprocess
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
            card = next(iter(json.loads(r.stdout).values()))
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "(" in str(v):
                    self.sclk.append(float(str(v).split("(")[1].split("M")[0]))
                elif "power" in kl and "(w)" in kl:
                    self.power.append(float(v))
            self.source = "rocm-smi, one call inside the region"
        except Exception:   # noqa: BLE001 -- a missing tool leaves the fields null
            pass

    def _run(self):
        if self._read_once():
            self.source = "sysfs (pp_dpm_sclk / hwmon), every 10 ms"
            while not self._stop.wait(0.01):
                self._read_once()
        else:
            self._smi_once()

    def __enter__(self):
        self._thread = self._threading.Thread(target=self._run, daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self._thread.join(timeout=15)

    def summary(self):
        med = lambda v: round(statistics.median(v), 1) if v else None
        return {"sclk_mhz": med(self.sclk), "sclk_mhz_min_max": [round(min(self.sclk), 1), round(max(self.sclk), 1)] if self.sclk else None,
                "power_w": med(self.power), "power_w_min_max": [round(min(self.power), 1), round(max(self.power), 1)] if self.power else None,
                "samples": max(len(self.sclk), len(self.power)), "clock_source": self.source, "clock_card": self.card}


def c4_child_leg(precision):
    """The Perch-sized model (BASELINE configs[3]) in a process of its own, bounded to a few steps, so that the DRIVER's line times it
    (C4 numbers were builder-side only): `bench.py --config c4 --steps 5 --warmup 2` without its extra legs."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", "c4", "--steps", "5", "--warmup", "2", "--no-extra-legs", "--no-cpu-baseline"]
    if precision in ("auto", "f16x3", "f32"):
        cmd += ["--precision", precision]
    t0 = time.perf_counter()
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        return {"value": d["value"], "unit": d["unit"], "metric": d["metric"], "steps": d["steps"], "ms_per_step": d["ms_per_step"],
                "median_of_5": d["repeats"]["median_of_5"], "fused_blocks": d["config"]["fused_blocks"], "roofline": d.get("roofline"),
                "wall_s": round(time.perf_counter() - t0, 1), "what": "bench.py --config c4 --steps 5 --warmup 2 in a child process: " + d["config"]["workload"]}
    except Exception as e:   # noqa: BLE001 -- the headline line must not die with this leg
        return {"value": None, "error": str(e)[:200], "wall_s": round(time.perf_counter() - t0, 1)}


PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16/f16 MFMA peak
PEAK_HBM_GBPS = 8000.0
DTYPE = {"f32": "f32", "f16x3": "f16x3 (f32 operands split into f16 hi + lo, f32 accumulate)", "f16": "f16 (f32 accumulate)",
         # the library's default (bh_config.flags = 0, BH_FLAG_AUTO): f16x3 compute; a row beyond the f16 range is re-run in f32
         "auto": "f16x3 (f32 operands split into f16 hi + lo, f32 accumulate; BH_FLAG_AUTO: rows beyond the f16 range re-run on the f32 kernels)"}


def mel_bytes_per_segment(m):
    """the segment read once + the spectrogram written once (v2.4: 576 000 + 392 448 = 968 448 B)."""
    return 4 * (m.sample_count + sum(b.n_mels * b.n_frames for b in m.branches))


def reference_ort_leg(model_path, hip_logits_first, sample_count):
    """SURVEY 8c / BASELINE.md 3.2: the true reference is ONNX Runtime (ORT_DYLIB_PATH, reference src/constants.rs:547)
    running the published birdnet.onnx on the CPU EP.  When both are on the box (and the model on disk was converted
    from that ONNX file: BIRDA_REFERENCE_ONNX), tools/ort_reference.py binds the ORT C API through ctypes and this
    returns its throughput and max |dlogit|; otherwise a note saying which piece is missing."""
    ort, onnx = os.environ.get("ORT_DYLIB_PATH", ""), os.environ.get("BIRDA_REFERENCE_ONNX", "")
    if not (ort and os.path.exists(ort)):
        return {"available": False, "note": "reference ORT path unavailable (no ORT_DYLIB_PATH on this box): compared against the CPU restatement"}
    if not (onnx and os.path.exists(onnx)):
        return {"available": False, "note": "libonnxruntime found but no BIRDA_REFERENCE_ONNX model: compared against the CPU restatement"}
    try:
        import numpy as np
        from tools import ort_reference
        r = ort_reference.run(ort, onnx, sample_count)
        ref = r.pop("logits_first16")
        # max |dlogit| against the TRUE reference needs the HIP library to run the same weights: the BHM1 file converted from
        # this ONNX file (tools/onnx_to_bhm.py), named by BIRDA_REFERENCE_BHM
        bhm = os.environ.get("BIRDA_REFERENCE_BHM", "")
        if bhm and os.path.exists(bhm):
            from birda_amd import synth
            from birda_amd.classifier import BirdClassifier
            r["max_abs_dlogit_vs_reference"] = {}
            for prec in ("f32", "f16x3"):
                clf = BirdClassifier(bhm, None, precision=prec)
                ctx = clf.create_batch_context(16)
                got = clf.predict_logits(ctx, synth.synth_segments(16, clf.sample_count(), clf.sample_rate()))
                ctx.close(); clf.close()
                d, sc = float(np.abs(got - ref).max()), float(max(1.0, np.abs(ref).max()))
                r["max_abs_dlogit_vs_reference"][prec] = {"max_abs_dlogit": round(d, 6), "max_abs_logit": round(sc, 3), "relative": float(f"{d / sc:.3e}")}
        else:
            r["note"] = "no BIRDA_REFERENCE_BHM (the model converted from this ONNX file): throughput only, no |dlogit|"
        return r
    except Exception as e:   # noqa: BLE001 -- the opportunistic leg must never take the bench down
        return {"available": False, "note": f"ORT reference leg failed: {e!r}"}


def usable_cores():
    """Host cores this process may actually use: the scheduler affinity mask, capped by the cgroup CPU quota when the container has
    one (os.cpu_count() reports the machine's 256 hardware threads whatever the quota; 256 OpenMP threads time-sliced onto a
    32-core quota is not a 256-thread baseline)."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.5)))
    return n, {"hardware_threads": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None,
               "cgroup_quota_cores": quota}


def cpu_baseline(model_path, sample_count, sample_rate, hip_logits=None):
    """The oracle (a port, not the reference: the reference's ORT path cannot run here) timed
    on this box's host cores over a bounded sample of the same synthetic workload.  As the checker it
    also gives BASELINE's second figure, max |dlogit| of the timed HIP path on the first segments."""
    import numpy as np
    from birda_amd import synth
    from oracle import oracle as O

    cores, core_facts = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    om = O.OracleModel(model_path)
    # the timed lists tile 64 distinct segments: all 64 go through the checker (16 on a small host, where 64 would take minutes)
    base = synth.synth_segments(64 if cores >= 8 else 16, sample_count, sample_rate)
    t = time.perf_counter()
    ref = om.forward(base)  # touch code / pages once; reference logits of segments 0..; its time sizes the timed sample
    warm_rate = base.shape[0] / max(time.perf_counter() - t, 1e-3)
    parity = None
    if hip_logits:
        parity = {}
        for name, got in hip_logits.items():
            k = min(len(ref), len(got))
            scale = float(max(1.0, np.abs(ref[:k]).max()))
            d = float(np.abs(got[:k] - ref[:k]).max())
            parity[name] = {"max_abs_dlogit": round(d, 6), "max_abs_logit": round(scale, 3), "relative": float(f"{d / scale:.3e}"),
                            "segments": k, "top1_agree": bool((got[:k].argmax(1) == ref[:k].argmax(1)).all())}
    n = int(min(2048, max(64, 12.0 * warm_rate)))  # ~12 s of CPU work at the rate the warm-up pass showed (<= 1.2 GB of segments)
    segs = np.tile(base, (n // base.shape[0] + 1, 1))[:n]
    t = time.perf_counter()
    om.forward(segs)
    dt = time.perf_counter() - t
    return {"value": round(n / dt, 2), "unit": "segments/s", "cores": cores, "kind": "port", "core_facts": core_facts,
            "per_core": round(n / dt / cores, 3),
            "reference": reference_ort_leg(model_path, hip_logits, sample_count),
            "readme_context": "reference README: 183 segments/s for BirdNET v2.4 on a 24-thread i7-13700K (ORT CPU EP, batch 8)",
            "sample": f"{n} synthetic segments, oracle/birda_oracle.c, OpenMP across segments "
                      f"({cores} threads), fp32, {dt:.1f} s",
            "max_abs_dlogit_vs_oracle": parity}


def layers_have_fused_stem(m, layer_tot):
    """True when layer 0 (the stem conv) heads a fused block: it was launched and layers 1, 2 were not."""
    from birda_amd import modelfile as mf
    L = m.layers
    return (len(L) > 2 and L[0].op == mf.OP_CONV and L[1].op == mf.OP_DWCONV and L[2].op == mf.OP_PWCONV
            and layer_tot[0][1] > 0 and layer_tot[1][1] == 0 and layer_tot[2][1] == 0)


def pmc_traffic(kernel_key):
    """HBM bytes per launch of a kernel from the newest committed PMC summary (profiles/*_traffic.json,
    collected with tools/profile_round.sh on this same command: two rocprofv3 --pmc passes, FETCH_SIZE
    doubled for gfx950 as MI355X_MICROARCH.md prescribes).  kernel_key names the EXACT instantiation
    (e.g. "bh::mel_kernel<6, 3>", "mbconv<3,2,16,...>"); None when no summary holds it."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        v = d.get(kernel_key)
        if v:
            return {"bytes_per_launch": v["hbm_bytes_per_launch"], "read": v["read_bytes_per_launch"],
                    "write": v["write_bytes_per_launch"], "source": os.path.basename(path)}
    return None


def se_blocks(m):
    """The squeeze-excite blocks of a layer table as the library matches them (api_plan.hip match_se_block): [expand | stem] ->
    depthwise -> pool -> 1x1 -> 1x1 -> scale -> project.  Returns (first layer, depthwise, pool, project) index tuples."""
    from birda_amd import modelfile as mf
    L, out, i = m.layers, [], 0
    while i < len(L):
        noexp = L[i].op == mf.OP_DWCONV
        d = i if noexp else i + 1
        if (d + 5 < len(L) and L[d].op == mf.OP_DWCONV and L[d + 1].op == mf.OP_GAP and L[d + 2].op == mf.OP_PWCONV and L[d + 3].op == mf.OP_PWCONV
                and L[d + 4].op == mf.OP_SCALE and L[d + 5].op == mf.OP_PWCONV and (noexp or L[i].op in (mf.OP_PWCONV, mf.OP_CONV))):
            out.append((i, d, d + 1, d + 5))
            i = d + 6
        else:
            i += 1
    return out


def analyse_se(clf, m, fused, layer_tot, segs_done, steps, slices_per_step, precision):
    """The per-kernel picture of a plan whose blocks carry squeeze-excite gates (config 4, round 5): every block is pass A of the
    fused kernel (booked on the block's first layer), the gate (booked on the pool layer) and the gated project GEMM (booked on the
    project layer).  `roofline` = the kernel with the largest total time; for the Perch-sized plan that is the streaming gated GEMM
    of the early blocks -- bound by HBM: per pixel it reads the K floats of the depthwise output (and N of the residual) and writes N."""
    from birda_amd import modelfile as mf
    per_step = steps * slices_per_step
    groups, a_us, b_us, g_us = {}, {}, {}, 0.0
    fi = 0
    for (i0, d, g, p) in se_blocks(m):
        if layer_tot[i0][1] == 0 or layer_tot[p][1] == 0 or (i0 != d and layer_tot[d][1] != 0):
            continue    # (not run as a fused squeeze-excite block)
        E, D, P = m.layers[i0], m.layers[d], m.layers[p]
        px = D.out_h * D.out_w
        kname = clf.fused_kernel_name(fused[fi], se=True) if fi < len(fused) else "mbconv<?>"
        fi += 1
        a_macs = (0 if i0 == d else (E.out_h * E.out_w * (E.kh * E.kw * E.cin) * E.cout)) + px * D.kh * D.kw * D.cout
        ga = groups.setdefault(("A", kname), {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "kind": "mfma"})
        ga["ms"] += layer_tot[i0][0]; ga["launches"] += layer_tot[i0][1]; ga["flops"] += 2.0 * a_macs * segs_done
        a_us[kname] = a_us.get(kname, 0.0) + layer_tot[i0][0] * 1e3 / segs_done * 1000
        g_us += layer_tot[g][0] * 1e3 / segs_done * 1000
        K, N = P.cin, P.cout
        nt = -(-N // 16)
        # which kernel launch_pw_gemm16_gated (kernels_conv.hip) takes, named as rocprofv3 prints it
        rows = px * (segs_done // per_step)
        thin = nt <= 3 and -(-K // 32) * nt * 2048 <= 65536 and rows >= 4096 and N % 4 == 0
        # (the row-streaming kernel takes N > 144 from 40 960 rows up only, and not when a pass's segments' gates and its W
        #  pieces exceed a CU's LDS -- kernels_conv.hip launch_pw_gemm16_gated; below, the staged tiles run)
        rb_, pf_ = (3, 3) if nt <= 7 else (2, 4) if nt <= 9 else (2, 3)
        gs_max = (8 * rb_ * 16 + px - 2) // px + 1
        wide_lds = max(pf_ * nt * 2048 + gs_max * (-(-K // 32) * 32) * 4, 8 * 16 * nt * 16 * 4)
        wide = 4 <= nt <= 15 and rows >= (40960 if nt >= 10 else 4096) and N % 4 == 0 and wide_lds <= 160 * 1024
        terms = 3 if precision in ("auto", "f16x3") else 1
        if i0 == d:     # a block without an expand convolution: D is computed again by the gated one-launch block (kernels.hpp MbDesc::gate)
            bname, bkind = clf.fused_kernel_name(fused[fi - 1], se=False) + " (no-expand block: depthwise x gate -> project + residual in one launch)", "hbm"
        elif precision == "f32":
            bname, bkind = "gated project GEMM %d -> %d (pw_gemm_kernel, f32 MFMA)" % (K, N), "mfma"
        elif thin:
            bname, bkind = "pw_gemm16_thin_kernel<%d, %d, %s, 0>" % (terms, nt, "true" if K <= 32 else "false"), "hbm"
        elif wide:
            rb, pf = (3, 3) if nt <= 7 else (2, 4) if nt <= 9 else (2, 3)
            bname, bkind = "pw_gemm16_wide_kernel<%d, %d, %d, %d, 0>" % (terms, nt, rb, pf), "hbm"
        else:
            bname, bkind = "gated project GEMM %d -> %d (pw_gemm16s_kernel, 128 x 128 staged tiles)" % (K, N), "mfma"
        gb = groups.setdefault(("B", bname), {"ms": 0.0, "launches": 0, "flops": 0.0, "bytes": 0.0, "kind": bkind})
        gb["ms"] += layer_tot[p][0]; gb["launches"] += layer_tot[p][1]
        gb["flops"] += 2.0 * px * K * N * segs_done
        gb["bytes"] += 4.0 * px * (K + N + (N if (P.res_tensor != mf.NO_TENSOR and i0 != d) else 0)) * segs_done     # (no-expand: the residual IS the input it reads)
        b_us["%d->%d @%dx%d" % (K, N, D.out_h, D.out_w)] = b_us.get("%d->%d @%dx%d" % (K, N, D.out_h, D.out_w), 0.0) + layer_tot[p][0] * 1e3 / segs_done * 1000
    out = {}
    if not groups:
        return out
    (kind, name), dom = max(groups.items(), key=lambda kv: kv[1]["ms"])
    avg_us = dom["ms"] * 1e3 / max(dom["launches"], 1)
    if dom["kind"] == "hbm":
        gbps = dom["bytes"] / (dom["ms"] * 1e-3) / 1e9
        out["roofline"] = {"kernel": name + " (gated project convolution of squeeze-excite blocks: D x gate -> 1x1, rows streamed once)",
                           "bound": "hbm", "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                           "traffic": pmc_traffic("bh::" + name), "rocprof_name": "bh::" + name, "launches": dom["launches"], "avg_launch_us": round(avg_us, 2),
                           "algorithmic_gb_per_launch": round(dom["bytes"] / max(dom["launches"], 1) / 1e9, 4),
                           "algorithmic_bytes": "per pixel: K floats of the depthwise output read + N written (+ N of the residual read)"}
    else:
        peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS / (3.0 if precision in ("auto", "f16x3") else 1.0)
        tflops = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": ("mbconv_kernel pass A (expand -> depthwise of a squeeze-excite block) " if kind == "A" else "") + name,
                           "bound": "mfma", "achieved": round(tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(tflops / peak, 4),
                           "traffic": None, "launches": dom["launches"], "avg_launch_us": round(avg_us, 2)}
    out["squeeze_excite"] = {"blocks": fi, "pass_a_us_per_1000_segments": {k: round(v, 1) for k, v in a_us.items()},
                             "gate_us_per_1000_segments": round(g_us, 1),
                             "gated_project_us_per_1000_segments": {k: round(v, 1) for k, v in b_us.items()},
                             "us_per_segment": {"pass_a": round(sum(a_us.values()) / 1000, 3), "gate": round(g_us / 1000, 3), "gated_project": round(sum(b_us.values()) / 1000, 3)}}
    return out


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
    if n_stem_blocks:
        E, D, P = layers[0], layers[1], layers[2]
        groups[("stem", E.cin, E.cout, P.cout, D.kh, D.sh, E.out_h, E.out_w)] = {
            "ms": layer_tot[0][0], "launches": layer_tot[0][1], "macs": macs(E) + macs(D) + macs(P),
            "kernel": clf.fused_kernel_name(fused[0]), "stem": True}
        i = 3
    while fused and i + 2 < len(layers) and bi + n_stem_blocks < len(fused):
        E, D, P = layers[i], layers[i + 1], layers[i + 2]
        if E.op == mf.OP_DWCONV and D.op == mf.OP_PWCONV and D.in_tensor == i + 1 and layer_tot[i][1] > 0 and layer_tot[i + 1][1] == 0:
            # a fused block WITHOUT an expand convolution (depthwise -> project): layers i, i + 1
            key = ("noexp", E.cout, D.cout, E.kh, E.sh, E.in_h, E.in_w)
            g = groups.setdefault(key, {"ms": 0.0, "launches": 0, "macs": macs(E) + macs(D),
                                        "kernel": clf.fused_kernel_name(fused[bi + n_stem_blocks]), "stem": False})
            g["ms"] += layer_tot[i][0]
            g["launches"] += layer_tot[i][1]
            bi += 1
            i += 2
            continue
        if (E.op == mf.OP_PWCONV and D.op == mf.OP_DWCONV and P.op == mf.OP_PWCONV and D.in_tensor == i + 1
                and P.in_tensor == i + 2 and layer_tot[i][1] > 0 and layer_tot[i + 1][1] == 0):
            key = (E.cin, E.cout, P.cout, D.kh, D.sh, E.in_h, E.in_w)
            g = groups.setdefault(key, {"ms": 0.0, "launches": 0, "macs": macs(E) + macs(D) + macs(P),
                                        "kernel": clf.fused_kernel_name(fused[bi + n_stem_blocks]), "stem": False})
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
        args_ = dom["kernel"].rstrip(">").split(",")
        dom_prec = int(args_[15])   # (mbconv's 17th argument marks a persistent instantiation)
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
        if dom_key[0] == "noexp":
            desc = "depthwise %dx%d s%d -> project 1x1, %d -> %d at %dx%d" % (dom_key[3], dom_key[3], dom_key[4], dom_key[1], dom_key[2], dom_key[5], dom_key[6])
        elif dom["stem"]:
            desc = "stem conv %dx%d s%d (im2col GEMM) -> depthwise -> project 1x1, %d -> %d -> %d at %dx%d" % (
                3, 3, 2, dom_key[1], dom_key[2], dom_key[3], dom_key[6], dom_key[7])
        else:
            desc = "expand 1x1 -> depthwise %dx%d s%d -> project 1x1, Cin %d -> %d -> %d at %dx%d" % (
                dom_key[3], dom_key[3], dom_key[4], dom_key[0], dom_key[1], dom_key[2], dom_key[5], dom_key[6])
        out["roofline"] = {
            "kernel": "mbconv_kernel (fused %s, %s)" % (desc, insn),
            "bound": "mfma", "achieved": round(tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(tflops / peak, 4), "peak_note": note, "traffic": pmc_traffic(dom["kernel"]),
            "rocprof_name": "bh::mbconv_kernel<" + dom["kernel"][len("mbconv<"):],
            "launches": dom["launches"], "avg_launch_us": round(dom["ms"] * 1e3 / max(dom["launches"], 1), 2),
            "algorithmic_gflop_per_launch": round(total_flops / max(dom["launches"], 1) / 1e9, 3)}
        # What actually bounds this kernel in the f16 modes: issuing its activation.  Algorithmic count: one
        # activation per expanded value and one per depthwise output (no tile halo); price: the THROUGHPUT of a SIMD shared by 3-4
        # waves, measured on this chip (tools/microbench/valu_throughput.hip, profiles/r3_c_valu_throughput.txt): 27.4 ns per
        # wave-instruction PAIR of GELUs (2 v_med3 + 7 v_pk_fma + 2 v_exp; 128 values), 23.4 ns for swish; 1 024 SIMDs.
        if dom_prec != 0:
            if dom_key[0] == "noexp":
                i_dom = next(i for i in range(len(layers) - 1) if layers[i].op == mf.OP_DWCONV and layers[i + 1].op == mf.OP_PWCONV and
                             ("noexp", layers[i].cout, layers[i + 1].cout, layers[i].kh, layers[i].sh, layers[i].in_h, layers[i].in_w) == dom_key)
                E = D = layers[i_dom]
            elif dom["stem"]:
                E, D = layers[0], layers[1]
            else:
                i_dom = next(i for i in range(len(layers) - 2) if layers[i].op == mf.OP_PWCONV and layers[i + 1].op == mf.OP_DWCONV and
                             (layers[i].cin, layers[i].cout, layers[i + 2].cout, layers[i + 1].kh, layers[i + 1].sh, layers[i].in_h, layers[i].in_w) == dom_key)
                E, D = layers[i_dom], layers[i_dom + 1]
            acts = ((0 if dom_key[0] == "noexp" else E.out_h * E.out_w * E.cout) + D.out_h * D.out_w * D.cout) * (segs_done / (steps * slices_per_step))   # per launch
            ns_pair = 23.4 if D.act == mf.ACT_SWISH else 27.4
            floor_us = acts / 128 * ns_pair / (256 * 4) * 1e-3
            avg_us = dom["ms"] * 1e3 / max(dom["launches"], 1)
            out["roofline"]["vector_issue"] = {
                "activations_per_launch": int(acts), "ns_per_wave_instruction_pair": ns_pair, "simds": 1024,
                "floor_us": round(floor_us, 1), "frac": round(floor_us / avg_us, 4),
                "note": "activation issue alone at the measured multi-wave VALU throughput (profiles/r3_c_valu_throughput.txt); tile halo, "
                        "depthwise taps, f16 splits and index arithmetic come on top -- the kernel's vector pipe is 77-90 % busy "
                        "(profiles/r3_e_sq_counters.txt)"}
        if mb_ms > 0:
            flops = 2.0 * sum(g["macs"] * max(1, g["launches"] // (steps * slices_per_step)) for g in groups.values()) * segs_done
            out["all_fused_blocks"] = {"achieved": round(flops / (mb_ms * 1e-3) / 1e12, 2), "unit": "TFLOP/s (algorithmic)",
                                       "launches": mb_launches, "us_per_segment": round(mb_ms * 1e3 / segs_done, 3)}
            out["fused_block_us_per_1000_segments"] = {g["kernel"]: round(g["ms"] * 1e3 / segs_done * 1000 / max(1, g["launches"] // (steps * slices_per_step)), 1)
                                                       for g in groups.values()}
    if "roofline" not in out and fused and any(L.op == mf.OP_SCALE for L in layers):
        out.update(analyse_se(clf, m, fused, layer_tot, segs_done, steps, slices_per_step, precision))
    mel_ms, mel_launches = stage_tot["mel"]
    mel_b = mel_bytes_per_segment(m)
    mel_gbps = mel_b * segs_done / (mel_ms * 1e-3) / 1e9
    f16_fe = precision != "f32" and os.environ.get("BIRDA_HIP_MEL_F32") != "1"
    mel_key = clf.mel_kernel_name() if hasattr(clf, "mel_kernel_name") else None
    out["roofline_mel"] = {"kernel": "%s (folded STFT x mel, %s)" % (mel_key or "mel_kernel", "split f16 x3 MFMA" if f16_fe else "v_mfma_f32_16x16x4_f32"),
                           "bound": "hbm", "achieved": round(mel_gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                           "frac": round(mel_gbps / PEAK_HBM_GBPS, 4), "algorithmic_bytes_per_segment": mel_b,
                           "traffic": pmc_traffic(mel_key) if mel_key else None,
                           "launches": mel_launches, "avg_launch_us": round(mel_ms * 1e3 / max(mel_launches, 1), 2),
                           "mfma_tflops": round(info.mel_flops_per_segment * segs_done / (mel_ms * 1e-3) / 1e12, 2)}
    out["stage_us_per_segment"] = {k: round(v[0] * 1e3 / segs_done, 3) for k, v in stage_tot.items()}
    return out


def torch_gather_rows(rows, n_total, rank, world, on_host):
    """The one collective of the path: packed int32 top-k rows of every rank to rank 0, in segment order
    (all_gather_into_tensor over RCCL; gloo gather in the CPU tests and the 1-GPU dry run)."""
    import torch
    import torch.distributed as dist
    from birda_amd import sharding
    if world == 1:
        return rows
    sizes = [sharding.shard_range(n_total, r, world) for r in range(world)]
    max_n = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((max_n, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad[: rows.shape[0]] = rows
    if not on_host and dist.get_backend() == "nccl":
        out = torch.empty((world * max_n, rows.shape[1]), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(out, pad)
        if rank != 0:
            return None
        return torch.cat([out[r * max_n: r * max_n + (hi - lo)] for r, (lo, hi) in enumerate(sizes)], 0)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return torch.cat([bufs[r][: hi - lo] for r, (lo, hi) in enumerate(sizes)], 0)


def host_legs(clf, m, model_path, precision, tmp):
    """H2D-inclusive entry points and the end-to-end per-file pipeline (rank 0, N = 1; BASELINE.md 3.5 (b), reference metric
    definition processor.rs:771-788).  Bounded: three calls each on 1 000 segments."""
    import numpy as np
    from birda_amd import pipeline, synth
    n = SEGMENTS_PER_GPU
    uniq = synth.synth_segments(16, m.sample_count, m.sample_rate)
    host = np.ascontiguousarray(np.tile(uniq, (n // 16 + 1, 1))[:n])
    ctx = clf.create_batch_context(n)
    out = {}
    import ctypes as C
    from birda_amd._lib import BhResult, check
    arr = (BhResult * n)()

    def timed(fn, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
        return statistics.median(ts)
    t = timed(lambda: check(clf._L.bh_predict_batch_contig(clf._h, ctx._h, host.ctypes.data, n, arr)))
    out["bh_predict_batch_contig"] = {"value": round(n / t, 1), "unit": "segments/s", "input": "pageable f32 host segments",
                                      "pcm_gb_per_s": round(n * m.sample_count * 4 / t / 1e9, 2)}
    # birda's own call: process_batch hands predict_batch_with_context (predict_batch without a context) a
    # Vec<&[f32]> of INDEPENDENT pageable slices, padded to the effective batch -- 512 = bh_default_batch_size for this backend
    # (processor.rs:240-277, classifier.rs:478-488, 571-582); one slice per segment, each its own allocation
    nb = 512
    slices = [np.array(host[i], copy=True) for i in range(nb)]
    ptrs = (C.c_void_p * nb)(*[a.ctypes.data for a in slices])
    cx512 = clf.create_batch_context(nb)
    t = timed(lambda: check(clf._L.bh_predict_batch_with_context(clf._h, cx512._h, ptrs, nb, m.sample_count, arr)), reps=5)
    out["bh_predict_batch_with_context"] = {"value": round(nb / t, 1), "unit": "segments/s", "n": nb, "ms_per_call": round(t * 1e3, 3),
                                            "input": "512 independent pageable f32 slices (the reference's &[&[f32]]), one call = one process_batch"}
    t = timed(lambda: check(clf._L.bh_predict_batch(clf._h, ptrs, nb, m.sample_count, arr)), reps=5)
    out["bh_predict_batch"] = {"value": round(nb / t, 1), "unit": "segments/s", "n": nb, "ms_per_call": round(t * 1e3, 3),
                               "input": "the same slices without a caller-owned context (Perch's path, processor.rs:582-603)"}
    cx512.close()
    del slices, ptrs
    from birda_amd.classifier import PinnedSegments
    pin = PinnedSegments(n, m.sample_count)
    pin.array[:] = host
    t = timed(lambda: check(clf._L.bh_predict_batch_contig(clf._h, ctx._h, pin.array.ctypes.data, n, arr)))
    out["bh_predict_batch_contig_pinned"] = {"value": round(n / t, 1), "unit": "segments/s",
                                             "input": "f32 host segments in pinned memory (bh_host_alloc): uploaded without the gather copy",
                                             "pcm_gb_per_s": round(n * m.sample_count * 4 / t / 1e9, 2)}
    pin.close()
    pcm = np.clip(np.round(host.reshape(-1).astype(np.float64) * 32767.0), -32768, 32767).astype(np.int16)
    # (the C entry point itself, as the contiguous legs above: the Python mirror's result objects cost 2-3 ms per 1 000 segments)
    starts = (C.c_uint64 * n)()
    n_out = C.c_size_t()
    pcm16 = lambda: check(clf._L.bh_predict_pcm16(clf._h, ctx._h, pcm.ctypes.data, pcm.shape[0], 1, m.sample_rate, 0, arr, n, C.byref(n_out), starts))
    t = timed(pcm16)
    assert n_out.value == n
    out["bh_predict_pcm16"] = {"value": round(n / t, 1), "unit": "segments/s", "input": "decoded int16 stream, scaled / windowed on the device",
                               "pcm_gb_per_s": round(pcm.nbytes / t / 1e9, 2)}
    check(clf._L.bh_host_register(pcm.ctypes.data, pcm.nbytes))
    t = timed(pcm16)
    check(clf._L.bh_host_unregister(pcm.ctypes.data))
    out["bh_predict_pcm16_pinned"] = {"value": round(n / t, 1), "unit": "segments/s",
                                      "input": "the same stream in registered (pinned) host memory: uploaded without the gather copy",
                                      "pcm_gb_per_s": round(pcm.nbytes / t / 1e9, 2)}
    ctx.close()
    # end to end: WAV file in, CSV out (decode + segment + classify + threshold + sort + write), device front end
    wav = os.path.join(tmp, "bench_1000_segments.wav")
    synth.write_wav_pcm16(wav, host.reshape(-1), m.sample_rate)
    labels = os.path.join(tmp, "bench_labels.txt")
    if not os.path.exists(labels):
        synth.write_labels(labels, m.n_classes)
    # ONE classifier per process, as in a birda process (lib.rs:1003-1100 builds one and shares it by reference): the timed
    # classifier itself, which main() builds with this label file.  (Rounds 3-4 built a second one here, beside the first and its
    # parked 4-GB contexts: these legs then ran 5-20 % below the same code in a process of its own.)
    c2 = clf
    e2e = {}
    for fe, reps in (("device", 3), ("host", 1)):
        pipeline.process_file(c2, wav, tmp, front_end=fe)
        rs = [pipeline.process_file(c2, wav, tmp, front_end=fe) for _ in range(reps)]
        r = sorted(rs, key=lambda x: x.segments_per_sec)[len(rs) // 2]
        e2e[fe] = {"value": round(r.segments_per_sec, 1), "unit": "segments/s", "segments": r.segments, "detections": r.detections,
                   "batch": r.effective_batch, "realtime_factor": round(r.audio_duration_secs / r.duration_secs, 1)}
    # many short recordings (the common field-recorder case: one-minute files = 20 segments): one file at a time as the reference
    # does (lib.rs:1003-1100), and packed into shared uploads / forwards (bhh_process_files)
    short_dir = os.path.join(tmp, "short")
    os.makedirs(short_dir, exist_ok=True)
    per_file = 20
    shorts = []
    for k in range(50):
        p = os.path.join(short_dir, "rec_%03d.wav" % k)
        synth.write_wav_pcm16(p, host[(k * per_file) % (n - per_file): (k * per_file) % (n - per_file) + per_file].reshape(-1), m.sample_rate)
        shorts.append(p)
    out_a, out_b = os.path.join(tmp, "short_single"), os.path.join(tmp, "short_packed")
    os.makedirs(out_a, exist_ok=True); os.makedirs(out_b, exist_ok=True)
    for f in shorts[:3]:
        pipeline.process_file(c2, f, out_a)
    t = time.perf_counter()
    segs_a = sum(pipeline.process_file(c2, f, out_a).segments for f in shorts)
    t_single = time.perf_counter() - t
    pipeline.process_files_packed(c2, shorts, out_b)
    t = time.perf_counter()
    res_b, status_b = pipeline.process_files_packed(c2, shorts, out_b)
    t_packed = time.perf_counter() - t
    segs_b = sum(r.segments for r in res_b)
    same = all(open(pipeline.output_path_for(f, out_a, "csv"), "rb").read() == open(pipeline.output_path_for(f, out_b, "csv"), "rb").read()
               for f in shorts)
    e2e["short_files"] = {"what": "50 PCM16 WAV files of %d segments each -> 50 CSV files" % per_file,
                          "one_file_at_a_time": {"value": round(segs_a / t_single, 1), "unit": "segments/s"},
                          "packed": {"value": round(segs_b / t_packed, 1), "unit": "segments/s", "entry_point": "bhh_process_files"},
                          "identical_outputs": bool(same and not any(status_b))}
    # the same long file eight times over through bhh_process_files: each file is a pack of its own; three packs are in flight, so a
    # file's upload and forward run under its neighbours' forwards, the previous file's output writing and the next one's copy into
    # a free context's staging buffer (median of three runs: one run is 70 ms of wall time)
    longs = []
    for k in range(8):
        p = os.path.join(tmp, "bench_long_%d.wav" % k)
        if not os.path.exists(p):
            os.link(wav, p)
        longs.append(p)
    out_l = os.path.join(tmp, "long_packed")
    os.makedirs(out_l, exist_ok=True)
    pipeline.process_files_packed(c2, longs, out_l)
    rates, ok_l = [], True
    for _ in range(3):
        t = time.perf_counter()
        res_l, status_l = pipeline.process_files_packed(c2, longs, out_l)
        rates.append(sum(r.segments for r in res_l) / (time.perf_counter() - t))
        ok_l = ok_l and not any(status_l)
    e2e["files_pipelined"] = {"what": "8 files of %d segments each through bhh_process_files (three in flight), median of 3 runs" % n,
                              "value": round(sorted(rates)[1], 1), "unit": "segments/s", "runs": [round(r, 1) for r in rates], "ok": ok_l}
    # The same eight files in a process of its own -- one classifier and nothing else, which is what a birda process is.  The HIP
    # runtime deals a process's streams onto a few hardware queues by what exists, and this process has
    # created and destroyed dozens by now; a child process (started, never exec'ed into) shows the pipeline without that history.
    try:
        import subprocess
        child = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-files", model_path, labels, precision, out_l] + longs,
                               capture_output=True, text=True, timeout=300)
        got = json.loads(child.stdout.strip().splitlines()[-1]) if child.returncode == 0 and child.stdout.strip() else None
    except Exception as ex:  # noqa: BLE001 -- a leg that cannot run is reported, not fatal
        got = {"error": str(ex)[:200]}
    e2e["files_pipelined_own_process"] = got
    out["end_to_end"] = {"what": "bhh_process_file on a synthetic %d-segment PCM16 WAV -> CSV (reference metric: segments / wall seconds, "
                                 "processor.rs:771-788), default batch size" % n, **e2e}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--micro-batch", type=int, default=int(os.environ.get("BIRDA_HIP_MICRO_BATCH", "1000")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the f32, H2D-inclusive and end-to-end legs (profiling runs)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs: c2 = configs[1] (the bench line), c3 = 10 000 segments as 8 shards through bh_multi_*, "
                         "c4 = Perch-sized model, c5 = mixed-rate input -> resampler -> f16 MFMA")
    ap.add_argument("--precision", default=os.environ.get("BIRDA_HIP_BENCH_PRECISION", ""),
                    choices=["", "auto", "f32", "f16x3", "f16"],
                    help="GEMM operands: f16x3 = f32 values split into f16 hi + lo, three f16 MFMAs per product, "
                         "f32 accumulate (same fp32 logit tolerance as f32); f32 = v_mfma_f32_16x16x4_f32 everywhere")
    args = ap.parse_args()
    if not args.precision:
        args.precision = "f16" if args.config == "c5" else "auto"    # auto = the library's default (bh_config.flags = 0)

    import numpy as np
    import torch
    import torch.distributed as dist

    from birda_amd import modelfile as mf, sharding, synth
    from birda_amd.classifier import BirdClassifier

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    n_visible = torch.cuda.device_count()
    if world > 1 and args.gpus != world:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                 f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...) or unset WORLD_SIZE")
    if os.environ.get("BIRDA_BENCH_DRYRUN_ONE_DEVICE") != "1" and n_visible < (args.gpus if world == 1 else 1):
        sys.exit(f"bench.py: --gpus {args.gpus} asked for, but {n_visible} HIP device(s) are visible: refusing to report a line "
                 f"that says n_gpus={args.gpus} (there is no CPU path and no silent single-GPU fallback)")
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

    # seeded synthetic model (no real weights exist offline)
    tmp = tempfile.mkdtemp(prefix=f"birda_bench_r{rank}_")
    kind = "perch_v2" if args.config == "c4" else "birdnet_v24"
    model_path = os.path.join(tmp, f"{kind}_synth.bhm")
    m = synth.build_model(kind)
    mf.write_model(model_path, m)

    if world == 1 and (args.gpus > 1 or args.config == "c3"):
        # `--gpus N` without torch.distributed.run (and config 3): N shards in this process through bh_multi_*
        if args.config in ("c4", "c5"):
            sys.exit(f"bench.py: --config {args.config} --gpus {args.gpus} needs torch.distributed.run (one rank per GPU); "
                     "the in-process bh_multi_* path serves c2 and c3")
        if args.config == "c3":
            devices = [g * args.gpus // 8 for g in range(8)]
            return bench_inproc_multi(args, m, model_path, tmp, devices, 10000, "strong",
                                      "configs[2]: 10 000 synthetic 3 s/48 kHz segments as 8 contiguous shards in ONE process through "
                                      "bh_multi_forward_device (host threads, one context + stream per shard, packed top-k gather); "
                                      "results include the unpack into bh_result rows")
        return bench_inproc_multi(args, m, model_path, tmp, list(range(args.gpus)), SEGMENTS_PER_GPU * args.gpus, "weak",
                                  "configs[1] per GPU: 1000 synthetic 3 s/48 kHz segments per GPU per step, HBM-resident, one shard per GPU, "
                                  "packed top-k rows gathered to the host")

    # (with labels: the end-to-end legs write CSV rows through this same classifier)
    labels_path = os.path.join(tmp, "bench_labels.txt")
    synth.write_labels(labels_path, m.n_classes)
    clf = BirdClassifier(model_path, labels_path, top_k=5, min_confidence=0.1, device=local_rank, precision=args.precision)
    ctx = clf.create_batch_context(args.micro_batch)
    info = clf.info
    fused = clf.fused_blocks()

    n_local = 10000 // world if args.config == "c3" else SEGMENTS_PER_GPU
    n_total = n_local * world
    lo, hi = sharding.shard_range(n_total, rank, world)
    logits = torch.empty((min(n_local, args.micro_batch) if args.config == "c3" else n_local, m.n_classes), device="cuda")
    tk_idx = torch.empty((n_local, 5), dtype=torch.int32, device="cuda")
    tk_conf = torch.empty((n_local, 5), device="cuda")
    rates = [m.sample_rate]
    if args.config == "c5":
        # the same signal synthesised at 22 050 / 44 100 / 48 000 Hz round-robin over the global list; on the
        # device the segments of one rate sit together (one resampler launch per rate), results return in list order
        from birda_amd import pipeline
        rates = [22050, 44100, 48000]
        groups = []
        for r in rates:
            ids = [j for j in range(n_local) if (lo + j) % 3 == rates.index(r)]
            src_len = pipeline.source_samples(m.sample_count, r, m.sample_rate)
            uniq = synth.synth_segments(16, src_len, r, start=0)
            host = np.stack([uniq[(lo + j) % 16] for j in ids])
            groups.append({"rate": r, "src_len": src_len, "ids": torch.tensor(ids, device="cuda"), "x": torch.from_numpy(host).cuda(),
                           "n": len(ids)})
        x48 = torch.empty((n_local, m.sample_count), device="cuda")      # resampled (or copied) segments, grouped by rate
        order = torch.cat([g["ids"] for g in groups])
        inv = torch.empty_like(order)
        inv[order] = torch.arange(n_local, device="cuda")
        x = None
    else:
        # segment i of the global list; 64 distinct seeds tiled keeps host prep short
        uniq = synth.synth_segments(64, m.sample_count, m.sample_rate, start=0)
        host = np.stack([uniq[(lo + j) % 64] for j in range(n_local)])
        x = torch.from_numpy(host).cuda()

    # The result gather runs on torch's stream; the forward runs on the context's own HIP stream.  Order them with
    # events (stream-to-stream, no host synchronise inside a step) when torch can wrap the context stream.
    ctx_stream = None
    if world > 1:
        try:
            ctx_stream = torch.cuda.ExternalStream(ctx.stream(), device=torch.device("cuda", local_rank))
        except Exception:   # noqa: BLE001 -- fall back to a host synchronise per step
            ctx_stream = None

    def forward_all(src_ptr):
        if args.config == "c3":      # logits are scratch here: one micro-batch of rows
            for b0 in range(0, n_local, args.micro_batch):
                nb = min(args.micro_batch, n_local - b0)
                clf.forward_device(ctx, src_ptr + b0 * m.sample_count * 4, nb, logits.data_ptr(), tk_idx.data_ptr() + b0 * 20, tk_conf.data_ptr() + b0 * 20)
        else:
            clf.forward_device(ctx, src_ptr, n_local, logits.data_ptr(), tk_idx.data_ptr(), tk_conf.data_ptr())

    def local_step():
        if args.config == "c5":
            off = 0
            for g in groups:     # decode_and_stream's resample_chunk + resize per segment (processor.rs:84-87), on the context stream
                clf.resample_device(ctx, g["x"].data_ptr(), g["src_len"], g["src_len"], g["rate"], m.sample_rate,
                                    x48.data_ptr() + off * m.sample_count * 4, m.sample_count, m.sample_count, g["n"])
                off += g["n"]
            forward_all(x48.data_ptr())
        else:
            forward_all(x.data_ptr())

    def step():
        local_step()
        if world > 1:
            cur = torch.cuda.current_stream()
            if ctx_stream is not None:
                done = torch.cuda.Event()
                done.record(ctx_stream)
                cur.wait_event(done)                      # the gather's inputs are complete
            else:
                ctx.synchronize()
            packed = torch.cat([tk_idx, tk_conf.view(torch.int32)], 1)     # indices stay integers; confidences travel as bit patterns
            if ctx_stream is not None:
                read = torch.cuda.Event()
                read.record(cur)
                ctx_stream.wait_event(read)               # the next forward's top-k must not overwrite them earlier
            torch_gather_rows(packed.cpu() if dryrun else packed, n_total, rank, world, dryrun)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed_region(k):
        sync_all()
        t0 = time.perf_counter()
        for _ in range(k):
            step()           # nothing in the timed region waits for the profiling events: they are read after it
        sync_all()
        e = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([e], device="cpu" if dryrun else "cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            e = float(t.item())
        return e

    # Shape warm-up, as the reference does it (classifier.rs:414-466 `ensure_warm`: one dummy inference per distinct batch size
    # before the first timed batch; lib.rs:1049 at start-up) -- here the registry call plus PRE_WARM_S seconds of untimed forwards
    # of the timed shape, so that the contract's FIRST region runs at the clock the chip sustains under this load (it ramps from
    # its idle 570 MHz over the first ~0.3 s: round 4's driver line had regions 1-2 at 141-142 k and 3-5 at 149-151 k -- and on a box
    # whose first GPU process this is for ~2 s: pre_warm() goes on, a quarter of a second at a time, while the steps still get
    # faster, 3 s at most; `config.pre_warm_s` says how long it took).  Untimed; the --warmup steps follow as the contract says.
    clf.ensure_warm(min(n_local, args.micro_batch))
    pre_warm_s = pre_warm(local_step, ctx.synchronize)
    for _ in range(args.warmup):
        step()
    sync_all()

    ctx.set_profiling(True)     # HIP events around every launch, on the context stream
    with ClockSampler(local_rank) as clocks:      # which mode the box is in WHILE the contract's region runs (host-side sysfs reads)
        elapsed = timed_region(args.steps)
    stage_tot = {k: [ms, n] for k, (ms, n) in ctx.stage_ms().items()}
    layer_tot = [[ms, n] for (ms, n) in ctx.layer_ms()]
    ctx.set_profiling(False)
    # four more K-step regions: median of 5; `value` stays the first, the contract's region
    more = [timed_region(args.steps) for _ in range(4)]
    all_values = [n_total * args.steps / e for e in [elapsed] + more]

    if args.config == "c5":
        first = logits[inv[:16]].cpu().numpy()            # list order
    else:
        first = logits[:64].cpu().numpy()                 # every distinct segment of the timed batch
    hip_logits = {args.precision: first} if rank == 0 else None   # segments 0.. of the global list
    value = n_total * args.steps / elapsed
    segs_done = n_local * args.steps
    slices_per_step = max(1, -(-n_local // args.micro_batch))
    workloads = {
        "c2": "configs[1]: 1000 synthetic 3 s/48 kHz segments per GPU per step, HBM-resident, seeded synthetic BirdNET-v2.4-shaped "
              "model (EfficientNet-B0-like, 6522 classes)",
        "c3": "configs[2]: 10 000 synthetic 3 s/48 kHz segments sharded over the ranks (contiguous blocks), HBM-resident, top-k gather to rank 0",
        "c4": "configs[3]: 1000 synthetic 5 s/32 kHz segments per GPU per step, HBM-resident, seeded synthetic Perch-v2-SIZED model "
              "(one 128-mel branch, EfficientNet-B3 stage plan with swish, 1536-d embedding, 6144-wide hidden layer, 14 795 classes, "
              "softmax: 437 MB, 2.7 GFLOP per segment)",
        "c5": "configs[4]: 1000 segments per GPU per step synthesised at 22.05/44.1/48 kHz round-robin, HBM-resident at the SOURCE rate -> "
              "device polyphase resampler -> BirdNET-v2.4-shaped model with f16 MFMA operands"}
    metric = {"c4": "5s/32kHz segments/sec (Perch-v2-sized)"}.get(args.config, "3s/48kHz segments/sec (BirdNET v2.4)")
    out = {
        "metric": metric, "value": round(value, 1), "unit": "segments/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE[args.precision].split(" ")[0], "data": "synthetic",
        "config": {"workload": workloads[args.config],
                   "segments_per_gpu": n_local, "micro_batch": args.micro_batch, "pre_warm_s": round(pre_warm_s, 2),
                   "gflop_per_segment": round((2 * info.macs_per_segment + info.mel_flops_per_segment) / 1e9, 3),
                   "fused_blocks": len(fused), "precision": args.precision, "dtype_note": DTYPE[args.precision],
                   **clocks.summary(),
                   "gemm": {"f32": "v_mfma_f32_16x16x4_f32 (exact f32 fmaf chains)",
                            "f16x3": "f32 operands split into f16 hi + lo, 3 x v_mfma_f32_16x16x32_f16 per product, f32 "
                                     "accumulate (|err| ~1e-7 of sum|a b|, same fp32 logit tolerance as the f32 MFMA path, "
                                     "tests/test_parity_gpu.py)",
                            "f16": "operands rounded to f16, v_mfma_f32_16x16x32_f16, f32 accumulate"}["f16x3" if args.precision == "auto" else args.precision]},
        "repeats": {"values": [round(v, 1) for v in all_values], "median_of_5": round(statistics.median(all_values), 1),
                    "note": "five regions of exactly --steps steps each; `value` is the first"},
    }
    out.update(analyse(clf, m, info, fused, stage_tot, layer_tot, segs_done, args.steps, slices_per_step, args.precision))
    extra = rank == 0 and world == 1 and not args.no_extra_legs
    if extra and args.config in ("c2", "c4"):
        # what birda's own batch sizes reach (-b 1..512, constants.rs:44,55; bh_default_batch_size = 256): the same 1 000 resident
        # segments through micro-batches of 256 and 512
        for mb in (256, 512):
            cx = clf.create_batch_context(mb)
            for _ in range(2):
                clf.forward_device(cx, x.data_ptr(), n_local, logits.data_ptr(), tk_idx.data_ptr(), tk_conf.data_ptr())
            cx.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                clf.forward_device(cx, x.data_ptr(), n_local, logits.data_ptr(), tk_idx.data_ptr(), tk_conf.data_ptr())
            cx.synchronize()
            out["value_at_batch_%d" % mb] = round(n_local * args.steps / (time.perf_counter() - t0), 1)
            cx.close()
    if extra and args.config == "c2":
        # (the child starts while this process holds its classifier: two processes share the device for those seconds, after
        #  every timed region of the headline)
        out["value_c4"] = None
        out.update(h2d_inclusive=None)
        legs = host_legs(clf, m, model_path, args.precision, tmp)
        out["end_to_end"] = legs.pop("end_to_end")
        out["h2d_inclusive"] = legs
    if extra and args.precision != "f32" and args.config in ("c2", "c4"):
        # the exact-f32 MFMA path, measured beside it on the same inputs for the same number of steps (not the headline)
        ctx.close()
        clf.close()
        clf = BirdClassifier(model_path, None, top_k=5, min_confidence=0.1, device=local_rank, precision="f32")
        ctx = clf.create_batch_context(args.micro_batch)
        for _ in range(2):
            step()
        ctx.set_profiling(True)
        e2 = timed_region(args.steps)
        st2 = {k: [ms, n] for k, (ms, n) in ctx.stage_ms().items()}
        ly2 = [[ms, n] for (ms, n) in ctx.layer_ms()]
        ctx.set_profiling(False)
        f32 = {"value": round(n_local * args.steps / e2, 1), "unit": "segments/s", "steps": args.steps,
               "gemm": "v_mfma_f32_16x16x4_f32 in every kernel"}
        f32.update(analyse(clf, m, info, clf.fused_blocks(), st2, ly2, n_local * args.steps, args.steps, slices_per_step, "f32"))
        out["f32_mfma_path"] = f32
        hip_logits["f32"] = logits[:64].cpu().numpy()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.config == "c5":       # the checker's path for this workload: oracle resampler -> oracle forward on the first segments
            out["cpu_baseline"] = cpu_baseline_c5(model_path, m, hip_logits)
        else:
            out["cpu_baseline"] = cpu_baseline(model_path, m.sample_count, m.sample_rate, hip_logits)
    elif rank == 0:
        out["cpu_baseline"] = None
    ctx.close()
    clf.close()
    if extra and args.config == "c2":
        out["small_calls"] = small_calls_leg(model_path, args.precision)
        out["value_c4"] = c4_child_leg(args.precision)      # (after this process has let go of its contexts)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def cpu_baseline_c5(model_path, m, hip_logits):
    """Config 5's CPU leg: the oracle's block-FFT resampler (rubato restatement) + forward on a bounded sample, and
    max |dlogit| of the f16-MFMA HIP path against it (tolerance 3e-3 of the logit scale, tests/test_parity_gpu.py)."""
    import numpy as np
    from birda_amd import pipeline, synth
    from oracle import oracle as O
    cores, _ = usable_cores()
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    om = O.OracleModel(model_path)
    rates = [22050, 44100, 48000]
    n = int(min(384, max(48, 2 * cores)))
    segs = np.zeros((n, m.sample_count), np.float32)
    t = time.perf_counter()
    for i in range(n):
        r = rates[i % 3]
        src = synth.synth_segments(1, pipeline.source_samples(m.sample_count, r, m.sample_rate), r, start=i % 16)[0]
        y = src if r == m.sample_rate else O.resample(src, r, m.sample_rate)
        segs[i, : min(len(y), m.sample_count)] = y[: m.sample_count]
    ref = om.forward(segs)
    dt = time.perf_counter() - t
    parity = {}
    for name, got in (hip_logits or {}).items():
        k = min(len(got), n)
        scale = float(max(1.0, np.abs(ref[:k]).max()))
        d = float(np.abs(got[:k] - ref[:k]).max())
        parity[name] = {"max_abs_dlogit": round(d, 6), "max_abs_logit": round(scale, 3), "relative": float(f"{d / scale:.3e}"), "segments": k,
                        "top1_agree": bool((got[:k].argmax(1) == ref[:k].argmax(1)).all())}
    return {"value": round(n / dt, 2), "unit": "segments/s", "cores": cores, "kind": "port",
            "sample": f"{n} mixed-rate segments: oracle resampler (serial) + oracle forward (OpenMP, {cores} threads), {dt:.1f} s",
            "max_abs_dlogit_vs_oracle": parity}


def bench_inproc_multi(args, m, model_path, tmp, devices, n_total, scaling, workload):
    """Several shards in ONE process through the C ABI (bh_multi_forward_device: one host thread + context + stream per
    shard, packed top-k gather by RCCL all-gather when every shard has its own device, else hipMemcpyDtoH per shard).
    Serves `--gpus N` without torch.distributed.run (one shard per device, 1 000 segments each: weak scaling) and
    `--config c3` (10 000 segments as 8 contiguous shards; with fewer than 8 GPUs the shards are logical devices on the
    GPUs that exist, SURVEY.md section 0, which measures the sharding / gather machinery, not 8-GPU scaling).
    A step = one bh_multi_forward_device call: every shard's forward + the gather + the unpack into bh_result rows; the
    call returns when all shards are done (the barrier), so the wall time of K calls is the max over shards."""
    import ctypes as C
    import numpy as np
    import torch
    from birda_amd import sharding, synth
    from birda_amd._lib import BhResult
    from birda_amd.multi import MultiClassifier
    G = len(devices)
    mc = MultiClassifier(model_path, None, devices=devices, top_k=5, min_confidence=0.1, precision=args.precision,
                         max_batch=min(args.micro_batch, -(-n_total // G)))
    uniq = synth.synth_segments(64, m.sample_count, m.sample_rate, start=0)
    xs, counts = [], []
    for g in range(G):
        lo, hi = sharding.shard_range(n_total, g, G)
        host = np.stack([uniq[j % 64] for j in range(lo, hi)])
        xs.append(torch.from_numpy(host).to(f"cuda:{devices[g]}"))
        counts.append(hi - lo)
    ptrs = [x.data_ptr() for x in xs]
    c_ptrs = (C.c_void_p * G)(*ptrs)
    c_counts = np.asarray(counts, np.uint64)
    c_res = (BhResult * n_total)()

    def step():
        rc = mc._L.bh_multi_forward_device(mc._h, c_ptrs, c_counts.ctypes.data, c_res)
        if rc != 0:
            raise RuntimeError(mc._L.bh_multi_last_error().decode())

    def sync_all():
        for d in sorted(set(devices)):
            torch.cuda.synchronize(d)
    res = mc.forward_device(ptrs, counts)
    pre_warm_s = pre_warm(step, sync_all)      # the chips at their sustained clock before the contract's region (see main())
    for _ in range(max(1, args.warmup)):
        step()
    clf0, ctx0 = mc.shard_classifier(0), mc.shard_context(0)
    vals, stage_tot, layer_tot = [], None, None
    for rep in range(5):
        if rep == 0:
            ctx0.set_profiling(True)     # HIP events around shard 0's launches, on its own stream
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        vals.append(n_total * args.steps / (time.perf_counter() - t0))
        if rep == 0:
            stage_tot = {k: [ms, n] for k, (ms, n) in ctx0.stage_ms().items()}
            layer_tot = [[ms, n] for (ms, n) in ctx0.layer_ms()]
            ctx0.set_profiling(False)
    n_gpus = len(set(devices))
    max_batch = min(args.micro_batch, -(-n_total // G))
    out = {"metric": "3s/48kHz segments/sec (BirdNET v2.4)", "value": round(vals[0], 1), "unit": "segments/s", "n_gpus": n_gpus,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(n_total / vals[0] * 1e3, 3), "higher_is_better": True,
           "scaling": scaling, "vs_baseline": None, "dtype": DTYPE[args.precision].split(" ")[0], "data": "synthetic",
           "config": {"workload": workload, "launch": "one process, bh_multi_* (no torch.distributed)",
                      "shards": G, "shard_devices": devices, "segments_per_shard": counts, "micro_batch": max_batch,
                      "gather_backend": mc.gather_backend(), "precision": args.precision,
                      "gflop_per_segment": round((2 * mc.info.macs_per_segment + mc.info.mel_flops_per_segment) / 1e9, 3),
                      "note": ("one shard per GPU" if n_gpus == G else
                               "shards that share a GPU are logical devices: this measures the sharding machinery, not multi-GPU scaling")},
           "repeats": {"values": [round(v, 1) for v in vals], "median_of_5": round(statistics.median(vals), 1),
                       "note": "five regions of exactly --steps steps each; `value` is the first (the one that carries the profiling events)"},
           "checks": {"results": len(res), "segments_with_predictions": sum(1 for r in res if r.predictions)}}
    fused = clf0.fused_blocks()
    out.update(analyse(clf0, m, mc.info, fused, stage_tot, layer_tot, counts[0] * args.steps, args.steps,
                       max(1, -(-counts[0] // max_batch)), args.precision))
    for k in ("roofline", "roofline_mel"):
        if k in out:
            out[k]["measured_on"] = "shard 0 (device %d), which shares its GPU with %d other shard(s)" % (devices[0], devices.count(devices[0]) - 1)
    out["cpu_baseline"] = None
    if n_gpus == 1 and not args.no_cpu_baseline:
        hip = {args.precision: clf0.predict_logits(ctx0, uniq)}          # the 64 distinct segments of the timed lists
        out["cpu_baseline"] = cpu_baseline(model_path, m.sample_count, m.sample_rate, hip)
    mc.close()
    print(json.dumps(out), flush=True)


def child_latency(argv):
    """`bench.py --child-latency model precision`: what a SMALL call lasts (forward_device + synchronise, device-resident input, median of
    30) for 1 / 8 / 20 / 32 segments -- a one-minute file is 20 -- with the library's default flags and with BH_FLAG_LOW_LATENCY
    (late blocks split over their expanded channels), in a process of its own."""
    import numpy as np
    import torch
    from birda_amd import modelfile as mf, synth
    from birda_amd.classifier import BirdClassifier
    model_path, precision = argv[0], argv[1]
    m = mf.read_model(model_path)
    base = synth.synth_segments(16, m.sample_count, m.sample_rate)
    res = {}
    for ll in (False, True):
        clf = BirdClassifier(model_path, None, precision=precision, low_latency=ll)
        for n in (1, 8, 20, 32):
            ctx = clf.create_batch_context(n)
            x = torch.from_numpy(np.ascontiguousarray(np.tile(base, (n // 16 + 1, 1))[:n])).cuda()
            logits = torch.empty((n, m.n_classes), device="cuda")
            idx = torch.empty((n, 5), dtype=torch.int32, device="cuda")
            conf = torch.empty((n, 5), device="cuda")
            ts = []
            for it in range(36):
                t = time.perf_counter()
                clf.forward_device(ctx, x.data_ptr(), n, logits.data_ptr(), idx.data_ptr(), conf.data_ptr())
                ctx.synchronize()
                if it >= 6:
                    ts.append(time.perf_counter() - t)
            res.setdefault(str(n), {})["low_latency" if ll else "default"] = round(statistics.median(ts) * 1e3, 3)
            ctx.close()
        clf.close()
    # ... and the reference's per-file loop on one-minute recordings (process_file per file, processor.rs:418-796) under the flag:
    # 50 PCM16 WAV files of 20 segments each -> 50 CSV files, one at a time
    files_rate = None
    try:
        from birda_amd import pipeline
        d = tempfile.mkdtemp(prefix="birda_bench_ll_")
        labels = os.path.join(d, "labels.txt")
        synth.write_labels(labels, m.n_classes)
        wavs = []
        for k in range(50):
            f = os.path.join(d, f"r{k:03d}.wav")
            synth.write_wav_pcm16(f, np.tile(base, (2, 1))[:20].reshape(-1), m.sample_rate)
            wavs.append(f)
        out_dir = os.path.join(d, "out")
        os.makedirs(out_dir)
        clf = BirdClassifier(model_path, labels, top_k=5, min_confidence=0.1, precision=precision, low_latency=True)
        rates = []
        for rep in range(3):
            t = time.perf_counter()
            n = sum(pipeline.process_file(clf, f, out_dir).segments for f in wavs)
            rates.append(n / (time.perf_counter() - t))
        clf.close()
        files_rate = {"value": round(sorted(rates)[1], 1), "unit": "segments/s", "what": "50 PCM16 WAV files of 20 segments each, bhh_process_file one at a time, "
                      "BH_FLAG_LOW_LATENCY classifier, median of 3 passes (the default-flags figure is end_to_end.short_files.one_file_at_a_time)"}
    except Exception as e:   # noqa: BLE001
        files_rate = {"error": str(e)[:200]}
    print(json.dumps({"unit": "ms per call (forward_device + synchronise, median of 30)", "segments": res, "one_minute_files_low_latency": files_rate,
                      "note": "BH_FLAG_LOW_LATENCY: forwards of at most 32 segments split the late blocks' expanded channels over 2-8 workgroups each "
                              "(fixed-order partial sums: a rounding of its own, launch-invariant within the regime)"}))


def small_calls_leg(model_path, precision):
    import subprocess
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child-latency", model_path, precision if precision in ("auto", "f16x3", "f32", "f16") else "auto"],
                           capture_output=True, text=True, timeout=180)
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        d["wall_s"] = round(time.perf_counter() - t0, 1)
        return d
    except Exception as e:   # noqa: BLE001
        return {"error": str(e)[:200]}


def child_files(argv):
    """`bench.py --child-files model labels precision out_dir wav...`: the files_pipelined leg in a process of its own (no torch)."""
    model_path, labels, precision, out_dir, wavs = argv[0], argv[1], argv[2], argv[3], argv[4:]
    from birda_amd import pipeline
    from birda_amd.classifier import BirdClassifier
    c = BirdClassifier(model_path, labels, top_k=5, min_confidence=0.1, precision=precision)
    pipeline.process_files_packed(c, wavs, out_dir)
    rates, ok = [], True
    for _ in range(3):
        t = time.perf_counter()
        res, status = pipeline.process_files_packed(c, wavs, out_dir)
        rates.append(sum(r.segments for r in res) / (time.perf_counter() - t))
        ok = ok and not any(status)
    c.close()
    print(json.dumps({"what": "%d files through bhh_process_files in a process of its own, median of 3 runs" % len(wavs),
                      "value": round(sorted(rates)[1], 1), "unit": "segments/s", "runs": [round(r, 1) for r in rates], "ok": ok}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child-files":
        child_files(sys.argv[2:])
    elif len(sys.argv) > 1 and sys.argv[1] == "--child-latency":
        child_latency(sys.argv[2:])
    else:
        main()
