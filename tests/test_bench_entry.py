"""bench.py's launch contract: `--gpus N` is honoured or refused, never ignored (round 2 parsed it and ran one GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=600):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_more_gpus_than_devices_is_refused_without_a_json_line():
    import torch
    n = torch.cuda.device_count()
    p = _run(["--gpus", str(n + 1), "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert f"--gpus {n + 1} asked for, but {n} HIP device(s) are visible" in p.stderr
    assert "n_gpus" not in p.stdout
    p = _run(["--gpus", "0"])
    assert p.returncode != 0 and "--gpus must be >= 1" in p.stderr


def test_gpus_and_world_size_must_agree():
    p = _run(["--gpus", "2"], env={"WORLD_SIZE": "4", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=4" in p.stderr


@pytest.mark.gpu
def test_in_process_multi_shard_line_reports_the_devices_that_ran():
    """`--config c3` through bh_multi_* on whatever is visible: n_gpus = distinct devices that hosted a shard."""
    import torch
    n = torch.cuda.device_count()
    p = _run(["--config", "c3", "--gpus", str(n), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--micro-batch", "256"])
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == n == len(set(line["config"]["shard_devices"]))
    assert line["config"]["shards"] == 8 and sum(line["config"]["segments_per_shard"]) == 10000
    assert line["config"]["gather_backend"].startswith(("rccl", "host"))
    assert line["checks"]["results"] == 10000 and line["value"] > 0
    assert line["roofline"]["frac"] > 0 and line["roofline_mel"]["bound"] == "hbm"


@pytest.mark.gpu
def test_one_rank_per_gpu_launch_as_the_driver_runs_it(tmp_path):
    """The N > 1 launch of the contract -- `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` -- on whatever is
    here: two ranks sharing device 0 (BIRDA_BENCH_DRYRUN_ONE_DEVICE=1: gloo instead of RCCL for the gather and the timing
    reduction, everything else as on an 8-GPU node).  Rank 0 prints ONE line: the whole job's segments over the slowest rank's time."""
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        e.pop(k, None)
    e["BIRDA_BENCH_DRYRUN_ONE_DEVICE"] = "1"
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29517", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=e, timeout=900, cwd=str(tmp_path))
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 2
    assert line["config"]["segments_per_gpu"] == 1000
    assert abs(line["value"] - 2 * 1000 / (line["ms_per_step"] / 1e3)) <= 1e-3 * line["value"]     # whole-job aggregate
    assert "cpu_baseline" not in line or line["cpu_baseline"] is None or "value" not in (line["cpu_baseline"] or {})   # rank 0 at N = 1 only
