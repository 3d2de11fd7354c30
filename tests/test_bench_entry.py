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
