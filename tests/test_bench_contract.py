"""bench.py's contract for N > 1: the default `--gpus N` line must be on a config BASELINE.json names — configs[1] (1e9 rows,
one GPU) at N = 1, configs[4] (1e10 rows row-sharded over 8 GPUs = 1.25e9 per GPU) at N = 8 — and say so in config.workload.  Host logic here;
`-m gpu`: the whole N = 2 path (launcher, sharding, the reduce, the JSON line) as two HIP processes on the box's one GPU over gloo."""
import argparse
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**kw):
    return argparse.Namespace(**dict(dict(rows=None, total_rows=None), **kw))


def test_default_rows_are_the_baseline_configs():
    import bench
    rows, scaling, what = bench.rows_of_rank(_args(), 0, 1)
    assert rows == 1_000_000_000 and scaling == "weak" and "configs[1]" in what
    for world in (2, 4, 8):
        per_rank = [bench.rows_of_rank(_args(), r, world) for r in range(world)]
        assert {p[0] for p in per_rank} == {1_250_000_000} and per_rank[0][1] == "weak" and "configs[4]" in per_rank[0][2]
    assert 8 * bench.rows_of_rank(_args(), 0, 8)[0] == 10_000_000_000 and "1e10 rows row-sharded over 8 GPUs" in bench.rows_of_rank(_args(), 0, 8)[2]
    # --rows: per GPU, weak; --total-rows: one table split over the ranks, strong, every row on exactly one rank
    assert bench.rows_of_rank(_args(rows=3e6), 1, 4)[:2] == (3_000_000, "weak")
    parts = [bench.rows_of_rank(_args(total_rows=1e9 + 3), r, 8) for r in range(8)]
    assert sum(p[0] for p in parts) == 1_000_000_003 and max(p[0] for p in parts) - min(p[0] for p in parts) <= 1 and parts[0][1] == "strong"
    assert json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"][4].startswith("1e10-row 2-D count+mean 256")


def _bench(*argv, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_rank_dry_run_on_one_gpu_over_gloo():
    line = _bench("--gpus", "2", "--backend", "gloo", "--rows", "3e7", "--steps", "3", "--warmup", "1", "--cpu-rows", "2e6")
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["unit"] == "rows/s" and line["value"] > 0
    cfg = line["config"]
    assert cfg["rows_per_gpu"] == 30_000_000 and cfg["total_rows"] == 60_000_000 and cfg["backend"] == "gloo" and "row-sharded x2" in cfg["parallelism"]
    assert abs(line["value"] - 60_000_000 / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert line["roofline"]["allreduce_ms"] > 0 and 0 < line["roofline"]["frac_incl_allreduce"] < 1
    assert line["cpu_baseline"]["parity_on_sample"] is True
    (c2,) = line["configs"]
    assert c2["config"] == "count2d" and c2["rows"] == 60_000_000 and c2["rows_per_s"] > 0 and c2["roofline"]["bytes_per_row"] == 16


@pytest.mark.gpu
def test_strong_scaling_split_of_one_table():
    line = _bench("--gpus", "2", "--backend", "gloo", "--total-rows", "40000001", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-configs")
    assert line["scaling"] == "strong" and line["config"]["total_rows"] == 40_000_001 and line["config"]["rows_per_gpu"] == 20_000_001 and "configs" not in line
