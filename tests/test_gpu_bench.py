"""bench.py on the GPU, every leg of its default run switched on (short): the line must come out whole. (Round 6: the full run
crashed on a value read after the step's tensors had been released for the secondary configurations, while every flag
combination the tests and the profiling scripts used went through.)"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_default_legs_produce_the_whole_line():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--psnr-steps", "20",
                          "--nerf-steps", "10", "--cpu-samples", "4096", "--cpu-budget-s", "1"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["metric"].startswith("hash-grid samples/sec") and rec["n_gpus"] == 1 and rec["steps"] == 5
    assert rec["value"] > 0 and rec["ms_per_step"] > 0 and rec["dtype"] == "f32" and rec["vs_baseline"] is None
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and set(roof["operators"]) == {"forward", "backward"}
    assert rec["cpu_baseline"]["kind"] == "port" and rec["cpu_baseline"]["value"] > 0
    assert rec["config"]["plan_bytes"] > 0 and rec["config"]["backward_workspace_bytes"] > 0
    assert len(rec["other_configs_1gpu"]) >= 8 and "psnr" in rec and "psnr_nerf" in rec
