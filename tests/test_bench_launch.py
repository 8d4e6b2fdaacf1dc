"""bench.py --gpus N started plainly must start N ranks itself: launch self-test, no GPU needed."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_self_launches_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch", "--scaling",
                          "strong"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == "strong" and rec["config"]["scaling"] == "strong"
    assert rec["config"]["allreduce_check"] is True
    # the line proves by itself that the backend joined both ranks
    assert rec["config"]["ranks_seen"] == 2 and rec["config"]["backend"] == "gloo"
    assert rec["config"]["rank_devices"] == [-1, -1] and "nccl_version" in rec["config"]
    # the line says what the metric is: top-level roofline = forward + backward (the BASELINE metric),
    # operators nested, counter traffic with its source, HBM utilisation -- the keys exist in the self-test line too (no values)
    roof = rec["roofline"]
    assert {"frac", "achieved", "peak", "traffic", "traffic_source", "hbm_utilisation", "operators", "dominant",
            "bytes_per_sample"} <= set(roof)
    assert roof["frac"] is None and roof["bytes_per_sample"] == 2328 and set(roof["operators"]) == {"forward", "backward"}
    assert {"achieved", "frac", "ms_per_launch", "traffic"} <= set(roof["operators"]["backward"]) and "cpu_baseline" in rec


def test_roofline_record_prices_the_metric_itself():
    """roofline.frac is forward + backward (algorithmic bytes of both operators over both times), the per-operator figures
    are nested, and hbm_utilisation uses the counter bytes."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = 1 << 20
    r = bench.roofline_record(0.3, 0.5, 1164, 1164, n, traffic_fwd=1.0e9, traffic_bwd=2.0e9, traffic_source="test")
    assert abs(r["achieved"] - 2328 * n / 0.8e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["dominant"] == "backward" and abs(r["operators"]["backward"]["frac"] - 1164 * n / 0.5e-3 / 1e9 / 8000.0) < 1e-12
    assert r["traffic"] == 3.0e9 and abs(r["hbm_utilisation"] - 3.0e9 / 0.8e-3 / 1e9 / 8000.0) < 1e-12
    assert r["traffic_source"] == "test"


def test_bench_self_launch_eight_ranks_never_touches_the_gpu_in_the_parent():
    """--gpus 8: the launcher counts devices without the HIP runtime (shacira_amd.dist.visible_gpu_count) and spawns its
    ranks as child processes; rank 0's line reports the all-reduce over 8 ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--selftest-launch"], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 8 and rec["config"]["allreduce_check"] is True and rec["config"]["ranks_seen"] == 8


def test_visible_gpu_count_reads_the_environment(monkeypatch):
    from shacira_amd import dist as sdist
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,2,3,4,5,6,7")
    assert sdist.visible_gpu_count() == 8
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert sdist.visible_gpu_count() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "2")
    assert sdist.visible_gpu_count() == 1


def test_sweep_control_flow_four_ranks():
    """bench.py --gpus 4 --sweep (the one command for the first multi-GPU node-hour): every configuration runs in fresh child
    processes, prints its own JSON line tagged with config.sweep / nccl_algo / ar_chunks, and a summary line closes the run.
    Launch self-test mode (gloo, no kernels), three of the ten configurations to keep the CPU suite short."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "NCCL_ALGO")}
    names = ["weak_S1/allreduce_ring", "weak_S1/rs_ag", "strong_D/allreduce_chunks3"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--sweep", "--selftest-launch",
                          "--sweep-filter", ",".join(names)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(recs) == len(names) + 1
    for rec, name in zip(recs, names):
        assert rec["n_gpus"] == 4 and rec["config"]["sweep"] == name and rec["config"]["allreduce_check"] is True
    assert recs[0]["config"]["nccl_algo"] == "Ring" and recs[1]["config"]["collective"] == "rs_ag"
    assert recs[2]["config"]["ar_chunks"] == 3 and recs[2]["scaling"] == "strong"
    assert sorted(recs[-1]["sweep_summary"]) == sorted(names)
    # an unknown name is refused before anything is launched
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--sweep", "--selftest-launch",
                          "--sweep-filter", "weak_S1/nope"], env=env, capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "unknown configuration" in bad.stderr


def test_ranks_proof_refuses_a_short_world(monkeypatch):
    """ranks_proof() must end the run (exit code 3) when the all-reduce of ones does not see `world` ranks."""
    import importlib.util
    import pytest
    import torch
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    class FakeDist:
        @staticmethod
        def is_initialized():
            return True

        @staticmethod
        def all_reduce(t):
            t.fill_(3.0)          # three ranks answered ...

        @staticmethod
        def get_backend():
            return "gloo"

        @staticmethod
        def all_gather(out, t):
            for o in out:
                o.copy_(t)
    monkeypatch.setattr(bench, "dist", FakeDist)
    with pytest.raises(SystemExit) as exc:
        bench.ranks_proof(0, 4, torch.device("cpu"))   # ... of four
    assert exc.value.code == 3
    assert bench.ranks_proof(0, 3, torch.device("cpu"))["ranks_seen"] == 3
