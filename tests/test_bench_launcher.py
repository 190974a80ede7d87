"""bench.py's own multi-rank launch (`python bench.py --gpus N` with no torchrun around it), rehearsed on the CPU with
a stand-in step: N rank processes are started, meet at the barriers, and rank 0 reports n_gpus = N with the MAX over
ranks as the step time."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=180)


def test_self_launch_two_ranks_stub():
    p = _run(["--gpus", "2", "--steps", "5", "--warmup", "1"], ARREAU_BENCH_STUB="1")
    assert p.returncode == 0, p.stderr
    lines = p.stdout.splitlines()
    assert len(lines) == 1, p.stdout  # ONE line on stdout: rank 0's result, nothing a library printed (gloo announces itself)
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and len(d["per_rank_ms"]) == 2
    assert d["per_rank_ms"][1] > d["per_rank_ms"][0]  # the stub's rank 1 is slower ...
    assert d["ms_per_step"] >= max(d["per_rank_ms"])  # ... and the reported time covers the slowest rank
    # the line proves its ranks: backend and size as the process group sees them, one record per rank from distinct processes
    r = d["ranks"]
    assert r["backend"] == "gloo" and r["world_size"] == 2 and r["data_path_collectives"] == 0
    assert [x["rank"] for x in r["per_rank"]] == [0, 1] and len({x["pid"] for x in r["per_rank"]}) == 2


def test_self_launch_refuses_when_gpus_are_missing():
    """Without the stub, asking for more GPUs than are visible must fail loudly (never a silent 1-GPU run)."""
    sys.path.insert(0, ROOT)
    import bench
    want = bench.visible_gpu_count() + 2
    p = _run(["--gpus", str(want), "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0
    assert "GPU(s) are visible" in p.stderr and not p.stdout.strip()


def test_gpu_count_comes_from_sysfs_and_visibility_lists(monkeypatch):
    """The launcher parent counts devices without calling into HIP: KFD topology + *_VISIBLE_DEVICES."""
    sys.path.insert(0, ROOT)
    import bench
    base = bench.visible_gpu_count()
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0")
    assert bench.visible_gpu_count() == min(base, 1)
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count() == 0
    src = open(BENCH).read()
    body = src[src.index("def launch_ranks"):src.index("def main()")]
    assert "import torch" not in body and "device_count" not in body


def test_world_size_mismatch_is_an_error():
    p = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], ARREAU_BENCH_STUB="1", WORLD_SIZE="2", RANK="0")
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
