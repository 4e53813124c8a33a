"""`python bench.py --gpus N` must start its own ranks (the driver's SCALE command has no launcher
around it).  Run here on CPU with --dry-launch: gloo instead of RCCL, the PyTorch model definition
instead of the HIP path -- what is under test is the launcher, the node-balanced sharding of one
global batch, the counter reductions and the one-line JSON contract."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, capture_output=True, text=True, env=env,
                          timeout=300)


def test_self_launch_two_ranks_one_json_line():
    p = _run(["--gpus", "2", "--dry-launch", "--workload", "tiny", "--steps", "3", "--warmup", "1", "--repeats", "3",
              "--batches", "2", "--shard", "one-batch"])
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["config"]["rccl_ranks"] == 2 and r["dry_launch"] is True
    assert r["steps"] == 3 and r["warmup"] == 1 and r["scaling"] == "weak" and r["unit"] == "graphs/s"
    assert r["repeats"]["n"] == 3 and len(r["repeats"]["ms_per_step_all"]) == 3
    assert r["repeats"]["value_min"] <= r["value"] <= r["repeats"]["value_max"]
    # one global batch of 2 x 64 graphs per step, cut in two: the ranks together did 3 x 128 graphs
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 * r["steps"] - 3 * 128) < 1e-6 * 3 * 128
    assert r["config"]["shard"] == "one-batch"


def test_rank_distinct_batches_and_world_mismatch():
    p = _run(["--gpus", "2", "--dry-launch", "--workload", "tiny", "--steps", "2", "--warmup", "0", "--repeats", "1",
              "--batches", "2"])
    assert p.returncode == 0, p.stderr
    r = json.loads([l for l in p.stdout.splitlines() if l.strip()][-1])
    assert r["n_gpus"] == 2 and abs(r["value"] * r["ms_per_step"] * 1e-3 * 2 - 2 * 2 * 64) < 1e-3
    # a launcher-provided world size that disagrees with --gpus is refused, not silently accepted
    p = _run(["--gpus", "2", "--dry-launch", "--workload", "tiny"], {"WORLD_SIZE": "4", "RANK": "0"})
    assert p.returncode == 2 and "WORLD_SIZE=4" in p.stderr


def test_parent_makes_no_gpu_call_before_spawning():
    """Static check: everything main() does before self_launch() is argument parsing -- no torch import,
    no HIP library load (a parent that initialised the GPU and then started ranks would be refused on the pool)."""
    src = (ROOT / "bench.py").read_text()
    main_src = src[src.index("def main():"):]
    head = main_src[:main_src.index("self_launch(args.gpus")]
    assert "import torch" not in head and "load_library" not in head and "cuda" not in head
    launch = src[src.index("def self_launch("):src.index("def main():")]
    code = launch[launch.index('"""', launch.index('"""') + 3) + 3:]  # body without the docstring
    assert "os.exec" not in code and "torch" not in code and "load_library" not in code


def test_a_rank_that_dies_at_startup_fails_the_launch_within_seconds():
    """Rank 1 exits 3 before the rendezvous: rank 0 would wait in init_process_group for the collective
    timeout (minutes).  The parent polls every child, terminates the survivors and exits 1."""
    import time

    t0 = time.monotonic()
    p = _run(["--gpus", "2", "--dry-launch", "--workload", "tiny", "--steps", "2", "--warmup", "0", "--repeats", "1",
              "--batches", "2"], {"BENCH_TEST_FAIL_RANK": "1"})
    dt = time.monotonic() - t0
    assert p.returncode == 1, (p.returncode, p.stderr)
    assert dt < 30.0, dt
    assert "ranks failed" in p.stderr and "(1, 3)" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip()]  # no JSON line from a failed launch
