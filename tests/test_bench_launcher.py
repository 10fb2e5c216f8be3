"""`python bench.py --gpus N` must start by itself (no torch.distributed.run around it): the parent spawns one child per
rank with the rendezvous environment set and never touches the GPU.  Rehearsed here on CPU with BENCH_DRY_RUN (gloo)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_rendezvous():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], BENCH_DRY_RUN="1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints the JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["dist_ranks"] == [0, 1]


def test_self_launch_propagates_a_failing_rank():
    r = _run(["--gpus", "3"], BENCH_DRY_RUN="fail_last")
    assert r.returncode != 0


def test_under_an_external_launcher_it_does_not_relaunch():
    # RANK present (what torch.distributed.run sets): the process is a rank itself
    r = _run(["--gpus", "1"], BENCH_DRY_RUN="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_PORT="29611")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["dist_ranks"] == [0]


def test_self_launch_deadline_ends_a_hung_rank():
    # a rank that never comes back (stuck in a collective): the launcher kills the ranks at its deadline and reports failure
    r = _run(["--gpus", "2"], BENCH_DRY_RUN="hang_last", BENCH_LAUNCH_TIMEOUT_S="20")
    assert r.returncode == 124, (r.returncode, r.stderr[-1000:])
