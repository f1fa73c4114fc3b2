"""`python bench.py --gpus N` without a torchrun environment must start its N ranks itself:
fresh child processes, rendezvous on 127.0.0.1, rank 0's JSON line relayed, a failing child reported by the exit code.
CPU-only: `--launch-rehearsal` runs the launcher, the process group, the fences and the max-over-ranks reduction around
an empty step (no model, no GPU, value null)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "HERMNET_BENCH_CHILD")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300)


def test_plain_command_line_starts_its_own_ranks():
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-rehearsal")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout                      # ONE JSON line, nothing else on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == "strong" and out["value"] is None


def test_single_rank_rehearsal_needs_no_launcher():
    p = _run("--launch-rehearsal")
    assert p.returncode == 0, p.stderr[-2000:]
    assert json.loads(p.stdout.strip())["n_gpus"] == 1


def test_failing_rank_gives_nonzero_exit_and_no_line():
    # without a GPU the real workload cannot start: every rank dies in torch.cuda.set_device
    p = _run("--gpus", "2", "--steps", "1", "--warmup", "0", "--no-secondary", env={"HIP_VISIBLE_DEVICES": "",
                                                                                  "CUDA_VISIBLE_DEVICES": ""})
    assert p.returncode != 0
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]


def test_mismatched_world_is_refused_inside_a_job():
    p = _run("--gpus", "4", "--launch-rehearsal", env={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert p.returncode != 0 and "inside a 2-rank job" in p.stderr


def test_self_launch_is_refused_under_a_profiler():
    # a profiler attached to the relay process would see an idle parent and miss the ranks
    p = _run("--gpus", "2", "--launch-rehearsal", env={"HERMNET_PROFILER_HINT": "/opt/rocm/lib/librocprofiler-sdk-tool.so"})   # (the real variables would load the tool)
    assert p.returncode == 2 and "Profile one rank instead" in p.stderr and not p.stdout.strip()


def test_hung_child_job_is_killed_at_the_launch_timeout():
    # HERMNET_REHEARSAL_HANG makes every rank of the rehearsal sleep: the launcher must end the job and say so
    p = _run("--gpus", "2", "--launch-rehearsal", "--launch-timeout", "8", env={"HERMNET_REHEARSAL_HANG": "600"})
    assert p.returncode == 124 and "launch-timeout" in p.stderr
