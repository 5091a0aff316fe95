"""`python bench.py --gpus N` with no launcher around it (the shape of the driver's command when WORLD_SIZE is unset) must start N
ranks itself and print ONE line with n_gpus = N; under torch.distributed.run it must use the ranks it is given.  Run here on the CPU
through bench.py's rehearsal mode (toy env, gloo): the launcher, the rendezvous, ppo.train's multi-rank path and the
rollout / learner / all-reduce split [REF brax_rodent_run_ppo.py:27,43-47: batch scaled by the device count; SURVEY.md 8(d) config 3/4]."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, **env):
    e = dict(os.environ, RR_BENCH_CPU_REHEARSAL="1", **env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout       # ONE line on stdout: nothing else may greet there (gloo does, unless redirected)
    return json.loads(lines[0])


def test_gpus_2_without_a_launcher_starts_two_ranks():
    out = _run([sys.executable, "bench.py", "--gpus", "2", "--config", "3", "--steps", "2", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 2
    c = out["config"]
    assert c["env_steps_per_training_step"] == 32 * 5 * 2           # global batch of both ranks
    assert c["allreduce_s_per_training_step"] > 0 and c["allreduce_s_per_training_step"] < c["learner_s_per_training_step"]


def test_gpus_1_is_one_process_and_reports_no_allreduce():
    out = _run([sys.executable, "bench.py", "--gpus", "1", "--config", "3", "--steps", "1", "--warmup", "1"])
    assert out["n_gpus"] == 1 and out["config"]["allreduce_s_per_training_step"] == 0


def test_mismatch_between_gpus_and_world_size_is_refused():
    e = dict(os.environ, RR_BENCH_CPU_REHEARSAL="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
