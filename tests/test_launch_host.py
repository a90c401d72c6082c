"""`bench.py --gpus N` starts its own N ranks (katome_amd/launch.py): launch logic only, no GPU call in the parent.
CPU tests: the rendezvous environment every child gets, one relayed JSON line, failure propagation, and bench.py itself
refusing to report a one-GPU number for --gpus 2."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _child(tmp_path, body):
    p = tmp_path / "child.py"
    p.write_text(textwrap.dedent(body))
    return [sys.executable, str(p)]


def test_children_get_rank_env_and_one_line_is_relayed(tmp_path):
    from katome_amd.launch import launch_ranks, relay_one_json_line
    argv = _child(tmp_path, """
        import json, os, sys
        rank = int(os.environ["RANK"])
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "HSA_ENABLE_IPC_MODE_LEGACY")}
        open(os.path.join(%r, "env%%d.json" %% rank), "w").write(json.dumps(env))
        print("banner from rank %%d" %% rank)                  # only rank 0's stdout is relayed
        if rank == 0:
            print(json.dumps({"n_gpus": int(os.environ["WORLD_SIZE"]), "value": 1.0}))
    """ % str(tmp_path))
    rc, out = launch_ranks(2, argv, timeout=60)
    assert rc == 0
    envs = [json.load(open(tmp_path / ("env%d.json" % r))) for r in range(2)]
    assert [e["RANK"] for e in envs] == ["0", "1"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1"]
    assert all(e["WORLD_SIZE"] == "2" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert envs[0]["MASTER_PORT"] == envs[1]["MASTER_PORT"] and int(envs[0]["MASTER_PORT"]) > 0
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)
    assert "rank 1" not in out                                 # other ranks' stdout never reaches the relayed text
    line = relay_one_json_line(out)
    assert json.loads(line) == {"n_gpus": 2, "value": 1.0}
    assert sum(1 for l in out.splitlines() if l.strip().startswith("{")) == 1


def test_a_failing_rank_fails_the_job_and_stops_the_others(tmp_path):
    from katome_amd.launch import launch_ranks
    argv = _child(tmp_path, """
        import os, sys, time
        if os.environ["RANK"] == "1":
            sys.exit(7)
        time.sleep(60)                                         # would hang the job if it were not terminated
    """)
    rc, out = launch_ranks(3, argv, timeout=30)
    assert rc == 7 and out == ""


def test_timeout_is_a_failure(tmp_path):
    from katome_amd.launch import launch_ranks
    argv = _child(tmp_path, "import time; time.sleep(60)")
    rc, _ = launch_ranks(2, argv, timeout=1.0)
    assert rc == 124


def _run_bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout)


def test_bench_gpus_2_without_a_launcher_starts_two_ranks_and_never_reports_one_gpu():
    """On a box without two GPUs (this container has none) `--gpus 2` must fail loudly: non-zero exit, no JSON line on
    stdout, both ranks named on stderr -- not a silent single-GPU run printing n_gpus=1 (round 1's behaviour)."""
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "1000", "--no-cpu-baseline"])
    err = r.stderr.decode(errors="replace")
    assert r.returncode != 0
    assert r.stdout.strip() == b""
    assert "--gpus 2" in err and ("rank 0" in err or "rank 1" in err)


def test_bench_rejects_a_world_that_is_not_gpus():
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                   {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == b""
    assert b"--gpus 2 but WORLD_SIZE=1" in r.stderr
    r = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                   {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and b"--gpus 1 but WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with exactly one GPU")
    r = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--reads", "100000", "--no-cpu-baseline"])
    assert r.returncode != 0 and r.stdout.strip() == b""
    assert b"only 1 GPU(s) visible" in r.stderr
