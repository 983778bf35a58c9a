"""CPU: `python3 bench.py --gpus N` starts its own ranks (tuatara_amd/launch.py), the process fan-out that stands where the reference
starts its six recogniser threads and joins them (/root/reference/tuatara.cpp:461-475).  Driven here at world size 2 with a rank body
that needs no GPU (TUATARA_BENCH_STUB): exit codes, the single JSON line, the deadline, the ranks' stage watchdog."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(stub, extra=(), timeout=60):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TUATARA_RANK_CHILD")}
    env["TUATARA_BENCH_STUB"] = stub
    t0 = time.time()
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", *extra], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    return p.returncode, p.stdout, p.stderr, time.time() - t0


def test_two_ranks_one_json_line_status_zero():
    rc, out, err, _ = _run("ok")
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, out                                  # rank 0's chatter goes to stderr, the result line alone to stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert "rank 0 chatter" in err


def test_a_failing_rank_fails_the_launch_and_ends_the_others():
    rc, out, err, dt = _run("fail")
    assert rc == 7, (rc, err)
    assert out.strip() == ""
    assert "rank 1 left with status 7" in err
    assert dt < 20                                               # rank 0 was ended, not waited for (its stub sleeps 30 s)


def test_the_deadline_ends_a_hung_launch():
    rc, out, err, dt = _run("hang", ["--deadline", "2"])
    assert rc == 124, (rc, err)
    assert out.strip() == "" and "deadline" in err
    assert dt < 20


def test_the_stage_watchdog_names_the_stage():
    rc, out, err, dt = _run("stage", ["--deadline", "30"])
    assert rc == 3, (rc, err)
    assert '"stage": "stub: communicator set-up"' in err and '"rank": 1' in err
    assert dt < 20


def test_an_external_launcher_is_left_alone():
    """under torch.distributed.run (WORLD_SIZE set by the launcher) bench.py does not launch again: a rank is a rank"""
    env = dict(os.environ, TUATARA_BENCH_STUB="ok", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert p.returncode == 0, p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["n_gpus"] == 2


def test_run_ranks_needs_a_json_line():
    from tuatara_amd import launch
    import io
    script = os.path.join(ROOT, "tests", "data", "_silent_rank.py")
    os.makedirs(os.path.dirname(script), exist_ok=True)
    with open(script, "w") as f:
        f.write("print('no result here')\n")
    try:
        out, err = io.StringIO(), io.StringIO()
        rc, js = launch.run_ranks(script, [], 2, deadline_s=20, out=out, err=err)
        assert rc == 1 and js is None and out.getvalue() == ""
    finally:
        os.remove(script)
