"""bench.py's launch logic, without a GPU: --gpus N started without a launcher must start N
ranks itself (one per GPU, rendezvous on 127.0.0.1), and a rank count that disagrees with
--gpus must be an error rather than a silently smaller job."""
import json
import os
import subprocess
import sys

from helpers import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env.update(extra)
    return env


def test_gpus_n_without_launcher_builds_n_rank_commands():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-launch",
                        "--master-port", "29777"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    plan = json.loads(r.stdout.strip().splitlines()[-1])["dry_launch"]
    assert len(plan) == 2
    for rank, ent in enumerate(plan):
        env, argv = ent["env"], ent["argv"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(rank)
        assert env["WORLD_SIZE"] == "2" and env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] == "29777"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert argv[1] == BENCH and "--dry-launch" not in argv
        assert argv[2:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--master-port", "29777"]


def test_rank_count_must_match_gpus():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "WORLD_SIZE=2" in r.stderr


def test_host_cpu_info_reads_this_machine():
    sys.path.insert(0, ROOT)
    import bench
    model, physical, logical = bench.host_cpu_info()
    assert isinstance(model, str) and model
    assert logical and logical >= 1
    assert physical is None or 1 <= physical <= logical


def _fake_ranks(*bodies):
    return [([sys.executable, "-c", b], {"RANK": str(r)}) for r, b in enumerate(bodies)]


def test_a_failing_rank_stops_the_others_within_seconds(tmp_path, capsys):
    """--gpus N started by bench.py itself: one rank that dies must not leave the others waiting in
    a rendezvous or a collective until an outer limit; the parent reports it and returns non-zero."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    cmds = _fake_ranks("import time; print('rank0 up', flush=True); time.sleep(120)",
                       # (dies after rank 0 has had time to start and print, however loaded the machine is)
                       "import sys, time; time.sleep(1.5); sys.stderr.write('boom from rank 1\\n'); sys.exit(3)",
                       "import time; time.sleep(120)")
    t0 = time.monotonic()
    rc, out0 = bench.supervise_ranks(cmds, rank_timeout=60.0, grace_s=2.0, log_dir=str(tmp_path))
    assert rc != 0
    assert time.monotonic() - t0 < 15.0
    assert "rank0 up" in out0
    err = capsys.readouterr().err
    assert "rank 1 exited with code 3" in err and "boom from rank 1" in err
    assert (tmp_path / "rank2.err").exists()


def test_ranks_that_hang_are_stopped_at_the_rank_timeout(tmp_path, capsys):
    import time
    sys.path.insert(0, ROOT)
    import bench
    # rank 1 ignores SIGTERM: it must be killed after the grace period
    cmds = _fake_ranks("import time; time.sleep(120)",
                       "import signal, time; signal.signal(signal.SIGTERM, signal.SIG_IGN); time.sleep(120)")
    t0 = time.monotonic()
    rc, _ = bench.supervise_ranks(cmds, rank_timeout=1.0, grace_s=1.0, log_dir=str(tmp_path))
    assert rc != 0
    assert time.monotonic() - t0 < 15.0
    assert "--rank-timeout" in capsys.readouterr().err


def test_all_ranks_ok_forwards_rank0_stdout(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    rc, out0 = bench.supervise_ranks(_fake_ranks("print('{\"value\": 1}')", "print('ignored')"), rank_timeout=30.0,
                                     log_dir=str(tmp_path))
    assert rc == 0 and out0.strip() == '{"value": 1}'
