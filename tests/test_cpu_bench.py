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
