"""Two ranks on one GPU (gloo): the sharded, exchanged gradient of the native trainer equals
the single-process gradient of the whole batch, for the post-backward and the overlapped
exchange."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("mode", ["after", "overlap"])
def test_two_rank_gradient_equals_single_process(tmp_path, mode):
    port = str(29600 + os.getpid() % 1000 + (0 if mode == "after" else 1))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs, outs = [], []
    for r in range(2):
        out = tmp_path / f"r{r}.json"
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "ddp_worker.py"), str(r), "2", port,
                                       str(out), mode], env=env))
    try:
        for p in procs:
            assert p.wait(timeout=90) == 0
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    res = [json.loads(o.read_text()) for o in outs]
    assert res[0]["norm"] > 0
    assert res[0]["rel_err"] < 1e-4, res
    assert max(r["rank_spread"] for r in res) == 0.0


def test_bench_exchange_over_rccl_single_rank():
    """bench.py --force-exchange: the overlapped gradient exchange issued as real RCCL collectives (one
    rank is all a one-GPU box allows) inside the benchmark's own step loop; the line it prints must be
    a valid result with a finite loss."""
    root = os.path.dirname(HERE)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29700 + os.getpid() % 200))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-exchange", "--steps", "3",
                        "--warmup", "1", "--no-cpu-baseline", "--no-profile"],
                       env=env, capture_output=True, text=True, timeout=110)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["rccl_ranks"] == 1 and out["n_gpus"] == 1
    assert out["data"] == "synthetic"
    loss = out["config"]["final_l1_loss"]
    assert loss == loss and 0.0 < loss < 2.0
