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
