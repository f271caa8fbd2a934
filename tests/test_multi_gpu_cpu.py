"""N>1 path on CPU: world_size-2 (and 3) gloo runs of the row-sharded render + reduce."""
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,exchange", [(2, "gather"), (2, "reduce"), (3, "gather"), (3, "reduce"), (4, "gather")])
def test_row_sharded_exchange_is_bit_identical(tmp_path, world, exchange):
    port = _free_port()
    out = tmp_path / "result.txt"
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(out), exchange], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    logs = []
    for p in procs:
        o, _ = p.communicate(timeout=240)
        logs.append(o.decode(errors="replace"))
        assert p.returncode == 0, logs[-1][-2000:]
    assert out.read_text() == "OK", logs


def test_bench_launches_its_own_ranks_and_rejects_a_mismatched_world():
    """`python bench.py --gpus N` started plainly (no RANK in the environment) spawns the N ranks itself as a child
    `torch.distributed.run` and relays the exit code.  Without a GPU every rank must stop with the no-CPU-fallback
    message -- which here proves that N ranks were started.  A pre-launched world of the wrong size is refused."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["PT_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    text = r.stderr + r.stdout
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert text.count("bench.py needs a GPU") >= 2, text[-3000:]
    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "does not match --gpus" in (r.stderr + r.stdout)
