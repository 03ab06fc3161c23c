"""bench.py's N > 1 path, rehearsed on the one GPU of the box: two ranks on cuda:0, collectives through gloo (host-staged),
a small corpus, and the bench's own check of the sharded result against an unsharded search of the same queries.  The
driver runs this path for real (RCCL, 2/4/8 GPUs) only at round end; a line that cannot be produced there is a lost
scaling curve (round 2: the sweep timing hook missed the packed entry point and the line divided by zero)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_on_one_gpu():
    env = dict(os.environ, ABC_BENCH_ONE_GPU="1", ABC_BENCH_BACKEND="gloo", ABC_BENCH_CHECK="1", ABC_BENCH_WATCHDOG="150")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29577", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
           "--nc", "196608"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["queries_per_step"] == 2 * 8192
    assert r["rehearsal_check"] == "sharded == unsharded (bit-exact)"
    assert r["value"] > 0 and r["roofline"]["achieved"] > 0 and r["roofline"]["kernel"].startswith("topk_sweep16_kernel")
