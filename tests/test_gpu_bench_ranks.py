"""bench.py --gpus 2 end to end (torchrun launch, transport agreement in exchange.init_comm, ghost refresh, assembly, ghost-row
reduction, checksum assert against the single-rank assembly, per-rank roofline blocks): two processes sharing the one GPU of a
gpurun box with IGX_BENCH_BACKEND=gloo (the host-callback transport; the product transport is RCCL).  The driver's multi-GPU
run must not be the first execution of this script."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("form,size", [("poisson", 64), ("elasticity", 48), ("cahnhilliard", 64), ("nsvms", 32)])
def test_bench_two_ranks_checksums(form, size):
    env = dict(os.environ, IGX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29900 + (os.getpid() + len(form) * 7) % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--form", form, "--size", str(size), "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["value"] > 0
    chk = line["config"]["checksum_check"]
    assert chk is not None and chk["size"] == size and max(chk["rel_diff"]) < 1e-9, chk
    assert line["config"]["partition"] == [1, 1, 2]
    assert line["config"]["transport"] == "host"
    per_rank = line["roofline_per_rank"]
    assert [r_["rank"] for r_ in per_rank] == [0, 1] and sum(r_["local_elements"] for r_ in per_rank) == size ** 3
    assert all(r_["frac"] is not None and r_["avg_launch_ms"] > 0 for r_ in per_rank)
