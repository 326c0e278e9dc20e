"""bench.py --gpus 2 end to end (its own rank launch, transport agreement in exchange.init_comm, ghost refresh, assembly,
ghost-row reduction, checksum assert against the single-rank assembly, per-rank roofline blocks): two processes sharing the one GPU
of a gpurun box with IGX_BENCH_BACKEND=gloo (the host-callback transport; the product transport is RCCL).  The driver's multi-GPU
run must not be the first execution of this script -- neither as plain `python bench.py --gpus N` (bench.py starts the ranks
itself) nor under torch.distributed.run."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


GRID = {2: [1, 1, 2], 4: [1, 2, 2], 8: [2, 2, 2]}


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfake_rccl.so")


def _env(transport):
    """gloo for torch.distributed (ranks share the one GPU); the library's exchange on the host-callback transport, or on its
    product transport (grouped ncclSend / ncclRecv) bound to tests/fake_rccl's double of librccl.so"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(IGX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if transport == "rccl":
        env.update(IGX_BENCH_TRANSPORT="rccl", IGX_RCCL_LIB=FAKE_RCCL, FAKE_RCCL_TIMEOUT_S="240")
    return env


def _check_line(r, form, size, ranks=2, transport="host"):
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert len(r.stdout.strip().splitlines()) == 1, r.stdout[-2000:]      # stdout carries the line and nothing else (gloo / RCCL notes go to stderr)
    line = json.loads(lines[0])
    assert line["n_gpus"] == ranks and line["value"] > 0
    chk = line["config"]["checksum_check"]
    assert chk is not None and chk["size"] == size and max(chk["rel_diff"]) < 1e-9, chk
    assert line["config"]["partition"] == GRID[ranks]
    assert line["config"]["transport"] == transport and line["config"]["transport_ranks"] == ranks
    assert line["config"]["rccl_ranks"] == (ranks if transport == "rccl" else None)      # ncclCommCount of the library's communicator
    # what the face-first decision rested on: measured over the transport at the communicator's creation (RCCL branch), the
    # constant for the host transport; and the passes the last assembly made because of it
    cfg = line["config"]
    if transport == "rccl":
        assert cfg["exchange_link_source"] == "measured" and cfg["exchange_link_gbs"] > 0 and "face message" in cfg["exchange_link_probe"], cfg
    else:
        assert cfg["exchange_link_source"] == "constant" and cfg["exchange_link_gbs"] == 60.0, cfg
    assert cfg["face_passes"] in (1, 2, 3, 4), cfg
    per_rank = line["roofline_per_rank"]
    assert [r_["rank"] for r_ in per_rank] == list(range(ranks)) and sum(r_["local_elements"] for r_ in per_rank) == size ** 3
    assert all(r_["frac"] is not None and r_["avg_launch_ms"] > 0 for r_ in per_rank)
    assert line["ms_per_step_min"] <= line["ms_per_step_median"] <= line["ms_per_step_max"] and len(line["per_step"]["ms"]) == line["steps"]


@pytest.mark.gpu
@pytest.mark.parametrize("form,size,transport", [("poisson", 64, "rccl"), ("elasticity", 48, "host"), ("cahnhilliard", 64, "host"), ("nsvms", 32, "rccl")])
def test_bench_two_ranks_checksums(form, size, transport):
    """The way the driver starts it: `python bench.py --gpus 2 ...`, no launcher, WORLD_SIZE unset."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--form", form, "--size", str(size), "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env(transport), capture_output=True, text=True, timeout=900)
    _check_line(r, form, size, 2, transport)


@pytest.mark.gpu
@pytest.mark.parametrize("ranks,form,size", [(8, "cahnhilliard", 48), (4, "nsvms", 32), (8, "poisson", 64)])
def test_bench_four_and_eight_ranks(ranks, form, size):
    """[1,2,2] and [2,2,2] as processes sharing the one GPU: the ghost refresh of a nonlinear form (its lists differ in length from the
    reduction's: the exchange buffers of round 3 did not survive that) and the whole N-rank flow of the driver's SCALE run --
    through the library's RCCL branch (comm.hpp kind == 1) on the test double: rccl_ranks == N in the line"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1", "--form", form, "--size", str(size), "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=_env("rccl"), capture_output=True, text=True, timeout=900)
    _check_line(r, form, size, ranks, "rccl")


@pytest.mark.gpu
def test_bench_two_ranks_under_torchrun():
    env = dict(os.environ, IGX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    port = 29900 + os.getpid() % 90
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--form", "poisson", "--size", "64", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    _check_line(r, "poisson", 64)


def test_bench_more_ranks_than_gpus_fails_loudly():
    """--gpus 2 over RCCL on a box with fewer than two GPUs (this container: none; a gpurun box: one): every rank leaves with a
    message before any rendezvous -- a non-zero exit code within seconds, not a hang."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs here: the RCCL path itself would run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "IGX_BENCH_BACKEND")}
    t = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "16", "--no-cpu-baseline"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "needs 2 GPUs" in r.stderr, r.stderr[-2000:]
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert time.time() - t < 240


def test_bench_world_size_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
