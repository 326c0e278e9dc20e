"""CPU tests of the multi-GPU path: (a) the ghost-row exchange plan is consistent across ranks (every send has
exactly one matching receive of the same size), (b) the point-to-point pattern of petiga_amd/exchange.py runs
deadlock-free on 2 gloo ranks, (c) the oracle's rank-local assemblies add up to the single-rank matrix."""
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_api as O
import petiga_amd as P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def make_rank(dim, dof, p, N, periodic, size, rank):
    g = P.IGX(dim, dof)
    for i in range(dim):
        g.axis_uniform(i, p, N[i], periodic=periodic[i])
    g.set_comm(size, rank)
    g.setup()
    return g


@pytest.mark.parametrize("dim,dof,p,N,periodic", [(3, 1, 3, (8, 8, 8), (0, 0, 0)), (3, 2, 2, (9, 7, 8), (0, 0, 0)), (2, 1, 2, (8, 12), (0, 0)),
                                                   (3, 1, 2, (8, 8, 8), (1, 0, 1)), (1, 3, 3, (16,), (0,))])
@pytest.mark.parametrize("size", [2, 4, 8])
def test_exchange_plan_is_consistent(dim, dof, p, N, periodic, size):
    # (dim 1, 8 ranks: 2 elements of degree 3 per rank -- a rank's 3 ghost nodes reach past its neighbour's 2 owned ones; since round 4
    #  the layer is split over the two ranks above, one message each)
    ranks = [make_rank(dim, dof, p, N, [bool(x) for x in periodic], size, r) for r in range(size)]
    sends = {(r, peer): (m, v) for r, g in enumerate(ranks) for peer, m, v in g.neighbors(True)}
    recvs = {(peer, r): (m, v) for r, g in enumerate(ranks) for peer, m, v in g.neighbors(False)}
    assert sends == recvs
    for r, g in enumerate(ranks):
        peers = [peer for peer, _, _ in g.neighbors(True)]
        assert len(peers) == len(set(peers)) and r not in peers        # one message per peer, never to itself
        assert len(peers) <= (7 if dim == 3 else 3)
    # every node is owned by exactly one rank, ghost rows are exactly the not-owned ones
    sz = ranks[0].sizes()
    owners = np.zeros(sz["node_sizes"][:dim], dtype=int)
    for g in ranks:
        s = g.sizes()
        sl = tuple(slice(s["node_lstart"][d], s["node_lstart"][d] + s["node_lwidth"][d]) for d in range(dim))
        owners[sl] += 1
    assert np.all(owners == 1)


def test_oracle_rank_local_assemblies_add_up():
    def build(size, rank):
        g = O.OracleIGA(3, 1)
        for i in range(3):
            g.axis_uniform(i, 2, 6)
        if size > 1:
            g.set_partition(size, rank)
        g.setup()
        for d in range(3):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 1.0)
        return g.compute_system("orc_form_poisson")
    A1, b1 = build(1, 0)
    acc, accb = np.zeros_like(A1.val), np.zeros_like(b1)
    for r in range(8):
        A, b = build(8, r)
        acc += A.val
        accb += b
    assert np.abs(acc - A1.val).max() < 1e-13 * np.abs(A1.val).max()
    assert np.abs(accb - b1).max() < 1e-13 * np.abs(b1).max()


WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
import petiga_amd as P
from petiga_amd import exchange
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
g = P.IGX(3, 1)
for i in range(3):
    g.axis_uniform(i, 3, 8)
g.set_comm(world, rank)
g.setup()
s, r = exchange.plan(g)
sb = [torch.full((n,), float(rank * 100 + p), dtype=torch.float64) for p, n in s]
rb = [torch.empty(n, dtype=torch.float64) for p, n in r]
exchange.p2p_exchange(sb, [p for p, _ in s], rb, [p for p, _ in r])
for (p, n), b in zip(r, rb):
    assert b.numel() == n and torch.all(b == float(p * 100 + rank)), (rank, p)
tot = torch.tensor([float(sum(n for _, n in s)), float(sum(n for _, n in r))])
dist.all_reduce(tot)
assert tot[0] == tot[1] and tot[0] > 0
dist.destroy_process_group()
print("rank", rank, "ok", len(s), len(r))
'''


@pytest.mark.parametrize("world", [2])
def test_p2p_pattern_on_gloo(world, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


@pytest.mark.parametrize("p,N,size", [(3, 16, 4), (3, 16, 5), (3, 16, 6), (2, 8, 4), (2, 9, 8), (1, 8, 8), (3, 7, 7), (3, 12, 5)])
def test_ghost_rows_reach_their_owner_however_thin_the_ranks(p, N, size):
    """Ranks with fewer than p elements own fewer nodes than their lower neighbour's ghost layer: the layer is split over the ranks
    above, one message each (the reference's stash routes every row to its true owner, src/petiga.c:1172-1208).  Every ghost node of
    every rank is sent exactly once, to the rank that owns it, and lands on the owner's local index of the same global node; the
    send list of a rank and the receive lists of its peers agree in sizes."""
    import petiga_amd as P
    gs = []
    for rank in range(size):
        g = P.IGX(1, 1)
        g.set_comm(size, rank)
        g.axis_uniform(0, p, N)
        g.setup()
        gs.append(g)
    sizes = [g.sizes() for g in gs]
    sends = [g.neighbors(True) for g in gs]
    recvs = [g.neighbors(False) for g in gs]
    for r in range(size):
        nghost = sizes[r]["node_gwidth"][0] - sizes[r]["node_lwidth"][0]
        assert sum(v for _, _, v in sends[r]) == nghost          # (vec_doubles = nodes of the piece at dof 1)
        peers = [q for q, _, _ in sends[r]]
        assert len(set(peers)) == len(peers) and all(q > r for q in peers)
        for q, m, v in sends[r]:
            back = [(m2, v2) for r2, m2, v2 in recvs[q] if r2 == r]
            assert back == [(m, v)], (r, q, back)
        # the thin case really occurs in this list: some rank sends to a rank two or more above
    if N // size < p and size > 2:
        assert any(q - r >= 2 for r in range(size) for q, _, _ in sends[r])


def test_a_periodic_ghost_layer_never_wraps_onto_its_own_rank():
    """The reference's ghost indices wrap unconditionally (src/petigagrid.c:160-163), and so do the exchange's: on a split periodic
    axis ranks are taken unwrapped (exchange.hpp), a ghost layer may cross the seam and reach several thin ranks.  The one thing the
    exchange cannot serve -- a layer that reaches around to its own rank, IGX_ERR_SUP 56 in exchange_supported() -- needs the OTHER
    ranks of the axis to own fewer than p nodes together.  IGA_Distribute balances the elements (src/petigapart.c:170-202), so the
    others own at least floor(N / 2) >= p nodes once the axis has the 2p + 1 functions a split periodic axis needs to hold its own
    stencil (refused at IGXSetUp otherwise).  Swept here: every degree, element count and rank count -- never refused; every ghost
    node is sent exactly once and the send / receive lists of all ranks agree.  (The wrapped thin-rank case runs against the oracle
    on the GPU: tests/test_gpu_comm.py, poisson-p2-5ranks-periodic-wrap.)"""
    import petiga_amd as P
    for p in (1, 2, 3, 4):
        for N in range(2 * p + 1, 2 * p + 8):
            for size in range(2, N + 1):
                gs = []
                for rank in range(size):
                    g = P.IGX(1, 1)
                    g.set_comm(size, rank)
                    g.axis_uniform(0, p, N, periodic=True)
                    g.setup()
                    gs.append(g)
                sends = [g.neighbors(True) for g in gs]          # (raises IGXError 56 if a layer wrapped onto its own rank)
                recvs = [g.neighbors(False) for g in gs]
                for r in range(size):
                    sz = gs[r].sizes()
                    assert sum(v for _, _, v in sends[r]) == sz["node_gwidth"][0] - sz["node_lwidth"][0]
                    assert all(q != r for q, _, _ in sends[r])
                    for q, m, v in sends[r]:
                        assert [(m2, v2) for r2, m2, v2 in recvs[q] if r2 == r] == [(m, v)], (p, N, size, r, q)
    # fewer than 2p + 1 functions on a split periodic axis: refused at set-up, with a message, before any exchange entry point
    g = P.IGX(1, 1)
    g.set_comm(2, 0)
    g.axis_uniform(0, 3, 4, periodic=True)
    with pytest.raises(P.IGXError) as e:
        g.setup()
    assert "2p+1" in str(e.value)
