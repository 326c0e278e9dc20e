"""CPU tests of the product's host side (no GPU needed): the C++ set-up behind the C ABI must agree
bit-exactly with the oracle on every integer quantity (ranges, partition, colouring) and the library must
export every symbol include/petiga_amd.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_api as O
import petiga_amd as P
from common import make_pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "petiga_amd.h")).read()
    names = set(re.findall(r"\b(IGX[A-Za-z0-9]+)\s*\(", hdr))
    assert len(names) > 50
    L = C.CDLL(os.path.join(ROOT, "petiga_amd", "libpetiga_amd.so"))
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


@pytest.mark.parametrize("dim,p,N,C_,periodic", [(1, 3, 7, -1, False), (2, [2, 3], [8, 5], [0, 1], False), (3, 3, [6, 7, 8], -1, False),
                                                 (3, 2, [9, 6, 7], -1, [True, False, True]), (2, 4, 5, 2, [False, True])])
@pytest.mark.parametrize("size", [1, 2, 4, 8])
def test_ranges_match_oracle(dim, p, N, C_, periodic, size):
    for rank in range(size):
        ls = lambda v: v if isinstance(v, list) else [v] * dim
        orc, eng = O.OracleIGA(dim, 1), P.IGX(dim, 1)
        for i in range(dim):
            orc.axis_uniform(i, ls(p)[i], ls(N)[i], ls(C_)[i], periodic=ls(periodic)[i])
            eng.axis_uniform(i, ls(p)[i], ls(N)[i], ls(C_)[i], periodic=ls(periodic)[i])
        try:
            orc.set_partition(size, rank)
        except RuntimeError:
            eng.set_comm(size, rank)
            with pytest.raises(P.IGXError):
                eng.setup()
            continue
        orc.setup()
        eng.set_comm(size, rank)
        eng.setup()
        ro, re_ = orc.ranges(), eng.sizes()
        for k in ("proc_sizes", "proc_ranks", "elem_sizes", "elem_start", "elem_width", "node_sizes", "node_lstart", "node_lwidth", "node_gstart", "node_gwidth"):
            assert ro[k] == re_[k][:dim], (k, rank, ro[k], re_[k])


def test_partition_golden_grids_on_product():
    for N in (16, 128, 192, 256):
        for size, grid in ((1, [1, 1, 1]), (2, [1, 1, 2]), (4, [1, 2, 2]), (8, [2, 2, 2])):
            g = P.IGX(3, 1)
            for i in range(3):
                g.axis_uniform(i, 2, N)
            g.set_comm(size, 0)
            g.setup()
            assert g.sizes()["proc_sizes"] == grid


def test_coloring_is_conflict_free_and_bit_exact():
    """Colour = e mod (p+1) per axis (plus private colours for the wrap on a periodic axis); two elements of
    the same colour never share a (wrapped) basis function."""
    for p, N, periodic in ((3, 13, False), (2, 9, False), (2, 10, True), (3, 8, True), (3, 9, True), (1, 5, True)):
        orc, eng = make_pair(1, 1, p, N, periodic=periodic, engine=True)
        nc = eng.coloring()[0]
        col = [eng.element_color(0, e) for e in range(N)]
        if not periodic:
            assert col == [e % (p + 1) for e in range(N)] and nc == min(N, p + 1)
        off = orc.basis(0)["offset"]
        nnp = orc.axis(0)["nnp"]
        nodes = [set((off[e] + a) % nnp for a in range(p + 1)) for e in range(N)]
        for e in range(N):
            for f in range(e + 1, N):
                if col[e] == col[f]:
                    assert not (nodes[e] & nodes[f]), (p, N, periodic, e, f)
        assert max(col) + 1 == nc


def test_error_codes_without_gpu():
    g = P.IGX()
    with pytest.raises(P.IGXError) as e:
        g.set_dim(4)
    assert e.value.code == 63
    g.set_dim(2)
    g.set_dof(1)
    with pytest.raises(P.IGXError) as e:          # IGAAxisInitUniform before IGAAxisSetDegree: PETSC_ERR_ORDER
        P._ck(P.lib().IGXAxisInitUniform(g.h, 0, 4, 0.0, 1.0, -1))
    assert e.value.code == 58
    with pytest.raises(P.IGXError) as e:          # IGASetUp without axes
        g.setup()
    assert e.value.code == 58
    g.axis_uniform(0, 2, 4)
    with pytest.raises(P.IGXError) as e:
        P._ck(P.lib().IGXAxisInitUniform(g.h, 0, 4, 1.0, 0.0, -1))    # Ui >= Uf: PETSC_ERR_ARG_WRONG
    assert e.value.code == 62
    with pytest.raises(P.IGXError) as e:
        P._ck(P.lib().IGXSetBoundaryValue(g.h, 5, 0, 0, 1.0))
    assert e.value.code == 63
