"""The headline pencil kernel on MAPPED geometries (gram_mfma.hpp, GEO variant): the metric tensor JW F^-1 F^-T of every Gauss
point comes from the wavefront's own geometry evaluation (Rationalize + GeometryMap + InverseMap, src/petigarat.f90.in,
src/petigamapgeo.f90.in, src/petigainv.f90.in) and enters the B operand of the MFMA contraction; the A operand keeps the
tensor-product form.  Engine vs oracle, 1e-11 (mapped geometry), polynomial and rational (NURBS) maps, p = 2 and 3."""
import ctypes as C

import numpy as np
import pytest

from common import compare_mats, make_pair, rel_err, warped_geometry

pytestmark = pytest.mark.gpu
TOL = 1e-11


def _poison(mat):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _, _, val = mat.device_ptrs()
    assert hip.hipMemset(val, 0xFF, mat.nblocks * mat.bs * mat.bs * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


def _bc(objs, kind):
    for g in objs:
        k = 0
        for d in range(3):
            for s in range(2):
                if kind == "all" or (kind == "some" and (d + s) % 2 == 0):
                    g.set_boundary_value(d, s, 0, 0.5 + 0.25 * k)
                k += 1


@pytest.mark.parametrize("rational", [False, True])
@pytest.mark.parametrize("p,N,bc,form", [(3, (9, 4, 5), "all", "poisson"), (3, (8, 3, 3), "none", "poisson"), (3, (12, 5, 4), "some", "poisson_f"),
                                         (2, (10, 5, 6), "all", "poisson"), (2, (8, 4, 3), "some", "poisson"), (3, (70, 3, 3), "all", "poisson")])
def test_pencil_kernel_on_mapped_geometry(p, N, bc, form, rational):
    orc, eng = make_pair(3, 1, p, list(N))
    X, W = warped_geometry(orc, 3, seed=p * 7 + N[0], rational=rational, amp=0.12)
    orc.set_geometry(X, W)
    eng.set_geometry(X, W)
    _bc((orc, eng), bc)
    A_o, b_o = orc.compute_system("orc_form_" + form)
    eng.set_form(form)
    eng.set_kernel(2)                      # the pencil kernel or an error: no silent change of kernel
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)
    eng.compute_system(A, b)
    eng.synchronize()
    assert "gram_pencil" in eng.kernel_name() and "mapped geometry" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, TOL)
    assert np.abs(b.get() - b_o).max() <= TOL * max(np.abs(b_o).max(), 1e-300)
    # IGAComputeMatrix on the same kernel, and the automatic choice picks it as well
    orc.clear_boundary()
    A_o2, _ = orc.compute_system("orc_form_" + form)
    eng.set_kernel(0)
    _poison(A)
    eng.compute_matrix(A)
    eng.synchronize()
    assert "gram_pencil" in eng.kernel_name()
    compare_mats(A, A_o2, TOL)


def test_tangled_geometry_is_reported():
    import petiga_amd as P
    orc, eng = make_pair(3, 1, 3, [8, 3, 3])
    X, _ = warped_geometry(orc, 3, rational=False)
    X[:, 0] *= -1.0
    eng.set_geometry(X)
    eng.set_form("poisson")
    eng.set_kernel(2)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    with pytest.raises(P.IGXError) as e:      # non-positive Jacobian: PETSC_ERR_USER as src/petigaelem.c:989-993
        eng.synchronize()
    assert e.value.code == 83


def test_repeatable_and_matches_the_feature_kernel_on_a_larger_mesh():
    import petiga_amd as P
    outs = []
    for kernel in (2, 2, 3):
        g = P.IGX(3, 1)
        for i, n in enumerate((40, 12, 12)):
            g.axis_uniform(i, 3, n)
        g.setup()
        from common import greville
        gv = [greville(np.concatenate([[0.0] * 4, np.arange(1, n) / n, [1.0] * 4]), 3) for n in (40, 12, 12)]
        mesh = np.meshgrid(*gv[::-1], indexing="ij")[::-1]
        X = np.stack([m.copy() for m in mesh], axis=-1)
        X[..., 0] += 0.05 * np.sin(2 * np.pi * mesh[1]); X[..., 1] += 0.05 * np.sin(2 * np.pi * mesh[2])
        Wt = 1.0 + 0.1 * np.cos(2 * np.pi * mesh[0])
        g.set_geometry(X.reshape(-1, 3), Wt.reshape(-1))
        for d in range(3):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 1.0)
        g.set_form("poisson")
        g.set_kernel(kernel)
        A, b = g.create_mat(), g.create_vec()
        g.compute_system(A, b)
        g.synchronize()
        outs.append((A.host(True), b.get()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    scale = np.abs(outs[2][0]).max()
    assert np.abs(outs[0][0] - outs[2][0]).max() <= 1e-12 * scale
    assert np.abs(outs[0][1] - outs[2][1]).max() <= 1e-12 * np.abs(outs[2][1]).max()


@pytest.mark.parametrize("p,N,periodic,nseg,driver", [
    (3, (9, 4, 5), (True, False, False), 0, "system"),       # the walk axis periodic and wrapped inside the rank
    (3, (21, 8, 4), (True, True, False), 3, "system"),       # ... in three segments, a second wrapped axis
    (2, (8, 5, 7), (True, False, True), 0, "matrix"),
    (2, (17, 4, 4), (True, False, False), 2, "system"),
    (3, (8, 4, 4), (True, False, False), 4, "system"),       # segments of two elements, shorter than their halo of p (the floor since round 6)
    (3, (9, 4, 5), (False, False, False), 4, "system"),      # ... on an open axis: lengths 3, 3, 3
])
def test_walk_axis_wrapped_inside_the_rank(p, N, periodic, nseg, driver, monkeypatch):
    """gram_pencil on a periodic axis 0 held by one rank: elements and node layers modulo the axis, every segment re-computes the p
    elements before its start, each band row written once (first-touch stores on a NaN-poisoned matrix)"""
    import ctypes as C
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, p, list(N), periodic=list(periodic))
    for g in (orc, eng):
        for d in range(3):
            if not periodic[d]:
                g.set_boundary_value(d, 0, 0, 0.5 + d); g.set_boundary_value(d, 1, 0, -1.0)
    Ao, bo = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A, b = eng.create_mat(), eng.create_vec()
    hip = C.CDLL("libamdhip64.so"); hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    assert hip.hipMemset(A.device_ptrs()[2], 0xFF, A.nblocks * 8) == 0 and hip.hipDeviceSynchronize() == 0
    if driver == "system":
        eng.compute_system(A, b)
    else:
        orc.clear_boundary()
        Ao, bo = orc.compute_system("orc_form_poisson")
        eng.compute_matrix(A)
    eng.synchronize()
    assert "gram_pencil" in eng.kernel_name() and "walk=0" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, Ao, 1e-12)
    if driver == "system":
        assert rel_err(b.get(), bo) < 1e-12
