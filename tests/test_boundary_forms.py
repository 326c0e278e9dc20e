"""Boundary-form passes: IGASetBoundaryForm faces, normals, detS.  Known answers are the reference's own
run rules: demo/NitscheMethod.c with -check_error 1e-6 (demo/makefile:218-219) and demo/BoundaryIntegral.c's -check_error
(error <= 1e-3, exact solution x_axis + 1 resp. 2 - x_axis)."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse.linalg as sla

import oracle_api as O
from common import compare_mats, make_pair, rel_err, warped_geometry

TOL = 1e-12


def _nitsche_pair(dim, p, N, geo=None, engine=True):
    orc, eng = make_pair(dim, 1, p, N, engine=engine)
    if geo:
        X, W = warped_geometry(orc, dim, seed=4, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W)
        if eng: eng.set_geometry(X, W)
    for a in range(dim):
        for s in range(2):
            orc.set_boundary_form(a, s, True)
            if eng: eng.set_boundary_form(a, s, True)
    return orc, eng


@pytest.mark.parametrize("dim", [1, 2])
def test_oracle_nitsche_known_answer(dim):
    # demo/makefile:218-219: -iga_dim {1,2} -iga_degree 2 -check_error 1e-6 (default 16 elements per axis)
    orc, _ = _nitsche_pair(dim, 2, 16, engine=False)
    A, b = orc.compute_system("orc_form_nitsche", C.c_int(2))
    x = sla.spsolve(A.scipy().tocsc(), b)
    err = np.sqrt(orc.compute_scalar("orc_scalar_x2err", 1, U=x)[0])
    assert err < 1e-6


@pytest.mark.parametrize("dim,axis,side", [(2, 0, 1), (2, 1, 0), (3, 2, 1)])
def test_oracle_boundary_integral_known_answer(dim, axis, side):
    orc, _ = make_pair(dim, 1, 2, 8 if dim == 2 else 4, engine=False)
    orc.set_boundary_value(axis, 1 - side, 0, 1.0)      # demo/BoundaryIntegral.c:172-176
    orc.set_boundary_form(axis, side, True)
    A, b = orc.compute_system("orc_form_boundary_integral")
    x = sla.spsolve(A.scipy().tocsc(), b).reshape([orc.axis(i)["nnp"] for i in range(dim)][::-1])
    from common import greville
    g = greville(orc.axis(axis)["U"], 2)
    exact = (g + 1) if side == 1 else (1 - g + 1)       # Exact(), :124-134 -- linear, so control values = Greville values
    xs = np.moveaxis(x, dim - 1 - axis, -1)
    assert np.abs(xs - exact).max() < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("dim,p,N,geo", [(2, 2, 16, None), (2, 3, 5, "nurbs"), (3, 2, 4, None), (3, 2, 3, "poly"), (3, 3, 3, "nurbs"), (3, (3, 2, 1), (3, 4, 5), "nurbs")])
def test_nitsche_on_device(dim, p, N, geo):
    orc, eng = _nitsche_pair(dim, p, N, geo)
    k = max(p) if isinstance(p, tuple) else p
    Ao, bo = orc.compute_system("orc_form_nitsche", C.c_int(k))
    eng.set_form("nitsche", (float(k),))
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    assert "feature_assemble" in eng.kernel_name()
    tol = 1e-12 if geo is None else 2e-11
    compare_mats(A, Ao, tol)
    assert rel_err(b.get(), bo) < tol
    if geo is None and dim == 2 and p == 2:   # the reference's run rule on the device-assembled system
        x = sla.spsolve(A.to_scipy_global().tocsc(), b.get())
        assert np.sqrt(eng.compute_scalar("x2err", eng.create_vec().set(x))[0]) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("dim,axis,side,geo", [(2, 0, 1, None), (2, 1, 0, "nurbs"), (3, 2, 1, None), (3, 0, 0, "nurbs"), (3, 1, 1, "poly")])
def test_boundary_integral_on_device(dim, axis, side, geo):
    # Dirichlet face + visited face on the same axis: the boundary pass must not repeat the unit diagonal / the values
    orc, eng = make_pair(dim, 1, 2, [5, 4, 6][:dim])
    if geo:
        X, W = warped_geometry(orc, dim, seed=8, rational=(geo == "nurbs"), amp=0.1)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    for g in (orc, eng):
        g.set_boundary_value(axis, 1 - side, 0, 1.0)
        g.set_boundary_value((axis + 1) % dim, 0, 0, -0.5)        # a Dirichlet face that cuts the visited face
        g.set_boundary_form(axis, side, True)
    Ao, bo = orc.compute_system("orc_form_boundary_integral")
    eng.set_form("boundary_integral")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    tol = 1e-12 if geo is None else 2e-11
    compare_mats(A, Ao, tol)
    assert np.abs(b.get() - bo).max() <= tol * max(np.abs(bo).max(), 1.0)
    # a form without a boundary branch is simply integrated over the face as well (the callback ignores atboundary)
    eng.set_form("mass"); A2, b2 = eng.create_mat(), eng.create_vec(); eng.compute_system(A2, b2)
    A2o, b2o = orc.compute_system("orc_form_mass")
    compare_mats(A2, A2o, tol)
    assert np.abs(b2.get() - b2o).max() <= tol * max(np.abs(b2o).max(), 1.0)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,p,N,geo", [(1, 2, 16, None), (1, 3, 7, None), (2, 2, 6, "nurbs"), (3, 2, 3, "poly"), (3, 4, 2, None), (3, 4, 2, "nurbs")])
def test_boundary_passes_on_the_generic_kernel(dim, p, N, geo):
    """dim 1 and nen > 64 (p = 4 in 3-D: 125 functions) have no MFMA kernel: the point-form kernel makes the boundary passes
    itself (IGAElementNextForm, src/petigaelem.c:427-447); the other cases run it by choice (IGXSetKernel 1)."""
    orc, eng = _nitsche_pair(dim, p, N, geo)
    Ao, bo = orc.compute_system("orc_form_nitsche", C.c_int(p))
    eng.set_form("nitsche", (float(p),))
    eng.set_kernel(1)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    assert "generic_assemble" in eng.kernel_name()
    tol = 1e-12 if geo is None else 2e-11
    compare_mats(A, Ao, tol)
    assert rel_err(b.get(), bo) < tol
    if dim == 1 and p == 2:      # demo/makefile:218: -iga_dim 1 -iga_degree 2 -check_error 1e-6
        x = sla.spsolve(A.to_scipy_global().tocsc(), b.get())
        assert np.sqrt(eng.compute_scalar("x2err", eng.create_vec().set(x))[0]) < 1e-6
    # a Dirichlet face next to a visited one, and a form without a boundary branch integrated over the face
    orc2, eng2 = make_pair(dim, 1, p, N)
    for g in (orc2, eng2):
        g.set_boundary_value(0, 0, 0, 1.0)
        g.set_boundary_form(0, 1, True)
        if dim > 1:
            g.set_boundary_value(1, 0, 0, -0.5)
    for oform, eform in (("orc_form_boundary_integral", "boundary_integral"), ("orc_form_mass", "mass")):
        Ao, bo = orc2.compute_system(oform)
        eng2.set_form(eform)
        eng2.set_kernel(1)
        A, b = eng2.create_mat(), eng2.create_vec()
        eng2.compute_system(A, b)
        compare_mats(A, Ao, 1e-12)
        assert np.abs(b.get() - bo).max() <= 1e-12 * max(np.abs(bo).max(), 1.0)


@pytest.mark.gpu
def test_boundary_functional_in_one_dimension():
    # IGAComputeScalarFull (test/IGAGeometryMap.c:391-450) in 1-D: length of the interval and one unit "area" per visited end
    _, eng = make_pair(1, 1, 2, 6)
    eng.set_boundary_form(0, 0, True)
    eng.set_boundary_form(0, 1, True)
    S = eng.compute_scalar("volume")
    assert abs(S[0] - 1.0) < 1e-14 and abs(S[1] - 2.0) < 1e-14
