"""GPU parity of state_pencil (petiga_amd/csrc/gram_mfma.hpp): the Tangent of demo/CahnHilliard3D.c:111-179 through
IGAComputeIJacobian (src/petigats.c:112-170) on the axis-0 pencil walk -- the state's value, gradient and Laplacian at the Gauss
points by sum factorisation across the wavefront, all (p+1)^2 tile pairs (the Tangent is not symmetric), band rows written once
per pencil.  Engine vs oracle on identical inputs: pattern bit-exact, values to 1e-11 of max|K| over the rows without a Dirichlet
condition; a NaN-poisoned matrix comes back with the same bits (first-touch stores reach every entry)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair

pytestmark = pytest.mark.gpu

CH = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)


def _poison(mat):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _, _, val = mat.device_ptrs()
    assert hip.hipMemset(val, 0xFF, mat.nblocks * mat.bs * mat.bs * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("p,N,periodic,bc,nseg", [
    (2, (9, 4, 5), (False, False, False), False, 0),        # demo/CahnHilliard3D.c at p = 2 (config 4's discretisation)
    (2, (12, 3, 4), (False, False, False), True, 0),        # Dirichlet values on four faces: FixValues + FixJacobian inside the walk
    (2, (17, 4, 3), (False, False, False), True, 3),        # three segments along the walk
    (2, (8, 4, 3), (False, False, False), True, 4),         # segments of two elements (the floor since round 6)
    (3, (9, 3, 4), (False, False, False), True, 4),         # p = 3: segments of three elements, the halo as long
    (2, (8, 5, 6), (False, True, True), False, 0),          # the demo's periodic box on the two axes the walk does not follow
    (3, (9, 4, 4), (False, False, False), False, 0),
    (3, (10, 3, 5), (False, False, True), True, 2),
    (2, (9, 5, 6), (True, True, True), False, 0),           # the demo with -iga_periodic on one rank: the walk axis wrapped inside the rank
    (2, (19, 5, 5), (True, False, True), True, 3),          # ... in three segments (each re-computes the p elements before its start)
    (3, (8, 7, 4), (True, True, False), False, 2),
])
def test_cahn_hilliard_tangent_vs_oracle(p, N, periodic, bc, nseg, monkeypatch):
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, p, list(N), periodic=list(periodic))
    if bc:
        for g in (orc, eng):
            if not periodic[0]:
                g.set_boundary_value(0, 0, 0, 0.6)
                g.set_boundary_value(0, 1, 0, 0.66)
            if not periodic[1]:
                g.set_boundary_value(1, 1, 0, 0.61)
            if not periodic[2]:
                g.set_boundary_value(2, 0, 0, 0.65)
    ctx = O.CahnHilliardCtx(*CH)
    rng = np.random.default_rng(3)
    n = orc.global_size()
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    eng.set_form("cahnhilliard", CH)
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    _poison(J)
    eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "state_pencil<CahnHilliard>" in eng.kernel_name(), eng.kernel_name()
    J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, 250.0, V, 0.0, U)
    compare_mats(J, J_o, 1e-11)
    rows, cols, vals = J.to_coo_global()
    _poison(J)
    eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J)
    assert np.array_equal(J.to_coo_global()[2], vals)


@pytest.mark.parametrize("p,N,periodic,driver", [
    (2, (9, 5, 4), (False, False, False), "jacobian"),      # demo/Bratu.c: Dirichlet u = 0 on every face, Jacobian (SNES) driver
    (3, (8, 4, 5), (False, False, False), "ijacobian"),
    (2, (11, 4, 6), (False, False, True), "ijacobian"),
])
def test_bratu_jacobian_vs_oracle(p, N, periodic, driver):
    orc, eng = make_pair(3, 1, p, list(N), periodic=list(periodic))
    for g in (orc, eng):
        for d in range(3):
            if not periodic[d]:
                for side in range(2):
                    g.set_boundary_value(d, side, 0, 0.1 * d * side)
    lam = C.c_double(3.5)
    rng = np.random.default_rng(8)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n)
    eng.set_form("bratu", (3.5,))
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    _poison(J)
    if driver == "jacobian":
        eng.compute_jacobian(Uv, J)
        J_o = orc.compute_jacobian("orc_form_bratu_jacobian", lam, U)
    else:
        eng.compute_ijacobian(4.0, Vv, 0.0, Uv, J)
        J_o = orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, 4.0, V, 0.0, U)
    eng.synchronize()
    assert "state_pencil<Bratu>" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-12)


@pytest.mark.parametrize("form,N,periodic,rational,bc,nseg,amp", [
    ("ch", (9, 4, 5), (False, False, False), False, False, 0, 0.05),     # a polynomial map
    ("ch", (9, 4, 5), (False, False, False), True, False, 0, 0.05),      # NURBS: the weights enter the second derivatives (Rationalize at order 2)
    ("ch", (12, 3, 4), (False, False, False), True, True, 0, 0.08),      # Dirichlet values on four faces
    ("ch", (17, 4, 3), (False, False, False), True, True, 3, 0.05),      # three segments along the walk
    ("ch", (8, 5, 6), (False, False, False), False, False, 0, 0.0),      # the identity map given as a geometry (Greville net)
    ("ch", (8, 5, 6), (False, True, True), True, False, 0, 0.05),        # periodic across the walk (the net of a periodic axis is not wrapped: src/petiga.c IGASetGeometry)
    ("ch", (19, 5, 5), (True, False, True), True, False, 3, 0.05),       # the walk axis periodic and wrapped inside the rank, three segments
    ("bratu", (9, 5, 4), (False, False, False), True, True, 0, 0.06),    # a first-order form: no second derivatives summed
    ("bratu", (11, 4, 6), (False, False, False), False, True, 2, 0.06),
])
def test_tangent_on_a_mapped_geometry_vs_oracle(form, N, periodic, rational, bc, nseg, amp, monkeypatch):
    """state_pencil_geo (p = 2): the Tangent on a mapped geometry -- the physical gradient and Laplacian of the basis functions and of
    the state need the second derivatives of the map (and of the NURBS weight function): IGAElement's shape functions at order 2,
    src/petigageo.f90.in + src/petigarat.f90.in.  Engine vs oracle on the same warped net, 1e-10 of max|K|."""
    from common import warped_geometry
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, 2, list(N), periodic=list(periodic))
    X, W = warped_geometry(orc, 3, seed=6, rational=rational, amp=amp)
    orc.set_geometry(X, W); eng.set_geometry(X, W)
    if bc:
        assert not any(periodic)
        for g in (orc, eng):
            g.set_boundary_value(0, 0, 0, 0.6); g.set_boundary_value(0, 1, 0, 0.66)
            g.set_boundary_value(1, 1, 0, 0.61); g.set_boundary_value(2, 0, 0, 0.65)
    rng = np.random.default_rng(13)
    n = orc.global_size()
    V = rng.standard_normal(n)
    if form == "ch":
        U = 0.63 + 0.05 * (2 * rng.random(n) - 1)
        eng.set_form("cahnhilliard", CH)
        J_o = orc.compute_ijacobian("orc_form_ch_tangent", O.CahnHilliardCtx(*CH), 250.0, V, 0.0, U)
    else:
        U = 0.3 * rng.standard_normal(n)
        eng.set_form("bratu", (3.5,))
        J_o = orc.compute_ijacobian("orc_form_bratu_ijacobian", C.c_double(3.5), 250.0, V, 0.0, U)
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    _poison(J)
    eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "state_pencil<" in eng.kernel_name() and "mapped geometry" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-10)
    vals = J.to_coo_global()[2]
    _poison(J)
    eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J)
    assert np.array_equal(J.to_coo_global()[2], vals)


@pytest.mark.parametrize("geo", [None, "nurbs"])
def test_layer_pair_tiles_stay_alive(geo, monkeypatch):
    """IGX_P2_PACK=0: the p = 2 Tangents with one tile per pair of node layers (state_pencil<2> / state_pencil_geo<2>: what ran before
    the packed tiles of round 5 and still is the code of p = 3 and of run-time structs under that switch) against the oracle, and the
    same matrix from the packed kernel to rounding."""
    from common import warped_geometry
    ctx = O.CahnHilliardCtx(*CH)
    vals = {}
    for pack in ("0", "1"):
        monkeypatch.setenv("IGX_P2_PACK", pack)
        orc, eng = make_pair(3, 1, 2, [12, 4, 5])
        if geo:
            X, W = warped_geometry(orc, 3, seed=8, rational=True, amp=0.06)
            orc.set_geometry(X, W); eng.set_geometry(X, W)
        for g in (orc, eng):
            g.set_boundary_value(0, 0, 0, 0.6); g.set_boundary_value(2, 1, 0, 0.64)
        rng = np.random.default_rng(21)
        n = orc.global_size()
        U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
        eng.set_form("cahnhilliard", CH)
        Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
        _poison(J)
        eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J)
        eng.synchronize()
        assert "state_pencil<CahnHilliard>" in eng.kernel_name() and (("packed tiles" in eng.kernel_name()) == (pack == "1")), eng.kernel_name()
        J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, 250.0, V, 0.0, U)
        compare_mats(J, J_o, 1e-10 if geo else 1e-11)
        vals[pack] = J.to_coo_global()[2]
    assert np.abs(vals["0"] - vals["1"]).max() <= 1e-10 * np.abs(vals["0"]).max()


def test_switch_and_fallbacks(monkeypatch):
    """IGX_STATE_PENCIL=0, a short walk axis and a mapped geometry at p = 3 keep the feature kernel"""
    from common import warped_geometry
    ctx = O.CahnHilliardCtx(*CH)
    for tag, N, periodic, env, geo, p in (("off", (9, 4, 4), (False,) * 3, "0", False, 2), ("short", (5, 4, 4), (False,) * 3, None, False, 2),
                                          ("short wrapped", (7, 5, 5), (True, False, False), None, False, 2), ("mapped", (9, 4, 4), (False,) * 3, None, True, 3),
                                          ("mapped, off", (9, 4, 4), (False,) * 3, "0", True, 2)):
        if env is None:
            monkeypatch.delenv("IGX_STATE_PENCIL", raising=False)
        else:
            monkeypatch.setenv("IGX_STATE_PENCIL", env)
        orc, eng = make_pair(3, 1, p, list(N), periodic=list(periodic))
        if geo:
            X, W = warped_geometry(orc, 3, seed=4, rational=False, amp=0.05)
            orc.set_geometry(X, W); eng.set_geometry(X, W)
        rng = np.random.default_rng(5)
        n = orc.global_size()
        U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
        eng.set_form("cahnhilliard", CH)
        Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
        eng.compute_ijacobian(10.0, Vv, 0.0, Uv, J)
        eng.synchronize()
        assert "state_pencil" not in eng.kernel_name(), (tag, eng.kernel_name())
        compare_mats(J, orc.compute_ijacobian("orc_form_ch_tangent", ctx, 10.0, V, 0.0, U), 1e-11)
        if tag == "mapped":      # asked for by name (IGXSetKernel(2)) the walk says why it cannot: PETSC_ERR_SUP, not another kernel's numbers
            import petiga_amd as P
            eng.set_kernel(2)
            with pytest.raises(P.IGXError) as e:
                eng.compute_ijacobian(10.0, Vv, 0.0, Uv, J)
            assert e.value.code == 56 and "p = 2 only" in str(e.value), str(e.value)


@pytest.mark.parametrize("form,N,periodic,bc,nseg,fixtable", [
    ("ch", (9, 4, 5), (False, False, False), False, 0, False),        # config 4's discretisation
    ("ch", (12, 3, 4), (False, False, False), True, 0, False),        # Dirichlet values on four faces: F_k = nelem (u - v), V dropped at fixed nodes
    ("ch", (17, 4, 3), (False, False, False), True, 3, True),         # three segments, the values from a fix table
    ("ch", (8, 5, 6), (False, True, True), False, 0, False),
    ("ch", (19, 5, 5), (True, False, True), True, 3, False),          # the walk axis wrapped inside the rank
    ("bratu", (9, 5, 4), (False, False, False), True, 0, False),
    ("bratu", (11, 4, 6), (False, False, True), True, 2, False),
    ("bratu-snes", (9, 5, 4), (False, False, False), True, 0, False),  # Function + Jacobian (no V)
])
def test_fused_function_and_jacobian_vs_oracle(form, N, periodic, bc, nseg, fixtable, monkeypatch):
    """IGXComputeIFunctionIJacobian (state_pencil_kr): the Residual rides on the Tangent's MFMAs as operand column 27 of tile 1 and is
    summed over the pencil in a ring of its own.  F against the oracle's IFunction (demo/CahnHilliard3D.c:55-109 through
    src/petigats.c:23-90: FixValues / DelValues / FixFunction), J against its IJacobian, both from ONE call; then the same call with
    IGX_FUSE_RESID=0 semantics (the two drivers) through the separate entry points gives the same numbers."""
    monkeypatch.setenv("IGX_FUSE_RESID", "1")        # (read at IGXCreate; the default runs the two drivers: the fused walk measured 3 % slower)
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, 2, list(N), periodic=list(periodic))
    if bc:
        for g in (orc, eng):
            if not periodic[0]:
                g.set_boundary_value(0, 0, 0, 0.6)
                g.set_boundary_value(0, 1, 0, 0.66)
            if not periodic[1]:
                g.set_boundary_value(1, 1, 0, 0.61)
            if not periodic[2]:
                g.set_boundary_value(2, 0, 0, 0.65)
    rng = np.random.default_rng(21)
    n = orc.global_size()
    V = rng.standard_normal(n)
    if fixtable:
        tab = 0.63 + 0.04 * (2 * rng.random(n) - 1)
        orc.set_fixtable(tab)
        eng.set_fixtable(eng.create_vec().set(tab))
    shift = 250.0
    if form == "ch":
        U = 0.63 + 0.05 * (2 * rng.random(n) - 1)
        eng.set_form("cahnhilliard", CH)
        ctx = O.CahnHilliardCtx(*CH)
        F_o = orc.compute_ifunction("orc_form_ch_residual", ctx, shift, V, 0.0, U)
        J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, shift, V, 0.0, U)
    else:
        U = 0.3 * rng.standard_normal(n)
        eng.set_form("bratu", (3.5,))
        lam = C.c_double(3.5)
        if form == "bratu":
            F_o = orc.compute_ifunction("orc_form_bratu_ifunction", lam, shift, V, 0.0, U)
            J_o = orc.compute_ijacobian("orc_form_bratu_ijacobian", lam, shift, V, 0.0, U)
        else:
            F_o = orc.compute_function("orc_form_bratu_function", lam, U)
            J_o = orc.compute_jacobian("orc_form_bratu_jacobian", lam, U)
    Uv, Vv, J, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat(), eng.create_vec()
    F.set(np.full(n, 7.0))          # the driver zeroes F itself
    _poison(J)
    if form == "bratu-snes":
        eng.compute_function_jacobian(Uv, F, J)
    else:
        eng.compute_ifunction_ijacobian(shift, Vv, 0.0, Uv, F, J)
    eng.synchronize()
    assert "+Residual>" in eng.kernel_name() and "packed tiles" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-11)
    scale = np.abs(F_o).max()
    assert np.abs(F.get() - F_o).max() <= 1e-11 * scale, np.abs(F.get() - F_o).max() / scale
    # bit-repeatable, and the two-call path agrees
    Fb, vals = F.get().copy(), J.to_coo_global()[2].copy()
    _poison(J)
    if form == "bratu-snes":
        eng.compute_function_jacobian(Uv, F, J)
    else:
        eng.compute_ifunction_ijacobian(shift, Vv, 0.0, Uv, F, J)
    assert np.array_equal(F.get(), Fb) and np.array_equal(J.to_coo_global()[2], vals)
    F2, J2 = eng.create_vec(), eng.create_mat()
    if form == "bratu-snes":
        eng.compute_function(Uv, F2); eng.compute_jacobian(Uv, J2)
    else:
        eng.compute_ifunction(shift, Vv, 0.0, Uv, F2); eng.compute_ijacobian(shift, Vv, 0.0, Uv, J2)
    eng.synchronize()
    assert np.abs(F2.get() - Fb).max() <= 1e-12 * scale
    assert np.abs(J2.to_coo_global()[2] - vals).max() <= 1e-12 * np.abs(vals).max()


def test_fused_call_falls_back_to_the_two_drivers(monkeypatch):
    """Where no fused kernel covers the case (p = 3; a mapped geometry) or it is not asked for (the default) the call runs IFunction and
    IJacobian one after the other: same results, both kernels named."""
    for p, geo, fuse in ((3, False, "1"), (2, True, "1"), (2, False, None)):
        if fuse:
            monkeypatch.setenv("IGX_FUSE_RESID", fuse)
        else:
            monkeypatch.delenv("IGX_FUSE_RESID", raising=False)
        orc, eng = make_pair(3, 1, p, [9, 4, 4])
        if geo:
            from common import warped_geometry
            X, W = warped_geometry(orc, 3, seed=6, rational=True, amp=0.05)
            orc.set_geometry(X, W); eng.set_geometry(X, W)
        rng = np.random.default_rng(2)
        n = orc.global_size()
        U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
        eng.set_form("cahnhilliard", CH)
        ctx = O.CahnHilliardCtx(*CH)
        F_o = orc.compute_ifunction("orc_form_ch_residual", ctx, 250.0, V, 0.0, U)
        J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, 250.0, V, 0.0, U)
        Uv, Vv, J, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat(), eng.create_vec()
        eng.compute_ifunction_ijacobian(250.0, Vv, 0.0, Uv, F, J)
        eng.synchronize()
        assert " | " in eng.kernel_name() and "vec_sumfact" in eng.kernel_name() and "state_pencil" in eng.kernel_name(), eng.kernel_name()
        compare_mats(J, J_o, 1e-10)
        assert np.abs(F.get() - F_o).max() <= 1e-10 * np.abs(F_o).max()
