"""GPU parity of the feature kernel's pencil mode (combine before write: a workgroup walks a pencil of elements along mesh
axis 0 and keeps the accumulator tiles across elements; feature_mfma.hpp).  The automatic choice takes it only when a
colour holds enough pencils to fill the chip, so the small meshes the oracle can check force it with IGX_COMBINE=1
(read when the IGX is created).  Same tolerances as tests/test_gpu_parity.py."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, warped_geometry

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def force_pencil(monkeypatch):
    monkeypatch.setenv("IGX_COMBINE", "1")
    monkeypatch.setenv("IGX_KERNEL", "3")


def _poison(mat):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _, _, val = mat.device_ptrs()
    assert hip.hipMemset(val, 0xFF, mat.nblocks * mat.bs * mat.bs * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


def _close(a, b, tol):
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)


def _bc(objs, kind, dof):
    for g in objs:
        if kind == "all":
            for d in range(3):
                for s in range(2):
                    for f in range(dof):
                        g.set_boundary_value(d, s, f, 0.25 + 0.5 * d + 0.125 * s + f)
        elif kind == "axis0":
            for f in range(dof):
                g.set_boundary_value(0, 0, f, 0.0)
            g.set_boundary_value(0, 1, 0, 1.0)
        elif kind == "partial":
            g.set_boundary_value(1, 0, 0, 2.0)
            g.set_boundary_value(2, 1, dof - 1, -1.0)
            g.set_boundary_load(0, 1, 0, 0.75)


@pytest.mark.parametrize("form,dof,N,bc,geo", [("poisson", 1, (9, 5, 6), "all", False), ("poisson", 1, (4, 4, 4), "partial", True), ("poisson", 1, (13, 3, 2), "none", True),
                                                ("elasticity", 3, (6, 5, 4), "axis0", False), ("elasticity", 3, (7, 4, 5), "all", True), ("elasticity", 3, (5, 4, 9), "partial", False),
                                                ("mass", 2, (8, 4, 5), "partial", True), ("errnorm", 4, (5, 6, 4), "all", False)])
def test_pencil_system_forms(form, dof, N, bc, geo):
    orc, eng = make_pair(3, dof, 3, list(N))
    if geo:
        X, W = warped_geometry(orc, 3, seed=sum(N), rational=True, amp=0.1)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    _bc((orc, eng), bc, dof)
    octx, prm = None, ()
    if form == "elasticity":
        octx, prm = O.ElasticityCtx(2.5, 0.7), (2.5, 0.7)
    A_o, b_o = orc.compute_system("orc_form_" + form, octx)
    eng.set_form(form, prm)
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)                      # first-touch stores must reach every entry
    eng.compute_system(A, b)
    eng.synchronize()
    assert "pencil walk" in eng.kernel_name(), eng.kernel_name()
    tol = 1e-11 if geo else 1e-12
    compare_mats(A, A_o, tol)
    _close(b.get(), b_o, tol)
    # Matrix driver (no fix-up) on the same walk
    orc.clear_boundary()
    A_o2, _ = orc.compute_system("orc_form_" + form, octx)
    _poison(A)
    eng.compute_matrix(A)
    eng.synchronize()
    compare_mats(A, A_o2, tol)


@pytest.mark.parametrize("geo", [False, True])
@pytest.mark.parametrize("periodic", [(True, False, True), (False, False, True), (False, True, False)])
def test_pencil_navier_stokes_vms_p3(geo, periodic):
    orc, eng = make_pair(3, 4, 3, [8, 7 if periodic[1] else 4, 7], periodic=list(periodic))
    if geo:
        X, W = warped_geometry(orc, 3, seed=3, rational=True, amp=0.08)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):
        for d in range(3):
            if not periodic[d]:
                for side in range(2):
                    for f in range(3):
                        g.set_boundary_value(d, side, f, 0.0)
    nu, fx, dt = 1.472e-4, 3.37204e-3, 1e-2
    ctx, params = O.NSVMSCtx(nu, fx, 0.0, 0.0, dt), (nu, fx, 0.0, 0.0, dt)
    rng = np.random.default_rng(29)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n) * 0.1
    shift = 2.0 / dt
    J_o = orc.compute_ijacobian("orc_form_ns_tangent", ctx, shift, V, 0.0, U)
    eng.set_form("nsvms", params)
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    _poison(J)
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    if not geo:      # (next to a mapped geometry's point arrays the staging buffers do not fit the LDS: the element mode takes over)
        assert "pencil walk" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-11)


def test_pencil_bratu_jacobian_and_repeatability():
    orc, eng = make_pair(3, 1, 3, [10, 4, 5])
    for g in (orc, eng):
        for d in range(3):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 0.0)
    lam = C.c_double(6.8)
    rng = np.random.default_rng(1)
    U = rng.standard_normal(orc.global_size()) * 0.4
    J_o = orc.compute_jacobian("orc_form_bratu_jacobian", lam, U)
    eng.set_form("bratu", (6.8,))
    Uv = eng.create_vec().set(U)
    outs = []
    for _ in range(2):
        J = eng.create_mat()
        _poison(J)
        eng.compute_jacobian(Uv, J)
        eng.synchronize()
        assert "pencil walk" in eng.kernel_name()
        compare_mats(J, J_o, 1e-12)
        outs.append(J.host(True))
    assert np.array_equal(outs[0], outs[1])           # deterministic: bitwise repeatable


def test_pencil_matches_element_mode_on_a_larger_mesh(monkeypatch):
    """Beyond the oracle's reach: pencil mode against one-element-per-workgroup mode of the same kernel (32 x 12 x 12,
    elasticity, clamped face): same matrix to rounding."""
    import petiga_amd as P

    def build(combine):
        monkeypatch.setenv("IGX_COMBINE", combine)
        g = P.IGX(3, 3)
        for i, n in enumerate((32, 12, 12)):
            g.axis_uniform(i, 3, n)
        g.setup()
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
        g.set_form("elasticity", (1.0, 1.0))
        A, b = g.create_mat(), g.create_vec()
        g.compute_system(A, b)
        g.synchronize()
        return g.kernel_name(), A.host(True), b.get()
    k1, a1, b1 = build("1")
    k0, a0, b0 = build("0")
    assert "pencil walk" in k1 and "pencil walk" not in k0
    assert np.abs(a1 - a0).max() <= 1e-12 * np.abs(a0).max()
    assert np.abs(b1 - b0).max() <= 1e-12 * max(np.abs(b0).max(), 1.0)
