"""Quadrature rules other than Gauss-Legendre on the device path: IGASetRuleType(IGA_RULE_LOBATTO) (src/petiga.c:500,
src/petigarule.c:321-459) and a user-defined rule (IGARuleSetRule, src/petigarule.c:145) against the oracle.  The kernels read
the rule from the 1-D tables (points, weights, basis rows), so every kernel family must take it: the pencil walks included
(Lobatto with p + 1 points has the point count they need)."""
import numpy as np
import pytest

from common import compare_mats, make_pair, warped_geometry

pytestmark = pytest.mark.gpu

UX = np.array([-0.93, -0.41, 0.08, 0.66])
UW = np.array([0.31, 0.62, 0.71, 0.36])


def _pair(dim, dof, p, N, rule, nq=None):
    orc, eng = make_pair(dim, dof, p, N)
    for g in (orc, eng):
        for i in range(dim):
            if rule in ("lobatto", "reduced"):
                g.set_rule_type(i, rule)
                if nq:
                    g.set_quadrature(i, nq)
            else:
                g.set_rule(i, UX[:nq or 4], UW[:nq or 4])
        g.setup()
    return orc, eng


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("dim,p,N,rule,nq,geo", [(2, 2, [7, 6], "lobatto", None, "none"), (2, 2, [6, 5], "lobatto", 5, "nurbs"), (3, 3, [9, 4, 4], "lobatto", None, "none"),
                                                 (3, 2, [9, 5, 4], "lobatto", None, "nurbs"), (3, 3, [8, 4, 5], "user", 4, "poly"), (3, 2, [10, 5, 4], "user", 3, "none"),
                                                 (1, 3, [9], "lobatto", 6, "none"),
                                                 # IGA_RULE_REDUCED (src/petigabasis.c:144-171): one point less on the interior elements of every axis
                                                 (3, 3, [9, 4, 5], "reduced", None, "none"), (3, 2, [9, 5, 4], "reduced", None, "nurbs"), (2, 3, [6, 7], "reduced", 5, "poly"),
                                                 (3, 2, [8, 2, 5], "reduced", None, "none"), (1, 2, [7], "reduced", 2, "none")])
def test_poisson_system_with_other_rules(dim, p, N, rule, nq, geo, kernel):
    orc, eng = _pair(dim, 1, p, N, rule, nq)
    eng.set_kernel(kernel)
    if geo != "none":
        X, W = warped_geometry(orc, dim, seed=5, rational=(geo == "nurbs"), amp=0.08)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    for g in (orc, eng):
        for d in range(dim):
            g.set_boundary_value(d, 0, 0, 0.25 * (d + 1))
        g.set_boundary_load(0, 1, 0, 1.5)
    A_o, b_o = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    if kernel == 0 and dim == 3 and (nq in (None, p + 1)):
        assert "pencil" in eng.kernel_name(), eng.kernel_name()      # the walk takes the rule from the tables like any other kernel
    tol = 1e-12 if geo == "none" else 2e-11
    compare_mats(A, A_o, tol)
    assert np.abs(b.get() - b_o).max() <= tol * max(np.abs(b_o).max(), 1.0)


@pytest.mark.parametrize("rule", ["lobatto", "reduced"])
@pytest.mark.parametrize("form,dof", [("elasticity", 3), ("cahnhilliard", 1)])
def test_multi_field_and_nonlinear_forms_with_lobatto(form, dof, rule):
    p = 3 if form == "elasticity" else 2
    orc, eng = _pair(3, dof, p, [9, 4, 4], rule)
    if form == "elasticity":
        import ctypes as C
        import oracle_api as O
        for g in (orc, eng):
            for c in range(3):
                g.set_boundary_value(0, 0, c, 0.0)
            g.set_boundary_value(0, 1, 0, 1.0)
        A_o, b_o = orc.compute_system("orc_form_elasticity", ctx=O.ElasticityCtx(1.0, 1.0))
        eng.set_form("elasticity", [1.0, 1.0])
        A, b = eng.create_mat(), eng.create_vec()
        eng.compute_system(A, b)
        eng.synchronize()
        compare_mats(A, A_o, 1e-12)
        assert np.abs(b.get() - b_o).max() <= 1e-12 * max(np.abs(b_o).max(), 1.0)
        return
    # Cahn-Hilliard (demo/CahnHilliard3D.c:55-179): IFunction + IJacobian, the state strictly inside (0, 1)
    import oracle_api as O
    CH = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)
    ctx = O.CahnHilliardCtx(*CH)
    rng = np.random.default_rng(11)
    n = orc.global_size()
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    eng.set_form("cahnhilliard", CH)
    Uv, Vv, J, F = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat(), eng.create_vec()
    eng.compute_ifunction(250.0, Vv, 0.0, Uv, F)
    eng.compute_ijacobian(250.0, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "state_pencil" in eng.kernel_name(), eng.kernel_name()
    F_o = orc.compute_ifunction("orc_form_ch_residual", ctx, 250.0, V, 0.0, U)
    J_o = orc.compute_ijacobian("orc_form_ch_tangent", ctx, 250.0, V, 0.0, U)
    compare_mats(J, J_o, 1e-11)
    assert np.abs(F.get() - F_o).max() <= 1e-11 * np.abs(F_o).max()
