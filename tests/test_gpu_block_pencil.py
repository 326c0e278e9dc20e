"""GPU parity of the block pencil kernel (petiga_amd/csrc/block_pencil.hpp): constant-coefficient multi-field forms
(demo/Elasticity3D.c) assembled as band rows, one node layer of a pencil at a time, with the coefficient transform, the
IGAElementFixSystem fix-up (src/petigaelem.c:1360-1389) and a coalesced read-add-write behind an LDS stage.

IGXSetKernel(4) forces the kernel (an uncovered case is an error, never a silent fall-back).  Engine vs oracle on identical
inputs: pattern bit-exact, values to 1e-12 of max|K| over the rows without a Dirichlet condition (tests/common.py)."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair

pytestmark = pytest.mark.gpu


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
        if kind == "demo":            # demo/Elasticity3D.c:66-71
            for f in range(dof):
                g.set_boundary_value(0, 0, f, 0.0)
            g.set_boundary_value(0, 1, 0, 1.0)
        elif kind == "all":           # every face, every field, distinct non-zero values: lifting through all band tiles
            for d in range(3):
                for s in range(2):
                    for f in range(dof):
                        g.set_boundary_value(d, s, f, 0.25 + 0.5 * d + 0.125 * s + f)
        elif kind == "partial":       # single fields on single faces + a boundary load
            g.set_boundary_value(1, 0, 0, 2.0)
            g.set_boundary_value(2, 1, dof - 1, -1.0)
            g.set_boundary_load(0, 1, 0, 0.75)
            g.set_boundary_load(1, 1, 1, -0.5)
        elif kind == "override":      # the same node fixed by two faces: the later face wins (IGAElementBuildFix order)
            g.set_boundary_value(0, 0, 0, 1.0)
            g.set_boundary_value(1, 0, 0, 2.0)
            g.set_boundary_value(2, 0, 0, 3.0)
            g.set_boundary_value(2, 0, 1, -3.0)


@pytest.mark.parametrize("N,bc,nseg", [((8, 4, 4), "demo", 0), ((9, 5, 6), "all", 0), ((8, 4, 9), "partial", 0), ((11, 4, 5), "override", 0),
                                       ((13, 5, 4), "none", 0), ((16, 4, 4), "all", 3), ((10, 8, 4), "demo", 5)])
def test_elasticity_vs_oracle(N, bc, nseg, monkeypatch):
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))     # cut the pencils into segments of node layers (read at IGXCreate)
    orc, eng = make_pair(3, 3, 3, list(N))
    _bc((orc, eng), bc, 3)
    octx, prm = O.ElasticityCtx(2.5, 0.7), (2.5, 0.7)
    A_o, b_o = orc.compute_system("orc_form_elasticity", octx)
    eng.set_form("elasticity", prm)
    eng.set_kernel(4)
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)                      # first-touch stores must reach every entry
    eng.compute_system(A, b)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, 1e-12)
    _close(b.get(), b_o, 1e-12)
    # Matrix driver (no fix-up) on the same kernel
    orc.clear_boundary()
    A_o2, _ = orc.compute_system("orc_form_elasticity", octx)
    _poison(A)
    eng.compute_matrix(A)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name()
    compare_mats(A, A_o2, 1e-12)


@pytest.mark.parametrize("N,bc,nseg", [((8, 4, 4), "demo", 0), ((9, 5, 6), "all", 0), ((8, 4, 9), "partial", 0), ((13, 5, 4), "none", 0), ((16, 4, 4), "all", 3)])
def test_elasticity_with_a_body_force(N, bc, nseg, monkeypatch):
    """F != 0 on the band-row kernel (round 6: the review's third request): demo/Elasticity3D.c's K with F[a][i] = N_a f_i.  The form's
    vec() comes from a sum-factorised vector pass ahead of the band rows (vec_sumfact, fixed rows left at zero); the band-row
    kernel adds the Dirichlet lifting and value x multiplicity of IGAElementFixSystem (src/petigaelem.c:1377-1387) as for F = 0."""
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 3, 3, list(N))
    _bc((orc, eng), bc, 3)
    prm = (2.5, 0.7, 0.3, -1.25, 2.0)
    A_o, b_o = orc.compute_system("orc_form_elasticity_f", (C.c_double * 5)(*prm))
    eng.set_form("elasticity_f", prm)
    eng.set_kernel(4)
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)
    eng.compute_system(A, b)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, 1e-12)
    _close(b.get(), b_o, 1e-12)
    eng.set_kernel(0)               # the automatic choice takes the same path
    A2, b2 = eng.create_mat(), eng.create_vec()
    eng.compute_system(A2, b2)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name(), eng.kernel_name()
    assert np.array_equal(b2.get(), b.get())
    eng.compute_vector(b2)          # IGAComputeVector: no fix-up at all
    eng.synchronize()
    orc.clear_boundary()
    _, b_o2 = orc.compute_system("orc_form_elasticity_f", (C.c_double * 5)(*prm))
    _close(b2.get(), b_o2, 1e-12)


@pytest.mark.parametrize("N,bc,nseg", [((8, 4, 4), "demo", 0), ((9, 5, 6), "all", 0), ((11, 4, 5), "override", 3)])
def test_elasticity_with_a_fix_table(N, bc, nseg, monkeypatch):
    """IGASetFixTable (src/petigaform.c:273-298): the Dirichlet values of the fixed dofs come from a vector, per node and field.  The
    band-row kernel reads them in its fix-up (lifting of F through the fixed columns, value x multiplicity in the fixed rows) instead
    of leaving the case to the feature kernel."""
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 3, 3, list(N))
    _bc((orc, eng), bc, 3)
    table = np.random.default_rng(17).standard_normal(orc.global_size())
    orc.set_fixtable(table)
    eng.set_fixtable(eng.create_vec().set(table))
    octx, prm = O.ElasticityCtx(2.5, 0.7), (2.5, 0.7)
    A_o, b_o = orc.compute_system("orc_form_elasticity", octx)
    eng.set_form("elasticity", prm)
    eng.set_kernel(4)               # the band-row kernel or an error
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)
    eng.compute_system(A, b)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, 1e-12)
    _close(b.get(), b_o, 1e-12)
    orc.set_fixtable(None)          # the table is what fixes the values: with the faces' constants F differs
    _, b_const = orc.compute_system("orc_form_elasticity", octx)
    assert np.abs(b_const - b_o).max() > 1e-3
    eng.set_kernel(0)               # ... and it is the automatic choice
    eng.compute_system(A, b)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name(), eng.kernel_name()


def test_nonuniform_knots_and_zeroed_matrix(monkeypatch):
    """Stretched knot vectors on every axis (per-element Jacobians differ); IGX_NO_FIRST_TOUCH: MatZeroEntries + read-add-write."""
    monkeypatch.setenv("IGX_NO_FIRST_TOUCH", "1")
    rng = np.random.default_rng(5)
    knots = []
    for n in (9, 5, 4):
        x = np.sort(rng.uniform(0.05, 0.95, n - 1))
        knots.append(np.concatenate([[0.0] * 4, x, [1.0] * 4]))
    orc, eng = make_pair(3, 3, 3, [9, 5, 4], knots=knots)
    _bc((orc, eng), "demo", 3)
    octx, prm = O.ElasticityCtx(1.3, 0.9), (1.3, 0.9)
    A_o, b_o = orc.compute_system("orc_form_elasticity", octx)
    eng.set_form("elasticity", prm)
    eng.set_kernel(4)
    A, b = eng.create_mat(), eng.create_vec()
    eng.compute_system(A, b)
    eng.synchronize()
    assert "block_pencil" in eng.kernel_name()
    compare_mats(A, A_o, 1e-12)
    _close(b.get(), b_o, 1e-12)


def test_automatic_choice_and_repeatability():
    """The automatic kernel choice takes the block pencil for Elasticity3D at p = 3; two assemblies are bitwise identical."""
    orc, eng = make_pair(3, 3, 3, [12, 6, 5])
    _bc((orc, eng), "demo", 3)
    octx, prm = O.ElasticityCtx(1.0, 1.0), (1.0, 1.0)
    A_o, b_o = orc.compute_system("orc_form_elasticity", octx)
    eng.set_form("elasticity", prm)
    outs = []
    for _ in range(2):
        A, b = eng.create_mat(), eng.create_vec()
        _poison(A)
        eng.compute_system(A, b)
        eng.synchronize()
        assert "block_pencil" in eng.kernel_name(), eng.kernel_name()
        compare_mats(A, A_o, 1e-12)
        _close(b.get(), b_o, 1e-12)
        outs.append((A.host(True), b.get()))
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


def test_uncovered_cases_are_refused_not_rerouted():
    """Forced kernel 4 on a case no band-row kernel covers (p = 4; boundary loads on a mapped geometry) fails with PETSC_ERR_SUP."""
    import petiga_amd as P
    from common import warped_geometry
    for case in ("degree", "loads"):
        orc, g = make_pair(3, 3, 4 if case == "degree" else 3, [8, 4, 4])
        if case == "loads":
            X, W = warped_geometry(orc, 3, seed=1, rational=True, amp=0.05)
            g.set_geometry(X, W)
            g.set_boundary_load(1, 1, 0, 1.0)
        g.set_form("elasticity", (1.0, 1.0))
        g.set_kernel(4)
        A, b = g.create_mat(), g.create_vec()
        with pytest.raises(P.IGXError) as e:
            g.compute_system(A, b)
        assert e.value.code == 56


@pytest.mark.parametrize("form,dof,N,geo,bc,p", [
    ("elasticity", 3, (8, 4, 4), "nurbs", "demo", 3),          # demo/Elasticity3D.c on a NURBS patch: the canonical IGA case
    ("elasticity", 3, (9, 5, 4), "poly", "all", 3),            # every face, every field fixed: the lifting through SystemVectorOf
    ("elasticity", 3, (10, 4, 5), "nurbs", "override", 3),
    ("elasticity_f", 3, (8, 5, 4), "nurbs", "all", 3),         # ... with a body force
    ("elasticity_f", 3, (9, 4, 4), "none", "demo", 3),         # (the identity geometry through the same kernel: IGX_BLOCK_PENCIL=... not needed, kernel 4 + no block pencil below)
    ("mass", 2, (8, 4, 5), "nurbs", "demo2", 3),               # two fields, F = N: vec() and the lifting in one vector pass
    ("elasticity", 3, (9, 4, 5), "nurbs", "demo", 2),          # p = 2 (on request: the 27 functions in 4 x 4 x 4 tile slots)
])
def test_constant_coefficient_forms_on_a_mapped_geometry(form, dof, N, geo, bc, p, monkeypatch):
    """Round 6 (asked for in rounds 3, 4, 5): 2- and 3-field constant-coefficient forms off the identity geometry take a band-row
    kernel -- band_pt's point records and physical operands, the Gram pairs as accumulators, block_pencil's coefficient transform,
    whole blocks of dof^2 values per lane -- instead of the feature kernel.  The System driver's vector comes from a sum-factorised
    pass that never sees a K_e (SystemVectorOf: vec() minus the lifting of the Dirichlet values; a fixed row takes its value per
    element).  Engine vs oracle: pattern bit-exact, values to 2e-11 of max|K| over the free rows (mapped geometry)."""
    from common import warped_geometry
    if geo == "none":
        monkeypatch.setenv("IGX_BLOCK_PENCIL", "2")      # (2: band_pt only -- the identity geometry is block_pencil's otherwise)
    orc, eng = make_pair(3, dof, p, list(N))
    if geo != "none":
        X, W = warped_geometry(orc, 3, seed=4, rational=(geo == "nurbs"), amp=0.07)
        orc.set_geometry(X, W); eng.set_geometry(X, W)
    if bc == "demo2":
        for g in (orc, eng):
            g.set_boundary_value(0, 0, 0, 0.5); g.set_boundary_value(0, 0, 1, -0.25); g.set_boundary_value(2, 1, 1, 1.5)
    else:
        _bc((orc, eng), bc, dof)
    if form == "elasticity":
        octx, prm, oname = O.ElasticityCtx(2.5, 0.7), (2.5, 0.7), "orc_form_elasticity"
    elif form == "elasticity_f":
        prm = (2.5, 0.7, 0.3, -1.25, 2.0); octx, oname = (C.c_double * 5)(*prm), "orc_form_elasticity_f"
    else:
        octx, prm, oname = None, (), "orc_form_mass"
    A_o, b_o = orc.compute_system(oname, octx)
    eng.set_form(form, prm)
    eng.set_kernel(4)
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)
    eng.compute_system(A, b)
    eng.synchronize()
    assert "band_pt" in eng.kernel_name() and "dof=%d" % dof in eng.kernel_name(), eng.kernel_name()
    tol = 1e-12 if geo == "none" else 2e-11
    compare_mats(A, A_o, tol)
    _close(b.get(), b_o, tol)
    if p == 3 and geo != "none":      # the automatic choice takes it too
        eng.set_kernel(0)
        A2, b2 = eng.create_mat(), eng.create_vec()
        eng.compute_system(A2, b2)
        eng.synchronize()
        assert "band_pt" in eng.kernel_name(), eng.kernel_name()
        assert np.array_equal(A2.host(True), A.host(True)) and np.array_equal(b2.get(), b.get())
    # Matrix driver: no fix-up
    orc.clear_boundary()
    A_o2, _ = orc.compute_system(oname, octx)
    _poison(A)
    eng.set_kernel(4)
    eng.compute_matrix(A)
    eng.synchronize()
    assert "band_pt" in eng.kernel_name()
    compare_mats(A, A_o2, tol)


def test_matches_feature_kernel_on_a_larger_mesh(monkeypatch):
    """Beyond the oracle's reach: block pencil against the element mode of the feature kernel (40 x 12 x 12, clamped face,
    loaded face): same matrix and vector to rounding; symmetric; rigid translations in the null space of the Matrix driver."""
    import petiga_amd as P

    def build(kernel):
        g = P.IGX(3, 3)
        for i, n in enumerate((40, 12, 12)):
            g.axis_uniform(i, 3, n)
        g.setup()
        for f in range(3):
            g.set_boundary_value(0, 0, f, 0.0)
        g.set_boundary_value(0, 1, 0, 1.0)
        g.set_boundary_load(1, 1, 2, 0.5)
        g.set_form("elasticity", (1.0, 1.0))
        g.set_kernel(kernel)
        A, b = g.create_mat(), g.create_vec()
        _poison(A)
        g.compute_system(A, b)
        g.synchronize()
        return g.kernel_name(), A.host(True), b.get()
    monkeypatch.setenv("IGX_COMBINE", "0")
    k1, a1, b1 = build(4)
    k0, a0, b0 = build(3)
    assert "block_pencil" in k1 and "feature_assemble" in k0
    assert np.abs(a1 - a0).max() <= 1e-12 * np.abs(a0).max()
    assert np.abs(b1 - b0).max() <= 1e-12 * max(np.abs(b0).max(), 1.0)
