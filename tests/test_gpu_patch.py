"""gram_patch (petiga_amd/csrc/gram_patch.hpp, round 6): the p = 2 Gram walk in patches of 4 x 3 pencils whose twelve wavefronts add into ONE
window of band rows in LDS -- combined across all three axes before a run reaches memory, 4 colours instead of 9.  The default for
these cases since round 6 (IGX_PATCH=0: the pencil walk); the one path that is not bit-repeatable -- the order of the LDS atomics of the
twelve wavefronts is not fixed.
Engine vs oracle (demo/Poisson3D.c:3-23 through IGAComputeMatrix / IGAComputeSystem): pattern bit-exact, values to 1e-12."""
import ctypes as C

import numpy as np
import pytest

from common import compare_mats, make_pair

pytestmark = pytest.mark.gpu


def _poison(mat):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _, _, val = mat.device_ptrs()
    assert hip.hipMemset(val, 0xFF, mat.nblocks * mat.bs * mat.bs * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("N,nseg", [((9, 8, 6), 0),        # whole patches: 2 x 2
                                    ((10, 9, 7), 0),       # partial patches on both axes (9 = 2 x 4 + 1, 7 = 2 x 3 + 1)
                                    ((17, 5, 4), 3),       # three segments along the walk: halo elements, owned rows
                                    ((8, 4, 3), 0),        # one patch
                                    ((8, 5, 4), 4),        # segments of two elements (the floor since round 6: small meshes are bound by the walk's length)
                                    ((12, 13, 11), 2)])
def test_matrix_driver_vs_oracle(N, nseg, monkeypatch):
    monkeypatch.setenv("IGX_PATCH", "1")
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, 2, list(N))
    A_o, _ = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A = eng.create_mat()
    _poison(A)
    eng.compute_matrix(A)
    eng.synchronize()
    assert "gram_patch" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, 1e-12)
    vals = A.host(True).copy()
    _poison(A)
    eng.compute_matrix(A)
    eng.synchronize()
    # NOT bit-repeatable (up to nine wavefronts add to a window entry with LDS atomics: the order of the adds is not fixed) -- the one
    # path of the library that is not, hence a switch; two assemblies agree to a few ulps of the largest entry
    assert np.abs(A.host(True) - vals).max() <= 4e-15 * np.abs(vals).max()


@pytest.mark.parametrize("N,bc,nseg", [((9, 8, 6), "all", 0), ((10, 9, 7), "all", 0), ((17, 5, 4), "some", 3), ((12, 13, 11), "all", 2), ((9, 5, 4), "none", 0), ((8, 5, 4), "all", 4)])
def test_system_driver_vs_oracle(N, bc, nseg, monkeypatch):
    """demo/Poisson3D.c:37-51: Dirichlet values on the faces (IGAElementFixSystem on the combined runs: fixed rows and columns emptied, the
    diagonal counting the elements of the patch's walk that hold the node, the lifting of a row gathered from its runs), F = N * 1, a
    boundary load; first-touch stores on a NaN-poisoned matrix."""
    monkeypatch.setenv("IGX_PATCH", "1")
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, 2, list(N))
    for g in (orc, eng):
        if bc == "all":
            for d in range(3):
                for s in range(2):
                    g.set_boundary_value(d, s, 0, 1.0 + 0.5 * d - 0.25 * s)
        elif bc == "some":
            g.set_boundary_value(0, 0, 0, 0.5); g.set_boundary_value(1, 1, 0, -1.0); g.set_boundary_value(2, 0, 0, 2.0)
            g.set_boundary_load(2, 1, 0, 1.5)
    A_o, b_o = orc.compute_system("orc_form_poisson")
    eng.set_form("poisson")
    A, b = eng.create_mat(), eng.create_vec()
    _poison(A)
    eng.compute_system(A, b)
    eng.synchronize()
    assert "gram_patch" in eng.kernel_name(), eng.kernel_name()
    compare_mats(A, A_o, 1e-12)
    assert np.abs(b.get() - b_o).max() <= 1e-12 * max(np.abs(b_o).max(), 1.0)
    vals, bv = A.host(True).copy(), b.get().copy()
    _poison(A)
    eng.compute_system(A, b)
    eng.synchronize()
    assert np.abs(A.host(True) - vals).max() <= 4e-15 * np.abs(vals).max() and np.abs(b.get() - bv).max() <= 4e-15 * max(np.abs(bv).max(), 1.0)


CH = (1.5, 200.0, 0.63, 1.0, 1.0 / 48.0, 1.0)      # (the parameters of tests/test_gpu_state_pencil.py)


@pytest.mark.parametrize("form,N,bc,nseg", [
    ("cahnhilliard", (9, 8, 6), False, 0),        # whole patches of 4 x 2 pencils
    ("cahnhilliard", (10, 9, 7), True, 0),        # partial patches on both axes, Dirichlet values on four faces
    ("cahnhilliard", (17, 5, 3), True, 3),        # three segments along the walk
    ("cahnhilliard", (8, 4, 2), False, 0),        # one patch
    ("cahnhilliard", (9, 4, 3), True, 4),         # segments of three, three and three elements, two of halo each
    ("bratu", (12, 13, 11), True, 2),
    ("bratu", (9, 5, 4), False, 0),
])
def test_tangent_patch_walk_vs_oracle(form, N, bc, nseg, monkeypatch):
    """state_patch_p2: the Tangent of demo/CahnHilliard3D.c:111-179 / demo/Bratu.c through IGAComputeIJacobian on patches of 4 x 2 pencils
    (IGX_PATCH_STATE=1), IGAElementFixJacobian on the combined runs, first-touch stores on a NaN-poisoned matrix."""
    import oracle_api as O
    monkeypatch.setenv("IGX_PATCH_STATE", "1")
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 1, 2, list(N))
    if bc:
        for g in (orc, eng):
            g.set_boundary_value(0, 0, 0, 0.6); g.set_boundary_value(0, 1, 0, 0.66)
            g.set_boundary_value(1, 1, 0, 0.61); g.set_boundary_value(2, 0, 0, 0.65)
    rng = np.random.default_rng(5)
    n = orc.global_size()
    U, V = 0.63 + 0.05 * (2 * rng.random(n) - 1), rng.standard_normal(n)
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    if form == "cahnhilliard":
        eng.set_form("cahnhilliard", CH)
        J_o = orc.compute_ijacobian("orc_form_ch_tangent", O.CahnHilliardCtx(*CH), 250.0, V, 0.0, U)
        shift = 250.0
    else:
        eng.set_form("bratu", (3.5,))
        J_o = orc.compute_ijacobian("orc_form_bratu_ijacobian", C.c_double(3.5), 4.0, V, 0.0, U)
        shift = 4.0
    _poison(J)
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "state_patch" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-11)
    vals = J.host(True).copy()
    _poison(J)
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert np.abs(J.host(True) - vals).max() <= 4e-15 * np.abs(vals).max()      # (not bit-repeatable, like gram_patch)
