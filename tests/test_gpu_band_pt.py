"""GPU parity of band_pt (petiga_amd/csrc/band_pt.hpp): the Tangent of demo/NavierStokesVMS.c:166-244 assembled as band rows, one
node layer of a pencil at a time, with the physical basis features built per k-step from the 1-D rows and the point records of
band_points (JW, inverse Jacobian, rational data, state, tau_M / tau_C).  IGXSetKernel(4) insists on the kernel.  Engine vs
oracle on identical inputs: pattern bit-exact, values to 1e-11 of max|K| over the rows without a Dirichlet condition."""
import ctypes as C

import numpy as np
import pytest

import oracle_api as O
from common import compare_mats, make_pair, warped_geometry

pytestmark = pytest.mark.gpu

NU, FX, DT = 1.472e-4, 3.37204e-3, 1e-2


def _poison(mat):
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    _, _, val = mat.device_ptrs()
    assert hip.hipMemset(val, 0xFF, mat.nblocks * mat.bs * mat.bs * 8) == 0
    assert hip.hipDeviceSynchronize() == 0


def _walls(objs, periodic, kind):
    for g in objs:
        if kind == "noslip":          # demo/NavierStokesVMS.c:362-385: u = 0 on the faces of the non-periodic axes
            for d in range(3):
                if not periodic[d]:
                    for side in range(2):
                        for f in range(3):
                            g.set_boundary_value(d, side, f, 0.0)
        elif kind == "mixed":         # values on single fields of single faces, pressure included
            for d in range(3):
                if not periodic[d]:
                    g.set_boundary_value(d, 0, d, 0.3)
                    g.set_boundary_value(d, 1, 3, -0.2)


@pytest.mark.parametrize("p", [3, 2])
@pytest.mark.parametrize("N,periodic,geo,bc,nseg", [
    ((8, 4, 4), (False, False, False), None, "noslip", 0),
    ((9, 5, 4), (False, False, False), "nurbs", "mixed", 0),
    ((10, 4, 5), (False, False, False), "poly", "none", 3),
    ((8, 4, 7), (True, False, True), None, "noslip", 0),         # config 5's topology on one rank: axes 0 and 2 wrapped inside the rank
    ((9, 3, 8), (True, False, True), "nurbs", "noslip", 0),
    ((11, 4, 7), (True, False, False), "nurbs", "mixed", 2),
    ((8, 5, 9), (False, False, True), "poly", "noslip", 0),
])
def test_ns_vms_tangent_vs_oracle(N, periodic, geo, bc, nseg, p, monkeypatch):
    if nseg:
        monkeypatch.setenv("IGX_NSEG", str(nseg))
    orc, eng = make_pair(3, 4, p, list(N), periodic=list(periodic))
    if geo:
        X, W = warped_geometry(orc, 3, seed=sum(N), rational=(geo == "nurbs"), amp=0.08)
        orc.set_geometry(X, W)
        eng.set_geometry(X, W)
    _walls((orc, eng), periodic, bc)
    ctx, params = O.NSVMSCtx(NU, FX, 0.0, 0.0, DT), (NU, FX, 0.0, 0.0, DT)
    rng = np.random.default_rng(29)
    n = orc.global_size()
    U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n) * 0.1
    shift = 2.0 / DT
    J_o = orc.compute_ijacobian("orc_form_ns_tangent", ctx, shift, V, 0.0, U)
    eng.set_form("nsvms", params)
    eng.set_kernel(4)
    Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
    _poison(J)
    eng.compute_ijacobian(shift, Vv, 0.0, Uv, J)
    eng.synchronize()
    assert "band_pt" in eng.kernel_name(), eng.kernel_name()
    compare_mats(J, J_o, 1e-11)


def test_automatic_choice_repeatability_and_the_feature_kernel():
    """The automatic choice takes band_pt for the NS-VMS IJacobian at p = 3; two assemblies are bitwise identical; the element
    mode of the feature kernel gives the same matrix to rounding on a mesh beyond the oracle's reach."""
    import petiga_amd as P
    outs = {}
    for kernel in (0, 0, 3):
        g = P.IGX(3, 4)
        for i, (n, per) in enumerate(((24, True), (8, False), (12, True))):
            g.axis_uniform(i, 3, n, periodic=per)
        g.setup()
        for side in range(2):
            for f in range(3):
                g.set_boundary_value(1, side, f, 0.0)
        g.set_form("nsvms", (NU, FX, 0.0, 0.0, DT))
        g.set_kernel(kernel)
        rng = np.random.default_rng(3)
        J = g.create_mat()
        U, V = g.create_vec().set(rng.standard_normal(J.nbrows * 4) * 0.3), g.create_vec().set(np.zeros(J.nbrows * 4))
        _poison(J)
        g.compute_ijacobian(2.0 / DT, V, 0.0, U, J)
        g.synchronize()
        outs.setdefault(kernel, []).append((J.host(True), g.kernel_name()))
    assert all("band_pt" in k for _, k in outs[0]) and "feature_assemble" in outs[3][0][1]
    assert np.array_equal(outs[0][0][0], outs[0][1][0])
    scale = np.abs(outs[3][0][0]).max()
    assert np.abs(outs[0][0][0] - outs[3][0][0]).max() <= 1e-11 * scale
