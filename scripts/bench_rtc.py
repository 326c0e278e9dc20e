#!/usr/bin/env python3
"""Run-time compiled forms against the same forms built in: one assembly each (not the headline bench)."""
import sys
import time

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests"); sys.path.insert(0, "oracle")
import petiga_amd as P
from test_rtc_forms import USER_ELASTICITY

POISSON = r"""
struct UserPoisson {   // demo/Poisson3D.c:3-23
  static constexpr int DOF = 1, ORDER = 1; static constexpr unsigned NEED = 0;
  static constexpr unsigned MAT_TEST_MASK = 0xEu;
  static __device__ void mat(const PtView &, const double *Na, const double *Nb, double *T) { T[0] = Na[1] * Nb[1] + Na[2] * Nb[2] + Na[3] * Nb[3]; }
  static __device__ void vec(const PtView &, const double *Na, double *R) { R[0] = Na[0]; }
};
"""


def run(label, dof, p, n, setform, kernel=0):
    g = P.IGX(3, dof)
    for i in range(3):
        g.axis_uniform(i, p, n)
    g.setup()
    g.set_kernel(kernel)
    for f in range(dof):
        g.set_boundary_value(0, 0, f, 0.0)
    setform(g)
    A, b = g.create_mat(), g.create_vec()
    ts = []
    for _ in range(3):
        g.synchronize(); t = time.perf_counter()
        g.compute_system(A, b); g.synchronize()
        ts.append(time.perf_counter() - t)
    print("%-46s first %7.1f ms, then %8.2f ms %7.2f M el/s  %s" % (label, ts[0] * 1e3, min(ts[1:]) * 1e3, n ** 3 / min(ts[1:]) / 1e6, g.kernel_name()))


for p, n in ((2, 64), (3, 48)):
    run("Poisson p=%d %d^3 built in (feature kernel)" % (p, n), 1, p, n, lambda g: g.set_form("poisson"), kernel=3)
    run("Poisson p=%d %d^3 from source" % (p, n), 1, p, n, lambda g: g.set_form_source(POISSON, "UserPoisson"))
    run("Poisson p=%d %d^3 from source, point-form kernel" % (p, n), 1, p, n, lambda g: g.set_form_source(POISSON, "UserPoisson"), kernel=1)
run("Elasticity p=3 48^3 built in (band rows)", 3, 3, 48, lambda g: g.set_form("elasticity", (1.0, 1.0)))
from test_rtc_boundary_scalar import ELASTICITY_BANDS
run("Elasticity p=3 48^3 from source, band rows", 3, 3, 48, lambda g: g.set_form_source(ELASTICITY_BANDS, "UserElasticityBands", (1.0, 1.0)))
run("Elasticity p=3 48^3 from source, Gram (element mode)", 3, 3, 48, lambda g: g.set_form_source(USER_ELASTICITY, "UserElasticity<1>", (1.0, 1.0)))
run("Elasticity p=3 48^3 from source, plain", 3, 3, 48, lambda g: g.set_form_source(USER_ELASTICITY, "UserElasticity<0>", (1.0, 1.0)))
run("Elasticity p=3 48^3 from source, point-form kernel", 3, 3, 48, lambda g: g.set_form_source(USER_ELASTICITY, "UserElasticity<0>", (1.0, 1.0)), kernel=1)


# NavierStokesVMS Tangent (demo/NavierStokesVMS.c:166-244), 48^3 on the bench's NURBS map, axes 0 and 2 periodic: built in, the built-in
# struct's text as source, a plain struct (the Tangent written once in mat_c) -- all on band_pt -- and the plain struct on the feature kernel
def run_vms(label, setform, kernel=0):
    import bench
    g = P.IGX(3, 4)
    for i, per in enumerate((True, False, True)):
        g.axis_uniform(i, 3, 48, periodic=per)
    g.setup()
    g.set_kernel(kernel)
    X, W = bench._bench_geometry(3, 48, [True, False, True])
    g.set_geometry(X, W)
    for side in range(2):
        for f in range(3):
            g.set_boundary_value(1, side, f, 0.0)
    setform(g)
    J = g.create_mat()
    rng = np.random.default_rng(5)
    U, V = g.create_vec().set(0.1 + 0.05 * rng.standard_normal(J.nbrows * 4)), g.create_vec().set(np.zeros(J.nbrows * 4))
    ts = []
    for _ in range(4):
        g.synchronize(); t = time.perf_counter()
        g.compute_ijacobian(200.0, V, 0.0, U, J); g.synchronize()
        ts.append(time.perf_counter() - t)
    print("%-46s first %7.1f ms, then %8.2f ms %7.2f M el/s  %s" % (label, ts[0] * 1e3, min(ts[1:]) * 1e3, 48 ** 3 / min(ts[1:]) / 1e6, g.kernel_name()))


from test_rtc_band_pt import PLAIN_VMS, builtin_text_as_source
VMS = (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2)
run_vms("NS-VMS Tangent p=3 48^3 NURBS built in", lambda g: g.set_form("nsvms", VMS))
_src, _name = builtin_text_as_source()
run_vms("NS-VMS Tangent 48^3 built-in text as source", lambda g: g.set_form_source(_src, _name, VMS))
run_vms("NS-VMS Tangent 48^3 plain struct as source", lambda g: g.set_form_source(PLAIN_VMS, "UserVMS", VMS))
run_vms("NS-VMS Tangent 48^3 plain struct, feature", lambda g: g.set_form_source(PLAIN_VMS, "UserVMS", VMS), kernel=3)
