"""Measurement support of the C ABI (not reference functions): the shader-clock probe bench.py reports next to the roofline."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_clock_probe_reports_a_plausible_shader_clock(monkeypatch):
    """IGXGetClockProbe (include/petiga_amd.h): measurement support for bench.py's roofline block, not a reference function."""
    import petiga_amd as P

    def problem():
        g = P.IGX(3, 1)
        for i in range(3):
            g.axis_uniform(i, 3, 16)
        g.setup()
        for d in range(3):
            for s in range(2):
                g.set_boundary_value(d, s, 0, 1.0)
        g.set_form("poisson")
        A, b = g.create_mat(), g.create_vec()
        g.compute_system(A, b)
        g.synchronize()
        return g, A, b

    monkeypatch.delenv("IGX_CLOCK_PROBE", raising=False)
    g, A, b = problem()
    with pytest.raises(P.IGXError):
        g.clock_probe()
    ref = A.host(True).copy()
    monkeypatch.setenv("IGX_CLOCK_PROBE", "1")
    g, A, b = problem()
    assert "pencil" in g.kernel_name()
    mhz, ne = g.clock_probe()
    assert 300.0 < mhz < 3000.0 and ne > 0
    assert np.array_equal(A.host(True), ref)          # the probe does not touch the results
    with pytest.raises(P.IGXError):                   # the sums were cleared
        g.clock_probe()


@pytest.mark.parametrize("geo", [False, True])
def test_fused_and_separate_launches_of_the_row_field_groups_agree_bitwise(monkeypatch, geo):
    """NS-VMS p=3 (dof 4 at 4x4 tiles) forms its two groups of row fields in one launch (feature_mfma.hpp, FUSE); IGX_FUSE_GROUPS=0
    keeps the one-launch-per-group path alive: same arithmetic in the same order, so the matrices are identical bit for bit.
    (Both are compared with the oracle in test_gpu_round2.py::test_navier_stokes_vms_p3.)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import petiga_amd as P
    from common import make_pair, warped_geometry
    mats, names = [], []
    rng = np.random.default_rng(9)
    U = V = None
    for fuse in ("1", "0"):
        monkeypatch.setenv("IGX_FUSE_GROUPS", fuse)
        orc, eng = make_pair(3, 4, 3, [5, 3, 4], periodic=[True, False, True])
        if geo:
            X, W = warped_geometry(orc, 3, seed=5, rational=True, amp=0.08)
            eng.set_geometry(X, W)
        for side in range(2):
            for f in range(3):
                eng.set_boundary_value(1, side, f, 0.0)
        eng.set_form("nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, 1e-2))
        if U is None:
            n = orc.global_size()
            U, V = rng.standard_normal(n) * 0.3, rng.standard_normal(n) * 0.1
        Uv, Vv, J = eng.create_vec().set(U), eng.create_vec().set(V), eng.create_mat()
        eng.compute_ijacobian(200.0, Vv, 0.0, Uv, J)
        eng.synchronize()
        mats.append(J.host(True).copy()); names.append(eng.kernel_name())
    assert "fused" in names[0] and "fused" not in names[1]
    assert np.array_equal(mats[0], mats[1])
