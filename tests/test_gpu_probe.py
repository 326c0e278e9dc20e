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
