"""The upper-face-first assembly (an assembly with a communicator forms the elements next to the upper face of
axis 2 first, marks the stream, then the rest; the first-touch rule is per pass) against the single-pass assembly of the same
rank: same local matrix and vector up to the order of the additions.  Seeded sweep over degrees, sizes, periodic axes, forms and
kernels; the exchange itself is covered by tests/test_gpu_comm.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(rng):
    p = int(rng.integers(1, 4))
    per = [bool(rng.random() < 0.2), bool(rng.random() < 0.2), bool(rng.random() < 0.3)]
    size = int(rng.choice([2, 3, 4]))
    form = str(rng.choice(["poisson", "mass", "elasticity"]))
    dof = 1 if form == "poisson" else (3 if form == "elasticity" else int(rng.integers(1, 3)))
    N = []
    for d in range(3):
        lo = 2 * p + 1 if per[d] else 2
        N.append(int(rng.integers(max(lo, 2), lo + 6)))
    N[2] = int(rng.integers(2 * (p + 1), 2 * (p + 1) + 5)) * size          # every rank keeps >= 2(p+1) elements on axis 2
    if rng.random() < 0.4:
        N[0] = max(N[0], 8)                                                  # long enough for the axis-0 walks
    return dict(p=p, periodic=per, size=size, form=form, dof=dof, N=N, kernel=int(rng.choice([0, 0, 3])), combine=bool(rng.random() < 0.3))


@pytest.mark.parametrize("seed", range(40))
def test_two_pass_assembly_equals_single_pass(seed, monkeypatch):
    import petiga_amd as P
    c = _case(np.random.default_rng(7000 + seed))
    if c["combine"]:
        monkeypatch.setenv("IGX_COMBINE", "1")
    results = []
    for overlap in ("1", "0"):
        monkeypatch.setenv("IGX_OVERLAP", overlap)
        g = P.IGX(3, c["dof"])
        g.set_comm(c["size"], 0)                       # rank 0 always has an upper neighbour on axis 2 (the partition cuts axis 2 first)
        for i in range(3):
            g.axis_uniform(i, c["p"], c["N"][i], periodic=c["periodic"][i])
        g.setup()
        if g.sizes()["proc_sizes"][2] < 2:
            pytest.skip("the partition did not cut axis 2")
        g.set_kernel(c["kernel"])
        for d in range(3):
            if not c["periodic"][d]:
                g.set_boundary_value(d, 0, 0, 0.5 + d)
        g.set_form(c["form"], (1.7, 0.6) if c["form"] == "elasticity" else ())
        g.comm_init_transport(lambda send, recv: None)      # a communicator is all the assembly looks at; nothing is exchanged here
        A, b = g.create_mat(), g.create_vec()
        g.compute_system(A, b)
        g.synchronize()
        results.append((A.host(True).copy(), b.get().copy(), g.dominant_kernel()["launches"], g.kernel_name()))
    (v1, b1, l1, k1), (v0, b0, l0, k0) = results
    assert k1 == k0
    if ("gram_pencil" in k1 and "walk=0" in k1) or "feature_assemble" in k1 or "block_pencil" in k1:     # (the other walks of the pencil kernel keep one pass)
        assert l1 > l0, (c, k1, l1, l0)                 # the face pass adds launches
    scale = np.abs(v0).max()
    assert np.abs(v1 - v0).max() <= 1e-13 * scale, c
    assert np.abs(b1 - b0).max() <= 1e-13 * max(np.abs(b0).max(), 1.0), c
