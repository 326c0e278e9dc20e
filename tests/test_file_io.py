"""IGA / Vec file formats either side of the path: IGXRead / IGXWrite against files written
independently in the test (layout of IGASave / IGALoad, src/petigaio.c:11-139, and of VecView binary)."""
import numpy as np
import pytest

import petiga_amd as P
from common import IGA_FILE_CLASSID, VEC_FILE_CLASSID, iga_file_bytes, make_pair, rel_err, vec_file_bytes, warped_geometry, compare_mats


def _case(dim, p, N, rational, seed=1):
    orc, _ = make_pair(dim, 1, p, N, engine=False)
    X, W = warped_geometry(orc, dim, seed=seed, rational=rational)
    U = [np.array(orc.axis(i)["U"]) for i in range(dim)]
    return orc, U, X, W


@pytest.mark.parametrize("dim,p,N,rational", [(1, 2, 5, False), (2, 2, 4, True), (3, 3, 3, True), (3, 2, 4, False)])
def test_read_then_write_is_identity(tmp_path, dim, p, N, rational):
    _, U, X, W = _case(dim, p, N, rational)
    src = iga_file_bytes([p] * dim, U, X, W)
    f = tmp_path / "geo.dat"; f.write_bytes(src)
    g = P.IGX(); g.set_dof(1)
    g.read(f)
    g.setup()
    sz = g.sizes()
    assert list(sz["elem_sizes"]) == [N] * dim + [1] * (3 - dim)
    out = tmp_path / "out.dat"
    g.write(out)
    got = out.read_bytes()
    assert len(got) == len(src)
    if rational:     # X -> X*w -> X/w -> X*w : within an ulp of the file
        a = np.frombuffer(src[-8 * (len(X) * (dim + 1)):], dtype=">f8"); b = np.frombuffer(got[-8 * (len(X) * (dim + 1)):], dtype=">f8")
        assert src[:-8 * a.size] == got[:-8 * a.size]
        assert np.allclose(a, b, rtol=4e-16, atol=0)
    else:
        assert got == src


@pytest.mark.parametrize("dim,p,N,geo", [(2, 2, 4, True), (3, 2, 3, False), (1, 3, 5, True)])
def test_property_array_travels_with_the_file(tmp_path, dim, p, N, geo):
    """IGASave / IGALoad with info bit 1 (src/petigaio.c:38,65-70,105,130-135): the property dimension and the Vec in natural order
    behind the geometry block."""
    _, U, X, W = _case(dim, p, N, True)
    rng = np.random.default_rng(11)
    A = rng.standard_normal((len(X), 3))
    src = iga_file_bytes([p] * dim, U, X if geo else None, W if geo else None, A)
    f = tmp_path / "prop.dat"; f.write_bytes(src)
    g = P.IGX(); g.set_dof(1); g.read(f); g.setup()
    assert g.property_dim() == 3
    out = tmp_path / "out.dat"; g.write(out)
    got = out.read_bytes()
    assert len(got) == len(src) and got[-8 * A.size:] == src[-8 * A.size:]      # the array itself bit for bit
    assert got[:12] == src[:12]
    g.set_property(None)                                                          # dropped: the file loses bit 1
    assert g.property_dim() == 0
    g.write(out)
    assert out.read_bytes()[4:8] == np.array([1 if geo else 0], dtype=">i4").tobytes()
    trunc = tmp_path / "t.dat"; trunc.write_bytes(src[:-8])
    with pytest.raises(P.IGXError) as e:
        P.IGX().read(trunc)
    assert e.value.code == 66


def test_no_geometry_and_header_checks(tmp_path):
    U = [np.array([0, 0, 0, .25, .5, .5, 1, 1, 1.])]
    f = tmp_path / "a.dat"; f.write_bytes(iga_file_bytes([2], U))
    g = P.IGX(); g.set_dof(2); g.read(f); g.setup()
    assert g.sizes()["node_sizes"][0] == 6
    out = tmp_path / "b.dat"; g.write(out)
    assert out.read_bytes() == f.read_bytes()
    bad = tmp_path / "bad.dat"; bad.write_bytes(vec_file_bytes([1.0, 2.0]))
    with pytest.raises(P.IGXError) as e:
        g.read(bad)                       # "Not an IGA in file": PETSC_ERR_ARG_WRONG (src/petigaio.c:32)
    assert e.value.code == 62
    with pytest.raises(P.IGXError) as e:
        g.read(tmp_path / "missing.dat")  # PETSC_ERR_FILE_OPEN
    assert e.value.code == 65
    trunc = tmp_path / "t.dat"; trunc.write_bytes(f.read_bytes()[:-4])
    with pytest.raises(P.IGXError) as e:
        g.read(trunc)                     # PETSC_ERR_FILE_READ
    assert e.value.code == 66


def test_uniform_weights_are_not_rational(tmp_path):
    # weights equal within 100 eps -> polynomial geometry (src/petigaio.c:253-255); X = Xw/w still applied
    _, U, X, _ = _case(2, 2, 3, False)
    W = np.full(len(X), 2.0)
    f = tmp_path / "g.dat"; f.write_bytes(iga_file_bytes([2, 2], U, X, W))
    g = P.IGX(); g.set_dof(1); g.read(f); g.setup()
    out = tmp_path / "o.dat"; g.write(out)
    assert out.read_bytes() == iga_file_bytes([2, 2], U, X, None)


@pytest.mark.gpu
@pytest.mark.parametrize("dim,p,N,rational", [(2, 2, 6, True), (3, 3, 4, True), (3, 2, 5, False)])
def test_poisson_on_geometry_read_from_file(tmp_path, dim, p, N, rational):
    orc, U, X, W = _case(dim, p, N, rational, seed=7)
    f = tmp_path / "geo.dat"; f.write_bytes(iga_file_bytes([p] * dim, U, X, W))
    eng = P.IGX(); eng.set_dof(1); eng.read(f); eng.setup()
    orc.set_geometry(X, W)
    for d in range(dim):
        for side in range(2):
            orc.set_boundary_value(d, side, 0, 0.5 * d + side); eng.set_boundary_value(d, side, 0, 0.5 * d + side)
    eng.set_form("poisson")
    A = eng.create_mat(); b = eng.create_vec()
    eng.compute_system(A, b)
    Ao, bo = orc.compute_system("orc_form_poisson")
    compare_mats(A, Ao, 1e-12)
    assert rel_err(b.get(), bo) < 1e-12


@pytest.mark.gpu
def test_vec_file_round_trip(tmp_path):
    _, eng = make_pair(2, 2, 2, 4)
    v = eng.create_vec()
    x = np.random.default_rng(0).standard_normal(v.n)
    v.set(x)
    f = tmp_path / "v.dat"
    eng.write_vec(v, f)
    assert f.read_bytes() == vec_file_bytes(x)
    w = eng.create_vec()
    eng.read_vec(w, f)
    assert np.array_equal(w.get(), x)
    _, other = make_pair(2, 1, 2, 4)
    with pytest.raises(P.IGXError) as e:
        other.read_vec(other.create_vec(), f)
    assert e.value.code == 62


def test_corrupt_headers_are_errors_and_leave_the_iga_untouched(tmp_path):
    """A hostile or truncated file comes back as PETSC_ERR_FILE_READ (66) / ARG_WRONG, never as an exception through the C ABI,
    and a failed read leaves the discretisation as it was."""
    import struct
    import petiga_amd as P
    g = P.IGX(2, 1)
    for i in range(2):
        g.axis_uniform(i, 2, 5)
    g.setup()
    before = g.sizes()
    U = np.array([0, 0, 0, 0.5, 1, 1, 1.0])
    good = iga_file_bytes([2, 2], [U, U])
    cases = {
        "huge-knot-count": struct.pack(">iii", IGA_FILE_CLASSID, 0, 2) + struct.pack(">ii", 2, 2 ** 30),
        "truncated-knots": good[:len(good) - 20],
        "huge-geometry": good[:4] + struct.pack(">i", 1) + good[8:] + struct.pack(">iii", 2, VEC_FILE_CLASSID, 2 ** 30),
        "negative-geometry": good[:4] + struct.pack(">i", 1) + good[8:] + struct.pack(">iii", 2, VEC_FILE_CLASSID, -5),
        "bad-dim": struct.pack(">iii", IGA_FILE_CLASSID, 0, 7),
    }
    for name, blob in cases.items():
        f = tmp_path / (name + ".dat")
        f.write_bytes(blob)
        with pytest.raises(P.IGXError) as e:
            g.read(f)
        assert e.value.code in (62, 66), (name, e.value.code)
        assert g.sizes() == before, name          # still set up, unchanged
    f = tmp_path / "good.dat"
    f.write_bytes(good)
    g.read(f)
    g.setup()
    assert g.sizes()["elem_sizes"][:2] == [2, 2]
