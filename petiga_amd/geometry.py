"""NURBS geometry helpers on the data-format side of the path: build / knot-refine a control net and
write it in the layout IGALoad reads (src/petigaio.c:11-139), so rational geometries reach IGXRead without igakit.

Arrays follow the reference's natural order: control net `Pw[n2][n1][n0][nsd+1]` in homogeneous coordinates
(x*w, y*w, z*w, w), axis 0 fastest in memory when flattened.  Host-side numpy only; nothing here is on the
assembly path."""
import numpy as np

IGA_FILE_CLASSID, VEC_FILE_CLASSID = 1211299, 1211214      # include/petiga.h:394, PETSc's VEC_FILE_CLASSID


def greville(U, p):
    """Greville abscissae of a knot vector: the mean of p consecutive knots (IGA_Greville, src/petigaaxis.c)."""
    U = np.asarray(U, dtype=np.float64)
    n = len(U) - p - 1
    return np.array([U[i + 1:i + p + 1].mean() for i in range(n)])


def find_span(p, U, u):
    """Index k with U[k] <= u < U[k+1] (last non-empty span for u == U[-1]); IGA_FindSpan, src/petigabsp.F90."""
    n = len(U) - p - 2
    if u >= U[n + 1]:
        return n
    return int(np.searchsorted(U, u, side="right") - 1)


def insert_knot(p, U, Pw, u, axis):
    """Boehm's single knot insertion along `axis` of the net (axis counted in parametric order: axis 0 is the LAST
    array dimension before the coordinate dimension).  Returns (U', Pw')."""
    U = np.asarray(U, dtype=np.float64)
    k = find_span(p, U, u)
    ax = Pw.ndim - 2 - axis
    P = np.moveaxis(Pw, ax, 0)
    n = P.shape[0]
    Q = np.empty((n + 1,) + P.shape[1:])
    Q[: k - p + 1] = P[: k - p + 1]
    Q[k + 1:] = P[k:]
    for i in range(k - p + 1, k + 1):
        a = (u - U[i]) / (U[i + p] - U[i])
        Q[i] = a * P[i] + (1 - a) * P[i - 1]
    return np.insert(U, k + 1, u), np.moveaxis(Q, 0, ax)


def refine(degrees, knots, Pw, new_knots):
    """Insert every value of new_knots[axis] (list per axis) once."""
    knots = [np.asarray(U, dtype=np.float64) for U in knots]
    for axis, add in enumerate(new_knots):
        for u in add:
            knots[axis], Pw = insert_knot(degrees[axis], knots[axis], Pw, float(u), axis)
    return knots, Pw


def refine_uniform(degrees, knots, Pw, elements):
    """Split every axis (assumed to be one span [0,1]) into `elements[axis]` equal spans."""
    return refine(degrees, knots, Pw, [np.arange(1, n) / n for n in elements])


def quarter_annulus(dim=2, height=2.0):
    """The geometry of test/IGAGeometryMap.c:18-32: radii 1..2 (axis 0, degree 2), a quarter circle (axis 1,
    degree 2, rational), extruded to z in [0, height] (axis 2, degree 1) when dim == 3.  One element."""
    s = np.sqrt(2.0) / 2
    PX = np.array([[1.0, 1.0, 0.0], [1.5, 1.5, 0.0], [2.0, 2.0, 0.0]])     # [i0 radial][i1 angular]
    PY = np.array([[0.0, 1.0, 1.0], [0.0, 1.5, 1.5], [0.0, 2.0, 2.0]])
    PW = np.array([[1.0, s, 1.0]] * 3)
    nz = 2 if dim == 3 else 1
    Pw = np.zeros((nz, 3, 3, dim + 1)) if dim == 3 else np.zeros((3, 3, dim + 1))
    for k in range(nz):
        for j in range(3):
            for i in range(3):
                w = PW[i][j]
                c = [PX[i][j] * w, PY[i][j] * w] + ([height * k * w] if dim == 3 else []) + [w]
                if dim == 3:
                    Pw[k, j, i] = c
                else:
                    Pw[j, i] = c
    degrees = [2, 2] + ([1] if dim == 3 else [])
    knots = [np.array([0, 0, 0, 1, 1, 1.0]), np.array([0, 0, 0, 1, 1, 1.0])] + ([np.array([0, 0, 1, 1.0])] if dim == 3 else [])
    return degrees, knots, Pw


def split_net(Pw):
    """Homogeneous net -> (X[nnodes][nsd], W[nnodes]) in natural order, as IGXSetGeometry takes them."""
    nsd = Pw.shape[-1] - 1
    flat = Pw.reshape(-1, nsd + 1)
    W = flat[:, nsd].copy()
    return flat[:, :nsd] / W[:, None], W


def write_iga(filename, degrees, knots, Pw=None):
    """IGASave layout, big-endian: classid, info, dim, {p, len(U), U}, [nsd, Vec(classid, n, (x*w.., w) per point)]."""
    with open(filename, "wb") as f:
        f.write(np.array([IGA_FILE_CLASSID, 1 if Pw is not None else 0, len(degrees)], dtype=">i4").tobytes())
        for p, U in zip(degrees, knots):
            f.write(np.array([p, len(U)], dtype=">i4").tobytes())
            f.write(np.asarray(U, dtype=">f8").tobytes())
        if Pw is not None:
            nsd = Pw.shape[-1] - 1
            f.write(np.array([nsd, VEC_FILE_CLASSID, Pw.size], dtype=">i4").tobytes())
            f.write(np.ascontiguousarray(Pw, dtype=">f8").tobytes())


def read_iga(filename):
    """Inverse of write_iga: (degrees, knots, Pw or None)."""
    b = open(filename, "rb").read()
    pos = 0

    def ints(n):
        nonlocal pos
        v = np.frombuffer(b, dtype=">i4", count=n, offset=pos); pos += 4 * n
        return [int(x) for x in v]

    def reals(n):
        nonlocal pos
        v = np.frombuffer(b, dtype=">f8", count=n, offset=pos).astype(np.float64); pos += 8 * n
        return v
    cid, info, dim = ints(3)
    if cid != IGA_FILE_CLASSID:
        raise ValueError("Not an IGA in file")
    degrees, knots = [], []
    for _ in range(dim):
        p, nk = ints(2)
        degrees.append(p); knots.append(reals(nk))
    Pw = None
    if info & 1:
        nsd, vid, n = ints(3)
        if vid != VEC_FILE_CLASSID:
            raise ValueError("bad Vec header")
        shape = [len(U) - p - 1 for p, U in zip(degrees, knots)][::-1] + [nsd + 1]
        Pw = reals(n).reshape(shape)
    return degrees, knots, Pw
