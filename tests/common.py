"""Shared helpers: build the same discretisation in the CPU oracle and in the engine."""
import numpy as np

import oracle_api as O
from petiga_amd.geometry import greville      # noqa: F401  (the tests take it from here)


def make_pair(dim, dof, p, N, C=None, periodic=None, nqp=None, order=None, knots=None, engine=True):
    """Returns (oracle, engine) set up identically.  p/N/C/periodic are per-axis lists or scalars."""
    ls = lambda v, d=None: (list(v) if isinstance(v, (list, tuple)) else [v] * dim) if v is not None else [d] * dim
    p, N, C, periodic, nqp = ls(p), ls(N), ls(C, -1), ls(periodic, False), ls(nqp, None)
    orc = O.OracleIGA(dim, dof)
    eng = None
    if engine:
        import petiga_amd as P
        eng = P.IGX(dim, dof)
    for i in range(dim):
        if knots is not None and knots[i] is not None:
            orc.axis_knots(i, p[i], knots[i], periodic=periodic[i])
            if eng: eng.axis_knots(i, p[i], knots[i], periodic=periodic[i])
        else:
            orc.axis_uniform(i, p[i], N[i], C[i], periodic=periodic[i])
            if eng: eng.axis_uniform(i, p[i], N[i], C[i], periodic=periodic[i])
        if nqp[i] is not None:
            orc.set_quadrature(i, nqp[i])
            if eng: eng.set_quadrature(i, nqp[i])
    if order is not None:
        orc.set_order(order)
        if eng: eng.set_order(order)
    orc.setup()
    if eng: eng.setup()
    return orc, eng




def warped_geometry(orc, dim, seed=0, rational=True, amp=0.15):
    """A valid (positive Jacobian) smooth geometry: Greville grid + smooth warp, optional NURBS weights."""
    rng = np.random.default_rng(seed)
    g = [greville(orc.axis(i)["U"], orc.axis(i)["p"]) for i in range(dim)]
    shape = [len(x) for x in g][::-1]
    mesh = np.meshgrid(*g[::-1], indexing="ij")[::-1]          # mesh[i] varies along axis i, arrays [n2][n1][n0]
    X = np.stack([m.copy() for m in mesh], axis=-1)
    ph = rng.uniform(0, 2 * np.pi, size=(dim, dim))
    for i in range(dim):
        for j in range(dim):
            if i != j:
                X[..., i] += amp / dim * np.sin(2 * np.pi * mesh[j] + ph[i, j]) * (0.5 + 0.5 * mesh[i])
    X[..., 0] *= 1.3
    W = rng.uniform(0.8, 1.2, size=shape) if rational else None
    return X.reshape(-1, dim), (None if W is None else W.reshape(-1))


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    d = np.abs(a - b).max() if a.size else 0.0
    return d / max(np.abs(b).max(), 1e-300) if b.size else d


def free_row_scale(rows, cols, vals):
    """max|K| over the rows that are NOT Dirichlet rows (a fixed row holds only its diagonal, the element multiplicity -- up to
    16 and more -- while stiffness entries are O(h^(dim-2)): scaling by the global maximum would loosen a stated tolerance on
    the real entries by 10^2-10^3).  rows / cols / vals: coordinate list of the reference matrix."""
    n = int(rows.max()) + 1 if rows.size else 0
    offdiag = np.zeros(n, dtype=bool)
    nzoff = (rows != cols) & (vals != 0.0)
    offdiag[rows[nzoff]] = True
    free = offdiag[rows]
    return np.abs(vals[free]).max() if free.any() else np.abs(vals).max()


def compare_mats(eng_mat, orc_mat, tol):
    """Engine matrix (device block CSR) vs oracle CSR: identical pattern (explicit zeros included, as
    IGACreateMat preallocates it), values within tol of max|K| over the rows without a Dirichlet condition."""
    rows, cols, vals = eng_mat.to_coo_global()
    n = orc_mat.nrows
    ke = rows * n + cols
    o = np.argsort(ke, kind="stable")
    ke, vals = ke[o], vals[o]
    orow = np.repeat(np.arange(n, dtype=np.int64), np.diff(orc_mat.rowptr))
    ko = orow * n + orc_mat.colidx.astype(np.int64)
    oo = np.argsort(ko, kind="stable")
    ko, vo = ko[oo], orc_mat.val[oo]
    assert ke.size == ko.size and np.array_equal(ke, ko), "sparsity pattern differs"
    # Scale: max|K| over the rows that are NOT Dirichlet rows.  A fixed row holds only its diagonal (the element
    # multiplicity, up to 2^dim * ... = 16 and more), while stiffness entries are O(h^(dim-2)): scaling by the global maximum
    # would loosen the stated tolerance on the real entries by 10^2-10^3.
    scale = free_row_scale(ko // n, ko % n, vo)
    err = np.abs(vals - vo).max()
    assert err <= tol * scale, "matrix values differ: %g (scale %g)" % (err, scale)
    return err / scale


# ---- PETSc-binary IGA files, written independently of the library (format of IGASave, src/petigaio.c:75-139) ----
IGA_FILE_CLASSID, VEC_FILE_CLASSID = 1211299, 1211214


def iga_file_bytes(degrees, knots, X=None, W=None, A=None):
    """Big-endian: classid, info, dim, {p, nk, U}, [nsd, Vec(classid, n, (x*w.., w) per control point)], [npd, Vec(classid, n, A[node][npd])]
    (IGASave, src/petigaio.c:75-139: info bit 0 geometry, bit 1 property)."""
    out = [np.array([IGA_FILE_CLASSID, (1 if X is not None else 0) | (2 if A is not None else 0), len(degrees)], dtype=">i4").tobytes()]
    for p, U in zip(degrees, knots):
        out.append(np.array([p, len(U)], dtype=">i4").tobytes())
        out.append(np.asarray(U, dtype=">f8").tobytes())
    if X is not None:
        X = np.asarray(X, dtype=np.float64)
        nsd = X.shape[-1]
        w = np.ones(len(X)) if W is None else np.asarray(W, dtype=np.float64)
        xw = np.concatenate([X * w[:, None], w[:, None]], axis=1)
        out.append(np.array([nsd, VEC_FILE_CLASSID, xw.size], dtype=">i4").tobytes())
        out.append(xw.astype(">f8").tobytes())
    if A is not None:
        A = np.asarray(A, dtype=np.float64)
        out.append(np.array([A.shape[-1], VEC_FILE_CLASSID, A.size], dtype=">i4").tobytes())
        out.append(A.astype(">f8").tobytes())
    return b"".join(out)


def vec_file_bytes(v):
    v = np.asarray(v, dtype=np.float64)
    return np.array([VEC_FILE_CLASSID, v.size], dtype=">i4").tobytes() + v.astype(">f8").tobytes()
