"""ctypes binding of the CPU oracle (oracle/libigaoracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package petiga_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int)


class OrcBC(C.Structure):
    _fields_ = [("count", C.c_int), ("field", C.c_int * 64), ("value", C.c_double * 64)]


class OrcAxis(C.Structure):
    _fields_ = [("p", C.c_int), ("m", C.c_int), ("U", c_dp), ("periodic", C.c_int),
                ("nel", C.c_int), ("nnp", C.c_int), ("span", c_ip)]


class OrcBasis(C.Structure):
    _fields_ = [("nel", C.c_int), ("nqp", C.c_int), ("nen", C.c_int), ("offset", c_ip),
                ("detJac", c_dp), ("weight", c_dp), ("point", c_dp), ("value", c_dp),
                ("bnd_point", C.c_double * 2), ("bnd_weight", C.c_double), ("bnd_detJac", C.c_double),
                ("bnd_value", c_dp * 2)]


class OrcIGAStruct(C.Structure):
    _fields_ = [("dim", C.c_int), ("dof", C.c_int), ("order", C.c_int), ("nsd", C.c_int), ("rational", C.c_int),
                ("axis", OrcAxis * 3), ("rule_nqp", C.c_int * 3), ("basis", OrcBasis * 3),
                ("proc_sizes", C.c_int * 3), ("proc_ranks", C.c_int * 3),
                ("elem_sizes", C.c_int * 3), ("elem_start", C.c_int * 3), ("elem_width", C.c_int * 3),
                ("node_sizes", C.c_int * 3), ("node_lstart", C.c_int * 3), ("node_lwidth", C.c_int * 3),
                ("node_gstart", C.c_int * 3), ("node_gwidth", C.c_int * 3),
                ("geometryX", c_dp), ("rationalW", c_dp),
                ("value", (OrcBC * 2) * 3), ("load", (OrcBC * 2) * 3), ("visit", (C.c_int * 2) * 3),
                ("fixtable", C.c_int), ("fixtableU", c_dp), ("setup", C.c_int),
                ("rule_type", C.c_int * 3), ("rule_user_n", C.c_int * 3), ("rule_x", c_dp * 3), ("rule_w", c_dp * 3),
                ("property", C.c_int), ("propertyA", c_dp)]


class OrcMat(C.Structure):
    _fields_ = [("nrows", C.c_int64), ("rowptr", C.POINTER(C.c_int64)), ("colidx", C.POINTER(C.c_int32)), ("val", c_dp)]


class OrcElemView(C.Structure):
    _fields_ = [("nqp", C.c_int), ("nen", C.c_int), ("dim", C.c_int), ("nsd", C.c_int),
                ("weight", c_dp), ("detJac", c_dp), ("point", c_dp), ("normal", c_dp), ("detX", c_dp), ("detS", c_dp),
                ("basis", c_dp * 5), ("shape", c_dp * 5), ("mapU", c_dp * 5), ("mapX", c_dp * 5),
                ("mapping", c_ip), ("geometryX", c_dp), ("rationalW", c_dp)]


class ElasticityCtx(C.Structure):
    _fields_ = [("lambda_", C.c_double), ("mu", C.c_double)]


class CahnHilliardCtx(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("theta", "alpha", "cbar", "L0", "lambda_", "tau")]


class NSVMSCtx(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("nu", "fx", "fy", "fz", "dt")]


def build(force=False):
    so = os.path.join(_HERE, "libigaoracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("igaoracle.c", "igaforms.c", "igaoracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libigaoracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        P = C.POINTER(OrcIGAStruct)
        L.orc_create.restype = P
        L.orc_create.argtypes = [C.c_int, C.c_int]
        L.orc_destroy.argtypes = [P]
        L.orc_axis_set_degree.argtypes = [P, C.c_int, C.c_int]
        L.orc_axis_set_periodic.argtypes = [P, C.c_int, C.c_int]
        L.orc_axis_init_uniform.argtypes = [P, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
        L.orc_axis_set_knots.argtypes = [P, C.c_int, C.c_int, c_dp]
        L.orc_set_quadrature.argtypes = [P, C.c_int, C.c_int]
        L.orc_set_order.argtypes = [P, C.c_int]
        L.orc_set_partition.argtypes = [P, C.c_int, C.c_int]
        L.orc_setup.argtypes = [P]
        L.orc_set_geometry.argtypes = [P, C.c_int, c_dp, c_dp]
        L.orc_set_property.argtypes = [P, C.c_int, c_dp]
        L.orc_set_boundary_value.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_set_boundary_load.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_double]
        L.orc_set_boundary_form.argtypes = [P, C.c_int, C.c_int, C.c_int]
        L.orc_clear_boundary.argtypes = [P]
        L.orc_set_fixtable.argtypes = [P, c_dp]
        L.orc_gauss_legendre.argtypes = [C.c_int, c_dp, c_dp]
        L.orc_gauss_lobatto.argtypes = [C.c_int, c_dp, c_dp]
        L.orc_set_rule_type.argtypes = [P, C.c_int, C.c_int]
        L.orc_set_rule.argtypes = [P, C.c_int, C.c_int, c_dp, c_dp]
        L.orc_bspline_ders.restype = None
        L.orc_bspline_ders.argtypes = [C.c_int, C.c_double, C.c_int, C.c_int, c_dp, c_dp]
        L.orc_partition.argtypes = [C.c_int, C.c_int, C.c_int, c_ip, c_ip, c_ip]
        L.orc_distribute.restype = None
        L.orc_distribute.argtypes = [C.c_int, c_ip, c_ip, c_ip, c_ip, c_ip]
        L.orc_global_size.restype = C.c_int64
        L.orc_global_size.argtypes = [P]
        L.orc_mat_create.restype = C.POINTER(OrcMat)
        L.orc_mat_create.argtypes = [P]
        L.orc_mat_destroy.argtypes = [C.POINTER(OrcMat)]
        M = C.POINTER(OrcMat)
        V = C.c_void_p
        L.orc_compute_system.argtypes = [P, V, V, M, c_dp]
        L.orc_compute_matrix.argtypes = [P, V, V, M]
        L.orc_compute_vector.argtypes = [P, V, V, c_dp]
        L.orc_compute_function.argtypes = [P, V, V, c_dp, c_dp]
        L.orc_compute_jacobian.argtypes = [P, V, V, c_dp, M]
        L.orc_compute_ifunction.argtypes = [P, V, V, C.c_double, c_dp, C.c_double, c_dp, c_dp]
        L.orc_compute_ijacobian.argtypes = [P, V, V, C.c_double, c_dp, C.c_double, c_dp, M]
        L.orc_compute_scalar.argtypes = [P, c_dp, C.c_int, c_dp, V, V, C.c_int]
        L.orc_element_tabulate.argtypes = [P, c_ip, C.c_int, C.POINTER(OrcElemView)]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(c_dp) if a is not None else None


def _fn(name):
    return C.cast(getattr(lib(), name), C.c_void_p)


def gauss_legendre(q):
    X = np.zeros(q)
    W = np.zeros(q)
    rc = lib().orc_gauss_legendre(q, _dp(X), _dp(W))
    if rc:
        raise ValueError("rule size %d not implemented" % q)
    return X, W


def gauss_lobatto(q):
    X = np.zeros(q)
    W = np.zeros(q)
    if lib().orc_gauss_lobatto(q, _dp(X), _dp(W)):
        raise ValueError("rule size %d not implemented" % q)
    return X, W


def bspline_ders(k, u, p, d, U):
    U = np.ascontiguousarray(U, dtype=np.float64)
    B = np.zeros((p + 1, 5))
    lib().orc_bspline_ders(k, u, p, d, _dp(U), _dp(B))
    return B


def partition(size, rank, N):
    dim = len(N)
    Nc = (C.c_int * 3)(*(list(N) + [1] * (3 - dim)))
    n = (C.c_int * 3)(-1, -1, -1)
    i = (C.c_int * 3)(0, 0, 0)
    rc = lib().orc_partition(size, rank, dim, Nc, n, i)
    if rc:
        raise ValueError("bad partition")
    return list(n)[:dim], list(i)[:dim]


def distribute(sizes, ranks, N):
    dim = len(N)
    a = lambda v: (C.c_int * 3)(*(list(v) + [1] * (3 - dim)))
    n = (C.c_int * 3)()
    s = (C.c_int * 3)()
    lib().orc_distribute(dim, a(sizes), a(ranks), a(N), n, s)
    return list(n)[:dim], list(s)[:dim]


class Mat:
    """The oracle's global natural-order CSR; numpy views over C memory."""

    def __init__(self, ptr):
        self.ptr = ptr
        m = ptr.contents
        self.nrows = m.nrows
        self.rowptr = np.ctypeslib.as_array(m.rowptr, shape=(self.nrows + 1,))
        nnz = int(self.rowptr[-1])
        self.nnz = nnz
        self.colidx = np.ctypeslib.as_array(m.colidx, shape=(nnz,))
        self.val = np.ctypeslib.as_array(m.val, shape=(nnz,))

    def scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val.copy(), self.colidx.copy(), self.rowptr.copy()), shape=(self.nrows, self.nrows))

    def __del__(self):
        try:
            lib().orc_mat_destroy(self.ptr)
        except Exception:
            pass


class OracleIGA:
    """Mirrors the slice of the PetIGA user API the assembly path needs
    (IGACreate/SetDim/SetDof, IGAAxis*, IGASetUp, IGASetBoundaryValue,
    IGASetFixTable, IGACreateMat, IGACompute*)."""

    def __init__(self, dim, dof=1):
        self.L = lib()
        self.p = self.L.orc_create(dim, dof)
        self.dim, self.dof = dim, dof

    def __del__(self):
        try:
            self.L.orc_destroy(self.p)
        except Exception:
            pass

    def _ck(self, rc):
        if rc:
            raise RuntimeError("oracle error %d" % rc)

    # -- discretisation
    def axis_uniform(self, i, p, N, C_=-1, Ui=0.0, Uf=1.0, periodic=False):
        self._ck(self.L.orc_axis_set_degree(self.p, i, p))
        self._ck(self.L.orc_axis_set_periodic(self.p, i, int(periodic)))
        self._ck(self.L.orc_axis_init_uniform(self.p, i, N, Ui, Uf, C_))

    def axis_knots(self, i, p, U, periodic=False):
        U = np.ascontiguousarray(U, dtype=np.float64)
        self._ck(self.L.orc_axis_set_degree(self.p, i, p))
        self._ck(self.L.orc_axis_set_periodic(self.p, i, int(periodic)))
        self._ck(self.L.orc_axis_set_knots(self.p, i, len(U) - 1, _dp(U)))

    def set_quadrature(self, i, q):
        self._ck(self.L.orc_set_quadrature(self.p, i, q))

    def set_rule_type(self, i, kind):
        self._ck(self.L.orc_set_rule_type(self.p, i, dict(legendre=0, lobatto=1, reduced=2)[kind] if isinstance(kind, str) else kind))

    def set_rule(self, i, x, w):
        x, w = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(w, dtype=np.float64)
        self._ck(self.L.orc_set_rule(self.p, i, len(x), _dp(x), _dp(w)))

    def set_order(self, o):
        self._ck(self.L.orc_set_order(self.p, o))

    def set_partition(self, size, rank):
        self._ck(self.L.orc_set_partition(self.p, size, rank))

    def setup(self):
        self._ck(self.L.orc_setup(self.p))

    def set_geometry(self, X, W=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        W = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
        self._keep = (X, W)
        self._ck(self.L.orc_set_geometry(self.p, X.shape[-1], _dp(X), _dp(W)))

    def set_property(self, A):
        if A is None:
            self._ck(self.L.orc_set_property(self.p, 0, None)); return
        A = np.ascontiguousarray(A, dtype=np.float64)
        self._keep_prop = A
        self._ck(self.L.orc_set_property(self.p, A.shape[-1], _dp(A)))

    def set_boundary_value(self, axis, side, field, value):
        self._ck(self.L.orc_set_boundary_value(self.p, axis, side, field, value))

    def set_boundary_load(self, axis, side, field, value):
        self._ck(self.L.orc_set_boundary_load(self.p, axis, side, field, value))

    def set_boundary_form(self, axis, side, flag=True):
        self._ck(self.L.orc_set_boundary_form(self.p, axis, side, int(flag)))

    def clear_boundary(self):
        self.L.orc_clear_boundary(self.p)

    def set_fixtable(self, U):
        U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        self._ck(self.L.orc_set_fixtable(self.p, _dp(U)))

    # -- introspection
    @property
    def s(self):
        return self.p.contents

    def axis(self, i):
        ax = self.s.axis[i]
        return dict(p=ax.p, m=ax.m, U=np.ctypeslib.as_array(ax.U, shape=(ax.m + 1,)).copy(), nel=ax.nel, nnp=ax.nnp,
                    span=np.ctypeslib.as_array(ax.span, shape=(ax.nel,)).copy(), periodic=bool(ax.periodic))

    def basis(self, i):
        b = self.s.basis[i]
        g = lambda ptr, shape: np.ctypeslib.as_array(ptr, shape=shape).copy()
        return dict(nel=b.nel, nqp=b.nqp, nen=b.nen, offset=g(b.offset, (b.nel,)), detJac=g(b.detJac, (b.nel,)),
                    weight=g(b.weight, (b.nel, b.nqp)), point=g(b.point, (b.nel, b.nqp)),
                    value=g(b.value, (b.nel, b.nqp, b.nen, 5)))

    def ranges(self):
        s = self.s
        f = lambda a: list(a)[:self.dim]
        return dict(proc_sizes=f(s.proc_sizes), proc_ranks=f(s.proc_ranks), elem_sizes=f(s.elem_sizes),
                    elem_start=f(s.elem_start), elem_width=f(s.elem_width), node_sizes=f(s.node_sizes),
                    node_lstart=f(s.node_lstart), node_lwidth=f(s.node_lwidth),
                    node_gstart=f(s.node_gstart), node_gwidth=f(s.node_gwidth))

    def global_size(self):
        return int(self.L.orc_global_size(self.p))

    def create_mat(self):
        return Mat(self.L.orc_mat_create(self.p))

    def element(self, ID, boundary_id=-1):
        """Tabulate one element; returns numpy copies of every element array."""
        v = OrcElemView()
        ids = (C.c_int * 3)(*(list(ID) + [0] * (3 - len(ID))))
        self._ck(self.L.orc_element_tabulate(self.p, ids, boundary_id, C.byref(v)))
        nqp, nen, dim, nsd = v.nqp, v.nen, v.dim, v.nsd
        g = lambda ptr, shape: np.ctypeslib.as_array(ptr, shape=shape).copy()
        out = dict(nqp=nqp, nen=nen, dim=dim, nsd=nsd,
                   weight=g(v.weight, (nqp,)), detJac=g(v.detJac, (nqp,)), point=g(v.point, (nqp, dim)),
                   normal=g(v.normal, (nqp, nsd)), detX=g(v.detX, (nqp,)), detS=g(v.detS, (nqp,)),
                   mapping=g(v.mapping, (nen,)), geometryX=g(v.geometryX, (nen, nsd)), rationalW=g(v.rationalW, (nen,)))
        for k in range(5):
            out["basis%d" % k] = g(v.basis[k], (nqp, nen) + (dim,) * k)
            out["shape%d" % k] = g(v.shape[k], (nqp, nen) + (nsd,) * k)
            out["mapX%d" % k] = g(v.mapX[k], (nqp, nsd) + (dim,) * k)
            out["mapU%d" % k] = g(v.mapU[k], (nqp, dim) + (nsd,) * k)
        return out

    # -- assembly
    def compute_system(self, form, ctx=None, A=None):
        A = A or self.create_mat()
        B = np.zeros(self.global_size())
        self._ck(self.L.orc_compute_system(self.p, _fn(form), C.cast(C.byref(ctx), C.c_void_p) if ctx is not None else None, A.ptr, _dp(B)))
        return A, B

    def compute_function(self, form, ctx, U):
        U = np.ascontiguousarray(U, dtype=np.float64)
        F = np.zeros(self.global_size())
        self._ck(self.L.orc_compute_function(self.p, _fn(form), C.cast(C.byref(ctx), C.c_void_p), _dp(U), _dp(F)))
        return F

    def compute_jacobian(self, form, ctx, U, A=None):
        U = np.ascontiguousarray(U, dtype=np.float64)
        A = A or self.create_mat()
        self._ck(self.L.orc_compute_jacobian(self.p, _fn(form), C.cast(C.byref(ctx), C.c_void_p), _dp(U), A.ptr))
        return A

    def compute_ifunction(self, form, ctx, a, V, t, U):
        V = np.ascontiguousarray(V, dtype=np.float64)
        U = np.ascontiguousarray(U, dtype=np.float64)
        F = np.zeros(self.global_size())
        self._ck(self.L.orc_compute_ifunction(self.p, _fn(form), C.cast(C.byref(ctx), C.c_void_p), a, _dp(V), t, _dp(U), _dp(F)))
        return F

    def compute_ijacobian(self, form, ctx, a, V, t, U, A=None):
        V = np.ascontiguousarray(V, dtype=np.float64)
        U = np.ascontiguousarray(U, dtype=np.float64)
        A = A or self.create_mat()
        self._ck(self.L.orc_compute_ijacobian(self.p, _fn(form), C.cast(C.byref(ctx), C.c_void_p), a, _dp(V), t, _dp(U), A.ptr))
        return A

    def compute_scalar(self, scalar, n, U=None, ctx=None, full=False):
        U = None if U is None else np.ascontiguousarray(U, dtype=np.float64)
        S = np.zeros(n)
        cp = None
        if ctx is not None:
            cp = C.cast(C.byref(ctx), C.c_void_p)
        self._ck(self.L.orc_compute_scalar(self.p, _dp(U), n, _dp(S), _fn(scalar), cp, int(full)))
        return S
