"""petiga_amd -- MI355X-native IGA element-assembly engine (one hot path of dalcinl/PetIGA).

This package is a thin ctypes view of the C ABI in include/petiga_amd.h; all compute is in
libpetiga_amd.so (hand-written HIP for gfx950).  There is no CPU or PyTorch fallback: if the
library is missing or no GPU is visible, the compute calls fail loudly.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SCALARS = dict(volume=1, x2err=2, errnorm=3)
FORMS = dict(none=0, poisson=1, mass=2, l2proj_x2=3, poisson_f=4, errnorm=5, elasticity=6, cahnhilliard=7, nsvms=8, boundary_integral=9, nitsche=10, bratu=11, elasticity_f=12, der3=13, property=14, surface=15)
RULE_TYPES = dict(legendre=0, lobatto=1, reduced=2, user=3)      # IGARuleType, include/petiga.h:82-87

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class IGXAxisTables(C.Structure):
    _fields_ = [("p", C.c_int), ("m", C.c_int), ("periodic", C.c_int), ("nel", C.c_int), ("nnp", C.c_int),
                ("U", _dp), ("span", _ip), ("nqp", C.c_int), ("nen", C.c_int), ("offset", _ip),
                ("detJac", _dp), ("weight", _dp), ("point", _dp), ("value", _dp)]


class IGXTables(C.Structure):
    _fields_ = [("dim", C.c_int), ("dof", C.c_int), ("order", C.c_int), ("axis", IGXAxisTables * 3),
                ("proc_sizes", C.c_int * 3), ("proc_ranks", C.c_int * 3),
                ("elem_sizes", C.c_int * 3), ("elem_start", C.c_int * 3), ("elem_width", C.c_int * 3),
                ("node_sizes", C.c_int * 3), ("node_lstart", C.c_int * 3), ("node_lwidth", C.c_int * 3),
                ("node_gstart", C.c_int * 3), ("node_gwidth", C.c_int * 3),
                ("nsd", C.c_int), ("rational", C.c_int), ("geometryX", _dp), ("rationalW", _dp),
                ("property", C.c_int), ("propertyA", _dp)]


# IGXTransportFn (include/petiga_amd.h): the host-callback transport of the ghost-row exchange
TRANSPORT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, _ip, C.POINTER(C.c_void_p), C.POINTER(C.c_int64))


class IGXError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("IGX error %d: %s" % (code, msg))
        self.code = code


def lib(build_if_needed=False):
    """Load libpetiga_amd.so.  torch (if importable) is imported first so both share one HIP runtime."""
    global _LIB
    if _LIB is not None:
        return _LIB
    so = os.path.join(_HERE, "libpetiga_amd_debug.so" if os.environ.get("IGX_USE_DEBUG_LIB") else "libpetiga_amd.so")   # the -DIGX_DEBUG experiment build
    named = bool(os.environ.get("IGX_LIB"))
    if named:      # another build of this same library (A/B measurements, scripts/headline_ab.py): it is the one that is loaded, never rebuilt
        so = os.path.abspath(os.environ["IGX_LIB"])
    elif build_if_needed:
        so = _build.build()
    if not os.path.exists(so):
        raise ImportError("libpetiga_amd.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    try:
        import torch  # noqa: F401  (same libamdhip64 for both)
    except Exception:
        pass
    L = C.CDLL(so)
    V = C.c_void_p
    L.IGXGetLastError.restype = C.c_char_p
    L.IGXGetElementCount.restype = C.c_int64
    L.IGXGetElementCount.argtypes = [V]
    sig = {
        "IGXCreate": [C.POINTER(V)], "IGXDestroy": [C.POINTER(V)], "IGXSetDim": [V, C.c_int], "IGXSetDof": [V, C.c_int],
        "IGXSetOrder": [V, C.c_int], "IGXSetQuadrature": [V, C.c_int, C.c_int], "IGXSetRuleType": [V, C.c_int, C.c_int], "IGXSetRuleSize": [V, C.c_int, C.c_int],
        "IGXSetRule": [V, C.c_int, C.c_int, _dp, _dp], "IGXGetRule": [V, C.c_int, _ip, _dp, _dp],
        "IGXGetBasis": [V, C.c_int, _ip, _ip, _ip, _ip, _dp, _dp, _dp, _dp], "IGXSetProcessors": [V, C.c_int, C.c_int],
        "IGXSetComm": [V, C.c_int, C.c_int], "IGXAxisSetDegree": [V, C.c_int, C.c_int], "IGXAxisSetPeriodic": [V, C.c_int, C.c_int],
        "IGXAxisInitUniform": [V, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int], "IGXAxisSetKnots": [V, C.c_int, C.c_int, _dp],
        "IGXSetUp": [V], "IGXSetGeometry": [V, C.c_int, _dp, _dp], "IGXSetProperty": [V, C.c_int, _dp], "IGXGetPropertyDim": [V, C.POINTER(C.c_int)],
        "IGXComputeScalar": [V, V, C.c_int, _dp, C.c_int, C.c_int, _dp],
        "IGXComputeScalarSource": [V, V, C.c_char_p, C.c_char_p, _dp, C.c_int, C.c_int, _dp],
        "IGXRead": [V, C.c_char_p], "IGXWrite": [V, C.c_char_p], "IGXWriteVec": [V, V, C.c_char_p], "IGXReadVec": [V, V, C.c_char_p],
        "IGXSetBoundaryValue": [V, C.c_int, C.c_int, C.c_int, C.c_double], "IGXSetBoundaryLoad": [V, C.c_int, C.c_int, C.c_int, C.c_double],
        "IGXClearBoundary": [V], "IGXSetBoundaryForm": [V, C.c_int, C.c_int, C.c_int], "IGXSetFixTable": [V, V], "IGXSetForm": [V, C.c_int, _dp, C.c_int],
        "IGXGetSizes": [V] + [_ip] * 8, "IGXGetProcessors": [V, _ip, _ip],
        "IGXCreateMat": [V, C.POINTER(V)], "IGXMatDestroy": [C.POINTER(V)],
        "IGXMatGetInfo": [V, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _ip],
        "IGXMatGetDeviceArrays": [V, C.POINTER(V), C.POINTER(V), C.POINTER(V)],
        "IGXMatCopyToHost": [V, V, V, V], "IGXMatGetLayout": [V, _ip, _ip], "IGXMatGetAxisMaps": [V, C.c_int, _ip, _ip],
        "IGXCreateVec": [V, C.POINTER(V)], "IGXVecDestroy": [C.POINTER(V)], "IGXVecGetSize": [V, C.POINTER(C.c_int64)],
        "IGXVecGetDeviceArray": [V, C.POINTER(V)], "IGXVecCopyToHost": [V, _dp], "IGXVecCopyFromHost": [V, _dp],
        "IGXComputeSystem": [V, V, V], "IGXComputeMatrix": [V, V], "IGXComputeVector": [V, V],
        "IGXComputeFunction": [V, V, V], "IGXComputeJacobian": [V, V, V],
        "IGXComputeIFunction": [V, C.c_double, V, C.c_double, V, V], "IGXComputeIJacobian": [V, C.c_double, V, C.c_double, V, V],
        "IGXComputeIFunctionIJacobian": [V, C.c_double, V, C.c_double, V, V, V], "IGXComputeFunctionJacobian": [V, V, V, V],
        "IGXSetStream": [V, V], "IGXSynchronize": [V], "IGXSetKernel": [V, C.c_int], "IGXGetKernelName": [V, C.c_char_p, C.c_int],
        "IGXSetTiming": [V, C.c_int], "IGXGetLastTiming": [V, _dp, _dp, _ip],
        "IGXGetDominantKernelTiming": [V, C.c_char_p, C.c_int, _dp, _ip, C.POINTER(C.c_int64), _dp],
        "IGXGetColoring": [V, _ip], "IGXGetElementColor": [V, C.c_int, C.c_int],
        "IGXGetNeighborCount": [V, _ip, _ip], "IGXGetNeighborInfo": [V, C.c_int, C.c_int, _ip, C.POINTER(C.c_int64), C.POINTER(C.c_int64)],
        "IGXPackGhostRows": [V, V, V, C.c_int, V], "IGXUnpackGhostRows": [V, V, V, C.c_int, V], "IGXRowOwned": [V, C.c_int, C.c_int, C.c_int],
        "IGXPackOwnerValues": [V, V, C.c_int, V], "IGXUnpackGhostValues": [V, V, C.c_int, V],
        "IGXChecksum": [V, V, V, _dp], "IGXCommGetOverlap": [V, _dp], "IGXCheckFormSource": [V, C.c_int, C.c_int], "IGXGetClockProbe": [V, _dp, C.POINTER(C.c_int64)], "IGXSetFormSource": [V, C.c_char_p, C.c_char_p, _dp, C.c_int],
        "IGXMatGetCOO": [V, C.c_int, C.c_int, V, V, C.c_int], "IGXMatGetCOODevice": [V, C.c_int, C.c_int, C.c_int, C.POINTER(V), C.POINTER(V)], "IGXMatFreeCOO": [V],
        "IGXVecGetIndices": [V, C.c_int, C.c_int, V, C.c_int],
        "IGXVecGetGhostedSize": [V, C.POINTER(C.c_int64)], "IGXVecCopyFromGhosted": [V, V, C.c_int], "IGXVecCopyToGhosted": [V, V, C.c_int],
        "IGXCommGetUniqueId": [C.c_void_p, C.c_char_p], "IGXCommInitRCCL": [V, C.c_void_p, C.c_char_p], "IGXCommInitTransport": [V, TRANSPORT_FN, C.c_void_p],
        "IGXCommDestroy": [V], "IGXReduceGhostRows": [V, V, V], "IGXRefreshGhosts": [V, V], "IGXCommGetLastBytes": [V, C.POINTER(C.c_int64)],
        "IGXCommLoopbackTest": [V, C.c_int64, _dp], "IGXCommGetRanks": [V, C.POINTER(C.c_int), C.POINTER(C.c_int)], "IGXCommGetEarlyPhases": [V, C.POINTER(C.c_int)],
        "IGXCommGetLinkRate": [V, _dp, _ip, _dp, _ip], "IGXGetFacePasses": [V, _ip],
        "IGXGetDeviceInfo": [C.c_char_p, C.c_int], "IGXCreateFromTables": [V, C.POINTER(V)],
    }
    missing = []
    for name, args in sig.items():
        if named and not hasattr(L, name):      # an older build of the library under A/B: it lacks newer entry points
            missing.append(name)
            continue
        f = getattr(L, name)
        f.argtypes = args
        f.restype = C.c_int
    if named:      # say which library this is and what it cannot do, here, not as an AttributeError far from the cause
        import sys
        print("petiga_amd: loaded IGX_LIB=%s%s" % (so, (" -- it lacks %d entry point(s) of include/petiga_amd.h: %s" % (len(missing), ", ".join(missing))) if missing else ""), file=sys.stderr)
    _LIB = L
    return L


def _ck(rc):
    if rc:
        raise IGXError(rc, lib().IGXGetLastError().decode())


class Vec:
    def __init__(self, iga):
        self.iga = iga
        self.h = C.c_void_p()
        _ck(lib().IGXCreateVec(iga.h, C.byref(self.h)))
        n = C.c_int64()
        _ck(lib().IGXVecGetSize(self.h, C.byref(n)))
        self.n = n.value

    def __del__(self):
        try:
            lib().IGXVecDestroy(C.byref(self.h))
        except Exception:
            pass

    def set(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n
        _ck(lib().IGXVecCopyFromHost(self.h, a.ctypes.data_as(_dp)))
        return self

    def get(self):
        a = np.empty(self.n)
        _ck(lib().IGXVecCopyToHost(self.h, a.ctypes.data_as(_dp)))
        return a

    def device_ptr(self):
        p = C.c_void_p()
        _ck(lib().IGXVecGetDeviceArray(self.h, C.byref(p)))
        return p.value

    def indices(self, numbering=0, owned_only=False):
        """Global index of every entry (natural or PETSc numbering), -1 for not-owned rows when owned_only."""
        idx = np.empty(self.n, dtype=np.int64)
        _ck(lib().IGXVecGetIndices(self.h, numbering, int(owned_only), idx.ctypes.data, 0))
        return idx

    def ghosted_size(self):
        n = C.c_int64()
        _ck(lib().IGXVecGetGhostedSize(self.h, C.byref(n)))
        return n.value

    def set_from_ghosted(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.ghosted_size()
        _ck(lib().IGXVecCopyFromGhosted(self.h, a.ctypes.data, 0))
        return self

    def get_ghosted(self):
        a = np.empty(self.ghosted_size())
        _ck(lib().IGXVecCopyToGhosted(self.h, a.ctypes.data, 0))
        return a


class Mat:
    def __init__(self, iga):
        self.iga = iga
        self.h = C.c_void_p()
        _ck(lib().IGXCreateMat(iga.h, C.byref(self.h)))
        a, b, c = C.c_int64(), C.c_int64(), C.c_int()
        _ck(lib().IGXMatGetInfo(self.h, C.byref(a), C.byref(b), C.byref(c)))
        self.nbrows, self.nblocks, self.bs = a.value, b.value, c.value

    def __del__(self):
        try:
            lib().IGXMatDestroy(C.byref(self.h))
        except Exception:
            pass

    def device_ptrs(self):
        a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _ck(lib().IGXMatGetDeviceArrays(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def host(self, values_only=False):
        val = np.empty(self.nblocks * self.bs * self.bs)
        if values_only:
            _ck(lib().IGXMatCopyToHost(self.h, None, None, val.ctypes.data))
            return val
        rp = np.empty(self.nbrows + 1, dtype=np.int64)
        ci = np.empty(self.nblocks, dtype=np.int32)
        _ck(lib().IGXMatCopyToHost(self.h, rp.ctypes.data, ci.ctypes.data, val.ctypes.data))
        return rp, ci, val

    def coo(self, numbering=0, owned_only=False):
        """(coo_i, coo_j) of every stored scalar in the order of the value array (IGXMatGetCOO)."""
        n = self.nblocks * self.bs * self.bs
        ci, cj = np.empty(n, dtype=np.int64), np.empty(n, dtype=np.int64)
        _ck(lib().IGXMatGetCOO(self.h, numbering, int(owned_only), ci.ctypes.data, cj.ctypes.data, 0))
        return ci, cj

    def coo_device(self, numbering=0, owned_only=False, index_bytes=8):
        """Device pointers of the coordinate lists the matrix keeps for the hand-back (IGXMatGetCOODevice); free_coo() drops them."""
        pi, pj = C.c_void_p(), C.c_void_p()
        _ck(lib().IGXMatGetCOODevice(self.h, numbering, int(owned_only), index_bytes, C.byref(pi), C.byref(pj)))
        return pi.value, pj.value

    def free_coo(self): _ck(lib().IGXMatFreeCOO(self.h))

    def layout(self):
        nrow, ncol = (C.c_int * 3)(), (C.c_int * 3)()
        _ck(lib().IGXMatGetLayout(self.h, nrow, ncol))
        maps = []
        for d in range(3):
            r = np.empty(nrow[d], dtype=np.int32)
            c = np.empty(ncol[d], dtype=np.int32)
            _ck(lib().IGXMatGetAxisMaps(self.h, d, r.ctypes.data_as(_ip), c.ctypes.data_as(_ip)))
            maps.append((r, c))
        return list(nrow), list(ncol), maps

    def to_coo_global(self):
        """(rows, cols, vals) in global natural numbering (node*dof+field), one entry per stored scalar."""
        rp, ci, val = self.host()
        nrow, ncol, maps = self.layout()
        ns = self.iga.sizes()["node_sizes"]
        bs = self.bs
        r = np.arange(self.nbrows, dtype=np.int64)
        r0, r1, r2 = r % nrow[0], (r // nrow[0]) % nrow[1], r // (nrow[0] * nrow[1])
        grow = maps[0][0][r0].astype(np.int64) + ns[0] * (maps[1][0][r1].astype(np.int64) + ns[1] * maps[2][0][r2].astype(np.int64))
        c = ci.astype(np.int64)
        c0, c1, c2 = c % ncol[0], (c // ncol[0]) % ncol[1], c // (ncol[0] * ncol[1])
        gcol = maps[0][1][c0].astype(np.int64) + ns[0] * (maps[1][1][c1].astype(np.int64) + ns[1] * maps[2][1][c2].astype(np.int64))
        brow = np.repeat(grow, np.diff(rp))
        ii, jj = np.meshgrid(np.arange(bs), np.arange(bs), indexing="ij")
        rows = (brow[:, None, None] * bs + ii[None]).reshape(-1)
        cols = (gcol[:, None, None] * bs + jj[None]).reshape(-1)
        return rows, cols, val.copy()

    def to_scipy_global(self):
        """Global natural-order CSR (node*dof+field) -- the Mat PETSc would hold on one rank."""
        import scipy.sparse as sp
        rows, cols, vals = self.to_coo_global()
        n = int(np.prod(self.iga.sizes()["node_sizes"])) * self.bs
        return sp.coo_matrix((vals, (rows, cols)), shape=(n, n)).tocsr()


class IGX:
    """Mirror of the PetIGA calls a driver program makes around IGAComputeSystem & friends."""

    def __init__(self, dim=None, dof=None, _handle=None):
        self.h = C.c_void_p()
        if _handle is not None:
            self.h = _handle
            return
        _ck(lib().IGXCreate(C.byref(self.h)))
        if dim is not None:
            self.set_dim(dim)
        if dof is not None:
            self.set_dof(dof)

    def __del__(self):
        try:
            lib().IGXDestroy(C.byref(self.h))
        except Exception:
            pass

    @classmethod
    def from_tables(cls, tables):
        """IGXCreateFromTables: the route a set-up PetIGA `IGA` takes."""
        h = C.c_void_p()
        _ck(lib().IGXCreateFromTables(C.byref(tables), C.byref(h)))
        g = cls(_handle=h)
        g.dim, g.dof = tables.dim, tables.dof
        return g

    def set_dim(self, dim): _ck(lib().IGXSetDim(self.h, dim)); self.dim = dim
    def set_dof(self, dof): _ck(lib().IGXSetDof(self.h, dof)); self.dof = dof
    def set_order(self, o): _ck(lib().IGXSetOrder(self.h, o))
    def set_quadrature(self, i, q): _ck(lib().IGXSetQuadrature(self.h, i, q))
    def set_rule_type(self, i, kind): _ck(lib().IGXSetRuleType(self.h, i, RULE_TYPES[kind] if isinstance(kind, str) else kind))
    def set_rule_size(self, i, q): _ck(lib().IGXSetRuleSize(self.h, i, q))

    def set_rule(self, i, x, w):
        x, w = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(w, dtype=np.float64)
        assert x.shape == w.shape and x.ndim == 1
        _ck(lib().IGXSetRule(self.h, i, len(x), x.ctypes.data_as(_dp), w.ctypes.data_as(_dp)))

    def rule(self, i):
        q = C.c_int(0)
        _ck(lib().IGXGetRule(self.h, i, C.byref(q), None, None))
        x, w = np.zeros(q.value), np.zeros(q.value)
        _ck(lib().IGXGetRule(self.h, i, C.byref(q), x.ctypes.data_as(_dp), w.ctypes.data_as(_dp)))
        return x, w
    def set_comm(self, size, rank):
        _ck(lib().IGXSetComm(self.h, size, rank))
        self._comm = (size, rank)

    def comm_size(self): return getattr(self, "_comm", (1, 0))[0]
    def set_processors(self, i, n): _ck(lib().IGXSetProcessors(self.h, i, n))

    def axis_uniform(self, i, p, N, C_=-1, Ui=0.0, Uf=1.0, periodic=False):
        _ck(lib().IGXAxisSetDegree(self.h, i, p))
        _ck(lib().IGXAxisSetPeriodic(self.h, i, int(periodic)))
        _ck(lib().IGXAxisInitUniform(self.h, i, N, Ui, Uf, C_))

    def axis_knots(self, i, p, U, periodic=False):
        U = np.ascontiguousarray(U, dtype=np.float64)
        _ck(lib().IGXAxisSetDegree(self.h, i, p))
        _ck(lib().IGXAxisSetPeriodic(self.h, i, int(periodic)))
        _ck(lib().IGXAxisSetKnots(self.h, i, len(U) - 1, U.ctypes.data_as(_dp)))

    def setup(self): _ck(lib().IGXSetUp(self.h))
    def read(self, filename): _ck(lib().IGXRead(self.h, str(filename).encode()))
    def write(self, filename): _ck(lib().IGXWrite(self.h, str(filename).encode()))
    def write_vec(self, vec, filename): _ck(lib().IGXWriteVec(self.h, vec.h, str(filename).encode()))
    def read_vec(self, vec, filename): _ck(lib().IGXReadVec(self.h, vec.h, str(filename).encode()))

    def set_geometry(self, X, W=None):
        X = np.ascontiguousarray(X, dtype=np.float64)
        Wp = None if W is None else np.ascontiguousarray(W, dtype=np.float64)
        _ck(lib().IGXSetGeometry(self.h, X.shape[-1], X.ctypes.data_as(_dp), None if Wp is None else Wp.ctypes.data_as(_dp)))

    def set_property(self, A):
        """IGASetPropertyDim + the property array on the geometry grid, natural order [..][npd] (None drops it)."""
        if A is None:
            _ck(lib().IGXSetProperty(self.h, 0, None)); return
        A = np.ascontiguousarray(A, dtype=np.float64)
        _ck(lib().IGXSetProperty(self.h, A.shape[-1], A.ctypes.data_as(_dp)))

    def property_dim(self):
        n = C.c_int(0); _ck(lib().IGXGetPropertyDim(self.h, C.byref(n))); return n.value

    def set_boundary_value(self, axis, side, field, value): _ck(lib().IGXSetBoundaryValue(self.h, axis, side, field, value))
    def set_boundary_load(self, axis, side, field, value): _ck(lib().IGXSetBoundaryLoad(self.h, axis, side, field, value))
    def clear_boundary(self): _ck(lib().IGXClearBoundary(self.h))
    def set_boundary_form(self, axis, side, flag=True): _ck(lib().IGXSetBoundaryForm(self.h, axis, side, int(bool(flag))))
    def set_fixtable(self, vec): _ck(lib().IGXSetFixTable(self.h, vec.h if vec is not None else None))

    def set_form(self, kind, params=()):
        p = np.ascontiguousarray(params, dtype=np.float64)
        _ck(lib().IGXSetForm(self.h, FORMS[kind] if isinstance(kind, str) else kind, p.ctypes.data_as(_dp) if p.size else None, p.size))

    def set_form_source(self, source, struct_name, params=()):
        """A user point form as HIP source (the contract of petiga_amd/csrc/forms.hpp), compiled at run time with hiprtc."""
        p = np.ascontiguousarray(params, dtype=np.float64)
        _ck(lib().IGXSetFormSource(self.h, source.encode(), struct_name.encode(), p.ctypes.data_as(_dp) if p.size else None, p.size))

    def sizes(self):
        arrs = [(C.c_int * 3)() for _ in range(8)]
        _ck(lib().IGXGetSizes(self.h, *arrs))
        names = ["elem_sizes", "elem_start", "elem_width", "node_sizes", "node_lstart", "node_lwidth", "node_gstart", "node_gwidth"]
        out = {k: list(a) for k, a in zip(names, arrs)}
        ps, pr = (C.c_int * 3)(), (C.c_int * 3)()
        _ck(lib().IGXGetProcessors(self.h, ps, pr))
        out["proc_sizes"], out["proc_ranks"] = list(ps), list(pr)
        return out

    def basis(self, i):
        """The 1-D tables of axis i (IGAGetBasis): offset, detJac, weight, point, value[nel][nqp][nen][5]."""
        n = [C.c_int(0) for _ in range(3)]
        _ck(lib().IGXGetBasis(self.h, i, C.byref(n[0]), C.byref(n[1]), C.byref(n[2]), None, None, None, None, None))
        nel, nqp, nen = (v.value for v in n)
        off = np.zeros(nel, dtype=np.int32)
        J, w, pt, val = np.zeros(nel), np.zeros((nel, nqp)), np.zeros((nel, nqp)), np.zeros((nel, nqp, nen, 5))
        _ck(lib().IGXGetBasis(self.h, i, None, None, None, off.ctypes.data_as(_ip), J.ctypes.data_as(_dp), w.ctypes.data_as(_dp), pt.ctypes.data_as(_dp), val.ctypes.data_as(_dp)))
        return dict(nel=nel, nqp=nqp, nen=nen, offset=off, detJac=J, weight=w, point=pt, value=val)

    def compute_scalar(self, kind, U=None, params=()):
        """IGAComputeScalar for one of the built-in functionals; returns the rank-local sums."""
        k = SCALARS[kind] if isinstance(kind, str) else kind
        n = 4 if k == SCALARS["errnorm"] else (2 if k == SCALARS["volume"] else 1)
        out = np.zeros(n)
        p = np.ascontiguousarray(params, dtype=np.float64)
        _ck(lib().IGXComputeScalar(self.h, U.h if U is not None else None, k, p.ctypes.data_as(_dp) if p.size else None, p.size, n, out.ctypes.data_as(_dp)))
        return out

    def compute_scalar_source(self, source, struct_name, n, U=None, params=()):
        """IGAComputeScalar with the user's point functional as HIP source (a struct with NSCALAR = n and scalar(p, S))."""
        out = np.zeros(n)
        p = np.ascontiguousarray(params, dtype=np.float64)
        _ck(lib().IGXComputeScalarSource(self.h, U.h if U is not None else None, source.encode(), struct_name.encode(),
                                         p.ctypes.data_as(_dp) if p.size else None, p.size, n, out.ctypes.data_as(_dp)))
        return out

    def element_count(self): return lib().IGXGetElementCount(self.h)
    def create_mat(self): return Mat(self)
    def create_vec(self): return Vec(self)

    def compute_system(self, A, b): _ck(lib().IGXComputeSystem(self.h, A.h, b.h))
    def compute_matrix(self, A): _ck(lib().IGXComputeMatrix(self.h, A.h))
    def compute_vector(self, b): _ck(lib().IGXComputeVector(self.h, b.h))
    def compute_function(self, U, F): _ck(lib().IGXComputeFunction(self.h, U.h, F.h))
    def compute_jacobian(self, U, J): _ck(lib().IGXComputeJacobian(self.h, U.h, J.h))
    def compute_ifunction(self, a, V, t, U, F): _ck(lib().IGXComputeIFunction(self.h, a, V.h, t, U.h, F.h))
    def compute_ijacobian(self, a, V, t, U, J): _ck(lib().IGXComputeIJacobian(self.h, a, V.h, t, U.h, J.h))

    def compute_ifunction_ijacobian(self, a, V, t, U, F, J):
        """F and J of one Newton step in one pass of the walk where a fused kernel exists (IGXComputeIFunctionIJacobian)."""
        _ck(lib().IGXComputeIFunctionIJacobian(self.h, a, V.h, t, U.h, F.h, J.h))

    def compute_function_jacobian(self, U, F, J): _ck(lib().IGXComputeFunctionJacobian(self.h, U.h, F.h, J.h))

    def set_stream(self, stream): _ck(lib().IGXSetStream(self.h, stream))
    def synchronize(self): _ck(lib().IGXSynchronize(self.h))
    def set_kernel(self, which): _ck(lib().IGXSetKernel(self.h, which))
    def set_timing(self, flag=True): _ck(lib().IGXSetTiming(self.h, int(flag)))

    def kernel_name(self):
        buf = C.create_string_buffer(1024)
        _ck(lib().IGXGetKernelName(self.h, buf, 1024))
        return buf.value.decode()

    def last_timing(self):
        a, b, n = C.c_double(), C.c_double(), C.c_int()
        _ck(lib().IGXGetLastTiming(self.h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def dominant_kernel(self):
        buf = C.create_string_buffer(128)
        ms, n, el, fl = C.c_double(), C.c_int(), C.c_int64(), C.c_double()
        _ck(lib().IGXGetDominantKernelTiming(self.h, buf, 128, C.byref(ms), C.byref(n), C.byref(el), C.byref(fl)))
        return dict(name=buf.value.decode(), ms=ms.value, launches=n.value, elements=el.value, executed_flop_per_element=fl.value)

    def neighbors(self, send):
        """[(peer rank, matrix doubles, vector doubles)] of the send (upper) or receive (lower) list."""
        ns, nr = C.c_int(), C.c_int()
        _ck(lib().IGXGetNeighborCount(self.h, C.byref(ns), C.byref(nr)))
        out = []
        for k in range(ns.value if send else nr.value):
            r, m, v = C.c_int(), C.c_int64(), C.c_int64()
            _ck(lib().IGXGetNeighborInfo(self.h, int(send), k, C.byref(r), C.byref(m), C.byref(v)))
            out.append((r.value, m.value, v.value))
        return out

    # -- the exchange inside the library (RCCL, or a host-callback transport)
    @staticmethod
    def comm_unique_id(librccl=None):
        buf = C.create_string_buffer(128)
        _ck(lib().IGXCommGetUniqueId(buf, librccl.encode() if librccl else None))
        return buf.raw

    def comm_init_rccl(self, unique_id, librccl=None):
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _ck(lib().IGXCommInitRCCL(self.h, buf, librccl.encode() if librccl else None))

    def comm_init_transport(self, pyfn):
        """pyfn(send=[(peer, devptr, count)], recv=[(peer, devptr, count)]) moves the packed device buffers."""
        def tramp(ctx, ns, sp, sb, sn, nr, rp, rb, rn):
            try:
                pyfn([(sp[i], sb[i], sn[i]) for i in range(ns)], [(rp[i], rb[i], rn[i]) for i in range(nr)])
                return 0
            except Exception as e:      # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._transport = TRANSPORT_FN(tramp)     # keep the trampoline alive
        _ck(lib().IGXCommInitTransport(self.h, self._transport, None))

    def comm_destroy(self): _ck(lib().IGXCommDestroy(self.h))
    def reduce_ghost_rows(self, A=None, b=None): _ck(lib().IGXReduceGhostRows(self.h, A.h if A is not None else None, b.h if b is not None else None))
    def refresh_ghosts(self, v): _ck(lib().IGXRefreshGhosts(self.h, v.h))

    def comm_last_bytes(self):
        n = C.c_int64()
        _ck(lib().IGXCommGetLastBytes(self.h, C.byref(n)))
        return n.value

    def comm_ranks(self):
        """(transport kind: "rccl" / "host" / None, ranks the transport itself reports: ncclCommCount for RCCL)."""
        k, n = C.c_int(0), C.c_int(0)
        _ck(lib().IGXCommGetRanks(self.h, C.byref(k), C.byref(n)))
        return {0: None, 1: "rccl", 2: "host"}[k.value], n.value

    def comm_loopback_test(self, n=1 << 20):
        d = C.c_double()
        _ck(lib().IGXCommLoopbackTest(self.h, n, C.byref(d)))
        return d.value

    def pack_ghost_rows(self, A, b, k, devptr): _ck(lib().IGXPackGhostRows(self.h, A.h if A else None, b.h if b else None, k, devptr))
    def unpack_ghost_rows(self, A, b, k, devptr): _ck(lib().IGXUnpackGhostRows(self.h, A.h if A else None, b.h if b else None, k, devptr))
    def pack_owner_values(self, v, k, devptr): _ck(lib().IGXPackOwnerValues(self.h, v.h, k, devptr))
    def unpack_ghost_values(self, v, k, devptr): _ck(lib().IGXUnpackGhostValues(self.h, v.h, k, devptr))
    def row_owned(self, r0, r1=0, r2=0): return bool(lib().IGXRowOwned(self.h, r0, r1, r2))

    def checksum(self, A=None, b=None):
        """[sum A, sum |A|, sum b, sum b^2] over the rows this rank owns."""
        out = np.zeros(4)
        _ck(lib().IGXChecksum(self.h, A.h if A is not None else None, b.h if b is not None else None, out.ctypes.data_as(_dp)))
        return out

    def check_form_source(self, with_matrix=True, gram=False):
        """Compile-only check of the run-time form against the kernels the drivers would launch for the current degrees (no GPU needed)."""
        # gram: False / True = the struct declares MAT_PAIR_MASK; 2, 3, 4 = the pencil walk / the vector kernel / state_pencil instead
        _ck(lib().IGXCheckFormSource(self.h, 1 if with_matrix else 0, int(gram)))

    def comm_overlap_ms(self):
        """ms by which the upper face of axis 2 was packed before the end of the assembly in the last reduce_ghost_rows."""
        ms = C.c_double(0)
        _ck(lib().IGXCommGetOverlap(self.h, C.byref(ms)))
        return ms.value

    def comm_link_rate(self):
        """(GB/s per direction, source, probe ms, faces): what the face-first decision uses (IGXCommGetLinkRate)."""
        gbs, src, ms, faces = C.c_double(0), C.c_int(0), C.c_double(0), C.c_int(0)
        _ck(lib().IGXCommGetLinkRate(self.h, C.byref(gbs), C.byref(src), C.byref(ms), C.byref(faces)))
        return gbs.value, ("constant", "measured", "env")[src.value], ms.value, faces.value

    def face_passes(self):
        n = C.c_int(0)
        _ck(lib().IGXGetFacePasses(self.h, C.byref(n)))
        return n.value

    def comm_early_phases(self):
        """phases (upper faces of axes 2, 1, 0) of the last reduce_ghost_rows that were packed behind a face mark of the assembly"""
        n = C.c_int(0)
        _ck(lib().IGXCommGetEarlyPhases(self.h, C.byref(n)))
        return n.value

    def clock_probe(self):
        """(shader MHz, elements walked) over the probe workgroups of the pencil-kernel launches since the last call; needs IGX_CLOCK_PROBE=1 at creation."""
        mhz, ne = C.c_double(0), C.c_int64(0)
        _ck(lib().IGXGetClockProbe(self.h, C.byref(mhz), C.byref(ne)))
        return mhz.value, ne.value

    def coloring(self):
        nc = (C.c_int * 3)()
        _ck(lib().IGXGetColoring(self.h, nc))
        return list(nc)

    def element_color(self, axis, e): return lib().IGXGetElementColor(self.h, axis, e)


def device_info():
    buf = C.create_string_buffer(256)
    lib().IGXGetDeviceInfo(buf, 256)
    return buf.value.decode()
