"""The hand-back the PETSc adapter performs (adapter/petiga_amd_petsc.c), call for call through ctypes:
IGXCreateFromTables (the tables a set-up PetIGA IGA holds) -> boundary tables -> IGXSetForm -> IGXVecCopyFromGhosted (the
ghosted local array of IGAGetLocalVecArray) -> IGXCompute* -> IGXMatGetCOO / IGXVecGetIndices in PETSc numbering -> scatter-add
of the engine's value array through those coordinate lists (what MatSetValuesCOO does, duplicates added, rows of not-owned
nodes routed to their owners).  PetIGA's AO (AOCreateMemoryScalable over the ranks' owned boxes in rank order,
src/petigagrid.c:185-199 with IGA_Grid_LocalIndices) is restated here with numpy from the partition; the result, permuted
back to natural numbering, must be the single-rank oracle matrix."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import oracle_api as O
from common import make_pair, warped_geometry

pytestmark = pytest.mark.gpu


def petsc_numbering(ranks_sizes, node_sizes):
    """natural node index -> PETSc node index: ranks in order, each rank's owned box in natural (i fastest) order."""
    ao = np.full(int(np.prod(node_sizes)), -1, dtype=np.int64)
    start = 0
    for s in ranks_sizes:
        ls, lw = s["node_lstart"], s["node_lwidth"]
        k, j, i = np.meshgrid(*[np.arange(ls[d], ls[d] + lw[d]) for d in (2, 1, 0)], indexing="ij")
        nat = (i + node_sizes[0] * (j + node_sizes[1] * k)).reshape(-1)
        ao[nat] = start + np.arange(nat.size)
        start += nat.size
    assert (ao >= 0).all()
    return ao


def tables_from_oracle(orc, P):
    """IGXTables of one rank from the oracle's struct (the layout of PetIGA's struct _p_IGA)."""
    s = orc.s
    t = P.IGXTables()
    t.dim, t.dof, t.order = s.dim, s.dof, s.order
    for i in range(s.dim):
        ax, bd, ta = s.axis[i], s.basis[i], t.axis[i]
        ta.p, ta.m, ta.periodic, ta.nel, ta.nnp, ta.U, ta.span = ax.p, ax.m, ax.periodic, ax.nel, ax.nnp, ax.U, ax.span
        ta.nqp, ta.nen, ta.offset, ta.detJac, ta.weight, ta.point, ta.value = bd.nqp, bd.nen, bd.offset, bd.detJac, bd.weight, bd.point, bd.value
    for name in ("proc_sizes", "proc_ranks", "elem_sizes", "elem_start", "elem_width", "node_sizes", "node_lstart", "node_lwidth", "node_gstart", "node_gwidth"):
        for i in range(3):
            getattr(t, name)[i] = getattr(s, name)[i]
    t.nsd, t.rational, t.geometryX, t.rationalW = s.nsd, s.rational, s.geometryX, s.rationalW
    t.property, t.propertyA = s.property, s.propertyA
    return t


@pytest.mark.parametrize("size,dim,dof,p,N,periodic,form", [(1, 3, 1, 2, (5, 4, 6), (0, 0, 0), "poisson"), (4, 3, 3, 2, (6, 7, 5), (0, 0, 0), "elasticity"),
                                                             (8, 3, 1, 3, (8, 9, 8), (0, 0, 0), "poisson+nurbs"), (2, 3, 1, 2, (6, 6, 8), (1, 0, 1), "bratu"),
                                                             (3, 2, 2, 2, (9, 7), (0, 1), "mass")])
def test_adapter_call_sequence_reassembles_the_oracle_matrix(size, dim, dof, p, N, periodic, form):
    import petiga_amd as P
    periodic = [bool(x) for x in periodic]
    geo = form.endswith("+nurbs")
    form = form.split("+")[0]
    ref, _ = make_pair(dim, dof, p, list(N), periodic=periodic, engine=False)
    Xg = Wg = None
    if geo:
        Xg, Wg = warped_geometry(ref, dim, seed=3, rational=True, amp=0.1)
        ref.set_geometry(Xg, Wg)

    def bcs(g):
        if form in ("poisson", "bratu"):
            for d in range(dim):
                if not periodic[d]:
                    g.set_boundary_value(d, 0, 0, 0.5)
            g.set_boundary_load(dim - 1, 1, 0, 0.75) if not periodic[dim - 1] else None
        elif form == "elasticity":
            for f in range(3):
                g.set_boundary_value(0, 0, f, 0.0)
            g.set_boundary_value(0, 1, 0, 1.0)
    bcs(ref)
    n = ref.global_size()
    rng = np.random.default_rng(4)
    Ug = rng.standard_normal(n) * 0.3
    lam = C.c_double(2.0)
    params = {"elasticity": (1.5, 0.8), "bratu": (2.0,)}.get(form, ())
    if form == "bratu":
        A_o, b_o = ref.compute_jacobian("orc_form_bratu_jacobian", lam, Ug), ref.compute_function("orc_form_bratu_function", lam, Ug)
    else:
        A_o, b_o = ref.compute_system("orc_form_" + form, O.ElasticityCtx(1.5, 0.8) if form == "elasticity" else None)
    node_sizes = list(ref.ranges()["node_sizes"]) + [1] * (3 - dim)

    ranks, pieces, vec_pieces = [], [], []
    for r in range(size):
        orc = O.OracleIGA(dim, dof)
        for i in range(dim):
            orc.axis_uniform(i, p, N[i], periodic=periodic[i])
        orc.set_partition(size, r)
        orc.setup()
        if geo:
            orc.set_geometry(Xg, Wg)
        g = P.IGX.from_tables(tables_from_oracle(orc, P))                 # IGAGetAmd
        bcs(g)                                                             # boundary tables of the IGAForm
        g.set_form(form, params)                                           # IGASetFormAMD
        A, b = g.create_mat(), g.create_vec()
        s = g.sizes()
        ranks.append(s)
        if form == "bratu":
            # IGAAmdSetState: the ghosted local array [gw2][gw1][gw0][dof] of U (periodic wrap = the lgmap's)
            gs, gw = s["node_gstart"], s["node_gwidth"]
            k, j, i = np.meshgrid(*[np.arange(gs[d], gs[d] + gw[d]) % node_sizes[d] for d in (2, 1, 0)], indexing="ij")
            nat = (i + node_sizes[0] * (j + node_sizes[1] * k)).reshape(-1)
            ghosted = Ug.reshape(-1, dof)[nat].reshape(-1)
            U = g.create_vec().set_from_ghosted(ghosted)
            assert np.array_equal(U.get_ghosted(), ghosted)
            g.compute_jacobian(U, A)
            g.compute_function(U, b)
        else:
            g.compute_system(A, b)
        g.synchronize()
        ci, cj = A.coo(numbering=1)                                        # IGAAmdHandBackMat: coordinate list, PETSc numbering
        nat_i, nat_j = A.coo(numbering=0)
        rows, cols, vals = A.to_coo_global()
        assert np.array_equal(nat_i, rows) and np.array_equal(nat_j, cols)  # natural numbering = the layout maps
        pieces.append((ci, cj, A.host(True), nat_i, nat_j))
        vec_pieces.append((b.indices(numbering=1), b.get(), b.indices(numbering=0), b.indices(numbering=1, owned_only=True)))
        ci_own, _ = A.coo(numbering=1, owned_only=True)
        assert ((ci_own == -1) | (ci_own == ci)).all()

    ao = petsc_numbering(ranks, node_sizes)
    # the library's PETSc numbering is PetIGA's AO
    for (ci, cj, vals, ni, nj), (vi, vv, vn, vown) in zip(pieces, vec_pieces):
        assert np.array_equal(ci, ao[ni // dof] * dof + ni % dof) and np.array_equal(cj, ao[nj // dof] * dof + nj % dof)
        assert np.array_equal(vi, ao[vn // dof] * dof + vn % dof)
    # every PETSc row is owned by exactly one rank's owned_only list
    owned = np.concatenate([v[3][v[3] >= 0] for v in vec_pieces])
    assert np.array_equal(np.sort(owned), np.arange(n))
    # MatSetValuesCOO / VecSetValuesCOO: add duplicates, over all ranks
    M = sp.coo_matrix((np.concatenate([x[2] for x in pieces]), (np.concatenate([x[0] for x in pieces]), np.concatenate([x[1] for x in pieces]))), shape=(n, n)).tocsr()
    bp = np.zeros(n)
    for vi, vv, _, _ in vec_pieces:
        np.add.at(bp, vi, vv)
    # back to natural numbering with the AO: the oracle's single-rank system
    perm = (ao[:, None] * dof + np.arange(dof)[None, :]).reshape(-1)      # natural scalar index -> PETSc scalar index
    Mn = M[perm][:, perm]
    Ao = A_o.scipy()
    scale = np.abs(Ao.data).max()
    assert abs(Mn - Ao).max() <= 1e-11 * scale
    assert np.abs(bp[perm] - b_o).max() <= 1e-11 * max(np.abs(b_o).max(), 1.0)


def test_device_coordinate_lists_match_the_host_lists():
    """IGXMatGetCOODevice (what the adapter hands to MatSetPreallocationCOO of a device Mat type): the lists kept on the device, in
    64-bit and in 32-bit PetscInt width, equal the host lists of IGXMatGetCOO; a problem whose indices do not fit 32 bits is
    refused (PETSC_ERR_ARG_OUTOFRANGE = 63), not truncated."""
    import torch
    import petiga_amd as P
    g = P.IGX(3, 2)
    g.set_comm(4, 1)
    for i, n in enumerate((6, 7, 9)):
        g.axis_uniform(i, 2, n)
    g.setup()
    A = g.create_mat()
    n = A.nblocks * A.bs * A.bs

    class _Dev:
        def __init__(self, ptr, count, typestr):
            self.__cuda_array_interface__ = dict(shape=(count,), typestr=typestr, data=(ptr, False), version=2)
    for numbering in (0, 1):
        for owned in (False, True):
            hi, hj = A.coo(numbering=numbering, owned_only=owned)
            for nbytes, ts in ((8, "<i8"), (4, "<i4")):
                pi, pj = A.coo_device(numbering=numbering, owned_only=owned, index_bytes=nbytes)
                di = torch.as_tensor(_Dev(pi, n, ts), device="cuda").cpu().numpy()
                dj = torch.as_tensor(_Dev(pj, n, ts), device="cuda").cpu().numpy()
                assert np.array_equal(di, hi) and np.array_equal(dj, hj)
    A.free_coo()
    with pytest.raises(P.IGXError) as e:
        A.coo_device(index_bytes=2)
    assert e.value.code == 63
    # 1300^3 nodes x 1 field > 2^31: set-up alone (no matrix is created) is enough to see the refusal through a tiny stand-in:
    big = P.IGX(3, 1)
    for i in range(3):
        big.axis_uniform(i, 1, 1300)
    big.setup()
    assert int(np.prod(big.sizes()["node_sizes"])) > 2 ** 31
