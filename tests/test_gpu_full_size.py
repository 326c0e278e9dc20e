"""The BASELINE sizes (no oracle can run there): size-independent properties -- 3-D p=3 Poisson on 256^3 elements (5.84e9
non-zeros) and p=2 on 128^3, checked on the device through torch views of the library's arrays -- and, for the identity-geometry
configs, VALUES: every distinct kind of row against the CPU oracle's matrix of a small mesh scaled by the element size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _DevArray:
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = dict(shape=(n,), typestr=typestr, data=(ptr, False), version=2)


def _views(A):
    import torch
    rp, ci, val = A.device_ptrs()
    return (torch.as_tensor(_DevArray(rp, A.nbrows + 1, "<i8"), device="cuda"),
            torch.as_tensor(_DevArray(ci, A.nblocks, "<i4"), device="cuda"),
            torch.as_tensor(_DevArray(val, A.nblocks, "<f8"), device="cuda"))


def _rowsums(rp, val, rows_per_chunk=2_000_000):
    """Row sums in chunks below 2^31 entries (torch.segment_reduce indexes with 32 bits)."""
    import torch
    out = torch.empty(rp.numel() - 1, dtype=torch.float64, device=val.device)
    for r0 in range(0, rp.numel() - 1, rows_per_chunk):
        r1 = min(r0 + rows_per_chunk, rp.numel() - 1)
        lo, hi = int(rp[r0]), int(rp[r1])
        out[r0:r1] = torch.segment_reduce(val[lo:hi], "sum", lengths=rp[r0 + 1:r1 + 1] - rp[r0:r1])
    return out


def _int1d(p, N):
    """integral of every basis function of the uniform open knot vector over [0,1] = (U[i+p+1]-U[i])/(p+1)"""
    U = np.concatenate([np.zeros(p + 1), np.arange(1, N) / N, np.ones(p + 1)])
    return (U[p + 1:] - U[:-p - 1]) / (p + 1)


# ---- VALUES at full size.  On the identity geometry with uniform open knot vectors the entry K[(i), (i + d)] of a node depends only on
# where the node sits relative to the two ends of each axis (the first / last 2p nodes see the repeated end knots, every other node is
# "interior") and on the element size h: K scales with h^(dim - 2) per field pair, F with h^dim.  So a SMALL oracle mesh (3p + 3
# elements per axis) holds, up to that scale, every distinct row of the 256^3 matrix: rows of the big matrix are sampled over all
# combinations of per-axis position classes -- 0 .. 2p-1 from either end and a middle node -- and every one of their entries (343
# per row at p = 3) is compared with the oracle's entry at the corresponding small-mesh node pair.  A coefficient that is wrong but
# conservative (row sums still zero, pattern intact) does not survive this.
def _axis_classes(p, n_big, n_small):
    """(sampled big-mesh indices, the small-mesh index of each)"""
    m = 2 * p
    big = list(range(m)) + [n_big // 2] + list(range(n_big - m, n_big))
    small = list(range(m)) + [n_small // 2] + list(range(n_small - m, n_small))
    return np.array(big), np.array(small)


def _sampled_rows_vs_scaled_oracle(A, p, N, dof, A_small, Ns, scale, tol=1e-11):
    import torch
    n, ns, W = N + p, Ns + p, 2 * p + 1
    rp, ci, val = A.device_ptrs()
    bs2 = dof * dof
    rpt = torch.as_tensor(_DevArray(rp, A.nbrows + 1, "<i8"), device="cuda")
    cit = torch.as_tensor(_DevArray(ci, A.nblocks, "<i4"), device="cuda")
    vt = torch.as_tensor(_DevArray(val, A.nblocks * bs2, "<f8"), device="cuda").view(-1, bs2)
    big, small = _axis_classes(p, n, ns)
    B0, B1, B2 = np.meshgrid(big, big, big, indexing="ij")
    S0, S1, S2 = np.meshgrid(small, small, small, indexing="ij")
    rows = (B0 + n * (B1 + n * B2)).reshape(-1)                      # axis 0 fastest
    srow = np.stack([S0.reshape(-1), S1.reshape(-1), S2.reshape(-1)], axis=1)
    rows_t = torch.as_tensor(rows, device="cuda")
    lo, hi = rpt[rows_t].cpu().numpy(), rpt[rows_t + 1].cpu().numpy()
    ref = A_small.tocsr()
    worst, vmax, nent = 0.0, 0.0, 0
    for k, r in enumerate(rows):
        cols = cit[lo[k]:hi[k]].cpu().numpy().astype(np.int64)
        vals = vt[lo[k]:hi[k]].cpu().numpy()
        c0, c1, c2 = cols % n, (cols // n) % n, cols // (n * n)
        r0, r1, r2 = r % n, (r // n) % n, r // (n * n)
        sc = (srow[k, 0] + (c0 - r0)) + ns * ((srow[k, 1] + (c1 - r1)) + ns * (srow[k, 2] + (c2 - r2)))     # the same offsets from the small-mesh node
        sr = srow[k, 0] + ns * (srow[k, 1] + ns * srow[k, 2])
        for i in range(dof):
            dense_row = ref[sr * dof + i].toarray().ravel()          # (one row of the small matrix, dense: (Ns + p)^3 dof numbers)
            for j in range(dof):
                want = dense_row[sc * dof + j] * scale
                got = vals[:, i * dof + j]
                worst = max(worst, float(np.abs(got - want).max()))
                vmax = max(vmax, float(np.abs(want).max()))
        nent += cols.size * bs2
    assert worst <= tol * vmax, (worst, vmax)
    return rows, srow, nent


@pytest.mark.parametrize("p,N", [(3, 256), (2, 128)])
def test_poisson_full_size_values_vs_scaled_oracle(p, N):
    """Every distinct kind of row of the full-size Poisson System -- with and without the Dirichlet data of demo/Poisson3D.c:37-43 --
    against the CPU oracle's matrix of a (3p + 3)^3 mesh scaled by the ratio of the element sizes; F likewise (h^3)."""
    import petiga_amd as P
    import oracle_api as O
    Ns = 3 * p + 3
    g, orc = P.IGX(3, 1), O.OracleIGA(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, N)
        orc.axis_uniform(i, p, Ns)
    g.setup(); orc.setup()
    g.set_form("poisson")
    A, b = g.create_mat(), g.create_vec()
    s = Ns / N                                   # h_big / h_small
    for bc in (False, True):
        if bc:
            for x in (g, orc):
                for d in range(3):
                    for sd in range(2):
                        x.set_boundary_value(d, sd, 0, 1.0)
        g.compute_system(A, b); g.synchronize()
        assert "pencil" in g.kernel_name()
        A_o, b_o = orc.compute_system("orc_form_poisson")
        M = A_o.scipy()
        if bc:      # a fixed row holds the element multiplicity on its diagonal (no length in it): undo the scale there
            n_s = Ns + p
            idx = np.arange(n_s ** 3)
            i0, i1, i2 = idx % n_s, (idx // n_s) % n_s, idx // (n_s * n_s)
            onb = (i0 == 0) | (i0 == n_s - 1) | (i1 == 0) | (i1 == n_s - 1) | (i2 == 0) | (i2 == n_s - 1)
            import scipy.sparse as sp
            D = sp.diags(np.where(onb, 1.0 / s, 1.0))
            M = (D @ M).tocsr()
        rows, srow, nent = _sampled_rows_vs_scaled_oracle(A, p, N, 1, M, Ns, s)
        assert len(rows) == (4 * p + 1) ** 3 and nent >= len(rows) * (p + 1) ** 3      # (a corner row has (p + 1)^3 entries, an interior one (2p + 1)^3)
        if not bc:
            n_s = Ns + p
            sr = srow[:, 0] + n_s * (srow[:, 1] + n_s * srow[:, 2])
            assert np.abs(b.get()[rows] - b_o[sr] * s ** 3).max() <= 1e-12 * np.abs(b_o).max() * s ** 3


@pytest.mark.parametrize("p,N", [(3, 256), (2, 128)])
def test_poisson_full_size_properties(p, N):
    import torch
    import petiga_amd as P
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, N)
    g.setup()
    g.set_form("poisson")
    A, b = g.create_mat(), g.create_vec()
    n = N + p
    assert A.nbrows == n ** 3 and A.nblocks == (n * (2 * p + 1) - p * (p + 1)) ** 3      # 17 373 979 rows / 5 841 725 401 nnz at p=3
    # --- no Dirichlet data: every row sums to zero (partition of unity), F_a = integral of N_a (closed form)
    g.compute_system(A, b); g.synchronize()
    assert "pencil" in g.kernel_name() or "gram_patch" in g.kernel_name()
    patch = "gram_patch" in g.kernel_name()      # config 2's default since round 6: up to nine wavefronts add to a window entry, in no fixed order
    rp, ci, val = _views(A)
    scale = float(val.abs().max())
    rowsum = _rowsums(rp, val)
    assert float(rowsum.abs().max()) <= 1e-11 * scale
    w = _int1d(p, N)
    F = (w[:, None, None] * w[None, :, None] * w[None, None, :]).reshape(-1)       # axis 0 fastest: index = i0 + n*(i1 + n*i2); symmetric in the axes
    assert np.abs(b.get() - F).max() <= 1e-13 * F.max()
    del rowsum
    # --- with u = 1 on the six faces: repeatable bit for bit, stale values never survive (first-touch stores),
    #     Dirichlet rows carry the element multiplicity on the diagonal and multiplicity * value on the right-hand side
    for d in range(3):
        for s in range(2):
            g.set_boundary_value(d, s, 0, 1.0)
    g.compute_system(A, b); g.synchronize()
    keep = val.clone(); bk = b.get()
    val.fill_(float("nan"))
    g.compute_system(A, b); g.synchronize()
    if patch:      # stale values never survive, and two assemblies agree to a few ulps of the largest entry (the pencil walks: bit for bit)
        assert bool(torch.isfinite(val).all()) and float((val - keep).abs().max()) <= 4e-15 * float(keep.abs().max()) and np.abs(b.get() - bk).max() <= 4e-15 * np.abs(bk).max()
    else:
        assert torch.equal(val, keep) and np.array_equal(b.get(), bk)
    del keep
    idx = np.arange(n ** 3)
    i0, i1, i2 = idx % n, (idx // n) % n, idx // (n * n)
    onb = (i0 == 0) | (i0 == n - 1) | (i1 == 0) | (i1 == n - 1) | (i2 == 0) | (i2 == n - 1)
    mult = lambda i: np.minimum(np.minimum(i + 1, p + 1), np.minimum(n - i, p + 1))
    m = (mult(i0) * mult(i1) * mult(i2)).astype(float)
    assert np.array_equal(bk[onb], m[onb])                       # F_k = multiplicity * 1.0
    rowsum = _rowsums(rp, val).cpu().numpy()
    assert np.array_equal(rowsum[onb], m[onb])                   # a fixed row holds only its diagonal
    deep = (np.minimum(i0, n - 1 - i0) > p) & (np.minimum(i1, n - 1 - i1) > p) & (np.minimum(i2, n - 1 - i2) > p)
    assert np.abs(rowsum[deep]).max() <= 1e-11 * scale           # rows that see no fixed column still sum to zero
    # lifting: b_i = F_i - sum_k K_ik over the fixed columns, and the full row sums to zero => b_i = F_i + (row sum after fix-up)
    free_near = ~onb & ~deep
    assert np.abs(bk[free_near] - (F[free_near] + rowsum[free_near])).max() <= 1e-11 * max(scale, 1.0)


def test_elasticity_full_size_values_vs_scaled_oracle():
    """Config 3 at 128^3: the 3 x 3 blocks of every distinct kind of row against the oracle's demo/Elasticity3D.c:13-46 on a 12^3 mesh,
    scaled by the ratio of the element sizes (the blocks are sums of grad N . grad N terms: h^1 in 3-D), with the demo's boundary
    data (demo/Elasticity3D.c:67-70: the face x = 0 clamped, u_x = 1 on the face x = 1)."""
    import petiga_amd as P
    import oracle_api as O
    p, N, Ns = 3, 128, 12
    g, orc = P.IGX(3, 3), O.OracleIGA(3, 3)
    for i in range(3):
        g.axis_uniform(i, p, N)
        orc.axis_uniform(i, p, Ns)
    g.setup(); orc.setup()
    for x in (g, orc):
        for f in range(3):
            x.set_boundary_value(0, 0, f, 0.0)
        x.set_boundary_value(0, 1, 0, 1.0)
    g.set_form("elasticity", (1.3, 0.7))
    A, b = g.create_mat(), g.create_vec()
    g.compute_system(A, b); g.synchronize()
    assert "block_pencil(mfma" in g.kernel_name()
    A_o, _ = orc.compute_system("orc_form_elasticity", O.ElasticityCtx(1.3, 0.7))
    s = Ns / N
    M = A_o.scipy()
    n_s = Ns + p
    idx = np.arange(n_s ** 3)
    i0 = idx % n_s
    fixed = np.zeros((n_s ** 3, 3), dtype=bool)
    fixed[i0 == 0, :] = True
    fixed[i0 == n_s - 1, 0] = True
    import scipy.sparse as sp
    M = (sp.diags(np.where(fixed.reshape(-1), 1.0 / s, 1.0)) @ M).tocsr()      # fixed rows: the multiplicity on the diagonal carries no length
    rows, _, nent = _sampled_rows_vs_scaled_oracle(A, p, N, 3, M, Ns, s)
    assert len(rows) == 13 ** 3 and nent >= 9 * 64 * len(rows)


def test_elasticity_full_size_properties():
    """Config 3 (Elasticity3D p=3, 128^3, 53 GB of values): rigid translations are in the null space of the
    unconstrained operator (every 3x3 block row sums to zero over the row), F = 0, bitwise repeatable."""
    import torch
    import petiga_amd as P
    N, p = 128, 3
    g = P.IGX(3, 3)
    for i in range(3):
        g.axis_uniform(i, p, N)
    g.setup()
    g.set_form("elasticity", (1.0, 1.0))
    A, b = g.create_mat(), g.create_vec()
    n = N + p
    assert A.nbrows == n ** 3 and A.bs == 3 and A.nblocks == (n * 7 - 12) ** 3
    g.compute_system(A, b); g.synchronize()
    assert "block_pencil(mfma" in g.kernel_name()
    rp, ci, val = A.device_ptrs()
    rpt = torch.as_tensor(_DevArray(rp, A.nbrows + 1, "<i8"), device="cuda")
    v = torch.as_tensor(_DevArray(val, A.nblocks * 9, "<f8"), device="cuda").view(-1, 9)
    scale = float(v.abs().max())
    worst = 0.0
    step = 1_000_000
    for r0 in range(0, A.nbrows, step):
        r1 = min(r0 + step, A.nbrows)
        lo, hi = int(rpt[r0]), int(rpt[r1])
        sums = torch.segment_reduce(v[lo:hi], "sum", lengths=rpt[r0 + 1:r1 + 1] - rpt[r0:r1], axis=0)
        worst = max(worst, float(sums.abs().max()))
    assert worst <= 1e-11 * scale
    assert np.abs(b.get()).max() == 0.0
    chk = float(v.sum()), float(v.abs().sum())
    v.fill_(float("nan"))
    g.compute_system(A, b); g.synchronize()
    assert (float(v.sum()), float(v.abs().sum())) == chk        # same bits -> same sums


def _colsums(ci, val, ncols, bs=1, chunk=200_000_000):
    """Column sums of a (block) CSR on the device: scalar column = block column * bs + j."""
    import torch
    out = torch.zeros(ncols * bs, dtype=torch.float64, device=val.device)
    v = val.view(-1, bs, bs) if bs > 1 else None
    for b0 in range(0, ci.numel(), chunk // (bs * bs)):
        b1 = min(b0 + chunk // (bs * bs), ci.numel())
        c = ci[b0:b1].to(torch.int64)
        if bs == 1:
            out.index_add_(0, c, val[b0:b1])
        else:
            blk = v[b0:b1].sum(dim=1)                       # sum over the block's rows: [blocks][bs columns]
            out.view(ncols, bs).index_add_(0, c, blk)
    return out


def test_cahn_hilliard_full_size_properties():
    """Config 4 (CahnHilliard3D p=2 C1, 256^3 elements, natural boundaries: 17.2 M rows, 2.1e9 non-zeros) on the device, the
    invariants tests/test_oracle_invariants.py checks on the oracle: every term of the residual but N_a c_t carries grad N_a or
    lap N_a, which sum to zero over a (partition of unity), so  sum_a R_a = int c_t = V . int N  and the column sums of the
    tangent are shift * int N_b (demo/CahnHilliard3D.c:55-179); NaN-poisoned matrix reproduced bit for bit."""
    import torch
    import petiga_amd as P
    N, p = 256, 2
    g = P.IGX(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, N)
    g.setup()
    g.set_form("cahnhilliard", (1.5, 200.0, 0.63, 1.0, 1.0 / (3.0 * N * N), 1.0))
    A, b = g.create_mat(), g.create_vec()
    n = N + p
    assert A.nbrows == n ** 3 and A.nblocks == (n * 5 - 6) ** 3           # 17 173 512 rows, 2 116 874 304 non-zeros
    rng = np.random.default_rng(3)
    Uh, Vh = 0.63 + 0.05 * (2 * rng.random(n ** 3) - 1), rng.standard_normal(n ** 3)
    U, V = g.create_vec().set(Uh), g.create_vec().set(Vh)
    shift = 250.0
    g.compute_ifunction(shift, V, 0.0, U, b)
    g.compute_ijacobian(shift, V, 0.0, U, A); g.synchronize()
    assert "state_pencil" in g.kernel_name()
    w = _int1d(p, N)
    bN = (w[None, None, :] * w[None, :, None] * w[:, None, None]).reshape(-1)
    F = b.get()
    assert abs(F.sum() - Vh @ bN) <= 1e-11 * np.abs(F).sum()
    rp, ci, val = _views(A)
    scale = float(val.abs().max())
    cs = _colsums(ci, val, n ** 3).cpu().numpy()
    assert np.abs(cs - shift * bN).max() <= 1e-10 * scale
    chk = float(val.sum()), float(val.abs().sum())
    val.fill_(float("nan"))
    g.compute_ijacobian(shift, V, 0.0, U, A); g.synchronize()
    assert (float(val.sum()), float(val.abs().sum())) == chk            # first-touch stores reach every entry; same bits -> same sums


def test_cahn_hilliard_full_size_values_vs_scaled_oracle():
    """Config 4's Tangent at 256^3 by VALUE.  At a uniform state c the Tangent of demo/CahnHilliard3D.c:111-179 is
    shift N_a N_b + M mu'(c) grad N_a . grad N_b + M lap N_a lap N_b with mu' proportional to L0^2 / lambda: three terms that scale
    with h^3, h, 1/h.  Give the small oracle mesh shift s^4 and lambda / s^2 (s = h_big / h_small) and the big matrix is the small one
    times 1/s, row class by row class (see _sampled_rows_vs_scaled_oracle).  The state-dependent terms (grad c, lap c) are zero here:
    they are compared with the oracle at oracle sizes, on random states (tests/test_gpu_state_pencil.py)."""
    import petiga_amd as P
    import oracle_api as O
    N, p, Ns = 256, 2, 9
    theta, alpha, cbar, L0, lam, tau = 1.5, 200.0, 0.63, 1.0, 1.0 / (3.0 * N * N), 1.0
    g, orc = P.IGX(3, 1), O.OracleIGA(3, 1)
    for i in range(3):
        g.axis_uniform(i, p, N)
        orc.axis_uniform(i, p, Ns)
    g.setup(); orc.setup()
    g.set_form("cahnhilliard", (theta, alpha, cbar, L0, lam, tau))
    A = g.create_mat()
    n = N + p
    U, V = g.create_vec().set(np.full(n ** 3, cbar)), g.create_vec().set(np.zeros(n ** 3))
    shift = 250.0
    g.compute_ijacobian(shift, V, 0.0, U, A); g.synchronize()
    assert "state_pencil" in g.kernel_name() and "packed" in g.kernel_name(), g.kernel_name()
    s = Ns / N
    ns = Ns + p
    J_o = orc.compute_ijacobian("orc_form_ch_tangent", O.CahnHilliardCtx(theta, alpha, cbar, L0, lam / (s * s), tau), shift * s ** 4,
                                np.zeros(ns ** 3), 0.0, np.full(ns ** 3, cbar))
    rows, _, nent = _sampled_rows_vs_scaled_oracle(A, p, N, 1, J_o.scipy(), Ns, 1.0 / s, tol=1e-10)
    assert len(rows) == 9 ** 3 and nent >= 27 * len(rows)


def test_cahn_hilliard_on_a_nurbs_patch_properties():
    """Cahn-Hilliard at 128^3 on the bench's rational map (state_pencil_geo + vec_sumfact at second order; no oracle runs at this
    size).  NURBS functions are a partition of unity as well, so sum_a R_a = int c_t dx = V . m and the column sums of the tangent
    are shift * m with m_b = int R_b dx -- taken from an independent kernel: the load vector of the Poisson System driver (f = 1, no
    Dirichlet faces) on the same geometry (gram_pencil, oracle-checked at small sizes); NaN-poisoned matrix reproduced bit for bit."""
    import os
    import sys
    import petiga_amd as P
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    N, p = 128, 2
    X, W = bench._bench_geometry(p, N, (False, False, False))
    def space():
        g = P.IGX(3, 1)
        for i in range(3):
            g.axis_uniform(i, p, N)
        g.setup()
        g.set_geometry(X, W)
        return g
    gp = space()
    gp.set_form("poisson")
    Ap, bp = gp.create_mat(), gp.create_vec()
    gp.compute_system(Ap, bp); gp.synchronize()
    assert "mapped geometry" in gp.kernel_name()
    m = bp.get().copy()
    del Ap, bp, gp
    assert abs(m.sum() - 1.0) < 0.2 and m.min() > 0           # (the volume of the warped unit cube)
    g = space()
    g.set_form("cahnhilliard", (1.5, 200.0, 0.63, 1.0, 1.0 / (3.0 * N * N), 1.0))
    A, b = g.create_mat(), g.create_vec()
    n = N + p
    rng = np.random.default_rng(3)
    Uh, Vh = 0.63 + 0.05 * (2 * rng.random(n ** 3) - 1), rng.standard_normal(n ** 3)
    U, V = g.create_vec().set(Uh), g.create_vec().set(Vh)
    shift = 250.0
    g.compute_ifunction(shift, V, 0.0, U, b); g.synchronize()
    assert "vec_sumfact" in g.kernel_name(), g.kernel_name()
    g.compute_ijacobian(shift, V, 0.0, U, A); g.synchronize()
    assert "state_pencil<CahnHilliard>" in g.kernel_name() and "mapped geometry" in g.kernel_name(), g.kernel_name()
    F = b.get()
    assert abs(F.sum() - Vh @ m) <= 1e-10 * np.abs(F).sum()
    rp, ci, val = _views(A)
    scale = float(val.abs().max())
    cs = _colsums(ci, val, n ** 3).cpu().numpy()
    assert np.abs(cs - shift * m).max() <= 1e-9 * scale
    chk = float(val.sum()), float(val.abs().sum())
    val.fill_(float("nan"))
    g.compute_ijacobian(shift, V, 0.0, U, A); g.synchronize()
    assert (float(val.sum()), float(val.abs().sum())) == chk


def test_navier_stokes_vms_one_gpu_share_properties():
    """Config 5's share of one GPU (NavierStokesVMS p=3, 96^3 elements of the 192^3 mesh, 4 fields, the bench's rational NURBS
    map, axes 0 and 2 periodic, no-slip walls on axis 1: 3.65 M block rows, 40 GB of values) on the device
    (demo/NavierStokesVMS.c:78-244):
      * a uniform flow (zero pressure: axis 1 has faces, a constant pressure would leave p * int N_a n dS on them) without
        forcing and without walls is a steady solution: R = 0 on the mapped geometry;
      * with the walls, for any state the pressure rows sum to int div u = 0 (every other term of Rp carries grad N_a; u = 0 on
        the walls, the other axes are periodic), whatever the geometry;
      * IJacobian: a no-slip dof's row and column hold nothing but the diagonal = the number of elements that hold the node (16);
      * NaN-poisoned matrix reproduced bit for bit."""
    import os
    import sys
    import torch
    import petiga_amd as P
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import _bench_geometry
    N, p = 96, 3
    per = (True, False, True)

    def build(walls):
        g = P.IGX(3, 4)
        for i in range(3):
            g.axis_uniform(i, p, N, periodic=per[i])
        g.setup()
        X, W = _bench_geometry(p, N, per)
        g.set_geometry(X, W)
        if walls:
            for s in range(2):
                for f in range(3):
                    g.set_boundary_value(1, s, f, 0.0)
        return g
    nn = N * (N + p) * N
    dt = 1e-2
    # --- uniform flow, no forcing, no walls
    g = build(False)
    g.set_form("nsvms", (1.472e-4, 0.0, 0.0, 0.0, dt))
    b = g.create_vec()
    U = g.create_vec().set(np.tile([0.3, -0.2, 0.5, 0.0], nn))
    V = g.create_vec().set(np.zeros(4 * nn))
    g.compute_ifunction(2.0 / dt, V, 0.0, U, b); g.synchronize()
    assert "vec_sumfact" in g.kernel_name()
    assert np.abs(b.get()).max() <= 1e-12
    del g, b, U, V
    # --- walls, random state
    g = build(True)
    g.set_form("nsvms", (1.472e-4, 3.37204e-3, 0.0, 0.0, dt))
    rng = np.random.default_rng(11)
    U, V = g.create_vec().set(rng.standard_normal(4 * nn) * 0.3), g.create_vec().set(rng.standard_normal(4 * nn) * 0.1)
    b, A = g.create_vec(), g.create_mat()
    assert A.nbrows == nn and A.bs == 4 and A.nblocks == nn // (N + p) * ((N + p) * 7 - 12) * 49
    g.compute_ifunction(2.0 / dt, V, 0.0, U, b)
    F = b.get()
    assert abs(F[3::4].sum()) <= 1e-10 * np.abs(F[3::4]).sum()
    g.compute_ijacobian(2.0 / dt, V, 0.0, U, A); g.synchronize()
    rp, ci, val = A.device_ptrs()
    rpt = torch.as_tensor(_DevArray(rp, A.nbrows + 1, "<i8"), device="cuda")
    cit = torch.as_tensor(_DevArray(ci, A.nblocks, "<i4"), device="cuda")
    v = torch.as_tensor(_DevArray(val, A.nblocks * 16, "<f8"), device="cuda").view(-1, 4, 4)
    # wall nodes: axis-1 index 0 and N+p-1; row index = i0 + N*(i1 + (N+p)*i2)
    i0, i2 = torch.arange(N, device="cuda"), torch.arange(N, device="cuda")
    for i1 in (0, N + p - 1):
        rows = (i0[None, :] + N * (i1 + (N + p) * i2[:, None])).reshape(-1)
        for r in rows[:: 97].tolist():                       # a sample of wall rows
            lo, hi = int(rpt[r]), int(rpt[r + 1])
            blk, cols = v[lo:hi], cit[lo:hi]
            diag = blk[cols == r][0]
            off = blk.clone(); off[cols == r] = 0
            assert float(off[:, :3, :].abs().max()) == 0.0                     # fixed rows (u, v, w) hold nothing off the diagonal block
            assert torch.equal(diag[:3, :3], 16.0 * torch.eye(3, dtype=torch.float64, device="cuda")) and float(diag[:3, 3].abs().max()) == 0.0
    # fixed columns: a sample of interior rows next to the wall has zeros in the wall nodes' velocity columns
    wall = torch.zeros(nn, dtype=torch.bool, device="cuda")
    idx = torch.arange(nn, device="cuda")
    i1_of = (idx // N) % (N + p)
    wall[(i1_of == 0) | (i1_of == N + p - 1)] = True
    r = int(5 + N * (1 + (N + p) * 7))
    lo, hi = int(rpt[r]), int(rpt[r + 1])
    wc = wall[cit[lo:hi].to(torch.int64)]
    assert bool(wc.any()) and float(v[lo:hi][wc][:, :, :3].abs().max()) == 0.0
    vf = v.view(-1)
    chk = float(vf.sum()), float(vf.abs().sum())
    vf.fill_(float("nan"))
    g.compute_ijacobian(2.0 / dt, V, 0.0, U, A); g.synchronize()
    assert (float(vf.sum()), float(vf.abs().sum())) == chk
