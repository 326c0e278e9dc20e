"""Size-independent properties at the BASELINE sizes (no oracle can run there): 3-D p=3 Poisson on 256^3 elements
(5.84e9 non-zeros) and p=2 on 128^3, checked on the device through torch views of the library's arrays."""
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
    assert "pencil" in g.kernel_name()
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
