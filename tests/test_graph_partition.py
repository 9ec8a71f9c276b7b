"""-seg_metis without METIS (host/graph_partition.cpp): the built-in partitioner, the part-file reader and the reference's
post-processing of a part vector (stable sort of the rows by part, symmetric permutation, work_sharing_arr from the part sizes;
code/mpi_funcs.hpp:494-598, sortPerm code/utilities.hpp:1833-1840, ScsData::permute code/classes_structs.hpp:1620-1700)."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import make_x, mtx_path


def _csr(coo):
    I, J, V = coo.arrays()
    return sp.coo_matrix((V, (I, J)), shape=(coo.n_rows, coo.n_cols)).tocsr()


def _cut(A, part):
    C = A.tocoo()
    return int(np.sum(part[C.row] != part[C.col]))


@pytest.mark.parametrize("name,P", [("FDM-2d-16", 2), ("FDM-2d-16", 4), ("bcsstk13", 3), ("bcsstk13", 8)])
def test_partition_is_balanced_and_applied_like_the_reference(pkg, orc, name, P):
    m = pkg.read_mtx(mtx_path(name))
    n = m.n_rows
    part = pkg.graph_partition(m, P)
    sizes = np.bincount(part, minlength=P)
    assert part.min() >= 0 and part.max() < P and sizes.min() > 0
    assert sizes.max() - sizes.min() <= max(2, int(0.07 * n / P) + 2), sizes          # vertex counts balanced (3 % slack per side)
    pm, wsa, perm = pkg.apply_partition(m, P, part)
    # the reference's contract: perm = stable sort by part, wsa = running sizes
    assert np.array_equal(perm, np.argsort(part, kind="stable"))
    assert np.array_equal(wsa, np.concatenate([[0], np.cumsum(sizes)]))
    inv = np.empty(n, np.int64); inv[perm] = np.arange(n)
    # symmetric permutation: B[inv[i], inv[j]] = A[i, j]; rows ascending, the order of a row's entries kept
    I, J, V = m.arrays()
    I2, J2, V2 = pm.arrays()
    assert np.all(np.diff(I2) >= 0)
    order = np.argsort(inv[I], kind="stable")
    assert np.array_equal(I2, inv[I][order]) and np.array_equal(J2, inv[J][order]) and np.array_equal(V2, V[order])
    # y of the permuted matrix is the permuted y, BIT FOR BIT (same entries per row in the same order)
    x = make_x(n)
    s0 = pkg.convert_to_scs(m, 1, 1); a0 = s0.arrays()
    s1 = pkg.convert_to_scs(pm, 1, 1); a1 = s1.arrays()
    y0 = orc.spmv_scs(1, s0.n_chunks, a0["chunk_ptrs"], a0["chunk_lengths"], a0["col_idxs"], a0["values"], x)
    y1 = orc.spmv_scs(1, s1.n_chunks, a1["chunk_ptrs"], a1["chunk_lengths"], a1["col_idxs"], a1["values"], x[perm])
    assert np.array_equal(y1, y0[perm])
    # the blocks of the permuted matrix are the parts: every row of block p has part p
    for p in range(P):
        assert np.all(part[perm[wsa[p]:wsa[p + 1]]] == p)


def test_partition_recovers_locality_of_a_shuffled_grid(pkg):
    """a 2-d grid matrix whose rows were shuffled: contiguous row blocks (-seg_rows) cut almost every edge, the graph partition finds
    the slabs again"""
    g = pkg.gen_stencil27(24, 24, 1)
    n = g.n_rows
    rng = np.random.default_rng(5)
    sh = rng.permutation(n)
    I, J, V = g.arrays()
    order = np.argsort(sh[I], kind="stable")
    m = pkg.Coo.from_arrays(n, n, sh[I][order], sh[J][order], V[order])
    P = 4
    part = pkg.graph_partition(m, P)
    A = _csr(m)
    naive = np.minimum(np.arange(n) * P // n, P - 1)
    assert _cut(A, part) * 4 < _cut(A, naive), (_cut(A, part), _cut(A, naive))
    sizes = np.bincount(part, minlength=P)
    assert sizes.max() - sizes.min() <= int(0.07 * n / P) + 2


def test_partition_file_and_edge_cases(pkg, tmp_path):
    m = pkg.read_mtx(mtx_path("FDM-2d-16"))
    n = m.n_rows
    part = (np.arange(n) % 3).astype(np.int32)
    f = tmp_path / "part.txt"
    f.write_text("\n".join(str(int(v)) for v in part) + "\n")
    assert np.array_equal(pkg.read_partition(f, n, 3), part)
    with pytest.raises(pkg.UspmvError):
        pkg.read_partition(f, n, 2)                  # part id 2 outside [0, 2)
    with pytest.raises(pkg.UspmvError):
        pkg.read_partition(f, n + 1, 3)              # too few ids
    with pytest.raises(pkg.UspmvError):
        pkg.read_partition(tmp_path / "missing.txt", n, 3)
    # last part empty: the reference's fix-up shifts the inner boundaries by one (code/mpi_funcs.hpp:602-606)
    part2 = np.zeros(n, np.int32); part2[n // 2:] = 1
    _, wsa, _ = pkg.apply_partition(m, 3, part2)
    assert wsa.tolist() == [0, n // 2 - 1, n - 1, n]
    # disconnected graph (two copies of the matrix side by side) and P = 1
    I, J, V = m.arrays()
    mm = pkg.Coo.from_arrays(2 * n, 2 * n, np.concatenate([I, I + n]), np.concatenate([J, J + n]), np.concatenate([V, V]))
    p2 = pkg.graph_partition(mm, 2)
    assert np.bincount(p2).tolist() == [n, n]
    assert np.all(pkg.graph_partition(m, 1) == 0)


def test_tiny_graphs_never_lose_a_part_and_malformed_entries_are_refused(pkg):
    """with n < 2 P the refinement's lower bound n / P - slack was 0 and a vertex could leave a part of size one; entries
    outside the square indexed the degree array out of bounds."""
    for n, P in ((5, 4), (7, 4), (3, 3), (9, 8), (2, 2)):
        I = np.arange(n - 1, dtype=np.int32); J = I + 1
        rows = np.concatenate([I, J, np.arange(n, dtype=np.int32)]); cols = np.concatenate([J, I, np.arange(n, dtype=np.int32)])
        order = np.argsort(rows, kind="stable")
        m = pkg.Coo.from_arrays(n, n, rows[order], cols[order], np.ones(rows.size))
        part = pkg.graph_partition(m, P)
        assert part.min() >= 0 and part.max() < P
        sizes = np.bincount(part, minlength=P)
        assert sizes.min() >= 1, (n, P, sizes)
    import ctypes
    from ultimate_spmv_amd import binding as B
    # a handle whose arrays were tampered with after creation (uspmv_coo_create itself refuses such entries)
    m = pkg.Coo.from_arrays(4, 4, [0, 1, 2, 3], [1, 2, 3, 0], [1.0, 1.0, 1.0, 1.0])
    Ip, Jp, Vp = B._i32p(), B._i32p(), B._f64p()
    assert B.lib().uspmv_coo_arrays(m.h, ctypes.byref(Ip), ctypes.byref(Jp), ctypes.byref(Vp)) == 0
    Jp[2] = 9
    part = np.zeros(4, np.int32)
    rc = B.lib().uspmv_graph_partition(m.h, 2, part.ctypes.data_as(B._i32p))
    assert rc != 0 and b"outside" in B.lib().uspmv_last_error()
