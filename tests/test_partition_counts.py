"""uspmv_seg_from_row_counts / uspmv_gen_stencil27_row_counts: the ranks of a distributed run agree on the reference's partition
(seg_work_sharing_arr, code/mpi_funcs.hpp:424-622) of a generated matrix from per-row entry counts alone -- bit-identical to the
rule applied to the row-sorted COO, for both seg methods, and to the golden work_sharing_arr of the reference on a real matrix."""
import json
import os

import numpy as np

from conftest import GOLDEN, mtx_path


def test_seg_from_row_counts_equals_the_coo_rule(pkg):
    from ultimate_spmv_amd.distributed import seg_from_row_counts as py_seg
    for shape, dof in (((7, 5, 9), 1), ((16, 16, 16), 1), ((3, 3, 40), 2), ((12, 1, 1), 3)):
        coo = pkg.gen_stencil27(*shape, dof=dof)
        cnt = pkg.gen_stencil27_row_counts(*shape, dof=dof)
        I, J, V = coo.arrays()
        assert np.array_equal(np.bincount(I, minlength=coo.n_rows), cnt)
        part = pkg.gen_stencil27_row_counts(*shape, dof=dof, row_begin=5, row_end=11)
        assert np.array_equal(part, cnt[5:11])
        for P in (1, 2, 3, 4, 7, 8):
            if coo.n_rows < P:
                continue
            for method in ("seg-rows", "seg-nnz"):
                a = pkg.seg_work_sharing_arr(coo, method, P)
                assert np.array_equal(a, pkg.seg_from_row_counts(cnt, method, P)), (shape, dof, P, method)
                assert np.array_equal(a, py_seg(cnt, method, P)), (shape, dof, P, method)


def test_seg_from_row_counts_on_the_reference_golden(pkg):
    meta = json.load(open(os.path.join(GOLDEN, "halo_meta.json")))
    m = pkg.read_mtx(mtx_path("bcsstk13"))
    I, J, V = m.arrays()
    cnt = np.bincount(I, minlength=m.n_rows).astype(np.int32)
    # bcsstk13, P = 4, seg-nnz -> {0, 724, 1183, 1600, 2003}, from the reference's own 4-rank run
    assert pkg.seg_from_row_counts(cnt, "seg-nnz", 4).tolist() == [0, 724, 1183, 1600, 2003]
    assert isinstance(meta, (dict, list))
