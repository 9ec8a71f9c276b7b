"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/_ref/libuspmv_ref*.so, i.e. the
GENUINE reference compiled from /root/reference/code by oracle/Makefile (driver: ref_shim.cpp).

Used by oracle/make_golden.py (golden vectors), tests (live cross-check when the prebuilt .so is
present) and bench.py's cpu_baseline leg (kind "reference").  Never by the product.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_vp = C.c_void_p


def available(variant="colwise"):
    return os.path.exists(_path(variant))


def _path(variant):
    name = {"colwise": "libuspmv_ref.so", "rowwise": "libuspmv_ref_rowwise.so", "mpi": "libuspmv_ref_mpi.so"}
    return os.path.join(_HERE, "_ref", name[variant])


def lib(variant="colwise"):
    if variant in _LIBS:
        return _LIBS[variant]
    L = C.CDLL(_path(variant))
    L.ref_read_mtx.argtypes = [C.c_char_p]; L.ref_read_mtx.restype = _vp
    L.ref_mtx_from_coo.argtypes = [C.c_long, C.c_long, C.c_long, _i32p, _i32p, _f64p]
    L.ref_mtx_from_coo.restype = _vp
    L.ref_mtx_dims.argtypes = [_vp, C.POINTER(C.c_long)]
    L.ref_mtx_arrays.argtypes = [_vp, _i32p, _i32p, _f64p]
    L.ref_mtx_free.argtypes = [_vp]
    L.ref_mtx_to_f32.argtypes = [_vp]; L.ref_mtx_to_f32.restype = _vp
    L.ref_mtx_f32_free.argtypes = [_vp]
    L.ref_mtx_f32_dims.argtypes = [_vp, C.POINTER(C.c_long)]
    L.ref_mtx_f32_arrays.argtypes = [_vp, _i32p, _i32p, _f32p]
    if hasattr(L, "ref_equilibrate_matrix"):
        L.ref_equilibrate_matrix.argtypes = [_vp]
    L.ref_partition_precisions_dpsp.argtypes = [_vp, C.c_double, C.POINTER(_vp), C.POINTER(_vp)]
    for suf, fp in (("f64", _f64p), ("f32", _f32p)):
        f = getattr(L, f"ref_convert_to_scs_{suf}"); f.argtypes = [_vp, C.c_long, C.c_long, _vp]; f.restype = _vp
        getattr(L, f"ref_scs_meta_{suf}").argtypes = [_vp, C.POINTER(C.c_long)]
        getattr(L, f"ref_scs_arrays_{suf}").argtypes = [_vp, _i32p, _i32p, _i32p, fp, _i32p, _i32p]
        getattr(L, f"ref_permute_scs_cols_{suf}").argtypes = [_vp, _i32p]
        getattr(L, f"ref_scs_free_{suf}").argtypes = [_vp]
        getattr(L, f"ref_apply_permutation_{suf}").argtypes = [fp, fp, _i32p, C.c_int]
        getattr(L, f"ref_spmv_omp_csr_{suf}").argtypes = [C.c_long, _i32p, _i32p, fp, fp, fp]
        getattr(L, f"ref_spmv_omp_scs_{suf}").argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p, fp, fp, fp]
        getattr(L, f"ref_spmv_omp_scs_adv_{suf}").argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p, fp, fp, fp]
        getattr(L, f"ref_block_spmv_omp_scs_general_{suf}").argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p,
                                                                       fp, fp, fp, C.c_int, C.c_int]
        getattr(L, f"ref_block_spmv_omp_csr_{suf}").argtypes = [C.c_long, _i32p, _i32p, fp, fp, fp, C.c_int,
                                                               C.c_int]
    ap = [C.c_long, C.c_long, _i32p, _i32p, _i32p, _f64p, _f64p, _f64p, _i32p, _i32p, _i32p, _f32p, _f32p, _f32p]
    L.ref_spmv_omp_scs_ap_adv.argtypes = ap
    L.ref_spmv_omp_scs_ap.argtypes = ap
    L.ref_spmv_omp_csr_apdpsp.argtypes = [C.c_long, _i32p, _i32p, _f64p, _f64p, _f64p, _i32p, _i32p, _f32p,
                                          _f32p, _f32p]
    if hasattr(L, "ref_random_init_f64"):
        L.ref_random_init_f64.argtypes = [C.c_double, C.c_double, C.c_long, _f64p]
        L.ref_random_init_f32.argtypes = [C.c_double, C.c_double, C.c_long, _f32p]
    if variant == "mpi":
        L.ref_seg_work_sharing_arr.argtypes = [_vp, C.c_char_p, C.c_int, _i32p]
        L.ref_seg_local_mtx.argtypes = [_vp, _i32p, C.c_int]; L.ref_seg_local_mtx.restype = _vp
        L.ref_collect_local_needed_heri.argtypes = [_vp, _i32p, C.c_int, C.c_int, _i32p, _i32p, _i32p]
        L.ref_collect_local_needed_heri.restype = C.c_int
    _LIBS[variant] = L
    return L


def _c(a, dt=None):
    return np.ascontiguousarray(a, dtype=dt)


class RefMtx:
    """MtxData<double,int> living inside the reference library."""

    def __init__(self, handle, variant="colwise"):
        self.h, self.variant = handle, variant
        d = (C.c_long * 3)()
        lib(variant).ref_mtx_dims(handle, d)
        self.n_rows, self.n_cols, self.nnz = int(d[0]), int(d[1]), int(d[2])

    @classmethod
    def read(cls, path, variant="colwise"):
        return cls(lib(variant).ref_read_mtx(os.fsencode(path)), variant)

    @classmethod
    def from_coo(cls, n_rows, n_cols, I, J, vals, variant="colwise"):
        I = _c(I, np.int32); J = _c(J, np.int32); v = _c(vals, np.float64)
        return cls(lib(variant).ref_mtx_from_coo(n_rows, n_cols, len(I), I, J, v), variant)

    def equilibrate(self):
        """equilibrate_matrix (code/utilities.hpp:2667-2685), in place."""
        lib(self.variant).ref_equilibrate_matrix(self.h)

    def arrays(self):
        I = np.zeros(self.nnz, np.int32); J = np.zeros(self.nnz, np.int32); v = np.zeros(self.nnz, np.float64)
        lib(self.variant).ref_mtx_arrays(self.h, I, J, v)
        return I, J, v


class RefScs:
    """ScsData<VT,int> living inside the reference library."""

    def __init__(self, handle, suf, variant="colwise"):
        self.h, self.suf, self.variant = handle, suf, variant
        m = (C.c_long * 8)()
        getattr(lib(variant), f"ref_scs_meta_{suf}")(handle, m)
        (self.C, self.sigma, self.n_rows, self.n_cols, self.n_rows_padded, self.n_chunks, self.n_elements,
         self.nnz) = [int(v) for v in m]

    def arrays(self):
        dt = np.float64 if self.suf == "f64" else np.float32
        cp = np.zeros(self.n_chunks + 1, np.int32); cl = np.zeros(self.n_chunks, np.int32)
        ci = np.zeros(self.n_elements, np.int32); va = np.zeros(self.n_elements, dt)
        o2n = np.zeros(self.n_rows, np.int32); n2o = np.zeros(self.n_rows, np.int32)
        getattr(lib(self.variant), f"ref_scs_arrays_{self.suf}")(self.h, cp, cl, ci, va, o2n, n2o)
        return dict(chunk_ptrs=cp, chunk_lengths=cl, col_idxs=ci, values=va, old_to_new_idx=o2n,
                    new_to_old_idx=n2o)

    def permute_cols(self, perm):
        getattr(lib(self.variant), f"ref_permute_scs_cols_{self.suf}")(self.h, _c(perm, np.int32))


def convert_to_scs(mtx, Cc, sigma, dtype="f64", fixed_perm=None):
    """mtx: RefMtx (f64) or a raw MtxData<float,int>* handle for dtype f32 made by mtx_to_f32()."""
    L = lib(mtx.variant if isinstance(mtx, RefMtx) else "colwise")
    variant = mtx.variant if isinstance(mtx, RefMtx) else "colwise"
    fp = None
    if fixed_perm is not None:
        fixed_perm = _c(fixed_perm, np.int32)
        fp = fixed_perm.ctypes.data
    if dtype == "f64":
        return RefScs(L.ref_convert_to_scs_f64(mtx.h, Cc, sigma, fp), "f64", variant)
    h32 = L.ref_mtx_to_f32(mtx.h) if isinstance(mtx, RefMtx) else mtx
    return RefScs(L.ref_convert_to_scs_f32(h32, Cc, sigma, fp), "f32", variant)


def partition_precisions_dpsp(mtx, threshold):
    """-> (RefMtx dp, raw f32 handle sp, (I,J,vals) of sp)."""
    L = lib(mtx.variant)
    dp, sp = _vp(), _vp()
    L.ref_partition_precisions_dpsp(mtx.h, float(threshold), C.byref(dp), C.byref(sp))
    d = (C.c_long * 3)()
    L.ref_mtx_f32_dims(sp, d)
    n = int(d[2])
    I = np.zeros(n, np.int32); J = np.zeros(n, np.int32); v = np.zeros(n, np.float32)
    L.ref_mtx_f32_arrays(sp, I, J, v)
    return RefMtx(dp.value, mtx.variant), sp.value, (I, J, v)


def _suf(a):
    return "f64" if a.dtype == np.float64 else "f32"


def apply_permutation(vec, perm, variant="colwise"):
    vec = _c(vec); perm = _c(perm, np.int32)
    out = np.zeros(len(perm), vec.dtype)
    getattr(lib(variant), f"ref_apply_permutation_{_suf(vec)}")(out, vec, perm, len(perm))
    return out


def spmv_scs(kind, Cc, n_chunks, cp, cl, ci, va, x, variant="colwise"):
    """kind: 'adv' (spmv_omp_scs_adv) or 'gen' (spmv_omp_scs)."""
    va = _c(va); x = _c(x, va.dtype).copy()
    y = np.zeros(n_chunks * Cc, va.dtype)
    name = {"adv": "ref_spmv_omp_scs_adv_", "gen": "ref_spmv_omp_scs_"}[kind] + _suf(va)
    getattr(lib(variant), name)(Cc, n_chunks, _c(cp, np.int32), _c(cl, np.int32), _c(ci, np.int32), va, x, y)
    return y


def spmv_csr(n_rows, rp, ci, va, x, variant="colwise"):
    va = _c(va); x = _c(x, va.dtype).copy()
    y = np.zeros(n_rows, va.dtype)
    getattr(lib(variant), f"ref_spmv_omp_csr_{_suf(va)}")(n_rows, _c(rp, np.int32), _c(ci, np.int32), va, x, y)
    return y


def spmmv_scs_general(Cc, n_chunks, cp, cl, ci, va, X, b, ld, rowwise):
    variant = "rowwise" if rowwise else "colwise"
    va = _c(va); X = _c(X, va.dtype).ravel().copy()
    Y = np.zeros(X.size, va.dtype)
    getattr(lib(variant), f"ref_block_spmv_omp_scs_general_{_suf(va)}")(
        Cc, n_chunks, _c(cp, np.int32), _c(cl, np.int32), _c(ci, np.int32), va, X, Y, b, ld)
    return Y


def spmmv_csr(n_rows, rp, ci, va, X, b, ld, rowwise):
    variant = "rowwise" if rowwise else "colwise"
    va = _c(va); X = _c(X, va.dtype).ravel().copy()
    Y = np.zeros(X.size, va.dtype)
    getattr(lib(variant), f"ref_block_spmv_omp_csr_{_suf(va)}")(n_rows, _c(rp, np.int32), _c(ci, np.int32), va, X,
                                                                Y, b, ld)
    return Y


def spmv_scs_ap(kind, Cc, n_chunks, dp, sp, x, x_sp=None):
    """kind 'adv' | 'gen'.  dp/sp = (chunk_ptrs, chunk_lengths, col_idxs, values)."""
    x = _c(x, np.float64).copy()
    x_sp = x.astype(np.float32) if x_sp is None else _c(x_sp, np.float32).copy()
    y = np.zeros(n_chunks * Cc, np.float64); ysp = np.zeros(n_chunks * Cc, np.float32)
    f = lib().ref_spmv_omp_scs_ap_adv if kind == "adv" else lib().ref_spmv_omp_scs_ap
    f(Cc, n_chunks, _c(dp[0], np.int32), _c(dp[1], np.int32), _c(dp[2], np.int32), _c(dp[3], np.float64), x, y,
      _c(sp[0], np.int32), _c(sp[1], np.int32), _c(sp[2], np.int32), _c(sp[3], np.float32), x_sp, ysp)
    return y


def spmv_csr_apdpsp(n_rows, dp, sp, x):
    x = _c(x, np.float64).copy()
    xs = x.astype(np.float32)
    y = np.zeros(n_rows, np.float64); ys = np.zeros(n_rows, np.float32)
    lib().ref_spmv_omp_csr_apdpsp(n_rows, _c(dp[0], np.int32), _c(dp[1], np.int32), _c(dp[2], np.float64), x, y,
                                  _c(sp[0], np.int32), _c(sp[1], np.int32), _c(sp[2], np.float32), xs, ys)
    return y


# ------------------------------------------------------------------ fake-rank halo set-up (mpi variant)
def random_init(matrix_min, matrix_max, n, dtype=np.float64, variant="colwise"):
    """-rand_x 1 vector of the reference (random_init, code/utilities.hpp:880-912)."""
    out = np.empty(n, dtype)
    getattr(lib(variant), "ref_random_init_" + _suf(out))(matrix_min, matrix_max, n, out)
    return out


def seg_work_sharing_arr(mtx, method, P):
    wsa = np.zeros(P + 1, np.int32)
    lib("mpi").ref_seg_work_sharing_arr(mtx.h, method.encode(), P, wsa)
    return wsa


def seg_local_mtx(mtx, wsa, rank):
    return RefMtx(lib("mpi").ref_seg_local_mtx(mtx.h, _c(wsa, np.int32), rank), "mpi")


def collect_local_needed_heri(scs, wsa, rank, P):
    cumsum = np.zeros(P + 1, np.int32)
    flat = np.zeros(max(scs.n_cols, 1), np.int32)
    counts = np.zeros(P, np.int32)
    n = lib("mpi").ref_collect_local_needed_heri(scs.h, _c(wsa, np.int32), rank, P, cumsum, flat, counts)
    out, o = [], 0
    for p in range(P):
        out.append(flat[o:o + counts[p]].copy()); o += counts[p]
    return int(n), out, cumsum
