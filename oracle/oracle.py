"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of the parity oracle (oracle/liboracle.so).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module,
and only as the checker.  The product (ultimate-spmv_amd/) never does.

Every function forwards to the plain-C restatement in oracle/uspmv_oracle.c, whose functions
cite the reference file:line they follow.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build():
    """Compile liboracle.so (and, when /root/reference is present, oracle/_ref)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "all"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        L = _LIB
        for suf, fp in (("f64", _f64p), ("f32", _f32p)):
            getattr(L, f"orc_spmv_scs_{suf}").argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p, fp, fp, fp]
            getattr(L, f"orc_spmv_csr_{suf}").argtypes = [C.c_long, _i32p, _i32p, fp, fp, fp]
            getattr(L, f"orc_spmmv_scs_{suf}").argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p, fp, fp, fp,
                                                          C.c_int, C.c_long, C.c_int]
            getattr(L, f"orc_spmmv_csr_{suf}").argtypes = [C.c_long, _i32p, _i32p, fp, fp, fp, C.c_int, C.c_long,
                                                          C.c_int]
            getattr(L, f"orc_apply_permutation_{suf}").argtypes = [fp, fp, _i32p, C.c_long]
            getattr(L, f"orc_pack_send_buf_{suf}").argtypes = [fp, fp, _i32p, _i32p, C.c_long, C.c_long]
        L.orc_spmv_scs_ap_adv.argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p, _f64p, _i32p, _i32p, _i32p,
                                          _f32p, _f64p, _f64p]
        L.orc_spmv_scs_ap.argtypes = [C.c_long, C.c_long, _i32p, _i32p, _i32p, _f64p, _i32p, _i32p, _i32p, _f32p,
                                      _f64p, _f32p, _f64p]
        L.orc_spmv_csr_apdpsp.argtypes = [C.c_long, _i32p, _i32p, _f64p, _i32p, _i32p, _f32p, _f64p, _f64p]
        L.orc_permute_scs_cols.argtypes = [C.c_long, C.c_long, _i32p, _i32p]
        L.orc_convert_to_scs.argtypes = [C.c_long, C.c_long, _i32p, _i32p, _f64p, C.c_long, C.c_long, C.c_void_p,
                                         _i32p, _i32p, _i32p, _i32p, C.c_void_p, C.c_void_p]
        L.orc_convert_to_scs.restype = C.c_long
        L.orc_partition_precisions_dpsp.argtypes = [C.c_long, _f64p, C.c_double, _u8p]
        L.orc_partition_precisions_dpsp.restype = C.c_long
        L.orc_equilibrate_matrix.argtypes = [C.c_long, C.c_long, C.c_long, _i32p, _i32p, _f64p]
        L.orc_seg_work_sharing_arr.argtypes = [C.c_int, C.c_long, C.c_long, _i32p, C.c_int, _i32p]
        L.orc_collect_local_needed_heri.argtypes = [C.c_long, _i32p, _i32p, C.c_int, C.c_int, C.c_long, _i32p,
                                                    _i32p, _i32p]
        L.orc_collect_local_needed_heri.restype = C.c_long
    return _LIB


def _suf(a):
    return {np.dtype(np.float64): "f64", np.dtype(np.float32): "f32"}[a.dtype]


def _c(a, dt=None):
    return np.ascontiguousarray(a, dtype=dt)


class Scs:
    """Plain container for a SELL-C-sigma matrix (field names follow ScsData,
    code/classes_structs.hpp:1313-1339)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    @property
    def n_rows_padded(self):
        return self.n_chunks * self.C


def convert_to_scs(n_rows, n_cols, I, J, vals, Cc, sigma, fixed_perm=None, dtype=np.float64):
    L = lib()
    I = _c(I, np.int32); J = _c(J, np.int32); v = _c(vals, np.float64)
    nnz = len(I)
    n_chunks = (n_rows + Cc - 1) // Cc
    cp = np.zeros(n_chunks + 1, np.int32); cl = np.zeros(n_chunks, np.int32)
    o2n = np.zeros(n_rows, np.int32); n2o = np.zeros(n_rows, np.int32)
    fp = None if fixed_perm is None else _c(fixed_perm, np.int32)
    fpp = None if fp is None else fp.ctypes.data
    ne = L.orc_convert_to_scs(n_rows, nnz, I, J, v, Cc, sigma, fpp, cp, cl, o2n, n2o, None, None)
    ci = np.zeros(ne, np.int32); va = np.zeros(ne, np.float64)
    L.orc_convert_to_scs(n_rows, nnz, I, J, v, Cc, sigma, fpp, cp, cl, o2n, n2o, ci.ctypes.data, va.ctypes.data)
    return Scs(C=Cc, sigma=sigma, n_rows=n_rows, n_cols=n_cols, n_chunks=n_chunks, n_elements=int(ne), nnz=nnz,
               chunk_ptrs=cp, chunk_lengths=cl, col_idxs=ci, values=va.astype(dtype), old_to_new_idx=o2n,
               new_to_old_idx=n2o)


def permute_scs_cols(scs, perm):
    lib().orc_permute_scs_cols(scs.n_elements, scs.n_rows, scs.col_idxs, _c(perm, np.int32))


def apply_permutation(vec, perm):
    vec = _c(vec); perm = _c(perm, np.int32)
    out = np.empty(len(perm), vec.dtype)
    getattr(lib(), f"orc_apply_permutation_{_suf(vec)}")(out, vec, perm, len(perm))
    return out


def spmv_scs(Cc, n_chunks, cp, cl, ci, va, x):
    va = _c(va); x = _c(x, va.dtype)
    y = np.zeros(n_chunks * Cc, va.dtype)
    getattr(lib(), f"orc_spmv_scs_{_suf(va)}")(Cc, n_chunks, _c(cp, np.int32), _c(cl, np.int32),
                                               _c(ci, np.int32), va, x, y)
    return y


def spmv_csr(n_rows, rp, ci, va, x):
    va = _c(va); x = _c(x, va.dtype)
    y = np.zeros(n_rows, va.dtype)
    getattr(lib(), f"orc_spmv_csr_{_suf(va)}")(n_rows, _c(rp, np.int32), _c(ci, np.int32), va, x, y)
    return y


def spmmv_scs(Cc, n_chunks, cp, cl, ci, va, X, b, ld, rowwise):
    va = _c(va); X = _c(X, va.dtype)
    Y = np.zeros(X.size, va.dtype)
    getattr(lib(), f"orc_spmmv_scs_{_suf(va)}")(Cc, n_chunks, _c(cp, np.int32), _c(cl, np.int32),
                                                _c(ci, np.int32), va, X.ravel(), Y, b, ld, int(rowwise))
    return Y


def spmmv_csr(n_rows, rp, ci, va, X, b, ld, rowwise):
    va = _c(va); X = _c(X, va.dtype)
    Y = np.zeros(X.size, va.dtype)
    getattr(lib(), f"orc_spmmv_csr_{_suf(va)}")(n_rows, _c(rp, np.int32), _c(ci, np.int32), va, X.ravel(), Y, b,
                                                ld, int(rowwise))
    return Y


def spmv_scs_ap_adv(Cc, n_chunks, dp, sp, x):
    """dp, sp: tuples (chunk_ptrs, chunk_lengths, col_idxs, values)."""
    y = np.zeros(n_chunks * Cc, np.float64)
    lib().orc_spmv_scs_ap_adv(Cc, n_chunks, _c(dp[0], np.int32), _c(dp[1], np.int32), _c(dp[2], np.int32),
                              _c(dp[3], np.float64), _c(sp[0], np.int32), _c(sp[1], np.int32),
                              _c(sp[2], np.int32), _c(sp[3], np.float32), _c(x, np.float64), y)
    return y


def spmv_scs_ap(Cc, n_chunks, dp, sp, x, x_sp):
    y = np.zeros(n_chunks * Cc, np.float64)
    lib().orc_spmv_scs_ap(Cc, n_chunks, _c(dp[0], np.int32), _c(dp[1], np.int32), _c(dp[2], np.int32),
                          _c(dp[3], np.float64), _c(sp[0], np.int32), _c(sp[1], np.int32), _c(sp[2], np.int32),
                          _c(sp[3], np.float32), _c(x, np.float64), _c(x_sp, np.float32), y)
    return y


def spmv_csr_apdpsp(n_rows, dp, sp, x):
    """dp, sp: tuples (row_ptrs, col_idxs, values)."""
    y = np.zeros(n_rows, np.float64)
    lib().orc_spmv_csr_apdpsp(n_rows, _c(dp[0], np.int32), _c(dp[1], np.int32), _c(dp[2], np.float64),
                              _c(sp[0], np.int32), _c(sp[1], np.int32), _c(sp[2], np.float32),
                              _c(x, np.float64), y)
    return y


def pack_send_buf(x, perm, send_idxs, block_offset=0):
    x = _c(x); send_idxs = _c(send_idxs, np.int32)
    out = np.zeros(len(send_idxs), x.dtype)
    getattr(lib(), f"orc_pack_send_buf_{_suf(x)}")(out, x, _c(perm, np.int32), send_idxs, len(send_idxs),
                                                   block_offset)
    return out


def partition_precisions_dpsp(vals, threshold):
    vals = _c(vals, np.float64)
    m = np.zeros(len(vals), np.uint8)
    lib().orc_partition_precisions_dpsp(len(vals), vals, float(threshold), m)
    return m.astype(bool)


def equilibrate_matrix(n_rows, n_cols, I, J, vals):
    """equilibrate_matrix (code/utilities.hpp:2667-2685): returns the scaled values."""
    v = _c(vals, np.float64).copy()
    I = _c(I, np.int32); J = _c(J, np.int32)
    lib().orc_equilibrate_matrix(n_rows, n_cols, len(v), I, J, v)
    return v


def seg_work_sharing_arr(method, n_rows, I, P):
    wsa = np.zeros(P + 1, np.int32)
    I = _c(I, np.int32)
    lib().orc_seg_work_sharing_arr({"seg-rows": 0, "seg-nnz": 1}[method], n_rows, len(I), I, P, wsa)
    return wsa


def collect_local_needed_heri(col_idxs, wsa, rank, P, n_cols):
    """Rewrites col_idxs in place; returns (n_halo, recv_idxs_by_owner(list), recv_cumsum)."""
    wsa = _c(wsa, np.int32)
    flat = np.zeros(max(int(n_cols), 1), np.int32)
    counts = np.zeros(P, np.int32)
    cumsum = np.zeros(P + 1, np.int32)
    n = lib().orc_collect_local_needed_heri(len(col_idxs), col_idxs, wsa, rank, P, n_cols, flat, counts, cumsum)
    out, o = [], 0
    for p in range(P):
        out.append(flat[o:o + counts[p]].copy()); o += counts[p]
    return int(n), out, cumsum
