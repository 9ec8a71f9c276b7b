"""ctypes front-end of libuspmv.so (include/uspmv.h).

Names follow the reference's library API (code/interface.hpp, API_doc.md:7-24): MtxData -> Coo,
ScsData -> Scs, convert_to_scs, partition_precisions, apply_permutation, permute_scs_cols,
uspmv_scs_gpu, uspmv_csr_gpu.  Device vectors are torch tensors (torch is used for device memory,
streams and torch.distributed only); their raw pointers are handed to the C ABI.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

F64, F32 = 0, 1
COLWISE, ROWWISE = 0, 1
SEG_ROWS, SEG_NNZ = 0, 1

_vp = C.c_void_p
_i64 = C.c_int64
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


class UspmvError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"libuspmv status {status}: {text}")
        self.status = status


def library_path():
    # USPMV_LIB: another build of the same library (A/B measurements of kernel variants); default = the in-tree build
    return os.environ.get("USPMV_LIB") or os.path.join(_HERE, "libuspmv.so")


def build_library(force=False):
    """Compile libuspmv.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-s", "-C", _HERE, "clean"])
    subprocess.check_call(["make", "-s", "-j4", "-C", _HERE, "all"])
    return library_path()


# every symbol include/uspmv.h declares: name -> (restype, argtypes)
_SIGS = {
    "uspmv_status_string": (C.c_char_p, [C.c_int]),
    "uspmv_last_error": (C.c_char_p, []),
    "uspmv_version": (C.c_char_p, []),
    "uspmv_read_mtx": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "uspmv_coo_save": (C.c_int, [_vp, C.c_char_p]),
    "uspmv_coo_write_mtx": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "uspmv_coo_load": (C.c_int, [C.c_char_p, C.POINTER(_vp)]),
    "uspmv_coo_equilibrate": (C.c_int, [_vp]),
    "uspmv_coo_create": (C.c_int, [_i64, _i64, _i64, _vp, _vp, _vp, C.POINTER(_vp)]),
    "uspmv_coo_dims": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_coo_arrays": (C.c_int, [_vp, C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_f64p)]),
    "uspmv_coo_free": (None, [_vp]),
    "uspmv_gen_stencil27": (C.c_int, [_i64, _i64, _i64, C.c_int, C.c_uint64, C.c_double, _i64, _i64, C.POINTER(_vp)]),
    "uspmv_gen_banded_random": (C.c_int, [_i64, C.c_int, _i64, C.c_uint64, C.c_double, _i64, _i64, C.POINTER(_vp)]),
    "uspmv_gen_kkt": (C.c_int, [_i64, C.c_uint64, _i64, _i64, C.POINTER(_vp)]),
    "uspmv_gen_kkt_row_counts": (C.c_int, [_i64, _i64, _i64, _vp]),
    "uspmv_convert_to_scs": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, C.POINTER(_vp)]),
    "uspmv_scs_meta": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_scs_dtype": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "uspmv_scs_arrays": (C.c_int, [_vp, C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_vp),
                                   C.POINTER(_i32p), C.POINTER(_i32p)]),
    "uspmv_scs_col_idxs_mut": (C.c_int, [_vp, C.POINTER(_i32p)]),
    "uspmv_permute_scs_cols": (C.c_int, [_vp, _vp]),
    "uspmv_scs_free": (None, [_vp]),
    "uspmv_apply_permutation": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int]),
    "uspmv_partition_precisions": (C.c_int, [_vp, C.c_double, C.POINTER(_vp), C.POINTER(_vp)]),
    "uspmv_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "uspmv_set_device": (C.c_int, [C.c_int]),
    "uspmv_stream_synchronize": (C.c_int, [_vp]),
    "uspmv_dmat_upload": (C.c_int, [_vp, C.POINTER(_vp)]),
    "uspmv_convert_to_scs_device": (C.c_int, [_vp, _i64, _i64, C.c_int, _vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp)]),
    "uspmv_dmat_download": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "uspmv_dmat_wrap": (C.c_int, [_i64, _i64, _i64, C.c_int, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "uspmv_dmat_free": (None, [_vp]),
    "uspmv_dmat_set_crs": (C.c_int, [_vp, C.c_int]),
    "uspmv_dmat_optimize": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_spmv": (C.c_int, [_vp, _vp, _vp, _vp]),
    "uspmv_spmv_chunks": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "uspmv_spmv_tiles": (C.c_int, [_vp, _vp, _i64, _vp, _vp, _vp]),
    "uspmv_dmat_tile_rows": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "uspmv_dmat_index_bits": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "uspmv_dmat_optimize_device": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_optimize_device_ap": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_plan_download": (C.c_int, [_vp, C.POINTER(_i64), _vp, _vp, _vp, _vp]),
    "uspmv_dmat_optimize_block": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_optimize_block_device": (C.c_int, [_vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_optimize_ap": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_optimize_sweep": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_optimize_sweep_ap": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_plan_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_block_plan_info": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_dmat_block_plan_digest": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "uspmv_dmat_block_plan_staged": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_dmat_stream_info": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_dmat_plan_granularity": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "uspmv_dmat_plan_rows_dealt": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "uspmv_dmat_optimize_sweep_device": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_dmat_sweep_plan_digest": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.POINTER(_i64)]),
    "uspmv_spmmv": (C.c_int, [_vp, _vp, _vp, C.c_int, _i64, C.c_int, _vp]),
    "uspmv_spmv_ap": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "uspmv_spmv_ap_generic": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "uspmv_scs_gpu_f64": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uspmv_scs_gpu_f32": (C.c_int, [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uspmv_csr_gpu_f64": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uspmv_csr_gpu_f32": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "uspmv_peek_i64": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_peek_i32": (C.c_int, [_vp, C.POINTER(C.c_int32)]),
    "uspmv_apply_permutation_dev": (C.c_int, [_vp, _vp, _vp, _i64, C.c_int, _vp]),
    "uspmv_raw_plan_cache_clear": (None, []),
    "uspmv_set_tuning": (C.c_int, [C.c_char_p, C.c_int]),
    "uspmv_get_tuning": (C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    "uspmv_seg_work_sharing_arr": (C.c_int, [_vp, C.c_int, C.c_int, _vp]),
    "uspmv_seg_local_coo": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(_vp)]),
    "uspmv_graph_partition": (C.c_int, [_vp, C.c_int, _vp]),
    "uspmv_read_partition": (C.c_int, [C.c_char_p, _i64, C.c_int, _vp]),
    "uspmv_coo_apply_partition": (C.c_int, [_vp, C.c_int, _vp, C.POINTER(_vp), _vp, _vp]),
    "uspmv_halo_discover": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.POINTER(_vp)]),
    "uspmv_halo_meta": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(_i32p), C.POINTER(_i32p), C.POINTER(_i32p)]),
    "uspmv_halo_free": (None, [_vp]),
    "uspmv_scs_chunk_classes": (C.c_int, [_vp, _i64, C.POINTER(C.c_uint8), C.POINTER(C.c_int32)]),
    "uspmv_scs_split_chunks": (C.c_int, [_vp, _i64, C.POINTER(_i32p), C.POINTER(_i64), C.POINTER(_i32p),
                                         C.POINTER(_i64)]),
    "uspmv_free": (None, [_vp]),
    "uspmv_pack_send_buf": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _vp, C.c_int, _vp]),
    "uspmv_seg_from_row_counts": (C.c_int, [_vp, _i64, C.c_int, C.c_int, _vp]),
    "uspmv_gen_stencil27_row_counts": (C.c_int, [_i64, _i64, _i64, C.c_int, _i64, _i64, _vp]),
    "uspmv_comm_unique_id": (C.c_int, [_vp]),
    "uspmv_dist_create": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _i64, C.c_int, C.POINTER(_vp)]),
    "uspmv_dist_create_from_coo": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _i64, _i64, C.c_int, C.c_int, C.POINTER(_vp)]),
    "uspmv_dist_info": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_dist_spmmv_info": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_dist_pad_info": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_convert_to_scs_device_from_arrays": (C.c_int, [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _i64, C.c_int, _vp, C.c_int, C.c_int, _vp,
                                                          C.POINTER(_vp), _vp, _vp, C.POINTER(_vp)]),
    "uspmv_dmat_plan_addresses": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "uspmv_dmat_optimize_block_sweep": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(_i64), C.POINTER(_i64)]),
    "uspmv_spmmv_x_prepared": (C.c_int, [_vp, _vp, C.c_int, _i64, _vp]),
    "uspmv_spmmv_x_release": (C.c_int, [_vp]),
    "uspmv_dmat_meta": (C.c_int, [_vp, C.POINTER(_i64)]),
    "uspmv_dist_comm_count": (C.c_int, [_vp, C.POINTER(C.c_int)]),
    "uspmv_dist_autotune": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _i32p, _vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "uspmv_dist_check_reference": (C.c_int, [_vp, _i32p, C.c_int, C.c_int, C.c_int, _vp]),
    "uspmv_dist_parts": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "uspmv_dist_set_overlap": (C.c_int, [_vp, C.c_int]),
    "uspmv_dist_set_no_pack": (C.c_int, [_vp, C.c_int]),
    "uspmv_dist_spmv": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "uspmv_dist_run": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp]),
    "uspmv_dist_spmmv": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "uspmv_dist_barrier": (C.c_int, [_vp, _vp]),
    "uspmv_dist_allreduce_max": (C.c_int, [_vp, C.POINTER(C.c_double), _vp]),
    "uspmv_dist_allgather_i64": (C.c_int, [_vp, _i64, C.POINTER(_i64), _vp]),
    "uspmv_dist_free": (None, [_vp]),
    "uspmv_dist_create_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _i64, _vp, _i64, C.c_int, _vp, C.POINTER(_vp)]),
    "uspmv_dist_create_from_coo_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _i64, _i64, C.c_int, C.c_int, _vp, C.POINTER(_vp)]),
    "uspmv_dist_comm_plan": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.POINTER(_i64)), C.POINTER(_i32p), C.POINTER(C.POINTER(_i64))]),
    "uspmv_dist_set_option": (C.c_int, [_vp, C.c_char_p, C.c_int]),
    "uspmv_dist_check": (C.c_int, [_vp, _vp, _vp, _vp, _vp, C.c_int, _vp, C.POINTER(_i64), C.POINTER(C.c_double)]),
    "uspmv_runtime_versions": (C.c_int, [C.POINTER(C.c_int)]),
    "uspmv_debug_backtrace_on_crash": (C.c_int, [C.c_int]),
    "uspmv_hostcomm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_double, C.POINTER(_vp)]),
    "uspmv_hostcomm_info": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
    "uspmv_hostcomm_barrier": (C.c_int, [_vp]),
    "uspmv_hostcomm_abort": (C.c_int, [_vp]),
    "uspmv_hostcomm_bcast": (C.c_int, [_vp, _vp, _i64, C.c_int]),
    "uspmv_hostcomm_allgather": (C.c_int, [_vp, _vp, _vp, _i64]),
    "uspmv_hostcomm_alltoallv": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "uspmv_hostcomm_allreduce_max_f64": (C.c_int, [_vp, C.POINTER(C.c_double)]),
    "uspmv_hostcomm_transport": (C.c_int, [_vp, _vp]),
    "uspmv_hostcomm_free": (None, [_vp]),
    "uspmv_comm_plan_create": (C.c_int, [_vp, _vp, C.POINTER(_vp)]),
    "uspmv_comm_plan_meta": (C.c_int, [_vp, C.POINTER(_i64), C.POINTER(C.POINTER(_i64)), C.POINTER(_i32p), C.POINTER(C.POINTER(_i64))]),
    "uspmv_comm_plan_free": (None, [_vp]),
    "uspmv_stream_copy": (C.c_int, [_vp, _vp, _i64, _vp]),
    "uspmv_stream_triad": (C.c_int, [_vp, _vp, _vp, C.c_double, _i64, _vp]),
    "uspmv_stream_read": (C.c_int, [_vp, _i64, _vp, _vp]),
    "uspmv_stream_gather_lines": (C.c_int, [_vp, _i64, _i64, _i64, C.c_int, _vp, _vp, C.POINTER(_i64)]),
    "uspmv_time_launches": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _vp, _vp, _i64, C.c_int, _i64, C.c_int, _vp,
                                      C.POINTER(C.c_double)]),
}


def lib():
    """Load libuspmv.so.  Fails loudly (no fallback) when the extension has not been built."""
    global _LIB
    if _LIB is None:
        # torch (when installed) must be imported BEFORE libuspmv.so is loaded: torch ships its
        # own HIP/HSA runtime and publishes it RTLD_GLOBAL; loaded first, it is the one runtime the
        # whole process (torch + libuspmv kernels + RCCL) shares.  Loaded second, the process ends up
        # with two HSA runtimes and the later one finds no device.
        if not os.environ.get("USPMV_NO_TORCH"):   # (USPMV_NO_TORCH=1: a torch-free process binds to the system's /opt/rocm runtime)
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        path = library_path()
        if not os.path.exists(path):
            raise UspmvError(-1, f"{path} is missing: build it with __graft_entry__.build() "
                                 f"(make -C ultimate-spmv_amd); there is no CPU fallback")
        L = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            f = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            f.restype, f.argtypes = res, args
        _LIB = L
        if os.environ.get("USPMV_BACKTRACE"):
            L.uspmv_debug_backtrace_on_crash(1)
    return _LIB


def _ck(rc):
    if rc != 0:
        raise UspmvError(rc, lib().uspmv_last_error().decode())


def _np_ptr(a):
    return a.ctypes.data if a is not None else None


def _view(ptr, n, dtype, owner=None):
    """numpy view of library-owned memory; `owner` (the Python wrapper whose handle owns the
    memory) is pinned on the ctypes pointer, which the array keeps alive through .base."""
    if n == 0:
        return np.zeros(0, dtype)
    ptr._uspmv_owner = owner
    return np.ctypeslib.as_array(ptr, shape=(n,)).view(dtype)


# ---------------------------------------------------------------------------------------- COO
class Coo:
    """Host COO matrix (MtxData, code/classes_structs.hpp:1169-1238)."""

    def __init__(self, handle):
        self.h = handle
        a, b, c = _i64(), _i64(), _i64()
        _ck(lib().uspmv_coo_dims(handle, C.byref(a), C.byref(b), C.byref(c)))
        self.n_rows, self.n_cols, self.nnz = a.value, b.value, c.value

    @classmethod
    def from_arrays(cls, n_rows, n_cols, I, J, values):
        I = np.ascontiguousarray(I, np.int32); J = np.ascontiguousarray(J, np.int32)
        v = np.ascontiguousarray(values, np.float64)
        h = _vp()
        _ck(lib().uspmv_coo_create(n_rows, n_cols, len(I), _np_ptr(I), _np_ptr(J), _np_ptr(v), C.byref(h)))
        return cls(h)

    def equilibrate(self):
        """equilibrate_matrix (-equilibrate 1, code/utilities.hpp:2667-2685), in place."""
        _ck(lib().uspmv_coo_equilibrate(self.h))

    def save(self, path):
        """Binary cache of this matrix (uspmv_coo_save)."""
        _ck(lib().uspmv_coo_save(self.h, os.fsencode(path)))

    @classmethod
    def load(cls, path):
        h = _vp()
        _ck(lib().uspmv_coo_load(os.fsencode(path), C.byref(h)))
        return cls(h)

    def write_mtx(self, path, symmetric=False):
        """MatrixMarket file of this matrix (uspmv_coo_write_mtx)"""
        _ck(lib().uspmv_coo_write_mtx(self.h, os.fsencode(path), int(bool(symmetric))))

    def arrays(self):
        """(I, J, values) as numpy views borrowed from the library (valid while self lives)."""
        I, J, v = _i32p(), _i32p(), _f64p()
        _ck(lib().uspmv_coo_arrays(self.h, C.byref(I), C.byref(J), C.byref(v)))
        return (_view(I, self.nnz, np.int32, self), _view(J, self.nnz, np.int32, self),
                _view(v, self.nnz, np.float64, self))

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.uspmv_coo_free(self.h)
            self.h = None


def read_mtx(path):
    h = _vp()
    _ck(lib().uspmv_read_mtx(os.fsencode(path), C.byref(h)))
    return Coo(h)


def gen_stencil27(nx, ny, nz, dof=1, seed=0x5EED, magnitude_decades=0.0, row_begin=0, row_end=None):
    n = nx * ny * nz * dof
    h = _vp()
    _ck(lib().uspmv_gen_stencil27(nx, ny, nz, dof, seed, magnitude_decades, row_begin,
                                  n if row_end is None else row_end, C.byref(h)))
    return Coo(h)


def gen_banded_random(n, nnz_per_row, band, seed=0x5EED, magnitude_decades=0.0, row_begin=0, row_end=None):
    h = _vp()
    _ck(lib().uspmv_gen_banded_random(n, nnz_per_row, band, seed, magnitude_decades, row_begin, n if row_end is None else row_end, C.byref(h)))
    return Coo(h)


def graph_partition(coo, P):
    """part id per row from the built-in -seg_metis stand-in (uspmv_graph_partition)"""
    part = np.empty(coo.n_rows, np.int32)
    _ck(lib().uspmv_graph_partition(coo.h, P, _np_ptr(part)))
    return part


def read_partition(path, n_rows, P):
    part = np.empty(n_rows, np.int32)
    _ck(lib().uspmv_read_partition(str(path).encode(), n_rows, P, _np_ptr(part)))
    return part


def apply_partition(coo, P, part):
    """(permuted matrix, wsa[P+1], perm with new row r = old row perm[r]): the reference's seg-metis post-processing"""
    part = np.ascontiguousarray(part, np.int32)
    h = _vp()
    wsa = np.empty(P + 1, np.int32)
    perm = np.empty(coo.n_rows, np.int32)
    _ck(lib().uspmv_coo_apply_partition(coo.h, P, _np_ptr(part), C.byref(h), _np_ptr(wsa), _np_ptr(perm)))
    return Coo(h), wsa, perm


def gen_kkt(N, seed=0x5EED, row_begin=0, row_end=None):
    """nlpkkt-class KKT matrix [H A^T; A 0] on an N^3 grid with boundary controls (n = 2 N^3 + 6 N^2), uspmv_gen_kkt."""
    n = 2 * N ** 3 + 6 * N ** 2
    h = _vp()
    _ck(lib().uspmv_gen_kkt(N, seed, row_begin, n if row_end is None else row_end, C.byref(h)))
    return Coo(h)


def gen_kkt_row_counts(N, row_begin=0, row_end=None):
    n = 2 * N ** 3 + 6 * N ** 2
    row_end = n if row_end is None else row_end
    out = np.empty(row_end - row_begin, np.int32)
    _ck(lib().uspmv_gen_kkt_row_counts(N, row_begin, row_end, _np_ptr(out)))
    return out


# ---------------------------------------------------------------------------------------- SCS
class Scs:
    """Host SELL-C-sigma matrix (ScsData, code/classes_structs.hpp:1313-1339)."""

    def __init__(self, handle):
        self.h = handle
        m = (_i64 * 8)()
        _ck(lib().uspmv_scs_meta(handle, m))
        (self.C, self.sigma, self.n_rows, self.n_cols, self.n_rows_padded, self.n_chunks, self.n_elements,
         self.nnz) = [int(v) for v in m]
        d = C.c_int()
        _ck(lib().uspmv_scs_dtype(handle, C.byref(d)))
        self.dtype = d.value

    @property
    def np_dtype(self):
        return np.float64 if self.dtype == F64 else np.float32

    def arrays(self):
        """dict of numpy views borrowed from the library (valid while self lives)."""
        cp, cl, ci, o2n, n2o = _i32p(), _i32p(), _i32p(), _i32p(), _i32p()
        va = _vp()
        _ck(lib().uspmv_scs_arrays(self.h, C.byref(cp), C.byref(cl), C.byref(ci), C.byref(va), C.byref(o2n),
                                   C.byref(n2o)))
        vp = C.cast(va, C.POINTER(C.c_double if self.dtype == F64 else C.c_float))
        full = bool(ci) or self.n_elements == 0        # layout-only structs (convert_to_scs_device) carry no host entries
        return dict(chunk_ptrs=_view(cp, self.n_chunks + 1, np.int32, self),
                    chunk_lengths=_view(cl, self.n_chunks, np.int32, self),
                    col_idxs=_view(ci, self.n_elements, np.int32, self) if full else None,
                    values=_view(vp, self.n_elements, self.np_dtype, self) if full else None,
                    old_to_new_idx=_view(o2n, self.n_rows, np.int32, self),
                    new_to_old_idx=_view(n2o, self.n_rows, np.int32, self))

    def split_chunks(self, n_local):
        """(interior chunk ids, boundary chunk ids): boundary chunks touch a column >= n_local."""
        a, b = _i32p(), _i32p()
        na, nb = _i64(), _i64()
        _ck(lib().uspmv_scs_split_chunks(self.h, n_local, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        ia = _view(a, na.value, np.int32).copy()
        ib = _view(b, nb.value, np.int32).copy()
        lib().uspmv_free(a); lib().uspmv_free(b)
        return ia, ib

    def chunk_classes(self, n_local):
        """(classes[n_chunks], pad_col): 0 no halo column, 1 halo only through +0.0 padding on column pad_col, 2 other halo references"""
        cls = np.zeros(max(self.n_chunks, 1), np.uint8)
        pc = C.c_int32(-1)
        _ck(lib().uspmv_scs_chunk_classes(self.h, n_local, cls.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(pc)))
        return cls[:self.n_chunks], pc.value

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.uspmv_scs_free(self.h)
            self.h = None


def convert_to_scs(coo, Cc, sigma, dtype=F64, fixed_permutation=None):
    fp = None if fixed_permutation is None else np.ascontiguousarray(fixed_permutation, np.int32)
    h = _vp()
    _ck(lib().uspmv_convert_to_scs(coo.h, Cc, sigma, dtype, _np_ptr(fp), C.byref(h)))
    return Scs(h)


def permute_scs_cols(scs, perm):
    perm = np.ascontiguousarray(perm, np.int32)
    _ck(lib().uspmv_permute_scs_cols(scs.h, _np_ptr(perm)))


def apply_permutation(vec, perm):
    """out[i] = vec[perm[i]] on the host (numpy float64 / float32)."""
    vec = np.ascontiguousarray(vec); perm = np.ascontiguousarray(perm, np.int32)
    dt = {np.dtype(np.float64): F64, np.dtype(np.float32): F32}[vec.dtype]
    out = np.empty(len(perm), vec.dtype)
    _ck(lib().uspmv_apply_permutation(_np_ptr(out), _np_ptr(vec), _np_ptr(perm), len(perm), dt))
    return out


def partition_precisions(coo, threshold_1):
    dp, sp = _vp(), _vp()
    _ck(lib().uspmv_partition_precisions(coo.h, threshold_1, C.byref(dp), C.byref(sp)))
    return Coo(dp), Coo(sp)


# ---------------------------------------------------------------------------------------- halo set-up
def seg_work_sharing_arr(coo, method, P):
    wsa = np.zeros(P + 1, np.int32)
    m = {"seg-rows": SEG_ROWS, "seg-nnz": SEG_NNZ}.get(method, method)
    _ck(lib().uspmv_seg_work_sharing_arr(coo.h, m, P, _np_ptr(wsa)))
    return wsa


def seg_local_coo(coo, wsa, rank):
    wsa = np.ascontiguousarray(wsa, np.int32)
    h = _vp()
    _ck(lib().uspmv_seg_local_coo(coo.h, _np_ptr(wsa), rank, C.byref(h)))
    return Coo(h)


def seg_from_row_counts(row_nnz, method, P):
    """work_sharing_arr from per-row entry counts alone (uspmv_seg_from_row_counts)."""
    rn = np.ascontiguousarray(row_nnz, np.int32)
    wsa = np.zeros(P + 1, np.int32)
    m = {"seg-rows": SEG_ROWS, "seg-nnz": SEG_NNZ}.get(method, method)
    _ck(lib().uspmv_seg_from_row_counts(_np_ptr(rn), len(rn), m, P, _np_ptr(wsa)))
    return wsa


def gen_stencil27_row_counts(nx, ny, nz, dof=1, row_begin=0, row_end=None):
    n = nx * ny * nz * dof
    row_end = n if row_end is None else row_end
    out = np.empty(row_end - row_begin, np.int32)
    _ck(lib().uspmv_gen_stencil27_row_counts(nx, ny, nz, dof, row_begin, row_end, _np_ptr(out)))
    return out


def comm_unique_id():
    """128 bytes of an RCCL unique id (rank 0 makes it, every rank of the communicator gets a copy)."""
    buf = (C.c_ubyte * 128)()
    _ck(lib().uspmv_comm_unique_id(buf))
    return bytes(buf)


class _BorrowedScs(Scs):
    """Scs view of a struct owned by another library object (kept alive through `owner`)."""

    def __init__(self, handle, owner):
        super().__init__(handle)
        self._owner = owner

    def __del__(self):
        self.h = None


class Transport(C.Structure):
    """uspmv_transport_t (include/uspmv.h, L4a)"""
    _fields_ = [("ctx", _vp), ("rank", C.c_int), ("size", C.c_int), ("alltoallv", _vp), ("allgather", _vp), ("barrier", _vp)]


class DistOptions(C.Structure):
    """uspmv_dist_options_t"""
    _fields_ = [("transport", C.POINTER(Transport)), ("exchange", C.c_int)]


EXCHANGE_RCCL, EXCHANGE_HOST = 0, 1


class HostComm:
    """Host communicator of one node (uspmv_hostcomm_*, host/hostcomm.cpp): barrier / broadcast / all-gather / all-to-all-v between
    the ranks of a job through a memory-mapped segment.  No GPU involved."""

    def __init__(self, job, rank, size, timeout_s=120.0):
        h = _vp()
        _ck(lib().uspmv_hostcomm_create(str(job).encode(), rank, size, float(timeout_s), C.byref(h)))
        self.h, self.rank, self.size = h, rank, size
        self.transport = Transport()
        _ck(lib().uspmv_hostcomm_transport(h, C.byref(self.transport)))

    @property
    def nonce(self):
        n = C.c_uint64()
        _ck(lib().uspmv_hostcomm_info(self.h, None, None, C.byref(n)))
        return n.value

    def barrier(self):
        _ck(lib().uspmv_hostcomm_barrier(self.h))

    def abort(self):
        lib().uspmv_hostcomm_abort(self.h)

    def bcast(self, arr, root=0):
        """in place on a contiguous numpy array (same shape and dtype on every rank)"""
        assert arr.flags["C_CONTIGUOUS"]
        _ck(lib().uspmv_hostcomm_bcast(self.h, _np_ptr(arr), arr.nbytes, root))
        return arr

    def allgather(self, arr):
        arr = np.ascontiguousarray(arr)
        out = np.empty((self.size,) + arr.shape, arr.dtype)
        _ck(lib().uspmv_hostcomm_allgather(self.h, _np_ptr(arr), _np_ptr(out), arr.nbytes))
        return out

    def alltoallv(self, send, send_counts, recv_counts):
        """element counts per peer; returns the received array"""
        send = np.ascontiguousarray(send)
        isz = send.dtype.itemsize
        so = np.zeros(self.size + 1, np.int64); so[1:] = np.cumsum(send_counts) * isz
        ro = np.zeros(self.size + 1, np.int64); ro[1:] = np.cumsum(recv_counts) * isz
        recv = np.empty(int(ro[-1] // isz), send.dtype)
        _ck(lib().uspmv_hostcomm_alltoallv(self.h, _np_ptr(send) if send.size else None, _np_ptr(so), _np_ptr(recv) if recv.size else None, _np_ptr(ro)))
        return recv

    def allreduce_max(self, v):
        d = C.c_double(float(v))
        _ck(lib().uspmv_hostcomm_allreduce_max_f64(self.h, C.byref(d)))
        return d.value

    def close(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.uspmv_hostcomm_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


def _plan_meta(fn, handle, P):
    n = _i64()
    so, ro = C.POINTER(_i64)(), C.POINTER(_i64)()
    si = _i32p()
    _ck(fn(handle, C.byref(n), C.byref(so), C.byref(si), C.byref(ro)))
    send_off = np.ctypeslib.as_array(so, shape=(P + 1,)).copy()
    recv_off = np.ctypeslib.as_array(ro, shape=(P + 1,)).copy()
    send_idxs = np.ctypeslib.as_array(si, shape=(n.value,)).copy() if n.value else np.zeros(0, np.int32)
    return n.value, send_off, send_idxs, recv_off


class CommPlan:
    """uspmv_comm_plan_create: what this rank must send to whom (the reference's comm_send_idxs / send cumsums,
    code/mpi_funcs.hpp:117-232), derived collectively over a transport.  No GPU involved."""

    def __init__(self, transport, halo):
        h = _vp()
        _ck(lib().uspmv_comm_plan_create(C.byref(transport), halo.h, C.byref(h)))
        self.h = h
        self.n_send, self.send_off, self.send_idxs, self.recv_off = _plan_meta(lib().uspmv_comm_plan_meta, h, halo.P)

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.uspmv_comm_plan_free(self.h)
            self.h = None


def dist_check_reference(local, wsa, rank, P, dtype=F64):
    """y of the local rows (original order) for x_global[j] = 1 + 1e-3 (j mod 1000): entry-ordered FMA chains on the host (uspmv_dist_check_reference)"""
    w = np.ascontiguousarray(wsa, np.int32)
    y = np.zeros(local.n_rows, np.float64 if dtype == F64 else np.float32)
    _ck(lib().uspmv_dist_check_reference(local.h, w.ctypes.data_as(_i32p), int(rank), int(P), int(dtype), y.ctypes.data_as(_vp)))
    return y


def runtime_versions():
    """(HIP build, HIP runtime, RCCL build, RCCL runtime) version codes of libuspmv.so in this process"""
    v = (C.c_int * 4)()
    _ck(lib().uspmv_runtime_versions(v))
    return tuple(int(a) for a in v)


class DistNative:
    """The distributed SpMV object of the C ABI (uspmv_dist_*, csrc/uspmv_dist_api.hip): partition block `rank` of P on an RCCL
    communicator of comm_size ranks (comm_size == P, or 1 = loopback).  Everything per step happens in C++.  use_graph (one
    hipGraphLaunch per step) works in processes bound to the system's RCCL (the uspmv CLI); under torch's bundled RCCL / HIP runtime
    the capture of an RCCL group crashes in hipStreamEndCapture, so it is off by default here."""

    def __init__(self, local_coo, wsa, C_, sigma, rank, P, comm_id=None, comm_rank=None, comm_size=None, dtype=F64, tlc=True,
                 hostcomm=None, host_exchange=False):
        """hostcomm: a HostComm that carries the set-up exchanges (default: the RCCL communicator made from comm_id);
        host_exchange=True additionally stages the per-step halo exchange through it (USPMV_EXCHANGE_HOST: no RCCL communicator,
        P processes may share one GPU)."""
        import torch
        self.rank, self.P = rank, P
        wsa = np.ascontiguousarray(wsa, np.int32)
        self._wsa = wsa
        h = _vp()
        idbuf = (C.c_ubyte * 128).from_buffer_copy(comm_id) if comm_id is not None else None
        self._hostcomm = hostcomm
        opt = None
        if hostcomm is not None:
            self._opt = DistOptions(C.pointer(hostcomm.transport), EXCHANGE_HOST if host_exchange else EXCHANGE_RCCL)
            opt = C.byref(self._opt)
        elif host_exchange:
            raise ValueError("host_exchange needs a HostComm")
        _ck(lib().uspmv_dist_create_from_coo_ex(idbuf, rank if comm_rank is None else comm_rank, P if comm_size is None else comm_size,
                                                rank, P, local_coo.h, _np_ptr(wsa), C_, sigma, dtype, int(bool(tlc)), opt, C.byref(h)))
        self.h = h
        s, a, hl = _vp(), _vp(), _vp()
        _ck(lib().uspmv_dist_parts(h, C.byref(s), C.byref(a), C.byref(hl)))
        self.scs = _BorrowedScs(s, self)
        self._A = a
        self.tdtype = torch.float64 if dtype == F64 else torch.float32
        self.stream = torch.cuda.Stream()
        self._refresh()
        arr = self.scs.arrays()
        self.old_to_new = arr["old_to_new_idx"].copy()
        self.new_to_old = arr["new_to_old_idx"].copy()

    def _refresh(self):
        m = (_i64 * 12)()
        _ck(lib().uspmv_dist_info(self.h, m))
        (self.n_local, self.n_halo, self.padded_vec_size, self.n_send, self.n_interior, self.n_boundary, tiles, self.n_rows_padded,
         loop, graph, self.graph_launches, self.eager_steps) = [int(v) for v in m]
        self.use_tiles, self.loopback, self.graph_captured = bool(tiles), bool(loop), bool(graph)

    def plan_info(self):
        k, a, b = C.c_int(), _i64(), _i64()
        _ck(lib().uspmv_dmat_plan_info(self._A, C.byref(k), C.byref(a), C.byref(b)))
        return k.value, a.value, b.value

    def new_x(self, x_local_orig):
        import torch
        xp = apply_permutation(np.ascontiguousarray(x_local_orig, self.scs.np_dtype), self.new_to_old)
        x = torch.zeros(self.padded_vec_size, dtype=self.tdtype, device="cuda")
        x[:self.n_local] = torch.from_numpy(xp).cuda()
        return x

    def new_y(self):
        import torch
        return torch.zeros(self.padded_vec_size, dtype=self.tdtype, device="cuda")

    def y_to_original_order(self, y):
        return apply_permutation(y.detach().cpu().numpy(), self.old_to_new)

    def set_overlap(self, on):
        _ck(lib().uspmv_dist_set_overlap(self.h, int(bool(on))))

    def set_option(self, key, value):
        _ck(lib().uspmv_dist_set_option(self.h, key.encode(), int(value)))

    def comm_plan(self):
        """(n_send, send_off[P+1], send_idxs, recv_off[P+1]) of the object's exchange plan"""
        return _plan_meta(lib().uspmv_dist_comm_plan, self.h, self.P)

    def check(self, local_coo, x, y, use_graph=False):
        """uspmv_dist_check: one step with x_global[j] = 1 + 1e-3 (j mod 1000), y of the local rows compared bitwise with the
        entry-ordered FMA chains of the block's COO.  Overwrites x and y.  Returns (mismatching rows, checksum of the local y)."""
        bad, cs = _i64(), C.c_double()
        self._order(x, y)
        _ck(lib().uspmv_dist_check(self.h, local_coo.h, _np_ptr(self._wsa), _dp(x), _dp(y), int(bool(use_graph)), self.stream.cuda_stream,
                                   C.byref(bad), C.byref(cs)))
        return bad.value, cs.value

    def _order(self, *tensors):
        """the object's stream is non-blocking: make it wait for whatever torch's current stream still does to the tensors
        (zero fills, copies of new_x / new_y), and tell the caching allocator the tensors are used on it"""
        import torch
        self.stream.wait_stream(torch.cuda.current_stream())
        for t in tensors:
            t.record_stream(self.stream)

    def spmv(self, x, y, comm_halos=True):
        """one eager step on the object's stream"""
        self._order(x, y)
        _ck(lib().uspmv_dist_spmv(self.h, _dp(x), _dp(y), int(bool(comm_halos)), self.stream.cuda_stream))
        return y

    def spmmv(self, X, Y, b, layout=COLWISE, mode=0, comm_halos=True):
        """Y = A X for b vectors of leading dimension padded_vec_size; mode 0 bulkvec | 1 multivec | 2 singlevec (uspmv_dist_spmmv)."""
        self._order(X, Y)
        _ck(lib().uspmv_dist_spmmv(self.h, _dp(X), _dp(Y), int(b), int(layout), int(mode), int(bool(comm_halos)), self.stream.cuda_stream))
        return Y

    STEP_FORMS = ("overlap", "plain", "pad", "fused")

    def comm_count(self):
        """ranks of the RCCL communicator the exchange runs on (ncclCommCount); 0 with the host-staged exchange"""
        n = C.c_int(0)
        _ck(lib().uspmv_dist_comm_count(self.h, C.byref(n)))
        return n.value

    def autotune(self, x, y, use_graph=False, local=None, wsa=None, all_forms=False):
        """Time the arrangements of the step on this machine and keep the fastest (uspmv_dist_autotune; COLLECTIVE: every rank calls it).
        Candidates: overlap | plain; all_forms adds pad (and fused for eager steps).  Returns (name of the chosen form, {form: ms per
        step}); with `local` (the block's Coo) and `wsa` a pad / fused winner must pass the bitwise self-check."""
        self._order(x, y)
        _ck(lib().uspmv_dist_set_option(self.h, b"autotune_all", int(bool(all_forms))))
        form = C.c_int(0)
        ms = (C.c_double * 4)()
        w = None if wsa is None else np.ascontiguousarray(wsa, np.int32)
        _ck(lib().uspmv_dist_autotune(self.h, _dp(x), _dp(y), int(bool(use_graph)), None if local is None else local.h,
                                      None if w is None else w.ctypes.data_as(_i32p), self.stream.cuda_stream, C.byref(form), ms))
        return self.STEP_FORMS[form.value], {self.STEP_FORMS[k]: float(ms[k]) for k in range(4) if ms[k] != 0}

    def pad_info(self):
        """dict(pad_tiles, real_boundary_tiles, pad_col, reruns) (uspmv_dist_pad_info)"""
        m = (C.c_int64 * 4)()
        _ck(lib().uspmv_dist_pad_info(self.h, m))
        return dict(zip(("pad_tiles", "real_boundary_tiles", "pad_col", "reruns"), (int(v) for v in m)))

    def spmmv_info(self):
        """dict(two_part, one_part, plan_b, plan_tiles, plan_boundary_tiles, parts) (uspmv_dist_spmmv_info)"""
        m = (C.c_int64 * 6)()
        _ck(lib().uspmv_dist_spmmv_info(self.h, m))
        return dict(zip(("two_part", "one_part", "plan_b", "plan_tiles", "plan_boundary_tiles", "parts"), (int(v) for v in m)))

    def new_X(self, X_local_orig, b, layout=COLWISE):
        import torch
        ld = self.padded_vec_size
        X = torch.zeros(b * ld, dtype=self.tdtype, device="cuda")
        for v in range(b):
            xp = torch.from_numpy(apply_permutation(np.ascontiguousarray(X_local_orig[v], self.scs.np_dtype), self.new_to_old)).cuda()
            if layout == ROWWISE:
                X[v:self.n_local * b:b] = xp
            else:
                X[v * ld:v * ld + self.n_local] = xp
        return X

    def run(self, x, y, n_steps, use_graph=False):
        self._order(x, y)
        _ck(lib().uspmv_dist_run(self.h, _dp(x), _dp(y), int(n_steps), int(bool(use_graph)), self.stream.cuda_stream))
        return y

    def synchronize(self):
        self.stream.synchronize()

    def barrier(self):
        _ck(lib().uspmv_dist_barrier(self.h, self.stream.cuda_stream))

    def allreduce_max(self, v):
        d = C.c_double(float(v))
        _ck(lib().uspmv_dist_allreduce_max(self.h, C.byref(d), self.stream.cuda_stream))
        return d.value

    def close(self):
        if getattr(self, "h", None) and _LIB is not None:
            self.scs = None
            _LIB.uspmv_dist_free(self.h)
            self.h = None

    def __del__(self):
        self.close()


class HaloPlan:
    """Result of collect_local_needed_heri on one rank (code/mpi_funcs.hpp:242-415)."""

    def __init__(self, scs, wsa, rank, P):
        wsa = np.ascontiguousarray(wsa, np.int32)
        h = _vp()
        _ck(lib().uspmv_halo_discover(scs.h, _np_ptr(wsa), rank, P, C.byref(h)))
        self.h, self.P, self.rank = h, P, rank
        n = _i64()
        cum, idx, cnt = _i32p(), _i32p(), _i32p()
        _ck(lib().uspmv_halo_meta(h, C.byref(n), C.byref(cum), C.byref(idx), C.byref(cnt)))
        self.n_halo = n.value
        self.recv_counts_cumsum = _view(cum, P + 1, np.int32).copy()
        self.recv_counts = _view(cnt, P, np.int32).copy()
        self.recv_idxs = _view(idx, self.n_halo, np.int32).copy()
        self.n_local = int(wsa[rank + 1] - wsa[rank])

    def recv_idxs_of(self, owner):
        o = int(self.recv_counts[:owner].sum())
        return self.recv_idxs[o:o + int(self.recv_counts[owner])]

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.uspmv_halo_free(self.h)
            self.h = None


# ---------------------------------------------------------------------------------------- device
def device_count():
    n = C.c_int()
    _ck(lib().uspmv_device_count(C.byref(n)))
    return n.value


def set_tuning(**kw):
    for k, v in kw.items():
        _ck(lib().uspmv_set_tuning(k.encode(), int(v)))


def get_tuning(key):
    v = C.c_int()
    _ck(lib().uspmv_get_tuning(key.encode(), C.byref(v)))
    return v.value


def _stream_ptr(stream):
    if stream is None:
        import torch
        return torch.cuda.current_stream().cuda_stream
    return getattr(stream, "cuda_stream", stream)


def _dp(t):
    return None if t is None else t.data_ptr()


class DeviceMatrix:
    """SELL-C-sigma matrix resident in HBM (what assign_spmv_kernel_gpu_data stages,
    code/utilities.hpp:3721-3811).  Arrays are torch tensors owned by this object."""

    def __init__(self, scs, device="cuda", crs=False, tlc=False, tlc_max_lines=0, block_tlc=0, _handle=None):
        import torch
        if _handle is not None:          # arrays owned by the library (convert_to_scs_device)
            self.C, self.n_chunks, self.n_elements, self.dtype = scs.C, scs.n_chunks, scs.n_elements, scs.dtype
            self.n_rows, self.n_rows_padded, self.nnz = scs.n_rows, scs.n_rows_padded, scs.nnz
            self.torch_dtype = torch.float64 if scs.dtype == F64 else torch.float32
            self.h = _handle
            self.tlc_tiles = self.tlc_staged = self.tile_rows = self.block_tiles = self.block_staged = 0
            return
        a = scs.arrays()
        self.C, self.n_chunks, self.n_elements, self.dtype = scs.C, scs.n_chunks, scs.n_elements, scs.dtype
        self.n_rows, self.n_rows_padded, self.nnz = scs.n_rows, scs.n_rows_padded, scs.nnz
        self.torch_dtype = torch.float64 if scs.dtype == F64 else torch.float32
        dev = torch.device(device)
        self.chunk_ptrs = torch.from_numpy(a["chunk_ptrs"].copy()).to(dev)
        self.chunk_lengths = torch.from_numpy(a["chunk_lengths"].copy()).to(dev)
        self.col_idxs = torch.from_numpy(a["col_idxs"]).to(dev)      # H2D straight from library memory
        self.values = torch.from_numpy(a["values"]).to(dev)
        h = _vp()
        _ck(lib().uspmv_dmat_wrap(self.C, self.n_chunks, self.n_elements, self.dtype, _dp(self.chunk_ptrs),
                                  _dp(self.chunk_lengths), _dp(self.col_idxs), _dp(self.values), C.byref(h)))
        self.h = h
        self.tlc_tiles = self.tlc_staged = self.tile_rows = 0
        if crs:
            _ck(lib().uspmv_dmat_set_crs(h, 1))
        if tlc:
            self.optimize(scs, tlc_max_lines)
        self.block_tiles = self.block_staged = 0
        if block_tlc:
            self.optimize_block(scs, block_tlc)

    def optimize_block_sweep(self, scs, b, wlog=0, tile_rows=0):
        """Block-vector column-window sweep plan for 64-byte X rows (uspmv_dmat_optimize_block_sweep); returns (n_tiles, n_sweep_tiles):
        installed iff they are equal."""
        a, n = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize_block_sweep(self.h, scs.h, int(b), int(wlog), int(tile_rows), C.byref(a), C.byref(n)))
        return a.value, n.value

    def optimize_device(self, max_lines=0):
        """Build the tile-local-column plan on the device from the handle's own arrays (uspmv_dmat_optimize_device)."""
        a, b = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize_device(self.h, max_lines, C.byref(a), C.byref(b)))
        self.tlc_tiles, self.tlc_staged = a.value, b.value
        tr = C.c_int()
        _ck(lib().uspmv_dmat_tile_rows(self.h, C.byref(tr)))
        self.tile_rows = tr.value
        return a.value, b.value

    def plan_granularity(self):
        """x elements per list entry of the tile-local-column plan: 16 (lines), 1 (single elements), 0 (no plan): uspmv_dmat_plan_granularity."""
        g = C.c_int()
        _ck(lib().uspmv_dmat_plan_granularity(self.h, C.byref(g)))
        return g.value

    def plan_rows_dealt(self):
        """True when the tile-local-column plan runs on rows dealt to its tiles by the matrix graph (uspmv_dmat_plan_rows_dealt)."""
        g = C.c_int()
        _ck(lib().uspmv_dmat_plan_rows_dealt(self.h, C.byref(g)))
        return bool(g.value)

    def index_bits(self):
        """Bits per tile-local column index the plan's kernel streams (16 | 12; 0 without a plan): uspmv_dmat_index_bits."""
        b = C.c_int()
        _ck(lib().uspmv_dmat_index_bits(self.h, C.byref(b)))
        return b.value

    def plan_download(self):
        """Host copies of the tile-local-column plan (tests): dict or None when the handle has no plan."""
        meta = (_i64 * 4)()
        _ck(lib().uspmv_dmat_plan_download(self.h, meta, None, None, None, None))
        nt, nl, n16, mx = [int(v) for v in meta]
        if nt == 0:
            return None
        lp = np.empty(nt + 1, np.int32); tl = np.empty(nl, np.int32)
        cp = np.empty(self.n_chunks + 1, np.uint32); c16 = np.empty(n16, np.uint16)
        _ck(lib().uspmv_dmat_plan_download(self.h, meta, _np_ptr(lp), _np_ptr(tl), _np_ptr(cp), _np_ptr(c16)))
        return dict(tile_line_ptr=lp, tile_lines=tl, c16_ptrs=cp, col16=c16, max_lines_used=mx)

    def optimize_block(self, scs, block_vec_size):
        """Build the LDS-staged SpMMV plan for block vectors of this width (uspmv_dmat_optimize_block)."""
        a, b = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize_block(self.h, scs.h, int(block_vec_size), C.byref(a), C.byref(b)))
        self.block_tiles, self.block_staged = a.value, b.value
        return a.value, b.value

    def optimize_sweep(self, scs, wlog=0, tile_rows=0):
        """Build the column-window sweep plan (uspmv_dmat_optimize_sweep); returns (n_tiles, n_sweep_tiles)."""
        a, b = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize_sweep(self.h, scs.h, int(wlog), int(tile_rows), C.byref(a), C.byref(b)))
        return a.value, b.value

    def plan_info(self):
        """(kind, n_tiles, n_planned) of the single-vector plan in use: kind 0 none, 1 tile-local-column, 2 column-window sweep."""
        k, a, b = C.c_int(), _i64(), _i64()
        _ck(lib().uspmv_dmat_plan_info(self.h, C.byref(k), C.byref(a), C.byref(b)))
        return k.value, a.value, b.value

    def block_plan_digest(self):
        d = (C.c_uint64 * 8)()
        _ck(lib().uspmv_dmat_block_plan_digest(self.h, d))
        return [int(v) for v in d]

    def optimize_sweep_device(self, sp=None, wlog=0, tile_rows=0):
        """the column-window sweep plan from the device arrays alone (uspmv_dmat_optimize_sweep_device); returns (tiles, sweep tiles)"""
        a, b = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize_sweep_device(self.h, sp.h if sp is not None else None, wlog, tile_rows, C.byref(a), C.byref(b)))
        return a.value, b.value

    def sweep_plan_digest(self):
        """(digests of the sweep plan's device arrays, meta) -- equal for equal plans"""
        d, m = (C.c_uint64 * 16)(), (_i64 * 8)()
        _ck(lib().uspmv_dmat_sweep_plan_digest(self.h, d, m))
        return [int(v) for v in d], [int(v) for v in m]

    def block_plan_info(self):
        """dict of the handle's block-vector plans (uspmv_dmat_block_plan_info)"""
        m = (_i64 * 10)()
        _ck(lib().uspmv_dmat_block_plan_info(self.h, m))
        keys = ("list_plan", "phased_plan", "line_plan", "tiles", "phases", "line_phases", "line_rows_staged", "idx8", "device_built", "max_rows")
        d = dict(zip(keys, [int(v) for v in m]))
        n = _i64()
        _ck(lib().uspmv_dmat_block_plan_staged(self.h, C.byref(n)))
        d["rows_staged"] = n.value
        m2 = (_i64 * 2)()
        _ck(lib().uspmv_dmat_stream_info(self.h, m2))
        d["stream_grid"], d["stream_descriptors"] = int(m2[0]), int(m2[1])
        return d

    def optimize_block_device(self, block_vec_size):
        """The block plan from the handle's device arrays alone (uspmv_dmat_optimize_block_device)."""
        a, b = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize_block_device(self.h, int(block_vec_size), C.byref(a), C.byref(b)))
        self.block_tiles, self.block_staged = a.value, b.value
        return a.value, b.value

    def optimize(self, scs, max_lines=0):
        """Build the tile-local-column plan (uspmv_dmat_optimize); returns (n_tiles, n_staged_tiles)."""
        a, b = _i64(), _i64()
        _ck(lib().uspmv_dmat_optimize(self.h, scs.h, max_lines, C.byref(a), C.byref(b)))
        self.tlc_tiles, self.tlc_staged = a.value, b.value
        tr = C.c_int()
        _ck(lib().uspmv_dmat_tile_rows(self.h, C.byref(tr)))
        self.tile_rows = tr.value          # 0: no plan was built
        return a.value, b.value

    def __del__(self):
        if getattr(self, "h", None) and _LIB is not None:
            _LIB.uspmv_dmat_free(self.h)
            self.h = None


def convert_to_scs_device(coo, C_, sigma, dtype=F64, fixed_permutation=None, permute_cols=True, device="cuda"):
    """GPU-side convert_to_scs (+ permute_scs_cols): returns (layout-only Scs, DeviceMatrix)."""
    import torch
    dev = torch.device(device)
    if dev.index is not None:
        torch.cuda.set_device(dev)
    fp = None if fixed_permutation is None else np.ascontiguousarray(fixed_permutation, np.int32)
    hs, hA = _vp(), _vp()
    _ck(lib().uspmv_convert_to_scs_device(coo.h, C_, sigma, dtype, _np_ptr(fp), int(bool(permute_cols)), C.byref(hs), C.byref(hA)))
    s = Scs(hs)
    return s, DeviceMatrix(s, device, _handle=hA)


SORT_HOST, SORT_DEVICE_STABLE = 0, 1


def convert_to_scs_device_from_arrays(d_I, d_J, d_V, n_rows, n_cols, C_, sigma, dtype=F64, fixed_permutation=None, permute_cols=True,
                                      sort=SORT_HOST, want_layout=True, stream=None):
    """convert_to_scs (+ permute_scs_cols) from DEVICE-resident COO arrays (torch int32 / int32 / float64 tensors, entries sorted by row):
    returns (layout-only Scs or None, DeviceMatrix, old_to_new, new_to_old) -- the permutations as int32 device tensors."""
    import torch
    assert d_I.dtype == torch.int32 and d_J.dtype == torch.int32 and d_V.dtype == torch.float64 and d_I.is_cuda
    nnz = int(d_I.numel())
    o2n = torch.empty(n_rows, dtype=torch.int32, device=d_I.device)
    n2o = torch.empty(n_rows, dtype=torch.int32, device=d_I.device)
    hs, hA = _vp(), _vp()
    _ck(lib().uspmv_convert_to_scs_device_from_arrays(_dp(d_I), _dp(d_J), _dp(d_V), n_rows, n_cols, nnz, C_, sigma, dtype,
                                                      None if fixed_permutation is None else _dp(fixed_permutation), int(bool(permute_cols)), int(sort),
                                                      _stream_ptr(stream), C.byref(hs) if want_layout else None, _dp(o2n), _dp(n2o), C.byref(hA)))
    if want_layout:
        s = Scs(hs)
        return s, DeviceMatrix(s, d_I.device, _handle=hA), o2n, n2o

    class _Meta:      # what DeviceMatrix needs to know about a handle that has no host struct
        pass
    m = (_i64 * 4)()
    _ck(lib().uspmv_dmat_meta(hA, m))
    s = _Meta()
    s.C, s.n_chunks, s.n_elements, s.dtype = int(m[0]), int(m[1]), int(m[2]), int(m[3])
    s.n_rows, s.n_rows_padded, s.nnz = n_rows, s.n_chunks * s.C, nnz
    return None, DeviceMatrix(s, d_I.device, _handle=hA), o2n, n2o


def spmmv_x_prepared(A, X, b, ld, stream=None):
    """column-major X unchanged between calls: re-lay it out once (uspmv_spmmv_x_prepared); spmmv_x_release(A) ends it"""
    _ck(lib().uspmv_spmmv_x_prepared(A.h, _dp(X), int(b), int(ld), _stream_ptr(stream)))


def spmmv_x_release(A):
    _ck(lib().uspmv_spmmv_x_release(A.h))


def dmat_download(A):
    """Host copies of a DeviceMatrix's arrays (tests / debugging)."""
    cp = np.empty(A.n_chunks + 1, np.int32); cl = np.empty(A.n_chunks, np.int32)
    ci = np.empty(A.n_elements, np.int32); va = np.empty(A.n_elements, np.float64 if A.dtype == F64 else np.float32)
    _ck(lib().uspmv_dmat_download(A.h, _np_ptr(cp), _np_ptr(cl), _np_ptr(ci), _np_ptr(va)))
    return dict(chunk_ptrs=cp, chunk_lengths=cl, col_idxs=ci, values=va)


def optimize_ap(A_dp, A_sp, scs_dp, scs_sp, max_lines=0):
    """Shared tile-local-column plan for an ap[dp_sp] pair; returns (n_tiles, n_staged_tiles)."""
    a, b = _i64(), _i64()
    _ck(lib().uspmv_dmat_optimize_ap(A_dp.h, A_sp.h, scs_dp.h, scs_sp.h, max_lines, C.byref(a), C.byref(b)))
    for A in (A_dp, A_sp):
        A.tlc_tiles, A.tlc_staged = a.value, b.value
    return a.value, b.value


def optimize_device_ap(A_dp, A_sp, max_lines=0):
    """Shared tile-local-column plan of an ap[dp_sp] pair built on the device from the handles' own arrays."""
    a, b = _i64(), _i64()
    _ck(lib().uspmv_dmat_optimize_device_ap(A_dp.h, A_sp.h, max_lines, C.byref(a), C.byref(b)))
    for A in (A_dp, A_sp):
        A.tlc_tiles, A.tlc_staged = a.value, b.value
    return a.value, b.value


def optimize_sweep_ap(A_dp, A_sp, scs_dp, scs_sp, wlog=0, tile_rows=0):
    """Shared column-window sweep plan for an ap[dp_sp] pair; returns (n_tiles, n_sweep_tiles)."""
    a, b = _i64(), _i64()
    _ck(lib().uspmv_dmat_optimize_sweep_ap(A_dp.h, A_sp.h, scs_dp.h, scs_sp.h, int(wlog), int(tile_rows), C.byref(a), C.byref(b)))
    return a.value, b.value


def spmv(A, x, y, stream=None):
    """y = A x on the device (x: padded_vec_size, y: n_rows_padded elements)."""
    assert x.dtype == A.torch_dtype and y.dtype == A.torch_dtype and y.numel() >= A.n_rows_padded
    _ck(lib().uspmv_spmv(A.h, _dp(x), _dp(y), _stream_ptr(stream)))
    return y


def spmv_chunks(A, chunk_ids, x, y, stream=None):
    assert chunk_ids.dtype.is_floating_point is False and x.dtype == A.torch_dtype
    _ck(lib().uspmv_spmv_chunks(A.h, _dp(chunk_ids), chunk_ids.numel(), _dp(x), _dp(y), _stream_ptr(stream)))
    return y


def spmv_tiles(A, tile_ids, x, y, stream=None):
    _ck(lib().uspmv_spmv_tiles(A.h, _dp(tile_ids), tile_ids.numel(), _dp(x), _dp(y), _stream_ptr(stream)))
    return y


def spmmv(A, X, Y, b, ld, layout=COLWISE, stream=None):
    assert X.dtype == A.torch_dtype and Y.dtype == A.torch_dtype
    _ck(lib().uspmv_spmmv(A.h, _dp(X), _dp(Y), b, ld, layout, _stream_ptr(stream)))
    return Y


def spmv_ap(A_dp, A_sp, x, y, stream=None, x_sp=None):
    """Adaptive precision dp+sp.  x_sp (float copy of x) selects the reference's generic-C variant."""
    if x_sp is None:
        _ck(lib().uspmv_spmv_ap(A_dp.h, A_sp.h, _dp(x), _dp(y), _stream_ptr(stream)))
    else:
        _ck(lib().uspmv_spmv_ap_generic(A_dp.h, A_sp.h, _dp(x), _dp(x_sp), _dp(y), _stream_ptr(stream)))
    return y


def uspmv_scs_gpu(Cc, n_chunks, chunk_ptrs, chunk_lengths, col_idxs, values, x, y, stream=None):
    """Raw-array form, argument list of uspmv_scs_gpu (code/interface.hpp:1766-1793)."""
    import torch
    f = lib().uspmv_scs_gpu_f64 if values.dtype == torch.float64 else lib().uspmv_scs_gpu_f32
    _ck(f(Cc, n_chunks, _dp(chunk_ptrs), _dp(chunk_lengths), _dp(col_idxs), _dp(values), _dp(x), _dp(y),
          _stream_ptr(stream)))
    return y


def uspmv_csr_gpu(n_rows, row_ptrs, col_idxs, values, x, y, stream=None):
    """Raw-array form, argument list of uspmv_csr_gpu (code/interface.hpp:1741-1760)."""
    import torch
    f = lib().uspmv_csr_gpu_f64 if values.dtype == torch.float64 else lib().uspmv_csr_gpu_f32
    _ck(f(n_rows, _dp(row_ptrs), _dp(col_idxs), _dp(values), _dp(x), _dp(y), _stream_ptr(stream)))
    return y


def apply_permutation_dev(out, vec, perm, stream=None):
    import torch
    dt = F64 if vec.dtype == torch.float64 else F32
    _ck(lib().uspmv_apply_permutation_dev(_dp(out), _dp(vec), _dp(perm), perm.numel(), dt, _stream_ptr(stream)))
    return out


def pack_send_buf(x, perm, send_idxs, out, block_offset=0, stream=None):
    """out[i] = x[perm[send_idxs[i]] + block_offset] for the concatenated send list (one launch)."""
    import torch
    dt = F64 if x.dtype == torch.float64 else F32
    _ck(lib().uspmv_pack_send_buf(_dp(x), _dp(perm), _dp(send_idxs), send_idxs.numel(), block_offset, _dp(out), dt,
                                  _stream_ptr(stream)))
    return out


def time_launches(what, reps, A=None, B=None, x=None, y=None, n=0, b=1, ld=0, layout=COLWISE, stream=None):
    """Average ms per launch of `reps` back-to-back launches, HIP events on the launch stream."""
    ms = C.c_double()
    _ck(lib().uspmv_time_launches(what, reps, A.h if A is not None else None, B.h if B is not None else None,
                                  _dp(x), _dp(y), n, b, ld, layout, _stream_ptr(stream), C.byref(ms)))
    return ms.value
