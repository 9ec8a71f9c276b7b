"""uspmv-mi355x: MI355X-native SELL-C-sigma SpMV / SpMMV engine.

The product is the C-ABI shared library ``libuspmv.so`` (host data layer in C++, hand-written
HIP kernels for gfx950; see include/uspmv.h).  This package is only the thin ctypes front-end used
by tests/, bench.py and __graft_entry__.py, plus the torch.distributed (RCCL) halo-exchange
driver.  It never imports anything from ``oracle/`` and has no CPU fallback: device entry points
raise UspmvError when the HIP extension or a GPU is missing.

The directory name contains a hyphen (repository convention), so import it through
``__graft_entry__.load_package()`` (module name ``ultimate_spmv_amd``).
"""
from .binding import (  # noqa: F401
    COLWISE, F32, F64, ROWWISE, SEG_NNZ, SEG_ROWS, Coo, DeviceMatrix, HaloPlan, Scs, UspmvError, apply_permutation,
    build_library, convert_to_scs, convert_to_scs_device, convert_to_scs_device_from_arrays, SORT_HOST, SORT_DEVICE_STABLE, device_count, dmat_download, gen_banded_random, graph_partition, read_partition, apply_partition, gen_kkt, gen_kkt_row_counts, gen_stencil27, get_tuning, lib, library_path, optimize_ap, optimize_device_ap, optimize_sweep_ap, pack_send_buf,
    partition_precisions, permute_scs_cols, read_mtx, seg_work_sharing_arr, seg_local_coo, set_tuning, spmmv_x_prepared, spmmv_x_release, spmmv, spmv, spmv_ap,
    spmv_chunks, spmv_tiles, uspmv_csr_gpu, uspmv_scs_gpu, DistNative, HostComm, CommPlan, Transport, DistOptions, EXCHANGE_HOST, EXCHANGE_RCCL, runtime_versions, dist_check_reference, comm_unique_id, seg_from_row_counts, gen_stencil27_row_counts,
)
