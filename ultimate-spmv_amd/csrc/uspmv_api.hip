// Device half of the C ABI (include/uspmv.h): handles, tuning, plans, GPU-side conversion, the small gather /
// stream kernels and the entry points that dispatch into spmv_kernels.hip, spmmv_kernels.hip and ap_kernels.hip.
#include "uspmv_device.hpp"

#include <mutex>

using namespace uspmv_dev;

namespace uspmv_dev { thread_local int tl_measure_off = 0; }
namespace uspmv_dev {

Tuning g_tune;

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n < 1) {
        (void)hipGetLastError();
        return uspmv::fail(USPMV_ERR_NO_DEVICE, "no HIP device is visible (hipGetDeviceCount: %s); "
                           "libuspmv has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    }
    return USPMV_OK;
}

int check_dmat(const uspmv_dmat *A, const char *who) {
    if (!A) return uspmv::fail(USPMV_ERR_INVALID, "%s: NULL matrix", who);
    if (A->C < 1 || A->n_chunks < 0) return uspmv::fail(USPMV_ERR_INVALID, "%s: corrupt matrix handle", who);
    if (A->n_chunks * A->C > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "%s: padded rows exceed int32", who);
    return USPMV_OK;
}


}  // namespace uspmv_dev

namespace {


// COO -> SELL-C-sigma scatter of uspmv_convert_to_scs_device: one thread per COO entry k (entries sorted by
// row, order inside a row preserved): slot = k - row_start[row], destination as convert_to_scs
// (code/utilities.hpp:2013-2036).  perm != nullptr folds permute_scs_cols (:1802-1831) into the same pass.
template <typename VT>
__global__ void scs_fill_kernel(const long nnz, const int C, const int n_rows, const int *__restrict__ I,
                                const int *__restrict__ J, const double *__restrict__ V,
                                const int *__restrict__ row_start, const int *__restrict__ row_map,
                                const int *__restrict__ perm, const int *__restrict__ chunk_ptrs,
                                int *__restrict__ col_idxs, VT *__restrict__ values) {
    const long k = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz) return;
    const int r = I[k];
    const int slot = (int)(k - row_start[r]);
    const int row = row_map[r];
    const int c = row / C;
    const long dst = (long)chunk_ptrs[c] + (long)slot * C + (row - c * C);
    int col = J[k];
    if (perm && col < n_rows) col = perm[col];
    col_idxs[dst] = col;
    values[dst] = (VT)V[k];
}

// out[i] = in[perm[idx ? idx[i] : i] + offset]   (pack_send_buf with idx, apply_permutation without)
template <typename VT>
__global__ void gather_kernel(VT *__restrict__ out, const VT *__restrict__ in, const int *__restrict__ perm,
                              const int *__restrict__ idx, const long n, const long offset) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[(long)perm[idx ? idx[i] : (int)i] + offset];
}

// STREAM-style calibrators.  copy / triad: every thread moves eight 16-byte pieces, all loads issued before the first store
// (32 KiB in flight per 256-thread workgroup), non-temporal loads and stores -- the shape the guide's 6.3 TB/s copy has; the
// grid-stride form of round 1 (one 16-byte piece in flight per thread, plain stores) stopped at 4.9 TB/s.
__global__ void __launch_bounds__(256) stream_copy_kernel(double2 *__restrict__ a, const double2 *__restrict__ b, const long n2) {
    const long base = (long)blockIdx.x * (256 * 8) + threadIdx.x;
    double v[16];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long i = base + u * 256;
        if (i < n2) { const double *p = (const double *)(b + i); v[2 * u] = __builtin_nontemporal_load(p); v[2 * u + 1] = __builtin_nontemporal_load(p + 1); }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long i = base + u * 256;
        if (i < n2) { double *p = (double *)(a + i); __builtin_nontemporal_store(v[2 * u], p); __builtin_nontemporal_store(v[2 * u + 1], p + 1); }
    }
}
__global__ void __launch_bounds__(256) stream_triad_kernel(double2 *__restrict__ a, const double2 *__restrict__ b,
                                                          const double2 *__restrict__ c, const double s, const long n2) {
    const long base = (long)blockIdx.x * (256 * 4) + threadIdx.x;
    double vb[8], vc[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256;
        if (i < n2) {
            const double *pb = (const double *)(b + i), *pc = (const double *)(c + i);
            vb[2 * u] = __builtin_nontemporal_load(pb); vb[2 * u + 1] = __builtin_nontemporal_load(pb + 1);
            vc[2 * u] = __builtin_nontemporal_load(pc); vc[2 * u + 1] = __builtin_nontemporal_load(pc + 1);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256;
        if (i < n2) {
            double *p = (double *)(a + i);
            __builtin_nontemporal_store(vb[2 * u] + s * vc[2 * u], p); __builtin_nontemporal_store(vb[2 * u + 1] + s * vc[2 * u + 1], p + 1);
        }
    }
}
__global__ void stream_read_kernel(const double2 *__restrict__ b, const long n2, double *__restrict__ partial) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        const double *pb = (const double *)(b + i);
        acc += __builtin_nontemporal_load(pb) + __builtin_nontemporal_load(pb + 1);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) partial[((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6] = acc;
}

// Second yardstick (what the tile-local-column kernel asks of the cache hierarchy beyond a stream): as many workgroups as the SpMV has
// tiles, mapped to the XCDs like its tiles, each reading the x lines a `rows`-row tile of a 27-point stencil touches -- nine runs of
// rows + 2 elements (rounded out to whole 128-byte lines) at the offsets {-plane, 0, +plane} + {-line, 0, +line} around its own rows --
// with the plain 16-byte loads of the kernel's staging phase.  Every element of x is asked for by ~9 workgroups, ~3 of them far apart
// in launch order (the neighbouring planes): what comes from the XCD's L2, what from the fabric, is exactly the SpMV's x traffic,
// without its matrix stream.  Reported as gathered bytes per second.
__global__ void __launch_bounds__(256) stream_gather_lines_kernel(const double2 *__restrict__ x, const long n, const long plane, const long line,
                                                                  const int rows, const int xcd_remap, double *__restrict__ partial) {
    const unsigned tile = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long base = (long)tile * rows;
    double acc = 0.0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy) {
            long lo = base + dz * plane + dy * line - 1, hi = base + rows + dz * plane + dy * line + 1;
            lo = lo < 0 ? 0 : (lo & ~15L);
            hi = hi > n ? n : hi;
            hi = (hi + 15) & ~15L;
            if (hi > (n & ~15L)) hi = n & ~15L;
            for (long i = lo / 2 + threadIdx.x; i < hi / 2; i += 256) { const double2 v = x[i]; acc += v.x + v.y; }
        }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) partial[(long)blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}

}  // namespace

namespace uspmv_dev {
// the O(nnz) scatter of convert_to_scs for callers in other files (csrc/convert_kernels.hip)
int launch_scs_fill(int dtype, long nnz, int C, int n_rows, const int *I, const int *J, const double *V, const int *row_start, const int *row_map,
                    const int *perm, const int *chunk_ptrs, int *col_idxs, void *values, hipStream_t st) {
    const unsigned grid = (unsigned)((nnz + 255) / 256);
    if (dtype == USPMV_F64)
        hipLaunchKernelGGL(scs_fill_kernel<double>, dim3(grid), dim3(256), 0, st, nnz, C, n_rows, I, J, V, row_start, row_map, perm, chunk_ptrs, col_idxs, (double *)values);
    else
        hipLaunchKernelGGL(scs_fill_kernel<float>, dim3(grid), dim3(256), 0, st, nnz, C, n_rows, I, J, V, row_start, row_map, perm, chunk_ptrs, col_idxs, (float *)values);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}
}  // namespace uspmv_dev

// ============================================================================================ C ABI
extern "C" {

int uspmv_device_count(int *count) {
    if (!count) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_device_count: NULL argument");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return USPMV_OK;
}

int uspmv_set_device(int device) {
    if (int rc = require_device()) return rc;
    HIP_TRY(hipSetDevice(device));
    return USPMV_OK;
}

int uspmv_stream_synchronize(void *stream) {
    if (int rc = require_device()) return rc;
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return USPMV_OK;
}

int uspmv_set_tuning(const char *key, int value) {
    if (!key) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_set_tuning: NULL key");
    if (!strcmp(key, "unroll")) {
        if (value != 1 && value != 2 && value != 4 && value != 8) return uspmv::fail(USPMV_ERR_INVALID, "unroll must be 1|2|4|8");
        g_tune.unroll = value;
    } else if (!strcmp(key, "nontemporal")) g_tune.nontemporal = value != 0;
    else if (!strcmp(key, "xcd_remap")) {
        if (value < 0 || value > 65536) return uspmv::fail(USPMV_ERR_INVALID, "xcd_remap must be 0, 1 or a group size <= 65536");
        g_tune.xcd_remap = (g_tune.xcd_remap & ~0xFFFFF) | value;
    } else if (!strcmp(key, "xcd_stagger")) {     // XCD k starts its group value*k tiles in (remap_block); 0 = all XCDs in step
        if (value < 0 || value > 2047) return uspmv::fail(USPMV_ERR_INVALID, "xcd_stagger must be in [0, 2047]");
        g_tune.xcd_remap = (g_tune.xcd_remap & 0xFFFFF) | (value << 20);
    } else if (!strcmp(key, "ablate")) g_tune.ablate = value;
    else if (!strcmp(key, "spmmv_prefetch")) g_tune.spmmv_prefetch = value != 0;
    else if (!strcmp(key, "spmmv_swizzle")) g_tune.spmmv_swizzle = value != 0;
    else if (!strcmp(key, "spmmv_reorder")) g_tune.spmmv_reorder = value < 0 ? 0 : value > 4 ? 4 : (int)value;
    else if (!strcmp(key, "spmmv_brick_stride")) g_tune.spmmv_brick_stride = value < 0 ? 0 : (long)value;
    else if (!strcmp(key, "spmmv_brick_lines")) g_tune.spmmv_brick_lines = value < 1 ? 1 : (int)value;
    else if (!strcmp(key, "spmmv_phase_dp")) g_tune.spmmv_phase_dp = value < 0 ? 0 : (int)value;
    else if (!strcmp(key, "tlc_elem")) g_tune.tlc_elem = value < 0 ? 0 : value > 2 ? 2 : (int)value;
    else if (!strcmp(key, "tlc_elem_rows")) g_tune.tlc_elem_rows = value < 0 ? 0 : (int)value;
    else if (!strcmp(key, "tlc_elem_seg_rows")) g_tune.tlc_elem_seg_rows = value < 65536 ? 65536 : (int)value;
    else if (!strcmp(key, "tlc_elem_cap")) g_tune.tlc_elem_cap = value < 64 ? 64 : value > 16384 ? 16384 : (int)value;
    else if (!strcmp(key, "spmmv_stream")) g_tune.spmmv_stream = value < 0 ? 0 : value >= 99 ? 99 : value > 5 ? 5 : (int)value;
    else if (!strcmp(key, "spmmv_stream_waves")) g_tune.spmmv_stream_waves = value >= 5 ? 5 : 4;
    else if (!strcmp(key, "spmmv_stream_xcd")) g_tune.spmmv_stream_xcd = value != 0;
    else if (!strcmp(key, "spmmv_stream_depth")) g_tune.spmmv_stream_depth = value >= 2 ? 2 : 1;
    else if (!strcmp(key, "spmmv_phased")) g_tune.spmmv_phased = value != 0;
    else if (!strcmp(key, "spmmv_xcol")) g_tune.spmmv_xcol = value != 0;
    else if (!strcmp(key, "spmmv_ycol_nt")) g_tune.spmmv_ycol_nt = value != 0;
    else if (!strcmp(key, "spmmv_xline")) g_tune.spmmv_xline = value != 0;
    else if (!strcmp(key, "block_plan_device")) g_tune.block_plan_device = value != 0;
    else if (!strcmp(key, "spmmv_unscramble")) g_tune.spmmv_unscramble = value != 0;
    else if (!strcmp(key, "spmmv_phase_rows")) g_tune.spmmv_phase_rows = value == 512 ? 512 : 256;
    else if (!strcmp(key, "spmmv_idx8")) g_tune.spmmv_idx8 = value != 0;
    else if (!strcmp(key, "spmmv_list_plan")) g_tune.spmmv_list_plan = value != 0;
    else if (!strcmp(key, "sweep")) g_tune.sweep = value != 0;
    else if (!strcmp(key, "sweep_nbuf")) g_tune.sweep_nbuf = value == 1 ? 1 : 2;
    else if (!strcmp(key, "sweep_unroll")) g_tune.sweep_unroll = value >= 8 ? 8 : value >= 4 ? 4 : 2;
    else if (!strcmp(key, "sweep_pair")) g_tune.sweep_pair = value < 0 ? 0 : value > 2 ? 2 : value;
    else if (!strcmp(key, "sweep_remap")) g_tune.sweep_remap = value < 0 ? 0 : value;
    else if (!strcmp(key, "sweep_wlog")) {
        if (value != 0 && (value < 8 || value > 16)) return uspmv::fail(USPMV_ERR_INVALID, "sweep_wlog must be 0 or 8..16");
        g_tune.sweep_wlog = value;
    }
    else if (!strcmp(key, "sweep_tile_rows")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024 && value != 2048 && value != 4096) return uspmv::fail(USPMV_ERR_INVALID, "sweep_tile_rows must be 0|256|512|1024|2048|4096");
        g_tune.sweep_tile_rows = value;
    }
    else if (!strcmp(key, "sweep_threads")) g_tune.sweep_threads = (value == 256 || value == 512 || value == 1024) ? (int)value : 0;
    else if (!strcmp(key, "sweep_max_stage")) g_tune.sweep_max_stage = value < 0 ? 0 : value;
    else if (!strcmp(key, "raw_plan_cache")) g_tune.raw_plan_cache = value != 0;
    else if (!strcmp(key, "spmmv_tile_rows")) g_tune.spmmv_tile_rows = value == 64 ? 64 : value == 32 ? 32 : 0;
    else if (!strcmp(key, "spmmv_lds_kb")) g_tune.spmmv_lds_kb = value < 0 ? 0 : value;
    else if (!strcmp(key, "spmmv_variant")) {
        if (value < 0 || value > 9 || value == 7) return uspmv::fail(USPMV_ERR_INVALID, "spmmv_variant must be 0..6, 8 or 9 (9: the block-vector window sweep only)");
        g_tune.spmmv_variant = value;
    }
    else if (!strcmp(key, "tail_batch")) g_tune.tail_batch = value != 0;
    else if (!strcmp(key, "spmmv_unroll")) g_tune.spmmv_unroll = value;
    else if (!strcmp(key, "tlc")) g_tune.tlc = value != 0;
    else if (!strcmp(key, "rechunk")) g_tune.rechunk = value != 0;
    else if (!strcmp(key, "tlc_auto_tile")) g_tune.tlc_auto_tile = value != 0;
    else if (!strcmp(key, "tlc_measure_tile")) g_tune.tlc_measure_tile = value != 0;
    else if (!strcmp(key, "tlc_idx12")) g_tune.tlc_idx12 = value < 0 ? 0 : value > 2 ? 2 : (int)value;
    else if (!strcmp(key, "tlc_tile_rows")) {
        if (value != 0 && value != 256 && value != 512 && value != 1024) return uspmv::fail(USPMV_ERR_INVALID, "tlc_tile_rows must be 0|256|512|1024");
        g_tune.tlc_tile_rows = value;
    }
    else if (!strcmp(key, "block")) {
        if (value != 64 && value != 128 && value != 256 && value != 512 && value != 1024)
            return uspmv::fail(USPMV_ERR_INVALID, "block must be 64|128|256|512|1024");
        g_tune.block = value;
    } else if (!strcmp(key, "spmv_variant")) {
        if (value < 0 || value > 2) return uspmv::fail(USPMV_ERR_INVALID, "spmv_variant must be 0|1|2");
        g_tune.spmv_variant = value;
    } else if (!strcmp(key, "csr_lanes")) {
        if (value < 0 || value > 64 || (value & (value - 1))) return uspmv::fail(USPMV_ERR_INVALID, "csr_lanes must be 0 or a power of two <= 64");
        g_tune.csr_lanes = value;
    } else return uspmv::fail(USPMV_ERR_INVALID, "uspmv_set_tuning: unknown key '%s'", key);
    return USPMV_OK;
}

int uspmv_get_tuning(const char *key, int *value) {
    if (!key || !value) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_get_tuning: NULL argument");
    if (!strcmp(key, "unroll")) *value = g_tune.unroll;
    else if (!strcmp(key, "nontemporal")) *value = g_tune.nontemporal;
    else if (!strcmp(key, "xcd_remap")) *value = g_tune.xcd_remap & 0xFFFFF;
    else if (!strcmp(key, "xcd_stagger")) *value = g_tune.xcd_remap >> 20;
    else if (!strcmp(key, "block")) *value = g_tune.block;
    else if (!strcmp(key, "spmv_variant")) *value = g_tune.spmv_variant;
    else if (!strcmp(key, "csr_lanes")) *value = g_tune.csr_lanes;
    else if (!strcmp(key, "ablate")) *value = g_tune.ablate;
    else if (!strcmp(key, "spmmv_prefetch")) *value = g_tune.spmmv_prefetch;
    else if (!strcmp(key, "spmmv_swizzle")) *value = g_tune.spmmv_swizzle;
    else if (!strcmp(key, "spmmv_reorder")) *value = g_tune.spmmv_reorder;
    else if (!strcmp(key, "spmmv_brick_stride")) *value = g_tune.spmmv_brick_stride;
    else if (!strcmp(key, "spmmv_brick_lines")) *value = g_tune.spmmv_brick_lines;
    else if (!strcmp(key, "spmmv_phase_dp")) *value = g_tune.spmmv_phase_dp;
    else if (!strcmp(key, "tlc_elem")) *value = g_tune.tlc_elem;
    else if (!strcmp(key, "tlc_elem_rows")) *value = g_tune.tlc_elem_rows;
    else if (!strcmp(key, "tlc_elem_seg_rows")) *value = g_tune.tlc_elem_seg_rows;
    else if (!strcmp(key, "tlc_elem_cap")) *value = g_tune.tlc_elem_cap;
    else if (!strcmp(key, "spmmv_stream")) *value = g_tune.spmmv_stream;
    else if (!strcmp(key, "spmmv_stream_xcd")) *value = g_tune.spmmv_stream_xcd;
    else if (!strcmp(key, "spmmv_stream_waves")) *value = g_tune.spmmv_stream_waves;
    else if (!strcmp(key, "spmmv_stream_depth")) *value = g_tune.spmmv_stream_depth;
    else if (!strcmp(key, "spmmv_phased")) *value = g_tune.spmmv_phased;
    else if (!strcmp(key, "spmmv_xcol")) *value = g_tune.spmmv_xcol;
    else if (!strcmp(key, "spmmv_ycol_nt")) *value = g_tune.spmmv_ycol_nt;
    else if (!strcmp(key, "spmmv_xline")) *value = g_tune.spmmv_xline;
    else if (!strcmp(key, "block_plan_device")) *value = g_tune.block_plan_device;
    else if (!strcmp(key, "spmmv_unscramble")) *value = g_tune.spmmv_unscramble;
    else if (!strcmp(key, "spmmv_phase_rows")) *value = g_tune.spmmv_phase_rows;
    else if (!strcmp(key, "spmmv_idx8")) *value = g_tune.spmmv_idx8;
    else if (!strcmp(key, "spmmv_list_plan")) *value = g_tune.spmmv_list_plan;
    else if (!strcmp(key, "sweep")) *value = g_tune.sweep;
    else if (!strcmp(key, "sweep_nbuf")) *value = g_tune.sweep_nbuf;
    else if (!strcmp(key, "sweep_unroll")) *value = g_tune.sweep_unroll;
    else if (!strcmp(key, "sweep_pair")) *value = g_tune.sweep_pair;
    else if (!strcmp(key, "sweep_remap")) *value = g_tune.sweep_remap;
    else if (!strcmp(key, "sweep_wlog")) *value = g_tune.sweep_wlog;
    else if (!strcmp(key, "sweep_tile_rows")) *value = g_tune.sweep_tile_rows;
    else if (!strcmp(key, "sweep_threads")) *value = g_tune.sweep_threads;
    else if (!strcmp(key, "sweep_max_stage")) *value = g_tune.sweep_max_stage;
    else if (!strcmp(key, "raw_plan_cache")) *value = g_tune.raw_plan_cache;
    else if (!strcmp(key, "spmmv_tile_rows")) *value = g_tune.spmmv_tile_rows;
    else if (!strcmp(key, "spmmv_lds_kb")) *value = g_tune.spmmv_lds_kb;
    else if (!strcmp(key, "spmmv_variant")) *value = g_tune.spmmv_variant;
    else if (!strcmp(key, "tail_batch")) *value = g_tune.tail_batch;
    else if (!strcmp(key, "spmmv_unroll")) *value = g_tune.spmmv_unroll;
    else if (!strcmp(key, "tlc")) *value = g_tune.tlc;
    else if (!strcmp(key, "rechunk")) *value = g_tune.rechunk;
    else if (!strcmp(key, "tlc_tile_rows")) *value = g_tune.tlc_tile_rows;
    else if (!strcmp(key, "tlc_auto_tile")) *value = g_tune.tlc_auto_tile;
    else if (!strcmp(key, "tlc_measure_tile")) *value = g_tune.tlc_measure_tile;
    else if (!strcmp(key, "tlc_idx12")) *value = g_tune.tlc_idx12;
    else return uspmv::fail(USPMV_ERR_INVALID, "uspmv_get_tuning: unknown key '%s'", key);
    return USPMV_OK;
}

int uspmv_dmat_upload(const uspmv_scs_t *s, uspmv_dmat_t **out) {
    if (!s || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_upload: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_upload: layout-only struct (its entries already live on the device)");
    if (int rc = require_device()) return rc;
    auto *A = new uspmv_dmat;
    A->C = s->C; A->n_chunks = s->n_chunks; A->n_elements = s->n_elements; A->dtype = s->dtype; A->owns = true;
    A->n_store = (long)(s->n_chunks * s->C);
    const size_t vsz = s->dtype == USPMV_F64 ? 8 : 4;
    void *cp = nullptr, *cl = nullptr, *ci = nullptr, *va = nullptr;
    const size_t ne = (size_t)std::max<int64_t>(s->n_elements, 1);
    hipError_t e;
    if ((e = hipMalloc(&cp, sizeof(int32_t) * (size_t)(s->n_chunks + 1))) != hipSuccess ||
        (e = hipMalloc(&cl, sizeof(int32_t) * (size_t)std::max<int64_t>(s->n_chunks, 1))) != hipSuccess ||
        (e = hipMalloc(&ci, sizeof(int32_t) * ne)) != hipSuccess || (e = hipMalloc(&va, vsz * ne)) != hipSuccess) {
        (void)hipFree(cp); (void)hipFree(cl); (void)hipFree(ci); (void)hipFree(va);
        delete A;
        return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_upload: hipMalloc failed: %s", hipGetErrorString(e));
    }
    A->chunk_ptrs = (const int32_t *)cp; A->chunk_lengths = (const int32_t *)cl;
    A->col_idxs = (const int32_t *)ci; A->values = va;
    e = hipMemcpy(cp, s->chunk_ptrs.data(), sizeof(int32_t) * (size_t)(s->n_chunks + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(cl, s->chunk_lengths.data(), sizeof(int32_t) * (size_t)s->n_chunks, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(ci, s->col_idxs.data(), sizeof(int32_t) * (size_t)s->n_elements, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(va, s->values_ptr(), vsz * (size_t)s->n_elements, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        uspmv_dmat_free(A);
        return uspmv::fail(USPMV_ERR_HIP, "uspmv_dmat_upload: hipMemcpy failed: %s", hipGetErrorString(e));
    }
    *out = A;
    return USPMV_OK;
}

int uspmv_dmat_wrap(int64_t C, int64_t n_chunks, int64_t n_elements, int dtype, const int32_t *d_chunk_ptrs,
                    const int32_t *d_chunk_lengths, const int32_t *d_col_idxs, const void *d_values,
                    uspmv_dmat_t **out) {
    if (!out || C < 1 || n_chunks < 0 || n_elements < 0 || (dtype != USPMV_F64 && dtype != USPMV_F32) ||
        !d_chunk_ptrs || (n_chunks > 0 && !d_chunk_lengths) || (n_elements > 0 && (!d_col_idxs || !d_values)))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_wrap: bad argument");
    auto *A = new uspmv_dmat;
    A->C = C; A->n_chunks = n_chunks; A->n_elements = n_elements; A->dtype = dtype; A->n_store = (long)(n_chunks * C);
    A->chunk_ptrs = d_chunk_ptrs; A->chunk_lengths = d_chunk_lengths; A->col_idxs = d_col_idxs; A->values = d_values;
    A->owns = false;
    *out = A;
    return USPMV_OK;
}

int uspmv_convert_to_scs_device(const uspmv_coo_t *m, int64_t C, int64_t sigma, int dtype, const int32_t *fixed_permutation,
                                int permute_cols, uspmv_scs_t **layout, uspmv_dmat_t **out) {
    if (!m || !layout || !out) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_convert_to_scs_device: NULL argument");
    if (int rc = require_device()) return rc;
    auto *s = new uspmv_scs;
    std::vector<int64_t> row_start;
    if (int rc = uspmv_scs_layout(m, C, sigma, dtype, fixed_permutation, s, &row_start, "uspmv_convert_to_scs_device")) { delete s; return rc; }
    if (row_start.empty() && m->nnz > 0) {
        delete s;
        return uspmv::fail(USPMV_ERR_UNSUPPORTED, "uspmv_convert_to_scs_device: COO entries must be sorted by row "
                                                  "(uspmv_read_mtx and the generators produce that order)");
    }
    if (m->nnz > INT32_MAX) { delete s; return uspmv::fail(USPMV_ERR_OVERFLOW, "uspmv_convert_to_scs_device: nnz exceeds int32"); }
    std::vector<int32_t> rs32(row_start.begin(), row_start.end());
    const int32_t *row_map = fixed_permutation ? fixed_permutation : s->old_to_new_idx.data();
    auto *A = new uspmv_dmat;
    A->C = s->C; A->n_chunks = s->n_chunks; A->n_elements = s->n_elements; A->dtype = dtype; A->owns = true;
    A->n_store = (long)(s->n_chunks * s->C);
    const size_t vsz = dtype == USPMV_F64 ? 8 : 4;
    const size_t ne = (size_t)std::max<int64_t>(s->n_elements, 1), nz = (size_t)std::max<int64_t>(m->nnz, 1);
    void *cp = nullptr, *cl = nullptr, *ci = nullptr, *va = nullptr;
    int32_t *dI = nullptr, *dJ = nullptr, *drs = nullptr, *dmap = nullptr, *dperm = nullptr;
    double *dV = nullptr;
    hipError_t e = hipSuccess;
    auto up = [&](const void *h, size_t bytes, void **d) {
        if (e != hipSuccess) return;
        e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    up(s->chunk_ptrs.data(), 4 * s->chunk_ptrs.size(), &cp);
    up(s->chunk_lengths.data(), 4 * s->chunk_lengths.size(), &cl);
    if (e == hipSuccess) e = hipMalloc(&ci, 4 * ne);
    if (e == hipSuccess) e = hipMalloc(&va, vsz * ne);
    // padding: value 0, column 0 -- which permute_scs_cols maps like any other local column (code/utilities.hpp:1820-1826)
    const int pad_col = (permute_cols && m->n_rows > 0) ? s->old_to_new_idx[0] : 0;
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)ci, pad_col, ne, nullptr);
    if (e == hipSuccess) e = hipMemsetAsync(va, 0, vsz * ne, nullptr);
    up(m->I.data(), 4 * (size_t)m->nnz, (void **)&dI);
    up(m->J.data(), 4 * (size_t)m->nnz, (void **)&dJ);
    up(m->values.data(), 8 * (size_t)m->nnz, (void **)&dV);
    up(rs32.data(), 4 * rs32.size(), (void **)&drs);
    up(row_map, 4 * (size_t)m->n_rows, (void **)&dmap);
    if (permute_cols) up(s->old_to_new_idx.data(), 4 * (size_t)m->n_rows, (void **)&dperm);
    A->chunk_ptrs = (const int32_t *)cp; A->chunk_lengths = (const int32_t *)cl; A->col_idxs = (const int32_t *)ci; A->values = va;
    if (e == hipSuccess && m->nnz > 0) {
        const unsigned grid = (unsigned)((nz + 255) / 256);
        if (dtype == USPMV_F64)
            hipLaunchKernelGGL(scs_fill_kernel<double>, dim3(grid), dim3(256), 0, nullptr, (long)m->nnz, (int)C, (int)m->n_rows, dI, dJ, dV,
                               drs, dmap, dperm, (const int *)cp, (int *)ci, (double *)va);
        else
            hipLaunchKernelGGL(scs_fill_kernel<float>, dim3(grid), dim3(256), 0, nullptr, (long)m->nnz, (int)C, (int)m->n_rows, dI, dJ, dV,
                               drs, dmap, dperm, (const int *)cp, (int *)ci, (float *)va);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    (void)hipFree(dI); (void)hipFree(dJ); (void)hipFree(dV); (void)hipFree(drs); (void)hipFree(dmap); (void)hipFree(dperm);
    if (e != hipSuccess) {
        uspmv_dmat_free(A); delete s;
        return uspmv::fail(USPMV_ERR_HIP, "uspmv_convert_to_scs_device: %s", hipGetErrorString(e));
    }
    *layout = s;
    *out = A;
    return USPMV_OK;
}

int uspmv_dmat_plan_addresses(const uspmv_dmat_t *A, uint64_t addr[8]) {
    if (!A || !addr) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_plan_addresses: NULL argument");
    const uspmv_dmat *M = A->alt ? A->alt : A;
    addr[0] = (uint64_t)(uintptr_t)M->tlc_col16; addr[1] = (uint64_t)(uintptr_t)M->tlc_lines; addr[2] = (uint64_t)(uintptr_t)M->tlc_line_ptr;
    addr[3] = (uint64_t)(uintptr_t)M->tlc_c16_ptrs; addr[4] = (uint64_t)(uintptr_t)M->values; addr[5] = (uint64_t)(uintptr_t)M->col_idxs;
    addr[6] = (uint64_t)(uintptr_t)M->chunk_ptrs; addr[7] = (uint64_t)(uintptr_t)M->chunk_lengths;
    return USPMV_OK;
}

int uspmv_dmat_meta(const uspmv_dmat_t *A, int64_t meta[4]) {
    if (!A || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_meta: NULL argument");
    meta[0] = A->C; meta[1] = A->n_chunks; meta[2] = A->n_elements; meta[3] = A->dtype;
    return USPMV_OK;
}

int uspmv_dmat_download(const uspmv_dmat_t *A, int32_t *chunk_ptrs, int32_t *chunk_lengths, int32_t *col_idxs, void *values) {
    if (int rc = check_dmat(A, "uspmv_dmat_download")) return rc;
    if (int rc = require_device()) return rc;
    const size_t vsz = A->dtype == USPMV_F64 ? 8 : 4;
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess && chunk_ptrs) e = hipMemcpy(chunk_ptrs, A->chunk_ptrs, 4 * (size_t)(A->n_chunks + 1), hipMemcpyDeviceToHost);
    if (e == hipSuccess && chunk_lengths) e = hipMemcpy(chunk_lengths, A->chunk_lengths, 4 * (size_t)A->n_chunks, hipMemcpyDeviceToHost);
    if (e == hipSuccess && col_idxs) e = hipMemcpy(col_idxs, A->col_idxs, 4 * (size_t)A->n_elements, hipMemcpyDeviceToHost);
    if (e == hipSuccess && values) e = hipMemcpy(values, A->values, vsz * (size_t)A->n_elements, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return uspmv::fail(USPMV_ERR_HIP, "uspmv_dmat_download: %s", hipGetErrorString(e));
    return USPMV_OK;
}

static void sw_release(uspmv_dmat_t *A);
static int sweep_plan_install_device(uspmv_dmat_t *A, uspmv_dmat_t *B, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep, const char *who);
static int sweep_plan_install(uspmv_dmat_t *A, uspmv_dmat_t *B, const uspmv_scs_t *s, const uspmv_scs_t *sB, int wlog, int tile_rows,
                              int64_t *n_tiles, int64_t *n_sweep, const char *who);
static void tlc_release(uspmv_dmat_t *A) {
    (void)hipFree(A->tlc_line_ptr); (void)hipFree(A->tlc_lines); (void)hipFree(A->tlc_c16_ptrs); (void)hipFree(A->tlc_col16);
    (void)hipFree(A->tlc_c12_ptrs); (void)hipFree(A->tlc_col12);
    A->tlc_line_ptr = A->tlc_lines = nullptr; A->tlc_c16_ptrs = nullptr; A->tlc_col16 = nullptr; A->tlc_c12_ptrs = A->tlc_col12 = nullptr;
    A->tlc_elem = false;
    (void)hipFree(A->tlc_values); (void)hipFree(A->tlc_row_map); (void)hipFree(A->tlc_cols); A->tlc_values = nullptr; A->tlc_row_map = A->tlc_cols = nullptr;
    A->tlc = false; A->tlc_plan_id = 0;
}

// rows per tile of the next tile-local-column plan (g_tune.tlc_tile_rows = 0: by kind)
static int plan_tile_rows(bool ap) { return g_tune.tlc_tile_rows ? g_tune.tlc_tile_rows : (ap ? 512 : 256); }

// Rows per tile by what the 256-row plan turned out to be (tuning tlc_auto_tile, default on; only when tlc_tile_rows is 0).  The x lines
// of a tile live in LDS (128 B each): when the largest 256-row tile needs more than 250 of them, at most 4 workgroups = 16 waves fit a
// CU, too few to cover the staging latency, and every line is fetched by several neighbouring tiles.  1024-row tiles (or 512-row ones)
// fetch each line fewer times and keep 32 (16) waves per CU when their lines still fit; they are taken when they stage >= 99 % of the
// tiles.  Measured (tools/tile_rows_sweep.py, profiles/r03/tile_rows_sweep.txt): KKT N = 200 0.82 -> 0.73 ms, banded 30 per row over
// +-2000 columns 0.27 -> 0.22 ms; matrices whose 256-row tiles need <= 217 lines (all the stencils) are fastest at 256 and stay there.
static bool tile_rows_grow(int rows, int lines_used) { return g_tune.tlc_auto_tile && g_tune.tlc_tile_rows == 0 && rows == 256 && lines_used > 250; }
static bool tile_rows_accept(int64_t n_tiles, int64_t n_staged) { return n_staged * 100 >= n_tiles * 99; }

static int device_plan_install_rows(uspmv_dmat_t *A, uspmv_dmat_t *B, int max_lines, int R, int64_t *n_tiles, int64_t *n_staged, const char *who);

// For LARGE single structs the rows per tile are MEASURED (tuning tlc_measure_tile, default on; only when tlc_tile_rows is 0): the plan
// is built on the device for 256, 512 and 1024 rows (two passes over the column indices each), the kernel timed three times on a zero
// vector, and a larger tile kept when it is more than 3 % ahead of 256.  Why: which size wins depends on how far apart the x lines of
// neighbouring tiles lie -- the 27-point stencil on 253^3 is fastest at 256 rows, the same stencil on 304^3 (planes of 739 instead of
// 512 KB: more of the x lines miss the XCD's L2) at 512 (1.249 against 1.341 ms, profiles/r03/tile_rows_sweep.txt).  The choice is
// remembered per (shape, size) for the life of the process, so the host and the device planner of one matrix agree.  0 = no opinion.
static int measured_tile_rows(uspmv_dmat_t *A, uspmv_dmat_t *B, int max_lines, const char *who) {
    if (uspmv_dev::tl_measure_off > 0) return 0;
    // one measurement at a time, and the verdict table only read / written under the lock (the verdict is keyed on the struct's shape and
    // size, not its content: two matrices with equal counts share it -- the price of host and device planner of ONE matrix agreeing)
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    if (!g_tune.tlc_measure_tile || g_tune.tlc_tile_rows != 0 || A->alt || A->C > 256 || 256 % A->C != 0) return 0;
    if (A->n_chunks * A->C < (int64_t)1 << 20) return 0;
    if (B && (B->alt || B->C != A->C || B->n_chunks != A->n_chunks || A->dtype != USPMV_F64 || B->dtype != USPMV_F32)) return 0;
    struct Key { int64_t nc, ne, ne2, C; int dtype, ml; };
    static std::vector<std::pair<Key, int>> seen;
    const int64_t ne2 = B ? B->n_elements : -1;
    for (auto &kv : seen)
        if (kv.first.nc == A->n_chunks && kv.first.ne == A->n_elements && kv.first.ne2 == ne2 && kv.first.C == A->C && kv.first.dtype == A->dtype && kv.first.ml == max_lines)
            return kv.second;
    const size_t vsz = A->dtype == USPMV_F64 ? 8 : 4;
    void *x = nullptr, *y = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int best = 0;
    auto done = [&]() {
        (void)hipFree(x); (void)hipFree(y);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (A->tlc) tlc_release(A);
        if (B && B->tlc) tlc_release(B);
    };
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { done(); (void)hipGetLastError(); return 0; }
    // the size to beat: 256 rows for one struct, 512 for an ap[dp_sp] pair (two entry streams per row, profiles/r02/ap_tile_rows.txt);
    // it is timed first and once more at the end (the first candidate may have met a cold clock)
    const int base = B ? 512 : 256;
    const int order[4] = {base, base == 256 ? 512 : 256, 1024, base};
    float tmin[3] = {0, 0, 0};                                  // best time seen for 256 / 512 / 1024 rows (0: not usable)
    auto slot_of = [](int R) { return R == 256 ? 0 : R == 512 ? 1 : 2; };
    for (int k = 0; k < 4; ++k) {
        const int R = order[k], slot = slot_of(R), bs = slot_of(base);
        if (k == 3) {                                           // re-check the base only when something is about to beat it
            bool beaten = false;
            for (int o = 0; o < 3; ++o) beaten |= o != bs && tmin[o] > 0 && tmin[bs] > 0 && tmin[o] < 0.97f * tmin[bs];
            if (!beaten) break;
        }
        int64_t nt = 0, ns = 0;
        if (device_plan_install_rows(A, B, max_lines, R, &nt, &ns, who) != USPMV_OK || !A->tlc) { (void)hipGetLastError(); continue; }
        if (!tile_rows_accept(nt, ns)) continue;
        if (!x) {
            const size_t xb = vsz * (size_t)std::max<int64_t>(A->tlc_x_len + 16, 16), yb = vsz * (size_t)std::max<int64_t>(A->n_chunks * A->C, 1);
            if (hipMalloc(&x, xb) != hipSuccess || hipMalloc(&y, yb) != hipSuccess || hipMemset(x, 0, xb) != hipSuccess) { done(); (void)hipGetLastError(); return 0; }
        }
        float ms = 0;
        bool ok = true;
        for (int rep = 0; rep < 2 && ok; ++rep) {             // (first round warms up)
            ok = hipEventRecord(e0, nullptr) == hipSuccess;
            for (int l = 0; l < 3 && ok; ++l) {
                if (B) ok = launch_spmv_ap(A, B, (const double *)x, nullptr, (double *)y, nullptr) == USPMV_OK;
                else ok = (A->dtype == USPMV_F64 ? launch_spmv_tlc<double>(A, nullptr, (long)A->tlc_n_tiles, (const double *)x, (double *)y, nullptr)
                                                 : launch_spmv_tlc<float>(A, nullptr, (long)A->tlc_n_tiles, (const float *)x, (float *)y, nullptr)) == USPMV_OK;
            }
            ok = ok && hipEventRecord(e1, nullptr) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess;
        }
        if (!ok) { (void)hipGetLastError(); continue; }
        if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] measured tile size%s: %d rows -> %.4f ms per SpMV (%lld of %lld tiles staged, %d lines at most)\n", B ? " (ap pair)" : "",
                                             R, ms / 3, (long long)ns, (long long)nt, A->tlc_max_lines);
        tmin[slot] = tmin[slot] > 0 ? std::min(tmin[slot], ms) : ms;
    }
    // the base size unless another one is more than 3 % ahead of it (the fastest of those that are)
    {
        const int bs = slot_of(base);
        float tbest = tmin[bs] > 0 ? 0.97f * tmin[bs] : 1e30f;
        best = tmin[bs] > 0 ? base : 0;
        const int sizes[3] = {256, 512, 1024};
        for (int o = 0; o < 3; ++o)
            if (o != bs && tmin[o] > 0 && tmin[o] < tbest) { best = sizes[o]; tbest = tmin[o]; }
    }
    done();
    seen.push_back({Key{A->n_chunks, A->n_elements, ne2, A->C, A->dtype, max_lines}, best});
    return best;
}

// The plan's local indices once more in 12 bits (single structs whose tiles list at most 256 lines, i.e. local indices below 4096; even C):
// what scs_spmv_tlc then streams instead of the 16-bit array -- 1.5 instead of 2 bytes per non-zero.  The 16-bit array stays (the
// adaptive-precision kernels, uspmv_dmat_plan_download and the plan digests read it).  cl: the chunk lengths when the caller has them on
// the host, else they are copied back (4 bytes per chunk).
static int tlc_pack12(uspmv_dmat_t *A, const std::vector<int32_t> *cl, const char *who) {
    if (!A->tlc || !g_tune.tlc_idx12 || (A->tlc_elem ? A->tlc_max_lines > 4096 : A->tlc_max_lines > 256) || A->C < 2 || A->C % 2 != 0 || A->n_chunks < 1) return USPMV_OK;
    std::vector<int32_t> own;
    if (!cl || (int64_t)cl->size() != A->n_chunks) {
        own.resize((size_t)A->n_chunks);
        HIP_TRY(hipMemcpy(own.data(), A->chunk_lengths, 4 * (size_t)A->n_chunks, hipMemcpyDeviceToHost));
        cl = &own;
    }
    const int64_t C = A->C, nc = A->n_chunks;
    std::vector<uint32_t> p12((size_t)nc + 1);
    int64_t tot = 0;                                             // dwords
    for (int64_t c = 0; c < nc; ++c) {
        p12[(size_t)c] = (uint32_t)tot;
        const int64_t ngt = ((int64_t)(*cl)[(size_t)c] + 3) / 4;
        tot += (ngt / 2) * 3 * C + (ngt & 1) * (C + C / 2);
        if (tot > (int64_t)UINT32_MAX) return USPMV_OK;          // (too large for 32-bit offsets: the 16-bit array serves)
    }
    p12[(size_t)nc] = (uint32_t)tot;
    hipError_t e = hipMalloc((void **)&A->tlc_c12_ptrs, 4 * ((size_t)nc + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->tlc_col12, 4 * (size_t)std::max<int64_t>(tot, 1));
    if (e == hipSuccess) e = hipMemcpy(A->tlc_c12_ptrs, p12.data(), 4 * ((size_t)nc + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess && uspmv_dev::launch_plan_pack12(A, A->tlc_c16_ptrs, A->tlc_col16, A->tlc_c12_ptrs, A->tlc_col12, nullptr) != USPMV_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        (void)hipFree(A->tlc_c12_ptrs); (void)hipFree(A->tlc_col12); A->tlc_c12_ptrs = A->tlc_col12 = nullptr;
        return uspmv::fail(USPMV_ERR_HIP, "%s: packing the local indices to 12 bits failed: %s", who, hipGetErrorString(e));
    }
    // Keep it?  Rows of a dozen entries gain or lose a per cent either way (one more load instruction per row for an odd last group), long
    // rows gain 1-15 % depending on matrix and box (profiles/r04/idx12_probe_*.txt: the 253^3 stencil 0.763 -> 0.691 ms on a slow box, 0.710 ->
    // 0.700 on a fast one; 304^3 between -15 % and +2 %).  The rule is a fixed one -- mean row length >= 8 -- and not a timing on the spot
    // (which was built first): a bench run, its counter passes and its profiler run must execute the same kernel, and a 1-2 % verdict
    // flips under a profiler's overhead.  "tlc_idx12" 2 keeps it regardless, 0 never builds it.
    if (g_tune.tlc_idx12 != 2 && (double)A->n_elements < 8.0 * (double)(nc * C)) {
        (void)hipFree(A->tlc_c12_ptrs); (void)hipFree(A->tlc_col12); A->tlc_c12_ptrs = A->tlc_col12 = nullptr;
    }
    return USPMV_OK;
}

// A quick look before an element plan is built in full (a sort per tile over all entries): of ~64 tiles spread over the struct, how many list more distinct
// columns than `cap`?  true: more than a tenth of them -- the element plan would be turned down anyway (wide irregular rows: the sweep's matrices).
static double elements_over_cap_frac(const uspmv_scs_t *s, int cap, int tile_rows) {
    const int64_t C = s->C, T = std::max<int64_t>(1, tile_rows / C), nt = (s->n_chunks + T - 1) / T;
    const int64_t step = std::max<int64_t>(1, nt / 64);
    int64_t seen = 0, over = 0;
    std::vector<int32_t> cols;
    for (int64_t t = step / 2; t < nt; t += step) {
        const int64_t c0 = t * T, c1 = std::min<int64_t>(c0 + T, s->n_chunks);
        cols.assign(s->col_idxs.begin() + s->chunk_ptrs[(size_t)c0], s->col_idxs.begin() + s->chunk_ptrs[(size_t)c1]);
        std::sort(cols.begin(), cols.end());
        const int64_t n = (int64_t)(std::unique(cols.begin(), cols.end()) - cols.begin());
        ++seen; over += n > cap;
    }
    return seen > 0 ? (double)over / (double)seen : 1.0;
}
static bool elements_over_cap(const uspmv_scs_t *s, int cap, int tile_rows) { return elements_over_cap_frac(s, cap, tile_rows) > 0.1; }

int uspmv_dmat_optimize(uspmv_dmat_t *A, const uspmv_scs_t *s, int max_lines, int64_t *n_tiles, int64_t *n_staged) {
    if (!A || !s) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize: layout-only struct; the plan builder needs the host column indices");
    if (A->C != s->C || A->n_chunks != s->n_chunks || A->dtype != s->dtype)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize: handle and host struct do not describe the same matrix");
    if (int rc = require_device()) return rc;
    if (A->tlc) tlc_release(A);
    if (A->alt) { uspmv_dmat_free(A->alt); A->alt = nullptr; }
    if (s->C < 32 && 32 % s->C == 0 && g_tune.rechunk) {
        // narrow chunks (incl. crs = C 1): run on an internal C = 32 re-chunking with the same row order
        uspmv_scs r;
        int rc = uspmv_scs_rechunk32(s, &r);
        if (rc == USPMV_OK && (double)r.n_elements <= 1.25 * (double)std::max<int64_t>(s->n_elements, 1) + 4096) {
            uspmv_dmat_t *alt = nullptr;
            if (int rc2 = uspmv_dmat_upload(&r, &alt)) return rc2;
            alt->n_store = (long)(s->n_chunks * s->C);      // y of the caller has only the original padded rows
            rc = uspmv_dmat_optimize(alt, &r, max_lines, n_tiles, n_staged);   // (C = 32: does not re-enter this branch)
            if (rc) { uspmv_dmat_free(alt); return rc; }
            A->alt = alt;
            return USPMV_OK;
        }
    }
    const bool own_budget = max_lines > 0;                     // (a caller with a line budget of its own keeps the line plan: no element fallback)
    if (max_lines <= 0) max_lines = 512;                       // 64 KiB of doubles: 2 workgroups per CU at worst
    const int cap = (int)(160 * 1024 / (16 * (s->dtype == USPMV_F64 ? 8 : 4)));
    if (max_lines > cap) max_lines = cap;
    uspmv_tlc_plan p;
    const int R_meas = measured_tile_rows(A, nullptr, max_lines, "uspmv_dmat_optimize");
    if (int rc = uspmv_build_tlc_plan(s, nullptr, max_lines, R_meas ? R_meas : plan_tile_rows(false), &p)) return rc;
    if (!R_meas && p.valid && tile_rows_grow(p.tile_rows, p.max_lines_used))
        for (int R : {1024, 512}) {
            uspmv_tlc_plan q;
            if (int rc = uspmv_build_tlc_plan(s, nullptr, max_lines, R, &q)) return rc;
            if (q.valid && tile_rows_accept(q.n_tiles, q.n_staged_tiles)) { p = std::move(q); break; }
        }
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_staged) *n_staged = p.valid ? p.n_staged_tiles : 0;
    if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] tlc plan: tile_rows=%d tiles=%lld staged=%lld max_lines=%d lines_total=%zu col16=%zu\n",
                                         p.tile_rows, (long long)p.n_tiles, (long long)p.n_staged_tiles, p.max_lines_used, p.tile_lines.size(), p.col16.size());
    if (A->sw) sw_release(A);
    bool elem = false;
    if (((!own_budget && (!p.valid || p.n_staged_tiles * 10 < p.n_tiles * 9)) || g_tune.tlc_elem == 2) && g_tune.tlc_elem && uspmv_dev::tl_measure_off == 0) {   // (2: measurement aid, always try)
        // columns scattered over many lines (x in a numbering that is only loosely related to the rows'): the line plan leaves a tenth of the tiles or
        // more to the gather path.  List the tile's distinct ELEMENTS instead -- taken when (nearly) every tile fits and an element serves four
        // entries or more on average (else the line plan stays, or the column-window sweep takes over below).  Measured (tools/numbering_probe.py,
        // profiles/r04/numbering_probe_*.txt): 27-point x 3 dof stencil with x renumbered at random inside blocks of 1 000 / 5 000 / 20 000 nodes 0.97 / 0.90 /
        // 0.89 of the roofline against 0.74 (line plan, 69 % of the tiles staged) / 0.70 / 0.61 (sweep); 1 dof, 4.2 entries per element: 0.65 against 0.59;
        // on a regular numbering the line plan is 20 % ahead (0.683 against 0.819 ms on the 253^3 stencil), which is why this is a fallback only.
        uspmv_tlc_plan q;
        const int ecap = std::min(g_tune.tlc_elem_cap, (int)(64 * 1024 / (s->dtype == USPMV_F64 ? 8 : 4)));
        if (!elements_over_cap(s, ecap, 256))
            if (int rc = uspmv_build_tlc_plan(s, nullptr, ecap, 256, &q, /*line_shift=*/0)) return rc;
        if (q.valid && tile_rows_accept(q.n_tiles, q.n_staged_tiles) && ((double)q.tile_lines.size() * 4.0 <= (double)s->n_elements || g_tune.tlc_elem == 2)) {
            p = std::move(q); elem = true;
            if (n_tiles) *n_tiles = p.n_tiles;
            if (n_staged) *n_staged = p.n_staged_tiles;
            if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] tlc plan over single x elements: tiles=%lld staged=%lld max_elements=%d elements_total=%zu (%.1f entries per element)\n",
                                                 (long long)p.n_tiles, (long long)p.n_staged_tiles, p.max_lines_used, p.tile_lines.size(), (double)s->n_elements / (double)std::max<size_t>(p.tile_lines.size(), 1));
        }
    }
    // ... and when the ROWS of a tile are scattered as well (rows and columns renumbered alike: a tile of 256 consecutive rows is no compact piece of the
    // mesh any more): deal the rows to the tiles by the matrix graph first, as the block plan does (uspmv_scs_reorder_rows mode 4: rows change places
    // only with rows of equal-length chunks, every row keeps its slot sequence), then the element plan on that order -- a private copy of the values
    // (8 / 4 bytes per element of HBM), of the column indices (for the few tiles that do not stage) and a row map for y.
    uspmv_scs rr;
    std::vector<int32_t> rr_map;
    bool reordered = false;
    if (!elem && !own_budget && (!p.valid || p.n_staged_tiles * 10 < p.n_tiles * 9) && g_tune.tlc_elem && g_tune.tlc_elem_rows && uspmv_dev::tl_measure_off == 0 && s->n_rows == s->n_cols) {
        const int ecap = std::min(g_tune.tlc_elem_cap, (int)(64 * 1024 / (s->dtype == USPMV_F64 ? 8 : 4)));
        // first with the clusters confined to segments of tlc_elem_seg_rows rows (64 Ki: many segments in parallel, and a trial on a sample of them that stops
        // irregular matrices early); when that leaves some, but not most, of the sampled tiles over the cap -- related rows further apart than a segment --
        // once more with segments of 2^20 rows (a second or more of clustering per million rows on few threads: only where it looks promising)
        const int64_t seg_stage[2] = {(int64_t)g_tune.tlc_elem_seg_rows, (int64_t)1 << 20};
        for (int stage = 0; stage < 2 && !reordered; ++stage) {
            if (stage == 1 && seg_stage[1] <= seg_stage[0]) break;
            if (uspmv_scs_reorder_rows(s, g_tune.tlc_elem_rows == 4 ? 4 : 2, &rr, &rr_map, g_tune.tlc_elem_rows == 4 ? 64 : 256, seg_stage[stage]) != 1) break;
            const double over = elements_over_cap_frac(&rr, ecap, 256);
            uspmv_tlc_plan q;
            if (over <= 0.1)
                if (int rc = uspmv_build_tlc_plan(&rr, nullptr, ecap, 256, &q, /*line_shift=*/0)) return rc;
            if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] element plan on the graph-dealt rows (segments of %lld rows): %.0f %% of the sampled tiles over the cap; valid=%d tiles=%lld staged=%lld max_elements=%d (cap %d) elements_total=%zu\n",
                                                 (long long)seg_stage[stage], 100.0 * over, (int)q.valid, (long long)q.n_tiles, (long long)q.n_staged_tiles, q.max_lines_used, ecap, q.tile_lines.size());
            // (19 of 20 tiles staged is enough here: what would run instead -- sweep or gather kernel -- is 2 x slower on such matrices)
            if (q.valid && q.n_staged_tiles * 20 >= q.n_tiles * 19 && (double)q.tile_lines.size() * 4.0 <= (double)s->n_elements) {
                p = std::move(q); elem = true; reordered = true;
                if (n_tiles) *n_tiles = p.n_tiles;
                if (n_staged) *n_staged = p.n_staged_tiles;
                if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] tlc plan over single x elements, rows dealt to the tiles by the matrix graph: tiles=%lld max_elements=%d elements_total=%zu (%.1f entries per element)\n",
                                                     (long long)p.n_tiles, p.max_lines_used, p.tile_lines.size(), (double)s->n_elements / (double)std::max<size_t>(p.tile_lines.size(), 1));
            } else if (over > 0.6) break;                    // most tiles far over the cap: larger segments will not repair that
        }
    }
    if (!elem && (!p.valid || p.n_staged_tiles * 2 < p.n_tiles) && g_tune.sweep) {
        // wide, irregular rows: most tiles touch too many x lines to stage them.  Try the column-window sweep; it takes over
        // when it covers at least half of the rows.
        int64_t swt = 0, sws = 0;
        if (int rc = sweep_plan_install(A, nullptr, s, nullptr, 0, 0, &swt, &sws, "uspmv_dmat_optimize")) return rc;
        if (A->sw && sws * 2 >= swt) return USPMV_OK;
        if (A->sw) sw_release(A);
    }
    if (!p.valid) return USPMV_OK;                              // nothing worth staging: plain kernel stays
    auto up = [&](const void *h, size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(p.tile_line_ptr.data(), p.tile_line_ptr.size() * 4, (void **)&A->tlc_line_ptr);
    if (e == hipSuccess) e = up(p.tile_lines.data(), p.tile_lines.size() * 4, (void **)&A->tlc_lines);
    if (e == hipSuccess) e = up(p.c16_ptrs.data(), p.c16_ptrs.size() * 4, (void **)&A->tlc_c16_ptrs);
    if (e == hipSuccess) e = up(p.col16.data(), p.col16.size() * 2, (void **)&A->tlc_col16);
    if (e == hipSuccess && reordered) {
        e = up(rr.values_ptr(), (size_t)rr.n_elements * (rr.dtype == USPMV_F64 ? 8 : 4), &A->tlc_values);
        if (e == hipSuccess) e = up(rr_map.data(), rr_map.size() * 4, (void **)&A->tlc_row_map);
        if (e == hipSuccess && p.n_staged_tiles < p.n_tiles) e = up(rr.col_idxs.data(), (size_t)rr.n_elements * 4, (void **)&A->tlc_cols);
    }
    if (e != hipSuccess) {
        tlc_release(A);
        return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize: device copy failed: %s", hipGetErrorString(e));
    }
    A->tlc = true; A->tlc_tile_rows = p.tile_rows; A->tlc_max_lines = p.max_lines_used; A->tlc_x_len = p.x_len_min; A->tlc_n_tiles = p.n_tiles;
    A->tlc_staged = p.n_staged_tiles; A->tlc_elem = elem;
    return tlc_pack12(A, &s->chunk_lengths, "uspmv_dmat_optimize");
}

// 16-bit index offsets per chunk from the chunk lengths (O(n_chunks) on the host); false: too large for 32-bit offsets
static bool c16_offsets(const std::vector<int32_t> &cl, int64_t C, std::vector<uint32_t> *c16p, int64_t *tot16) {
    const int64_t nc = (int64_t)cl.size();
    c16p->assign((size_t)nc + 1, 0);
    int64_t tot = 0;
    for (int64_t c = 0; c < nc; ++c) {
        (*c16p)[(size_t)c] = (uint32_t)tot;
        tot += ((int64_t)(cl[(size_t)c] + 3) / 4) * 4 * C;
        if (tot > (int64_t)UINT32_MAX) return false;
    }
    (*c16p)[(size_t)nc] = (uint32_t)tot;
    *tot16 = tot;
    return true;
}

// the tile-local-column plan of A (and of the pair A + B sharing one line list when B != nullptr), built on the device
// ... with the rows per tile chosen as uspmv_dmat_optimize chooses them (measured_tile_rows / tile_rows_grow above)
static int device_plan_install(uspmv_dmat_t *A, uspmv_dmat_t *B, int max_lines, int64_t *n_tiles, int64_t *n_staged, const char *who) {
    if (!B) {
        const int ml = std::min(std::min(max_lines <= 0 ? 512 : max_lines, (int)(160 * 1024 / (16 * (A->dtype == USPMV_F64 ? 8 : 4)))), 4096);
        if (const int R_meas = measured_tile_rows(A, nullptr, ml, who)) return device_plan_install_rows(A, nullptr, max_lines, R_meas, n_tiles, n_staged, who);
    } else {
        const int ml = std::min(std::min(max_lines <= 0 ? 512 : max_lines, (int)(160 * 1024 / (16 * 8))), 1280);
        if (const int R_meas = measured_tile_rows(A, B, ml, who)) return device_plan_install_rows(A, B, max_lines, R_meas, n_tiles, n_staged, who);
    }
    const int R0 = plan_tile_rows(B != nullptr);
    if (int rc = device_plan_install_rows(A, B, max_lines, R0, n_tiles, n_staged, who)) return rc;
    if (B || !A->tlc || !tile_rows_grow(R0, A->tlc_max_lines)) return USPMV_OK;
    for (int R : {1024, 512}) {
        int64_t nt = 0, ns = 0;
        if (int rc = device_plan_install_rows(A, nullptr, max_lines, R, &nt, &ns, who)) return rc;
        if (A->tlc && tile_rows_accept(nt, ns)) { if (n_tiles) *n_tiles = nt; if (n_staged) *n_staged = ns; return USPMV_OK; }
    }
    return device_plan_install_rows(A, nullptr, max_lines, R0, n_tiles, n_staged, who);
}

static int device_plan_install_rows(uspmv_dmat_t *A, uspmv_dmat_t *B, int max_lines, const int R, int64_t *n_tiles, int64_t *n_staged, const char *who) {
    if (A->tlc) tlc_release(A);
    if (B && B->tlc) tlc_release(B);
    if (n_tiles) *n_tiles = 0;
    if (n_staged) *n_staged = 0;
    const int64_t C = A->C, nc = A->n_chunks;
    if (C > 256 || 256 % C != 0 || nc < 1) return USPMV_OK;                    // shape without a plan
    if (max_lines <= 0) max_lines = 512;
    max_lines = std::min(max_lines, (int)(160 * 1024 / (16 * (A->dtype == USPMV_F64 ? 8 : 4))));
    max_lines = std::min(max_lines, B ? 1280 : 4096);
    const int64_t T = R / C, nt = (nc + T - 1) / T;
    std::vector<int32_t> cl((size_t)nc);
    std::vector<uint32_t> c16p, c16p_b;
    int64_t tot16 = 0, tot16_b = 0;
    HIP_TRY(hipMemcpy(cl.data(), A->chunk_lengths, 4 * (size_t)nc, hipMemcpyDeviceToHost));
    if (!c16_offsets(cl, C, &c16p, &tot16)) return USPMV_OK;
    if (B) {
        HIP_TRY(hipMemcpy(cl.data(), B->chunk_lengths, 4 * (size_t)nc, hipMemcpyDeviceToHost));
        if (!c16_offsets(cl, C, &c16p_b, &tot16_b)) return USPMV_OK;
    }
    int *d_n = nullptr, *d_max = nullptr;
    hipError_t e = hipMalloc((void **)&d_n, 4 * (size_t)nt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_max, 4);
    if (e == hipSuccess) e = hipMemset(d_max, 0, 4);
    if (e != hipSuccess) { (void)hipFree(d_n); (void)hipFree(d_max); return uspmv::fail(USPMV_ERR_ALLOC, "%s: %s", who, hipGetErrorString(e)); }
    int rc = launch_plan_count(A, (long)nt, max_lines, d_n, d_max, nullptr, B, R);
    std::vector<int32_t> lp((size_t)nt + 1, 0);
    int max_col = 0;
    if (!rc) {
        e = hipMemcpy(lp.data() + 1, d_n, 4 * (size_t)nt, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(&max_col, d_max, 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    }
    (void)hipFree(d_n); (void)hipFree(d_max);
    if (rc) return rc;
    int64_t staged = 0, total = 0;
    int used = 0;
    for (int64_t t = 0; t < nt; ++t) {
        const int n = lp[(size_t)t + 1];
        staged += n > 0; used = std::max(used, n);
        total += n;
        if (total > INT32_MAX) return USPMV_OK;
        lp[(size_t)t + 1] = (int32_t)total;
    }
    if (n_tiles) *n_tiles = nt;
    if (n_staged) *n_staged = staged;
    if (staged == 0) return USPMV_OK;
    e = hipMalloc((void **)&A->tlc_line_ptr, 4 * ((size_t)nt + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->tlc_lines, 4 * (size_t)std::max<int64_t>(total, 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->tlc_c16_ptrs, 4 * ((size_t)nc + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->tlc_col16, 2 * (size_t)std::max<int64_t>(tot16, 1));
    if (e == hipSuccess) e = hipMemset(A->tlc_col16, 0, 2 * (size_t)std::max<int64_t>(tot16, 1));   // padded slots: index 0
    if (e == hipSuccess) e = hipMemcpy(A->tlc_line_ptr, lp.data(), 4 * ((size_t)nt + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(A->tlc_c16_ptrs, c16p.data(), 4 * ((size_t)nc + 1), hipMemcpyHostToDevice);
    if (B) {
        if (e == hipSuccess) e = hipMalloc((void **)&B->tlc_c16_ptrs, 4 * ((size_t)nc + 1));
        if (e == hipSuccess) e = hipMalloc((void **)&B->tlc_col16, 2 * (size_t)std::max<int64_t>(tot16_b, 1));
        if (e == hipSuccess) e = hipMemset(B->tlc_col16, 0, 2 * (size_t)std::max<int64_t>(tot16_b, 1));
        if (e == hipSuccess) e = hipMemcpy(B->tlc_c16_ptrs, c16p_b.data(), 4 * ((size_t)nc + 1), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && launch_plan_write(A, (long)nt, A->tlc_line_ptr, A->tlc_c16_ptrs, A->tlc_lines, A->tlc_col16, nullptr, B,
                                             B ? B->tlc_c16_ptrs : nullptr, B ? B->tlc_col16 : nullptr, R) != USPMV_OK)
        e = hipErrorUnknown;
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) {
        tlc_release(A);
        if (B) tlc_release(B);
        return uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    }
    static uint64_t next_dev_plan_id = (uint64_t)1 << 40;
    const uint64_t id = B ? next_dev_plan_id++ : 0;
    for (uspmv_dmat_t *M : {A, B}) {
        if (!M) continue;
        M->tlc = true; M->tlc_tile_rows = R; M->tlc_max_lines = used; M->tlc_x_len = (int64_t)max_col + 1; M->tlc_n_tiles = nt;
        M->tlc_staged = staged; M->tlc_plan_id = id;
    }
    if (!B) return tlc_pack12(A, nullptr, who);
    return USPMV_OK;
}

// as uspmv_dmat_optimize[_ap]: when the tile-local-column plan stages fewer than half of the tiles (wide, irregular rows), try the
// column-window sweep -- built on the device as well -- and let it take over when it covers at least half of the tiles
static int device_sweep_if_irregular(uspmv_dmat_t *A, uspmv_dmat_t *B, int64_t * /*n_tiles*/, int64_t * /*n_staged*/, const char *who) {
    if (A->sw) sw_release(A);
    if (B && B->sw) sw_release(B);
    if (!g_tune.sweep || (A->tlc && A->tlc_staged * 2 >= A->tlc_n_tiles)) return USPMV_OK;
    int64_t swt = 0, sws = 0;
    if (int rc = sweep_plan_install_device(A, B, 0, 0, &swt, &sws, who)) return rc;
    if (A->sw && sws * 2 >= swt) {
        if (A->tlc) tlc_release(A);         // (n_tiles / n_staged keep describing the tile-local-column attempt, as in uspmv_dmat_optimize;
        if (B && B->tlc) tlc_release(B);    //  uspmv_dmat_plan_info tells which plan the handle ended up with)
        return USPMV_OK;
    }
    if (A->sw) { sw_release(A); if (B) sw_release(B); }
    return USPMV_OK;
}

int uspmv_dmat_optimize_device(uspmv_dmat_t *A, int max_lines, int64_t *n_tiles, int64_t *n_staged) {
    if (int rc = check_dmat(A, "uspmv_dmat_optimize_device")) return rc;
    if (int rc = require_device()) return rc;
    if (A->alt) { uspmv_dmat_free(A->alt); A->alt = nullptr; }
    if (A->C < 32 && 32 % A->C == 0 && g_tune.rechunk && A->n_chunks > 0) {
        // narrow chunks (incl. crs = C 1): the internal C = 32 re-chunking of uspmv_dmat_optimize, built on the device --
        // O(n_chunks) layout on the host, the O(n_elements) copy by rechunk32_kernel
        const int64_t C = A->C, nc_old = A->n_chunks, per = 32 / C, nc = (nc_old + per - 1) / per;
        std::vector<int32_t> cl_old((size_t)nc_old), cl((size_t)nc, 0), cp((size_t)nc + 1, 0);
        int32_t last = 0;
        HIP_TRY(hipMemcpy(cl_old.data(), A->chunk_lengths, 4 * (size_t)nc_old, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(&last, A->chunk_ptrs + nc_old, 4, hipMemcpyDeviceToHost));
        int64_t cur = 0;
        bool fits = true;
        for (int64_t k = 0; k < nc && fits; ++k) {
            int32_t L = 0;
            for (int64_t c = k * per; c < std::min((k + 1) * per, nc_old); ++c) L = std::max(L, cl_old[(size_t)c]);
            cl[(size_t)k] = L; cp[(size_t)k] = (int32_t)cur;
            cur += (int64_t)L * 32;
            fits = cur <= INT32_MAX;
        }
        if (fits && (double)cur <= 1.25 * (double)std::max<int64_t>(last, 1) + 4096) {
            cp[(size_t)nc] = (int32_t)cur;
            auto *alt = new uspmv_dmat;
            alt->C = 32; alt->n_chunks = nc; alt->n_elements = cur; alt->dtype = A->dtype; alt->owns = true;
            alt->n_store = (long)(nc_old * C);              // y of the caller has only the original padded rows
            const size_t vsz = A->dtype == USPMV_F64 ? 8 : 4, ne = (size_t)std::max<int64_t>(cur, 1);
            void *d_cp = nullptr, *d_cl = nullptr, *d_ci = nullptr, *d_va = nullptr;
            hipError_t e = hipMalloc(&d_cp, 4 * ((size_t)nc + 1));
            if (e == hipSuccess) e = hipMalloc(&d_cl, 4 * (size_t)nc);
            if (e == hipSuccess) e = hipMalloc(&d_ci, 4 * ne);
            if (e == hipSuccess) e = hipMalloc(&d_va, vsz * ne);
            alt->chunk_ptrs = (const int32_t *)d_cp; alt->chunk_lengths = (const int32_t *)d_cl; alt->col_idxs = (const int32_t *)d_ci; alt->values = d_va;
            if (e == hipSuccess) e = hipMemcpy(d_cp, cp.data(), 4 * ((size_t)nc + 1), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemcpy(d_cl, cl.data(), 4 * (size_t)nc, hipMemcpyHostToDevice);
            if (e == hipSuccess) e = hipMemsetAsync(d_ci, 0, 4 * ne, nullptr);
            if (e == hipSuccess) e = hipMemsetAsync(d_va, 0, vsz * ne, nullptr);
            int rc = e == hipSuccess ? launch_rechunk32(A, (const int *)d_cp, (int *)d_ci, d_va, nullptr) : uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize_device: %s", hipGetErrorString(e));
            if (!rc) rc = device_plan_install(alt, nullptr, max_lines, n_tiles, n_staged, "uspmv_dmat_optimize_device");
            if (!rc) rc = device_sweep_if_irregular(alt, nullptr, n_tiles, n_staged, "uspmv_dmat_optimize_device");
            if (rc) { uspmv_dmat_free(alt); return rc; }
            if (A->tlc) tlc_release(A);
            A->alt = alt;
            return USPMV_OK;
        }
    }
    if (int rc = device_plan_install(A, nullptr, max_lines, n_tiles, n_staged, "uspmv_dmat_optimize_device")) return rc;
    return device_sweep_if_irregular(A, nullptr, n_tiles, n_staged, "uspmv_dmat_optimize_device");
}

int uspmv_dmat_optimize_device_ap(uspmv_dmat_t *dp, uspmv_dmat_t *sp, int max_lines, int64_t *n_tiles, int64_t *n_staged) {
    if (int rc = check_dmat(dp, "uspmv_dmat_optimize_device_ap")) return rc;
    if (int rc = check_dmat(sp, "uspmv_dmat_optimize_device_ap")) return rc;
    if (dp->dtype != USPMV_F64 || sp->dtype != USPMV_F32 || dp->C != sp->C || dp->n_chunks != sp->n_chunks)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_device_ap: handles do not form a dp+sp pair");
    if (int rc = require_device()) return rc;
    if (int rc = device_plan_install(dp, sp, max_lines, n_tiles, n_staged, "uspmv_dmat_optimize_device_ap")) return rc;
    return device_sweep_if_irregular(dp, sp, n_tiles, n_staged, "uspmv_dmat_optimize_device_ap");
}

int uspmv_dmat_plan_download(const uspmv_dmat_t *A, int64_t meta[4], int32_t *tile_line_ptr, int32_t *tile_lines, uint32_t *c16_ptrs,
                             uint16_t *col16) {
    if (int rc = check_dmat(A, "uspmv_dmat_plan_download")) return rc;
    if (!meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_plan_download: NULL meta");
    meta[0] = meta[1] = meta[2] = meta[3] = 0;
    if (!A->tlc) return USPMV_OK;
    if (int rc = require_device()) return rc;
    int32_t last = 0; uint32_t last16 = 0;
    HIP_TRY(hipMemcpy(&last, A->tlc_line_ptr + A->tlc_n_tiles, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&last16, A->tlc_c16_ptrs + A->n_chunks, 4, hipMemcpyDeviceToHost));
    meta[0] = A->tlc_n_tiles; meta[1] = last; meta[2] = last16; meta[3] = A->tlc_max_lines;
    if (tile_line_ptr) HIP_TRY(hipMemcpy(tile_line_ptr, A->tlc_line_ptr, 4 * ((size_t)A->tlc_n_tiles + 1), hipMemcpyDeviceToHost));
    if (tile_lines && last) HIP_TRY(hipMemcpy(tile_lines, A->tlc_lines, 4 * (size_t)last, hipMemcpyDeviceToHost));
    if (c16_ptrs) HIP_TRY(hipMemcpy(c16_ptrs, A->tlc_c16_ptrs, 4 * ((size_t)A->n_chunks + 1), hipMemcpyDeviceToHost));
    if (col16 && last16) HIP_TRY(hipMemcpy(col16, A->tlc_col16, 2 * (size_t)last16, hipMemcpyDeviceToHost));
    return USPMV_OK;
}

static void part_release(uspmv_dmat_t *A, int order) {
    (void)hipFree(A->part_len[order][0]); (void)hipFree(A->part_len[order][1]);
    A->part_len[order][0] = A->part_len[order][1] = nullptr;
}

extern "C++" {
namespace uspmv_dev {

namespace { struct FlagBuf { unsigned char *p = nullptr; ~FlagBuf() { (void)hipFree(p); } }; }

static int part_build(uspmv_dmat *A, int order, int rows_per_flag, const unsigned char *d_flags) {
    part_release(A, order);
    const size_t bytes = 4 * (size_t)std::max<int64_t>(A->n_chunks, 1);
    hipError_t e = hipMalloc((void **)&A->part_len[order][0], bytes);
    if (e == hipSuccess) e = hipMalloc((void **)&A->part_len[order][1], bytes);
    int rc = e == hipSuccess ? launch_part_len_fill(A, rows_per_flag, d_flags, A->part_len[order][0], A->part_len[order][1], nullptr)
                             : uspmv::fail(USPMV_ERR_ALLOC, "two-part SpMMV: %s", hipGetErrorString(e));
    if (!rc && (e = hipStreamSynchronize(nullptr)) != hipSuccess) rc = uspmv::fail(USPMV_ERR_HIP, "two-part SpMMV: %s", hipGetErrorString(e));
    if (rc) part_release(A, order);
    return rc;
}

int dmat_part_set_chunks(uspmv_dmat *A, const unsigned char *h_chunk_flags) {
    FlagBuf f;
    HIP_TRY(hipMalloc((void **)&f.p, (size_t)std::max<int64_t>(A->n_chunks, 1)));
    if (A->n_chunks) HIP_TRY(hipMemcpy(f.p, h_chunk_flags, (size_t)A->n_chunks, hipMemcpyHostToDevice));
    return part_build(A, 0, (int)A->C, f.p);
}

int dmat_part_set_plan(uspmv_dmat *A, long n_local, int64_t *n_boundary_tiles) {
    part_release(A, 1);
    if (n_boundary_tiles) *n_boundary_tiles = 0;
    if (!A->pb || A->pb_n_tiles == 0) return USPMV_OK;
    FlagBuf f;
    HIP_TRY(hipMalloc((void **)&f.p, (size_t)A->pb_n_tiles));
    if (int rc = launch_block_tile_class(A, n_local, f.p, nullptr)) return rc;
    if (n_boundary_tiles) {
        std::vector<unsigned char> h((size_t)A->pb_n_tiles);
        HIP_TRY(hipMemcpy(h.data(), f.p, h.size(), hipMemcpyDeviceToHost));
        for (unsigned char v : h) *n_boundary_tiles += v;
    }
    return part_build(A, 1, 64, f.p);
}

}  // namespace uspmv_dev
}  // extern "C++"

static void bt_release(uspmv_dmat_t *A) {
    part_release(A, 1);                                         // (classified per tile of the plan that goes away)
    (void)hipFree(A->bt_line_ptr); (void)hipFree(A->bt_xrows); (void)hipFree(A->bt_c16_ptrs); (void)hipFree(A->bt_col16);
    (void)hipFree(A->bt_values); (void)hipFree(A->bt_cols); (void)hipFree(A->bt_row_map);
    A->bt_values = nullptr; A->bt_cols = A->bt_row_map = nullptr;
    (void)hipFree(A->pb_ph_ptr); (void)hipFree(A->pb_g0); (void)hipFree(A->pb_list_ptr); (void)hipFree(A->pb_xrows); (void)hipFree(A->pb_c16_ptrs); (void)hipFree(A->pb_col16);
    (void)hipFree(A->pb_values); A->pb_values = nullptr; A->pb_idx8 = false; A->pb_device_built = false;
    dmat_stream_release(A);
    A->pb_ph_ptr = A->pb_g0 = A->pb_list_ptr = A->pb_xrows = nullptr; A->pb_c16_ptrs = nullptr; A->pb_col16 = nullptr; A->pb = false;
    A->bt_line_ptr = A->bt_xrows = nullptr; A->bt_c16_ptrs = nullptr; A->bt_col16 = nullptr;
    A->bt = false;
    (void)hipFree(A->pl_ph_ptr); (void)hipFree(A->pl_g0); (void)hipFree(A->pl_list_ptr); (void)hipFree(A->pl_lines); (void)hipFree(A->pl_col8);
    A->pl_ph_ptr = A->pl_g0 = A->pl_list_ptr = A->pl_lines = nullptr; A->pl_col8 = nullptr; A->pl = false;
    (void)hipFree(A->pu_ph_ptr); (void)hipFree(A->pu_g0); (void)hipFree(A->pu_list_ptr); (void)hipFree(A->pu_xrows); (void)hipFree(A->pu_col8); (void)hipFree(A->pu_perm);
    A->pu_ph_ptr = A->pu_g0 = A->pu_list_ptr = A->pu_xrows = A->pu_perm = nullptr; A->pu_col8 = nullptr; A->pu = false;
}

extern "C++" { namespace uspmv_dev { void dmat_block_plan_release(uspmv_dmat *A) { if (A->bt || A->pb) bt_release(A); } } }

int uspmv_dmat_optimize_block(uspmv_dmat_t *A, const uspmv_scs_t *s, int block_vec_size, int64_t *n_tiles, int64_t *n_staged) {
    if (!A || !s) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: layout-only struct; the plan builder needs the host column indices");
    if (A->C != s->C || A->n_chunks != s->n_chunks || A->dtype != s->dtype)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: handle and host struct do not describe the same matrix");
    if (block_vec_size < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: block_vec_size must be >= 1");
    if (int rc = require_device()) return rc;
    if (A->bt || A->pb) bt_release(A);
    if (n_tiles) *n_tiles = 0;
    if (n_staged) *n_staged = 0;
    // a struct rebuilt from the handle's device arrays (uspmv_dmat_optimize_block_device) carries the indices only: the plan's private
    // copies of the VALUES are then gathered on the device from the handle's own array (launch_block_values_gather)
    const bool host_values = (int64_t)(s->dtype == USPMV_F64 ? s->values_f64.size() : s->values_f32.size()) == s->n_elements;
    const size_t row_bytes = (size_t)block_vec_size * (s->dtype == USPMV_F64 ? 8 : 4);
    // only the 16-byte-piece kernels (b*sizeof(VT) in {16,32,64,128}) read the plan, compiled for C = 32 and 64
    if (row_bytes % 16 != 0 || (row_bytes & (row_bytes - 1)) != 0 || row_bytes > 128 || (s->C != 32 && s->C != 64)) return USPMV_OK;
    const size_t cap = g_tune.spmmv_lds_kb > 0 ? std::min<size_t>((size_t)g_tune.spmmv_lds_kb * 1024, BT_LDS_CAP) : BT_LDS_CAP;
    const int max_rows = (int)(cap / row_bytes);
    // rows of >= 64 bytes on C = 32: 32-row tiles, two lanes per row (half the LDS per tile, twice the tiles per CU)
    const int tile_rows = (s->C == 32 && g_tune.spmmv_tile_rows != 64 && (row_bytes >= 128 || (row_bytes >= 32 && g_tune.spmmv_tile_rows == 32))) ? 32 : 64;
    uspmv_tlc_plan p;
    uspmv_scs r;                       // private copy with the sigma sort's ties undone (only kept when rows moved)
    std::vector<int32_t> row_map;
    const bool moved = g_tune.spmmv_reorder && (g_tune.spmmv_reorder == 3 ? uspmv_scs_reorder_bricks(s, g_tune.spmmv_brick_stride, g_tune.spmmv_brick_lines, &r, &row_map)
                                                                          : uspmv_scs_reorder_rows(s, g_tune.spmmv_reorder == 2 ? 2 : (g_tune.spmmv_reorder == 4 && !g_tune.spmmv_xline) ? 4 : 1, &r, &row_map)) == 1;
    // (with the line plan requested -- "spmmv_xline": X staged by 128-byte lines of the column-major vector -- the rows stay in original order,
    //  ties undone: a patch of several mesh lines touches more LINES of X than a run of consecutive rows)
    // 64-byte rows: the phased plan over the same (re-ordered) entries -- what uspmv_spmmv runs by default.  When the phased kernel
    // can take it (at most 512 rows per phase), the one-list-per-tile plan of the older kernels and its column-major copy of the entries
    // (8 + 6 bytes per non-zero of HBM, a second or two of planning) are only built on request ("spmmv_list_plan" 1).
    uspmv_phased_plan pp;
    if (row_bytes == 64 && tile_rows == 64 && g_tune.spmmv_phased)
        if (int rc = uspmv_build_phased_plan(moved ? &r : s, g_tune.spmmv_phase_rows, 8, &pp, 0, g_tune.spmmv_phase_dp)) return rc;
    const bool phased_ok = pp.valid && pp.ngp <= 8 && (pp.max_rows_used * 4 + 255) / 256 <= 8;
    // ... and once more with LINE lists for column-major block vectors (no re-layout pass over X): kept when no phase needs more
    // than 256 rows' worth of lines and the lines staged stay below twice the rows the row plan stages
    uspmv_phased_plan pl;
    if (phased_ok && g_tune.spmmv_xline && g_tune.spmmv_phase_rows == 256) {
        const int shift = s->dtype == USPMV_F64 ? 4 : 5;
        if (int rc = uspmv_build_phased_plan(moved ? &r : s, 256, 8, &pl, shift)) return rc;
        if (getenv("USPMV_VERBOSE")) {
            int64_t over = 0;
            for (int64_t ph = 0; ph < pl.n_phases; ++ph) over += (pl.ph_list_ptr[(size_t)ph + 1] - pl.ph_list_ptr[(size_t)ph]) > (256 >> shift);
            fprintf(stderr, "[uspmv] line plan candidate: valid=%d phases=%lld lines_total=%zu (= %zu rows; row plan stages %zu) max_rows=%d phases over the cap: %lld\n",
                    (int)pl.valid, (long long)pl.n_phases, pl.xrows.size(), pl.xrows.size() << shift, pp.xrows.size(), pl.max_rows_used, (long long)over);
        }
        if (pl.valid && (pl.max_rows_used > 256 || pl.ngp > 8 || ((int64_t)pl.xrows.size() << shift) > 2 * (int64_t)pp.xrows.size())) pl.valid = false;
    }
    // ... and once more over ORIGINAL X-row numbering (column index c -> new_to_old[c]) for column-major callers: their re-layout pass
    // undoes the sigma permutation on the way, so the rows a tile needs are runs of the workspace again
    uspmv_phased_plan pu;
    const bool have_perm = host_values && s->sigma > 1 && (int64_t)s->new_to_old_idx.size() >= s->n_rows && (int64_t)s->old_to_new_idx.size() >= s->n_rows;
    if (phased_ok && g_tune.spmmv_unscramble && have_perm && pp.max_rows_used <= 256 && g_tune.spmmv_idx8) {
        uspmv_scs u;                                    // indices only: a struct with the renumbered columns
        const uspmv_scs *src = moved ? &r : s;
        u.C = src->C; u.sigma = src->sigma; u.n_rows = src->n_rows; u.n_cols = src->n_cols; u.n_rows_padded = src->n_rows_padded; u.n_chunks = src->n_chunks;
        u.n_elements = src->n_elements; u.nnz = src->nnz; u.dtype = src->dtype;
        u.chunk_ptrs = src->chunk_ptrs; u.chunk_lengths = src->chunk_lengths;
        u.col_idxs.resize(src->col_idxs.size());
        const int32_t *n2o = s->new_to_old_idx.data();
        const int64_t nr = s->n_rows;
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < (int64_t)src->col_idxs.size(); ++k) { const int32_t c = src->col_idxs[(size_t)k]; u.col_idxs[(size_t)k] = c < nr ? n2o[c] : c; }
        if (int rc = uspmv_build_phased_plan(&u, 256, 8, &pu, 0, g_tune.spmmv_phase_dp)) return rc;
        if (pu.valid && (pu.max_rows_used > 256 || pu.ngp > 8)) pu.valid = false;
    }
    const bool list_plan = !phased_ok || g_tune.spmmv_list_plan;
    if (list_plan) {
        if (int rc = uspmv_build_tlc_plan(moved ? &r : s, nullptr, max_rows, tile_rows, &p, /*line_shift=*/0)) return rc;
        if (n_tiles) *n_tiles = p.n_tiles;
        if (n_staged) *n_staged = p.valid ? p.n_staged_tiles : 0;
        if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] block plan: b=%d tile_rows=%d tiles=%lld staged=%lld max_rows=%d (cap %d) rows_total=%zu\n",
                                             block_vec_size, p.tile_rows, (long long)p.n_tiles, (long long)p.n_staged_tiles, p.max_lines_used, max_rows, p.tile_lines.size());
        if (!p.valid && !pp.valid) return USPMV_OK;
    } else {
        if (n_tiles) *n_tiles = pp.n_tiles;
        if (n_staged) *n_staged = pp.n_tiles;
    }
    auto up = [&](const void *h, size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = hipSuccess;
    if (moved) e = up(row_map.data(), row_map.size() * 4, (void **)&A->bt_row_map);
    if (list_plan && p.valid) {
        if (e == hipSuccess) e = up(p.tile_line_ptr.data(), p.tile_line_ptr.size() * 4, (void **)&A->bt_line_ptr);
        if (e == hipSuccess) e = up(p.tile_lines.data(), p.tile_lines.size() * 4, (void **)&A->bt_xrows);
        if (e == hipSuccess) e = up(p.c16_ptrs.data(), p.c16_ptrs.size() * 4, (void **)&A->bt_c16_ptrs);
        if (e == hipSuccess) e = up(p.col16.data(), p.col16.size() * 2, (void **)&A->bt_col16);
        if (e == hipSuccess && moved) {
            if (host_values) e = up(r.values_ptr(), (size_t)r.n_elements * (r.dtype == USPMV_F64 ? 8 : 4), &A->bt_values);
            else {
                e = hipMalloc(&A->bt_values, std::max<size_t>((size_t)r.n_elements, 1) * (r.dtype == USPMV_F64 ? 8 : 4));
                if (e == hipSuccess && launch_block_values_gather(A, A->bt_row_map, nullptr, A->bt_values, false, nullptr) != USPMV_OK) e = hipErrorUnknown;
            }
            if (e == hipSuccess && (p.n_staged_tiles < p.n_tiles || g_tune.spmmv_variant == 5)) e = up(r.col_idxs.data(), (size_t)r.n_elements * 4, (void **)&A->bt_cols);
        }
    }
    if (e == hipSuccess && pp.valid) {
        {
            e = up(pp.ph_ptr.data(), pp.ph_ptr.size() * 4, (void **)&A->pb_ph_ptr);
            if (e == hipSuccess) e = up(pp.ph_g0.data(), pp.ph_g0.size() * 4, (void **)&A->pb_g0);
            if (e == hipSuccess) e = up(pp.ph_list_ptr.data(), pp.ph_list_ptr.size() * 4, (void **)&A->pb_list_ptr);
            if (e == hipSuccess) e = up(pp.xrows.data(), pp.xrows.size() * 4, (void **)&A->pb_xrows);
            if (e == hipSuccess) e = up(pp.c16_ptrs.data(), pp.c16_ptrs.size() * 4, (void **)&A->pb_c16_ptrs);
            A->pb_idx8 = g_tune.spmmv_idx8 && pp.max_rows_used <= 256;
            if (e == hipSuccess && A->pb_idx8) {             // same layout, one byte per entry
                std::vector<uint8_t> c8(pp.col16.size());
                for (size_t k = 0; k < c8.size(); ++k) c8[k] = (uint8_t)pp.col16[k];
                e = up(c8.data(), c8.size(), (void **)&A->pb_col16);
            } else if (e == hipSuccess) e = up(pp.col16.data(), pp.col16.size() * 2, (void **)&A->pb_col16);
            if (e == hipSuccess && !host_values) {
                const size_t vs = s->dtype == USPMV_F64 ? 8 : 4;
                e = hipMalloc(&A->pb_values, std::max<size_t>(pp.col16.size(), 1) * vs);
                if (e == hipSuccess) e = hipMemset(A->pb_values, 0, std::max<size_t>(pp.col16.size(), 1) * vs);     // padded slots of the last group of a chunk
                if (e == hipSuccess && launch_block_values_gather(A, moved ? A->bt_row_map : nullptr, A->pb_c16_ptrs, A->pb_values, true, nullptr) != USPMV_OK) e = hipErrorUnknown;
                if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
            } else if (e == hipSuccess) {
                // the entries once more, group-major like the indices (what scs_spmmv_quadph streams)
                const uspmv_scs *src = moved ? &r : s;
                const size_t vs = src->dtype == USPMV_F64 ? 8 : 4;
                std::vector<unsigned char> gv(pp.col16.size() * vs, 0);
                const int64_t C_ = src->C;
#pragma omp parallel for schedule(static)
                for (int64_t c = 0; c < src->n_chunks; ++c) {
                    const int64_t cs = src->chunk_ptrs[(size_t)c], L = src->chunk_lengths[(size_t)c];
                    const size_t base = pp.c16_ptrs[(size_t)c];
                    for (int64_t j = 0; j < L; ++j)
                        for (int64_t i = 0; i < C_; ++i) {
                            const size_t dst = base + (size_t)((j / 4) * 4 * C_ + i * 4 + (j % 4)), from = (size_t)(cs + j * C_ + i);
                            if (vs == 8) ((double *)gv.data())[dst] = src->values_f64[from]; else ((float *)gv.data())[dst] = src->values_f32[from];
                        }
                }
                e = up(gv.data(), gv.size(), &A->pb_values);
            }
            if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] phased block plan: tiles=%lld phases=%lld rows_total=%zu max_rows=%d (cap %d)\n",
                                                 (long long)pp.n_tiles, (long long)pp.n_phases, pp.xrows.size(), pp.max_rows_used, pp.cap_rows);
            if (e == hipSuccess && pu.valid) {
                e = up(pu.ph_ptr.data(), pu.ph_ptr.size() * 4, (void **)&A->pu_ph_ptr);
                if (e == hipSuccess) e = up(pu.ph_g0.data(), pu.ph_g0.size() * 4, (void **)&A->pu_g0);
                if (e == hipSuccess) e = up(pu.ph_list_ptr.data(), pu.ph_list_ptr.size() * 4, (void **)&A->pu_list_ptr);
                if (e == hipSuccess) e = up(pu.xrows.data(), pu.xrows.size() * 4, (void **)&A->pu_xrows);
                if (e == hipSuccess) e = up(s->old_to_new_idx.data(), (size_t)s->n_rows * 4, (void **)&A->pu_perm);
                if (e == hipSuccess) {
                    std::vector<uint8_t> c8(pu.col16.size());
                    for (size_t k = 0; k < c8.size(); ++k) c8[k] = (uint8_t)pu.col16[k];
                    e = up(c8.data(), c8.size(), (void **)&A->pu_col8);
                }
                if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] unscrambled plan (column-major X behind the permuting re-layout): phases=%lld rows_total=%zu (scrambled: %zu) max_rows=%d\n",
                                                     (long long)pu.n_phases, pu.xrows.size(), pp.xrows.size(), pu.max_rows_used);
            }
            if (e == hipSuccess && pl.valid) {
                e = up(pl.ph_ptr.data(), pl.ph_ptr.size() * 4, (void **)&A->pl_ph_ptr);
                if (e == hipSuccess) e = up(pl.ph_g0.data(), pl.ph_g0.size() * 4, (void **)&A->pl_g0);
                if (e == hipSuccess) e = up(pl.ph_list_ptr.data(), pl.ph_list_ptr.size() * 4, (void **)&A->pl_list_ptr);
                if (e == hipSuccess) e = up(pl.xrows.data(), pl.xrows.size() * 4, (void **)&A->pl_lines);
                if (e == hipSuccess) {
                    std::vector<uint8_t> c8(pl.col16.size());
                    for (size_t k = 0; k < c8.size(); ++k) c8[k] = (uint8_t)pl.col16[k];
                    e = up(c8.data(), c8.size(), (void **)&A->pl_col8);
                }
                if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] line plan (column-major X): phases=%lld lines_total=%zu (= %zu rows) max_rows=%d\n",
                                                     (long long)pl.n_phases, pl.xrows.size(), pl.xrows.size() << pl.line_shift, pl.max_rows_used);
            }
        }
    }
    if (e != hipSuccess) {
        bt_release(A);
        return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize_block: device copy failed: %s", hipGetErrorString(e));
    }
    if (pp.valid && pu.valid) { A->pu = true; A->pu_max_rows = pu.max_rows_used; A->pu_n_phases = pu.n_phases; A->pu_n_perm = s->n_rows; }
    if (pp.valid && pl.valid) { A->pl = true; A->pl_shift = pl.line_shift; A->pl_max_rows = pl.max_rows_used; A->pl_n_phases = pl.n_phases; A->pl_rows_staged = (int64_t)pl.xrows.size() << pl.line_shift; }
    if (pp.valid) { A->pb = true; A->pb_cap_rows = pp.cap_rows; A->pb_ngp = pp.ngp; A->pb_max_rows = pp.max_rows_used; A->pb_n_tiles = pp.n_tiles; A->pb_n_phases = pp.n_phases; A->pb_rows_staged = (int64_t)pp.xrows.size(); }
    if (list_plan && p.valid) { A->bt = true; A->bt_tile_rows = p.tile_rows; A->bt_max_rows = p.max_lines_used; A->bt_n_tiles = p.n_tiles; A->bt_staged = p.n_staged_tiles; }
    if (A->pb && g_tune.spmmv_stream > 0) return dmat_stream_schedule(A, g_tune.spmmv_stream);
    return USPMV_OK;
}

// The phased block plan built entirely on the device (csrc/block_plan_kernels.hip): row order, phases, X-row lists, one-byte indices and
// the group-major value copy; the host sees the chunk lengths (offsets of the index array) and two integers per tile (exclusive scans).
// Returns 1 when the shape / tuning is not the default one this builder covers (the caller then plans the index part on the host).
static int block_plan_install_device(uspmv_dmat_t *A, int block_vec_size, int64_t *n_tiles, int64_t *n_staged) {
    const size_t vsz = A->dtype == USPMV_F64 ? 8 : 4;
    const size_t row_bytes = (size_t)block_vec_size * vsz;
    if (row_bytes != 64 || (A->C != 32 && A->C != 64) || !g_tune.spmmv_phased || g_tune.spmmv_phase_rows != 256 || g_tune.spmmv_list_plan ||
        (g_tune.spmmv_reorder != 1 && g_tune.spmmv_reorder != 4) || !g_tune.spmmv_idx8 || g_tune.spmmv_tile_rows == 32 || g_tune.spmmv_xline || !g_tune.block_plan_device) return 1;
    const int64_t C = A->C, nc = A->n_chunks, n_pad = nc * C, nt = (n_pad + 63) / 64;
    if (A->bt || A->pb) bt_release(A);
    std::vector<int32_t> cl((size_t)nc);
    std::vector<uint32_t> c16p;
    int64_t tot16 = 0;
    HIP_TRY(hipMemcpy(cl.data(), A->chunk_lengths, 4 * (size_t)nc, hipMemcpyDeviceToHost));
    if (!c16_offsets(cl, C, &c16p, &tot16)) return USPMV_OK;
    int *d_changed = nullptr, *d_tph = nullptr, *d_tl = nullptr, *d_max = nullptr;
    hipError_t e = hipMalloc((void **)&d_changed, 8);
    d_max = d_changed ? d_changed + 1 : nullptr;
    if (e == hipSuccess) e = hipMemset(d_changed, 0, 8);
    if (e == hipSuccess) e = hipMalloc((void **)&A->bt_row_map, 4 * (size_t)std::max<int64_t>(n_pad, 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->pb_c16_ptrs, 4 * ((size_t)nc + 1));
    if (e == hipSuccess) e = hipMemcpy(A->pb_c16_ptrs, c16p.data(), 4 * ((size_t)nc + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&d_tph, 4 * (size_t)nt);
    if (e == hipSuccess) e = hipMalloc((void **)&d_tl, 4 * (size_t)nt);
    auto fail_out = [&](const char *what) {
        (void)hipFree(d_changed); (void)hipFree(d_tph); (void)hipFree(d_tl);
        bt_release(A);
        return uspmv::fail(USPMV_ERR_HIP, "uspmv_dmat_optimize_block_device: %s: %s", what, hipGetErrorString(e));
    };
    if (e != hipSuccess) return fail_out("allocation");
    // ---- row order (ties of the sigma sort undone by first column), then the phases: count, scan, write
    int rc = launch_block_reorder(A, A->bt_row_map, d_changed, nullptr);
    int changed = 0;
    if (!rc) { e = hipMemcpy(&changed, d_changed, 4, hipMemcpyDeviceToHost); if (e != hipSuccess) return fail_out("row order"); }
    const int *rmap = changed ? A->bt_row_map : nullptr;
    if (!changed) { (void)hipFree(A->bt_row_map); A->bt_row_map = nullptr; }
    if (!rc) rc = launch_block_phase_plan(A, false, 256, 8, rmap, A->pb_c16_ptrs, d_tph, d_tl, nullptr, nullptr, nullptr, nullptr, d_max, nullptr);
    std::vector<int32_t> tph((size_t)nt + 1, 0), tl((size_t)nt + 1, 0);
    if (!rc) {
        e = hipMemcpy(tph.data(), d_tph, 4 * (size_t)nt, hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(tl.data(), d_tl, 4 * (size_t)nt, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail_out("phase counts");
    }
    if (rc) { (void)hipFree(d_changed); (void)hipFree(d_tph); (void)hipFree(d_tl); bt_release(A); return rc; }
    int64_t n_ph = 0, n_list = 0;
    for (int64_t t = 0; t < nt; ++t) {   // exclusive scans (ph_ptr of the plan; list bases)
        const int32_t a = tph[(size_t)t], b = tl[(size_t)t];
        tph[(size_t)t] = (int32_t)n_ph; tl[(size_t)t] = (int32_t)n_list;
        n_ph += a; n_list += b;
        if (n_ph > INT32_MAX || n_list > INT32_MAX) { (void)hipFree(d_changed); (void)hipFree(d_tph); (void)hipFree(d_tl); bt_release(A); return USPMV_OK; }
    }
    tph[(size_t)nt] = (int32_t)n_ph; tl[(size_t)nt] = (int32_t)n_list;
    if (n_tiles) *n_tiles = nt;
    if (n_staged) *n_staged = n_ph > 0 ? nt : 0;
    if (n_ph == 0) { (void)hipFree(d_changed); (void)hipFree(d_tph); (void)hipFree(d_tl); bt_release(A); return USPMV_OK; }
    e = hipMalloc((void **)&A->pb_ph_ptr, 4 * ((size_t)nt + 1));
    if (e == hipSuccess) e = hipMemcpy(A->pb_ph_ptr, tph.data(), 4 * ((size_t)nt + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_tl, tl.data(), 4 * (size_t)nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&A->pb_g0, 4 * (size_t)n_ph);
    if (e == hipSuccess) e = hipMalloc((void **)&A->pb_list_ptr, 4 * ((size_t)n_ph + 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->pb_xrows, 4 * (size_t)std::max<int64_t>(n_list, 1));
    if (e == hipSuccess) e = hipMalloc((void **)&A->pb_col16, (size_t)std::max<int64_t>(tot16, 1));
    if (e == hipSuccess) e = hipMemset(A->pb_col16, 0, (size_t)std::max<int64_t>(tot16, 1));
    if (e == hipSuccess) { const int32_t last = (int32_t)n_list; e = hipMemcpy(A->pb_list_ptr + n_ph, &last, 4, hipMemcpyHostToDevice); }
    if (e == hipSuccess) e = hipMalloc(&A->pb_values, (size_t)std::max<int64_t>(tot16, 1) * vsz);
    if (e == hipSuccess) e = hipMemset(A->pb_values, 0, (size_t)std::max<int64_t>(tot16, 1) * vsz);
    if (e != hipSuccess) return fail_out("plan arrays");
    rc = launch_block_phase_plan(A, true, 256, 8, rmap, A->pb_c16_ptrs, A->pb_ph_ptr, d_tl, A->pb_g0, A->pb_list_ptr, A->pb_xrows, (unsigned char *)A->pb_col16, d_max, nullptr);
    if (!rc) rc = launch_block_values_gather(A, rmap, A->pb_c16_ptrs, A->pb_values, true, nullptr);
    int max_rows = 0;
    if (!rc) { e = hipMemcpy(&max_rows, d_max, 4, hipMemcpyDeviceToHost); if (e != hipSuccess) return fail_out("plan kernels"); }
    (void)hipFree(d_changed); (void)hipFree(d_tph); (void)hipFree(d_tl);
    if (rc) { bt_release(A); return rc; }
    A->pb = true; A->pb_idx8 = true; A->pb_device_built = true; A->pb_cap_rows = 256; A->pb_ngp = 8; A->pb_max_rows = max_rows; A->pb_n_tiles = nt; A->pb_n_phases = n_ph; A->pb_rows_staged = n_list;
    if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] phased block plan (device builder): tiles=%lld phases=%lld rows_total=%lld max_rows=%d rows %s\n",
                                         (long long)nt, (long long)n_ph, (long long)n_list, max_rows, changed ? "re-ordered" : "in the caller's order");
    if (g_tune.spmmv_stream > 0) return dmat_stream_schedule(A, g_tune.spmmv_stream);
    return USPMV_OK;
}

// The block plan for a handle whose arrays exist only in HBM (uspmv_dmat_wrap around a harness' own device arrays -- what the
// function-pointer launchers hold): the INDEX arrays (4 of the 12 bytes per non-zero) are copied to the host once, the index part of
// the plan (row order, phases, X-row lists, local indices) is built there like in uspmv_dmat_optimize_block and uploaded; the
// plan's copies of the VALUES (8 bytes per non-zero, group-major under the plan's row map) are gathered on the device from the
// handle's own array and never cross the bus.  Without the caller's permutation the tie re-ordering orders the rows of equal-length
// chunks by their first column (uspmv_scs_reorder_ties), which for locally numbered matrices restores the original row order.
int uspmv_dmat_optimize_block_device(uspmv_dmat_t *A, int block_vec_size, int64_t *n_tiles, int64_t *n_staged) {
    if (int rc = check_dmat(A, "uspmv_dmat_optimize_block_device")) return rc;
    if (block_vec_size < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block_device: block_vec_size must be >= 1");
    if (int rc = require_device()) return rc;
    uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;      // narrow chunks: the internal C = 32 re-chunking is what uspmv_spmmv runs on
    if (n_tiles) *n_tiles = 0;
    if (n_staged) *n_staged = 0;
    if ((M->C != 32 && M->C != 64) || M->n_chunks < 1) return USPMV_OK;
    {   // the default shape: everything on the device
        const int rc = block_plan_install_device(M, block_vec_size, n_tiles, n_staged);
        if (rc != 1) return rc;
    }
    uspmv_scs s;
    s.C = M->C; s.sigma = 0; s.n_chunks = M->n_chunks; s.n_rows = s.n_rows_padded = M->n_chunks * M->C; s.dtype = M->dtype;
    s.chunk_ptrs.resize((size_t)M->n_chunks + 1); s.chunk_lengths.resize((size_t)M->n_chunks);
    HIP_TRY(hipMemcpy(s.chunk_ptrs.data(), M->chunk_ptrs, 4 * ((size_t)M->n_chunks + 1), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(s.chunk_lengths.data(), M->chunk_lengths, 4 * (size_t)M->n_chunks, hipMemcpyDeviceToHost));
    s.n_elements = s.chunk_ptrs[(size_t)M->n_chunks]; s.nnz = s.n_elements;
    s.col_idxs.resize((size_t)s.n_elements);
    HIP_TRY(hipMemcpy(s.col_idxs.data(), M->col_idxs, 4 * (size_t)s.n_elements, hipMemcpyDeviceToHost));
    int32_t mc = 0;
    for (int32_t c : s.col_idxs) mc = std::max(mc, c);
    s.n_cols = (int64_t)mc + 1;
    // (the values stay where they are: uspmv_dmat_optimize_block gathers the plan's private copies on the device)
    return uspmv_dmat_optimize_block(M, &s, block_vec_size, n_tiles, n_staged);
}

int uspmv_dmat_optimize_ap(uspmv_dmat_t *dp, uspmv_dmat_t *sp, const uspmv_scs_t *s_dp, const uspmv_scs_t *s_sp,
                           int max_lines, int64_t *n_tiles, int64_t *n_staged) {
    if (!dp || !sp || !s_dp || !s_sp) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_ap: NULL argument");
    if (!uspmv::scs_has_entries(s_dp) || !uspmv::scs_has_entries(s_sp))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_ap: layout-only struct; the plan builder needs the host column indices");
    if (dp->C != s_dp->C || dp->n_chunks != s_dp->n_chunks || dp->dtype != USPMV_F64 || s_dp->dtype != USPMV_F64 ||
        sp->C != s_sp->C || sp->n_chunks != s_sp->n_chunks || sp->dtype != USPMV_F32 || s_sp->dtype != USPMV_F32 ||
        dp->C != sp->C || dp->n_chunks != sp->n_chunks)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_ap: handles / host structs do not form a dp+sp pair");
    if (int rc = require_device()) return rc;
    if (dp->tlc) tlc_release(dp);
    if (sp->tlc) tlc_release(sp);
    if (max_lines <= 0) max_lines = 512;
    if (max_lines > 1280) max_lines = 1280;
    uspmv_tlc_plan p;
    const int R_meas = measured_tile_rows(dp, sp, max_lines, "uspmv_dmat_optimize_ap");
    if (int rc = uspmv_build_tlc_plan(s_dp, s_sp, max_lines, R_meas ? R_meas : plan_tile_rows(true), &p)) return rc;
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_staged) *n_staged = p.valid ? p.n_staged_tiles : 0;
    if (dp->sw) sw_release(dp);
    if (sp->sw) sw_release(sp);
    if ((!p.valid || p.n_staged_tiles * 2 < p.n_tiles) && g_tune.sweep) {   // as in uspmv_dmat_optimize
        int64_t swt = 0, sws = 0;
        if (int rc = sweep_plan_install(dp, sp, s_dp, s_sp, 0, 0, &swt, &sws, "uspmv_dmat_optimize_ap")) return rc;
        if (dp->sw && sws * 2 >= swt) return USPMV_OK;
        if (dp->sw) { sw_release(dp); sw_release(sp); }
    }
    if (!p.valid) return USPMV_OK;
    auto up = [&](const void *h, size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(p.tile_line_ptr.data(), p.tile_line_ptr.size() * 4, (void **)&dp->tlc_line_ptr);
    if (e == hipSuccess) e = up(p.tile_lines.data(), p.tile_lines.size() * 4, (void **)&dp->tlc_lines);
    if (e == hipSuccess) e = up(p.c16_ptrs.data(), p.c16_ptrs.size() * 4, (void **)&dp->tlc_c16_ptrs);
    if (e == hipSuccess) e = up(p.col16.data(), p.col16.size() * 2, (void **)&dp->tlc_col16);
    if (e == hipSuccess) e = up(p.c16_ptrs_b.data(), p.c16_ptrs_b.size() * 4, (void **)&sp->tlc_c16_ptrs);
    if (e == hipSuccess) e = up(p.col16_b.data(), p.col16_b.size() * 2, (void **)&sp->tlc_col16);
    if (e != hipSuccess) {
        tlc_release(dp); tlc_release(sp);
        return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize_ap: device copy failed: %s", hipGetErrorString(e));
    }
    static uint64_t next_plan_id = 1;
    const uint64_t id = next_plan_id++;
    for (uspmv_dmat_t *A : {dp, sp}) {
        A->tlc = true; A->tlc_tile_rows = p.tile_rows; A->tlc_max_lines = p.max_lines_used; A->tlc_x_len = p.x_len_min;
        A->tlc_n_tiles = p.n_tiles; A->tlc_staged = p.n_staged_tiles; A->tlc_plan_id = id;
    }
    return USPMV_OK;
}


static void sw_release(uspmv_dmat_t *A) {
    (void)hipFree(A->sw_tile_ids); (void)hipFree(A->sw_smin); (void)hipFree(A->sw_S); (void)hipFree(A->sw_pad); (void)hipFree(A->sw_pad_b);
    (void)hipFree(A->sw_rest); (void)hipFree(A->sw_cnt_off); (void)hipFree(A->sw_wave_off); (void)hipFree(A->sw_wave_off_b);
    (void)hipFree(A->sw_cnt); (void)hipFree(A->sw_cnt_b); (void)hipFree(A->sw_vals); (void)hipFree(A->sw_vals_b); (void)hipFree(A->sw_idx); (void)hipFree(A->sw_idx_b);
    A->sw_tile_ids = A->sw_smin = A->sw_S = A->sw_pad = A->sw_pad_b = A->sw_rest = nullptr;
    A->sw_cnt_off = nullptr; A->sw_wave_off = A->sw_wave_off_b = nullptr; A->sw_cnt = A->sw_cnt_b = nullptr;
    A->sw_vals = nullptr; A->sw_vals_b = nullptr; A->sw_idx = A->sw_idx_b = nullptr;
    A->sw = false; A->sw_plan_id = 0; A->sw_n_tiles = A->sw_all_tiles = A->sw_n_rest = 0;
}

// builds and uploads a sweep plan for A (and, when B/sB are given, for the dp+sp pair A/B); returns the number of sweep tiles
static int sweep_plan_install(uspmv_dmat_t *A, uspmv_dmat_t *B, const uspmv_scs_t *s, const uspmv_scs_t *sB, int wlog, int tile_rows,
                              int64_t *n_tiles, int64_t *n_sweep, const char *who) {
    if (A->sw) sw_release(A);
    if (B && B->sw) sw_release(B);
    if (n_tiles) *n_tiles = 0;
    if (n_sweep) *n_sweep = 0;
    const size_t vsz = s->dtype == USPMV_F64 ? 8 : 4;
    if (wlog <= 0) wlog = g_tune.sweep_wlog;
    const int nbuf = g_tune.sweep_nbuf == 2 ? 2 : 1;
    // window: as much of the LDS as one buffer per workgroup allows (128 KiB; 64 KiB each when double-buffered) -- on config 4b every
    // doubling from 8 KiB up paid (1.51 / 0.97 / 0.74 / 0.61 ms for the ap kernel at 2^10 .. 2^13 elements, 0.55 at 2^14 with 4 096-row
    // tiles; profiles/r02/config4b_sweep_variants.txt): fewer, longer rounds per wave and fewer barriers
    if (wlog <= 0) wlog = (vsz == 8 ? 13 : 14) + (nbuf == 1 ? 1 : 0);
    if (((size_t)1 << wlog) * vsz * (size_t)nbuf > 160 * 1024)
        return uspmv::fail(USPMV_ERR_INVALID, "%s: %d window buffer(s) of 2^%d elements do not fit the 160 KB of LDS", who, nbuf, wlog);
    if (tile_rows <= 0) tile_rows = g_tune.sweep_tile_rows;
    // rows per tile: the windows are staged once per tile, so more rows = fewer staged bytes per non-zero (a lane owns up to four rows);
    // but at least ~1.5 tiles per CU
    if (tile_rows <= 0) {
        const int64_t n_pad = s->n_chunks * s->C;
        tile_rows = 4096;
        while (tile_rows > 1024 && n_pad / tile_rows < 384) tile_rows /= 2;
    }
    uspmv_sweep_plan p;
    const double max_stage = g_tune.sweep_max_stage > 0 ? (double)g_tune.sweep_max_stage : 24.0;
    if (int rc = uspmv_build_sweep_plan(s, sB, wlog, tile_rows, max_stage, &p)) return rc;
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_sweep) *n_sweep = p.valid ? p.n_sweep_tiles : 0;
    if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] sweep plan: tile_rows=%d wlog=%d tiles=%lld sweep=%lld rest_chunks=%zu elements=%zu cnt_bytes=%zu\n",
                                         p.tile_rows, p.wlog, (long long)p.n_tiles, (long long)p.n_sweep_tiles, p.rest_chunks.size(), p.idx.size(), p.cnt.size());
    if (!p.valid) return USPMV_OK;
    hipError_t e = hipSuccess;
    auto up = [&](const void *h, size_t bytes, void **d) {
        if (e != hipSuccess) return;
        e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    up(p.tile_ids.data(), p.tile_ids.size() * 4, (void **)&A->sw_tile_ids);
    up(p.t_smin.data(), p.t_smin.size() * 4, (void **)&A->sw_smin);
    up(p.t_S.data(), p.t_S.size() * 4, (void **)&A->sw_S);
    up(p.t_cnt_off.data(), p.t_cnt_off.size() * 8, (void **)&A->sw_cnt_off);
    up(p.wave_off.data(), p.wave_off.size() * 4, (void **)&A->sw_wave_off);
    up(p.cnt.data(), p.cnt.size(), (void **)&A->sw_cnt);
    up(vsz == 8 ? (const void *)p.vals_f64.data() : (const void *)p.vals_f32.data(), p.idx.size() * vsz, &A->sw_vals);
    up(p.idx.data(), p.idx.size() * 2, (void **)&A->sw_idx);
    up(p.pad_col.data(), p.pad_col.size() * 4, (void **)&A->sw_pad);
    up(p.rest_chunks.data(), p.rest_chunks.size() * 4, (void **)&A->sw_rest);
    if (B) {
        up(p.wave_off_b.data(), p.wave_off_b.size() * 4, (void **)&A->sw_wave_off_b);
        up(p.cnt_b.data(), p.cnt_b.size(), (void **)&A->sw_cnt_b);
        up(p.vals_b_f32.data(), p.idx_b.size() * 4, (void **)&A->sw_vals_b);
        up(p.idx_b.data(), p.idx_b.size() * 2, (void **)&A->sw_idx_b);
        up(p.pad_col_b.data(), p.pad_col_b.size() * 4, (void **)&A->sw_pad_b);
    }
    if (e != hipSuccess) {
        sw_release(A);
        return uspmv::fail(USPMV_ERR_ALLOC, "%s: device copy failed: %s", who, hipGetErrorString(e));
    }
    static uint64_t next_sweep_id = 1;
    const uint64_t id = next_sweep_id++;
    A->sw = true; A->sw_tile_rows = p.tile_rows; A->sw_wlog = p.wlog; A->sw_n_tiles = p.n_sweep_tiles; A->sw_all_tiles = p.n_tiles;
    A->sw_x_len = p.x_len_min; A->sw_n_rest = (int64_t)p.rest_chunks.size(); A->sw_plan_id = id;
    A->sw_n_vals = (int64_t)p.idx.size() - 64; A->sw_n_vals_b = B ? (int64_t)p.idx_b.size() - 64 : 0; A->sw_cnt_bytes = (int64_t)p.cnt.size();
    if (B) { B->sw = true; B->sw_plan_id = id; B->sw_n_tiles = p.n_sweep_tiles; B->sw_all_tiles = p.n_tiles; }
    return USPMV_OK;
}

// The same plan from the handle's DEVICE arrays (csrc/sweep_plan_kernels.hip): a scan kernel per struct, the tile decisions and the
// offsets on the host (O(n_tiles); 16 bytes per 64-row group come back), a fill kernel per struct.  Same defaults, same criteria and
// -- by construction of the fill kernel -- the same arrays as sweep_plan_install builds from a host struct.
static int sweep_plan_install_device(uspmv_dmat_t *A, uspmv_dmat_t *B, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep, const char *who) {
    if (A->sw) sw_release(A);
    if (B && B->sw) sw_release(B);
    if (n_tiles) *n_tiles = 0;
    if (n_sweep) *n_sweep = 0;
    const int64_t C = A->C, nc = A->n_chunks;
    const size_t vsz = A->dtype == USPMV_F64 ? 8 : 4;
    if (wlog <= 0) wlog = g_tune.sweep_wlog;
    const int nbuf = g_tune.sweep_nbuf == 2 ? 2 : 1;
    if (wlog <= 0) wlog = (vsz == 8 ? 13 : 14) + (nbuf == 1 ? 1 : 0);
    if (((size_t)1 << wlog) * vsz * (size_t)nbuf > 160 * 1024)
        return uspmv::fail(USPMV_ERR_INVALID, "%s: %d window buffer(s) of 2^%d elements do not fit the 160 KB of LDS", who, nbuf, wlog);
    if (tile_rows <= 0) tile_rows = g_tune.sweep_tile_rows;
    const int64_t n_pad = nc * C;
    if (tile_rows <= 0) {
        tile_rows = 4096;
        while (tile_rows > 1024 && n_pad / tile_rows < 384) tile_rows /= 2;
    }
    if (tile_rows != 256 && tile_rows != 512 && tile_rows != 1024 && tile_rows != 2048 && tile_rows != 4096) tile_rows = 1024;
    if (C < 1 || C > 64 || 64 % C != 0 || nc < 1 || wlog < 8 || wlog > 16) return USPMV_OK;
    if (A->n_elements > (int64_t)UINT32_MAX || (B && B->n_elements > (int64_t)UINT32_MAX)) return USPMV_OK;
    const int64_t R = tile_rows, nt = (n_pad + R - 1) / R, wpt = R / 64, n_groups = (n_pad + 63) / 64;
    const int ns = B ? 2 : 1;
    const uspmv_dmat_t *M[2] = {A, B};
    const double max_stage = g_tune.sweep_max_stage > 0 ? (double)g_tune.sweep_max_stage : 24.0;
    // ---- scan
    int *d_le[2] = {nullptr, nullptr}, *d_pad[2] = {nullptr, nullptr}, *d_grp[2] = {nullptr, nullptr}, *d_max = nullptr;
    auto scratch_free = [&]() { for (int w = 0; w < 2; ++w) { (void)hipFree(d_le[w]); (void)hipFree(d_pad[w]); (void)hipFree(d_grp[w]); } (void)hipFree(d_max); };
    hipError_t e = hipMalloc((void **)&d_max, 4);
    if (e == hipSuccess) e = hipMemset(d_max, 0, 4);
    for (int w = 0; w < ns && e == hipSuccess; ++w) {
        e = hipMalloc((void **)&d_le[w], 4 * (size_t)n_groups * 64);
        if (e == hipSuccess) e = hipMalloc((void **)&d_pad[w], 4 * (size_t)n_groups * 64);
        if (e == hipSuccess) e = hipMalloc((void **)&d_grp[w], 16 * (size_t)n_groups);
    }
    if (e != hipSuccess) { scratch_free(); return uspmv::fail(USPMV_ERR_ALLOC, "%s: %s", who, hipGetErrorString(e)); }
    std::vector<int32_t> grp[2];
    int max_col = 0;
    int rc = USPMV_OK;
    for (int w = 0; w < ns && !rc; ++w) rc = launch_sweep_scan(M[w], wlog, d_le[w], d_pad[w], d_grp[w], d_max, nullptr);
    for (int w = 0; w < ns && !rc; ++w) {
        grp[w].resize((size_t)n_groups * 4);
        e = hipMemcpy(grp[w].data(), d_grp[w], 16 * (size_t)n_groups, hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    }
    if (!rc && hipMemcpy(&max_col, d_max, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = uspmv::fail(USPMV_ERR_HIP, "%s: scan results", who);
    if (rc) { scratch_free(); return rc; }
    // ---- which tiles sweep (sweep_plan.cpp pass 1), offsets
    std::vector<int32_t> tile_ids, t_smin, t_S, rest;
    std::vector<uint64_t> t_cnt_off;
    int64_t cnt_bytes = 0, tot[2] = {0, 0};
    for (int64_t t = 0; t < nt; ++t) {
        int32_t lo = INT32_MAX, hi = -1;
        bool good = true;
        int64_t nnz_t = 0;
        for (int w = 0; w < ns; ++w)
            for (int64_t g = t * wpt; g < std::min((t + 1) * wpt, n_groups); ++g) {
                const int32_t *q = grp[w].data() + (size_t)g * 4;
                nnz_t += q[0]; lo = std::min(lo, q[1]); hi = std::max(hi, q[2]); good = good && !q[3];
            }
        const int64_t nS = (int64_t)hi - lo + 1;
        const bool ok = good && hi >= 0 && (double)nS * (double)((int64_t)1 << wlog) * (double)vsz <= max_stage * (double)std::max<int64_t>(nnz_t, 1) && nS <= 4096;
        if (!ok) { for (int64_t c = t * R / C; c < std::min((t + 1) * R / C, nc); ++c) rest.push_back((int32_t)c); continue; }
        tile_ids.push_back((int32_t)t); t_smin.push_back(lo); t_S.push_back((int32_t)nS); t_cnt_off.push_back((uint64_t)cnt_bytes);
        cnt_bytes += nS * R;
    }
    const int64_t nsw = (int64_t)tile_ids.size();
    if (n_tiles) *n_tiles = nt;
    if (n_sweep) *n_sweep = nsw;
    if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] sweep plan (device builder): tile_rows=%d wlog=%d tiles=%lld sweep=%lld rest_chunks=%zu cnt_bytes=%lld\n",
                                         tile_rows, wlog, (long long)nt, (long long)nsw, rest.size(), (long long)cnt_bytes);
    if (nsw == 0) { scratch_free(); return USPMV_OK; }
    std::vector<uint32_t> wave_off[2];
    for (int w = 0; w < ns; ++w) {
        wave_off[w].assign((size_t)(nsw * wpt), 0);
        for (int64_t k = 0; k < nsw; ++k)
            for (int64_t v = 0; v < wpt; ++v) {
                wave_off[w][(size_t)(k * wpt + v)] = (uint32_t)tot[w];
                const int64_t g = (int64_t)tile_ids[(size_t)k] * wpt + v;
                if (g < n_groups) tot[w] += grp[w][(size_t)g * 4];
            }
        if (tot[w] > (int64_t)UINT32_MAX) { scratch_free(); return USPMV_OK; }
    }
    // ---- device arrays of the plan
    constexpr size_t SPARE = 64;
    auto up = [&](const void *h, size_t bytes, void **d) {
        if (e != hipSuccess) return;
        e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    auto zeroed = [&](size_t bytes, void **d) {
        if (e != hipSuccess) return;
        e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemset(*d, 0, bytes);
    };
    up(tile_ids.data(), tile_ids.size() * 4, (void **)&A->sw_tile_ids);
    up(t_smin.data(), t_smin.size() * 4, (void **)&A->sw_smin);
    up(t_S.data(), t_S.size() * 4, (void **)&A->sw_S);
    up(t_cnt_off.data(), t_cnt_off.size() * 8, (void **)&A->sw_cnt_off);
    up(wave_off[0].data(), wave_off[0].size() * 4, (void **)&A->sw_wave_off);
    up(rest.data(), rest.size() * 4, (void **)&A->sw_rest);
    zeroed((size_t)cnt_bytes, (void **)&A->sw_cnt);
    zeroed(((size_t)tot[0] + SPARE) * vsz, &A->sw_vals);
    zeroed(((size_t)tot[0] + SPARE) * 2, (void **)&A->sw_idx);
    zeroed((size_t)(nsw * R) * 4, (void **)&A->sw_pad);
    if (B) {
        up(wave_off[1].data(), wave_off[1].size() * 4, (void **)&A->sw_wave_off_b);
        zeroed((size_t)cnt_bytes, (void **)&A->sw_cnt_b);
        zeroed(((size_t)tot[1] + SPARE) * 4, (void **)&A->sw_vals_b);
        zeroed(((size_t)tot[1] + SPARE) * 2, (void **)&A->sw_idx_b);
        zeroed((size_t)(nsw * R) * 4, (void **)&A->sw_pad_b);
    }
    if (e == hipSuccess && launch_sweep_fill(A, wlog, (int)R, (long)nsw, A->sw_tile_ids, A->sw_smin, A->sw_S, (const unsigned long long *)A->sw_cnt_off, A->sw_wave_off,
                                             d_le[0], d_pad[0], A->sw_cnt, A->sw_vals, A->sw_idx, A->sw_pad, nullptr) != USPMV_OK) e = hipErrorUnknown;
    if (e == hipSuccess && B && launch_sweep_fill(B, wlog, (int)R, (long)nsw, A->sw_tile_ids, A->sw_smin, A->sw_S, (const unsigned long long *)A->sw_cnt_off, A->sw_wave_off_b,
                                                  d_le[1], d_pad[1], A->sw_cnt_b, A->sw_vals_b, A->sw_idx_b, A->sw_pad_b, nullptr) != USPMV_OK) e = hipErrorUnknown;
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    scratch_free();
    if (e != hipSuccess) {
        sw_release(A);
        return uspmv::fail(USPMV_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    }
    static uint64_t next_dev_sweep_id = (uint64_t)1 << 41;
    const uint64_t id = next_dev_sweep_id++;
    A->sw = true; A->sw_tile_rows = tile_rows; A->sw_wlog = wlog; A->sw_n_tiles = nsw; A->sw_all_tiles = nt;
    A->sw_x_len = (int64_t)max_col + 1; A->sw_n_rest = (int64_t)rest.size(); A->sw_plan_id = id;
    A->sw_n_vals = tot[0]; A->sw_n_vals_b = B ? tot[1] : 0; A->sw_cnt_bytes = cnt_bytes;
    if (B) { B->sw = true; B->sw_plan_id = id; B->sw_n_tiles = nsw; B->sw_all_tiles = nt; }
    return USPMV_OK;
}

int uspmv_dmat_optimize_sweep(uspmv_dmat_t *A, const uspmv_scs_t *s, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep) {
    if (!A || !s) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep: layout-only struct; the plan builder needs the host entries");
    if (A->C != s->C || A->n_chunks != s->n_chunks || A->dtype != s->dtype)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep: handle and host struct do not describe the same matrix");
    if (int rc = require_device()) return rc;
    return sweep_plan_install(A, nullptr, s, nullptr, wlog, tile_rows, n_tiles, n_sweep, "uspmv_dmat_optimize_sweep");
}

int uspmv_dmat_optimize_sweep_ap(uspmv_dmat_t *dp, uspmv_dmat_t *sp, const uspmv_scs_t *s_dp, const uspmv_scs_t *s_sp, int wlog, int tile_rows,
                                 int64_t *n_tiles, int64_t *n_sweep) {
    if (!dp || !sp || !s_dp || !s_sp) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep_ap: NULL argument");
    if (!uspmv::scs_has_entries(s_dp) || !uspmv::scs_has_entries(s_sp))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep_ap: layout-only struct; the plan builder needs the host entries");
    if (dp->C != s_dp->C || dp->n_chunks != s_dp->n_chunks || dp->dtype != USPMV_F64 || s_dp->dtype != USPMV_F64 ||
        sp->C != s_sp->C || sp->n_chunks != s_sp->n_chunks || sp->dtype != USPMV_F32 || s_sp->dtype != USPMV_F32 ||
        dp->C != sp->C || dp->n_chunks != sp->n_chunks)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep_ap: handles / host structs do not form a dp+sp pair");
    if (int rc = require_device()) return rc;
    return sweep_plan_install(dp, sp, s_dp, s_sp, wlog, tile_rows, n_tiles, n_sweep, "uspmv_dmat_optimize_sweep_ap");
}

static void bw_release(uspmv_dmat_t *A) {
    (void)hipFree(A->bw_tile_ids); (void)hipFree(A->bw_win_ptr); (void)hipFree(A->bw_wins); (void)hipFree(A->bw_pad); (void)hipFree(A->bw_cnt_off);
    (void)hipFree(A->bw_wave_off); (void)hipFree(A->bw_cnt); (void)hipFree(A->bw_vals); (void)hipFree(A->bw_idx);
    A->bw_tile_ids = A->bw_win_ptr = A->bw_wins = A->bw_pad = nullptr; A->bw_cnt_off = nullptr; A->bw_wave_off = nullptr; A->bw_cnt = nullptr;
    A->bw_vals = nullptr; A->bw_idx = nullptr;
    A->bw = false; A->bw_n_tiles = A->bw_all_tiles = 0; A->bw_b = 0;
}

// The block-vector column-window sweep plan (host/sweep_plan.cpp: uspmv_build_block_sweep_plan; kernel csrc/spmmv_sweep.hip) for 64-byte X
// rows.  Installed only when EVERY tile sweeps (rows column-sorted at window granularity, staging within "spmmv_sweep_max_stage" bytes
// per non-zero); otherwise the handle keeps whatever block plan it has.  wlog / tile_rows 0 = defaults (2^11 rows = 128 KiB windows, one
// buffer; 4 096-row tiles, fewer when the matrix is small).
int uspmv_dmat_optimize_block_sweep(uspmv_dmat_t *A, const uspmv_scs_t *s, int block_vec_size, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep) {
    if (!A || !s) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block_sweep: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block_sweep: layout-only struct; the plan builder needs the host entries");
    if (A->C != s->C || A->n_chunks != s->n_chunks || A->dtype != s->dtype)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block_sweep: handle and host struct do not describe the same matrix");
    if (int rc = require_device()) return rc;
    if (A->bw) bw_release(A);
    if (n_tiles) *n_tiles = 0;
    if (n_sweep) *n_sweep = 0;
    const size_t vsz = s->dtype == USPMV_F64 ? 8 : 4;
    if ((size_t)block_vec_size * vsz != 64) return USPMV_OK;                 // the kernel is written for 64-byte X rows
    // defaults: the largest window (2^11 rows = 128 KiB, one buffer) and 4 096-row tiles measured best on the Queen_4147-class matrix
    // (0.998 / 1.016 ms row- / column-wise; 2^9-row windows with two buffers 1.26 / 1.29: profiles/r04/spmmv_sweep_probe.txt)
    if (wlog <= 0) wlog = 11;
    if (((size_t)1 << wlog) * 64 > 160 * 1024) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block_sweep: a window of 2^%d rows does not fit the LDS", wlog);
    if (tile_rows <= 0) {
        tile_rows = 4096;
        while (tile_rows > 1024 && s->n_chunks * s->C / tile_rows < 384) tile_rows /= 2;   // (at least ~1.5 tiles per CU)
    }
    uspmv_block_sweep_plan p;
    const double max_stage = g_tune.sweep_max_stage > 0 ? (double)g_tune.sweep_max_stage : 24.0;
    if (int rc = uspmv_build_block_sweep_plan(s, wlog, tile_rows, 64, max_stage, &p)) return rc;
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_sweep) *n_sweep = p.valid ? p.n_sweep_tiles : 0;
    if (getenv("USPMV_VERBOSE"))
        fprintf(stderr, "[uspmv] block sweep plan: tile_rows=%d wlog=%d tiles=%lld sweep=%lld rest_chunks=%zu elements=%zu windows staged=%lld (%.2f X rows per matrix row) cnt_bytes=%zu\n",
                p.tile_rows, p.wlog, (long long)p.n_tiles, (long long)p.n_sweep_tiles, p.rest_chunks.size(), p.idx.size(), (long long)p.windows_staged,
                (double)p.windows_staged * (double)((int64_t)1 << p.wlog) / (double)std::max<int64_t>(s->n_chunks * s->C, 1), p.cnt.size());
    if (!p.valid || p.n_sweep_tiles != p.n_tiles || !p.rest_chunks.empty()) return USPMV_OK;
    hipError_t e = hipSuccess;
    auto up = [&](const void *h, size_t bytes, void **d) {
        if (e != hipSuccess) return;
        e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
    };
    up(p.tile_ids.data(), p.tile_ids.size() * 4, (void **)&A->bw_tile_ids);
    up(p.t_win_ptr.data(), p.t_win_ptr.size() * 4, (void **)&A->bw_win_ptr);
    up(p.wins.data(), p.wins.size() * 4, (void **)&A->bw_wins);
    up(p.t_cnt_off.data(), p.t_cnt_off.size() * 8, (void **)&A->bw_cnt_off);
    up(p.wave_off.data(), p.wave_off.size() * 4, (void **)&A->bw_wave_off);
    up(p.cnt.data(), p.cnt.size(), (void **)&A->bw_cnt);
    up(vsz == 8 ? (const void *)p.vals_f64.data() : (const void *)p.vals_f32.data(), p.idx.size() * vsz, &A->bw_vals);
    up(p.idx.data(), p.idx.size() * 2, (void **)&A->bw_idx);
    up(p.pad_col.data(), p.pad_col.size() * 4, (void **)&A->bw_pad);
    if (e != hipSuccess) { bw_release(A); (void)hipGetLastError(); return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize_block_sweep: %s", hipGetErrorString(e)); }
    A->bw = true; A->bw_tile_rows = p.tile_rows; A->bw_wlog = p.wlog; A->bw_b = block_vec_size;
    A->bw_n_tiles = p.n_sweep_tiles; A->bw_all_tiles = p.n_tiles; A->bw_x_rows = p.x_rows_min; A->bw_windows = p.windows_staged;
    return USPMV_OK;
}

void uspmv_dmat_free(uspmv_dmat_t *A) {
    if (!A) return;
    if (A->bw) bw_release(A);
    if (A->sw) sw_release(A);
    if (A->alt) { uspmv_dmat_free(A->alt); A->alt = nullptr; }
    if (A->tlc) tlc_release(A);
    if (A->bt || A->pb) bt_release(A);
    part_release(A, 0); part_release(A, 1);
    if (A->ws) (void)hipFree(A->ws);
    if (A->owns) {
        (void)hipFree((void *)A->chunk_ptrs); (void)hipFree((void *)A->chunk_lengths);
        (void)hipFree((void *)A->col_idxs); (void)hipFree((void *)A->values);
    }
    delete A;
}

int uspmv_dmat_set_crs(uspmv_dmat_t *A, int on) {
    if (!A) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_set_crs: NULL matrix");
    if (on && A->C != 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_set_crs: crs needs C = 1 (got %lld)", (long long)A->C);
    A->crs = on != 0;
    return USPMV_OK;
}

int uspmv_spmv(const uspmv_dmat_t *A, const void *d_x, void *d_y, void *stream) {
    if (int rc = check_dmat(A, "uspmv_spmv")) return rc;
    if (!d_x || !d_y) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv: NULL vector");
    if (int rc = require_device()) return rc;
    if (A->alt && g_tune.tlc && g_tune.rechunk && g_tune.spmv_variant == 0 && !g_tune.ablate) {
        if (A->dtype == USPMV_F64) return launch_spmv_scs<double>(A->alt, nullptr, 0, (const double *)d_x, (double *)d_y, (hipStream_t)stream);
        return launch_spmv_scs<float>(A->alt, nullptr, 0, (const float *)d_x, (float *)d_y, (hipStream_t)stream);
    }
    if (A->crs) {
        if (A->dtype == USPMV_F64)
            return launch_csr<double>((long)A->n_chunks, (long)A->n_elements, A->chunk_ptrs, A->col_idxs,
                                      (const double *)A->values, (const double *)d_x, (double *)d_y, (hipStream_t)stream);
        return launch_csr<float>((long)A->n_chunks, (long)A->n_elements, A->chunk_ptrs, A->col_idxs,
                                 (const float *)A->values, (const float *)d_x, (float *)d_y, (hipStream_t)stream);
    }
    if (A->dtype == USPMV_F64) return launch_spmv_scs<double>(A, nullptr, 0, (const double *)d_x, (double *)d_y, (hipStream_t)stream);
    return launch_spmv_scs<float>(A, nullptr, 0, (const float *)d_x, (float *)d_y, (hipStream_t)stream);
}

int uspmv_spmv_chunks(const uspmv_dmat_t *A, const int32_t *d_chunk_ids, int64_t n_ids, const void *d_x, void *d_y,
                      void *stream) {
    if (int rc = check_dmat(A, "uspmv_spmv_chunks")) return rc;
    if (n_ids < 0 || n_ids > A->n_chunks || (n_ids > 0 && !d_chunk_ids) || !d_x || !d_y)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv_chunks: bad argument");
    if (int rc = require_device()) return rc;
    if (n_ids == 0) return USPMV_OK;
    if (A->dtype == USPMV_F64) return launch_spmv_scs<double>(A, d_chunk_ids, n_ids, (const double *)d_x, (double *)d_y, (hipStream_t)stream);
    return launch_spmv_scs<float>(A, d_chunk_ids, n_ids, (const float *)d_x, (float *)d_y, (hipStream_t)stream);
}

int uspmv_spmv_tiles(const uspmv_dmat_t *A, const int32_t *d_tile_ids, int64_t n_ids, const void *d_x, void *d_y,
                     void *stream) {
    if (int rc = check_dmat(A, "uspmv_spmv_tiles")) return rc;
    if (!A->tlc || A->tlc_plan_id != 0) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv_tiles: handle has no tile-local-column plan (uspmv_dmat_optimize)");
    if (n_ids < 0 || n_ids > A->tlc_n_tiles || (n_ids > 0 && !d_tile_ids) || !d_x || !d_y)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv_tiles: bad argument");
    if ((uintptr_t)d_x % 16) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv_tiles: x must be 16-byte aligned");
    if (int rc = require_device()) return rc;
    if (A->dtype == USPMV_F64) return launch_spmv_tlc<double>(A, d_tile_ids, (long)n_ids, (const double *)d_x, (double *)d_y, (hipStream_t)stream);
    return launch_spmv_tlc<float>(A, d_tile_ids, (long)n_ids, (const float *)d_x, (float *)d_y, (hipStream_t)stream);
}

int uspmv_dmat_optimize_sweep_device(uspmv_dmat_t *A, uspmv_dmat_t *sp, int wlog, int tile_rows, int64_t *n_tiles, int64_t *n_sweep) {
    if (int rc = check_dmat(A, "uspmv_dmat_optimize_sweep_device")) return rc;
    if (sp) {
        if (int rc = check_dmat(sp, "uspmv_dmat_optimize_sweep_device")) return rc;
        if (A->dtype != USPMV_F64 || sp->dtype != USPMV_F32 || A->C != sp->C || A->n_chunks != sp->n_chunks)
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_sweep_device: handles do not form a dp+sp pair");
    }
    if (int rc = require_device()) return rc;
    return sweep_plan_install_device(A, sp, wlog, tile_rows, n_tiles, n_sweep, "uspmv_dmat_optimize_sweep_device");
}

// FNV-1a digests of the sweep plan's device arrays (tests: a plan built on the device must equal the host planner's)
int uspmv_dmat_sweep_plan_digest(const uspmv_dmat_t *A, uint64_t digest[16], int64_t meta[8]) {
    if (int rc = check_dmat(A, "uspmv_dmat_sweep_plan_digest")) return rc;
    if (!digest || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_sweep_plan_digest: NULL argument");
    for (int k = 0; k < 16; ++k) digest[k] = 0;
    meta[0] = A->sw; meta[1] = A->sw_tile_rows; meta[2] = A->sw_wlog; meta[3] = A->sw_n_tiles; meta[4] = A->sw_all_tiles; meta[5] = A->sw_n_rest;
    meta[6] = A->sw_n_vals; meta[7] = A->sw_n_vals_b;
    if (!A->sw || !A->sw_tile_ids) return USPMV_OK;
    std::vector<unsigned char> buf;
    auto fnv = [&](const void *d, size_t bytes, uint64_t *out) -> int {
        uint64_t h = 1469598103934665603ull;
        if (d && bytes) {
            buf.resize(bytes);
            HIP_TRY(hipMemcpy(buf.data(), d, bytes, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < bytes; ++k) { h ^= buf[k]; h *= 1099511628211ull; }
        }
        *out = h;
        return USPMV_OK;
    };
    const size_t nsw = (size_t)A->sw_n_tiles, wpt = (size_t)A->sw_tile_rows / 64, vsz = A->dtype == USPMV_F64 ? 8 : 4;
    int rc = fnv(A->sw_tile_ids, nsw * 4, &digest[0]);
    if (!rc) rc = fnv(A->sw_smin, nsw * 4, &digest[1]);
    if (!rc) rc = fnv(A->sw_S, nsw * 4, &digest[2]);
    if (!rc) rc = fnv(A->sw_cnt_off, nsw * 8, &digest[3]);
    if (!rc) rc = fnv(A->sw_wave_off, nsw * wpt * 4, &digest[4]);
    if (!rc) rc = fnv(A->sw_cnt, (size_t)A->sw_cnt_bytes, &digest[5]);
    if (!rc) rc = fnv(A->sw_vals, (size_t)A->sw_n_vals * vsz, &digest[6]);
    if (!rc) rc = fnv(A->sw_idx, (size_t)A->sw_n_vals * 2, &digest[7]);
    if (!rc) rc = fnv(A->sw_pad, nsw * (size_t)A->sw_tile_rows * 4, &digest[8]);
    if (!rc) rc = fnv(A->sw_rest, (size_t)A->sw_n_rest * 4, &digest[9]);
    if (!rc && A->sw_idx_b) {
        rc = fnv(A->sw_wave_off_b, nsw * wpt * 4, &digest[10]);
        if (!rc) rc = fnv(A->sw_cnt_b, (size_t)A->sw_cnt_bytes, &digest[11]);
        if (!rc) rc = fnv(A->sw_vals_b, (size_t)A->sw_n_vals_b * 4, &digest[12]);
        if (!rc) rc = fnv(A->sw_idx_b, (size_t)A->sw_n_vals_b * 2, &digest[13]);
        if (!rc) rc = fnv(A->sw_pad_b, nsw * (size_t)A->sw_tile_rows * 4, &digest[14]);
    }
    return rc;
}

int uspmv_dmat_plan_info(const uspmv_dmat_t *A, int *kind, int64_t *n_tiles, int64_t *n_planned) {
    if (!A) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_plan_info: NULL matrix");
    const uspmv_dmat_t *M = A->alt ? A->alt : A;
    int k = 0; int64_t nt = 0, np = 0;
    if (M->sw) { k = 2; nt = M->sw_all_tiles; np = M->sw_n_tiles; }
    else if (M->tlc) { k = 1; nt = M->tlc_n_tiles; np = M->tlc_staged; }
    if (kind) *kind = k;
    if (n_tiles) *n_tiles = nt;
    if (n_planned) *n_planned = np;
    return USPMV_OK;
}

// FNV-1a digests of the phased block plan's device arrays (tests: device-built == host-planned)
int uspmv_dmat_block_plan_digest(const uspmv_dmat_t *A0, uint64_t digest[8]) {
    if (int rc = check_dmat(A0, "uspmv_dmat_block_plan_digest")) return rc;
    if (!digest) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_block_plan_digest: NULL argument");
    const uspmv_dmat_t *A = (A0->alt && g_tune.rechunk) ? A0->alt : A0;
    for (int k = 0; k < 8; ++k) digest[k] = 0;
    if (!A->pb) return USPMV_OK;
    std::vector<unsigned char> buf;
    auto fnv = [&](const void *d, size_t bytes, uint64_t *out) -> int {
        uint64_t h = 1469598103934665603ull;
        if (d && bytes) {
            buf.resize(bytes);
            HIP_TRY(hipMemcpy(buf.data(), d, bytes, hipMemcpyDeviceToHost));
            for (size_t k = 0; k < bytes; ++k) { h ^= buf[k]; h *= 1099511628211ull; }
        }
        *out = h;
        return USPMV_OK;
    };
    const size_t nt = (size_t)A->pb_n_tiles, nph = (size_t)A->pb_n_phases, nc = (size_t)A->n_chunks, vsz = A->dtype == USPMV_F64 ? 8 : 4;
    int32_t n_list = 0;
    uint32_t tot16 = 0;
    HIP_TRY(hipMemcpy(&n_list, A->pb_list_ptr + nph, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&tot16, A->pb_c16_ptrs + nc, 4, hipMemcpyDeviceToHost));
    int rc = fnv(A->pb_ph_ptr, (nt + 1) * 4, &digest[0]);
    if (!rc) rc = fnv(A->pb_g0, nph * 4, &digest[1]);
    if (!rc) rc = fnv(A->pb_list_ptr, (nph + 1) * 4, &digest[2]);
    if (!rc) rc = fnv(A->pb_xrows, (size_t)n_list * 4, &digest[3]);
    if (!rc) rc = fnv(A->pb_c16_ptrs, (nc + 1) * 4, &digest[4]);
    if (!rc) rc = fnv(A->pb_col16, (size_t)tot16 * (A->pb_idx8 ? 1 : 2), &digest[5]);
    if (!rc) rc = fnv(A->pb_values, (size_t)tot16 * vsz, &digest[6]);
    if (!rc) rc = fnv(A->bt_row_map, A->bt_row_map ? nc * (size_t)A->C * 4 : 0, &digest[7]);
    return rc;
}

int uspmv_dmat_block_plan_staged(const uspmv_dmat_t *A, int64_t *rows_staged) {
    if (!A || !rows_staged) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_block_plan_staged: NULL argument");
    const uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;
    *rows_staged = M->pb ? M->pb_rows_staged : 0;
    return USPMV_OK;
}

int uspmv_dmat_block_plan_info(const uspmv_dmat_t *A, int64_t meta[10]) {
    if (!A || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_block_plan_info: NULL argument");
    const uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;
    meta[0] = M->bt; meta[1] = M->pb; meta[2] = M->pl; meta[3] = M->pb_n_tiles; meta[4] = M->pb_n_phases; meta[5] = M->pl_n_phases;
    meta[6] = M->pl_rows_staged; meta[7] = M->pb_idx8; meta[8] = M->pb_device_built; meta[9] = M->pb_max_rows;
    return USPMV_OK;
}

int uspmv_dmat_plan_granularity(const uspmv_dmat_t *A, int *elements_per_list_entry) {
    if (!A || !elements_per_list_entry) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_plan_granularity: NULL argument");
    const uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;
    *elements_per_list_entry = !M->tlc ? 0 : M->tlc_elem ? 1 : 16;
    return USPMV_OK;
}

int uspmv_dmat_plan_rows_dealt(const uspmv_dmat_t *A, int *dealt) {
    if (!A || !dealt) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_plan_rows_dealt: NULL argument");
    const uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;
    *dealt = M->tlc && M->tlc_row_map != nullptr;
    return USPMV_OK;
}

int uspmv_dmat_stream_info(const uspmv_dmat_t *A, int64_t meta[2]) {
    if (!A || !meta) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_stream_info: NULL argument");
    const uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;
    meta[0] = M->ps_desc ? M->ps_grid : 0; meta[1] = M->ps_desc ? M->ps_n_desc : 0;
    return USPMV_OK;
}

int uspmv_dmat_tile_rows(const uspmv_dmat_t *A, int *tile_rows) {
    if (!A || !tile_rows) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_tile_rows: NULL argument");
    *tile_rows = A->tlc ? A->tlc_tile_rows : 0;
    return USPMV_OK;
}

int uspmv_dmat_index_bits(const uspmv_dmat_t *A, int *bits) {
    if (!A || !bits) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_index_bits: NULL argument");
    const uspmv_dmat_t *M = (A->alt && g_tune.rechunk) ? A->alt : A;
    *bits = !M->tlc ? 0 : M->tlc_col12 ? 12 : 16;
    return USPMV_OK;
}

int uspmv_spmmv(const uspmv_dmat_t *A, const void *d_X, void *d_Y, int b, int64_t ld, int layout, void *stream) {
    if (int rc = check_dmat(A, "uspmv_spmmv")) return rc;
    if (!d_X || !d_Y || b < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmmv: bad argument");
    if (layout != USPMV_COLWISE && layout != USPMV_ROWWISE) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmmv: unknown layout %d", layout);
    if (layout == USPMV_COLWISE && ld < A->n_chunks * A->C)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmmv: ld=%lld smaller than n_rows_padded=%lld", (long long)ld,
                           (long long)(A->n_chunks * A->C));
    if (int rc = require_device()) return rc;
    // narrow-chunk handles optimised by uspmv_dmat_optimize carry an internal C = 32 re-chunking with the same row
    // order (crs: SELL-32-1): coalesced matrix stream for the block kernels too; its stores stop at the caller's rows
    if (A->alt && g_tune.rechunk) A = A->alt;
    if (A->dtype == USPMV_F64) return launch_spmmv<double>(A, (const double *)d_X, (double *)d_Y, b, (long)ld, layout, (hipStream_t)stream);
    return launch_spmmv<float>(A, (const float *)d_X, (float *)d_Y, b, (long)ld, layout, (hipStream_t)stream);
}

int uspmv_spmmv_x_prepared(const uspmv_dmat_t *A, const void *d_X, int b, int64_t ld, void *stream) {
    if (int rc = check_dmat(A, "uspmv_spmmv_x_prepared")) return rc;
    if (!d_X || b < 1 || ld < A->n_chunks * A->C) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmmv_x_prepared: bad argument");
    if (int rc = require_device()) return rc;
    if (A->alt && g_tune.rechunk) A = A->alt;
    A->xprep_ptr = nullptr;
    const int rc = A->dtype == USPMV_F64 ? prepare_x<double>(A, (const double *)d_X, b, (long)ld, (hipStream_t)stream)
                                         : prepare_x<float>(A, (const float *)d_X, b, (long)ld, (hipStream_t)stream);
    if (rc > 0) return USPMV_OK;                              // (this width / alignment has no re-layout pass: nothing to prepare, nothing skipped)
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_spmmv_x_release(const uspmv_dmat_t *A) {
    if (int rc = check_dmat(A, "uspmv_spmmv_x_release")) return rc;
    A->xprep_ptr = nullptr;
    if (A->alt) A->alt->xprep_ptr = nullptr;
    return USPMV_OK;
}

static int spmv_ap_impl(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, const float *d_x_sp,
                        double *d_y, void *stream, const char *who) {
    if (int rc = check_dmat(dp, who)) return rc;
    if (int rc = check_dmat(sp, who)) return rc;
    if (dp->dtype != USPMV_F64 || sp->dtype != USPMV_F32)
        return uspmv::fail(USPMV_ERR_INVALID, "%s: expects a double and a float struct", who);
    if (dp->C != sp->C || dp->n_chunks != sp->n_chunks)
        return uspmv::fail(USPMV_ERR_INVALID, "%s: dp and sp structs must share C and n_chunks", who);
    if (!d_x || !d_y) return uspmv::fail(USPMV_ERR_INVALID, "%s: NULL vector", who);
    if (int rc = require_device()) return rc;
    if (dp->n_chunks == 0) return USPMV_OK;
    return launch_spmv_ap(dp, sp, d_x, d_x_sp, d_y, (hipStream_t)stream);
}

int uspmv_spmv_ap(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, double *d_y, void *stream) {
    return spmv_ap_impl(dp, sp, d_x, nullptr, d_y, stream, "uspmv_spmv_ap");
}

int uspmv_spmv_ap_generic(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, const float *d_x_sp,
                          double *d_y, void *stream) {
    if (!d_x_sp) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv_ap_generic: NULL float x");
    return spmv_ap_impl(dp, sp, d_x, d_x_sp, d_y, stream, "uspmv_spmv_ap_generic");
}

// Opt-in plan cache of the raw-array entry points (tuning key "raw_plan_cache"): the reference's function-pointer
// seam passes the same device arrays on every call (code/classes_structs.hpp:997-1034), so the first call wraps them,
// builds the tile-local-column plan on the device and later calls run the plan kernel.  Keyed on the array
// addresses and shape -- the caller promises not to put a different matrix behind the same pointers
// (uspmv_raw_plan_cache_clear() after freeing or rewriting them).  Mutex-guarded; a handful of entries.
namespace {
struct RawKey { const void *cp, *cl, *ci, *va; int64_t C, n_chunks; int dtype; };
struct RawEntry { RawKey k; uspmv_dmat_t *A; };
std::vector<RawEntry> g_raw_cache;
std::mutex g_raw_mutex;

const uspmv_dmat *raw_cached(const RawKey &k) {
    std::lock_guard<std::mutex> lock(g_raw_mutex);
    for (const RawEntry &e : g_raw_cache)
        if (e.k.cp == k.cp && e.k.cl == k.cl && e.k.ci == k.ci && e.k.va == k.va && e.k.C == k.C && e.k.n_chunks == k.n_chunks && e.k.dtype == k.dtype)
            return e.A;
    int32_t last = 0;
    if (hipMemcpy(&last, (const int32_t *)k.cp + k.n_chunks, 4, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    uspmv_dmat_t *A = nullptr;
    if (uspmv_dmat_wrap(k.C, k.n_chunks, last, k.dtype, (const int32_t *)k.cp, (const int32_t *)k.cl, (const int32_t *)k.ci, k.va, &A)) return nullptr;
    if (uspmv_dmat_optimize_device(A, 0, nullptr, nullptr)) { uspmv_dmat_free(A); return nullptr; }
    if (g_raw_cache.size() >= 16) { uspmv_dmat_free(g_raw_cache.front().A); g_raw_cache.erase(g_raw_cache.begin()); }
    g_raw_cache.push_back(RawEntry{k, A});
    return A;
}
}  // namespace

void uspmv_raw_plan_cache_clear(void) {
    std::lock_guard<std::mutex> lock(g_raw_mutex);
    for (RawEntry &e : g_raw_cache) uspmv_dmat_free(e.A);
    g_raw_cache.clear();
}

#define RAW_SCS(SUF, VT, DT)                                                                                        \
    int uspmv_scs_gpu_##SUF(int64_t C, int64_t n_chunks, const int32_t *cp, const int32_t *cl, const int32_t *ci,   \
                            const VT *va, const VT *x, VT *y, void *stream) {                                       \
        if (C < 1 || n_chunks < 0 || !cp || (n_chunks > 0 && (!cl || !ci || !va || !x || !y)))                      \
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_scs_gpu_" #SUF ": bad argument");                          \
        if (int rc = require_device()) return rc;                                                                   \
        uspmv_dmat A;                                                                                               \
        A.C = C; A.n_chunks = n_chunks; A.dtype = DT; A.chunk_ptrs = cp; A.chunk_lengths = cl; A.col_idxs = ci;     \
        A.n_store = (long)(C * n_chunks);                                                                           \
        A.values = va;                                                                                              \
        if (int rc = check_dmat(&A, "uspmv_scs_gpu_" #SUF)) return rc;                                              \
        if (g_tune.raw_plan_cache && n_chunks > 0)                                                                  \
            if (const uspmv_dmat *P = raw_cached(RawKey{cp, cl, ci, va, C, n_chunks, DT}))                          \
                return launch_spmv_scs<VT>(P, nullptr, 0, x, y, (hipStream_t)stream);                               \
        return launch_spmv_scs<VT>(&A, nullptr, 0, x, y, (hipStream_t)stream);                                      \
    }                                                                                                               \
    int uspmv_csr_gpu_##SUF(int64_t n_rows, const int32_t *rp, const int32_t *ci, const VT *va, const VT *x, VT *y, \
                            void *stream) {                                                                         \
        if (n_rows < 0 || !rp || (n_rows > 0 && (!x || !y)))                                                        \
            return uspmv::fail(USPMV_ERR_INVALID, "uspmv_csr_gpu_" #SUF ": bad argument");                          \
        if (int rc = require_device()) return rc;                                                                   \
        return launch_csr<VT>((long)n_rows, 0, rp, ci, va, x, y, (hipStream_t)stream);                              \
    }
RAW_SCS(f64, double, USPMV_F64)
RAW_SCS(f32, float, USPMV_F32)
#undef RAW_SCS

static int peek_bytes(const void *p, void *out, size_t n, const char *who) {
    if (!p || !out) return uspmv::fail(USPMV_ERR_INVALID, "%s: NULL argument", who);
    hipPointerAttribute_t at;
    hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) { (void)hipGetLastError(); memcpy(out, p, n); return USPMV_OK; }   // not known to the runtime: plain host memory
    if (at.type == hipMemoryTypeDevice || at.type == hipMemoryTypeManaged) {
        HIP_TRY(hipMemcpy(out, p, n, hipMemcpyDeviceToHost));
        return USPMV_OK;
    }
    memcpy(out, p, n);
    return USPMV_OK;
}
int uspmv_peek_i64(const void *p, int64_t *out) { return peek_bytes(p, out, 8, "uspmv_peek_i64"); }
int uspmv_peek_i32(const void *p, int32_t *out) { return peek_bytes(p, out, 4, "uspmv_peek_i32"); }

int uspmv_apply_permutation_dev(void *d_out, const void *d_in, const int32_t *d_perm, int64_t n, int dtype,
                                void *stream) {
    if (!d_out || !d_in || !d_perm || n < 0) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_apply_permutation_dev: bad argument");
    if (int rc = require_device()) return rc;
    if (n == 0) return USPMV_OK;
    const unsigned grid = grid_for(n, 256);
    if (dtype == USPMV_F64)
        hipLaunchKernelGGL((gather_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)d_out,
                           (const double *)d_in, d_perm, (const int *)nullptr, (long)n, 0L);
    else if (dtype == USPMV_F32)
        hipLaunchKernelGGL((gather_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)d_out,
                           (const float *)d_in, d_perm, (const int *)nullptr, (long)n, 0L);
    else return uspmv::fail(USPMV_ERR_INVALID, "uspmv_apply_permutation_dev: unknown dtype %d", dtype);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_pack_send_buf(const void *d_x, const int32_t *d_perm, const int32_t *d_send_idxs, int64_t n,
                        int64_t block_offset, void *d_send, int dtype, void *stream) {
    if (n < 0 || (n > 0 && (!d_x || !d_perm || !d_send_idxs || !d_send)))
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_pack_send_buf: bad argument");
    if (int rc = require_device()) return rc;
    if (n == 0) return USPMV_OK;
    const unsigned grid = grid_for(n, 256);
    if (dtype == USPMV_F64)
        hipLaunchKernelGGL((gather_kernel<double>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (double *)d_send,
                           (const double *)d_x, d_perm, d_send_idxs, (long)n, (long)block_offset);
    else if (dtype == USPMV_F32)
        hipLaunchKernelGGL((gather_kernel<float>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)d_send,
                           (const float *)d_x, d_perm, d_send_idxs, (long)n, (long)block_offset);
    else return uspmv::fail(USPMV_ERR_INVALID, "uspmv_pack_send_buf: unknown dtype %d", dtype);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_stream_copy(double *a, const double *b, int64_t n, void *stream) {
    if (!a || !b || n < 0 || (n & 1)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_stream_copy: bad argument (n must be even)");
    if (int rc = require_device()) return rc;
    hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)((n / 2 + 2047) / 2048)), dim3(256), 0, (hipStream_t)stream, (double2 *)a, (const double2 *)b, (long)(n / 2));
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_stream_triad(double *a, const double *b, const double *c, double s, int64_t n, void *stream) {
    if (!a || !b || !c || n < 0 || (n & 1)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_stream_triad: bad argument (n must be even)");
    if (int rc = require_device()) return rc;
    hipLaunchKernelGGL(stream_triad_kernel, dim3((unsigned)((n / 2 + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, (double2 *)a, (const double2 *)b, (const double2 *)c, s, (long)(n / 2));
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_stream_read(const double *b, int64_t n, double *partial, void *stream) {
    if (!b || !partial || n < 0 || (n & 1)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_stream_read: bad argument (n must be even; partial needs 8192 doubles)");
    if (int rc = require_device()) return rc;
    hipLaunchKernelGGL(stream_read_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (const double2 *)b, (long)(n / 2), partial);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_stream_gather_lines(const double *x, int64_t n, int64_t plane, int64_t line, int rows, double *partial, void *stream, int64_t *bytes) {
    if (!x || !partial || n < 4096 || rows < 16 || rows > 4096 || plane < 0 || line < 0)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_stream_gather_lines: bad argument (partial needs 4 * ceil(n / rows) doubles)");
    if (int rc = require_device()) return rc;
    const long tiles = (long)((n + rows - 1) / rows);
    hipLaunchKernelGGL(stream_gather_lines_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const double2 *)x, (long)n, (long)plane, (long)line, rows,
                       g_tune.xcd_remap, partial);
    HIP_TRY(hipGetLastError());
    if (bytes) *bytes = tiles * 9 * (int64_t)(((rows + 2 + 15) / 16 + 1) * 128);   // (nine runs of whole lines per tile; the ends of the vector ask for a little less)
    return USPMV_OK;
}

int uspmv_time_launches(int what, int reps, const uspmv_dmat_t *A, const uspmv_dmat_t *B, const void *d_x, void *d_y,
                        int64_t n, int b, int64_t ld, int layout, void *stream, double *avg_ms) {
    if (reps < 1 || !avg_ms) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_time_launches: bad argument");
    if (int rc = require_device()) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipStream_t st = (hipStream_t)stream;
    int rc = USPMV_OK;
    for (int r = -2; r < reps && rc == USPMV_OK; ++r) {      // two untimed launches first (page tables, caches, clocks)
        if (r == 0) HIP_TRY(hipEventRecord(e0, st));
        switch (what) {
            case 0: rc = uspmv_spmv(A, d_x, d_y, stream); break;
            case 1: rc = uspmv_stream_copy((double *)d_y, (const double *)d_x, n, stream); break;
            case 2: rc = uspmv_stream_triad((double *)d_y, (const double *)d_x, (const double *)d_x + n, 3.0, n, stream); break;
            case 3: rc = uspmv_stream_read((const double *)d_x, n, (double *)d_y, stream); break;
            case 4: rc = uspmv_spmv_ap(A, B, (const double *)d_x, (double *)d_y, stream); break;
            case 5: rc = uspmv_spmmv(A, d_x, d_y, b, ld, layout, stream); break;
            case 6: rc = uspmv_stream_gather_lines((const double *)d_x, n, ld, (int64_t)b, layout, (double *)d_y, stream, nullptr); break;
            default: rc = uspmv::fail(USPMV_ERR_INVALID, "uspmv_time_launches: unknown kind %d", what);
        }
    }
    hipError_t e = hipEventRecord(e1, st);
    if (e == hipSuccess) e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (rc != USPMV_OK) return rc;
    if (e != hipSuccess) return uspmv::fail(USPMV_ERR_HIP, "uspmv_time_launches: %s", hipGetErrorString(e));
    *avg_ms = (double)ms / reps;
    return USPMV_OK;
}

}  // extern "C"
