// Hand-written HIP kernels for gfx950 (CDNA4, MI355X) and the device half of the C ABI.
// Written for 64-lane wavefronts and the 8-XCD / per-XCD-L2 memory system; no CUDA path exists.
//
// Kernels (reference CPU twin each one reproduces, paths relative to the reference root):
//   scs_spmv_rows      spmv_omp_scs / scs_impl_cpu<C>            code/kernels.hpp:159-258
//   scs_spmv_split2    same maths, two lanes per row (C = 32)     (tolerance variant)
//   csr_spmv_vector    spmv_omp_csr                               code/kernels.hpp:22-63
//   scs_spmmv_rows     block_spmv_omp_scs_general                 code/kernels.hpp:306-398
//   scs_spmv_ap_rows   scs_ap_impl_cpu<C>                         code/ap_kernels.hpp:24-82
//   gather_kernel      pack_send_buf / apply_permutation          code/classes_structs.hpp:813-818,
//                                                                 code/utilities.hpp:1768-1782
//
// Data layout in HBM (identical to the reference's ScsData, code/classes_structs.hpp:1313-1339):
// element (row-in-chunk i, slot j) of chunk c at chunk_ptrs[c] + j*C + i.  A wavefront that owns
// 64/C consecutive chunks (lane <-> row) therefore reads, per slot j, 64/C contiguous segments
// of C*sizeof(VT) bytes of `values` and C*4 bytes of `col_idxs`: the matrix stream is perfectly
// coalesced and read exactly once; it is issued with non-temporal loads so that it does not
// evict the x vector, whose irregular 8-byte gathers are served by the XCD's L2 / the
// Infinity Cache.  One lane walks one row in slot order j = 0,1,2,... with one fused
// multiply-add per element, which is bit-for-bit the summation the reference's CPU kernels
// perform (g++ -O3 contracts `tmp += a*b` to an FMA).
//
// Workgroup -> chunk mapping: hardware deals workgroups round-robin over the 8 XCDs; with
// xcd_remap the logical block id is permuted so that every XCD walks its own contiguous eighth of
// the chunk range -- the x window of a region is then fetched into one L2 instead of eight.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../host/uspmv_internal.hpp"

struct uspmv_dmat {
    int64_t C = 0, n_chunks = 0, n_elements = 0;
    int dtype = USPMV_F64;
    const int32_t *chunk_ptrs = nullptr, *chunk_lengths = nullptr, *col_idxs = nullptr;
    const void *values = nullptr;
    bool owns = false;
    bool crs = false;
    long n_store = 0;              // rows of y the kernels may write (= n_chunks*C unless re-chunked)
    uspmv_dmat *alt = nullptr;     // internal C = 32 re-chunking of a C in {1,2,4,8,16} struct (same row order)
    // scratch for the internal row-major copies of column-major block vectors (uspmv_spmmv); grown
    // on demand, released with the handle.  Not thread-safe per handle, like the reference's kernel object.
    mutable void *ws = nullptr;
    mutable size_t ws_bytes = 0;
    // tile-local-column plan (host/tlc_plan.cpp), device copies owned by the handle
    bool tlc = false;
    int tlc_max_lines = 0, tlc_tile_rows = 256;
    int64_t tlc_x_len = 0, tlc_n_tiles = 0, tlc_staged = 0;
    uint64_t tlc_plan_id = 0;   // structs planned together (ap pair) carry the same non-zero id
    int32_t *tlc_line_ptr = nullptr, *tlc_lines = nullptr;
    uint32_t *tlc_c16_ptrs = nullptr;
    uint16_t *tlc_col16 = nullptr;
    // block (SpMMV) plan: 64-row tiles, per tile the list of X rows it touches (uspmv_dmat_optimize_block)
    bool bt = false;
    int bt_max_rows = 0, bt_tile_rows = 64;
    int64_t bt_n_tiles = 0, bt_staged = 0;
    int32_t *bt_line_ptr = nullptr, *bt_xrows = nullptr;
    uint32_t *bt_c16_ptrs = nullptr;
    uint16_t *bt_col16 = nullptr;
};

namespace {

struct Tuning {
    // defaults = fastest of the interleaved sweep on the nlpkkt200-class matrix (profiles/r01_sweep253.txt)
    int unroll = 8;
    int nontemporal = 1;
    int xcd_remap = 256;  // groups of 256 consecutive workgroups per XCD (profiles/r01/sweepH.txt)
    int block = 256;
    int spmv_variant = 0;
    int csr_lanes = 0;  // 0 = choose from average row length
    int ablate = 0;     // measurement only
    int tlc = 1;            // use the tile-local-column kernel when the handle carries a plan
    int rechunk = 1;        // uspmv_dmat_optimize may re-chunk C < 32 structs to C = 32 internally
    int tlc_tile_rows = 256;  // rows (= threads) per tile used by the NEXT uspmv_dmat_optimize
    int tail_batch = 0;     // ragged tail of a chunk as one predicated batch
    int spmmv_unroll = 0;   // 0 = auto (256 bytes of X rows per lane and batch)
    int spmmv_lds_kb = 0;    // block plan: LDS budget per tile in KiB for the NEXT uspmv_dmat_optimize_block (0 = 80)
    int spmmv_tile_rows = 0; // block plan: 0 = auto (32-row tiles for >= 64-byte rows on C = 32), 64 = always 64
    int spmmv_prefetch = 1; // row-major lane-per-row kernel: request batch k+1's matrix entries behind batch k's X rows
    int spmmv_variant = 0;  // 0 = auto (= 3 where a B-specialised kernel exists); 1 = generic kernel; 2 = row-major with transposing X phase; 3 = row-major, lane per row
};
Tuning g_tune;

#define HIP_TRY(call)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return uspmv::fail(USPMV_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                               __FILE__, __LINE__);                                                \
    } while (0)

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

// ------------------------------------------------------------------------------------------
template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T *p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
// y store.  A y vector is ~2 % of the bytes of an SpMV, but its HBM write stream costs 15-18 % of the
// kernel when it goes through the write-back L2 (profiles/r01_microbench.txt: 0.74 ms without the
// store, 0.88 ms with plain stores, 0.83 ms write-through).  Relaxed agent-scope atomic stores
// compile to `global_store_dword[x2] ... sc1` (write-through, line not kept in L2).
template <bool WT, typename T>
__device__ __forceinline__ void st_y(T *p, T v) {
    if constexpr (WT) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// logical block id.  Hardware deals blocks round-robin over the 8 XCDs (b, b+8, b+16, ... share
// one).  mode 0: identity.  mode 1: every XCD walks one contiguous eighth of the grid.
// mode G >= 2: groups of G consecutive logical blocks per XCD, the 8 groups of a super-block
// of 8*G blocks being processed concurrently (keeps all XCDs inside one moving DRAM window while
// neighbouring blocks -- which share x lines -- share an L2).
__device__ __forceinline__ unsigned remap_block(unsigned b, unsigned nb, int mode) {
    if (mode == 0 || nb < 16) return b;
    if (mode == 1) {
        const unsigned xcd = b & 7u, q = nb >> 3, r = nb & 7u;
        const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        return base + (b >> 3);
    }
    const unsigned G = (unsigned)mode, SG = 8u * G;
    const unsigned full = (nb / SG) * SG;
    if (b >= full) return b;
    const unsigned sup = b / SG, rem = b - sup * SG;
    return sup * SG + (rem & 7u) * G + (rem >> 3);
}

// ------------------------------------------------------------------------------------------
// SELL-C-sigma SpMV, one lane per row.  CT > 0: compile-time C; CT == 0: C passed at run time.
// IDS: virtual chunk v -> chunk_ids[v] (interior / boundary subsets).
// ABL != 0: measurement-only ablations (WRONG results): 1 = every gather hits one 512-byte window
// of x (keeps the instruction stream, removes L1 misses), 2 = no gather at all.
template <typename VT, int CT, int U, bool NT, bool IDS, int ABL = 0, bool TAILB = false>
__global__ void scs_spmv_rows(const long n_work_chunks, const int C_rt, const int *__restrict__ chunk_ptrs,
                              const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                              const VT *__restrict__ values, const VT *__restrict__ x, VT *__restrict__ y,
                              const int *__restrict__ chunk_ids, const int xcd_remap, const long n_store) {
    const int C = CT > 0 ? CT : C_rt;
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long vrow = (long)lb * blockDim.x + threadIdx.x;
    const long vc = vrow / C;
    const int i = (int)(vrow - vc * C);
    if (vc >= n_work_chunks) return;
    const long c = IDS ? (long)chunk_ids[vc] : vc;
    const long cs = chunk_ptrs[c];
    const int L = chunk_lengths[c];
    const VT *vp = values + cs + i;
    const int *cp = col_idxs + cs + i;
    VT acc = VT(0);
    int j = 0;
    for (; j + U <= L; j += U) {
        VT v[U];
        int ci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = ld_stream<NT>(vp + (long)(j + u) * C);
            ci[u] = ld_stream<NT>(cp + (long)(j + u) * C);
        }
        VT xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = ABL == 0 ? x[ci[u]] : ABL == 1 ? x[ci[u] & 63] : (VT)ci[u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fma_t(v[u], xv[u], acc);
    }
    if (TAILB) {
        // ragged tail (< U slots) as ONE predicated batch: all its loads issue back to back under the
        // lane mask, then all its gathers -- 2 dependent round trips instead of 2 per leftover slot
        if (j < L) {
            VT v[U];
            int ci[U];
            VT xv[U];
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                v[u] = VT(0); ci[u] = 0;
                if (j + u < L) { v[u] = ld_stream<NT>(vp + (long)(j + u) * C); ci[u] = ld_stream<NT>(cp + (long)(j + u) * C); }
            }
#pragma unroll
            for (int u = 0; u < U - 1; ++u) {
                xv[u] = VT(0);
                if (j + u < L) xv[u] = ABL == 0 ? x[ci[u]] : ABL == 1 ? x[ci[u] & 63] : (VT)ci[u];
            }
#pragma unroll
            for (int u = 0; u < U - 1; ++u)
                if (j + u < L) acc = fma_t(v[u], xv[u], acc);
        }
    } else {
        for (; j < L; ++j) {
            const VT v = ld_stream<NT>(vp + (long)j * C);
            const int ci = ld_stream<NT>(cp + (long)j * C);
            acc = fma_t(v, ABL == 0 ? x[ci] : ABL == 1 ? x[ci & 63] : (VT)ci, acc);
        }
    }
    if (c * C + i < n_store) st_y<NT>(y + (c * C + i), acc);   // n_store < n_rows_padded only for re-chunked structs
}

// Software-pipelined form of scs_spmv_rows (same lane <-> row mapping, same FMA chain, bit-exact):
//   * the loop runs over wave-uniform batches of U slots up to the longest chunk of the wave; a
//     lane past the end of its own chunk re-reads its last slot (clamped index, always in bounds)
//     and its accumulate is predicated off -- so the ragged tail is ONE masked batch instead of up
//     to U-1 dependent single-slot round trips;
//   * the matrix stream of batch k+1 is issued BEHIND the x gathers of batch k.  vmcnt retires in
//     order, so the gathers (issued first) are waited for with the 2*U prefetch loads still in
//     flight: while a wave waits for its gathers it already has its next 64*U*12 bytes coming.
//     The steady-state body is straight-line code (no divergent branch), which is what lets the
//     compiler emit the counted s_waitcnt vmcnt(2*U + ...) instead of vmcnt(0).
// Dependent memory round trips per wave: 2 + ceil(L/U) instead of 2 + 2*(L/U) + 2*(L%U).
template <typename VT, int CT, int U, bool NT, bool IDS>
__global__ void scs_spmv_rows_pipe(const long n_work_chunks, const int C_rt, const int *__restrict__ chunk_ptrs,
                                   const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                                   const VT *__restrict__ values, const VT *__restrict__ x, VT *__restrict__ y,
                                   const int *__restrict__ chunk_ids, const int xcd_remap, const long n_store) {
    const int C = CT > 0 ? CT : C_rt;
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long vrow = (long)lb * blockDim.x + threadIdx.x;
    const long vc = vrow / C;
    const int i = (int)(vrow - vc * C);
    const bool valid = vc < n_work_chunks;
    long c = 0;
    int cs = 0, L = 0;
    if (valid) {
        c = IDS ? (long)chunk_ids[vc] : vc;
        cs = chunk_ptrs[c];
        L = chunk_lengths[c];
    }
    int Lmax = L;  // longest chunk of this wavefront (wave-uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o, 64));
    Lmax = __builtin_amdgcn_readfirstlane(Lmax);
    VT acc = VT(0);
    if (Lmax > 0) {
        // lanes with an empty chunk (or past the grid) stream element 0 of the arrays: always valid
        const long base = L > 0 ? (long)cs + i : 0;
        const int last = L > 0 ? L - 1 : 0;
        const VT *vp = values + base;
        const int *cp = col_idxs + base;
        VT v0[U];
        int c0[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long o = (long)(u < last ? u : last) * C;
            v0[u] = ld_stream<NT>(vp + o);
            c0[u] = ld_stream<NT>(cp + o);
        }
        int j = 0;
        for (; j + U < Lmax; j += U) {
            VT xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = x[c0[u]];
            __builtin_amdgcn_sched_barrier(0);  // gathers first: they are what the FMAs below wait for
            VT v1[U];
            int c1[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int jj = j + U + u;
                const long o = (long)(jj < last ? jj : last) * C;
                v1[u] = ld_stream<NT>(vp + o);
                c1[u] = ld_stream<NT>(cp + o);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the FMA block (hipcc sinks it otherwise)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const VT t = fma_t(v0[u], xv[u], acc);
                acc = (j + u < L) ? t : acc;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) { v0[u] = v1[u]; c0[u] = c1[u]; }
        }
        VT xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[c0[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const VT t = fma_t(v0[u], xv[u], acc);
            acc = (j + u < L) ? t : acc;
        }
    }
    if (valid && c * C + i < n_store) st_y<NT>(y + (c * C + i), acc);
}

// SpMV over a tile-local-column plan (host/tlc_plan.cpp).  One 256-thread workgroup = one tile of
// 256/C chunks.  Phase 1: the workgroup copies the tile's x lines (16 elements each, listed in
// tile_lines) into LDS with coalesced 16-byte loads.  Phase 2: lane <-> row as in scs_spmv_rows,
// but the column stream is the 2-byte LDS-local index array (four slots per 8-byte load) and the x
// operand comes from LDS (ds_read) instead of a 64-lane global gather.  Same slot-ordered FMA chain
// per row -> bit-exact.  Tiles without a line list (footprint too wide) take the global-gather path.
template <typename VT, int CT, bool NT, bool IDS>
__global__ void __launch_bounds__(1024) scs_spmv_tlc(const long n_chunks, const int C_rt, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        const VT *__restrict__ x, VT *__restrict__ y, const int *__restrict__ tile_line_ptr,
        const int *__restrict__ tile_lines, const unsigned *__restrict__ c16_ptrs,
        const unsigned short *__restrict__ col16, const long x_len, const int *__restrict__ tile_ids,
        const int xcd_remap, const long n_store) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tlc_smem[];
    VT *xs = (VT *)tlc_smem;
    constexpr int EPL = 16 / (int)sizeof(VT);   // elements per 16-byte load
    constexpr int LPL = 16 / EPL;               // lanes that copy one 16-element line
    typedef VT vec_t __attribute__((ext_vector_type(EPL)));
    const int C = CT > 0 ? CT : C_rt;
    const unsigned lbt = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const unsigned tile = IDS ? (unsigned)tile_ids[lbt] : lbt;
    const int lp0 = tile_line_ptr[tile];
    const int nl = tile_line_ptr[tile + 1] - lp0;
    const long row = (long)tile * blockDim.x + threadIdx.x;
    const long c = row / C;
    const int i = (int)(row - c * C);
    const bool valid = c < n_chunks;
    int cs = 0, L = 0;
    unsigned q0 = 0;
    if (valid) { cs = chunk_ptrs[c]; L = chunk_lengths[c]; q0 = c16_ptrs[c]; }
    VT acc = VT(0);
    if (nl > 0) {
        const int sub = threadIdx.x % LPL, lk = threadIdx.x / LPL;
        for (int k = lk; k < nl; k += blockDim.x / LPL) {
            const long idx = (long)tile_lines[lp0 + k] * 16 + sub * EPL;
            vec_t v;
            if (idx + EPL <= x_len) {
                v = *(const vec_t *)(x + idx);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] = idx + e < x_len ? x[idx + e] : VT(0);
            }
            *(vec_t *)(xs + k * 16 + sub * EPL) = v;
        }
        __syncthreads();
        if (L > 0) {
            const VT *vp = values + (long)cs + i;
            const unsigned long long *cq = (const unsigned long long *)(col16 + q0) + i;
            const int ng = L >> 2;
            int g = 0;
            for (; g + 2 <= ng; g += 2) {
                VT v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
                const unsigned long long qa = ld_stream<NT>(cq + (long)g * C), qb = ld_stream<NT>(cq + (long)(g + 1) * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fma_t(v[u], xs[(qa >> (16 * u)) & 0xFFFFu], acc);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fma_t(v[4 + u], xs[(qb >> (16 * u)) & 0xFFFFu], acc);
            }
            for (; g < ng; ++g) {
                VT v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
                const unsigned long long qa = ld_stream<NT>(cq + (long)g * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = fma_t(v[u], xs[(qa >> (16 * u)) & 0xFFFFu], acc);
            }
            const int rem = L & 3;
            if (rem) {
                const unsigned long long qa = ld_stream<NT>(cq + (long)ng * C);
                for (int u = 0; u < rem; ++u) acc = fma_t(ld_stream<NT>(vp + (long)(4 * ng + u) * C), xs[(qa >> (16 * u)) & 0xFFFFu], acc);
            }
        }
    } else if (L > 0) {  // wide-footprint tile: 32-bit columns, global gathers
        const VT *vp = values + (long)cs + i;
        const int *cp = col_idxs + (long)cs + i;
        int j = 0;
        for (; j + 8 <= L; j += 8) {
            VT v[8]; int ci[8]; VT xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = ld_stream<NT>(vp + (long)(j + u) * C); ci[u] = ld_stream<NT>(cp + (long)(j + u) * C); }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x[ci[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma_t(v[u], xv[u], acc);
        }
        for (; j < L; ++j) acc = fma_t(ld_stream<NT>(vp + (long)j * C), x[ld_stream<NT>(cp + (long)j * C)], acc);
    }
    if (valid && row < n_store) st_y<NT>(y + row, acc);
}

// Adaptive precision dp+sp over a tile-local-column plan shared by the two structs (one line list
// per tile covering the columns of both): x lines staged once, then the dp chain (8-byte values +
// 2-byte local indices) and the sp chain (4-byte values + 2-byte local indices), y = dp + sp.
// 10 and 6 bytes per non-zero instead of 12 and 8; numerics of scs_ap_impl_cpu, bit-exact.
template <int CT, bool NT>
__global__ void __launch_bounds__(1024) scs_spmv_ap_tlc(const long n_chunks, const int C_rt,
        const int *__restrict__ dp_cp, const int *__restrict__ dp_cl, const int *__restrict__ dp_ci, const double *__restrict__ dp_va,
        const int *__restrict__ sp_cp, const int *__restrict__ sp_cl, const int *__restrict__ sp_ci, const float *__restrict__ sp_va,
        const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ tile_line_ptr,
        const int *__restrict__ tile_lines, const unsigned *__restrict__ dp_c16p, const unsigned short *__restrict__ dp_c16,
        const unsigned *__restrict__ sp_c16p, const unsigned short *__restrict__ sp_c16, const long x_len, const int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tlc_smem[];
    double *xs = (double *)tlc_smem;
    typedef double vec_t __attribute__((ext_vector_type(2)));
    const int C = CT > 0 ? CT : C_rt;
    const unsigned tile = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int lp0 = tile_line_ptr[tile];
    const int nl = tile_line_ptr[tile + 1] - lp0;
    const long row = (long)tile * blockDim.x + threadIdx.x;
    const long c = row / C;
    const int i = (int)(row - c * C);
    const bool valid = c < n_chunks;
    int dcs = 0, Ld = 0, scs_ = 0, Ls = 0;
    unsigned dq0 = 0, sq0 = 0;
    if (valid) { dcs = dp_cp[c]; Ld = dp_cl[c]; scs_ = sp_cp[c]; Ls = sp_cl[c]; dq0 = dp_c16p[c]; sq0 = sp_c16p[c]; }
    double dt = 0.0, st = 0.0;
    if (nl > 0) {
        const int sub = threadIdx.x & 7, lk = threadIdx.x >> 3;
        for (int k = lk; k < nl; k += blockDim.x >> 3) {
            const long idx = (long)tile_lines[lp0 + k] * 16 + sub * 2;
            vec_t v;
            if (idx + 2 <= x_len) v = *(const vec_t *)(x + idx);
            else { v[0] = idx < x_len ? x[idx] : 0.0; v[1] = 0.0; }
            *(vec_t *)(xs + k * 16 + sub * 2) = v;
        }
        __syncthreads();
        if (Ld > 0) {
            const double *vp = dp_va + (long)dcs + i;
            const unsigned long long *cq = (const unsigned long long *)(dp_c16 + dq0) + i;
            const int ng = Ld >> 2;
            int g = 0;
            for (; g + 2 <= ng; g += 2) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
                const unsigned long long qa = ld_stream<NT>(cq + (long)g * C), qb = ld_stream<NT>(cq + (long)(g + 1) * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) dt = __builtin_fma(v[u], xs[(qa >> (16 * u)) & 0xFFFFu], dt);
#pragma unroll
                for (int u = 0; u < 4; ++u) dt = __builtin_fma(v[4 + u], xs[(qb >> (16 * u)) & 0xFFFFu], dt);
            }
            for (; g < ng; ++g) {
                double v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
                const unsigned long long qa = ld_stream<NT>(cq + (long)g * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) dt = __builtin_fma(v[u], xs[(qa >> (16 * u)) & 0xFFFFu], dt);
            }
            const int rem = Ld & 3;
            if (rem) {
                const unsigned long long qa = ld_stream<NT>(cq + (long)ng * C);
                for (int u = 0; u < rem; ++u) dt = __builtin_fma(ld_stream<NT>(vp + (long)(4 * ng + u) * C), xs[(qa >> (16 * u)) & 0xFFFFu], dt);
            }
        }
        if (Ls > 0) {
            const float *vp = sp_va + (long)scs_ + i;
            const unsigned long long *cq = (const unsigned long long *)(sp_c16 + sq0) + i;
            const int ng = Ls >> 2;
            int g = 0;
            for (; g + 2 <= ng; g += 2) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
                const unsigned long long qa = ld_stream<NT>(cq + (long)g * C), qb = ld_stream<NT>(cq + (long)(g + 1) * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) st = __builtin_fma((double)v[u], xs[(qa >> (16 * u)) & 0xFFFFu], st);
#pragma unroll
                for (int u = 0; u < 4; ++u) st = __builtin_fma((double)v[4 + u], xs[(qb >> (16 * u)) & 0xFFFFu], st);
            }
            for (; g < ng; ++g) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
                const unsigned long long qa = ld_stream<NT>(cq + (long)g * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) st = __builtin_fma((double)v[u], xs[(qa >> (16 * u)) & 0xFFFFu], st);
            }
            const int rem = Ls & 3;
            if (rem) {
                const unsigned long long qa = ld_stream<NT>(cq + (long)ng * C);
                for (int u = 0; u < rem; ++u) st = __builtin_fma((double)ld_stream<NT>(vp + (long)(4 * ng + u) * C), xs[(qa >> (16 * u)) & 0xFFFFu], st);
            }
        }
    } else {  // wide-footprint tile: 32-bit columns, global gathers
        const double *dvp = dp_va + (long)dcs + i;
        const int *dcp = dp_ci + (long)dcs + i;
        for (int j = 0; j < Ld; ++j) dt = __builtin_fma(ld_stream<NT>(dvp + (long)j * C), x[ld_stream<NT>(dcp + (long)j * C)], dt);
        const float *svp = sp_va + (long)scs_ + i;
        const int *scp = sp_ci + (long)scs_ + i;
        for (int j = 0; j < Ls; ++j) st = __builtin_fma((double)ld_stream<NT>(svp + (long)j * C), x[ld_stream<NT>(scp + (long)j * C)], st);
    }
    if (valid) st_y<NT>(y + row, dt + st);
}

// C = 32, one wavefront per chunk, two lanes per row: lane l owns row l & 31 and the slots
// j == (l >> 5) (mod 2), so every wave-instruction of the matrix stream is one contiguous
// 512-byte (values) / 256-byte (col_idxs) segment.  The two partial sums are combined with one
// cross-lane add: NOT the sequential chain -> compared against the oracle with a tolerance.
template <typename VT, int U, bool NT>
__global__ void scs_spmv_split2(const long n_chunks, const int *__restrict__ chunk_ptrs,
                                const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                                const VT *__restrict__ values, const VT *__restrict__ x, VT *__restrict__ y,
                                const int xcd_remap) {
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long c = ((long)lb * blockDim.x + threadIdx.x) >> 6;
    if (c >= n_chunks) return;
    const int lane = threadIdx.x & 63;
    const long cs = chunk_ptrs[c];
    const int L = chunk_lengths[c];
    const VT *vp = values + cs + lane;   // slot pair t: element index t*64 + lane
    const int *cp = col_idxs + cs + lane;
    const int h = lane >> 5;
    const int T = (L + 1 - h) >> 1;       // number of slots j = 2t + h < L
    VT acc = VT(0);
    int t = 0;
    for (; t + U <= T; t += U) {
        VT v[U];
        int ci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = ld_stream<NT>(vp + (long)(t + u) * 64);
            ci[u] = ld_stream<NT>(cp + (long)(t + u) * 64);
        }
        VT xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[ci[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) acc = fma_t(v[u], xv[u], acc);
    }
    for (; t < T; ++t) {
        const VT v = ld_stream<NT>(vp + (long)t * 64);
        const int ci = ld_stream<NT>(cp + (long)t * 64);
        acc = fma_t(v, x[ci], acc);
    }
    const VT other = __shfl_xor(acc, 32, 64);
    if (h == 0) st_y<NT>(y + (c * 32 + lane), acc + other);
}

// CRS SpMV: G lanes per row (G = power of two <= 64), lane-strided partial sums, shuffle
// reduction.  The reference's own loop is `omp simd`-reassociated (code/kernels.hpp:49), so
// there is no canonical order to be bit-exact with; compared with a tolerance.
template <typename VT, int G, bool NT>
__global__ void csr_spmv_vector(const long n_rows, const int *__restrict__ row_ptrs,
                                const int *__restrict__ col_idxs, const VT *__restrict__ values,
                                const VT *__restrict__ x, VT *__restrict__ y) {
    const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gt / G;
    const int g = (int)(gt % G);
    VT acc = VT(0);
    if (row < n_rows) {
        const int b = row_ptrs[row], e = row_ptrs[row + 1];
        for (int k = b + g; k < e; k += G) acc = fma_t(ld_stream<NT>(values + k), x[ld_stream<NT>(col_idxs + k)], acc);
    }
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < n_rows && g == 0) y[row] = acc;
}

// SELL-C-sigma SpMMV (block of b vectors), one lane per row, VB vectors per pass held in registers.
// colwise: X[col + v*ld], Y[row + v*ld];  rowwise: X[col*b + v], Y[row*b + v].
template <typename VT, int VB, bool ROWWISE, bool NT>
__global__ void scs_spmmv_rows(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
                               const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                               const VT *__restrict__ values, const VT *__restrict__ X, VT *__restrict__ Y,
                               const int b, const long ld, const int xcd_remap) {
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long row = (long)lb * blockDim.x + threadIdx.x;
    const long c = row / C;
    const int i = (int)(row - c * C);
    if (c >= n_chunks) return;
    const long cs = chunk_ptrs[c];
    const int L = chunk_lengths[c];
    const VT *vp = values + cs + i;
    const int *cp = col_idxs + cs + i;
    for (int v0 = 0; v0 < b; v0 += VB) {
        VT acc[VB];
#pragma unroll
        for (int v = 0; v < VB; ++v) acc[v] = VT(0);
        for (int j = 0; j < L; ++j) {
            const VT a = ld_stream<NT>(vp + (long)j * C);
            const long col = ld_stream<NT>(cp + (long)j * C);
#pragma unroll
            for (int v = 0; v < VB; ++v) {
                if (v0 + v < b) {
                    const VT xv = ROWWISE ? X[col * b + v0 + v] : X[col + (long)(v0 + v) * ld];
                    acc[v] = fma_t(a, xv, acc[v]);
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VB; ++v) {
            if (v0 + v < b) {
                if (ROWWISE) st_y<NT>(Y + (row * b + v0 + v), acc[v]);
                else st_y<NT>(Y + (row + (long)(v0 + v) * ld), acc[v]);
            }
        }
    }
}

constexpr size_t BT_LDS_CAP = 80 * 1024;  // LDS per single-wave SpMMV tile (block plan): two tiles per CU at worst

// SpMMV with ROW-MAJOR block vectors of compile-time width B (X[col*B + v]): one lane per row, B
// accumulators per lane.  Per slot a lane reads its whole X row -- B*sizeof(VT) contiguous bytes --
// with 16-byte loads, so one wave-instruction moves 1 KiB of X instead of 512 B of eight-byte
// column gathers: 1 + 1 + B*sizeof(VT)/16 vector-memory instructions per 64 non-zeros (the
// column-major form needs 2 + B, each fetching a whole 64-byte sector per lane for 8 useful bytes).
// Every (row, v) accumulator is still the slot-ordered FMA chain of block_spmv_omp_scs_general.
// YCOL: write Y column-major (Y[row + v*ld]) straight from the accumulators -- per vector one
// coalesced 64-lane store -- so that column-major callers only pay the X re-layout.
template <typename VT, int B, int U, bool NT, bool YCOL, bool PF>
__global__ void scs_spmmv_rowmajor(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
                                   const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                                   const VT *__restrict__ values, const VT *__restrict__ X, VT *__restrict__ Y,
                                   const long ld, const int xcd_remap) {
    constexpr int VW = 16 / (int)sizeof(VT);  // elements per 16-byte load
    constexpr int NV = B / VW;
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long row = (long)lb * blockDim.x + threadIdx.x;
    const long c = row / C;
    const int i = (int)(row - c * C);
    if (c >= n_chunks) return;
    const long cs = chunk_ptrs[c];
    const int L = chunk_lengths[c];
    const VT *vp = values + cs + i;
    const int *cp = col_idxs + cs + i;
    VT acc[B];
#pragma unroll
    for (int v = 0; v < B; ++v) acc[v] = VT(0);
    int j = 0;
    if (PF) {
        // PF: the (value, column) pairs of batch k+1 are requested right after the X rows of batch k, so a
        // wave has both round trips in flight instead of one after the other (the kernel is latency-bound:
        // 8 waves per SIMD x 2 dependent misses per batch).  Loads retire in order, so waiting for the X
        // rows does not wait for the prefetch.  Past the end the prefetch re-reads slot L-1 and is ignored.
        if (L >= U) {
            VT a[U];
            int ci[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = ld_stream<NT>(vp + (long)u * C); ci[u] = ld_stream<NT>(cp + (long)u * C); }
            for (; j + U <= L; j += U) {
                vec_t xr[U][NV];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const vec_t *xp = (const vec_t *)(X + (long)ci[u] * B);
#pragma unroll
                    for (int k = 0; k < NV; ++k) xr[u][k] = xp[k];
                }
                __builtin_amdgcn_sched_barrier(0);
                VT an[U];
                int cn[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int jj = min(j + U + u, L - 1);
                    an[u] = ld_stream<NT>(vp + (long)jj * C); cn[u] = ld_stream<NT>(cp + (long)jj * C);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < U; ++u)
#pragma unroll
                    for (int k = 0; k < NV; ++k)
#pragma unroll
                        for (int w = 0; w < VW; ++w) acc[k * VW + w] = fma_t(a[u], xr[u][k][w], acc[k * VW + w]);
#pragma unroll
                for (int u = 0; u < U; ++u) { a[u] = an[u]; ci[u] = cn[u]; }
            }
        }
    } else {
        for (; j + U <= L; j += U) {
            VT a[U];
            int ci[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = ld_stream<NT>(vp + (long)(j + u) * C); ci[u] = ld_stream<NT>(cp + (long)(j + u) * C); }
            vec_t xr[U][NV];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const vec_t *xp = (const vec_t *)(X + (long)ci[u] * B);
#pragma unroll
                for (int k = 0; k < NV; ++k) xr[u][k] = xp[k];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < NV; ++k)
#pragma unroll
                    for (int w = 0; w < VW; ++w) acc[k * VW + w] = fma_t(a[u], xr[u][k][w], acc[k * VW + w]);
        }
    }
    for (; j < L; ++j) {
        const VT a = ld_stream<NT>(vp + (long)j * C);
        const vec_t *xp = (const vec_t *)(X + (long)ld_stream<NT>(cp + (long)j * C) * B);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const vec_t xv = xp[k];
#pragma unroll
            for (int w = 0; w < VW; ++w) acc[k * VW + w] = fma_t(a, xv[w], acc[k * VW + w]);
        }
    }
    if (YCOL) {
#pragma unroll
        for (int v = 0; v < B; ++v) st_y<NT>(Y + (row + (long)v * ld), acc[v]);
    } else {
        vec_t *yp = (vec_t *)(Y + row * B);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            vec_t t;
#pragma unroll
            for (int w = 0; w < VW; ++w) t[w] = acc[k * VW + w];
            yp[k] = t;
        }
    }
}

// Row-major SpMMV, "transposing" form.  rocprofv3 counters on scs_spmmv_rowmajor (profiles/r01/spmmv_pmc.txt)
// show the vector L1 saturated (846 M 64-byte accesses = 54 GB per launch for 25 GB of useful bytes):
// a lane that owns a whole 64-byte X row fetches it as four 16-byte pieces in four instructions, and
// every piece costs a full 64-byte L1 access.  Here the matrix stream keeps its lane <-> row mapping
// (one coalesced 512-byte / 256-byte load per slot and wave), but the X phase runs in P = B*sizeof(VT)/16
// rounds over 64/P rows each with P adjacent lanes per row: a round's loads fetch whole contiguous X
// rows (one L1 access per row), the (value, column) pairs reaching the gathering lanes through
// ds_bpermute.  Lane (r, g) accumulates piece g of rows r, r + 64/P, ...; every (row, v) chain is still
// slot-ordered -> bit-exact.
template <typename VT, int B, int U, bool NT, bool YCOL, bool PF>
__global__ void scs_spmmv_xpose(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
                                const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                                const VT *__restrict__ values, const VT *__restrict__ X, VT *__restrict__ Y,
                                const long ld, const int xcd_remap) {
    constexpr int VW = 16 / (int)sizeof(VT);
    constexpr int P = B / VW;          // 16-byte pieces per X row = rounds
    constexpr int RPR = 64 / P;        // rows per round
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int lane = threadIdx.x & 63;
    const long row = (long)lb * blockDim.x + threadIdx.x;      // streaming role: this lane's row
    const long wrow0 = row - lane;
    const long c = row / C;
    const int i = (int)(row - c * C);
    int L = 0;
    long cs = 0;
    if (c < n_chunks) { cs = chunk_ptrs[c]; L = chunk_lengths[c]; }
    int Lmax = L;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) Lmax = max(Lmax, __shfl_xor(Lmax, o, 64));
    Lmax = __builtin_amdgcn_readfirstlane(Lmax);
    const VT *vp = values + cs + i;
    const int *cp = col_idxs + cs + i;
    const int rl = lane / P, g = lane % P;                      // gathering role: row-in-round, piece
    const vec_t *Xg = (const vec_t *)X + g;
    vec_t acc[P];
#pragma unroll
    for (int q = 0; q < P; ++q)
#pragma unroll
        for (int w = 0; w < VW; ++w) acc[q][w] = VT(0);
    // one batch of U slots of this lane's row: value 0 / column -1 past the end of the chunk
    auto load_batch = [&](int j, VT (&a)[U], int (&ci)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            a[u] = VT(0); ci[u] = -1;
            if (j + u < L) { a[u] = ld_stream<NT>(vp + (long)(j + u) * C); ci[u] = ld_stream<NT>(cp + (long)(j + u) * C); }
        }
    };
    VT a[U];
    int ci[U];
    if (PF) load_batch(0, a, ci);
    for (int j = 0; j < Lmax; j += U) {
        if (!PF) load_batch(j, a, ci);
        vec_t xv[U][P];
        VT aa[U][P];
        int cc[U][P];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < P; ++q) {
                cc[u][q] = __shfl(ci[u], q * RPR + rl, 64);
                aa[u][q] = __shfl(a[u], q * RPR + rl, 64);
                xv[u][q] = Xg[(long)(cc[u][q] < 0 ? 0 : cc[u][q]) * P];
            }
        if (PF) {   // next batch's matrix entries requested behind this batch's X rows (see scs_spmmv_rowmajor)
            __builtin_amdgcn_sched_barrier(0);
            load_batch(j + U, a, ci);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < P; ++q)
#pragma unroll
                for (int w = 0; w < VW; ++w) {
                    const VT t = fma_t(aa[u][q], xv[u][q][w], acc[q][w]);
                    acc[q][w] = cc[u][q] >= 0 ? t : acc[q][w];
                }
    }
    const long n_pad = n_chunks * (long)C;
#pragma unroll
    for (int q = 0; q < P; ++q) {
        const long r = wrow0 + q * RPR + rl;
        if (r < n_pad) {
            if (YCOL) {
#pragma unroll
                for (int w = 0; w < VW; ++w) st_y<NT>(Y + (r + (long)(g * VW + w) * ld), acc[q][w]);
            } else {
                ((vec_t *)Y)[r * P + g] = acc[q];
            }
        }
    }
}

// Row-major SpMMV over a block plan (uspmv_dmat_optimize_block): one 64-row tile per single-wave
// workgroup.  The gather kernels above stop at the L2 -> CU path (every non-zero pulls its 16*NV-byte X
// row through L1: 16.6 GB per launch on config 3); here a tile's distinct X rows (listed
// by the plan, 6-8x fewer than its non-zeros) cross that path once, by LDS-DMA (global_load_lds_dwordx4:
// per-lane source address, lane-linear destination, no VGPRs), and every non-zero reads its operand with
// ds_read_b128 through the 2-byte local index stream.  LDS holds 2-3 tiles per CU, so latency is hidden
// by depth instead of occupancy: the row list first (a 4-byte DMA into LDS), then up to NB register
// batches of 4*G slots of matrix entries and the whole X DMA are in flight together.  X rows sit piece-swizzled in LDS (physical
// piece = piece ^ f(row), applied to the DMA's source address and to the reads) so that the 16 lanes
// of a ds_read_b128 group spread over all 64 banks.  C is a template parameter (32 | 64) so that a
// batch addresses its slots with immediate offsets.  Same slot-ordered FMA chain per (row, v):
// bit-exact.  Tiles without a row list (footprint too large for LDS) gather from global memory.
// HS = 2 (C = 32, rows of >= 64 bytes): 32-row tiles, two lanes per row with half of the B columns each --
// half the LDS per tile, so twice the tiles per CU to overlap one tile's staging with another's arithmetic.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

template <typename VT, int B, bool NT, bool YCOL, int G, int C, int HS>
__global__ void __launch_bounds__(64) scs_spmmv_tlc(const long n_chunks, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        const VT *__restrict__ X, VT *__restrict__ Y, const long ld, const int *__restrict__ tile_line_ptr,
        const int *__restrict__ tile_xrows, const unsigned *__restrict__ c16_ptrs,
        const unsigned short *__restrict__ col16, const long x_rows, const int xcd_remap, const long n_store, const int x_bytes) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tlc_smem[];
    constexpr int VW = 16 / (int)sizeof(VT);  // elements per 16-byte piece
    constexpr int NV = B / VW;                // pieces per X row (1, 2, 4, 8)
    constexpr int NVS = NV == 1 ? 0 : NV == 2 ? 1 : NV == 4 ? 2 : 3;
    constexpr int SWS = 4 - NVS;              // rows 2^SWS apart start on the same bank
    constexpr int NB = 4;                     // register batches in flight
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    typedef unsigned long long u64;
    const vec_t *xs = (const vec_t *)tlc_smem;
    const unsigned tile = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int lp0 = tile_line_ptr[tile];
    const int nl = tile_line_ptr[tile + 1] - lp0;
    const int lane = threadIdx.x;
    // HS = 2: two lanes per row (lane and lane + 32), each owning half of the row's B columns; a tile is 32 rows
    constexpr int NVH = NV / HS;              // 16-byte pieces of an X row per lane
    constexpr int BH = B / HS;                // accumulators per lane
    const int h = HS == 2 ? lane >> 5 : 0;
    const long row = (long)tile * (64 / HS) + (HS == 2 ? lane & 31 : lane);
    const long c = row / C;
    const int i = (int)(row - c * C);
    const bool valid = c < n_chunks;
    int cs = 0, L = 0;
    unsigned q0 = 0;
    if (valid) { cs = chunk_ptrs[c]; L = chunk_lengths[c]; q0 = c16_ptrs[c]; }
    VT acc[BH];
#pragma unroll
    for (int v = 0; v < BH; ++v) acc[v] = VT(0);
    const VT *vp = values + (long)cs + i;
    auto fma_row = [&](const VT a, const vec_t *xp, const unsigned sw) {
#pragma unroll
        for (int k = 0; k < NVH; ++k) {
            const vec_t xv = xp[(unsigned)(k + h * NVH) ^ sw];
#pragma unroll
            for (int w = 0; w < VW; ++w) acc[k * VW + w] = fma_t(a, xv[w], acc[k * VW + w]);
        }
    };
    auto fma_local = [&](const VT a, const unsigned local) {
        fma_row(a, xs + local * NV, NV > 1 ? (local >> SWS) & (NV - 1) : 0u);
    };
    if (nl > 0) {
        const u64 *cq = (const u64 *)(col16 + q0) + i;
        int Lmin = L, Lmax = L;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { Lmin = min(Lmin, __shfl_xor(Lmin, o, 64)); Lmax = max(Lmax, __shfl_xor(Lmax, o, 64)); }
        Lmin = __builtin_amdgcn_readfirstlane(Lmin);
        Lmax = __builtin_amdgcn_readfirstlane(Lmax);
        const int ngf = Lmin >> 2;            // groups of four slots every lane of the wave has in full
        const int nbt = ngf / G;              // register batches
        // ---- 1. the tile's row list -> LDS (behind the X rows), by DMA as well: one latency, no registers
        const int np = nl << NVS;
        int *rl = (int *)(tlc_smem + x_bytes);
#pragma unroll 1
        for (int r0 = 0; r0 < nl; r0 += 64)
            if (r0 + lane < nl)
                __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(tile_xrows + lp0 + r0 + lane), (lds_void_t *)(rl + r0), 4, 0, 0);
        __syncthreads();                      // (drains the DMA: vmcnt(0) + barrier)
        // ---- 2. matrix entries: up to NB batches requested before anything is waited for
        auto load_batch = [&](const int bi, VT (&a)[4 * G], u64 (&q)[G]) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const long gg = (long)bi * G + g;
                q[g] = ld_stream<NT>(cq + gg * C);
#pragma unroll
                for (int u = 0; u < 4; ++u) a[4 * g + u] = ld_stream<NT>(vp + (4 * gg + u) * C);
            }
        };
        auto compute_batch = [&](const VT (&a)[4 * G], const u64 (&q)[G]) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    fma_local(a[4 * g + u], (unsigned)(q[g] >> (16 * u)) & 0xFFFFu);
                    if (NV >= 8 || u == 3) __builtin_amdgcn_sched_barrier(0);   // at most 64 VGPRs of LDS reads ahead of their FMAs
                }
            }
        };
        VT a0[4 * G], a1[4 * G], a2[NB > 2 ? 4 * G : 1], a3[NB > 2 ? 4 * G : 1];
        u64 qa[G], qb[G], qc[NB > 2 ? G : 1], qd[NB > 2 ? G : 1];
        if (nbt > 0) load_batch(0, a0, qa);
        if (nbt > 1) load_batch(1, a1, qb);
        if constexpr (NB > 2) {
            if (nbt > 2) load_batch(2, a2, qc);
            if (nbt > 3) load_batch(3, a3, qd);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. X rows -> LDS by DMA: LDS position p = (row k, physical piece pp) takes logical piece pp ^ f(k)
#pragma unroll 1
        for (int t0 = 0; t0 * 64 < np; t0 += 8) {
            int xr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = (t0 + u) * 64 + lane;
                xr[u] = p < np ? rl[p >> NVS] : 0;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = (t0 + u) * 64 + lane;
                if (p < np) {
                    const unsigned k = (unsigned)p >> NVS;
                    const unsigned piece = ((unsigned)p & (NV - 1)) ^ (NV > 1 ? (k >> SWS) & (NV - 1) : 0u);
                    const VT *src = X + (long)xr[u] * B + piece * VW;
                    __builtin_amdgcn_global_load_lds((glb_cvoid_t *)src, (lds_void_t *)(tlc_smem + (t0 + u) * 1024), 16, 0, 0);
                }
            }
        }
        __syncthreads();                      // drains the DMA (vmcnt) and the batches requested before it
        for (int bi = 0; bi < nbt; bi += NB) {
            compute_batch(a0, qa);
            if (bi + NB < nbt) load_batch(bi + NB, a0, qa);
            if (bi + 1 < nbt) { compute_batch(a1, qb); if (bi + 1 + NB < nbt) load_batch(bi + 1 + NB, a1, qb); }
            if constexpr (NB > 2) {
                if (bi + 2 < nbt) { compute_batch(a2, qc); if (bi + 2 + NB < nbt) load_batch(bi + 2 + NB, a2, qc); }
                if (bi + 3 < nbt) { compute_batch(a3, qd); if (bi + 3 + NB < nbt) load_batch(bi + 3 + NB, a3, qd); }
            }
        }
        for (int g = nbt * G; g < ngf; ++g) {  // full groups that do not fill a batch
            const u64 q = ld_stream<NT>(cq + (long)g * C);
            VT a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = ld_stream<NT>(vp + (long)(4 * g + u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) fma_local(a[u], (unsigned)(q >> (16 * u)) & 0xFFFFu);
        }
        const unsigned short *c16 = col16 + q0;
        for (int j = 4 * ngf; j < Lmax; ++j)   // ragged rest: lanes past their chunk's length sit out
            if (j < L) fma_local(ld_stream<NT>(vp + (long)j * C), c16[(long)(j >> 2) * 4 * C + i * 4 + (j & 3)]);
    } else if (L > 0) {  // wide-footprint tile: 32-bit columns, X rows gathered from global memory
        const int *cp = col_idxs + (long)cs + i;
        for (int j = 0; j < L; ++j)
            fma_row(ld_stream<NT>(vp + (long)j * C), (const vec_t *)(X + (long)ld_stream<NT>(cp + (long)j * C) * B), 0u);
    }
    if (!valid || row >= n_store) return;
    if (YCOL) {
#pragma unroll
        for (int v = 0; v < BH; ++v) st_y<NT>(Y + (row + (long)(h * BH + v) * ld), acc[v]);
    } else {
        vec_t *yp = (vec_t *)(Y + row * B) + h * NVH;
#pragma unroll
        for (int k = 0; k < NVH; ++k) {
            vec_t t;
#pragma unroll
            for (int w = 0; w < VW; ++w) t[w] = acc[k * VW + w];
            yp[k] = t;
        }
    }
}

// colwise (b vectors of leading dimension ld) <-> row-major (n rows of B) re-layout, one lane per row
template <typename VT, int B, bool TO_ROWMAJOR>
__global__ void block_vector_relayout(const VT *__restrict__ in, VT *__restrict__ out, const long n, const long ld) {
    const long r = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    if (TO_ROWMAJOR) {
        VT t[B];
#pragma unroll
        for (int v = 0; v < B; ++v) t[v] = in[r + (long)v * ld];
#pragma unroll
        for (int v = 0; v < B; ++v) out[r * B + v] = t[v];
    } else {
        VT t[B];
#pragma unroll
        for (int v = 0; v < B; ++v) t[v] = in[r * B + v];
#pragma unroll
        for (int v = 0; v < B; ++v) out[r + (long)v * ld] = t[v];
    }
}

// Adaptive precision dp+sp, one lane per row: the dp chain, then the sp chain (float value widened,
// times the DOUBLE x, accumulated in double), y = dp + sp  (code/ap_kernels.hpp:59-75).
// SPX: the generic-C reference kernel spmv_omp_scs_ap multiplies the sp values with the FLOAT copy
// of x (float product, rounded, then widened and added; code/ap_kernels.hpp:619-623).
template <int U, bool NT, bool SPX>
__global__ void scs_spmv_ap_rows(const long n_chunks, const int C, const int *__restrict__ dp_cp,
                                 const int *__restrict__ dp_cl, const int *__restrict__ dp_ci,
                                 const double *__restrict__ dp_va, const int *__restrict__ sp_cp,
                                 const int *__restrict__ sp_cl, const int *__restrict__ sp_ci,
                                 const float *__restrict__ sp_va, const double *__restrict__ x,
                                 const float *__restrict__ x_sp, double *__restrict__ y, const int xcd_remap) {
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long row = (long)lb * blockDim.x + threadIdx.x;
    const long c = row / C;
    const int i = (int)(row - c * C);
    if (c >= n_chunks) return;
    double dt = 0.0, st = 0.0;
    const long dcs = dp_cp[c], scs_ = sp_cp[c];
    const int Ld = dp_cl[c], Ls = sp_cl[c];
    const double *dvp = dp_va + dcs + i;
    const int *dcp = dp_ci + dcs + i;
    const float *svp = sp_va + scs_ + i;
    const int *scp = sp_ci + scs_ + i;
    int jd = 0, js = 0;
    // fused part: one dp batch and one sp batch per trip -- the two accumulators are independent
    // chains, so their streams and gathers are issued together (twice the bytes in flight per
    // wave); each chain still runs in slot order, i.e. bit-identical to dp-then-sp.
    for (; jd + U <= Ld && js + U <= Ls; jd += U, js += U) {
        double dv[U]; int dci[U]; float sv[U]; int sci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dv[u] = ld_stream<NT>(dvp + (long)(jd + u) * C); dci[u] = ld_stream<NT>(dcp + (long)(jd + u) * C);
            sv[u] = ld_stream<NT>(svp + (long)(js + u) * C); sci[u] = ld_stream<NT>(scp + (long)(js + u) * C);
        }
        double dx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) dx[u] = x[dci[u]];
        if constexpr (SPX) {
            float sx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) sx[u] = x_sp[sci[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) { dt = __builtin_fma(dv[u], dx[u], dt); st = st + (double)__fmul_rn(sv[u], sx[u]); }
        } else {
            double sx[U];
#pragma unroll
            for (int u = 0; u < U; ++u) sx[u] = x[sci[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) { dt = __builtin_fma(dv[u], dx[u], dt); st = __builtin_fma((double)sv[u], sx[u], st); }
        }
    }
    for (; jd + U <= Ld; jd += U) {
        double v[U]; int ci[U]; double xv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = ld_stream<NT>(dvp + (long)(jd + u) * C); ci[u] = ld_stream<NT>(dcp + (long)(jd + u) * C); }
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = x[ci[u]];
#pragma unroll
        for (int u = 0; u < U; ++u) dt = __builtin_fma(v[u], xv[u], dt);
    }
    for (; jd < Ld; ++jd) dt = __builtin_fma(ld_stream<NT>(dvp + (long)jd * C), x[ld_stream<NT>(dcp + (long)jd * C)], dt);
    for (; js + U <= Ls; js += U) {
        float v[U]; int ci[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { v[u] = ld_stream<NT>(svp + (long)(js + u) * C); ci[u] = ld_stream<NT>(scp + (long)(js + u) * C); }
        if constexpr (SPX) {
            float xs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xs[u] = x_sp[ci[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) st = st + (double)__fmul_rn(v[u], xs[u]);
        } else {
            double xv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) xv[u] = x[ci[u]];
#pragma unroll
            for (int u = 0; u < U; ++u) st = __builtin_fma((double)v[u], xv[u], st);
        }
    }
    for (; js < Ls; ++js) {
        const float v = ld_stream<NT>(svp + (long)js * C);
        const int ci = ld_stream<NT>(scp + (long)js * C);
        if constexpr (SPX) st = st + (double)__fmul_rn(v, x_sp[ci]);
        else st = __builtin_fma((double)v, x[ci], st);
    }
    st_y<NT>(y + row, dt + st);
}

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

// STREAM-style calibrators: 16 bytes per lane, grid-stride.
__global__ void stream_copy_kernel(double2 *__restrict__ a, const double2 *__restrict__ b, const long n2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) a[i] = b[i];
}
__global__ void stream_triad_kernel(double2 *__restrict__ a, const double2 *__restrict__ b,
                                    const double2 *__restrict__ c, const double s, const long n2) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
        double2 bb = b[i], cc = c[i];
        a[i] = make_double2(bb.x + s * cc.x, bb.y + s * cc.y);
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

// ------------------------------------------------------------------------------------------ launchers
inline unsigned grid_for(long work_items, int block) { return (unsigned)((work_items + block - 1) / block); }

template <typename VT, int CT, int U, bool NT>
void launch_rows_ids(bool ids, unsigned grid, int block, hipStream_t st, long nwc, int C, const uspmv_dmat *A,
                     const VT *x, VT *y, const int *chunk_ids) {
    if (g_tune.spmv_variant == 2) {
        if (ids)
            hipLaunchKernelGGL((scs_spmv_rows_pipe<VT, CT, U, NT, true>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                               A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
        else
            hipLaunchKernelGGL((scs_spmv_rows_pipe<VT, CT, U, NT, false>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                               A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
        return;
    }
    if (g_tune.tail_batch && U > 1) {
        if (ids)
            hipLaunchKernelGGL((scs_spmv_rows<VT, CT, U, NT, true, 0, true>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                               A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
        else
            hipLaunchKernelGGL((scs_spmv_rows<VT, CT, U, NT, false, 0, true>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                               A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
        return;
    }
    if (ids)
        hipLaunchKernelGGL((scs_spmv_rows<VT, CT, U, NT, true>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                           A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
    else
        hipLaunchKernelGGL((scs_spmv_rows<VT, CT, U, NT, false>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                           A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
}

template <typename VT, int CT, int U>
void launch_rows_nt(bool ids, unsigned grid, int block, hipStream_t st, long nwc, int C, const uspmv_dmat *A,
                    const VT *x, VT *y, const int *chunk_ids) {
    if (g_tune.nontemporal) launch_rows_ids<VT, CT, U, true>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids);
    else launch_rows_ids<VT, CT, U, false>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids);
}

template <typename VT, int CT>
void launch_rows_unroll(bool ids, unsigned grid, int block, hipStream_t st, long nwc, int C, const uspmv_dmat *A,
                        const VT *x, VT *y, const int *chunk_ids) {
    switch (g_tune.unroll) {
        case 1: launch_rows_nt<VT, CT, 1>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
        case 2: launch_rows_nt<VT, CT, 2>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
        case 8: launch_rows_nt<VT, CT, 8>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
        default: launch_rows_nt<VT, CT, 4>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
    }
}

template <typename VT>
int launch_spmv_tlc(const uspmv_dmat *A, const int *tile_ids, long n_tiles, const VT *x, VT *y, hipStream_t st) {
    if (n_tiles == 0) return USPMV_OK;
    const int C = (int)A->C;
    const unsigned grid = (unsigned)n_tiles;
    const size_t lds = (size_t)A->tlc_max_lines * 16 * sizeof(VT);
#define TLC_LAUNCH(CTV, NTV, IDSV)                                                                                    \
    do {                                                                                                              \
        auto kfn = scs_spmv_tlc<VT, CTV, NTV, IDSV>;                                                                 \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(A->tlc_tile_rows), lds, st, (long)A->n_chunks, C, A->chunk_ptrs,        \
                           A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, A->tlc_line_ptr, A->tlc_lines,   \
                           A->tlc_c16_ptrs, A->tlc_col16, (long)A->tlc_x_len, tile_ids, g_tune.xcd_remap, A->n_store);              \
    } while (0)
#define TLC_LAUNCH_C(NTV, IDSV) do { if (C == 32) TLC_LAUNCH(32, NTV, IDSV); else TLC_LAUNCH(0, NTV, IDSV); } while (0)
    if (tile_ids) { if (g_tune.nontemporal) TLC_LAUNCH_C(true, true); else TLC_LAUNCH_C(false, true); }
    else { if (g_tune.nontemporal) TLC_LAUNCH_C(true, false); else TLC_LAUNCH_C(false, false); }
#undef TLC_LAUNCH_C
#undef TLC_LAUNCH
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

template <typename VT>
int launch_spmv_scs(const uspmv_dmat *A, const int *chunk_ids, long n_ids, const VT *x, VT *y, hipStream_t st) {
    const bool ids = chunk_ids != nullptr;
    const long nwc = ids ? n_ids : A->n_chunks;
    if (nwc == 0) return USPMV_OK;
    const int C = (int)A->C;
    const int block = g_tune.block;
    if (!ids && A->tlc && A->tlc_plan_id == 0 && g_tune.tlc && !g_tune.ablate && g_tune.spmv_variant == 0 && ((uintptr_t)x % 16 == 0))
        return launch_spmv_tlc<VT>(A, nullptr, A->tlc_n_tiles, x, y, st);
    if (!ids && C == 32 && g_tune.spmv_variant == 1) {
        const unsigned grid = grid_for(nwc * 64, block);
        if (g_tune.nontemporal)
            hipLaunchKernelGGL((scs_spmv_split2<VT, 4, true>), dim3(grid), dim3(block), 0, st, nwc, A->chunk_ptrs,
                               A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, g_tune.xcd_remap);
        else
            hipLaunchKernelGGL((scs_spmv_split2<VT, 4, false>), dim3(grid), dim3(block), 0, st, nwc, A->chunk_ptrs,
                               A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, g_tune.xcd_remap);
    } else {
        const unsigned grid = grid_for(nwc * C, block);
        if (g_tune.ablate && C == 32 && !ids) {  // measurement-only (results are wrong by construction)
            if (g_tune.ablate == 1)
                hipLaunchKernelGGL((scs_spmv_rows<VT, 32, 8, true, false, 1>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                                   A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
            else
                hipLaunchKernelGGL((scs_spmv_rows<VT, 32, 8, true, false, 2>), dim3(grid), dim3(block), 0, st, nwc, C, A->chunk_ptrs,
                                   A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, chunk_ids, g_tune.xcd_remap, A->n_store);
            HIP_TRY(hipGetLastError());
            return USPMV_OK;
        }
        switch (C) {  // host-side dispatch on C (the reference switches inside the __global__, code/kernels.hpp:735-753)
            case 1: launch_rows_unroll<VT, 1>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 2: launch_rows_unroll<VT, 2>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 4: launch_rows_unroll<VT, 4>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 8: launch_rows_unroll<VT, 8>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 16: launch_rows_unroll<VT, 16>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 32: launch_rows_unroll<VT, 32>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 64: launch_rows_unroll<VT, 64>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            case 128: launch_rows_unroll<VT, 128>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
            default: launch_rows_unroll<VT, 0>(ids, grid, block, st, nwc, C, A, x, y, chunk_ids); break;
        }
    }
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

template <typename VT, int G>
void launch_csr_g(long n_rows, const int *rp, const int *ci, const VT *va, const VT *x, VT *y, hipStream_t st) {
    const unsigned grid = grid_for(n_rows * G, 256);
    if (g_tune.nontemporal)
        hipLaunchKernelGGL((csr_spmv_vector<VT, G, true>), dim3(grid), dim3(256), 0, st, n_rows, rp, ci, va, x, y);
    else
        hipLaunchKernelGGL((csr_spmv_vector<VT, G, false>), dim3(grid), dim3(256), 0, st, n_rows, rp, ci, va, x, y);
}

template <typename VT>
int launch_csr(long n_rows, long nnz_hint, const int *rp, const int *ci, const VT *va, const VT *x, VT *y,
               hipStream_t st) {
    if (n_rows == 0) return USPMV_OK;
    int G = g_tune.csr_lanes;
    if (G <= 0) {
        const double avg = nnz_hint > 0 ? (double)nnz_hint / (double)n_rows : 16.0;
        G = 2;
        while (G < 64 && G < avg / 2) G <<= 1;
    }
    switch (G) {
        case 1: launch_csr_g<VT, 1>(n_rows, rp, ci, va, x, y, st); break;
        case 2: launch_csr_g<VT, 2>(n_rows, rp, ci, va, x, y, st); break;
        case 4: launch_csr_g<VT, 4>(n_rows, rp, ci, va, x, y, st); break;
        case 8: launch_csr_g<VT, 8>(n_rows, rp, ci, va, x, y, st); break;
        case 16: launch_csr_g<VT, 16>(n_rows, rp, ci, va, x, y, st); break;
        case 32: launch_csr_g<VT, 32>(n_rows, rp, ci, va, x, y, st); break;
        default: launch_csr_g<VT, 64>(n_rows, rp, ci, va, x, y, st); break;
    }
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

template <typename VT, int VB>
void launch_spmmv_vb(const uspmv_dmat *A, const VT *X, VT *Y, int b, long ld, int layout, hipStream_t st) {
    const int block = g_tune.block;
    const unsigned grid = grid_for(A->n_chunks * A->C, block);
    const bool nt = g_tune.nontemporal != 0;
#define SPMMV_LAUNCH(RW, NTV)                                                                                     \
    hipLaunchKernelGGL((scs_spmmv_rows<VT, VB, RW, NTV>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks,      \
                       (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, b, ld, \
                       g_tune.xcd_remap)
    if (layout == USPMV_ROWWISE) { if (nt) SPMMV_LAUNCH(true, true); else SPMMV_LAUNCH(true, false); }
    else { if (nt) SPMMV_LAUNCH(false, true); else SPMMV_LAUNCH(false, false); }
#undef SPMMV_LAUNCH
}

template <typename VT, int B, int U>
void launch_spmmv_rowmajor_u(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const int block = g_tune.block;
    const unsigned grid = grid_for(A->n_chunks * A->C, block);
#define RM_LAUNCH(NTV, YC)                                                                                          \
    do {                                                                                                            \
        if (g_tune.spmmv_prefetch)                                                                                  \
            hipLaunchKernelGGL((scs_spmmv_rowmajor<VT, B, U, NTV, YC, true>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, ld, \
                               g_tune.xcd_remap);                                                                   \
        else                                                                                                        \
            hipLaunchKernelGGL((scs_spmmv_rowmajor<VT, B, U, NTV, YC, false>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, ld, \
                               g_tune.xcd_remap);                                                                   \
    } while (0)
    if (g_tune.nontemporal) { if (ycol) RM_LAUNCH(true, true); else RM_LAUNCH(true, false); }
    else { if (ycol) RM_LAUNCH(false, true); else RM_LAUNCH(false, false); }
#undef RM_LAUNCH
}

template <typename VT, int B, int U>
void launch_spmmv_xpose_u(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const int block = g_tune.block;
    const unsigned grid = grid_for(A->n_chunks * A->C, block);
#define XP_LAUNCH(NTV, YC)                                                                                          \
    do {                                                                                                            \
        if (g_tune.spmmv_prefetch)                                                                                  \
            hipLaunchKernelGGL((scs_spmmv_xpose<VT, B, U, NTV, YC, true>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, ld, \
                               g_tune.xcd_remap);                                                                   \
        else                                                                                                        \
            hipLaunchKernelGGL((scs_spmmv_xpose<VT, B, U, NTV, YC, false>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, ld, \
                               g_tune.xcd_remap);                                                                   \
    } while (0)
    if (g_tune.nontemporal) { if (ycol) XP_LAUNCH(true, true); else XP_LAUNCH(true, false); }
    else { if (ycol) XP_LAUNCH(false, true); else XP_LAUNCH(false, false); }
#undef XP_LAUNCH
}

template <typename VT, int B, int G, int CT, int HS>
void launch_spmmv_tlc_g(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const size_t x_bytes = (size_t)A->bt_max_rows * B * sizeof(VT);
    const size_t lds = x_bytes + (((size_t)A->bt_max_rows * 4 + 15) & ~(size_t)15);   // X rows + the tile's row list
#define BT_LAUNCH(NTV, YC)                                                                                              \
    do {                                                                                                                \
        auto kfn = scs_spmmv_tlc<VT, B, NTV, YC, G, CT, HS>;                                                                \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->bt_n_tiles), dim3(64), lds, st, (long)A->n_chunks,                    \
                           A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, ld, A->bt_line_ptr, \
                           A->bt_xrows, A->bt_c16_ptrs, A->bt_col16, ld, g_tune.xcd_remap, (long)A->n_store, (int)x_bytes); \
    } while (0)
    if (g_tune.nontemporal) { if (ycol) BT_LAUNCH(true, true); else BT_LAUNCH(true, false); }
    else { if (ycol) BT_LAUNCH(false, true); else BT_LAUNCH(false, false); }
#undef BT_LAUNCH
}

template <typename VT, int B>
void launch_spmmv_rowmajor(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    constexpr int RB = B * (int)sizeof(VT);          // bytes per X row
    if constexpr (RB >= 16 && RB <= 128) {
        // block plan staged in LDS, if the handle carries one whose tiles fit this row width
        // auto takes the plan for rows of <= 32 bytes only: there 4+ tiles fit a CU and the kernel is 12-15 % ahead of
        // the gather form; with 64-byte rows (2-3 tiles per CU) each tile's chain of dependent fetches is exposed and
        // it is 20 % behind (profiles/r01/spmmv_probe13.txt).  Variant 4 forces it.
        if (A->bt && ((g_tune.spmmv_variant == 0 && RB <= 32) || g_tune.spmmv_variant == 4) && (size_t)A->bt_max_rows * RB <= BT_LDS_CAP) {
            if (A->bt_tile_rows == 32) {
                if constexpr (RB >= 64) { launch_spmmv_tlc_g<VT, B, 4, 32, 2>(A, X, Y, ld, ycol, st); return; }
            } else {
                if (A->C == 32) launch_spmmv_tlc_g<VT, B, 4, 32, 1>(A, X, Y, ld, ycol, st);
                else launch_spmmv_tlc_g<VT, B, 4, 64, 1>(A, X, Y, ld, ycol, st);
                return;
            }
        }
    }
    if constexpr (RB >= 32) {                        // at least two 16-byte pieces per X row
        if (g_tune.spmmv_variant == 2) {             // transposing X phase: 2-7 % over the plain lane-per-row loop,
            int Up = g_tune.spmmv_unroll ? g_tune.spmmv_unroll : 4;         // level with its prefetching form (spmmv_probe7.txt)
            if (g_tune.spmmv_prefetch && RB >= 64 && Up > 2) Up = 2;        // 4 prefetching slots of 64-byte rows spill
            if (Up >= 4) launch_spmmv_xpose_u<VT, B, 4>(A, X, Y, ld, ycol, st);
            else if (Up >= 2) launch_spmmv_xpose_u<VT, B, 2>(A, X, Y, ld, ycol, st);
            else launch_spmmv_xpose_u<VT, B, 1>(A, X, Y, ld, ycol, st);
            return;
        }
    }
    // auto: 256 bytes of X rows per lane and batch (more spills the prefetching form at 128 VGPRs)
    constexpr int UMAX = RB >= 128 ? 2 : RB >= 64 ? 4 : 8;
    int U = g_tune.spmmv_unroll ? g_tune.spmmv_unroll : UMAX;
    if (U > UMAX) U = UMAX;
    if constexpr (UMAX >= 8) { if (U >= 8) { launch_spmmv_rowmajor_u<VT, B, 8>(A, X, Y, ld, ycol, st); return; } }
    if constexpr (UMAX >= 4) { if (U >= 4) { launch_spmmv_rowmajor_u<VT, B, 4>(A, X, Y, ld, ycol, st); return; } }
    if (U >= 2) launch_spmmv_rowmajor_u<VT, B, 2>(A, X, Y, ld, ycol, st);
    else launch_spmmv_rowmajor_u<VT, B, 1>(A, X, Y, ld, ycol, st);
}

// B-specialised path: row-major kernel; column-major callers get X re-laid out once into the handle's
// scratch and Y written column-major directly by the kernel.
template <typename VT, int B>
int spmmv_fast(const uspmv_dmat *A, const VT *X, VT *Y, long ld, int layout, hipStream_t st) {
    if (layout == USPMV_ROWWISE) {
        launch_spmmv_rowmajor<VT, B>(A, X, Y, ld, false, st);
        return USPMV_OK;
    }
    const size_t need = sizeof(VT) * (size_t)B * (size_t)ld;
    if (A->ws_bytes < need) {
        if (A->ws) (void)hipFree(A->ws);
        A->ws = nullptr; A->ws_bytes = 0;
        hipError_t e = hipMalloc(&A->ws, need);
        if (e != hipSuccess) return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_spmmv: workspace of %zu bytes: %s", need, hipGetErrorString(e));
        A->ws_bytes = need;
    }
    VT *Xr = (VT *)A->ws;
    hipLaunchKernelGGL((block_vector_relayout<VT, B, true>), dim3(grid_for(ld, 256)), dim3(256), 0, st, X, Xr, ld, ld);
    launch_spmmv_rowmajor<VT, B>(A, Xr, Y, ld, true, st);
    return USPMV_OK;
}

template <typename VT>
int launch_spmmv(const uspmv_dmat *A, const VT *X, VT *Y, int b, long ld, int layout, hipStream_t st) {
    if (A->n_chunks == 0) return USPMV_OK;
    int rc = -1;
    if (g_tune.spmmv_variant != 1 && ((uintptr_t)X % 16 == 0) && ((uintptr_t)Y % 16 == 0)) {
        constexpr int VW = 16 / (int)sizeof(VT);
        switch (b) {
            case 2: if (VW <= 2) rc = spmmv_fast<VT, 2>(A, X, Y, ld, layout, st); break;
            case 4: rc = spmmv_fast<VT, 4>(A, X, Y, ld, layout, st); break;
            case 8: rc = spmmv_fast<VT, 8>(A, X, Y, ld, layout, st); break;
            case 16: rc = spmmv_fast<VT, 16>(A, X, Y, ld, layout, st); break;
            default: break;
        }
    }
    if (rc > 0) return rc;
    if (rc < 0) {  // generic width / layout
        if (b <= 1) launch_spmmv_vb<VT, 1>(A, X, Y, b, ld, layout, st);
        else if (b <= 2) launch_spmmv_vb<VT, 2>(A, X, Y, b, ld, layout, st);
        else if (b <= 4) launch_spmmv_vb<VT, 4>(A, X, Y, b, ld, layout, st);
        else launch_spmmv_vb<VT, 8>(A, X, Y, b, ld, layout, st);
    }
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int check_dmat(const uspmv_dmat *A, const char *who) {
    if (!A) return uspmv::fail(USPMV_ERR_INVALID, "%s: NULL matrix", who);
    if (A->C < 1 || A->n_chunks < 0) return uspmv::fail(USPMV_ERR_INVALID, "%s: corrupt matrix handle", who);
    if (A->n_chunks * A->C > INT32_MAX) return uspmv::fail(USPMV_ERR_OVERFLOW, "%s: padded rows exceed int32", who);
    return USPMV_OK;
}

}  // namespace

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
        g_tune.xcd_remap = value;
    } else if (!strcmp(key, "ablate")) g_tune.ablate = value;
    else if (!strcmp(key, "spmmv_prefetch")) g_tune.spmmv_prefetch = value != 0;
    else if (!strcmp(key, "spmmv_tile_rows")) g_tune.spmmv_tile_rows = value == 64 ? 64 : 0;
    else if (!strcmp(key, "spmmv_lds_kb")) g_tune.spmmv_lds_kb = value < 0 ? 0 : value;
    else if (!strcmp(key, "spmmv_variant")) {
        if (value < 0 || value > 4) return uspmv::fail(USPMV_ERR_INVALID, "spmmv_variant must be 0|1|2|3|4");
        g_tune.spmmv_variant = value;
    }
    else if (!strcmp(key, "tail_batch")) g_tune.tail_batch = value != 0;
    else if (!strcmp(key, "spmmv_unroll")) g_tune.spmmv_unroll = value;
    else if (!strcmp(key, "tlc")) g_tune.tlc = value != 0;
    else if (!strcmp(key, "rechunk")) g_tune.rechunk = value != 0;
    else if (!strcmp(key, "tlc_tile_rows")) {
        if (value != 256 && value != 512 && value != 1024) return uspmv::fail(USPMV_ERR_INVALID, "tlc_tile_rows must be 256|512|1024");
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
    else if (!strcmp(key, "xcd_remap")) *value = g_tune.xcd_remap;
    else if (!strcmp(key, "block")) *value = g_tune.block;
    else if (!strcmp(key, "spmv_variant")) *value = g_tune.spmv_variant;
    else if (!strcmp(key, "csr_lanes")) *value = g_tune.csr_lanes;
    else if (!strcmp(key, "ablate")) *value = g_tune.ablate;
    else if (!strcmp(key, "spmmv_prefetch")) *value = g_tune.spmmv_prefetch;
    else if (!strcmp(key, "spmmv_tile_rows")) *value = g_tune.spmmv_tile_rows;
    else if (!strcmp(key, "spmmv_lds_kb")) *value = g_tune.spmmv_lds_kb;
    else if (!strcmp(key, "spmmv_variant")) *value = g_tune.spmmv_variant;
    else if (!strcmp(key, "tail_batch")) *value = g_tune.tail_batch;
    else if (!strcmp(key, "spmmv_unroll")) *value = g_tune.spmmv_unroll;
    else if (!strcmp(key, "tlc")) *value = g_tune.tlc;
    else if (!strcmp(key, "rechunk")) *value = g_tune.rechunk;
    else if (!strcmp(key, "tlc_tile_rows")) *value = g_tune.tlc_tile_rows;
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

static void tlc_release(uspmv_dmat_t *A) {
    (void)hipFree(A->tlc_line_ptr); (void)hipFree(A->tlc_lines); (void)hipFree(A->tlc_c16_ptrs); (void)hipFree(A->tlc_col16);
    A->tlc_line_ptr = A->tlc_lines = nullptr; A->tlc_c16_ptrs = nullptr; A->tlc_col16 = nullptr;
    A->tlc = false; A->tlc_plan_id = 0;
}

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
            g_tune.rechunk = 0;                             // (no recursion)
            rc = uspmv_dmat_optimize(alt, &r, max_lines, n_tiles, n_staged);
            g_tune.rechunk = 1;
            if (rc) { uspmv_dmat_free(alt); return rc; }
            A->alt = alt;
            return USPMV_OK;
        }
    }
    if (max_lines <= 0) max_lines = 512;                       // 64 KiB of doubles: 2 workgroups per CU at worst
    const int cap = (int)(160 * 1024 / (16 * (s->dtype == USPMV_F64 ? 8 : 4)));
    if (max_lines > cap) max_lines = cap;
    uspmv_tlc_plan p;
    if (int rc = uspmv_build_tlc_plan(s, nullptr, max_lines, g_tune.tlc_tile_rows, &p)) return rc;
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_staged) *n_staged = p.valid ? p.n_staged_tiles : 0;
    if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] tlc plan: tile_rows=%d tiles=%lld staged=%lld max_lines=%d lines_total=%zu col16=%zu\n",
                                         p.tile_rows, (long long)p.n_tiles, (long long)p.n_staged_tiles, p.max_lines_used, p.tile_lines.size(), p.col16.size());
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
    if (e != hipSuccess) {
        tlc_release(A);
        return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize: device copy failed: %s", hipGetErrorString(e));
    }
    A->tlc = true; A->tlc_tile_rows = p.tile_rows; A->tlc_max_lines = p.max_lines_used; A->tlc_x_len = p.x_len_min; A->tlc_n_tiles = p.n_tiles;
    A->tlc_staged = p.n_staged_tiles;
    return USPMV_OK;
}

static void bt_release(uspmv_dmat_t *A) {
    (void)hipFree(A->bt_line_ptr); (void)hipFree(A->bt_xrows); (void)hipFree(A->bt_c16_ptrs); (void)hipFree(A->bt_col16);
    A->bt_line_ptr = A->bt_xrows = nullptr; A->bt_c16_ptrs = nullptr; A->bt_col16 = nullptr;
    A->bt = false;
}

int uspmv_dmat_optimize_block(uspmv_dmat_t *A, const uspmv_scs_t *s, int block_vec_size, int64_t *n_tiles, int64_t *n_staged) {
    if (!A || !s) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: NULL argument");
    if (!uspmv::scs_has_entries(s)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: layout-only struct; the plan builder needs the host column indices");
    if (A->C != s->C || A->n_chunks != s->n_chunks || A->dtype != s->dtype)
        return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: handle and host struct do not describe the same matrix");
    if (block_vec_size < 1) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_optimize_block: block_vec_size must be >= 1");
    if (int rc = require_device()) return rc;
    if (A->bt) bt_release(A);
    if (n_tiles) *n_tiles = 0;
    if (n_staged) *n_staged = 0;
    const size_t row_bytes = (size_t)block_vec_size * (s->dtype == USPMV_F64 ? 8 : 4);
    // only the 16-byte-piece kernels (b*sizeof(VT) in {16,32,64,128}) read the plan, compiled for C = 32 and 64
    if (row_bytes % 16 != 0 || (row_bytes & (row_bytes - 1)) != 0 || row_bytes > 128 || (s->C != 32 && s->C != 64)) return USPMV_OK;
    const size_t cap = g_tune.spmmv_lds_kb > 0 ? std::min<size_t>((size_t)g_tune.spmmv_lds_kb * 1024, BT_LDS_CAP) : BT_LDS_CAP;
    const int max_rows = (int)(cap / row_bytes);
    // rows of >= 64 bytes on C = 32: 32-row tiles, two lanes per row (half the LDS per tile, twice the tiles per CU)
    const int tile_rows = (s->C == 32 && row_bytes >= 64 && g_tune.spmmv_tile_rows != 64) ? 32 : 64;
    uspmv_tlc_plan p;
    if (int rc = uspmv_build_tlc_plan(s, nullptr, max_rows, tile_rows, &p, /*line_shift=*/0)) return rc;
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_staged) *n_staged = p.valid ? p.n_staged_tiles : 0;
    if (getenv("USPMV_VERBOSE")) fprintf(stderr, "[uspmv] block plan: b=%d tile_rows=%d tiles=%lld staged=%lld max_rows=%d (cap %d) rows_total=%zu\n",
                                         block_vec_size, p.tile_rows, (long long)p.n_tiles, (long long)p.n_staged_tiles, p.max_lines_used, max_rows, p.tile_lines.size());
    if (!p.valid) return USPMV_OK;
    auto up = [&](const void *h, size_t bytes, void **d) -> hipError_t {
        hipError_t e = hipMalloc(d, bytes ? bytes : 4);
        if (e == hipSuccess && bytes) e = hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(p.tile_line_ptr.data(), p.tile_line_ptr.size() * 4, (void **)&A->bt_line_ptr);
    if (e == hipSuccess) e = up(p.tile_lines.data(), p.tile_lines.size() * 4, (void **)&A->bt_xrows);
    if (e == hipSuccess) e = up(p.c16_ptrs.data(), p.c16_ptrs.size() * 4, (void **)&A->bt_c16_ptrs);
    if (e == hipSuccess) e = up(p.col16.data(), p.col16.size() * 2, (void **)&A->bt_col16);
    if (e != hipSuccess) {
        bt_release(A);
        return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_dmat_optimize_block: device copy failed: %s", hipGetErrorString(e));
    }
    A->bt = true; A->bt_tile_rows = p.tile_rows; A->bt_max_rows = p.max_lines_used; A->bt_n_tiles = p.n_tiles; A->bt_staged = p.n_staged_tiles;
    return USPMV_OK;
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
    if (int rc = uspmv_build_tlc_plan(s_dp, s_sp, max_lines, g_tune.tlc_tile_rows, &p)) return rc;
    if (n_tiles) *n_tiles = p.n_tiles;
    if (n_staged) *n_staged = p.valid ? p.n_staged_tiles : 0;
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

void uspmv_dmat_free(uspmv_dmat_t *A) {
    if (!A) return;
    if (A->alt) { uspmv_dmat_free(A->alt); A->alt = nullptr; }
    if (A->tlc) tlc_release(A);
    if (A->bt) bt_release(A);
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

int uspmv_dmat_tile_rows(const uspmv_dmat_t *A, int *tile_rows) {
    if (!A || !tile_rows) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_dmat_tile_rows: NULL argument");
    *tile_rows = A->tlc ? A->tlc_tile_rows : 0;
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
    if (A->dtype == USPMV_F64) return launch_spmmv<double>(A, (const double *)d_X, (double *)d_Y, b, (long)ld, layout, (hipStream_t)stream);
    return launch_spmmv<float>(A, (const float *)d_X, (float *)d_Y, b, (long)ld, layout, (hipStream_t)stream);
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
    if (!d_x_sp && dp->tlc && sp->tlc && dp->tlc_plan_id != 0 && dp->tlc_plan_id == sp->tlc_plan_id && g_tune.tlc &&
        ((uintptr_t)d_x % 16 == 0)) {
        const size_t lds = (size_t)dp->tlc_max_lines * 16 * sizeof(double);
        const int C = (int)dp->C;
#define APT_LAUNCH(CTV, NTV)                                                                                          \
    do {                                                                                                              \
        auto kfn = scs_spmv_ap_tlc<CTV, NTV>;                                                                        \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)dp->tlc_n_tiles), dim3(dp->tlc_tile_rows), lds, (hipStream_t)stream,     \
                           (long)dp->n_chunks, C, dp->chunk_ptrs, dp->chunk_lengths, dp->col_idxs, (const double *)dp->values, \
                           sp->chunk_ptrs, sp->chunk_lengths, sp->col_idxs, (const float *)sp->values, d_x, d_y,         \
                           dp->tlc_line_ptr, dp->tlc_lines, dp->tlc_c16_ptrs, dp->tlc_col16, sp->tlc_c16_ptrs,           \
                           sp->tlc_col16, (long)dp->tlc_x_len, g_tune.xcd_remap);                                        \
    } while (0)
        if (g_tune.nontemporal) { if (C == 32) APT_LAUNCH(32, true); else APT_LAUNCH(0, true); }
        else { if (C == 32) APT_LAUNCH(32, false); else APT_LAUNCH(0, false); }
#undef APT_LAUNCH
        HIP_TRY(hipGetLastError());
        return USPMV_OK;
    }
    const int block = g_tune.block;
    const unsigned grid = grid_for(dp->n_chunks * dp->C, block);
#define AP_LAUNCH_U(UU, NTV, SPXV)                                                                                   \
    hipLaunchKernelGGL((scs_spmv_ap_rows<UU, NTV, SPXV>), dim3(grid), dim3(block), 0, (hipStream_t)stream,           \
                       (long)dp->n_chunks, (int)dp->C, dp->chunk_ptrs, dp->chunk_lengths, dp->col_idxs,              \
                       (const double *)dp->values, sp->chunk_ptrs, sp->chunk_lengths, sp->col_idxs,                  \
                       (const float *)sp->values, d_x, d_x_sp, d_y, g_tune.xcd_remap)
#define AP_LAUNCH(NTV, SPXV)                                                                                         \
    do { if (g_tune.unroll >= 8) AP_LAUNCH_U(8, NTV, SPXV); else if (g_tune.unroll == 4) AP_LAUNCH_U(4, NTV, SPXV);  \
         else AP_LAUNCH_U(2, NTV, SPXV); } while (0)
    if (d_x_sp) { if (g_tune.nontemporal) AP_LAUNCH(true, true); else AP_LAUNCH(false, true); }
    else { if (g_tune.nontemporal) AP_LAUNCH(true, false); else AP_LAUNCH(false, false); }
#undef AP_LAUNCH_U
#undef AP_LAUNCH
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_spmv_ap(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, double *d_y, void *stream) {
    return spmv_ap_impl(dp, sp, d_x, nullptr, d_y, stream, "uspmv_spmv_ap");
}

int uspmv_spmv_ap_generic(const uspmv_dmat_t *dp, const uspmv_dmat_t *sp, const double *d_x, const float *d_x_sp,
                          double *d_y, void *stream) {
    if (!d_x_sp) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_spmv_ap_generic: NULL float x");
    return spmv_ap_impl(dp, sp, d_x, d_x_sp, d_y, stream, "uspmv_spmv_ap_generic");
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
    hipLaunchKernelGGL(stream_copy_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (double2 *)a, (const double2 *)b, (long)(n / 2));
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int uspmv_stream_triad(double *a, const double *b, const double *c, double s, int64_t n, void *stream) {
    if (!a || !b || !c || n < 0 || (n & 1)) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_stream_triad: bad argument (n must be even)");
    if (int rc = require_device()) return rc;
    hipLaunchKernelGGL(stream_triad_kernel, dim3(256 * 8), dim3(256), 0, (hipStream_t)stream, (double2 *)a, (const double2 *)b, (const double2 *)c, s, (long)(n / 2));
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

int uspmv_time_launches(int what, int reps, const uspmv_dmat_t *A, const uspmv_dmat_t *B, const void *d_x, void *d_y,
                        int64_t n, int b, int64_t ld, int layout, void *stream, double *avg_ms) {
    if (reps < 1 || !avg_ms) return uspmv::fail(USPMV_ERR_INVALID, "uspmv_time_launches: bad argument");
    if (int rc = require_device()) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    hipStream_t st = (hipStream_t)stream;
    int rc = USPMV_OK;
    HIP_TRY(hipEventRecord(e0, st));
    for (int r = 0; r < reps && rc == USPMV_OK; ++r) {
        switch (what) {
            case 0: rc = uspmv_spmv(A, d_x, d_y, stream); break;
            case 1: rc = uspmv_stream_copy((double *)d_y, (const double *)d_x, n, stream); break;
            case 2: rc = uspmv_stream_triad((double *)d_y, (const double *)d_x, (const double *)d_x + n, 3.0, n, stream); break;
            case 3: rc = uspmv_stream_read((const double *)d_x, n, (double *)d_y, stream); break;
            case 4: rc = uspmv_spmv_ap(A, B, (const double *)d_x, (double *)d_y, stream); break;
            case 5: rc = uspmv_spmmv(A, d_x, d_y, b, ld, layout, stream); break;
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
