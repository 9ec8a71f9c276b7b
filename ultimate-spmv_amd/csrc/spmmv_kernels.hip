// SpMMV kernels (block vectors) and their dispatch: reference twin block_spmv_omp_scs_general
// (code/kernels.hpp:306-398) and block_spmv_omp_csr (:68-154).  See uspmv_device.hpp / DESIGN.md 5.
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

// SELL-C-sigma SpMMV (block of b vectors), one lane per row, VB vectors per pass held in registers.
// colwise: X[col + v*ld], Y[row + v*ld];  rowwise: X[col*b + v], Y[row*b + v].
template <typename VT, int VB, bool ROWWISE, bool NT>
__global__ void scs_spmmv_rows(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
                               const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                               const VT *__restrict__ values, const VT *__restrict__ X, VT *__restrict__ Y,
                               const int b, const long ld, const int xcd_remap, const long n_store) {
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long row = (long)lb * blockDim.x + threadIdx.x;
    const long c = row / C;
    const int i = (int)(row - c * C);
    if (c >= n_chunks) return;
    const long cs = chunk_ptrs[c];
    const int L = chunk_lengths[c];
    if (L < 0) return;                        // (USPMV_SKIP_LEN: the chunk belongs to the other part of a two-part SpMMV)
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
            if (v0 + v < b && row < n_store) {
                if (ROWWISE) st_y<NT>(Y + (row * b + v0 + v), acc[v]);
                else st_y<NT>(Y + (row + (long)(v0 + v) * ld), acc[v]);
            }
        }
    }
}

// SpMMV with ROW-MAJOR block vectors of compile-time width B (X[col*B + v]): one lane per row, B
// accumulators per lane.  Per slot a lane reads its whole X row -- B*sizeof(VT) contiguous bytes --
// with 16-byte loads, so one wave-instruction moves 1 KiB of X instead of 512 B of eight-byte
// column gathers: 1 + 1 + B*sizeof(VT)/16 vector-memory instructions per 64 non-zeros (the
// column-major form needs 2 + B, each fetching a whole 64-byte sector per lane for 8 useful bytes).
// Every (row, v) accumulator is still the slot-ordered FMA chain of block_spmv_omp_scs_general.
// YCOL: write Y column-major (Y[row + v*ld]) straight from the accumulators -- per vector one
// coalesced 64-lane store -- so that column-major callers only pay the X re-layout.
template <typename VT, int B, int U, bool NT, bool YCOL, bool PF>
__global__ void __launch_bounds__(256) scs_spmmv_rowmajor(const long n_chunks, const int C, const int *__restrict__ chunk_ptrs,
                                   const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs,
                                   const VT *__restrict__ values, const VT *__restrict__ X, VT *__restrict__ Y,
                                   const long ld, const int xcd_remap, const long n_store, const int *__restrict__ row_map = nullptr) {
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
    if (L < 0) return;                        // (USPMV_SKIP_LEN: the chunk belongs to the other part of a two-part SpMMV)
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
    const long yrow = row_map ? (long)row_map[row] : row;   // (tie-reordered private copy: plan row -> caller's row)
    if (yrow >= n_store) return;              // (re-chunked handles: rows past the caller's padded rows)
    if (YCOL) {
#pragma unroll
        for (int v = 0; v < B; ++v) st_y<NT>(Y + (yrow + (long)v * ld), acc[v]);
    } else {
        vec_t *yp = (vec_t *)(Y + yrow * B);
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
                                const long ld, const int xcd_remap, const long n_store) {
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
        if (r < n_pad && r < n_store) {
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

template <typename VT, int B, bool NT, bool YCOL, int G, int C, int HS, bool SWZ>
__global__ void __launch_bounds__(64) scs_spmmv_tlc(const long n_chunks, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        const VT *__restrict__ X, VT *__restrict__ Y, const long ld, const int *__restrict__ tile_line_ptr,
        const int *__restrict__ tile_xrows, const unsigned *__restrict__ c16_ptrs,
        const unsigned short *__restrict__ col16, const long x_rows, const int xcd_remap, const long n_store, const int x_bytes,
        const int *__restrict__ row_map) {
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
        fma_row(a, xs + local * NV, (SWZ && NV > 1) ? (local >> SWS) & (NV - 1) : 0u);
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
                    const unsigned piece = ((unsigned)p & (NV - 1)) ^ ((SWZ && NV > 1) ? (k >> SWS) & (NV - 1) : 0u);
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
    if (!valid) return;
    const long yrow = row_map ? (long)row_map[row] : row;   // (plan rows are a permutation of the caller's rows)
    if (yrow >= n_store) return;
    if (YCOL) {
#pragma unroll
        for (int v = 0; v < BH; ++v) st_y<NT>(Y + (yrow + (long)(h * BH + v) * ld), acc[v]);
    } else {
        vec_t *yp = (vec_t *)(Y + yrow * B) + h * NVH;
#pragma unroll
        for (int k = 0; k < NVH; ++k) {
            vec_t t;
#pragma unroll
            for (int w = 0; w < VW; ++w) t[w] = acc[k * VW + w];
            yp[k] = t;
        }
    }
}


// Row-major SpMMV over the block plan, FOUR LANES PER ROW (rows of four 16-byte pieces: b = 8 in double, 16 in single
// precision).  The single-wave form above keeps a whole 64-byte X row per lane: B accumulators, deep register batches
// (it compiles to 486 registers, 700 accvgpr moves per tile) and ONE wave per 50 KB of LDS -- three waves per CU whose
// compute phase cannot overlap anything.  Here a 64-row tile is a 256-thread workgroup: wave w owns rows 16w..16w+15, lane
// (r, q) owns piece q (two doubles) of row r's accumulators.  The matrix stream stays coalesced -- lane (r, q) loads slot
// 4g+q of row r, one 8-byte value and one 2-byte local index -- and a slot's (value, index) reaches the row's four lanes by
// a DPP quad broadcast (no LDS traffic); the X operand is one ds_read_b128 per lane and slot.  Per (row, column) the FMA
// chain still runs over the slots in order: bit-identical to block_spmv_omp_scs_general (code/kernels.hpp:306-398).
// Four waves share one staged copy of the tile's X rows, ~25 registers each, so a CU holds 12 waves on 3 tiles and the
// arithmetic of a tile takes a quarter of the time.
template <typename VT, int B, bool NT, bool YCOL, int C, bool SWZ, int NG, int MAXP>
__global__ void __launch_bounds__(256) scs_spmmv_quad(const long n_chunks, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        const VT *__restrict__ X, VT *__restrict__ Y, const long ld, const int *__restrict__ tile_line_ptr,
        const int *__restrict__ tile_xrows, const unsigned *__restrict__ c16_ptrs, const unsigned short *__restrict__ col16,
        const int xcd_remap, const long n_store, const int x_bytes, const int *__restrict__ row_map, const int ablate) {
    // ablate (measurement only, WRONG results): 1 = no X staging, 2 = no arithmetic, 3 = no row list and no X staging
    // NG: full groups (of four slots) a lane keeps in registers per pass.  A wave's share of a tile is 16 rows x L slots =
    // L/4 (value, index) pairs per lane; with NG = 20 the whole share of an 81-slot stencil tile is requested in ONE burst
    // before the X rows are staged (12 waves x 21 x 640 B = 160 KB in flight per CU) and the arithmetic never waits for HBM.
    extern __shared__ __attribute__((aligned(16))) unsigned char tlc_smem[];
    constexpr int VW = 16 / (int)sizeof(VT);
    static_assert(B == 4 * VW, "four 16-byte pieces per X row");
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    const unsigned tile = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int lp0 = tile_line_ptr[tile];
    const int nl = tile_line_ptr[tile + 1] - lp0;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane >> 2, q = lane & 3;
    const long row = (long)tile * 64 + wave * 16 + r;       // plan row (16 | C: a wave's rows sit in one chunk)
    const long c = row / C;
    const int i = (int)(row - c * C);
    const bool valid = c < n_chunks;
    int cs = 0, L = 0;
    unsigned q0 = 0;
    if (valid) { cs = chunk_ptrs[c]; L = chunk_lengths[c]; q0 = c16_ptrs[c]; }
    L = __builtin_amdgcn_readfirstlane(L);                   // (wave-uniform; an invalid wave has L = 0)
    vec_t acc;
#pragma unroll
    for (int w = 0; w < VW; ++w) acc[w] = VT(0);
    const int ngf = L >> 2, rem = L & 3;                     // full groups of four slots, slots of the last partial group
    const VT *vp = values + (long)cs + i + (long)q * C;     // slot 4g+q of this lane's row: + g*4*C
#define QUAD_STEP(UU, AV, IV, XOF)                                                                            \
    {                                                                                                         \
        const VT aa = quad_bcast<UU>(AV);                                                                     \
        const unsigned li = (unsigned)quad_bcast<UU>((int)(IV));                                              \
        const vec_t xv = XOF(li);                                                                             \
        _Pragma("unroll") for (int w = 0; w < VW; ++w) acc[w] = fma_t(aa, xv[w], acc[w]);                     \
    }
    if (nl > 0) {
        const unsigned short *ip = col16 + q0 + (long)i * 4 + q;   // + g*4*C
        // ---- 1. the first pass of matrix entries is requested before anything else (it does not depend on the X rows)
        VT a[NG], at = VT(0);
        unsigned ix[NG], ixt = 0u;
        auto load_pass = [&](const int gb) {
#pragma unroll
            for (int d = 0; d < NG; ++d) {
                a[d] = VT(0); ix[d] = 0u;
                if (gb + d < ngf) {
                    if (ablate != 4) ix[d] = ld_stream<NT>(ip + (long)(gb + d) * 4 * C);      // (ablate 4 / 5: no index / no value loads)
                    if (ablate != 5) a[d] = ld_stream<NT>(vp + (long)(gb + d) * 4 * C);
                }
            }
        };
        // ---- 1. the X-row list entries this lane needs for its DMA pieces, straight into registers (piece p = (wave + 4k)*64 + lane
        //         <-> list entry p >> 2): the one round trip nothing else can overlap
        const int np = (ablate == 1 || ablate == 3) ? 0 : nl << 2;
        int xr[MAXP];
#pragma unroll
        for (int k = 0; k < MAXP; ++k) {
            const int p = (wave + 4 * k) * 64 + lane;
            xr[k] = -1;
            if (p < np) xr[k] = tile_xrows[lp0 + (p >> 2)];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // ---- 2. X rows -> LDS by DMA (64 pieces of 16 bytes per wave-instruction) and, behind them, the wave's whole share of
        //         matrix entries: both bursts are in flight together
#pragma unroll
        for (int k = 0; k < MAXP; ++k)
            if (xr[k] >= 0) {
                const unsigned kk = (unsigned)((wave + 4 * k) * 64 + lane) >> 2;
                const unsigned piece = (unsigned)q ^ (SWZ ? (kk >> 2) & 3u : 0u);
                __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + (long)xr[k] * B + piece * VW), (lds_void_t *)(tlc_smem + (wave + 4 * k) * 1024), 16, 0, 0);
            }
        load_pass(0);
        if (rem) { ixt = ld_stream<NT>(ip + (long)ngf * 4 * C); if (q < rem) at = ld_stream<NT>(vp + (long)ngf * 4 * C); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- 4. arithmetic from registers
        const vec_t *xs = (const vec_t *)tlc_smem;
#define X_LDS(li) xs[(li) * 4 + ((unsigned)q ^ (SWZ ? ((li) >> 2) & 3u : 0u))]
        if (ablate != 2) {
            for (int gb = 0;; gb += NG) {
#pragma unroll
                for (int d = 0; d < NG; ++d) {
                    if (gb + d < ngf) {
                        QUAD_STEP(0, a[d], ix[d], X_LDS) QUAD_STEP(1, a[d], ix[d], X_LDS) QUAD_STEP(2, a[d], ix[d], X_LDS) QUAD_STEP(3, a[d], ix[d], X_LDS)
                    }
                }
                if (gb + NG >= ngf) break;
                load_pass(gb + NG);                          // rows longer than 4*NG slots: further passes (these do wait)
            }
            if (rem > 0) QUAD_STEP(0, at, ixt, X_LDS)
            if (rem > 1) QUAD_STEP(1, at, ixt, X_LDS)
            if (rem > 2) QUAD_STEP(2, at, ixt, X_LDS)
        }
#undef X_LDS
    } else if (L > 0) {   // wide-footprint tile: 32-bit columns, X pieces gathered from global memory
        const int *cp = col_idxs + (long)cs + i + (long)q * C;
#define X_GLB(col) (*((const vec_t *)(X + (long)(col) * B) + q))
        for (int g = 0; g < ngf; ++g) {
            const VT av = ld_stream<NT>(vp + (long)g * 4 * C);
            const int cv = ld_stream<NT>(cp + (long)g * 4 * C);
            QUAD_STEP(0, av, cv, X_GLB) QUAD_STEP(1, av, cv, X_GLB) QUAD_STEP(2, av, cv, X_GLB) QUAD_STEP(3, av, cv, X_GLB)
        }
        if (rem) {
            VT av = VT(0);
            int cv = 0;
            if (q < rem) { av = ld_stream<NT>(vp + (long)ngf * 4 * C); cv = ld_stream<NT>(cp + (long)ngf * 4 * C); }
            if (rem > 0) QUAD_STEP(0, av, cv, X_GLB)
            if (rem > 1) QUAD_STEP(1, av, cv, X_GLB)
            if (rem > 2) QUAD_STEP(2, av, cv, X_GLB)
        }
#undef X_GLB
    }
#undef QUAD_STEP
    if (!valid) return;
    const long yrow = row_map ? (long)row_map[row] : row;
    if (yrow >= n_store) return;
    if (YCOL) {
#pragma unroll
        for (int w = 0; w < VW; ++w) st_y<NT>(Y + (yrow + (long)(q * VW + w) * ld), acc[w]);
    } else {
        *((vec_t *)(Y + yrow * B) + q) = acc;
    }
}

// Column-major -> row-major re-layout through LDS: 256 rows per workgroup.  Reads are B coalesced element streams (one per column);
// the rows are assembled in LDS (16 bytes of padding per row against bank conflicts) and written back as 16-byte pieces in linear
// order, so every store instruction of a wave covers 1 KiB of contiguous output (the lane-per-row form above it replaces wrote
// B scalars per lane, B*sizeof(VT) bytes apart: 4.4 TB/s for the 2 x 262 MB of config 3).  Row-major `out` must be 16-byte aligned.
// PERM: out row r = in row perm[r] for r < n_perm (identity beyond): the re-layout pass of a column-major caller undoes the sigma
// permutation on the way (perm = old_to_new_idx), so that the phased kernel's X rows are runs of the workspace (the handle's
// "unscrambled" plan).  The reads of a workgroup stay inside one or two sigma windows per column.
template <typename VT, int B, bool PERM = false>
__global__ void __launch_bounds__(256) block_vector_to_rowmajor(const VT *__restrict__ in, VT *__restrict__ out, const long n, const long ld,
                                                                const int *__restrict__ perm = nullptr, const long n_perm = 0) {
    constexpr int RB = B * (int)sizeof(VT), PPR = RB / 16, STRIDE = RB + 16;     // bytes per row, 16-byte pieces per row, padded LDS row
    __shared__ __attribute__((aligned(16))) unsigned char tile[256 * STRIDE];
    typedef VT vec_t __attribute__((ext_vector_type(16 / (int)sizeof(VT))));
    const long r0 = (long)blockIdx.x * 256, r = r0 + threadIdx.x;
    if (r < n) {
        VT tv[B];
        const long rs = (PERM && r < n_perm) ? (long)perm[r] : r;
#pragma unroll
        for (int v = 0; v < B; ++v) tv[v] = PERM ? in[rs + (long)v * ld] : __builtin_nontemporal_load(in + rs + (long)v * ld);
#pragma unroll
        for (int v = 0; v < B; ++v) *(VT *)(tile + threadIdx.x * STRIDE + v * (int)sizeof(VT)) = tv[v];
    }
    __syncthreads();
    const long n_here = min((long)256, n - r0);
#pragma unroll
    for (int k = 0; k < PPR; ++k) {
        const int j = k * 256 + (int)threadIdx.x;           // piece number inside the workgroup's block of output
        const int row = j / PPR, piece = j % PPR;
        if (row < n_here) {
            const vec_t v = *(const vec_t *)(tile + row * STRIDE + piece * 16);
            __builtin_nontemporal_store(v, (vec_t *)(out + r0 * B) + j);
        }
    }
}

// colwise (b vectors of leading dimension ld) <-> row-major (n rows of B) re-layout, one lane per row (kept for the reverse direction)
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

template <typename VT, int VB>
void launch_spmmv_vb(const uspmv_dmat *A, const VT *X, VT *Y, int b, long ld, int layout, hipStream_t st) {
    const int block = g_tune.block;
    const unsigned grid = grid_for(A->n_chunks * A->C, block);
    const bool nt = g_tune.nontemporal != 0;
#define SPMMV_LAUNCH(RW, NTV)                                                                                     \
    hipLaunchKernelGGL((scs_spmmv_rows<VT, VB, RW, NTV>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks,      \
                       (int)A->C, A->chunk_ptrs, part_lengths(A, 0), A->col_idxs, (const VT *)A->values, X, Y, b, ld, \
                       g_tune.xcd_remap, (long)A->n_store)
    if (layout == USPMV_ROWWISE) { if (nt) SPMMV_LAUNCH(true, true); else SPMMV_LAUNCH(true, false); }
    else { if (nt) SPMMV_LAUNCH(false, true); else SPMMV_LAUNCH(false, false); }
#undef SPMMV_LAUNCH
}

template <typename VT, int B, int U>
void launch_spmmv_rowmajor_u(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const int block = std::min(g_tune.block, 256);      // (__launch_bounds__(256): deep batches may use > 128 VGPRs)
    const unsigned grid = grid_for(A->n_chunks * A->C, block);
    // variant 5: the gather kernel over the block plan's tie-reordered copy of the entries (neighbouring lanes then read
    // neighbouring X rows, which is what L1 can exploit)
    const bool ro = g_tune.spmmv_variant == 5 && A->bt_values && A->bt_cols && A->bt_row_map && !A->part;
    const int *cols = ro ? A->bt_cols : A->col_idxs;
    const VT *vals = (const VT *)(ro ? A->bt_values : A->values);
    const int *rmap = ro ? A->bt_row_map : nullptr;
#define RM_LAUNCH(NTV, YC)                                                                                          \
    do {                                                                                                            \
        if (g_tune.spmmv_prefetch)                                                                                  \
            hipLaunchKernelGGL((scs_spmmv_rowmajor<VT, B, U, NTV, YC, true>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, part_lengths(A, 0), cols, vals, X, Y, ld, \
                               g_tune.xcd_remap, (long)A->n_store, rmap);                                                             \
        else                                                                                                        \
            hipLaunchKernelGGL((scs_spmmv_rowmajor<VT, B, U, NTV, YC, false>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, part_lengths(A, 0), cols, vals, X, Y, ld, \
                               g_tune.xcd_remap, (long)A->n_store, rmap);                                                             \
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
                               g_tune.xcd_remap, (long)A->n_store);                                                                   \
        else                                                                                                        \
            hipLaunchKernelGGL((scs_spmmv_xpose<VT, B, U, NTV, YC, false>), dim3(grid), dim3(block), 0, st, (long)A->n_chunks, \
                               (int)A->C, A->chunk_ptrs, A->chunk_lengths, A->col_idxs, (const VT *)A->values, X, Y, ld, \
                               g_tune.xcd_remap, (long)A->n_store);                                                                   \
    } while (0)
    if (g_tune.nontemporal) { if (ycol) XP_LAUNCH(true, true); else XP_LAUNCH(true, false); }
    else { if (ycol) XP_LAUNCH(false, true); else XP_LAUNCH(false, false); }
#undef XP_LAUNCH
}

template <typename VT, int B, int G, int CT, int HS, bool SWZ>
void launch_spmmv_tlc_g(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const size_t x_bytes = (size_t)A->bt_max_rows * B * sizeof(VT);
    const size_t lds = x_bytes + (((size_t)A->bt_max_rows * 4 + 15) & ~(size_t)15);   // X rows + the tile's row list
#define BT_LAUNCH(NTV, YC)                                                                                              \
    do {                                                                                                                \
        auto kfn = scs_spmmv_tlc<VT, B, NTV, YC, G, CT, HS, SWZ>;                                                                \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->bt_n_tiles), dim3(64), lds, st, (long)A->n_chunks,                    \
                           A->chunk_ptrs, A->chunk_lengths, A->bt_cols ? A->bt_cols : A->col_idxs,                       \
                           (const VT *)(A->bt_values ? A->bt_values : A->values), X, Y, ld, A->bt_line_ptr,                \
                           A->bt_xrows, A->bt_c16_ptrs, A->bt_col16, ld, g_tune.xcd_remap, (long)A->n_store, (int)x_bytes, \
                           (const int *)A->bt_row_map);                                                                   \
    } while (0)
    if (g_tune.nontemporal) { if (ycol) BT_LAUNCH(true, true); else BT_LAUNCH(true, false); }
    else { if (ycol) BT_LAUNCH(false, true); else BT_LAUNCH(false, false); }
#undef BT_LAUNCH
}


template <typename VT, int B, int CT, bool SWZ, int PD, int MAXP>
void launch_spmmv_quad_m(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const size_t x_bytes = (size_t)MAXP * 4 * 1024;          // whole DMA pieces: MAXP per wave, four waves
    const size_t lds = x_bytes;
#define QD_LAUNCH(NTV, YC)                                                                                              \
    do {                                                                                                                \
        auto kfn = scs_spmmv_quad<VT, B, NTV, YC, CT, SWZ, PD, MAXP>;                                                   \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->bt_n_tiles), dim3(256), lds, st, (long)A->n_chunks,                   \
                           A->chunk_ptrs, A->chunk_lengths, A->bt_cols ? A->bt_cols : A->col_idxs,                       \
                           (const VT *)(A->bt_values ? A->bt_values : A->values), X, Y, ld, A->bt_line_ptr,                \
                           A->bt_xrows, A->bt_c16_ptrs, A->bt_col16, g_tune.xcd_remap, (long)A->n_store, (int)x_bytes,    \
                           (const int *)A->bt_row_map, g_tune.ablate);                                                    \
    } while (0)
    if (g_tune.nontemporal) { if (ycol) QD_LAUNCH(true, true); else QD_LAUNCH(true, false); }
    else { if (ycol) QD_LAUNCH(false, true); else QD_LAUNCH(false, false); }
#undef QD_LAUNCH
}

template <typename VT, int B, int CT, bool SWZ, int PD>
void launch_spmmv_quad_g(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const int pieces = (A->bt_max_rows * 4 + 255) / 256;     // DMA pieces per wave for the largest tile (BT_LDS_CAP = 80 KiB: <= 20)
    if (pieces <= 8) launch_spmmv_quad_m<VT, B, CT, SWZ, PD, 8>(A, X, Y, ld, ycol, st);
    else if (pieces <= 13) launch_spmmv_quad_m<VT, B, CT, SWZ, PD, 13>(A, X, Y, ld, ycol, st);
    else launch_spmmv_quad_m<VT, B, CT, SWZ, PD, 20>(A, X, Y, ld, ycol, st);
}

template <typename VT, int B>
void launch_spmmv_quad(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    const bool swz = g_tune.spmmv_swizzle != 0;
#define QD_PD(CTV, SW) launch_spmmv_quad_g<VT, B, CTV, SW, 20>(A, X, Y, ld, ycol, st)   /* 20 groups of four slots per lane and pass */
    if (A->C == 32) { if (swz) QD_PD(32, true); else QD_PD(32, false); }
    else { if (swz) QD_PD(64, true); else QD_PD(64, false); }
#undef QD_PD
}

// the phased-plan kernels live in spmmv_phased.hip (64-byte rows only: dp b = 8, sp b = 16)
template <typename VT, int B>
bool launch_spmmv_quadph(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, int xmode, hipStream_t st) {
    return uspmv_dev::spmmv_phased(A, X, Y, ld, ycol, xmode, st);
}
template <typename VT, int B>
void launch_spmmv_rowmajor(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    constexpr int RB = B * (int)sizeof(VT);          // bytes per X row
    if constexpr (RB == 64) {
        // 64-byte rows, phased plan (variant 8; auto when the handle carries one): eight workgroups per CU
        if ((g_tune.spmmv_variant == 8 || (g_tune.spmmv_variant == 0 && !g_tune.ablate)) && launch_spmmv_quadph<VT, B>(A, X, Y, ld, ycol, 0, st)) return;
        // 64-byte rows: the four-lanes-per-row kernel over 64-row tiles of the block plan (variant 6; auto when the plan is there)
        if (A->bt && !A->part && A->bt_tile_rows == 64 && (g_tune.spmmv_variant == 0 || g_tune.spmmv_variant == 6) && (size_t)A->bt_max_rows * RB <= BT_LDS_CAP) {
            // one tile per workgroup, three workgroups per CU (a persistent, software-pipelined form of it -- 222 registers, two workgroups
            // per CU -- measured 5-7 % slower and was removed: profiles/r02/spmmv_variants.txt)
            launch_spmmv_quad<VT, B>(A, X, Y, ld, ycol, st);
            return;
        }
    }
    if constexpr (RB >= 16 && RB <= 128) {
        // block plan staged in LDS, if the handle carries one whose tiles fit this row width
        // auto takes the plan for rows of <= 32 bytes only: there 4+ tiles fit a CU and the kernel is 12-15 % ahead of
        // the gather form; with 64-byte rows (2-3 tiles per CU) each tile's chain of dependent fetches is exposed and
        // it is 20 % behind (profiles/r01/spmmv_probe13.txt).  Variant 4 forces it.
        if (A->bt && !A->part && ((g_tune.spmmv_variant == 0 && RB <= 32) || g_tune.spmmv_variant == 4) && (size_t)A->bt_max_rows * RB <= BT_LDS_CAP) {
            const bool swz = g_tune.spmmv_swizzle != 0;
            if (A->bt_tile_rows == 32) {
                if constexpr (RB >= 32) {
                    if (swz) launch_spmmv_tlc_g<VT, B, 4, 32, 2, true>(A, X, Y, ld, ycol, st);
                    else launch_spmmv_tlc_g<VT, B, 4, 32, 2, false>(A, X, Y, ld, ycol, st);
                    return;
                }
            } else {
                if (A->C == 32) { if (swz) launch_spmmv_tlc_g<VT, B, 4, 32, 1, true>(A, X, Y, ld, ycol, st); else launch_spmmv_tlc_g<VT, B, 4, 32, 1, false>(A, X, Y, ld, ycol, st); }
                else { if (swz) launch_spmmv_tlc_g<VT, B, 4, 64, 1, true>(A, X, Y, ld, ycol, st); else launch_spmmv_tlc_g<VT, B, 4, 64, 1, false>(A, X, Y, ld, ycol, st); }
                return;
            }
        }
    }
    if constexpr (RB >= 32) {                        // at least two 16-byte pieces per X row
        if (g_tune.spmmv_variant == 2 && !A->part) { // transposing X phase: 2-7 % over the plain lane-per-row loop,
            int Up = g_tune.spmmv_unroll ? g_tune.spmmv_unroll : 4;         // level with its prefetching form (spmmv_probe7.txt)
            if (g_tune.spmmv_prefetch && RB >= 64 && Up > 2) Up = 2;        // 4 prefetching slots of 64-byte rows spill
            if (Up >= 4) launch_spmmv_xpose_u<VT, B, 4>(A, X, Y, ld, ycol, st);
            else if (Up >= 2) launch_spmmv_xpose_u<VT, B, 2>(A, X, Y, ld, ycol, st);
            else launch_spmmv_xpose_u<VT, B, 1>(A, X, Y, ld, ycol, st);
            return;
        }
    }
    // auto: 256 bytes of X rows per lane and batch (more spills the prefetching form at 128 VGPRs)
    constexpr int UMAX = RB >= 128 ? 4 : 8;
    constexpr int UDEF = RB >= 128 ? 2 : RB >= 64 ? 4 : 8;
    int U = g_tune.spmmv_unroll ? g_tune.spmmv_unroll : UDEF;
    if (U > UMAX) U = UMAX;
    if constexpr (UMAX >= 8) { if (U >= 8) { launch_spmmv_rowmajor_u<VT, B, 8>(A, X, Y, ld, ycol, st); return; } }
    if constexpr (UMAX >= 4) { if (U >= 4) { launch_spmmv_rowmajor_u<VT, B, 4>(A, X, Y, ld, ycol, st); return; } }
    if (U >= 2) launch_spmmv_rowmajor_u<VT, B, 2>(A, X, Y, ld, ycol, st);
    else launch_spmmv_rowmajor_u<VT, B, 1>(A, X, Y, ld, ycol, st);
}

// The re-layout pass of a column-major X into the handle's row-major workspace.  *form: 1 = rows in the caller's (sigma-permuted)
// numbering, 2 = the pass also undid the sigma permutation ("spmmv_unscramble": the plan over original X-row numbering runs on it).
template <typename VT, int B>
int relayout_x(const uspmv_dmat *A, const VT *X, long ld, hipStream_t st, int *form, bool plain_only = false) {
    const size_t need = sizeof(VT) * (size_t)B * (size_t)ld;
    if (A->ws_bytes < need) {
        if (A->ws) (void)hipFree(A->ws);
        A->ws = nullptr; A->ws_bytes = 0; A->xprep_ptr = nullptr;
        hipError_t e = hipMalloc(&A->ws, need);
        if (e != hipSuccess) return uspmv::fail(USPMV_ERR_ALLOC, "uspmv_spmmv: workspace of %zu bytes: %s", need, hipGetErrorString(e));
        A->ws_bytes = need;
    }
    VT *Xr = (VT *)A->ws;
    if constexpr (B * (int)sizeof(VT) == 64) {
        // the re-layout pass undoes the sigma permutation, the kernel runs on the plan over original X-row numbering
        if (!plain_only && (g_tune.spmmv_variant == 0 || g_tune.spmmv_variant == 8) && !g_tune.ablate && !A->part && g_tune.spmmv_unscramble && A->pu && A->pu_perm && A->pu_n_perm <= ld) {
            hipLaunchKernelGGL((block_vector_to_rowmajor<VT, B, true>), dim3(grid_for(ld, 256)), dim3(256), 0, st, X, Xr, ld, ld, (const int *)A->pu_perm, (long)A->pu_n_perm);
            *form = 2;
            return USPMV_OK;
        }
    }
    // (two-part SpMMV: the interior part needs the local X rows only, the boundary part brings the halo rows after the exchange)
    const long split = std::min<long>(std::max<long>(A->part_split, 0), ld);
    const long r0 = A->part == 2 ? split : 0, nr = (A->part == 1 ? split : ld) - r0;
    if (nr > 0) {
        if constexpr ((B * (int)sizeof(VT)) % 16 == 0)
            hipLaunchKernelGGL((block_vector_to_rowmajor<VT, B>), dim3(grid_for(nr, 256)), dim3(256), 0, st, X + r0, Xr + r0 * B, nr, ld);
        else
            hipLaunchKernelGGL((block_vector_relayout<VT, B, true>), dim3(grid_for(nr, 256)), dim3(256), 0, st, X + r0, Xr + r0 * B, nr, ld);
    }
    *form = 1;
    return USPMV_OK;
}

// B-specialised path: row-major kernel; column-major callers get X re-laid out once into the handle's
// scratch and Y written column-major directly by the kernel.
template <typename VT, int B>
int spmmv_fast(const uspmv_dmat *A, const VT *X, VT *Y, long ld, int layout, hipStream_t st) {
    if constexpr (B * (int)sizeof(VT) == 64) {
        // the block-vector window sweep, when the handle carries its plan (uspmv_dmat_optimize_block_sweep): both layouts straight from
        // the caller's vectors -- column-major X is staged column by column, no re-layout pass
        if (A->bw && (g_tune.spmmv_variant == 0 || g_tune.spmmv_variant == 9) && !g_tune.ablate) {
            const int rc = launch_spmmv_sweep<VT>(A, X, Y, B, ld, layout == USPMV_COLWISE, layout == USPMV_COLWISE, st);
            if (rc <= 0) return rc;
        }
    }
    if (layout == USPMV_ROWWISE) {
        launch_spmmv_rowmajor<VT, B>(A, X, Y, ld, false, st);
        return USPMV_OK;
    }
    if constexpr (B * (int)sizeof(VT) == 64) {
        // 64-byte rows with a phased plan: the kernel stages X straight from the column-major vector -- by 128-byte lines when the
        // handle carries the line plan (default), through registers with "spmmv_xcol" 1 -- no re-layout pass, no workspace
        if ((g_tune.spmmv_variant == 0 || g_tune.spmmv_variant == 8) && !g_tune.ablate && !A->part) {
            if (g_tune.spmmv_xcol && launch_spmmv_quadph<VT, B>(A, X, Y, ld, true, 1, st)) return USPMV_OK;
            if (g_tune.spmmv_xline && launch_spmmv_quadph<VT, B>(A, X, Y, ld, true, 2, st)) return USPMV_OK;
        }
    }
    // a caller whose X is unchanged since uspmv_spmmv_x_prepared (the reference's bench loop) skips the re-layout pass
    const bool prepared = A->xprep_ptr == (const void *)X && A->xprep_b == B && A->xprep_ld == ld && A->ws && !A->part;
    int form = A->xprep_form;
    if (!prepared) {
        A->xprep_ptr = nullptr;                               // (the workspace is about to hold another X)
        if (int rc = relayout_x<VT, B>(A, X, ld, st, &form)) return rc;
    }
    VT *Xr = (VT *)A->ws;
    if constexpr (B * (int)sizeof(VT) == 64) {
        if (form == 2 && launch_spmmv_quadph<VT, B>(A, Xr, Y, ld, true, 3, st)) return USPMV_OK;
        if (form == 2) {      // (the plan over original X-row numbering turned the launch down: plain re-layout, plain plan)
            if (int rc = relayout_x<VT, B>(A, X, ld, st, &form, true)) return rc;
        }
    }
    launch_spmmv_rowmajor<VT, B>(A, Xr, Y, ld, true, st);
    return USPMV_OK;
}

}  // namespace

namespace uspmv_dev {

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

// uspmv_spmmv_x_prepared: the re-layout now, remembered on the handle.  > 0: this (dtype, b) has no re-layout pass (nothing to prepare)
template <typename VT>
int prepare_x(const uspmv_dmat *A, const VT *X, int b, long ld, hipStream_t st) {
    constexpr int VW = 16 / (int)sizeof(VT);
    if (((uintptr_t)X % 16) != 0 || A->part) return 1;
    int form = 0, rc = 1;
    switch (b) {
        case 2: if (VW <= 2) rc = relayout_x<VT, 2>(A, X, ld, st, &form); break;
        case 4: rc = relayout_x<VT, 4>(A, X, ld, st, &form); break;
        case 8: rc = relayout_x<VT, 8>(A, X, ld, st, &form); break;
        case 16: rc = relayout_x<VT, 16>(A, X, ld, st, &form); break;
        default: break;
    }
    if (rc == USPMV_OK) { A->xprep_ptr = X; A->xprep_b = b; A->xprep_ld = ld; A->xprep_form = form; }
    return rc;
}
template int prepare_x<double>(const uspmv_dmat *, const double *, int, long, hipStream_t);
template int prepare_x<float>(const uspmv_dmat *, const float *, int, long, hipStream_t);

template int launch_spmmv<double>(const uspmv_dmat *, const double *, double *, int, long, int, hipStream_t);
template int launch_spmmv<float>(const uspmv_dmat *, const float *, float *, int, long, int, hipStream_t);

}  // namespace uspmv_dev
