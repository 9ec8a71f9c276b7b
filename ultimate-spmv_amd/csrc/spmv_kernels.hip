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
// Workgroup -> chunk mapping: hardware deals workgroups round-robin over the 8 XCDs; with xcd_remap = G >= 2 (default 256) the
// logical block id is permuted so that every XCD processes GROUPS of G consecutive workgroups, the eight groups of a super-block
// running side by side: neighbouring tiles -- which share x lines -- share an L2, and all XCDs stay inside one moving window of
// DRAM.  (xcd_remap = 1, one contiguous eighth of the grid per XCD, measured 4 % SLOWER than hardware order; groups of 64-1024
// measure 1-3 % faster: DESIGN.md 5.3, profiles/r01/sweepH.txt.)
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

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
template <typename T>
__device__ __forceinline__ T ld_now(const T *p) { return *(const volatile T *)p; }   // a load the compiler may neither cache nor hoist
// system-scope (sc0 sc1) load: served from memory, not from an L1 / L2 line that another kernel or GPU has made stale meanwhile
__device__ __forceinline__ double ld_sys(const double *p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}
__device__ __forceinline__ float ld_sys(const float *p) {
    return __uint_as_float(__hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
}

// I12: c16_ptrs / col16 are the handle's 12-BIT arrays (uspmv_device.hpp: tlc_c12_ptrs in dwords, tlc_col12): per pair of slot groups three
// consecutive dwords per row (8 indices, one global_load_dwordx3), an odd last group a dword + a ushort -- 1.5 instead of 2 bytes of index
// per non-zero.
// ELEM: the tile's list holds single x ELEMENTS (uspmv_build_tlc_plan with line_shift 0): thread k gathers element k of the list into LDS -- one
// 8-byte gather per DISTINCT column of the tile instead of one per entry; local indices = positions in the list.
template <typename VT, int CT, bool NT, bool IDS, int SYNC = 0, bool I12 = false, bool ELEM = false>
__global__ void __launch_bounds__(1024) scs_spmv_tlc(const long n_chunks, const int C_rt, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const int *__restrict__ col_idxs, const VT *__restrict__ values,
        const VT *__restrict__ x_arg, VT *__restrict__ y, const int *__restrict__ tile_line_ptr,
        const int *__restrict__ tile_lines, const unsigned *__restrict__ c16_ptrs,
        const unsigned short *__restrict__ col16, const long x_len, const int *__restrict__ tile_ids,
        const int xcd_remap, const long n_store, const StepArgs sa, const int *__restrict__ row_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tlc_smem[];
    VT *xs = (VT *)tlc_smem;
    constexpr int EPL = 16 / (int)sizeof(VT);   // elements per 16-byte load
    constexpr int LPL = 16 / EPL;               // lanes that copy one 16-element line
    typedef VT vec_t __attribute__((ext_vector_type(EPL)));
    const int C = CT > 0 ? CT : C_rt;
    const unsigned lbt = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const VT *x = x_arg;
    bool coh = false;                              // (SYNC 1: a late entry that runs inside the launch; wave-uniform)
    unsigned tile;
    if constexpr (SYNC == 0) {
        tile = IDS ? (unsigned)tile_ids[lbt] : lbt;
        if (IDS && (int)tile < 0) return;          // (an entry of a conditional tile list that was switched off: uspmv_dist's padding re-run)
    } else {
        // one-launch distributed step (StepSync in uspmv_device.hpp)
        __shared__ int s_go;
        long pos = lbt;
        if constexpr (SYNC == 2) {                 // the launch behind the exchange: entry lbt of this step's deferred list
            const int e = ld_now(&sa.ss->flag);
            if (lbt == 0 && threadIdx.x == 0) { sa.ss->count[(e + 1) & 1] = 0; sa.ss->expect = e + 1; }   // (nobody reads these before this launch ends)
            if ((long)lbt >= (long)ld_now(&sa.ss->count[e & 1])) return;
            pos = sa.defer[(long)(e & 1) * sa.defer_cap + lbt];
        } else if (pos >= sa.late0 && pos < sa.late0 + sa.n_real + sa.n_cond) {
            if (threadIdx.x == 0) {
                const int e = ld_now(&sa.ss->expect);
                // (relaxed: an acquire load would invalidate the caches like the fence below; the x loads that depend on it are issued
                //  after the barrier and go to memory themselves)
                const int go = __hip_atomic_load(&sa.ss->flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= e;
                if (!go) sa.defer[(long)(e & 1) * sa.defer_cap + atomicAdd(&sa.ss->count[e & 1], 1)] = (int)pos;
                s_go = go;
            }
            __syncthreads();
            if (!s_go) return;
            // the tail of x was written by another kernel (RCCL, possibly a peer GPU) while this one was running: this workgroup reads x
            // with system-scope loads, which are served from memory instead of a stale L1 / L2 line.  (An acquire FENCE here would be
            // correct too, but every such fence invalidates the XCD's L2 under the 12 000 other workgroups of the launch: 425 instead
            // of 178 us for the step's kernel, profiles/r03/dist_step_fused.txt.)
            coh = true;
        }
        if (pos >= sa.late0 + sa.n_real && pos < sa.late0 + sa.n_real + sa.n_cond) {
            // a padding tile again -- only when the delivered x[pad_col] differs in sign / finiteness from what the slot held
            const VT s = ld_now((const VT *)sa.stale), c = ld_now(x + sa.pad_col);
            if (isfinite(s) && isfinite(c) && (signbit(s) == signbit(c))) return;
            if (pos == sa.late0 + sa.n_real && threadIdx.x == 0) atomicAdd(&sa.ss->reruns, 1);
        }
        tile = (unsigned)tile_ids[pos];
    }
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
        if constexpr (ELEM) {
            for (int k = threadIdx.x; k < nl; k += blockDim.x) {
                const long idx = tile_lines[lp0 + k];
                xs[k] = idx < x_len ? x[idx] : VT(0);
            }
        } else
        for (int k = lk; k < nl; k += blockDim.x / LPL) {
            const long idx = (long)tile_lines[lp0 + k] * 16 + sub * EPL;
            vec_t v;
            if (SYNC == 1 && coh) {
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] = idx + e < x_len ? ld_sys(x + idx + e) : VT(0);
            } else if (idx + EPL <= x_len) {
                v = *(const vec_t *)(x + idx);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e) v[e] = idx + e < x_len ? x[idx + e] : VT(0);
            }
            *(vec_t *)(xs + k * 16 + sub * EPL) = v;
        }
        __syncthreads();
        if constexpr (I12) {
            if (L > 0) {
                const VT *vp = values + (long)cs + i;
                const unsigned *cw = (const unsigned *)col16 + q0 + (long)i * 3;
                const int ngt = (L + 3) >> 2, np = ngt >> 1;
#define TLC12_UNPACK(D0, D1, D2)                                                                                      \
                const unsigned ix[8] = {(D0) & 0xFFFu, ((D0) >> 12) & 0xFFFu, ((D0) >> 24) | (((D1) & 0xFu) << 8), ((D1) >> 4) & 0xFFFu, \
                                        ((D1) >> 16) & 0xFFFu, ((D1) >> 28) | (((D2) & 0xFFu) << 4), ((D2) >> 8) & 0xFFFu, (D2) >> 20};
                int p = 0;
                for (; 8 * p + 8 <= L; ++p) {
                    VT v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld_stream<NT>(vp + (long)(8 * p + u) * C);
                    const unsigned *t3 = cw + (long)p * 3 * C;
                    const unsigned d0 = ld_stream<NT>(t3), d1 = ld_stream<NT>(t3 + 1), d2 = ld_stream<NT>(t3 + 2);   // (one global_load_dwordx3)
                    TLC12_UNPACK(d0, d1, d2)
#pragma unroll
                    for (int u = 0; u < 8; ++u) acc = fma_t(v[u], xs[ix[u]], acc);
                }
                if (p < np) {                                    // the last pair holds fewer than eight slots of the row
                    const unsigned *t3 = cw + (long)p * 3 * C;
                    const unsigned d0 = ld_stream<NT>(t3), d1 = ld_stream<NT>(t3 + 1), d2 = ld_stream<NT>(t3 + 2);   // (one global_load_dwordx3)
                    TLC12_UNPACK(d0, d1, d2)
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (8 * p + u < L) acc = fma_t(ld_stream<NT>(vp + (long)(8 * p + u) * C), xs[ix[u]], acc);
                } else if (ngt & 1) {                            // an odd last group: 48 bits per row
                    const unsigned *t = (const unsigned *)col16 + q0 + (long)3 * np * C;
                    const unsigned e0 = ld_stream<NT>(t + i), e1 = ld_stream<NT>((const unsigned short *)(t + C) + i);
                    const unsigned jx[4] = {e0 & 0xFFFu, (e0 >> 12) & 0xFFFu, (e0 >> 24) | ((e1 & 0xFu) << 8), e1 >> 4};
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (8 * np + u < L) acc = fma_t(ld_stream<NT>(vp + (long)(8 * np + u) * C), xs[jx[u]], acc);
                }
#undef TLC12_UNPACK
            }
        } else
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
            for (int u = 0; u < 8; ++u) xv[u] = (SYNC == 1 && coh) ? ld_sys(x + ci[u]) : x[ci[u]];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fma_t(v[u], xv[u], acc);
        }
        for (; j < L; ++j) {
            const int cj = ld_stream<NT>(cp + (long)j * C);
            acc = fma_t(ld_stream<NT>(vp + (long)j * C), (SYNC == 1 && coh) ? ld_sys(x + cj) : x[cj], acc);
        }
    }
    if constexpr (ELEM) {
        if (row_map) {                                       // rows dealt to the tiles by the matrix graph: y through the plan's row map (plain stores: the L2 merges them)
            if (valid) { const long yr = row_map[row]; if (yr < n_store) y[yr] = acc; }
            return;
        }
    }
    if (valid && row < n_store) st_y<NT>(y + row, acc);
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

// ------------------------------------------------------------------------------------------ launchers

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

}  // namespace

namespace uspmv_dev {

template <typename VT>
int launch_spmv_tlc(const uspmv_dmat *A, const int *tile_ids, long n_tiles, const VT *x, VT *y, hipStream_t st) {
    if (n_tiles == 0) return USPMV_OK;
    const int C = (int)A->C;
    const unsigned grid = (unsigned)n_tiles;
    const bool elem = A->tlc_elem;
    const size_t lds = (size_t)A->tlc_max_lines * (elem ? 1 : 16) * sizeof(VT);
    const bool i12 = A->tlc_col12 != nullptr;                 // (12-bit local indices: uspmv_api.hip tlc_pack12)
    const unsigned *iptrs = i12 ? A->tlc_c12_ptrs : A->tlc_c16_ptrs;
    const unsigned short *idata = i12 ? (const unsigned short *)A->tlc_col12 : A->tlc_col16;
#define TLC_LAUNCH(CTV, NTV, IDSV)                                                                                    \
    do {                                                                                                              \
        auto kfn = i12 ? scs_spmv_tlc<VT, CTV, NTV, IDSV, 0, true> : scs_spmv_tlc<VT, CTV, NTV, IDSV>;                \
        if (elem) kfn = i12 ? scs_spmv_tlc<VT, CTV, NTV, IDSV, 0, true, true> : scs_spmv_tlc<VT, CTV, NTV, IDSV, 0, false, true>; \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3(grid), dim3(A->tlc_tile_rows), lds, st, (long)A->n_chunks, C, A->chunk_ptrs,        \
                           A->chunk_lengths, A->tlc_cols ? A->tlc_cols : A->col_idxs, (const VT *)(A->tlc_values ? A->tlc_values : A->values), x, y, A->tlc_line_ptr, A->tlc_lines,   \
                           iptrs, idata, (long)A->tlc_x_len, tile_ids, g_tune.xcd_remap, A->n_store, StepArgs{}, (const int *)A->tlc_row_map);  \
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
int launch_spmv_tlc_step(const uspmv_dmat *A, const int *step_ids, const StepArgs &sa, int sync, const VT *x, VT *y, hipStream_t st) {
    const long n = sync == 1 ? sa.n_early + sa.n_real + sa.n_cond : sa.n_real + sa.n_cond;
    if (n == 0) return USPMV_OK;
    if (A->tlc_elem) return uspmv::fail(USPMV_ERR_UNSUPPORTED, "one-launch distributed step: not on a plan over single x elements");
    const int C = (int)A->C;
    const size_t lds = (size_t)A->tlc_max_lines * 16 * sizeof(VT);
    const bool i12 = A->tlc_col12 != nullptr;
    const unsigned *iptrs = i12 ? A->tlc_c12_ptrs : A->tlc_c16_ptrs;
    const unsigned short *idata = i12 ? (const unsigned short *)A->tlc_col12 : A->tlc_col16;
#define TLC_STEP(CTV, SY)                                                                                             \
    do {                                                                                                              \
        auto kfn = i12 ? scs_spmv_tlc<VT, CTV, true, true, SY, true> : scs_spmv_tlc<VT, CTV, true, true, SY>;         \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)n), dim3(A->tlc_tile_rows), lds, st, (long)A->n_chunks, C, A->chunk_ptrs,  \
                           A->chunk_lengths, A->col_idxs, (const VT *)A->values, x, y, A->tlc_line_ptr, A->tlc_lines,   \
                           iptrs, idata, (long)A->tlc_x_len, step_ids, g_tune.xcd_remap, A->n_store, sa, (const int *)nullptr); \
    } while (0)
    if (sync == 1) { if (C == 32) TLC_STEP(32, 1); else TLC_STEP(0, 1); }
    else { if (C == 32) TLC_STEP(32, 2); else TLC_STEP(0, 2); }
#undef TLC_STEP
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
    if (!ids && A->sw && A->sw_tile_ids && !A->sw_idx_b && g_tune.sweep && !g_tune.ablate && g_tune.spmv_variant == 0 && ((uintptr_t)x % 16 == 0)) {
        // column-window sweep over the tiles that qualify, lane-per-row gather kernel over the chunks that are left
        if (int rc = launch_spmv_sweep<VT>(A, x, y, st)) return rc;
        return A->sw_n_rest ? launch_spmv_scs<VT>(A, A->sw_rest, (long)A->sw_n_rest, x, y, st) : USPMV_OK;
    }
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

template int launch_spmv_scs<double>(const uspmv_dmat *, const int *, long, const double *, double *, hipStream_t);
template int launch_spmv_scs<float>(const uspmv_dmat *, const int *, long, const float *, float *, hipStream_t);
template int launch_spmv_tlc<double>(const uspmv_dmat *, const int *, long, const double *, double *, hipStream_t);
template int launch_spmv_tlc<float>(const uspmv_dmat *, const int *, long, const float *, float *, hipStream_t);
template int launch_spmv_tlc_step<double>(const uspmv_dmat *, const int *, const StepArgs &, int, const double *, double *, hipStream_t);
template int launch_spmv_tlc_step<float>(const uspmv_dmat *, const int *, const StepArgs &, int, const float *, float *, hipStream_t);
template int launch_csr<double>(long, long, const int *, const int *, const double *, const double *, double *, hipStream_t);
template int launch_csr<float>(long, long, const int *, const int *, const float *, const float *, float *, hipStream_t);

}  // namespace uspmv_dev
