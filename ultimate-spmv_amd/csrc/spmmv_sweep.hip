// SpMMV over the block-vector column-window sweep plan (host/sweep_plan.cpp: uspmv_build_block_sweep_plan): 64-byte X rows (dp b = 8, sp b = 16).
// Reference loop: block_spmv_omp_scs_general (code/kernels.hpp:306-398): per row and column v the slot-ordered chain tmp += val * X[col][v].
//
// One workgroup = one tile of RPL * blockDim.x consecutive rows (lane <-> rows tid, tid + blockDim.x, ...), every row with its B accumulators
// in registers.  The workgroup walks the windows of X rows ITS rows touch, ascending; per window the 2^wlog rows are copied into LDS by
// LDS-DMA, double buffered, and every wave runs its compacted entry stream for that window (scs_spmv_sweep's stream: in round k the lanes
// whose row has more than k entries in the window own an element; one ballot gives position and advance).  Windows ascending = slots
// ascending for a column-sorted row, so each (row, v) accumulator sees its entries in slot order: the reference's FMA chain, bit for bit.
//
// Why: the phased plan (spmmv_phased.hip) stages 11.9 X rows per matrix row on the Queen_4147-class matrix -- every neighbour line once
// per slot range -- and the volume that passes L2 -> LDS is what bounds it.  Here a tile stages every window once: 3.7-6.0
// rows per matrix row.  MEASURED (profiles/r04/spmmv_sweep_probe.txt): bit-exact in both layouts, but 1.00-1.13 ms against the phased
// kernel's 0.78 ms row-wise / 0.91 ms column-wise -- a lane per row reads its 64-byte X row alone (four ds_read_b128 per entry) and the
// rounds of a wave are as long as its longest row in the window.  So this is an OPTION (uspmv_dmat_optimize_block_sweep installs the
// plan, uspmv_spmmv then prefers it), not the default; what it shows is that the staging volume can be cut 2-3 x with the reference's
// bits intact, and that column-major X and Y need no re-layout pass on this path (1.07 ms column-wise).
//
// LDS: row r of the window at r * 64 bytes, its four 16-byte pieces XOR-swizzled by (r >> 1) & 3, so that the 64 lanes of a wave --
// 64 different rows, the same piece -- spread over all bank groups (the swizzle is free on the DMA side: a lane's GLOBAL address is its own).
// XCOL: X is the caller's COLUMN-major block vector; the window is staged column by column (B runs of 2^wlog elements) and a row is read
// as B elements 2^wlog apart -- no re-layout pass, no workspace.  YCOL: Y column-major (each of the B stores of a wave is contiguous).
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

__device__ __forceinline__ unsigned lanes_below_b(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// one window of one lane's row: rounds in batches of U (see sweep_window in sweep_kernels.hip); acc[] = the row's B accumulators
template <typename VT, int B, int U, bool NT, bool XCOL>
__device__ __forceinline__ void sweep_window_block(const unsigned char *__restrict__ win, const int wlog, const int c, const VT *__restrict__ &vp,
                                                   const unsigned short *__restrict__ &ip, VT (&acc)[B]) {
    constexpr int VW = 16 / (int)sizeof(VT);
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    VT v[U];
    unsigned ix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u] = VT(0); ix[u] = 0u; }
    for (int k0 = 0;; k0 += U) {
        unsigned long long m[U];
        unsigned first[U + 1];
        first[0] = 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m[u] = __ballot(k0 + u < c);
            first[u + 1] = first[u] + (unsigned)__popcll(m[u]);
        }
        if (m[0] == 0ull) break;                             // wave-uniform: every row of the wave is through this window
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < c) {
                const unsigned off = first[u] + lanes_below_b(m[u]);
                v[u] = ld_stream_g<NT>(vp + off); ix[u] = ld_stream_g<NT>(ip + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // a lane that sits the round out reads some valid row and DISCARDS the product (select, not a branch: with a divergent branch per
            // round the compiler waits for each round's LDS reads on their own -- 1.03 -> 1.84 ms, profiles/r04/spmmv_sweep_probe.txt)
            const bool act = k0 + u < c;
            const unsigned r = ix[u];
            if constexpr (XCOL) {
                const VT *col = (const VT *)win + r;
#pragma unroll
                for (int w = 0; w < B; ++w) {
                    const VT t = fma_t(v[u], col[(size_t)w << wlog], acc[w]);
                    acc[w] = act ? t : acc[w];
                }
            } else {
                const unsigned base = r << 6, s4 = ((r >> 1) & 3u) << 4;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const vec_t xv = *(const vec_t *)(win + base + (s4 ^ ((unsigned)k << 4)));
#pragma unroll
                    for (int w = 0; w < VW; ++w) {
                        const VT t = fma_t(v[u], xv[w], acc[k * VW + w]);
                        acc[k * VW + w] = act ? t : acc[k * VW + w];
                    }
                }
            }
        }
        vp += first[U];
        ip += first[U];
    }
}

template <typename VT, int B, bool NT, bool XCOL, bool YCOL, int NBUF, int U, int RPL>
__global__ void __launch_bounds__(1024) scs_spmmv_sweep(const int wlog, const int *__restrict__ tile_ids, const int *__restrict__ t_win_ptr,
        const int *__restrict__ wins, const unsigned long long *__restrict__ t_cnt_off, const unsigned *__restrict__ wave_off,
        const unsigned char *__restrict__ cnt, const VT *__restrict__ vals, const unsigned short *__restrict__ idx, const int *__restrict__ pad_col,
        const VT *__restrict__ X, VT *__restrict__ Y, const long ld, const long x_rows, const long n_store, const int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char bsw_smem[];
    static_assert(B * (int)sizeof(VT) == 64, "64-byte X rows");
    constexpr int VW = 16 / (int)sizeof(VT);
    const unsigned bt = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int T = blockDim.x, nw = T >> 6;
    const long R = (long)T * RPL;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = tile_ids[bt], wp0 = t_win_ptr[bt], S = t_win_ptr[bt + 1] - wp0;
    const long W = 1L << wlog;
    const long wbytes = W * 64;
    const unsigned char *cp = cnt + t_cnt_off[bt] + threadIdx.x;                 // + h*T + s*R
    const VT *__restrict__ vp[RPL];
    const unsigned short *__restrict__ ip[RPL];
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane(wave_off[bt * (nw * RPL) + h * nw + wave]);
        vp[h] = vals + o; ip[h] = idx + o;
    }

    auto stage = [&](const int s, const int b) {
        const long g0 = (long)wins[wp0 + s] << wlog;          // first X row of the window
        unsigned char *dst = bsw_smem + (long)b * wbytes;
        if constexpr (XCOL) {
            // column v of the window: W elements from X + v*ld + g0; one wave instruction = 64 pieces of 16 bytes = 64*VW rows of one column
            const int ppc = (int)(W / (64 * VW));             // wave instructions per column
            for (int p = wave; p < B * ppc; p += nw) {
                const int v = p / ppc, q = p - v * ppc;
                const long r = (long)q * 64 * VW + (long)lane * VW;               // first row of this lane's piece, window-local
                const long gi = (long)v * ld + g0 + r;
                const long lo = ((long)v << wlog) + (long)q * 64 * VW;            // first element of the wave's 1-KiB run in LDS
                if (g0 + r + VW <= x_rows) {
                    __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + gi), (lds_void_t *)(dst + lo * (long)sizeof(VT)), 16, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < VW; ++e)
                        if (g0 + r + e < x_rows) ((VT *)dst)[lo + lane * VW + e] = X[gi + e];
                }
            }
        } else {
            // one wave instruction = 64 pieces = 16 rows; LDS position P = 64 p + lane holds piece (P & 3) ^ swizzle of row P >> 2
            const int n_inst = (int)(W >> 4);
            for (int p = wave; p < n_inst; p += nw) {
                const long r = (long)p * 16 + (lane >> 2);
                const int k = (lane & 3) ^ (int)((r >> 1) & 3);
                if (g0 + r < x_rows)
                    __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + (g0 + r) * B + k * VW), (lds_void_t *)(dst + (long)p * 1024), 16, 0, 0);
            }
        }
    };

    VT acc[RPL][B];
    int c_cur[RPL];
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
#pragma unroll
        for (int w = 0; w < B; ++w) acc[h][w] = VT(0);
        c_cur[h] = S > 0 ? cp[h * T] : 0;
    }
    if (NBUF == 2 && S > 0) stage(0, 0);
    for (int s = 0; s < S; ++s) {
        const int cb = NBUF == 2 ? (s & 1) : 0;
        const unsigned char *cur = bsw_smem + (long)cb * wbytes;
        if (NBUF == 1) {
            __syncthreads();                              // everybody is through with window s-1
            stage(s, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of window s has landed
        __syncthreads();
        int c_next[RPL];
#pragma unroll
        for (int h = 0; h < RPL; ++h) c_next[h] = 0;
        if (s + 1 < S) {
            if (NBUF == 2) stage(s + 1, cb ^ 1);              // that buffer was read last in window s-1: all waves are past it
#pragma unroll
            for (int h = 0; h < RPL; ++h) c_next[h] = cp[(long)(s + 1) * R + h * T];
        }
#pragma unroll
        for (int h = 0; h < RPL; ++h) {
            sweep_window_block<VT, B, U, NT, XCOL>(cur, wlog, c_cur[h], vp[h], ip[h], acc[h]);
            c_cur[h] = c_next[h];
        }
    }
    // trailing padding of the row, applied once per column (see sweep_plan.cpp), then the store
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        const long row = (long)tile * R + h * T + threadIdx.x;
        const int pc = pad_col[(long)bt * R + h * T + threadIdx.x];
        if (pc >= 0) {
#pragma unroll
            for (int w = 0; w < B; ++w) acc[h][w] = fma_t(VT(0), XCOL ? X[(long)pc + (long)w * ld] : X[(long)pc * B + w], acc[h][w]);
        }
        if (row < n_store) {
            if constexpr (YCOL) {
#pragma unroll
                for (int w = 0; w < B; ++w) st_y<NT>(Y + row + (long)w * ld, acc[h][w]);
            } else {
                typedef VT vec_t __attribute__((ext_vector_type(VW)));
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    vec_t o;
#pragma unroll
                    for (int w = 0; w < VW; ++w) o[w] = acc[h][k * VW + w];
                    *((vec_t *)(Y + row * B) + k) = o;
                }
            }
        }
    }
}

template <typename VT, int B>
int launch_bsw(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool xcol, bool ycol, hipStream_t st) {
    const long W = 1L << A->bw_wlog;
    const int nbuf = (g_tune.sweep_nbuf == 2 && 2 * (size_t)W * 64 <= 160 * 1024) ? 2 : 1;
    const size_t lds = (size_t)nbuf * (size_t)W * 64;
    const int threads = std::min<int>(A->bw_tile_rows, 1024);
    const int rpl = A->bw_tile_rows / threads;
    constexpr int VW = 16 / (int)sizeof(VT);
    if (xcol && (ld % VW != 0 || ((uintptr_t)X % 16) != 0 || W < 64 * VW)) return 1;
    if (((uintptr_t)X % 16) != 0 || ((uintptr_t)Y % 16) != 0) return 1;
    const long x_rows = xcol ? ld : std::max<long>(A->bw_x_rows, ld);      // rows of X that exist (row-major: the caller's padded_vec_size)
#define BSW_LAUNCH(XC, YC, NB, RP)                                                                                                    \
    do {                                                                                                                              \
        auto kfn = g_tune.nontemporal ? scs_spmmv_sweep<VT, B, true, XC, YC, NB, 4, RP> : scs_spmmv_sweep<VT, B, false, XC, YC, NB, 4, RP>; \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->bw_n_tiles), dim3(threads), lds, st, A->bw_wlog, A->bw_tile_ids, A->bw_win_ptr, A->bw_wins, \
                           (const unsigned long long *)A->bw_cnt_off, A->bw_wave_off, A->bw_cnt, (const VT *)A->bw_vals, A->bw_idx, A->bw_pad, \
                           X, Y, ld, x_rows, (long)A->n_store, g_tune.sweep_remap);                                                   \
    } while (0)
#define BSW_R(XC, YC, NB) do { if (rpl == 4) BSW_LAUNCH(XC, YC, NB, 4); else if (rpl == 2) BSW_LAUNCH(XC, YC, NB, 2); else BSW_LAUNCH(XC, YC, NB, 1); } while (0)
#define BSW_B(XC, YC) do { if (nbuf == 2) BSW_R(XC, YC, 2); else BSW_R(XC, YC, 1); } while (0)
    if (xcol) { if (ycol) BSW_B(true, true); else BSW_B(true, false); }
    else { if (ycol) BSW_B(false, true); else BSW_B(false, false); }
#undef BSW_B
#undef BSW_R
#undef BSW_LAUNCH
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

}  // namespace

namespace uspmv_dev {

template <typename VT>
int launch_spmmv_sweep(const uspmv_dmat *A, const VT *X, VT *Y, int b, long ld, bool xcol, bool ycol, hipStream_t st) {
    if (!A->bw || A->bw_b != b || A->part || A->bw_n_tiles != A->bw_all_tiles) return 1;
    if (A->bw_n_tiles == 0) return USPMV_OK;
    if constexpr (sizeof(VT) == 8) { if (b == 8) return launch_bsw<VT, 8>(A, X, Y, ld, xcol, ycol, st); }
    else { if (b == 16) return launch_bsw<VT, 16>(A, X, Y, ld, xcol, ycol, st); }
    return 1;
}
template int launch_spmmv_sweep<double>(const uspmv_dmat *, const double *, double *, int, long, bool, bool, hipStream_t);
template int launch_spmmv_sweep<float>(const uspmv_dmat *, const float *, float *, int, long, bool, bool, hipStream_t);

}  // namespace uspmv_dev
