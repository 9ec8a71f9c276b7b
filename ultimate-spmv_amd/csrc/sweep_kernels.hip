// Column-window sweep SpMV (plan: host/sweep_plan.cpp): for matrices whose rows are wide and irregular, where the
// gather kernels pay one 64-byte L2 sector per 8-byte x operand and the tile-local-column plan stages nothing.
// Reference twins: scs_impl_cpu<C> (code/kernels.hpp:216-258) and scs_ap_impl_cpu<C> (code/ap_kernels.hpp:24-82);
// per row the entries are consumed in slot order with one FMA each, so y is bit-identical to them.
//
// One workgroup = one tile of blockDim.x consecutive rows (lane <-> row).  x is cut into windows of W = 2^wlog
// elements; the workgroup walks the windows its tile touches in ascending order.  Per window: the W elements are
// copied into LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no registers), double
// buffered -- window s+1 lands while window s is consumed -- and every wave runs its COMPACTED entry stream for
// that window: in round k only the lanes whose row has more than k entries in the window are active; a lane's
// element sits at base + (active lanes below it), base advancing by the number of active lanes (one ballot, one
// s_bcnt1, one v_mbcnt pair).  The stream is therefore contiguous per wave and free of padding: sizeof(VT) + 2
// bytes per non-zero, read once, non-temporally; the x operand is a ds_read from the staged window.
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

// active lanes strictly below this one
__device__ __forceinline__ unsigned lanes_below(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// One window of one wave: batches of U rounds.  AT = type of the stored values (float for the sp part of ap[dp_sp]), XT = type
// of x and of the accumulator.  vp / ip point at the wave's next element of the compacted stream and stay WAVE-UNIFORM (scalar
// registers): in round u the active lanes are those with more than k0+u entries in this window -- one ballot m[u] --, a lane's
// element sits `lanes below it in m[u]` behind the round's first element, and the round's first element is popcount(m[0..u-1])
// behind vp.  An inactive lane must not even add a signed zero.
// LOOP 0: loads under the lane mask, FMA result selected (the form measured first: 0.63 / 0.66 ms on config 4b);
// LOOP 1: scalar stream pointers (`scalar base + 32-bit lane offset` loads) and both loads and FMAs under the lane mask.
// (A branch-free form -- inactive lanes load the batch's first element, every FMA selected -- let the compiler request the whole
// batch before the first wait, but measured 0.92 / 0.84 ms: the rounds it cannot skip cost more than the waits it saves.)
template <typename AT, typename XT, int U, bool NT, int LOOP>
__device__ __forceinline__ void sweep_window(const XT *__restrict__ xs, const int c, const AT *__restrict__ &vp,
                                             const unsigned short *__restrict__ &ip, XT &acc) {
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
        AT v[U];
        unsigned ix[U];
        if constexpr (LOOP == 0) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = AT(0); ix[u] = 0u;
                if (k0 + u < c) {
                    const unsigned off = first[u] + lanes_below(m[u]);
                    v[u] = ld_stream<NT>(vp + off); ix[u] = ld_stream<NT>(ip + off);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const XT t = fma_t((XT)v[u], xs[ix[u]], acc);
                acc = (k0 + u < c) ? t : acc;
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (k0 + u < c) {
                    const unsigned off = first[u] + lanes_below(m[u]);
                    v[u] = ld_stream<NT>(vp + off); ix[u] = ld_stream<NT>(ip + off);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (k0 + u < c) acc = fma_t((XT)v[u], xs[ix[u]], acc);
        }
        vp += first[U];
        ip += first[U];
    }
}

template <typename VT, bool AP, bool NT, int NBUF, int U, int LOOP>
__global__ void __launch_bounds__(1024) scs_spmv_sweep(const int wlog, const int *__restrict__ tile_ids, const int *__restrict__ t_smin,
        const int *__restrict__ t_S, const unsigned long long *__restrict__ t_cnt_off,
        const unsigned *__restrict__ wave_off, const unsigned char *__restrict__ cnt, const VT *__restrict__ vals, const unsigned short *__restrict__ idx,
        const int *__restrict__ pad_col,
        const unsigned *__restrict__ wave_off_b, const unsigned char *__restrict__ cnt_b, const float *__restrict__ vals_b,
        const unsigned short *__restrict__ idx_b, const int *__restrict__ pad_col_b,
        const VT *__restrict__ x, VT *__restrict__ y, const long x_len, const long n_store, const int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sweep_smem[];
    constexpr int EPL = 16 / (int)sizeof(VT);           // elements per 16-byte DMA lane
    constexpr int EPP = 1024 / (int)sizeof(VT);         // elements per 1-KiB piece (one wave-instruction)
    const unsigned bt = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int R = blockDim.x, nw = R >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = tile_ids[bt], smin = t_smin[bt], S = t_S[bt];
    const long row = (long)tile * R + threadIdx.x;
    const long W = 1L << wlog;
    const int n_pieces = (int)(W / EPP);
    const unsigned char *cp = cnt + t_cnt_off[bt] + threadIdx.x;
    const unsigned char *cpb = AP ? cnt_b + t_cnt_off[bt] + threadIdx.x : nullptr;
    const VT *__restrict__ vp = vals + (unsigned)__builtin_amdgcn_readfirstlane(wave_off[bt * nw + wave]);
    const unsigned short *__restrict__ ip = idx + (unsigned)__builtin_amdgcn_readfirstlane(wave_off[bt * nw + wave]);
    const float *__restrict__ vpb = vals_b;
    const unsigned short *__restrict__ ipb = idx_b;
    if constexpr (AP) {
        vpb = vals_b + (unsigned)__builtin_amdgcn_readfirstlane(wave_off_b[bt * nw + wave]);
        ipb = idx_b + (unsigned)__builtin_amdgcn_readfirstlane(wave_off_b[bt * nw + wave]);
    }
    VT *const xs_all = (VT *)sweep_smem;                 // buffer b starts at element b * W

    auto stage = [&](const int s, const int b) {
        const long g0 = (long)(smin + s) << wlog;
        for (int p = wave; p < n_pieces; p += nw) {
            const long gi = g0 + (long)p * EPP + lane * EPL;
            const long lo = (long)b * W + (long)p * EPP;   // first element of the piece in LDS
            if (gi + EPL <= x_len) {
                __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(x + gi), (lds_void_t *)(sweep_smem + lo * (long)sizeof(VT)), 16, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < EPL; ++e)
                    if (gi + e < x_len) xs_all[lo + lane * EPL + e] = x[gi + e];
            }
        }
    };

    VT acc = VT(0);
    double acc_b = 0.0;
    int c_cur = 0, cb_cur = 0;
    if (S > 0) { c_cur = cp[0]; if (AP) cb_cur = cpb[0]; }
    if (NBUF == 2 && S > 0) stage(0, 0);
    for (int s = 0; s < S; ++s) {
        const int cb = NBUF == 2 ? (s & 1) : 0;
        const VT *cur = xs_all + (long)cb * W;
        if (NBUF == 1) {
            __syncthreads();                              // everybody is through with window s-1
            stage(s, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's share of window s has landed
        __syncthreads();
        int c_next = 0, cb_next = 0;
        if (s + 1 < S) {
            if (NBUF == 2) stage(s + 1, cb ^ 1);              // that buffer was read last in window s-1: all waves are past it
            c_next = cp[(long)(s + 1) * R];
            if (AP) cb_next = cpb[(long)(s + 1) * R];
        }
        sweep_window<VT, VT, U, NT, LOOP>(cur, c_cur, vp, ip, acc);
        if constexpr (AP) sweep_window<float, double, U, NT, LOOP>((const double *)cur, cb_cur, vpb, ipb, acc_b);
        c_cur = c_next; cb_cur = cb_next;
    }
    // trailing padding of the row, applied once (see sweep_plan.cpp)
    const int pc = pad_col[(long)bt * R + threadIdx.x];
    if (pc >= 0) acc = fma_t(VT(0), x[pc], acc);
    if constexpr (AP) {
        const int pcb = pad_col_b[(long)bt * R + threadIdx.x];
        if (pcb >= 0) acc_b = __builtin_fma((double)0.0f, (double)x[pcb], acc_b);
        acc = (VT)((double)acc + acc_b);
    }
    if (row < n_store) st_y<NT>(y + row, acc);
}

template <typename VT, bool AP>
int launch_sweep(const uspmv_dmat *A, const VT *x, VT *y, hipStream_t st) {
    const long W = 1L << A->sw_wlog;
    const int nbuf = g_tune.sweep_nbuf == 1 ? 1 : 2;
    const size_t lds = (size_t)nbuf * (size_t)W * sizeof(VT);
    const int remap = g_tune.sweep_remap;
#define SW_LAUNCH(NTV, NB, UU)                                                                                              \
    do {                                                                                                                    \
        auto kfn = g_tune.sweep_loop == 1 ? scs_spmv_sweep<VT, AP, NTV, NB, UU, 1> : scs_spmv_sweep<VT, AP, NTV, NB, UU, 0>;                 \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->sw_n_tiles), dim3(A->sw_tile_rows), lds, st, A->sw_wlog, A->sw_tile_ids,   \
                           A->sw_smin, A->sw_S, (const unsigned long long *)A->sw_cnt_off, A->sw_wave_off, A->sw_cnt,        \
                           (const VT *)A->sw_vals, A->sw_idx, A->sw_pad, A->sw_wave_off_b, A->sw_cnt_b, A->sw_vals_b,        \
                           A->sw_idx_b, A->sw_pad_b, x, y, (long)A->sw_x_len, (long)A->n_store, remap);                      \
    } while (0)
#define SW_LAUNCH_U(NTV, NB) do { if (g_tune.sweep_unroll >= 8) SW_LAUNCH(NTV, NB, 8); else if (g_tune.sweep_unroll >= 4) SW_LAUNCH(NTV, NB, 4); else SW_LAUNCH(NTV, NB, 2); } while (0)
    if (g_tune.nontemporal) { if (nbuf == 2) SW_LAUNCH_U(true, 2); else SW_LAUNCH_U(true, 1); }
    else { if (nbuf == 2) SW_LAUNCH_U(false, 2); else SW_LAUNCH_U(false, 1); }
#undef SW_LAUNCH_U
#undef SW_LAUNCH
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

}  // namespace

namespace uspmv_dev {

template <typename VT>
int launch_spmv_sweep(const uspmv_dmat *A, const VT *x, VT *y, hipStream_t st) {
    if (A->sw_n_tiles == 0) return USPMV_OK;
    return launch_sweep<VT, false>(A, x, y, st);
}

int launch_spmv_sweep_ap(const uspmv_dmat *dp, const double *x, double *y, hipStream_t st) {
    if (dp->sw_n_tiles == 0) return USPMV_OK;
    return launch_sweep<double, true>(dp, x, y, st);
}

template int launch_spmv_sweep<double>(const uspmv_dmat *, const double *, double *, hipStream_t);
template int launch_spmv_sweep<float>(const uspmv_dmat *, const float *, float *, hipStream_t);

}  // namespace uspmv_dev
