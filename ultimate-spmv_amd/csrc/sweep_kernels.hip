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
// behind vp.  Loads run under the lane mask; the FMA result is selected (an inactive lane must not even add a signed zero).
// Forms that were built, measured on config 4b and dropped (profiles/r02/config4b_sweep_variants.txt): both loads and FMAs under the
// lane mask (0.72 / 0.84 ms against 0.62 / 0.65); a branch-free form in which inactive lanes load the batch's first element, so
// that the compiler requests the whole batch before its first wait (0.92 / 0.84: the rounds it cannot skip cost more than the
// waits it saves); the first batch of window s+1 requested before the barrier that ends window s (dp 0.649 vs 0.667, but the
// registers cost the ap kernel its second workgroup per CU: 0.81).
template <typename AT, typename XT, int U, bool NT>
__device__ __forceinline__ void sweep_window(const XT *__restrict__ xs, const int c, const AT *__restrict__ &vp,
                                             const unsigned short *__restrict__ &ip, XT &acc) {
    // (entry registers of lanes that sit a round out keep what an earlier round left there -- a valid window index and a value whose
    //  product is discarded -- instead of being zeroed every round: three moves less per round)
    AT v[U];
    unsigned ix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { v[u] = AT(0); ix[u] = 0u; }
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
                const unsigned off = first[u] + lanes_below(m[u]);
                v[u] = ld_stream_g<NT>(vp + off); ix[u] = ld_stream_g<NT>(ip + off);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const XT t = fma_t((XT)v[u], xs[ix[u]], acc);
            acc = (k0 + u < c) ? t : acc;
        }
        vp += first[U];
        ip += first[U];
    }
}

// A batch of U fused multiply-adds, each under ITS round's lane mask: EXEC is set to the round's ballot, so that a lane that sits the
// round out keeps its accumulator -- not even a signed zero is added -- without the copy + two selects per round that the
// select form costs (v_mov_b64, v_fmac_f64, 2 x v_cndmask_b32: the rounds are issue-bound, profiles/r03/config4b.txt).  All lanes
// of the wave are active around the call (the kernel's control flow is wave-uniform); EXEC is saved and restored regardless.
template <int U>
__device__ __forceinline__ void masked_fma_batch(double &acc, const double (&v)[U], const double (&x)[U], const unsigned long long (&m)[U]) {
    static_assert(U == 4 || U == 8, "batch of 4 or 8 rounds");
    unsigned long long save;
    if constexpr (U == 8)
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, %[m0]\n\tv_fmac_f64 %[a], %[v0], %[x0]\n\t"
                     "s_mov_b64 exec, %[m1]\n\tv_fmac_f64 %[a], %[v1], %[x1]\n\t"
                     "s_mov_b64 exec, %[m2]\n\tv_fmac_f64 %[a], %[v2], %[x2]\n\t"
                     "s_mov_b64 exec, %[m3]\n\tv_fmac_f64 %[a], %[v3], %[x3]\n\t"
                     "s_mov_b64 exec, %[m4]\n\tv_fmac_f64 %[a], %[v4], %[x4]\n\t"
                     "s_mov_b64 exec, %[m5]\n\tv_fmac_f64 %[a], %[v5], %[x5]\n\t"
                     "s_mov_b64 exec, %[m6]\n\tv_fmac_f64 %[a], %[v6], %[x6]\n\t"
                     "s_mov_b64 exec, %[m7]\n\tv_fmac_f64 %[a], %[v7], %[x7]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(save)
                     : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [m4] "s"(m[4]), [m5] "s"(m[5]), [m6] "s"(m[6]), [m7] "s"(m[7]),
                       [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]),
                       [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]));
    else
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, %[m0]\n\tv_fmac_f64 %[a], %[v0], %[x0]\n\t"
                     "s_mov_b64 exec, %[m1]\n\tv_fmac_f64 %[a], %[v1], %[x1]\n\t"
                     "s_mov_b64 exec, %[m2]\n\tv_fmac_f64 %[a], %[v2], %[x2]\n\t"
                     "s_mov_b64 exec, %[m3]\n\tv_fmac_f64 %[a], %[v3], %[x3]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(save)
                     : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]),
                       [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]));
}
template <int U>
__device__ __forceinline__ void masked_fma_batch(float &acc, const float (&v)[U], const float (&x)[U], const unsigned long long (&m)[U]) {
    static_assert(U == 4 || U == 8, "batch of 4 or 8 rounds");
    unsigned long long save;
    if constexpr (U == 8)
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, %[m0]\n\tv_fmac_f32 %[a], %[v0], %[x0]\n\t"
                     "s_mov_b64 exec, %[m1]\n\tv_fmac_f32 %[a], %[v1], %[x1]\n\t"
                     "s_mov_b64 exec, %[m2]\n\tv_fmac_f32 %[a], %[v2], %[x2]\n\t"
                     "s_mov_b64 exec, %[m3]\n\tv_fmac_f32 %[a], %[v3], %[x3]\n\t"
                     "s_mov_b64 exec, %[m4]\n\tv_fmac_f32 %[a], %[v4], %[x4]\n\t"
                     "s_mov_b64 exec, %[m5]\n\tv_fmac_f32 %[a], %[v5], %[x5]\n\t"
                     "s_mov_b64 exec, %[m6]\n\tv_fmac_f32 %[a], %[v6], %[x6]\n\t"
                     "s_mov_b64 exec, %[m7]\n\tv_fmac_f32 %[a], %[v7], %[x7]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(save)
                     : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]), [m4] "s"(m[4]), [m5] "s"(m[5]), [m6] "s"(m[6]), [m7] "s"(m[7]),
                       [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]), [v7] "v"(v[7]),
                       [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]), [x4] "v"(x[4]), [x5] "v"(x[5]), [x6] "v"(x[6]), [x7] "v"(x[7]));
    else
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b64 exec, %[m0]\n\tv_fmac_f32 %[a], %[v0], %[x0]\n\t"
                     "s_mov_b64 exec, %[m1]\n\tv_fmac_f32 %[a], %[v1], %[x1]\n\t"
                     "s_mov_b64 exec, %[m2]\n\tv_fmac_f32 %[a], %[v2], %[x2]\n\t"
                     "s_mov_b64 exec, %[m3]\n\tv_fmac_f32 %[a], %[v3], %[x3]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [a] "+v"(acc), [sv] "=&s"(save)
                     : [m0] "s"(m[0]), [m1] "s"(m[1]), [m2] "s"(m[2]), [m3] "s"(m[3]),
                       [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [x0] "v"(x[0]), [x1] "v"(x[1]), [x2] "v"(x[2]), [x3] "v"(x[3]));
}

// TWO independent chains of one lane through the same window in one loop -- two of the lane's rows (dp / sp kernels with several rows
// per lane) or the dp and the sp part of one row (ap[dp_sp]: their accumulators only meet after the last window).  Per batch both
// chains' ballots are taken, then BOTH batches of loads are issued before the first wait: twice the entries in flight per wave
// and half as many dependent HBM round trips per window (a wave walks ~4 batches per window and chain; with 16 waves per CU and
// one workgroup per CU -- the window takes the LDS -- those round trips are what the kernel waits for, profiles/r02/pmc_cfg4b.txt:
// 51 % of the wave cycles).  Each chain still sees its entries in slot order: the FMA chains of the reference, bit for bit.
// Pointers travel by value and come back through the struct (by-reference pointer arrays made the compiler shuffle 64-bit
// scalar pairs on every batch).
template <typename A0, typename A1>
struct SweepPtrs { const A0 *v0; const unsigned short *i0; const A1 *v1; const unsigned short *i1; };

template <typename A0, typename A1, typename X0, typename X1, int U, bool NT, bool MF>
__device__ __forceinline__ SweepPtrs<A0, A1> sweep_window2(const X0 *__restrict__ xs0, const X1 *__restrict__ xs1, const int c0, const int c1,
                                                            SweepPtrs<A0, A1> p, X0 &acc0, X1 &acc1) {
    A0 v0[U];
    A1 v1[U];
    unsigned ix0[U], ix1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { v0[u] = A0(0); v1[u] = A1(0); ix0[u] = 0u; ix1[u] = 0u; }
    for (int k0 = 0;; k0 += U) {
        unsigned long long m0[U], m1[U];
        unsigned f0[U + 1], f1[U + 1];
        f0[0] = 0u; f1[0] = 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            m0[u] = __ballot(k0 + u < c0);
            m1[u] = __ballot(k0 + u < c1);
            f0[u + 1] = f0[u] + (unsigned)__popcll(m0[u]);
            f1[u + 1] = f1[u] + (unsigned)__popcll(m1[u]);
        }
        if ((m0[0] | m1[0]) == 0ull) break;                  // wave-uniform: both chains of every lane are through this window
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (k0 + u < c0) {
                const unsigned off = f0[u] + lanes_below(m0[u]);
                v0[u] = ld_stream_g<NT>(p.v0 + off); ix0[u] = ld_stream_g<NT>(p.i0 + off);
            }
            if (k0 + u < c1) {
                const unsigned off = f1[u] + lanes_below(m1[u]);
                v1[u] = ld_stream_g<NT>(p.v1 + off); ix1[u] = ld_stream_g<NT>(p.i1 + off);
            }
        }
        if constexpr (MF) {      // FMAs under the rounds' lane masks (EXEC), see masked_fma_batch
            {
                X0 w0[U], x0v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { w0[u] = (X0)v0[u]; x0v[u] = xs0[ix0[u]]; }
                masked_fma_batch<U>(acc0, w0, x0v, m0);
            }
            {
                X1 w1[U], x1v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { w1[u] = (X1)v1[u]; x1v[u] = xs1[ix1[u]]; }
                masked_fma_batch<U>(acc1, w1, x1v, m1);
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const X0 t0 = fma_t((X0)v0[u], xs0[ix0[u]], acc0);
                acc0 = (k0 + u < c0) ? t0 : acc0;
                const X1 t1 = fma_t((X1)v1[u], xs1[ix1[u]], acc1);
                acc1 = (k0 + u < c1) ? t1 : acc1;
            }
        }
        p.v0 += f0[U]; p.i0 += f0[U];
        p.v1 += f1[U]; p.i1 += f1[U];
    }
    return p;
}

// RPL rows per lane: a tile is RPL * blockDim.x rows, lane <-> rows tid, tid + blockDim.x, ...  The windows of x are staged once per
// tile, so RPL = 2 halves the staging traffic per non-zero (the workgroup is already 1 024 threads).
template <typename VT, bool AP, bool NT, int NBUF, int U, int RPL, int PAIRM = 0>
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
    const int T = blockDim.x, nw = T >> 6;              // threads, waves of the workgroup
    const long R = (long)T * RPL;                       // rows of the tile
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int tile = tile_ids[bt], smin = t_smin[bt], S = t_S[bt];
    const long W = 1L << wlog;
    const int n_pieces = (int)(W / EPP);
    const unsigned char *cp = cnt + t_cnt_off[bt] + threadIdx.x;                 // + h*T + s*R
    const unsigned char *cpb = AP ? cnt_b + t_cnt_off[bt] + threadIdx.x : nullptr;
    const VT *__restrict__ vp[RPL];
    const unsigned short *__restrict__ ip[RPL];
    const float *__restrict__ vpb[RPL];
    const unsigned short *__restrict__ ipb[RPL];
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        const unsigned o = (unsigned)__builtin_amdgcn_readfirstlane(wave_off[bt * (nw * RPL) + h * nw + wave]);
        vp[h] = vals + o; ip[h] = idx + o;
        vpb[h] = vals_b; ipb[h] = idx_b;
        if constexpr (AP) {
            const unsigned ob = (unsigned)__builtin_amdgcn_readfirstlane(wave_off_b[bt * (nw * RPL) + h * nw + wave]);
            vpb[h] = vals_b + ob; ipb[h] = idx_b + ob;
        }
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

    VT acc[RPL];
    double acc_b[RPL];
    int c_cur[RPL], cb_cur[RPL];
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        acc[h] = VT(0); acc_b[h] = 0.0; c_cur[h] = 0; cb_cur[h] = 0;
        if (S > 0) { c_cur[h] = cp[h * T]; if (AP) cb_cur[h] = cpb[h * T]; }
    }
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
        int c_next[RPL], cb_next[RPL];
#pragma unroll
        for (int h = 0; h < RPL; ++h) { c_next[h] = 0; cb_next[h] = 0; }
        if (s + 1 < S) {
            if (NBUF == 2) stage(s + 1, cb ^ 1);              // that buffer was read last in window s-1: all waves are past it
#pragma unroll
            for (int h = 0; h < RPL; ++h) {
                c_next[h] = cp[(long)(s + 1) * R + h * T];
                if (AP) cb_next[h] = cpb[(long)(s + 1) * R + h * T];
            }
        }
        constexpr bool PAIR = PAIRM != 0, MF = PAIRM == 2;
        if constexpr (PAIR && AP) {            // the dp and the sp chain of a row side by side
#pragma unroll
            for (int h = 0; h < RPL; ++h) {
                SweepPtrs<double, float> pr{(const double *)vp[h], ip[h], vpb[h], ipb[h]};
                double a0 = (double)acc[h];
                pr = sweep_window2<double, float, double, double, U, NT, MF>((const double *)cur, (const double *)cur, c_cur[h], cb_cur[h], pr, a0, acc_b[h]);
                acc[h] = (VT)a0;
                vp[h] = (const VT *)pr.v0; ip[h] = pr.i0; vpb[h] = pr.v1; ipb[h] = pr.i1;
                c_cur[h] = c_next[h]; cb_cur[h] = cb_next[h];
            }
        } else if constexpr (PAIR && RPL >= 2) {   // two of the lane's rows side by side
#pragma unroll
            for (int h = 0; h < RPL; h += 2) {
                SweepPtrs<VT, VT> pr{vp[h], ip[h], vp[h + 1], ip[h + 1]};
                pr = sweep_window2<VT, VT, VT, VT, U, NT, MF>(cur, cur, c_cur[h], c_cur[h + 1], pr, acc[h], acc[h + 1]);
                vp[h] = pr.v0; ip[h] = pr.i0; vp[h + 1] = pr.v1; ip[h + 1] = pr.i1;
                c_cur[h] = c_next[h]; c_cur[h + 1] = c_next[h + 1];
            }
        } else {
#pragma unroll
            for (int h = 0; h < RPL; ++h) {
                sweep_window<VT, VT, U, NT>(cur, c_cur[h], vp[h], ip[h], acc[h]);
                if constexpr (AP) sweep_window<float, double, U, NT>((const double *)cur, cb_cur[h], vpb[h], ipb[h], acc_b[h]);
                c_cur[h] = c_next[h]; cb_cur[h] = cb_next[h];
            }
        }
    }
    // trailing padding of the row, applied once (see sweep_plan.cpp)
#pragma unroll
    for (int h = 0; h < RPL; ++h) {
        const long row = (long)tile * R + h * T + threadIdx.x;
        const int pc = pad_col[(long)bt * R + h * T + threadIdx.x];
        if (pc >= 0) acc[h] = fma_t(VT(0), x[pc], acc[h]);
        if constexpr (AP) {
            const int pcb = pad_col_b[(long)bt * R + h * T + threadIdx.x];
            if (pcb >= 0) acc_b[h] = __builtin_fma((double)0.0f, (double)x[pcb], acc_b[h]);
            acc[h] = (VT)((double)acc[h] + acc_b[h]);
        }
        if (row < n_store) st_y<NT>(y + row, acc[h]);
    }
}

template <typename VT, bool AP>
int launch_sweep(const uspmv_dmat *A, const VT *x, VT *y, hipStream_t st) {
    const long W = 1L << A->sw_wlog;
    // (two buffers only where the plan's window leaves room for them)
    const int nbuf = (g_tune.sweep_nbuf == 2 && 2 * (size_t)W * sizeof(VT) <= 160 * 1024) ? 2 : 1;
    const size_t lds = (size_t)nbuf * (size_t)W * sizeof(VT);
    const int remap = g_tune.sweep_remap;
    // threads per workgroup: 1 024 (or the tile, if smaller) unless "sweep_threads" asks for fewer -- a lane then owns more rows
    int threads = std::min<int>(A->sw_tile_rows, g_tune.sweep_threads > 0 ? g_tune.sweep_threads : 1024);
    if (A->sw_tile_rows / threads > 4) threads = A->sw_tile_rows / 4;
    const int rpl = A->sw_tile_rows / threads;
    const bool pair = g_tune.sweep_pair != 0 && (AP || rpl >= 2);
#define SW_LAUNCH(NTV, NB, UU, RP)                                                                                          \
    do {                                                                                                                    \
        auto kfn = scs_spmv_sweep<VT, AP, NTV, NB, UU, RP>;                                                                 \
        if (pair) kfn = g_tune.sweep_pair == 2 ? scs_spmv_sweep<VT, AP, NTV, NB, UU, RP, 2> : scs_spmv_sweep<VT, AP, NTV, NB, UU, RP, 1>; \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->sw_n_tiles), dim3(threads), lds, st, A->sw_wlog, A->sw_tile_ids,           \
                           A->sw_smin, A->sw_S, (const unsigned long long *)A->sw_cnt_off, A->sw_wave_off, A->sw_cnt,        \
                           (const VT *)A->sw_vals, A->sw_idx, A->sw_pad, A->sw_wave_off_b, A->sw_cnt_b, A->sw_vals_b,        \
                           A->sw_idx_b, A->sw_pad_b, x, y, (long)A->sw_x_len, (long)A->n_store, remap);                      \
    } while (0)
#define SW_LAUNCH_R(NTV, NB, UU) do { if (rpl == 4) SW_LAUNCH(NTV, NB, UU, 4); else if (rpl == 2) SW_LAUNCH(NTV, NB, UU, 2); else SW_LAUNCH(NTV, NB, UU, 1); } while (0)
#define SW_LAUNCH_U(NTV, NB) do { if (g_tune.sweep_unroll >= 8) SW_LAUNCH_R(NTV, NB, 8); else SW_LAUNCH_R(NTV, NB, 4); } while (0)
    if (g_tune.nontemporal) { if (nbuf == 2) SW_LAUNCH_U(true, 2); else SW_LAUNCH_U(true, 1); }
    else { if (nbuf == 2) SW_LAUNCH_U(false, 2); else SW_LAUNCH_U(false, 1); }
#undef SW_LAUNCH_U
#undef SW_LAUNCH_R
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
