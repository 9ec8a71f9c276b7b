// Adaptive-precision dp+sp SpMV kernels: reference twins scs_ap_impl_cpu<C> (code/ap_kernels.hpp:24-82) and
// spmv_omp_scs_ap (:562-634).  See uspmv_device.hpp / DESIGN.md 5.
#include "uspmv_device.hpp"

using namespace uspmv_dev;

namespace {

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

// Adaptive precision dp+sp, one lane per row: the dp chain, then the sp chain (float value widened,
// times the DOUBLE x, accumulated in double), y = dp + sp  (code/ap_kernels.hpp:59-75).
// SPX: the generic-C reference kernel spmv_omp_scs_ap multiplies the sp values with the FLOAT copy
// of x (float product, rounded, then widened and added; code/ap_kernels.hpp:619-623).
template <int U, bool NT, bool SPX, int CT, bool IDS = false>
__global__ void scs_spmv_ap_rows(const long n_chunks, const int C_rt, const int *__restrict__ dp_cp,
                                 const int *__restrict__ dp_cl, const int *__restrict__ dp_ci,
                                 const double *__restrict__ dp_va, const int *__restrict__ sp_cp,
                                 const int *__restrict__ sp_cl, const int *__restrict__ sp_ci,
                                 const float *__restrict__ sp_va, const double *__restrict__ x,
                                 const float *__restrict__ x_sp, double *__restrict__ y, const int xcd_remap,
                                 const int *__restrict__ chunk_ids = nullptr) {
    const int C = CT > 0 ? CT : C_rt;        // CT = 32: slot strides become immediate offsets
    const unsigned lb = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const long vrow = (long)lb * blockDim.x + threadIdx.x;
    const long vc = vrow / C;                // IDS: virtual chunk vc -> chunk_ids[vc] (the chunks a sweep plan leaves over)
    const int i = (int)(vrow - vc * C);
    if (vc >= n_chunks) return;
    const long c = IDS ? (long)chunk_ids[vc] : vc;
    const long row = c * C + i;
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

}  // namespace

namespace uspmv_dev {

int launch_spmv_ap_chunks(const uspmv_dmat *dp, const uspmv_dmat *sp, const int *chunk_ids, long n_ids, const double *d_x,
                          double *d_y, hipStream_t stream) {
    if (n_ids == 0) return USPMV_OK;
    const unsigned grid = grid_for(n_ids * dp->C, 256);
    if (g_tune.nontemporal)
        hipLaunchKernelGGL((scs_spmv_ap_rows<4, true, false, 0, true>), dim3(grid), dim3(256), 0, stream, n_ids, (int)dp->C, dp->chunk_ptrs,
                           dp->chunk_lengths, dp->col_idxs, (const double *)dp->values, sp->chunk_ptrs, sp->chunk_lengths, sp->col_idxs,
                           (const float *)sp->values, d_x, (const float *)nullptr, d_y, g_tune.xcd_remap, chunk_ids);
    else
        hipLaunchKernelGGL((scs_spmv_ap_rows<4, false, false, 0, true>), dim3(grid), dim3(256), 0, stream, n_ids, (int)dp->C, dp->chunk_ptrs,
                           dp->chunk_lengths, dp->col_idxs, (const double *)dp->values, sp->chunk_ptrs, sp->chunk_lengths, sp->col_idxs,
                           (const float *)sp->values, d_x, (const float *)nullptr, d_y, g_tune.xcd_remap, chunk_ids);
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

int launch_spmv_ap(const uspmv_dmat *dp, const uspmv_dmat *sp, const double *d_x, const float *d_x_sp, double *d_y,
                   hipStream_t stream) {
    if (!d_x_sp && dp->sw && sp->sw && dp->sw_tile_ids && dp->sw_idx_b && dp->sw_plan_id == sp->sw_plan_id && g_tune.sweep &&
        ((uintptr_t)d_x % 16 == 0)) {
        if (int rc = launch_spmv_sweep_ap(dp, d_x, d_y, stream)) return rc;
        return launch_spmv_ap_chunks(dp, sp, dp->sw_rest, (long)dp->sw_n_rest, d_x, d_y, stream);
    }
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
#define AP_LAUNCH_UC(UU, NTV, SPXV, CTV)                                                                             \
    hipLaunchKernelGGL((scs_spmv_ap_rows<UU, NTV, SPXV, CTV>), dim3(grid), dim3(block), 0, (hipStream_t)stream,      \
                       (long)dp->n_chunks, (int)dp->C, dp->chunk_ptrs, dp->chunk_lengths, dp->col_idxs,              \
                       (const double *)dp->values, sp->chunk_ptrs, sp->chunk_lengths, sp->col_idxs,                  \
                       (const float *)sp->values, d_x, d_x_sp, d_y, g_tune.xcd_remap)
#define AP_LAUNCH_U(UU, NTV, SPXV) do { if (dp->C == 32) AP_LAUNCH_UC(UU, NTV, SPXV, 32); else AP_LAUNCH_UC(UU, NTV, SPXV, 0); } while (0)
#define AP_LAUNCH(NTV, SPXV)                                                                                         \
    do { if (g_tune.unroll >= 8) AP_LAUNCH_U(8, NTV, SPXV); else if (g_tune.unroll == 4) AP_LAUNCH_U(4, NTV, SPXV);  \
         else AP_LAUNCH_U(2, NTV, SPXV); } while (0)
    if (d_x_sp) { if (g_tune.nontemporal) AP_LAUNCH(true, true); else AP_LAUNCH(false, true); }
    else { if (g_tune.nontemporal) AP_LAUNCH(true, false); else AP_LAUNCH(false, false); }
#undef AP_LAUNCH_U
#undef AP_LAUNCH_UC
#undef AP_LAUNCH
    HIP_TRY(hipGetLastError());
    return USPMV_OK;
}

}  // namespace uspmv_dev
