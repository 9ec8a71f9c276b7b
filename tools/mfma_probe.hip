// MFMA probe for the SpMMV path (BASELINE config 3, -block_vec_size 8; north_star: "MFMA only for the block-vector path where the
// per-chunk dense panel really is a contraction"; SURVEY 7 step 6: evaluate v_mfma_f64_16x16x4_f64).  Measurement tool, not product.
//
// Work unit = 16 consecutive rows of a Queen_4147-class matrix (27-point stencil, 3 dof per node, 81 entries per interior row) times a
// block vector of b = 8 columns.  The rows of one node share their 81 columns; neighbouring nodes share two thirds of them: per
// (y,z)-neighbour line (9 of them) the 16 rows touch ONE run of 24 consecutive X rows, each row 9 of the 24.  That is the best case
// this matrix class offers a matrix core: a 16 x 24 panel per run, 37.5 % dense, times a 24 x 8 panel of X.
//   fma   : the arithmetic of scs_spmmv_quadph -- four lanes per row, compact entries (value + one-byte local index, group-major),
//           X rows from LDS, one FMA per entry and column in slot order: the reference's chain (code/kernels.hpp:306-398)
//   mfma  : per run six v_mfma_f64_16x16x4_f64 (K = 24 in steps of 4) on the panel stored DENSE with explicit zeros in the MFMA
//           operand layout (A[lane & 15][k = lane >> 4]) -- 2.67 x the value bytes; B = X rows from LDS, columns 8-15 of the 16 x 16
//           result unused (b = 8)
//   mfma_l2: the same with the dense panel of ONE unit reused by all (L2-resident): the matrix-core path without its extra HBM bytes,
//           i.e. an upper bound for any scheme that expands the compact entries on the fly
// Prints milliseconds per SpMMV-equivalent (all 4.1 M rows), registers, and max |delta| of the MFMA results against the FMA chain.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe && ./tools/mfma_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v4d __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

constexpr int RUNS = 9, RUNK = 24, NNZ_ROW = 81, GROUPS4 = (NNZ_ROW + 3) / 4;   // 21 groups of four slots (84 slots, 3 padded)
constexpr int WG_UNITS = 4;                                                      // units (waves) per workgroup

__device__ __forceinline__ long run_base(long g, int j, long n_xrows) {
    const long off = ((long)(j % 3 - 1) * 111 + (long)(j / 3 - 1) * 111 * 111) * 3;
    long b = 16 * g - 3 + off;
    return b < 0 ? 0 : (b > n_xrows - RUNK ? n_xrows - RUNK : b);
}

template <int U>
__device__ __forceinline__ int quad_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, U * 0x55, 0xf, 0xf, true); }
template <int U>
__device__ __forceinline__ double quad_bcast_d(double v) {
    return __hiloint2double(quad_bcast_i<U>(__double2hiint(v)), quad_bcast_i<U>(__double2loint(v)));
}

// stage the unit's 9 runs of 24 X rows (64 bytes each) into LDS: 216 rows = 864 pieces of 16 bytes, by LDS-DMA
__device__ __forceinline__ void stage_runs(const double *__restrict__ X, long g, long n_xrows, unsigned char *smem, int lane) {
#pragma unroll
    for (int k = 0; k < (RUNS * RUNK * 4 + 63) / 64; ++k) {
        const int pp = k * 64 + lane;
        if (pp < RUNS * RUNK * 4) {
            const int row = pp >> 2, j = row / RUNK, rr = row % RUNK;
            __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + (run_base(g, j, n_xrows) + rr) * 8 + (pp & 3) * 2), (lds_void_t *)(smem + k * 1024), 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// compact entries, group-major: vals[unit][group of four slots][row 0..15][slot % 4], idx likewise (local X row 0..215)
__global__ void __launch_bounds__(64 * WG_UNITS) k_fma(const double *__restrict__ vals, const unsigned char *__restrict__ idx, const double *__restrict__ X,
                                                      double *__restrict__ Y, long n_units, long n_xrows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane >> 2, q = lane & 3;
    const long g = (long)blockIdx.x * WG_UNITS + wave;
    if (g >= n_units) return;
    unsigned char *smem = smem_all + wave * (RUNS * RUNK * 64);
    stage_runs(X, g, n_xrows, smem, lane);
    const v2d *xs = (const v2d *)smem;
    const double *vp = vals + (g * GROUPS4 * 16 + r) * 4 + q;
    const unsigned char *ip = idx + (g * GROUPS4 * 16 + r) * 4 + q;
    v2d acc = {0.0, 0.0};
    double a[GROUPS4];
    unsigned ix[GROUPS4];
#pragma unroll
    for (int d = 0; d < GROUPS4; ++d) { a[d] = __builtin_nontemporal_load(vp + d * 64); ix[d] = __builtin_nontemporal_load(ip + d * 64); }
#define STEP(UU, D) { const double aa = quad_bcast_d<UU>(a[D]); const unsigned li = (unsigned)quad_bcast_i<UU>((int)ix[D]); const v2d xv = xs[li * 4 + q]; \
                      acc[0] = __builtin_fma(aa, xv[0], acc[0]); acc[1] = __builtin_fma(aa, xv[1], acc[1]); }
#pragma unroll
    for (int d = 0; d < GROUPS4; ++d) {
        STEP(0, d)
        if (d * 4 + 1 < NNZ_ROW) STEP(1, d)
        if (d * 4 + 2 < NNZ_ROW) STEP(2, d)
        if (d * 4 + 3 < NNZ_ROW) STEP(3, d)
    }
#undef STEP
    *((v2d *)(Y + (g * 16 + r) * 8) + q) = acc;
}

// dense panels in the MFMA A-operand layout: apad[unit][run][kstep 0..5][lane] = A[row = lane & 15][k = 4 kstep + (lane >> 4)]
template <bool SHARED_PANEL>
__global__ void __launch_bounds__(64 * WG_UNITS) k_mfma(const double *__restrict__ apad, const double *__restrict__ X, double *__restrict__ Y, long n_units, long n_xrows) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long g = (long)blockIdx.x * WG_UNITS + wave;
    if (g >= n_units) return;
    unsigned char *smem = smem_all + wave * (RUNS * RUNK * 64);
    stage_runs(X, g, n_xrows, smem, lane);
    const double *xs = (const double *)smem;
    const double *ap = apad + (SHARED_PANEL ? 0 : g * (long)(RUNS * 6 * 64)) + lane;
    const int n = lane & 15, k = lane >> 4;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    double a[RUNS * 6];
#pragma unroll
    for (int s = 0; s < RUNS * 6; ++s) a[s] = SHARED_PANEL ? ap[s * 64] : __builtin_nontemporal_load(ap + s * 64);
#pragma unroll
    for (int s = 0; s < RUNS * 6; ++s) {
        const int xr = (s / 6) * RUNK + (s % 6) * 4 + k;                     // local X row of this lane's B element
        const double b = n < 8 ? xs[xr * 8 + n] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, acc, 0, 0, 0);
    }
    if (n < 8) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) Y[(g * 16 + (lane >> 4) + 4 * rg) * 8 + n] = acc[rg];
    }
}

static inline unsigned long long mix(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}

int main(int argc, char **argv) {
    const long n_rows = argc > 1 ? atol(argv[1]) : 4102893;
    const long n_units = n_rows / 16, n_xrows = n_rows;
    const int reps = 20;
    printf("MFMA probe: %ld units of 16 rows x 81 entries (Queen_4147-class, b = 8), %ld X rows\n", n_units, n_xrows);
    // ---- host data: one unit's pattern is the same for all (rows i use columns 3*(i/3) .. +8 of every run), values hashed
    std::vector<double> hv((size_t)n_units * GROUPS4 * 64, 0.0), hpad((size_t)n_units * RUNS * 6 * 64, 0.0), hX((size_t)n_xrows * 8);
    std::vector<unsigned char> hi((size_t)n_units * GROUPS4 * 64, 0);
    for (size_t k = 0; k < hX.size(); ++k) hX[k] = 1.0 + 1e-3 * (double)(mix(k) % 1000);
#pragma omp parallel for schedule(static)
    for (long g = 0; g < n_units; ++g)
        for (int i = 0; i < 16; ++i)
            for (int e = 0; e < NNZ_ROW; ++e) {
                const int j = e / 9, c = 3 * (i / 3) + e % 9;                 // run, column inside the run (ascending: the row's slot order)
                const double v = 2.0 * (double)(mix((unsigned long long)(g * 16 + i) * 131 + e) >> 11) / 9007199254740992.0 - 1.0;
                hv[((size_t)g * GROUPS4 + e / 4) * 64 + i * 4 + e % 4] = v;
                hi[((size_t)g * GROUPS4 + e / 4) * 64 + i * 4 + e % 4] = (unsigned char)(j * RUNK + c);
                hpad[(((size_t)g * RUNS + j) * 6 + c / 4) * 64 + (c % 4) * 16 + i] = v;
            }
    double *dv, *dpad, *dX, *dY0, *dY1;
    unsigned char *di;
    HK(hipMalloc(&dv, hv.size() * 8)); HK(hipMalloc(&dpad, hpad.size() * 8)); HK(hipMalloc(&dX, hX.size() * 8)); HK(hipMalloc(&di, hi.size()));
    HK(hipMalloc(&dY0, (size_t)n_units * 128 * 8)); HK(hipMalloc(&dY1, (size_t)n_units * 128 * 8));
    HK(hipMemcpy(dv, hv.data(), hv.size() * 8, hipMemcpyHostToDevice)); HK(hipMemcpy(dpad, hpad.data(), hpad.size() * 8, hipMemcpyHostToDevice));
    HK(hipMemcpy(dX, hX.data(), hX.size() * 8, hipMemcpyHostToDevice)); HK(hipMemcpy(di, hi.data(), hi.size(), hipMemcpyHostToDevice));
    const size_t lds = (size_t)WG_UNITS * RUNS * RUNK * 64;
    const dim3 grid((unsigned)((n_units + WG_UNITS - 1) / WG_UNITS)), block(64 * WG_UNITS);
    HK(hipFuncSetAttribute((const void *)k_fma, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HK(hipFuncSetAttribute((const void *)k_mfma<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HK(hipFuncSetAttribute((const void *)k_mfma<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    HK(hipEventCreate(&e0)); HK(hipEventCreate(&e1));
    auto timeit = [&](auto launch) -> float {
        launch(); launch();
        (void)hipEventRecord(e0);
        for (int k = 0; k < reps; ++k) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / reps;
    };
    const float t_fma = timeit([&] { hipLaunchKernelGGL(k_fma, grid, block, lds, 0, dv, di, dX, dY0, n_units, n_xrows); });
    const float t_mfma = timeit([&] { hipLaunchKernelGGL(k_mfma<false>, grid, block, lds, 0, dpad, dX, dY1, n_units, n_xrows); });
    HK(hipDeviceSynchronize());
    std::vector<double> y0((size_t)n_units * 128), y1((size_t)n_units * 128);
    HK(hipMemcpy(y0.data(), dY0, y0.size() * 8, hipMemcpyDeviceToHost)); HK(hipMemcpy(y1.data(), dY1, y1.size() * 8, hipMemcpyDeviceToHost));
    double max_abs = 0, max_rel = 0; long differing = 0;
    for (size_t k = 0; k < y0.size(); ++k) {
        const double d = std::fabs(y0[k] - y1[k]);
        if (d > 0) ++differing;
        max_abs = std::max(max_abs, d); max_rel = std::max(max_rel, d / std::max(1e-300, std::fabs(y0[k])));
    }
    const float t_l2 = timeit([&] { hipLaunchKernelGGL(k_mfma<true>, grid, block, lds, 0, dpad, dX, dY1, n_units, n_xrows); });
    hipFuncAttributes fa0, fa1;
    HK(hipFuncGetAttributes(&fa0, (const void *)k_fma)); HK(hipFuncGetAttributes(&fa1, (const void *)k_mfma<false>));
    const double nnz = (double)n_units * 16 * NNZ_ROW, flops = 2.0 * nnz * 8;
    printf("  fma     (compact entries, 9 B per non-zero)          %8.4f ms  %7.0f GF/s useful   %3d VGPRs\n", t_fma, flops / t_fma / 1e6, fa0.numRegs);
    printf("  mfma    (dense 16x24 panels, explicit zeros from HBM) %8.4f ms  %7.0f GF/s useful   %3d VGPRs   issued flops = %.2f x useful\n", t_mfma, flops / t_mfma / 1e6,
           fa1.numRegs, (double)RUNS * 6 * 2048 / (16.0 * NNZ_ROW * 16));
    printf("  mfma_l2 (one panel reused by all units: no A bytes)   %8.4f ms  %7.0f GF/s useful\n", t_l2, flops / t_l2 / 1e6);
    printf("  bytes per unit: fma %d, mfma %d (values only; + 13.8 KB of X rows staged by either)\n", GROUPS4 * 64 * 9, RUNS * 6 * 64 * 8);
    printf("  MFMA vs FMA chain: %ld of %zu results differ, max |delta| = %.3e, max relative = %.3e\n", differing, y0.size(), max_abs, max_rel);
    return 0;
}
