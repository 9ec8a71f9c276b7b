// Standalone HBM streaming micro-benchmark (measurement tool, not product code).
// Reads a "values" array (8 B/elt) and a "cols" array (4 B/elt) with different per-lane widths
// and kernel structures, to find what the SELL stream can reach on MI355X.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o tools/microbench && tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T> __device__ __forceinline__ T ldnt(const T *p) { return __builtin_nontemporal_load(p); }

// persistent, VW doubles + CW ints per lane per step, unroll U
template <int VW, int CW, int U, bool NT>
__global__ void k_persist(const double *__restrict__ v, const int *__restrict__ c, long n, double *out) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nt = (long)gridDim.x * blockDim.x;
    double acc = 0; int iacc = 0;
    // each step a wave covers 64*VW doubles and 64*VW ints (same element count), so CW == VW here
    for (long base = tid * VW; base + (long)(U - 1) * nt * VW + VW <= n; base += nt * VW * U) {
        double a[U][VW]; int b[U][CW];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long o = base + (long)u * nt * VW;
#pragma unroll
            for (int w = 0; w < VW; ++w) a[u][w] = NT ? ldnt(v + o + w) : v[o + w];
#pragma unroll
            for (int w = 0; w < CW; ++w) b[u][w] = NT ? ldnt(c + o + w) : c[o + w];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int w = 0; w < VW; ++w) acc += a[u][w];
#pragma unroll
            for (int w = 0; w < CW; ++w) iacc += b[u][w];
        }
    }
    if (acc == 12345.678 && iacc == 77) out[0] = acc;
}

// SELL-like: one lane per row, C = 32, every chunk L slots, no metadata load, no gather
template <int U, bool NT, bool PERSIST>
__global__ void k_sell(const double *__restrict__ v, const int *__restrict__ c, long n_chunks, int L, double *out) {
    const long row0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = PERSIST ? (long)gridDim.x * blockDim.x : n_chunks * 32;
    double acc = 0; int iacc = 0;
    for (long row = row0; row < n_chunks * 32; row += stride) {
        const long ch = row >> 5; const int i = row & 31;
        const double *vp = v + ch * L * 32 + i; const int *cp = c + ch * L * 32 + i;
        int j = 0;
        for (; j + U <= L; j += U) {
            double a[U]; int b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = NT ? ldnt(vp + (j + u) * 32) : vp[(j + u) * 32]; b[u] = NT ? ldnt(cp + (j + u) * 32) : cp[(j + u) * 32]; }
#pragma unroll
            for (int u = 0; u < U; ++u) { acc += a[u]; iacc += b[u]; }
        }
        for (; j < L; ++j) { acc += NT ? ldnt(vp + j * 32) : vp[j * 32]; iacc += NT ? ldnt(cp + j * 32) : cp[j * 32]; }
    }
    if (acc == 12345.678 && iacc == 77) out[0] = acc;
}

// as k_sell (non persistent) + the two dependent metadata loads (chunk_ptrs, chunk_lengths) + y store
template <int U, bool NT, bool META, int STORE, bool FMA>
__global__ void k_sell_meta(const double *__restrict__ v, const int *__restrict__ c, const int *__restrict__ cptr,
                            const int *__restrict__ clen, long n_chunks, int Lc, double *__restrict__ y, double *out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long ch = row >> 5; const int i = row & 31;
    if (ch >= n_chunks) return;
    const long cs = META ? (long)cptr[ch] : ch * Lc * 32;
    const int L = META ? clen[ch] : Lc;
    const double *vp = v + cs + i; const int *cp = c + cs + i;
    double acc = 0; int iacc = 0;
    int j = 0;
    for (; j + U <= L; j += U) {
        double a[U]; int b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { a[u] = NT ? ldnt(vp + (long)(j + u) * 32) : vp[(long)(j + u) * 32]; b[u] = NT ? ldnt(cp + (long)(j + u) * 32) : cp[(long)(j + u) * 32]; }
#pragma unroll
        for (int u = 0; u < U; ++u) { if (FMA) acc = __builtin_fma(a[u], (double)b[u], acc); else { acc += a[u]; iacc += b[u]; } }
    }
    for (; j < L; ++j) {
        const double a = NT ? ldnt(vp + (long)j * 32) : vp[(long)j * 32]; const int b = NT ? ldnt(cp + (long)j * 32) : cp[(long)j * 32];
        if (FMA) acc = __builtin_fma(a, (double)b, acc); else { acc += a; iacc += b; }
    }
    const double r = acc + iacc;
    if (STORE == 1) y[row] = r;
    else if (STORE == 2) __builtin_nontemporal_store(r, y + row);
    else if (STORE == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(y + row), "v"(r) : "memory");
    else if (STORE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(y + row), "v"(r) : "memory");
    else if (STORE == 7) y[row & 8191] = r;                       // same 64 KB: stays in L2, no HBM write stream
    else if (STORE == 8) { if ((ch & 7) == 0) y[row] = r; }       // 1/8 of the rows
    else if (STORE == 9) { if ((ch & 1) == 0) y[row] = r; }       // 1/2 of the rows
    else if (STORE == 10) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" :: "v"(y + row), "v"(r) : "memory");
    else if (STORE == 11) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" :: "v"(y + row), "v"(r) : "memory");
    else if (STORE == 5) {  // pair lanes: even lanes store 16 B
        const double o = __shfl_down(r, 1, 64);
        if ((threadIdx.x & 1) == 0) { double2 t = make_double2(r, o); *(double2 *)(y + row) = t; }
    } else if (STORE == 6) {  // stage through LDS, 64 lanes store 16 B each for 2 waves' worth? -> one wave: 32 lanes x 16 B
        const double o = __shfl_down(r, 1, 64);
        if ((threadIdx.x & 1) == 0) { double *p = y + row; __builtin_nontemporal_store(r, p); __builtin_nontemporal_store(o, p + 1); }
    }
    else if (acc == 12345.678 && iacc == 77) out[0] = acc;
}

// persistent form: every wave walks chunk pairs with a grid stride; y stores are fire-and-forget
template <int U, bool NT, int STORE>
__global__ void k_sell_pers(const double *__restrict__ v, const int *__restrict__ c, const int *__restrict__ cptr,
                            const int *__restrict__ clen, long n_chunks, double *__restrict__ y, double *out) {
    const long nrows = n_chunks * 32;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long row = (long)blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += stride) {
        const long ch = row >> 5; const int i = row & 31;
        const long cs = cptr[ch];
        const int L = clen[ch];
        const double *vp = v + cs + i; const int *cp = c + cs + i;
        double acc = 0;
        int j = 0;
        for (; j + U <= L; j += U) {
            double a[U]; int b[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { a[u] = NT ? ldnt(vp + (long)(j + u) * 32) : vp[(long)(j + u) * 32]; b[u] = NT ? ldnt(cp + (long)(j + u) * 32) : cp[(long)(j + u) * 32]; }
#pragma unroll
            for (int u = 0; u < U; ++u) acc = __builtin_fma(a[u], (double)b[u], acc);
        }
        for (; j < L; ++j) {
            const double a = NT ? ldnt(vp + (long)j * 32) : vp[(long)j * 32]; const int b = NT ? ldnt(cp + (long)j * 32) : cp[(long)j * 32];
            acc = __builtin_fma(a, (double)b, acc);
        }
        if (STORE == 1) y[row] = acc;
        else if (STORE == 2) __builtin_nontemporal_store(acc, y + row);
        else if (STORE == 4) asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(y + row), "v"(acc) : "memory");
    }
}

// each block owns G consecutive groups of 8 chunks, stages their y in LDS and writes G*2 KB in one
// burst of 16-byte stores at the end
template <int U, int G, int STORE>
__global__ void k_sell_burst(const double *__restrict__ v, const int *__restrict__ c, const int *__restrict__ cptr,
                             const int *__restrict__ clen, long n_chunks, double *__restrict__ y) {
    __shared__ double ybuf[G * 256];
    const long row_base = (long)blockIdx.x * (G * 256);
    for (int g = 0; g < G; ++g) {
        const long row = row_base + g * 256 + threadIdx.x;
        const long ch = row >> 5; const int i = row & 31;
        double acc = 0;
        if (ch < n_chunks) {
            const long cs = cptr[ch];
            const int L = clen[ch];
            const double *vp = v + cs + i; const int *cp = c + cs + i;
            int j = 0;
            for (; j + U <= L; j += U) {
                double a[U]; int b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { a[u] = ldnt(vp + (long)(j + u) * 32); b[u] = ldnt(cp + (long)(j + u) * 32); }
#pragma unroll
                for (int u = 0; u < U; ++u) acc = __builtin_fma(a[u], (double)b[u], acc);
            }
            for (; j < L; ++j) acc = __builtin_fma(ldnt(vp + (long)j * 32), (double)ldnt(cp + (long)j * 32), acc);
        }
        ybuf[g * 256 + threadIdx.x] = acc;
    }
    __syncthreads();
    const long nrows = n_chunks * 32;
    for (int k = threadIdx.x * 2; k < G * 256; k += 512) {
        const long r = row_base + k;
        if (r + 1 < nrows) {
            if (STORE == 1) *(double2 *)(y + r) = make_double2(ybuf[k], ybuf[k + 1]);
            else { typedef double d2 __attribute__((ext_vector_type(2))); d2 t; t.x = ybuf[k]; t.y = ybuf[k + 1];
                   asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(y + r), "v"(t) : "memory"); }
        }
    }
}

// chip-scale write combining: persistent workgroups (one per CU, WAVES waves) walk groups of 8*WAVES/4
// chunks block-cyclically, park y in LDS and flush M groups at a time.  All workgroups progress at
// the same rate, so the flushes of the whole chip coincide: HBM sees long pure-read phases separated
// by short pure-write bursts instead of a continuous read/write mix.
template <int U, int M, int STORE>
__global__ void __launch_bounds__(1024) k_sell_wc(const double *__restrict__ v, const int *__restrict__ c, const int *__restrict__ cptr,
                          const int *__restrict__ clen, long n_chunks, double *__restrict__ y) {
    extern __shared__ double ybuf[];                       // M * blockDim.x doubles
    const long nrows = n_chunks * 32;
    const long ngroups = (nrows + blockDim.x - 1) / blockDim.x;
    long g = blockIdx.x;
    while (g < ngroups) {
        int m = 0;
        long g0 = g;
        for (; m < M && g < ngroups; ++m, g += gridDim.x) {
            const long row = g * blockDim.x + threadIdx.x;
            const long ch = row >> 5; const int i = row & 31;
            double acc = 0;
            if (ch < n_chunks) {
                const long cs = cptr[ch];
                const int L = clen[ch];
                const double *vp = v + cs + i; const int *cp = c + cs + i;
                int j = 0;
                for (; j + U <= L; j += U) {
                    double a[U]; int b[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { a[u] = ldnt(vp + (long)(j + u) * 32); b[u] = ldnt(cp + (long)(j + u) * 32); }
#pragma unroll
                    for (int u = 0; u < U; ++u) acc = __builtin_fma(a[u], (double)b[u], acc);
                }
                for (; j < L; ++j) acc = __builtin_fma(ldnt(vp + (long)j * 32), (double)ldnt(cp + (long)j * 32), acc);
            }
            ybuf[m * blockDim.x + threadIdx.x] = acc;       // each lane re-reads only its own slots: no barrier needed
        }
        for (int k = 0; k < m; ++k) {
            const long row = (g0 + (long)k * gridDim.x) * blockDim.x + threadIdx.x;
            if (row < nrows) {
                const double r = ybuf[k * blockDim.x + threadIdx.x];
                if (STORE == 1) y[row] = r;
                else asm volatile("global_store_dwordx2 %0, %1, off sc1" :: "v"(y + row), "v"(r) : "memory");
            }
        }
    }
}

// slot-blocked SELL: lane = row, per step one double2 (2 slots) + one int2, chunk stored [j/2][i][j%2]
template <int U, bool NT>
__global__ void k_sell_b2(const double2 *__restrict__ v, const int2 *__restrict__ c, long n_chunks, int L2, double *out) {
    const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= n_chunks * 32) return;
    const long ch = row >> 5; const int i = row & 31;
    const double2 *vp = v + ch * L2 * 32 + i; const int2 *cp = c + ch * L2 * 32 + i;
    double acc = 0; int iacc = 0;
    int j = 0;
    for (; j + U <= L2; j += U) {
        double2 a[U]; int2 b[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const double *pa = (const double *)(vp + (j + u) * 32); const int *pb = (const int *)(cp + (j + u) * 32);
            a[u] = make_double2(NT ? ldnt(pa) : pa[0], NT ? ldnt(pa + 1) : pa[1]);
            b[u] = make_int2(NT ? ldnt(pb) : pb[0], NT ? ldnt(pb + 1) : pb[1]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { acc += a[u].x + a[u].y; iacc += b[u].x + b[u].y; }
    }
    for (; j < L2; ++j) { double2 a = vp[j * 32]; int2 b = cp[j * 32]; acc += a.x + a.y; iacc += b.x + b.y; }
    if (acc == 12345.678 && iacc == 77) out[0] = acc;
}

template <typename F> float timeit(F f, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); f();
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / reps;
}

int main() {
    const long n_chunks = 506072, L = 27;             // nlpkkt200-class
    const long n = n_chunks * 32 * 28;                // room for the L=28 blocked layout
    double *v; int *c; double *out;
    CK(hipMalloc(&v, n * 8)); CK(hipMalloc(&c, n * 4)); CK(hipMalloc(&out, 64));
    {   // random contents (zero-filled HBM reads can run at a different power/clock point)
        std::vector<double> hv(n); std::vector<int> hc(n);
        unsigned long long z = 88172645463325252ull;
        for (long k = 0; k < n; ++k) { z ^= z << 13; z ^= z >> 7; z ^= z << 17; hv[k] = (double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5; hc[k] = (int)(z & 0xFFFFF); }
        CK(hipMemcpy(v, hv.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(c, hc.data(), n * 4, hipMemcpyHostToDevice));
    }
    int *cptr, *clen; double *y;
    CK(hipMalloc(&cptr, (n_chunks + 1) * 4)); CK(hipMalloc(&clen, n_chunks * 4)); CK(hipMalloc(&y, n_chunks * 32 * 8));
    {
        std::vector<int> hp(n_chunks + 1), hl(n_chunks, (int)L);
        for (long k = 0; k <= n_chunks; ++k) hp[k] = (int)(k * L * 32);
        CK(hipMemcpy(cptr, hp.data(), (n_chunks + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(clen, hl.data(), n_chunks * 4, hipMemcpyHostToDevice));
    }
    const long nel = n_chunks * 32 * L;
    const double gb = nel * 12.0 / 1e9, gb28 = n_chunks * 32 * 28 * 12.0 / 1e9;
    auto rep = [&](const char *name, float ms, double g) { printf("%-44s %8.4f ms  %7.0f GB/s\n", name, ms, g / ms * 1e3); fflush(stdout); };
    for (int round = 0; round < 2; ++round) {
#define P(VW, U, NT, G) rep("persist VW=" #VW " U=" #U " nt=" #NT " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_persist<VW, VW, U, NT>), dim3(G), dim3(256), 0, 0, v, c, nel, out); }, 20), gb)

#define S(U, NT, PERS, G) rep("sell U=" #U " nt=" #NT " persist=" #PERS " grid=" #G, timeit([&] { hipLaunchKernelGGL((k_sell<U, NT, PERS>), dim3(G), dim3(256), 0, 0, v, c, n_chunks, (int)L, out); }, 20), gb)
        const int gfull = (int)((n_chunks * 32 + 255) / 256);
        S(8, true, false, gfull);
#define M(U, META, STORE, FMA) rep("sell U=" #U " meta=" #META " store=" #STORE " fma=" #FMA, timeit([&] { hipLaunchKernelGGL((k_sell_meta<U, true, META, STORE, FMA>), dim3(gfull), dim3(256), 0, 0, v, c, cptr, clen, n_chunks, (int)L, y, out); }, 20), gb)
        M(8, true, 1, true); M(8, true, 4, true);
#define WC(U, M, STORE, G, T) rep("sell-wc U=" #U " M=" #M " store=" #STORE " grid=" #G " threads=" #T, timeit([&] { hipLaunchKernelGGL((k_sell_wc<U, M, STORE>), dim3(G), dim3(T), M * T * 8, 0, v, c, cptr, clen, n_chunks, y); }, 20), gb)
        WC(4, 16, 4, 256, 1024); WC(4, 1, 4, 256, 1024); WC(4, 4, 4, 256, 1024); WC(4, 19, 4, 256, 1024); WC(4, 16, 1, 256, 1024); WC(2, 16, 4, 256, 1024); WC(4, 16, 4, 512, 512); WC(4, 16, 4, 1024, 256); WC(4, 8, 4, 2048, 256); WC(4, 1, 4, 2048, 256); WC(9, 16, 4, 256, 1024); WC(3, 16, 4, 256, 1024);
#define BU(G, STORE) rep("sell-burst G=" #G " store=" #STORE, timeit([&] { hipLaunchKernelGGL((k_sell_burst<8, G, STORE>), dim3((unsigned)((n_chunks * 32 + G * 256 - 1) / (G * 256))), dim3(256), 0, 0, v, c, cptr, clen, n_chunks, y); }, 20), gb)
        BU(32, 4);
#define B2(U, NT) rep("sell slot-blocked-2 (L=28) U=" #U " nt=" #NT, timeit([&] { hipLaunchKernelGGL((k_sell_b2<U, NT>), dim3(gfull), dim3(256), 0, 0, (const double2 *)v, (const int2 *)c, n_chunks, 14, out); }, 20), gb28)

        printf("----\n");
    }
    return 0;
}
