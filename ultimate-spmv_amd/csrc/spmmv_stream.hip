// SpMMV over the phased block plan as a STREAM (64-byte X rows, C = 32, one-byte phase-local indices): persistent workgroups walk a flat
// schedule of phase descriptors, and everything a phase needs is requested one phase (its X-row list: two, its descriptor: three) ahead --
// the X rows by DMA into the other half of a double LDS buffer, the matrix entries into a second register set -- so that a workgroup's
// round trips (descriptor -> list -> X rows + entries) lie behind the arithmetic of its OWN previous phase instead of only behind other
// workgroups, and one barrier per phase is left.  The arithmetic is that of scs_spmmv_quadph (spmmv_phased.hip): four lanes per row, a
// slot's (value, index) quad-broadcast, one ds_read_b128 per lane and slot, every row its slot-ordered FMA chain -- bit-identical to
// block_spmv_omp_scs_general (code/kernels.hpp:306-398).
#include "uspmv_device.hpp"
#include <vector>

using namespace uspmv_dev;

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

// one phase of one tile, 32 bytes, in the order its workgroup meets it
struct alignas(32) PhDesc {
    unsigned va, vb;     // element offset (values and index bytes alike) of the phase's first group in the tile's two chunks
    int lp, nl;          // the phase's X-row list: first entry, entries
    unsigned pk;         // bits 0-3 / 4-7: full groups of chunk A / B in the phase; 8-9 / 10-11: slots of their partial last group when it lies in the phase; 12: first phase of its tile; 13: last
    int tile;
    int spare0, spare1;
};
constexpr unsigned PK_FIRST = 1u << 12, PK_LAST = 1u << 13;

__global__ void __launch_bounds__(256) stream_desc_fill(const long n_tiles, const long n_chunks, const int *__restrict__ chunk_lengths,
        const unsigned *__restrict__ c16_ptrs, const int *__restrict__ ph_ptr, const int *__restrict__ ph_g0, const int *__restrict__ ph_list_ptr,
        const int *__restrict__ slot, PhDesc *__restrict__ out) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const long cA = 2 * t, cB = 2 * t + 1;
    const int LA = cA < n_chunks ? chunk_lengths[cA] : 0, LB = cB < n_chunks ? chunk_lengths[cB] : 0;
    const unsigned qA = cA < n_chunks ? c16_ptrs[cA] : 0u, qB = cB < n_chunks ? c16_ptrs[cB] : 0u;
    const int fA = LA >> 2, rA = LA & 3, fB = LB >> 2, rB = LB & 3;
    const int p0 = ph_ptr[t], p1 = ph_ptr[t + 1];
    PhDesc *o = out + slot[t];
    if (p0 >= p1) {                                          // a tile of empty rows: its y rows are still written
        PhDesc d{0u, 0u, 0, 0, PK_FIRST | PK_LAST, (int)t, 0, 0};
        *o = d;
        return;
    }
    for (int ph = p0; ph < p1; ++ph) {
        const int g0 = ph_g0[ph], g1 = ph + 1 < p1 ? ph_g0[ph + 1] : 0x7fffffff;
        const int nfa = max(min(g1, fA) - g0, 0), nfb = max(min(g1, fB) - g0, 0);
        const int ra = (rA && fA >= g0 && fA < g1) ? rA : 0, rb = (rB && fB >= g0 && fB < g1) ? rB : 0;
        PhDesc d;
        d.va = nfa + ra > 0 ? qA + (unsigned)g0 * 128u : 0u;      // (0: nothing of the chunk in this phase -- the depth-2 kernel loads unconditionally, from a valid address)
        d.vb = nfb + rb > 0 ? qB + (unsigned)g0 * 128u : 0u;
        d.lp = ph_list_ptr[ph]; d.nl = ph_list_ptr[ph + 1] - d.lp;
        d.pk = (unsigned)nfa | ((unsigned)nfb << 4) | ((unsigned)ra << 8) | ((unsigned)rb << 10) | (ph == p0 ? PK_FIRST : 0u) | (ph + 1 == p1 ? PK_LAST : 0u);
        d.tile = (int)t; d.spare0 = d.spare1 = 0;
        o[ph - p0] = d;
    }
}

// MW: waves per SIMD the register allocation aims at (4: four workgroups per CU; 5: five, the most 32 KiB of LDS each allow).  xcd_remap != 0: the
// schedule has ONE TILE PER WORKGROUP ("spmmv_stream" 99: the hardware deals the tiles out as workgroups finish, like the default kernel, and only the
// phases of a tile are pipelined) and workgroup b takes entry remap_block(b) of it.
template <typename VT, int B, bool NT, bool YCOL, bool YNT, int ABL, int MW = 4>
__global__ void __launch_bounds__(256, MW) scs_spmmv_pstream(const PhDesc *__restrict__ desc, const int *__restrict__ wg_ptr,
        const VT *__restrict__ values, const unsigned char *__restrict__ col8, const int *__restrict__ xrows, const VT *__restrict__ X,
        VT *__restrict__ Y, const long ld, const long n_rows_pad, const long n_store, const int *__restrict__ row_map, long long *__restrict__ wg_clock,
        const int xcd_remap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];     // two buffers of 256 X rows (16 KiB each)
    if (wg_clock && threadIdx.x == 0) wg_clock[blockIdx.x] = (long long)wall_clock64();      // (measurement aid: USPMV_STREAM_CLOCK)
    constexpr int VW = 16 / (int)sizeof(VT);
    static_assert(B == 4 * VW, "four 16-byte pieces per X row");
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane >> 2, q = lane & 3;
    const int sel = wave >> 1, hoff = (wave & 1) * 64 + lane;                  // the wave's chunk of the tile; its lanes' element inside a group
    const unsigned wg = xcd_remap ? remap_block(blockIdx.x, gridDim.x, xcd_remap) : blockIdx.x;
    const int d_beg = wg_ptr[wg], d_end = wg_ptr[wg + 1];
    if (d_beg >= d_end) return;
    typedef int v8i __attribute__((ext_vector_type(8)));
    // descriptors of phases s, s + 1, s + 2 as scalars: {va, vb, lp, nl, pk, tile, -, -}; dl = the one in flight (phase s + 3 once iteration s has
    // asked for it): a VECTOR load of the same address by every lane, made scalar behind the next iteration's wait.  (A scalar load has to be
    // waited for where it is used -- as plain C++ that was an immediate wait that also covered the X rows and entries just requested -- and by
    // inline asm the register allocator copies its destination while the load is still in flight.)
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    v8i D0 = {0, 0, 0, 0, 0, 0, 0, 0}, D1 = D0, D2 = D0;
    v4i dl0 = {0, 0, 0, 0};
    v2i dl1 = {0, 0};
    VT a[2][8];
    unsigned ix[2][8];
    int xr[4] = {-1, -1, -1, -1};
    int y0 = -1, y1 = -1, y2 = -1;                                             // y rows of the lane for the tiles of phases s, s + 1, s + 2
    vec_t acc;
#pragma unroll
    for (int w = 0; w < VW; ++w) acc[w] = VT(0);

#define PS_STEP(UU, AV, IV)                                                                                   \
    {                                                                                                         \
        const VT aa = quad_bcast<UU>(AV);                                                                     \
        const unsigned li = (unsigned)quad_bcast<UU>((int)(IV));                                              \
        const vec_t xv = xs[li * 4u + (unsigned)q];                                                           \
        _Pragma("unroll") for (int w = 0; w < VW; ++w) acc[w] = fma_t(aa, xv[w], acc[w]);                     \
    }
    // one iteration; P = register set that holds phase s (the other one receives phase s + 1)
#define PS_BODY(P)                                                                                                                   \
    {                                                                                                                                \
        __builtin_amdgcn_s_waitcnt(0);                                 /* X rows and entries of phase s, list of s + 1, descriptor of s + 2 */ \
        __syncthreads();                                               /* ... of every wave; and every wave is through with phase s - 1 */ \
        D0 = D1; D1 = D2; y0 = y1; y1 = y2;                                                                                          \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) D2[j] = __builtin_amdgcn_readfirstlane(dl0[j]);                                \
        D2[4] = __builtin_amdgcn_readfirstlane(dl1[0]); D2[5] = __builtin_amdgcn_readfirstlane(dl1[1]);                              \
        if (s + 3 >= d_beg && s + 3 < d_end) {                                                                                       \
            const int *dp = (const int *)(desc + (s + 3));                                                                           \
            dl0 = *(const v4i *)dp; dl1 = *(const v2i *)(dp + 4);                                                                    \
        }                                                                                                                            \
        if (s + 1 >= d_beg && s + 1 < d_end) {                                                                                       \
            unsigned char *buf = ps_smem + ((s + 1) & 1) * 16384;                                                                    \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                                            \
                if (!(ABL & 1) && xr[k] >= 0)                                                                                        \
                    __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + (long)xr[k] * B + q * VW), (lds_void_t *)(buf + (wave + 4 * k) * 1024), 16, 0, 0); \
            const unsigned pk1 = (unsigned)D1[4];                                                                                    \
            const unsigned vo = (unsigned)(sel ? D1[1] : D1[0]) + (unsigned)hoff;                                                    \
            const int ngl = (int)((pk1 >> (4 * sel)) & 15u) + (((pk1 >> (8 + 2 * sel)) & 3u) ? 1 : 0);                               \
            _Pragma("unroll") for (int d = 0; d < 8; ++d)                                                                            \
                if (d < ngl) {                                                                                                       \
                    if (!(ABL & 8)) ix[(P) ^ 1][d] = ld_stream<NT>(col8 + vo + d * 128);                                             \
                    if (!(ABL & 4)) a[(P) ^ 1][d] = ld_stream<NT>(values + vo + d * 128);                                            \
                }                                                                                                                    \
        }                                                                                                                            \
        if (s + 2 >= d_beg && s + 2 < d_end) {                                                                                       \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                          \
                const int e = ((wave + 4 * k) * 64 + lane) >> 2;                                                                     \
                xr[k] = -1;                                                                                                          \
                if (e < D2[3]) xr[k] = xrows[D2[2] + e];                                                                             \
            }                                                                                                                        \
            y2 = y1;                                                                                                                 \
            if ((unsigned)D2[4] & PK_FIRST) {                                                                                        \
                const long row = (long)D2[5] * 64 + wave * 16 + r;                                                                   \
                y2 = -1;                                                                                                             \
                if (row < n_rows_pad) y2 = row_map ? row_map[row] : (int)row;                                                        \
            }                                                                                                                        \
        }                                                                                                                            \
        if (s >= d_beg) {                                                                                                            \
            const vec_t *xs = (const vec_t *)(ps_smem + (s & 1) * 16384);                                                            \
            const unsigned pk0 = (unsigned)D0[4];                                                                                    \
            const int nf = (int)((pk0 >> (4 * sel)) & 15u), rem = (int)((pk0 >> (8 + 2 * sel)) & 3u);                                \
            if (!(ABL & 2)) {                                                                                                        \
                _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                                      \
                    if (d < nf) { PS_STEP(0, a[P][d], ix[P][d]) PS_STEP(1, a[P][d], ix[P][d]) PS_STEP(2, a[P][d], ix[P][d]) PS_STEP(3, a[P][d], ix[P][d]) } \
                    else if (d == nf && rem) {                                                                                       \
                        PS_STEP(0, a[P][d], ix[P][d])                                                                                \
                        if (rem > 1) PS_STEP(1, a[P][d], ix[P][d])                                                                   \
                        if (rem > 2) PS_STEP(2, a[P][d], ix[P][d])                                                                   \
                    }                                                                                                                \
                }                                                                                                                    \
            }                                                                                                                        \
            if (pk0 & PK_LAST) {                                                                                                     \
                if (y0 >= 0 && y0 < n_store) {                                                                                       \
                    if (YCOL) { _Pragma("unroll") for (int w = 0; w < VW; ++w) st_y<YNT>(Y + ((long)y0 + (long)(q * VW + w) * ld), acc[w]); } \
                    else *((vec_t *)(Y + (long)y0 * B) + q) = acc;                                                                   \
                }                                                                                                                    \
                _Pragma("unroll") for (int w = 0; w < VW; ++w) acc[w] = VT(0);                                                       \
            }                                                                                                                        \
        }                                                                                                                            \
        ++s;                                                                                                                         \
    }

    int s = d_beg - 3;
    while (s < d_end) {
        PS_BODY(0)
        if (s >= d_end) break;
        PS_BODY(1)
    }
    if (wg_clock && threadIdx.x == 0) wg_clock[gridDim.x + blockIdx.x] = (long long)wall_clock64();
#undef PS_BODY
#undef PS_STEP
}

// Depth 2: the X rows and entries of phase s + 2 are requested while phase s is multiplied (three LDS buffers, three register sets), and the wait at the
// top of an iteration is PARTIAL -- vmcnt(20): everything but the 4 DMA + 16 entry loads of the previous iteration -- so that a workgroup has
// requests in flight also while it waits and while it computes (with depth 1 the workgroups of a CU fall into step: they wait together and compute
// together, and the parts of the time add up, profiles/r04/stream/).  For the count to be a constant every load of an iteration is unconditional: list
// entries, rows and groups beyond the phase's are fetched from valid dummy addresses and never used.  Order inside an iteration: the small loads
// (descriptor, list, y row) first, the 20 large ones last -- loads complete in order, so "at most 20 outstanding" means every older one has landed.
template <typename VT, int B, bool NT, bool YCOL, bool YNT, bool RM>
__global__ void __launch_bounds__(256, 3) scs_spmmv_pstream2(const PhDesc *__restrict__ desc, const int *__restrict__ wg_ptr,
        const VT *__restrict__ values, const unsigned char *__restrict__ col8, const int *__restrict__ xrows, const VT *__restrict__ X,
        VT *__restrict__ Y, const long ld, const long n_rows_pad, const long n_store, const int *__restrict__ row_map, long long *__restrict__ wg_clock,
        const int /*xcd_remap*/) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ps_smem[];     // three buffers of 256 X rows (16 KiB each)
    if (wg_clock && threadIdx.x == 0) wg_clock[blockIdx.x] = (long long)wall_clock64();
    constexpr int VW = 16 / (int)sizeof(VT);
    static_assert(B == 4 * VW, "four 16-byte pieces per X row");
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    typedef int v8i __attribute__((ext_vector_type(8)));
    typedef int v4i __attribute__((ext_vector_type(4)));
    typedef int v2i __attribute__((ext_vector_type(2)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane >> 2, q = lane & 3;
    const int sel = wave >> 1, hoff = (wave & 1) * 64 + lane;
    const int d_beg = wg_ptr[blockIdx.x], d_end = wg_ptr[blockIdx.x + 1];
    if (d_beg >= d_end) return;
    v8i D0 = {0, 0, 0, 0, 0, 0, 0, 0}, D1 = D0, D2 = D0, D3 = D0;               // descriptors of phases s .. s + 3
    v4i dl0 = {0, 0, 0, 0};
    v2i dl1 = {0, 0};
    VT a[3][8];
    unsigned ix[3][8];
    int xr[4] = {0, 0, 0, 0};
    int y0 = -1, y1 = -1, y2 = -1, y3 = -1;
    vec_t acc;
#pragma unroll
    for (int w = 0; w < VW; ++w) acc[w] = VT(0);
    const long row_last = n_rows_pad - 1;

#define PS_STEP(UU, AV, IV)                                                                                   \
    {                                                                                                         \
        const VT aa = quad_bcast<UU>(AV);                                                                     \
        const unsigned li = (unsigned)quad_bcast<UU>((int)(IV));                                              \
        const vec_t xv = xs[li * 4u + (unsigned)q];                                                           \
        _Pragma("unroll") for (int w = 0; w < VW; ++w) acc[w] = fma_t(aa, xv[w], acc[w]);                     \
    }
    // P = register set / LDS buffer of phase s; phase s + 2 goes to (P + 2) % 3
#define PS2_BODY(P)                                                                                                                  \
    {                                                                                                                                \
        __builtin_amdgcn_s_waitcnt(0x4F74);                            /* vmcnt(20), the other counters left alone */                \
        asm volatile("" ::: "memory");                                                                                               \
        __builtin_amdgcn_s_barrier();                                                                                                \
        asm volatile("" ::: "memory");                                                                                               \
        D0 = D1; D1 = D2; D2 = D3; y0 = y1; y1 = y2; y2 = y3;                                                                        \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) D3[j] = __builtin_amdgcn_readfirstlane(dl0[j]);                                \
        D3[4] = __builtin_amdgcn_readfirstlane(dl1[0]); D3[5] = __builtin_amdgcn_readfirstlane(dl1[1]);                              \
        int xu[4];                                                                                                                   \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) xu[k] = xr[k];                                                                 \
        /* ---- the small loads: descriptor of s + 4, list of s + 3, y row of s + 3's tile */                                        \
        {                                                                                                                            \
            const int di = min(max(s + 4, d_beg), d_end - 1);                                                                        \
            const int *dp = (const int *)(desc + di);                                                                                \
            dl0 = *(const v4i *)dp; dl1 = *(const v2i *)(dp + 4);                                                                    \
            _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                          \
                const int e = ((wave + 4 * k) * 64 + lane) >> 2;                                                                     \
                xr[k] = xrows[e < D3[3] ? D3[2] + e : 0];                                                                            \
            }                                                                                                                        \
            const long row = min((long)D3[5] * 64 + wave * 16 + r, row_last);                                                        \
            y3 = RM ? row_map[row] : (int)row;                         /* (every phase asks for its tile's y row: no select on a value in flight) */ \
        }                                                                                                                            \
        /* ---- the large ones: X rows and entries of phase s + 2 (descriptor D2, list xu) */                                        \
        {                                                                                                                            \
            unsigned char *buf = ps_smem + (((P) + 2) % 3) * 16384;                                                                  \
            _Pragma("unroll") for (int k = 0; k < 4; ++k)                                                                            \
                __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + (long)xu[k] * B + q * VW), (lds_void_t *)(buf + (wave + 4 * k) * 1024), 16, 0, 0); \
            const unsigned pk2 = (unsigned)D2[4];                                                                                    \
            const unsigned vo = (unsigned)(sel ? D2[1] : D2[0]) + (unsigned)hoff;                                                    \
            const int ngl = (int)((pk2 >> (4 * sel)) & 15u) + (((pk2 >> (8 + 2 * sel)) & 3u) ? 1 : 0);                               \
            _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                                          \
                const unsigned od = vo + (d < ngl ? d * 128u : 0u);                                                                  \
                ix[((P) + 2) % 3][d] = ld_stream<NT>(col8 + od);                                                                     \
                a[((P) + 2) % 3][d] = ld_stream<NT>(values + od);                                                                    \
            }                                                                                                                        \
        }                                                                                                                            \
        if (s >= d_beg) {                                                                                                            \
            const vec_t *xs = (const vec_t *)(ps_smem + (P) * 16384);                                                                \
            const unsigned pk0 = (unsigned)D0[4];                                                                                    \
            const int nf = (int)((pk0 >> (4 * sel)) & 15u), rem = (int)((pk0 >> (8 + 2 * sel)) & 3u);                                \
            _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                                          \
                if (d < nf) { PS_STEP(0, a[P][d], ix[P][d]) PS_STEP(1, a[P][d], ix[P][d]) PS_STEP(2, a[P][d], ix[P][d]) PS_STEP(3, a[P][d], ix[P][d]) } \
                else if (d == nf && rem) {                                                                                           \
                    PS_STEP(0, a[P][d], ix[P][d])                                                                                    \
                    if (rem > 1) PS_STEP(1, a[P][d], ix[P][d])                                                                       \
                    if (rem > 2) PS_STEP(2, a[P][d], ix[P][d])                                                                       \
                }                                                                                                                    \
            }                                                                                                                        \
            if (pk0 & PK_LAST) {                                                                                                     \
                if ((long)D0[5] * 64 + wave * 16 + r <= row_last && y0 < n_store) {                                                  \
                    if (YCOL) { _Pragma("unroll") for (int w = 0; w < VW; ++w) st_y<YNT>(Y + ((long)y0 + (long)(q * VW + w) * ld), acc[w]); } \
                    else *((vec_t *)(Y + (long)y0 * B) + q) = acc;                                                                   \
                }                                                                                                                    \
                _Pragma("unroll") for (int w = 0; w < VW; ++w) acc[w] = VT(0);                                                       \
            }                                                                                                                        \
        }                                                                                                                            \
        ++s;                                                                                                                         \
    }

    // iteration s asks for the descriptor of s + 4, the list of s + 3, the rows and entries of s + 2 and multiplies phase s: five iterations of
    // lead-in (their loads go to clamped, valid addresses; nothing is multiplied or stored before s = d_beg)
    int s = d_beg - 5;
    s -= ((s - d_beg) % 3 + 3) % 3;                                        // (phase d_beg + 3k in register set 0)
    while (s < d_end) {
        PS2_BODY(0)
        if (s >= d_end) break;
        PS2_BODY(1)
        if (s >= d_end) break;
        PS2_BODY(2)
    }
#undef PS2_BODY
#undef PS_STEP
    if (wg_clock && threadIdx.x == 0) wg_clock[gridDim.x + blockIdx.x] = (long long)wall_clock64();
}

template <typename VT, int B>
bool launch_pstream(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, hipStream_t st) {
    if (!A->ps_desc || !A->ps_wg_ptr || A->ps_grid <= 0 || A->part || A->C != 32 || !A->pb_idx8 || A->pb_max_rows > 256 || A->pb_ngp > 8) return false;
    const size_t lds = 2 * 16384;
    const bool per_tile = (int64_t)A->ps_grid == A->pb_n_tiles && A->ps_per_tile;
    // measurement aid: USPMV_STREAM_CLOCK=<file> -- every workgroup's start and end time (100 MHz counter) of THIS launch, written as text
    static const char *clock_file = getenv("USPMV_STREAM_CLOCK");
    long long *d_clock = nullptr;
    if (clock_file && hipMalloc((void **)&d_clock, 16 * (size_t)A->ps_grid) != hipSuccess) d_clock = nullptr;
    struct ClockOut {
        long long *d; int G; hipStream_t st; const char *file;
        ~ClockOut() {
            if (!d) return;
            std::vector<long long> h(2 * (size_t)G);
            if (hipStreamSynchronize(st) == hipSuccess && hipMemcpy(h.data(), d, 16 * (size_t)G, hipMemcpyDeviceToHost) == hipSuccess) {
                if (FILE *f = fopen(file, "w")) {
                    long long t0 = h[0];
                    for (int w = 0; w < G; ++w) t0 = std::min(t0, h[(size_t)w]);
                    for (int w = 0; w < G; ++w) fprintf(f, "%d %lld %lld\n", w, h[(size_t)w] - t0, h[(size_t)G + w] - t0);
                    fclose(f);
                }
            }
            (void)hipFree(d);
        }
    } clock_out{d_clock, A->ps_grid, st, clock_file};
#define PS_ARGS (const PhDesc *)A->ps_desc, (const int *)A->ps_wg_ptr, (const VT *)A->pb_values, (const unsigned char *)A->pb_col16, (const int *)A->pb_xrows, X, Y, ld, \
                (long)(A->n_chunks * A->C), (long)A->n_store, (const int *)A->bt_row_map, d_clock, per_tile ? g_tune.xcd_remap : 0
#define PS_LAUNCH(NTV, YC, YN, AB) hipLaunchKernelGGL((scs_spmmv_pstream<VT, B, NTV, YC, YN, AB>), dim3((unsigned)A->ps_grid), dim3(256), lds, st, PS_ARGS)
    if constexpr (sizeof(VT) == 8) {
        if (g_tune.ablate >= 1 && !ycol) {          // measurement only (results wrong by construction)
            switch (g_tune.ablate) {
                case 1: PS_LAUNCH(true, false, true, 1); return true;
                case 2: PS_LAUNCH(true, false, true, 2); return true;
                case 4: PS_LAUNCH(true, false, true, 4); return true;
                case 12: PS_LAUNCH(true, false, true, 12); return true;
                case 13: PS_LAUNCH(true, false, true, 13); return true;
                case 3: PS_LAUNCH(true, false, true, 3); return true;
                default: break;
            }
        }
    }
    const bool nt = g_tune.nontemporal != 0, ynt = nt && (!ycol || g_tune.spmmv_ycol_nt);
    if (g_tune.spmmv_stream_depth >= 2) {
        const size_t lds3 = 3 * 16384;
#define PS2_LAUNCH(NTV, YC, YN) do { if (A->bt_row_map) hipLaunchKernelGGL((scs_spmmv_pstream2<VT, B, NTV, YC, YN, true>), dim3((unsigned)A->ps_grid), dim3(256), lds3, st, PS_ARGS); \
                                     else hipLaunchKernelGGL((scs_spmmv_pstream2<VT, B, NTV, YC, YN, false>), dim3((unsigned)A->ps_grid), dim3(256), lds3, st, PS_ARGS); } while (0)
        if (nt) {
            if (ycol) { if (ynt) PS2_LAUNCH(true, true, true); else PS2_LAUNCH(true, true, false); }
            else PS2_LAUNCH(true, false, true);
        } else {
            if (ycol) PS2_LAUNCH(false, true, false); else PS2_LAUNCH(false, false, false);
        }
#undef PS2_LAUNCH
        return true;
    }
    if (nt && g_tune.spmmv_stream_waves >= 5) {
#define PS_LAUNCH5(NTV, YC, YN) hipLaunchKernelGGL((scs_spmmv_pstream<VT, B, NTV, YC, YN, 0, 5>), dim3((unsigned)A->ps_grid), dim3(256), lds, st, PS_ARGS)
        if (ycol) { if (ynt) PS_LAUNCH5(true, true, true); else PS_LAUNCH5(true, true, false); }
        else PS_LAUNCH5(true, false, true);
#undef PS_LAUNCH5
    } else if (nt) {
        if (ycol) { if (ynt) PS_LAUNCH(true, true, true, 0); else PS_LAUNCH(true, true, false, 0); }
        else PS_LAUNCH(true, false, true, 0);
    } else {
        if (ycol) PS_LAUNCH(false, true, false, 0); else PS_LAUNCH(false, false, false, 0);
    }
#undef PS_LAUNCH
#undef PS_ARGS
    return true;
}

}  // namespace

namespace uspmv_dev {

void dmat_stream_release(uspmv_dmat *A) {
    (void)hipFree(A->ps_desc); (void)hipFree(A->ps_wg_ptr);
    A->ps_desc = nullptr; A->ps_wg_ptr = nullptr; A->ps_grid = 0; A->ps_n_desc = 0; A->ps_per_tile = false;
}

// The flat schedule of the handle's phased plan for `wgs_per_cu` persistent workgroups per CU: workgroup w walks tiles w, w + G, w + 2G, ...
// (the whole grid moves through the matrix as one front, like the one-tile-per-workgroup launch), its descriptors contiguous.
int dmat_stream_schedule(uspmv_dmat *A, int wgs_per_cu) {
    dmat_stream_release(A);
    if (!A->pb || A->C != 32 || !A->pb_idx8 || A->pb_n_tiles <= 0 || wgs_per_cu <= 0) return USPMV_OK;
    int dev = 0, cus = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t nt = A->pb_n_tiles;
    const bool per_tile = wgs_per_cu >= 99 && nt <= INT32_MAX;          // one tile per workgroup: only the phases of a tile are pipelined
    const int G = per_tile ? (int)nt : (int)std::min<int64_t>((int64_t)std::max(cus, 1) * std::min(wgs_per_cu, 5), nt);
    std::vector<int32_t> php((size_t)nt + 1), slot((size_t)nt), wgp((size_t)G + 1, 0);
    HIP_TRY(hipMemcpy(php.data(), A->pb_ph_ptr, 4 * ((size_t)nt + 1), hipMemcpyDeviceToHost));
    // workgroup w runs on XCD w % 8 (round-robin dispatch): with "spmmv_stream_xcd" the G / 8 workgroups of an XCD take CONSECUTIVE tiles of
    // every super-block of G tiles -- neighbouring tiles share most of their X rows, which then meet in one L2 -- otherwise tile t goes to
    // workgroup t % G
    const bool by_xcd = !per_tile && g_tune.spmmv_stream_xcd && G % 8 == 0 && G >= 16;
    const int per = G / 8;
    int64_t pos = 0;
    for (int w = 0; w < G; ++w) {
        wgp[(size_t)w] = (int32_t)pos;
        const int64_t first = by_xcd ? (int64_t)(w % 8) * per + w / 8 : w;
        for (int64_t t = first; t < nt; t += G) {
            slot[(size_t)t] = (int32_t)pos;
            pos += std::max(php[(size_t)t + 1] - php[(size_t)t], 1);
        }
        if (pos > INT32_MAX - 8) return USPMV_OK;
    }
    wgp[(size_t)G] = (int32_t)pos;
    int32_t *d_slot = nullptr;
    hipError_t e = hipMalloc((void **)&d_slot, 4 * (size_t)nt);
    if (e == hipSuccess) e = hipMemcpy(d_slot, slot.data(), 4 * (size_t)nt, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc((void **)&A->ps_wg_ptr, 4 * ((size_t)G + 1));
    if (e == hipSuccess) e = hipMemcpy(A->ps_wg_ptr, wgp.data(), 4 * ((size_t)G + 1), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&A->ps_desc, sizeof(PhDesc) * (size_t)pos);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(stream_desc_fill, dim3(grid_for(nt, 256)), dim3(256), 0, nullptr, (long)nt, (long)A->n_chunks, A->chunk_lengths,
                           (const unsigned *)A->pb_c16_ptrs, (const int *)A->pb_ph_ptr, (const int *)A->pb_g0, (const int *)A->pb_list_ptr, (const int *)d_slot, (PhDesc *)A->ps_desc);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    (void)hipFree(d_slot);
    if (e != hipSuccess) {
        dmat_stream_release(A);
        return uspmv::fail(USPMV_ERR_HIP, "uspmv_dmat_optimize_block: stream schedule: %s", hipGetErrorString(e));
    }
    A->ps_grid = G; A->ps_n_desc = pos; A->ps_per_tile = per_tile;
    return USPMV_OK;
}

bool spmmv_stream(const uspmv_dmat *A, const double *X, double *Y, long ld, bool ycol, hipStream_t st) { return launch_pstream<double, 8>(A, X, Y, ld, ycol, st); }
bool spmmv_stream(const uspmv_dmat *A, const float *X, float *Y, long ld, bool ycol, hipStream_t st) { return launch_pstream<float, 16>(A, X, Y, ld, ycol, st); }

}  // namespace uspmv_dev
