// SpMMV over the PHASED block plan (64-byte X rows: dp b = 8, sp b = 16): four lanes per row, one tile per workgroup
// (scs_spmmv_quadph) or persistent workgroups walking a flat schedule of the same phases (scs_spmmv_quadpp).
// Reference loop: code/kernels.hpp:277-333 (block_colwise / block_rowwise SELL-C-sigma SpMMV); same slot order per row.
#include "uspmv_device.hpp"
#include <vector>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

using namespace uspmv_dev;

namespace {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_cvoid_t;

// Four lanes per row over the PHASED block plan (uspmv_build_phased_plan): the arithmetic of scs_spmmv_quad, but a tile's slots are
// processed phase by phase -- at most NGP groups of four slots whose entries touch at most MAXP*64 distinct X rows -- and only the
// current phase's X rows are staged: 16 KB of LDS and 40 registers per 256-thread workgroup instead of 50 KB and 88, so EIGHT
// workgroups (the wave limit) instead of three share a CU and one workgroup's round trips (list, X rows + its matrix entries) hide
// behind the arithmetic of seven others.  Every row still walks its slots in order: bit-identical FMA chains.
// XCOL: X is the caller's COLUMN-MAJOR block vector (X[col + v*ld]) -- no re-layout pass, no workspace: a phase's X rows are
// assembled in LDS by the workgroup itself (thread <-> list entry: B coalesced element loads, one per column, written as one
// row of LDS; rows sit 16 bytes apart from a power-of-two stride so that the column-strided writes spread over the banks).
// XM == 2 (XLINE): X is the caller's COLUMN-MAJOR block vector and the plan's lists hold LINES of 128 bytes (16 doubles / 32 floats of
// one column; uspmv_build_phased_plan with line_shift): per listed line and column one whole 128-byte line travels from L2 into
// LDS by DMA -- the same number of DMA instructions and bytes as the row-major form, but no re-layout pass over X and no
// workspace.  LDS layout [line][column][128 bytes], the eight 16-byte pieces of a (line, column) block XOR-swizzled by the
// column's quarter (free: a DMA lane's GLOBAL address is its own), so that the four lanes of a row -- same X row, four column
// pairs -- hit four different bank groups.  An entry's 8-bit local index is line << LS | row-in-line; its LDS byte offset
// (line << (7 + log2 B)) | (row-in-line * sizeof(VT)) is computed once per loaded index BEFORE the quad broadcast, the lane's
// column / swizzle bits are XORed in after it, and the VW elements of the lane's columns sit 128 bytes apart (ds_read2).
// ABL (measurement only, results wrong by construction), bits: 1 no X staging, 2 no arithmetic, 4 no value loads, 8 no index loads, 16 no list
template <typename VT, typename IT, int B, bool NT, bool YCOL, int C, int NGP, int MAXP, int XM, int ABL = 0, bool YNT = NT>
__global__ void __launch_bounds__(256) scs_spmmv_quadph(const long n_chunks, const int *__restrict__ chunk_ptrs,
        const int *__restrict__ chunk_lengths, const VT *__restrict__ values, const VT *__restrict__ X, VT *__restrict__ Y, const long ld,
        const int *__restrict__ ph_ptr, const int *__restrict__ ph_g0, const int *__restrict__ ph_list_ptr, const int *__restrict__ xrows,
        const unsigned *__restrict__ c16_ptrs, const IT *__restrict__ col16, const int xcd_remap, const long n_store,
        const int *__restrict__ row_map) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tlc_smem[];
    constexpr int VW = 16 / (int)sizeof(VT);
    constexpr bool XCOL = XM == 1, XLINE = XM == 2;
    constexpr int LS = sizeof(VT) == 8 ? 4 : 5;                  // XLINE: log2 of the X rows per 128-byte line
    constexpr int LB = B == 8 ? 3 : 4, LVS = sizeof(VT) == 8 ? 3 : 2, LVW = sizeof(VT) == 8 ? 1 : 2;
    static_assert(B == 4 * VW, "four 16-byte pieces per X row");
    static_assert(!XLINE || ((1 << LB) == B && (1 << LVW) == VW), "XLINE shapes");
    typedef VT vec_t __attribute__((ext_vector_type(VW)));
    const unsigned tile = remap_block(blockIdx.x, gridDim.x, xcd_remap);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int r = lane >> 2, q = lane & 3;
    const long row = (long)tile * 64 + wave * 16 + r;
    const long c = row / C;
    const int i = (int)(row - c * C);
    const bool valid = c < n_chunks;
    int cs = 0, L = 0;
    unsigned q0 = 0;
    if (valid) { cs = chunk_ptrs[c]; L = chunk_lengths[c]; q0 = c16_ptrs[c]; }
    L = __builtin_amdgcn_readfirstlane(L);
    if (L < 0) return;                                           // USPMV_SKIP_LEN: the tile belongs to the other part of a two-part SpMMV (all of its chunks do)
    const int ngf = L >> 2, rem = L & 3;
    const int p0 = ph_ptr[tile], p1 = ph_ptr[tile + 1];
    vec_t acc;
#pragma unroll
    for (int w = 0; w < VW; ++w) acc[w] = VT(0);
    // the plan's own copy of the entries is GROUP-major like the indices ([chunk][group of four slots][row][slot % 4]): the four
    // lanes of a row read 32 contiguous bytes and a wave 512 -- four cache lines per load instead of the sixteen (one per lane
    // quad and slot) a column-major chunk costs, which is what the L1 tag rate could not take (profiles/r02/spmmv_phase_ablation.txt)
    const VT *vp = values + q0 + (long)i * 4 + q;
    const IT *ip = col16 + q0 + (long)i * 4 + q;             // (IT: phase-local index type, 8 bits when no phase lists more than 256 rows)
    const vec_t *xs = (const vec_t *)tlc_smem;
    constexpr unsigned RS = XCOL ? 5u : 4u;                  // 16-byte pieces per staged row (XCOL: 80-byte row stride)
    const unsigned lanec = ((unsigned)q << (7 + LVW)) | ((unsigned)q << 5);   // XLINE: the lane's column pair / quad + its swizzle
#define QUAD_STEP(UU, AV, IV)                                                                                 \
    {                                                                                                         \
        const VT aa = quad_bcast<UU>(AV);                                                                     \
        const unsigned li = (unsigned)quad_bcast<UU>((int)(IV));                                              \
        vec_t xv;                                                                                             \
        if constexpr (XLINE) {                                                                                \
            const unsigned char *xb = tlc_smem + (li ^ lanec);                                                \
            _Pragma("unroll") for (int w = 0; w < VW; ++w) xv[w] = *(const VT *)(xb + w * 128);               \
        } else xv = xs[li * RS + (unsigned)q];                                                                \
        _Pragma("unroll") for (int w = 0; w < VW; ++w) acc[w] = fma_t(aa, xv[w], acc[w]);                     \
    }
    // XLINE: local index (line << LS | row in line) -> LDS byte offset of (line, column 0, row in line)
    auto lds_off = [](unsigned ix) -> unsigned { return XLINE ? (((ix >> LS) << (7 + LB)) | ((ix & ((1u << LS) - 1u)) << LVS)) : ix; };
    int g0 = p0 < p1 ? ph_g0[p0] : 0, lp = p0 < p1 ? ph_list_ptr[p0] : 0;
    // (Built and measured level in round 4, then removed: the X-row list of phase ph + 1 requested together with the X rows and entries of
    //  phase ph -- one dependent round trip per phase instead of two, at 72 instead of 64 registers: 0.7468 against 0.7486 ms.  Neither the
    //  list's round trip nor the bytes the X staging takes from the fabric -- halved by a tile -> XCD group of plane/8, PMC-confirmed --
    //  bound this kernel: profiles/r04/spmmv_listahead_timing.txt, pmc_xcd_cfg3/summary.txt.)
    for (int ph = p0; ph < p1; ++ph) {
        const int g1 = ph + 1 < p1 ? ph_g0[ph + 1] : 0x7fffffff;
        const int lp1 = ph_list_ptr[ph + 1];
        const int np = (lp1 - lp) << 2;
        if constexpr (XLINE) {
            // ---- lines of the phase -> LDS: piece pp = (wave + 4k)*64 + lane = ((line in list * B + column) * 8 + slot); slot j holds
            //      the line's piece j ^ swizzle(column)
            const int npl = (lp1 - lp) << (3 + LB);
            long ge[MAXP];
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                const int pp = (wave + 4 * k) * 64 + lane;
                ge[k] = -1;
                if (!(ABL & 16) && pp < npl) {
                    const int v = (pp >> 3) & (B - 1);
                    const int gj = (pp & 7) ^ ((v >> LVW) << 1);
                    ge[k] = ((long)xrows[lp + (pp >> (3 + LB))] << LS) + (long)gj * VW + (long)v * ld;   // element index of the piece in X
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ph > p0) __syncthreads();                    // every wave is through with the previous phase's lines
#pragma unroll
            for (int k = 0; k < MAXP; ++k)
                if (!(ABL & 1) && ge[k] >= 0) {
                    const int pp = (wave + 4 * k) * 64 + lane;
                    const long v = (long)((pp >> 3) & (B - 1));
                    if (ge[k] - v * ld + VW <= ld)           // the piece lies inside its column
                        __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + ge[k]), (lds_void_t *)(tlc_smem + (wave + 4 * k) * 1024), 16, 0, 0);
                    else {
#pragma unroll
                        for (int w = 0; w < VW; ++w)
                            if (ge[k] - v * ld + w < ld) ((VT *)(tlc_smem + (size_t)pp * 16))[w] = X[ge[k] + w];
                    }
                }
        } else if constexpr (!XCOL) {
            // ---- the list entries this lane needs for its DMA pieces (piece p = (wave + 4k)*64 + lane <-> list entry p >> 2)
            int xr[MAXP];
#pragma unroll
            for (int k = 0; k < MAXP; ++k) {
                const int pp = (wave + 4 * k) * 64 + lane;
                xr[k] = -1;
                if (!(ABL & 16) && pp < np) xr[k] = xrows[lp + (pp >> 2)];
                // measurement only (ablate 64): the same number of X rows, but CONSECUTIVE ones -- whole 128-byte lines, 1 KiB per wave
                // instruction -- instead of the list's: what the staging would cost if only its bytes counted, not its requests
                if ((ABL & 64) && pp < np) xr[k] = (int)(((long)tile * 97 + (pp >> 2)) % (ld > 1024 ? ld - 1024 : 1));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (ph > p0) __syncthreads();                    // every wave is through with the previous phase's rows
            // ---- X rows of the phase -> LDS by DMA, and behind them this wave's matrix entries of the phase
#pragma unroll
            for (int k = 0; k < MAXP; ++k)
                if (!(ABL & 1) && xr[k] >= 0)
                    __builtin_amdgcn_global_load_lds((glb_cvoid_t *)(X + (long)xr[k] * B + q * VW), (lds_void_t *)(tlc_smem + (wave + 4 * k) * 1024), 16, 0, 0);
        } else {
            // ---- column-major X: thread <-> list entry (MAXP*64/256 entries per thread), its B elements loaded column by column
            constexpr int EPT = MAXP / 4;                    // list entries per thread (cap rows / 256 threads)
            int xe[EPT];
            VT xv_[EPT][B];
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const int e = k * 256 + (int)threadIdx.x;
                xe[k] = -1;
                if (e < (np >> 2)) xe[k] = xrows[lp + e];
            }
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (xe[k] >= 0) {
#pragma unroll
                    for (int v = 0; v < B; ++v) xv_[k][v] = X[(long)xe[k] + (long)v * ld];
                }
            if (ph > p0) __syncthreads();                    // every wave is through with the previous phase's rows
#pragma unroll
            for (int k = 0; k < EPT; ++k)
                if (xe[k] >= 0) {
                    VT *dst = (VT *)(tlc_smem + (size_t)(k * 256 + (int)threadIdx.x) * (RS * 16));
#pragma unroll
                    for (int v = 0; v < B; ++v) dst[v] = xv_[k][v];
                }
        }
        const int ge = min(g1, ngf);                         // full groups of this wave's rows in the phase: [g0, ge)
        VT a[NGP], at = VT(0);
        unsigned ix[NGP], ixt = 0u;
        const bool tail_here = rem && ngf >= g0 && ngf < g1;  // the partial last group of this wave's rows belongs to this phase
        if constexpr ((ABL & 32) != 0) {
            // measurement only (ablate 32): the n = ge - g0 <= NGP full groups in register slots NGP - n .. NGP - 1, one uniform switch
            // into a straight line of NGP blocks instead of a compare + branch + zero fill per group.  Same chains (ascending slots
            // = ascending groups).  78 instead of 64 registers and MORE scalar instructions (a branch tree, no jump table): see
            // profiles/r03/config3_colwise.txt, block 5.
            const int skip = NGP - max(min(ge - g0, NGP), 0);
            const VT *vpg = vp + (long)(g0 - skip) * 4 * C;      // (never dereferenced below slot `skip`)
            const IT *ipg = ip + (long)(g0 - skip) * 4 * C;
#define PH_LOAD(K) { ix[K] = lds_off(ld_stream<NT>(ipg + (K) * 4 * C)); a[K] = ld_stream<NT>(vpg + (K) * 4 * C); }
            static_assert(NGP == 8, "eight slots per phase");
            switch (skip) {
                case 0: PH_LOAD(0) [[fallthrough]];
                case 1: PH_LOAD(1) [[fallthrough]];
                case 2: PH_LOAD(2) [[fallthrough]];
                case 3: PH_LOAD(3) [[fallthrough]];
                case 4: PH_LOAD(4) [[fallthrough]];
                case 5: PH_LOAD(5) [[fallthrough]];
                case 6: PH_LOAD(6) [[fallthrough]];
                case 7: PH_LOAD(7) [[fallthrough]];
                default: break;
            }
#undef PH_LOAD
            if (tail_here) { ixt = lds_off(ld_stream<NT>(ip + (long)ngf * 4 * C)); if (q < rem) at = ld_stream<NT>(vp + (long)ngf * 4 * C); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#define PH_STEPS(K) { QUAD_STEP(0, a[K], ix[K]) QUAD_STEP(1, a[K], ix[K]) QUAD_STEP(2, a[K], ix[K]) QUAD_STEP(3, a[K], ix[K]) }
            switch (skip) {
                case 0: PH_STEPS(0) [[fallthrough]];
                case 1: PH_STEPS(1) [[fallthrough]];
                case 2: PH_STEPS(2) [[fallthrough]];
                case 3: PH_STEPS(3) [[fallthrough]];
                case 4: PH_STEPS(4) [[fallthrough]];
                case 5: PH_STEPS(5) [[fallthrough]];
                case 6: PH_STEPS(6) [[fallthrough]];
                case 7: PH_STEPS(7) [[fallthrough]];
                default: break;
            }
#undef PH_STEPS
        } else {
#pragma unroll
            for (int d = 0; d < NGP; ++d) {
                a[d] = VT(0); ix[d] = 0u;
                if (g0 + d < ge) {
                    if (!(ABL & 8)) ix[d] = lds_off(ld_stream<NT>(ip + (long)(g0 + d) * 4 * C));
                    if (!(ABL & 4)) a[d] = ld_stream<NT>(vp + (long)(g0 + d) * 4 * C);
                }
            }
            if (tail_here) { ixt = lds_off(ld_stream<NT>(ip + (long)ngf * 4 * C)); if (q < rem) at = ld_stream<NT>(vp + (long)ngf * 4 * C); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll
            for (int d = 0; d < NGP; ++d) {
                if (ABL & 2) { if (ix[d] == 0xdeadbeefu) acc[0] += a[d]; continue; }
                if (g0 + d < ge) { QUAD_STEP(0, a[d], ix[d]) QUAD_STEP(1, a[d], ix[d]) QUAD_STEP(2, a[d], ix[d]) QUAD_STEP(3, a[d], ix[d]) }
            }
        }
        if (tail_here) {
            if (rem > 0) QUAD_STEP(0, at, ixt)
            if (rem > 1) QUAD_STEP(1, at, ixt)
            if (rem > 2) QUAD_STEP(2, at, ixt)
        }
        g0 = g1; lp = lp1;
    }
#undef QUAD_STEP
    if (!valid) return;
    const long yrow = row_map ? (long)row_map[row] : row;
    if (yrow >= n_store) return;
    if (YCOL) {
#pragma unroll
        for (int w = 0; w < VW; ++w) st_y<YNT>(Y + (yrow + (long)(q * VW + w) * ld), acc[w]);
    } else {
        *((vec_t *)(Y + yrow * B) + q) = acc;
    }
}

template <typename VT, typename IT, int B, int CT, int MAXP>
void launch_spmmv_quadph_m(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, int xmode, hipStream_t st) {
    const bool xcol = xmode == 1;
    const size_t lds = xcol ? (size_t)MAXP * 64 * 80 : (size_t)MAXP * 4 * 1024;   // MAXP*64 rows of 64 (row-major X / lines, DMA pieces) or 80 bytes
#define QH_ARGS(PH, G0, LP, XR, C16) (long)A->n_chunks, A->chunk_ptrs, part_lengths(A, 1), (const VT *)A->pb_values, X, Y, ld, PH, G0, LP, XR, A->pb_c16_ptrs, \
                           (const IT *)C16, g_tune.xcd_remap, (long)A->n_store, (const int *)A->bt_row_map
#define QH_LAUNCH(NTV, YC)                                                                                              \
    do {                                                                                                                \
        auto kfn = xcol ? scs_spmmv_quadph<VT, IT, B, NTV, YC, CT, 8, MAXP, 1> : scs_spmmv_quadph<VT, IT, B, NTV, YC, CT, 8, MAXP, 0>; \
        if (YC && NTV && !xcol && !g_tune.spmmv_ycol_nt) kfn = scs_spmmv_quadph<VT, IT, B, NTV, YC, CT, 8, MAXP, 0, 0, false>;            \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(kfn, dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st, QH_ARGS(A->pb_ph_ptr, A->pb_g0, A->pb_list_ptr, A->pb_xrows, A->pb_col16)); \
    } while (0)
    if constexpr (MAXP == 4 && sizeof(IT) == 1) {
        if (xmode == 3) {   // row-major workspace in ORIGINAL row numbering (the re-layout pass undid the sigma permutation): the handle's third plan
            if (g_tune.nontemporal && g_tune.spmmv_ycol_nt)
                hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, true, true, CT, 8, MAXP, 0>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st,
                                   QH_ARGS(A->pu_ph_ptr, A->pu_g0, A->pu_list_ptr, A->pu_xrows, A->pu_col8));
            else if (g_tune.nontemporal)
                hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, true, true, CT, 8, MAXP, 0, 0, false>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st,
                                   QH_ARGS(A->pu_ph_ptr, A->pu_g0, A->pu_list_ptr, A->pu_xrows, A->pu_col8));
            else
                hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, false, true, CT, 8, MAXP, 0>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st,
                                   QH_ARGS(A->pu_ph_ptr, A->pu_g0, A->pu_list_ptr, A->pu_xrows, A->pu_col8));
            return;
        }
        if (xmode == 2) {   // column-major X staged by lines (the handle's second phased plan); Y column-major as well
            if (g_tune.nontemporal && g_tune.spmmv_ycol_nt)
                hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, true, true, CT, 8, MAXP, 2>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st,
                                   QH_ARGS(A->pl_ph_ptr, A->pl_g0, A->pl_list_ptr, A->pl_lines, A->pl_col8));
            else if (g_tune.nontemporal)
                hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, true, true, CT, 8, MAXP, 2, 0, false>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st,
                                   QH_ARGS(A->pl_ph_ptr, A->pl_g0, A->pl_list_ptr, A->pl_lines, A->pl_col8));
            else
                hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, false, true, CT, 8, MAXP, 2>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st,
                                   QH_ARGS(A->pl_ph_ptr, A->pl_g0, A->pl_list_ptr, A->pl_lines, A->pl_col8));
            return;
        }
    }
    if constexpr (sizeof(VT) == 8 && CT == 32 && MAXP == 4 && sizeof(IT) == 1) {
        if (g_tune.ablate >= 1 && !xcol && !ycol) {   // measurement only
#define QH_ABL(N) case N: hipLaunchKernelGGL((scs_spmmv_quadph<VT, IT, B, true, false, CT, 8, MAXP, 0, N>), dim3((unsigned)A->pb_n_tiles), dim3(256), lds, st, \
                           QH_ARGS(A->pb_ph_ptr, A->pb_g0, A->pb_list_ptr, A->pb_xrows, A->pb_col16)); break;
            switch (g_tune.ablate) { QH_ABL(1) QH_ABL(2) QH_ABL(4) QH_ABL(8) QH_ABL(17) QH_ABL(14) QH_ABL(3) QH_ABL(19) QH_ABL(32) QH_ABL(64) QH_ABL(78) default: break; }
#undef QH_ABL
            return;
        }
    }
    if (g_tune.nontemporal) { if (ycol) QH_LAUNCH(true, true); else QH_LAUNCH(true, false); }
    else { if (ycol) QH_LAUNCH(false, true); else QH_LAUNCH(false, false); }
#undef QH_LAUNCH
#undef QH_ARGS
}

// false: the handle's phased plan does not fit the compiled shapes.  xmode: 0 = X row-major, 1 = column-major X assembled through
// registers (measured slower, kept as "spmmv_xcol" 1), 2 = column-major X staged by 128-byte lines (needs the handle's line plan)
template <typename VT, int B>
bool launch_spmmv_quadph(const uspmv_dmat *A, const VT *X, VT *Y, long ld, bool ycol, int xmode, hipStream_t st) {
    if (!A->pb || !A->pb_values || A->pb_ngp > 8 || !part_ok(A, 1)) return false;
    if (A->part && xmode != 0) return false;                   // (the two-part form runs on the row-major plan only)
    if (xmode == 3) {
        if (!A->pu || !ycol || !A->pu_col8 || A->pu_max_rows > 256) return false;
#define QU_C(CTV) launch_spmmv_quadph_m<VT, unsigned char, B, CTV, 4>(A, X, Y, ld, ycol, 3, st)
        if (A->C == 32) QU_C(32); else if (A->C == 64) QU_C(64); else if (A->C == 16) QU_C(16); else return false;
#undef QU_C
        return true;
    }
    if (xmode == 2) {
        constexpr int VW = 16 / (int)sizeof(VT);
        if (!A->pl || !ycol || !A->pl_col8 || A->pl_max_rows > 256 || ld % VW != 0 || ((uintptr_t)X % 16) != 0) return false;
#define QL_C(CTV) launch_spmmv_quadph_m<VT, unsigned char, B, CTV, 4>(A, X, Y, ld, ycol, 2, st)
        if (A->C == 32) QL_C(32); else if (A->C == 64) QL_C(64); else if (A->C == 16) QL_C(16); else return false;
#undef QL_C
        return true;
    }
    const int pieces = (A->pb_max_rows * 4 + 255) / 256;
#define QH_C(CTV) do { if (pieces <= 4 && A->pb_idx8) launch_spmmv_quadph_m<VT, unsigned char, B, CTV, 4>(A, X, Y, ld, ycol, xmode, st); \
        else if (pieces <= 4) launch_spmmv_quadph_m<VT, unsigned short, B, CTV, 4>(A, X, Y, ld, ycol, xmode, st); \
        else if (pieces <= 8) launch_spmmv_quadph_m<VT, unsigned short, B, CTV, 8>(A, X, Y, ld, ycol, xmode, st); else return false; } while (0)
    if (A->C == 32) QH_C(32); else if (A->C == 64) QH_C(64); else if (A->C == 16) QH_C(16); else return false;
#undef QH_C
    return true;
}

}  // namespace

namespace uspmv_dev {

bool spmmv_phased(const uspmv_dmat *A, const double *X, double *Y, long ld, bool ycol, int xmode, hipStream_t st) {
    if (xmode == 0 && g_tune.spmmv_stream > 0 && A->ps_desc && spmmv_stream(A, X, Y, ld, ycol, st)) return true;
    return launch_spmmv_quadph<double, 8>(A, X, Y, ld, ycol, xmode, st);
}
bool spmmv_phased(const uspmv_dmat *A, const float *X, float *Y, long ld, bool ycol, int xmode, hipStream_t st) {
    if (xmode == 0 && g_tune.spmmv_stream > 0 && A->ps_desc && spmmv_stream(A, X, Y, ld, ycol, st)) return true;
    return launch_spmmv_quadph<float, 16>(A, X, Y, ld, ycol, xmode, st);
}

}  // namespace uspmv_dev
