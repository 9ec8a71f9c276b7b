// Column-window sweep plan: an MI355X-specific device layout for matrices whose rows are wide and irregular
// (HV15R-class, SURVEY.md 8(d): 140 entries scattered over a +-50 000 band).  A 256-row tile of such a matrix
// touches every x line of its 100 000-column window, so the tile-local-column plan (tlc_plan.cpp) stages nothing
// and the gather kernel pays one 64-byte L2 sector per 8-byte operand.  Nothing like this exists in the reference;
// the SCS arrays of the caller stay untouched, this is a private re-layout of the same entries.
//
// Idea: x is cut into WINDOWS of W = 2^wlog consecutive elements (a global grid: window of column c = c >> wlog).
// A tile = tile_rows consecutive rows = one workgroup (above 1 024 rows a lane owns several rows).  If, in every row of the tile, the window index never
// decreases from one slot to the next (true whenever a row's entries are column-sorted, e.g. a general-pattern
// MatrixMarket file or the generators of gen_matrix.cpp; sigma-sorting permutes columns only inside aligned
// sigma blocks, so windows that are multiples of sigma keep the property), the workgroup can SWEEP the windows in
// ascending order -- stage window s of x in LDS with coalesced 16-byte loads, let every lane run the entries of
// its row that fall into window s, go on to s+1 -- and every row still sees its entries in slot order: the
// same FMA chain as scs_impl_cpu (code/kernels.hpp:216-258), bit for bit.
//
// Rows of one wave have different numbers of entries in a window.  To keep the matrix stream free of padding, the
// entries of a wave are stored COMPACTED: for window s and round k only the lanes with more than k entries in
// that window own an element, lanes ascending; a lane finds its element at base + (number of active lanes below
// it), and base advances by the number of active lanes -- both from one ballot.  The stream is then exactly
// nnz * (sizeof(VT) + 2) bytes, contiguous per wave, plus one count byte per (row, window).
//
// Trailing padding of a row (value +0, one repeated column; code/utilities.hpp:1991-2002) is not stored: k >= 1
// applications of acc = fma(+0, x[c], acc) equal one application (the second adds the same signed zero again, or
// meets the NaN the first one made), so the kernel applies it once per row that had any (pad_col >= 0).
#include <algorithm>
#include <cstring>

#include "uspmv_internal.hpp"

namespace {

template <typename VT>
inline bool is_pos_zero(VT v) {
    if (sizeof(VT) == 8) { uint64_t b; std::memcpy(&b, &v, 8); return b == 0; }
    uint32_t b; std::memcpy(&b, &v, 4); return b == 0;
}

// effective length of a row: the chunk length minus the trailing run of (+0, same column as the last slot)
template <typename VT>
inline int64_t effective_len(const int32_t *ci, const VT *va, int64_t cs, int64_t L, int64_t i, int64_t C, int32_t *pad_col) {
    *pad_col = -1;
    if (L == 0) return 0;
    const int32_t pc = ci[cs + (L - 1) * C + i];
    int64_t le = L;
    while (le > 0 && ci[cs + (le - 1) * C + i] == pc && is_pos_zero(va[cs + (le - 1) * C + i])) --le;
    if (le < L) *pad_col = pc;
    return le;
}

}  // namespace

int uspmv_build_sweep_plan(const uspmv_scs *s, const uspmv_scs *s2, int wlog, int tile_rows, double max_stage_bytes_per_nnz,
                           uspmv_sweep_plan *p) {
    p->valid = false;
    const int64_t C = s->C, nc = s->n_chunks;
    if (tile_rows != 256 && tile_rows != 512 && tile_rows != 1024 && tile_rows != 2048 && tile_rows != 4096) tile_rows = 1024;
    if (C < 1 || C > 64 || 64 % C != 0 || nc < 1) return USPMV_OK;          // a wave covers whole chunks
    if (s2 && (s2->C != C || s2->n_chunks != nc)) return USPMV_OK;
    if (wlog < 8 || wlog > 16) return USPMV_OK;                              // 16-bit local indices
    if (s->n_elements > (int64_t)UINT32_MAX || (s2 && s2->n_elements > (int64_t)UINT32_MAX)) return USPMV_OK;
    const int64_t R = tile_rows, n_pad = nc * C, n_tiles = (n_pad + R - 1) / R, wpt = R / 64;
    const int ns = s2 ? 2 : 1;
    const uspmv_scs *ss[2] = {s, s2};
    const size_t vsz = s->dtype == USPMV_F64 ? 8 : 4;
    p->tile_rows = tile_rows; p->wlog = wlog; p->n_tiles = n_tiles;

    // ---- pass 1: which tiles sweep, their window range, entries per wave
    std::vector<int32_t> smin((size_t)n_tiles, 0), S((size_t)n_tiles, 0);
    std::vector<char> ok((size_t)n_tiles, 0);
    std::vector<int64_t> wave_n[2];
    for (int w = 0; w < ns; ++w) wave_n[w].assign((size_t)(n_tiles * wpt), 0);
    int32_t max_col = 0;
#pragma omp parallel
    {
        int32_t my_max = 0;
#pragma omp for schedule(dynamic, 8)
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int64_t q0 = t * R, q1 = std::min(q0 + R, n_pad);
            int32_t lo = INT32_MAX, hi = -1;
            bool good = true;
            int64_t nnz_t = 0;
            for (int w = 0; w < ns && good; ++w) {
                const int32_t *ci = ss[w]->col_idxs.data();
                for (int64_t q = q0; q < q1 && good; ++q) {
                    const int64_t c = q / C, i = q % C, cs = ss[w]->chunk_ptrs[(size_t)c], L = ss[w]->chunk_lengths[(size_t)c];
                    int32_t pc;
                    const int64_t le = ss[w]->dtype == USPMV_F64 ? effective_len(ci, ss[w]->values_f64.data(), cs, L, i, C, &pc)
                                                                 : effective_len(ci, ss[w]->values_f32.data(), cs, L, i, C, &pc);
                    if (pc >= 0) my_max = std::max(my_max, pc);
                    int32_t prev = -1;
                    int run = 0;
                    for (int64_t j = 0; j < le; ++j) {
                        const int32_t col = ci[cs + j * C + i];
                        const int32_t sw = col >> wlog;
                        my_max = std::max(my_max, col);
                        if (sw < prev) { good = false; break; }
                        run = sw == prev ? run + 1 : 1;
                        if (run > 255) { good = false; break; }     // one count byte per (row, window)
                        prev = sw;
                        lo = std::min(lo, sw); hi = std::max(hi, sw);
                    }
                    nnz_t += le;
                    wave_n[w][(size_t)(t * wpt + (q - q0) / 64)] += le;
                }
            }
            if (!good || hi < 0) continue;
            const int64_t nS = (int64_t)hi - lo + 1;
            // staging cost: every window is copied once per tile
            if ((double)nS * (double)((int64_t)1 << wlog) * (double)vsz > max_stage_bytes_per_nnz * (double)std::max<int64_t>(nnz_t, 1)) continue;
            if (nS > 4096) continue;
            ok[(size_t)t] = 1; smin[(size_t)t] = lo; S[(size_t)t] = (int32_t)nS;
        }
#pragma omp critical
        max_col = std::max(max_col, my_max);
    }
    // ---- compact list of sweep tiles, offsets
    p->tile_ids.clear(); p->t_smin.clear(); p->t_S.clear(); p->t_cnt_off.clear();
    int64_t cnt_bytes = 0, tot[2] = {0, 0};
    for (int64_t t = 0; t < n_tiles; ++t) {
        if (!ok[(size_t)t]) continue;
        p->tile_ids.push_back((int32_t)t); p->t_smin.push_back(smin[(size_t)t]); p->t_S.push_back(S[(size_t)t]);
        p->t_cnt_off.push_back((uint64_t)cnt_bytes);
        cnt_bytes += (int64_t)S[(size_t)t] * R;
    }
    const int64_t nsw = (int64_t)p->tile_ids.size();
    p->n_sweep_tiles = nsw;
    p->x_len_min = (int64_t)max_col + 1;
    // chunks the sweep does not cover (gather kernel over this id list)
    p->rest_chunks.clear();
    for (int64_t t = 0; t < n_tiles; ++t)
        if (!ok[(size_t)t])
            for (int64_t c = t * R / C; c < std::min((t + 1) * R / C, nc); ++c) p->rest_chunks.push_back((int32_t)c);
    if (nsw == 0) return USPMV_OK;
    for (int w = 0; w < ns; ++w) {
        auto &wo = w == 0 ? p->wave_off : p->wave_off_b;
        wo.assign((size_t)(nsw * wpt), 0);
        for (int64_t k = 0; k < nsw; ++k)
            for (int64_t v = 0; v < wpt; ++v) {
                wo[(size_t)(k * wpt + v)] = (uint32_t)tot[w];
                tot[w] += wave_n[w][(size_t)(p->tile_ids[(size_t)k] * wpt + v)];
            }
        if (tot[w] > (int64_t)UINT32_MAX) return USPMV_OK;
    }
    p->cnt.assign((size_t)cnt_bytes, 0);
    constexpr size_t SPARE = 64;   // inactive lanes of the kernel load the batch's first element: keep that address valid at the very end
    p->idx.assign((size_t)tot[0] + SPARE, 0);
    p->pad_col.assign((size_t)(nsw * R), -1);
    if (s->dtype == USPMV_F64) p->vals_f64.assign((size_t)tot[0] + SPARE, 0.0); else p->vals_f32.assign((size_t)tot[0] + SPARE, 0.0f);
    if (s2) {
        p->cnt_b.assign((size_t)cnt_bytes, 0);
        p->idx_b.assign((size_t)tot[1] + SPARE, 0);
        p->pad_col_b.assign((size_t)(nsw * R), -1);
        if (s2->dtype == USPMV_F64) p->vals_b_f64.assign((size_t)tot[1] + SPARE, 0.0); else p->vals_b_f32.assign((size_t)tot[1] + SPARE, 0.0f);
    }
    // ---- pass 2: counts and the compacted entry stream
#pragma omp parallel
    {
        std::vector<int64_t> le((size_t)R), pos((size_t)R);
#pragma omp for schedule(dynamic, 8)
        for (int64_t k = 0; k < nsw; ++k) {
            const int64_t t = p->tile_ids[(size_t)k], q0 = t * R, q1 = std::min(q0 + R, n_pad);
            const int32_t lo = p->t_smin[(size_t)k];
            const int64_t nS = p->t_S[(size_t)k];
            for (int w = 0; w < ns; ++w) {
                const uspmv_scs *m = ss[w];
                const int32_t *ci = m->col_idxs.data();
                uint8_t *cnt = (w == 0 ? p->cnt.data() : p->cnt_b.data()) + p->t_cnt_off[(size_t)k];
                int32_t *padc = (w == 0 ? p->pad_col.data() : p->pad_col_b.data()) + k * R;
                uint16_t *idx = w == 0 ? p->idx.data() : p->idx_b.data();
                const auto &wo = w == 0 ? p->wave_off : p->wave_off_b;
                for (int64_t q = q0; q < q1; ++q) {
                    const int64_t c = q / C, i = q % C, cs = m->chunk_ptrs[(size_t)c], L = m->chunk_lengths[(size_t)c];
                    int32_t pc;
                    le[(size_t)(q - q0)] = m->dtype == USPMV_F64 ? effective_len(ci, m->values_f64.data(), cs, L, i, C, &pc)
                                                                  : effective_len(ci, m->values_f32.data(), cs, L, i, C, &pc);
                    padc[q - q0] = pc;
                    pos[(size_t)(q - q0)] = 0;
                    for (int64_t j = 0; j < le[(size_t)(q - q0)]; ++j) ++cnt[(size_t)(((ci[cs + j * C + i] >> wlog) - lo) * R + (q - q0))];
                }
                for (int64_t v = 0; v < wpt; ++v) {
                    const int64_t r0 = v * 64, r1 = std::min<int64_t>(r0 + 64, q1 - q0);
                    if (r0 >= r1) break;
                    int64_t out = wo[(size_t)(k * wpt + v)];
                    for (int64_t sw = 0; sw < nS; ++sw) {
                        int mx = 0;
                        for (int64_t r = r0; r < r1; ++r) mx = std::max<int>(mx, cnt[(size_t)(sw * R + r)]);
                        for (int kk = 0; kk < mx; ++kk)
                            for (int64_t r = r0; r < r1; ++r) {
                                if (cnt[(size_t)(sw * R + r)] <= kk) continue;
                                const int64_t q = q0 + r, c = q / C, i = q % C, cs = m->chunk_ptrs[(size_t)c];
                                const int64_t j = pos[(size_t)r]++;
                                const int64_t src = cs + j * C + i;
                                idx[(size_t)out] = (uint16_t)(ci[src] - ((int32_t)(lo + sw) << wlog));
                                if (m->dtype == USPMV_F64) (w == 0 ? p->vals_f64 : p->vals_b_f64)[(size_t)out] = m->values_f64[(size_t)src];
                                else (w == 0 ? p->vals_f32 : p->vals_b_f32)[(size_t)out] = m->values_f32[(size_t)src];
                                ++out;
                            }
                    }
                }
            }
        }
    }
    p->valid = true;
    return USPMV_OK;
}


// The sweep for BLOCK vectors (row-major X rows of row_bytes = block_vec_size * sizeof(VT) bytes; csrc/spmmv_sweep.hip): the same
// compacted per-wave entry stream and per-(row, window) counts, with two differences.  A window is 2^wlog X ROWS.  And a tile stages
// only the windows its rows touch (t_win_ptr / wins), not the whole range between the first and the last: a tile of a 3-D grid matrix
// reaches into three plane-sized index ranges far apart.  Why it exists: the slot-ordered phases of the phased block plan stage every
// neighbour line of a tile once per (dy, dz) -- 11.9 X rows per matrix row on the Queen_4147-class matrix whatever the tile size --
// while phases ordered by X-row windows stage 3 (k + 2) lines for a tile of k grid lines.  Rows stay in slot order
// (windows ascending = slots ascending for column-sorted rows), so every (row, column) accumulator is the reference's FMA chain
// (block_spmv_omp_scs_general, code/kernels.hpp:306-398).
int uspmv_build_block_sweep_plan(const uspmv_scs *s, int wlog, int tile_rows, int row_bytes, double max_stage_bytes_per_nnz,
                                 uspmv_block_sweep_plan *p) {
    p->valid = false;
    const int64_t C = s->C, nc = s->n_chunks;
    if (tile_rows != 1024 && tile_rows != 2048 && tile_rows != 4096) tile_rows = 2048;
    if (C < 1 || C > 64 || 64 % C != 0 || nc < 1) return USPMV_OK;          // a wave covers whole chunks
    if (wlog < 6 || wlog > 12) return USPMV_OK;
    if (s->n_elements > (int64_t)UINT32_MAX) return USPMV_OK;
    const int64_t R = tile_rows, n_pad = nc * C, n_tiles = (n_pad + R - 1) / R, wpt = R / 64;
    p->tile_rows = tile_rows; p->wlog = wlog; p->n_tiles = n_tiles;
    const int32_t *ci = s->col_idxs.data();
    auto eff = [&](int64_t cs, int64_t L, int64_t i, int32_t *pc) {
        return s->dtype == USPMV_F64 ? effective_len(ci, s->values_f64.data(), cs, L, i, C, pc) : effective_len(ci, s->values_f32.data(), cs, L, i, C, pc);
    };

    // ---- pass 1: which tiles sweep, the windows they touch, entries per wave
    std::vector<std::vector<int32_t>> twins((size_t)n_tiles);
    std::vector<char> ok((size_t)n_tiles, 0);
    std::vector<int64_t> wave_n((size_t)(n_tiles * wpt), 0);
    int32_t max_col = 0;
#pragma omp parallel
    {
        int32_t my_max = 0;
        std::vector<int32_t> seen;
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int64_t q0 = t * R, q1 = std::min(q0 + R, n_pad);
            bool good = true;
            int64_t nnz_t = 0;
            seen.clear();
            for (int64_t q = q0; q < q1 && good; ++q) {
                const int64_t c = q / C, i = q % C, cs = s->chunk_ptrs[(size_t)c], L = s->chunk_lengths[(size_t)c];
                int32_t pc;
                const int64_t le = eff(cs, L, i, &pc);
                if (pc >= 0) my_max = std::max(my_max, pc);
                int32_t prev = -1;
                int run = 0;
                for (int64_t j = 0; j < le; ++j) {
                    const int32_t col = ci[cs + j * C + i];
                    const int32_t sw = col >> wlog;
                    my_max = std::max(my_max, col);
                    if (sw < prev) { good = false; break; }
                    run = sw == prev ? run + 1 : 1;
                    if (run > 255) { good = false; break; }     // one count byte per (row, window)
                    if (sw != prev) seen.push_back(sw);
                    prev = sw;
                }
                nnz_t += le;
                wave_n[(size_t)(t * wpt + (q - q0) / 64)] += le;
            }
            if (!good) continue;
            std::sort(seen.begin(), seen.end());
            seen.erase(std::unique(seen.begin(), seen.end()), seen.end());
            // staging cost: every touched window is copied once per tile
            if ((double)seen.size() * (double)((int64_t)1 << wlog) * (double)row_bytes > max_stage_bytes_per_nnz * (double)std::max<int64_t>(nnz_t, 1)) continue;
            if (seen.size() > 4096) continue;
            ok[(size_t)t] = 1;
            twins[(size_t)t] = seen;
        }
#pragma omp critical
        max_col = std::max(max_col, my_max);
    }
    p->tile_ids.clear(); p->t_win_ptr.assign(1, 0); p->wins.clear(); p->t_cnt_off.clear(); p->rest_chunks.clear();
    int64_t cnt_bytes = 0, tot = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
        if (!ok[(size_t)t]) {
            for (int64_t c = t * R / C; c < std::min((t + 1) * R / C, nc); ++c) p->rest_chunks.push_back((int32_t)c);
            continue;
        }
        p->tile_ids.push_back((int32_t)t);
        p->wins.insert(p->wins.end(), twins[(size_t)t].begin(), twins[(size_t)t].end());
        p->t_win_ptr.push_back((int32_t)p->wins.size());
        p->t_cnt_off.push_back((uint64_t)cnt_bytes);
        cnt_bytes += (int64_t)twins[(size_t)t].size() * R;
    }
    const int64_t nsw = (int64_t)p->tile_ids.size();
    p->n_sweep_tiles = nsw;
    p->windows_staged = (int64_t)p->wins.size();
    p->x_rows_min = (int64_t)max_col + 1;
    if (nsw == 0) return USPMV_OK;
    p->wave_off.assign((size_t)(nsw * wpt), 0);
    for (int64_t k = 0; k < nsw; ++k)
        for (int64_t v = 0; v < wpt; ++v) {
            p->wave_off[(size_t)(k * wpt + v)] = (uint32_t)tot;
            tot += wave_n[(size_t)(p->tile_ids[(size_t)k] * wpt + v)];
        }
    if (tot > (int64_t)UINT32_MAX) return USPMV_OK;
    constexpr size_t SPARE = 64;
    p->cnt.assign((size_t)cnt_bytes, 0);
    p->idx.assign((size_t)tot + SPARE, 0);
    p->pad_col.assign((size_t)(nsw * R), -1);
    if (s->dtype == USPMV_F64) p->vals_f64.assign((size_t)tot + SPARE, 0.0); else p->vals_f32.assign((size_t)tot + SPARE, 0.0f);

    // ---- pass 2: counts and the compacted entry stream (windows of the tile ascending, rounds, lanes ascending: what the kernel walks)
#pragma omp parallel
    {
        std::vector<int64_t> le((size_t)R), pos((size_t)R);
#pragma omp for schedule(dynamic, 4)
        for (int64_t k = 0; k < nsw; ++k) {
            const int64_t t = p->tile_ids[(size_t)k], q0 = t * R, q1 = std::min(q0 + R, n_pad);
            const int32_t *tw = p->wins.data() + p->t_win_ptr[(size_t)k];
            const int64_t nS = p->t_win_ptr[(size_t)k + 1] - p->t_win_ptr[(size_t)k];
            uint8_t *cnt = p->cnt.data() + p->t_cnt_off[(size_t)k];
            int32_t *padc = p->pad_col.data() + k * R;
            for (int64_t q = q0; q < q1; ++q) {
                const int64_t c = q / C, i = q % C, cs = s->chunk_ptrs[(size_t)c], L = s->chunk_lengths[(size_t)c];
                int32_t pc;
                le[(size_t)(q - q0)] = eff(cs, L, i, &pc);
                padc[q - q0] = pc;
                pos[(size_t)(q - q0)] = 0;
                for (int64_t j = 0; j < le[(size_t)(q - q0)]; ++j) {
                    const int32_t sw = ci[cs + j * C + i] >> wlog;
                    const int64_t at = std::lower_bound(tw, tw + nS, sw) - tw;
                    ++cnt[(size_t)(at * R + (q - q0))];
                }
            }
            for (int64_t v = 0; v < wpt; ++v) {
                const int64_t r0 = v * 64, r1 = std::min<int64_t>(r0 + 64, q1 - q0);
                if (r0 >= r1) break;
                int64_t out = p->wave_off[(size_t)(k * wpt + v)];
                for (int64_t sw = 0; sw < nS; ++sw) {
                    int mx = 0;
                    for (int64_t r = r0; r < r1; ++r) mx = std::max<int>(mx, cnt[(size_t)(sw * R + r)]);
                    for (int kk = 0; kk < mx; ++kk)
                        for (int64_t r = r0; r < r1; ++r) {
                            if (cnt[(size_t)(sw * R + r)] <= kk) continue;
                            const int64_t q = q0 + r, c = q / C, i = q % C, cs = s->chunk_ptrs[(size_t)c];
                            const int64_t j = pos[(size_t)r]++;
                            const int64_t src = cs + j * C + i;
                            p->idx[(size_t)out] = (uint16_t)(ci[src] - (tw[sw] << wlog));
                            if (s->dtype == USPMV_F64) p->vals_f64[(size_t)out] = s->values_f64[(size_t)src];
                            else p->vals_f32[(size_t)out] = s->values_f32[(size_t)src];
                            ++out;
                        }
                }
            }
        }
    }
    p->valid = true;
    return USPMV_OK;
}
