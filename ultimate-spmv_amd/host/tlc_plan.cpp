// Tile-local-column (TLC) plan: an MI355X-specific device layout derived from a SELL-C-sigma
// struct at upload time.  Nothing like it exists in the reference; the SCS arrays themselves stay
// bit-identical to the reference's, this is an additional index structure for the SpMV kernel.
//
// A tile = the tile_rows/C consecutive chunks one workgroup of tile_rows (256 | 512 | 1024) threads
// processes.  For every tile the
// planner lists the distinct 16-element lines of x its column indices touch (sorted).  If there are
// at most `max_lines` of them, the workgroup can stage exactly those lines into LDS with coalesced
// 16-byte loads and every column index of the tile becomes a 16-bit LDS-local index
//      local = (position of the line in the tile's list) * 16 + (col & 15).
// That replaces the 4-byte global column stream by a 2-byte one (12 -> 10 bytes per non-zero in
// double precision) and the 64-lane eight-byte global gathers by ds_read_b64.  Tiles whose
// footprint is too wide keep the plain 32-bit gather path (n_lines = 0).
//
// col16 layout: per chunk, slots in groups of four, [group][row i][slot % 4], so that one lane
// reads the four indices of its next four slots with a single 8-byte load.
#include <algorithm>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <omp.h>

#include "uspmv_internal.hpp"

int uspmv_build_tlc_plan(const uspmv_scs *s, const uspmv_scs *s2, int max_lines, int tile_rows, uspmv_tlc_plan *p, int line_shift) {
    // s2 (optional): a second struct with the same row layout (the sp part of an ap[dp_sp] pair);
    // the line list of a tile then covers the columns of both, each struct gets its own col16.
    p->valid = false;
    const int64_t C = s->C;
    // line_shift: log2 of the elements per line -- 4 (16-element lines) for SpMV; 0 for the block-vector
    // plan, where an "element" is a whole X row of b values and a tile is one wave (64 rows) tall.
    if (line_shift < 0 || line_shift > 4) line_shift = 4;
    const int LS = line_shift;
    const int32_t LM = (1 << LS) - 1;
    if (tile_rows != 32 && tile_rows != 64 && tile_rows != 128 && tile_rows != 256 && tile_rows != 512 && tile_rows != 1024) tile_rows = 256;
    if (C < 1 || C > tile_rows || tile_rows % C != 0 || s->n_chunks < 1) return USPMV_OK;  // unsupported shape: no plan
    if (s2 && (s2->C != C || s2->n_chunks != s->n_chunks)) return USPMV_OK;
    if (max_lines < 1) return USPMV_OK;
    if (max_lines > (65536 >> LS)) max_lines = 65536 >> LS;  // 16-bit local indices
    p->line_shift = LS;
    const int64_t T = tile_rows / C;
    p->tile_rows = tile_rows;
    const int64_t n_tiles = (s->n_chunks + T - 1) / T;
    p->chunks_per_tile = (int)T;
    p->n_tiles = n_tiles;
    const uspmv_scs *ss[2] = {s, s2};
    std::vector<uint32_t> *ptrs[2] = {&p->c16_ptrs, &p->c16_ptrs_b};
    std::vector<uint16_t> *c16[2] = {&p->col16, &p->col16_b};
    const int ns = s2 ? 2 : 1;
    for (int w = 0; w < ns; ++w) {
        ptrs[w]->assign((size_t)s->n_chunks + 1, 0);
        int64_t tot16 = 0;
        for (int64_t c = 0; c < s->n_chunks; ++c) {
            (*ptrs[w])[(size_t)c] = (uint32_t)tot16;
            tot16 += ((int64_t)(ss[w]->chunk_lengths[(size_t)c] + 3) / 4) * 4 * C;
            if (tot16 > (int64_t)UINT32_MAX) return USPMV_OK;  // too large for 32-bit offsets: no plan
        }
        (*ptrs[w])[(size_t)s->n_chunks] = (uint32_t)tot16;
        c16[w]->assign((size_t)tot16, 0);
    }
    std::vector<std::vector<int32_t>> tile_lines((size_t)n_tiles);
    int32_t max_col = 0;
#pragma omp parallel
    {
        std::vector<int32_t> lines, pos;
        int32_t my_max = 0;
#pragma omp for schedule(dynamic, 16)
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int64_t c0 = t * T, c1 = std::min<int64_t>(c0 + T, s->n_chunks);
            lines.clear();
            int32_t lo = INT32_MAX, hi = -1;
            int64_t n_el = 0;
            for (int w = 0; w < ns; ++w) {
                const int32_t *ci = ss[w]->col_idxs.data();
                const int64_t e0 = ss[w]->chunk_ptrs[(size_t)c0], e1 = ss[w]->chunk_ptrs[(size_t)c1];
                n_el += e1 - e0;
                for (int64_t k = e0; k < e1; ++k) {
                    const int32_t l = ci[k] >> LS;
                    lo = std::min(lo, l); hi = std::max(hi, l);
                    my_max = std::max(my_max, ci[k]);
                }
            }
            if (hi < 0) continue;                       // tile without elements: nothing to stage
            const int64_t range = (int64_t)hi - lo + 1;
            const bool dense = range <= 65536;          // dense marking over the tile's line range
            int32_t n = 0;
            if (dense) {
                pos.assign((size_t)range, -1);
                for (int w = 0; w < ns; ++w) {
                    const int32_t *ci = ss[w]->col_idxs.data();
                    for (int64_t k = ss[w]->chunk_ptrs[(size_t)c0]; k < ss[w]->chunk_ptrs[(size_t)c1]; ++k) pos[(size_t)((ci[k] >> LS) - lo)] = 0;
                }
                for (int64_t r = 0; r < range && n <= max_lines; ++r)
                    if (pos[(size_t)r] == 0) { pos[(size_t)r] = n++; lines.push_back((int32_t)(lo + r)); }
            } else {                                    // wide footprint: sort + unique
                lines.reserve((size_t)n_el);
                for (int w = 0; w < ns; ++w) {
                    const int32_t *ci = ss[w]->col_idxs.data();
                    for (int64_t k = ss[w]->chunk_ptrs[(size_t)c0]; k < ss[w]->chunk_ptrs[(size_t)c1]; ++k) lines.push_back(ci[k] >> LS);
                }
                std::sort(lines.begin(), lines.end());
                lines.erase(std::unique(lines.begin(), lines.end()), lines.end());
                n = (int32_t)lines.size();
            }
            if (n > max_lines) continue;                // gather path for this tile
            auto local_of = [&](int32_t col) -> uint16_t {
                const int32_t l = col >> LS;
                const int32_t pl = dense ? pos[(size_t)(l - lo)]
                                         : (int32_t)(std::lower_bound(lines.begin(), lines.end(), l) - lines.begin());
                return (uint16_t)((pl << LS) | (col & LM));
            };
            for (int w = 0; w < ns; ++w) {
                const int32_t *ci = ss[w]->col_idxs.data();
                for (int64_t c = c0; c < c1; ++c) {
                    const int64_t cs = ss[w]->chunk_ptrs[(size_t)c], L = ss[w]->chunk_lengths[(size_t)c];
                    uint16_t *q = c16[w]->data() + (*ptrs[w])[(size_t)c];
                    for (int64_t j = 0; j < L; ++j)
                        for (int64_t i = 0; i < C; ++i)
                            q[(j / 4) * 4 * C + i * 4 + (j % 4)] = local_of(ci[cs + j * C + i]);
                }
            }
            tile_lines[(size_t)t] = lines;
        }
#pragma omp critical
        max_col = std::max(max_col, my_max);
    }
    p->tile_line_ptr.assign((size_t)n_tiles + 1, 0);
    int64_t tot = 0;
    int mx = 0;
    int64_t n_staged = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
        p->tile_line_ptr[(size_t)t] = (int32_t)tot;
        tot += (int64_t)tile_lines[(size_t)t].size();
        mx = std::max<int>(mx, (int)tile_lines[(size_t)t].size());
        n_staged += !tile_lines[(size_t)t].empty();
        if (tot > INT32_MAX) return USPMV_OK;
    }
    p->tile_line_ptr[(size_t)n_tiles] = (int32_t)tot;
    p->tile_lines.resize((size_t)tot);
    for (int64_t t = 0; t < n_tiles; ++t)
        std::copy(tile_lines[(size_t)t].begin(), tile_lines[(size_t)t].end(), p->tile_lines.begin() + p->tile_line_ptr[(size_t)t]);
    p->max_lines_used = mx;
    p->n_staged_tiles = n_staged;
    p->x_len_min = (int64_t)max_col + 1;
    p->valid = n_staged > 0;
    return USPMV_OK;
}


// Re-chunk a SELL-C-sigma struct with C in {1,2,4,8,16} into chunks of 32 rows WITHOUT touching the row
// order: new chunk k = old chunks [k*32/C, (k+1)*32/C), its length the longest of theirs.  Row r keeps
// its position r and its slot order, so y and the per-row FMA chain are unchanged; the matrix stream
// becomes 256/128-byte coalesced segments (a CRS struct, C = 1, turns into SELL-32-1).  Used internally
// by uspmv_dmat_optimize when the extra padding stays small.
int uspmv_scs_rechunk32(const uspmv_scs *s, uspmv_scs *o) {
    const int64_t C = s->C;
    if (C < 1 || C >= 32 || 32 % C != 0) return USPMV_ERR_UNSUPPORTED;
    const int64_t per = 32 / C, n_pad_old = s->n_chunks * C;
    const int64_t nc = (s->n_chunks + per - 1) / per;
    o->C = 32; o->sigma = s->sigma; o->n_rows = s->n_rows; o->n_cols = s->n_cols; o->nnz = s->nnz; o->dtype = s->dtype;
    o->n_chunks = nc; o->n_rows_padded = nc * 32;
    o->chunk_lengths.assign((size_t)nc, 0);
    o->chunk_ptrs.assign((size_t)nc + 1, 0);
    int64_t cur = 0;
    for (int64_t k = 0; k < nc; ++k) {
        int32_t L = 0;
        for (int64_t c = k * per; c < std::min((k + 1) * per, s->n_chunks); ++c) L = std::max(L, s->chunk_lengths[(size_t)c]);
        o->chunk_lengths[(size_t)k] = L;
        o->chunk_ptrs[(size_t)k] = (int32_t)cur;
        cur += (int64_t)L * 32;
        if (cur > INT32_MAX) return USPMV_ERR_OVERFLOW;
    }
    o->chunk_ptrs[(size_t)nc] = (int32_t)cur;
    o->n_elements = cur;
    o->col_idxs.assign((size_t)cur, 0);
    if (s->dtype == USPMV_F64) o->values_f64.assign((size_t)cur, 0.0); else o->values_f32.assign((size_t)cur, 0.0f);
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < nc; ++k) {
        const int64_t base = o->chunk_ptrs[(size_t)k];
        for (int64_t i = 0; i < 32; ++i) {
            const int64_t row = k * 32 + i;
            if (row >= n_pad_old) break;
            const int64_t co = row / C, io = row % C;
            const int64_t cs = s->chunk_ptrs[(size_t)co], L = s->chunk_lengths[(size_t)co];
            for (int64_t j = 0; j < L; ++j) {
                const int64_t src = cs + j * C + io, dst = base + j * 32 + i;
                o->col_idxs[(size_t)dst] = s->col_idxs[(size_t)src];
                if (s->dtype == USPMV_F64) o->values_f64[(size_t)dst] = s->values_f64[(size_t)src];
                else o->values_f32[(size_t)dst] = s->values_f32[(size_t)src];
            }
        }
    }
    return USPMV_OK;
}

// Tie re-ordering for the block (SpMMV) plan.  convert_to_scs sorts the rows of a sigma window by length with
// std::sort (code/utilities.hpp:1930-1941), which leaves rows of EQUAL length in an arbitrary order: a 64-row
// tile of a 512-row window then holds rows from all over the window and touches almost twice the X rows a tile
// of neighbouring rows would.  Rows that sit in chunks of equal length can be exchanged without changing any
// chunk length or pointer, so the plan's private copy of the entries puts them back in original-row order
// inside every run of equal-length chunks of a window; row_map[new position] = position in the caller's
// struct (= where y goes).  Per-row slot order is untouched, hence the same FMA chain per (row, column).
// Row CLUSTERING for the block plan (modes 2 and 4 of uspmv_scs_reorder_rows).  The X rows a 64-row tile stages are the union of its
// rows' columns; for a matrix from a 2-D / 3-D mesh a tile of 64 CONSECUTIVE rows is a line segment of the mesh and touches
// ~10 X rows per row, a compact patch of the mesh about 6.  Rows may be exchanged freely between positions that sit in chunks of
// equal length (nothing about the chunk structure changes, every row keeps its own padded slot sequence -> same FMA chain), so the
// tiles are re-filled greedily: a tile takes the next unassigned row of its chunk-length class (in the order of `base`: ties undone)
// and grows a breadth-first ball around it over rows of the same class (the matrix' own column indices are the graph).
// flat (mode 4): the ball grows only along the slots AROUND THE DIAGONAL that one phase of the plan can hold (32 slots at most, an
// equal share of the row when it needs several phases) -- for column-sorted rows of a mesh those are the neighbours whose own rows
// share X rows with this one IN THE SAME PHASES, so for a 3-D mesh the balls come out as patches of one plane (three phases, each
// staging one plane's patch) instead of 3 x 3 x 3 blocks (each phase staging three planes' worth).  Rows of one phase: all slots.
// row_map[new position] = position in the caller's struct.  O(elements); the matrix is cut into segments of whole sigma windows that
// are clustered independently (in parallel; a ball does not cross a segment's end).
// seg_stride > 1: only every seg_stride-th segment is clustered (the others keep `base`): the trial run that decides whether the whole
// matrix is worth it.  Returns the chunks per segment.
static int64_t cluster_row_map(const uspmv_scs *s, const std::vector<int32_t> &base, int64_t window_chunks, std::vector<int32_t> *row_map, bool flat, int64_t seg_stride = 1,
                               int64_t tile_rows = 64, int64_t seg_rows = 65536) {
    const int64_t C = s->C, nc = s->n_chunks, n_pad = nc * C;
    const int64_t T = std::max<int64_t>(1, tile_rows / C);
    int64_t seg_chunks = std::max<int64_t>(window_chunks, T);
    while (seg_chunks * C < seg_rows) seg_chunks *= 2;
    const int64_t n_seg = (nc + seg_chunks - 1) / seg_chunks;
    if (seg_stride > 1) *row_map = base;
    std::vector<char> assigned((size_t)n_pad, 0);
    std::vector<int32_t> stamp((size_t)n_pad, -1), cnt((size_t)n_pad, 0);
#pragma omp parallel
    {
        std::vector<int32_t> cls_rows;
        std::vector<std::pair<int32_t, int32_t>> heap;      // (rows of the tile pointing at u, -u)
        std::vector<int64_t> cls_begin, cursor;
#pragma omp for schedule(dynamic, 1)
        for (int64_t sg = seg_stride / 2; sg < n_seg; sg += seg_stride) {
            const int64_t cA = sg * seg_chunks, cB = std::min(nc, cA + seg_chunks), lo = cA * C, hi = cB * C;
            int32_t max_len = 0;
            for (int64_t c = cA; c < cB; ++c) max_len = std::max(max_len, s->chunk_lengths[(size_t)c]);
            // the segment's rows of every class in `base` order, and a cursor to the first one that may still be unassigned
            cls_begin.assign((size_t)max_len + 2, 0);
            for (int64_t c = cA; c < cB; ++c) cls_begin[(size_t)s->chunk_lengths[(size_t)c] + 1] += C;
            for (size_t l = 0; l + 1 < cls_begin.size(); ++l) cls_begin[l + 1] += cls_begin[l];
            cls_rows.resize((size_t)(hi - lo));
            cursor.assign(cls_begin.begin(), cls_begin.end() - 1);
            for (int64_t c = cA; c < cB; ++c) {
                const int32_t l = s->chunk_lengths[(size_t)c];
                for (int64_t i = 0; i < C; ++i) cls_rows[(size_t)cursor[(size_t)l]++] = base[(size_t)(c * C + i)];
            }
            cursor.assign(cls_begin.begin(), cls_begin.end() - 1);
            int32_t blob = 0;
            for (int64_t c0 = cA; c0 < cB;) {
                const int64_t tile_end = std::min(cB, (c0 / T + 1) * T);
                int64_t c1 = c0 + 1;
                const int32_t l = s->chunk_lengths[(size_t)c0];
                while (c1 < tile_end && s->chunk_lengths[(size_t)c1] == l) ++c1;
                const int64_t need = (c1 - c0) * C;
                int32_t *out = row_map->data() + c0 * C;
                if (l == 0) {                                    // empty chunks: rows stay where `base` has them
                    for (int64_t k = 0; k < need; ++k) { out[k] = base[(size_t)(c0 * C + k)]; assigned[(size_t)out[k]] = 1; }
                    c0 = c1;
                    continue;
                }
                const int64_t n_ph = (l + 31) / 32, width = flat && n_ph > 1 ? (l + n_ph - 1) / n_ph : l;
                int64_t got = 0;
                heap.clear();
                ++blob;
                while (got < need) {
                    int32_t v = -1;
                    while (!heap.empty()) {                      // the unassigned row most of the tile's rows point at (lowest position on a tie)
                        std::pop_heap(heap.begin(), heap.end());
                        const std::pair<int32_t, int32_t> top = heap.back();
                        heap.pop_back();
                        const int32_t u = -top.second;
                        if (!assigned[(size_t)u] && cnt[(size_t)u] == top.first) { v = u; break; }   // (else: taken, or an older count of u)
                    }
                    if (v < 0) {                                 // (re)seed: the next unassigned row of the class
                        int64_t &cu = cursor[(size_t)l];
                        while (cu < cls_begin[(size_t)l + 1] && assigned[(size_t)cls_rows[(size_t)cu]]) ++cu;
                        if (cu >= cls_begin[(size_t)l + 1]) break;   // (cannot happen: the class has exactly as many rows as positions)
                        v = cls_rows[(size_t)cu];
                    }
                    assigned[(size_t)v] = 1;
                    out[got++] = v;
                    const int64_t vc = v / C, vi = v % C, vcs = s->chunk_ptrs[(size_t)vc];
                    int64_t ja = 0, jb = l;
                    if (width < l) {                             // the slots of one phase around the diagonal (the middle of the row when it has none)
                        int64_t d = l / 2;
                        for (int64_t j = 0; j < l; ++j)
                            if (s->col_idxs[(size_t)(vcs + j * C + vi)] == v) { d = j; break; }
                        ja = std::min(std::max<int64_t>(0, d - width / 2), l - width);
                        jb = ja + width;
                    }
                    for (int64_t j = ja; j < jb; ++j) {
                        const int64_t u = s->col_idxs[(size_t)(vcs + j * C + vi)];
                        if (u >= lo && u < hi && !assigned[(size_t)u] && s->chunk_lengths[(size_t)(u / C)] == l) {
                            if (stamp[(size_t)u] != blob) { stamp[(size_t)u] = blob; cnt[(size_t)u] = 0; }
                            heap.emplace_back(++cnt[(size_t)u], (int32_t)-u);
                            std::push_heap(heap.begin(), heap.end());
                        }
                    }
                }
                std::sort(out, out + got);                       // inside the run: position order (neighbouring lanes <-> neighbouring y rows)
                c0 = c1;
            }
        }
    }
    return seg_chunks;
}

// X rows the 64-row tiles of a sample (every `step`-th tile of every seg_stride-th segment of seg_tiles tiles) touch under a row order:
// what a clustering is accepted or refused by
static int64_t sample_tile_columns(const uspmv_scs *s, const std::vector<int32_t> &row_map, int64_t step, int64_t seg_tiles = 0, int64_t seg_stride = 1, int64_t tile_rows = 64) {
    const int64_t C = s->C, nc = s->n_chunks, T = std::max<int64_t>(1, tile_rows / C), n_tiles = (nc + T - 1) / T;
    int64_t total = 0;
#pragma omp parallel reduction(+ : total)
    {
        std::vector<int32_t> cols;
#pragma omp for schedule(dynamic, 4)
        for (int64_t t = 0; t < n_tiles; t += step) {
            if (seg_stride > 1 && (t / seg_tiles) % seg_stride != seg_stride / 2) continue;
            cols.clear();
            for (int64_t c = t * T; c < std::min(nc, (t + 1) * T); ++c) {
                const int64_t L = s->chunk_lengths[(size_t)c];
                for (int64_t i = 0; i < C; ++i) {
                    const int64_t v = row_map[(size_t)(c * C + i)], vcs = s->chunk_ptrs[(size_t)(v / C)], vi = v % C;
                    for (int64_t j = 0; j < L; ++j) cols.push_back(s->col_idxs[(size_t)(vcs + j * C + vi)]);
                }
            }
            std::sort(cols.begin(), cols.end());
            total += (int64_t)(std::unique(cols.begin(), cols.end()) - cols.begin());
        }
    }
    return total;
}

// Measurement aid (needs the caller's permutation and a known line stride): rows dealt to the tiles as FLAT bricks of `lines`
// neighbouring mesh lines -- rows of a chunk-length class ordered by (block of `lines` lines, position in the line, line), cut into the
// class's positions in order, every run of equal-length chunks of a tile then sorted by position.
static bool brick_row_map(const uspmv_scs *s, int64_t stride, int64_t lines, std::vector<int32_t> *row_map) {
    const int64_t C = s->C, nc = s->n_chunks, n_pad = nc * C;
    if (stride < 1 || lines < 1 || (int64_t)s->new_to_old_idx.size() < s->n_rows) return false;
    int32_t max_len = 0;
    for (int64_t c = 0; c < nc; ++c) max_len = std::max(max_len, s->chunk_lengths[(size_t)c]);
    std::vector<int64_t> cls_begin((size_t)max_len + 2, 0);
    for (int64_t c = 0; c < nc; ++c) cls_begin[(size_t)s->chunk_lengths[(size_t)c] + 1] += C;
    for (size_t l = 0; l + 1 < cls_begin.size(); ++l) cls_begin[l + 1] += cls_begin[l];
    std::vector<int32_t> cls_pos((size_t)n_pad);
    {
        std::vector<int64_t> fill(cls_begin.begin(), cls_begin.end() - 1);
        for (int64_t c = 0; c < nc; ++c) {
            const int32_t l = s->chunk_lengths[(size_t)c];
            for (int64_t i = 0; i < C; ++i) cls_pos[(size_t)fill[(size_t)l]++] = (int32_t)(c * C + i);
        }
    }
    const int32_t *n2o = s->new_to_old_idx.data();
    const int64_t n_rows = s->n_rows, blk = lines * stride;
    auto key = [&](int32_t q) -> int64_t {
        if (q >= n_rows) return ((int64_t)1 << 60) | q;
        const int64_t o = n2o[q];
        return (o / blk) * blk + (o % stride) * lines + (o / stride) % lines;
    };
    std::vector<int32_t> rows(cls_pos);
    for (size_t l = 1; l + 1 < cls_begin.size(); ++l) {
        int32_t *b = rows.data() + cls_begin[l], *e = rows.data() + cls_begin[l + 1];
        if (b == e) continue;
        std::vector<std::pair<int64_t, int32_t>> kv((size_t)(e - b));
#pragma omp parallel for schedule(static)
        for (int64_t k = 0; k < (int64_t)kv.size(); ++k) kv[(size_t)k] = {key(b[k]), b[k]};
        std::sort(kv.begin(), kv.end());
        for (size_t k = 0; k < kv.size(); ++k) b[k] = kv[k].second;
    }
    for (int64_t k = 0; k < n_pad; ++k) (*row_map)[(size_t)cls_pos[(size_t)k]] = rows[(size_t)k];
    const int64_t T = std::max<int64_t>(1, 64 / C);
    for (int64_t c0 = 0; c0 < nc;) {
        const int64_t tile_end = std::min(nc, (c0 / T + 1) * T);
        int64_t c1 = c0 + 1;
        while (c1 < tile_end && s->chunk_lengths[(size_t)c1] == s->chunk_lengths[(size_t)c0]) ++c1;
        std::sort(row_map->data() + c0 * C, row_map->data() + c1 * C);
        c0 = c1;
    }
    return true;
}

int uspmv_scs_reorder_bricks(const uspmv_scs *s, int64_t stride, int64_t lines, uspmv_scs *r, std::vector<int32_t> *row_map) {
    row_map->resize((size_t)(s->n_chunks * s->C));
    if (!brick_row_map(s, stride, lines, row_map)) return uspmv_scs_reorder_rows(s, 1, r, row_map);
    return uspmv_scs_reorder_rows(s, -1, r, row_map);
}

int uspmv_scs_reorder_ties(const uspmv_scs *s, uspmv_scs *r, std::vector<int32_t> *row_map) { return uspmv_scs_reorder_rows(s, 1, r, row_map); }

// row_map of the tie re-ordering (mode 1); *window_chunks = chunks per window the rows were ordered in.  true when a row moved
static bool tie_row_map(const uspmv_scs *s, std::vector<int32_t> *row_map, int64_t *window_chunks) {
    const int64_t C = s->C, nc = s->n_chunks, n_pad = nc * C;
    bool changed = false;
    row_map->resize((size_t)n_pad);
    for (int64_t q = 0; q < n_pad; ++q) (*row_map)[(size_t)q] = (int32_t)q;
    const bool have_perm = s->sigma > 1 && std::max<int64_t>(s->sigma, C) % C == 0 && (int64_t)s->new_to_old_idx.size() >= s->n_rows;
    // with the caller's permutation: original row order inside a sigma window.  Without it (a struct rebuilt from device arrays,
    // uspmv_dmat_optimize_block_device): rows ordered by their first column inside runs of at most 16 equal-length chunks -- for
    // locally numbered (banded, stencil-like) matrices that is the original order again.
    const int64_t W = have_perm ? std::max<int64_t>(s->sigma, C) : 16 * C;
    const int64_t cpw = W / C;
    if (window_chunks) *window_chunks = cpw;
    const int32_t *n2o = have_perm ? s->new_to_old_idx.data() : nullptr;
    const int64_t n_rows = s->n_rows;
    std::vector<char> ch((size_t)((nc + cpw - 1) / cpw), 0);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t w = 0; w < (nc + cpw - 1) / cpw; ++w) {
        const int64_t c_end = std::min(nc, (w + 1) * cpw);
        int64_t c0 = w * cpw;
        while (c0 < c_end) {
            int64_t c1 = c0 + 1;
            while (c1 < c_end && s->chunk_lengths[(size_t)c1] == s->chunk_lengths[(size_t)c0]) ++c1;
            int32_t *b = row_map->data() + c0 * C, *e = row_map->data() + c1 * C;
            auto key = [&](int32_t q) -> int64_t {
                if (n2o) return q < n_rows ? (int64_t)n2o[q] : (int64_t)INT32_MAX + q;
                const int64_t cq = q / C;
                return s->chunk_lengths[(size_t)cq] > 0 ? (int64_t)s->col_idxs[(size_t)(s->chunk_ptrs[(size_t)cq] + q % C)] : (int64_t)INT32_MAX + q;
            };
            auto less = [&](int32_t a, int32_t bb) { const int64_t ka = key(a), kb = key(bb); return ka < kb || (ka == kb && a < bb); };
            if (!std::is_sorted(b, e, less)) {
                std::sort(b, e, less);
                ch[(size_t)w] = 1;
            }
            c0 = c1;
        }
    }
    for (char v : ch) changed = changed || v;
    return changed;
}

// mode 1: tie re-ordering; mode 2 / 4: row clustering (balls / flat patches; kept only where a sample of tiles then touches fewer
// X rows than under mode 1, which it falls back to); mode -1: copy under the caller's row_map
// tile_rows: rows of the tiles the clusters are grown for (64: the block plan's; 256: the SpMV plan's, uspmv_dmat_optimize's fallback for unfriendly numberings)
// seg_rows: rows of the segments that are clustered independently of each other (a cluster does not cross a segment's end; 64 Ki by default: many
// segments in parallel; larger: numberings that scatter related rows further apart, fewer threads at work)
int uspmv_scs_reorder_rows(const uspmv_scs *s, int mode, uspmv_scs *r, std::vector<int32_t> *row_map, int tile_rows, int64_t seg_rows) {
    if (tile_rows < 64 || tile_rows % 64 != 0) tile_rows = 64;
    if (seg_rows < 65536) seg_rows = 65536;
    const int64_t C = s->C, nc = s->n_chunks;
    bool changed = false;
    const bool verbose = getenv("USPMV_VERBOSE") != nullptr;
    double t_last = omp_get_wtime();
    auto lap = [&](const char *what) { if (verbose) { const double t = omp_get_wtime(); fprintf(stderr, "[uspmv] block plan row order: %s %.2f s\n", what, t - t_last); t_last = t; } };
    if (mode == -1) changed = true;                          // row_map given by the caller: only the copy below
    else {
        int64_t cpw = 1;
        changed = tie_row_map(s, row_map, &cpw);
        lap("ties undone");
        // (clustering works on the caller's struct: there a column index below the row count IS a row position)
        if ((mode == 2 || mode == 4) && nc * C <= (int64_t)INT32_MAX) {
            std::vector<int32_t> cl((size_t)(nc * C));
            const int64_t T = std::max<int64_t>(1, (int64_t)tile_rows / C), n_tiles = (nc + T - 1) / T;
            // a trial on every 16th segment first (matrices of more than 32 segments): irregular matrices, which gain nothing, stop there
            const int64_t seg_guess = (nc * C + seg_rows - 1) / seg_rows;
            bool worth = true;
            if (seg_guess > 32) {
                const int64_t seg_chunks = cluster_row_map(s, *row_map, cpw, &cl, mode == 4, 16, tile_rows, seg_rows), seg_tiles = std::max<int64_t>(1, seg_chunks / T);
                const int64_t step = std::max<int64_t>(1, seg_tiles / 128);
                const int64_t before = sample_tile_columns(s, *row_map, step, seg_tiles, 16, tile_rows), after = sample_tile_columns(s, cl, step, seg_tiles, 16, tile_rows);
                worth = after * 100 < before * 95;
                if (verbose) fprintf(stderr, "[uspmv] block plan row clustering (mode %d), trial on every 16th segment: %lld X rows against %lld with the ties undone -> %s\n",
                                     mode, (long long)after, (long long)before, worth ? "go on" : "not worth it");
                lap("trial segments");
            }
            if (worth) {
                cluster_row_map(s, *row_map, cpw, &cl, mode == 4, 1, tile_rows, seg_rows);
                lap("clusters grown");
                const int64_t step = std::max<int64_t>(1, n_tiles / 2048);
                const int64_t before = sample_tile_columns(s, *row_map, step, 0, 1, tile_rows), after = sample_tile_columns(s, cl, step, 0, 1, tile_rows);
                if (verbose)
                    fprintf(stderr, "[uspmv] block plan row clustering (mode %d): sampled tiles touch %lld X rows against %lld with the ties undone -> %s\n", mode,
                            (long long)after, (long long)before, after * 100 < before * 95 ? "kept" : "not kept");
                if (after * 100 < before * 95) { row_map->swap(cl); changed = true; }
                lap("sample compared");
            }
        }
    }
    r->C = C; r->sigma = s->sigma; r->n_rows = s->n_rows; r->n_cols = s->n_cols; r->nnz = s->nnz; r->dtype = s->dtype;
    r->n_chunks = nc; r->n_rows_padded = s->n_rows_padded; r->n_elements = s->n_elements;
    r->chunk_ptrs = s->chunk_ptrs; r->chunk_lengths = s->chunk_lengths;
    uspmv_resize_huge(r->col_idxs, (size_t)s->n_elements);
    // (a struct rebuilt from device arrays may carry the indices only -- the values then stay on the device and the caller gathers them
    //  there under row_map: uspmv_dmat_optimize_block_device)
    const bool has_values = (int64_t)(s->dtype == USPMV_F64 ? s->values_f64.size() : s->values_f32.size()) == s->n_elements;
    if (has_values) { if (s->dtype == USPMV_F64) uspmv_resize_huge(r->values_f64, (size_t)s->n_elements); else uspmv_resize_huge(r->values_f32, (size_t)s->n_elements); }
    lap("copy allocated");
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < nc; ++c) {
        const int64_t cs = s->chunk_ptrs[(size_t)c], L = s->chunk_lengths[(size_t)c];
        int64_t from[64];                                    // (C <= 64 wherever a block plan is built; wider chunks take the row loop below)
        if (C <= 64) {
            // slot by slot: the C destinations of a slot are contiguous and their sources sit in the few chunks the tile's rows come from
            for (int64_t i = 0; i < C; ++i) {
                const int64_t src_row = (*row_map)[(size_t)(c * C + i)];
                from[i] = s->chunk_ptrs[(size_t)(src_row / C)] + src_row % C;
            }
            for (int64_t j = 0; j < L; ++j) {
                const int64_t dst = cs + j * C, off = j * C;
                for (int64_t i = 0; i < C; ++i) r->col_idxs[(size_t)(dst + i)] = s->col_idxs[(size_t)(from[i] + off)];
                if (!has_values) continue;
                if (s->dtype == USPMV_F64) for (int64_t i = 0; i < C; ++i) r->values_f64[(size_t)(dst + i)] = s->values_f64[(size_t)(from[i] + off)];
                else for (int64_t i = 0; i < C; ++i) r->values_f32[(size_t)(dst + i)] = s->values_f32[(size_t)(from[i] + off)];
            }
            continue;
        }
        for (int64_t i = 0; i < C; ++i) {
            const int64_t src_row = (*row_map)[(size_t)(c * C + i)];
            const int64_t sc = src_row / C, si = src_row % C, scs = s->chunk_ptrs[(size_t)sc];
            for (int64_t j = 0; j < L; ++j) {
                const int64_t src = scs + j * C + si, dst = cs + j * C + i;
                r->col_idxs[(size_t)dst] = s->col_idxs[(size_t)src];
                if (!has_values) continue;
                if (s->dtype == USPMV_F64) r->values_f64[(size_t)dst] = s->values_f64[(size_t)src];
                else r->values_f32[(size_t)dst] = s->values_f32[(size_t)src];
            }
        }
    }
    lap("entries copied");
    return changed ? 1 : 0;
}

// Phased block plan (SpMMV with 64-byte X rows, csrc/spmmv_kernels.hip: scs_spmmv_quadph).  The one-list-per-tile block plan
// needs 40-50 KB of LDS per 64-row tile for b = 8 in double precision: three workgroups per CU, whose phases (list, X rows,
// matrix entries, arithmetic) barely overlap.  Here a tile's slot range is cut into PHASES -- runs of at most `ngp` groups of four
// slots whose entries touch at most `cap_rows` distinct X rows -- and every phase has its own sorted X-row list and phase-local
// 16-bit indices.  The workgroup stages one phase at a time (cap_rows * row bytes of LDS: 16 KB -> eight workgroups per CU) and
// still walks every row's slots in order, so the FMA chains are unchanged.  An X row needed in two phases is staged twice; for
// matrices whose rows are column-sorted the phases' row sets are (nearly) disjoint.
int uspmv_build_phased_plan(const uspmv_scs *s, int cap_rows, int ngp, uspmv_phased_plan *p, int line_shift, int phase_cost) {
    // line_shift > 0: the lists hold LINES of 2^line_shift consecutive X rows (what a column-major block vector is staged by: one
    // 128-byte line per list entry and column); a phase lists at most cap_rows >> line_shift lines and the local index of an entry
    // is (position of its line) << line_shift | (column & (2^line_shift - 1)).
    p->valid = false;
    p->line_shift = line_shift;
    if (line_shift < 0 || line_shift > 8 || (cap_rows >> line_shift) < 1) return USPMV_OK;
    const int cap_items = cap_rows >> line_shift;
    const int32_t sub_mask = (1 << line_shift) - 1;
    const int64_t C = s->C, nc = s->n_chunks;
    if ((C != 32 && C != 64 && C != 16) || nc < 1 || cap_rows < 256 || cap_rows > 65536 || ngp < 1) return USPMV_OK;
    if (s->n_cols > (int64_t)INT32_MAX) return USPMV_OK;
    const int64_t TR = 64, T = TR / C, n_tiles = (nc + T - 1) / T;
    p->cap_rows = cap_rows; p->ngp = ngp; p->n_tiles = n_tiles;
    p->c16_ptrs.assign((size_t)nc + 1, 0);
    int64_t tot16 = 0;
    for (int64_t c = 0; c < nc; ++c) {
        p->c16_ptrs[(size_t)c] = (uint32_t)tot16;
        tot16 += ((int64_t)(s->chunk_lengths[(size_t)c] + 3) / 4) * 4 * C;
        if (tot16 > (int64_t)UINT32_MAX) return USPMV_OK;
    }
    p->c16_ptrs[(size_t)nc] = (uint32_t)tot16;
    p->col16.assign((size_t)tot16, 0);
    std::vector<std::vector<int32_t>> t_g0((size_t)n_tiles), t_len((size_t)n_tiles), t_rows((size_t)n_tiles);
    int64_t max_col_seen = 0;
    for (int64_t k = 0; k < s->n_elements; ++k) max_col_seen = std::max<int64_t>(max_col_seen, s->col_idxs[(size_t)k]);
    const size_t ncol = (size_t)(max_col_seen >> line_shift) + 1;      // items: X rows, or lines of them
#pragma omp parallel
    {
        std::vector<int64_t> stamp(ncol, -1), gstamp(ncol, -1);
        std::vector<int32_t> pos(ncol, 0), gc, cur;
        int64_t phase_id = 0, group_id = 0;
#pragma omp for schedule(dynamic, 16)
        for (int64_t t = 0; t < n_tiles; ++t) {
            const int64_t c0 = t * T, c1 = std::min(c0 + T, nc);
            int64_t ng = 0;
            for (int64_t c = c0; c < c1; ++c) ng = std::max<int64_t>(ng, (s->chunk_lengths[(size_t)c] + 3) / 4);
            auto &g0s = t_g0[(size_t)t]; auto &lens = t_len[(size_t)t]; auto &rows = t_rows[(size_t)t];
            if (ng == 0) continue;
            auto close_phase = [&](int64_t first_group, int64_t end_group) {
                // sorted row list of the phase, phase-local indices of its entries
                std::sort(cur.begin(), cur.end());
                for (size_t k = 0; k < cur.size(); ++k) pos[(size_t)cur[k]] = (int32_t)k;
                for (int64_t c = c0; c < c1; ++c) {
                    const int64_t cs = s->chunk_ptrs[(size_t)c], L = s->chunk_lengths[(size_t)c];
                    uint16_t *q = p->col16.data() + p->c16_ptrs[(size_t)c];
                    for (int64_t j = first_group * 4; j < std::min(end_group * 4, L); ++j)
                        for (int64_t i = 0; i < C; ++i) {
                            const int32_t col = s->col_idxs[(size_t)(cs + j * C + i)];
                            q[(j / 4) * 4 * C + i * 4 + (j % 4)] = (uint16_t)((pos[(size_t)(col >> line_shift)] << line_shift) | (col & sub_mask));
                        }
                }
                g0s.push_back((int32_t)first_group); lens.push_back((int32_t)cur.size());
                rows.insert(rows.end(), cur.begin(), cur.end());
                cur.clear();
            };
            if (phase_cost > 0) {
                // cuts by dynamic programming: least (staged items + phase_cost per phase) over all cuts with <= ngp groups and <= cap_items
                // items per phase (the greedy form below fills every phase to the brim, which for column-sorted rows of a mesh
                // cuts through the groups of slots that share their X rows)
                std::vector<std::vector<int32_t>> gcols((size_t)ng);
                for (int64_t g = 0; g < ng; ++g) {
                    ++group_id;
                    for (int64_t c = c0; c < c1; ++c) {
                        const int64_t cs = s->chunk_ptrs[(size_t)c], L = s->chunk_lengths[(size_t)c];
                        for (int64_t j = g * 4; j < std::min(g * 4 + 4, L); ++j)
                            for (int64_t i = 0; i < C; ++i) {
                                const int32_t col = s->col_idxs[(size_t)(cs + j * C + i)] >> line_shift;
                                if (gstamp[(size_t)col] != group_id) { gstamp[(size_t)col] = group_id; gcols[(size_t)g].push_back(col); }
                            }
                    }
                }
                std::vector<int64_t> best((size_t)ng + 1, INT64_MAX);
                std::vector<int32_t> from((size_t)ng + 1, -1);
                best[0] = 0;
                for (int64_t i = 0; i < ng; ++i) {
                    if (best[(size_t)i] == INT64_MAX) continue;
                    ++group_id;
                    int64_t cnt = 0;
                    for (int64_t j = i; j < std::min<int64_t>(ng, i + ngp); ++j) {
                        for (int32_t col : gcols[(size_t)j])
                            if (gstamp[(size_t)col] != group_id) { gstamp[(size_t)col] = group_id; ++cnt; }
                        if (j > i && cnt > cap_items) break;
                        const int64_t v = best[(size_t)i] + cnt + phase_cost;
                        if (v < best[(size_t)j + 1]) { best[(size_t)j + 1] = v; from[(size_t)j + 1] = (int32_t)i; }
                    }
                }
                std::vector<int32_t> cuts;
                for (int64_t j = ng; j > 0; j = from[(size_t)j]) cuts.push_back((int32_t)j);
                int64_t first = 0;
                for (size_t k = cuts.size(); k-- > 0;) {
                    ++phase_id;
                    for (int64_t g = first; g < cuts[k]; ++g)
                        for (int32_t col : gcols[(size_t)g])
                            if (stamp[(size_t)col] != phase_id) { stamp[(size_t)col] = phase_id; cur.push_back(col); }
                    close_phase(first, cuts[k]);
                    first = cuts[k];
                }
                continue;
            }
            int64_t first = 0;
            ++phase_id;
            for (int64_t g = 0; g < ng; ++g) {
                gc.clear(); ++group_id;
                for (int64_t c = c0; c < c1; ++c) {
                    const int64_t cs = s->chunk_ptrs[(size_t)c], L = s->chunk_lengths[(size_t)c];
                    for (int64_t j = g * 4; j < std::min(g * 4 + 4, L); ++j)
                        for (int64_t i = 0; i < C; ++i) {
                            const int32_t col = s->col_idxs[(size_t)(cs + j * C + i)] >> line_shift;
                            if (gstamp[(size_t)col] != group_id) { gstamp[(size_t)col] = group_id; gc.push_back(col); }
                        }
                }
                int64_t fresh = 0;
                for (int32_t col : gc) fresh += stamp[(size_t)col] != phase_id;
                if (g > first && ((int64_t)cur.size() + fresh > cap_items || g - first >= ngp)) {
                    close_phase(first, g);
                    first = g; ++phase_id;
                }
                for (int32_t col : gc)
                    if (stamp[(size_t)col] != phase_id) { stamp[(size_t)col] = phase_id; cur.push_back(col); }
            }
            close_phase(first, ng);
        }
    }
    p->ph_ptr.assign((size_t)n_tiles + 1, 0);
    int64_t n_ph = 0, n_rows_tot = 0;
    int mx = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
        p->ph_ptr[(size_t)t] = (int32_t)n_ph;
        n_ph += (int64_t)t_g0[(size_t)t].size();
        n_rows_tot += (int64_t)t_rows[(size_t)t].size();
        if (n_ph > INT32_MAX || n_rows_tot > INT32_MAX) return USPMV_OK;
    }
    p->ph_ptr[(size_t)n_tiles] = (int32_t)n_ph;
    p->ph_g0.resize((size_t)n_ph); p->ph_list_ptr.resize((size_t)n_ph + 1); p->xrows.resize((size_t)n_rows_tot);
    int64_t ph = 0, off = 0;
    for (int64_t t = 0; t < n_tiles; ++t) {
        size_t ro = 0;
        for (size_t k = 0; k < t_g0[(size_t)t].size(); ++k, ++ph) {
            p->ph_g0[(size_t)ph] = t_g0[(size_t)t][k];
            p->ph_list_ptr[(size_t)ph] = (int32_t)off;
            const int len = t_len[(size_t)t][k];
            std::copy(t_rows[(size_t)t].begin() + (long)ro, t_rows[(size_t)t].begin() + (long)(ro + (size_t)len), p->xrows.begin() + off);
            ro += (size_t)len; off += len; mx = std::max(mx, len << line_shift);
        }
    }
    p->ph_list_ptr[(size_t)n_ph] = (int32_t)off;
    p->n_phases = n_ph; p->max_rows_used = mx;
    p->valid = n_ph > 0;
    return USPMV_OK;
}
