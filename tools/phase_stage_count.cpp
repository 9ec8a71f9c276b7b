// Host-only: X rows the phased block plan stages on the config-3 class matrix under a row order / phase-cut choice.
//   g++ -O2 -std=c++17 -fopenmp -I include -I ultimate-spmv_amd/host tools/phase_stage_count.cpp -L ultimate-spmv_amd -luspmv -o gpurun_out/scratch/psc
//   psc G mode(1|2|3|4) stride lines phase_cost
#include "uspmv.h"
#include "uspmv_internal.hpp"
#include <cstdio>
#include <cstdlib>
#include <chrono>
int main(int argc, char **argv) {
    const int g = argc > 1 ? atoi(argv[1]) : 111, mode = argc > 2 ? atoi(argv[2]) : 1;
    const long stride = argc > 3 ? atol(argv[3]) : 3L * g;
    const int lines = argc > 4 ? atoi(argv[4]) : 4, pc = argc > 5 ? atoi(argv[5]) : 0;
    uspmv_coo_t *coo = nullptr; uspmv_scs_t *s = nullptr;
    if (uspmv_gen_stencil27(g, g, g, 3, 0x5EED, 0.0, 0, 3L * g * g * g, &coo)) return 1;
    if (uspmv_convert_to_scs(coo, 32, 512, USPMV_F64, nullptr, &s)) return 1;
    if (uspmv_permute_scs_cols(s, s->old_to_new_idx.data())) return 1;
    uspmv_scs r; std::vector<int32_t> map;
    auto t0 = std::chrono::steady_clock::now();
    int moved = mode == 3 ? uspmv_scs_reorder_bricks(s, stride, lines, &r, &map) : uspmv_scs_reorder_rows(s, mode, &r, &map);
    auto t1 = std::chrono::steady_clock::now();
    uspmv_phased_plan p;
    if (uspmv_build_phased_plan(moved == 1 ? &r : s, 256, 8, &p, 0, pc)) return 1;
    auto t2 = std::chrono::steady_clock::now();
    printf("g=%d mode=%d stride=%ld lines=%d phase_cost=%d: tiles=%lld phases=%lld (%.2f per tile) staged=%zu (%.2f per row) max_rows=%d reorder %.1fs plan %.1fs\n",
           g, mode, stride, lines, pc, (long long)p.n_tiles, (long long)p.n_phases, (double)p.n_phases / p.n_tiles, p.xrows.size(),
           (double)p.xrows.size() / s->n_rows, p.max_rows_used, std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(t2 - t1).count());
    return 0;
}
