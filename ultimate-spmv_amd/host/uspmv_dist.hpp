// Multi-rank (one process per GPU, RCCL) path of the `uspmv` harness -- see uspmv_dist.cpp.
#pragma once
#include <string>

#include "uspmv.h"

struct DistConfig {
    long C = 32, sigma = 1;
    bool seg_metis = false;                    // -seg_metis: graph partition + symmetric permutation on rank 0 (code/mpi_funcs.hpp:494-598)
    std::string part_file;                     // ... part ids from this file instead of the built-in partitioner
    bool seg_nnz = false, comm_halos = true, ba_synch = true, tlc = true, verbose = false;
    bool no_overlap = false, use_graph = true, print_comm_vol = false, no_pack = false;
    int block_vec_size = 1, layout = USPMV_COLWISE, vec_mode = USPMV_BULKVEC;   // -block_vec_size, -block_vec_layout, -mpi_mode
    double bench_time = 5.0;
    int bench_steps = 0, bench_warmup = -1;   // -bench_steps K: time exactly K steps between barriers (0 = the reference's doubling loop); -bench_warmup W (-1 = 100)
    bool check_y = false;                      // -check_y 1: bitwise self-check of one distributed step (uspmv_dist_check)
    // -step_form overlap|plain|pad|fused|auto: how the single-vector step is arranged around the exchange.  overlap = interior tiles
    // during the exchange, boundary tiles after it; plain = exchange, then the whole matrix (the reference's order); pad = overlap with
    // the padding-only tiles in front of the exchange; fused = pad in one launch (eager steps only); auto = time each for a few steps
    // on THIS machine, all ranks agreeing on the slowest rank's clock, and keep the fastest (all forms give the same bits).
    std::string step_form = "auto";
    char random_init_x = '0';                  // -rand_x 0 | 1 | m: DefaultValues 5.0 | default-seeded mt19937 over [min, max] of |values| of the WHOLE matrix
                                               // (rank 0 extracts, MPI_Bcast: code/utilities.hpp:2502-2540; the same sequence on every rank, :880-912) | their midpoint
    bool equilibrate = false;                  // -equilibrate 1: per-rank scaling of the block (code/main.cpp:1117-1125)
    bool sp = false;                           // -sp: single precision matrix, vectors and exchange
    char mode = 'b';                           // -mode s: the reference's COMM-spmv-SWAP loop (code/main.cpp:528-607), -rev iterations, no timing
    unsigned long n_repetitions = 1;
    std::string dump_y;                        // -dump_y <file>: every rank writes y of its rows (original order, raw doubles) to <file>.<rank>
    std::string json;                          // -json <file|->: one JSON line with everything measured (rank 0)
    std::string matrix_name;
};

bool uspmv_dist_requested();                                        // WORLD_SIZE > 1, USPMV_FORCE_DIST or USPMV_LOOPBACK=P
int uspmv_run_distributed(const DistConfig &c);   // bench mode, scs, -dp, single vector; every rank builds only its row block
