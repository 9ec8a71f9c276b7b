// Multi-rank (one process per GPU, RCCL) path of the `uspmv` harness -- see uspmv_dist.cpp.
#pragma once
#include <string>

#include "uspmv.h"

struct DistConfig {
    long C = 32, sigma = 1;
    bool seg_nnz = false, comm_halos = true, ba_synch = true, tlc = true, verbose = false;
    double bench_time = 5.0;
    std::string matrix_name;
};

bool uspmv_dist_requested();                                        // WORLD_SIZE > 1 (or USPMV_FORCE_DIST)
int uspmv_run_distributed(const DistConfig &c, uspmv_coo_t *total); // bench mode, scs, -dp, single vector
