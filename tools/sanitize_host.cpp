// sanitizer pass over the host data layer (no HIP): every golden matrix through read / convert / plan / halo / split
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "uspmv.h"
#include <unistd.h>

#include "uspmv_internal.hpp"
#define CK(x) do { int rc_ = (x); if (rc_) { fprintf(stderr, "FAIL %s: %s\n", #x, uspmv_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
    for (int a = 1; a < argc; ++a) {
        uspmv_coo_t *m = nullptr;
        CK(uspmv_read_mtx(argv[a], &m));
        int64_t n, nc, nnz; CK(uspmv_coo_dims(m, &n, &nc, &nnz));
        for (auto cs : std::vector<std::pair<int,int>>{{1,1},{4,8},{16,512},{32,512},{64,64},{128,256},{5,7}}) {
            for (int dt : {USPMV_F64, USPMV_F32}) {
                uspmv_scs_t *s = nullptr;
                CK(uspmv_convert_to_scs(m, cs.first, cs.second, dt, nullptr, &s));
                const int32_t *o2n; CK(uspmv_scs_arrays(s, nullptr, nullptr, nullptr, nullptr, &o2n, nullptr));
                std::vector<int32_t> perm(o2n, o2n + n);
                if (n == nc) CK(uspmv_permute_scs_cols(s, perm.data()));
                uspmv_tlc_plan p;
                if (256 % cs.first == 0) { CK(uspmv_build_tlc_plan(s, nullptr, 512, 256, &p)); CK(uspmv_build_tlc_plan(s, nullptr, 8, 256, &p)); }
                if (cs.first == 32 || cs.first == 64) CK(uspmv_build_tlc_plan(s, nullptr, 300, 64, &p, 0));
                if (cs.first < 32 && 32 % cs.first == 0) { uspmv_scs r; CK(uspmv_scs_rechunk32(s, &r)); }
                if (256 % cs.first == 0) {                                   // column-window sweep plan (two window sizes, small tiles)
                    uspmv_sweep_plan sp;
                    CK(uspmv_build_sweep_plan(s, nullptr, 6, 256, 1e9, &sp));
                    CK(uspmv_build_sweep_plan(s, nullptr, 10, 1024, 24.0, &sp));
                }
                if (cs.first == 32 || cs.first == 64 || cs.first == 16) {   // phased block plan over the tie-re-ordered copy, and the line form
                    uspmv_scs r; std::vector<int32_t> rm;
                    const int moved = uspmv_scs_reorder_rows(s, 1, &r, &rm);
                    if (moved < 0) return 1;
                    uspmv_phased_plan pp;
                    CK(uspmv_build_phased_plan(moved ? &r : s, 256, 8, &pp));
                    CK(uspmv_build_phased_plan(moved ? &r : s, 256, 8, &pp, dt == USPMV_F64 ? 4 : 5));
                    uspmv_scs r2; std::vector<int32_t> rm2;
                    if (uspmv_scs_reorder_rows(s, 2, &r2, &rm2) < 0) return 1;
                    uspmv_scs r4; std::vector<int32_t> rm4;                 // the default: flat patches + cuts by dynamic programming
                    const int moved4 = uspmv_scs_reorder_rows(s, 4, &r4, &rm4);
                    if (moved4 < 0) return 1;
                    CK(uspmv_build_phased_plan(moved4 ? &r4 : s, 256, 8, &pp, 0, 24));
                    CK(uspmv_build_phased_plan(moved4 ? &r4 : s, 512, 8, &pp, 0, 200));
                }
                uspmv_scs_free(s);
            }
        }
        if (n == nc) for (int P : {2, 3, 4}) {
            for (int method : {USPMV_SEG_ROWS, USPMV_SEG_NNZ}) {
                std::vector<int32_t> wsa(P + 1);
                if (uspmv_seg_work_sharing_arr(m, method, P, wsa.data())) continue;   // tiny matrices: flaw in wsa is a legal refusal
                for (int r = 0; r < P; ++r) {
                    uspmv_coo_t *loc = nullptr; CK(uspmv_seg_local_coo(m, wsa.data(), r, &loc));
                    uspmv_scs_t *s = nullptr; if (uspmv_convert_to_scs(loc, 32, 512, USPMV_F64, nullptr, &s)) { uspmv_coo_free(loc); continue; }  // empty block: legal refusal
                    uspmv_halo_t *h = nullptr; CK(uspmv_halo_discover(s, wsa.data(), r, P, &h));
                    int32_t *ia, *ib; int64_t na, nb;
                    CK(uspmv_scs_split_chunks(s, wsa[r + 1] - wsa[r], &ia, &na, &ib, &nb));
                    int64_t sm8[8]; CK(uspmv_scs_meta(s, sm8));
                    std::vector<uint8_t> cls((size_t)std::max<int64_t>(sm8[5], 1)); int32_t pad_col = -2;
                    CK(uspmv_scs_chunk_classes(s, wsa[r + 1] - wsa[r], cls.data(), &pad_col));
                    std::vector<double> yref((size_t)std::max<int32_t>(wsa[r + 1] - wsa[r], 1));
                    CK(uspmv_dist_check_reference(loc, wsa.data(), r, P, USPMV_F64, yref.data()));
                    uspmv_free(ia); uspmv_free(ib); uspmv_halo_free(h); uspmv_scs_free(s); uspmv_coo_free(loc);
                }
            }
        }
        if (n == nc && n >= 8) for (int P : {2, 5}) {                      // graph partition + the reference's post-processing
            std::vector<int32_t> part((size_t)n), wsa2(P + 1), perm2((size_t)n);
            CK(uspmv_graph_partition(m, P, part.data()));
            uspmv_coo_t *pm = nullptr;
            CK(uspmv_coo_apply_partition(m, P, part.data(), &pm, wsa2.data(), perm2.data()));
            uspmv_coo_free(pm);
        }
        uspmv_coo_t *dp, *sp; CK(uspmv_partition_precisions(m, 1.0, &dp, &sp)); uspmv_coo_free(dp); uspmv_coo_free(sp);
        std::string f = std::string("/tmp/uspmv_sanitize_cache.bin"); CK(uspmv_coo_save(m, f.c_str()));
        uspmv_coo_t *m2; CK(uspmv_coo_load(f.c_str(), &m2)); uspmv_coo_free(m2);
        if (n <= nc) CK(uspmv_coo_equilibrate(m));
        uspmv_coo_free(m);
        printf("ok %s\n", argv[a]);
    }
    uspmv_coo_t *g = nullptr; CK(uspmv_gen_stencil27(9, 8, 7, 3, 0x5EED, 4.0, 10, 900, &g)); uspmv_coo_free(g);
    CK(uspmv_gen_kkt(5, 0x5EED, 3, 300, &g)); uspmv_coo_free(g);
    CK(uspmv_gen_banded_random(500, 7, 40, 0x5EED, 3.0, 17, 400, &g)); uspmv_coo_free(g);
    {   // host communicator (one rank) and the exchange plan over its transport
        uspmv_hostcomm_t *hc = nullptr;
        char job[64]; snprintf(job, sizeof job, "san%d", (int)getpid());
        CK(uspmv_hostcomm_create(job, 0, 1, 30.0, &hc));
        CK(uspmv_hostcomm_barrier(hc));
        double v = 2.5; CK(uspmv_hostcomm_allreduce_max_f64(hc, &v));
        int64_t w[2] = {7, 9}; CK(uspmv_hostcomm_bcast(hc, w, sizeof w, 0));
        uspmv_transport_t tr; CK(uspmv_hostcomm_transport(hc, &tr));
        uspmv_coo_t *one = nullptr; CK(uspmv_gen_stencil27(6, 5, 4, 1, 0x5EED, 0.0, 0, 120, &one));
        int32_t wsa1[2] = {0, 120};
        uspmv_scs_t *s1 = nullptr; CK(uspmv_convert_to_scs(one, 32, 64, USPMV_F64, nullptr, &s1));
        uspmv_halo_t *h1 = nullptr; CK(uspmv_halo_discover(s1, wsa1, 0, 1, &h1));
        uspmv_comm_plan_t *cp = nullptr; CK(uspmv_comm_plan_create(&tr, h1, &cp));
        uspmv_comm_plan_free(cp); uspmv_halo_free(h1); uspmv_scs_free(s1); uspmv_coo_free(one);
        uspmv_hostcomm_free(hc);
    }
    return 0;
}
