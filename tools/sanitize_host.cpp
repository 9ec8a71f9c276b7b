// sanitizer pass over the host data layer (no HIP): every golden matrix through read / convert / plan / halo / split
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "uspmv.h"
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
                    uspmv_free(ia); uspmv_free(ib); uspmv_halo_free(h); uspmv_scs_free(s); uspmv_coo_free(loc);
                }
            }
        }
        uspmv_coo_t *dp, *sp; CK(uspmv_partition_precisions(m, 1.0, &dp, &sp)); uspmv_coo_free(dp); uspmv_coo_free(sp);
        std::string f = std::string("/tmp/uspmv_sanitize_cache.bin"); CK(uspmv_coo_save(m, f.c_str()));
        uspmv_coo_t *m2; CK(uspmv_coo_load(f.c_str(), &m2)); uspmv_coo_free(m2);
        if (n <= nc) CK(uspmv_coo_equilibrate(m));
        uspmv_coo_free(m);
        printf("ok %s\n", argv[a]);
    }
    uspmv_coo_t *g = nullptr; CK(uspmv_gen_stencil27(9, 8, 7, 3, 0x5EED, 4.0, 10, 900, &g)); uspmv_coo_free(g);
    return 0;
}
