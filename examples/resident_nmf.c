/* examples/resident_nmf.c -- the C ABI from plain C: keep A resident in HBM and factor it several
 * times (the pattern hierclust / parameter sweeps use).
 *
 *   gcc -std=c99 -Iinclude examples/resident_nmf.c -o resident_nmf \
 *       -Lsmallk_amd/lib -lsmallk_amd -Wl,-rpath,$PWD/smallk_amd/lib
 *   ./resident_nmf            (needs an MI355X)
 */
#include <stdio.h>
#include <stdlib.h>

#include "smallk_amd.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != SMK_OK) {                                                     \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, smk_last_error());    \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void)
{
    const int m = 4096, n = 2048;
    CHECK(smk_initialize(-1));

    /* A lives on the device; here it is generated there (a host buffer would go through
     * smk_matrix_upload_f64).  SMK_STORE_BF16 halves the bytes every iteration streams. */
    smk_matrix* a = NULL;
    CHECK(smk_matrix_create(&a, m, n, 0, n, SMK_STORE_F32));
    CHECK(smk_matrix_fill_uniform(a, 42));

    const int ranks[3] = {8, 16, 32};
    for (int t = 0; t < 3; ++t) {
        const int k = ranks[t];
        smk_options o = {0};
        o.tol = 0.005; o.algorithm = SMK_ALG_HALS; o.prog_est_algorithm = SMK_PROG_PG_RATIO;
        o.height = m; o.width = n; o.k = k;
        o.min_iter = 5; o.max_iter = 200; o.tolcount = 1; o.normalize = 1;

        double* W = (double*)malloc(sizeof(double) * (size_t)m * k);
        double* H = (double*)malloc(sizeof(double) * (size_t)k * n);
        smk_uniform_fill_host(W, m, m, k, 0, 0, m, 1, 0);
        smk_uniform_fill_host(H, k, k, n, 0, 0, k, 2, 0);
        for (long i = 0; i < (long)k * n; ++i) H[i] *= 2.0 / k;      /* E[W H] = E[A] */

        smk_solver* s = NULL;
        smk_stats st = {0, 0};
        CHECK(smk_solver_create(&s, &o, a));
        CHECK(smk_solver_set_factors(s, W, m, H, k));
        int rc = smk_solver_run(s, &st);                              /* NmfSolve with the stopping rule */
        if (rc != SMK_OK && rc != SMK_FAILURE) { fprintf(stderr, "run: %s\n", smk_last_error()); return 1; }
        CHECK(smk_solver_get_factors(s, 0, W, m, H, k));
        double metric = 0.0;
        smk_solver_progress(s, &metric);
        printf("k = %2d: %s after %d iterations, %.2f ms, projected-gradient ratio %.3g, W[0][0] = %.6f\n", k,
               rc == SMK_OK ? "converged" : "stopped", st.iteration_count, st.elapsed_us / 1000.0, metric, W[0]);
        smk_solver_destroy(s);
        free(W);
        free(H);
    }
    smk_matrix_destroy(a);
    smk_finalize();
    return 0;
}
