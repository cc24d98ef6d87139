/* Plain-C client of include/mfgp.h: the drop-in boundary exercised from a host language other than Python.
 * Reads a case from a text file, runs fit-arithmetic + predict through the C-ABI, prints the numbers.
 *   abi_client <case.txt>
 * case.txt:  N D n_parts Ns  /  parts (type c0 c1 term) x n_parts  /  theta (2*n_parts)  noise  /  X (N*D)  /  Y (N)  /  Xs (Ns*D)
 * Built and run by tests/test_gpu_c_abi.py (gcc -I include ... -L <package dir> -lmfgp_hip). */
#include <stdio.h>
#include <stdlib.h>
#include "mfgp.h"

static void die(mfgp_handle* h, const char* what, int rc) {
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, mfgp_last_error(h));
    exit(2);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s case.txt\n", argv[0]); return 1; }
    FILE* f = fopen(argv[1], "r");
    if (!f) { perror("case file"); return 1; }
    long N; int D, P; long Ns;
    if (fscanf(f, "%ld %d %d %ld", &N, &D, &P, &Ns) != 4) return 1;
    mfgp_kern_part parts[MFGP_MAX_PARTS];
    for (int p = 0; p < P; ++p)
        if (fscanf(f, "%d %d %d %d", &parts[p].type, &parts[p].col_begin, &parts[p].col_end, &parts[p].term) != 4) return 1;
    double theta[2 * MFGP_MAX_PARTS], noise;
    for (int p = 0; p < 2 * P; ++p) if (fscanf(f, "%lf", &theta[p]) != 1) return 1;
    if (fscanf(f, "%lf", &noise) != 1) return 1;
    double* X = malloc(sizeof(double) * N * D);
    double* Y = malloc(sizeof(double) * N);
    double* Xs = malloc(sizeof(double) * Ns * D);
    for (long i = 0; i < N * D; ++i) if (fscanf(f, "%lf", &X[i]) != 1) return 1;
    for (long i = 0; i < N; ++i) if (fscanf(f, "%lf", &Y[i]) != 1) return 1;
    for (long i = 0; i < Ns * D; ++i) if (fscanf(f, "%lf", &Xs[i]) != 1) return 1;
    fclose(f);

    mfgp_handle* h = NULL;
    int rc = mfgp_create(0, &h);
    if (rc) die(NULL, "mfgp_create", rc);
    printf("device %s\n", mfgp_device_info(h));
    if ((rc = mfgp_set_data(h, X, N, D, Y))) die(h, "mfgp_set_data", rc);
    if ((rc = mfgp_set_kernel(h, parts, P))) die(h, "mfgp_set_kernel", rc);
    double nlml, grad[2 * MFGP_MAX_PARTS + 1];
    if ((rc = mfgp_eval(h, theta, noise, 1e-8, 1, &nlml, grad))) die(h, "mfgp_eval", rc);
    printf("nlml %.17g\n", nlml);
    for (int p = 0; p <= 2 * P; ++p) printf("grad %.17g\n", grad[p]);
    double* mean = malloc(sizeof(double) * Ns);
    double* var = malloc(sizeof(double) * Ns);
    if ((rc = mfgp_predict(h, Xs, Ns, mean, var, 1, 1))) die(h, "mfgp_predict", rc);
    for (long i = 0; i < Ns; ++i) printf("pred %.17g %.17g\n", mean[i], var[i]);
    /* round 4: three hyper-parameter points in ONE batched pass (mfgp_eval_batch) -- set 0 is the point above: its numbers must be
     * the very same bits; and the build id of the library (the hash profiles/ files are stamped with) */
    {
        const int np_ = mfgp_num_params(parts, P);
        double thetas[3 * 2 * MFGP_MAX_PARTS], noises[3], jit[3] = {1e-8, 1e-8, 1e-8}, f[3], g[3 * (2 * MFGP_MAX_PARTS + 1)];
        int32_t st[3];
        for (int b = 0; b < 3; ++b) {
            noises[b] = noise * (1.0 + 0.5 * b);
            for (int i = 0; i < np_; ++i) thetas[b * np_ + i] = theta[i] * (1.0 + 0.1 * b);
        }
        if ((rc = mfgp_eval_batch(h, 3, thetas, noises, jit, 1, f, g, st))) die(h, "mfgp_eval_batch", rc);
        int same = (f[0] == nlml) && st[0] == 0 && st[1] == 0 && st[2] == 0;
        for (int i = 0; i <= np_; ++i) same = same && (g[i] == grad[i]);
        printf("batch_set0_bitwise %d\n", same);
        printf("batch_nlml %.17g %.17g %.17g\n", f[0], f[1], f[2]);
        if ((rc = mfgp_predict(h, Xs, Ns, mean, var, 1, 1))) die(h, "mfgp_predict after batch", rc);   /* the handle's own factorisation survives */
        printf("build_id %s\n", mfgp_build_id());
    }
    /* error behaviour: a call that cannot succeed reports, it does not crash */
    rc = mfgp_predict(h, NULL, Ns, mean, var, 1, 1);
    printf("null_predict_rc %d (%s)\n", rc, mfgp_last_error(h));
    mfgp_destroy(h);
    free(X); free(Y); free(Xs); free(mean); free(var);
    return 0;
}
