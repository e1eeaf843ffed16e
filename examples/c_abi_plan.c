/*
 * The C ABI from plain C (no Python, no C++): build a plan from a mapping's
 * triplets, remap a batch of fields, check every value against the
 * sequential multiply-then-add scipy's csr_matvecs performs
 * (pyremap/remapper/remap_numpy.py:134-137 and :258-278).
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude \
 *       examples/c_abi_plan.c -o c_abi_plan \
 *       -Lpyremap_amd/_lib -lremap_hip -L/opt/rocm/lib -lamdhip64 \
 *       -Wl,-rpath,$PWD/pyremap_amd/_lib -Wl,-rpath,/opt/rocm/lib -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "remap_hip.h"

#define CHECK_HIP(x)                                                      \
    do {                                                                  \
        hipError_t e_ = (x);                                              \
        if (e_ != hipSuccess) {                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));       \
            return 2;                                                     \
        }                                                                 \
    } while (0)
#define CHECK_REMAP(x)                                                    \
    do {                                                                  \
        if ((x) != REMAP_OK) {                                            \
            fprintf(stderr, "%s: %s\n", #x, remap_last_error());          \
            return 3;                                                     \
        }                                                                 \
    } while (0)

int main(void)
{
    /* a 12 x 16 destination grid fed by 300 source cells: every row gets 1-5
     * entries with pseudo-random columns and weights, one row stays empty,
     * one has frac_b = 0; entries are given unsorted, 1-based, with one
     * duplicate (summed by the library, as scipy does) */
    enum { NA = 300, MY = 12, MX = 16, NB = MY * MX, K = 160 };
    static int32_t row[NB * 5 + 1], col[NB * 5 + 1];
    static double S[NB * 5 + 1], frac_b[NB], X[NA * K], Y[NB * K], ref[NB * K];
    uint64_t state = 88172645463325252ull;
    int64_t n_s = 0;
    for (int i = NB - 1; i >= 0; --i) {          /* rows in descending order */
        int n = 1 + (int)((state = state * 6364136223846793005ull + 1442695040888963407ull) >> 33) % 5;
        if (i == 7)
            n = 0;
        double sum = 0.0;
        for (int j = 0; j < n; ++j) {
            state = state * 6364136223846793005ull + 1442695040888963407ull;
            row[n_s] = i + 1;
            col[n_s] = 1 + (int32_t)((i * 3 + j * 37 + (state >> 40)) % NA);
            S[n_s] = 0.1 + (double)((state >> 20) & 0xffff) / 65536.0;
            sum += S[n_s];
            ++n_s;
        }
        frac_b[i] = i == 11 ? 0.0 : (n ? sum : 0.5);
    }
    row[n_s] = row[0]; col[n_s] = col[0]; S[n_s] = 0.25; ++n_s;  /* duplicate */
    for (int i = 0; i < NA * K; ++i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        X[i] = (double)(int64_t)(state >> 11) / 9007199254740992.0 - 0.5;
    }

    /* the expected result: CSR order = ascending column within a row, equal
     * (row, col) summed in input order; y += a * x one entry after the other */
    memset(ref, 0, sizeof(ref));
    for (int i = 0; i < NB; ++i) {
        int32_t cols[8];
        double w[8];
        int n = 0;
        for (int64_t t = 0; t < n_s; ++t) {
            if (row[t] != i + 1)
                continue;
            int at = -1;
            for (int q = 0; q < n; ++q)
                if (cols[q] == col[t])
                    at = q;
            if (at >= 0) {
                w[at] += S[t];
            } else {
                cols[n] = col[t];
                w[n++] = S[t];
            }
        }
        for (int a = 1; a < n; ++a)                /* insertion sort by col */
            for (int b = a; b > 0 && cols[b - 1] > cols[b]; --b) {
                int32_t tc = cols[b]; cols[b] = cols[b - 1]; cols[b - 1] = tc;
                double tw = w[b]; w[b] = w[b - 1]; w[b - 1] = tw;
            }
        for (int k = 0; k < K; ++k) {
            double acc = 0.0;
            for (int q = 0; q < n; ++q) {
                volatile double prod = w[q] * X[(cols[q] - 1) * K + k];
                acc = acc + prod;
            }
            ref[i * K + k] = frac_b[i] > 0.0 ? acc / frac_b[i] : NAN;
        }
    }

    remap_plan *plan = NULL;
    const int64_t dims[2] = {MY, MX};
    CHECK_REMAP(remap_plan_create(NB, NA, n_s, row, col, S, 1, frac_b,
                                  1 /* host arrays */, dims, 2, NULL, &plan));
    remap_plan_info info;
    CHECK_REMAP(remap_plan_query(plan, &info));
    double *dX, *dY;
    CHECK_HIP(hipMalloc((void **)&dX, sizeof(X)));
    CHECK_HIP(hipMalloc((void **)&dY, sizeof(Y)));
    CHECK_HIP(hipMemcpy(dX, X, sizeof(X), hipMemcpyHostToDevice));
    remap_field f;
    memset(&f, 0, sizeof(f));
    f.X = dX;
    f.x_dtype = REMAP_DTYPE_F64;
    f.mode = REMAP_MODE_FRACB;
    f.n_batch = 1;
    f.k_inner = K;
    f.x_row_stride = f.y_row_stride = K;
    f.Y = dY;
    CHECK_REMAP(remap_plan_apply(plan, &f, NULL));
    CHECK_HIP(hipMemcpy(Y, dY, sizeof(Y), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < NB * K; ++i) {
        const int both_nan = isnan(Y[i]) && isnan(ref[i]);
        if (!both_nan && memcmp(&Y[i], &ref[i], 8) != 0)
            ++bad;
    }
    printf("%s (ABI %d): %lld triplets -> %lld entries, schedule family %d, "
           "%d x %d values, %d differ\n", remap_arch(), remap_abi_version(),
           (long long)n_s, (long long)info.nnz, info.family, NB, K, bad);

    /* the same fields laid out (Time, nCells) -- MPAS's 2-D time series, the
     * reference's most common input, which it flattens with a transpose copy
     * (remap_numpy.py:254-256).  Here: addressed in place, n_batch = Time,
     * k_inner = 1, through the LDS-staged lanes-across-rows kernel once its
     * patch plan has been prepared; the output is (Time, lat, lon). */
    static double Xt[K * NA], Yt[K * NB];
    for (int a = 0; a < NA; ++a)
        for (int k = 0; k < K; ++k)
            Xt[k * NA + a] = X[a * K + k];
    CHECK_REMAP(remap_plan_prepare_short_runs(plan, NULL));
    CHECK_HIP(hipMemcpy(dX, Xt, sizeof(Xt), hipMemcpyHostToDevice));
    f.n_batch = K;
    f.k_inner = 1;
    f.x_row_stride = f.y_row_stride = 1;
    f.x_batch_stride = NA;
    f.y_batch_stride = NB;
    CHECK_REMAP(remap_plan_apply(plan, &f, NULL));
    CHECK_HIP(hipMemcpy(Yt, dY, sizeof(Yt), hipMemcpyDeviceToHost));
    int bad_t = 0;
    for (int i = 0; i < NB; ++i)
        for (int k = 0; k < K; ++k) {
            const double y = Yt[k * NB + i], r = ref[i * K + k];
            if (!(isnan(y) && isnan(r)) && memcmp(&y, &r, 8) != 0)
                ++bad_t;
        }
    printf("(Time = %d, nCells) in place: %d differ\n", K, bad_t);
    bad += bad_t;
    remap_plan_destroy(plan);
    hipFree(dX);
    hipFree(dY);
    return bad ? 1 : 0;
}
