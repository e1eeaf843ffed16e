/*
 * remap_oracle.c -- CPU restatement of the pyremap weight-application path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and there only as the checker / reported CPU baseline.
 * The product path (pyremap_amd) never imports, links or calls it.
 *
 * Parity status: PINNED.  oracle/make_goldens.py runs the reference's own
 * `_load_mapping` / `_remap_numpy_array` / `_remap_numpy`
 * (/root/reference/pyremap/remapper/remap_numpy.py, imported by file path in
 * the build container) on seeded inputs and stores inputs + outputs under
 * tests/golden/*.npz; tests/test_oracle_golden.py checks this restatement
 * bit-for-bit against every one of them.
 *
 * What is restated, with the reference lines each function follows
 * (paths relative to /root/reference):
 *
 *  oracle_coo_to_csr   pyremap/remapper/remap_numpy.py:134-137
 *        `csr_matrix((S, (row-1, col-1)), shape=(n_b, n_a))`.  The arithmetic
 *        lives in scipy (unpinned in pyproject.toml:31; 1.15.3 in the build
 *        container): coo -> csr is a stable counting sort by row, then
 *        `sum_duplicates()` = per-row sort by column + left-to-right sum of
 *        equal (row, col) entries; explicit zeros are kept.  Restated here as
 *        stable sort by (row, col) + sequential sum in input order, which is
 *        what scipy yields whenever a row holds <= 16 entries (std::sort is
 *        an insertion sort there, hence stable) or a (row, col) key repeats
 *        at most twice (a + b == b + a).
 *  oracle_csr_matvecs  the `matrix.dot(dense)` calls at remap_numpy.py:264,
 *        265, 268 -> scipy sparsetools `csr_matvecs`: Y zeroed, then for each
 *        row, for jj in CSR order, y[k] += a * x[k] -- a separate multiply
 *        and add (no FMA in the x86-64 wheels), single thread.
 *  oracle_remap_flat   remap_numpy.py:258-278: masked / unmasked branches,
 *        `mask = den > thr` (or `frac_b > 0.0`), `out[mask] /= den[mask]`,
 *        result masked where ~mask (the undivided value stays underneath).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no -ffast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- COO -> CSR -------------------------------------------------------- */

typedef struct {
    int32_t col;
    int64_t pos; /* position in the input: tie-break keeps the sort stable */
} oracle_key_t;

static int oracle_key_cmp(const void *pa, const void *pb)
{
    const oracle_key_t *a = (const oracle_key_t *)pa;
    const oracle_key_t *b = (const oracle_key_t *)pb;
    if (a->col != b->col)
        return a->col < b->col ? -1 : 1;
    if (a->pos != b->pos)
        return a->pos < b->pos ? -1 : 1;
    return 0;
}

/*
 * row, col: 0-based.  indptr has n_rows + 1 entries; indices / data must have
 * room for nnz entries; *nnz_out receives the number kept after duplicates
 * are summed.  Returns 0, or -1 on a bad index, -2 on allocation failure.
 */
int oracle_coo_to_csr(int64_t n_rows, int64_t n_cols, int64_t nnz,
                      const int32_t *row, const int32_t *col, const double *S,
                      int64_t *indptr, int32_t *indices, double *data,
                      int64_t *nnz_out)
{
    int64_t i, n, k;
    int64_t *fill;
    oracle_key_t *keys;

    for (n = 0; n < nnz; n++) {
        if (row[n] < 0 || row[n] >= n_rows || col[n] < 0 || col[n] >= n_cols)
            return -1;
    }
    fill = (int64_t *)calloc((size_t)n_rows + 1, sizeof(int64_t));
    keys = (oracle_key_t *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(*keys));
    if (!fill || !keys) {
        free(fill);
        free(keys);
        return -2;
    }
    /* stable counting sort by row */
    for (n = 0; n < nnz; n++)
        fill[row[n] + 1]++;
    for (i = 0; i < n_rows; i++)
        fill[i + 1] += fill[i];
    memcpy(indptr, fill, ((size_t)n_rows + 1) * sizeof(int64_t));
    for (n = 0; n < nnz; n++) {
        int64_t dst = fill[row[n]]++;
        keys[dst].col = col[n];
        keys[dst].pos = n;
    }
    /* per-row stable sort by column, then sum runs of equal columns */
    k = 0;
    for (i = 0; i < n_rows; i++) {
        int64_t a = indptr[i], b = indptr[i + 1], jj;
        qsort(keys + a, (size_t)(b - a), sizeof(*keys), oracle_key_cmp);
        indptr[i] = k;
        jj = a;
        while (jj < b) {
            int32_t c = keys[jj].col;
            double v = S[keys[jj].pos];
            jj++;
            while (jj < b && keys[jj].col == c) {
                v = v + S[keys[jj].pos];
                jj++;
            }
            indices[k] = c;
            data[k] = v;
            k++;
        }
    }
    indptr[n_rows] = k;
    *nnz_out = k;
    free(fill);
    free(keys);
    return 0;
}

/* ---- CSR x dense ------------------------------------------------------- */

/*
 * Y (n_row, K) = A (CSR) . X (n_col, K), both C-order.  Sequential in jj,
 * separate multiply and add, exactly as scipy's csr_matvecs.  nthreads > 1
 * splits ROWS over OpenMP threads (each row is still summed sequentially, so
 * the result does not depend on the thread count).
 */
void oracle_csr_matvecs(int64_t n_row, int64_t K, const int64_t *Ap,
                        const int32_t *Aj, const double *Ax, const double *X,
                        double *Y, int nthreads)
{
    int64_t i;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (i = 0; i < n_row; i++) {
        double *y = Y + K * i;
        int64_t jj, k;
        for (k = 0; k < K; k++)
            y[k] = 0.0;
        for (jj = Ap[i]; jj < Ap[i + 1]; jj++) {
            const double a = Ax[jj];
            const double *x = X + K * (int64_t)Aj[jj];
            for (k = 0; k < K; k++)
                y[k] = y[k] + a * x[k];
        }
    }
}

/* ---- the flat (n_a, K) -> (n_b, K) part of _remap_numpy_array ---------- */

/*
 * masked != 0 : remap_numpy.py:262-266.  valid = !isnan(X);
 *               num = A.(valid ? X : +0.0); den = A.(valid ? 1.0 : 0.0);
 *               ok = den > thr.
 * masked == 0 : remap_numpy.py:268-274.  num = A.X (NaNs propagate);
 *               den = frac_b[i]; ok = den > 0.0.
 * Then (277-278) out = ok ? num / den : num (undivided), mask_out = !ok.
 * mask_out uses numpy.ma's convention: 1 = masked (invalid).
 */
void oracle_remap_flat_mask(int64_t n_b, int64_t K, const int64_t *Ap,
                            const int32_t *Aj, const double *Ax,
                            const double *frac_b, const double *X,
                            const uint8_t *in_masked, int masked, double thr,
                            double *out, uint8_t *mask_out, int nthreads);

void oracle_remap_flat(int64_t n_b, int64_t K, const int64_t *Ap,
                       const int32_t *Aj, const double *Ax,
                       const double *frac_b, const double *X, int masked,
                       double thr, double *out, uint8_t *mask_out,
                       int nthreads)
{
    oracle_remap_flat_mask(n_b, K, Ap, Aj, Ax, frac_b, X, NULL, masked, thr,
                           out, mask_out, nthreads);
}

/*
 * The same with the input mask given EXPLICITLY (in_masked[a * K + k] != 0 =
 * masked, numpy.ma's convention) instead of read off the NaNs: the general
 * form of remap_numpy.py:262-266, where
 *     in_mask = array(logical_not(in_field.mask), float)
 *     num = matrix.dot(in_mask * in_field)      den = matrix.dot(in_mask)
 * `in_mask * in_field` holds +0.0 under the mask (numpy.ma keeps the FIRST
 * operand's data there) and 1.0 * data elsewhere -- so a NaN that is NOT
 * masked reaches the product and makes every destination cell it touches
 * NaN (unmasked), while den counts it as valid.  in_masked == NULL: the mask
 * is isnan(X), which is what _remap_data_array builds (:201-204).
 */
void oracle_remap_flat_mask(int64_t n_b, int64_t K, const int64_t *Ap,
                            const int32_t *Aj, const double *Ax,
                            const double *frac_b, const double *X,
                            const uint8_t *in_masked, int masked, double thr,
                            double *out, uint8_t *mask_out, int nthreads)
{
    int64_t i;
    (void)nthreads;
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
#endif
    {
        double *den = NULL;
        if (masked)
            den = (double *)malloc((size_t)(K > 0 ? K : 1) * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (i = 0; i < n_b; i++) {
            double *y = out + K * i;
            uint8_t *m = mask_out + K * i;
            int64_t jj, k;
            for (k = 0; k < K; k++)
                y[k] = 0.0;
            if (masked) {
                for (k = 0; k < K; k++)
                    den[k] = 0.0;
                for (jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                    const double a = Ax[jj];
                    const double *x = X + K * (int64_t)Aj[jj];
                    const uint8_t *im =
                        in_masked ? in_masked + K * (int64_t)Aj[jj] : NULL;
                    for (k = 0; k < K; k++) {
                        const int valid = im ? !im[k] : !isnan(x[k]);
                        const double xv = valid ? x[k] : 0.0;
                        const double mv = valid ? 1.0 : 0.0;
                        y[k] = y[k] + a * xv;
                        den[k] = den[k] + a * mv;
                    }
                }
                for (k = 0; k < K; k++) {
                    const int ok = den[k] > thr;
                    if (ok)
                        y[k] = y[k] / den[k];
                    m[k] = (uint8_t)!ok;
                }
            } else {
                const double d = frac_b[i];
                const int ok = d > 0.0;
                for (jj = Ap[i]; jj < Ap[i + 1]; jj++) {
                    const double a = Ax[jj];
                    const double *x = X + K * (int64_t)Aj[jj];
                    for (k = 0; k < K; k++)
                        y[k] = y[k] + a * x[k];
                }
                for (k = 0; k < K; k++) {
                    if (ok)
                        y[k] = y[k] / d;
                    m[k] = (uint8_t)!ok;
                }
            }
        }
        free(den);
    }
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
