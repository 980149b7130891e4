/*
 * oracle/libdistance_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the arithmetic of the reference's `libdistance`
 * vector-metric path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this file's shared object; the product path
 * (msmbuilder_amd/) never does.
 *
 * Parity status: PINNED.  Checked bit-for-bit against oracle/_ref (the
 * reference's own headers compiled by oracle/Makefile) in
 * tests/test_oracle_libdistance.py and against the committed golden vectors
 * in tests/golden/libdistance_*.npz.
 *
 * What is restated (all paths relative to /root/reference/msmbuilder/libdistance):
 *   metric kernels        src/distance_kernels.h:41-243
 *   metric dispatch       src/distance_kernels.h:245-293
 *   dist / dist_X_indices src/dist.hpp:4-80
 *   cdist                 src/cdist.hpp:4-49
 *   assign_nearest        src/assign.hpp:6-91
 *   pdist / _X_indices    src/pdist.hpp:4-88
 *   sumdist               src/sumdist.hpp:4-44
 *
 * Arithmetic contract being pinned (this is what "bit-exact labels" means):
 *   - float inputs: `u[i] - v[i]` and `u[i] + v[i]` are evaluated in FLOAT
 *     (both operands float), then widened to double.  The reference's
 *     translation unit is Cython C++ (setup.py:146-147): Python.h and numpy's
 *     headers come first, so <math.h> is libstdc++'s C++ wrapper and the
 *     global-namespace `fabs(float)` the kernels call is the FLOAT overload:
 *     canberra's `fabs(u[i]) + fabs(v[i])` is a FLOAT add, widened afterwards
 *     (pinned against oracle/_ref built in that context -- ref_shim.cpp
 *     includes Python.h + numpy/npy_common.h: a double add mismatches, a float
 *     add is bit-identical.  A translation unit that only sees C's <math.h>,
 *     as round 2's hand-typedef'd shim did, gets the double add).
 *     Everything accumulates in ONE double accumulator, features visited in
 *     order i = 0..n-1, multiply and add rounded separately.
 *   - euclidean = sqrt(sqeuclidean), compared AFTER the sqrt.
 *   - assign_nearest: min_d starts at DBL_MAX, strict `d < min_d`, so the
 *     lowest centre index wins ties and an all-NaN row keeps assignment 0;
 *     inertia is the sequential double sum of min_d.
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile); no
 * -ffast-math, no -march=native so mul/add stay separately rounded.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

typedef int64_t idx_t; /* npy_intp on LP64 */

enum {
    M_EUCLIDEAN = 0,
    M_SQEUCLIDEAN,
    M_CITYBLOCK,
    M_CHEBYSHEV,
    M_CANBERRA,
    M_BRAYCURTIS,
    M_HAMMING,
    M_JACCARD,
    M_UNKNOWN = -1
};

/* distance_kernels.h:245-268 (string dispatch; unknown name -> NULL) */
int oracle_metric_id(const char *metric)
{
    if (strcmp(metric, "euclidean") == 0) return M_EUCLIDEAN;
    if (strcmp(metric, "sqeuclidean") == 0) return M_SQEUCLIDEAN;
    if (strcmp(metric, "cityblock") == 0) return M_CITYBLOCK;
    if (strcmp(metric, "chebyshev") == 0) return M_CHEBYSHEV;
    if (strcmp(metric, "canberra") == 0) return M_CANBERRA;
    if (strcmp(metric, "braycurtis") == 0) return M_BRAYCURTIS;
    if (strcmp(metric, "hamming") == 0) return M_HAMMING;
    if (strcmp(metric, "jaccard") == 0) return M_JACCARD;
    return M_UNKNOWN;
}

/* ---- double kernels: distance_kernels.h:41-52,67-71,79-93,109-123,141-152,
 *      167-177,191-202,218-229 ---- */
static double metric_f64(int m, const double *u, const double *v, idx_t n)
{
    idx_t i;
    switch (m) {
    case M_EUCLIDEAN:
    case M_SQEUCLIDEAN: {
        double s = 0.0, d;
        for (i = 0; i < n; i++) {
            d = u[i] - v[i];
            s += d * d;
        }
        return m == M_EUCLIDEAN ? sqrt(s) : s;
    }
    case M_CITYBLOCK: {
        double s = 0.0, d;
        for (i = 0; i < n; i++) {
            d = fabs(u[i] - v[i]);
            s = s + d;
        }
        return s;
    }
    case M_CHEBYSHEV: {
        double d, maxv = 0.0;
        for (i = 0; i < n; i++) {
            d = fabs(u[i] - v[i]);
            if (d > maxv) maxv = d;
        }
        return maxv;
    }
    case M_CANBERRA: {
        double snum, sdenom, tot = 0.0;
        for (i = 0; i < n; i++) {
            snum = fabs(u[i] - v[i]);
            sdenom = fabs(u[i]) + fabs(v[i]);
            if (sdenom > 0.0) tot += snum / sdenom;
        }
        return tot;
    }
    case M_BRAYCURTIS: {
        double s1 = 0.0, s2 = 0.0;
        for (i = 0; i < n; i++) {
            s1 += fabs(u[i] - v[i]);
            s2 += fabs(u[i] + v[i]);
        }
        return s1 / s2;
    }
    case M_HAMMING: {
        double s = 0.0;
        for (i = 0; i < n; i++) s += (u[i] != v[i]);
        return s / n;
    }
    case M_JACCARD: {
        double denom = 0.0, num = 0.0;
        for (i = 0; i < n; i++) {
            num += (u[i] != v[i]) & ((u[i] != 0.0) | (v[i] != 0.0));
            denom += (u[i] != 0.0) | (v[i] != 0.0);
        }
        return num / denom;
    }
    }
    return NAN;
}

/* ---- float kernels: distance_kernels.h:54-65,73-77,95-107,125-139,154-165,
 *      179-189,204-215,231-242.  Differences/sums of two floats are FLOAT. ---- */
static double metric_f32(int m, const float *u, const float *v, idx_t n)
{
    idx_t i;
    switch (m) {
    case M_EUCLIDEAN:
    case M_SQEUCLIDEAN: {
        double s = 0.0, d;
        for (i = 0; i < n; i++) {
            float df = u[i] - v[i];
            d = df;
            s += d * d;
        }
        return m == M_EUCLIDEAN ? sqrt(s) : s;
    }
    case M_CITYBLOCK: {
        double s = 0.0, d;
        for (i = 0; i < n; i++) {
            float df = fabsf(u[i] - v[i]);
            d = df;
            s = s + d;
        }
        return s;
    }
    case M_CHEBYSHEV: {
        double d, maxv = 0.0;
        for (i = 0; i < n; i++) {
            float df = fabsf(u[i] - v[i]);
            d = df;
            if (d > maxv) maxv = d;
        }
        return maxv;
    }
    case M_CANBERRA: {
        double snum, sdenom, tot = 0.0;
        for (i = 0; i < n; i++) {
            float fn = fabsf(u[i] - v[i]);
            snum = fn;
            sdenom = (double)(float)(fabsf(u[i]) + fabsf(v[i])); /* FLOAT add: pinned vs oracle/_ref */
            if (sdenom > 0.0) tot += snum / sdenom;
        }
        return tot;
    }
    case M_BRAYCURTIS: {
        double s1 = 0.0, s2 = 0.0;
        for (i = 0; i < n; i++) {
            float a = fabsf(u[i] - v[i]);
            float b = fabsf(u[i] + v[i]);
            s1 += (double)a;
            s2 += (double)b;
        }
        return s1 / s2;
    }
    case M_HAMMING: {
        double s = 0.0;
        for (i = 0; i < n; i++) s += (u[i] != v[i]);
        return s / n;
    }
    case M_JACCARD: {
        double denom = 0.0, num = 0.0;
        for (i = 0; i < n; i++) {
            num += (u[i] != v[i]) & ((u[i] != 0.0) | (v[i] != 0.0));
            denom += (u[i] != 0.0) | (v[i] != 0.0);
        }
        return num / denom;
    }
    }
    return NAN;
}

/* dist.hpp:4-22 / 44-60 and the _X_indices forms :24-41 / :62-80.
 * Returns 0, or -1 for an unknown metric (the reference prints "Error" and
 * leaves `out` untouched). */
int oracle_dist_f64(const double *X, const double *y, const char *metric, idx_t n, idx_t m,
                    const idx_t *X_indices, idx_t n_X_indices, double *out)
{
    int id = oracle_metric_id(metric);
    idx_t i;
    if (id < 0) return -1;
    if (X_indices == NULL) {
        for (i = 0; i < n; i++) out[i] = metric_f64(id, X + m * i, y, m);
    } else {
        for (i = 0; i < n_X_indices; i++) out[i] = metric_f64(id, X + m * X_indices[i], y, m);
    }
    return 0;
}

int oracle_dist_f32(const float *X, const float *y, const char *metric, idx_t n, idx_t m,
                    const idx_t *X_indices, idx_t n_X_indices, double *out)
{
    int id = oracle_metric_id(metric);
    idx_t i;
    if (id < 0) return -1;
    if (X_indices == NULL) {
        for (i = 0; i < n; i++) out[i] = metric_f32(id, X + m * i, y, m);
    } else {
        for (i = 0; i < n_X_indices; i++) out[i] = metric_f32(id, X + m * X_indices[i], y, m);
    }
    return 0;
}

/* cdist.hpp:4-26 / 28-49: out[i*nb + j] */
int oracle_cdist_f64(const double *XA, const double *XB, const char *metric, idx_t na, idx_t nb,
                     idx_t m, double *out)
{
    int id = oracle_metric_id(metric);
    idx_t i, j, k = 0;
    if (id < 0) return -1;
    for (i = 0; i < na; i++)
        for (j = 0; j < nb; j++) out[k++] = metric_f64(id, XA + m * i, XB + m * j, m);
    return 0;
}

int oracle_cdist_f32(const float *XA, const float *XB, const char *metric, idx_t na, idx_t nb,
                     idx_t m, double *out)
{
    int id = oracle_metric_id(metric);
    idx_t i, j, k = 0;
    if (id < 0) return -1;
    for (i = 0; i < na; i++)
        for (j = 0; j < nb; j++) out[k++] = metric_f32(id, XA + m * i, XB + m * j, m);
    return 0;
}

/* pdist.hpp: condensed upper triangle, row by row; rows are X_indices[ii] when given. */
int oracle_pdist_f64(const double *X, const char *metric, idx_t n, idx_t m, const idx_t *X_indices,
                     idx_t n_X_indices, double *out)
{
    int id = oracle_metric_id(metric);
    idx_t ii, jj, k = 0, nn = X_indices == NULL ? n : n_X_indices;
    if (id < 0) return -1;
    for (ii = 0; ii < nn; ii++)
        for (jj = ii + 1; jj < nn; jj++) {
            idx_t i = X_indices == NULL ? ii : X_indices[ii], j = X_indices == NULL ? jj : X_indices[jj];
            out[k++] = metric_f64(id, X + m * i, X + m * j, m);
        }
    return 0;
}

int oracle_pdist_f32(const float *X, const char *metric, idx_t n, idx_t m, const idx_t *X_indices,
                     idx_t n_X_indices, double *out)
{
    int id = oracle_metric_id(metric);
    idx_t ii, jj, k = 0, nn = X_indices == NULL ? n : n_X_indices;
    if (id < 0) return -1;
    for (ii = 0; ii < nn; ii++)
        for (jj = ii + 1; jj < nn; jj++) {
            idx_t i = X_indices == NULL ? ii : X_indices[ii], j = X_indices == NULL ? jj : X_indices[jj];
            out[k++] = metric_f32(id, X + m * i, X + m * j, m);
        }
    return 0;
}

/* sumdist.hpp: sequential double sum over the listed pairs; -1 for an unknown metric */
double oracle_sumdist_f64(const double *X, const char *metric, idx_t n, idx_t m, const idx_t *pairs, idx_t p)
{
    int id = oracle_metric_id(metric);
    idx_t i;
    double s = 0;
    (void)n;
    if (id < 0) return -1;
    for (i = 0; i < p; i++) s += metric_f64(id, X + m * pairs[2 * i], X + m * pairs[2 * i + 1], m);
    return s;
}

double oracle_sumdist_f32(const float *X, const char *metric, idx_t n, idx_t m, const idx_t *pairs, idx_t p)
{
    int id = oracle_metric_id(metric);
    idx_t i;
    double s = 0;
    (void)n;
    if (id < 0) return -1;
    for (i = 0; i < p; i++) s += metric_f32(id, X + m * pairs[2 * i], X + m * pairs[2 * i + 1], m);
    return s;
}

/* assign.hpp:6-47 / 50-91.  `min_dist` (nullable) additionally returns the
 * per-row minimum so the GPU path can be compared element-wise; the return
 * value is the reference's sequential inertia sum, -1 for an unknown metric. */
double oracle_assign_nearest_f64(const double *X, const double *Y, const char *metric,
                                 const idx_t *X_indices, idx_t n_X, idx_t n_Y, idx_t n_features,
                                 idx_t n_X_indices, idx_t *assignments, double *min_dist)
{
    int id = oracle_metric_id(metric);
    double d, min_d, inertia = 0;
    idx_t i, j, n = X_indices == NULL ? n_X : n_X_indices;
    if (id < 0) return -1;
    for (i = 0; i < n; i++) {
        const double *x = X + (X_indices == NULL ? i : X_indices[i]) * n_features;
        min_d = DBL_MAX;
        for (j = 0; j < n_Y; j++) {
            d = metric_f64(id, x, Y + j * n_features, n_features);
            if (d < min_d) {
                min_d = d;
                assignments[i] = j;
            }
        }
        if (min_dist) min_dist[i] = min_d;
        inertia += min_d;
    }
    return inertia;
}

double oracle_assign_nearest_f32(const float *X, const float *Y, const char *metric,
                                 const idx_t *X_indices, idx_t n_X, idx_t n_Y, idx_t n_features,
                                 idx_t n_X_indices, idx_t *assignments, double *min_dist)
{
    int id = oracle_metric_id(metric);
    double d, min_d, inertia = 0;
    idx_t i, j, n = X_indices == NULL ? n_X : n_X_indices;
    if (id < 0) return -1;
    for (i = 0; i < n; i++) {
        const float *x = X + (X_indices == NULL ? i : X_indices[i]) * n_features;
        min_d = DBL_MAX;
        for (j = 0; j < n_Y; j++) {
            d = metric_f32(id, x, Y + j * n_features, n_features);
            if (d < min_d) {
                min_d = d;
                assignments[i] = j;
            }
        }
        if (min_dist) min_dist[i] = min_d;
        inertia += min_d;
    }
    return inertia;
}

/* _KCenters.fit, cluster/kcenters.py:79-102, restated in C so the GPU
 * k-centers driver has a fast CPU checker at full sizes:
 *   distances_ = +inf, labels_ = 0; for it in 0..K-1:
 *     d = dist(X, X[c]); strict d < distances_ updates; ids[it] = c;
 *     c = argmax(distances_) (first maximum wins, numpy semantics: a NaN
 *     would win, but distances_ never holds NaN because NaN < x is false).
 */
int oracle_kcenters_fit_f32(const float *X, idx_t n, idx_t m, idx_t K, const char *metric,
                            idx_t seed_index, idx_t *ids, idx_t *labels, double *distances)
{
    int id = oracle_metric_id(metric);
    idx_t i, it, c = seed_index;
    if (id < 0) return -1;
    for (i = 0; i < n; i++) {
        labels[i] = 0;
        distances[i] = INFINITY;
    }
    for (it = 0; it < K; it++) {
        const float *y = X + c * m;
        idx_t best = 0;
        double bestv;
        for (i = 0; i < n; i++) {
            double d = metric_f32(id, X + i * m, y, m);
            if (d < distances[i]) {
                distances[i] = d;
                labels[i] = it;
            }
        }
        ids[it] = c;
        bestv = distances[0];
        for (i = 1; i < n; i++)
            if (distances[i] > bestv) {
                bestv = distances[i];
                best = i;
            }
        c = best;
    }
    return 0;
}

int oracle_kcenters_fit_f64(const double *X, idx_t n, idx_t m, idx_t K, const char *metric,
                            idx_t seed_index, idx_t *ids, idx_t *labels, double *distances)
{
    int id = oracle_metric_id(metric);
    idx_t i, it, c = seed_index;
    if (id < 0) return -1;
    for (i = 0; i < n; i++) {
        labels[i] = 0;
        distances[i] = INFINITY;
    }
    for (it = 0; it < K; it++) {
        const double *y = X + c * m;
        idx_t best = 0;
        double bestv;
        for (i = 0; i < n; i++) {
            double d = metric_f64(id, X + i * m, y, m);
            if (d < distances[i]) {
                distances[i] = d;
                labels[i] = it;
            }
        }
        ids[it] = c;
        bestv = distances[0];
        for (i = 1; i < n; i++)
            if (distances[i] > bestv) {
                bestv = distances[i];
                best = i;
            }
        c = best;
    }
    return 0;
}
