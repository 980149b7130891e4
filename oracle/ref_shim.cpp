// oracle/ref_shim.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Build glue for oracle/_ref/libref_libdistance.so: it #includes the
// reference's own libdistance headers WHERE THEY LIE under /root/reference
// (the Makefile passes -I$(REF)/msmbuilder/libdistance/src; no reference
// source is copied into this repository) and exposes them with C linkage so
// ctypes can call the real reference arithmetic.  npy_intp and NPY_INLINE come
// from numpy's OWN header (numpy/npy_common.h, found by the Makefile through
// numpy.get_include() and Python's include directory), exactly what the
// Cython-generated translation unit pulls in in the reference build
// (libdistance.pyx:13 `cimport numpy`); nothing is typedef'ed by hand.
//
// Used only by tests/ (to pin oracle/libdistance_oracle.c) and optionally by
// bench.py's cpu_baseline leg (kind="reference").
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include <Python.h>
#include <numpy/npy_common.h>

#include "assign.hpp"
#include "cdist.hpp"
#include "dist.hpp"
#include "pdist.hpp"
#include "sumdist.hpp"

extern "C" {

double ref_assign_nearest_double(const double* X, const double* Y, const char* metric,
                                 const npy_intp* X_indices, npy_intp n_X, npy_intp n_Y,
                                 npy_intp n_features, npy_intp n_X_indices, npy_intp* assignments)
{
    return assign_nearest_double(X, Y, metric, X_indices, n_X, n_Y, n_features, n_X_indices,
                                 assignments);
}

double ref_assign_nearest_float(const float* X, const float* Y, const char* metric,
                                const npy_intp* X_indices, npy_intp n_X, npy_intp n_Y,
                                npy_intp n_features, npy_intp n_X_indices, npy_intp* assignments)
{
    return assign_nearest_float(X, Y, metric, X_indices, n_X, n_Y, n_features, n_X_indices,
                                assignments);
}

void ref_dist_double(const double* X, const double* y, const char* metric, npy_intp n, npy_intp m,
                     double* out)
{
    dist_double(X, y, metric, n, m, out);
}

void ref_dist_float(const float* X, const float* y, const char* metric, npy_intp n, npy_intp m,
                    double* out)
{
    dist_float(X, y, metric, n, m, out);
}

void ref_dist_double_X_indices(const double* X, const double* y, const char* metric, npy_intp n,
                               npy_intp m, const npy_intp* X_indices, npy_intp n_X_indices,
                               double* out)
{
    dist_double_X_indices(X, y, metric, n, m, X_indices, n_X_indices, out);
}

void ref_dist_float_X_indices(const float* X, const float* y, const char* metric, npy_intp n,
                              npy_intp m, const npy_intp* X_indices, npy_intp n_X_indices,
                              double* out)
{
    dist_float_X_indices(X, y, metric, n, m, X_indices, n_X_indices, out);
}

void ref_cdist_double(const double* XA, const double* XB, const char* metric, npy_intp na,
                      npy_intp nb, npy_intp m, double* out)
{
    cdist_double(XA, XB, metric, na, nb, m, out);
}

void ref_cdist_float(const float* XA, const float* XB, const char* metric, npy_intp na,
                     npy_intp nb, npy_intp m, double* out)
{
    cdist_float(XA, XB, metric, na, nb, m, out);
}

void ref_pdist_double(const double* X, const char* metric, npy_intp n, npy_intp m, double* out)
{
    pdist_double(X, metric, n, m, out);
}

void ref_pdist_float(const float* X, const char* metric, npy_intp n, npy_intp m, double* out)
{
    pdist_float(X, metric, n, m, out);
}

void ref_pdist_double_X_indices(const double* X, const char* metric, npy_intp n, npy_intp m,
                                const npy_intp* X_indices, npy_intp n_X_indices, double* out)
{
    pdist_double_X_indices(X, metric, n, m, X_indices, n_X_indices, out);
}

void ref_pdist_float_X_indices(const float* X, const char* metric, npy_intp n, npy_intp m,
                               const npy_intp* X_indices, npy_intp n_X_indices, double* out)
{
    pdist_float_X_indices(X, metric, n, m, X_indices, n_X_indices, out);
}

double ref_sumdist_double(const double* X, const char* metric, npy_intp n, npy_intp m,
                          const npy_intp* pairs, npy_intp p)
{
    return sumdist_double(X, metric, n, m, pairs, p);
}

double ref_sumdist_float(const float* X, const char* metric, npy_intp n, npy_intp m,
                         const npy_intp* pairs, npy_intp p)
{
    return sumdist_float(X, metric, n, m, pairs, p);
}

}  // extern "C"
