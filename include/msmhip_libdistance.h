/*
 * msmhip_libdistance.h -- the reference's own native seam, re-exported.
 *
 * These are, name for name and argument for argument, the functions that
 * /root/reference/msmbuilder/libdistance/libdistance.pyx:26-67 declares
 * `cdef extern ... nogil` from assign.hpp / dist.hpp / cdist.hpp / pdist.hpp / sumdist.hpp.  A maintainer
 * swaps the three `cdef extern from "....hpp"` blocks for
 * `cdef extern from "msmhip_libdistance.h"` and links libmsmhip.so (see
 * INTEGRATION.md); nothing else in libdistance.pyx changes.
 *
 * Semantics kept from the reference: host pointers, caller-owned buffers,
 * C-contiguous row-major arrays, npy_intp == int64 indices, results
 * bit-identical (labels and every distance; the returned inertia is an fp64
 * tree sum instead of a sequential one, |rel. diff| <= 1e-14).  Error
 * convention kept as well: an unknown metric makes assign_nearest_* return -1
 * and dist_* / cdist_* return without touching `out` (assign.hpp:15-18,
 * dist.hpp:11-14); the message goes to msm_last_error() instead of stderr.
 * Any other failure (no GPU, HIP error) also returns -1 / leaves `out`
 * untouched -- there is NO CPU fallback.
 */
#ifndef MSMHIP_LIBDISTANCE_H
#define MSMHIP_LIBDISTANCE_H

#include <stdint.h>

#ifndef NPY_INTP_DEFINED_BY_NUMPY
typedef intptr_t msm_npy_intp; /* == npy_intp */
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* assign.hpp:6-47 / 50-91 */
double assign_nearest_double(const double* X, const double* Y, const char* metric,
                             const msm_npy_intp* X_indices, msm_npy_intp n_X, msm_npy_intp n_Y,
                             msm_npy_intp n_features, msm_npy_intp n_X_indices,
                             msm_npy_intp* assignments);
double assign_nearest_float(const float* X, const float* Y, const char* metric,
                            const msm_npy_intp* X_indices, msm_npy_intp n_X, msm_npy_intp n_Y,
                            msm_npy_intp n_features, msm_npy_intp n_X_indices,
                            msm_npy_intp* assignments);
/* dist.hpp:4-22, 24-41, 44-60, 62-80 */
void dist_double(const double* X, const double* y, const char* metric, msm_npy_intp n,
                 msm_npy_intp m, double* out);
void dist_float(const float* X, const float* y, const char* metric, msm_npy_intp n,
                msm_npy_intp m, double* out);
void dist_double_X_indices(const double* X, const double* y, const char* metric, msm_npy_intp n,
                           msm_npy_intp m, const msm_npy_intp* X_indices,
                           msm_npy_intp n_X_indices, double* out);
void dist_float_X_indices(const float* X, const float* y, const char* metric, msm_npy_intp n,
                          msm_npy_intp m, const msm_npy_intp* X_indices,
                          msm_npy_intp n_X_indices, double* out);
/* cdist.hpp:4-26 / 28-49 */
void cdist_double(const double* XA, const double* XB, const char* metric, msm_npy_intp na,
                  msm_npy_intp nb, msm_npy_intp m, double* out);
void cdist_float(const float* XA, const float* XB, const char* metric, msm_npy_intp na,
                 msm_npy_intp nb, msm_npy_intp m, double* out);

/* pdist.hpp:4-88 */
void pdist_double(const double* X, const char* metric, msm_npy_intp n, msm_npy_intp m, double* out);
void pdist_float(const float* X, const char* metric, msm_npy_intp n, msm_npy_intp m, double* out);
void pdist_double_X_indices(const double* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                            const msm_npy_intp* X_indices, msm_npy_intp n_X_indices, double* out);
void pdist_float_X_indices(const float* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                           const msm_npy_intp* X_indices, msm_npy_intp n_X_indices, double* out);
/* sumdist.hpp:4-44 (returns -1 for an unknown metric or any failure) */
double sumdist_double(const double* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                      const msm_npy_intp* pairs, msm_npy_intp p);
double sumdist_float(const float* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                     const msm_npy_intp* pairs, msm_npy_intp p);

#ifdef __cplusplus
}
#endif
#endif /* MSMHIP_LIBDISTANCE_H */
