// libdistance_compat.hip -- the reference's native libdistance entry points
// (include/msmhip_libdistance.h), thin host-pointer wrappers over the msm_* ABI.
#include "../../include/msmhip_libdistance.h"
#include "common.h"

extern "C" {

double assign_nearest_double(const double* X, const double* Y, const char* metric,
                             const msm_npy_intp* X_indices, msm_npy_intp n_X, msm_npy_intp n_Y,
                             msm_npy_intp n_features, msm_npy_intp n_X_indices,
                             msm_npy_intp* assignments)
{
    double inertia = 0.0;
    int rc = msm_assign_nearest_f64(X, Y, metric, (const msm_idx_t*)X_indices, n_X, n_Y, n_features,
                                    n_X_indices, (msm_idx_t*)assignments, nullptr, &inertia, 0);
    return rc == MSM_OK ? inertia : -1.0;
}

double assign_nearest_float(const float* X, const float* Y, const char* metric,
                            const msm_npy_intp* X_indices, msm_npy_intp n_X, msm_npy_intp n_Y,
                            msm_npy_intp n_features, msm_npy_intp n_X_indices,
                            msm_npy_intp* assignments)
{
    double inertia = 0.0;
    int rc = msm_assign_nearest_f32(X, Y, metric, (const msm_idx_t*)X_indices, n_X, n_Y, n_features,
                                    n_X_indices, (msm_idx_t*)assignments, nullptr, &inertia, 0);
    return rc == MSM_OK ? inertia : -1.0;
}

void dist_double(const double* X, const double* y, const char* metric, msm_npy_intp n,
                 msm_npy_intp m, double* out)
{
    (void)msm_dist_f64(X, y, metric, n, m, nullptr, 0, out, 0);
}

void dist_float(const float* X, const float* y, const char* metric, msm_npy_intp n, msm_npy_intp m,
                double* out)
{
    (void)msm_dist_f32(X, y, metric, n, m, nullptr, 0, out, 0);
}

void dist_double_X_indices(const double* X, const double* y, const char* metric, msm_npy_intp n,
                           msm_npy_intp m, const msm_npy_intp* X_indices,
                           msm_npy_intp n_X_indices, double* out)
{
    (void)msm_dist_f64(X, y, metric, n, m, (const msm_idx_t*)X_indices, n_X_indices, out, 0);
}

void dist_float_X_indices(const float* X, const float* y, const char* metric, msm_npy_intp n,
                          msm_npy_intp m, const msm_npy_intp* X_indices,
                          msm_npy_intp n_X_indices, double* out)
{
    (void)msm_dist_f32(X, y, metric, n, m, (const msm_idx_t*)X_indices, n_X_indices, out, 0);
}

void cdist_double(const double* XA, const double* XB, const char* metric, msm_npy_intp na,
                  msm_npy_intp nb, msm_npy_intp m, double* out)
{
    (void)msm_cdist_f64(XA, XB, metric, na, nb, m, out, 0);
}

void cdist_float(const float* XA, const float* XB, const char* metric, msm_npy_intp na,
                 msm_npy_intp nb, msm_npy_intp m, double* out)
{
    (void)msm_cdist_f32(XA, XB, metric, na, nb, m, out, 0);
}

void pdist_double(const double* X, const char* metric, msm_npy_intp n, msm_npy_intp m, double* out)
{
    (void)msm_pdist_f64(X, metric, n, m, nullptr, 0, out, 0);
}

void pdist_float(const float* X, const char* metric, msm_npy_intp n, msm_npy_intp m, double* out)
{
    (void)msm_pdist_f32(X, metric, n, m, nullptr, 0, out, 0);
}

void pdist_double_X_indices(const double* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                            const msm_npy_intp* X_indices, msm_npy_intp n_X_indices, double* out)
{
    (void)msm_pdist_f64(X, metric, n, m, (const msm_idx_t*)X_indices, n_X_indices, out, 0);
}

void pdist_float_X_indices(const float* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                           const msm_npy_intp* X_indices, msm_npy_intp n_X_indices, double* out)
{
    (void)msm_pdist_f32(X, metric, n, m, (const msm_idx_t*)X_indices, n_X_indices, out, 0);
}

double sumdist_double(const double* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                      const msm_npy_intp* pairs, msm_npy_intp p)
{
    double s = 0.0;
    return msm_sumdist_f64(X, metric, n, m, (const msm_idx_t*)pairs, p, &s, 0) == MSM_OK ? s : -1.0;
}

double sumdist_float(const float* X, const char* metric, msm_npy_intp n, msm_npy_intp m,
                     const msm_npy_intp* pairs, msm_npy_intp p)
{
    double s = 0.0;
    return msm_sumdist_f32(X, metric, n, m, (const msm_idx_t*)pairs, p, &s, 0) == MSM_OK ? s : -1.0;
}

}  // extern "C"
