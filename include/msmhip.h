/*
 * msmhip.h -- C ABI of libmsmhip.so, the MI355X (gfx950) implementation of the
 * MSMBuilder tICA + geometric-clustering hot path.
 *
 * Plain C, caller-owned buffers, no exceptions, no stdout/stderr chatter.
 * Every entry point returns an int status (MSM_OK == 0, negative = error; the
 * message is available from msm_last_error()) unless noted.
 *
 * Pointer placement.  Functions that take `on_device` accept either host
 * pointers (on_device = 0: the library stages through its own device buffers)
 * or device pointers in the current HIP device's address space (on_device = 1:
 * e.g. torch.Tensor.data_ptr()).  `on_device` applies to the PER-ROW arrays
 * (X, X_indices, assignments, per-row distances, cdist's `out`); per-centre
 * arrays (y, Y, ids) and scalar outputs are always host memory.
 *
 * Threading: one host thread drives one device; handles are not thread-safe.
 * All work is enqueued on the stream set with msm_set_stream() (default: the
 * null stream, which is torch's default current stream).
 *
 * What each group replaces in the reference (paths under
 * /root/reference/msmbuilder):
 *   msm_tica_*        decomposition/tica.py:150-165 (_initialize), :401-424 (_fit),
 *                     :228-259 (the sums those properties read), :312-354 (transform)
 *   msm_dist_*        libdistance/src/dist.hpp:4-80     (via libdistance.pyx:229-270)
 *   msm_cdist_*       libdistance/src/cdist.hpp:4-49    (via libdistance.pyx:134-179)
 *   msm_assign_nearest_*  libdistance/src/assign.hpp:6-91 (via libdistance.pyx:82-131)
 *   msm_pdist_* / msm_sumdist_*  libdistance/src/pdist.hpp:4-88, sumdist.hpp:4-44 (via libdistance.pyx:182-226, 273-310)
 *   msm_kcenters_fit_*    cluster/kcenters.py:79-102 (_KCenters.fit's k-pass loop)
 *   msm_kmeans_* / msm_mbk_*  sklearn MiniBatchKMeans arithmetic behind
 *                     cluster/__init__.py:67-69 (third-party)
 * The exact reference signatures of libdistance are additionally exported,
 * unprefixed, from include/msmhip_libdistance.h.
 */
#ifndef MSMHIP_H
#define MSMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t msm_idx_t; /* npy_intp on LP64 */

enum msm_status {
    MSM_OK = 0,
    MSM_ERR_INVALID = -1,   /* bad argument (shape, null pointer, dtype code) */
    MSM_ERR_METRIC = -2,    /* unknown metric name (reference: prints "Error", returns -1) */
    MSM_ERR_HIP = -3,       /* a HIP runtime call failed */
    MSM_ERR_NODEVICE = -4,  /* no usable GPU */
    MSM_ERR_NONFINITE = -5, /* input contains NaN/Inf (reference: ValueError, validation.py:68-74) */
    MSM_ERR_STATE = -6      /* handle used before data / after destroy */
};

/* accumulate precision of the tICA covariance kernel */
enum msm_tica_mode {
    MSM_TICA_F32 = 0,  /* v_mfma_f32_32x32x2_f32, fp32 partials per <=4096-row chunk, fp64 merge */
    MSM_TICA_F64 = 1,  /* v_mfma_f64_16x16x4_f64 on fp64-widened inputs: the reference's arithmetic */
    MSM_TICA_BF16 = 2, /* v_mfma_f32_32x32x16_bf16 on inputs rounded to bf16, fp32 partials, fp64 merge */
    MSM_TICA_BF16X2 = 3 /* split x = hi + mid (two bf16 terms), four products on the bf16 MFMA: fp32-class sums */
};

/* ---- runtime ---------------------------------------------------------- */
const char* msm_last_error(void);
const char* msm_version(void);
int msm_device_count(void);          /* number of visible GPUs, 0 if none (never negative) */
int msm_init(int device);            /* hipSetDevice + arch check (gfx950) */
int msm_set_stream(void* hip_stream); /* hipStream_t; NULL = null stream */
int msm_synchronize(void);
int msm_device_info(char* name, int name_len, int* n_cu, int64_t* hbm_bytes);

/* device memory helpers (so callers without torch can keep data resident) */
int msm_malloc(void** dptr, size_t bytes);
int msm_free(void* dptr);
/* Blocking copies; bytes == 0 is a no-op whatever the pointers (an empty array has no address).  Large copies go through the
 * library's pinned staging ring, which has ONE submitter: like every other entry point these two must not be called from a
 * second host thread while another call of this library is in progress. */
int msm_memcpy_h2d(void* dst, const void* src, size_t bytes);
int msm_memcpy_d2h(void* dst, const void* src, size_t bytes);
int msm_memcpy_d2d(void* dst, const void* src, size_t bytes);
/* n host buffers back to back into one device buffer (dst must hold their sum), staged through the library's pinned ring
 * with no synchronisation between them; returns when the data is on the device.  What tICA.fit_transform uses to upload a
 * list of host trajectories ONCE for both passes. */
int msm_upload_list(void* dst, const void* const* src, const msm_idx_t* nbytes, msm_idx_t n);
/* out[i, :] = X[rows[i], :]  (X device or host per on_device; rows/out follow it) */
int msm_gather_rows(const void* X, int elem_size, msm_idx_t n_features, const msm_idx_t* rows,
                    msm_idx_t n_rows, void* out, int on_device);
/* HIP-event timing on the library's stream (bench.py's roofline leg) */
int msm_event_create(void** ev);
int msm_event_record(void* ev);
int msm_event_elapsed_ms(void* start, void* stop, float* ms); /* synchronises on `stop` */
int msm_event_destroy(void* ev);

/* ---- communicator: RCCL over xGMI, one process per GPU ---------------
 * The reference is a single process (no collective exists in /root/reference); these are the exchange steps the
 * sharded hot path adds: msm_tica_allreduce, msm_kcenters_fit_sharded_*, msm_mbk_allreduce / msm_mbk_run.
 * Bootstrap: rank 0 calls msm_comm_unique_id and ships the 128 bytes to the other ranks by any means (parallel.py:
 * a torch.distributed broadcast); every rank then calls msm_comm_init_rccl AFTER msm_init(device).  The collectives
 * run on the library stream on device buffers.
 * msm_comm_init_host installs a host-side transport instead (buffers are staged through pinned memory and
 * fn(op, send, recv, nbytes) is called: op 0 = all-reduce(sum) of nbytes/8 doubles in place (send == recv), op 1 =
 * all-gather of nbytes from every rank into recv[world][nbytes]; non-zero return = failure).  It exists for runs
 * where RCCL cannot form a communicator -- several ranks on ONE GPU, gloo-only test runs. */
typedef int (*msm_host_collective_fn)(int op, void* send, void* recv, int64_t nbytes);
int msm_comm_rccl_available(void); /* 1: librccl loadable and a device visible.  Ranks must agree on this (an all-reduce
                                      MIN over the bootstrap channel) before ANY of them calls msm_comm_init_rccl, because
                                      ncclCommInitRank blocks until every rank has joined */
int msm_comm_unique_id(char* id128);
int msm_comm_init_rccl(const char* id128, int rank, int world); /* gives up after MSM_COMM_TIMEOUT_S seconds (180) with
                                      MSM_ERR_STATE and a message naming the rank: ncclCommInitRank cannot be cancelled, a rank
                                      that never joins must not hang the others for good */
int msm_comm_init_host(msm_host_collective_fn fn, int rank, int world);
int msm_comm_selftest(int timeout_s); /* one all-reduce + one all-gather of a few doubles through the installed communicator,
                                      checked and time-limited (0 ok; MSM_ERR_STATE: wrong numbers / no answer, message names the
                                      rank).  Every rank calls it right after the communicator is built */
int msm_comm_destroy(void);
int msm_comm_info(int* rank, int* world, int* kind); /* kind: 0 none, 1 RCCL, 2 host callback */
int msm_comm_allreduce_f64(double* dbuf, msm_idx_t n);                    /* device buffer, in place; synchronises */
int msm_comm_allgather(const void* dsend, void* drecv, msm_idx_t bytes);  /* device buffers; synchronises */
int msm_comm_measure(int kind, msm_idx_t bytes, int reps, float* us_per_call); /* latency of `reps` queued collectives of `bytes`
                                      per rank (kind 0: all-reduce of doubles, 1: all-gather), microseconds per call; every
                                      rank calls it (bench.py reports these instead of assumptions when it runs on N > 1) */

/* ---- tICA second-moment accumulation ---------------------------------- */
typedef struct msm_tica msm_tica_t;

int msm_tica_create(msm_tica_t** h, msm_idx_t n_features, msm_idx_t lag_time, int mode);
int msm_tica_destroy(msm_tica_t* h);
int msm_tica_reset(msm_tica_t* h); /* zero all accumulators and counters */

/* One trajectory X[n_rows, n_features] (row stride ld elements), dtype_bytes = 4 (f32) or 8 (f64).
 * Lagged pairs never span calls.  n_rows <= lag_time is a no-op with *skipped = 1
 * (tica.py:410-412).  check_finite != 0: the call synchronises and returns
 * MSM_ERR_NONFINITE, state unchanged, if X holds NaN/Inf; check_finite == 0: fully
 * asynchronous, a sticky flag is kept (msm_tica_nonfinite). */
/* dtype_bytes = 2: bfloat16-STORED trajectories (BASELINE configs[4]: half the bytes per frame), accepted by the
 * MSM_TICA_BF16 / MSM_TICA_BF16X2 modes (MSM_ERR_INVALID otherwise), device-resident or host. */
int msm_tica_accumulate(msm_tica_t* h, const void* X, int dtype_bytes, msm_idx_t n_rows,
                        msm_idx_t ld, int on_device, int check_finite, int* skipped);
/* Many trajectories in one launch: X_ptrs[s] -> n_rows[s] x n_features, common ld.
 * X_ptrs / n_rows are host arrays of length n_seq; the trajectories themselves are
 * device-resident (on_device = 1) or host (0).  *n_skipped counts too-short ones. */
int msm_tica_accumulate_batch(msm_tica_t* h, const void* const* X_ptrs, const msm_idx_t* n_rows,
                              msm_idx_t n_seq, int dtype_bytes, msm_idx_t ld, int on_device,
                              int check_finite, msm_idx_t* n_skipped);
/* Slices of trajectories, for splitting ONE long trajectory over ranks (the sum over
 * lagged pairs of tica.py:417-422 restricted to the pairs whose LEFT frame a rank owns).
 * seg4[4*s + {0,1,2,3}] = {trajectory length, trajectory row of X_ptrs[s][0], own_begin, own_end}:
 * the call adds the terms of the left frames t in [own_begin, own_end) only and needs the slice to
 * reach row min(own_end + lag_time, length) - 1 (right halo; MSM_ERR_INVALID otherwise).
 * n_observations grows by own_end - own_begin, n_sequences by 1 where own_begin == 0, so the
 * all-reduced counters equal the unsplit fit's.  {n, 0, 0, n} is msm_tica_accumulate_batch. */
int msm_tica_accumulate_segments(msm_tica_t* h, const void* const* X_ptrs, const msm_idx_t* n_rows,
                                 const msm_idx_t* seg4, msm_idx_t n_seq, int dtype_bytes, msm_idx_t ld,
                                 int on_device, int check_finite, msm_idx_t* n_skipped);
int msm_tica_nonfinite(msm_tica_t* h, int* flag); /* synchronises; sticky until reset */
/* 1 when this handle accumulates fp32 input with the symmetric sum/difference kernel (MSM_TICA_F32,
 * 128 < n_features <= 3968, n_features % 4 == 0; MSM_TICA_SYM=0 disables): the "C" it exports is then
 * already (C + C^T) / 2 -- the only form the estimator reads (tica.py:234-241) -- and the raw
 * X[:-tau].T @ X[tau:] is not kept. */
int msm_tica_lagged_symmetrised(msm_tica_t* h, int* flag);
/* HIP-event duration (ms) of the most recent MFMA accumulation launch of this handle,
 * measured on the stream it ran on (bench.py's roofline leg); synchronises on it.  bf16 modes: the whole pipeline --
 * float32 rows: pack, then multiply, super-chunk after super-chunk in turn on one stream (sequential: overlapping them
 * was measured slower); with MSM_TICA_IMG_FUSED=1 on bfloat16-stored rows: the step-record kernel + the fused kernel. */
int msm_tica_last_kernel_ms(msm_tica_t* h, float* ms);
/* 1 when the most recent accumulation launch had no column-sum pass over X of its own: the sum/difference kernel's
 * staging lanes summed the left frames (float32 input, whole trajectories of >= 2 lag frames, n_features % 128 == 0, at
 * least MSM_TICA_FOLD_MIN = 2^26 elements; MSM_TICA_FOLD=0 disables).  The sums (tica.py:418-419) and the finite check
 * (utils/validation.py:68-74; a rejected launch is undone) are the same either way. */
int msm_tica_last_folded(msm_tica_t* h, int* flag);
/* 1 when the most recent accumulation launch of a bf16 / bf16x2 handle ran the FUSED kernel (round 5; opt-in,
 * MSM_TICA_IMG_FUSED=1): bfloat16-stored rows, n_features % 256 == 0, ld % 8 == 0, 16-byte aligned trajectories -- the MFMA
 * kernel's load role stages the raw rows in LDS and forms the pair frames itself, no packed image is written
 * (tica.py:402-422 in one streamed pass + a column-sum pass).  Bit-identical accumulators to the packed-image pipeline on such
 * input, half its fabric traffic, no image ring -- and measured slower, hence not the default. */
int msm_tica_last_img_fused(msm_tica_t* h, int* flag);
/* 1 when the most recent accumulation launch of a bf16 / bf16x2 handle packed its later super-chunks of the image INSIDE the
 * multiply kernel (round 6, the carried pack: whole 256-feature panels, 16-byte aligned rows, more than one super-chunk;
 * MSM_TICA_IMG_CARRY=0 disables): the packets, and hence the accumulators, are the pre-pass kernel's bit for bit. */
int msm_tica_last_img_carried(msm_tica_t* h, int* flag);
/* profiling: {shader-clock start, end, 100 MHz wall-clock start, end} of workgroup 0 of that launch */
int msm_tica_debug_clocks(msm_tica_t* h, long long* out4);
/* profiling builds (csrc built with -DMSM_TICA_PROFILE) only: out64[8 + 8*slot + i] = shader cycles wave 0 of
 * five sample workgroups spent in section i of the fp32 kernel {chunk prologue, step head, MFMA loop,
 * step tail, final merge, inter-chunk merge}; zeros in a product build */
int msm_tica_debug_profile(msm_tica_t* h, long long* out64);

/* Accumulators as the reference defines them (float64, row-major F x F / F):
 *   C    = sum_traj X[:-tau].T @ X[tau:]                    (tica.py:417; its symmetric part only when
 *                                                            msm_tica_lagged_symmetrised says so)
 *   G    = sum_traj X[:-tau].T @ X[:-tau] + X[tau:].T @ X[tau:]   (tica.py:421-422, only their sum is ever read: :245)
 *   s0   = sum_traj X[:-tau].sum(0)   stau = sum_traj X[tau:].sum(0)   (tica.py:418-419)
 *   n_observations, n_sequences                               (tica.py:414-415)
 * Host pointers. */
int msm_tica_export(msm_tica_t* h, double* C, double* G, double* s0, double* stau,
                    msm_idx_t* n_observations, msm_idx_t* n_sequences);
int msm_tica_import(msm_tica_t* h, const double* C, const double* G, const double* s0,
                    const double* stau, msm_idx_t n_observations, msm_idx_t n_sequences);
/* Packed form for one RCCL all-reduce(sum) over xGMI (torch.distributed):
 * doubles [C (F*F) | G (F*F) | s0 (F) | stau (F) | n_observations | n_sequences]. */
msm_idx_t msm_tica_packed_size(msm_tica_t* h); /* number of doubles */
int msm_tica_export_packed(msm_tica_t* h, double* buf, int on_device);
int msm_tica_import_packed(msm_tica_t* h, const double* buf, int on_device);

/* s0 / stau alone (host, F doubles each): the means without the two F x F downloads. */
int msm_tica_export_sums(msm_tica_t* h, double* s0, double* stau);

/* Device-resident finalisation + solve of the generalized eigenproblem of tica.py:167-259 (replaces the host-side
 * `_solve` body: offset_correlation_, covariance_ incl. the Rao-Blackwell Ledoit-Wolf shrinkage of tica.py:492-524,
 * and scipy.linalg.eigh(lhs, b=rhs, eigvals=(F-k, F-1)) of tica.py:188-194).  Nothing F x F crosses PCIe except, in
 * the hybrid form, the one reduced matrix.
 *   shrinkage < 0 (or NaN): the RBLW estimate with n = n_rblw (= n_observations_); otherwise the given intensity.
 *   scale: NULL, or F host doubles s of a folded input scaling x' = (x - m) / s (moments scale by 1 / (s_i s_j)).
 *   mu (host, F): the RAW means (s0 + stau) / 2N'.   info (host, 8 doubles, nullable): {rho, tr S, potrf info,
 *   syevd info, OC non-finite, S non-finite}.
 * Errors: MSM_ERR_NONFINITE with the reference's RuntimeError texts ("... is not symmetric") when a moment is not
 * finite; MSM_ERR_INVALID with LAPACK's "leading minor ... not positive definite" text; MSM_ERR_STATE before any data.
 *
 * msm_tica_reduce: B = L L^T, Cs = L^-1 OC L^-T on the device; Cs (host, F x F, symmetric up to rounding) is returned
 * for a top-k standard eigensolver on the host (LAPACK dsyevr: its tridiagonalisation is latency-bound on a GPU at
 * F = 512); msm_tica_backsolve then maps k eigenvectors y of Cs (rows of Y, host k x F) to v = L^-T y (rows of V),
 * B-orthonormal like LAPACK's dsygvx.  msm_tica_solve_device does everything on the device (rocSOLVER dsyevd; wins
 * from F ~ 1024) and returns the k largest eigenvalues (descending) and their eigenvectors as rows. */
int msm_tica_reduce(msm_tica_t* h, double shrinkage, msm_idx_t n_rblw, const double* scale, double* Cs, double* mu,
                    double* info);
int msm_tica_backsolve(msm_tica_t* h, const double* Y, msm_idx_t k, double* V);
/* The top-k solve on the device (128 <= n_features <= 1024, k <= 16): msm_tica_reduce's finalisation and reduction
 * (Cholesky by the library's own blocked kernel), a Chebyshev-filtered subspace iteration on the reduced matrix
 * (csrc/subspace.hip; its 32 x 32 Rayleigh-Ritz problems are solved on the host, so it synchronises a few times),
 * v = L^-T y.  vals[k] descending, vecs[k][F] rows B-orthonormal like dsygvx's, mu[F], info[12]: [0..5] as
 * msm_tica_reduce, info[6] = max_j ||Cs y_j - lambda_j y_j||_inf, info[7] = max(max_j | ||y_j||^2 - 1 |, max_{i<j} |y_i.y_j|),
 * info[8] = 1 if pairs were returned, info[9] = filtered iterations spent.
 * *status = 0: pairs returned and verified on the reduced matrix (residual, norm, mutual orthogonality); 1: the iteration
 * did not converge (flat spectra); 2: the check failed -- in both cases Cs (host, F x F) holds the reduced matrix for the
 * caller's LAPACK route (msm_tica_backsolve afterwards). */
int msm_tica_solve_topk(msm_tica_t* h, double shrinkage, msm_idx_t n_rblw, const double* scale, msm_idx_t k, double* vals,
                        double* vecs, double* Cs, double* mu, double* info, int* status);
/* Building block, exported for the tests: Cholesky B = U^T U on the row-major upper triangle (LAPACK dpotrf 'L' on the
 * column-major view), *info = first non-positive pivot (1-based) or 0. */
int msm_potrf(double* B, msm_idx_t n, int* info, int on_device);
int msm_tica_solve_device(msm_tica_t* h, double shrinkage, msm_idx_t n_rblw, const double* scale, msm_idx_t k,
                          double* vals, double* vecs, double* mu, double* info);

/* Sum the accumulators over all ranks of the library communicator: ONE all-reduce of the packed float64
 * [C | G | s0 | stau | n_obs | n_seq] (4.2 MB at F = 512), device to device; every rank ends with the global model.
 * Without a communicator (or with one rank) it is a no-op. */
int msm_tica_allreduce(msm_tica_t* h);
int msm_tica_counts(msm_tica_t* h, msm_idx_t* n_observations, msm_idx_t* n_sequences); /* frames / trajectories seen */

/* out[n, k] (float64) = (X - mean) @ comps.T, comps is k x F row-major, mean/comps host
 * float64 (tica.py:329-333; any kinetic/commute column scaling is folded into comps by
 * the caller).  X / out follow on_device.  check_finite as above.  dtype_bytes = 2: bfloat16-STORED rows, widened
 * (exactly) inside the kernel -- the result equals the projection of their float32 images. */
int msm_tica_project(const void* X, int dtype_bytes, msm_idx_t n_rows, msm_idx_t n_features,
                     msm_idx_t ld, const double* mean, const double* comps, msm_idx_t k,
                     double* out, int on_device, int check_finite);
/* The same for a LIST of device-resident trajectories in one launch per block of 16 components (tica.py:329-352 walks the
 * list): X_ptrs[s] is [n_rows[s], n_features] with row stride n_features, out_ptrs[s] its [n_rows[s], k] float64 output.
 * Rows must be whole 16-byte vectors at 16-byte aligned addresses; MSM_ERR_INVALID otherwise (project one by one then). */
int msm_tica_project_batch(const void* const* X_ptrs, double* const* out_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq,
                           int dtype_bytes, msm_idx_t n_features, const double* mean, const double* comps, msm_idx_t k,
                           int check_finite);
/* The same for a LIST of HOST trajectories of float32 / float64 rows (dtype_bytes 4 / 8, row stride n_features): `out`
 * (host) is ONE [sum of n_rows][k] float64 array, trajectory after trajectory.  Staged over PCIe in groups through two
 * device buffers, group g + 1 copied while group g is projected, one copy back at the end (tica.py:329-352 walks the
 * list of numpy arrays one by one). */
int msm_tica_project_host_list(const void* const* X_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int dtype_bytes,
                               msm_idx_t n_features, const double* mean, const double* comps, msm_idx_t k, double* out,
                               int check_finite);

/* ---- libdistance: exact-arithmetic vector metrics --------------------- */
/* metric in {"euclidean","sqeuclidean","cityblock","chebyshev","canberra",
 *            "braycurtis","hamming","jaccard"}; results are bit-identical to the
 * reference's scalar loops (float subtract -> double accumulate, feature order,
 * sqrt before compare, strict <, lowest index wins).  out/min_dist are float64. */
int msm_dist_f32(const float* X, const float* y, const char* metric, msm_idx_t n, msm_idx_t m,
                 const msm_idx_t* X_indices, msm_idx_t n_X_indices, double* out, int on_device);
int msm_dist_f64(const double* X, const double* y, const char* metric, msm_idx_t n, msm_idx_t m,
                 const msm_idx_t* X_indices, msm_idx_t n_X_indices, double* out, int on_device);
int msm_cdist_f32(const float* XA, const float* XB, const char* metric, msm_idx_t na,
                  msm_idx_t nb, msm_idx_t m, double* out, int on_device);
int msm_cdist_f64(const double* XA, const double* XB, const char* metric, msm_idx_t na,
                  msm_idx_t nb, msm_idx_t m, double* out, int on_device);
/* pdist.hpp:4-88: condensed upper triangle of the (indexed) rows, out has n(n-1)/2 entries;
 * sumdist.hpp:4-44: sum over the listed pairs (pairs is [p][2], follows on_device; the reference
 * sums sequentially, here an fp64 tree sum). */
int msm_pdist_f32(const float* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
                  msm_idx_t n_X_indices, double* out, int on_device);
int msm_pdist_f64(const double* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* X_indices,
                  msm_idx_t n_X_indices, double* out, int on_device);
int msm_sumdist_f32(const float* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                    msm_idx_t p, double* sum, int on_device);
int msm_sumdist_f64(const double* X, const char* metric, msm_idx_t n, msm_idx_t m, const msm_idx_t* pairs,
                    msm_idx_t p, double* sum, int on_device);
/* assignments[i] = argmin_j metric(X[i or X_indices[i]], Y[j]); min_dist nullable;
 * *inertia = sum_i min_dist[i] (fp64 tree sum; the reference sums sequentially). */
int msm_assign_nearest_f32(const float* X, const float* Y, const char* metric,
                           const msm_idx_t* X_indices, msm_idx_t n_X, msm_idx_t n_Y,
                           msm_idx_t n_features, msm_idx_t n_X_indices, msm_idx_t* assignments,
                           double* min_dist, double* inertia, int on_device);
int msm_assign_nearest_f64(const double* X, const double* Y, const char* metric,
                           const msm_idx_t* X_indices, msm_idx_t n_X, msm_idx_t n_Y,
                           msm_idx_t n_features, msm_idx_t n_X_indices, msm_idx_t* assignments,
                           double* min_dist, double* inertia, int on_device);

/* Gonzalez k-centers (kcenters.py:79-102): K fused dist + running-min + argmax passes,
 * no host round trip inside the loop.  ids (host, K) = chosen row indices, first =
 * seed_index; labels (int64) / distances (float64) per row follow on_device;
 * *inertia = sum(distances). */
int msm_kcenters_fit_f32(const float* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters,
                         const char* metric, msm_idx_t seed_index, msm_idx_t* ids,
                         msm_idx_t* labels, double* distances, double* inertia, int on_device);
int msm_kcenters_fit_f64(const double* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters,
                         const char* metric, msm_idx_t seed_index, msm_idx_t* ids,
                         msm_idx_t* labels, double* distances, double* inertia, int on_device);
/* the same fit, also returning the chosen rows themselves: centers[n_clusters][m] (HOST memory) = X[ids]
 * (kcenters.py:98 cluster_centers_), in the fit's final synchronisation */
int msm_kcenters_fit2_f32(const float* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                          msm_idx_t seed_index, msm_idx_t* ids, msm_idx_t* labels, double* distances, double* inertia,
                          int on_device, float* centers);
int msm_kcenters_fit2_f64(const double* X, msm_idx_t n, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                          msm_idx_t seed_index, msm_idx_t* ids, msm_idx_t* labels, double* distances, double* inertia,
                          int on_device, double* centers);

/* One externally driven k-centers pass for row-sharded data (one process per GPU): the centre's
 * coordinates y (host, m values; it may live on another rank) are supplied, distances_/labels_
 * (device) get the strict running-min update with label `it`, and the shard-local
 * (max distance, lowest local row attaining it, that row's m coordinates -- host, nullable)
 * come back for the cross-rank argmax: one small D2H per pass. */
int msm_kcenters_pass_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* y, msm_idx_t it,
                          const char* metric, msm_idx_t* labels, double* distances, double* max_dist,
                          msm_idx_t* argmax, float* argmax_row, int on_device);
int msm_kcenters_pass_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* y, msm_idx_t it,
                          const char* metric, msm_idx_t* labels, double* distances, double* max_dist,
                          msm_idx_t* argmax, double* argmax_row, int on_device);

/* The same pass with NOTHING on the host (multi-GPU driver, one collective per centre and no
 * synchronisation): y_dev is the centre on the device; after the pass the shard's candidate record
 * cand_dev[2 + m] (device, float64) = {max distance, row_offset + lowest local row attaining it,
 * that row's coordinates}, or {-1, -1, 0...} for an empty shard.  Ranks all-gather their records
 * (RCCL) into cands_dev[world][2 + m]; msm_kcenters_select picks the winner (largest distance, ties
 * to the lowest global row = numpy's argmax on the concatenated array) and writes its coordinates
 * to y_dev and to row `slot` of centers_dev, its global row to ids_dev[slot].  All asynchronous on
 * the library stream. */
int msm_kcenters_pass_dev_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* y_dev, msm_idx_t it,
                              const char* metric, msm_idx_t* labels, double* distances, msm_idx_t row_offset,
                              double* cand_dev);
int msm_kcenters_pass_dev_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* y_dev, msm_idx_t it,
                              const char* metric, msm_idx_t* labels, double* distances, msm_idx_t row_offset,
                              double* cand_dev);
int msm_kcenters_select_f32(const double* cands_dev, msm_idx_t world, msm_idx_t m, float* y_dev, float* centers_dev,
                            msm_idx_t* ids_dev, msm_idx_t slot);
int msm_kcenters_select_f64(const double* cands_dev, msm_idx_t world, msm_idx_t m, double* y_dev, double* centers_dev,
                            msm_idx_t* ids_dev, msm_idx_t slot);

/* The whole row-sharded k-centers fit in one call (every rank of the library communicator calls it with ITS block of
 * rows; ranks own consecutive blocks of the global array, this one starting at global row row_offset): per centre one
 * pass kernel, one candidate record, ONE all-gather of the world's records (RCCL on the library stream) and one select
 * kernel -- nothing returns to the host inside the loop.  seed_index is the GLOBAL row of centre 0.  labels / distances
 * (device, n_local) stay sharded; ids (host, K global rows), centers (host, K x m) and *inertia (sum over ALL rows) are
 * the same on every rank and equal the single-process fit of the concatenated array bit for bit (ties go to the lowest
 * global row, numpy's argmax: kcenters.py:91-97). */
int msm_kcenters_fit_sharded_f32(const float* X, msm_idx_t n_local, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                                 msm_idx_t seed_index, msm_idx_t row_offset, msm_idx_t* labels, double* distances,
                                 msm_idx_t* ids, float* centers, double* inertia);
int msm_kcenters_fit_sharded_f64(const double* X, msm_idx_t n_local, msm_idx_t m, msm_idx_t n_clusters, const char* metric,
                                 msm_idx_t seed_index, msm_idx_t row_offset, msm_idx_t* labels, double* distances,
                                 msm_idx_t* ids, double* centers, double* inertia);

/* What the last k-centers fit of this process streamed: out5 = {rows, plain passes, screened passes, bytes a plain pass
 * reads per row (the row + distances_ + labels_), bytes a screened pass reads per row (the low-precision copy + the
 * rounded-up distance)}.  bench.py derives its bytes-per-pass figure from it (exact re-evaluations of screen candidates
 * and the updates are not counted: a lower bound on the traffic). */
int msm_kcenters_last_stats(msm_idx_t* out5);
/* Wide screened passes (round 6: float32 rows, float64 rows of more than 16 features; csrc/distance_wscreen_dev.h), with
 * MSM_KC_STATS=1 in the environment of the fit: out2 = {rows re-evaluated exactly over all screened passes, rows that changed}. */
int msm_kcenters_last_wide_stats(msm_idx_t* out2);

/* ---- k-means labelling / mini-batch step (GEMM form on MFMA) ----
 * Element type: scikit-learn (the arithmetic behind msmbuilder.cluster.MiniBatchKMeans, cluster/__init__.py:67-69) works
 * in the type of X -- float32 rows give float32 centres / counts, everything else is float64 -- and the reference
 * pipeline feeds it the float64 output of tICA.transform (decomposition/tica.py:329-352).  Both types are built:
 * `_f32` entry points on v_mfma_f32_32x32x2_f32, `_f64` entry points on v_mfma_f64_16x16x4_f64 (round 6). */
/* labels[i] = argmin_j ||X[i]-C[j]||^2 computed as ||c||^2 - 2 x.c (+||x||^2 for the
 * inertia), in the rows' own type like scikit-learn's _labels_inertia; centers host [K, m].
 * labels int32 (sklearn's dtype) follow on_device; *inertia fp64 sum of per-row terms. */
int msm_kmeans_label_f32(const float* X, msm_idx_t n, msm_idx_t m, const float* centers,
                         msm_idx_t K, int32_t* labels, double* inertia, int on_device);
int msm_kmeans_label_f64(const double* X, msm_idx_t n, msm_idx_t m, const double* centers,
                         msm_idx_t K, int32_t* labels, double* inertia, int on_device);
/* k-means++ seeds (scikit-learn `_kmeans_plusplus`, sklearn/cluster/_kmeans.py:163-259; reached from
 * msmbuilder/cluster/__init__.py:67-69) of the n x F rows X (host or device per on_device): centre 0 = row
 * `first`, then K - 1 rounds of L candidates drawn by inverse-CDF sampling of the current squared distances with the
 * uniforms u[(K - 1) * L] (host float64 in [0, 1): the caller's RandomState stream), the candidate of lowest potential
 * wins.  float32 rows: distances in scikit-learn's float64-upcast arithmetic rounded to float32; float64 rows: float64
 * throughout.  centers[K * F] (the rows' type) and ids[K] (rows of X): host. */
int msm_kmeans_plusplus_f32(const float* X, msm_idx_t n, msm_idx_t F, msm_idx_t K, msm_idx_t first, const double* u, int L,
                            float* centers, msm_idx_t* ids, int on_device);
int msm_kmeans_plusplus_f64(const double* X, msm_idx_t n, msm_idx_t F, msm_idx_t K, msm_idx_t first, const double* u, int L,
                            double* centers, msm_idx_t* ids, int on_device);
/* One MiniBatchKMeans step on the rows X[batch_idx[b]] (batch_idx host int64, length B):
 * label, then per-centre streaming mean c <- (c*w + sum x)/(w + n) with cumulative
 * counts (sklearn _k_means_minibatch.pyx:59-109, unit sample weights).  centers
 * [K, m] and counts [K] are host, updated in place; sums/counts partials are
 * exported for the multi-GPU all-reduce when apply_update == 0:
 *   batch_sums [K, m] float64, batch_counts [K] float64 (host, nullable). */
int msm_mbk_step_f32(const float* X, msm_idx_t n, msm_idx_t m, const msm_idx_t* batch_idx,
                     msm_idx_t B, float* centers, float* counts, msm_idx_t K,
                     double* batch_inertia, double* batch_sums, double* batch_counts,
                     int apply_update, int on_device);
int msm_mbk_step_f64(const double* X, msm_idx_t n, msm_idx_t m, const msm_idx_t* batch_idx,
                     msm_idx_t B, double* centers, double* counts, msm_idx_t K,
                     double* batch_inertia, double* batch_sums, double* batch_counts,
                     int apply_update, int on_device);

/* Device-resident MiniBatchKMeans state (centres [K, m], cumulative counts [K] and ||c||^2 stay in
 * HBM for the whole fit).  A step ships the batch indices in and [inertia | counts] out.
 * The handle has an element type: msm_mbk_create -> float32, msm_mbk_create_f64 -> float64 (msm_mbk_is_f64 tells); every
 * `void*` below (X, centers, counts, counts_out) is an array of THAT type.
 *   msm_mbk_step(apply_update = 1): label the batch rows X[batch_idx[b]], streaming-mean update,
 *       *batch_inertia (before the update) and counts_out (host, K; nullable) after it.
 *   apply_update = 0 (multi-GPU): label + fp64 batch sums only; msm_mbk_export_packed() hands out
 *       [K*m sums | K counts | inertia] doubles for the RCCL all-reduce, msm_mbk_apply_packed()
 *       applies the reduced buffer identically on every rank.
 *   msm_mbk_reassign: centres[which[i]] = X[rows[i]], counts[which[i]] = new_count
 *       (scikit-learn's starved-centre reassignment, decided on the host with its RNG). */
typedef struct msm_mbk msm_mbk_t;
int msm_mbk_create(msm_mbk_t** h, msm_idx_t n_clusters, msm_idx_t n_features);
int msm_mbk_create_f64(msm_mbk_t** h, msm_idx_t n_clusters, msm_idx_t n_features);
int msm_mbk_is_f64(msm_mbk_t* h);
int msm_mbk_destroy(msm_mbk_t* h);
int msm_mbk_set(msm_mbk_t* h, const void* centers, const void* counts);
int msm_mbk_set_counts(msm_mbk_t* h, const void* counts);
int msm_mbk_get(msm_mbk_t* h, void* centers, void* counts);
int msm_mbk_step(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t B,
                 double* batch_inertia, void* counts_out, int apply_update, int on_device);
/* S consecutive plain steps (label + streaming-mean update, as msm_mbk_step with apply_update = 1) on device-resident X
 * with ONE host synchronisation: batch_idx is [S][B] (host), and sklearn's _mini_batch_convergence (_kmeans.py:1963-2027,
 * the tol == 0 / verbose == 0 branch) runs on the device after every step -- state6 = {ewa_inertia, ewa_inertia_min,
 * no_improvement, have_ewa, have_min, (out) steps executed}, alpha = min(1, 2 B / (n_samples + 1)), max_no_improvement < 0
 * = None.  Steps queued behind the one at which the criterion fires are not executed: *steps_done <= S says how many were,
 * inertias[0 .. steps_done) are their batch inertias, *converged the criterion.  first_step = index of the run's first
 * step in the fit (step 0 is excluded from the moving average, as in sklearn). */
int msm_mbk_run(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t S, msm_idx_t B,
                msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement, double* state6,
                msm_idx_t* steps_done, int* converged, double* inertias, void* counts_out);
/* msm_mbk_run in two halves: _begin queues the run and returns, _end waits and fetches (the host draws the next run's
 * indices in between); one run in flight per handle */
int msm_mbk_run_begin(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* batch_idx, msm_idx_t S, msm_idx_t B,
                      msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement, const double* state6);
int msm_mbk_run_end(msm_mbk_t* h, double* state6, msm_idx_t* steps_done, int* converged, double* inertias, void* counts_out);
/* The same for a ROW-SHARDED fit (one process per GPU, rows in consecutive blocks): the S batches are global and identical
 * on every rank; a rank passes the rows of each batch that IT owns as local row numbers -- local_idx (host) back to back,
 * offsets[S + 1] (host) delimiting the steps -- and the size B of the whole batch.  Per step: label + fp64 sums / counts /
 * inertia of the local rows, ONE all-reduce of the packed [K m | K | 1] buffer over the library communicator (RCCL on the
 * library stream, device to device), the identical update and convergence step on every rank (every rank stops at the same
 * step: the criterion sees the all-reduced inertia).  One host synchronisation per run; outputs as msm_mbk_run. */
int msm_mbk_run_sharded(msm_mbk_t* h, const void* X, msm_idx_t n_local, const msm_idx_t* local_idx, const msm_idx_t* offsets,
                        msm_idx_t S, msm_idx_t B, msm_idx_t first_step, double alpha, msm_idx_t max_no_improvement,
                        double* state6, msm_idx_t* steps_done, int* converged, double* inertias, void* counts_out);
msm_idx_t msm_mbk_packed_size(msm_mbk_t* h);
int msm_mbk_export_packed(msm_mbk_t* h, double* buf, int on_device);
int msm_mbk_apply_packed(msm_mbk_t* h, const double* buf, void* counts_out, int on_device);
/* Sharded step over the library communicator: every rank runs msm_mbk_step(apply_update = 0) on the batch rows it
 * owns (msm_mbk_zero_packed if it owns none), then msm_mbk_allreduce: ONE all-reduce of the device-resident
 * [K*m sums | K counts | inertia] (RCCL on the library stream, nothing staged through the host) and the identical
 * update on every rank.  *batch_inertia = global batch inertia, counts_out (host, K) = updated cumulative counts. */
int msm_mbk_zero_packed(msm_mbk_t* h);
int msm_mbk_allreduce(msm_mbk_t* h, double* batch_inertia, void* counts_out);
int msm_mbk_reassign(msm_mbk_t* h, const void* X, msm_idx_t n, const msm_idx_t* rows, const msm_idx_t* which,
                     msm_idx_t n_reassign, double new_count, int on_device);
int msm_mbk_label(msm_mbk_t* h, const void* X, msm_idx_t n, int32_t* labels, double* inertia, int on_device);

/* ------------------------------------------------------------------------------------------
 * Pre-tICA column scan and scaling.  Replaces the fit / transform arithmetic of
 * msmbuilder.preprocessing.{StandardScaler, MinMaxScaler, MaxAbsScaler}
 * (/root/reference/msmbuilder/preprocessing/__init__.py:56-83 -- mixins over scikit-learn's
 * scalers, base.py:14-199).
 * msm_colstats: one pass over n_seq trajectories X_ptrs[s] -> n_rows[s] x n_features (common row
 * stride ld, dtype_bytes 4 or 8; device-resident or host).  out5F[5][n_features] (host, float64) =
 * per-column {count of non-NaN values, mean, sum of squared deviations M2, min, max}; NaN is a
 * missing value (scikit-learn's convention), *has_inf != 0 if an infinity was seen.
 * msm_scale_apply: mode 0 out = (x - shift) / scale, mode 1 out = x * scale + shift, every step
 * computed in float64 and rounded to the array dtype like numpy's in-place operators; shift /
 * scale are host float64 [n_features] or NULL (step skipped); out may alias X. */
int msm_colstats(const void* const* X_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int dtype_bytes,
                 msm_idx_t n_features, msm_idx_t ld, int on_device, double* out5F, int* has_inf);
int msm_scale_apply(const void* X, int dtype_bytes, msm_idx_t n_rows, msm_idx_t n_features, msm_idx_t ld,
                    const double* shift, const double* scale, int mode, void* out, msm_idx_t ld_out,
                    int on_device);
/* One pass of the radix SELECT behind RobustScaler's per-column median / percentiles (exact order
 * statistics without a sort): DEVICE-resident trajectories only.  Values map to order-preserving keys
 * (sign-flipped IEEE bits, NaN skipped).  prefix == NULL: hist[0][f][d] = number of values of column f
 * whose key digit (key >> shift) & ((1 << bits) - 1) equals d.  prefix != NULL (host uint64
 * [n_targets][n_features]): hist[t][f][d] counts only the values whose higher bits
 * key >> (shift + bits) equal prefix[t][f].  hist is host int64 [n_targets or 1][n_features][1 << bits]. */
int msm_col_digit_hist(const void* const* X_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int dtype_bytes,
                       msm_idx_t n_features, msm_idx_t ld, const uint64_t* prefix, int n_targets, int shift, int bits,
                       int64_t* hist);

/* ------------------------------------------------------------------------------------------
 * Post-clustering transition counts.  Replaces the counting loop of
 * msmbuilder.msm._transition_counts (/root/reference/msmbuilder/msm/core.py:487-596) for integer
 * labels: y_ptrs[s] -> n_rows[s] int64 labels (device-resident, e.g. KCenters.labels_, or host).
 * msm_label_range: min / max label over all sequences (*hi < *lo if there are no labels).
 * msm_label_histogram: hist[b] (host, int64[n_bins]) = #labels equal to lo + b -- the class discovery
 *   behind np.unique (core.py:544); labels outside [lo, lo + n_bins) are ignored.
 * msm_transition_counts: counts (host, int64 [n_states][n_states], row-major) [i][j] = number of
 *   (t, t + lag_time) pairs inside one sequence with state(y[t]) = i and state(y[t+lag]) = j, where
 *   state(v) = remap[v - lo] (host int32[n_bins], -1 = no mapping: the pair is dropped, as for
 *   NaN / None upstream) or v - lo when remap is NULL (then n_bins must equal n_states).  The caller
 *   divides by lag_time for the sliding-window normalisation (core.py:594). */
int msm_label_range(const msm_idx_t* const* y_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                    msm_idx_t* lo, msm_idx_t* hi, msm_idx_t* n_total);
int msm_label_histogram(const msm_idx_t* const* y_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                        msm_idx_t lo, msm_idx_t n_bins, int64_t* hist);
int msm_transition_counts(const msm_idx_t* const* y_ptrs, const msm_idx_t* n_rows, msm_idx_t n_seq, int on_device,
                          msm_idx_t lag_time, msm_idx_t lo, const int32_t* remap, msm_idx_t n_bins,
                          msm_idx_t n_states, int64_t* counts);

/* ------------------------------------------------------------------------------------------
 * dir-npy trajectories straight into HBM.  MSMBuilder's NumpyDirDataset is a directory
 * of %08d.npy files read with np.load (/root/reference/msmbuilder/dataset.py:290-331).
 * msm_npy_info: parse a .npy header (format 1.0-3.0, little-endian bool/int/uint/float): item size,
 *   kind ('f','i','u','b'), fortran_order, ndim, shape (up to 4), byte offset of the payload.
 * Loader: n_buffers reader threads, each with one pinned host buffer of buffer_bytes and its own
 *   stream, pread() buffer-sized pieces of the submitted files and copy them to the caller's device
 *   pointer; submit() validates the header, queues the pieces and returns a job id at once, wait(job)
 *   returns when that job and every earlier one are completely in HBM, so disk, PCIe and the kernels
 *   on previously loaded trajectories overlap.  nbytes must equal the file's payload size (from
 *   msm_npy_info); read / copy errors surface at wait(). */
typedef struct msm_npy_loader msm_npy_loader_t;
int msm_npy_info(const char* path, int* dtype_bytes, int* kind, int* fortran_order, int* ndim, msm_idx_t* shape4,
                 msm_idx_t* data_offset);
int msm_npy_loader_create(msm_npy_loader_t** out, int n_buffers, size_t buffer_bytes);
int msm_npy_loader_submit(msm_npy_loader_t* h, const char* path, void* dptr, msm_idx_t nbytes, msm_idx_t* job_id);
int msm_npy_loader_wait(msm_npy_loader_t* h, msm_idx_t job_id);
int msm_npy_loader_destroy(msm_npy_loader_t* h);

/* ------------------------------------------------------------------------------------------
 * Device-side generalized eigensolve.  Replaces scipy.linalg.eigh(A, b=B,
 * eigvals=(n-k, n-1)) of /root/reference/msmbuilder/decomposition/tica.py:188-194: the k LARGEST
 * solutions of A v = lambda B v (A symmetric, B symmetric positive definite, float64 n x n, host or
 * device), evals[k] descending, evecs[k][n] row j = eigenvector j, normalised v^T B v = 1, sign
 * arbitrary (LAPACK's conventions).  Runs dpotrf + dtrsm + dsyevd of rocSOLVER / rocBLAS; the library is dlopen'ed at first use
 * (MSM_ERR_STATE if it is not installed).  MSM_ERR_INVALID with LAPACK's message if B is not positive
 * definite. */
int msm_sygv_top(const double* A, const double* B, msm_idx_t n, msm_idx_t k, double* evals, double* evecs,
                 int on_device);

#ifdef __cplusplus
}
#endif
#endif /* MSMHIP_H */
