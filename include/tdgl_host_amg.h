/* Host-side set-up helpers of the MI355X TDGL path: the two loops of the AMG set-up that NumPy / SciPy ran on
 * one thread -- the Lanczos estimate of rho(D^-1 A) and the MIS(2) aggregation (tdgl_amd/amg.py).  The AMG
 * hierarchy is this port's counterpart of the reference's sparse LU factorisation of the mu Laplacian
 * (tdgl/finite_volume/operators.py:305-308).  Plain C++ with std::thread (no OpenMP runtime: safe across fork),
 * built into libtdgl_mesh.so next to include/tdgl_host_mesh.h.  Results do not depend on the number of threads.
 * No C++ exception leaves these functions: running out of memory or of threads inside a call returns -5 (NULL from
 * tdgl_host_spgemm), the thread team is wound down first.
 */
#ifndef TDGL_HOST_AMG_H
#define TDGL_HOST_AMG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* `iters` Lanczos steps on D^-1/2 A D^-1/2 (A: CSR with n rows, 32-bit indices; dinv: 1 / diagonal, 0 where the
 * diagonal is not positive) from the unit start vector v0.  alpha[iters], beta[iters] receive the tridiagonal
 * matrix (beta[j] = norm of the j-th residual; a breakdown beta[j] <= 1e-12 max(|alpha[j]|, 1) ends the run with
 * beta[j] = 0), *steps the number of steps taken, *gershgorin max_i dinv_i sum_j |A_ij|.  Dot products are summed in
 * fixed blocks of 8192 entries, blocks in order: the same numbers for any thread count.  threads <= 0: up to 16 (or the
 * environment variable TDGL_HOST_THREADS), never more than the hardware reports.  Returns 0, or -1 on bad arguments. */
int tdgl_host_lanczos(int64_t n, const int32_t *indptr, const int32_t *indices, const double *data, const double *dinv,
                      int iters, const double *v0, int threads, double *alpha, double *beta, int *steps,
                      double *gershgorin);

/* MIS(2) aggregation of the symmetric strength graph S (CSR, no diagonal, `weight` = |S_ij|) with the distinct
 * priorities `priority[n]` (1..n): synchronous rounds -- an undecided node whose priority is the largest within
 * distance 2 becomes a root, every undecided node within distance 2 of a new root is excluded -- then two passes in
 * which a node without aggregate joins the aggregate of the neighbour it is most strongly coupled to (first such
 * neighbour in row order; decisions of a pass do not see each other).  Isolated nodes are roots.  agg[n] receives the
 * aggregate of every node (roots numbered in ascending node order), *n_agg their number: exactly what
 * tdgl_amd.amg.mis2_aggregate computes with NumPy.  Returns 0, -1 on bad arguments, -2 if 200 rounds do not decide
 * every node. */
int tdgl_host_mis2_aggregate(int64_t n, const int32_t *indptr, const int32_t *indices, const double *weight,
                             const int64_t *priority, int threads, int64_t *agg, int64_t *n_agg);

/* C = A B for CSR matrices with 32-bit indices (A: rows x inner, B: inner x cols): the Galerkin products and the
 * pre-multiplied operators of the AMG set-up, which SciPy computes on one thread.  Row blocks are dealt to threads;
 * every entry is accumulated in the order (entries of A's row) x (entries of B's row), exact zeros are not stored
 * and the columns of a row come out ascending, so the result does not depend on the number of threads.
 * `tdgl_host_spgemm` returns a handle (NULL on bad arguments) and the number of stored entries in *nnz;
 * `tdgl_host_spgemm_take` copies the result into indptr[rows + 1], indices[nnz], data[nnz] (a NULL indptr discards
 * it) and releases the handle. */
void *tdgl_host_spgemm(int64_t rows, int64_t cols, const int32_t *a_indptr, const int32_t *a_indices, const double *a_data,
                       const int32_t *b_indptr, const int32_t *b_indices, const double *b_data, int threads, int64_t *nnz);
int tdgl_host_spgemm_take(void *handle, int64_t *indptr, int32_t *indices, double *data);

#ifdef __cplusplus
}
#endif
#endif
