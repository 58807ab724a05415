/*
 * tdgl_hip.h -- C-ABI of the MI355X-native TDGL time-stepping core (libtdgl_hip.so).
 *
 * The reference (loganbvh/py-tdgl v0.8.3) has no FFI: its hot path is the Python method
 * seam TDGLSolver / MeshOperators.  Each entry point below names the reference interface
 * it replaces (file:line under the reference tree).  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - All arrays are caller-owned, C-contiguous host buffers, borrowed for the duration of
 *     the call.  double = IEEE fp64; complex arrays are interleaved (re, im) pairs of
 *     doubles (numpy complex128); indices are int32.
 *   - Sites and edges are numbered as in the reference's Mesh / EdgeMesh
 *     (tdgl/finite_volume/mesh.py:45-69, edge_mesh.py:25-42); the library applies its own
 *     locality permutation internally and inverts it on every output.
 *   - Every function returns a tdgl_status (0 = ok); tdgl_last_error() returns the message
 *     of the most recent failure on that context (or globally for tdgl_create).
 *   - One host thread per context; no re-entrancy.  Device buffers and the HIP stream are
 *     owned by the context.
 */
#ifndef TDGL_HIP_H
#define TDGL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tdgl_ctx tdgl_ctx;

typedef enum {
    TDGL_OK = 0,
    TDGL_ERR_HIP = 1,          /* HIP runtime / RCCL failure */
    TDGL_ERR_ARG = 2,          /* bad argument or call order */
    TDGL_ERR_PSI_RETRIES = 3,  /* psi update failed after max_solve_retries (reference:
                                  RuntimeError at tdgl/solver/solver.py:478-483) */
    TDGL_ERR_PCG = 4,          /* Poisson solve did not reach the tolerance */
    TDGL_ERR_NOT_READY = 5,    /* hierarchy / link variables / state not set yet */
    TDGL_ERR_SCREENING = 6     /* screening iteration did not converge (reference: RuntimeError
                                  at tdgl/solver/solver.py:657-663) */
} tdgl_status;

/* Mesh description: the arrays of the reference's Mesh/EdgeMesh plus the boundary
 * conditions MeshOperators is constructed with
 * (tdgl/finite_volume/operators.py:245-280, tdgl/solver/solver.py:258-276). */
typedef struct {
    int64_t n_sites;
    int64_t n_edges;
    int64_t n_boundary_edges;
    const int32_t *edges;                 /* [n_edges, 2], i < j, lexicographic        */
    const double *areas;                  /* [n_sites]  Voronoi cell areas             */
    const double *edge_lengths;           /* [n_edges]                                 */
    const double *dual_edge_lengths;      /* [n_edges]                                 */
    const double *directions;             /* [n_edges, 2]  r_j - r_i (un-normalised)   */
    const int32_t *boundary_edge_indices; /* [n_boundary_edges] edge ids               */
    const int32_t *fixed_sites;           /* [n_fixed] terminal sites (may be NULL)    */
    int64_t n_fixed;
    int32_t fix_psi;                      /* options.terminal_psi is not None          */
    const int32_t *site_perm;             /* [n_sites] internal -> reference site id
                                             (e.g. reverse Cuthill-McKee); NULL = identity */
    double u;                             /* Layer.u      (tdgl/device/layer.py:30)    */
    double gamma;                         /* Layer.gamma  (tdgl/device/layer.py:31)    */
    int64_t n_owned;                      /* one-process-per-GPU mode: sites [0, n_owned) are
                                             owned by this rank, [n_owned, n_sites) are ghost
                                             copies (site_perm must then be NULL); 0 = all    */
} tdgl_mesh_desc;

/* One level of the algebraic-multigrid hierarchy that preconditions the mu solve
 * (replaces the SuperLU factorisation of operators.py:305-308).  CSR, int32 indices, in
 * the INTERNAL (permuted) site order for level 0.  P/R are NULL on the last level. */
typedef struct {
    int64_t n;            /* rows of A on this level                                   */
    int64_t n_coarse;     /* rows of the next level (0 on the last level)              */
    const int32_t *A_indptr;  const int32_t *A_indices;  const double *A_data;
    const double *dinv;   /* [n] 1/diag(A)  ([n_cols] on a distributed level 0)          */
    double rho;           /* estimate of the spectral radius of D^-1 A                 */
    const int32_t *P_indptr;  const int32_t *P_indices;  const double *P_data; /* n x n_coarse */
    const int32_t *R_indptr;  const int32_t *R_indices;  const double *R_data; /* n_coarse x n */
    int64_t n_cols;       /* level 0 in distributed mode: columns of A / rows of P = owned +
                             ghost sites (A is n x n_cols, P is n_cols x n_coarse); 0 = n    */
    /* Two distributed levels (tdgl_set_deep_halo_plan below); all 0 otherwise.
     * Level 0: A has a_rows >= n rows (owned + first ghost layer: z is formed there too), P has p_rows >= a_rows rows
     * (+ second ghost layer: where x is formed), n_cols >= p_rows is the whole ghost zone (where r is received);
     * A's columns are < p_rows, R is not used (NULL allowed: the fused restriction replaces it).
     * Level 1 (explicit_only = 1): no A / P / R at all -- the level exists through its explicit operators
     * (tdgl_poisson_set_collapsed_level: M [n_coarse x <= n], tdgl_poisson_set_collapsed_up: W [n x n_cols],
     * V [n x n_coarse]); n = the level-1 rows this rank forms x1 on, n_cols = the level-1 entries it forms b1 on. */
    int64_t a_rows, p_rows;
    int32_t explicit_only;
} tdgl_amg_level;

/* Adaptive time-step controller = SolverOptions fields read by the step
 * (tdgl/solver/options.py:66-73; used at solver.py:316-320, 475-485, 698-707). */
typedef struct {
    double dt_init;
    double dt_max;
    int32_t adaptive;
    int32_t adaptive_window;
    int32_t max_solve_retries;
    double adaptive_time_step_multiplier;
} tdgl_controller;

typedef struct {
    double rtol;          /* ||b - A mu||_2 <= rtol * ||b||_2                          */
    int32_t max_iter;
    int32_t nu;           /* smoother degree: Chebyshev degree / number of Jacobi sweeps,
                             before and after the coarse correction                    */
    int32_t check_every;  /* host convergence check every k PCG iterations; 0 = auto:
                             first after (previous solve's count - 1), then every one   */
    int32_t edge_currents_every_step; /* 1: J_s, J_n are formed every step like the
                                         reference's update(); 0: only on tdgl_get_state */
    int32_t smoother;     /* 0 = damped Jacobi, 1 = Chebyshev in D^-1 A                */
    double cheb_lo;       /* Chebyshev interval [cheb_lo * rho, rho]                   */
    int32_t extrapolate;  /* initial guess: 0 = mu^n, 1 = linear, 2 = quadratic extrapolation
                             in time through the last two / three solutions, 3 (default) = the
                             combination of the last guess_window solutions with the smallest
                             residual (the matrix is the same every step; dot products and the
                             small solve in double-double arithmetic, see tdgl_get_guess_gram)  */
    int32_t nu_fine;      /* smoother degree on level 0 (0 = nu).  Default 1 with nu = 2:
                             halves the level-0 passes per cycle for ~5 % more iterations  */
    int32_t precond_fp32; /* 1 (default): the operators of the V-cycle (level 0: fused restriction,
                             prolongation, smoothing step; intermediate levels: A, R, RA, AP, P)
                             are stored and streamed in fp32 and the preconditioned residual z
                             is an fp32 vector; arithmetic is fp64 and the CG itself (A p,
                             residual recurrence, dot products, convergence test) uses fp64
                             data only, so the solution meets the same rtol.  Active with
                             nu_fine = 1 and a fused restriction (z is kept in fp64 in
                             one-process-per-GPU mode, where its ghosts are exchanged).
                             2 (default): in addition the three operator streams of level 0
                             (fused restriction, prolongation, the matrix of the smoothing step)
                             are stored in IEEE binary16 -- same PCG iteration count; falls back
                             to 1 when an entry exceeds binary16's range.  0: everything fp64.  */
    int32_t guess_window; /* extrapolate = 3: number of previous solutions kept, 1..16 (0 = 12) */
    int32_t flexible_cg;  /* 1: beta = z_{k+1}.(r_{k+1} - r_k) / z_k.r_k (Polak-Ribiere, the "flexible" CG), which
                             tolerates a preconditioner that is not exactly symmetric -- the V-cycle's
                             operators are rounded to fp32 / binary16 one by one.  0 (default): beta =
                             z_{k+1}.r_{k+1} / z_k.r_k (Fletcher-Reeves).  Measured (round 3): same
                             iteration counts on the benchmark meshes (8.97 per step at 1M sites with
                             either), 32 / 33 against 30 / 33 on a 600 : 1 graded mesh, and the extra
                             dot product costs 1.5 us per iteration -- hence off by default.
                             Single-GPU recurrence only                                             */
} tdgl_poisson_options;

/* ------------------------------------------------------------------ lifetime */
int tdgl_device_count(void);
const char *tdgl_version(void);
const char *tdgl_last_error(const tdgl_ctx *ctx); /* ctx may be NULL */

/* Replaces MeshOperators.__init__ + build_operators() for the A-independent operators
 * (operators.py:245-308): uploads the mesh, builds the SELL-64 site graph, divergence /
 * gradient / Neumann-boundary data. */
int tdgl_create(tdgl_ctx **out, const tdgl_mesh_desc *mesh, int device_id);
void tdgl_destroy(tdgl_ctx *ctx);
int tdgl_synchronize(tdgl_ctx *ctx);

/* Upload the AMG hierarchy (levels[0] is the finest) and the dense pseudo-inverse
 * [n_last, n_last] of the coarsest operator.  Replaces sp.linalg.factorized
 * (operators.py:305-308). */
int tdgl_poisson_set_hierarchy(tdgl_ctx *ctx, const tdgl_amg_level *levels, int32_t n_levels,
                               const double *coarse_pinv);
int tdgl_set_poisson_options(tdgl_ctx *ctx, const tdgl_poisson_options *opts);
/* Optional accelerator of the V-cycle: M = R_0 (I - c A_0 D_0^-1) as CSR [n_1, n_0], c = the
 * level-0 smoothing coefficient in use (degree-1 smoothing).  Restricting the pre-smoothed
 * residual then is one product with M instead of a pass over A_0 plus the restriction.  Pure
 * algebraic re-association: the V-cycle is the same operator.  indptr == NULL switches it off;
 * replaced hierarchies drop it.  One-process-per-GPU mode: built from the rank's slice (columns =
 * owned + ghost sites); the partial coarse right-hand sides are summed over ranks. */
int tdgl_poisson_set_fused_restriction(tdgl_ctx *ctx, int64_t n_rows, int64_t n_cols, const int32_t *indptr,
                                       const int32_t *indices, const double *data, double c);
/* Optional, for a level 1 <= level < n_levels - 1: R A [n_coarse x n] and A P [n x n_coarse] as CSR,
 * plus P's values laid out on A P's sparsity pattern (0 where P has no entry).  The restricted
 * residual R b - (R A) x and "prolongate, then first post-smoothing step" then take one launch
 * each; the coarse levels are launch-latency bound, not bandwidth bound.  ra_indptr == NULL
 * switches it off for that level. */
int tdgl_poisson_set_fused_level(tdgl_ctx *ctx, int32_t level, const int32_t *ra_indptr, const int32_t *ra_indices,
                                 const double *ra_data, const int32_t *ap_indptr, const int32_t *ap_indices,
                                 const double *ap_data, const double *p_on_ap_data);
/* Optional: the collapsed coarse chain.  Levels below level 0 move a few MB per kernel and are bound
 * by kernel boundaries and dependent memory round trips; the host (tdgl_amd/amg.py:
 * collapsed_operators) re-associates the SAME V-cycle into fewer, denser operators:
 *   - tdgl_poisson_set_collapsed_level(level, M): M = R (I - A S) [n_coarse x n] (CSR) of an
 *     intermediate level, S = the two-step pre-smoothing polynomial.  The next level's right-hand
 *     side M b and the pre-smoothing x = S b then share one launch.  m_indptr == NULL: off.
 *   - tdgl_poisson_set_collapsed_up(level, W, V): the way up of that level as explicit operators,
 *     e = W b + V e_next with W = T_x S + T_b [n x n] (pre- and post-smoothing as one polynomial in A)
 *     and V = T_x P [n x n_coarse] (CSR both): together with M the level takes two launches and its
 *     pre-smoothed iterate is never formed.  Needs M; w_indptr == NULL: off.
 *   - tdgl_poisson_set_collapsed_tail(t): everything from level t->level down.
 *       mode 0: e = G b, G = dense [n, n] (the cycle of that level formed explicitly, or the exact
 *               pseudo-inverse of its operator);
 *       mode 1: y = G b (G dense [g_rows, n]), e = W [b ; y] + V y[0 : v_cols] (W CSR over the
 *               concatenated vector, [n, n + g_rows]; V dense [n, v_cols], v_cols may be 0).
 *     nu / smoother / cheb_lo are the smoother settings the operators were built for; the library
 *     falls back to the plain kernel sequence while tdgl_poisson_options differ.  t == NULL: off.
 * Replaced hierarchies drop both. */
typedef struct {
    int32_t level;
    int32_t mode;
    const double *G;
    int64_t g_rows;                /* mode 1 */
    const int32_t *W_indptr;  const int32_t *W_indices;  const double *W_data;   /* mode 1 */
    const double *V;               /* mode 1: dense row-major [n, v_cols] (NULL when v_cols = 0) */
    int64_t v_cols;
    int32_t nu, smoother;
    double cheb_lo;
} tdgl_collapsed_tail;
int tdgl_poisson_set_collapsed_level(tdgl_ctx *ctx, int32_t level, const int32_t *m_indptr, const int32_t *m_indices,
                                     const double *m_data);
int tdgl_poisson_set_collapsed_up(tdgl_ctx *ctx, int32_t level, const int32_t *w_indptr, const int32_t *w_indices,
                                  const double *w_data, const int32_t *v_indptr, const int32_t *v_indices,
                                  const double *v_data);
int tdgl_poisson_set_collapsed_tail(tdgl_ctx *ctx, const tdgl_collapsed_tail *tail);
/* Direct mu solve for small meshes.  The reference factorises L_mu once (operators.py:305-308) and
 * back-substitutes every step (solver.py:516); on the GPU a triangular solve is a chain of dependent
 * launches, so the factorisation's counterpart here is the explicit pseudo-inverse G = pinv(A) of
 * A = -diag(a) L_mu (symmetric, null space = constants), dense [n, n] row major, in the site order of
 * the hierarchy's level 0, and every solve is ONE matrix-vector product mu = G b (k_dense_solve):
 * n^2 doubles streamed per step -- cheaper than the ~60 dependent launches of the AMG-PCG path up to
 * ~10^4 sites (268 MB / 55 us at 5.8k sites).  The zero-mean solution comes out by construction
 * (G annihilates constants on both sides).  While set, tdgl_run / tdgl_poisson_solve use it instead of
 * the PCG iteration (iteration counts read 0; tdgl_poisson_solve still reports the true relative
 * residual).  G == NULL switches back to AMG-PCG.  Single GPU only (TDGL_ERR_ARG otherwise);
 * a hierarchy must have been set (its level-0 matrix measures the residual). */
int tdgl_poisson_set_dense_inverse(tdgl_ctx *ctx, const double *G, int64_t n);
/* The same, with G computed on the device from the hierarchy's level-0 matrix (the set-up counterpart of
 * operators.py:305-308 without the host): M = A + (s/n) 1 1^T assembled as symmetric 128 x 128 tiles and
 * inverted in place by the library's own blocked symmetric sweep (Gauss-Jordan without pivoting on the
 * positive definite M, 64-row pivot blocks: k_gj_panel / k_gj_update, csrc/dense.inc -- no rocSOLVER /
 * rocBLAS is loaded), then 1 1^T / (s n) removed.  *seconds (may be NULL) = wall time of the call.
 * TDGL_ERR_ARG when a pivot is not positive (a mesh in several pieces has a larger null space: stay with
 * AMG-PCG, or pass a host-computed G to tdgl_poisson_set_dense_inverse). */
int tdgl_poisson_build_dense_inverse(tdgl_ctx *ctx, double *seconds);

/* Substructured direct mu solve for mid-size meshes (between the dense inverse above and AMG-PCG): one
 * level of nested dissection with every factor explicit, so that the reference's back-substitution
 * (solver.py:516 with the LU of operators.py:305-308) becomes four launches of dense row products.
 * The context's site order must be "interiors of the P parts, part by part, then the separator"
 * (tdgl_mesh_desc.site_perm; host layer: tdgl_amd/substructure.py).  With G_p = A_pp^-1,
 * E_p = G_p A_pS and the Schur complement S_c = A_SS - sum_p A_Sp E_p:
 *     w = [ G_p b_p | b_S - sum_p E_p^T b_p | (G_p 1)^T b_p ]       k_sub_down (rows of dense segments)
 *     x_S = pinv(S_c) w_S                                             k_dense_sym_tiles + k_dense_sym_finish
 *     x_p = w_p - E_p x_S,  minus the mean (sum x = sum_p w_{n+p} + u^T x_S)   k_sub_up
 * The way down is given as rows of segments over one value pool: output row r =
 * sum_{k in [seg_ptr[r], seg_ptr[r+1])} vals[seg_val[k] .. + seg_len[k]) . b[seg_x[k] .. + seg_len[k]),
 * rows [0, n_interior) = G rows, [n_interior, n) = separator rows (b_S as a one-entry segment with value 1,
 * minus the E^T rows), [n, n + n_parts) = (G_p 1)^T.  `schur` is the SINGULAR Schur complement; the
 * library forms its pseudo-inverse on the device (the blocked sweep of tdgl_poisson_build_dense_inverse).
 * s == NULL switches back to AMG-PCG.  Single GPU only; a hierarchy must have been set. */
typedef struct {
    int64_t n_interior, n_sep;
    int32_t n_parts;
    const int32_t *part_ptr;  /* [n_parts + 1] interior ranges                                  */
    const int32_t *seg_ptr;   /* [n_interior + n_sep + n_parts + 1]                             */
    const int64_t *seg_val;   /* [n_seg] offset into vals                                       */
    const int32_t *seg_x;     /* [n_seg] first entry of b                                       */
    const int32_t *seg_len;   /* [n_seg]                                                        */
    const double *vals;
    int64_t n_vals;
    const int32_t *sep_ptr;   /* [n_parts + 1] into sep_idx                                     */
    const int32_t *sep_idx;   /* separator-local indices of the sites part p touches            */
    const int64_t *e_off;     /* [n_parts] offset of E_p (n_p x s_p, row major) in e_vals       */
    const double *e_vals;     /* (checked for its size only: the way up reads the pool's -E_p^T blocks) */
    int64_t n_e;
    const double *u;          /* [n_sep]  -sum_p E_p^T 1                                        */
    const double *schur;      /* [n_sep, n_sep] row major                                       */
} tdgl_substructure;
int tdgl_poisson_set_substructure(tdgl_ctx *ctx, const tdgl_substructure *s, double *seconds);
/* Second level of the nested dissection, for meshes whose first-level separator is too large for a dense
 * matrix (~32k to ~650k sites).  Site order: part interiors, then the fine separators S'_Q of the Q
 * super-blocks (the parts of a super-block lie inside it), then the top separator T, which covers every
 * edge between two super-blocks (host layer: substructure.py: substructure_order2).  The first level is set
 * with tdgl_poisson_set_substructure and `schur == NULL` (n_sep = |S'| + |T|; its `u` is not used); this call
 * then describes the same construction on the first level's Schur complement S1 = A_SS - sum_p A_Sp E_p, a
 * matrix on [S'_0 .. S'_{Q-1} | T] in which the S'_Q are decoupled from each other: n_interior = |S'|,
 * n_sep = |T|, "parts" = the S'_Q, G_q = S1_qq^-1, E_q = G_q S1_qT, `schur` = S1_TT - sum_q S1_Tq E_q
 * (singular; its pseudo-inverse is formed on the device), segments over the first level's separator vector.
 * The gauge is carried down as a functional: the rows [n_interior + n_sep, + n_parts) are (G_q v_q)^T and
 * u = v_T - sum_q E_q^T v_q with v = 1_S - sum_p E_p^T 1 of the first level, so that
 * sum x = sum_p (G_p 1)^T b_p + sum_q (G_q v_q)^T r_q + u^T x_T.  A solve is six launches (down, down, the
 * dense pair on T, up, up); the time loop, the run-ahead loop and the in-loop guard use it like the one-level
 * form.  Until this call has succeeded the context keeps solving with AMG-PCG.
 * THREE levels: call it twice -- first with `schur == NULL` (the second level, whose separator T is then cut once more:
 * super-super-blocks, parts T'_R, top-top separator TT; order ... | T'_0 .. | TT), then with the third level on the
 * second level's Schur complement (n_interior = |T'|, n_sep = |TT|, functional = the second level's u), which
 * carries the dense `schur`.  The last level given is the one with the dense matrix. */
int tdgl_poisson_set_substructure_inner(tdgl_ctx *ctx, const tdgl_substructure *inner, double *seconds);
/* Optional, per level (0: the first, 1: the second): the sparse coupling block of the level's matrix, separator rows x
 * interior columns (A_SI as CSR [n_sep, n_interior]; for the second level S1_TS').  The level's separator right-hand
 * side is then formed as r_S = b_S - A_SI y_I by a sparse product BEHIND the dense one (y_I = G b_I must be complete:
 * a launch of its own) and the separator rows of the level's description must hold their identity segment only --
 * the -E_p^T rows are not streamed on the way down any more (a fifth of a two-level solve's bytes at 250k sites, for
 * two more launches: pays from ~200k sites on).  A level described that way is not used (the context keeps solving with
 * AMG-PCG) until its coupling block has arrived.  Call after the level's factors are set. */
/* Solver choice in the time loop, for contexts that hold BOTH a direct mu solve and the AMG hierarchy.  A direct
 * solve costs the same whatever the state; AMG-PCG from the projection guess costs next to nothing once the state is
 * stationary (a transport current through a strip: 0 iterations) and more than the direct solve while vortices move
 * (251k-site strip, stationary: 7.1k against 4.3k steps/s; 250k-site film in a field: 2.4k against 4.3k).  on = 1: while
 * |psi|^2 has changed by less than 1e-4 per step for 64 accepted steps -- or has been falling by 10 % or more from one
 * window of 64 steps to the next three times in a row, below 2e-3 -- the direct solve is paused (one
 * synchronisation per step, AMG-PCG); when the running mean of the PCG iterations exceeds 2.5 it comes back, and
 * the next pause has to wait twice as long (256 steps at first, and again after every pause that lasted 1,024 steps or more).  on = 0: off (the direct solve always); on < 0: query
 * only.  *switches / *paused (may be NULL): changes so far, the current state.  The host layer switches it on from
 * 150k sites.  The choice is a function of the run's recent history; tdgl_set_state, tdgl_begin_stage, tdgl_set_mu_boundary,
 * tdgl_set_epsilon and this call with on >= 0 return it to its starting point (direct solve, no windows, the 256-step
 * wait), so that a restart from a checkpoint takes the same path whatever preceded it.  While the direct solve is paused
 * mu meets pcg_rtol (the iterative solve's tolerance) instead of the factors' ~1e-14. */
int tdgl_direct_switching(tdgl_ctx *ctx, int32_t on, int64_t *switches, int32_t *paused);
int tdgl_poisson_set_substructure_coupling(tdgl_ctx *ctx, int32_t level, const int32_t *indptr, const int32_t *indices,
                                           const double *data);
/* The resident factors (every level and coupling block described) as the PRECONDITIONER of the CG instead of as the
 * solver -- the counterpart of the reference's `mu_laplacian_lu(rhs)` (tdgl/solver/solver.py:516, factorised at
 * tdgl/finite_volume/operators.py:305-308) for meshes of 0.65 - 1.3 million sites, where streaming the fp64 factors
 * (2.9 GB at 1M sites) costs as much as the 7-12 AMG-preconditioned iterations it would replace:
 *   - fp32_storage != 0: the value pools and the top separator's tiles are converted to fp32 (half the bytes; the
 *     fp64 pools are released); every multiply-add stays fp64.  One application then contracts the residual by
 *     ~1e-5 .. 1e-6 -- ONE CG iteration wherever the projection guess is good to 1e-4 --, and the CG around it (A p
 *     with the fp64 matrix, recurrences, dot products, convergence test) is unchanged, so mu meets the same pcg_rtol;
 *   - the context keeps ITS site order (reverse Cuthill-McKee: what the stencil kernels and the hierarchy are fastest
 *     in); the factors were described in the dissection's order, and site_map[i] = the caller's site at dissection
 *     position i ([n_sites], a permutation) lets an application gather the residual and scatter the result;
 *   - per solve the library takes the cheaper of the two preconditioners by PREDICTED cost: the decades to go (the
 *     residual of the projection guess, known on the host before anything is queued) over the decades per iteration
 *     observed for each, times what one application was MEASURED to take on this device at this call (*t_apply_us for
 *     the factors, *t_vcycle_us for the AMG V-cycle; may be NULL).
 * tdgl_poisson_precond_choice: 0 = by predicted cost (default), 1 = always the factors, 2 = never (tests, A/B runs).
 * A solve that the factors leave just above the tolerance is FINISHED by the V-cycle when that is predicted cheaper than a
 * second application (the same rule on the residual that is left; the CG restarts from the iterate with beta = 0).
 * tdgl_get_precond_direct_stats: out4 = {solves that began with the factors, CG iterations taken with the factors, solves
 * with the V-cycle alone, iterations taken with the V-cycle (hand-overs included)} since the last reset, out4d =
 * {t_apply_us, t_vcycle_us, decades per application observed, hand-overs since the last reset}.
 * With fp32 storage, a level whose parts all have at most 256 rows (and which has parts enough to fill the chip: 256) keeps
 * only the 16 x 16 tiles on or below the diagonal of its symmetric G blocks -- 56 % of the entries at 144 rows -- and its way
 * down uses every tile twice (k_sub_down_sym).  tdgl_get_precond_direct_layout: sym_rows3[k] = the rows of a part level k
 * stages in LDS when it is stored that way, 0 when it keeps whole blocks.  (TDGL_PD_SYM=0 in the environment: whole blocks
 * everywhere, for A/B measurements; =2: tiles also on levels of fewer than 256 parts, for tests on small meshes.) */
int tdgl_poisson_set_substructure_precond(tdgl_ctx *ctx, const int32_t *site_map, int32_t fp32_storage, double *t_apply_us,
                                          double *t_vcycle_us);
int tdgl_poisson_precond_choice(tdgl_ctx *ctx, int32_t mode);
/* The same preconditioner in one-process-per-GPU mode: RANK-LEVEL nested dissection (csrc/schur.inc; host side
 * tdgl_amd/schur_dd.py).  Gamma = a vertex cover of the edges between ranks (n_gamma sites, one global numbering);
 * every rank describes the factors of ITS interior block A_II -- its owned sites outside Gamma, in a local dissection
 * order; positive definite, so the last level's Schur complement gets a plain inverse and no gauge is carried --
 * between tdgl_poisson_schur_begin and tdgl_poisson_schur_finish with the calls of the single-GPU solve
 * (tdgl_poisson_set_substructure / _inner / _coupling on a vector of n_interior entries).
 *   tdgl_poisson_schur_complement: out[n_gamma, n_gamma] = A_GI A_II^-1 A_IG, formed column by column with the resident
 *     fp64 factors; the host layer sums S = A_GG - (these) over the ranks -- the ONLY communication of the set-up;
 *   tdgl_poisson_schur_finish: S (the same on every rank) is pseudo-inverted on the device, the factors are stored in
 *     fp32 (fp32_storage != 0), both preconditioners are timed (collectively: every rank calls it at the same point);
 *   tdgl_poisson_set_precond_times: the host layer hands every rank the SAME pair of times (the maximum over ranks), so
 *     that the per-solve choice of pcg_solve is the same everywhere.
 * One application: y_I = A_II^-1 r_I, t = r_G - sum_r A_GI y_I (ONE all-reduce of n_gamma doubles), x_G = S^+ t
 * (replicated), x_I = y_I - A_II^-1 A_IG x_G: with exact factors a direct solve (what the reference's LU is,
 * tdgl/solver/solver.py:516), with fp32 storage ~6 decades per application -- per step one such sum, one exchange of
 * z's ghost layer and the CG's one sum of 3 x 1024 partials, instead of ~9 AMG-preconditioned iterations of three
 * collectives each. */
typedef struct {
    int64_t n_interior;               /* owned sites outside Gamma */
    int64_t n_gamma;                  /* |Gamma|, the same on every rank */
    int64_t n_gamma_owned;
    const int32_t *interior;          /* [n_interior] local (owned) site at local dissection position i */
    const int32_t *gamma_owned_local; /* [n_gamma_owned] local site ... */
    const int32_t *gamma_owned_gid;   /* ... and its position in Gamma */
    const int32_t *gi_indptr, *gi_indices; /* A_GI [n_gamma x n_interior] CSR (columns: local dissection positions) */
    const double *gi_data;
    const int32_t *ig_indptr, *ig_indices; /* A_IG [n_interior x n_gamma] CSR */
    const double *ig_data;
} tdgl_schur_piece;
int tdgl_poisson_schur_begin(tdgl_ctx *ctx, const tdgl_schur_piece *piece);
int tdgl_poisson_schur_complement(tdgl_ctx *ctx, double *out);
int tdgl_poisson_schur_finish(tdgl_ctx *ctx, const double *S, int32_t fp32_storage, double *t_apply_us, double *t_vcycle_us);
int tdgl_poisson_set_precond_times(tdgl_ctx *ctx, double t_apply_us, double t_vcycle_us);
int tdgl_get_precond_direct_stats(tdgl_ctx *ctx, int64_t *out4, double *out4d, int32_t reset);
int tdgl_get_precond_direct_layout(tdgl_ctx *ctx, int32_t *sym_rows3);
/* The same storage for the direct SOLVE (fp64 factors; reference: the LU solve of tdgl/solver/solver.py:516): after every level
 * has been described, symmetric_tiles != 0 repacks the fp64 pool of each level that qualifies (all parts <= 256 rows, >= 256
 * parts) into tiles on or below the diagonal + -E^T rows on 64-byte lines and switches that level's way down to the tile
 * kernel; the other levels are left as uploaded.  Not after tdgl_poisson_set_substructure_precond (which lays the pools out
 * itself).  tdgl_get_precond_direct_layout reports the outcome. */
int tdgl_poisson_set_substructure_layout(tdgl_ctx *ctx, int32_t symmetric_tiles);
/* The same solve with every factor formed ON THE DEVICE from the hierarchy's level-0 matrix: the caller
 * passes index arrays only (host layer: substructure.plan_for_device).  Per part the interior block is
 * read from the resident SELL matrix and inverted by the batched form of the blocked symmetric sweep
 * (all parts at once, blockIdx.y = part); E_p = G_p A_pS and C_p = A_Sp E_p are sums of the two or
 * three rows of G_p / E_p that the entries of A_pS select (ent_*: the entries grouped by (part,
 * separator site) pair); the Schur complement is assembled row by row (A_SS as CSR minus the C_p rows of
 * the pairs that touch the site, node_*: in ascending pair order -- deterministic) straight into the
 * padded matrix the sweep inverts.  The value pool the way down reads is laid out as
 * [1.0 | G_p at g_off[p], n_p x n_p | -E_p^T at et_off[p], s_p x n_p | G_p 1 at gvec_off + row]. */
typedef struct {
    int64_t n_interior, n_sep;
    int32_t n_parts;
    const int32_t *part_ptr;   /* [n_parts + 1]                                                         */
    const int32_t *sep_ptr;    /* [n_parts + 1] pairs of each part                                      */
    const int32_t *sep_idx;    /* [n_pairs] separator-local site of each pair (ascending within a part) */
    const int32_t *ent_ptr;    /* [n_pairs + 1] entries of A_pS of each pair                            */
    const int32_t *ent_row;    /*   interior row (internal site index)                                  */
    const double *ent_val;     /*   A[row][separator site]                                              */
    const int32_t *node_ptr;   /* [n_sep + 1] pairs touching each separator site                        */
    const int32_t *node_pair;  /*   pair ids, ascending                                                 */
    const int32_t *ass_indptr; /* A_SS as CSR [n_sep, n_sep], sorted columns                            */
    const int32_t *ass_indices;
    const double *ass_data;
    const int32_t *seg_ptr;    /* the way down, as in tdgl_substructure                                 */
    const int64_t *seg_val;
    const int32_t *seg_x;
    const int32_t *seg_len;
    const int64_t *g_off;      /* [n_parts] */
    const int64_t *et_off;     /* [n_parts] */
    int64_t gvec_off, n_vals;
    const int64_t *e_off;      /* [n_parts] */
    int64_t n_e;
} tdgl_substructure_plan;
int tdgl_poisson_build_substructure(tdgl_ctx *ctx, const tdgl_substructure_plan *plan, double *seconds);
/* out3 = {solves that fell back from the fp32-stored to the fp64 operators, iterations of the last
 * solve, 0 (was: a captured iteration-pair hipGraph in use; the replay was removed in round 5, plain launches are
 * faster since ROCm 7.2 -- profiles/EXPERIMENTS.md)}. */
int tdgl_get_poisson_stats(tdgl_ctx *ctx, int64_t *out3);
/* How well the first batch of PCG iterations of a solve was sized.  The host queues a batch without looking at the
 * residual; iterations beyond convergence freeze themselves but still cost their launches, a batch that is too short
 * costs one more look per missing iteration.  The batch is ceil(log10(|r0| / (rtol |b|)) / rate - 1/4) with |r0| the
 * residual of the initial guess -- known on the host from the guess's Gram data before anything is queued -- and
 * `rate` the running mean of the decades per iteration observed so far (environment TDGL_PCG_PREDICT=last: the
 * previous solve's count, the rule of rounds 1-3).  out3 = {iterations queued, iterations up to convergence, looks
 * after the first batch} since the context was created; *rate (may be NULL) the current estimate. */
int tdgl_get_pcg_prediction_stats(tdgl_ctx *ctx, int64_t *out3, double *rate);

/* Storage of the V-cycle's operators in effect (after a solve): 0 fp64, 1 fp32, 2 fp32 and binary16 on
 * level 0 (tdgl_poisson_options.precond_fp32). */
int tdgl_get_precond_storage(tdgl_ctx *ctx, int32_t *mode);

/* Quality of the last solve's initial guess: number of basis vectors it was projected on
 * (extrapolate = 3; 0 = none) and ||b - A x0|| / ||b||. */
int tdgl_get_guess_stats(tdgl_ctx *ctx, int32_t *vectors, double *initial_relres);
/* The Gram matrix G_ij = y_i . y_j of the projection guess's window, y_j = A x_j (k <= 16 vectors,
 * row-major [k, k], oldest first, rounded to fp64 from the double-double sums the library keeps; the
 * newest vector's row and column are exact only after the next solve has started -- until then they
 * hold y_j . b_newest): a global quantity, identical on every rank of a decomposed run (tests). */
int tdgl_get_guess_gram(tdgl_ctx *ctx, int32_t *k, double *G_rowmajor);
/* Host-only: c = argmin ||b - Y c||_2 from G = Y^T Y and g = Y^T b given as double-double (hi, lo) pairs
 * (G_pairs [k, k, 2], g_pairs [k, 2], oldest vector first): L D L^T in double-double arithmetic from the
 * newest vector to the oldest; a vector whose pivot is below cut * G_jj is left out (c_j = 0).
 * *used (may be NULL) = vectors kept.  No device work. */
int tdgl_host_solve_gram(int32_t k, const double *G_pairs, const double *g_pairs, double cut, double *c,
                         int32_t *used);

/* ------------------------------------------------------------------ one process per GPU
 * The reference is single-process.  Here the mesh is cut into `world` pieces (host layer:
 * tdgl_amd/partition.py); each rank creates its context from its sub-mesh (owned sites first,
 * ghosts after, tdgl_mesh_desc.n_owned) and registers which owned sites every neighbour needs
 * and where each neighbour's values land among the ghosts.  Level 0 of the AMG hierarchy is
 * sliced per rank, coarser levels are replicated.  Per step the library exchanges ghost values
 * of psi, of the PCG direction / smoothing iterates and of mu, and sums dot products and the
 * restricted residual over ranks. */
typedef struct {
    int32_t rank, world;
    int64_t n_global;                /* total number of sites                             */
    int32_t n_neighbors;
    const int32_t *neighbor_ranks;   /* [n_neighbors]                                     */
    const int32_t *send_ptr;         /* [n_neighbors + 1] offsets into send_idx           */
    const int32_t *send_idx;         /* owned local site ids to send, grouped by neighbour */
    const int32_t *recv_ptr;         /* [n_neighbors + 1] offsets into the ghost range:
                                        neighbour k fills sites n_owned + recv_ptr[k] ...  */
} tdgl_halo_plan;
int tdgl_set_halo_plan(tdgl_ctx *ctx, const tdgl_halo_plan *plan);

/* Two distributed AMG levels with ONE vector exchange per PCG iteration (host: tdgl_amd/partition.py DeepPlanner).
 * The aggregates of level 0 lie inside ranks, so level 1 has owners; every rank forms redundantly what it would
 * otherwise receive (x1 on the level-1 rows its prolongation reads, b1 on the columns W1 reads there, z on its first
 * ghost layer) from the residual r on a DEEP ghost zone -- the closure of those stencils, 5-9 % of the owned rows at
 * 500k rows per rank.  Per iteration a rank then communicates three times: this exchange of r, the sum of the
 * partial level-2 right-hand sides, the CG's dot products (before: the ghosts of r and of z, a level-1-sized sum, the
 * dot products).  Local vector layout: [owned | ghost layer 1 (tdgl_halo_plan's ghosts) | layer 2 | deeper], n_ext
 * entries; neighbour k sends the owned entries send_idx[send_ptr[k] ...] and its values land in the ghost entries
 * recv_idx[recv_ptr[k] ...] (any order: an unpack kernel scatters them); a neighbour may have an empty send or
 * receive list.  Call after tdgl_set_halo_plan and before tdgl_poisson_set_hierarchy, whose level 0 must then
 * carry a_rows / p_rows / n_cols = n_ext and whose level 1 must be explicit_only. */
typedef struct {
    int64_t n_ext;
    int32_t n_neighbors;
    const int32_t *neighbor_ranks;   /* [n_neighbors] */
    const int32_t *send_ptr;         /* [n_neighbors + 1] */
    const int32_t *send_idx;         /* owned local ids */
    const int32_t *recv_ptr;         /* [n_neighbors + 1] */
    const int32_t *recv_idx;         /* ghost local ids in [n_owned, n_ext) */
    int64_t l1_interior;             /* leading level-1 rows whose restriction b1 = F r reads owned fine entries only:
                                        formed while the exchange is in flight (peer-mapped transport) */
} tdgl_deep_halo_plan;
int tdgl_set_deep_halo_plan(tdgl_ctx *ctx, const tdgl_deep_halo_plan *plan);
/* RCCL transport (xGMI): rank 0 makes the 128-byte id, the host layer broadcasts it, every rank
 * calls tdgl_comm_init_rccl.  Collectives run on the context's stream. */
int tdgl_comm_unique_id(char *out128);
int tdgl_comm_init_rccl(tdgl_ctx *ctx, const char *id128);
/* Host-callback transport (tests: lets two ranks share one GPU, which RCCL refuses).  Buffers are
 * host arrays of doubles; offsets are in doubles; return 0 on success.
 * allreduce op: 0 = sum, 1 = max (in place). */
typedef int (*tdgl_halo_fn)(void *user, const double *send, const int64_t *send_off, double *recv,
                            const int64_t *recv_off, int32_t n_neighbors, const int32_t *neighbor_ranks);
typedef int (*tdgl_allreduce_fn)(void *user, double *buf, int64_t count, int32_t op);
int tdgl_comm_init_callbacks(tdgl_ctx *ctx, tdgl_halo_fn halo, tdgl_allreduce_fn allreduce, void *user);
/* Peer-mapped transport (the ranks of ONE node; csrc/ipc.inc): every rank owns an inbox in device memory, exported
 * with hipIpcGetMemHandle and opened by the others -- on the same GPU or on a peer over xGMI.  A halo exchange is one
 * launch that stores the owned values straight into the neighbours' inboxes and raises a flag there, and one launch
 * that waits for the neighbours' flags and scatters the inbox into the ghost entries; an all-reduce is one launch that
 * stores the vector into slot [rank] of every inbox and one that adds the slots in rank order (the same bits on every
 * rank).  Everything runs on the context's one stream: no host synchronisation, no second stream, and work that reads
 * no ghost values is queued between the two launches of an exchange (the overlap is then always on).  Waits are bounded
 * (tdgl_comm_ipc_set_timeout; 120 s unless that call or TDGL_IPC_TIMEOUT_S says otherwise -- the host layer meets at
 * a barrier before it queues a batch, so the bound only has to cover what happens inside one): a rank that died turns
 * into TDGL_ERR_HIP at the end of tdgl_run, not into a hang -- on EVERY rank: the rank whose wait timed out stops
 * sending and poisons its flags at all peers, whose waits then fail at once instead of consuming stale ghosts.
 * Set-up (after tdgl_set_halo_plan / tdgl_set_deep_halo_plan): every rank calls tdgl_comm_ipc_export (handle64: the
 * 64-byte IPC handle of its inbox; table[4 + 6 world]: where in that inbox each rank's values land), the host layer
 * all-gathers both, every rank calls tdgl_comm_init_ipc with the `world` handles and tables in rank order. */
int tdgl_comm_ipc_export(tdgl_ctx *ctx, char *handle64, int64_t *table, int64_t table_len);
int tdgl_comm_init_ipc(tdgl_ctx *ctx, const char *handles, const int64_t *tables);
int tdgl_comm_ipc_set_timeout(tdgl_ctx *ctx, double seconds);
/* Overlap of the halo exchanges with the ghost-free rows: the stencil kernels that follow an
 * exchange (psi Laplacian + rhs; level-0 residual of the V-cycle; A p of the CG; edge currents)
 * run their leading ghost-free part on the compute stream while the exchange travels on a second
 * HIP stream, and the rest after it.  Needs the owned sites numbered interior first
 * (tdgl_amd.partition does); interior_rows reports the prefix the library found.
 * mode: 0 = never, 1 = automatic (default: only when the ghost-free part is large enough to pay
 * for the two cross-stream dependencies an overlapped exchange costs), 2 = always.
 * tdgl_get_comm_overlap reports whether exchanges are overlapped with the current plan. */
int tdgl_set_comm_overlap(tdgl_ctx *ctx, int32_t on);
int tdgl_get_comm_overlap(tdgl_ctx *ctx, int32_t *enabled, int64_t *interior_rows);
/* Transport self-test: ONE exchange of the ghost entries of a caller-supplied vector (vec_inout: n_sites * width
 * doubles, or n_ext with deep = 1 -- the owned entries go out, the ghost entries come back filled) / ONE in-place sum or
 * maximum of `count` doubles over the ranks (as_f32: carried as floats, like the level-2 right-hand side), through
 * whatever transport the context was given.  Every rank must call them in the same order.  The host layer
 * (DistributedTDGL.selftest, bench.py --selftest) fills the vectors with functions of the GLOBAL site id and checks what
 * arrives: the first contact of N real ranks then yields "rank 3, neighbour 5, entry 17" instead of a hang. */
int tdgl_comm_test_halo(tdgl_ctx *ctx, double *vec_inout, int32_t width, int32_t deep);
int tdgl_comm_test_allreduce(tdgl_ctx *ctx, double *buf_inout, int64_t count, int32_t op, int32_t as_f32);
/* Communication counters of this rank since creation / the last reset:
 * out4 = {halo exchanges, bytes sent in them, all-reduces, bytes all-reduced}. */
int tdgl_get_comm_stats(tdgl_ctx *ctx, int64_t *out4, int32_t reset);

/* ------------------------------------------------------------------ inputs */
/* MeshOperators.set_link_exponents (operators.py:310-383): A[n_edges, 2], dimensionless.
 * Recomputes U_e = exp(-i A_e . d_e) and the covariant Laplacian / gradient values. */
int tdgl_set_link_exponents(tdgl_ctx *ctx, const double *A);
/* Time-dependent applied vector potential (solver.py:626-642): A_new[n_edges, 2] evaluated at
 * this step's time, dt_prev = the previous step's dt (state["dt"]).  Stores
 * dA/dt|_e = ((A_new - A_prev) / dt_prev) . e_hat, which enters the Poisson right-hand side and
 * J_n of the following tdgl_run step (solver.py:508-510, 519), recomputes the link variables
 * (operators.py:346-383) and makes A_new the new A_prev. */
int tdgl_update_link_exponents(tdgl_ctx *ctx, const double *A_new, double dt_prev);
/* The common special case A(t) = f(t) * A_base (a time-dependent factor times a static field, e.g.
 * tdgl/sources/scaling.py LinearRamp * tdgl/sources/constant.py ConstantField -- the reference's
 * flagship field-ramp example) without re-uploading A every step:
 *   tdgl_set_link_exponents_base  uploads A_base [n_edges, 2] once and sets A = scale * A_base
 *                                 (static semantics: dA/dt = 0, like tdgl_set_link_exponents);
 *   tdgl_update_link_scale        A <- scale * A_base with dA/dt from the previous A, exactly
 *                                 tdgl_update_link_exponents(scale * A_base, dt_prev);
 *   tdgl_set_link_ramp            lets tdgl_run evaluate the factor itself before every step:
 *                                 f(t) = initial + (final - initial) * clip((t - tmin)/(tmax - tmin), 0, 1)
 *                                 (LinearRamp), with dt_prev = Runner.dt; on = 0 switches it off.
 *   tdgl_get_link_scale           the factor of the current A (for saving A_applied). */
int tdgl_set_link_exponents_base(tdgl_ctx *ctx, const double *A_base, double scale);
int tdgl_update_link_scale(tdgl_ctx *ctx, double scale, double dt_prev);
int tdgl_set_link_ramp(tdgl_ctx *ctx, int32_t on, double tmin, double tmax, double initial, double final_value);
int tdgl_get_link_scale(tdgl_ctx *ctx, double *scale);
/* self.epsilon (solver.py:191-216, 645-648). */
int tdgl_set_epsilon(tdgl_ctx *ctx, const double *epsilon);
/* self.mu_boundary (solver.py:289, 325-345): indexed by position in boundary_edge_indices. */
int tdgl_set_mu_boundary(tdgl_ctx *ctx, const double *mu_boundary);
/* Time-dependent terminal currents and disorder without a host round trip per step: piecewise-linear
 * tables (constant outside the node range) that tdgl_run evaluates at the time of every step, the way
 * tdgl_set_link_ramp does for A(t).
 *   tdgl_set_mu_boundary_table: update_mu_boundary (solver.py:325-345) for tabulated currents.  Group g
 *     (one per terminal) covers the boundary-edge POSITIONS group_pos[group_ptr[g] .. group_ptr[g+1])
 *     and carries the current density density[g * n_nodes + k] at times[k] (already
 *     -(1/length_g) * sum of the other terminals' currents); mu_boundary is refreshed only in steps
 *     where a density changed, like the reference.  Densities and mu_boundary start at 0.  In the run-ahead loop
 *     (tdgl_run) the device evaluates the table itself, with the host's arithmetic operation for operation.
 *   tdgl_set_epsilon_table: epsilon(r, t) = factor(t) * epsilon0(r) (update_epsilon, solver.py:364-381,
 *     for a separable disorder parameter).
 * n_nodes = 0 switches a table off. */
int tdgl_set_mu_boundary_table(tdgl_ctx *ctx, int32_t n_nodes, const double *times, int32_t n_groups,
                               const int32_t *group_ptr, const int32_t *group_pos, const double *density);
int tdgl_set_epsilon_table(tdgl_ctx *ctx, const double *epsilon0, int32_t n_nodes, const double *times,
                           const double *factor);
/* Initial / seed values of psi [n] complex and mu [n] (solver.py:284-288, 732-752). */
int tdgl_set_state(tdgl_ctx *ctx, const double *psi, const double *mu);
/* SolverOptions fields + resets the controller state (tentative_dt = dt_init,
 * d_psi_sq_vals = [], Runner.dt = dt_init; solver.py:316-320, runner.py:262). */
int tdgl_set_controller(tdgl_ctx *ctx, const tdgl_controller *c);
/* Host-only: np.mean(values[-window:]) in numpy's summation order (pairwise above 128 elements),
 * the arithmetic of the controller at solver.py:702-704; window == 0 averages the whole list like
 * Python's `vals[-0:]`.  No device work. */
double tdgl_host_mean_tail(const double *values, int64_t n, int32_t window);
/* Probe sites (device.probe_point_indices, solver.py:142, 691-694); n_probe may be 0. */
int tdgl_set_probes(tdgl_ctx *ctx, const int32_t *sites, int32_t n_probe);

/* ------------------------------------------------------------------ screening
 * options.include_screening (solver.py:304-314, 522-578, 654-688; kernel: tdgl/solver/screening.py).
 * Inside every step the induced vector potential A_ind[e] = sum_j K_site[j] area_j / |r_e - r_j|
 * is iterated to self-consistency with Polyak's heavy-ball update; the link variables use
 * A_applied + A_induced.  sites_xy [n,2] / edge_centers_xy [m,2] are the dimensionful positions
 * and site_areas [n] the areas already multiplied by the screening scale, exactly the arrays the
 * reference passes to its kernel (solver.py:307-313).  opts == NULL switches screening off. */
typedef struct {
    int32_t max_iterations;   /* SolverOptions.max_iterations_per_step */
    double tolerance;         /* SolverOptions.screening_tolerance      */
    double step_size;         /* SolverOptions.screening_step_size (alpha) */
    double step_drag;         /* SolverOptions.screening_step_drag (beta)  */
} tdgl_screening_options;
int tdgl_set_screening(tdgl_ctx *ctx, const tdgl_screening_options *opts, const double *sites_xy,
                       const double *edge_centers_xy, const double *site_areas);
/* The same in one-process-per-GPU mode (after tdgl_set_halo_plan): the 1/r sum runs over ALL sites,
 * so every rank passes the GLOBAL site coordinates / scaled areas (global order) and the global
 * ids of its owned sites, plus the centres of its local edges.  Per screening iteration the owned
 * site currents are scattered into a global array that is summed over ranks (all-reduce). */
int tdgl_set_screening_distributed(tdgl_ctx *ctx, const tdgl_screening_options *opts, const double *global_sites_xy,
                                   const double *global_site_areas, const int64_t *owned_global_ids,
                                   const double *edge_centers_xy);
/* A_induced [n_edges, 2] in reference edge order (seed / read-out; solver.py:738, 751). */
int tdgl_set_induced_vector_potential(tdgl_ctx *ctx, const double *A_induced);
int tdgl_get_induced_vector_potential(tdgl_ctx *ctx, double *A_induced);
/* One evaluation of the kernel of get_induced_vector_potential (solver.py:549-562;
 * tdgl/solver/screening.py:12-42, Mesh.get_quantity_on_site tdgl/finite_volume/mesh.py:203-243):
 * edge_current [n_edges] -> site average -> A_new [n_edges, 2] = sum_j <K>_j area_j / |r_e - r_j|.
 * No heavy-ball update, A_induced is not touched. */
int tdgl_induced_vector_potential(tdgl_ctx *ctx, const double *edge_current, double *A_new);

/* ------------------------------------------------------------------ the time loop */
/* Start a Runner stage (runner.py:294-297, 315-318): time = 0, stage step = 0.  Runner.dt
 * and the controller state are deliberately NOT reset. */
int tdgl_begin_stage(tdgl_ctx *ctx);

/* Runs TDGLSolver.update (solver.py:580-714) inside Runner._run_stage's loop
 * (runner.py:379-433) for up to max_steps iterations, stopping after the iteration in which
 * time >= end_time holds (the reference tests this AFTER stepping and BEFORE advancing
 * time, so one step past end_time is taken).
 *   out_dt[k]            dt used by iteration k                          (max_steps)
 *   out_mu_probe         mu at the probe sites        [max_steps, n_probe] (may be NULL)
 *   out_theta_probe      arg(psi) at the probe sites  [max_steps, n_probe] (may be NULL)
 *   out_pcg_iters[k]     Poisson iterations used                          (may be NULL)
 *   out_screening_iters[k]  screening iterations of step k (running_state
 *                        "screening_iterations", solver.py:695-696)       (may be NULL)
 *   steps_done           iterations executed
 *   reached_end          1 if the loop ended because time >= end_time
 * Returns TDGL_ERR_PSI_RETRIES with the reference's message when the psi update fails.
 * With a direct mu solve (tdgl_poisson_build_dense_inverse / _build_substructure), static link variables,
 * and no screening (the tables of tdgl_set_mu_boundary_table / tdgl_set_epsilon_table are fine: the device
 * evaluates them at the time of every attempt) the loop runs AHEAD of the host: the retry decision (solver.py:475-485), the
 * adaptive-dt controller (:698-707) and this loop's bookkeeping execute on the device and the host
 * synchronises once per batch of up to 64 attempts -- outputs, errors and the state left behind are
 * bit-identical to the one-synchronisation-per-step loop (environment TDGL_NO_RUN_AHEAD=1 forces that one). */
int tdgl_run(tdgl_ctx *ctx, int64_t max_steps, double end_time, double *out_dt,
             double *out_mu_probe, double *out_theta_probe, int32_t *out_pcg_iters,
             int64_t *steps_done, int32_t *reached_end, int32_t *out_screening_iters);

/* Work counters of tdgl_run since the last reset: out6 = {steps accepted, psi updates that failed and
 * were repeated with a smaller dt (solver.py:475-485), PCG iterations, host synchronisations,
 * nanoseconds the host was blocked in them, nanoseconds inside tdgl_run}. */
int tdgl_get_step_stats(tdgl_ctx *ctx, int64_t *out6, int32_t reset);
/* In-loop guard of the direct mu solves (explicit inverses are otherwise checked once, at set-up): once per
 * batch of queued attempts (one synchronisation per step: every 64th step) the residual of an accepted
 * step's solve, ||b - A mu|| / ||b|| with the resident level-0 matrix, is measured -- two small launches.
 * relres_max = the largest value seen, checks = steps checked, fell_back = 1 when a check exceeded the
 * limit (default 1e-9; the factors deliver 1e-14): the factors were released and every later step uses
 * AMG-PCG.  tdgl_set_direct_guard changes the limit (tests). */
int tdgl_get_direct_stats(tdgl_ctx *ctx, double *relres_max, int64_t *checks, int32_t *fell_back);
int tdgl_set_direct_guard(tdgl_ctx *ctx, double limit);

/* Loop state: stage step index i, Runner.time, Runner.dt (state["dt"] of the next
 * iteration), tentative_dt of the controller. */
int tdgl_get_loop_state(tdgl_ctx *ctx, int64_t *step, double *time, double *runner_dt,
                        double *tentative_dt);

/* Overwrite the loop state (used by the host-side TDGLSolver.update() compatibility shim,
 * whose caller owns step/time like the reference's Runner does, runner.py:381-384). */
int tdgl_set_loop_state(tdgl_ctx *ctx, int64_t step, double time, double runner_dt);

/* The adaptive-dt controller's state (solver.py:316-320, 698-707): tentative_dt and the persistent
 * list d_psi_sq_vals (the reference keeps it for the whole life of the solver; only the last
 * `adaptive_window` entries are ever read).  get: the newest min(capacity, n) entries, oldest first,
 * *n_history = entries held.  set: replaces both.  With tdgl_set_state / tdgl_set_loop_state this
 * restarts a run from a recorded point (checkpoint/restart; bench.py's parity replay). */
int tdgl_get_controller_state(tdgl_ctx *ctx, double *tentative_dt, double *history, int64_t capacity,
                              int64_t *n_history);
int tdgl_set_controller_state(tdgl_ctx *ctx, double tentative_dt, const double *history, int64_t n_history);

/* Current fields in reference ordering; any pointer may be NULL.  supercurrent and
 * normal_current are [n_edges] (operators.py:385-394, solver.py:519). */
int tdgl_get_state(tdgl_ctx *ctx, double *psi, double *mu, double *supercurrent,
                   double *normal_current);

/* ------------------------------------------------------------------ single operators
 * (parity-test entry points; each mirrors one reference call, host buffers in/out) */
/* psi_laplacian @ psi (operators.py:120-185 via :333-339). */
int tdgl_apply_psi_laplacian(tdgl_ctx *ctx, const double *psi, double *out);
/* MeshOperators.get_supercurrent (operators.py:385-394). */
int tdgl_supercurrent(tdgl_ctx *ctx, const double *psi, double *out);
/* TDGLSolver.solve_for_psi_squared (solver.py:383-439); *ok = 0 where the reference
 * returns None.  Uses the context's epsilon, u, gamma and link variables. */
int tdgl_psi_update(tdgl_ctx *ctx, const double *psi, const double *mu, double dt,
                    double *psi_out, double *abs_sq_out, int32_t *ok);
/* The right-hand side of the mu equation, divergence @ J_s - mu_boundary_laplacian @
 * mu_boundary (solver.py:508-510), for a given psi. */
int tdgl_poisson_rhs(tdgl_ctx *ctx, const double *psi, double *rhs);
/* mu_laplacian_lu(rhs) (solver.py:516) up to the additive constant: returns the
 * zero-mean solution.  mu_inout holds the initial guess on entry. */
int tdgl_poisson_solve(tdgl_ctx *ctx, const double *rhs, double *mu_inout, int32_t *iters,
                       double *relres);
/* The remaining MeshOperators attributes as operator applications (operators.py:282-345):
 *   psi_gradient @ psi                      [n_edges] complex   (operators.py:87-117, 340-344)
 *   divergence @ edge_field                 [n_sites]           (operators.py:59-84, 288)
 *   mu_laplacian @ mu                       [n_sites]           (operators.py:120-185, 285)
 *   mu_boundary_laplacian @ mu_boundary     [n_sites]           (operators.py:188-230, 286)
 * so that code written against the reference seam (`ops.divergence @ J`) keeps working. */
int tdgl_apply_psi_gradient(tdgl_ctx *ctx, const double *psi, double *out);
int tdgl_apply_divergence(tdgl_ctx *ctx, const double *edge_field, double *out);
int tdgl_apply_mu_laplacian(tdgl_ctx *ctx, const double *mu, double *out);
int tdgl_apply_mu_boundary_laplacian(tdgl_ctx *ctx, const double *mu_boundary, double *out);
/* -(mu_gradient @ mu) (solver.py:519, operators.py:287). */
int tdgl_normal_current(tdgl_ctx *ctx, const double *mu, double *out);
/* One application of the AMG V-cycle preconditioner z = M^-1 r on level-0 vectors given in
 * REFERENCE site order (for cross-checks against the host restatement of the cycle). */
int tdgl_vcycle(tdgl_ctx *ctx, const double *r, double *z);
/* The dot-product pass of the projection guess (k_multi_dot) on caller-supplied vectors [k, n] and b [n]
 * (k <= 16): out_pairs[2 a], out_pairs[2 a + 1] = (hi, lo) double-double sums of array a in {0: b . b,
 * 1: sum b, 2 + j: y_j . b, 18 + j: y_newest . y_j (newest = -1: zeros)}; 2 * 34 doubles. */
int tdgl_guess_dots(tdgl_ctx *ctx, int32_t k, int64_t n, const double *vectors, const double *b, int32_t newest,
                    double *out_pairs);

/* ------------------------------------------------------------------ measurement */
/* Average duration (ms) of `reps` back-to-back launches of one kernel on the context's
 * stream, timed with HIP events.  kernel: 0 = psi-Laplacian SpMV (K1), 1 = fused
 * psi-Laplacian + rhs, 2 = pointwise psi update, 3 = edge currents, 4 = level-0 Poisson
 * SpMV, 5 = level-0 V-cycle, 6 = copy (device memcpy ceiling, bytes = 16 * n_sites r+w),
 * 7 = induced vector potential (screening; n_edges * n_sites pairs, ~12 fp64 flops each),
 * 8 / 9 = two trivial dependent kernels across two streams (event record + wait each way) / in one
 * stream: their difference is the price of the two cross-stream dependencies an overlapped halo
 * exchange pays;
 * 10-15 = 100 device-wide barriers inside ONE launch (avg_ms is per launch): cooperative-groups
 * grid.sync() with 32 / 256 workgroups (10 / 11; the kernel syncs twice per round, i.e. 200), a
 * monotonic counter with agent-scope fences with 32 / 256 workgroups (12 / 13), with the 32 workgroups
 * of one XCD (14), the same with L2-local fences (15: NOT coherent, counts its stale reads);
 * tdgl_last_error then reports stale reads and the XCC_ID of workgroups 0-15;
 * 16-20 = HBM-cold streaming over a 1 GiB buffer: grid-stride read-only sum (16; 18 with 8 loads per lane
 * in flight, 19 with 16 workgroups per CU, 20 one workgroup per 512 entries), copy 512 MiB -> 512 MiB (17);
 * 21 = one workgroup of 1024 threads reading an L2-resident 2 MiB buffer 50 times (bytes one CU gets).
 * tools/bench_barrier.py prints all of them. */
int tdgl_time_kernel(tdgl_ctx *ctx, int32_t kernel, int32_t reps, double *avg_ms);
/* Enable/disable HIP-event timing of the fused psi-Laplacian kernel inside tdgl_run, and
 * read back the accumulated launches / milliseconds. */
int tdgl_profile_enable(tdgl_ctx *ctx, int32_t on);
int tdgl_profile_read(tdgl_ctx *ctx, int64_t *launches, double *total_ms);
/* The same for the kernel that dominates the run time, the CG's fused direction update + A p
 * (k_sell_axp): every 8th launch after tdgl_profile_enable is timed, up to 64 samples (single GPU). */
int tdgl_profile_read_pcg(tdgl_ctx *ctx, int64_t *launches, double *total_ms);
/* ... and for the launch sequence of a direct mu solve: one event pair per batch of the run-ahead loop brackets the
 * whole solve (k_sub_down ... k_sub_up / k_dense_sym_tiles + _finish).  Counted apart from the samples above, so
 * that a window in which the time loop changed the solver (tdgl_direct_switching) reports each kind by itself. */
int tdgl_profile_read_direct(tdgl_ctx *ctx, int64_t *solves, double *total_ms);
/* Mean reading of an event pair with nothing recorded in between (the markers' own cost, which
 * the in-run durations above contain and a profiler's dispatch durations do not). */
int tdgl_profile_event_overhead(tdgl_ctx *ctx, int32_t reps, double *avg_ms);

#ifdef __cplusplus
}
#endif
#endif /* TDGL_HIP_H */
