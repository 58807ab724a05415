// Internal declarations of libtdgl_hip (gfx950 only).  See include/tdgl_hip.h for the ABI
// and DESIGN.md for the data layout.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "tdgl_hip.h"

namespace tdgl {

typedef _Float16 half_t;        // IEEE binary16 storage (level-0 V-cycle operators)
constexpr int WAVE = 64;        // CDNA wavefront
constexpr int BLOCK = 256;      // 4 waves per workgroup
constexpr int XCDS = 8;         // MI355X accelerator complex dies (one L2 each)
constexpr int REDUCE_BLOCK = 1024;
constexpr int PROBE_RING_STEPS = 1024;  // steps of probe read-outs kept on the device between flushes

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------- device buffers
template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    void take(DevBuf &other) {  // (this buffer's memory is freed, the other's becomes this one's)
        release();
        p = other.p;
        n = other.n;
        other.p = nullptr;
        other.n = 0;
    }
    hipError_t alloc(size_t count, bool zero = true) {
        release();
        n = count;
        if (count == 0) return hipSuccess;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            p = nullptr;
            n = 0;
            return e;
        }
        if (zero) e = hipMemset(p, 0, count * sizeof(T));
        return e;
    }
    hipError_t upload(const std::vector<T> &h) {
        hipError_t e = alloc(h.size(), false);
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// SELL-C-sigma with C = 64 (one slice per wavefront, sigma = 1: row order is kept, the
// site permutation already groups rows of equal degree).  Entry k of row r of slice s is
// stored at (slice_off[s] + k) * 64 + (r % 64): a wavefront's loads are 64 consecutive
// elements.  Padding entries point at the row itself with value 0.
struct SellPattern {
    int64_t n_rows = 0, n_pad = 0;
    int32_t n_slices = 0;
    int64_t n_slots = 0;  // slice_off[n_slices] * 64
    DevBuf<int32_t> slice_off;
    DevBuf<int32_t> cols;
    // Square operators in RCM order: column - row fits 16 bits (the bandwidth is ~2 sqrt(n)), which
    // halves the index stream of the kernels that run at the HBM ceiling.  Built only on request
    // and only when every delta fits; kernels fall back to `cols` otherwise.
    DevBuf<int16_t> cols16;
    bool use16 = false;
    // Rectangular operators (the level-0 prolongation): column - slice_base[slice] as 16 unsigned
    // bits, when the columns of every 64-row slice span < 65536 (coarse numbering follows the fine one).
    DevBuf<uint16_t> offs16;
    DevBuf<int32_t> slice_base;
    bool use_off16 = false;
};

struct SellF64 {
    SellPattern pat;
    DevBuf<double> vals;
};

struct Csr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int32_t> indptr, indices;
    DevBuf<double> data;
    DevBuf<float> data32;  // fp32 copy of the values (mixed-precision preconditioner), on demand
};

// Level 0 (rows ~ 10^6, 7 entries each) streams through SELL-64, one lane per row.  Coarse
// levels and the restriction have few, longer rows: one lane per row would be a single long
// dependent gather chain per wave, so they use CSR with several lanes per row instead.
struct AmgLevel {
    int64_t n = 0, n_pad = 0, n_coarse = 0;
    int64_t n_cols = 0;              // entries of this level's right-hand side / iterate vectors (>= n: ghost entries)
    bool explicit_only = false;      // a distributed level 1: no A / P / R, only M, Wup, Vneg
    double rho = 2.0;
    SellF64 A;                       // level 0
    Csr Ac;                          // levels >= 1
    DevBuf<double> dinv;
    SellF64 P;                       // prolongation n x n_coarse (level 0)
    Csr Pc, R;                       // prolongation (levels >= 1); restriction n_coarse x n
    DevBuf<double> xa, xb, d, b, r;  // level vectors (level 0 borrows b from PCG)
    // fp32 copies of the level-0 V-cycle operators / iterates (mixed-precision preconditioner)
    DevBuf<float> A32, P32, dinv32, x32a, x32b;
    DevBuf<half_t> A16, P16;         // the same operators in binary16 (popt.precond_fp32 == 2)
    // optional pre-multiplied operators of a coarse level (tdgl_poisson_set_fused_level)
    bool fused = false;
    Csr RA;                          // R A   [n_coarse x n]
    Csr AP;                          // A P   [n x n_coarse]
    DevBuf<double> APp;              // P on the pattern of A P
    DevBuf<float> APp32;
    // collapsed coarse chain (tdgl_poisson_set_collapsed_level): M = R (I - A S) [n_coarse x n]
    Csr M;
    // ... and the way up as explicit operators (tdgl_poisson_set_collapsed_up): e = Wup b - Vneg e_next,
    // Wup = T_x S + T_b [n x n], Vneg = -T_x P [n x n_coarse]
    Csr Wup, Vneg;
    // compact forms for k_mid_up (when they fit): W columns = up_wbase[row] + up_woff, V columns 16-bit
    DevBuf<int32_t> up_wbase;
    DevBuf<uint16_t> up_woff, up_vidx;
    DevBuf<half_t> up_w16, up_v16;   // binary16 values (popt.precond_fp32 >= 2)
};

// scalars of the PCG recurrence, resident on the device
// (S_CONV_IT: first frozen iteration or -1; S_IT: iterations launched so far in this solve; S_TOL2:
// rtol^2 ||b||^2 -- kept on the device so that the iteration kernels take no per-iteration
// arguments and a pair of iterations can be replayed as a hipGraph)
// (S_CG_*: r.z and alpha of the previous iteration for the single-reduction CG, ping-pong by parity)
// (S_RR0: ||r_0||^2 of the current solve, recorded by the first update: the quality of the initial guess)
// (S_ALPHA: alpha of the last CG update, read by the next direction update's flexible beta)
enum Scal { S_RR = 0, S_BB, S_CONV_IT, S_IT, S_TOL2, S_CG_RZ0, S_CG_RZ1, S_CG_ALPHA0, S_CG_ALPHA1, S_RR0, S_ALPHA, S_COUNT = 12 };

// what a step reports back to the host at its synchronisation point
// arguments of the CG update x += alpha p, r -= alpha q (kernels.inc: xr_update_body)
struct XrArgs {
    const double *p, *q, *part_rz, *part_pq, *part_rr_in;
    double *scal, *x, *r, *part_rr_out;
    int64_t n;
    int npart;  // entries of every partial array in use (tdgl_ctx::npart)
    double *part_sx = nullptr;  // (optional) partials of sum x after the update: the zero-mean gauge needs no pass of its own
};

// up to 64 consecutive interior rows of ONE part of a substructure level, the work of one wavefront of the way up
// (kernels.inc: k_sub_up): where column `first row` of the part's -E_p^T block starts in the value pool, the part's
// size (= the block's row length), where the part's separator index list starts and how many entries it has,
// the chunk's first row and its rows
struct SubUpChunk {
    int64_t et;
    int32_t np, s0, cnt, row0, n_rows;
};

// the way down (kernels.inc: k_sub_down): a chunk of consecutive interior rows of ONE part -- where its first row of
// G_p starts in the pool, where b_p starts, the part's size, the chunk's first row and its rows -- and a row of the
// rest (separator rows, (G_p 1)^T rows) with up to four segments inline (nseg < 0: more, see the segment lists)
struct SubDownChunk {
    int64_t g;
    int32_t x0, ncols, row0, n_rows;
    int32_t ld, pad;  // (lane-per-row form: distance between consecutive rows of the part's G block in the pool; == ncols unless repacked)
};
// one part of a level while its fp64 pool is repacked into the padded fp32 pool (k_sub_repack)
struct SubRepack {
    int64_t old_g, new_g, old_et, new_et;
    int32_t np, sp, ld, pad;
};
struct SubDownRow {
    int32_t nseg, pad;
    int64_t val[4];
    int32_t x[4], len[4];
};

constexpr int GUESS_MAX = 16;  // maximal window of the projection guess (kernels.inc: GK)

// Run-ahead time loop of the direct solves (run.inc: run_ahead): the adaptive-dt controller and the loop's
// bookkeeping live on the device, so that the host can queue a batch of steps without waiting for each
// step's outcome.  `poisoned` stops everything behind a failed psi update (the host repeats that step the
// classic way with a smaller dt) or behind the step that reached end_time.
constexpr int RA_HIST_MAX = 128;   // adaptive_window up to here (numpy's pairwise sum has no recursion below 129 terms)
constexpr int RA_BATCH_MAX = 64;
struct StepCtl {
    double tentative_dt;   // dt the next STEP starts from (solver.py:316-320, 698-707)
    double attempt_dt;     // dt of the next attempt: tentative_dt, times the multiplier per failed attempt of the step
    double time;           // Runner.time (runner.py:433)
    double end_time;
    double dt_init, dt_cap, multiplier;
    long long stage_step;  // Runner step index of the current stage
    int adaptive, window, max_retries;
    int cur;               // which psi / L psi buffer holds psi^n
    int retries;           // failed attempts of the current step so far
    int poisoned;          // no further attempt may run (end reached, or the retry budget is spent)
    int live;              // the attempt being processed started un-poisoned (set by its psi-update kernel)
    int last_ok;           // ... and was accepted
    int reached_end, error;
    int n_done;            // attempts processed since the batch began
    int n_acc;             // ... of which accepted
    int hist_count;
    int pad;
    double hist[RA_HIST_MAX];  // the last min(window, count) values of max d|psi|^2, oldest first
    // field ramp evaluated inside the loop (k_ra_ramp_begin): A(t) = ramp(t) A_base
    double runner_dt;          // Runner.dt: dt of the previous accepted step (the dA/dt of this one divides by it)
    double ramp_tmin, ramp_tmax, ramp_initial, ramp_final;
    double link_scale, link_scale_prev;
    int has_dadt;              // a dynamic update has run (update_link_scale's early return needs one)
    int ramp_do;               // this attempt is the first of a step and the vector potential moves
};
struct StepRec {
    double dt, dmax;
    int ok, pad;
};

struct StepStatus {
    int32_t fail_flag;          // psi update: discriminant < 0 or non-finite somewhere
    int32_t pad;
    unsigned long long dmax_bits[8];  // max | |psi'|^2 - |psi|^2 | as ordered uint64 bits, 8 slots
    double scal[S_COUNT];
    // projection guess, double-double sums (hi at 2 a, lo at 2 a + 1): a = 0: b . b, 1: sum b, 2 + j: y_j . b,
    // 2 + GUESS_MAX + j: y_newest . y_j
    double gdot[2 * (2 * GUESS_MAX + 2)];
};

}  // namespace tdgl

struct IpcState;  // peer-mapped transport (ipc.inc)

struct tdgl_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    int64_t n = 0, m = 0, nb = 0, n_pad = 0, m_pad = 0;
    int64_t n_own = 0;     // rows this rank computes (sites [n_own, n) are ghost copies)
    int64_t n_global = 0;  // sites over all ranks
    double u = 5.79, gamma = 10.0;

    // ---- one-process-per-GPU mode (comm.inc) -----------------------------------------
    int rank = 0, world = 1;
    ncclComm_t comm = nullptr;
    tdgl_halo_fn cb_halo = nullptr;
    tdgl_allreduce_fn cb_allreduce = nullptr;
    void *cb_user = nullptr;
    std::vector<int32_t> nbr_ranks, send_ptr, recv_ptr;
    tdgl::DevBuf<int32_t> d_send_idx;
    tdgl::DevBuf<double> d_sendbuf, d_gstat;
    tdgl::DevBuf<float> d_red32;          // fp32 staging of the coarse right-hand side all-reduce
    double *h_sendbuf = nullptr, *h_recvbuf = nullptr;  // pinned (callback transport)
    int64_t h_buf_doubles = 0;
    bool fix_psi = true;
    // halo exchange overlapped with the ghost-free rows (comm.inc): second stream + events
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_pack = nullptr, ev_halo = nullptr;
    int int_tiles = 0;        // leading 256-row tiles whose rows have no ghost neighbour
    int64_t m_int = 0;        // leading edges (internal order) between two owned sites
    bool defer_mu_halo = false;  // pcg_solve leaves the exchange of mu's ghosts pending (run.inc)
    int overlap = 1;          // tdgl_set_comm_overlap: 0 off, 1 auto (by size), 2 always
    int64_t stat_halos = 0, stat_halo_bytes = 0, stat_allreduces = 0, stat_allreduce_bytes = 0;
    double *pend_v = nullptr; // exchange started by comm_halo_start, completed by comm_halo_wait
    int pend_width = 0;
    IpcState *ipc = nullptr;  // peer-mapped transport (tdgl_comm_init_ipc)
    // two distributed AMG levels (tdgl_set_deep_halo_plan): the exchange of r on the whole ghost zone
    bool deep = false;
    int64_t n_ext = 0;        // owned + every ghost layer (length of the PCG's level-0 vectors)
    int64_t l1_interior = 0;  // leading level-1 rows whose restriction reads owned fine entries only
    std::vector<int32_t> deep_nbrs, deep_send_ptr, deep_recv_ptr;
    tdgl::DevBuf<int32_t> d_deep_send_idx, d_deep_recv_idx;
    tdgl::DevBuf<double> d_deep_sendbuf, d_deep_recvbuf;
    double *h_deep_send = nullptr, *h_deep_recv = nullptr;  // pinned (callback transport)
    int64_t stat_deep_halos = 0;

    // permutations (host)
    std::vector<int32_t> perm, iperm;            // internal -> reference site, and inverse
    std::vector<int32_t> edge_perm, edge_iperm;  // internal -> reference edge, and inverse

    // ---- psi operators -------------------------------------------------------------
    tdgl::SellPattern lap_pat;            // off-diagonal pattern of the site graph
    tdgl::DevBuf<double2> lap_vals;       // (w_e / a_i) * U_ij, per slot
    tdgl::DevBuf<double> lap_slot_w;      // w_e / a_i per slot (static)
    tdgl::DevBuf<int32_t> lap_slot_edge;  // (internal edge << 1) | conj flag, -1 = padding
    tdgl::DevBuf<double> lap_diag;        // -sum_j w_ij / a_i
    tdgl::DevBuf<uint8_t> fixed_mask;     // 1 = identity row (terminal site with fix_psi)
    tdgl::DevBuf<double> area;            // a_i (0 on padding rows)
    // per edge (internal order, reference orientation)
    tdgl::DevBuf<int32_t> e0, e1;
    tdgl::DevBuf<double> e_inv_len, e_dirx, e_diry;
    tdgl::DevBuf<double2> e_U;
    tdgl::DevBuf<double> e_A, e_Aprev;    // link exponents now / at the previous step [2 * m_pad]
    tdgl::DevBuf<double> e_dAdt;          // dA/dt along the edges (time-dependent A), else unused
    bool has_dadt = false;
    tdgl::DevBuf<int32_t> link_block_changed, link_changed;  // allclose(new_A, old_A) test (k_dadt)
    // A(t) = scale(t) * A_base on the device (tdgl_set_link_exponents_base / _scale / _ramp)
    tdgl::DevBuf<double> e_Abase;
    bool have_base = false;
    bool ramp_on = false;
    double ramp_tmin = 0.0, ramp_tmax = 1.0, ramp_initial = 0.0, ramp_final = 1.0;
    double link_scale = 1.0, link_scale_prev = 1.0;  // scale of the current / previous A
    // Piecewise-linear time tables evaluated by tdgl_run itself before every step (no Python round
    // trip per step): terminal current densities -> mu_boundary (tdgl_set_mu_boundary_table) and a
    // time-dependent factor of epsilon (tdgl_set_epsilon_table)
    std::vector<double> tab_mu_t, tab_mu_dens;       // nodes; densities [n_groups x n_nodes]
    std::vector<int32_t> tab_mu_ptr, tab_mu_pos;     // groups of boundary-edge positions (CSR-like)
    std::vector<double> tab_mu_last, tab_mu_host;    // last densities applied; host copy of mu_boundary
    bool tab_mu_dev_synced = false;                  // b_mu on the device holds tab_mu_host at the positions no table covers
    // the same table on the device, for the run-ahead loop (k_ra_mu_table evaluates it at the device's own time)
    tdgl::DevBuf<double> d_tab_mu_t, d_tab_mu_dens;
    tdgl::DevBuf<int32_t> d_tab_mu_group;            // [nb]: group of a boundary position, -1 = not tabulated
    tdgl::DevBuf<int32_t> d_b_sites;                 // the sites that boundary edges touch, each once
    int32_t n_b_sites = 0;
    bool tab_mu_on_device = false;
    tdgl::DevBuf<double> d_tab_eps_t, d_tab_eps_f;   // the epsilon factor's table on the device (k_ra_eps_table)
    bool tab_eps_on_device = false;
    std::vector<double> tab_eps_t, tab_eps_f;
    tdgl::DevBuf<double> tab_eps0;                   // static part of epsilon, internal site order
    double tab_eps_last = NAN;
    // boundary term: c = mu_boundary_laplacian @ mu_boundary
    tdgl::DevBuf<int32_t> b_s0, b_s1;
    tdgl::DevBuf<double> b_c0, b_c1, b_mu;
    tdgl::DevBuf<double> cvec;            // boundary part
    tdgl::DevBuf<double> ceff;            // cvec + divergence(dA/dt): what the rhs kernel reads

    // ---- state ---------------------------------------------------------------------
    tdgl::DevBuf<double2> psi[2];
    int cur = 0;                          // psi[cur] is psi^n
    tdgl::DevBuf<double2> lap[2];         // lap[cur] = L_psi psi^n (cached between steps)
    bool lap_valid = false;
    tdgl::DevBuf<double> mu, eps, bvec;
    tdgl::DevBuf<double> js, jn;          // per edge, internal order
    bool currents_valid = false;
    bool have_links = false, have_state = false, have_eps = false;

    // ---- Poisson ---------------------------------------------------------------------
    std::vector<tdgl::AmgLevel *> levels;
    tdgl::Csr fusedR;                     // R0 (I - c A0 D0^-1): restriction of the pre-smoothed residual
    double fusedR_c = 0.0;                // the smoothing coefficient it was built for
    tdgl::DevBuf<float> fusedR32;         // its values in fp32
    tdgl::DevBuf<tdgl::half_t> fusedR16;  // ... in binary16
    bool level0_f16 = false;              // the level-0 V-cycle kernels read the binary16 copies
    double level0_absmax = 0.0;           // largest |entry| of A0, P0 and the fused restriction (binary16 range check)
    tdgl::DevBuf<uint16_t> fusedR_off16;  // its columns as 16-bit offsets from fusedR_base[row], when they fit
    tdgl::DevBuf<int32_t> fusedR_base;
    bool f32_ready = false;               // fp32 copies are current
    bool coarse32_ready = false;          // ... of the intermediate-level operators
    int64_t pcg_epoch = 0;                // bumped whenever an operator of the solve is replaced
    int64_t f32_fallbacks = 0;            // solves that had to be finished with the fp64 operators
    tdgl::DevBuf<double> coarse_pinv;
    int64_t n_coarsest = 0;
    // direct solve for small meshes (tdgl_poisson_set_dense_inverse): mu = G b, G = pinv(A) dense
    // [n, dense_ld] row major; replaces the PCG iteration while set (single GPU only)
    tdgl::DevBuf<double> denseG;
    int64_t dense_ld = 0;                 // > 0: in use
    int dense_tiles = 0;                  // > 0: symmetric packed storage, tiles per side (k_dense_sym_tiles)
    tdgl::DevBuf<double> dense_part;      // its per-tile contributions [dense_tiles][dense_tiles * DT]
    // substructured direct solve (tdgl_poisson_set_substructure): the Schur pseudo-inverse lives in
    // denseG / dense_part / dense_tiles (dense_n = n_sep)
    int64_t dense_n = 0;                  // order of the matrix in denseG (n, or n_sep)
    int32_t sub_parts = 0;                // > 0: in use
    int64_t sub_nI = 0, sub_nS = 0;
    tdgl::DevBuf<int32_t> sub_part_ptr, sub_seg_ptr, sub_seg_x, sub_seg_len, sub_sep_ptr, sub_sep_idx, sub_row_part;
    tdgl::DevBuf<int64_t> sub_seg_val, sub_e_off, sub_g_off;
    tdgl::DevBuf<double> sub_vals, sub_e, sub_u;
    tdgl::DevBuf<double> sub_w;           // [n + parts] result of the way down
    tdgl::DevBuf<double> sub_xs;          // [n_sep] separator solution before the mean is removed
    tdgl::DevBuf<double> sub_upart;       // per-workgroup partials of u . x_S
    int sub_nfin = 0;                     // workgroups of k_dense_sym_finish
    // further levels (tdgl_poisson_set_substructure_inner, once or twice): the same construction on the previous level's
    // Schur complement -- the second level's "parts" are the fine separators of the super-blocks, its separator the top
    // separator T; a third level cuts T the same way.  The LAST level's separator is the one whose pseudo-inverse is the
    // matrix in denseG.  sub_n_inner == 0: one level.
    struct SubInner {
        int32_t parts = 0;
        int64_t nI = 0, nS = 0;           // fine-separator sites / |T|; nI + nS = sub_nS
        tdgl::DevBuf<int32_t> part_ptr, seg_ptr, seg_x, seg_len, sep_ptr, sep_idx, row_part;
        tdgl::DevBuf<int64_t> seg_val, e_off, g_off;
        tdgl::DevBuf<double> vals, e, u;
        tdgl::DevBuf<float> vals32;       // (sub_fp32: the pool in fp32, `vals` released)
        tdgl::DevBuf<tdgl::SubUpChunk> chunks;
        tdgl::DevBuf<tdgl::SubDownChunk> down_chunks;
        tdgl::DevBuf<tdgl::SubDownRow> down_rows;
        tdgl::DevBuf<double> w;           // [sub_nS + parts] way down of the second level
        tdgl::DevBuf<double> xt;          // [|T|] top separator solution
        tdgl::Csr coupling;               // (optional) S1_TS' [|T| x nI]: r_T = r_T - coupling y_q, see sub_coupling
    } sub_in[2];
    int sub_n_inner = 0;                  // inner levels set so far (the last one carries the dense top separator)
    int sub_outer_parts_pending = 0;      // parts of the first level while it waits for its inner levels
    tdgl::DevBuf<tdgl::SubUpChunk> sub_chunks;
    tdgl::DevBuf<tdgl::SubDownChunk> sub_down_chunks;
    tdgl::DevBuf<tdgl::SubDownRow> sub_down_rows;
    tdgl::DevBuf<double> sub_mean;        // [1] two levels: the mean of the solution, left by the second level's way up
    // (optional, tdgl_poisson_set_substructure_coupling) A_SI [sub_nS x sub_nI]: the separator right-hand side of the way
    // down as r_S = b_S - A_SI y_I, a sparse product behind the dense one, instead of the -E_p^T rows inside it
    tdgl::Csr sub_coupling;
    // The factors as the CG's PRECONDITIONER (tdgl_poisson_set_substructure_precond; 650k - 1.3M sites): value pools and
    // the top separator's tiles stored in fp32 (sub_vals32 / SubInner::vals32 / denseG32; the fp64 pools are released),
    // every multiply-add fp64; the context stays in reverse Cuthill-McKee order, the dissection order lives inside the
    // application (sub_map[i] = the context's index of the site at dissection position i).  One application contracts
    // the residual by ~1e-5 .. 1e-6 (the rounding of the stored entries), so the CG needs ONE iteration wherever the
    // projection guess is good to 1e-4 -- at half the bytes of the fp64 factors.  pcg_solve chooses between this and the
    // AMG V-cycle per solve, by predicted cost (iterations x measured time per application).
    bool sub_precond = false;
    bool sub_fp32 = false;
    int sub_up_R[3] = {1, 1, 1};          // per level: chunks of 64 rows a workgroup of the way up takes (a whole part where > 1)
    int sub_sym_lds[3] = {0, 0, 0};       // per level: rows of b_p k_sub_down_sym stages (0: whole blocks, k_sub_down_lanes)
    bool sub_lanes = false;               // the ways down run k_sub_down_lanes (chunk lists rebuilt for it)
    bool sub_ident[3] = {false, false, false};  // level k's separator rows are their identity segment alone (out = b)
    tdgl::DevBuf<float> sub_vals32, denseG32;
    tdgl::DevBuf<int32_t> sub_map;
    tdgl::DevBuf<double> sub_bp, sub_xp, sub_z;   // [n] gathered residual / solution in dissection order / z in the context's order
    int32_t pd_choice = 0;                // 0: by predicted cost, 1: always the factors, 2: never (AMG V-cycle)
    double pd_rate = 4.5;                 // decades per iteration observed with the factors as preconditioner (running mean)
    double pd_t_apply_us = 0.0, pd_t_vcycle_us = 0.0;  // measured at set-up: one application of either preconditioner
    int64_t pd_solves = 0, pd_iters = 0, pd_amg_solves = 0, pd_amg_iters = 0;  // solves / iterations by preconditioner since the last reset
    bool pd_last = false;                 // the last solve used the factors
    int64_t pd_handovers = 0;             // solves that began with the factors and were finished by the V-cycle
    // Rank-level nested dissection (one process per GPU; schur.inc, tdgl_poisson_schur_begin / _complement / _finish): the
    // resident factors are those of THIS RANK'S INTERIOR block A_II (sub_n_local sites in the local dissection order, positive
    // definite: plain inverse of the top separator, no gauge), Gamma = the interface between the ranks (schur_ng sites,
    // global numbering), schurS32 = pinv of the interface complement in symmetric fp32 tiles on every rank.  One
    // application = two local solves + ONE all-reduce of schur_ng doubles + a replicated dense product.
    int64_t sub_n_local = 0;               // > 0: length of the vector the first level works on (instead of n)
    bool sub_nonsingular = false;
    bool schur_pending = false, schur_on = false;
    int64_t schur_ng = 0, schur_ngo = 0;
    tdgl::DevBuf<int32_t> schur_owner_local;   // [ng] local index of the Gamma site if this rank owns it, else -1
    tdgl::DevBuf<int32_t> schur_go_local, schur_go_gid;  // [ngo] owned Gamma sites: local index, position in Gamma
    tdgl::Csr schur_GI, schur_IG;          // A_GI [ng x n_I], A_IG [n_I x ng] (columns / rows in the local dissection order)
    tdgl::DevBuf<double> schurS64;         // tiles of the pseudo-inverse (fp64 until the conversion)
    tdgl::DevBuf<float> schurS32;
    tdgl::DevBuf<double> schur_part;
    int schur_tiles = 0;
    tdgl::DevBuf<double> schur_t, schur_rg, schur_xg, schur_y, schur_v, schur_c;  // [ng] x 3, [n_I] x 3
    bool sub_need_coupling[3] = {false, false, false};  // a level was described without its -E^T rows and its coupling block is not there yet
    // Solver choice in the time loop (tdgl_direct_switching; meshes where BOTH a direct solve and the hierarchy are
    // resident and large enough for the choice to matter).  A direct solve costs the same whatever the state; AMG-PCG
    // started from the projection guess costs next to nothing once the state is stationary (a transport current
    // through a strip: 0 iterations) and more than the direct solve while vortices move.  So: direct by default; while
    // |psi|^2 has changed by less than DIRECT_PAUSE_DMAX per step for DIRECT_PAUSE_WINDOW steps the direct solve is
    // paused (dense_on() is false: classic loop, AMG-PCG); when the iterations' running mean exceeds
    // DIRECT_RESUME_ITERS it comes back, and the next pause has to wait twice as long.
    bool direct_switch_on = false, direct_paused = false;
    int64_t direct_switch_steps = 0;       // accepted steps since the last switch
    int64_t direct_pause_hold = 256;       // steps the direct solve runs at least before it may be paused again
    int64_t direct_switches = 0;
    double direct_recent_dmax[64] = {0};   // ring of the last accepted steps' max d|psi|^2
    int64_t direct_recent_n = 0;
    double direct_win_max[4] = {0, 0, 0, 0};  // maxima of the last four complete windows of 64 steps, oldest first
    double direct_pcg_ema = 0.0;
    bool sub_wait_inner = false;          // first level set without a Schur complement: not usable before the second is
    // run-ahead time loop (direct solves, static links): device-resident controller + per-step records
    tdgl::DevBuf<tdgl::StepCtl> d_ctl;
    tdgl::DevBuf<tdgl::StepRec> d_rec;
    tdgl::StepCtl *h_ctl = nullptr;       // pinned
    tdgl::StepRec *h_rec = nullptr;       // pinned [RA_BATCH_MAX]
    int ra_batch = 4;                     // attempts queued per synchronisation: doubles up to RA_BATCH_MAX
    bool run_ahead_disabled = false;      // TDGL_NO_RUN_AHEAD, read once when the context is created (tests, A/B runs)
    int ra_retries = 0;                   // failed attempts of the step in progress when the last batch ended ...
    double ra_attempt_dt = 0.0;           // ... and the dt its next attempt takes (the retry state outlives a batch)
    int64_t stat_ra_batches = 0, stat_ra_dead = 0;
    bool currents_deferred = false;       // J of the last accepted step ride in the next step's psi-update launch
    bool spec_currents = false;           // step driver: queue the edge currents right behind the dense solve,
    bool spec_currents_done = false;      // before the host has seen the step's status (run.inc)
    // collapsed coarse chain (tdgl_poisson_set_collapsed_tail): everything from level `tail_level`
    // down as explicit operators, built for the smoother settings (tail_nu, tail_smoother, tail_cheb_lo)
    int tail_level = -1;                  // -1: off
    int tail_mode = 0;                    // 0: e = B b (dense [n_t, n_t]);  1: y = G b, e = W b + V y
    int64_t tail_g_rows = 0;              // rows of G
    int64_t tail_v_cols = 0;              // columns of the dense V (leading entries of G b)
    int64_t tail_ldg = 0, tail_ldv = 0;   // leading dimensions of G / V (padded to 4 entries)
    tdgl::DevBuf<double> tailG, tailV;    // dense row-major (mode 0: tailG holds B)
    tdgl::DevBuf<float> tailG32, tailV32;
    tdgl::Csr tailW;
    tdgl::DevBuf<uint16_t> tailW_idx16;   // its column indices in 16 bits (the tail level has < 65536 rows)
    int tail_nu = 0, tail_smoother = 0;
    double tail_cheb_lo = 0.0;
    tdgl::DevBuf<double> pcg_r, pcg_p, pcg_q;
    tdgl::DevBuf<double> pcg_p2;          // second direction buffer (fused direction update, k_sell_axp)
    // Per-workgroup partials of the dot products: arrays of stride NB of which the first `npart` entries
    // are written and summed -- the number of 256-row tiles rounded up to a multiple of 8, at most NB
    // (NB itself in one-process-per-GPU mode, where the arrays are summed over ranks of different sizes)
    int npart = 1024;
    tdgl::DevBuf<double> part_pair[2];    // 2 x NB partials each: [r.z | ||r||^2], ping-pong
    tdgl::DevBuf<double> part_pq, part_tmp;  // NB per-workgroup partials each
    tdgl::DevBuf<double> scal;            // tdgl::Scal (numbers the host reads)
    tdgl::DevBuf<double> mu_prev, mu_prev2;  // mu^{n-1}, mu^{n-2} for the extrapolated initial guess
    double prev_dt = 0.0, prev_dt2 = 0.0;    // dt of the steps that produced mu / mu_prev (0: no history)
    tdgl_poisson_options popt{1e-10, 500, 2, 0, 1, 1, 0.1, 3, 1, 2, 0, 0};
    // projection guess (popt.extrapolate == 3): window of previous solutions x_j and their images
    // y_j = A x_j (= b_j - r_j with the final residual of the CG recurrence), oldest first;
    // g_G[i][j] = y_i . y_j in window order, kept on the host as double-double numbers (hi, lo)
    tdgl::DevBuf<double> g_x[tdgl::GUESS_MAX], g_y[tdgl::GUESS_MAX];
    int g_slot[tdgl::GUESS_MAX] = {0};
    int g_count = 0;
    bool g_row_pending = false;           // the newest vector's Gram row arrives with the next solve's first status block
    bool mu_first_saved = false;          // mu_prev holds mu^n of a solve that started without a basis
    double g_G[tdgl::GUESS_MAX][tdgl::GUESS_MAX][2] = {{{0}}};
    double g_rhs[tdgl::GUESS_MAX][2] = {{0}};
    double g_bb[2] = {0, 0};              // (b - mean) . (b - mean) of the right-hand side being solved
    tdgl::DevBuf<double> part_gdot;       // 2 (2 GUESS_MAX + 2) x NB partials, see StepStatus::gdot
    tdgl::DevBuf<double> d_gdot;          // their sums
    tdgl::DevBuf<double> part_gdot_rank;  // one process per GPU: every rank's totals (kernels.inc: k_guess_rank_totals)
    // in-loop guard of the direct mu solves (run.inc: direct_guard_*): ||b - A mu|| / ||b|| of an accepted step,
    // measured with the resident level-0 matrix once per run-ahead batch / every DIRECT_GUARD_EVERY classic steps
    double direct_relres_max = 0.0;
    int64_t direct_checks = 0;
    int32_t direct_fell_back = 0;         // a check exceeded direct_guard_limit: the factors were released, AMG-PCG took over
    int32_t direct_guard_countdown = 0;
    double direct_guard_limit = 1e-9;
    int32_t last_guess_vectors = 0;       // basis size the last guess was formed from
    double last_guess_relres = 0.0;
    int32_t last_pcg_iters = 0;
    double last_relres = 0.0;
    // how many iterations the first batch of a solve queues before the host looks (pcg_solve): predicted from the
    // guess's residual, which the host knows from the Gram data, and the running contraction per iteration
    double pcg_rate = 0.6;                // decades of ||r|| per iteration (running mean of the observed ones)
    int32_t pcg_predict_from_guess = 1;   // 0: the previous solve's count (TDGL_PCG_PREDICT=last)
    int64_t stat_pcg_launched = 0;        // iterations queued, frozen ones included
    int64_t stat_pcg_needed = 0;          // iterations up to convergence
    int64_t stat_pcg_extra_syncs = 0;     // host looks after the first batch of a solve

    // ---- step status / probes --------------------------------------------------------
    tdgl::DevBuf<double> psi_dmax_part;    // per-workgroup max d|psi|^2 of the last k_psi_update
    tdgl::DevBuf<int32_t> psi_fail_part;   // per-workgroup failure flags
    int psi_blocks = 0;                    // its grid
    bool psi_status_pending = false;       // not yet reduced into d_status
    tdgl::DevBuf<tdgl::StepStatus> d_status;
    tdgl::StepStatus *h_status = nullptr;  // pinned: d_status is copied here at every host synchronisation
    tdgl::StepStatus *status_dev = nullptr;  // what the kernels write (= d_status.p)
    std::vector<int32_t> probes;           // internal site ids
    tdgl::DevBuf<int32_t> d_probes;
    tdgl::DevBuf<double> d_probe_out;      // ring buffer [PROBE_RING_STEPS][2 * n_probe]: mu | theta per step
    double *h_probe_out = nullptr;         // pinned, same shape
    int probe_ring_count = 0;              // steps written since the last flush (run.inc: probe_ring_flush)

    // ---- controller / loop state (host; mirrors solver.py:316-320, runner.py:260-263) --
    tdgl_controller ctl{1e-6, 1e-1, 1, 10, 10, 0.25};
    double tentative_dt = 1e-6, dt_cap = 1e-1;
    std::vector<double> d_psi_sq_vals;
    double runner_dt = 1e-6, time = 0.0;
    int64_t stage_step = 0;
    // counters since the last reset (tdgl_get_step_stats): steps accepted, failed psi updates that were
    // repeated with a smaller dt, PCG iterations, host synchronisations inside tdgl_run
    int64_t stat_steps = 0, stat_psi_retries = 0, stat_pcg_iters = 0, stat_host_syncs = 0;
    int64_t stat_wait_ns = 0, stat_run_ns = 0;  // host time blocked in those synchronisations / inside tdgl_run

    // ---- screening (screening.inc) -------------------------------------------------------
    bool scr_enabled = false;
    tdgl_screening_options scr{1000, 1e-3, 0.1, 0.5};
    tdgl::DevBuf<double> scr_site_xyw;   // per site: x, y, scaled area  [3 * n_pad]
    tdgl::DevBuf<double> scr_edge_xy;    // per edge (internal order): x, y  [2 * m_pad]
    tdgl::DevBuf<double> scr_inv_2deg;   // 1 / (2 * number of incident edges)  [n_pad]
    tdgl::DevBuf<double> scr_Jsite;      // site-averaged K = J_s + J_n  [2 * n_pad]
    tdgl::DevBuf<double> scr_Aind, scr_vel;  // [2 * m_pad], internal edge order
    tdgl::DevBuf<double> scr_Anew;           // [scr_chunks][2 * m_pad] partial sums over site chunks
    int scr_chunks = 1, scr_tiles_per_chunk = 1;
    tdgl::DevBuf<double> abs_sq_old;     // |psi^n|^2 of the step's starting psi [n_pad]
    tdgl::DevBuf<unsigned long long> scr_err_bits;
    // one-process-per-GPU mode: the 1/r sum runs over ALL sites; every rank scatters the site
    // currents of its owned sites into a global array that is summed over ranks
    int64_t scr_n_sites = 0;             // sites the kernel sums over (n, or n_global)
    tdgl::DevBuf<int64_t> scr_owned_gid; // global id of each owned site
    tdgl::DevBuf<double> scr_Jglobal;    // [2 * n_global_pad]
    int32_t last_screening_iters = 0;

    // ---- measurement -------------------------------------------------------------------
    bool profile = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int64_t prof_launches = 0;
    double prof_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_pending;
    std::vector<hipEvent_t> prof_pool;     // created by tdgl_profile_enable, recycled by profile_drain: no
                                           // hipEventCreate inside a timed region
    // the same for the kernel that dominates the run time (the CG's fused A p), sampled: only the
    // first prof2_budget launches after tdgl_profile_enable are bracketed by events
    int64_t prof2_launches = 0, prof2_budget = 0, prof2_seen = 0;
    double prof2_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof2_pending;
    // ... and for the launch sequence of a direct mu solve (one pair per run-ahead batch; its own counters, so that a
    // window in which the loop changed the solver does not divide one kind of bytes by the other kind of samples)
    int64_t prof3_launches = 0;
    double prof3_ms = 0.0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof3_pending;
};

// The time loop's solver choice (run.inc: direct_policy) is a function of the run's recent history.  Whatever makes that
// history meaningless -- a new state, a new stage, new boundary values or a new epsilon handed in by the caller --
// returns it to its starting point: the direct solve, no windows, the short wait.  So a restart from a checkpoint
// (tdgl_set_state + tdgl_set_loop_state + tdgl_set_controller_state) takes the same path whatever came before it.
static inline void direct_policy_reset(tdgl_ctx *ctx) {
    ctx->direct_paused = false;
    ctx->direct_switch_steps = 0;
    ctx->direct_recent_n = 0;
    ctx->direct_pause_hold = 256;
    ctx->direct_pcg_ema = 0.0;
    for (double &w : ctx->direct_win_max) w = 0.0;
}

