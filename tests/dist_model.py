"""Host model of the DISTRIBUTED step (test infrastructure; NumPy + torch.distributed/gloo).

It executes, with NumPy arithmetic, exactly the algorithm the HIP library runs in one-process-
per-GPU mode on the plan produced by `tdgl_amd.partition`:

* owned rows only, ghost copies refreshed by neighbour exchange where the library does it;
* PCG with global dot products (all-reduce), level 0 of the AMG hierarchy distributed, coarser
  levels replicated, the restricted residual summed over ranks.

`tests/test_distributed_cpu.py` runs it with world_size 2 and 3 under gloo and compares against
the single-domain oracle, which validates the partitioner, the halo send/receive lists and the
slicing of the hierarchy -- the data the GPU path consumes.
"""

import numpy as np
import torch
import torch.distributed as dist

from tdgl_amd.amg import smoother_coefficients, vcycle_host


def halo_exchange(lp, vec):
    """Refresh the ghost entries of the local vector ``vec`` (float64 or complex128) in place."""
    width = 2 if np.iscomplexobj(vec) else 1
    flat = vec.view(np.float64).reshape(len(vec), width)
    reqs, recv_bufs = [], {}
    for nb in lp.neighbors:
        a, b = lp.recv_range[nb]
        recv_bufs[nb] = torch.empty((b - a, width), dtype=torch.float64)
        reqs.append(dist.irecv(recv_bufs[nb], src=nb))
    for nb in lp.neighbors:
        reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(flat[lp.send_idx[nb]])), dst=nb))
    for r in reqs:
        r.wait()
    for nb in lp.neighbors:
        a, b = lp.recv_range[nb]
        flat[a:b] = recv_bufs[nb].numpy()


def allreduce_sum(x):
    t = torch.from_numpy(np.atleast_1d(np.asarray(x, dtype=np.float64)).copy())
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy() if t.numel() > 1 else float(t.item())


def vcycle_dist(h, loc0, lp, b_own, nu=2, smoother="chebyshev", cheb_lo=0.1, nu_fine=1):
    """One V-cycle: level 0 distributed (`loc0` from partition.local_hierarchy_level0), levels
    >= 1 replicated.  Exchanges happen exactly where csrc/poisson.inc places them."""
    A, dinv_loc, P, R = loc0["A"], loc0["dinv"], loc0["P"], loc0["R"]
    n_own = lp.n_own
    dinv = dinv_loc[:n_own]
    if nu_fine == 1:
        # degree-1 smoothing on level 0: ONE exchange (ghosts of the right-hand side) per cycle
        _, c2 = smoother_coefficients(loc0["rho"], 1, smoother, cheb_lo)
        bl = np.zeros(lp.n_loc)
        bl[:n_own] = b_own
        halo_exchange(lp, bl)
        r = b_own - c2[0] * (A @ (dinv_loc * bl))
        bc = allreduce_sum(R @ r)
        ec = vcycle_host(h, bc, nu, smoother, cheb_lo, lvl=1)
        x = c2[0] * dinv_loc * bl + P @ ec  # valid on ghost rows too
        return x[:n_own] + c2[0] * dinv * (b_own - A @ x)
    c1, c2 = smoother_coefficients(loc0["rho"], nu, smoother, cheb_lo)
    x = np.zeros(lp.n_loc)
    d = c2[0] * dinv * b_own
    x[:n_own] = d
    for k in range(1, nu):
        halo_exchange(lp, x)
        d = c1[k] * d + c2[k] * dinv * (b_own - A @ x)
        x[:n_own] += d
    halo_exchange(lp, x)
    r = b_own - A @ x
    bc = allreduce_sum(R @ r)  # partial restrictions summed over ranks
    ec = vcycle_host(h, bc, nu, smoother, cheb_lo, lvl=1) if len(h.levels) > 1 else None
    x += P @ ec  # ghost rows included: ghosts stay valid without an exchange
    for k in range(nu):
        if k > 0:
            halo_exchange(lp, x)
        d = c1[k] * d + c2[k] * dinv * (b_own - A @ x)
        x[:n_own] += d
    return x[:n_own].copy()


def pcg_dist(h, loc0, lp, b_own, x0_loc=None, rtol=1e-10, maxiter=200):
    """Distributed PCG for A mu = b; returns (mu_local incl. valid ghosts, iterations)."""
    A = loc0["A"]
    n_own, n_glob = lp.n_own, lp.n_global
    b = b_own - allreduce_sum(b_own.sum()) / n_glob
    x = np.zeros(lp.n_loc) if x0_loc is None else x0_loc.copy()
    r = b - A @ x
    bb = allreduce_sum(b @ b)
    rr = allreduce_sum(r @ r)
    if bb == 0:
        return np.zeros(lp.n_loc), 0
    p = np.zeros(lp.n_loc)
    rz_old, it = None, 0
    while rr > rtol * rtol * bb and it < maxiter:
        z = vcycle_dist(h, loc0, lp, r)
        rz = allreduce_sum(r @ z)
        p[:n_own] = z if rz_old is None else z + (rz / rz_old) * p[:n_own]
        halo_exchange(lp, p)
        q = A @ p
        alpha = rz / allreduce_sum(p[:n_own] @ q)
        x[:n_own] += alpha * p[:n_own]
        r -= alpha * q
        rr = allreduce_sum(r @ r)
        rz_old = rz
        it += 1
    x[:n_own] -= allreduce_sum(x[:n_own].sum()) / n_glob
    halo_exchange(lp, x)
    return x, it


def gather_global(lp, local_owned_values, n_global, dtype=np.float64):
    """Assemble a global site vector from every rank's owned values (all ranks get it)."""
    out = np.zeros(n_global, dtype=dtype)
    out[lp.local_to_global[: lp.n_own]] = local_owned_values
    flat = out.view(np.float64).copy()
    t = torch.from_numpy(flat)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.numpy().view(dtype)


# ---------------------------------------------------------------------------------------------------------------
# Two distributed AMG levels, one vector exchange per iteration (tdgl_amd.partition.DeepPlan; csrc/poisson.inc with
# tdgl_set_deep_halo_plan): the same sequence in NumPy.
def halo_exchange_deep(dp, vec):
    """Refresh ALL ghost entries ``vec[n_own:n_ext]`` (the residual's deep ghost zone) in place."""
    nbrs = sorted(set(dp.neighbors) | set(dp.send_idx))
    reqs, bufs = [], {}
    for nb in nbrs:
        k = len(dp.recv_idx.get(nb, ()))
        if k:
            bufs[nb] = torch.empty(k, dtype=torch.float64)
            reqs.append(dist.irecv(bufs[nb], src=nb))
    for nb in nbrs:
        idx = dp.send_idx.get(nb, ())
        if len(idx):
            reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(vec[idx])), dst=nb))
    for r in reqs:
        r.wait()
    for nb, t in bufs.items():
        vec[dp.recv_idx[nb]] = t.numpy()


def level2_cycle(h, plan, b2):
    """The replicated part of the collapsed chain: everything from level 2 down (host model)."""
    def level(k, bk):
        if k == plan["tail"]:
            if plan["mode"] == "dense":
                return plan["B"] @ bk
            y = plan["G"] @ bk
            return plan["W"] @ np.concatenate([bk, y]) + plan["V"] @ y[: plan["V"].shape[1]]
        W, V = plan["up"][k]
        return W @ bk + V @ level(k + 1, plan["mid"][k] @ bk)

    return level(2, b2)


def vcycle_deep(dp, h, plan, r_ext, counts=None):
    """z = M^-1 r on the owned rows AND the first ghost layer from r on the whole ghost zone (already exchanged):
    no further vector exchange, one sum over ranks (the partial level-2 right-hand sides)."""
    b1 = dp.F @ r_ext
    b2 = allreduce_sum(dp.M @ b1[: dp.l1_own])
    if counts is not None:
        counts["allreduce_values"] += len(b2)
    x1 = dp.W @ b1 + dp.V @ level2_cycle(h, plan, b2)
    x = dp.c * dp.dinv[: dp.n2] * r_ext[: dp.n2] + dp.P @ x1
    return x[: dp.n1] + dp.c * dp.dinv[: dp.n1] * (r_ext[: dp.n1] - dp.A @ x)


def pcg_deep(dp, lp, h, plan, b_own, x0_loc=None, rtol=1e-10, maxiter=200):
    """The single-reduction (Chronopoulos-Gear) PCG the library runs in this mode.  Returns (mu_local with valid
    first-layer ghosts, iterations, counts of what was communicated)."""
    n_own, n_glob = dp.n_own, lp.n_global
    A_own = dp.A[:n_own]
    counts = dict(deep_exchanges=0, thin_exchanges=0, allreduce_values=0)
    b = b_own - allreduce_sum(b_own.sum()) / n_glob
    x = np.zeros(dp.n1) if x0_loc is None else x0_loc.copy()
    xe = np.zeros(dp.n2)
    xe[: dp.n1] = x
    r = np.zeros(dp.n_ext)
    r[:n_own] = b - A_own @ xe
    bb = allreduce_sum(b @ b)
    rr = allreduce_sum(r[:n_own] @ r[:n_own])
    p, q = np.zeros(n_own), np.zeros(n_own)
    rz_old = alpha_old = None
    it = 0
    while rr > rtol * rtol * bb and it < maxiter:
        halo_exchange_deep(dp, r)
        counts["deep_exchanges"] += 1
        z = vcycle_deep(dp, h, plan, r, counts)            # valid on [0, n1)
        ze = np.zeros(dp.n2)
        ze[: dp.n1] = z
        w = A_own @ ze                                      # no exchange: z's first ghost layer was formed locally
        rz, zw = allreduce_sum(np.array([r[:n_own] @ z[:n_own], z[:n_own] @ w]))
        counts["allreduce_values"] += 2
        beta = 0.0 if rz_old is None else rz / rz_old
        alpha = rz / (zw - (0.0 if rz_old is None else beta * rz / alpha_old))
        p = z[:n_own] + beta * p
        q = w + beta * q
        x[:n_own] += alpha * p
        r[:n_own] -= alpha * q
        rr = allreduce_sum(r[:n_own] @ r[:n_own])
        rz_old, alpha_old = rz, alpha
        it += 1
    x[:n_own] -= allreduce_sum(x[:n_own].sum()) / n_glob
    halo_exchange(lp, x)
    counts["thin_exchanges"] += 1
    return x, it, counts


# ---------------------------------------------------------------- rank-level nested dissection (tdgl_amd/schur_dd.py)
def schur_setup_dist(lp, piece):
    """What `DistributedTDGL` does at set-up, with SciPy's LU standing in for the rank's local factors: the interface
    complement ``S = A_GG - sum_r A_GI_r A_II_r^-1 A_I_rG`` summed over the ranks (ONE sum of |Gamma|^2 values), its
    pseudo-inverse on every rank."""
    import scipy.sparse.linalg as spla

    from tdgl_amd.schur_dd import interface_pinv

    lu = spla.splu(piece.A_II.tocsc())
    touched = np.unique(piece.A_IG.indices)
    X = lu.solve(piece.A_IG[:, touched].toarray())
    S = piece.A_GG_owned.toarray()
    S[np.ix_(touched, touched)] -= piece.A_GI[touched] @ X
    t = torch.from_numpy(S)
    dist.all_reduce(t)
    return lu, interface_pinv(t.numpy())


def schur_apply_dist(lp, piece, lu, Spinv, r_own, counts=None):
    """One application ``z = M r`` on the owned rows: two local solves, ONE all-reduce of |Gamma| doubles, a replicated
    dense product (the sequence of csrc/poisson.inc: precond_schur_apply)."""
    r_loc = np.zeros(lp.n_loc)
    r_loc[: lp.n_own] = r_own
    y = lu.solve(r_loc[piece.interior])
    t = np.zeros(piece.n_gamma)
    t[piece.gamma_owned_gid] = r_loc[piece.gamma_owned_local]
    t -= piece.A_GI @ y
    t = allreduce_sum(t)
    if counts is not None:
        counts["allreduces"] = counts.get("allreduces", 0) + 1
        counts["allreduce_values"] = counts.get("allreduce_values", 0) + piece.n_gamma
    xg = Spinv @ t
    xi = y - lu.solve(piece.A_IG @ xg)
    z = np.zeros(lp.n_own)
    z[piece.interior] = xi  # (interior ids are owned ids: < n_own)
    z[piece.gamma_owned_local] = xg[piece.gamma_owned_gid]
    return z
