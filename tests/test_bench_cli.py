"""`bench.py` as the driver calls it, end to end (GPU): the decomposed path with its transport race and self-test on
ranks that share the test box's one GPU, and the single-GPU line's contract fields."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*flags, timeout=600):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout,
                       cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]  # ONE JSON line on stdout, and nothing else
    return json.loads(lines[0])


def test_two_ranks_sharing_the_gpu_race_the_transports_and_report_it():
    """`--gpus 2 --share-devices` (what the driver runs with one GPU per rank, here oversubscribed): the peer-mapped
    transport passes its self-test on both ranks, RCCL refuses two ranks on one device and is recorded as out, the
    line says which transport ran, what a step communicated, and that the timing is a dry run."""
    d = _run("--gpus", "2", "--share-devices", "--workload", "250k", "--steps", "10", "--warmup", "5", "--preroll", "60",
             "--no-cpu-baseline", "--config5", "off")
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["scaling"] == "strong" and d["value"] > 0
    tr = d["transport"]
    assert tr["used"] == "ipc"
    by = {t["transport"]: t for t in tr["selftest"]}
    assert by["ipc"]["ok"] and by["ipc"]["checks_per_rank"] >= 5 and by["ipc"]["probe_steps_per_s"] > 0
    assert not by["rccl"]["ok"] and by["rccl"]["failures"]
    c = d["comm_per_step"]
    assert "levels 0 and 1 distributed" in c["decomposition"] and c["overlap"] is True
    its = d["pcg"]["mean_iterations"]
    assert c["halo_exchanges"] <= 1.3 * its + 3.5 and c["deep_ghost_entries"] > c["ghost_sites"]
    assert "DRY RUN" in d["config"]["parallelism"] and "peer-mapped" in d["config"]["parallelism"]
    # current conservation of the final state assembled from both ranks: solver tolerance, not O(1)
    assert d["conservation"]["relative"] < 1e-7 and d["conservation"]["max_abs_current"] > 1e-3


def test_under_the_drivers_launcher_all_ranks_together_print_one_line():
    """The driver's literal multi-GPU command -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W` -- where EVERY rank's stdout is the launcher's: one
    JSON line from rank 0 and not a byte more from anybody (gloo announces every new process group on stdout, RCCL prints
    a banner: both are routed to stderr)."""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "5",
                        "--share-devices", "--workload", "60k", "--preroll", "45", "--no-cpu-baseline", "--config5", "off"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), p.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 5 and d["value"] > 0 and d["transport"]["used"] == "ipc"


def test_eight_ranks_sharing_the_gpu_run_the_drivers_command_shape():
    """The driver's multi-GPU command with the rank count of the scaling run (`--gpus 8`), rehearsed on the one GPU there
    is: the peer-mapped transport passes its self-test on all eight ranks, RCCL's refusal of ranks that share a device is
    RECORDED in `transport.selftest` (not discovered on the 8-GPU box), the decomposed path is followed by the oracle,
    and the final state conserves current across all seven cuts.  (The full-size rehearsal -- 1M sites cut 8 ways with
    config 5 attached, 233 s of wall time -- is profiles/BENCH_r08c_1M_8ranks_sharing_one_gpu_DRY_RUN.json.)"""
    d = _run("--gpus", "8", "--share-devices", "--workload", "60k", "--steps", "10", "--warmup", "5", "--preroll", "45",
             "--config5", "off", timeout=900)
    assert d["n_gpus"] == 8 and d["value"] > 0 and "DRY RUN" in d["config"]["parallelism"]
    by = {t["transport"]: t for t in d["transport"]["selftest"]}
    assert d["transport"]["used"] == "ipc" and by["ipc"]["ok"] and by["ipc"]["checks_per_rank"] >= 5
    assert not by["rccl"]["ok"] and len(by["rccl"]["failures"]) == 8 and all("RCCL" in f["error"] for f in by["rccl"]["failures"])
    assert d["parity_vs_oracle"]["ok"] and "8 rank(s)" in d["parity_vs_oracle"]["source"]
    assert d["conservation"]["relative"] < 1e-7


def test_decomposed_run_is_followed_by_the_oracle_from_the_same_state():
    """With the CPU leg on, a decomposed run's K timed steps are taken by the oracle too, from the state assembled from
    both ranks before the clock started: `parity_vs_oracle` of the decomposed path (no `cpu_baseline`: that is a
    single-GPU figure)."""
    d = _run("--gpus", "2", "--share-devices", "--workload", "60k", "--steps", "10", "--warmup", "5", "--preroll", "45",
             "--config5", "off")
    par = d["parity_vs_oracle"]
    assert par["ok"] and par["steps"] == 10 and "2 rank(s)" in par["source"], par
    assert max(par[k] for k in ("dt", "abs_sq_psi", "mu_zero_mean", "J_s", "J_n")) <= par["tolerance"] == 1e-8
    assert "cpu_baseline" not in d and d["transport"]["used"] == "ipc"


def test_when_no_device_transport_works_the_host_transport_still_yields_a_line():
    d = _run("--gpus", "2", "--share-devices", "--workload", "60k", "--steps", "5", "--warmup", "2", "--preroll", "45",
             "--no-cpu-baseline", "--config5", "off", "--debug-fail", "ipc,rccl")
    tr = d["transport"]
    assert tr["used"] == "gloo" and [t["ok"] for t in tr["selftest"]] == [False, False, True]
    assert d["value"] > 0 and "host-callback" in d["config"]["parallelism"]


def test_single_gpu_line_carries_the_contract_fields():
    d = _run("--workload", "60k", "--steps", "20", "--warmup", "5", "--preroll", "40", "--cpu-seconds", "2", "--vortex-window", "off")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["dtype"] == "f64" and d["vs_baseline"] is None and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["frac_net"] >= r["frac"] and r["frac_burst"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert d["parity_vs_oracle"]["ok"] and d["parity_vs_oracle"]["tolerance"] == 1e-8
    # 59k sites: the two-level direct solve in the run-ahead loop -- its own roofline object, from its own event pairs
    rd = d["roofline_direct"]
    assert rd is not None and rd["samples"] >= 1 and 0.0 < rd["frac"] <= 1.0 and d["roofline_pcg"] is None
    assert 0.0 < r["frac"] <= 1.0


def test_a_window_in_which_the_loop_has_paused_the_direct_solve_reports_no_direct_roofline():
    """The 251k-site strip relaxes into a stationary state; the time loop then pauses the direct mu solve
    (`tdgl_direct_switching`) and AMG-PCG from the projection guess takes over.  A timed window in that state holds no
    direct solve: `roofline_direct` must be null (round 5 divided the factors' bytes by event pairs that bracketed the
    CG's `A p` kernel: frac 2.9), and no roofline object of the line may claim more than the peak."""
    d = _run("--workload", "strip250k", "--steps", "20", "--warmup", "5", "--preroll", "1500", "--no-cpu-baseline",
             "--vortex-window", "off")
    sw = d["pcg"].get("direct_switching") or d.get("direct_switching")
    assert sw is not None and sw["paused"] and sw["switches"] >= 1, d["pcg"]
    assert d["roofline_direct"] is None
    for key in ("roofline", "roofline_pcg"):
        if d.get(key) is not None:
            assert 0.0 < d[key]["frac"] <= 1.0, (key, d[key])
