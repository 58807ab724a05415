"""Where does an 8-rank callback-transport run spend its time?  (rank 0 prints)"""
import os, sys, time, socket
sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
import numpy as np
import torch.multiprocessing as mp


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, "tests"); sys.path.insert(0, "py-tdgl_amd"); sys.path.insert(0, ".")
    from test_hip_distributed import _problem
    t0 = time.time()
    mesh, terms, A, mu_b, opts, probes, psi0 = _problem(120, 60)
    from tdgl_amd import _lib
    _lib.load()
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t1 = time.time()
    from tdgl_amd.distributed import DistributedTDGL
    run = DistributedTDGL(mesh, opts, A, 1.0, rank=rank, world=world, terminal_info=terms, mu_boundary=mu_b,
                          probe_points=probes, transport="gloo", device_id=0, overlap="auto")
    t2 = time.time()
    run.set_state(psi0, np.zeros(len(mesh.sites)))
    run.begin_stage()
    for chunk in (1, 4, 15):
        ta = time.time()
        res = run.run(chunk)
        tb = time.time()
        if rank == 0:
            print(f"{chunk} steps: {tb - ta:.2f} s, iters {res['pcg_iters'].tolist()}, comm {run.ctx.comm_stats(reset=True)}", flush=True)
    if rank == 0:
        print(f"import+mesh {t1 - t0:.1f} s, setup {t2 - t1:.1f} s, levels {run.ctx.hierarchy.sizes}", flush=True)
    run.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    mp.spawn(worker, args=(world, port), nprocs=world, join=True)
