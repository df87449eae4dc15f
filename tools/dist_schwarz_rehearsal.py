"""Iteration counts of the partitioned solves (Jacobi single-reduction PCG vs additive Schwarz with local AMG V-cycles)
for W ranks.  A 1-GPU box cannot host several RCCL ranks, so the ranks share GPU 0 and use gloo: iteration counts and
correctness are those of the 8-GPU run, times are not.   python tools/dist_schwarz_rehearsal.py [W ...]"""
import os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def worker(rank, world, port, nx, ny):
    import torch, torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pyfem_gpu_testflight_amd import pyfem
    from pyfem_gpu_testflight_amd.dist import PartitionedElasticity
    c = pyfem.ProblemCreator(nx + 1, ny + 1)
    conn, X, dof_fixed, force = c.create_linear_elasticity_problem()
    q = pyfem.QuadratureBilinear2D(); b = pyfem.BasisBilinear2D(q)
    rho = 0.3 + 0.7 * np.random.RandomState(1).rand(X.shape[0])
    pm = PartitionedElasticity(X, conn, dof_fixed, None, force, q, b, p=3.0)
    out = {}
    for pc in ("amg2", "amg", "jacobi"):
        torch.cuda.synchronize(); dist.barrier(); t0 = time.perf_counter()
        x = pm.solve(rho=rho, rtol=1e-8, precond=pc, check_every=(100 if pc == "jacobi" else 1))
        torch.cuda.synchronize(); dist.barrier()
        out[pc] = (pm.last_iters, time.perf_counter() - t0)
    if rank == 0:
        print(f"W={world} {nx}x{ny}: interface dofs {pm.iface.G}; " + "; ".join(f"{k}: {v[0]} iterations, {v[1]:.2f} s (gloo, shared GPU)" for k, v in out.items()), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    nx, ny = int(os.environ.get("NX", 512)), int(os.environ.get("NY", 256))
    for w in [int(a) for a in sys.argv[1:]] or [1, 2, 4]:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mp.spawn(worker, args=(w, port, nx, ny), nprocs=w, join=True)
