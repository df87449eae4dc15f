#!/usr/bin/env python
"""Headline benchmark: element-assemblies/sec + CG-iter/sec on the 4M-quad plane-stress mesh
(BASELINE.json metric), one process per GPU.

    python bench.py                                   # N=1, defaults finish in a few minutes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one device-resident `compute_jacobian()` pass over the whole mesh: RAMP density at the
quadrature points -> element stiffness rows -> scatter into the existing sparsity pattern (the
symbolic pattern is built once outside the timed region).  Inputs (mesh, density) are
resident in HBM before the timed region.  The CG leg runs a fixed number of Jacobi-PCG iterations
on the BC-applied matrix.  N>1 is STRONG scaling: the same 4M-quad mesh is split into N element
blocks (no data-path collective in assembly; one all-reduce per CG iteration).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling is ~6290


def algorithmic_bytes_assembly(nelems, nnodes, nnz, P=4, with_rho=False):
    """conn int32 + X + rho + K.data written once."""
    return 4 * P * nelems + 16 * nnodes + (8 * nnodes if with_rho else 0) + 8 * nnz


def algorithmic_bytes_spmv(nnz, nrows):
    """12 B per nonzero (value + int32 column) + 20 B per row."""
    return 12 * nnz + 20 * nrows


def cpu_baseline(n=1024, reps=3, cg_iters=40):
    """The oracle (numpy/scipy restatement of the reference's path) timed on this host: a bounded
    sample -- an n x n quad mesh of the same plane-stress workload."""
    from oracle import pyfem_oracle as orc  # checker/baseline only

    # BLAS threads = the CPUs this process may really use (affinity mask and cgroup quota), not the host's logical
    # CPU count: on a quota'd box one OpenBLAS thread per logical CPU only burns the quota and gets the process
    # throttled, which would understate the baseline.
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            usable = max(1, min(usable, int(int(quota) / int(period))))
    except Exception:
        pass
    threads = usable
    limiter = None
    try:
        from threadpoolctl import threadpool_limits

        limiter = threadpool_limits(limits=usable)
    except Exception:
        threads = os.cpu_count() or 1
    conn, X = orc.structured_mesh(n + 1, n + 1, "quad")
    dof_fixed, force = orc.elasticity_problem(n + 1, n + 1)
    conn_dof = orc.create_conn_dof(conn, 2)
    times = []
    for i in range(reps + 1):
        t0 = time.perf_counter()
        K = orc.assemble_csr(orc.elasticity_Ke(X, conn, "quad"), conn_dof)
        times.append(time.perf_counter() - t0)
    t_asm = float(np.median(times[1:]))
    rhs = orc.elasticity_rhs(K.shape[0], force)
    Kbc, rhs = orc.apply_dirichlet_bcs(K, rhs, dof_fixed)
    v = np.random.RandomState(0).rand(K.shape[0])
    t0 = time.perf_counter()
    for _ in range(5):
        Kbc.dot(v)
    t_spmv = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    try:
        orc.jacobi_pcg(Kbc, rhs, rtol=1e-30, maxiter=cg_iters)
    except RuntimeError:
        pass
    t_cg = (time.perf_counter() - t0) / cg_iters
    if limiter is not None:
        limiter.restore_original_limits()
    return {
        "value": conn.shape[0] / t_asm, "unit": "elements/s", "cores": int(threads), "kind": "port",
        "sample": f"{n}x{n}-quad plane-stress mesh ({conn.shape[0]} elements), oracle einsum Ke + coo->csr, "
                  f"median of {reps} warm passes; BLAS threads limited to the {threads} usable cpus of "
                  f"{os.cpu_count()} logical ones",
        "cg_iter_per_sec": 1.0 / t_cg, "spmv_GBps": algorithmic_bytes_spmv(Kbc.nnz, Kbc.shape[0]) / t_spmv / 1e9,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=2048, help="quads per side (2048 -> the 4M-quad metric mesh)")
    ap.add_argument("--cg-iters", type=int, default=200)
    ap.add_argument("--spmv-reps", type=int, default=50)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # rehearsal knobs (a 1-GPU box cannot host two RCCL ranks): PF_BENCH_BACKEND=gloo with
    # PF_BENCH_DEVICE=0 runs the N>1 code path with every rank on the same GPU
    if "PF_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["PF_BENCH_DEVICE"])
    backend = os.environ.get("PF_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from pyfem_gpu_testflight_amd import _lib, pyfem

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- workload (synthetic, ProblemCreator semantics of the reference) --------------------
    nx = args.nx
    creator = pyfem.ProblemCreator(nnodes_x=nx + 1, nnodes_y=nx + 1)
    conn, X, dof_fixed, nodal_force = creator.create_linear_elasticity_problem()
    nelems_total = conn.shape[0]
    quadrature = pyfem.QuadratureBilinear2D()
    basis = pyfem.BasisBilinear2D(quadrature)
    t0 = time.perf_counter()
    if world == 1:
        model = pyfem.LinearElasticity(X, conn, dof_fixed, None, nodal_force, quadrature, basis)
        pm = None
    else:
        from pyfem_gpu_testflight_amd.dist import PartitionedElasticity, dist_cg

        pm = PartitionedElasticity(X, conn, dof_fixed, None, nodal_force, quadrature, basis)
        model = pm.model
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t0
    h = model.hctx

    # ---- assembly: W warm-up + exactly K timed steps ------------------------------------------
    K = model.compute_jacobian()
    for _ in range(args.warmup):
        model.compute_jacobian(out=K)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        model.compute_jacobian(out=K)
    e1.record()
    barrier()
    t_asm = max_over_ranks(time.perf_counter() - t0)
    kern_ms = e0.elapsed_time(e1) / args.steps  # HIP events on the stream the kernel runs on
    value = nelems_total * args.steps / t_asm
    b_asm = algorithmic_bytes_assembly(h.nelems, h.nnodes, h.nnz)
    asm_gbs = b_asm / (kern_ms * 1e-3) / 1e9

    # ---- SpMV roofline leg (local block) ---------------------------------------------------------
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    xv = torch.rand(h.nrows, generator=g, dtype=torch.float64).to(dev)
    yv = torch.empty_like(xv)
    spmv_ms = h.time_kernel(_lib.PF_K_SPMV, args.spmv_reps, in0=K.vals, in1=xv, out0=yv)
    b_spmv = algorithmic_bytes_spmv(h.nnz, h.nrows)
    spmv_gbs = b_spmv / (spmv_ms * 1e-3) / 1e9

    # ---- CG leg: fixed iteration count on the BC-applied system -----------------------------------
    if world == 1:
        rhs = h.to_device(model.compute_rhs(), torch.float64)
        Kbc, b = model.apply_dirichlet_bcs(K, rhs, True)
        h.cg(Kbc.vals, b, rtol=0.0, maxiter=10, allow_maxiter=True)  # warm-up
        barrier()
        t0 = time.perf_counter()
        _, iters, relres = h.cg(Kbc.vals, b, rtol=0.0, maxiter=args.cg_iters, check_every=args.cg_iters,
                                allow_maxiter=True)
        barrier()
        t_cg = time.perf_counter() - t0
    else:
        rhs = pm.ops.from_numpy(model.compute_rhs(), torch.float64)
        Kbc, b = pm.apply_dirichlet_bcs(K, rhs)
        dinv = 1.0 / pm.iface.sum(pm.ops.diag(Kbc.vals))

        def run(n):
            try:
                dist_cg(pm.ops, Kbc.vals, b, dinv, pm.own_w, pm.iface, rtol=0.0, maxiter=n, check_every=10 ** 9)
            except RuntimeError:
                pass

        run(10)
        barrier()
        t0 = time.perf_counter()
        run(args.cg_iters)
        barrier()
        t_cg = max_over_ranks(time.perf_counter() - t0)
        iters = args.cg_iters
    cg_rate = iters / t_cg

    # HBM traffic per launch from the committed rocprofv3 --pmc passes of this same command
    # (profiles/r01_pmc_summary.json: WRITE_SIZE + FETCH_SIZE in KB, separate passes; FETCH_SIZE of the
    # 16-B/lane SpMV stream doubled as MI355X_MICROARCH.md prescribes; of the assembly's reads only the visit-record
    # stream is a 16-B/lane coalesced read, so the half of its 16 B x (4 nelems) that the counter misses is added back,
    # the gathers are left as counted).  null when the summary is missing or the mesh is not the profiled one.
    traffic_asm = traffic_spmv = None
    try:
        if nx == 2048 and world == 1:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_summary.json")))
            for k, v in pm.items():
                if "k_assemble_v3" in k:
                    traffic_asm = 1024.0 * (v["WRITE_SIZE"] + v["FETCH_SIZE"]) + 0.5 * 16.0 * 4 * h.nelems
                if "k_spmv" in k:
                    traffic_spmv = 1024.0 * (v["WRITE_SIZE"] + 2.0 * v["FETCH_SIZE"])
    except Exception:
        pass

    out = {
        "metric": "element-assemblies/sec (4M-quad plane-stress); CG-iter/sec alongside",
        "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * t_asm / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{nx}x{nx}-quad plane-stress mesh (ProblemCreator), E=10 nu=0.3 rho=1, "
                               f"{nelems_total} elements, ndof={2 * (nx + 1) ** 2}, element-block partition over "
                               f"{world} GPU(s)", "nelems": nelems_total, "cg_iters_timed": int(iters)},
        "cg_iter_per_sec": cg_rate,
        "roofline": {"bound": "hbm", "kernel": "k_assemble_v3<4,Elasticity<4,false>>", "achieved": asm_gbs,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": asm_gbs / HBM_PEAK_GBS, "traffic": traffic_asm,
                     "algorithmic_bytes_per_launch": b_asm, "avg_launch_ms": kern_ms},
        "roofline_spmv": {"bound": "hbm", "kernel": "k_spmv<2,1,3,640,true,double>", "achieved": spmv_gbs, "peak": HBM_PEAK_GBS,
                          "unit": "GB/s", "frac": spmv_gbs / HBM_PEAK_GBS, "traffic": traffic_spmv,
                          "algorithmic_bytes_per_launch": b_spmv, "avg_launch_ms": spmv_ms,
                          "note": "algorithmic bytes are CSR's 12 B per scalar nonzero (SURVEY 8d); the 2x2-block layout "
                                  "with 16-bit column offsets moves 8.5 B, so frac can exceed the real HBM utilisation "
                                  "(traffic / time)"},
        "setup_s": t_setup,
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
