#!/usr/bin/env python3
"""bench.py -- MDoF/s of NavierStokesMatrix::vmult (3D Q2/Q1) + HBM roofline on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N = 1 : BASELINE.json configs[1] -- 3D Beltrami Q2/Q1, uniform 128^3 hex mesh,
          one `vmult` (= one "step") of the Newton-linearised NS operator.
  N > 1 : launched by torch.distributed.run, one rank per GPU; weak scaling: every
          rank owns a 128^3-cell brick of a (px*128, py*128, pz*128) mesh (N = 8 is
          configs[2], the 256^3 mesh), ghost-DoF exchange over RCCL.
One JSON line on stdout (rank 0).  Inputs are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_GBS = 6290.0       # measured float4 copy ceiling (same guide)
B_ALG_PER_CELL_Q2 = 16 * (3 * 2 ** 3 + 1 ** 3) + 8 * 12 * 27   # 400 + 2592 B


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cells", type=int, default=128, help="cells per direction per GPU")
    ap.add_argument("--degree", type=int, default=2)
    ap.add_argument("--variant", type=int, default=1, help="0 generic kernels, 1 specialised")
    ap.add_argument("--chunk", type=int, default=0, help="Q2 kernel z-chunk (0 = heuristic)")
    ap.add_argument("--state-pad", type=int, default=-1, help="Q2 state skew padding in 16 B units")
    ap.add_argument("--linearization", default="coupled implicit Newton",
                    help="diagnostic only: e.g. 'coupled velocity explicit' times the kernel without q-state")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-cells", type=int, default=48)
    return ap.parse_args()


def beltrami_nodal(torch, mesh_lower, h, ncell, degree, t, device):
    """nodal interpolant of the Beltrami field (tests/beltrami.cc:82-172) on the device"""
    a = 0.25 * np.pi
    d = 2.0 * a
    ax = []
    for dim in range(3):
        n = degree * ncell[dim] + 1
        idx = torch.arange(n, device=device, dtype=torch.float64)
        if degree == 2:
            x = mesh_lower[dim] + 0.5 * h[dim] * idx            # GLL nodes 0, 1/2, 1
        elif degree == 1:
            x = mesh_lower[dim] + h[dim] * idx
        else:
            from adaflo_amd import _lib  # noqa: F401  (higher degrees: host GLL nodes)
            raise NotImplementedError
        ax.append(x)
    z, y, x = torch.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    if degree == 1:
        p = -a * a * 0.5 * (torch.exp(2 * a * x) + torch.exp(2 * a * y) + torch.exp(2 * a * z)
                            + 2 * torch.sin(a * x + d * y) * torch.cos(a * z + d * x) * torch.exp(a * (y + z))
                            + 2 * torch.sin(a * y + d * z) * torch.cos(a * x + d * y) * torch.exp(a * (z + x))
                            + 2 * torch.sin(a * z + d * x) * torch.cos(a * y + d * z) * torch.exp(a * (x + y)))
        return (p * np.exp(-2 * d * d * t)).reshape(-1).contiguous()
    f = np.exp(-d * d * t)
    u0 = -a * (torch.exp(a * x) * torch.sin(a * y + d * z) + torch.exp(a * z) * torch.cos(a * x + d * y)) * f
    u1 = -a * (torch.exp(a * y) * torch.sin(a * z + d * x) + torch.exp(a * x) * torch.cos(a * y + d * z)) * f
    u2 = -a * (torch.exp(a * z) * torch.sin(a * x + d * y) + torch.exp(a * y) * torch.cos(a * z + d * x)) * f
    return torch.stack([u0, u1, u2], dim=-1).reshape(-1).contiguous()


def cpu_baseline(sample_cells, budget_s=12.0):
    """time the CPU restatement (oracle/adaflo_oracle_fast.c) on a bounded sample"""
    from oracle import oracle as orc
    orc.build()
    orc.fast_set_threads(orc.usable_cores())          # the container's CPU quota, not the visible core count
    n = sample_cells
    mesh = orc.Mesh.make([n] * 3, [-1.0] * 3, [1.0] * 3)
    prm = orc.NSParams.make(weight=1.5 / 0.05, weight_old=-2 / 0.05, weight_old_old=0.5 / 0.05)
    rng = np.random.default_rng(20260515)
    nu, npr = mesh.n_nodes(2) * 3, mesh.n_nodes(1)
    su, sp = rng.uniform(-1, 1, nu), rng.uniform(-1, 1, npr)
    lin = rng.uniform(-1, 1, mesh.n_cells * 27 * 12)
    con_u = orc.boundary_mask(mesh, 2, 3)
    w = orc.ns_pressure_mass_weight(mesh, 2)
    modes = np.ones(npr)
    out = (np.empty(nu), np.empty(npr))
    orc.fast_ns_vmult(mesh, 2, prm, su, sp, con_u, None, lin=lin, weights=w, modes=modes, out=out)
    reps, t0 = 0, time.perf_counter()
    while True:
        orc.fast_ns_vmult(mesh, 2, prm, su, sp, con_u, None, lin=lin, weights=w, modes=modes, out=out)
        reps += 1
        el = time.perf_counter() - t0
        if el > budget_s or reps >= 5000:
            break
    return {"value": round((nu + npr) * reps / el / 1e6, 2), "unit": "MDoF/s",
            "cores": orc.fast_n_threads(), "kind": "port",
            "sample": "%d^3-cell Q2/Q1 brick, %d vmults of the sum-factorised OpenMP restatement "
                      "of the adaflo path (deal.II unavailable), %.1f s" % (n, reps, el)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run "
                             "(one rank per GPU)" % args.gpus)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP engine)")
    # functional dry run of the N > 1 path on a single GPU (not a measurement):
    #   ADAFLO_BENCH_BACKEND=gloo ADAFLO_BENCH_SINGLE_DEVICE=1 torchrun --nproc-per-node 2 bench.py --gpus 2 --cells 32
    backend = os.environ.get("ADAFLO_BENCH_BACKEND", "nccl")
    if os.environ.get("ADAFLO_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import adaflo_amd
    from adaflo_amd import build as _build
    from adaflo_amd import parallel
    if not os.path.exists(_build.LIB):
        _build.build()

    nc, k = args.cells, args.degree
    fp = adaflo_amd.FlowParameters(velocity_degree=k, time_step_size_start=0.05, end_time=1.0,
                                   linearization=args.linearization)
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()                                   # steady BDF-2 weights: gamma = 1.5/dt
    grid = parallel.brick_grid(world)
    part = parallel.BrickPartition(grid, rank, [nc] * 3, lower=[-1.0] * 3,
                                   upper=[-1.0 + 2.0 * g for g in grid] if world > 1 else [1.0] * 3)
    stream = torch.cuda.current_stream(device).cuda_stream   # 0 = the legacy default stream
    op = parallel.DistributedNavierStokesMatrix(fp, part, device=local_rank, stream=stream,
                                                group=dist.group.WORLD if world > 1 else None)
    op.initialize(ts, True)
    op.set_kernel_variant(args.variant)
    if args.chunk:
        op.local.set_q2_chunk(args.chunk)
    if args.state_pad >= 0:
        op.local.set_q2_state_pad(args.state_pad)

    mesh = op.local.mesh
    n_u, n_p = op.local.n_dofs_u(), op.local.n_dofs_p()
    # linearisation point = nodal interpolant of the Beltrami field at t = 0, pushed through
    # the residual kernel, which is the only producer of the q-point state in the reference
    u_lin = beltrami_nodal(torch, mesh.lower, mesh.h, mesh.ncell, k, 0.0, device)
    p_lin = torch.zeros(n_p, device=device, dtype=torch.float64)
    zeros_u = torch.zeros(n_u, device=device, dtype=torch.float64)
    V = adaflo_amd.DeviceVector.from_torch
    ctx = op.local._ctx
    rhs = adaflo_amd.BlockVector([V(ctx, torch.zeros_like(u_lin)), V(ctx, torch.zeros_like(p_lin))])
    op.local.residual(rhs, adaflo_amd.BlockVector([V(ctx, u_lin), V(ctx, p_lin)]), None,
                      adaflo_amd.BlockVector([V(ctx, u_lin)]), adaflo_amd.BlockVector([V(ctx, zeros_u)]))
    del rhs
    # deterministic pseudo-random source in [-1,1], identical global vector for any partitioning
    gen = torch.Generator(device=device)
    gen.manual_seed(20260515 + rank)
    src_u = torch.rand(n_u, device=device, dtype=torch.float64, generator=gen) * 2 - 1
    src_p = torch.rand(n_p, device=device, dtype=torch.float64, generator=gen) * 2 - 1
    dst_u, dst_p = torch.empty_like(src_u), torch.empty_like(src_p)
    src = adaflo_amd.BlockVector([V(ctx, src_u), V(ctx, src_p)])
    dst = adaflo_amd.BlockVector([V(ctx, dst_u), V(ctx, dst_p)])
    op.make_consistent(src)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        op.vmult(dst, src)
    op.local.get_kernel_statistics()
    op.local.get_matvec_statistics()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        op.vmult(dst, src)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ksec, kcount = op.local.get_kernel_statistics()
    msec, mcount = op.local.get_matvec_statistics()

    n_dofs_global = part.n_global_dofs(k)
    n_cells_local = op.local.n_cells()
    value = n_dofs_global * args.steps / elapsed / 1e6
    kernel_avg = ksec / max(kcount, 1)
    b_alg_launch = B_ALG_PER_CELL_Q2 * n_cells_local if k == 2 else None
    achieved = b_alg_launch / kernel_avg / 1e9 if b_alg_launch else None
    if not torch.isfinite(dst_u).all():
        raise SystemExit("non-finite result")

    traffic = None
    try:  # PMC-measured HBM bytes per launch of the dominant kernel (profiles/, collected with rocprofv3)
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            pmc = json.load(f)
        if pmc.get("workload") == "%dx%dx%d k=%d variant=%d" % (nc, nc, nc, k, args.variant):
            traffic = pmc["ns_q2_kernel"]["hbm_bytes"]
    except (OSError, KeyError, ValueError):
        pass
    out = {
        "metric": "MDoF/s for NavierStokesMatrix::vmult (3D Q2/Q1)",
        "value": round(value, 1), "unit": "MDoF/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "3D Beltrami Q%d/Q%d NavierStokes vmult, Newton linearisation, "
                               "uniform hex mesh %s cells (%d^3 per GPU), Dirichlet on all faces, "
                               "pressure mean projection" % (k, k - 1, "x".join(
                                   str(g * nc) for g in grid), nc),
                   "dofs": n_dofs_global, "cells_per_gpu": n_cells_local,
                   "partition": "x".join(str(g) for g in grid), "kernel_variant": args.variant},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1) if achieved else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                     "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 4) if achieved else None,
                     "traffic": traffic, "kernel": "ns_q2_kernel" if (k == 2 and args.variant == 1)
                     else "ns_cell_kernel", "kernel_ms": round(1e3 * kernel_avg, 4),
                     "alg_bytes_per_launch": b_alg_launch, "vmult_ms_device": round(1e3 * msec / max(mcount, 1), 4)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_sample_cells)
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
