"""End-to-end test of the multi-GPU path on ONE GPU: N ranks (gloo, all on cuda:0) each drive
the HIP engine on their brick, interface DoFs go through the native pack/unpack kernel
(messages staged through host memory because gloo cannot move device tensors); the assembled
result must equal the single-engine vmult on the global mesh.  On an 8-GPU node the same code
runs over RCCL (bench.py --gpus N)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import adaflo_amd
from adaflo_amd import parallel
from common import rel_l2

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _view(g, part, degree):
    sl = []
    for d in (2, 1, 0):
        lo = part.coords[d] * part.cells[d] * degree
        sl.append(slice(lo, lo + part.cells[d] * degree + 1))
    return g[tuple(sl)]


def _make(fp):
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    return ts


def _worker(rank, world, port, grid, cells, gu, gp, glin, ref_u, ref_p, results):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        k = 2
        lower, upper = [-1.0] * 3, [-1.0 + 0.5 * g for g in grid]
        part = parallel.BrickPartition(grid, rank, cells, lower, upper)
        fp = adaflo_amd.FlowParameters(velocity_degree=k)
        op = parallel.DistributedNavierStokesMatrix(fp, part, device=0)
        op.initialize(_make(fp), True)
        cs = tuple(slice(part.coords[d] * cells[d], (part.coords[d] + 1) * cells[d]) for d in (2, 1, 0))
        op.local.set_linearization(np.ascontiguousarray(glin[cs]).reshape(-1))
        halo = op.halo
        ou, opm = halo.owned_mask(0).numpy(), halo.owned_mask(1).numpy()
        lu, lp = _view(gu, part, k).reshape(-1), _view(gp, part, k - 1).reshape(-1)
        V = adaflo_amd.DeviceVector.from_torch
        ctx = op.local._ctx
        # stale replicas in src: the operator has to import the owners' values
        su = torch.from_numpy(np.where(ou > 0, lu, -5.0)).to(dev)
        sp = torch.from_numpy(np.where(opm > 0, lp, -5.0)).to(dev)
        du, dp = torch.full_like(su, 3.0), torch.full_like(sp, 3.0)
        src = adaflo_amd.BlockVector([V(ctx, su), V(ctx, sp)])
        dst = adaflo_amd.BlockVector([V(ctx, du), V(ctx, dp)])
        # Q2/Q1 sweep kernel with and without the overlapped (phased) schedule, generic kernel -- exchange
        # driven from Python (torch.distributed) and inside the library (adaflo_ns_vmult_distributed with the
        # gloo-staged transport callbacks)
        nat = parallel.DistributedNavierStokesMatrix(fp, part, device=0, group=dist.group.WORLD, native_comm=True)
        nat.initialize(_make(fp), True)
        nat.local.set_linearization(np.ascontiguousarray(glin[cs]).reshape(-1))
        nctx = nat.local._ctx
        nsrc = adaflo_amd.BlockVector([V(nctx, su), V(nctx, sp)])
        ndst = adaflo_amd.BlockVector([V(nctx, du), V(nctx, dp)])
        for variant, overlap, native in ((1, True, False), (1, False, False), (0, False, False), (1, True, True), (0, False, True)):
            o, a, b = (nat, ndst, nsrc) if native else (op, dst, src)
            o.set_kernel_variant(variant)
            o.overlap = overlap
            du.fill_(3.0)
            dp.fill_(3.0)
            su.copy_(torch.from_numpy(np.where(ou > 0, lu, -5.0)))
            sp.copy_(torch.from_numpy(np.where(opm > 0, lp, -5.0)))
            o.vmult(a, b)
            torch.cuda.synchronize()
            nu = [k * g * c + 1 for g, c in zip(grid, cells)]
            npn = [(k - 1) * g * c + 1 for g, c in zip(grid, cells)]
            ru = _view(ref_u.reshape(nu[2], nu[1], nu[0], 3), part, k).reshape(-1)
            rp = _view(ref_p.reshape(npn[2], npn[1], npn[0], 1), part, k - 1).reshape(-1)
            results[(rank, variant, overlap, native)] = (rel_l2(du.cpu().numpy(), ru), rel_l2(dp.cpu().numpy(), rp))
            # the replicas of an interface DoF are bitwise equal on all sharers: importing the owners' values
            # changes nothing (compress(add) sums in a fixed order on every rank)
            cu, cp = du.clone(), dp.clone()
            halo.update_ghost_values([cu, cp])
            torch.cuda.synchronize()
            assert torch.equal(cu, du) and torch.equal(cp, dp), (rank, variant, overlap, native)
        nat.comm.close()
    finally:
        dist.destroy_process_group()


# the larger bricks have workgroups in all three phases (interface / interior A / interior B)
@pytest.mark.parametrize("world,cells", [(2, (9, 8, 5)), (4, (8, 5, 6)), (8, (4, 5, 3)), (2, (40, 24, 12)),
                                         (8, (24, 17, 12))])
def test_distributed_vmult_on_one_gpu(world, cells):
    grid = parallel.brick_grid(world)
    k = 2
    rng = np.random.default_rng(5)
    gcells = [g * c for g, c in zip(grid, cells)]
    nu = [k * n + 1 for n in gcells]
    npn = [n + 1 for n in gcells]
    gu = rng.uniform(-1, 1, (nu[2], nu[1], nu[0], 3))
    gp = rng.uniform(-1, 1, (npn[2], npn[1], npn[0], 1))
    glin = rng.uniform(-1, 1, (gcells[2], gcells[1], gcells[0], 27 * 12))
    # reference: the same engine on the undivided mesh
    fp = adaflo_amd.FlowParameters(velocity_degree=k)
    ref = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh(gcells, [-1.0] * 3, [-1.0 + 0.5 * g for g in grid]))
    ref.initialize(_make(fp), True)
    ref.set_linearization(glin.reshape(-1))
    dst = ref.block_vector()
    ref.vmult(dst, ref.block_vector(gu.reshape(-1), gp.reshape(-1)))
    ref_u, ref_p = dst.numpy()
    del ref
    mgr = mp.Manager()
    results = mgr.dict()
    # Up to eight processes time-slice ONE GPU here.  A rank that dies with a device fault of the time-sliced
    # queue (seen once in ~25 runs of the 8-rank cases: "HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" at abort, no Python
    # exception) is a property of this test set-up, not of the exchange: such a run is repeated once.  Python
    # exceptions in a rank (wrong numbers, engine errors) arrive as ProcessRaisedException and are never retried.
    for attempt in range(2):
        try:
            results.clear()
            mp.spawn(_worker, args=(world, _free_port(), grid, list(cells), gu, gp, glin, ref_u, ref_p, results),
                     nprocs=world, join=True)
            break
        except mp.ProcessExitedException:
            if attempt == 1:
                raise
    assert len(results) == 5 * world
    for key, (eu, ep) in results.items():
        assert eu < 1e-12 and ep < 1e-12, (key, eu, ep)


def test_native_communicator_world_one_over_rccl():
    """adaflo_comm_create with the RCCL transport on a world of one rank (the only world a one-GPU box
    offers): the library resolves RCCL, opens a communicator, and adaflo_ns_vmult_distributed equals the
    local operator including the global form of the mean-value projection"""
    import ctypes as C
    from adaflo_amd import _lib
    lib = _lib.load()
    k, cells = 2, [9, 8, 5]
    rng = np.random.default_rng(3)
    fp = adaflo_amd.FlowParameters(velocity_degree=k)
    mesh = adaflo_amd.BrickMesh(cells, [-1.0] * 3, [1.0] * 3)
    lin = rng.uniform(-1, 1, int(np.prod(cells)) * 27 * 12)
    ref = adaflo_amd.NavierStokesMatrix(fp, mesh)
    ref.initialize(_make(fp), True)
    ref.set_linearization(lin)
    su, sp = rng.uniform(-1, 1, ref.n_dofs_u()), rng.uniform(-1, 1, ref.n_dofs_p())
    dst = ref.block_vector()
    ref.vmult(dst, ref.block_vector(su, sp))
    ref_u, ref_p = dst.numpy()
    op = adaflo_amd.NavierStokesMatrix(fp, mesh)
    op.initialize(_make(fp), False)                  # the communicator applies the projection
    op.set_linearization(lin)
    uid = _lib.CommUniqueId()
    assert lib.adaflo_comm_get_unique_id(C.byref(uid)) == 0
    comm = C.c_void_p()
    grid = (C.c_int * 3)(1, 1, 1)
    code = lib.adaflo_comm_create(op._ctx, C.byref(uid), 0, 1, grid, 1, C.byref(comm))
    assert code == 0, lib.adaflo_last_error(op._ctx)
    assert lib.adaflo_comm_interface_faces(comm) == 0
    src, dst = op.block_vector(su, sp), op.block_vector()
    assert lib.adaflo_ns_vmult_distributed(op._ctx, comm, dst.block(0).ptr, dst.block(1).ptr, src.block(0).ptr,
                                           src.block(1).ptr, 0) == 0
    got_u, got_p = dst.numpy()
    assert rel_l2(got_u, ref_u) < 1e-13 and rel_l2(got_p, ref_p) < 1e-12
    assert lib.adaflo_comm_destroy(comm) == 0
