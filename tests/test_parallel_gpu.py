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
RANK_DEATHS = 0   # ranks lost to a signal in this session (see _run_distributed_case)


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


def _flow_parameters(k, two_phase):
    """(the scheme of a case travels to the spawned ranks through the environment)"""
    return adaflo_amd.FlowParameters(velocity_degree=k, density_diff=0.5 if two_phase else 0.0,
                                     linearization=os.environ.get("ADAFLO_TEST_LINEARIZATION", "coupled implicit Newton"))


def _make(fp):
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()
    return ts


def _worker(rank, world, port, grid, cells, k, gu, gp, glin, gcoef, ref_u, ref_p, results, crumbs):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    op = nat = None
    try:
        dev = torch.device("cuda", 0)
        lower, upper = [-1.0] * 3, [-1.0 + 0.5 * g for g in grid]
        part = parallel.BrickPartition(grid, rank, cells, lower, upper)
        fp = _flow_parameters(k, gcoef is not None)
        op = parallel.DistributedNavierStokesMatrix(fp, part, device=0)
        op.initialize(_make(fp), True)
        cs = tuple(slice(part.coords[d] * cells[d], (part.coords[d] + 1) * cells[d]) for d in (2, 1, 0))

        def set_state(local):
            local.set_linearization(np.ascontiguousarray(glin[cs]).reshape(-1))
            if gcoef is not None:      # two-phase operator: density, viscosity, damping at the quadrature points
                local.set_coefficients(*[np.ascontiguousarray(c[cs]).reshape(-1) for c in gcoef])
        set_state(op.local)
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
        set_state(nat.local)
        nctx = nat.local._ctx
        nsrc = adaflo_amd.BlockVector([V(nctx, su), V(nctx, sp)])
        ndst = adaflo_amd.BlockVector([V(nctx, du), V(nctx, dp)])
        # (k > 2: variant 1 is the high-order sweep kernel; it has no phased schedule, `overlap` is then ignored)
        combos = ((1, True, False), (1, False, False), (0, False, False), (1, True, True), (0, False, True))
        if k == 4 and gcoef is None and adaflo_amd.NavierStokesMatrix.has_kernel_variant(3):
            combos += ((3, True, True),)                    # the plane-per-lane kernel under the two-stream schedule
        only = os.environ.get("ADAFLO_TEST_VARIANTS")       # (scripts/dev/stress_parallel.py: bisecting a fault)
        if only:
            combos = tuple(c for c in combos if str(c[0]) in only.split(","))
        for variant, overlap, native in combos:
            o, a, b = (nat, ndst, nsrc) if native else (op, dst, src)
            crumbs[rank] = "variant %d overlap %s native %s" % (variant, overlap, native)   # read after a rank death
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
            if not (torch.equal(cu, du) and torch.equal(cp, dp)):
                bad_u, bad_p = torch.nonzero(cu != du).reshape(-1), torch.nonzero(cp != dp).reshape(-1)
                raise AssertionError("replicas differ: rank %d variant %d overlap %s native %s: %d velocity entries "
                                     "(first %s: %r vs %r), %d pressure entries (first %s)"
                                     % (rank, variant, overlap, native, bad_u.numel(),
                                        bad_u[:4].tolist(), cu[bad_u[:4]].tolist(), du[bad_u[:4]].tolist(),
                                        bad_p.numel(), bad_p[:4].tolist()))
        crumbs[rank] = "teardown"
    finally:
        # explicit, ordered teardown: nothing of the engine is left to the interpreter's exit sequence
        torch.cuda.synchronize()
        if nat is not None:
            if nat.comm is not None:
                nat.comm.close()
            nat.local.clear()
        if op is not None:
            op.local.clear()
        torch.cuda.synchronize()
        dist.destroy_process_group()


def _reference(gcells, grid, k, two_phase, glin, gcoef, gu, gp):
    fp = _flow_parameters(k, two_phase)
    ref = adaflo_amd.NavierStokesMatrix(fp, adaflo_amd.BrickMesh(gcells, [-1.0] * 3, [-1.0 + 0.5 * g for g in grid]))
    ref.initialize(_make(fp), True)
    ref.set_linearization(glin.reshape(-1))
    if two_phase:
        ref.set_coefficients(*[c.reshape(-1) for c in gcoef])
    dst = ref.block_vector()
    ref.vmult(dst, ref.block_vector(gu.reshape(-1), gp.reshape(-1)))
    ref_u, ref_p = dst.numpy()
    ref.clear()
    del ref
    return ref_u, ref_p


def _oracle_reference(gcells, grid, k, two_phase, glin, gcoef, gu, gp):
    """the same global operator application by the CPU oracle (oracle/adaflo_oracle.c, the restatement of
    navier_stokes_matrix.cc:221-262,601-916) -- the distributed result is then pinned to the oracle directly, not only
    to the undivided engine (round-4 review, Weak #2)"""
    from common import BETA, LIN, PHYS
    from oracle import oracle as orc
    fp = _flow_parameters(k, two_phase)
    ts = _make(fp)
    mesh = orc.Mesh.make(list(gcells), [-1.0] * 3, [-1.0 + 0.5 * g for g in grid])
    inv = lambda table, v: [a for a, b in table.items() if b == v][0]   # noqa: E731
    prm = orc.NSParams.make(
        physical_type=inv(PHYS, fp.physical_type), linearization=inv(LIN, fp.linearization),
        beta=inv(BETA, fp.formulation_convective_term), tau_grad_div=fp.tau_grad_div, density=fp.density,
        viscosity=fp.viscosity, damping=-fp.damping, density_diff=fp.density_diff, weight=ts.weight(),
        weight_old=ts.weight_old(), weight_old_old=ts.weight_old_old(), tau1=ts.tau1(),
        extrap_old=ts.factor_extrapol_old, extrap_old_old=ts.factor_extrapol_old_old)
    con_u = orc.boundary_mask(mesh, k, 3, faces=range(6))
    con_p = orc.boundary_mask(mesh, k - 1, 1, faces=())
    w = orc.ns_pressure_mass_weight(mesh, k, con_p)
    co = {} if not two_phase else dict(rho=gcoef[0].reshape(-1), mu=gcoef[1].reshape(-1), damp=gcoef[2].reshape(-1))
    return orc.ns_vmult(mesh, k, prm, gu.reshape(-1), gp.reshape(-1), con_u, con_p, lin=glin.reshape(-1),
                        weights=w, modes=np.ones(mesh.n_nodes(k - 1)), **co)


def _reference_worker(rank, gcells, grid, k, two_phase, glin, gcoef, gu, gp, out):
    out["u"], out["p"] = _reference(gcells, grid, k, two_phase, glin, gcoef, gu, gp)


def _run_distributed_case(world, cells, k=2, two_phase=False, against_oracle=False):
    grid = parallel.brick_grid(world)
    rng = np.random.default_rng(5)
    gcells = [g * c for g, c in zip(grid, cells)]
    nq = (k + 1) ** 3
    nu = [k * n + 1 for n in gcells]
    npn = [(k - 1) * n + 1 for n in gcells]
    gu = rng.uniform(-1, 1, (nu[2], nu[1], nu[0], 3))
    gp = rng.uniform(-1, 1, (npn[2], npn[1], npn[0], 1))
    glin = rng.uniform(-1, 1, (gcells[2], gcells[1], gcells[0], nq * 12))
    gcoef = None
    if two_phase:
        gcoef = [rng.uniform(lo, hi, (gcells[2], gcells[1], gcells[0], nq)) for lo, hi in ((.5, 2.), (.5, 2.), (-.5, .5))]
    # reference: the same engine on the undivided mesh (ADAFLO_TEST_REF_IN_CHILD: computed by a child process, so that
    # the parent holds no GPU context of its own while the ranks run -- scripts/dev/stress_parallel.py uses it to
    # test whether the rare rank deaths need more GPU processes than the driver keeps resident at a time)
    if os.environ.get("ADAFLO_TEST_REF_IN_CHILD"):
        mgr0 = mp.Manager()
        out = mgr0.dict()
        mp.spawn(_reference_worker, args=(gcells, grid, k, two_phase, glin, gcoef, gu, gp, out), nprocs=1, join=True)
        ref_u, ref_p = out["u"], out["p"]
    else:
        ref_u, ref_p = _reference(gcells, grid, k, two_phase, glin, gcoef, gu, gp)
    if against_oracle:
        # the ranks are compared with the ORACLE's application of the global operator; the undivided engine has to agree
        # with it as well (it is what the other cases use)
        eng_u, eng_p = ref_u, ref_p
        ref_u, ref_p = _oracle_reference(gcells, grid, k, two_phase, glin, gcoef, gu, gp)
        assert rel_l2(eng_u, ref_u) < 1e-12 and rel_l2(eng_p, ref_p) < 1e-12
    mgr = mp.Manager()
    results, crumbs = None, None
    # Up to eight processes share ONE GPU here (production: one process per GPU).  A rank of an 8-process job dies
    # now and then with "Queue ... aborting with error: HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION" (never in 2- or 4-process
    # jobs).  Round 4 (scripts/dev/stress_parallel.py 2 8 big, logs in profiles/r04_stress_*.log): 3 deaths in 33 jobs
    # as before; 1 in 25 with the parent holding no GPU context -- and that rank died BEFORE its first vmult (the
    # breadcrumbs below were still empty), so the sweep kernels with their hand-issued LDS-DMA are not what faults;
    # 0 in 25 with GPU_MAX_HW_QUEUES=1 (one hardware queue per process instead of four).  The deaths need more
    # processes / hardware queues on one device than the driver keeps resident (it then preempts running waves to
    # time-slice them), a condition the product never creates.  So the 8-process jobs run with one hardware queue
    # per process; a death by SIGNAL is still reported loudly (warning + breadcrumbs of what every rank was doing).  It
    # is repeated ONCE, and only if the dead rank had not yet applied an operator (no breadcrumb: the observed deaths);
    # a rank that dies after it started operator applications fails the test at once, as does a second death.  Python
    # exceptions in a rank (wrong numbers, engine errors) arrive as ProcessRaisedException and are never retried.
    saved_queues = os.environ.get("GPU_MAX_HW_QUEUES")
    if world >= 8 and "ADAFLO_TEST_KEEP_QUEUES" not in os.environ:
        os.environ["GPU_MAX_HW_QUEUES"] = "1"          # read by the HIP runtime of the spawned ranks
    try:
        for attempt in range(2):
            try:
                # fresh dictionaries per attempt: ranks of an aborted attempt that are still being torn down must not be
                # able to write into what the repetition collects (round 4: the one wrong pressure block of the stress
                # series, profiles/r04_stress_8rank_refchild.log repetition 15, came out of exactly such a repetition --
                # a rank death, then the retry -- with the dictionary of the first attempt cleared and re-used)
                results, crumbs = mgr.dict(), mgr.dict()
                mp.spawn(_worker, args=(world, _free_port(), grid, list(cells), k, gu, gp, glin, gcoef, ref_u, ref_p,
                                        results, crumbs), nprocs=world, join=True)
                break
            except mp.ProcessExitedException as e:
                global RANK_DEATHS
                RANK_DEATHS += 1
                import warnings
                where = dict(crumbs)
                warnings.warn("a rank of the %d-process job died by signal (%s); death number %d of this session, attempt "
                              "%d; the ranks were at: %s" % (world, e, RANK_DEATHS, attempt + 1, where), RuntimeWarning)
                # only a death BEFORE the rank launched anything of this engine (no breadcrumb yet: runtime / context
                # start-up of one of eight processes on one device) is repeated, once; a rank that had started operator
                # applications may have died of a kernel fault and fails the test at once
                died = getattr(e, "error_index", None)
                if attempt == 1 or world < 8 or died is None or where.get(died) is not None:
                    raise
    finally:
        if saved_queues is None:
            os.environ.pop("GPU_MAX_HW_QUEUES", None)
        else:
            os.environ["GPU_MAX_HW_QUEUES"] = saved_queues
    only = os.environ.get("ADAFLO_TEST_VARIANTS")
    variants = (1, 1, 0, 1, 0) + ((3,) if (k == 4 and not two_phase and adaflo_amd.NavierStokesMatrix.has_kernel_variant(3)) else ())
    n_combos = len(variants) if not only else sum(1 for v in variants if str(v) in only.split(","))
    assert len(results) == n_combos * world, (len(results), n_combos, world)
    for key, (eu, ep) in results.items():
        assert eu < 1e-12 and ep < 1e-12, (key, eu, ep)


# the larger bricks have workgroups in all three phases (interface / interior A / interior B)
@pytest.mark.parametrize("world,cells", [(2, (9, 8, 5)), (4, (8, 5, 6)), (8, (4, 5, 3)), (2, (40, 24, 12)),
                                         (8, (24, 17, 12))])
def test_distributed_vmult_on_one_gpu(world, cells):
    _run_distributed_case(world, cells)


@pytest.mark.parametrize("world,cells,k,two_phase", [(2, (40, 24, 12), 2, False), (8, (24, 17, 12), 2, False),
                                                     (8, (4, 3, 3), 4, False), (2, (9, 8, 5), 2, True)])
def test_distributed_vmult_against_the_oracle(world, cells, k, two_phase):
    """2 and 8 ranks of the real engine (phased and plain schedules, Python-driven and native exchange) against the CPU
    ORACLE's application of the undivided operator, entry by entry on every rank's brick -- Q2/Q1 on bricks with workgroups
    in all three phases, Q4/Q3 on 2 x 2 x 2 bricks, the two-phase operator"""
    _run_distributed_case(world, cells, k=k, two_phase=two_phase, against_oracle=True)


@pytest.mark.parametrize("world,cells,k", [(2, (40, 24, 12), 2), (2, (9, 8, 5), 2), (2, (5, 4, 3), 4)])
def test_distributed_vmult_of_the_projection_scheme_against_the_oracle(world, cells, k, monkeypatch):
    """the projection scheme integrates no pressure row (navier_stokes_matrix.cc:902-907): dst_p = -src_p on constrained
    rows, 0 elsewhere, prepared by a small kernel of its own.  In the two-stream schedule of adaflo_ns_vmult_distributed
    that kernel has to run before the auxiliary stream packs and unpack-adds dst_p (it ran beside them)"""
    monkeypatch.setenv("ADAFLO_TEST_LINEARIZATION", "projection")
    _run_distributed_case(world, cells, k=k, against_oracle=True)


@pytest.mark.parametrize("world,cells", [(2, (5, 4, 3)), (8, (4, 3, 3))])
def test_distributed_vmult_q4_on_one_gpu(world, cells):
    """Q4/Q3 (BASELINE configs[4] runs on 8 GPUs): high-order sweep kernel and generic kernel on 2 / 8 bricks
    against the undivided engine"""
    _run_distributed_case(world, cells, k=4)


@pytest.mark.parametrize("world,cells", [(2, (9, 8, 5)), (8, (8, 9, 5))])
def test_distributed_two_phase_vmult_on_one_gpu(world, cells):
    """variable density / viscosity / damping at the quadrature points (the two-phase Jacobian), phased and plain"""
    _run_distributed_case(world, cells, two_phase=True)


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
