#!/usr/bin/env python3
"""bench.py -- MDoF/s of NavierStokesMatrix::vmult (3D Q2/Q1) + HBM roofline on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  --config beltrami (default)
      N = 1 : BASELINE.json configs[1] -- 3D Beltrami Q2/Q1, uniform 128^3 hex mesh,
              one `vmult` (= one "step") of the Newton-linearised NS operator.
      N > 1 : one rank per GPU; weak scaling: every rank owns a 128^3-cell brick of a
              (px*128, py*128, pz*128) mesh (N = 8 is configs[2], the 256^3 mesh),
              ghost-DoF exchange over RCCL.
  --config cavity
      BASELINE.json configs[4] -- Q4/Q3 driven-cavity operator, 64^3 cells on
      [0,1]x[0,1]x[0,3] (applications/drivencavity.cc:356-370), `incompressible stationary`,
      mu = 0.01 (drivencavity.prm:9-14); the 64^3 mesh is FIXED and cut into bricks
      (strong scaling: 32^3 cells per rank at N = 8).
One JSON line on stdout (rank 0).  Inputs are resident in HBM before the timed region.
Set-up (untimed) ends with about a quarter of a second of the same operator application, so that the W
warm-up and K timed steps the caller asks for run at the steady clocks of the device even for small W.

With N > 1 and no torch.distributed environment the script launches its own N ranks
(`python -m torch.distributed.run`) BEFORE anything touches the GPU, forwards the JSON line
and exits with the child's return code.  If the box has fewer than N GPUs the ranks share
device 0 and talk gloo: a functional dry run ("dry_run": true), not a measurement.

Every rank of an N > 1 job -- started by the driver's `python -m torch.distributed.run` or by the self-launch -- is a
SUPERVISOR that never touches the GPU: it starts the measuring process as a child (same environment) and forwards rank 0's
JSON line.  If the child job fails with the engine's own RCCL communicator (`--comm native`, the default) -- a non-zero exit
code or no end within ADAFLO_BENCH_ATTEMPT_TIMEOUT seconds on ANY rank -- all supervisors stop their children and start a
fresh child job that moves the ghost layers through torch.distributed point-to-point operations (`--comm torch`, still
RCCL); the line then carries "comm": "torch" and "native_error": the error text of the first attempt.  The supervisors
of a job (one node) agree through files in a scratch directory below /tmp; nothing is re-executed in a process that has
initialised the GPU.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md
HBM_COPY_GBS = 6290.0       # measured float4 copy ceiling (same guide)
FP64_PEAK_TFLOPS = 78.6     # FP64 vector peak: 256 CUs x 128 flop/clk x 2.4 GHz
SEED = 20260515


def b_alg_per_cell(k):
    """vectors 16 (3 k^3 + (k-1)^3) + Newton state 8 * 12 * (k+1)^3 bytes per cell"""
    return 16 * (3 * k ** 3 + (k - 1) ** 3) + 8 * 12 * (k + 1) ** 3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="beltrami", choices=["beltrami", "cavity"])
    ap.add_argument("--cells", type=int, default=0,
                    help="beltrami: cells per direction per GPU (128); cavity: GLOBAL cells per direction (64)")
    ap.add_argument("--degree", type=int, default=0, help="velocity degree (beltrami 2, cavity 4)")
    ap.add_argument("--variant", type=int, default=1, help="0 generic kernels, 1 specialised, 2 = 1 with the round-2 z-sweep kernel for Q3..Q5, "
                    "3 = 1 with the plane-per-lane kernel for Q4/Q3")
    ap.add_argument("--chunk", type=int, default=0, help="Q2 kernel z-chunk (0 = heuristic)")
    ap.add_argument("--state-pad", type=int, default=-1, help="Q2 state skew padding in 16 B units")
    ap.add_argument("--linearization", default="coupled implicit Newton",
                    help="diagnostic only: e.g. 'coupled velocity explicit' times the kernel without q-state")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: blocking exchange schedule")
    ap.add_argument("--comm", default="native", choices=["torch", "native"],
                    help="N > 1: ghost exchange inside the engine (default; adaflo_ns_vmult_distributed of the C ABI: "
                         "RCCL group send/recv on a second stream) or driven through torch.distributed point-to-point "
                         "operations.  If the job with the engine's communicator fails, the supervising parent processes run it once "
                         "more with --comm torch and the line carries the RCCL error string (native_error).")
    ap.add_argument("--src-consistent", action="store_true",
                    help="N > 1: skip the owner->ghost update of src in the timed vmult (valid inside a Krylov loop, "
                         "where the replicas of an interface DoF are bitwise identical after the rank-ordered "
                         "compress).  Default: the full cell_loop exchange of the reference, update_ghost_values(src) "
                         "+ compress(add) of dst (navier_stokes_matrix.cc:232-245)")
    ap.add_argument("--full-exchange", action="store_true", help="(default since round 3; kept for scripts)")
    ap.add_argument("--through-comm", action="store_true",
                    help="N = 1 only: run the vmult through adaflo_ns_vmult_distributed with the three-phase schedule "
                         "forced (packs, events, second stream, three launches): the fixed cost of the multi-GPU path")
    ap.add_argument("--no-supervisor", action="store_true",
                    help="N > 1: measure in this process (no child process, no fall-back to --comm torch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--print-steps", action="store_true", help="per-step device times on stderr")
    ap.add_argument("--cpu-sample-cells", type=int, default=48)
    args = ap.parse_args()
    if args.through_comm and args.comm != "native":
        ap.error("--through-comm measures adaflo_ns_vmult_distributed: it needs --comm native")
    return args


def self_launch(args):
    """N > 1 without a torch.distributed environment: start the ranks as a child job.  Nothing in
    this process has touched the GPU (device_count() does not initialise it)."""
    import torch
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if torch.cuda.device_count() < args.gpus:
        env["ADAFLO_BENCH_BACKEND"] = "gloo"
        env["ADAFLO_BENCH_SINGLE_DEVICE"] = "1"
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def supervise(args, script=None, argv=None):
    """One rank of an N > 1 job as a supervisor (see the module docstring): child process = the measurement; on failure of
    the native-communicator job one more child job with --comm torch.  Returns the exit code."""
    rank, world = int(os.environ["RANK"]), int(os.environ.get("WORLD_SIZE", "1"))
    limit = float(os.environ.get("ADAFLO_BENCH_ATTEMPT_TIMEOUT", "900"))
    box = "/tmp/adaflo_bench_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    os.makedirs(box, exist_ok=True)

    def put(name, text=""):
        tmp = os.path.join(box, ".%s.%d" % (name, rank))
        with open(tmp, "w") as f:
            f.write(text)
        os.replace(tmp, os.path.join(box, name))

    def ls(prefix):
        try:
            return sorted(n for n in os.listdir(box) if n.startswith(prefix))
        except OSError:
            return []

    current = []                                           # the running child: stopped with the supervisor

    def stop(signum, frame):
        for ch in current:
            if ch.poll() is None:
                ch.kill()
        sys.exit(128 + signum)
    import signal
    for sig in (signal.SIGTERM, signal.SIGINT):
        try:
            signal.signal(sig, stop)
        except ValueError:                                 # (not the main thread: tests that call supervise() directly)
            pass

    def attempt(no, extra):
        env = dict(os.environ, ADAFLO_BENCH_WORKER="1", ADAFLO_BENCH_ATTEMPT=str(no), ADAFLO_BENCH_BOX=box)
        cmd = [sys.executable, script or os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else argv) + extra
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if rank == 0 else None, text=True)
        current[:] = [child]
        lines, t0, rc = [], time.perf_counter(), None
        if rank == 0:                                      # (reader thread: the pipe must not fill up)
            import threading
            reader = threading.Thread(target=lambda: lines.extend(child.stdout.readlines()), daemon=True)
            reader.start()
        while rc is None:
            rc = child.poll()
            if rc is None:
                others_failed = bool(ls("a%d.fail." % no))
                if others_failed or time.perf_counter() - t0 > limit:
                    child.kill()
                    child.wait()
                    rc = -9 if others_failed else -99      # (-99: no end within the limit)
                else:
                    time.sleep(0.05)
        if rank == 0:
            reader.join(timeout=10)
        if rc != 0:
            put("a%d.fail.%d" % (no, rank), str(rc))
        put("a%d.rc.%d" % (no, rank), str(rc))
        t1 = time.perf_counter()                            # every supervisor has an exit code within a minute of the first
        while len(ls("a%d.rc." % no)) < world and time.perf_counter() - t1 < 120:
            time.sleep(0.05)
        return (0 if not ls("a%d.fail." % no) else (rc or 1)), lines

    rc, lines = attempt(1, [])
    out_lines, note = lines, None
    if rc != 0 and args.comm == "native":
        errs = []
        for n in ls("a1.err."):
            with open(os.path.join(box, n)) as f:
                errs.append("rank %s: %s" % (n.split(".")[-1], f.read().strip()[-400:]))
        note = "; ".join(errs) or "attempt 1 ended with exit code(s) " + ", ".join(
            open(os.path.join(box, n)).read() for n in ls("a1.rc."))
        if rank == 0:
            print("bench.py: the job with the native communicator failed (%s): once more with --comm torch" % note,
                  file=sys.stderr, flush=True)
        rc, out_lines = attempt(2, ["--comm", "torch"])
    if rank == 0:
        for l in out_lines:
            if l.startswith("{") and note is not None:
                try:
                    d = json.loads(l)
                    d["native_error"] = note
                    l = json.dumps(d) + "\n"
                except ValueError:
                    pass
            sys.stdout.write(l)
        sys.stdout.flush()
    put("done.%d" % rank)
    if rank == 0:                                           # the last one out removes the scratch directory
        t1 = time.perf_counter()
        while len(ls("done.")) < world and time.perf_counter() - t1 < 30:
            time.sleep(0.05)
        import shutil
        shutil.rmtree(box, ignore_errors=True)
    return rc


def _s64(x):
    """two's-complement int64 value of x mod 2^64"""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >> 63 else x


def hashed_uniform(torch, index, salt):
    """deterministic doubles in [-1, 1) from GLOBAL DoF indices (splitmix64 finaliser): every
    partitioning of the mesh sees the same global source vector without materialising it
    (SURVEY 8d asks for one std::mt19937_64 stream in global order: same purpose)"""
    def lsr(z, n):                                   # logical shift right on int64
        return (z >> n) & ((1 << (64 - n)) - 1)
    z = index + _s64(salt * 0x9E3779B97F4A7C15 + SEED)
    z = (z ^ lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _s64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    return lsr(z, 11).to(torch.float64) * (2.0 / (1 << 53)) - 1.0


def global_dof_index(torch, part, degree, ncomp, device):
    """global lexicographic index (node * ncomp + comp) of the local DoFs of a brick"""
    nn_g = [degree * g * c + 1 for g, c in zip(part.grid, part.cells)]
    ax = [torch.arange(degree * part.cells[d] + 1, device=device, dtype=torch.int64)
          + degree * part.coords[d] * part.cells[d] for d in range(3)]
    node = (ax[2][:, None, None] * nn_g[1] + ax[1][None, :, None]) * nn_g[0] + ax[0][None, None, :]
    idx = node.reshape(-1, 1) * ncomp + torch.arange(ncomp, device=device, dtype=torch.int64)[None, :]
    return idx.reshape(-1)


def beltrami_nodal(torch, lower, h, ncell, degree, t, device, pressure=False):
    """nodal interpolant of the Beltrami field (tests/beltrami.cc:82-172) on the device; FE_Q
    support points = Gauss-Lobatto points (navier_stokes.cc:95-106)"""
    from adaflo_amd.navier_stokes import gauss_lobatto_points
    a = 0.25 * np.pi
    d = 2.0 * a
    gl = torch.tensor(gauss_lobatto_points(degree + 1), device=device, dtype=torch.float64)
    ax = []
    for dim in range(3):
        c = torch.arange(ncell[dim], device=device, dtype=torch.float64)
        x = (c[:, None] + gl[None, :degree]).reshape(-1)
        x = torch.cat([x, torch.tensor([float(ncell[dim])], device=device, dtype=torch.float64)])
        ax.append(lower[dim] + h[dim] * x)
    z, y, x = torch.meshgrid(ax[2], ax[1], ax[0], indexing="ij")
    if pressure:
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


def cpu_baseline(sample_cells, k=2, budget_s=12.0):
    """time the CPU restatements of the reference's path on a bounded sample of the SAME element: the cell-batched one
    (oracle/adaflo_oracle_batched.c: W cells per SIMD register, state in the batched layout, compile-time loop bounds,
    with and without even-odd 1D kernels -- the data flow of deal.II's FEEvaluation) and, for the record, the scalar
    sum-factorised one of rounds 1-3 (oracle/adaflo_oracle_fast.c).  `value` is the best of them."""
    from oracle import oracle as orc
    orc.build()
    orc.fast_set_threads(orc.usable_cores())          # the container's CPU quota, not the visible core count
    n = sample_cells
    mesh = orc.Mesh.make([n] * 3, [-1.0] * 3, [1.0] * 3)
    prm = orc.NSParams.make(weight=1.5 / 0.05, weight_old=-2 / 0.05, weight_old_old=0.5 / 0.05)
    rng = np.random.default_rng(SEED)
    nu, npr = mesh.n_nodes(k) * 3, mesh.n_nodes(k - 1)
    su, sp = rng.uniform(-1, 1, nu), rng.uniform(-1, 1, npr)
    lin = rng.uniform(-1, 1, mesh.n_cells * (k + 1) ** 3 * 12)
    con_u = orc.boundary_mask(mesh, k, 3)
    w = orc.ns_pressure_mass_weight(mesh, k)
    modes = np.ones(npr)
    out = (np.empty(nu), np.empty(npr))
    batched = orc.BatchedNSVmult(mesh, k, con_u, None, lin)

    def rate(f, budget):
        f()
        reps, t0 = 0, time.perf_counter()
        while True:
            f()
            reps += 1
            el = time.perf_counter() - t0
            if el > budget or reps >= 5000:
                return (nu + npr) * reps / el / 1e6, reps, el
    variants, total_reps, total_s = {}, 0, 0.0
    for name, share, f in (
            ("batched", 0.4, lambda: batched.vmult(prm, su, sp, weights=w, modes=modes, even_odd=False, out=out)),
            ("batched_even_odd", 0.4, lambda: batched.vmult(prm, su, sp, weights=w, modes=modes, even_odd=True, out=out)),
            ("scalar_sum_factorised", 0.2, lambda: orc.fast_ns_vmult(mesh, k, prm, su, sp, con_u, None, lin=lin, weights=w,
                                                                     modes=modes, out=out))):
        v, reps, el = rate(f, share * budget_s)
        variants[name] = round(v, 2)
        total_reps, total_s = total_reps + reps, total_s + el
    best = max(variants, key=variants.get)
    return {"value": variants[best], "unit": "MDoF/s", "cores": orc.fast_n_threads(), "kind": "port",
            "variant": best, "isa": batched.isa, "simd_width": batched.width, "variants": variants,
            "sample": "%d^3-cell Q%d/Q%d brick, %d vmults of the OpenMP restatements of the adaflo CPU path (deal.II "
                      "unavailable): %d cells per %s register, batched state layout, with / without even-odd 1D kernels; "
                      "the scalar sum-factorised variant of rounds 1-3 beside them; %.1f s"
                      % (n, k, k - 1, total_reps, batched.width, batched.isa, total_s)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d inside a torch.distributed job of %d ranks" % (args.gpus, world))
    if world > 1 and not args.no_supervisor and os.environ.get("ADAFLO_BENCH_WORKER") != "1":
        sys.exit(supervise(args))
    try:
        measure(args, world, rank, local_rank)
    except BaseException as e:                              # noqa: BLE001 -- the supervisor reports why attempt 1 failed
        box = os.environ.get("ADAFLO_BENCH_BOX")
        if box and not (isinstance(e, SystemExit) and e.code in (0, None)):
            try:
                with open(os.path.join(box, "a%s.err.%d" % (os.environ.get("ADAFLO_BENCH_ATTEMPT", "1"), rank)), "w") as f:
                    f.write("%s: %s" % (type(e).__name__, e))
            except OSError:
                pass
        raise


def measure(args, world, rank, local_rank):
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the HIP engine)")
    # functional dry run of the N > 1 path on a single GPU (not a measurement): gloo messages
    # staged through host memory, all ranks on device 0 (set by self_launch when GPUs are missing)
    backend = os.environ.get("ADAFLO_BENCH_BACKEND", "nccl")
    dry_run = os.environ.get("ADAFLO_BENCH_SINGLE_DEVICE") == "1"
    if dry_run:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        kw = {"device_id": device} if backend == "nccl" else {}
        attempt = int(os.environ.get("ADAFLO_BENCH_ATTEMPT", "1"))
        if attempt > 1 and os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
            # the repetition of a failed job (see supervise): the launcher's store still holds the rendezvous keys of the
            # first attempt (addresses of processes that are gone) -- the same store under a prefix of its own
            import datetime
            base = dist.TCPStore(os.environ["MASTER_ADDR"], int(os.environ["MASTER_PORT"]), world, False,
                                 timeout=datetime.timedelta(seconds=300))
            dist.init_process_group(backend, store=dist.PrefixStore("adaflo_bench_attempt%d" % attempt, base), rank=rank,
                                    world_size=world, **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    import adaflo_amd
    from adaflo_amd import build as _build
    from adaflo_amd import parallel
    if not os.path.exists(_build.LIB):
        _build.build()

    grid = parallel.brick_grid(world)
    cavity = args.config == "cavity"
    if cavity:
        k = args.degree or 4
        n_global = args.cells or 64
        if any(n_global % g for g in grid):
            raise SystemExit("cavity: %d cells are not divisible by the process grid %s" % (n_global, grid))
        cells = [n_global // g for g in grid]
        lower, upper = [0.0, 0.0, 0.0], [1.0, 1.0, 3.0]          # drivencavity.cc:357-370
        fp = adaflo_amd.FlowParameters(velocity_degree=k, physical_type="incompressible stationary",
                                       viscosity=0.01, time_step_size_start=0.05, end_time=1.0,
                                       linearization=args.linearization)
        scaling = "strong"
    else:
        k = args.degree or 2
        nc = args.cells or 128
        cells = [nc] * 3
        lower, upper = [-1.0] * 3, ([-1.0 + 2.0 * g for g in grid] if world > 1 else [1.0] * 3)
        fp = adaflo_amd.FlowParameters(velocity_degree=k, time_step_size_start=0.05, end_time=1.0,
                                       linearization=args.linearization)
        scaling = "weak"
    ts = adaflo_amd.TimeStepping(fp)
    for _ in range(3):
        ts.next()                                   # steady BDF-2 weights: gamma = 1.5/dt
    part = parallel.BrickPartition(grid, rank, cells, lower=lower, upper=upper)
    stream = torch.cuda.current_stream(device).cuda_stream   # 0 = the legacy default stream
    op = parallel.DistributedNavierStokesMatrix(fp, part, device=local_rank, stream=stream,
                                                group=dist.group.WORLD if world > 1 else None,
                                                native_comm=args.comm == "native",
                                                through_comm=args.through_comm and world == 1)
    if world > 1 and args.comm == "native" and os.environ.get("ADAFLO_BENCH_INJECT_NATIVE_FAILURE") == "1" and rank == world - 1:
        raise RuntimeError("injected failure of the native communicator (tests/test_bench_contract.py)")
    op.initialize(ts, True)
    op.set_kernel_variant(args.variant)
    op.overlap = not args.no_overlap
    if args.chunk:
        op.local.set_q2_chunk(args.chunk)
    if args.state_pad >= 0:
        op.local.set_q2_state_pad(args.state_pad)

    mesh = op.local.mesh
    n_u, n_p = op.local.n_dofs_u(), op.local.n_dofs_p()
    # linearisation point = nodal interpolant of the Beltrami field at t = 0 (on the cavity box as
    # well), pushed through the residual kernel, which is the only producer of the
    # q-point state in the reference
    u_lin = beltrami_nodal(torch, mesh.lower, mesh.h, mesh.ncell, k, 0.0, device)
    p_lin = torch.zeros(n_p, device=device, dtype=torch.float64)
    zeros_u = torch.zeros(n_u, device=device, dtype=torch.float64)
    V = adaflo_amd.DeviceVector.from_torch
    ctx = op.local._ctx
    rhs = adaflo_amd.BlockVector([V(ctx, torch.zeros_like(u_lin)), V(ctx, torch.zeros_like(p_lin))])
    op.local.residual(rhs, adaflo_amd.BlockVector([V(ctx, u_lin), V(ctx, p_lin)]), None,
                      adaflo_amd.BlockVector([V(ctx, u_lin)]), adaflo_amd.BlockVector([V(ctx, zeros_u)]))
    del rhs, u_lin, zeros_u
    # deterministic pseudo-random source in [-1,1): a hash of the GLOBAL DoF index, so any
    # partitioning sees the same global vector (replicas of interface DoFs agree by construction)
    src_u = hashed_uniform(torch, global_dof_index(torch, part, k, 3, device), 0)
    src_p = hashed_uniform(torch, global_dof_index(torch, part, k - 1, 1, device), 1)
    dst_u, dst_p = torch.empty_like(src_u), torch.empty_like(src_p)
    src = adaflo_amd.BlockVector([V(ctx, src_u), V(ctx, src_p)])
    dst = adaflo_amd.BlockVector([V(ctx, dst_u), V(ctx, dst_p)])

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    # set-up, untimed: a fixed quarter of a second of the same operator application brings clocks, TLBs and caches
    # of the device to their steady state (a step takes 1.5 ms: five warm-up steps alone leave the first timed
    # steps 3-5 % slow); then the W warm-up steps the caller asked for
    # (the number of applications is agreed between the ranks: estimated from four timed ones, maximum over ranks)
    barrier()
    t_pre = time.perf_counter()
    for _ in range(4):
        op.vmult(dst, src, src_consistent=args.src_consistent)
    barrier()
    est = (time.perf_counter() - t_pre) / 4
    if world > 1:
        t = torch.tensor([est], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        est = float(t.item())
    for _ in range(int(min(160, max(8, 0.25 / max(est, 1e-6))))):
        op.vmult(dst, src, src_consistent=args.src_consistent)
    barrier()
    for _ in range(args.warmup):
        op.vmult(dst, src, src_consistent=args.src_consistent)
    op.local.get_kernel_statistics()
    op.local.get_matvec_statistics()
    # timed region: EXACTLY K steps between two barriers, nothing else on the stream (a torch timing event is a default
    # HIP event: its record carries a system-scope fence -- cache write-back + invalidate -- that the next step pays
    # for; the per-step times are therefore taken in a second, untimed pass below)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        op.vmult(dst, src, src_consistent=args.src_consistent)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ksec, kcount = op.local.get_kernel_statistics()
    if op.comm is not None:
        # get_matvec_statistics over the ranks (navier_stokes_matrix.cc:1194-1206): device time of the whole distributed
        # vmult per rank, exchanges included -- min / max over the ranks show the load imbalance of boundary bricks
        (rank_min, rank_max, rank_avg, rank_imin, rank_imax), mcount = op.get_matvec_statistics()
        msec = rank_avg
        op.local.get_matvec_statistics()
    else:
        msec, mcount = op.local.get_matvec_statistics()
        rank_min = rank_max = msec
        rank_imin = rank_imax = 0

    # per-step device times (min / median; not part of the timed region): events on the stream the engine launches on
    # (= torch's current stream)
    events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    for i in range(args.steps):
        events[i].record()
        op.vmult(dst, src, src_consistent=args.src_consistent)
    events[args.steps].record()
    barrier()
    step_ms = np.array([events[i].elapsed_time(events[i + 1]) for i in range(args.steps)])
    # device times of the pieces of the distributed operator on rank 0 (a third, untimed pass: ten more event records per
    # application): what overlaps with what, so that one driver run of an N > 1 job diagnoses itself
    phase_ms = None
    if op.comm is not None:
        op.comm.set_phase_timing(True)
        barrier()
        for i in range(args.steps):
            op.vmult(dst, src, src_consistent=args.src_consistent)
        barrier()
        n_ph, sec = op.comm.phase_statistics()
        op.comm.set_phase_timing(False)
        phase_ms = {name: round(1e3 * t / max(n_ph, 1), 4) for name, t in sec.items()}
    if args.print_steps and rank == 0:
        print("step ms:", " ".join("%.3f" % t for t in step_ms), file=sys.stderr)
    op.local.get_kernel_statistics()
    op.local.get_matvec_statistics()

    n_dofs_global = part.n_global_dofs(k)
    n_cells_local = op.local.n_cells()
    value = n_dofs_global * args.steps / elapsed / 1e6
    kernel_avg = ksec / max(kcount, 1)
    b_alg_launch = b_alg_per_cell(k) * n_cells_local
    achieved = b_alg_launch / kernel_avg / 1e9 if kernel_avg > 0 else None
    ms_per_step = 1e3 * elapsed / args.steps
    frac_vmult = b_alg_launch / (1e-3 * ms_per_step) / 1e9 / HBM_PEAK_GBS
    if not torch.isfinite(dst_u).all() and not os.environ.get("ADAFLO_BENCH_NOCHECK"):  # (diagnostic kernel builds)
        raise SystemExit("non-finite result")

    # the kernel a variant selects (adaflo_ns_set_kernel_variant): 0 generic; 1 / 4 Q2/Q1 sweep kernel (1 recomputes the
    # Newton state, 4 streams it) and the x-marching kernel for Q3..Q5; 2 / 3 the superseded Q3..Q5 kernels (development builds)
    if args.variant == 0 or k > 5:
        kernel_name = "ns_cell_kernel"
    elif k == 2:
        kernel_name = "ns_q2_kernel"
    elif args.variant == 2:
        kernel_name = "ns_ho_kernel"
    elif args.variant == 3 and k == 4:
        kernel_name = "ns_hop_kernel"
    else:
        kernel_name = "ns_hox_kernel"
    traffic = None
    try:  # PMC-measured HBM bytes per launch of the dominant kernel (profiles/, collected with rocprofv3)
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        key = "%s k=%d variant=%d" % ("x".join(str(c) for c in cells), k, args.variant)
        traffic = pmc[key][kernel_name]["hbm_bytes"]
    except (OSError, KeyError, ValueError, TypeError):
        pass
    mesh_str = "x".join(str(g * c) for g, c in zip(grid, cells))
    if cavity:
        workload = ("3D driven-cavity box [0,1]x[0,1]x[0,3] Q%d/Q%d NavierStokes vmult, incompressible stationary, "
                    "mu = 0.01, Newton linearisation about the Beltrami formula, uniform hex mesh %s cells "
                    "(%s per GPU), Dirichlet on all faces" % (k, k - 1, mesh_str, "x".join(map(str, cells))))
    else:
        workload = ("3D Beltrami Q%d/Q%d NavierStokes vmult, Newton linearisation, uniform hex mesh %s cells "
                    "(%d^3 per GPU), Dirichlet on all faces, pressure mean projection" % (k, k - 1, mesh_str, cells[0]))
    # Q2/Q1 since round 5: the Newton state (u_lin, grad u_lin) is not streamed but recomputed from the nodal linearisation
    # point the residual left (kernel variant 1; variant 4 streams).  `achieved` / `frac` stay what the contract defines --
    # SURVEY 8(d)'s algorithmic bytes per cell over the kernel time --; the bytes this kernel has to move are listed next
    # to them (vectors + 24 k^3 B per cell of nodal linearisation point), and `traffic` is what the PMC counters saw
    # (round 6: the Picard-type state as well -- Picard, semi-implicit and projection schemes; the explicit scheme has no state)
    newton = args.linearization == "coupled implicit Newton"
    recomputed = k == 2 and args.variant in (1, 2, 3) and args.linearization != "coupled velocity explicit"
    # FP64 work of the recompute mode, counted in the ISA of ns_q2_kernel<0,true,true,false,false,false,true,false>:
    # per cell layer and lane 770 v_fmac_f64 + 328 v_fma_f64 (2 flop) + 458 v_mul_f64 + 342 v_add_f64, four lanes per cell
    flop_per_cell = 4 * (2 * (770 + 328) + 458 + 342) if (recomputed and newton) else None
    b_moved_launch = (16 * (3 * k ** 3 + (k - 1) ** 3) + 24 * k ** 3) * n_cells_local if recomputed else b_alg_launch
    out = {
        "metric": "MDoF/s for NavierStokesMatrix::vmult (3D Q%d/Q%d)" % (k, k - 1),
        "value": round(value, 1), "unit": "MDoF/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
        "ms_per_step_min": round(float(step_ms.min()), 4), "ms_per_step_median": round(float(np.median(step_ms)), 4),
        "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": workload, "dofs": n_dofs_global, "cells_per_gpu": n_cells_local,
                   "partition": "x".join(str(g) for g in grid), "kernel_variant": args.variant,
                   "overlap": bool(op.overlap) if world > 1 else None,
                   "comm": args.comm if (world > 1 or args.through_comm) else None,
                   "through_comm": bool(args.through_comm) if world == 1 else None,
                   # N > 1: does a timed vmult include the owner->ghost update of src (the reference's
                   # update_ghost_values in cell_loop)?  False only with --src-consistent
                   "src_ghost_update": (not args.src_consistent) if world > 1 else None},
        "roofline": {"bound": "fp64" if recomputed else "hbm", "achieved": round(achieved, 1) if achieved else None,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None,
                     "frac_vmult": round(frac_vmult, 4),
                     "frac_of_copy_ceiling": round(achieved / HBM_COPY_GBS, 4) if achieved else None,
                     "traffic": traffic,
                     "traffic_source": "profiles/pmc_traffic.json (builder's rocprofv3 --pmc pass of this mesh and variant, "
                                       "not this run)" if traffic is not None else None,
                     "kernel": kernel_name, "kernel_ms": round(1e3 * kernel_avg, 4),
                     "alg_bytes_per_launch": b_alg_launch, "alg_bytes_per_dof": round(
                         b_alg_per_cell(k) / (3 * k ** 3 + (k - 1) ** 3), 1),
                     "state": "recomputed from the nodal linearisation point" if recomputed else "streamed",
                     "bytes_to_move_per_launch": b_moved_launch,
                     "frac_bytes_to_move": round(b_moved_launch / kernel_avg / 1e9 / HBM_PEAK_GBS, 4) if kernel_avg > 0 else None,
                     "vmult_ms_device": round(1e3 * msec / max(mcount, 1), 4)},
    }
    if recomputed and newton and kernel_avg > 0:
        # the kernel that recomputes the state is bound by FP64 issue, not by HBM: `achieved` / `frac` above stay
        # the contract's algorithmic bytes over the kernel time; this object prices the same launch against the FP64 vector peak
        tf = flop_per_cell * n_cells_local / kernel_avg / 1e12
        out["roofline"]["fp64"] = {"flop_per_launch": flop_per_cell * n_cells_local, "flop_per_cell": flop_per_cell,
                                   "achieved": round(tf, 2), "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(tf / FP64_PEAK_TFLOPS, 4),
                                   "source": "FP64 instructions of the kernel's ISA per cell layer x 4 lanes per cell"}
    if world > 1:
        out["ms_per_step_min_rank"] = round(1e3 * rank_min / max(mcount, 1), 4)
        out["ms_per_step_max_rank"] = round(1e3 * rank_max / max(mcount, 1), 4)
        out["min_rank"], out["max_rank"] = rank_imin, rank_imax
    if phase_ms is not None:
        # (rank 0; src_exchange, interface_cells and dst_exchange run on the auxiliary stream beside interior_cells)
        out["phase_ms_rank0"] = phase_ms
    if dry_run:
        out["dry_run"] = True       # ranks share one GPU, gloo messages: functional check only
    if not args.no_cpu_baseline:
        # same element as the metric; the sample holds about as many DoFs as the 48^3 Q2/Q1 default.  Timed on rank 0,
        # after the timed region; with N > 1 the other ranks sleep on the rendezvous store meanwhile (a blocking
        # socket wait: they do not spin on cores the baseline is using)
        store = dist.distributed_c10d._get_default_store() if world > 1 else None
        if rank == 0:
            out["cpu_baseline"] = cpu_baseline(max(4, args.cpu_sample_cells * 2 // k), k)
            if store is not None:
                store.set("adaflo_bench_cpu_baseline_done", "1")
        elif store is not None:
            import datetime
            store.wait(["adaflo_bench_cpu_baseline_done"], datetime.timedelta(seconds=600))
    elif rank == 0:
        out["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
